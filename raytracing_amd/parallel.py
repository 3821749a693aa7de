"""Multi-GPU frame sharding: one process per GPU (torch.distributed; backend "nccl" is RCCL).

The reference renders on one GPU (RayTracer.cs:135) and has nothing comparable.  Pixels are
independent and the scene is read-only, so the scene is replicated and the image is cut into
bands of `band_rows` rows dealt round-robin to the ranks (sky, floor and mesh rows are spread
evenly); every rank renders its bands into a compact buffer, one all-gather of equal-sized
contributions moves them over xGMI, and an un-permute pass restores scanline order.
"""
import numpy as np
import torch
import torch.distributed as dist


def local_rows(height, band_rows, rank, world):
    """Rows `rank` owns (same rule as rt_local_rows in include/rt_mi355x.h)."""
    n_bands = -(-height // band_rows)
    return sum(min(band_rows, height - b * band_rows) for b in range(rank, n_bands, world))


def padded_rows(height, band_rows, world):
    n_bands = -(-height // band_rows)
    return -(-n_bands // world) * band_rows


def global_rows_of(height, band_rows, rank, world):
    """Global row index of every local row of `rank`, in local order."""
    rows = []
    for lr in range(local_rows(height, band_rows, rank, world)):
        rows.append(((lr // band_rows) * world + rank) * band_rows + lr % band_rows)
    return np.asarray(rows, dtype=np.int64)


def gather_index(height, band_rows, world):
    """index[y] = row of the rank-major gathered buffer ([world * padded_rows] rows) holding scanline y."""
    pr = padded_rows(height, band_rows, world)
    idx = np.empty(height, dtype=np.int64)
    for r in range(world):
        g = global_rows_of(height, band_rows, r, world)
        idx[g] = r * pr + np.arange(len(g))
    return idx


def assemble_rows(gathered, height, band_rows, world):
    """Device-agnostic un-permute (torch index_select); the HIP path uses rt_unpermute_bands."""
    pr = padded_rows(height, band_rows, world)
    flat = gathered.reshape(world * pr, *gathered.shape[2:])
    idx = torch.from_numpy(gather_index(height, band_rows, world)).to(flat.device)
    return flat.index_select(0, idx)


def all_gather_bands(local, world, group=None):
    """local: [padded_rows, W, C] on every rank -> [world, padded_rows, W, C] on every rank."""
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if world == 1:
        out[0].copy_(local)
    else:
        # output viewed as the rank-major concatenation along dim 0 (the form every backend accepts)
        dist.all_gather_into_tensor(out.view((world * local.shape[0],) + tuple(local.shape[1:])),
                                    local.contiguous(), group=group)
    return out


def assemble_frames(gathered, height, band_rows, world):
    """gathered [world, n_frames, padded_rows, W, C] -> [n_frames, H, W, C] (torch; any device)."""
    idx = torch.from_numpy(gather_index(height, band_rows, world)).to(gathered.device)
    pr = gathered.shape[2]
    per_frame = gathered.permute(1, 0, 2, 3, 4).reshape(gathered.shape[1], world * pr, *gathered.shape[3:])
    return per_frame.index_select(1, idx)


class ShardedFrames:
    """Per-rank state of a batch of band-sharded RGBA8 frames that travel in ONE collective per step.

    A 1080p band set is ~1 MB per rank: one collective per frame is latency bound on xGMI, so a step that
    renders n_frames frames renders every frame's local bands with one launch (up to 8 frames; more go out
    in batches over a few HIP streams) and exchanges them together.  `assemble` says where frames end up:

      "owner"  n_frames a multiple of world: frame f is assembled on rank f % world -- concurrent gathers, one
               root each, as ONE all-to-all.  Every rank sends and receives (world-1)/world of n_frames / world
               frames per step, whatever the world size (xGMI is point to point: each peer's share travels its own
               link).  Inside, the frames are kept in owner-major order so that the chunk for rank d is contiguous.
      "root"   every frame is assembled on rank 0 (one gather): the strong-scaling form, one frame per step.
      "all"    every rank gets every frame (one all-gather); traffic per rank grows with world."""

    def __init__(self, rt, width, height, rank, world, band_rows, n_frames, group=None, device=None, n_streams=4,
                 assemble="all"):
        self.rt, self.W, self.H, self.F = rt, int(width), int(height), int(n_frames)
        self.rank, self.world, self.band, self.group = rank, world, int(band_rows), group
        if assemble not in ("all", "owner", "root"):
            raise ValueError("assemble must be 'all', 'owner' or 'root'")
        if assemble == "owner" and self.F % world != 0:
            raise ValueError("assemble='owner' needs a multiple of `world` frames")
        self.assemble = assemble
        # slot j of the local buffer holds frame order[j]: owner-major for "owner" (rank d's frames d, d + world, ...)
        per = self.F // world if assemble == "owner" else self.F
        self.order = [i * world + d for d in range(world) for i in range(per)] if assemble == "owner" and world > 1 else list(range(self.F))
        dev = device if device is not None else torch.device("cuda", rt.device_id)
        self.pr = padded_rows(self.H, self.band, world)
        self.local = torch.zeros((self.F, self.pr, self.W, 4), dtype=torch.uint8, device=dev)
        n_mine = {"all": self.F, "owner": self.F // world, "root": self.F if rank == 0 else 0}[assemble]
        self.n_mine = n_mine
        # what arrives here: [source rank][frame of mine][padded rows][W] (owner: one frame, root: rank 0 only)
        self.gathered = torch.empty((world, max(n_mine, 1), self.pr, self.W, 4), dtype=torch.uint8, device=dev)
        self.frames = torch.empty((max(n_mine, 1), self.H, self.W, 4), dtype=torch.uint8, device=dev)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(min(n_streams, self.F))] if self.F > 8 else []
        self.pending = None
        self.ready = False
        # backends that cannot move device tensors in this collective (gloo rehearsals) stage through the host
        self.staged = world > 1 and dist.is_initialized() and dist.get_backend(group) != "nccl" and dev.type != "cpu"
        self.on_gpu = dev.type != "cpu"   # host tensors: only the tests' stand-in renderer (no GPU in the container)

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream if self.on_gpu else 0

    def owned_frames(self):
        """(frame index, assembled [H, W, 4] tensor) of the frames this rank holds after finish()."""
        if self.assemble == "owner" and self.world > 1:
            return [(i * self.world + self.rank, self.frames[i]) for i in range(self.n_mine)]
        return [(f, self.frames[f]) for f in range(self.n_mine)]

    def render(self, cameras, max_bounces):
        """Launch all frames' local bands; on return the current stream waits for them.  The frames go to
        the library in batches of 8 (one launch each for the megakernels, rt_render_frames_device), batches
        alternating over the streams so that the tail of one launch overlaps the head of the next."""
        out = self.local if self.world > 1 else self.frames  # a single rank renders straight into the frames
        stride = out.stride(0) // 4  # pixels between consecutive frames
        cameras = np.ascontiguousarray(np.asarray(cameras, dtype=np.float32)[:self.F][self.order])
        batches = [(f0, min(f0 + 8, self.F)) for f0 in range(0, self.F, 8)]
        if len(batches) == 1 or not self.streams:
            self.rt.render_frames_device(cameras[:self.F], max_bounces, stride, rgba8=out.data_ptr(), band_rows=self.band,
                                         rank=self.rank, world=self.world, stream=self._stream())
            return
        main = torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(main)
        for k, (f0, f1) in enumerate(batches):
            s = self.streams[k % len(self.streams)]
            self.rt.render_frames_device(cameras[f0:f1], max_bounces, stride, rgba8=out[f0].data_ptr(),
                                         band_rows=self.band, rank=self.rank, world=self.world, stream=s.cuda_stream)
        for s in self.streams:
            main.wait_stream(s)

    def _collective(self, recv, send, async_op):
        flat_send = send.view(-1)
        if self.assemble == "all":
            return dist.all_gather_into_tensor(recv.view(-1), flat_send, group=self.group, async_op=async_op)
        if self.assemble == "owner":   # chunk d of `send` (frame d's bands) goes to rank d
            return dist.all_to_all_single(recv.view(-1), flat_send, group=self.group, async_op=async_op)
        parts = [recv[r].view(-1) for r in range(self.world)] if self.rank == 0 else None
        return dist.gather(flat_send, parts, dst=0, group=self.group, async_op=async_op)

    def gather_async(self):
        if self.world == 1:
            return
        if self.staged:   # rehearsal path: device -> host, collective on host tensors, host -> device
            send = self.local.cpu()
            recv = torch.empty(self.gathered.shape, dtype=torch.uint8)
            self._collective(recv, send, False)
            if self.n_mine:
                self.gathered.copy_(recv)
            self.ready = True
            return
        self.pending = self._collective(self.gathered, self.local, True)

    def wait_gather(self):
        """Make the current stream wait for this buffer's exchange (if one is in flight)."""
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
            self.ready = True

    def finish(self):
        """Wait for the exchange of this buffer (if any) and restore scanline order where frames are assembled."""
        self.wait_gather()
        if self.ready:
            self.ready = False
            if self.n_mine:
                self.rt.unpermute_frames(self.gathered.data_ptr(), self.frames.data_ptr(), 4, self.W, self.H, self.band,
                                         self.world, self.n_mine, stream=self._stream())
        return self.frames


class ShardedFrame:
    """Per-rank state of a tile-sharded RGBA8 frame on the GPU."""

    def __init__(self, rt, width, height, rank, world, band_rows=16, group=None, device=None):
        self.rt, self.W, self.H = rt, int(width), int(height)
        self.rank, self.world, self.band, self.group = rank, world, int(band_rows), group
        dev = device if device is not None else torch.device("cuda", rt.device_id)
        self.pr = padded_rows(self.H, self.band, world)
        self.local = torch.zeros((self.pr, self.W, 4), dtype=torch.uint8, device=dev)
        self.gathered = torch.empty((world, self.pr, self.W, 4), dtype=torch.uint8, device=dev)
        self.frame = torch.empty((self.H, self.W, 4), dtype=torch.uint8, device=dev)

    def render(self, camera, max_bounces, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream()
        self.rt.render_device(camera, max_bounces, rgba8=self.local.data_ptr(), band_rows=self.band,
                              rank=self.rank, world=self.world, stream=s.cuda_stream)

    def gather(self):
        if self.world == 1:
            self.gathered[0].copy_(self.local)
        else:
            dist.all_gather_into_tensor(self.gathered.view(self.world * self.pr, self.W, 4), self.local,
                                        group=self.group)

    def assemble(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream()
        self.rt.unpermute_bands(self.gathered.data_ptr(), self.frame.data_ptr(), 4, self.W, self.H, self.band,
                                self.world, stream=s.cuda_stream)
        return self.frame
