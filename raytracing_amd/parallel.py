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
    """Per-rank state of a batch of band-sharded RGBA8 frames that travel in ONE all-gather.

    A 1080p band set is ~1 MB per rank: one collective per frame is latency bound on xGMI, so a step
    that renders n_frames frames renders every frame's local bands first (alternating over a few HIP
    streams, so the tail of one small launch overlaps the head of the next) and exchanges them together."""

    def __init__(self, rt, width, height, rank, world, band_rows, n_frames, group=None, device=None, n_streams=4):
        self.rt, self.W, self.H, self.F = rt, int(width), int(height), int(n_frames)
        self.rank, self.world, self.band, self.group = rank, world, int(band_rows), group
        dev = device if device is not None else torch.device("cuda", rt.device_id)
        self.pr = padded_rows(self.H, self.band, world)
        self.local = torch.zeros((self.F, self.pr, self.W, 4), dtype=torch.uint8, device=dev)
        self.gathered = torch.empty((world, self.F, self.pr, self.W, 4), dtype=torch.uint8, device=dev)
        self.frames = torch.empty((self.F, self.H, self.W, 4), dtype=torch.uint8, device=dev)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(min(n_streams, self.F))] if self.F > 1 else []
        self.pending = None

    def render(self, cameras, max_bounces):
        """Launch all frames' local bands; on return the current stream waits for them.  The frames go to
        the library in batches of 8 (one launch each for the megakernels, rt_render_frames_device), batches
        alternating over the streams so that the tail of one launch overlaps the head of the next."""
        main = torch.cuda.current_stream()
        out = self.local if self.world > 1 else self.frames  # a single rank renders straight into the frames
        stride = out.stride(0) // 4  # pixels between consecutive frames
        batches = [(f0, min(f0 + 8, self.F)) for f0 in range(0, self.F, 8)]
        if len(batches) == 1 or not self.streams:
            self.rt.render_frames_device(cameras[:self.F], max_bounces, stride, rgba8=out.data_ptr(), band_rows=self.band,
                                         rank=self.rank, world=self.world, stream=main.cuda_stream)
            return
        for s in self.streams:
            s.wait_stream(main)
        for k, (f0, f1) in enumerate(batches):
            s = self.streams[k % len(self.streams)]
            self.rt.render_frames_device(cameras[f0:f1], max_bounces, stride, rgba8=out[f0].data_ptr(),
                                         band_rows=self.band, rank=self.rank, world=self.world, stream=s.cuda_stream)
        for s in self.streams:
            main.wait_stream(s)

    def gather_async(self):
        if self.world > 1:
            self.pending = dist.all_gather_into_tensor(
                self.gathered.view(self.world * self.F * self.pr, self.W, 4),
                self.local.view(self.F * self.pr, self.W, 4), group=self.group, async_op=True)

    def wait_gather(self):
        """Make the current stream wait for this buffer's exchange (if one is in flight)."""
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
            self.ready = True

    def finish(self):
        """Wait for the exchange of this buffer (if any) and restore scanline order on every rank."""
        self.wait_gather()
        if getattr(self, "ready", False):
            self.ready = False
            self.rt.unpermute_frames(self.gathered.data_ptr(), self.frames.data_ptr(), 4, self.W, self.H, self.band,
                                     self.world, self.F, stream=torch.cuda.current_stream().cuda_stream)
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
