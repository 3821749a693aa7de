"""GPU: rt_create_multi -- one context over several GPUs behind the C ABI.  The one-GPU box
drives it (a) with n = 1, the real RCCL path (ncclCommInitAll, grouped ncclSend/ncclRecv to the root), and
(b) with the same device named three times, which plays three ranks on the one GPU (bands move with
device-to-device copies, everything else is the multi-GPU code path).  Either way rt_render must return
the single-GPU frame, and the reference-rule counters summed over the devices must equal the oracle's."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_frame_matches, assert_stats_match, oracle_render
from raytracing_amd import RayTracer
from raytracing_amd import _native as N

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices,rccl", [([0], True), ([0, 0, 0], False)], ids=["n1-rccl", "n3-on-one-gpu"])
@pytest.mark.parametrize("name,params,depth,size", [("TestScene1", None, 10, (333, 190)), ("Soup", [8000, 80, 8], 4, (160, 90))])
def test_multi_context_renders_the_single_gpu_frame(devices, rccl, name, params, depth, size, rt, scene_of):
    sc = scene_of(name, params)
    cam = sc.camera_for(*size)
    rt.SendPrimitivesToShader(sc)
    rt.PrepareRenderOpenGL(cam)
    single = rt.RenderGL(depth)
    with RayTracer(devices) as m:
        assert m.multi_info() == {"n_devices": len(devices), "uses_rccl": rccl}
        m.SendPrimitivesToShader(sc)
        m.PrepareRenderOpenGL(cam)
        for _ in range(2):                      # twice: buffers are reused, streams stay ordered
            out = m.RenderGL(depth)
            for k in ("rgb", "rgba8", "primary_id", "final_id"):
                assert np.array_equal(out[k].view(np.uint8), single[k].view(np.uint8)), f"{name} {k}"
        slow = m.RenderGL(depth, stats=True)    # reference-order kernel on every device, counters summed
        ref = oracle_render(sc, cam, depth, threads=16)
        assert_frame_matches(slow, ref, f"{name} multi reference-order")
        assert_stats_match(slow["stats"], ref["counters"], f"{name} multi")
        if name == "Soup":                      # the staged pipeline on every device
            m.set_option(m.OPTION_TRAVERSAL, m.TRAVERSAL_WAVEFRONT)
            out = m.RenderGL(depth)
            assert np.array_equal(out["rgb"].view(np.uint8), single["rgb"].view(np.uint8))
            assert np.array_equal(out["final_id"], single["final_id"])


def test_multi_context_argument_errors():
    lib = N.device_lib()
    ctx = C.c_void_p()
    assert lib.rt_create_multi(None, 1, C.byref(ctx)) == -1 and not ctx
    ids = (C.c_int * 2)(0, 0)
    assert lib.rt_create_multi(ids, 0, C.byref(ctx)) == -1 and not ctx
    bad = (C.c_int * 2)(0, 9999)
    assert lib.rt_create_multi(bad, 2, C.byref(ctx)) == -4 and not ctx     # RT_ERR_DEVICE: no such GPU
    n, r = C.c_int(), C.c_int()
    with RayTracer(0) as one:
        assert one.multi_info() == {"n_devices": 1, "uses_rccl": False}


def test_rccl_that_cannot_be_loaded_is_an_error_code_not_a_crash():
    """the loader formed its message from two dlerror() calls (the second returns NULL).  A fresh
    process whose RCCL is forced to a missing file (RT_RCCL_LIBRARY) must get RT_ERR_DEVICE from rt_create_multi --
    for every distinct-device list, n = 1 included -- and go on working."""
    import os
    import subprocess
    import sys
    code = ("import ctypes as C, torch\n"
            "from raytracing_amd import _native as N, RayTracer\n"
            "lib = N.device_lib(); ctx = C.c_void_p(); ids = (C.c_int * 1)(0)\n"
            "rc = lib.rt_create_multi(ids, 1, C.byref(ctx))\n"
            "assert rc == -4 and not ctx, rc\n"
            "with RayTracer(0) as one: assert one.multi_info()['n_devices'] == 1\n"
            "print('survived')\n")
    env = dict(os.environ, RT_RCCL_LIBRARY="/nonexistent/librccl-missing.so")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "survived" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    assert "cannot load librccl.so" in p.stderr


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]], ids=["n1-rccl", "n3-on-one-gpu"])
def test_multi_context_device_resident_frame(devices, rt, scene_of):
    """rt_render_device on a multi-GPU context (whole frame): every device renders its bands, ALL requested planes
    travel in one exchange, and the assembled planes land in the caller's buffers on the root device, asynchronously
    on the caller's stream -- what `bench.py --capi-multi` times.  Equal to the single-GPU context bit for bit,
    twice in a row and after a change of size."""
    import torch
    sc = scene_of("TeapotTable")
    rt.SendPrimitivesToShader(sc)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with RayTracer(devices) as m:
        m.SendPrimitivesToShader(sc)
        for (w, h) in [(333, 190), (160, 90), (333, 190)]:
            cam = sc.camera_for(w, h)
            rt.PrepareRenderOpenGL(cam)
            single = rt.RenderGL(8)
            out = {"rgb": torch.full((h, w, 3), -7.0, dtype=torch.float32, device=dev),
                   "rgba8": torch.zeros((h, w, 4), dtype=torch.uint8, device=dev),
                   "primary_id": torch.full((h, w), -9, dtype=torch.int32, device=dev),
                   "final_id": torch.full((h, w), -9, dtype=torch.int32, device=dev)}
            torch.cuda.synchronize()
            with torch.cuda.stream(stream):
                m.render_device(cam, 8, stream=stream.cuda_stream, **{k: v.data_ptr() for k, v in out.items()})
                got = {k: v.clone() for k, v in out.items()}   # queued on the same stream: must see the finished planes
            stream.synchronize()
            for k in out:
                assert np.array_equal(got[k].cpu().numpy().view(np.uint8), single[k].view(np.uint8)), (devices, w, h, k)
            # one plane only
            only = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
            m.render_device(cam, 8, rgba8=only.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
            assert np.array_equal(only.cpu().numpy(), single["rgba8"])


def test_multi_context_with_trees_built_on_its_first_device(rt, scene_of):
    """RT_OPTION_GPU_BUILD_MIN: a multi-GPU context builds both derived trees on its first device (csrc/rt_gpu_bvh.h),
    brings them back to the host and replicates them -- same arrays as the host builder's, so the same frame, the
    same scene info and the same counters as the single-GPU context, here forced on for a small scene and a soup."""
    for name, params, depth in (("TeapotTable", None, 8), ("Soup", [30000, 300, 8], 4)):
        sc = scene_of(name, params)
        cam = sc.camera_for(192, 108)
        rt.set_option(rt.OPTION_GPU_BUILD_MIN, -1)      # the single-GPU context: host threads
        try:
            rt.SendPrimitivesToShader(sc)
            rt.PrepareRenderOpenGL(cam)
            one, info_one = rt.RenderGL(depth), rt.scene_info()
        finally:
            rt.set_option(rt.OPTION_GPU_BUILD_MIN, 20000)
        with RayTracer([0, 0, 0]) as m:
            m.set_option(m.OPTION_GPU_BUILD_MIN, 0)
            m.SendPrimitivesToShader(sc)
            m.PrepareRenderOpenGL(cam)
            out, info = m.RenderGL(depth), m.scene_info()
        for k in ("bvh_nodes", "bvh_levels", "tight_nodes", "tight_levels", "flags", "unbounded_triangles"):
            assert info[k] == info_one[k], (name, k)
        for plane in ("primary_id", "final_id"):
            assert np.array_equal(out[plane], one[plane]), (name, plane)
        assert np.array_equal(out["rgb"].view(np.uint32), one["rgb"].view(np.uint32)), name
