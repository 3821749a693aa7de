"""GPU: `python bench.py --gpus N` starts its own N ranks and runs the N > 1 step end to end.  The one-GPU box
plays every rank on the same card with the gloo backend (BENCH_BACKEND=gloo, exchange staged through the host);
--verify makes every rank compare the frames it assembled with whole-frame renders."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=240, **more_env):
    env = dict(os.environ, BENCH_BACKEND="gloo", **more_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def test_plain_invocation_starts_two_ranks_weak():
    out = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "teapot_1080p_d1", "--verify", "--no-cpu")
    assert out["n_gpus"] == 2 and out["n_ranks"] == 2 and out["backend"] == "gloo" and out["n_ranks_rccl"] == 0
    assert out["scaling"] == "weak" and 0 < out["value_traced"] <= out["value"]
    assert out["config"]["frames_per_step"] == 2 and out["value"] > 0
    assert out["config"]["rays_traced_per_step"] <= out["config"]["rays_per_step"]
    assert 0 < out["roofline"]["frac"] <= 1


def test_plain_invocation_strong_scaling_one_frame():
    out = _bench("--gpus", "3", "--strong", "--steps", "2", "--warmup", "1", "--workload", "testscene1_720p_d10",
                 "--verify", "--no-cpu")
    assert out["n_gpus"] == 3 and out["n_ranks"] == 3 and out["scaling"] == "strong"
    assert out["config"]["frames_per_step"] == 1
    assert 0 < out["roofline"]["frac"] <= 1


def test_config5_sweep_as_one_command_two_ranks():
    """BASELINE.json configs[4]: the 256-frame TeapotTable sweep at depth 8 -- 8 x N frames per step, frame f assembled
    on rank f % N by one all-to-all; --verify compares frames 0 / 64 / 128 / 192 (and the last step's) with whole-frame
    renders.  Two steps here; the default is one whole sweep."""
    out = _bench("--gpus", "2", "--workload", "teapottable_sweep256_d8", "--steps", "2", "--warmup", "1", "--verify", "--no-cpu",
                 timeout=400)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["frames_per_step"] == 16
    assert out["value"] > 0 and out["config"]["max_bounces"] == 8


@pytest.mark.parametrize("devices,rccl", [("0", 1), ("0,0,0", 0)], ids=["n1-rccl", "n3-on-one-gpu"])
def test_capi_multi_bench(devices, rccl):
    """`bench.py --capi-multi`: the in-library multi-GPU render (rt_create_multi) timed through rt_render_device;
    N = 1 goes over RCCL, three devices played on the one GPU go through the copy path."""
    n = len(devices.split(","))
    out = _bench("--capi-multi", "--gpus", str(n), "--workload", "teapot_1080p_d1", "--steps", "5", "--warmup", "2", "--verify",
                 BENCH_MULTI_DEVICES=devices)
    assert out["n_gpus"] == n and out["n_ranks_rccl"] == rccl * n and out["scaling"] == "strong" and out["verified"]
    assert out["value"] > 0 and 0 < out["roofline"]["frac"] <= 1
