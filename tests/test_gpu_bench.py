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


def _bench(*flags, timeout=240):
    env = dict(os.environ, BENCH_BACKEND="gloo")
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
    assert out["n_gpus"] == 2 and out["n_ranks_rccl"] == 2 and out["scaling"] == "weak"
    assert out["config"]["frames_per_step"] == 2 and out["value"] > 0
    assert out["config"]["rays_traced_per_step"] <= out["config"]["rays_per_step"]
    assert 0 < out["roofline"]["frac"] <= 1


def test_plain_invocation_strong_scaling_one_frame():
    out = _bench("--gpus", "3", "--strong", "--steps", "2", "--warmup", "1", "--workload", "testscene1_720p_d10",
                 "--verify", "--no-cpu")
    assert out["n_gpus"] == 3 and out["n_ranks_rccl"] == 3 and out["scaling"] == "strong"
    assert out["config"]["frames_per_step"] == 1
    assert 0 < out["roofline"]["frac"] <= 1
