"""Shared helpers of the parity tests."""
import zlib

import numpy as np

import oracle_lib as O

STAT_TO_COUNTER = {"primary_rays": "primary", "secondary_rays": "secondary", "shadow_rays": "shadow",
                   "node_tests": "node_tests", "refs_read": "refs_read", "sphere_tests": "sphere_tests",
                   "plane_tests": "plane_tests", "triangle_tests": "tri_tests", "material_fetches": "material",
                   "shadow_rays_zero_term": "zero_term"}

# north_star: pixel colours within max abs error <= 1/255 per channel.  The only operation that is
# not bit-identical between the oracle (glibc powf) and the kernels (OCML powf) is pow, so the
# tests hold the float colour to a far tighter bound and the stated tolerance is kept as the outer gate.
COLOUR_TOL = 1.0 / 255.0
COLOUR_TIGHT = 2e-5


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def oracle_render(sc, cam, depth, threads=8, **kw):
    return O.render(sc.spheresData.view(O.SPHERE_DT), sc.planesData.view(O.PLANE_DT),
                    sc.trianglesData.view(O.TRI_DT), sc.AccelerationStructureData, sc.lightsData, cam, depth,
                    n_threads=threads, **kw)


def assert_frame_matches(out, ref, what=""):
    """ids bit-exact, colour within tolerance, RGBA8 off by at most one code on a rounding tie."""
    assert np.array_equal(out["primary_id"], ref["primary_id"]), f"{what}: primary hit ids differ"
    assert np.array_equal(out["final_id"], ref["final_id"]), f"{what}: final hit ids differ"
    if "rgb" in out and "rgb" in ref:
        err = np.abs(out["rgb"].astype(np.float64) - ref["rgb"].astype(np.float64))
        scale = np.maximum(1.0, np.abs(ref["rgb"].astype(np.float64)))
        assert np.nanmax(err / scale) <= COLOUR_TIGHT, f"{what}: colour error {np.nanmax(err / scale)}"
        clipped = np.abs(np.clip(out["rgb"], 0, 1) - np.clip(ref["rgb"], 0, 1))
        assert np.nanmax(clipped) <= COLOUR_TOL, f"{what}: displayed colour error {np.nanmax(clipped)}"
    if "rgba8" in out and "rgba8" in ref:
        d = np.abs(out["rgba8"].astype(np.int16) - ref["rgba8"].astype(np.int16))
        assert d.max() <= 1, f"{what}: RGBA8 differs by {d.max()} codes"
        assert (d > 0).mean() < 1e-3, f"{what}: too many RGBA8 pixels differ ({(d > 0).mean()})"


def assert_stats_match(stats, counters, what=""):
    for k, v in STAT_TO_COUNTER.items():
        if v not in counters:   # fixtures written before a bookkeeping counter existed
            continue
        assert stats[k] == counters[v], f"{what}: {k} {stats[k]} != oracle {counters[v]}"
