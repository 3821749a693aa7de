"""The oracle against the committed golden vectors and against hand-derivable known answers.

The reference holds no tests or golden vectors: the goldens are the oracle's own
outputs (tests/golden/make_golden.py) and pin it against regressions; the known-answer tests pin
it against facts that can be derived by hand from the reference's source.
"""
import math

import numpy as np
import pytest

import oracle_lib as O
from golden_cases import CASES, intersect_inputs
from helpers import crc, oracle_render


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_oracle_reproduces_golden_frames(case, frames, checksums, scene_of):
    sc = scene_of(case["scene"], case.get("params"))
    cam = case["camera"](sc)
    meta = checksums["cases"][case["id"]]
    assert crc(sc.AccelerationStructureData) == meta["crc_accel"]
    assert crc(sc.trianglesData) == meta["crc_triangles"]
    assert crc(sc.spheresData) == meta["crc_spheres"]
    assert crc(cam) == meta["crc_camera"]
    r = oracle_render(sc, cam, case["depth"])
    k = case["id"]
    assert np.array_equal(r["primary_id"], frames[k + "/primary_id"])
    assert np.array_equal(r["final_id"], frames[k + "/final_id"])
    assert np.array_equal(r["bounces"], frames[k + "/bounces"])
    assert np.array_equal(r["rgba8"], frames[k + "/rgba8"])
    np.testing.assert_allclose(r["rgb"], frames[k + "/rgb"], rtol=1e-6, atol=1e-7)
    assert r["counters"] == meta["counters"]


def test_oracle_reproduces_golden_intersections(intersect_vectors):
    v = intersect_vectors
    inp = intersect_inputs()
    for k in inp:  # the generator is deterministic: inputs are reproducible from the seed
        assert np.array_equal(inp[k].view(np.uint32), v[k].view(np.uint32)), k
    assert np.array_equal(O.intersect_box(v["box_o"], v["box_d"], v["box_b"]), v["box_hit"])
    assert np.array_equal(O.intersect_sphere(v["sph_o"], v["sph_d"], v["sph_c"]).view(np.uint32),
                          v["sphere_out"].view(np.uint32))
    assert np.array_equal(O.intersect_plane(v["pl_o"], v["pl_d"], v["pl_p"], v["pl_n"]).view(np.uint32),
                          v["plane_t"].view(np.uint32))
    assert np.array_equal(
        O.intersect_triangle(v["tri_o"], v["tri_d"], v["tri_a"], v["tri_b"], v["tri_c"], v["tri_n"]).view(np.uint32),
        v["tri_t"].view(np.uint32))
    # the vector set really exercises both outcomes of every routine
    assert 0 < v["box_hit"].mean() < 1
    assert 0 < (v["sphere_out"][:, 3] > 0).mean() < 1
    assert 0 < (v["tri_t"] > 0).mean() < 1


# ---- known answers derived by hand from raytracer_fs.glsl ------------------------------------
def test_sphere_known_answers():
    # unit sphere at the origin seen from z = -5: b = -10, c = 24, D = 4, t2 = 4, normal (0,0,-1)
    r = O.intersect_sphere([[0, 0, -5]], [[0, 0, 1]], [[0, 0, 0, 1]])[0]
    assert r.tolist() == [0.0, 0.0, -1.0, 4.0]
    # from the centre: t2 = -1 < 0 so t1 = 1 is returned, normal at the exit point (0,0,1)
    r = O.intersect_sphere([[0, 0, 0]], [[0, 0, 1]], [[0, 0, 0, 1]])[0]
    assert r.tolist() == [0.0, 0.0, 1.0, 1.0]
    # sphere behind the origin: t1 < 0 -> (-1,-1,-1,-1)
    r = O.intersect_sphere([[0, 0, 5]], [[0, 0, 1]], [[0, 0, 0, 1]])[0]
    assert r.tolist() == [-1.0, -1.0, -1.0, -1.0]
    # miss: discriminant < 0
    r = O.intersect_sphere([[0, 3, -5]], [[0, 0, 1]], [[0, 0, 0, 1]])[0]
    assert r.tolist() == [-1.0, -1.0, -1.0, -1.0]


def test_plane_and_triangle_known_answers():
    # plane y = 0 from (0,2,0) looking down: t = 2; looking up: t = -2 (callers filter); parallel: -1
    assert O.intersect_plane([[0, 2, 0]], [[0, -1, 0]], [[0, 0, 0]], [[0, 1, 0]])[0] == 2.0
    assert O.intersect_plane([[0, 2, 0]], [[0, 1, 0]], [[0, 0, 0]], [[0, 1, 0]])[0] == -2.0
    assert O.intersect_plane([[0, 2, 0]], [[1, 0, 0]], [[0, 0, 0]], [[0, 1, 0]])[0] == -1.0
    A, B, C, N = [0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]
    tri = lambda o, d, n=N: float(O.intersect_triangle([o], [d], [A], [B], [C], [n])[0])  # noqa: E731
    assert tri([0.25, 0.25, -1], [0, 0, 1]) == 1.0          # inside
    assert tri([0.0, 0.0, -1], [0, 0, 1]) == 1.0            # exactly at vertex A: edges are inclusive
    assert tri([0.5, 0.0, -1], [0, 0, 1]) == 1.0            # on edge AB
    assert tri([0.5, 0.5, -1], [0, 0, 1]) == 1.0            # on the hypotenuse
    assert tri([0.75, 0.75, -1], [0, 0, 1]) == -1.0         # outside
    assert tri([0.25, 0.25, 1], [0, 0, 1]) == -1.0          # behind (t <= 0)
    assert tri([0.25, 0.25, -1], [0, 0, 1], [0, 0, -1]) == -1.0  # winding disagrees with the stored normal
    assert tri([0.25, 0.25, -1], [1, 0, 0]) == -1.0         # parallel


def test_box_known_answers():
    box = [[-1, -1, -1, 1, 1, 1]]
    hit = lambda o, d: int(O.intersect_box([o], [d], box)[0])  # noqa: E731
    assert hit([0, 0, -5], [0, 0.1, 1]) == 1
    assert hit([0, 3, -5], [0, 0, 1]) == 0
    # no t >= 0 test: a box BEHIND the origin still passes (raytracer_fs.glsl:134-161)
    assert hit([0, 0, 5], [0, 0.1, 1]) == 1
    # origin on a slab plane with a zero direction component: 0 * inf = NaN, every comparison is
    # false and the test falls through to "hit"
    assert hit([1, 0, -5], [0, 0, 1]) == 1


def test_struct_sizes():
    # PrimitiveStructsForBuffer.cs:8-74 <-> std430
    assert O.SPHERE_DT.itemsize == 48 and O.PLANE_DT.itemsize == 64 and O.TRI_DT.itemsize == 96
    assert O.SPHERE_DT.fields["pure"][1] == 28 and O.SPHERE_DT.fields["specularity"][1] == 44
    assert O.PLANE_DT.fields["normal"][1] == 16 and O.PLANE_DT.fields["pure"][1] == 44
    assert O.TRI_DT.fields["n"][1] == 48 and O.TRI_DT.fields["diffuse"][1] == 64 and O.TRI_DT.fields["pure"][1] == 76


def test_asset_triangle_counts(ref_assets, checksums):
    # OBJImportHelper drops the last line read: the teapot has 1024 f-lines, 1023 survive
    expect = {"cube": 12, "pyramid": 6, "table": 156, "teapot": 1023}
    for name, n in expect.items():
        raw = O.import_model(f"{ref_assets}/{name}.KarelModel", 1.0, (0, 0, 0))
        assert len(raw) == n
        assert checksums["assets"][name]["triangles"] == n
        assert crc(raw) == checksums["assets"][name]["crc_raw9"]


def test_obj_dialects(tmp_path):
    # single space: y/z swapped; double space: not swapped; last line never processed
    p = tmp_path / "m.KarelModel"
    p.write_text("v 1 2 3\nv 4 5 6\nv 7 8 10\nf 1 2 3\nf 3 2 1\n")   # ends with newline: all 5 lines + nothing
    raw = O.import_model(p, 2.0, (10, 20, 30))
    assert len(raw) == 1                                             # the second f-line is the last line read
    assert raw[0].tolist() == [12, 26, 34, 18, 32, 40, 24, 40, 46]   # Position + (x, z, y) * scale
    p.write_text("v  1  2  3\nv  4  5  6\nv  7  8  10\nf  1  2  3\n\n")
    raw = O.import_model(p, 2.0, (10, 20, 30))
    assert len(raw) == 1 and raw[0].tolist() == [12, 24, 36, 18, 30, 42, 24, 36, 50]


def test_triangle_and_camera_construction():
    t = O.make_triangle([0, 0, 0, 3, 0, 0, 0, 3, 0])
    assert t[3].tolist() == [0.0, 0.0, 1.0]                          # normal of a CCW triangle in z=0
    assert np.allclose(t[:3], [[0, 0, 0], [3, 0, 0], [0, 3, 0]], atol=1e-6)
    cam = O.camera_pack([1, 5, -5], [0, -1, 1], [1, 0, 0], 1.0, 1.6, 0.9, 1280, 720)
    s = 1 / math.sqrt(2)
    centre = np.array([1, 5 - s, -5 + s])
    up = np.array([0, s, s])
    bl = centre - up * 0.45 - np.array([0.8, 0, 0])
    assert np.allclose(cam[3:6], bl, atol=1e-6) and cam[12] == 1280 and cam[13] == 720
    assert np.allclose(cam[6:9] - cam[3:6], [1.6, 0, 0], atol=1e-6)
    assert np.allclose(cam[9:12] - cam[3:6], up * 0.9, atol=1e-6)


def test_oracle_counts_rays_by_definition(scene_of):
    # depth 0, no mirrors reached: rays = W*H primary + L shadow per shaded (non-sky) pixel
    sc = scene_of("TestScene1")
    cam = sc.camera_for(64, 36)
    r = oracle_render(sc, cam, 0)
    shaded = int((r["final_id"] >= 0).sum())
    assert r["counters"]["primary"] == 64 * 36
    assert r["counters"]["secondary"] == 0
    assert r["counters"]["shadow"] == shaded * sc.n_lights
    assert int(r["bounces"].max()) == 0


def test_oracle_region_equals_crop(scene_of):
    sc = scene_of("TestScene1")
    cam = sc.camera_for(96, 54)
    full = oracle_render(sc, cam, 10)
    part = oracle_render(sc, cam, 10, region=(17, 9, 61, 40))
    assert np.array_equal(part["final_id"], full["final_id"][9:40, 17:61])
    assert np.array_equal(part["rgb"], full["rgb"][9:40, 17:61])
