"""GPU edge cases and error behaviour of the C ABI."""
import numpy as np
import pytest

from helpers import assert_frame_matches, assert_stats_match, oracle_render
from raytracing_amd import RayTracer, RtError
from raytracing_amd import _native as N
from raytracing_amd.host import Scene

pytestmark = pytest.mark.gpu


def _empty_scene(camera, lights=None):
    return Scene(np.zeros(0, N.SPHERE_DT), np.zeros(0, N.PLANE_DT), np.zeros(0, N.TRIANGLE_DT),
                 np.zeros(0, np.float32), np.zeros(0, np.float32) if lights is None else lights, camera)


def test_render_before_upload_is_an_error():
    with RayTracer(0) as fresh:
        fresh.PrepareRenderOpenGL(np.array([0] * 12 + [16, 16], np.float32))
        with pytest.raises(RtError) as e:
            fresh.RenderGL(1)
        assert e.value.code == -3


def test_bad_arguments(rt, scene_of):
    sc = scene_of("TestScene1")
    rt.SendPrimitivesToShader(sc)
    for w, h in [(0, 10), (10.5, 10), (-4, 4), (np.nan, 4), (1e9, 4)]:
        cam = sc.cameraData.copy()
        cam[12], cam[13] = w, h
        rt.PrepareRenderOpenGL(cam)
        with pytest.raises(RtError) as e:
            rt.RenderGL(1)
        assert e.value.code == -1
    rt.PrepareRenderOpenGL(sc.camera_for(16, 16))
    with pytest.raises(RtError):
        rt.RenderGL(-1)
    with pytest.raises(RtError):  # lights_len not a multiple of 6
        rt.upload_raw(sc.spheresData, sc.planesData, sc.trianglesData, sc.AccelerationStructureData,
                      sc.lightsData[:5])
    with pytest.raises(RtError) as e:  # ids beyond the triangle array
        rt.upload_raw(sc.spheresData, sc.planesData, sc.trianglesData[:100], sc.AccelerationStructureData,
                      sc.lightsData)
    assert e.value.code == -2
    bad = sc.AccelerationStructureData.copy()
    bad[8] = 0
    with pytest.raises(RtError) as e:
        rt.upload_raw(sc.spheresData, sc.planesData, sc.trianglesData, bad, sc.lightsData)
    assert e.value.code == -2
    # a failed upload leaves the context without a scene rather than with a half-updated one
    with pytest.raises(RtError) as e:
        rt.RenderGL(1)
    assert e.value.code == -3
    rt.SendPrimitivesToShader(sc)
    rt.RenderGL(1, want=("primary_id",))


def test_empty_scene_is_all_sky(rt):
    cam = np.array([0, 1, 0, -0.8, 0.55, 1, 0.8, 0.55, 1, -0.8, 1.45, 1, 33, 17], np.float32)
    sc = _empty_scene(cam)
    rt.SendPrimitivesToShader(sc)
    rt.PrepareRenderOpenGL(cam)
    out = rt.RenderGL(10, stats=True)
    assert (out["primary_id"] == -1).all() and (out["final_id"] == -1).all()
    assert np.allclose(out["rgb"], [0.5, 0.6, 1.0])
    assert (out["rgba8"] == [128, 153, 255, 255]).all()
    assert out["stats"]["primary_rays"] == 33 * 17 and out["stats"]["shadow_rays"] == 0
    assert out["stats"]["node_tests"] == 0
    assert_frame_matches(out, oracle_render(sc, cam, 10), "empty")


def test_planes_only_and_no_lights(rt, scene_of):
    ts = scene_of("TestScene1")
    cam = ts.camera_for(64, 36)
    sc = Scene(np.zeros(0, N.SPHERE_DT), ts.planesData, np.zeros(0, N.TRIANGLE_DT), np.zeros(0, np.float32),
               np.zeros(0, np.float32), cam)
    rt.SendPrimitivesToShader(sc)
    rt.PrepareRenderOpenGL(cam)
    out = rt.RenderGL(10, stats=True)
    ref = oracle_render(sc, cam, 10)
    assert_frame_matches(out, ref, "planes only")
    assert_stats_match(out["stats"], ref["counters"], "planes only")
    assert out["stats"]["shadow_rays"] == 0
    # several planes, brute force in index order (ties resolved by the first)
    planes = np.repeat(ts.planesData, 3)
    planes[1]["Position"] = [0, 0, 0]                     # coincident with plane 0: plane 0 must win
    planes[2]["Position"], planes[2]["Normal"] = [0, 0, 8], [0, 0, -1]
    planes[2]["IsPureSpecular"] = 0
    sc = Scene(ts.spheresData, planes, np.zeros(0, N.TRIANGLE_DT), _spheres_only_accel(ts), ts.lightsData, cam)
    rt.SendPrimitivesToShader(sc)
    out = rt.RenderGL(10, stats=True)
    ref = oracle_render(sc, cam, 10)
    assert_frame_matches(out, ref, "three planes")
    assert_stats_match(out["stats"], ref["counters"], "three planes")
    assert not (out["primary_id"] == 8 + 0 + 1).any()


def _spheres_only_accel(ts):
    from raytracing_amd import host
    return host.accel_from_structs(ts.spheresData, np.zeros(0, N.TRIANGLE_DT))


@pytest.mark.parametrize("w,h", [(1, 1), (1, 40), (40, 1), (17, 16), (16, 17), (255, 3), (130, 67)])
def test_ragged_sizes(w, h, rt, scene_of):
    sc = scene_of("TestScene1")
    cam = sc.camera_for(w, h)
    rt.SendPrimitivesToShader(sc)
    rt.PrepareRenderOpenGL(cam)
    out = rt.RenderGL(10, stats=True)
    ref = oracle_render(sc, cam, 10)
    assert_frame_matches(out, ref, f"{w}x{h}")
    assert_stats_match(out["stats"], ref["counters"], f"{w}x{h}")
    assert_frame_matches(rt.RenderGL(10), ref, f"{w}x{h} fast")


def test_mixed_and_oversized_leaves(rt):
    # hand-written tree: a root branch over one leaf that mixes 3 spheres with 40 triangles (cut into
    # homogeneous pieces of <= 31 slots on upload) and one empty leaf
    g = np.random.default_rng(11)
    ns, nt = 3, 40
    sph = np.zeros(ns, N.SPHERE_DT)
    sph["Center"] = g.uniform(-2, 2, (ns, 3)) + [0, 2, 6]
    sph["Radius"] = 0.6
    sph["DiffuseColor"] = g.uniform(0.2, 1, (ns, 3))
    sph["SpecularColor"] = 0.5
    sph["Specularity"] = 10
    sph["IsPureSpecular"][0] = 1
    tri = np.zeros(nt, N.TRIANGLE_DT)
    a = g.uniform(-3, 3, (nt, 3)) + [0, 2, 7]
    b, c = a + g.uniform(-1, 1, (nt, 3)), a + g.uniform(-1, 1, (nt, 3))
    n = np.cross(b - a, c - a)
    tri["PointA"], tri["PointB"], tri["PointC"] = a, b, c
    tri["Normal"] = n / np.linalg.norm(n, axis=1, keepdims=True)
    tri["DiffuseColor"] = g.uniform(0.2, 1, (nt, 3))
    tri["SpecularColor"] = 0.3
    tri["Specularity"] = 5
    tri["IsPureSpecular"][::7] = 1
    box = [-6, -2, 0, 6, 6, 12]
    ids = list(g.permutation(ns + nt))
    leaf = box + [1, len(ids)] + [float(i) for i in ids]
    empty_leaf = [5, 5, 5, 6, 6, 6, 1, 0]
    root = box + [0, 2, 10.0, float(10 + len(leaf))]
    accel = np.array(root + leaf + empty_leaf, np.float32)
    cam = np.array([0, 2, -2, -0.8, 1.55, -1, 0.8, 1.55, -1, -0.8, 2.45, -1, 96, 54], np.float32)
    lights = np.array([0, 8, 0, 60, 60, 60, 4, 3, 2, 10, 20, 30], np.float32)
    sc = Scene(sph, np.zeros(0, N.PLANE_DT), tri, accel, lights, cam)
    rt.SendPrimitivesToShader(sc)
    info = rt.scene_info()
    assert info["n_nodes"] == 1 + 3 + 1 and info["n_tri_slots"] == nt and info["n_sphere_slots"] == ns
    rt.PrepareRenderOpenGL(cam)
    out = rt.RenderGL(6, stats=True)
    ref = oracle_render(sc, cam, 6)
    assert (ref["primary_id"] >= 0).mean() > 0.05
    assert len(np.unique(ref["primary_id"])) > 12
    assert_frame_matches(out, ref, "mixed leaf")
    assert_stats_match(out["stats"], ref["counters"], "mixed leaf")
    assert_frame_matches(rt.RenderGL(6), ref, "mixed leaf fast")


def test_equal_distance_tie_break(rt):
    # two coincident triangles and a coincident plane: the shader's order is spheres, planes,
    # triangles with strict <, so the plane wins over both triangles; without the plane the first
    # triangle in leaf order wins
    tri = np.zeros(2, N.TRIANGLE_DT)
    for k in range(2):
        tri[k]["PointA"], tri[k]["PointB"], tri[k]["PointC"], tri[k]["Normal"] = [-5, -5, 4], [5, -5, 4], [0, 5, 4], \
            [0, 0, 1]
        tri[k]["DiffuseColor"] = [1, 0, 0] if k == 0 else [0, 1, 0]
    pl = np.zeros(1, N.PLANE_DT)
    pl[0]["Position"], pl[0]["Normal"], pl[0]["DiffuseColor"] = [0, 0, 4], [0, 0, -1], [0, 0, 1]
    accel = np.array([-5, -5, 4, 5, 5, 4, 1, 2, 1, 0], np.float32)  # leaf lists triangle 1 BEFORE triangle 0
    cam = np.array([0, 0, 0, -0.8, -0.45, 1, 0.8, -0.45, 1, -0.8, 0.45, 1, 32, 18], np.float32)
    lights = np.array([0, 0, 0, 5, 5, 5], np.float32)
    for planes, want in [(pl, 2), (np.zeros(0, N.PLANE_DT), 1)]:
        sc = Scene(np.zeros(0, N.SPHERE_DT), planes, tri, accel, lights, cam)
        rt.SendPrimitivesToShader(sc)
        rt.PrepareRenderOpenGL(cam)
        ref = oracle_render(sc, cam, 2)
        for stats in (False, True):
            out = rt.RenderGL(2, stats=stats)
            assert_frame_matches(out, ref, "tie")
            assert out["primary_id"][9, 16] == want


def test_reupload_replaces_scene(rt, scene_of):
    a, b = scene_of("TestScene1"), scene_of("TeapotScene")
    cam = a.camera_for(80, 45)
    rt.SendPrimitivesToShader(a)
    rt.PrepareRenderOpenGL(cam)
    fa = rt.RenderGL(3, want=("final_id",))["final_id"]
    rt.SendPrimitivesToShader(b)
    fb = rt.RenderGL(3, want=("final_id",))["final_id"]
    rt.SendPrimitivesToShader(a)
    assert np.array_equal(rt.RenderGL(3, want=("final_id",))["final_id"], fa)
    assert not np.array_equal(fa, fb)
