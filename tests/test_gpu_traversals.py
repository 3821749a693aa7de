"""GPU parity of every fast traversal (wave walk, per-lane BVH megakernel, wavefront pipeline) through
the C ABI: each must reproduce the oracle's id planes bit for bit and its colours within tolerance,
and all must agree with the reference-order kernel float for float.  Default renders pick a
traversal by scene size (RT_TRAVERSAL_AUTO); here every mode is forced on every scene class."""
import numpy as np
import pytest

from helpers import assert_frame_matches, oracle_render
from raytracing_amd import _native as N
from raytracing_amd.host import Scene

pytestmark = pytest.mark.gpu

# 3 = the staged pipeline on the re-treed reference leaves with the reference's box formula, 4 = with the tight
# per-primitive tree as a filter in front (built for every scene size here; by default only scenes from 1000 node
# records get it); 0 = whatever AUTO picks
# 2 = the BVH megakernel (every pixel traces its own shadow rays), 5 = with the shadow rays of sparse bounces
# regrouped over the wave (RT_OPTION_COOP_SHADOWS)
MODES = [(1, "wave"), (2, "bvh"), (3, "wavefront"), (4, "wavefront+tight"), (5, "bvh+coop-shadows")]


def _tight_expected(info, mode):
    """The tight tree exists unless it is switched off (mode 3); small scenes only get it when asked (mode 4)."""
    return mode != 3 and (mode == 4 or info["n_nodes"] >= 1000)


def _set_mode(rt, mode):
    rt.set_option(rt.OPTION_TIGHT_TREE, 0 if mode == 3 else 1)
    rt.set_option(rt.OPTION_TIGHT_MIN_NODES, 0 if mode == 4 else 1000)
    rt.set_option(rt.OPTION_WAVEFRONT_MIN_NODES, 1000)
    rt.set_option(rt.OPTION_COOP_SHADOWS, {5: 1, 0: 2}.get(mode, 0))     # 2 = the default: where it measures faster
    rt.set_option(rt.OPTION_TRAVERSAL, {0: 0, 1: 1, 2: 2, 3: 3, 4: 3, 5: 2}[mode])


@pytest.fixture()
def rt_mode(rt):
    yield rt
    _set_mode(rt, 0)


def _render(rt, sc, cam, depth, **kw):
    rt.SendPrimitivesToShader(sc)
    rt.PrepareRenderOpenGL(cam)
    return rt.RenderGL(depth, **kw)


@pytest.mark.parametrize("mode,label", MODES, ids=[m[1] for m in MODES])
@pytest.mark.parametrize("name,params,depth", [("TestScene1", None, 10), ("TeapotTable", None, 8),
                                               ("Soup", [8000, 80, 8], 4)])
def test_every_traversal_matches_the_oracle(mode, label, name, params, depth, rt_mode, scene_of):
    from raytracing_amd import host
    rt = rt_mode
    sc = scene_of(name, params)
    _set_mode(rt, mode)
    g = np.random.default_rng(7)
    cams = [sc.camera_for(160, 90)]
    pos = g.uniform([-8, 1, -10], [8, 12, 10])
    view = g.uniform([-2, 0, -2], [2, 4, 6]) - pos
    cams.append(host.camera_pack(pos, view, np.cross(view, [0, -1, 0]), 1.0, 1.6, 0.9, 131, 77))
    for cam in cams:
        ref = oracle_render(sc, cam, depth, threads=16)
        out = _render(rt, sc, cam, depth)
        assert_frame_matches(out, ref, f"{name} {label}")
    info = rt.scene_info()
    assert info["flags"] == 3 and info["bvh_nodes"] > 0 and 0 < info["bvh_levels"] <= 32
    assert (info["tight_nodes"] > 0) == _tight_expected(info, mode) and info["tight_levels"] <= 32


@pytest.mark.parametrize("mode,label", MODES, ids=[m[1] for m in MODES])
def test_axis_aligned_rays(mode, label, rt_mode):
    """Rays with exactly zero direction components (1/d infinite, 0*inf = NaN in the box test) must
    take the exact tree: a camera looking straight down over an odd-sized frame, a light straight
    above the hit points, boxes whose faces pass through the ray origin."""
    rt = rt_mode
    g = np.random.default_rng(3)
    nt = 60
    tri = np.zeros(nt, N.TRIANGLE_DT)
    a = np.round(g.uniform(-4, 4, (nt, 3))) + [0, 3, 0]          # integer coordinates: faces line up with rays
    b, c = a + np.round(g.uniform(-2, 2, (nt, 3))), a + np.round(g.uniform(-2, 2, (nt, 3)))
    n = np.cross(b - a, c - a)
    keep = np.linalg.norm(n, axis=1) > 0
    tri = tri[keep]
    a, b, c, n = a[keep], b[keep], c[keep], n[keep]
    tri["PointA"], tri["PointB"], tri["PointC"] = a, b, c
    tri["Normal"] = n / np.linalg.norm(n, axis=1, keepdims=True)
    tri["DiffuseColor"] = g.uniform(0.2, 1, (len(tri), 3))
    tri["SpecularColor"], tri["Specularity"] = 0.4, 8
    tri["IsPureSpecular"][::5] = 1
    sph = np.zeros(4, N.SPHERE_DT)
    sph["Center"] = [[0, 2, 0], [2, 1, 2], [-3, 1, 1], [1, 4, -2]]
    sph["Radius"] = 1
    sph["DiffuseColor"], sph["SpecularColor"], sph["Specularity"] = 0.7, 0.5, 20
    pl = np.zeros(1, N.PLANE_DT)
    pl[0]["Position"], pl[0]["Normal"], pl[0]["DiffuseColor"] = [0, 0, 0], [0, 1, 0], [0.5, 0.5, 0.5]
    pl[0]["IsPureSpecular"], pl[0]["SpecularColor"], pl[0]["Specularity"] = 1, [0.6, 0.6, 0.6], 1
    from raytracing_amd import host
    accel = host.accel_from_structs(sph, tri)
    lights = np.array([0, 20, 0, 300, 300, 300, 2, 20, 2, 200, 100, 50], np.float32)
    # camera at (0, 20, 0) looking down -y: the middle column and row have a zero direction component,
    # the centre pixel has two
    cam = host.camera_pack([0, 20, 0], [0, -1, 0], [1, 0, 0], 1.0, 1.0, 1.0, 33, 33)
    sc = Scene(sph, pl, tri, accel, lights, cam)
    ref = oracle_render(sc, cam, 4)
    assert (ref["primary_id"] >= 0).all()
    _set_mode(rt, mode)
    out = _render(rt, sc, cam, 4)
    assert_frame_matches(out, ref, "axis-aligned " + label)
    slow = _render(rt, sc, cam, 4, stats=True)
    assert np.array_equal(out["rgb"].view(np.uint32), slow["rgb"].view(np.uint32))


@pytest.mark.parametrize("mode,label", MODES, ids=[m[1] for m in MODES])
def test_equal_t_across_leaves_takes_the_first_in_reference_order(mode, label, rt_mode):
    """Twelve coincident triangles spread over several leaves give bit-identical t on every ray: the
    reference accepts on strict <, so the first candidate of its visit order wins.  An ordered
    traversal meets the leaves in a ray-dependent order and must carry that rank explicitly."""
    from raytracing_amd import host
    rt = rt_mode
    nt = 12
    tri = np.zeros(nt, N.TRIANGLE_DT)
    tri["PointA"], tri["PointB"], tri["PointC"] = [-5, -4, 6], [5, -5, 5], [1, 5, 4]
    n = np.cross(np.array([10, -1, -1.0]), np.array([6, 9, -2.0]))
    tri["Normal"] = n / np.linalg.norm(n)
    tri["DiffuseColor"] = np.linspace(0.1, 1, nt)[:, None]
    accel = host.accel_from_structs(np.zeros(0, N.SPHERE_DT), tri)
    cam = np.array([0, 0, 0, -0.8, -0.45, 1, 0.8, -0.45, 1, -0.8, 0.45, 1, 64, 36], np.float32)
    lights = np.array([0, 0, 0, 5, 5, 5], np.float32)
    sc = Scene(np.zeros(0, N.SPHERE_DT), np.zeros(0, N.PLANE_DT), tri, accel, lights, cam)
    ref = oracle_render(sc, cam, 2)
    hit = ref["primary_id"][ref["primary_id"] >= 0]
    assert len(hit) > 200 and len(np.unique(hit)) == 1
    _set_mode(rt, mode)
    assert_frame_matches(_render(rt, sc, cam, 2), ref, "coincident " + label)


def test_hand_made_tree_without_nesting_falls_back(rt_mode, scene_of):
    """A tree whose child boxes stick out of their parent is legal input (the shader only ever tests
    the boxes it is given).  The re-treed BVH and the distance cull both rely on nesting, so the
    library must notice and render with the exact tree whatever traversal is asked for."""
    rt = rt_mode
    sc = scene_of("TeapotScene")
    accel = sc.AccelerationStructureData.copy()
    accel[0:3] += 0.75   # shrink the root box from below: children now stick out
    bad = Scene(sc.spheresData, sc.planesData, sc.trianglesData, accel, sc.lightsData, sc.cameraData)
    cam = bad.camera_for(160, 90)
    ref = oracle_render(bad, cam, 1)
    good_ref = oracle_render(sc, cam, 1)
    assert not np.array_equal(ref["primary_id"], good_ref["primary_id"])  # the shrunken root does hide hits
    for mode, label in MODES + [(0, "auto")]:
        _set_mode(rt, mode)
        out = _render(rt, bad, cam, 1)
        info = rt.scene_info()
        assert info["flags"] == 0 and info["bvh_nodes"] == 0 and info["tight_nodes"] == 0
        assert_frame_matches(out, ref, "not nested " + label)


def test_all_traversals_agree_at_full_size_soup1m(rt_mode, scene_of):
    """BASELINE config 3 (Soup1M, 3840x2160, depth 4): the id planes and float colours of the wavefront
    pipeline, the BVH megakernel and the reference-order kernel must be identical, and a crop must match
    the oracle.  (This is the frame on which two intersecting triangles produce bit-identical t.)"""
    rt = rt_mode
    sc = scene_of("Soup", [1_000_000, 10_000, 8])
    cam = sc.camera_for(3840, 2160)
    want = ("rgb", "primary_id", "final_id")
    slow = _render(rt, sc, cam, 4, want=want, stats=True)
    st = slow["stats"]
    assert st["primary_rays"] == 3840 * 2160
    for mode, label in [(3, "wavefront"), (2, "bvh"), (0, "auto")]:
        _set_mode(rt, mode)
        fast = _render(rt, sc, cam, 4, want=want)
        assert (rt.scene_info()["tight_nodes"] > 0) == (mode != 3)
        assert np.array_equal(fast["primary_id"], slow["primary_id"]), label
        assert np.array_equal(fast["final_id"], slow["final_id"]), label
        assert np.array_equal(fast["rgb"].view(np.uint32), slow["rgb"].view(np.uint32)), label
    x0, y0, x1, y1 = 1440, 960, 1504, 1000   # contains pixel (1472, 985), the equal-t pixel
    ref = oracle_render(sc, cam, 4, threads=16, region=(x0, y0, x1, y1))
    crop = {k: v[y0:y1, x0:x1] for k, v in slow.items() if k in want}
    assert_frame_matches(crop, ref, "soup1m crop")


def test_wavefront_renders_issued_on_two_streams_do_not_mix(rt_mode, scene_of):
    """The wavefront pipeline keeps its ray queues in per-context buffers; renders issued on different
    streams must be chained on the device, not interleaved (bench.py renders a step's frames over
    several streams)."""
    import torch
    from raytracing_amd import host
    rt = rt_mode
    sc = scene_of("Soup", [8000, 80, 8])
    rt.SendPrimitivesToShader(sc)
    w, h = 320, 180
    cams = [sc.camera_for(w, h), host.camera_pack([30, 25, -70], [-0.3, -0.2, 1], [1, 0, 0.3], 1.0, 1.6, 0.9, w, h)]
    _set_mode(rt, 4)
    main = torch.cuda.current_stream()
    ref = []
    for cam in cams:
        out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
        rt.render_device(cam, 4, rgba8=out.data_ptr(), stream=main.cuda_stream)
        torch.cuda.synchronize()
        ref.append(out)
    assert not torch.equal(ref[0], ref[1])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.empty((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(6)]
    for k, out in enumerate(outs):       # six renders back to back, alternating streams and cameras
        rt.render_device(cams[k % 2], 4, rgba8=out.data_ptr(), stream=streams[k % 2].cuda_stream)
    torch.cuda.synchronize()
    for k, out in enumerate(outs):
        assert torch.equal(out, ref[k % 2]), f"render {k}"


def test_normals_that_are_not_the_triangles_switch_the_cull_off(rt_mode, scene_of):
    """IntersectTriangle intersects the plane through A with the STORED normal, whatever it is.  The
    distance cull assumes hit points lie on the triangle, so a scene with skewed normals must be rendered
    without it (flags lose RT_SCENE_CULL) and still match the oracle in every traversal."""
    rt = rt_mode
    sc = scene_of("TeapotScene")
    tri = sc.trianglesData.copy()
    g = np.random.default_rng(5)
    n = tri["Normal"] + g.normal(scale=0.4, size=tri["Normal"].shape).astype(np.float32)
    tri["Normal"] = n / np.linalg.norm(n, axis=1, keepdims=True)
    skew = Scene(sc.spheresData, sc.planesData, tri, sc.AccelerationStructureData, sc.lightsData, sc.cameraData)
    cam = skew.camera_for(160, 90)
    ref = oracle_render(skew, cam, 1)
    for mode, label in MODES:
        _set_mode(rt, mode)
        out = _render(rt, skew, cam, 1)
        assert rt.scene_info()["flags"] == 1      # nested, but no cull
        assert rt.scene_info()["tight_nodes"] == 0
        assert_frame_matches(out, ref, "skewed normals " + label)


def test_slightly_skewed_normals_keep_the_cull_and_the_tight_tree(rt_mode, scene_of):
    """Normals a little off their triangles (the reference's own importer yields up to 3e-3 rad on
    slivers) stay inside the cull's tolerance.  IntersectTriangle then accepts points of the plane
    through A with the STORED normal whose projection along it falls into ABC: the tight boxes must be
    fitted to that region, not to the vertices."""
    rt = rt_mode
    sc = scene_of("TeapotScene")
    tri = sc.trianglesData.copy()
    g = np.random.default_rng(11)
    n = tri["Normal"] + g.normal(scale=0.002, size=tri["Normal"].shape).astype(np.float32)
    tri["Normal"] = n / np.linalg.norm(n, axis=1, keepdims=True)
    skew = Scene(sc.spheresData, sc.planesData, tri, sc.AccelerationStructureData, sc.lightsData, sc.cameraData)
    cam = skew.camera_for(320, 180)
    ref = oracle_render(skew, cam, 1, threads=16)
    for mode, label in MODES:
        _set_mode(rt, mode)
        out = _render(rt, skew, cam, 1)
        info = rt.scene_info()
        assert info["flags"] == 3 and (info["tight_nodes"] > 0) == _tight_expected(info, mode)
        assert_frame_matches(out, ref, "slightly skewed normals " + label)


def test_rays_from_far_outside_the_scene_skip_the_tight_tree(rt_mode, scene_of):
    """The tight boxes are padded for ray origins within 3 scene radii of the world origin; a camera
    further out must fall back to the leaf-box tree for its primary rays (and still use the tight tree
    for the shadow and bounce rays that start inside the scene)."""
    from raytracing_amd import host
    rt = rt_mode
    sc = scene_of("Soup", [8000, 80, 8])
    for dist in (100.0, 1000.0, 30000.0):
        cam = host.camera_pack([0, 0.3 * dist + 10, -dist], [0, -0.3, 1], [1, 0, 0], 1.0, 0.8 * 40 / dist, 0.45 * 40 / dist,
                               160, 90)
        ref = oracle_render(sc, cam, 4, threads=16)
        assert (ref["primary_id"] >= 0).mean() > 0.2
        _set_mode(rt, 4)
        out = _render(rt, sc, cam, 4)
        assert rt.scene_info()["tight_nodes"] > 0
        assert_frame_matches(out, ref, f"camera at {dist}")


def test_a_point_triangle_with_a_finite_normal_is_a_plane(rt_mode, scene_of):
    """Coincident vertices make every edge test 0 >= 0: the reference accepts the whole plane through A
    wherever the ray passes the leaf's box test, which has no distance bound -- hits far outside any
    box around the vertices.  The distance cull and the tight tree assume hits near their primitive, so
    the upload must mark such triangles (scene info: unbounded_triangles), never cull the subtrees that
    hold them, give them their padded reference leaf box in the tight tree (walked as a line test), and every traversal must still reproduce the oracle.
    (The reference's own table model has 19 triangles with collinear vertices and a finite normal.)"""
    from raytracing_amd import host
    rt = rt_mode
    sc = scene_of("TeapotScene")
    tri = sc.trianglesData.copy()
    k = len(tri) // 2
    p = tri["PointA"][k].copy()
    tri["PointB"][k], tri["PointC"][k] = p, p
    tri["Normal"][k] = [0.0, 0.8, -0.6]
    j = k + 7                                # and one with collinear vertices: accepted along a line
    tri["PointC"][j] = tri["PointA"][j] + 2.5 * (tri["PointB"][j] - tri["PointA"][j])
    accel = host.accel_from_structs(sc.spheresData, tri)
    odd = Scene(sc.spheresData, sc.planesData, tri, accel, sc.lightsData, sc.cameraData)
    cam = odd.camera_for(160, 90)
    ref = oracle_render(odd, cam, 1)
    tid = len(sc.spheresData) + k
    assert (ref["primary_id"] == tid).sum() >= 1      # a point would cover none
    for mode, label in MODES:
        _set_mode(rt, mode)
        out = _render(rt, odd, cam, 1)
        info = rt.scene_info()
        assert info["flags"] == 3 and info["unbounded_triangles"] == 2
        assert (info["tight_nodes"] > 0) == _tight_expected(info, mode)
        assert_frame_matches(out, ref, "point triangle " + label)


def test_slivers_keep_every_traversal_exact(rt_mode, scene_of):
    """Long thin triangles: rounding lets the reference's edge tests accept points past a sharp vertex (by about
    8 eps * longest edge / angle).  The tight boxes add that reach to their pad and the longest slivers are marked
    unbounded (tests/test_scene_prep.py); every traversal must reproduce the oracle on a scene full of them, seen
    from a camera that looks along many of their long axes."""
    from raytracing_amd import host
    rt = rt_mode
    sc = scene_of("TeapotScene")
    tri = sc.trianglesData.copy()
    g = np.random.default_rng(23)
    idx = np.nonzero(np.isfinite(tri["Normal"]).all(axis=1))[0]
    idx = g.choice(idx, 300, replace=False)
    for i in idx:
        a = tri["PointA"][i].astype(np.float64)
        n = tri["Normal"][i].astype(np.float64)
        u = np.cross(n, g.normal(size=3)); u /= np.linalg.norm(u)
        v = np.cross(n, u)
        L, angle = 10.0 ** g.uniform(-1, 1.3), 10.0 ** g.uniform(-4.5, -2)
        tri["PointB"][i] = (a + L * u).astype(np.float32)
        tri["PointC"][i] = (a + g.uniform(0.1, 0.9) * L * u + L * angle * v).astype(np.float32)
    accel = host.accel_from_structs(sc.spheresData, tri)
    odd = Scene(sc.spheresData, sc.planesData, tri, accel, sc.lightsData, sc.cameraData)
    cam = odd.camera_for(240, 135)
    ref = oracle_render(odd, cam, 2, threads=16)
    sliver_ids = set((len(sc.spheresData) + idx).tolist())
    assert len(sliver_ids & set(np.unique(ref["primary_id"]).tolist())) >= 5      # some slivers are actually seen
    for mode, label in MODES:
        _set_mode(rt, mode)
        out = _render(rt, odd, cam, 2)
        info = rt.scene_info()
        assert info["flags"] == 3 and info["unbounded_triangles"] > 0
        assert_frame_matches(out, ref, "slivers " + label)


@pytest.mark.parametrize("shift", [(0.0, 0.0, 0.0), (3000.0, 700.0, -5000.0)], ids=["at-origin", "far-from-origin"])
def test_rays_that_graze_the_root_box(shift, rt_mode, scene_of):
    """The per-lane kernels skip a ray's traversal set-up when its line certainly misses a sphere around the root
    box (rt_device.h misses_scene).  Cameras zoomed onto the silhouette of the root box -- corners, edges, from
    near and far -- and the same scene moved far from the world origin (a small box at large coordinates: the
    margins scale with the coordinates) must reproduce the oracle in every traversal."""
    from raytracing_amd import host
    rt = rt_mode
    sc = scene_of("TeapotScene")
    tri = sc.trianglesData.copy()
    sh = np.array(shift, dtype=np.float32)
    for k in ("PointA", "PointB", "PointC"):
        tri[k] = tri[k] + sh
    accel = host.accel_from_structs(sc.spheresData, tri)
    lights = np.asarray(sc.lightsData, dtype=np.float32).copy().reshape(-1, 6)
    lights[:, 0:3] += sh
    moved = Scene(sc.spheresData, sc.planesData[:0], tri, accel, lights.reshape(-1), sc.cameraData)
    # the vertices that define the root box lie ON its faces: look at each along a direction nearly inside that face
    pts = np.concatenate([tri["PointA"], tri["PointB"], tri["PointC"]]).astype(np.float64)
    size = np.linalg.norm(pts.max(0) - pts.min(0))
    seen = 0
    for axis in range(3):
        for sign, dist, fov in ((1, 2.0, 0.03), (-1, 60.0, 0.001)):
            v = pts[np.argmax(sign * pts[:, axis])]
            along = np.array([0.55, 0.33, -0.77])
            along[axis] = 0.004 * sign                      # 0.2 degrees off the face
            eye = v + dist * size * along / np.linalg.norm(along)
            fwd = (v - eye) / np.linalg.norm(v - eye)
            right = np.cross([0.0, 1.0, 0.0], fwd)
            right /= np.linalg.norm(right)
            cam = host.camera_pack(eye.tolist(), fwd.tolist(), right.tolist(), 1.0, fov * 1.6, fov * 0.9, 96, 54)
            ref = oracle_render(moved, cam, 1, threads=16)
            hit = (ref["primary_id"] >= 0).mean()
            seen += 0.0 < hit < 1.0                             # the model's silhouette crosses the frame
            for mode, label in MODES:
                _set_mode(rt, mode)
                out = _render(rt, moved, cam, 1)
                assert_frame_matches(out, ref, f"grazing the {'+' if sign > 0 else '-'}{'xyz'[axis]} face {label}")
    assert seen >= 4


def test_the_table_models_collinear_triangles_are_marked(rt_mode, scene_of):
    rt = rt_mode
    rt.SendPrimitivesToShader(scene_of("TeapotTable"))
    info = rt.scene_info()
    assert info["flags"] == 3 and info["unbounded_triangles"] == 21   # 19 with collinear vertices + 2 long slivers
    rt.SendPrimitivesToShader(scene_of("TeapotScene"))
    assert rt.scene_info()["unbounded_triangles"] == 0


@pytest.mark.parametrize("mode,label", MODES, ids=[m[1] for m in MODES])
def test_untraced_shadow_rays_leave_every_float_as_it_was(mode, label, rt_mode, scene_of):
    """The fast kernels do not trace a shadow ray when the light's term is exactly zero whichever way
    the ray ends.  That must hold bit for bit, also where zero is delicate: negative-zero and negative
    diffuse colours, negative light colours, black speculars, specularity 0 (pow(0, 0) = 1) -- compared
    float for float with the reference-order kernel, which traces everything."""
    rt = rt_mode
    sc = scene_of("Soup", [6000, 60, 8])
    g = np.random.default_rng(17)
    tri, sph, pl = sc.trianglesData.copy(), sc.spheresData.copy(), sc.planesData.copy()
    for rec in (tri, sph):
        n = len(rec)
        pick = g.integers(0, 6, n)
        rec["DiffuseColor"][pick == 0] = -0.0
        rec["DiffuseColor"][pick == 1] *= -1.0
        rec["DiffuseColor"][pick == 2] = 0.0
        rec["SpecularColor"][g.integers(0, 4, n) == 0] = 0.0
        rec["Specularity"][g.integers(0, 5, n) == 0] = 0.0
    pl["DiffuseColor"] = -0.0
    lights = sc.lightsData.copy().reshape(-1, 6)
    lights[1, 3:] *= -1.0
    lights[2, 3:] = 0.0
    odd = Scene(sph, pl, tri, sc.AccelerationStructureData, lights.ravel(), sc.cameraData)
    cam = odd.camera_for(256, 144)
    want = ("rgb", "primary_id", "final_id")
    slow = _render(rt, odd, cam, 4, want=want, stats=True)
    _set_mode(rt, mode)
    fast = _render(rt, odd, cam, 4, want=want)
    assert np.array_equal(fast["primary_id"], slow["primary_id"]) and np.array_equal(fast["final_id"], slow["final_id"])
    assert np.array_equal(fast["rgb"].view(np.uint32), slow["rgb"].view(np.uint32))
    assert_frame_matches(fast, oracle_render(odd, cam, 4, threads=16), "odd materials " + label)


def test_auto_times_both_kernels_on_mid_sized_scenes(rt_mode, scene_of):
    """Between 1000 and 100000 flattened nodes AUTO renders three frames each with the megakernel, the staged pipeline
    and the megakernel with cooperative shadow rays (a warm-up, then two between events) and then keeps the fastest: a
    field of teapots stays on the megakernel (most rays miss everything; the staged pipeline's per-ray bookkeeping
    would dominate); a triangle soup goes to whichever measured faster (the staged pipeline until the megakernel
    learnt to walk the tight tree, since then a close call at this size).  Every frame on the way is the same frame."""
    import torch
    rt = rt_mode
    sc_soup = scene_of("Soup", [10000, 100, 8])
    base = scene_of("TeapotScene")
    tris = []
    for k in range(12):
        t = base.trianglesData.copy()
        for f in ("PointA", "PointB", "PointC"):
            t[f] += np.array([(k % 6 - 2.5) * 3.0, 0.0, (k // 6) * 3.5 - 2.0], np.float32)
        tris.append(t)
    from raytracing_amd import host
    tri = np.concatenate(tris)
    field = Scene(base.spheresData, base.planesData, tri, host.accel_from_structs(base.spheresData, tri),
                  base.lightsData, base.cameraData)
    st = torch.cuda.current_stream().cuda_stream
    for sc, expect in ((sc_soup, (rt.TRAVERSAL_WAVEFRONT, rt.TRAVERSAL_BVH)), (field, (rt.TRAVERSAL_BVH,))):
        _set_mode(rt, 0)
        rt.set_option(rt.OPTION_WAVEFRONT_MIN_NODES, 1000)
        rt.SendPrimitivesToShader(sc)
        assert 1000 <= rt.scene_info()["n_nodes"] < 100000 and rt.scene_info()["tight_nodes"] > 0
        cam = sc.camera_for(640, 360)
        frames = []
        for _ in range(11):
            out = torch.empty((360, 640, 4), dtype=torch.uint8, device="cuda")
            rt.render_device(cam, 4, rgba8=out.data_ptr(), stream=st)
            torch.cuda.synchronize()
            frames.append(out)
        assert all(torch.equal(frames[0], f) for f in frames[1:])
        assert rt.scene_info()["traversal"] in expect


@pytest.mark.parametrize("mode,label", MODES, ids=[m[1] for m in MODES])
def test_edge_scenes_in_every_traversal(mode, label, rt_mode, scene_of):
    """Small and lopsided inputs through every traversal (the defaults would send them all to the
    megakernel): no lights, spheres only, one triangle, no geometry under several planes, depth 0 and a
    depth beyond the staged pipeline's bounce cap (falls back to the megakernel)."""
    from raytracing_amd import host
    rt = rt_mode
    ts = scene_of("TestScene1")
    cam = ts.camera_for(96, 54)
    none_s, none_t = np.zeros(0, N.SPHERE_DT), np.zeros(0, N.TRIANGLE_DT)
    one = ts.trianglesData[:1].copy()
    one["PointA"], one["PointB"], one["PointC"] = [-3, 1, 6], [3, 1, 6], [0, 6, 7]
    n = np.cross(np.array([6.0, 0, 0]), np.array([3.0, 5, 1]))
    one["Normal"] = n / np.linalg.norm(n)
    cases = [
        ("no lights", Scene(ts.spheresData, ts.planesData, ts.trianglesData, ts.AccelerationStructureData,
                            np.zeros(0, np.float32), cam), 10),
        ("spheres only", Scene(ts.spheresData, ts.planesData, none_t, host.accel_from_structs(ts.spheresData, none_t),
                               ts.lightsData, cam), 10),
        ("one triangle", Scene(none_s, ts.planesData, one, host.accel_from_structs(none_s, one), ts.lightsData, cam), 10),
        ("planes only", Scene(none_s, np.repeat(ts.planesData, 2), none_t, np.zeros(0, np.float32), ts.lightsData, cam), 10),
        ("depth 0", ts, 0),
        ("depth 65", ts, 65),
    ]
    for what, sc, depth in cases:
        _set_mode(rt, mode)
        out = _render(rt, sc, cam, depth)
        assert_frame_matches(out, oracle_render(sc, cam, depth), f"{what} {label}")


@pytest.mark.parametrize("tight_everywhere", [False, True], ids=["leaf-tree-only", "with-tight-tree"])
@pytest.mark.parametrize("name,depth", [("TestScene1", 10), ("TeapotTable", 8)])
def test_lds_resident_kernel_matches_the_oracle(name, depth, tight_everywhere, rt_mode, scene_of):
    """RT_OPTION_USE_LDS (k_render_bvh_lds, GeomBvhLds): the kernel's per-lane stacks and the host's LDS allocation
    must use the same depth -- the leaf tree's, the only tree it walks -- whether or not the scene also has a tight
    tree (they once disagreed, and the upper waves wrote past the allocation)."""
    rt = rt_mode
    sc = scene_of(name)
    rt.set_option(rt.OPTION_TIGHT_TREE, 1)
    rt.set_option(rt.OPTION_TIGHT_MIN_NODES, 0 if tight_everywhere else 30000)
    rt.set_option(rt.OPTION_TRAVERSAL, rt.TRAVERSAL_BVH)
    rt.set_option(rt.OPTION_USE_LDS, 1)
    try:
        for cam in (sc.camera_for(160, 90), sc.camera_for(333, 190)):   # 333x190: every wave of the 1024-thread groups works
            ref = oracle_render(sc, cam, depth, threads=16)
            out = _render(rt, sc, cam, depth)
            assert_frame_matches(out, ref, f"{name} lds")
        info = rt.scene_info()
        assert 0 < info["lds_scene_bytes"] <= 160 * 1024, "the scene must actually fit, or the LDS kernel did not run"
        assert (info["tight_nodes"] > 0) == tight_everywhere
    finally:
        rt.set_option(rt.OPTION_USE_LDS, 0)
        rt.set_option(rt.OPTION_TIGHT_MIN_NODES, 1000)


@pytest.mark.parametrize("n_lights", [1, 9, 20, 32, 33])
def test_cooperative_kernel_with_many_lights(n_lights, rt_mode, scene_of):
    """The cooperative kernel queues at most 8 lights' rays per fill (9, 20 and 32 lights: several fills per
    shading step, 64 lanes x 8 lights fill the queue exactly) and keeps one mask bit per light: from 33 lights on
    the plain megakernel renders the frame.  Lights behind surfaces, one with a negative colour, one black."""
    rt = rt_mode
    sc = scene_of("TestScene1")
    g = np.random.default_rng(n_lights)
    lights = np.zeros((n_lights, 6), np.float32)
    lights[:, 0:3] = g.uniform([-12, -2, -12], [12, 14, 12], (n_lights, 3))     # some below the floor
    lights[:, 3:6] = g.uniform(2, 30, (n_lights, 3))
    if n_lights > 2:
        lights[1, 3:] *= -1.0
        lights[2, 3:] = 0.0
    many = Scene(sc.spheresData, sc.planesData, sc.trianglesData, sc.AccelerationStructureData, lights.ravel(), sc.cameraData)
    cam = sc.camera_for(128, 72)
    ref = oracle_render(many, cam, 6, threads=16)
    _set_mode(rt, 5)
    out = _render(rt, many, cam, 6)
    assert_frame_matches(out, ref, f"{n_lights} lights, cooperative")
    _set_mode(rt, 2)
    plain = _render(rt, many, cam, 6)
    assert np.array_equal(out["rgb"].view(np.uint32), plain["rgb"].view(np.uint32))


@pytest.mark.parametrize("mode,label", [(2, "bvh"), (5, "bvh+coop-shadows")])
def test_order_by_the_previous_frame_changes_no_pixel(mode, label, rt_mode, scene_of):
    """Single frames of a scene whose root box spans the view (no priority columns) and whose paths may bounce three
    times or more are dealt out in an order ranked from the previous launch's wave lifetimes (k_rank_supers).  The
    first frame has no marks, the next ones do; a camera cut and a change of size leave stale marks behind.  Every
    frame must equal the reference-order kernel's, whatever order its tiles were rendered in."""
    from raytracing_amd import host
    rt = rt_mode
    sc = scene_of("TestScene1")
    _set_mode(rt, mode)
    rt.SendPrimitivesToShader(sc)
    cams = [sc.camera_for(320, 180)] * 3 + [host.camera_orbit(40, 96, 320, 180)] * 2 + [sc.camera_for(200, 120)] * 2 + \
           [sc.camera_for(320, 180)]
    try:
        for k, cam in enumerate(cams):
            # frames 4 and 5 with the feedback switched off (RT_OPTION_FRAME_FEEDBACK: timings independent of the frame before)
            rt.set_option(rt.OPTION_FRAME_FEEDBACK, 0 if k in (4, 5) else 1)
            rt.PrepareRenderOpenGL(cam)
            out = rt.RenderGL(10)
            ref = rt.RenderGL(10, stats=True)            # (stats: the reference-order kernel)
            for plane in ("rgb", "primary_id", "final_id", "rgba8"):
                assert np.array_equal(out[plane].view(np.uint8), ref[plane].view(np.uint8)), (label, k, plane)
    finally:
        rt.set_option(rt.OPTION_FRAME_FEEDBACK, 1)


@pytest.mark.parametrize("name,params,depth", [("TestScene1", None, 10), ("TeapotTable", None, 8), ("Soup", [20000, 200, 8], 4)])
def test_binned_path_lists_change_no_pixel(name, params, depth, rt_mode, scene_of):
    """RT_OPTION_BIN_PATHS sorts the lists of continuing paths by origin cell and direction octant between the bounces
    of the staged pipeline (counting sort on the device): every plane must stay bit-identical to the unsorted run's,
    from whichever bounce the sorting starts, and match the oracle."""
    rt = rt_mode
    sc = scene_of(name, params)
    _set_mode(rt, 4)
    cam = sc.camera_for(200, 120)
    try:
        rt.set_option(rt.OPTION_BIN_PATHS, 0)
        base = _render(rt, sc, cam, depth)
        assert_frame_matches(base, oracle_render(sc, cam, depth, threads=16), name + " unsorted")
        for first in (1, 2, depth, depth + 1):
            rt.set_option(rt.OPTION_BIN_PATHS, first)
            out = _render(rt, sc, cam, depth)
            for plane in ("primary_id", "final_id"):
                assert np.array_equal(out[plane], base[plane]), (name, first, plane)
            assert np.array_equal(out["rgb"].view(np.uint32), base["rgb"].view(np.uint32)), (name, first)
    finally:
        rt.set_option(rt.OPTION_BIN_PATHS, 0)


@pytest.mark.parametrize("name,params,depth", [("TestScene1", None, 10), ("TeapotTable", None, 8), ("Soup", [20000, 200, 8], 4)])
def test_packet_walks_change_no_pixel(name, params, depth, rt_mode, scene_of):
    """RT_OPTION_PACKET_WALKS: the primary rays of a tile, and the mirror bounce of its pixels on one pure-specular plane
    (the scenes' mirror floors), walk the tight tree as one packet.  Every plane must equal the per-ray walks' bit for bit
    -- on a frame with ragged tiles, from a camera low over the floor (most tiles bounce), from one inside the scene --
    and match the oracle."""
    from raytracing_amd import host
    rt = rt_mode
    sc = scene_of(name, params)
    _set_mode(rt, 4)
    cams = [sc.camera_for(333, 190),
            host.camera_pack([3.0, 0.4, -9.0], [-0.2, 0.05, 1.0], [1, 0, 0.2], 1.0, 1.6, 0.9, 161, 93),
            host.camera_pack([0.5, 2.0, 0.5], [0.3, -0.4, 1.0], [1, 0, -0.3], 1.0, 1.6, 0.9, 128, 72)]
    try:
        for cam in cams:
            rt.set_option(rt.OPTION_PACKET_WALKS, 0)
            base = _render(rt, sc, cam, depth)
            rt.set_option(rt.OPTION_PACKET_WALKS, 1)
            out = _render(rt, sc, cam, depth)
            for plane in ("primary_id", "final_id"):
                assert np.array_equal(out[plane], base[plane]), (name, plane)
            assert np.array_equal(out["rgb"].view(np.uint32), base["rgb"].view(np.uint32)), name
        assert_frame_matches(out, oracle_render(sc, cams[-1], depth, threads=16), name + " packets")
    finally:
        rt.set_option(rt.OPTION_PACKET_WALKS, 1)
