"""Debug overlay data: what the reference's Debug2D views draw (RayTracer.cs:144-940) as
arrays instead of pixels -- the R-tree boxes (a second reader of the serialised format, as
RayTracer.cs:353-370) and a fan of sample rays across the middle of the view with their hits."""
import numpy as np

from . import _native as N


def rtree_boxes(accel):
    """Serialised R-tree (RTree.cs:111-178) -> record array, one row per node in the shader's visit order:
    offset (float index of the node), depth, is_leaf, count, lo[3], hi[3]."""
    acc = np.asarray(accel, np.float32)
    dt = np.dtype([("offset", "<i8"), ("depth", "<i4"), ("is_leaf", "?"), ("count", "<i4"), ("lo", "<f4", 3),
                   ("hi", "<f4", 3)])
    rows, stack = [], ([(0, 0)] if len(acc) else [])
    while stack:
        pos, depth = stack.pop()                      # raytracer_fs.glsl:169-171
        leaf, cnt = acc[pos + 6] != 0, int(acc[pos + 7])
        rows.append((pos, depth, leaf, cnt, acc[pos:pos + 3], acc[pos + 3:pos + 6]))
        if not leaf:
            stack += [(pos + int(acc[pos + 8 + i]), depth + 1) for i in range(cnt)]   # last child pops first
    return np.array(rows, dtype=dt)


def sample_rays(rt, n=15, max_bounces=0):
    """n primary rays spread over the middle scanline of rt's current camera (the reference draws 15,
    RayTracer.cs:223-230): origin, directions [n, 3] and the id of what each one hits (-1 = nothing).
    Rendered by the library as an n x 1 frame of the same image plane."""
    cam = rt.cameraData.copy()
    cam[12], cam[13] = float(n), 1.0
    saved = rt.cameraData
    rt.PrepareRenderOpenGL(cam)
    try:
        ids = rt.RenderGL(max_bounces, want=("primary_id",))["primary_id"][0]
    finally:
        rt.PrepareRenderOpenGL(saved)
    o, bl, br, tl = cam[0:3], cam[3:6], cam[6:9], cam[9:12]
    x = (np.arange(n, dtype=np.float32) + np.float32(0.5)) / np.float32(n)
    d = (bl + x[:, None] * (br - bl) + np.float32(0.5) * (tl - bl)) - o   # raytracer_fs.glsl:302-313
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return {"origin": o.copy(), "directions": d.astype(np.float32), "primary_id": ids}


def hit_distance(scene, origin, direction, hit_id):
    """Distance along a sample ray to the primitive the library reported (for drawing the ray's end point)."""
    S, T = len(scene.spheresData), len(scene.trianglesData)
    o, d = np.asarray(origin, np.float64), np.asarray(direction, np.float64)
    if hit_id < 0:
        return np.inf
    if hit_id < S:
        s = scene.spheresData[hit_id]
        v = o - s["Center"].astype(np.float64)
        b, c = 2 * d.dot(v), v.dot(v) - float(s["Radius"]) ** 2
        disc = max(b * b - 4 * c, 0.0)
        t2 = (-b - np.sqrt(disc)) / 2
        return t2 if t2 >= 0 else (-b + np.sqrt(disc)) / 2
    if hit_id < S + T:
        t = scene.trianglesData[hit_id - S]
        p, n = t["PointA"].astype(np.float64), t["Normal"].astype(np.float64)
    else:
        pl = scene.planesData[hit_id - S - T]
        p, n = pl["Position"].astype(np.float64), pl["Normal"].astype(np.float64)
    return float((p - o).dot(n) / d.dot(n))
