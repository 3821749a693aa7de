"""rtree.h (fast, exact-equivalent) against the oracle's literal restatement of RTree.cs."""
import numpy as np
import pytest

import oracle_lib as O
from raytracing_amd import host
from raytracing_amd._native import SPHERE_DT, TRIANGLE_DT


def _oracle_accel(spheres, triangles):
    o = O.scene_from_structs(spheres.view(O.SPHERE_DT), np.zeros(0, O.PLANE_DT), triangles.view(O.TRI_DT),
                             np.zeros(0, np.float32))
    return o


@pytest.mark.parametrize("n_tri,n_sph", [(0, 0), (1, 0), (0, 1), (5, 0), (6, 0), (37, 3), (500, 20), (6000, 60),
                                         (30000, 300)])
def test_soup_trees_bit_identical(n_tri, n_sph):
    if n_tri + n_sph == 0:
        assert len(host.accel_from_structs(np.zeros(0, SPHERE_DT), np.zeros(0, TRIANGLE_DT))) == 0
        assert len(_oracle_accel(np.zeros(0, SPHERE_DT), np.zeros(0, TRIANGLE_DT)).accel) == 0
        return
    sc = host.make_scene("Soup", params=[n_tri, n_sph, 2, 1234 + n_tri])
    o = _oracle_accel(sc.spheresData, sc.trianglesData)
    assert o.error == ""
    assert o.accel.tobytes() == sc.AccelerationStructureData.tobytes()
    assert host.accel_from_structs(sc.spheresData, sc.trianglesData).tobytes() == o.accel.tobytes()


def test_clustered_and_duplicate_boxes():
    # many identical / nested boxes stress the FullyContains routing and repeated splits
    g = np.random.default_rng(3)
    tri = np.zeros(400, TRIANGLE_DT)
    base = g.uniform(-1, 1, (20, 3)).astype(np.float32)
    for i in range(400):
        c = base[i % 20]
        s = np.float32(0.1 + 0.05 * (i % 7))
        tri[i]["PointA"], tri[i]["PointB"], tri[i]["PointC"] = c, c + [s, 0, s / 2], c + [0, s, s]
    o = _oracle_accel(np.zeros(0, SPHERE_DT), tri)
    assert o.error == ""
    assert host.accel_from_structs(np.zeros(0, SPHERE_DT), tri).tobytes() == o.accel.tobytes()


def test_crash_corner_is_reported_by_both():
    # five coplanar axis-aligned triangles fill one leaf: every pair volume is 0 and RTree.cs:303
    # indexes primitivePointers[-1]; both restatements report it instead of crashing
    tri = np.zeros(6, TRIANGLE_DT)
    for i in range(6):
        tri[i]["PointA"], tri[i]["PointB"], tri[i]["PointC"] = [i, 0, 0], [i + 1, 0, 0], [i, 0, 1]
    o = _oracle_accel(np.zeros(0, SPHERE_DT), tri)
    assert "IndexOutOfRange" in o.error
    with pytest.raises(RuntimeError, match="IndexOutOfRange"):
        host.accel_from_structs(np.zeros(0, SPHERE_DT), tri)


def test_untilted_table_hits_the_crash_corner(ref_assets):
    # documents why TeapotTable tilts the table (scenes.h)
    raw = O.import_model(f"{ref_assets}/table.KarelModel", 0.15, (0, 0, 5.6))
    tri = np.zeros(len(raw), TRIANGLE_DT)
    for i, r in enumerate(raw):
        t = O.make_triangle(r)
        tri[i]["PointA"], tri[i]["PointB"], tri[i]["PointC"], tri[i]["Normal"] = t
    assert "IndexOutOfRange" in _oracle_accel(np.zeros(0, SPHERE_DT), tri).error
