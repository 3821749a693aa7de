"""The C-ABI library loads and exports every symbol include/rt_mi355x.h declares (no compute)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracing_amd import _native as N
from raytracing_amd.raytracer import RtError, check_accel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 14
    lib = C.CDLL(N.DEVICE_LIB)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rt_mi355x.h but not exported"
    assert set(names) == set(N.DEVICE_API), "ctypes table and header disagree"


def test_struct_layouts():
    assert C.sizeof(N.rt_stats) == 16 * 8
    assert C.sizeof(N.rt_scene_info) == 16 * 8
    assert (N.SPHERE_DT.itemsize, N.PLANE_DT.itemsize, N.TRIANGLE_DT.itemsize) == (48, 64, 96)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(N, "_device", None)
    monkeypatch.setattr(N, "DEVICE_LIB", str(tmp_path / "librt_mi355x.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        N.device_lib()


@pytest.mark.parametrize("h,band,world", [(1080, 16, 8), (720, 16, 3), (77, 8, 4), (5, 16, 2), (2160, 32, 8),
                                          (90, 7, 5), (16, 16, 1)])
def test_band_partition(h, band, world):
    lib = N.device_lib()
    rows = [lib.rt_local_rows(h, band, r, world) for r in range(world)]
    padded = lib.rt_padded_rows(h, band, world)
    assert sum(rows) == h and max(rows) <= padded and padded % band == 0
    owner = np.full(h, -1)
    for r in range(world):
        for lr in range(rows[r]):
            y = ((lr // band) * world + r) * band + lr % band
            assert 0 <= y < h and owner[y] == -1
            owner[y] = r
    assert (owner >= 0).all()


def test_check_accel_accepts_reference_trees(scene_of):
    for name, params in [("TestScene1", None), ("TeapotTable", None), ("Soup", [3000, 30, 4])]:
        sc = scene_of(name, params)
        info = check_accel(sc.AccelerationStructureData, len(sc.spheresData), len(sc.trianglesData))
        assert info["n_tri_slots"] == len(sc.trianglesData) and info["n_sphere_slots"] == len(sc.spheresData)
        assert info["n_nodes"] >= 1 and info["max_depth"] >= 1
    assert check_accel(np.zeros(0, np.float32), 0, 0)["n_nodes"] == 0


def test_retreed_bvh_sizes(scene_of):
    """rt_bvh.h through the host-only dry run: one BVH leaf per non-empty reference leaf piece, a binary
    tree over them plus the root holder, depth within the LDS stack bound; not built for trees whose
    boxes are not nested."""
    for name, params in [("TestScene1", None), ("TeapotTable", None), ("Soup", [30000, 300, 4])]:
        sc = scene_of(name, params)
        acc = sc.AccelerationStructureData
        info = check_accel(acc, len(sc.spheresData), len(sc.trianglesData))
        assert info["flags"] & 1
        # count the non-empty leaf pieces of the serialised tree (reference leaves hold <= 5 ids of one kind
        # unless they mix spheres and triangles)
        S = len(sc.spheresData)
        pieces, pos_stack = 0, [0]
        while pos_stack:
            pos = pos_stack.pop()
            cnt = int(acc[pos + 7])
            if acc[pos + 6] == 0:
                pos_stack += [pos + int(acc[pos + 8 + i]) for i in range(cnt)]
            else:
                ids = acc[pos + 8:pos + 8 + cnt]
                pieces += int((ids < S).any()) + int((ids >= S).any())
        assert info["bvh_nodes"] == pieces  # (pieces - 1) inner nodes + the root holder
        assert 1 <= info["bvh_levels"] <= 32
    sc = scene_of("TestScene1")
    acc = sc.AccelerationStructureData.copy()
    acc[3:6] -= 1.0  # root box no longer contains its children
    info = check_accel(acc, len(sc.spheresData), len(sc.trianglesData))
    assert info["flags"] == 0 and info["bvh_nodes"] == 0


def test_check_accel_rejects_malformed(scene_of):
    sc = scene_of("TestScene1")
    good = sc.AccelerationStructureData
    S, T = len(sc.spheresData), len(sc.trianglesData)

    def bad(mutate, match):
        a = good.copy()
        mutate(a)
        with pytest.raises(RtError, match=match) as e:
            check_accel(a, S, T)
        assert e.value.code == -2

    bad(lambda a: a.__setitem__(7, 3.5), "bad count")                    # fractional count
    bad(lambda a: a.__setitem__(7, -1), "bad count")
    bad(lambda a: a.__setitem__(7, 1e9), "bad count")                    # count past the end
    bad(lambda a: a.__setitem__(8, 0), "bad child offset")               # child = self: a cycle
    bad(lambda a: a.__setitem__(8, 3), "bad child offset")               # child inside own header
    bad(lambda a: a.__setitem__(9, float(len(good))), "bad child offset")  # child past the end
    bad(lambda a: a.__setitem__(9, np.nan), "bad child offset")
    with pytest.raises(RtError):
        check_accel(good[:5], S, T)                                      # shorter than a header
    with pytest.raises(RtError, match="bad primitive id"):
        check_accel(good, S, T - 400)                                    # ids beyond the primitive arrays
    # both root pointers name the same child: the walk would visit more nodes than the array can
    # hold, which is how shared children and cycles through forward pointers are caught
    a = good.copy()
    a[9] = a[8]
    with pytest.raises(RtError, match="shared or cyclic"):
        check_accel(a, S, T)
    # sharing the SMALL child stays within the budget: a valid walk, exactly what the shader would do
    a = good.copy()
    a[8] = a[9]
    check_accel(a, S, T)
