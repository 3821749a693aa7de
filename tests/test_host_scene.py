"""The C++ host mirror (csrc/host) against the oracle's independent restatement of the same C#."""
import numpy as np
import pytest

import oracle_lib as O
from helpers import crc
from raytracing_amd import host
from raytracing_amd._native import ASSET_DIR


def test_testscene1_buffers_match_golden(scene_of, checksums):
    sc = scene_of("TestScene1")
    g = checksums["testscene1"]
    assert (len(sc.spheresData), len(sc.planesData), len(sc.trianglesData)) == (8, 1, 1047)
    assert crc(sc.spheresData) == g["crc_spheres"]
    assert crc(sc.planesData) == g["crc_planes"]
    assert crc(sc.trianglesData) == g["crc_triangles"]
    assert crc(sc.lightsData) == g["crc_lights"]
    assert crc(sc.cameraData) == g["crc_camera"]
    assert len(sc.AccelerationStructureData) == g["accel_len"]
    assert crc(sc.AccelerationStructureData) == g["crc_accel"]
    assert sc.AccelerationStructureData[:64].tolist() == g["accel_first64"]


def test_testscene1_matches_oracle_builder(ref_assets, scene_of):
    o = O.testscene1(ref_assets)
    sc = scene_of("TestScene1")
    assert o.error == ""
    for a, b in [(o.spheres, sc.spheresData), (o.planes, sc.planesData), (o.triangles, sc.trianglesData),
                 (o.accel, sc.AccelerationStructureData), (o.lights, sc.lightsData), (o.camera, sc.cameraData)]:
        assert a.tobytes() == b.tobytes()


def test_testscene1_values(scene_of):
    sc = scene_of("TestScene1")
    # new Color4(50, 50, 255, 255) binds to the byte overload; the float-literal lights do not
    pyramid = sc.trianglesData[12]
    assert np.allclose(pyramid["SpecularColor"], [50 / 255, 50 / 255, 1.0])
    assert sc.lightsData[3:6].tolist() == [70.0, 70.0, 70.0]
    # Material.SpecularWidth clamps to >= 1 (teapot is built with 0.01, Material.cs:15-19)
    assert sc.trianglesData[-1]["Specularity"] == 1.0
    assert sc.planesData[0]["IsPureSpecular"] == 1 and np.allclose(sc.planesData[0]["Normal"], [0, 1, 0])
    # every branch of the serialised tree is binary, leaves hold <= 5 ids
    acc = sc.AccelerationStructureData
    seen, stack = 0, [0]
    ids = []
    while stack:
        pos = stack.pop()
        cnt = int(acc[pos + 7])
        if acc[pos + 6] == 0:
            assert cnt == 2
            stack += [pos + int(acc[pos + 8 + i]) for i in range(cnt)]
        else:
            assert 1 <= cnt <= 5
            ids += [int(acc[pos + 8 + i]) for i in range(cnt)]
        seen += 1
    assert sorted(ids) == list(range(8 + 1047))


def test_mesh_assets_equal_text_assets(ref_assets):
    for name, scale, pos in [("cube", 0.02, (3, 0, 1)), ("pyramid", 0.05, (3, 0, -2)), ("table", 0.15, (0, 0, 5)),
                             ("teapot", 0.03, (0, 2, 5))]:
        a = host.import_model(f"{ref_assets}/{name}.KarelModel", scale, pos)
        b = host.import_model(f"{ASSET_DIR}/{name}.mesh", scale, pos)
        assert a.tobytes() == b.tobytes()
        raw = O.import_model(f"{ref_assets}/{name}.KarelModel", scale, pos)
        want = np.stack([O.make_triangle(r) for r in raw])
        assert a.tobytes() == want.tobytes()


def test_mesh_asset_counts():
    expect = {"cube": 12, "pyramid": 6, "table": 156, "teapot": 1023}
    for name, n in expect.items():
        assert len(host.import_model(f"{ASSET_DIR}/{name}.mesh", 1.0, (0, 0, 0))) == n


def test_import_errors(tmp_path):
    with pytest.raises(RuntimeError, match="FileNotFound"):
        host.import_model(tmp_path / "nope.KarelModel", 1.0, (0, 0, 0))
    p = tmp_path / "bad.KarelModel"
    p.write_text("v 0 0 0\nv 1 0 0\nf 1 2 3\nf 1 2 3\n")  # face index beyond the vertex list
    with pytest.raises(RuntimeError, match="OutOfRange"):
        host.import_model(p, 1.0, (0, 0, 0))


def test_camera_pack_matches_oracle():
    g = np.random.default_rng(5)
    for _ in range(50):
        pos, view = g.uniform(-10, 10, 3), g.normal(size=3)
        right = np.cross(view, g.normal(size=3))
        a = host.camera_pack(pos, view, right, 1.3, 1.6, 0.9, 640, 400)
        b = O.camera_pack(pos, view, right, 1.3, 1.6, 0.9, 640, 400)
        assert a.tobytes() == b.tobytes()


def test_orbit_camera_is_orthonormal():
    for f in (0, 17, 64, 200):
        cam = host.camera_orbit(f, 256, 320, 180)
        right, up = cam[6:9] - cam[3:6], cam[9:12] - cam[3:6]
        assert abs(np.dot(right, up)) < 1e-5
        assert np.isclose(np.linalg.norm(right), 1.6, atol=1e-5) and np.isclose(np.linalg.norm(up), 0.9, atol=1e-5)


def test_unknown_scene():
    with pytest.raises(RuntimeError, match="unknown scene"):
        host.make_scene("NoSuchScene")
