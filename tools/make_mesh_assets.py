"""Convert the reference's text models into the pre-parsed ".mesh" form the scenes load where
the reference tree is absent (the GPU box).  A .mesh holds only what OBJImportHelper.ImportModel
processes: the "v " coordinates and "f" index triples of every line but the last
(OBJImportHelper.cs:24-25, 71), as data.  Run here, where /root/reference exists:

    python tools/make_mesh_assets.py [/root/reference/assets]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import host  # noqa: E402
from raytracing_amd._native import ASSET_DIR  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/assets"
os.makedirs(ASSET_DIR, exist_ok=True)
for name in ("cube", "pyramid", "table", "teapot"):
    dst = os.path.join(ASSET_DIR, name + ".mesh")
    host.convert_model(os.path.join(src, name + ".KarelModel"), dst)
    print(name, os.path.getsize(dst), "bytes")
