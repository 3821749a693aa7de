// host_capi.cpp -- C entry points of librt_host.so: lets the Python harness (tests, bench)
// drive the C++ host mirror (scenes.h) and fetch the packed buffers the C# host would hand to
// rt_upload_scene / rt_render.  CPU only; no HIP dependency.
#include <cstring>
#include <exception>
#include <string>

#include "scenes.h"

using namespace rth;

struct rth_scene {
    PackedScene packed;
    std::string error;
};

namespace {
thread_local std::string g_error;

template <class F>
rth_scene* guarded(F&& build) {
    auto* s = new rth_scene();
    try {
        build(*s);
    } catch (const std::exception& e) {
        s->error = e.what();
    }
    return s;
}
}  // namespace

extern "C" {

void rth_set_asset_directory(const char* dir) { OBJImportHelper::AssetDirectory() = dir ? dir : "."; }

// name: "TestScene1" | "TeapotScene" | "TeapotTable" | "Soup" (params: n_triangles, n_spheres,
// n_lights, seed; missing/negative = defaults 1e6, 1e4, 8, SoupScene::DefaultSeed).
// screen_w/h go to camera[12..13] (RayTracer.cs:42 fixes them at 1280x720).
rth_scene* rth_scene_create(const char* name, float screen_w, float screen_h, const double* params, int n_params) {
    const std::string n = name ? name : "";
    return guarded([&](rth_scene& s) {
        auto P = [&](int i, double dflt) { return (i < n_params && params && params[i] >= 0) ? params[i] : dflt; };
        if (n == "TestScene1") {
            TestScene1 sc;
            s.packed.Pack(sc, screen_w, screen_h);
        } else if (n == "TeapotScene") {
            TeapotScene sc;
            s.packed.Pack(sc, screen_w, screen_h);
        } else if (n == "TeapotTable") {
            TeapotTableScene sc;
            s.packed.Pack(sc, screen_w, screen_h);
        } else if (n == "Soup") {
            SoupScene sc((int)P(0, 1e6), (int)P(1, 1e4), (int)P(2, 8), (uint64_t)P(3, (double)SoupScene::DefaultSeed));
            s.packed.Pack(sc, screen_w, screen_h);
        } else {
            throw std::runtime_error("unknown scene: " + n);
        }
    });
}

// IScene.ActivateAccelerationStructure on caller-packed structs (tests compare with the oracle)
rth_scene* rth_scene_from_structs(const void* spheres, int ns, const void* tris, int nt) {
    return guarded([&](rth_scene& s) {
        const auto* sp = (const SphereStruct*)spheres;
        const auto* tr = (const TriangleStruct*)tris;
        std::vector<BoundingBox> boxes;
        boxes.reserve((size_t)ns + nt);
        for (int i = 0; i < ns; i++) {
            const float r = sp[i].Radius;
            boxes.emplace_back(Vector3(sp[i].Center[0] - r, sp[i].Center[1] - r, sp[i].Center[2] - r),
                               Vector3(sp[i].Center[0] + r, sp[i].Center[1] + r, sp[i].Center[2] + r));
        }
        for (int i = 0; i < nt; i++) {
            const float *a = tr[i].PointA, *b = tr[i].PointB, *c = tr[i].PointC;
            boxes.emplace_back(
                Vector3(MathF_Min(a[0], MathF_Min(b[0], c[0])), MathF_Min(a[1], MathF_Min(b[1], c[1])),
                        MathF_Min(a[2], MathF_Min(b[2], c[2]))),
                Vector3(MathF_Max(a[0], MathF_Max(b[0], c[0])), MathF_Max(a[1], MathF_Max(b[1], c[1])),
                        MathF_Max(a[2], MathF_Max(b[2], c[2]))));
        }
        RTree tree(std::move(boxes));
        for (int i = 0; i < ns + nt; i++) tree.AddPrimitive(i);
        s.packed.accelerationStructureData = tree.TurnIntoFloatArray();
        if (ns) s.packed.spheresData.assign(sp, sp + ns);
        if (nt) s.packed.trianglesData.assign(tr, tr + nt);
        std::memset(s.packed.cameraData, 0, sizeof s.packed.cameraData);
    });
}

// which: 0 spheres(48 B) 1 planes(64 B) 2 triangles(96 B) 3 accel(float) 4 lights(float) 5 camera(14 floats)
int rth_scene_get(const rth_scene* s, int which, const void** ptr, int64_t* count) {
    if (!s || !ptr || !count) return -1;
    const PackedScene& p = s->packed;
    switch (which) {
        case 0: *ptr = p.spheresData.data(); *count = (int64_t)p.spheresData.size(); return 0;
        case 1: *ptr = p.planesData.data(); *count = (int64_t)p.planesData.size(); return 0;
        case 2: *ptr = p.trianglesData.data(); *count = (int64_t)p.trianglesData.size(); return 0;
        case 3: *ptr = p.accelerationStructureData.data(); *count = (int64_t)p.accelerationStructureData.size(); return 0;
        case 4: *ptr = p.lightsData.data(); *count = (int64_t)p.lightsData.size(); return 0;
        case 5: *ptr = p.cameraData; *count = 14; return 0;
    }
    return -1;
}
const char* rth_scene_error(const rth_scene* s) { return s ? s->error.c_str() : "null scene"; }
void rth_scene_free(rth_scene* s) { delete s; }

// Camera value constructor + RayTracer.cs:947-954 packing
void rth_camera_pack(const float* pos3, const float* view3, const float* right3, float dist, float width,
                     float height, float screen_w, float screen_h, float* out14) {
    Camera cam(Vector3(pos3[0], pos3[1], pos3[2]), Vector3(view3[0], view3[1], view3[2]),
               Vector3(right3[0], right3[1], right3[2]), dist, width, height);
    cam.PackCameraData(screen_w, screen_h, out14);
}
// TeapotTable orbit (config 5)
void rth_camera_orbit(int frame, int n_frames, float screen_w, float screen_h, float* out14) {
    TeapotTableScene::OrbitCamera(frame, n_frames).PackCameraData(screen_w, screen_h, out14);
}

// OBJImportHelper.ImportModel: returns the triangle count and writes up to cap triangles as the
// 12 floats PointA, PointB, PointC, Normal; -1 on error (rth_last_error()).
int rth_import_model(const char* path, float scale, float px, float py, float pz, float* out12, int cap) {
    try {
        const auto tris = OBJImportHelper::ImportModel(path, scale, Vector3(px, py, pz), Material());
        for (int i = 0; i < (int)tris.size() && i < cap; i++) {
            const Vector3 v[4] = {tris[i].PointA(), tris[i].PointB(), tris[i].PointC(), tris[i].Normal};
            for (int k = 0; k < 4; k++)
                out12[12 * i + 3 * k] = v[k].X, out12[12 * i + 3 * k + 1] = v[k].Y, out12[12 * i + 3 * k + 2] = v[k].Z;
        }
        return (int)tris.size();
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}
// text model -> pre-parsed ".mesh" (tools/make_mesh_assets.py)
int rth_convert_model(const char* text_path, const char* mesh_path) {
    try {
        OBJImportHelper::WriteMesh(OBJImportHelper::ParseText(text_path), mesh_path);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}
const char* rth_last_error() { return g_error.c_str(); }

}  // extern "C"
