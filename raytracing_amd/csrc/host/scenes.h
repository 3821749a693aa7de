// scenes.h -- IScene (Scenes/IScene.cs:10-38), TestScene1 (Scenes/TestScene1.cs:8-78) and the
// builder-defined scenes of the benchmark configs (TeapotScene, Soup, TeapotTable),
// plus the packing of RayTracer.SendPrimitivesToShader (RayTracer.cs:957-1005).
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "obj_import.h"
#include "rtree.h"
#include "scene_elements.h"

namespace rth {

// ---- Scenes/IScene.cs:10-38 ----
class IScene {
  public:
    std::vector<Plane> PlanePrimitives;
    std::vector<Sphere> SpherePrimitives;
    std::vector<Triangle> TrianglePrimitives;
    std::vector<PointLight> PointLights;
    rth::Camera Camera{Vector3(0, 0, 0), Vector3(0, 0, 1), Vector3(1, 0, 0), 1, 1.6f, 0.9f};
    std::vector<float> AccelerationStructureData;
    virtual ~IScene() = default;
    virtual void Tick() {}

    // IScene.cs:22-38: spheres 0..S-1 first, then triangles S..S+T-1; planes are not inserted
    void ActivateAccelerationStructure() {
        std::vector<BoundingBox> boxes;
        boxes.reserve(SpherePrimitives.size() + TrianglePrimitives.size());
        for (const Sphere& s : SpherePrimitives) boxes.push_back(s.BoundingBox());
        for (const Triangle& t : TrianglePrimitives) boxes.push_back(t.BoundingBox());
        RTree AccelerationStructure(std::move(boxes));
        for (size_t i = 0; i < SpherePrimitives.size(); i++) AccelerationStructure.AddPrimitive((int)i);
        for (size_t i = 0; i < TrianglePrimitives.size(); i++)
            AccelerationStructure.AddPrimitive((int)(i + SpherePrimitives.size()));
        AccelerationStructureData = AccelerationStructure.TurnIntoFloatArray();
    }

  protected:
    void Concat(std::vector<Triangle> more) {
        TrianglePrimitives.insert(TrianglePrimitives.end(), more.begin(), more.end());
    }
};

// ---- the buffers RayTracer.SendPrimitivesToShader builds (RayTracer.cs:957-1005) ----
struct PackedScene {
    std::vector<SphereStruct> spheresData;
    std::vector<PlaneStruct> planesData;
    std::vector<TriangleStruct> trianglesData;
    std::vector<float> lightsData;
    std::vector<float> accelerationStructureData;
    float cameraData[14];

    static void Mat(const Material& m, float* d, uint32_t* pure, float* s, float* width) {
        d[0] = m.DiffuseColor.R, d[1] = m.DiffuseColor.G, d[2] = m.DiffuseColor.B;
        s[0] = m.SpecularColor.R, s[1] = m.SpecularColor.G, s[2] = m.SpecularColor.B;
        *pure = m.IsPureSpecular ? 1u : 0u;
        *width = m.SpecularWidth();
    }
    static void Put(float* dst, Vector3 v) { dst[0] = v.X, dst[1] = v.Y, dst[2] = v.Z; }

    // maxLights < 0 lifts the reference's cap of 50 (RayTracer.cs:960, 991-994)
    void Pack(const IScene& scene, float screenWidth, float screenHeight, int maxLights = -1) {
        planesData.assign(scene.PlanePrimitives.size(), PlaneStruct{});
        spheresData.assign(scene.SpherePrimitives.size(), SphereStruct{});
        trianglesData.assign(scene.TrianglePrimitives.size(), TriangleStruct{});
        for (size_t i = 0; i < planesData.size(); i++) {
            const Plane& p = scene.PlanePrimitives[i];
            PlaneStruct& o = planesData[i];
            Put(o.Position, p.Center), Put(o.Normal, p.Normal);
            Mat(p.Material_, o.DiffuseColor, &o.IsPureSpecular, o.SpecularColor, &o.Specularity);
        }
        for (size_t i = 0; i < spheresData.size(); i++) {
            const Sphere& s = scene.SpherePrimitives[i];
            SphereStruct& o = spheresData[i];
            Put(o.Center, s.Center), o.Radius = s.Radius;
            Mat(s.Material_, o.DiffuseColor, &o.IsPureSpecular, o.SpecularColor, &o.Specularity);
        }
        for (size_t i = 0; i < trianglesData.size(); i++) {
            const Triangle& t = scene.TrianglePrimitives[i];
            TriangleStruct& o = trianglesData[i];
            Put(o.PointA, t.PointA()), Put(o.PointB, t.PointB()), Put(o.PointC, t.PointC()), Put(o.Normal, t.Normal);
            Mat(t.Material_, o.DiffuseColor, &o.IsPureSpecular, o.SpecularColor, &o.Specularity);
        }
        size_t lightsAmount = scene.PointLights.size();
        if (maxLights >= 0 && lightsAmount > (size_t)maxLights) lightsAmount = (size_t)maxLights;
        lightsData.assign(lightsAmount * 6, 0.0f);
        for (size_t i = 0; i < lightsAmount; i++) {
            const PointLight& l = scene.PointLights[i];
            float* o = &lightsData[i * 6];
            o[0] = l.Position.X, o[1] = l.Position.Y, o[2] = l.Position.Z;
            o[3] = l.Intensity.R, o[4] = l.Intensity.G, o[5] = l.Intensity.B;
        }
        accelerationStructureData = scene.AccelerationStructureData;
        scene.Camera.PackCameraData(screenWidth, screenHeight, cameraData);
    }
};

inline Material MirrorFloorMaterial() { return Material(Color4::DimGray(), Color4::LightGray(), true, 1.0f); }

inline void AddTestScene1Lights(IScene& s) {
    // TestScene1.cs:42-50 (float-literal Color4 constructor: components used as given)
    s.PointLights = {
        PointLight(Vector3(-5.0f, 10.0f, 0.0f), Color4(70, 70, 70, 1.0f)),
        PointLight(Vector3(0.0f, 3.0f, 0.0f), Color4(0, 4, 4, 1.0f)),
        PointLight(Vector3(0.5f, 2.5f, -3.0f), Color4(3.0f, 0, 6, 1.0f)),
        PointLight(Vector3(8.0f, 5.0f, 2.0f), Color4(10.0f, 5.0f, 0.0f, 1.0f)),
        PointLight(Vector3(5.0f, 10.0f, 5.0f), Color4(30, 30, 30, 1.0f)),
        PointLight(Vector3(30.0f, 20.0f, 0.0f), Color4(300, 300, 300, 1.0f)),
    };
}

// ---- Scenes/TestScene1.cs:16-56 ----
class TestScene1 : public IScene {
  public:
    TestScene1() {
        Vector3 offset(10, 0, 0);
        Camera = rth::Camera(Vector3(1, 5.0f, -5), Vector3(0.0f, -1.0f, 1.0f), Vector3(1, 0.0f, 0), 1.0f, 1.6f, 0.9f);
        PlanePrimitives = {Plane(Vector3(0, 0, 0), Vector3(0, 1, 0), MirrorFloorMaterial())};
        SpherePrimitives = {
            Sphere(Vector3(-1, 0.5f, 0), 0.5f, Material(Color4::Red(), Color4::Black(), false, 50.0f)),
            Sphere(Vector3(0.0f, 1.0f, 2.0f), 1.0f, Material(Color4::Black(), Color4::Gray(), true, 100.0f)),
            Sphere(Vector3(-1, 0.3f, -1.0f), 0.3f, Material(Color4::Gold(), Color4::LightGray(), false, 50.0f)),
            Sphere(Vector3(-2, 0.8f, -2), 0.8f, Material(Color4::Purple(), Color4::Purple(), false, 15.0f)),
            Sphere(Vector3(-1, 0.5f, 0) + offset, 0.5f, Material(Color4::Red(), Color4::Gray(), false, 50.0f)),
            Sphere(Vector3(0.0f, 1.0f, 2.0f) + offset, 1.0f,
                   Material(Color4::DeepPink(), Color4::LightGray(), false, 50.0f)),
            Sphere(Vector3(-1, 0.3f, -1.0f) + offset, 0.3f,
                   Material(Color4::Yellow(), Color4::LightGray(), false, 100.0f)),
            Sphere(Vector3(-2, 0.8f, -2) + offset, 0.8f, Material(Color4::Black(), Color4::LightGray(), true, 5.0f)),
        };
        Concat(OBJImportHelper::ImportModel(OBJImportHelper::FilePath("cube"), 0.02f, Vector3(3, 0, 1),
                                            Material(Color4::Gold(), Color4::Gold(), true, 1.0f)));
        Concat(OBJImportHelper::ImportModel(OBJImportHelper::FilePath("pyramid"), 0.03f, Vector3(1, 0, 0),
                                            Material(Color4::Black(), Color4::FromBytes(50, 50, 255, 255), true, 1.0f)));
        Concat(OBJImportHelper::ImportModel(OBJImportHelper::FilePath("pyramid"), 0.05f, Vector3(3, 0, -2),
                                            Material(Color4::Turquoise(), Color4::White(), false, 1.0f)));
        Concat(OBJImportHelper::ImportModel(OBJImportHelper::FilePath("teapot"), 0.03f, Vector3(0, 2, 5),
                                            Material(Color4::Beige(), Color4::Gray(), false, 0.01f)));
        AddTestScene1Lights(*this);
        ActivateAccelerationStructure();
    }
};

// ---- config 2 (builder-defined): TestScene1's plane, lights and camera, only the teapot ----
class TeapotScene : public IScene {
  public:
    TeapotScene() {
        Camera = rth::Camera(Vector3(1, 5.0f, -5), Vector3(0.0f, -1.0f, 1.0f), Vector3(1, 0.0f, 0), 1.0f, 1.6f, 0.9f);
        PlanePrimitives = {Plane(Vector3(0, 0, 0), Vector3(0, 1, 0), MirrorFloorMaterial())};
        Concat(OBJImportHelper::ImportModel(OBJImportHelper::FilePath("teapot"), 0.03f, Vector3(0, 2, 5),
                                            Material(Color4::Beige(), Color4::Gray(), false, 0.01f)));
        AddTestScene1Lights(*this);
        ActivateAccelerationStructure();
    }
};

// splitmix64; a float draw is the top 24 bits / 2^24 (exact in binary32)
struct SplitMix64 {
    uint64_t state;
    explicit SplitMix64(uint64_t seed) : state(seed) {}
    uint64_t Next() {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    float NextFloat() { return (float)(Next() >> 40) / 16777216.0f; }
    float Uniform(float a, float b) { return a + (b - a) * NextFloat(); }
};

// ---- config 3 (builder-defined): triangle soup + spheres over a mirror floor ----
// Draw order (fixed; the fixtures depend on it): for every sphere, then for every triangle:
// centre x,y,z; [sphere: radius | triangle: 3 vertices x (dx,dy,dz), redrawn while the triangle
// has zero area]; diffuse r,g,b; pure-specular draw (< 0.2); specular r,g,b; specularity draw.
class SoupScene : public IScene {
  public:
    static constexpr uint64_t DefaultSeed = 0x5EED5EED2024ull;
    SoupScene(int nTriangles, int nSpheres, int nLights = 8, uint64_t seed = DefaultSeed) {
        SplitMix64 rng(seed);
        const float widths[5] = {1.0f, 5.0f, 15.0f, 50.0f, 100.0f};
        auto material = [&]() {
            const float dr = rng.Uniform(0.1f, 1.0f), dg = rng.Uniform(0.1f, 1.0f), db = rng.Uniform(0.1f, 1.0f);
            const bool pure = rng.NextFloat() < 0.2f;
            const float sr = rng.Uniform(0.3f, 0.9f), sg = rng.Uniform(0.3f, 0.9f), sb = rng.Uniform(0.3f, 0.9f);
            int wi = (int)(rng.NextFloat() * 5.0f);
            if (wi > 4) wi = 4;
            return Material(Color4(dr, dg, db, 1.0f), Color4(sr, sg, sb, 1.0f), pure, widths[wi]);
        };
        auto centre = [&]() {
            const float x = rng.Uniform(-40.0f, 40.0f), y = rng.Uniform(0.5f, 30.0f), z = rng.Uniform(-40.0f, 40.0f);
            return Vector3(x, y, z);
        };
        PlanePrimitives = {Plane(Vector3(0, 0, 0), Vector3(0, 1, 0), MirrorFloorMaterial())};
        SpherePrimitives.reserve(nSpheres);
        for (int i = 0; i < nSpheres; i++) {
            const Vector3 c = centre();
            const float r = rng.Uniform(0.05f, 0.4f);
            SpherePrimitives.emplace_back(c, r, material());
        }
        TrianglePrimitives.reserve(nTriangles);
        for (int i = 0; i < nTriangles; i++) {
            const Vector3 c = centre();
            Vector3 p[3];
            for (;;) {
                for (int k = 0; k < 3; k++) {
                    const float dx = rng.Uniform(-0.15f, 0.15f), dy = rng.Uniform(-0.15f, 0.15f),
                                dz = rng.Uniform(-0.15f, 0.15f);
                    p[k] = c + Vector3(dx, dy, dz);
                }
                const float area2 = Vector3::Cross(p[1] - p[0], p[2] - p[0]).Length();
                if (area2 > 0.0f && std::isfinite(area2)) break;
            }
            TrianglePrimitives.emplace_back(p[0], p[1], p[2], material());
        }
        for (int k = 0; k < nLights; k++) {
            const double ang = 2.0 * 3.14159265358979323846 * k / nLights;
            PointLights.emplace_back(Vector3((float)(60.0 * std::cos(ang)), 50.0f, (float)(60.0 * std::sin(ang))),
                                     Color4(600, 600, 600, 1.0f));
        }
        // looking from (0,35,-90) at (0,10,0)
        Camera = rth::Camera(Vector3(0, 35, -90), Vector3(0, -25, 90), Vector3(1, 0, 0), 1.0f, 1.6f, 0.9f);
        ActivateAccelerationStructure();
    }
};

// ---- config 5 (builder-defined): teapot on the table over the mirror floor; orbiting camera ----
class TeapotTableScene : public IScene {
  public:
    static Vector3 LookAt() { return Vector3(0, 5.0f, 5.0f); }
    // frame f of n: the camera circles LookAt() at radius 14, 5 above it, via the value
    // constructor (Camera.cs:92-99)
    static rth::Camera OrbitCamera(int frame, int nFrames) {
        const double ang = 2.0 * 3.14159265358979323846 * frame / nFrames;
        const float s = (float)std::sin(ang), c = (float)std::cos(ang);
        const Vector3 position = LookAt() + Vector3(14.0f * s, 5.0f, -14.0f * c);
        return rth::Camera(position, LookAt() - position, Vector3(c, 0, s), 1.0f, 1.6f, 0.9f);
    }
    TeapotTableScene() {
        Camera = OrbitCamera(0, 256);
        PlanePrimitives = {Plane(Vector3(0, 0, 0), Vector3(0, 1, 0), MirrorFloorMaterial())};
        // table.KarelModel (single-space dialect -> y/z swap): 40 x 40 x 40 units, top at y = 40*scale.
        // Imported as it stands the table crashes the reference's R-tree build: its flat faces put
        // five coplanar axis-aligned triangles into one leaf, every pair volume is 0 and
        // RTree.cs:303 indexes primitivePointers[-1].  Fallback: the table is turned
        // 12 degrees about the vertical through its centre and tipped 1.5 degrees, so that no face
        // is axis-aligned; rotation constants are binary32 products in the order written.
        const Vector3 pivot(0, 0, 5.0f);
        const float cy = 0.97814760f, sy = 0.20791169f;   // cos/sin 12 deg
        const float cx = 0.99965732f, sx = 0.02617695f;   // cos/sin 1.5 deg
        auto tilt = [=](Vector3 v) {
            const Vector3 p = v - pivot;
            const Vector3 q((p.X * cy) + (p.Z * sy), p.Y, (p.Z * cy) - (p.X * sy));   // about Y
            const Vector3 r(q.X, (q.Y * cx) - (q.Z * sx), (q.Z * cx) + (q.Y * sx));   // about X
            return r + pivot;
        };
        Concat(OBJImportHelper::ImportModel(OBJImportHelper::FilePath("table"), 0.15f, Vector3(0, 0.2f, 5.6f),
                                            Material(Color4::FromBytes(60, 40, 30), Color4::LightGray(), true, 1.0f),
                                            tilt));
        // teapot (scale 0.03 spans y -1.2..1.5 around its position) standing on the table top (y = 6.2)
        Concat(OBJImportHelper::ImportModel(OBJImportHelper::FilePath("teapot"), 0.03f, Vector3(0, 7.4f, 5),
                                            Material(Color4::Beige(), Color4::Gray(), false, 0.01f)));
        AddTestScene1Lights(*this);
        ActivateAccelerationStructure();
    }
};

}  // namespace rth
