// scene_elements.h -- C++ mirror of the reference's host-side scene API (same names, same
// argument meaning): Vector3/Color4 (the parts of OpenTK.Mathematics the path uses), Material,
// BoundingBox, IPrimitive, Sphere, Plane, Triangle, PointLight, Camera and the three buffer
// structs.  The reference is C# (.NET 6, not available here), so the host side above the
// C ABI is written in C++; INTEGRATION.md shows the P/Invoke edits for the real C# host.
//
// All arithmetic is strict binary32 in the order the C# expressions are written
// (-ffp-contract=off).  Citations are file:line in the reference tree.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace rth {

// MathF.Min / MathF.Max (.NET 6): NaN-propagating, -0 < +0
inline float MathF_Min(float a, float b) {
    if (a < b) return a;
    if (a != a) return a;
    if (a == b) return std::signbit(a) ? a : b;
    return b;
}
inline float MathF_Max(float a, float b) {
    if (a > b) return a;
    if (a != a) return a;
    if (a == b) return std::signbit(a) ? b : a;
    return b;
}

// ---- OpenTK.Mathematics.Vector3 (OpenTK 4.8.2; source not in the reference tree: the
// operator definitions below are the published ones and are "parity unpinned") ----
struct Vector3 {
    float X = 0, Y = 0, Z = 0;
    Vector3() = default;
    Vector3(float x, float y, float z) : X(x), Y(y), Z(z) {}
    float Length() const { return sqrtf((X * X) + (Y * Y) + (Z * Z)); }
    Vector3 Normalized() const {
        float scale = 1.0f / Length();
        return Vector3(X * scale, Y * scale, Z * scale);
    }
    static Vector3 Cross(const Vector3& l, const Vector3& r) {
        return Vector3((l.Y * r.Z) - (l.Z * r.Y), (l.Z * r.X) - (l.X * r.Z), (l.X * r.Y) - (l.Y * r.X));
    }
    static float Dot(const Vector3& l, const Vector3& r) { return (l.X * r.X) + (l.Y * r.Y) + (l.Z * r.Z); }
};
inline Vector3 operator+(Vector3 a, Vector3 b) { return Vector3(a.X + b.X, a.Y + b.Y, a.Z + b.Z); }
inline Vector3 operator-(Vector3 a, Vector3 b) { return Vector3(a.X - b.X, a.Y - b.Y, a.Z - b.Z); }
inline Vector3 operator-(Vector3 a) { return Vector3(-a.X, -a.Y, -a.Z); }
inline Vector3 operator*(Vector3 a, float s) { return Vector3(a.X * s, a.Y * s, a.Z * s); }
inline Vector3 operator*(float s, Vector3 a) { return Vector3(a.X * s, a.Y * s, a.Z * s); }
inline Vector3 operator/(Vector3 a, float s) { return Vector3(a.X / s, a.Y / s, a.Z / s); }

// ---- OpenTK.Mathematics.Color4 ----
struct Color4 {
    float R = 0, G = 0, B = 0, A = 1;
    Color4() = default;
    // float overload (picked when any argument is a float literal, TestScene1.cs:44-49)
    Color4(float r, float g, float b, float a) : R(r), G(g), B(b), A(a) {}
    // byte overload (picked when all four arguments are int literals, TestScene1.cs:39)
    static Color4 FromBytes(int r, int g, int b, int a = 255) {
        return Color4(r / 255.0f, g / 255.0f, b / 255.0f, a / 255.0f);
    }
    // named colours used by the scenes: the standard .NET/X11 byte triples
    static Color4 DimGray() { return FromBytes(105, 105, 105); }
    static Color4 LightGray() { return FromBytes(211, 211, 211); }
    static Color4 Red() { return FromBytes(255, 0, 0); }
    static Color4 Black() { return FromBytes(0, 0, 0); }
    static Color4 Gray() { return FromBytes(128, 128, 128); }
    static Color4 Gold() { return FromBytes(255, 215, 0); }
    static Color4 Purple() { return FromBytes(128, 0, 128); }
    static Color4 DeepPink() { return FromBytes(255, 20, 147); }
    static Color4 Yellow() { return FromBytes(255, 255, 0); }
    static Color4 Turquoise() { return FromBytes(64, 224, 208); }
    static Color4 White() { return FromBytes(255, 255, 255); }
    static Color4 Beige() { return FromBytes(245, 245, 220); }
};

// ---- Material.cs:9-43 ----
class Material {
    float specularWidth = 1;

  public:
    Color4 DiffuseColor, SpecularColor;
    bool IsPureSpecular = false;
    float SpecularWidth() const { return specularWidth; }
    void SetSpecularWidth(float v) { specularWidth = MathF_Max(v, 1); }  // Material.cs:15-19
    Material() = default;
    Material(Color4 diffuseColor, Color4 specularColor, bool isPureSpecular, float width)
        : DiffuseColor(diffuseColor), SpecularColor(specularColor), IsPureSpecular(isPureSpecular) {
        SetSpecularWidth(width);
    }
    explicit Material(Color4 diffuseColor) : DiffuseColor(diffuseColor), SpecularColor(Color4::Black()) {
        SetSpecularWidth(1);
    }
    // Material.cs:33-42 ("metal" constructor)
    Material(Color4 diffuseColor, bool isMetal, float specularity, bool isPureSpecular, float width)
        : DiffuseColor(diffuseColor), IsPureSpecular(isPureSpecular) {
        if (isMetal)
            SpecularColor =
                Color4(diffuseColor.R * specularity, diffuseColor.G * specularity, diffuseColor.B * specularity, 1.0f);
        else
            SpecularColor = Color4(specularity, specularity, specularity, 1.0f);
        SetSpecularWidth(width);
    }
};

// ---- BoundingBox.cs:6-102 ----
struct BoundingBox {
    Vector3 MinimumValues, MaximumValues;
    BoundingBox() = default;
    BoundingBox(Vector3 mn, Vector3 mx) : MinimumValues(mn), MaximumValues(mx) {}
    float GetSize() const {
        float x = MaximumValues.X - MinimumValues.X;
        float y = MaximumValues.Y - MinimumValues.Y;
        float z = MaximumValues.Z - MinimumValues.Z;
        return x * y * z;
    }
    bool FullyContains(const BoundingBox& c) const {
        return !(c.MinimumValues.X < MinimumValues.X) && !(c.MinimumValues.Y < MinimumValues.Y) &&
               !(c.MinimumValues.Z < MinimumValues.Z) && !(c.MaximumValues.X > MaximumValues.X) &&
               !(c.MaximumValues.Y > MaximumValues.Y) && !(c.MaximumValues.Z > MaximumValues.Z);
    }
    BoundingBox CalculatePotentialBoundingBox(BoundingBox inc) const {
        inc.MinimumValues.X = MathF_Min(inc.MinimumValues.X, MinimumValues.X);
        inc.MinimumValues.Y = MathF_Min(inc.MinimumValues.Y, MinimumValues.Y);
        inc.MinimumValues.Z = MathF_Min(inc.MinimumValues.Z, MinimumValues.Z);
        inc.MaximumValues.X = MathF_Max(inc.MaximumValues.X, MaximumValues.X);
        inc.MaximumValues.Y = MathF_Max(inc.MaximumValues.Y, MaximumValues.Y);
        inc.MaximumValues.Z = MathF_Max(inc.MaximumValues.Z, MaximumValues.Z);
        return inc;
    }
    float CalculatePotentialBoundingBoxSize(const BoundingBox& inc) const {
        return CalculatePotentialBoundingBox(inc).GetSize();
    }
    bool Overlap(const BoundingBox& b) const {
        return !(b.MinimumValues.X > MaximumValues.X) && !(b.MinimumValues.Y > MaximumValues.Y) &&
               !(b.MinimumValues.Z > MaximumValues.Z) && !(b.MaximumValues.X < MinimumValues.X) &&
               !(b.MaximumValues.Y < MinimumValues.Y) && !(b.MaximumValues.Z < MinimumValues.Z);
    }
    float OverlapSize(const BoundingBox& b) const {
        if (!Overlap(b)) return 0;
        float minX = MathF_Max(MinimumValues.X, b.MinimumValues.X);
        float minY = MathF_Max(MinimumValues.Y, b.MinimumValues.Y);
        float minZ = MathF_Max(MinimumValues.Z, b.MinimumValues.Z);
        float maxX = MathF_Min(MaximumValues.X, b.MaximumValues.X);
        float maxY = MathF_Min(MaximumValues.Y, b.MaximumValues.Y);
        float maxZ = MathF_Min(MaximumValues.Z, b.MaximumValues.Z);
        return (maxX - minX) * (maxY - minY) * (maxZ - minZ);
    }
};

// ---- IPrimitive.cs ----  (RayIntersect is dead code in the reference -- nothing calls it,
// SURVEY 8a-14 -- and the ray caster lives on the GPU, so only the data side is mirrored)
class IPrimitive {
  public:
    virtual ~IPrimitive() = default;
    virtual rth::BoundingBox BoundingBox() const = 0;
    Material Material_;
};

// ---- Sphere.cs:7-51 ----
class Sphere : public IPrimitive {
  public:
    Vector3 Center;
    float Radius;
    Sphere(Vector3 center, float radius, Material material) : Center(center), Radius(radius) { Material_ = material; }
    rth::BoundingBox BoundingBox() const override {
        return rth::BoundingBox(Vector3(Center.X - Radius, Center.Y - Radius, Center.Z - Radius),
                                Vector3(Center.X + Radius, Center.Y + Radius, Center.Z + Radius));
    }
};

// ---- Plane.cs:7-38 ----
class Plane : public IPrimitive {
  public:
    Vector3 Center, Normal;
    Plane(Vector3 center, Vector3 normal, Material material) : Center(center), Normal(normal.Normalized()) {
        Material_ = material;
    }
    rth::BoundingBox BoundingBox() const override { return rth::BoundingBox(Vector3(0, 0, 0), Vector3(0, 0, 0)); }
};

// ---- Triangle.cs:7-81 ----
class Triangle : public IPrimitive {
  public:
    Vector3 Center, Normal, VectorA, VectorB, VectorC;
    Vector3 PointA() const { return Center + VectorA; }
    Vector3 PointB() const { return Center + VectorB; }
    Vector3 PointC() const { return Center + VectorC; }
    Triangle(Vector3 pointA, Vector3 pointB, Vector3 pointC, Material material) {
        Center = (pointA + pointB + pointC) / 3;
        VectorA = pointA - Center;
        VectorB = pointB - Center;
        VectorC = pointC - Center;
        Vector3 cross = Vector3::Cross(pointB - pointA, pointC - pointA);
        Normal = cross / cross.Length();
        Material_ = material;
    }
    rth::BoundingBox BoundingBox() const override {
        const Vector3 a = PointA(), b = PointB(), c = PointC();
        return rth::BoundingBox(
            Vector3(MathF_Min(a.X, MathF_Min(b.X, c.X)), MathF_Min(a.Y, MathF_Min(b.Y, c.Y)),
                    MathF_Min(a.Z, MathF_Min(b.Z, c.Z))),
            Vector3(MathF_Max(a.X, MathF_Max(b.X, c.X)), MathF_Max(a.Y, MathF_Max(b.Y, c.Y)),
                    MathF_Max(a.Z, MathF_Max(b.Z, c.Z))));
    }
};

// ---- PointLight.cs:5-15 ----
class PointLight {
  public:
    Vector3 Position;
    Color4 Intensity;
    PointLight(Vector3 position, Color4 intensity) : Position(position), Intensity(intensity) {}
};

// ---- Camera.cs:9-124 (the value constructor and the derived plane corners; the quaternion
// rotations of 112-124 belong to interactive input, which is out of scope) ----
class Camera {
  public:
    Vector3 Position;
    float DistanceToCenter, Width, Height;
    Vector3 ViewDirection, RightDirection;
    Vector3 UpDirection() const { return Vector3::Cross(ViewDirection, RightDirection); }
    Camera(Vector3 position, Vector3 viewDirection, Vector3 rightDirection, float distanceToCenter, float width,
           float height)
        : Position(position),
          DistanceToCenter(distanceToCenter),
          Width(width),
          Height(height),
          ViewDirection(viewDirection.Normalized()),
          RightDirection(rightDirection.Normalized()) {}
    void SetDirection(Vector3 viewDirection, Vector3 rightDirection) {
        ViewDirection = viewDirection.Normalized();
        RightDirection = rightDirection.Normalized();
    }
    Vector3 TopLeftCameraPlane() const {
        return Position + (ViewDirection * DistanceToCenter) + (UpDirection() * Height / 2) +
               (-RightDirection * Width / 2);
    }
    Vector3 TopRightCameraPlane() const {
        return Position + (ViewDirection * DistanceToCenter) + (UpDirection() * Height / 2) +
               (RightDirection * Width / 2);
    }
    Vector3 BottomRightCameraPlane() const {
        return Position + (ViewDirection * DistanceToCenter) + (-UpDirection() * Height / 2) +
               (RightDirection * Width / 2);
    }
    Vector3 BottomLeftCameraPlane() const {
        return Position + (ViewDirection * DistanceToCenter) + (-UpDirection() * Height / 2) -
               (RightDirection * Width / 2);
    }
    // RayTracer.cs:947-954
    void PackCameraData(float screenWidth, float screenHeight, float out[14]) const {
        const Vector3 bl = BottomLeftCameraPlane(), br = BottomRightCameraPlane(), tl = TopLeftCameraPlane();
        const float v[14] = {Position.X, Position.Y, Position.Z, bl.X, bl.Y, bl.Z, br.X,
                             br.Y,       br.Z,       tl.X,       tl.Y, tl.Z, screenWidth, screenHeight};
        for (int i = 0; i < 14; i++) out[i] = v[i];
    }
};

// ---- PrimitiveStructsForBuffer.cs:8-74 (std430; the bool is one 32-bit word) ----
struct SphereStruct {
    float Center[3];
    float Radius;
    float DiffuseColor[3];
    uint32_t IsPureSpecular;
    float SpecularColor[3];
    float Specularity;
};
struct PlaneStruct {
    float Position[3];
    float padding0;
    float Normal[3];
    float padding1;
    float DiffuseColor[3];
    uint32_t IsPureSpecular;
    float SpecularColor[3];
    float Specularity;
};
struct TriangleStruct {
    float PointA[3];
    float padding0;
    float PointB[3];
    float padding1;
    float PointC[3];
    float padding2;
    float Normal[3];
    float padding3;
    float DiffuseColor[3];
    uint32_t IsPureSpecular;
    float SpecularColor[3];
    float Specularity;
};
static_assert(sizeof(SphereStruct) == 48 && sizeof(PlaneStruct) == 64 && sizeof(TriangleStruct) == 96,
              "buffer structs must keep the reference's std430 sizes");

}  // namespace rth
