// rt_math.h -- canonical binary32 arithmetic shared by the HIP kernels and the C++ host mirror.
//
// Rules (DESIGN.md "canonical arithmetic"): evaluation left to right as written in the
// reference, no FMA contraction (translation units are built with -ffp-contract=off and this
// header repeats it as a pragma), correctly rounded / and sqrt, NaN comparisons are false.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace rtm {

struct V3 {
    float x, y, z;
};
RT_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RT_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
RT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
RT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_HD float length(V3 a) { return sqrtf(dot(a, a)); }
// GLSL normalize / OpenTK Vector3.Normalized(): scale = 1 / length, then component * scale
RT_HD V3 normalize(V3 a) {
    float s = 1.0f / sqrtf(dot(a, a));
    return v3(a.x * s, a.y * s, a.z * s);
}
// GLSL max(0, x) = (0 < x) ? x : 0  (NaN -> 0)
RT_HD float max0(float x) { return (0.0f < x) ? x : 0.0f; }

}  // namespace rtm
