/*
 * rt_oracle.h -- C ABI of the CPU parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (raytracing_amd/, include/,
 * bench.py's GPU leg) may include, link, import or execute anything under oracle/.
 * Allowed users: tests/, __graft_entry__.smoke(), and bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference (Phlipsje/Raytracing @ v2) holds no tests, golden images or
 * known-answer vectors, and neither its C# host nor its GLSL shader can be built or run in
 * this image (no dotnet/mono/glslang, no GL device).  This oracle is therefore a line-by-line
 * C++ restatement of the reference's source, read as text, and is checked only against
 * hand-derivable facts (struct sizes, asset triangle counts, analytic ray/primitive cases).
 *
 * Canonical arithmetic (shared with the HIP kernels): IEEE binary32, evaluation left to right
 * as written in the reference, no FMA contraction (-ffp-contract=off), correctly rounded
 * / and sqrt, dot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z, length = sqrtf(dot),
 * normalize(v) = v * (1.0f / sqrtf(dot(v,v))), max(0,x) = (0 < x) ? x : 0, pow = powf.
 * Shader undefined behaviour is given defined meaning: candidate counters start at 0, the
 * traversal stack and candidate lists are unbounded.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* counters[] slots written by orc_render (all uint64) */
enum {
    ORC_C_PRIMARY = 0,      /* W*H of the rendered region                                  */
    ORC_C_SECONDARY = 1,    /* executed mirror-loop iterations (raytracer_fs.glsl:334-381) */
    ORC_C_SHADOW = 2,       /* ObjectInWayOfLight calls (raytracer_fs.glsl:350, 410)       */
    ORC_C_NODE_TESTS = 3,   /* IntersectBoundingBox calls (raytracer_fs.glsl:172)          */
    ORC_C_REFS_READ = 4,    /* accStruct[pos+8+i] reads (raytracer_fs.glsl:184, 191)       */
    ORC_C_SPHERE_TESTS = 5, /* IntersectSphere calls                                       */
    ORC_C_PLANE_TESTS = 6,  /* IntersectPlane calls on planes[] (not via triangles)        */
    ORC_C_TRI_TESTS = 7,    /* IntersectTriangle calls                                     */
    ORC_C_MATERIAL = 8,     /* accepted closer hits in FindClosestIntersection             */
    ORC_C_MAX_STACK = 9,    /* deepest traversal stack seen (shader's fixed size is 100)   */
    ORC_C_MAX_SPH_CAND = 10,/* longest sphere candidate list (shader's fixed size is 100)  */
    ORC_C_MAX_TRI_CAND = 11,/* longest triangle candidate list (shader's fixed size 500)   */
    ORC_C_ZERO_TERM = 12,   /* NOT a property of the reference: those of ORC_C_SHADOW whose light term is an exact
                               zero whether or not the ray is blocked (the shadow rays the HIP fast kernels leave
                               untraced); lets the tests check that count on the CPU                              */
    ORC_N_COUNTERS = 16
};

/* Render the pixel rectangle [x0,x1) x [y0,y1) of the W x H frame named by camera[12..13].
 * Row 0 is the BOTTOM row (gl_FragCoord).  Output arrays hold (x1-x0)*(y1-y0) pixels, row
 * major inside the rectangle; any of them may be NULL.  Returns 0, or <0 on malformed input. */
int orc_render(const void* spheres, int n_spheres, const void* planes, int n_planes,
               const void* triangles, int n_triangles, const float* accel, int64_t accel_len,
               const float* lights, int lights_len, const float* camera14, int max_bounces,
               int x0, int y0, int x1, int y1,
               float* rgb, uint8_t* rgba8, int32_t* primary_id, int32_t* final_id,
               int32_t* bounces, uint64_t* counters, int n_threads);

/* Per-function vectors (n independent cases each). */
void orc_intersect_box(int n, const float* o3, const float* d3, const float* box6, int32_t* hit);
void orc_intersect_sphere(int n, const float* o3, const float* d3, const float* c3r, float* out4);
void orc_intersect_plane(int n, const float* o3, const float* d3, const float* p3, const float* n3, float* t);
void orc_intersect_triangle(int n, const float* o3, const float* d3, const float* a3, const float* b3,
                            const float* c3, const float* n3, float* t);

/* Scene-side restatements. */
typedef struct orc_scene orc_scene;
/* TestScene1.cs:16-56, reading the four .KarelModel assets from asset_dir. */
orc_scene* orc_scene_testscene1(const char* asset_dir);
/* IScene.cs:22-38 on caller-packed structs (spheres 48 B, triangles 96 B). */
orc_scene* orc_scene_from_structs(const void* spheres, int n_spheres, const void* planes, int n_planes,
                                  const void* triangles, int n_triangles,
                                  const float* lights, int lights_len);
/* which: 0 spheres(48 B) 1 planes(64 B) 2 triangles(96 B) 3 accel(float) 4 lights(float)
 * 5 camera (14 floats, packed for screen 1280x720 as RayTracer.cs:42, 947-954). */
int orc_scene_get(const orc_scene* s, int which, const void** ptr, int64_t* count);
const char* orc_scene_error(const orc_scene* s);   /* non-empty if the C# code would have thrown */
void orc_scene_free(orc_scene* s);

/* OBJImportHelper.cs:19-75: returns triangle count, writes up to cap triangles as 9 floats
 * (the three raw constructor vertices after scale/position), or <0 on I/O error. */
int orc_import_model(const char* path, float scale, float px, float py, float pz, float* raw9, int cap);
/* Triangle.cs:27-36 + 14-16: raw vertices -> PointA,PointB,PointC (Center+VectorX), Normal. */
void orc_make_triangle(const float* raw9, float* out12);
/* Camera.cs:92-99, 17-30 + RayTracer.cs:947-954. */
void orc_camera_pack(const float* pos3, const float* view3, const float* right3, float dist,
                     float width, float height, float screen_w, float screen_h, float* out14);

#ifdef __cplusplus
}
#endif
#endif
