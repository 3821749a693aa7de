/*
 * rt_mi355x.h -- C ABI of the MI355X (gfx950) per-pixel ray caster.
 *
 * Drop-in boundary for the Phlipsje/Raytracing host (reference citations are file:line in
 * that tree).  The reference has no FFI; its seam is the OpenGL call sequence in
 * RayTracer.cs.  Each entry point below replaces one of those call sites, takes exactly the
 * buffers the C# host already builds (PrimitiveStructsForBuffer.cs:8-74 layouts, the
 * serialised R-tree of RTree.cs:111-178, the lights array of RayTracer.cs:993-1005 and the
 * 14-float camera array of RayTracer.cs:947-954) and is what a P/Invoke stub would bind
 * (INTEGRATION.md shows the stub).
 *
 * Conventions: every call returns 0 on success and a negative rt_status on failure; no
 * exception crosses the ABI; rt_last_error() returns the message of the last failure on that
 * context.  One host thread per context (as the reference's single render thread,
 * OpenTK.cs:152,186,240).  Host arrays are read during the call only (glBufferData
 * semantics).  Pixel row 0 is the BOTTOM row (gl_FragCoord, raytracer_fs.glsl:302-303).
 *
 * Hit ids (the reference outputs none; defined from its own R-tree pointer convention,
 * IScene.cs:27-35, raytracer_fs.glsl:191-195): sphere i -> i, triangle j -> S + j,
 * plane k -> S + T + k, miss -> -1.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_MALFORMED_ACCEL = -2, /* accel array fails validation (the shader would read out of bounds) */
    RT_ERR_NO_SCENE = -3,
    RT_ERR_DEVICE = -4,          /* a HIP call failed; rt_last_error() has hipGetErrorString */
    RT_ERR_OUT_OF_MEMORY = -5,
    RT_ERR_TOO_LARGE = -6        /* index does not fit the packed node format (>= 2^24 slots) */
} rt_status;

/* Deterministic work counters of one frame, counted by the REFERENCE algorithm's rules (every
 * candidate tested, no distance cull; raytracer_fs.glsl:163-297): identical to the oracle's. */
typedef struct rt_stats {
    uint64_t primary_rays;      /* W*H                                                        */
    uint64_t secondary_rays;    /* executed mirror-loop iterations, raytracer_fs.glsl:334-381 */
    uint64_t shadow_rays;       /* ObjectInWayOfLight calls, raytracer_fs.glsl:350, 410       */
    uint64_t node_tests;        /* IntersectBoundingBox calls, raytracer_fs.glsl:172          */
    uint64_t refs_read;         /* accStruct[pos+8+i] reads, raytracer_fs.glsl:184, 191       */
    uint64_t sphere_tests;      /* IntersectSphere calls                                      */
    uint64_t plane_tests;       /* IntersectPlane calls on planes[]                           */
    uint64_t triangle_tests;    /* IntersectTriangle calls                                    */
    uint64_t material_fetches;  /* accepted closer hits in FindClosestIntersection            */
    uint64_t shadow_rays_zero_term; /* not a reference counter: those of shadow_rays whose light term is an exact
                                   zero whichever way the ray ends (light behind the surface, no highlight).  The
                                   reference traces them, the fast kernels do not (same pixels):
                                   rays the fast kernels trace = primary + secondary + shadow - this           */
    uint64_t reserved[6];
} rt_stats;

/* Render flags */
enum {
    RT_RENDER_DEFAULT = 0,
    RT_RENDER_REFERENCE_ORDER = 1, /* run the exhaustive kernel that follows the reference's
                                      candidate rules literally (used for rt_stats; slower) */
    RT_RENDER_GLOBAL_MEMORY = 2    /* never use the LDS-resident kernel, even if RT_OPTION_USE_LDS is set */
};

/* Replaces GL.CreateProgram + shader load/compile/link, RayTracer.cs:49-62.
 * device_id: HIP device ordinal (one context per GPU; multi-GPU hosts create one per rank). */
int rt_create(int device_id, rt_ctx** out);
void rt_destroy(rt_ctx* ctx);
/* One context over several GPUs of one node (the reference's host draws one frame per call on one
 * GPU, RayTracer.cs:125-137, and has nothing comparable).  device_ids[0] is the root.  On the returned context
 *   rt_upload_scene  replicates the scene on every device (validated and flattened once on the host),
 *   rt_render        cuts the frame into 8-row bands dealt round-robin to the devices, renders every device's
 *                    bands on its own stream, gathers them on the root over RCCL (grouped ncclSend/ncclRecv: each
 *                    peer's bands travel its own xGMI link) and restores scanline order there: one call = one
 *                    finished frame, as the reference's draw; results are those of a single-GPU context,
 *   rt_set_option / rt_get_scene_info / rt_last_error / rt_destroy act on the whole context.
 *   rt_render_device with rank 0, world 1 and neither d_bounces nor d_stats does the same and leaves the assembled
 *                    planes in the caller's buffers on the ROOT device (all requested planes in one grouped
 *                    exchange), asynchronously on `stream`, a stream of the root device,
 * The other device-resident entry points (rt_render_device with a band split, rt_render_frames_device,
 * rt_render_device_noalloc, rt_unpermute_*) address the ROOT device only.
 * The exchange over RCCL has run with one device only (and with one device named several times): UNVERIFIED on
 * hardware for n > 1 distinct devices.
 * Single process, one host thread; RCCL (librccl.so) is loaded when the first such context is created
 * (ncclCommInitAll).  A list that names one device several times is accepted for testing: the ranks are then
 * played on that GPU and the bands move with device-to-device copies instead of RCCL. */
int rt_create_multi(const int* device_ids, int n_devices, rt_ctx** out);
/* Devices of a context (1 for rt_create) and whether its gather goes over RCCL. */
int rt_multi_info(const rt_ctx* ctx, int* n_devices, int* uses_rccl);
/* The reference only prints GL info logs (RayTracer.cs:53-56, 1054). */
const char* rt_last_error(const rt_ctx* ctx);

/* Tuning knobs of a context (no counterpart in the reference).  Results never depend on them. */
enum {
    RT_OPTION_USE_LDS = 1,   /* 1: small scenes (triangle/sphere slots + BVH nodes + per-lane stacks within the
                                CU's 160 KB LDS) are rendered by the LDS-resident persistent form of the BVH
                                kernel; 0 (default): geometry stays in HBM/L2 (measured faster) */
    RT_OPTION_SUPERTILE = 2, /* edge, in 8x8-pixel tiles, of the super-tiles dealt round-robin to the
                                XCDs by the global-memory kernels (default 2, 1..64) */
    RT_OPTION_TRAVERSAL = 3, /* RT_TRAVERSAL_*: which fast kernel renders (default AUTO: by scene size) */
    RT_OPTION_TIGHT_TREE = 4, /* 1 (default): uploads of scenes with at least RT_OPTION_TIGHT_MIN_NODES node records
                                whose primitives were verified to sit inside their leaf boxes also build a
                                per-primitive tree that the staged pipeline walks with a cheap conservative box
                                test, as a filter in front of the reference's leaf-box rule (32 B per primitive:
                                child boxes on a 16-bit grid; ~0.25 s of host time per million); 0: neither built
                                nor used */
    RT_OPTION_WAVEFRONT_MIN_NODES = 5, /* scenes with at least this many flattened node records (default 1000) are
                                candidates for RT_TRAVERSAL_WAVEFRONT under AUTO: the first renders of a (scene, frame
                                size, bounces) time the candidate kernels -- three frames each, a warm-up and two
                                between events -- and later ones use the fastest (size alone does not tell a triangle
                                soup from a field of meshes); from 100000 nodes on AUTO takes the staged pipeline
                                without asking */
    RT_OPTION_TIGHT_MIN_NODES = 6, /* scenes with fewer flattened node records get no tight tree (default 1000; only
                                the staged pipeline walks it -- in the megakernel it measured slower) */
    RT_OPTION_COOP_SHADOWS = 7 /* 1: the BVH megakernel keeps the lanes of a wave in step through the mirror loop and
                                regroups the shadow rays of every shading step over all 64 lanes (ballot compaction
                                into an LDS queue of the rays whose light term is not an exact zero), so pixels that
                                are still bouncing use the lanes of those that have finished and a step takes
                                ceil(needed rays / 64) rounds instead of one per light; 0: every pixel traces its own
                                shadow rays one after the other; 2 (default): whichever measures faster for the
                                (scene, frame size, bounces), timed over the first renders like the staged pipeline
                                above.  Scenes with bounces gain 5-20 % (TestScene1, TeapotTable), the teapot frame
                                loses 8-30 % (lock-step, light rays formed again per queue entry).  Results do not
                                depend on it */,
    RT_OPTION_FRAME_FEEDBACK = 9, /* 1 (default): single frames of a scene whose root box spans the view and whose paths may
                                bounce three times or more deal their tiles out in an order ranked from the PREVIOUS launch's
                                wave lifetimes (longest first; same scene, size and bounce count), so the second and later
                                frames of a view are faster than its first (TestScene1 720p depth 10: 0.31 against 0.40 ms).
                                0: every frame is dealt out in the fixed order -- what a caller sets who wants timings that
                                do not depend on the frame before.  Pixels never depend on it */
    RT_OPTION_BIN_PATHS = 10, /* staged pipeline only.  n > 0: the lists of continuing paths that iterations >= n read are
                                sorted by the cell of the paths' new origin and their direction octant (counting sort on
                                the device, two small kernels per bounce) before they are traced.  0 (default): lists stay
                                in compaction order.  Measured on the 1M-triangle soup at 4K: neither gains nor loses
                                (47.7 ms both ways, profiles/r03_soup1m_binning_per_bounce.txt), hence off.  Pixels never
                                depend on it; rt_render_device_noalloc ignores it */
    RT_OPTION_PACKET_WALKS = 12, /* staged pipeline only.  1 (default): the primary rays of every 8x8 tile walk the tight tree
                                as ONE packet (a shared walk, per-lane tests), and so does the mirror bounce of the tile's
                                pixels that lie on one pure-specular plane (the reference scenes' mirror floors); 0: every
                                ray walks on its own.  1M-triangle soup at 4K: 43.0 against 44.1 ms.  Pixels never depend
                                on it */
    RT_OPTION_GPU_BUILD_MIN = 11, /* scenes whose two derived trees (re-treed leaves, tight tree) have at least this many
                                items together (default 20000) get them built on the GPU: binned SAH, large nodes level by
                                level, subtrees of <= 64 items by one wave each -- node arrays bit-identical to the host
                                builder's (rt_check_gpu_build), which smaller scenes and trees the device declines still
                                take.  A multi-GPU context builds on its first device and replicates.  -1: never */
    RT_OPTION_GPU_FLATTEN_MIN = 8 /* accel arrays of at least this many floats (default 65536) are validated and
                                flattened on the GPU, level by level (bit-identical to the host's walk, which smaller
                                arrays take and which still words every refusal); -1: never */
};
enum {
    RT_TRAVERSAL_AUTO = 0,
    RT_TRAVERSAL_WAVE = 1,   /* wave-cooperative stackless walk of the flattened tree (cache-resident scenes) */
    RT_TRAVERSAL_BVH = 2,    /* per-lane ordered traversal of the re-treed leaves, stacks in LDS; falls back
                                to WAVE when the uploaded tree's boxes are not nested */
    RT_TRAVERSAL_WAVEFRONT = 3 /* the same traversal, staged: per bounce a closest-hit, a shadow and a shading
                                stage over ray queues, idle lanes refilled by ballot compaction (large
                                scenes); falls back like BVH, and to BVH when max_bounces > 64 */
};
int rt_set_option(rt_ctx* ctx, int option, int value);

/* Replaces RayTracer.SendPrimitivesToShader, RayTracer.cs:957-1046 (glBufferData x4 +
 * glUniform x2).  spheres: SphereStruct[n] 48 B; planes: PlaneStruct[n] 64 B; triangles:
 * TriangleStruct[n] 96 B; accel: IScene.AccelerationStructureData; lights: 6 floats per light
 * (position xyz, intensity rgb), lights_len = 6*L.  The reference caps L at 50
 * (RayTracer.cs:960); this library does not.  The 32-bit word at the IsPureSpecular offset is
 * read as != 0.  Validates the accel array completely (bounds, forward child offsets, ids)
 * and flattens it for the GPU; a malformed array is rejected with RT_ERR_MALFORMED_ACCEL. */
int rt_upload_scene(rt_ctx* ctx, const void* spheres, int n_spheres, const void* planes, int n_planes,
                    const void* triangles, int n_triangles, const float* accel, int64_t accel_len,
                    const float* lights, int lights_len);

/* Replaces PrepareRenderOpenGL + RenderGL, RayTracer.cs:943-956 and 125-137 (glUniform of the
 * camera + the full-screen draw).  camera: the 14 floats exactly as packed at
 * RayTracer.cs:947-954 (position, bottom-left, bottom-right, top-left, width, height).
 * max_bounces: the shader's maxBounces constant (raytracer_fs.glsl:7 fixes it at 10).
 * Outputs are HOST arrays of W*H pixels, any may be NULL: rgb 3 floats (unclamped
 * outputColor.rgb), rgba8 = round-half-even(clamp(c,0,1)*255) with A=255, primary_id /
 * final_id = hit id of the first ray / of the finally shaded hit, stats (forces the
 * reference-order kernel). */
int rt_render(rt_ctx* ctx, const float camera[14], int max_bounces, float* rgb, uint8_t* rgba8,
              int32_t* primary_id, int32_t* final_id, rt_stats* stats);

/* Device-resident form of rt_render for hosts that keep the frame in HBM (no counterpart in
 * the reference, whose frame stays in the GL framebuffer, RayTracer.cs:135).
 * Renders the rows owned by `rank` when the image is cut into bands of `band_rows` rows dealt
 * round-robin to `world` ranks (rank 0, world 1, any band_rows = whole frame).  Local row r
 * is global row ((r / band_rows) * world + rank) * band_rows + r % band_rows; outputs are
 * DEVICE pointers to compact arrays of rt_local_rows() * W pixels (NULL = not wanted).
 * d_bounces: int32 bounce count per pixel.  d_stats: device pointer to one rt_stats that is
 * ACCUMULATED into (caller zeroes it), only honoured with RT_RENDER_REFERENCE_ORDER.
 * stream: a hipStream_t (NULL = default stream); the call is asynchronous on it.  One context may
 * render on several streams at once; renders of large scenes (RT_TRAVERSAL_WAVEFRONT) keep their ray
 * queues in per-context device buffers (allocated or grown inside this call on first use of a frame
 * size: 80 B per local pixel + one byte per shadow ray) and are therefore run one after the other on
 * the device, whichever streams they were issued on. */
int rt_render_device(rt_ctx* ctx, const float camera[14], int max_bounces, int flags, int band_rows,
                     int rank, int world, float* d_rgb, uint8_t* d_rgba8, int32_t* d_primary_id,
                     int32_t* d_final_id, int32_t* d_bounces, rt_stats* d_stats, void* stream);

/* rt_render_device for callers that cannot allocate device memory: code under HIP graph capture (the call does no
 * allocation and no event operation: tests/test_zz_no_alloc.py records one render into a graph and replays it), tight
 * serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE (its frame buffer and
 * every SSBO are allocated by the GL driver, RayTracer.cs:1007-1045).  Same arguments, same results as
 * rt_render_device with flags = RT_RENDER_DEFAULT and no d_stats; the difference is where the staged
 * pipeline (RT_TRAVERSAL_WAVEFRONT, large scenes) keeps its ray queues and per-path state: not in per-context
 * buffers grown with hipMalloc on first use of a frame size, but in one fixed array of
 * RT_NOALLOC_SCRATCH_BYTES inside the library's code object (each GPU the library runs on holds its own copy
 * until the process ends).  This call performs no hipMalloc / hipFree / event creation.  A frame whose
 * path state (72 B per local pixel + one byte per light for every pixel of the frame's 8x8 tiles) exceeds the
 * array is rejected with RT_ERR_TOO_LARGE before anything is launched; the megakernels (small scenes) keep no
 * such state and accept any size.  AUTO does not time mid-sized scenes here (a verdict rt_render_device
 * measured earlier is reused, otherwise the megakernel renders).  All calls on one GPU share that GPU's
 * array: calls that overlap on one GPU -- two contexts, this context on two streams, replays of a graph beside
 * other renders -- race; the library does not order these calls, the caller does. */
#define RT_NOALLOC_SCRATCH_BYTES (96u << 20)
int rt_render_device_noalloc(rt_ctx* ctx, const float camera[14], int max_bounces, int band_rows, int rank,
                             int world, float* d_rgb, uint8_t* d_rgba8, int32_t* d_primary_id,
                             int32_t* d_final_id, int32_t* d_bounces, void* stream);

/* Batch form of rt_render_device for hosts that render several small frames per step (one rank of a
 * band-sharded multi-GPU render gets 1/world of every frame: a launch per frame would be latency bound).
 * cameras: n_frames x 14 floats (host), all of one width/height; frame f's pixels start
 * f * frame_stride PIXELS after frame 0's in every output plane (frame_stride >= rt_local_rows() * W).
 * The megakernels render up to 8 frames per launch; large scenes (RT_TRAVERSAL_WAVEFRONT), the LDS
 * kernel and RT_RENDER_REFERENCE_ORDER take one launch (sequence) per frame.  Results are those of
 * n_frames rt_render_device calls.  No counterpart in the reference (one frame per draw). */
int rt_render_frames_device(rt_ctx* ctx, const float* cameras, int n_frames, int max_bounces, int flags,
                            int band_rows, int rank, int world, int64_t frame_stride, float* d_rgb,
                            uint8_t* d_rgba8, int32_t* d_primary_id, int32_t* d_final_id, int32_t* d_bounces,
                            void* stream);

/* Number of rows rank renders (see rt_render_device), and the padded per-rank row count that
 * makes every rank's contribution equal-sized for a gather. */
int rt_local_rows(int height, int band_rows, int rank, int world);
int rt_padded_rows(int height, int band_rows, int world);

/* After an all-gather/gather of `world` equal contributions of rt_padded_rows()*W pixels of
 * elem_size bytes each (rank-major), restore scanline order into d_out (H*W pixels).
 * elem_size must be 4 or 12.  No counterpart in the reference (single GPU). */
int rt_unpermute_bands(rt_ctx* ctx, const void* d_gathered, void* d_out, int elem_size, int width,
                       int height, int band_rows, int world, void* stream);
/* The same for a batch of n_frames frames exchanged with ONE collective: d_gathered holds
 * [world][n_frames][rt_padded_rows()][W] pixels, d_out receives [n_frames][H][W].  Small frames are
 * latency bound on xGMI, so hosts that render several frames per step gather them together. */
int rt_unpermute_frames(rt_ctx* ctx, const void* d_gathered, void* d_out, int elem_size, int width,
                        int height, int band_rows, int world, int n_frames, void* stream);

/* Introspection for tests and the bench: sizes of the flattened structures on the device. */
typedef struct rt_scene_info {
    int64_t n_nodes;        /* flattened node records (32 B each)                  */
    int64_t n_tri_slots;    /* triangle geometry slots in traversal order (48 B)   */
    int64_t n_sphere_slots; /* sphere geometry slots in traversal order (16 B)     */
    int64_t max_depth;      /* deepest node of the R-tree                          */
    int64_t device_bytes;   /* bytes of HBM held by the scene                      */
    int64_t lds_scene_bytes;/* bytes the LDS-resident variant would need (0 = n/a) */
    int64_t bvh_nodes;      /* 64-byte nodes of the re-treed leaves (0 = not built)    */
    int64_t bvh_levels;     /* LDS stack entries per lane the BVH kernel reserves      */
    int64_t flags;          /* RT_SCENE_*                                              */
    int64_t traversal;      /* RT_TRAVERSAL_* a default render (max_bounces 10) uses; for a mid-sized scene under
                               AUTO the measured verdict once there is one (RT_OPTION_WAVEFRONT_MIN_NODES)       */
    int64_t tight_nodes;    /* 32-byte nodes of the per-primitive tree (0 = not built) */
    int64_t tight_levels;   /* its depth                                               */
    int64_t unbounded_triangles; /* triangle slots with (nearly) collinear or coincident vertices as seen
                               along their stored normal: the reference accepts those on a whole line or
                               plane, so the distance cull never skips the subtrees that hold them        */
    int64_t reserved[3];
} rt_scene_info;
enum {
    RT_SCENE_NESTED = 1,    /* every box lies inside its parent's (true for every RTree.cs tree) */
    RT_SCENE_CULL = 2       /* ... and every primitive inside its leaf box: distance culling is on */
};
int rt_get_scene_info(const rt_ctx* ctx, rt_scene_info* out);

/* Host-only validation + flatten dry run of an accel array: the same checks rt_upload_scene
 * applies, without a context or a GPU.  info (may be NULL) receives the flattened sizes; err
 * (may be NULL) receives the message.  Returns RT_OK, RT_ERR_MALFORMED_ACCEL or RT_ERR_TOO_LARGE.
 * The reference never validates: a bad array makes the shader read out of bounds
 * (raytracer_fs.glsl:65-68, 173-195). */
int rt_check_accel(const float* accel, int64_t accel_len, int n_spheres, int n_triangles, rt_scene_info* info,
                   char* err, size_t err_cap);
/* The same with the primitives: everything rt_upload_scene decides on the host (cull preconditions ->
 * flags, unbounded_triangles, the re-treed leaves, the tight tree for scenes of at least tight_min_nodes
 * flattened nodes), followed by a structural self-check of both trees (exact unions, every slot reached
 * once, depth bound, tight boxes contain their primitive's accepted region + pad) and a second build of both
 * from the same items in another order, which must give the same node arrays (the device builder,
 * RT_OPTION_GPU_BUILD_MIN, relies on that).  No context, no GPU. */
int rt_check_scene(const void* spheres, int n_spheres, const void* triangles, int n_triangles, const float* accel,
                   int64_t accel_len, int tight_min_nodes, rt_scene_info* info, char* err, size_t err_cap);

/* Self-check of the GPU flatten (RT_OPTION_GPU_FLATTEN_MIN): flattens the array on the host and on ctx's GPU,
 * whatever its size, and compares node records, slot maps, parent links, depth and the nested flag bit for bit.
 * Returns 1 (identical; the device took the array), 0 (an array too short to hold a node, or a tree of more than
 * 1024 levels, which the device leaves to the host's linear walk: host only), the host's
 * refusal (RT_ERR_MALFORMED_ACCEL / RT_ERR_TOO_LARGE) when the device declined it too -- an upload then reports
 * the host's message -- or RT_ERR_DEVICE with the first difference in err / rt_last_error.  Used by the tests. */
int rt_check_gpu_flatten(rt_ctx* ctx, const float* accel, int64_t accel_len, int n_spheres, int n_triangles, char* err,
                         size_t err_cap);

/* Self-check of the GPU tree builds (RT_OPTION_GPU_BUILD_MIN): builds the re-treed leaves and the tight tree of the
 * scene (tight_min_nodes as in rt_check_scene) on the host threads and on ctx's GPU, whatever their size, and compares
 * the node arrays, the depths, the second root holder and the 16-bit grid form bit for bit.  Returns 1 (identical), 0
 * (the device declined a tree: a large node of coincident centroids, which it leaves to the host's median split), a
 * refusal of the accel array as rt_upload_scene words it, or RT_ERR_DEVICE with the first difference in err /
 * rt_last_error.  Also compares the device's form of the cull preconditions (verdict, unbounded triangles, never-cull
 * marks) and the slot -> leaf maps.  The context's scene, if any, is gone afterwards (its buffers are the route's
 * staging area): upload again.  Used by the tests.  No counterpart in the reference (its tree arrives built,
 * RTree.cs:180-377). */
int rt_check_gpu_build(rt_ctx* ctx, const void* spheres, int n_spheres, const void* triangles, int n_triangles, const float* accel,
                       int64_t accel_len, int tight_min_nodes, char* err, size_t err_cap);

/* Per-function device evaluators of the four intersection routines, one case per element
 * (raytracer_fs.glsl:134-161, 93-113, 114-120, 121-131).  Host arrays; used by the parity
 * tests to pin the device arithmetic bit-for-bit against the oracle. */
int rt_eval_box(rt_ctx* ctx, int n, const float* o3, const float* d3, const float* box6, int32_t* hit);
int rt_eval_sphere(rt_ctx* ctx, int n, const float* o3, const float* d3, const float* c3r, float* out4);
int rt_eval_plane(rt_ctx* ctx, int n, const float* o3, const float* d3, const float* p3, const float* n3,
                  float* t);
int rt_eval_triangle(rt_ctx* ctx, int n, const float* o3, const float* d3, const float* a3, const float* b3,
                     const float* c3, const float* n3, float* t);

#ifdef __cplusplus
}
#endif
#endif
