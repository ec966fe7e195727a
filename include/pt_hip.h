/* pt_hip.h -- C-ABI of the MI355X-native path tracer (libpt_hip.so).
 *
 * Drop-in boundary for ONE hot path of DeveloperDenis/PathTracer: the span the reference times as
 * "PathTracing" (src/main.cpp:412-481), i.e. every call of run_thread_batch (src/main.cpp:334-362) and what
 * it calls -- Camera::get_ray (src/camera.cpp:57-79), cast_ray (src/main.cpp:171-307), the sphere / plane /
 * BVH intersection (src/geometry.cpp:39-115, src/main.cpp:53-91) and the diffuse / metal / dielectric scatter
 * with its samplers (src/utils.h:71-138).  The reference has no plugin or FFI interface; the
 * one seam is the thread-pool work item `run_thread_batch(.., ThreadData*, ..)` whose inputs are a read-only
 * World + Camera + BVH and whose output is Image.pixels.  The functions below take exactly those inputs
 * (flattened to plain arrays) and produce that output; INTEGRATION.md shows the 30-line change to the
 * reference's main() that binds them.  Two stated deviations: the RNG is Philox, not the
 * reference's thread-local MT19937, so agreement is bit-exact per function on identical uniforms and statistical per
 * image; and the sphere query returns the true closest hit, not the result of the reference BVH's pairwise combine
 * (src/main.cpp:78-87), which drops 0.1-0.2 % of secondary hits -- converged images sit at 44.0 dB PSNR against the
 * reference renderer where two reference runs agree to 50.6 dB (mean relative error 0.36 % vs 0.31 %).
 *
 * Conventions: every entry point returns an int status (0 = PT_OK, negative = pt_status); nothing aborts or
 * throws across the boundary (the reference is built /EHa- /GR-, src/build.bat:7): every entry point runs behind a
 * catch-all barrier (a failed host allocation comes back as PT_ERR_NO_MEMORY, or NULL from the functions that return a
 * handle), and a worker thread that cannot be started degrades to fewer threads, down to the calling one.  The caller owns all host
 * buffers before and after each call; the library copies on upload and owns its device memory inside pt_ctx.
 * A pt_ctx is bound to one GPU and is driven by one host thread at a time, with ONE render in flight at a time: the
 * unit cursor, statistics, partial images and timing events are per context, so a render enqueued on another stream
 * waits (on the device) for the context's previous one; distinct contexts are independent.
 * The SAMPLES of a frame depend only on (scene, params) -- never on the GPU count, launch geometry or scheduling --
 * because the RNG is Philox4x32-10 keyed by (seed, pixel, global sample index) (DESIGN.md "RNG").  The ORDER in which
 * a pixel's samples are added is the run's by default (sums equal to a few ulp); PT_FLAG_REPRODUCIBLE fixes it too.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_ctx pt_ctx;
typedef struct pt_scene pt_scene;

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,     /* NULL pointer, zero size, inconsistent counts */
    PT_ERR_NO_DEVICE = -2,   /* no such GPU / HIP runtime unavailable */
    PT_ERR_HIP = -3,         /* a HIP call failed; pt_hip_last_error(ctx) has the text */
    PT_ERR_NO_SCENE = -4,    /* render / probe before pt_hip_upload_scene */
    PT_ERR_UNSUPPORTED = -5, /* e.g. more planes than the reference's 64 (src/render_world.h:88) */
    PT_ERR_IO = -6,          /* file could not be written */
    PT_ERR_NO_MEMORY = -7,   /* a host allocation failed inside the library (std::bad_alloc caught at the boundary) */
    PT_ERR_INTERNAL = -8     /* any other C++ exception caught at the boundary; pt_hip_last_error(ctx) has its text */
} pt_status;

/* Material::Type, src/render_world.h:32-38 (same numeric values, same spelling of DIALECTRIC) */
typedef enum pt_material_type { PT_MAT_NONE = 0, PT_MAT_DIFFUSE = 1, PT_MAT_METAL = 2, PT_MAT_DIALECTRIC = 3 } pt_material_type;

/* Flattened World + Camera (src/render_world.h:57-96, src/camera.h:9-38).  All arrays are caller-owned. */
typedef struct pt_scene_desc {
    uint32_t n_spheres;
    const float* sphere_pos_r;       /* 4*n : Sphere{pos.xyz, radius}                 src/geometry.h:16-23   */
    const float* sphere_vel;         /* 3*n : SphereObject::velocity, or NULL = static src/render_world.h:64  */
    const uint32_t* sphere_mat_type; /*   n : Material::type                           src/render_world.h:40  */
    const float* sphere_mat_rgba;    /* 4*n : Material::colour (alpha included)        src/render_world.h:41  */
    const float* sphere_mat_param;   /* 2*n : Material::roughness, Material::n         src/render_world.h:44,47 */
    uint32_t n_planes;               /* <= 64                                          src/render_world.h:88  */
    const float* plane_n_d;          /* 4*np: Plane{normal.xyz, offset}                src/geometry.h:25-29   */
    const uint32_t* plane_mat_type;
    const float* plane_mat_rgba;
    const float* plane_mat_param;
    float t0, t1;                    /* World::startTime, endTime                      src/render_world.h:91-92 */
    float camera[13];                /* pos[3] dir[3] up[3] fov(rad) aspect focusDistance lensRadius  src/camera.h:14-21,37 */
} pt_scene_desc;

/* The reference's compile-time settings (src/main.cpp:29-42) as run-time parameters. */
typedef struct pt_params {
    uint32_t width, height;       /* Image::width/height, src/image.h:4-9 */
    uint32_t spp_begin, spp_end;  /* global sample indices [begin,end) of SAMPLES_PER_PIXEL rendered by this call */
    uint32_t max_depth;           /* MAX_RAY_DEPTH */
    uint32_t flags;               /* PT_FLAG_* */
    uint64_t seed;                /* Philox key */
} pt_params;

/* How the closest-sphere query is answered (all give the same definition of the answer):
 *   default            : BVH traversal with the tree staged in LDS when it fits (book-1 scene: 18 KB), else from HBM
 *                        (from 32768 spheres up as 4-wide 64-byte nodes on a 16-bit grid; leaf tests stay exact)
 *   PT_FLAG_BRUTE_FORCE: two-level brute force over the LDS-resident SoA scene in packed FP32 (the FP32-VALU-bound
 *                        formulation of BASELINE.json's north star; approximate search + exact re-evaluation)
 *   PT_FLAG_STRICT_SEARCH: brute force in reference arithmetic for every sphere (slow; parity tests)
 *   PT_FLAG_FORCE_BVH  : BVH traversal from HBM even if the scene fits in LDS
 *   PT_FLAG_FORCE_LDS  : fail with PT_ERR_UNSUPPORTED rather than read the scene from HBM
 * How the per-pixel sums are formed (the samples are the same either way, each exactly once):
 *   default            : a wave starts its next (tile, sample-chunk) unit while the last paths of the previous one are
 *                        still under way, and the units of a pixel's sample chunks ADD their sums to the output with
 *                        float atomics; the order of the additions into a pixel then depends on the run, and two
 *                        renders of one frame can differ in the last bits of a sum (a few ulp).  No scratch images.
 *   PT_FLAG_REPRODUCIBLE: every unit is finished before the next one starts and each sample chunk goes to a partial image
 *                        of its own (context-owned scratch, width*height*16 bytes per chunk, allocated on first use and
 *                        kept), folded in chunk order: the same launch gives the same bits, run to run (~6 % slower on
 *                        the bench workload) */
#define PT_FLAG_STRICT_SEARCH 1u
#define PT_FLAG_FORCE_BVH 2u
#define PT_FLAG_FORCE_LDS 4u
#define PT_FLAG_BRUTE_FORCE 8u
#define PT_FLAG_REPRODUCIBLE 16u
/* Compatibility mode: answer the sphere query the way the reference does -- its recursive
 * traversal of ITS tree (src/main.cpp:53-91: exact hit_test per inner node, both children always, no pruning, raw t at
 * the leaves) with the pairwise combine of src/main.cpp:78-87, which discards a sibling's valid hit when the other
 * child returns a t <= 0.001 that is smaller.  Needs the reference-topology tree (pt_hip_build_reference_tree; built
 * with rng_skip = 0 on first use otherwise); scenes of at most PT_REFERENCE_MAX_SPHERES spheres.  Several times slower
 * than the default; bit-equal to the reference's own query on the same tree (tests/golden q_t / q_obj), and converged
 * images then meet the reference renderer at its own noise floor (50.6 dB on the book-1 scene, not D1's 44 dB). */
#define PT_FLAG_REFERENCE_QUERY 32u
/* Progressive / resumable rendering (the reference's only analogue is its progress print,
 * src/main.cpp:321-332): the frame's per-pixel sums are ADDED to what the output buffer already holds instead of replacing
 * it, so a caller renders [0, 4096) in pieces into one resident buffer (zero it before the first piece).  pt_hip_render_device
 * / _static and pt_node_render honour it; with PT_FLAG_REPRODUCIBLE the piece is folded in its fixed order first and added
 * once, so piece k of a sequence leaves the same bits run to run. */
#define PT_FLAG_ACCUMULATE 64u
#define PT_REFERENCE_MAX_SPHERES 4192u /* World::objects[4192], src/render_world.h:85 */

typedef struct pt_stats {
    uint64_t samples;       /* width*height*(spp_end-spp_begin) */
    uint64_t segments;      /* loop iterations of cast_ray over all samples (S-bar = segments/samples) */
    uint64_t sphere_tests;  /* ray-sphere evaluations executed (LDS kernel: groups searched x 8 x 64 lanes; BVH kernel: leaf tests) */
    uint64_t node_visits;   /* bound tests executed (LDS kernel: segments x groups; BVH kernel: box tests) */
    uint64_t wave_iterations; /* trips of the per-wave segment loop; lane utilisation = segments / (64 * wave_iterations) */
    uint64_t phase_cycles[3]; /* diagnostic builds (-DPT_STAMPS) only: shader cycles in refill / closest-hit / shading; else 0 */
    double kernel_ms;       /* HIP-event time of the render kernel on its stream */
    uint32_t kernel;        /* 0 = LDS brute force (fast), 1 = BVH in HBM, 2 = LDS brute force (strict), 3 = BVH in LDS, 4 = reference tree + pairwise combine, 5 = straight loop over leaf blocks (scenes of at most 8 spheres) */
    uint32_t node_bytes;    /* bytes per tree node the BVH kernels read: 64 (float boxes, or four 16-bit grid boxes) or 32 (two 16-bit grid boxes); 0 without a tree */
    uint32_t node_width;    /* child boxes per node: 2, or 4 (big trees in HBM); 0 without a tree */
    uint32_t reserved;
} pt_stats;

/* ---- device / context -------------------------------------------------------------------------------- */
int pt_hip_device_count(void);
int pt_hip_create(int device, pt_ctx** out);
void pt_hip_destroy(pt_ctx* ctx);
const char* pt_hip_strerror(int status);
const char* pt_hip_last_error(pt_ctx* ctx);

/* Replaces: World/Camera being handed to the workers through ThreadData (src/main.cpp:421-427) and
 * build_bvh_tree (src/main.cpp:402).  Copies the scene, lays it out for the device, builds the BVH if needed. */
int pt_hip_upload_scene(pt_ctx* ctx, const pt_scene_desc* scene);

/* The same upload in two steps, for callers that put ONE scene on SEVERAL GPUs (pt_node_upload_scene does): the
 * reference builds its tree once per process (src/main.cpp:402) and every worker reads it.
 *   pt_scene_prepare    host only, touches no device: validates the scene, builds the BVH and every array the kernels
 *                       read, in their device layout.  The result is immutable and may be committed to any number of
 *                       contexts, from any thread.
 *   pt_hip_commit_scene allocations + copies on the context's GPU; the prepared scene may be freed right after it.
 *   pt_tree_build_count diagnostic: how many times this process has run the BVH builder (tests: one build, N commits). */
typedef struct pt_prepared pt_prepared;
int pt_scene_prepare(const pt_scene_desc* scene, pt_prepared** out);
int pt_hip_commit_scene(pt_ctx* ctx, const pt_prepared* prepared);
void pt_prepared_free(pt_prepared* prepared);
uint64_t pt_tree_build_count(void);

/* Replaces: build_bvh_tree (src/main.cpp:131-159, called at src/main.cpp:402) in the reference's own recipe, for
 * PT_FLAG_REFERENCE_QUERY: random axis per node from the main thread's std::mt19937 stream (src/utils.h:38-63), in-place
 * first-pivot quicksort of the sub-range (src/main.cpp:93-129), midpoint split, swept leaf boxes.  rng_skip = number of
 * random_f32 draws the caller's scene construction took from that stream before the build (the reference builds the
 * tree right after init_test_scene_N on the same thread, src/main.cpp:396-402); 0 = a fresh stream, which is what the
 * fixtures under tests/golden were generated with.  Call after pt_hip_upload_scene; PT_ERR_UNSUPPORTED beyond
 * PT_REFERENCE_MAX_SPHERES spheres. */
int pt_hip_build_reference_tree(pt_ctx* ctx, uint32_t rng_skip);
/* Host-only (touches no device): the same tree for inspection / export -- the reference-topology exporter of SURVEY.md
 * 8f rank 2.  Nodes in the order the reference allocates them (children before their parent, the root last):
 * boxes6[6*i] = Rect3f{pos.xyz, halfWidth, halfHeight, halfLength} (src/geometry.h:31-52), links3[3*i] = {left, right,
 * object} as node / sphere indices into the CALLER's arrays, -1 for none; order[k] = index of the sphere that the
 * reference's in-place sorts leave at position k of World::objects (n_spheres entries).  Any output may be NULL.
 * Returns the node count 2*n_spheres - 1 (outputs are written only if it is <= cap_nodes), 0 on invalid input or more
 * than PT_REFERENCE_MAX_SPHERES spheres. */
uint32_t pt_reference_bvh(const pt_scene_desc* scene, uint32_t rng_skip, uint32_t cap_nodes, float* boxes6, int32_t* links3,
                          uint32_t* order, uint32_t* depth_out);

/* Replaces: the thread-pool submit + join over all 32x32 blocks (src/main.cpp:415-477).
 * accum_rgba_host: width*height*4 floats, row-major top-down like Image.pixels (src/image.h:11-14); receives the
 * per-pixel SUM of the samples [spp_begin,spp_end) (not the mean: partial sums of several GPUs / calls add up;
 * pt_resolve() turns the total into the reference's clamp(mean) pixels).  Synchronous. */
int pt_hip_render(pt_ctx* ctx, const pt_params* params, float* accum_rgba_host);

/* Same, into caller-owned DEVICE memory on a caller-owned stream (hipStream_t as void*; NULL = default stream).
 * Asynchronous: returns after enqueueing (a PT_FLAG_REPRODUCIBLE frame bigger than any before it on this context first
 * grows the context's scratch with hipMalloc; the default mode never allocates).  image width*height < 2^31.  This is what a multi-GPU driver all-reduces (RCCL) before resolving.
 * accum_rgba_dev: 16-byte aligned, ordinary device memory of this GPU (hipMalloc / a torch CUDA tensor): without
 * PT_FLAG_REPRODUCIBLE the kernel may add to it with hardware FP32 atomics, which fine-grained or host-mapped
 * allocations do not guarantee. */
int pt_hip_render_device(pt_ctx* ctx, const pt_params* params, float* accum_rgba_dev, void* stream);

/* The same frame as pt_hip_render_device, bit for bit, through an entry point that ALLOCATES NO DEVICE MEMORY: no
 * hipMalloc or hipFree on any path.  Host side: the first call for a new (kernel, workgroup shape) asks the runtime for
 * that kernel's occupancy and remembers the answer in storage reserved at pt_hip_create; error texts are formatted into
 * storage reserved there too; nothing else touches the heap.  Under stream capture the context's timing events are left
 * out of the graph (pt_hip_last_stats then has nothing for that launch).  No counterpart in the reference; it is for
 * callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  The scratch that
 * pt_hip_render_device keeps in (and grows inside) the context -- the unit cursor and per-wave statistics records the
 * persistent waves use, and the per-chunk partial images of a PT_FLAG_REPRODUCIBLE frame -- lives in a file-scope
 * `static __device__` array instead (1 MiB of cursor/statistics for up to 8192 resident waves + 32 MiB of partial
 * images per GPU, held until the process ends).  Limits: a PT_FLAG_REPRODUCIBLE frame whose sample chunks need more
 * than 32 MiB of partial images (width*height*16 bytes per chunk) is refused with PT_ERR_UNSUPPORTED and a message in
 * pt_hip_last_error; without that flag any frame size is served.  Every call on a GPU -- from any context -- shares
 * that GPU's array: calls that overlap in time on one GPU race (serialise them on one stream, or use
 * pt_hip_render_device for the concurrent ones); pt_hip_last_stats after it reads the shared array, so take the
 * statistics before the next static call on that GPU.  accum_rgba_dev and params as for pt_hip_render_device. */
int pt_hip_render_device_static(pt_ctx* ctx, const pt_params* params, float* accum_rgba_dev, void* stream);

/* ---- one node, several GPUs --------------------------------------------------------------- */
/* Replaces: the thread-pool submit + join (src/main.cpp:448-477) at node scale.  A pt_node owns one pt_ctx, one stream,
 * one f32 sum buffer and (with more than one GPU) one RCCL communicator per GPU, devices 0 .. n_gpus-1, all driven by the
 * calling host thread.  pt_node_render splits the samples [spp_begin, spp_end) of EVERY pixel into n_gpus contiguous
 * slices (pt_spp_slice), renders slice g on GPU g (pt_hip_render_device), joins the per-GPU sums with ONE
 * ncclAllReduce(sum, f32) over xGMI, resolves on device 0 and copies back what was asked for: the f32 sums
 * (width*height*4, may be NULL) and / or the packed 0xAARRGGBB pixels (width*height, may be NULL).  Synchronous.  The
 * sample set is that of a one-GPU render of the same range (Philox is keyed by the global sample index); the sums
 * agree to float summation order.  RCCL is bound at run time: PT_ERR_UNSUPPORTED if n_gpus > 1 and it cannot be found.
 * pt_node_upload_scene builds the scene's tree ONCE on the host (pt_scene_prepare) and commits it to every GPU, as the
 * reference builds once per process (src/main.cpp:402).  A call that fails drains every GPU's stream before it returns. */
typedef struct pt_node pt_node;
int pt_node_create(int n_gpus, pt_node** out);
int pt_node_upload_scene(pt_node* node, const pt_scene_desc* scene);
int pt_node_render(pt_node* node, const pt_params* params, float* accum_rgba_host, uint32_t* bgra_host);
void pt_node_destroy(pt_node* node);
const char* pt_node_last_error(pt_node* node);
int pt_node_gpu_count(const pt_node* node);
pt_ctx* pt_node_ctx(pt_node* node, int gpu);   /* GPU g's context (statistics, reference tree, ...); owned by the node */
/* Progressive / resumable rendering through the node (reference analogue: the per-block progress
 * print, src/main.cpp:321-332).  pt_node_render with PT_FLAG_ACCUMULATE in params->flags (both host outputs NULL) renders
 * its sample range like any other call -- split over the GPUs -- but ADDS each GPU's slice to that GPU's resident running
 * sums and returns after enqueueing: no collective, no copy-back, so a C caller renders 4096 spp in pieces with no host-side
 * sums.  pt_node_snapshot joins the running sums (one all-reduce into a second buffer: the per-GPU sums stay partial and
 * keep growing), resolves by the samples rendered so far and copies back what is asked for (either may be NULL); *spp_out
 * = samples per pixel in the snapshot.  All pieces of a sequence share one frame shape.  pt_node_reset forgets the
 * resident sums (the next accumulating call starts from zero); so do pt_node_upload_scene and a plain pt_node_render. */
int pt_node_snapshot(pt_node* node, float* accum_rgba_host, uint32_t* bgra_host, uint64_t* spp_out);
int pt_node_reset(pt_node* node);
/* Replaces: `image.pixels = memory_alloc(...)` (src/main.cpp:385), which the reference does BEFORE its timed span
 * (src/main.cpp:412-481).  Allocates what a frame of this shape needs on every GPU -- the sums and device 0's pinned
 * image -- ahead of the first render.  Optional: pt_node_render / pt_node_snapshot allocate on demand. */
int pt_node_reserve(pt_node* node, uint32_t width, uint32_t height);
/* The one-call form SURVEY.md 8(b) proposes: create + upload + render + destroy. */
int pt_hip_render_node(const pt_scene_desc* scene, const pt_params* params, int n_gpus, float* accum_rgba_host);
/* Host-only helper: the slice of the samples [spp_begin, spp_end) that part `part` of `n_parts` renders -- contiguous,
 * disjoint, covering, sizes differing by at most one. */
void pt_spp_slice(uint32_t spp_begin, uint32_t spp_end, uint32_t n_parts, uint32_t part, uint32_t* out_begin, uint32_t* out_end);

/* Statistics of the last render on this context (synchronises with it). */
int pt_hip_last_stats(pt_ctx* ctx, pt_stats* out);
/* Host-only helper (no reference counterpart; touches no device): the sample-chunk schedule pt_hip_render_device uses
 * for `nspp` samples per pixel when a tile should hand out about `ratio_q16 / 65536` of the remaining samples per chunk
 * and never fewer than `min_spp`.  Writes up to `cap` chunk sizes to out_sizes and returns the number of chunks; the
 * sizes sum to nspp.  Exposed so that the schedule the kernels walk can be checked without a GPU. */
uint32_t pt_chunk_schedule(uint32_t nspp, uint32_t ratio_q16, uint32_t min_spp, uint32_t* out_sizes, uint32_t cap);

/* Host-only helper (no reference counterpart; touches no device): builds the BVH pt_hip_upload_scene would build for
 * `scene`, in both node formats, and checks what the parity argument rests on -- every sphere's swept box inside the
 * box of its leaf, every leaf reachable exactly once, every 32-byte grid node enclosing the 64-byte node it encodes.
 * out8: [0] nodes [1] depth [2] leaves [3] largest leaf [4] spheres found in leaves [5] spheres outside their leaf box
 * [6] grid boxes that fail to enclose their float box [7] references out of range.  Returns PT_OK when [4] equals the
 * sphere count and [5..7] are zero, PT_ERR_INVALID otherwise. */
int pt_bvh_selfcheck(const pt_scene_desc* scene, uint64_t out8[8]);

/* Diagnostics (no reference counterpart): the raw device counter words of the last render, for the tuning tools under
 * tools/.  [0] segments [1] wave iterations [2..4] phase cycles [5] groups searched [6] node visits [7] sphere tests
 * [8] units handed out [9] wave-level inner-step trips [10] wave-level leaf-step trips [11..12] their cycles (stamps build)
 * [13] paths that ended after their unit's tile was flushed (each adds its rgba to the image with four float atomics). */
int pt_hip_debug_counters(pt_ctx* ctx, uint64_t out16[16]);

/* Replaces: `clamp(pixelColour/SAMPLES_PER_PIXEL, 0, 1)` (src/main.cpp:358) and the per-pixel part of
 * write_image_to_bmp (src/file_io.h:29-45: gamma-2 sqrt, (u8)(255*c) truncation, 0xAARRGGBB) -- on the device, so
 * only 4 bytes per pixel need to cross PCIe.  Either output may be NULL.  Device pointers, asynchronous.
 * An output may also be PINNED HOST memory the device can address (hipHostMalloc / hipHostRegister): the kernel then
 * writes the finished image straight over PCIe -- 0.16 ms for 1920x1080, what the copy engine takes for the same bytes --
 * and the frame loop needs no hipMemcpyAsync (whose lazy set-up inside the runtime stalls a kernel in flight for ~8 ms,
 * twice per process: profiles/README.md); the host reads it after synchronising with the stream. */
int pt_hip_resolve_device(pt_ctx* ctx, const float* accum_rgba_dev, uint32_t n_pixels, uint32_t spp_total,
                          float* pixels_rgba_dev, uint32_t* bgra_dev, void* stream);

/* ---- host-side pieces that stay host-side in the reference ------------------------------------------- */
/* clamp(sum * (1.0f/spp), 0, 1) on the host (src/main.cpp:358, src/vectors.cpp:100-103, src/utils.h:17-20). */
int pt_resolve(const float* accum_rgba, uint32_t n_pixels, uint32_t spp_total, float* pixels_rgba);
/* src/file_io.h:29-45 per pixel. */
int pt_pack_bgra(const float* pixels_rgba, uint32_t n_pixels, uint32_t* bgra);
/* write_image_to_bmp (src/file_io.h:4-52): 58-byte header + top-down 32-bpp BGRA.  Byte-identical format. */
int pt_bmp_write(const char* path, const float* pixels_rgba, uint32_t width, uint32_t height);
int pt_bmp_write_bgra(const char* path, const uint32_t* bgra, uint32_t width, uint32_t height);
/* Header only (58 bytes), for tests. */
int pt_bmp_header(uint32_t width, uint32_t height, uint8_t* out58);

/* Scene builders: init_test_scene_{1,2,3} (src/scene_init.cpp:35-177) re-expressed on std::mt19937 so they
 * produce the scenes the reference produces under libstdc++; id 4 = the materials/DOF test scene (C2) and
 * id 5 = n_spheres random spheres (C5) defined in SURVEY.md 8(d).  The returned handle owns the arrays that
 * pt_scene_get_desc() points into. */
pt_scene* pt_scene_builtin(int scene_id, float aspect_ratio, uint32_t n_spheres_for_id5);
int pt_scene_get_desc(const pt_scene* scene, pt_scene_desc* out);
void pt_scene_free(pt_scene* scene);
/* Scene files.  The reference hard-codes its scene as a function call (src/main.cpp:396) and its
 * settings as macros (src/main.cpp:29-42); these two give the same World + Camera a text form (format: pt_world.cpp),
 * exact to the bit for every float.  pt_scene_load returns NULL on a missing or malformed file. */
int pt_scene_save(const pt_scene_desc* scene, const char* path);
pt_scene* pt_scene_load(const char* path);

/* ---- parity probes: the product's own device functions on caller-supplied inputs ---------------------- */
/* All pointers are HOST pointers; n records; synchronous.  They run the same device code as the render kernels
 * (same LDS staging, same search, same scatter) with the RNG replaced by an explicit tape of uniforms, so the
 * reference, the oracle and the GPU can be compared on IDENTICAL inputs (BASELINE.json north_star, gate 1). */

/* intersection_test(Ray,Sphere) and intersection_test(Ray,Plane) (src/geometry.cpp:39-81) record by record, as the
 * kernels evaluate them for a chosen hit.  rays 6*n; spheres 4*n (centre, radius); planes 4*n (normal, offset). */
int pt_hip_probe_primitives(pt_ctx* ctx, uint32_t n, const float* rays, const float* spheres, const float* planes,
                            float* out_t_sphere, float* out_t_plane);
/* Closest hit of planes then spheres as cast_ray orders it (src/main.cpp:180-216).
 * rays: 6*n (origin, unit dir); times: n; out_prim: sphere index >= 0, plane i -> -1-i, miss -> INT32_MIN. */
int pt_hip_probe_hits(pt_ctx* ctx, uint32_t n, const float* rays, const float* times, uint32_t flags,
                      float* out_t, int32_t* out_prim, float* out_point, float* out_normal);
/* Camera::get_ray (src/camera.cpp:57-79) for n (u,v) pairs; ray i draws from tape[ofs[i] .. ofs[i+1]). */
int pt_hip_probe_get_rays(pt_ctx* ctx, uint32_t n, const float* uv, const float* tape, const uint64_t* ofs, float* out_rays);
/* One scatter event (src/main.cpp:219-298) per record.  out_event: 0 scattered, 3 absorbed, 4 NONE material;
 * bit 31 set = tape overrun. */
int pt_hip_probe_scatter(pt_ctx* ctx, uint32_t n, const float* dirs, const float* points, const float* normals,
                         const uint32_t* mat_type, const float* mat_rgba, const float* mat_param,
                         const float* tape32, const uint64_t* ofs32, const double* tape64, const uint64_t* ofs64,
                         float* out_dir, uint32_t* out_event);
/* cast_ray (src/main.cpp:171-307) for n rays on per-ray tapes. out_rgba 4*n; out_segments n (may be NULL). */
int pt_hip_probe_paths(pt_ctx* ctx, uint32_t n, const float* rays, const float* times, uint32_t max_depth, uint32_t flags,
                       const float* tape32, const uint64_t* ofs32, const double* tape64, const uint64_t* ofs64,
                       float* out_rgba, uint32_t* out_segments);
/* One sample of run_thread_batch's loop (src/main.cpp:349-355) per record on the Philox stream of
 * (params->seed, pixels[i], samples[i]); out_rgba 4*n; out_segments n (may be NULL). */
int pt_hip_probe_samples(pt_ctx* ctx, const pt_params* params, uint32_t n, const uint32_t* pixels, const uint32_t* samples,
                         float* out_rgba, uint32_t* out_segments);
/* First n words of the Philox stream of (seed, pixel, sample), generated on the device. */
int pt_hip_probe_philox(pt_ctx* ctx, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t* out_words);

#ifdef __cplusplus
}
#endif
#endif /* PT_HIP_H */
