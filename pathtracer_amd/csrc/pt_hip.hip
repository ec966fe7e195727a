// pt_hip.hip -- kernels and C-ABI of libpt_hip.so (gfx950 only; see include/pt_hip.h for the boundary).
//
// Kernels
//   k_render{,_w6,_w7}<MOVING,MODE>  the hot path (one body, three register budgets).  Persistent waves pull work units
//                                -- (8x8 pixel tile, sample chunk) -- from a global cursor; inside a unit the lanes pull
//                                (pixel, sample) items from a wave-local pool, so a lane whose path ended starts its
//                                next sample in the same trip (path regeneration).  MODE picks the sphere query: own BVH
//                                staged in LDS (default) or read from HBM (big scenes), packed brute force over the
//                                LDS-resident scene, or the strict linear search (tests).
//   k_fold_chunks                adds the per-chunk partial images in chunk order.
//   k_probe_*                    the same device functions on caller-supplied records (parity harness).
//   k_resolve                    clamp(mean) + gamma-2 + 8-bit BGRA pack.
// Host code below the kernels implements include/pt_hip.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <cmath>
#include <atomic>
#include <new>
#include <string>
#include <vector>

#include "../../include/pt_hip.h"
#include "pt_guard.h"
#include "pt_device.h"
#include "pt_host.h"
#include "pt_bvh.h"

using namespace ptd;

// ================================================================================================ kernel arguments
struct SceneArgs
{
    const float4* blob; // HBM copy of the LDS image
    uint32_t blob_f4;   // its length in float4
    uint32_t off_velp, off_groups, off_smat, off_planes, off_pmat, off_cam; // off_cam: the folded DeviceCamera (4 float4)
    uint32_t ns, npairs, ngroups, np, idx_mask;
    float t0, t1;
    BvhView bvh;        // MODE_BVH: nodes / spheres / materials in HBM.  MODE_BVH_LDS: only root_ref / n_spheres are used,
    uint32_t bvh_off[4]; //   the arrays sit inside the blob at these float4 offsets (nodes, sph, vel, mat)
    uint32_t stack_entries; // per-lane traversal stack capacity carved from LDS (BVH modes)
    uint32_t leaf_batch;    // lanes that must wait at a leaf before the wave runs the leaf code (BVH modes)
    uint32_t refill_min;
    uint32_t exit_lanes;    // a trip's traversal phase is left once fewer lanes than this still traverse and one is ready to shade
    RefView ref;            // MODE_REF: the reference-topology tree (PT_FLAG_REFERENCE_QUERY)
    uint32_t carry_max;     // CARRY kernels: a unit whose pool is empty is left once at most this many of its paths are still under way
};

// LDS carve-up behind the scene image: per wave, [64 pixels x rgba] accumulators (render kernel) and, in the BVH modes,
// a (tree depth + 1) x 64 traversal stack.
template <int MODE> __device__ __forceinline__ QueryCtx make_query(const SceneArgs& a, const float4* smem, uint32_t* stackBase)
{
    QueryCtx q;
    q.bvh = a.bvh;
    if (MODE == MODE_BVH_LDS || MODE == MODE_FEW)
    {
        q.bvh.nodes = smem + a.bvh_off[0]; q.bvh.sph = smem + a.bvh_off[1]; q.bvh.vel = smem + a.bvh_off[2];
        if (a.bvh_off[3] != 0xFFFFFFFFu) q.bvh.mat = smem + a.bvh_off[3]; // else: materials stay in HBM (touched once per hit)
    }
    q.ref = a.ref;
    q.ovf = nullptr; q.stackCap = a.stack_entries;
    // MODE_BVH_LDS: the wave's stack area (stack_entries x 64 dwords) holds 2 x stack_entries levels of 16-bit entries
    q.stack16 = reinterpret_cast<int16_t*>(stackBase - (threadIdx.x & 63u)) + (threadIdx.x & 63u); q.laneTests = 0;
    q.stack = stackBase; q.searched = 0; q.nodeVisits = 0; q.sphereTests = 0; q.leafBatch = a.leaf_batch;
#ifdef PT_STAMPS
    q.innerIters = 0; q.leafIters = 0; q.cycInner = 0; q.cycLeaf = 0; q.leafSlotIters = 0; q.leafExactIters = 0;
#endif
    return q;
}

__device__ __forceinline__ SceneView make_view(const float4* base, const SceneArgs& a)
{
    SceneView s;
    s.pairs = base; s.velp = base + a.off_velp; s.groups = base + a.off_groups; s.smat = base + a.off_smat;
    s.planes = base + a.off_planes; s.pmat = base + a.off_pmat;
    s.ns = a.ns; s.npairs = a.npairs; s.ngroups = a.ngroups; s.np = a.np; s.idx_mask = a.idx_mask;
    return s;
}

__device__ __forceinline__ void stage_scene(float4* smem, const SceneArgs& a)
{
    for (uint32_t i = threadIdx.x; i < a.blob_f4; i += blockDim.x) smem[i] = a.blob[i];
}

struct RenderArgs
{
    SceneArgs sc;
    uint32_t width, height, tiles_x, ntiles, tile_stride;
    uint32_t spp_begin, nspp, max_depth;
    uint64_t seed;
    uint32_t nchunks, nunits;       // work units = ntiles x nchunks; chunk c covers the samples chunk_range() gives it
    uint32_t chunk_ratio_q16, chunk_min; // the guided schedule's two parameters (chunk_size)
    float* out_sum;                 // width*height*4
    float* partial;                 // PT_FLAG_REPRODUCIBLE and nchunks > 1: [nchunks][height][width][4] per-chunk sums, folded in chunk order by k_fold_chunks
    uint32_t chunk_stride;          // pixels between the images of two chunks: width*height with `partial`, else 0
    uint32_t accumulate;            // nchunks > 1 without `partial`: every unit ADDS its tile to out_sum (zeroed before the launch)
    unsigned long long* stats;      // [0] segments ... [8] low word: the unit cursor the waves pull from
};

// Guided schedule of the sample chunks: big chunks first, small ones last.  Every unit has fixed costs (this walk, the
// tile flush, a partial image to fold, the cursor's round trip), so few units are wanted; the launch's tail is as long
// as its last units, so small ones are wanted at the end.  A chunk takes the share `ratio` of the samples still to do (at most half, at least
// `minSpp`; a remainder below 2 x minSpp goes in one piece).  Host and device run the same recurrence.
__host__ __device__ inline uint32_t chunk_size(uint32_t remaining, uint32_t ratio_q16, uint32_t minSpp)
{
    if (minSpp == 0u || remaining < 2u * minSpp) return remaining; // (minSpp 0 cannot advance: one piece)
    uint32_t u = (uint32_t)(((unsigned long long)remaining * ratio_q16) >> 16);
    if (u > remaining / 2u) u = remaining / 2u;
    if (u < minSpp) u = minSpp;
    return u;
}

constexpr uint32_t kStatWords = 16;        // stats buffer: totals [0..7] counters, [8] unit cursor, [9..12] stamps; then one such record per wave
constexpr uint32_t kWavesPerBlock = 4;     // LDS modes: four waves share one staged copy of the scene
constexpr uint32_t kWavesPerBlockBvh = 1;  // tree in HBM: nothing to share, one wave per workgroup packs best

// Scratch of pt_hip_render_device_static, the entry point that allocates nothing: the statistics words -- the unit cursor
// the waves pull from and one record per wave (a 256-CU chip holds 8192 waves) -- and room for the per-chunk partial
// images of a PT_FLAG_REPRODUCIBLE frame.  One copy per GPU, held until the process ends; in host code the name is a
// stand-in, the array's address on a device comes from hipGetSymbolAddress (pt_hip_create) and is kept in the context.
constexpr uint32_t kStaticSlots = 8192;
constexpr size_t kStaticStatsBytes = (((size_t)kStatWords * (1u + kStaticSlots) * sizeof(unsigned long long)) + 255u) / 256u * 256u;
constexpr size_t kStaticPartialBytes = 32u << 20;
static __device__ __align__(256) unsigned char g_static_scratch[kStaticStatsBytes + kStaticPartialBytes];

// ================================================================================================ the hot kernel
template <bool MOVING, int MODE, bool RESUME, bool CARRY>
__device__ __forceinline__ void render_body(const RenderArgs& a)
{
    extern __shared__ float4 smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    stage_scene(smem, a.sc);
    float* acc = reinterpret_cast<float*>(smem + a.sc.blob_f4) + wave * 256u; // [64 pixels][rgba] of this wave's tile
    uint32_t* stack = reinterpret_cast<uint32_t*>(smem + a.sc.blob_f4 + nwaves * 64u) + wave * (a.sc.stack_entries * 64u) + lane;
    QueryCtx qc = make_query<MODE>(a.sc, smem, stack);
    uint32_t ovfStack[MODE == MODE_BVH_Q4 ? kOvfEntries : 1]; // levels of the traversal stack beyond its LDS part (bvh_run_wide)
    qc.ovf = ovfStack;
    __syncthreads();
    const SceneView s = make_view(smem, a.sc);
    PhiloxRng rng; rng.init(a.seed, 0, 0);
    uint32_t segs = 0, iters = 0, ncarried = 0;
#ifdef PT_STAMPS
    unsigned long long cyc_refill = 0, cyc_hit = 0, cyc_shade = 0;
#define PT_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
    unsigned long long ts0, ts1, ts2, ts3;
#else
#define PT_STAMP(var) do { } while (0)
#endif

    // Persistent waves: the grid is sized to what the chip holds at once and every wave pulls work units -- (tile,
    // sample chunk) pairs -- from one global cursor until it runs dry.  A workgroup's staged copy of the scene then
    // serves many tiles, and no wave's slot idles behind a slower sibling.  Every wave reaches unit >= nunits: the
    // cursor only grows.
    // (A wave's first unit is its own index -- no 7000 atomics on one address at the start of the launch; the cursor
    // hands out the units behind those.)
    const uint32_t waveId = pt_uniform(blockIdx.x * nwaves + wave), cursorBase = gridDim.x * nwaves;
    // Per-lane path state.  It outlives the unit: when a unit's pool runs dry the paths still under way are CARRIED into
    // the next unit's trips (no drain with a handful of lanes); such a path adds its result straight to its own unit's
    // pixel in HBM (kCarried | pixel index in p) instead of the tile accumulator, which by then belongs to the next unit.
    bool alive = false;
    // The live ray is kept as six scalars: as a struct it stays an un-promoted alloca (scratch traffic in the loop).
    float rox = 0, roy = 0, roz = 0, rdx = 0, rdy = 0, rdz = 1;
    V4 T = v4(1, 1, 1, 1);
    float time = 0.0f;
    uint32_t depth = 0, p = 0;
    constexpr uint32_t kCarried = 0x80000000u;
    float* const image = a.partial ? a.partial : a.out_sum;
    // BVH kernels: a lane's sphere query may span several trips (bvh_run's early exit); ts.cur != kBvhDone marks a
    // lane whose query is under way.
    constexpr bool kBvh = RESUME && mode_is_bvh(MODE);
    constexpr uint32_t kPendingShade = 0x40000000u; // in ts.pk: the finished query's hit has not got its scatter direction yet
    TravState ts; ts.cur = kBvhDone; ts.best = 0.0f; ts.bestSlot = -1; ts.pk = 0;
    for (uint32_t unit = waveId;;)
    {
        const bool haveUnit = unit < a.nunits;     // false: the launch's last trips, carried paths only
        if (!haveUnit && pt_ballot(alive) == 0ull) break;
        const uint32_t chunk = haveUnit ? unit / a.ntiles : 0u, slot = unit - chunk * a.ntiles;
        // Neighbouring tiles cost alike (sky rows are cheap, the sphere field is not): a multiplicative shuffle of the
        // tile order mixes cheap and expensive tiles over the whole launch.
        const uint32_t tile = haveUnit ? (uint32_t)(((unsigned long long)slot * a.tile_stride) % a.ntiles) : 0u;
        const uint32_t tx = tile % a.tiles_x, ty = tile / a.tiles_x;
        uint32_t s0 = 0, s1 = 0; // samples [s0, s1) of the call: walk the schedule up to this chunk (scalar work, once per unit)
        if (haveUnit)
            for (uint32_t c = 0;; ++c)
            {
                s1 = s0 + chunk_size(a.nspp - s0, a.chunk_ratio_q16, a.chunk_min);
                if (c == chunk) break;
                s0 = s1;
            }
        acc[lane] = 0.0f; acc[lane + 64] = 0.0f; acc[lane + 128] = 0.0f; acc[lane + 192] = 0.0f;

        const uint32_t total = 64u * (s1 - s0); // work items of this unit: item i = (pixel i&63 of the tile, sample s0 + (i>>6))
        uint32_t next = 0;                      // wave-uniform cursor into the pool
        bool retired = false;

        for (;;)
        {
            PT_STAMP(ts0);
            // ---- refill: every lane without a live path takes the next work item (main.cpp:347-354)
            const bool want = !alive && !retired;
            const unsigned long long need = pt_ballot(want);
            if (need && ((uint32_t)__popcll(need) >= a.sc.refill_min || next >= total || pt_ballot(alive) == 0ull))
            {
                if (want)
                {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                    const uint32_t item = next + rank;
                    if (item < total)
                    {
                        p = item & 63u;
                        const uint32_t px = tx * 8u + (p & 7u), py = ty * 8u + (p >> 3);
                        if (px < a.width && py < a.height) // ragged edge tiles: the item is skipped
                        {
                            rng.init(a.seed, py * a.width + px, a.spp_begin + s0 + (item >> 6));
                            const Ray r0 = begin_sample(*reinterpret_cast<const DeviceCamera*>(smem + a.sc.off_cam), px, py, a.width, a.height, a.sc.t0, a.sc.t1, rng, &time);
                            rox = r0.o.x; roy = r0.o.y; roz = r0.o.z; rdx = r0.d.x; rdy = r0.d.y; rdz = r0.d.z;
                            T = v4(1, 1, 1, 1);
                            depth = a.max_depth;
                            alive = depth != 0;
                            if (!alive) atomicAdd(&acc[p * 4u + 3u], 1.0f); // maxDepth 0 -> BLACK (0,0,0,1) (main.cpp:175-176)
                        }
                    }
                    else
                        retired = true;
                }
                next += (uint32_t)__popcll(need);
            }
            if (CARRY && haveUnit && next >= total && (uint32_t)__popcll(pt_ballot(alive)) <= a.sc.carry_max)
            {
                // The pool is empty: whoever is still under way is carried (its tile pixel becomes an index into the image).
                if (alive && !(p & kCarried))
                    p = kCarried | (chunk * a.chunk_stride + (ty * 8u + (p >> 3)) * a.width + tx * 8u + (p & 7u));
                break;
            }
            if (pt_ballot(alive) == 0ull)
            {
                if (CARRY ? !haveUnit : next >= total) break;
                continue;
            }
            // ---- one segment for every lane (lanes without a live path ride along on stale data)
            ++iters;
            PT_STAMP(ts1);
            Ray ray; ray.o = v3(rox, roy, roz); ray.d = v3(rdx, rdy, rdz);
            Hit h;
            bool ready = alive;
            if constexpr (kBvh)
            {
                // a new segment: planes first, then the tree from its root (a lane without a path parks on the sentinel);
                // a lane whose hit still waits for its scatter direction (kPendingShade) keeps the finished query
                if (ts.cur == kBvhDone && !(ts.pk & kPendingShade))
                {
                    int32_t planeIdx = -1;
                    const float planeT = closest_plane(s, ray, &planeIdx);
                    trav_begin<MODE == MODE_BVH_LDS>(qc.bvh, planeT, planeIdx, alive, qc, ts);
                }
                run_traversal<MOVING, MODE>(qc.bvh, ray, time, alive, qc, ts, a.sc.exit_lanes);
                ready = alive && ts.cur == kBvhDone;
                if (ready) h = make_hit_bvh<MOVING, MODE == MODE_BVH_LDS>(s, ray, time, qc, ts);
            }
            else
                h = closest_hit<MOVING, MODE>(s, ray, time, alive, qc);
            PT_STAMP(ts2);
            if (ready)
            {
                V4 terminal;
                bool pending = false;
                const bool goes_on = shade_step(ray, h, T, rng, &terminal, kBvh && kScatterAttempts != 0u ? &pending : nullptr);
                rng.park(); // the next scatter aligns to a fresh block anyway: drop the buffered words from the live state
                if constexpr (kBvh)
                {
                    // The unit-sphere rejection loop is left after kScatterAttempts Philox blocks: a lane still without a point
                    // keeps its hit (the traversal state is live anyway) and draws on in the wave's next trip.
                    ts.pk = pending ? (ts.pk | kPendingShade) : (ts.pk & ~kPendingShade);
                }
                if (!pending)
                {
                ++segs;
                rox = ray.o.x; roy = ray.o.y; roz = ray.o.z; rdx = ray.d.x; rdy = ray.d.y; rdz = ray.d.z;
                bool ended = !goes_on;
                if (goes_on && --depth == 0) { terminal = v4(0, 0, 0, 1); ended = true; } // main.cpp:175-176
                if (ended)
                {
                    const V4 c = hadamard(T, terminal);
                    if (p & kCarried)
                    {
                        ncarried += (uint32_t)__popcll(pt_ballot(true)); // (wave-level tally: a scalar add)
                        float* dst = image + 4u * (size_t)(p & ~kCarried);
                        unsafeAtomicAdd(dst, c.x); unsafeAtomicAdd(dst + 1, c.y); unsafeAtomicAdd(dst + 2, c.z); unsafeAtomicAdd(dst + 3, c.w);
                    }
                    else
                    {
                        atomicAdd(&acc[p * 4u + 0u], c.x); atomicAdd(&acc[p * 4u + 1u], c.y);
                        atomicAdd(&acc[p * 4u + 2u], c.z); atomicAdd(&acc[p * 4u + 3u], c.w);
                    }
                    alive = false;
                }
                } // !pending
            }
#ifdef PT_STAMPS
            PT_STAMP(ts3);
            cyc_refill += ts1 - ts0; cyc_hit += ts2 - ts1; cyc_shade += ts3 - ts2;
#endif
        }

        // ---- the unit's tile goes to the image (plus the carried paths' atomics: the only HBM traffic of the render
        // besides staging the scene).
        if (haveUnit)
        {
            if (a.accumulate)
            {
                // Several chunks, one image: the tile is ADDED to it (the host zeroed it).  One instruction covers two
                // tile rows of 8 pixels x rgba = 2 x 128 contiguous bytes (the shape float atomics run at full rate in,
                // MI355X_MICROARCH.md "Global float atomics"); lane l of instruction j holds acc[64 j + l].  Atomic
                // adds commute: no ordering against the carried paths' adds, nothing to wait for.
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j)
                {
                    const uint32_t py = ty * 8u + 2u * j + (lane >> 5), px = tx * 8u + ((lane & 31u) >> 2);
                    if (px < a.width && py < a.height)
                        unsafeAtomicAdd(image + 4u * ((size_t)py * a.width + tx * 8u) + (lane & 31u), acc[j * 64u + lane]);
                }
            }
            else
            {
                // one 16-byte store per pixel and unit; it must have landed before a carried path of this unit adds to it
                const uint32_t px = tx * 8u + (lane & 7u), py = ty * 8u + (lane >> 3);
                if (px < a.width && py < a.height)
                {
                    float4 v = make_float4(acc[lane * 4u], acc[lane * 4u + 1u], acc[lane * 4u + 2u], acc[lane * 4u + 3u]);
                    reinterpret_cast<float4*>(image)[(size_t)chunk * a.chunk_stride + (size_t)py * a.width + px] = v;
                }
                __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
            }
            if (lane == 0) unit = cursorBase + atomicAdd(reinterpret_cast<unsigned int*>(a.stats + 8), 1u);
            unit = pt_uniform(unit);
        }
    } // units
    // ---- statistics: every wave leaves its tallies in a slot of its own (plain stores; k_sum_stats adds the slots up).
    // Atomics on shared counters would be ~36 000 same-address operations in the launch's last microseconds: measured,
    // they were 0.7 ms -- the whole fixed cost of a small launch.
    for (int o = 32; o > 0; o >>= 1) { segs += __shfl_xor(segs, o); qc.laneTests += __shfl_xor(qc.laneTests, o); }
    if (MODE == MODE_BVH_LDS) qc.sphereTests = qc.laneTests;
    if (lane == 0)
    {
        unsigned long long* slot = a.stats + kStatWords * (1u + waveId);
        slot[0] = segs; slot[1] = iters; slot[5] = qc.searched; slot[13] = ncarried;
        slot[6] = (mode_is_bvh(MODE) || MODE == MODE_REF) ? qc.nodeVisits : 0u; // both tallies are wave-level already
        slot[7] = (mode_is_bvh(MODE) || MODE == MODE_REF) ? qc.sphereTests : 0u;
#ifdef PT_STAMPS
        slot[2] = cyc_refill; slot[3] = cyc_hit; slot[4] = cyc_shade;
        slot[9] = qc.innerIters; slot[10] = qc.leafIters; slot[11] = qc.cycInner; slot[12] = qc.cycLeaf;
        slot[14] = qc.leafSlotIters; slot[15] = qc.leafExactIters;
#endif
    }
}

// totals[w] += sum over the wave slots (word 8, the unit cursor, is left alone; the totals were zeroed before the
// render).  A workgroup takes 256 slots: thread (g, w) adds word w of 16 of them (independent loads), LDS folds the 16
// groups, 15 atomics per workgroup -- a few hundred in all.
__global__ void __launch_bounds__(256) k_sum_stats(unsigned long long* stats, uint32_t nslots)
{
    __shared__ unsigned long long part[256];
    const uint32_t w = threadIdx.x & 15u, g = threadIdx.x >> 4;
    unsigned long long sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16u; ++j)
    {
        const uint32_t slot = blockIdx.x * 256u + g + 16u * j;
        if (slot < nslots) sum += stats[kStatWords * (1u + slot) + w];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x < 16u && threadIdx.x != 8u)
    {
        unsigned long long t = 0;
        for (uint32_t k = 0; k < 16u; ++k) t += part[k * 16u + threadIdx.x];
        if (t) atomicAdd(stats + threadIdx.x, t);
    }
}

// The kernel proper, in three register budgets.  More resident waves hide the LDS / L1 latency of the traversal's
// dependent loads; the measured optimum (book-1 1080p, tools/ab.py) is 7 waves per SIMD (<= 72 VGPRs) for the tree in
// LDS and 6 (<= 80) for the tree in HBM and for the brute-force search; the strict search is a test oracle and is left
// to the compiler.
// RESUME: sphere queries that may span trips (bvh_run); scenes with a handful of spheres do without (the carried state
// costs them 3 %).
// CARRY: paths still under way when a unit's pool runs dry ride along in the next unit (false: PT_FLAG_REPRODUCIBLE).
template <bool MOVING, int MODE, bool CARRY> __global__ void __launch_bounds__(512) k_render(RenderArgs a) { render_body<MOVING, MODE, false, CARRY>(a); }
#ifdef PT_NO_EU
#define PT_EU(n)
#else
#define PT_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif
#ifndef PT_HBM_WAVES
#define PT_HBM_WAVES 7 // (6 until round 3: C5 1 675 -> 1 705 Msamples/s with 8 stack levels in LDS, scene 2 +1 %)  experiments: make variant DEFS=-DPT_HBM_WAVES=6
#endif
// (the name is historical: these kernels ran 6 waves per SIMD until round 3)
template <bool MOVING, int MODE, bool RESUME, bool CARRY> __global__ void __launch_bounds__(512) PT_EU(PT_HBM_WAVES) k_render_w6(RenderArgs a) { render_body<MOVING, MODE, RESUME, CARRY>(a); }
#ifndef PT_LDS_WAVES
#define PT_LDS_WAVES 7 // experiments: make variant DEFS=-DPT_LDS_WAVES=6
#endif
template <bool MOVING, int MODE, bool RESUME, bool CARRY> __global__ void __launch_bounds__(512) PT_EU(PT_LDS_WAVES) k_render_w7(RenderArgs a) { render_body<MOVING, MODE, RESUME, CARRY>(a); }
using RenderKernel = void (*)(RenderArgs);
template <bool MOVING, int MODE, bool CARRY> RenderKernel render_kernel_for(bool resume)
{
    if constexpr (MODE == MODE_BVH_LDS) return resume ? &k_render_w7<MOVING, MODE, true, CARRY> : &k_render_w7<MOVING, MODE, false, CARRY>;
    else if constexpr (mode_in_hbm(MODE)) return resume ? &k_render_w6<MOVING, MODE, true, CARRY> : &k_render_w6<MOVING, MODE, false, CARRY>;
    else if constexpr (MODE == MODE_LDS_STRICT || MODE == MODE_REF) return &k_render<MOVING, MODE, CARRY>; // test / compatibility modes: left to the compiler
    else if constexpr (MODE == MODE_FEW) return &k_render_w7<MOVING, MODE, false, CARRY>; // (8 waves per SIMD, <= 64 VGPRs: measured equal)
    else return &k_render_w6<MOVING, MODE, false, CARRY>;
}
template <bool MOVING, bool CARRY> RenderKernel render_kernel_mc(int mode, bool resume)
{
    if (mode == MODE_BVH_LDS) return render_kernel_for<MOVING, MODE_BVH_LDS, CARRY>(resume);
    if (mode == MODE_BVH) return render_kernel_for<MOVING, MODE_BVH, CARRY>(resume);
    if (mode == MODE_BVH_Q) return render_kernel_for<MOVING, MODE_BVH_Q, CARRY>(resume);
    if (mode == MODE_BVH_Q4) return render_kernel_for<MOVING, MODE_BVH_Q4, CARRY>(resume);
    if (mode == MODE_LDS_STRICT) return render_kernel_for<MOVING, MODE_LDS_STRICT, CARRY>(resume);
    if (mode == MODE_REF) return render_kernel_for<MOVING, MODE_REF, CARRY>(resume);
    if (mode == MODE_FEW) return render_kernel_for<MOVING, MODE_FEW, CARRY>(resume);
    return render_kernel_for<MOVING, MODE_LDS_FAST, CARRY>(resume);
}
static RenderKernel render_kernel(bool moving, int mode, bool resume, bool carry)
{
    if (moving) return carry ? render_kernel_mc<true, true>(mode, resume) : render_kernel_mc<true, false>(mode, resume);
    return carry ? render_kernel_mc<false, true>(mode, resume) : render_kernel_mc<false, false>(mode, resume);
}

// ================================================================================================ probes
struct ProbeHitsArgs { SceneArgs sc; uint32_t n; const float* rays; const float* times; float* t; int32_t* prim; float* point; float* normal; };
template <bool MOVING, int MODE> __global__ void __launch_bounds__(256) k_probe_hits(ProbeHitsArgs a)
{
    extern __shared__ float4 smem[];
    stage_scene(smem, a.sc);
    QueryCtx qc = make_query<MODE>(a.sc, smem, reinterpret_cast<uint32_t*>(smem + a.sc.blob_f4) + (threadIdx.x >> 6) * (a.sc.stack_entries * 64u) + (threadIdx.x & 63u));
    uint32_t ovfStack[MODE == MODE_BVH_Q4 ? kOvfEntries : 1]; qc.ovf = ovfStack;
    __syncthreads();
    const SceneView s = make_view(smem, a.sc);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    Ray r; r.o = v3(a.rays[6*i], a.rays[6*i+1], a.rays[6*i+2]); r.d = v3(a.rays[6*i+3], a.rays[6*i+4], a.rays[6*i+5]);
    Hit h = closest_hit<MOVING, MODE>(s, r, a.times[i], true, qc);
    a.t[i] = h.t; a.prim[i] = h.prim;
    if (h.prim == kPrimMiss) { h.point = v3(0, 0, 0); h.normal = v3(0, 0, 0); }   // (make_hit leaves them undefined on a miss)
    a.point[3*i] = h.point.x; a.point[3*i+1] = h.point.y; a.point[3*i+2] = h.point.z;
    a.normal[3*i] = h.normal.x; a.normal[3*i+1] = h.normal.y; a.normal[3*i+2] = h.normal.z;
}

__global__ void __launch_bounds__(256) k_probe_primitives(uint32_t n, const float* rays, const float* spheres, const float* planes, float* ts, float* tp)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r; r.o = v3(rays[6*i], rays[6*i+1], rays[6*i+2]); r.d = v3(rays[6*i+3], rays[6*i+4], rays[6*i+5]);
    const float radius = spheres[4*i+3];
    ts[i] = intersect_sphere_exact(r, v3(spheres[4*i], spheres[4*i+1], spheres[4*i+2]), radius * radius); // geometry.cpp:47
    tp[i] = intersect_plane_exact(r, v3(planes[4*i], planes[4*i+1], planes[4*i+2]), planes[4*i+3]);
}

struct ProbeRaysArgs { DeviceCamera cam; uint32_t n; const float* uv; const float* tape; const unsigned long long* ofs; float* rays; };
__global__ void __launch_bounds__(256) k_probe_get_rays(ProbeRaysArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    TapeRng rng; rng.init(a.tape + a.ofs[i], (uint32_t)(a.ofs[i+1] - a.ofs[i]), nullptr, 0);
    Ray r = get_ray(a.cam, a.uv[2*i], a.uv[2*i+1], rng);
    float* o = a.rays + 6 * (size_t)i;
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
}

struct ProbeScatterArgs
{
    uint32_t n; const float* dirs; const float* points; const float* normals; const uint32_t* type; const float* rgba; const float* param;
    const float* tape32; const unsigned long long* ofs32; const double* tape64; const unsigned long long* ofs64; float* out_dir; uint32_t* out_event;
};
__global__ void __launch_bounds__(256) k_probe_scatter(ProbeScatterArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    Ray r; r.o = v3(0, 0, 0); r.d = v3(a.dirs[3*i], a.dirs[3*i+1], a.dirs[3*i+2]);
    Hit h; h.t = 1.0f; h.prim = 0;
    h.point = v3(a.points[3*i], a.points[3*i+1], a.points[3*i+2]); h.normal = v3(a.normals[3*i], a.normals[3*i+1], a.normals[3*i+2]);
    h.m.type = a.type[i]; h.m.colour = v4(a.rgba[4*i], a.rgba[4*i+1], a.rgba[4*i+2], a.rgba[4*i+3]); h.m.roughness = a.param[2*i]; h.m.n = a.param[2*i+1];
    TapeRng rng; rng.init(a.tape32 + a.ofs32[i], (uint32_t)(a.ofs32[i+1] - a.ofs32[i]), a.tape64 + a.ofs64[i], (uint32_t)(a.ofs64[i+1] - a.ofs64[i]));
    Ray next; next.o = v3(0, 0, 0); next.d = v3(0, 0, 0); uint32_t ev;
    scatter(r, h, rng, &next, &ev);
    a.out_dir[3*i] = next.d.x; a.out_dir[3*i+1] = next.d.y; a.out_dir[3*i+2] = next.d.z;
    a.out_event[i] = ev | (rng.overrun() ? 0x80000000u : 0u);
}

struct ProbePathsArgs
{
    SceneArgs sc; uint32_t n, max_depth; const float* rays; const float* times;
    const float* tape32; const unsigned long long* ofs32; const double* tape64; const unsigned long long* ofs64; float* rgba; uint32_t* segs;
};
template <bool MOVING, int MODE> __global__ void __launch_bounds__(256) k_probe_paths(ProbePathsArgs a)
{
    extern __shared__ float4 smem[];
    stage_scene(smem, a.sc);
    QueryCtx qc = make_query<MODE>(a.sc, smem, reinterpret_cast<uint32_t*>(smem + a.sc.blob_f4) + (threadIdx.x >> 6) * (a.sc.stack_entries * 64u) + (threadIdx.x & 63u));
    uint32_t ovfStack[MODE == MODE_BVH_Q4 ? kOvfEntries : 1]; qc.ovf = ovfStack;
    __syncthreads();
    const SceneView s = make_view(smem, a.sc);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    Ray r; r.o = v3(a.rays[6*i], a.rays[6*i+1], a.rays[6*i+2]); r.d = v3(a.rays[6*i+3], a.rays[6*i+4], a.rays[6*i+5]);
    TapeRng rng; rng.init(a.tape32 + a.ofs32[i], (uint32_t)(a.ofs32[i+1] - a.ofs32[i]), a.tape64 + a.ofs64[i], (uint32_t)(a.ofs64[i+1] - a.ofs64[i]));
    uint32_t n = 0;
    V4 c = cast_ray<MOVING, MODE>(s, qc, r, a.max_depth, a.times[i], rng, &n);
    a.rgba[4*i] = c.x; a.rgba[4*i+1] = c.y; a.rgba[4*i+2] = c.z; a.rgba[4*i+3] = c.w;
    if (a.segs) a.segs[i] = n | (rng.overrun() ? 0x80000000u : 0u);
}

struct ProbeSamplesArgs { SceneArgs sc; uint32_t n, width, height, max_depth; uint64_t seed; const uint32_t* pixels; const uint32_t* samples; float* rgba; uint32_t* segs; };
template <bool MOVING, int MODE> __global__ void __launch_bounds__(256) k_probe_samples(ProbeSamplesArgs a)
{
    extern __shared__ float4 smem[];
    stage_scene(smem, a.sc);
    QueryCtx qc = make_query<MODE>(a.sc, smem, reinterpret_cast<uint32_t*>(smem + a.sc.blob_f4) + (threadIdx.x >> 6) * (a.sc.stack_entries * 64u) + (threadIdx.x & 63u));
    uint32_t ovfStack[MODE == MODE_BVH_Q4 ? kOvfEntries : 1]; qc.ovf = ovfStack;
    __syncthreads();
    const SceneView s = make_view(smem, a.sc);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t pixel = a.pixels[i];
    PhiloxRng rng; rng.init(a.seed, pixel, a.samples[i]);
    float time;
    Ray r = begin_sample(*reinterpret_cast<const DeviceCamera*>(smem + a.sc.off_cam), pixel % a.width, pixel / a.width, a.width, a.height, a.sc.t0, a.sc.t1, rng, &time);
    uint32_t n = 0;
    V4 c = cast_ray<MOVING, MODE>(s, qc, r, a.max_depth, time, rng, &n);
    a.rgba[4*i] = c.x; a.rgba[4*i+1] = c.y; a.rgba[4*i+2] = c.z; a.rgba[4*i+3] = c.w;
    if (a.segs) a.segs[i] = n;
}

__global__ void k_probe_philox(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t* out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    PhiloxRng rng; rng.init(seed, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.word();
}

// ================================================================================================ resolve
// main.cpp:358 (mean, clamp) and file_io.h:29-45 (gamma 2, truncating quantise, 0xAARRGGBB).
__device__ __forceinline__ float clamp01(float v) { if (v < 0.0f) return 0.0f; else if (v > 1.0f) return 1.0f; return v; } // utils.h:8-15
__global__ void __launch_bounds__(256) k_resolve(const float4* __restrict__ sum, uint32_t n, float inv_spp, float4* __restrict__ pixels, uint32_t* __restrict__ bgra)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        float4 v = sum[i];
        v.x = clamp01(v.x * inv_spp); v.y = clamp01(v.y * inv_spp); v.z = clamp01(v.z * inv_spp); v.w = clamp01(v.w * inv_spp);
        if (pixels) pixels[i] = v;
        if (bgra)
        {
            uint32_t r = (uint8_t)(255.0f * sqrtf(v.x)), g = (uint8_t)(255.0f * sqrtf(v.y)), b = (uint8_t)(255.0f * sqrtf(v.z)), al = (uint8_t)(255.0f * sqrtf(v.w));
            bgra[i] = (al << 24) | (r << 16) | (g << 8) | b;
        }
    }
}

// Sum of the per-chunk partial images in chunk order (a fixed order: the frame does not depend on which wave ran which unit).
// add: the frame's sums join what `out` holds (PT_FLAG_ACCUMULATE) -- the frame is folded first, then added once.
__global__ void __launch_bounds__(256) k_fold_chunks(const float4* __restrict__ partial, uint32_t nchunks, uint32_t n, float4* __restrict__ out, uint32_t add)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        float4 v = partial[i];
        for (uint32_t c = 1; c < nchunks; ++c)
        {
            const float4 w = partial[(size_t)c * n + i];
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        if (add) { const float4 o = out[i]; v.x = o.x + v.x; v.y = o.y + v.y; v.z = o.z + v.z; v.w = o.w + v.w; }
        out[i] = v;
    }
}

// ================================================================================================ host side
struct pt_ctx
{
    int device = -1;
    std::string err;
    bool has_scene = false, moving = false, lds_ok = false, bvh_lds_ok = false;
    DeviceCamera cam{};          // folded camera of the uploaded scene (probe_get_rays; the kernels read the copy in the scene image)
    float4* blob_dev = nullptr; float4* blob_planes_dev = nullptr; float4* blob_bvh_dev = nullptr; // brute-force image, planes only, planes + tree
    float4* nodes_dev = nullptr; float4* sph_dev = nullptr; float4* vel_dev = nullptr; float4* mat_dev = nullptr;
    float4* mat_lds_dev = nullptr;  // materials in the order of the LDS tree's leaf blocks
    float4* blob_few_dev = nullptr; float4* mat_few_dev = nullptr; bool few_ok = false; SceneArgs sc_few{}; // MODE_FEW: planes + every sphere as leaf blocks
    uint32_t bvh_nodes = 0, bvh_depth = 0;
    SceneArgs sc_lds{}, sc_bvh{}, sc_bvh_lds{}, sc_ref{};
    pt_scene_desc desc{};                       // the uploaded scene's description: COPIES of the caller's arrays (below), for pt_hip_build_reference_tree
    std::vector<float> h_pos_r, h_vel;
    std::vector<uint32_t> slot_of_sphere;       // caller's sphere index -> slot in sph_dev / vel_dev / mat_dev (leaf order of the own tree)
    float4* ref_nodes_dev = nullptr; bool has_ref = false;
    float* accum_dev = nullptr; size_t accum_floats = 0;
    float* partial_dev = nullptr; size_t partial_floats = 0; // per-chunk sums of a chunked launch (grown on demand, kept)
    int num_cus = 0;
    unsigned long long* stats_dev = nullptr;
    unsigned long long* last_stats_dev = nullptr; // where the last launch left its counters (stats_dev or the static scratch)
    unsigned char* static_scratch = nullptr;      // this device's copy of g_static_scratch (pt_hip_render_device_static)
    uint32_t stat_slots = 0;
    bool bvh_q = false;   // the tree in HBM is stored as 32-byte grid nodes (MODE_BVH_Q)
    bool bvh_wide = false; // ... as 64-byte 4-wide grid nodes (MODE_BVH_Q4)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int last_mode = MODE_LDS_FAST;
    pt_stats last{};
    size_t lds_limit = 0;
    struct Occ { const void* kern; int block; size_t lds; int wgs; };
    std::vector<Occ> occ;   // answers of the occupancy query (render_occupancy)
};

// A scene laid out for the device, still on the host (pt_scene_prepare): every array pt_hip_commit_scene copies, and the
// kernel arguments with their device pointers null.  Immutable once built: any number of contexts may commit from it.
struct pt_prepared
{
    std::string err;
    DeviceCamera cam{};
    bool moving = false, lds_ok = false, bvh_lds_ok = false, bvh_q = false, bvh_wide = false;
    pth::Blob blob, blob_planes;           // brute-force image of the whole scene (if it fits LDS); planes + camera only
    std::vector<float> nodes_f;            // the tree read from HBM: 64-byte float nodes ...
    std::vector<uint32_t> nodes_u;         // ... or grid nodes (32-byte 2-wide / 64-byte 4-wide)
    std::vector<float> sph, vel, mat;      // sphere payload in leaf order
    std::vector<float> few_img, mat_few; bool few_ok = false; SceneArgs sc_few{}; // MODE_FEW (scenes of at most kFewMax spheres)
    std::vector<float> mat_lds;            // the materials once more in the order of the LDS tree's leaf blocks (slot = 4 x leaf + position)
    std::vector<float> bvh_lds_img;        // planes + tree + spheres staged in LDS by the MODE_BVH_LDS kernels
    SceneArgs sc_lds{}, sc_bvh{}, sc_bvh_lds{};
    uint32_t bvh_nodes = 0, bvh_depth = 0;
    pt_scene_desc desc{}; std::vector<float> h_pos_r, h_vel; std::vector<uint32_t> slot_of_sphere;
};

extern "C" int pt_hip_build_reference_tree(pt_ctx* ctx, uint32_t rng_skip);

namespace {

#define HIPCHK(ctx, call)                                                                                       \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            if (ctx) (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
            return PT_ERR_HIP;                                                                                  \
        }                                                                                                       \
    } while (0)

// Every extern "C" body runs between these two (pt_guard.h): no C++ exception leaves the library.
#define PT_ENTER(errp, name) return ptg::guard_status((errp), name, [&]() -> int {
#define PT_LEAVE });

struct Bind // make the context's device current for the duration of a call
{
    int prev = -1; bool ok = false;
    explicit Bind(pt_ctx* c) { if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(c->device) == hipSuccess) ok = true; }
    ~Bind() { if (prev >= 0) (void)hipSetDevice(prev); }
};

constexpr size_t kMaxLdsBytes = 128 * 1024; // of the 160 KiB per CU
constexpr size_t kAccBytesPerWave = 64 * 16;
constexpr uint32_t kMinResidentWaves = 16; // per CU: below this the LDS-resident tree loses to the tree in HBM
// BVH traversal: bvh_run's minActive for the tree in LDS and in HBM (book-1 1080p: 105.1 -> 102.2 ms with 8, flat
// from 6 to 12; 10^6 spheres 4K: 431 -> 594 -> 669 -> 700 Msamples/s with 0, 8, 16, 24); a single-leaf tree (up to 4 spheres) has no stragglers and skips it.
constexpr uint32_t kExitLanes = 10, kExitLanesHbm = 24, kResumeMinSpheres = 4; // (tree in LDS: 8 until the joint sweep at the end of round 2; 10 with leaf batch 20: +0.2-0.5 % at 64-512 spp, C1 +5 %)
constexpr uint32_t kGridNodeMinSpheres = 32768; // from here on the tree in HBM uses grid nodes (4-wide)
constexpr uint32_t kWideStackLds = 12;          // 4-wide traversal: stack levels kept in LDS, deeper ones in per-lane scratch (C5, 37 levels needed; round 3, 7 waves per SIMD: 6 / 8 / 10 / 12 / 16 levels 1 656 / 1 703 / 1 716 / 1 721 / 1 628 Msamples/s; round 2, 6 waves: 10-20 levels equal, 24 slower -- fewer resident waves)
// Lanes that must be out of work before the wave runs the refill code (render_body).  Scene in LDS: flat from 20 to 36 on
// book-1, +2-4 % at 28-36 on the small scenes; tree in HBM (trips dear, refill cheap by comparison): 12.
constexpr uint32_t kRefillMin = 28, kRefillMinHbm = 12;
// A unit whose pool is empty is left (its unfinished paths ride along in the next unit and add their results with four
// float atomics each) once at most this many paths are still under way; until then the unit runs on.  Book-1 1080p
// 512 spp: 64 (always leave) / 32 / 24 / 16 / 12 / 8 / 0 (finish everything): 89.79 / 89.95 / 90.05 / 90.11 / 90.5 / 90.68 /
// 95.16 ms with 17.5 / 5.2 / 3.0 / 1.4 / 0.9 / 0.4 / 0 million carried paths and 4.24 / - / 2.33 / - / 1.78 / - / 0.99 GB of
// memory-side writes per launch (WRITE_SIZE: float atomics execute at the memory side; the tile flushes alone are the
// 0.99 GB).  24 = the knee: a fifth of the scattered atomics for 0.3 % of the time.  Trees in HBM (trips dear): 32.
constexpr uint32_t kCarryMax = 24, kCarryMaxHbm = 32;
constexpr uint32_t kLeafBatchLds = 20;
constexpr uint32_t kLeafBatch = 16; // BVH traversal: lanes gathered at leaves before the leaf code runs (pt_device.h, bvh_closest)
// 4-wide nodes over single spheres (10^6 spheres 4K, 8 spp): a leaf step is one sphere test, so lanes wait for fewer
// companions (leaf batch 1 / 4 / 8 / 16 / 32: 1106 / 1212 / 1242 / 1171 / 938 Msamples/s at exit 40) and the traversal
// phase is left as soon as the finished lanes outnumber the others (exit 16 / 24 / 40-64: 1154 / 1187 / 1230-1245).
constexpr uint32_t kLeafBatchWide = 8, kExitLanesWide = 48;
constexpr uint32_t kExitMul = 4, kExitMulWide = 6; // traversal phase left when finished lanes x this > traversing lanes x 4 (wide: two finished for every three traversing)
constexpr uint32_t kRefillMinWide = 8, kCarryMaxWide = 48; // wide-node kernel, swept together with kExitMulWide at the end of round 2: C5 1 599 -> 1 636 Msamples/s

uint32_t pick_stride(uint32_t ntiles)
{
    // a multiplier coprime to ntiles, near the golden ratio of it
    if (ntiles < 3) return 1;
    auto gcd = [](uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; };
    uint32_t s = (uint32_t)((double)ntiles * 0.6180339887498949);
    if (s < 1) s = 1;
    while (gcd(s, ntiles) != 1) ++s;
    return s;
}

template <class K> int set_lds(pt_ctx* ctx, K kernel, size_t bytes)
{
    if (bytes > 64 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PT_OK;
}

int check_scene(pt_ctx* ctx)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->has_scene) return PT_ERR_NO_SCENE;
    return PT_OK;
}

// Which kernel family answers the sphere query for this call, and with which scene arguments.
// waves = waves per workgroup of the kernel about to be launched; acc = it needs the per-wave tile accumulators.
int select_mode(pt_ctx* ctx, uint32_t flags, uint32_t waves, bool acc, int* mode, const SceneArgs** sc, size_t* lds)
{
    const bool brute = (flags & (PT_FLAG_BRUTE_FORCE | PT_FLAG_STRICT_SEARCH)) != 0;
    if ((flags & PT_FLAG_FORCE_BVH) && ((flags & PT_FLAG_FORCE_LDS) || brute)) return PT_ERR_UNSUPPORTED;
    if (flags & PT_FLAG_REFERENCE_QUERY)
    {
        if (brute || (flags & (PT_FLAG_FORCE_BVH | PT_FLAG_FORCE_LDS))) return PT_ERR_UNSUPPORTED;
        if (!ctx->has_ref) { int st = pt_hip_build_reference_tree(ctx, 0); if (st != PT_OK) return st; }
        *mode = MODE_REF; *sc = &ctx->sc_ref;
        *lds = (size_t)(*sc)->blob_f4 * 16 + waves * ((acc ? kAccBytesPerWave : 0) + (size_t)(*sc)->stack_entries * 64 * 4);
        return *lds > ctx->lds_limit ? PT_ERR_UNSUPPORTED : PT_OK;
    }
    if (brute)
    {
        if (!ctx->lds_ok) return PT_ERR_UNSUPPORTED;
        *mode = (flags & PT_FLAG_STRICT_SEARCH) ? MODE_LDS_STRICT : MODE_LDS_FAST; *sc = &ctx->sc_lds;
    }
    else if (ctx->few_ok && !(flags & PT_FLAG_FORCE_BVH)) { *mode = MODE_FEW; *sc = &ctx->sc_few; }
    else if (ctx->bvh_lds_ok && !(flags & PT_FLAG_FORCE_BVH)) { *mode = MODE_BVH_LDS; *sc = &ctx->sc_bvh_lds; }
    else
    {
        if (flags & PT_FLAG_FORCE_LDS) return PT_ERR_UNSUPPORTED;
        *mode = ctx->bvh_wide ? MODE_BVH_Q4 : ctx->bvh_q ? MODE_BVH_Q : MODE_BVH; *sc = &ctx->sc_bvh;
    }
    const size_t stack = mode_is_bvh(*mode) ? (size_t)(*sc)->stack_entries * 64 * 4 : 0;
    *lds = (size_t)(*sc)->blob_f4 * 16 + waves * ((acc ? kAccBytesPerWave : 0) + stack);
    if (*lds > ctx->lds_limit) return PT_ERR_UNSUPPORTED;
    return PT_OK;
}

// Launch KERNEL<moving, mode> with dynamic LDS `lds` on `stream`.
#define PT_LAUNCH(KERNEL, moving, mode, grid, block, lds, stream, args)                                           \
    do {                                                                                                        \
        int st_ = PT_OK;                                                                                        \
        auto go_ = [&](auto kern) { st_ = set_lds(ctx, kern, lds); if (st_ == PT_OK) hipLaunchKernelGGL(kern, grid, block, lds, stream, args); }; \
        if (moving) { if (mode == MODE_BVH_LDS) go_(KERNEL<true, MODE_BVH_LDS>); else if (mode == MODE_BVH) go_(KERNEL<true, MODE_BVH>); else if (mode == MODE_BVH_Q) go_(KERNEL<true, MODE_BVH_Q>); else if (mode == MODE_BVH_Q4) go_(KERNEL<true, MODE_BVH_Q4>); else if (mode == MODE_LDS_STRICT) go_(KERNEL<true, MODE_LDS_STRICT>); else if (mode == MODE_REF) go_(KERNEL<true, MODE_REF>); else if (mode == MODE_FEW) go_(KERNEL<true, MODE_FEW>); else go_(KERNEL<true, MODE_LDS_FAST>); } \
        else        { if (mode == MODE_BVH_LDS) go_(KERNEL<false, MODE_BVH_LDS>); else if (mode == MODE_BVH) go_(KERNEL<false, MODE_BVH>); else if (mode == MODE_BVH_Q) go_(KERNEL<false, MODE_BVH_Q>); else if (mode == MODE_BVH_Q4) go_(KERNEL<false, MODE_BVH_Q4>); else if (mode == MODE_LDS_STRICT) go_(KERNEL<false, MODE_LDS_STRICT>); else if (mode == MODE_REF) go_(KERNEL<false, MODE_REF>); else if (mode == MODE_FEW) go_(KERNEL<false, MODE_FEW>); else go_(KERNEL<false, MODE_LDS_FAST>); } \
        if (st_ != PT_OK) return st_;                                                                           \
    } while (0)

constexpr uint64_t kGuide = 24;        // guided schedule: a unit is at most 1/kGuide of a resident wave's even share of what is left (tile cost varies ~10x; 48 while the
                                       // smallest chunk was 8 spp; with 4: 24 is +1 % at 512 spp, +2.5 % at 64-128 spp, 16 and below lose at 512)
constexpr uint32_t kMinChunkSpp = 4;   // never cut a tile's samples finer than this (every unit has fixed costs).  8 until the end of round 2: equal on
                                       // long frames (1080p, 512 spp), but short ones want more units than waves -- 800x450 at 32 spp 1.26 -> 1.14 ms,
                                       // 1080p at 16 spp 3.9 -> 3.5 ms (profiles/README.md); frames with less than kTinyFrame tile-samples per
                                       // resident wave go down to 2
constexpr uint64_t kTinyFrame = 16;
constexpr uint32_t kFewSpheres = 16;
constexpr uint32_t kFewMax = 8;        // up to this many spheres the query is a straight loop over leaf blocks, no tree (MODE_FEW)
constexpr uint32_t kMaxChunks = 4096;  // (the walk in the kernel is linear in the chunk index)
constexpr uint64_t kMaxPartialBytes = 4ull << 30; // per-chunk partial images the context may hold (PT_FLAG_REPRODUCIBLE frames only)

// Workgroups of the render kernel for (moving, mode) one CU holds at this block size and LDS footprint.
// (Asked once per kernel, shape and footprint, then remembered: a frame loop repeats the same few questions.)
int render_occupancy(pt_ctx* ctx, bool moving, int mode, bool resume, bool carry, int block, size_t lds, int* out)
{
    const RenderKernel kern = render_kernel(moving, mode, resume, carry);
    for (const pt_ctx::Occ& o : ctx->occ)
        if (o.kern == reinterpret_cast<const void*>(kern) && o.block == block && o.lds == lds) { *out = o.wgs; return PT_OK; }
    int st = set_lds(ctx, kern, lds);
    if (st != PT_OK) return st;
    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(out, kern, block, lds));
    ctx->occ.push_back(pt_ctx::Occ{reinterpret_cast<const void*>(kern), block, lds, *out});
    return PT_OK;
}

// device scratch that frees itself
struct DevBuf
{
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

void free_scene(pt_ctx* c)
{
    float4** ptrs[] = {&c->blob_dev, &c->blob_planes_dev, &c->blob_bvh_dev, &c->nodes_dev, &c->sph_dev, &c->vel_dev, &c->mat_dev, &c->mat_lds_dev, &c->blob_few_dev, &c->mat_few_dev};
    for (float4** p : ptrs) if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (c->ref_nodes_dev) { (void)hipFree(c->ref_nodes_dev); c->ref_nodes_dev = nullptr; }
    c->has_ref = false;
    c->has_scene = false;
}

int upload_f4(pt_ctx* ctx, const void* host, size_t bytes, float4** dev)
{
    HIPCHK(ctx, hipMalloc((void**)dev, bytes ? bytes : 16));
    if (bytes) HIPCHK(ctx, hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return PT_OK;
}

void fill_scene_args(SceneArgs* sc, const pth::Blob& b, const float4* dev, const pt_scene_desc* d, uint32_t np)
{
    sc->blob = dev; sc->blob_f4 = (uint32_t)b.f4.size();
    sc->off_velp = b.off_velp; sc->off_groups = b.off_groups; sc->off_smat = b.off_smat; sc->off_planes = b.off_planes; sc->off_pmat = b.off_pmat; sc->off_cam = b.off_cam;
    sc->ns = 2 * b.npairs; sc->npairs = b.npairs; sc->ngroups = b.ngroups; sc->np = np; sc->idx_mask = b.idx_mask;
    sc->t0 = d->t0; sc->t1 = d->t1;
    sc->refill_min = kRefillMin;
    sc->carry_max = kCarryMax;
    if (const char* e = getenv("PT_CARRY_MAX")) { int v = atoi(e); if (v >= 0 && v <= 64) sc->carry_max = (uint32_t)v; } // tuning experiments only
    if (const char* e = getenv("PT_REFILL_MIN")) { int v = atoi(e); if (v >= 1 && v <= 64) sc->refill_min = (uint32_t)v; } // tuning experiments only
}


std::atomic<uint64_t> g_tree_builds{0};    // runs of pth::build_bvh in this process (pt_tree_build_count: "one build, N commits")

// Host half of a scene upload: every array the kernels will read, laid out, and the kernel arguments with their device
// pointers still null.  Touches no device.
int prepare_scene(const pt_scene_desc* d, pt_prepared* P)
{
    int st = pth::validate_desc(d);
    if (st != PT_OK) return st;
    pth::make_device_camera(d->camera, &P->cam);
    // (1) LDS image of the whole scene, if it can fit beside the accumulators of a 4-wave workgroup
    const size_t roughBytes = (size_t)d->n_spheres * (16 + 16 + 32 + 2) + (size_t)d->n_planes * 48 + 4096;
    P->lds_ok = false;
    if (roughBytes <= kMaxLdsBytes)
    {
        pth::build_blob(d, &P->blob);
        P->lds_ok = P->blob.f4.size() * 16 + kWavesPerBlock * kAccBytesPerWave <= kMaxLdsBytes;
    }
    if (P->lds_ok)
    {
        fill_scene_args(&P->sc_lds, P->blob, nullptr, d, d->n_planes);
        P->sc_lds.bvh = BvhView{};
        P->moving = P->blob.moving;
    }
    else
        P->blob = pth::Blob{};
    // (2) BVH in HBM (always built: it is what PT_FLAG_FORCE_BVH and every scene beyond LDS use) + planes-only LDS image
    pth::BvhData bv;
    pth::build_bvh(d, &bv);
    g_tree_builds.fetch_add(1);
    pt_scene_desc planesOnly = *d;
    planesOnly.n_spheres = 0;
    pth::build_blob(&planesOnly, &P->blob_planes);
    // Big trees are read from HBM as 32-byte grid nodes (half the loads per step: the L1 is what binds that kernel);
    // small ones, resident in L1/L2 and bound by instruction issue, keep the 64-byte float nodes.
    // ... and from kGridNodeMinSpheres up as 64-byte 4-wide grid nodes over single spheres (MODE_BVH_Q4): half the
    // dependent fetches per ray.  PT_GRID_NODES (tests / tuning): 0 float nodes, 1 32-byte 2-wide, 2 4-wide over
    // the binary tree's leaves, 3 4-wide over single spheres (the default for big trees).
    int gridKind = bv.n_spheres >= kGridNodeMinSpheres ? 3 : 0;
    if (const char* e = getenv("PT_GRID_NODES")) gridKind = atoi(e);
    if (gridKind < 0 || gridKind > 3) gridKind = 0;
    if (bv.n_spheres == 0) gridKind = 0;
    P->bvh_q = gridKind == 1 && bv.n_nodes != 0;
    P->bvh_wide = false;
    pth::NodeGrid grid{};
    pth::WideData wd;
    if (gridKind >= 2)
    {
        pth::widen_nodes(bv, d, gridKind == 3, &grid, &wd);
        // (a tree too deep for the stack's LDS + scratch levels keeps the 2-wide grid nodes)
        if (wd.n_nodes != 0 && 3u * wd.depth + 1u <= kWideStackLds + kOvfEntries) P->bvh_wide = true;
        else P->bvh_q = bv.n_nodes != 0;
    }
    if (P->bvh_wide) P->nodes_u.swap(wd.nodes);
    else if (P->bvh_q) pth::quantise_nodes(bv, &grid, &P->nodes_u);
    fill_scene_args(&P->sc_bvh, P->blob_planes, nullptr, d, d->n_planes);
    P->sc_bvh.bvh = BvhView{nullptr, nullptr, nullptr, nullptr, P->bvh_wide ? wd.root_ref : bv.root_ref, bv.n_spheres,
                            {grid.org[0], grid.org[1], grid.org[2]}, {grid.step[0], grid.step[1], grid.step[2]}};
    P->sc_bvh.stack_entries = bv.depth + 1 < 4 ? 4 : bv.depth + 1; // the sentinel + at most one pushed sibling per inner node of a root-to-leaf path (bv.depth of them)
    if (P->bvh_wide)
    {
        // up to three pushed siblings per level; the first kWideStackLds levels in LDS, the rest in scratch (pt_device.h)
        uint32_t capLds = kWideStackLds;
        if (const char* e = getenv("PT_WIDE_STACK")) { int v = atoi(e); if (v >= 2 && v <= 255) capLds = (uint32_t)v; } // tuning experiments only
        const uint32_t need = 3u * wd.depth + 1u;
        P->sc_bvh.stack_entries = need < capLds ? (need < 4 ? 4 : need) : capLds;
        if (need > P->sc_bvh.stack_entries + kOvfEntries) { P->err = "wide BVH deeper than the traversal stack"; return PT_ERR_UNSUPPORTED; }
    }
    P->sc_bvh.leaf_batch = kLeafBatch;
    int exitEnv = -1;
    if (const char* e = getenv("PT_EXIT_LANES")) { int v = atoi(e); if (v >= 0 && v <= 64) exitEnv = v; } // tuning experiments only
    const bool resumable = bv.n_spheres > kResumeMinSpheres;

    int exitMul = -1;
    if (const char* e = getenv("PT_EXIT_MUL")) { int v = atoi(e); if (v >= 1 && v <= 255) exitMul = v; } // tuning experiments only
    P->sc_bvh.exit_lanes = !resumable ? 0u : pack_exit_rule(exitEnv >= 0 ? (uint32_t)exitEnv : P->bvh_wide ? kExitLanesWide : kExitLanesHbm,
                                                           exitMul > 0 ? (uint32_t)exitMul : P->bvh_wide ? kExitMulWide : kExitMul);
    if (P->bvh_wide) P->sc_bvh.leaf_batch = kLeafBatchWide;
    const uint32_t refillLds = P->sc_bvh.refill_min; // (fill_scene_args: the default, or PT_REFILL_MIN)
    if (!getenv("PT_REFILL_MIN")) P->sc_bvh.refill_min = P->bvh_wide ? kRefillMinWide : kRefillMinHbm;
    const uint32_t carryLds = P->sc_bvh.carry_max; // (fill_scene_args: the default, or PT_CARRY_MAX)
    if (!getenv("PT_CARRY_MAX")) P->sc_bvh.carry_max = P->bvh_wide ? kCarryMaxWide : kCarryMaxHbm;
    if (const char* e = getenv("PT_LEAF_BATCH")) { int v = atoi(e); if (v >= 1 && v <= 64) P->sc_bvh.leaf_batch = (uint32_t)v; } // tuning experiments only
    P->bvh_nodes = bv.n_nodes; P->bvh_depth = bv.depth;
    P->moving = bv.moving;
    // what pt_hip_build_reference_tree needs later: the spheres as the caller numbered them, and where each one sits now
    P->desc = *d;
    P->h_pos_r.assign(d->sphere_pos_r, d->sphere_pos_r + 4 * (size_t)d->n_spheres);
    if (d->sphere_vel) P->h_vel.assign(d->sphere_vel, d->sphere_vel + 3 * (size_t)d->n_spheres); else P->h_vel.clear();
    P->desc.sphere_pos_r = nullptr; P->desc.sphere_vel = nullptr; // (commit points them at the context's own copies)
    P->desc.sphere_mat_type = nullptr; P->desc.sphere_mat_rgba = nullptr; P->desc.sphere_mat_param = nullptr;
    P->desc.plane_n_d = nullptr; P->desc.plane_mat_type = nullptr; P->desc.plane_mat_rgba = nullptr; P->desc.plane_mat_param = nullptr;
    P->slot_of_sphere.assign(d->n_spheres, 0u);
    for (uint32_t slot = 0; slot < bv.n_spheres; ++slot) P->slot_of_sphere[pth::f2u(bv.mat[8 * (size_t)slot + 3])] = slot;
    if (bv.depth + 1 > kBvhStack) { P->err = "BVH deeper than the traversal stack"; return PT_ERR_UNSUPPORTED; }
    // (3) the same tree staged in LDS behind the planes, when planes + tree + leaf blocks + 4 waves' stacks fit.  The LDS
    // flavour (pt_device.h, "the tree in LDS"): compact references, 16-bit stack entries (two levels per dword and lane),
    // the spheres as one structure-of-arrays block per leaf; materials stay in HBM (read once per hit), in block order.
    const uint32_t levels = bv.depth + 1 < 4 ? 4 : bv.depth + 1;  // the sentinel + at most one pushed sibling per inner node of a root-to-leaf path
    const uint32_t entries = (levels + 1) / 2;                     // ... in dwords per lane
    std::vector<float> ldsNodes(bv.nodes), blocks, velBlocks;
    uint32_t nLeaves = 0;
    bool blocksOk = bv.n_nodes < 0x8000u;
    auto add_leaf = [&](uint32_t ref) -> uint32_t {
        const uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 7u) + 1u, leaf = nLeaves++;
        if (count > 4u || leaf >= 0x1000u || (uint64_t)first + count > bv.n_spheres) { blocksOk = false; return kLdsLeafBase; }
        blocks.resize(16 * (size_t)nLeaves); P->mat_lds.resize(32 * (size_t)nLeaves, 0.0f);
        if (bv.moving) velBlocks.resize(12 * (size_t)nLeaves, 0.0f);
        float* blk = blocks.data() + 16 * (size_t)leaf;
        for (uint32_t k = 0; k < 4u; ++k)
        {
            const bool real = k < count;
            const float* sp = bv.sph.data() + 4 * (size_t)(first + (real ? k : 0u));   // an unused slot: the leaf's first centre, r2 = -1e30 (never a candidate)
            blk[k] = sp[0]; blk[4 + k] = sp[1]; blk[8 + k] = sp[2]; blk[12 + k] = real ? sp[3] : -1.0e30f;
            float* m = P->mat_lds.data() + 8 * (size_t)(4u * leaf + k);
            if (real) memcpy(m, bv.mat.data() + 8 * (size_t)(first + k), 32); else m[3] = pth::u2f(0xFFFFFFFFu);
            if (bv.moving && real) { const float* v = bv.vel.data() + 4 * (size_t)(first + k); float* vb = velBlocks.data() + 12 * (size_t)leaf; vb[k] = v[0]; vb[4 + k] = v[1]; vb[8 + k] = v[2]; }
        }
        return kLdsLeafBase | ((count - 1u) << 12) | leaf;
    };
    uint32_t ldsRoot = 0;
    if (bv.n_spheres != 0 && bv.n_nodes == 0) ldsRoot = add_leaf(bv.root_ref);
    for (uint32_t node = 0; node < bv.n_nodes; ++node)
        for (int c = 12; c <= 13; ++c)
        {
            const uint32_t ref = pth::f2u(ldsNodes[16 * (size_t)node + c]);
            if (ref & pth::kLeafFlag) ldsNodes[16 * (size_t)node + c] = pth::u2f(add_leaf(ref));
        }
    const size_t f4s = P->blob_planes.f4.size() + (ldsNodes.size() + blocks.size() + velBlocks.size()) / 4;
    P->bvh_lds_ok = blocksOk && f4s * 16 + kWavesPerBlock * (kAccBytesPerWave + (size_t)entries * 256) <= kMaxLdsBytes;
    if (P->bvh_lds_ok)
    {
        std::vector<float>& img = P->bvh_lds_img;
        img.resize(P->blob_planes.f4.size() * 4);
        if (!img.empty()) memcpy(img.data(), P->blob_planes.f4.data(), img.size() * 4);
        uint32_t off[4] = {0, 0, 0, 0xFFFFFFFFu};
        const std::vector<float>* parts[3] = {&ldsNodes, &blocks, &velBlocks};
        for (int k = 0; k < 3; ++k) { off[k] = (uint32_t)(img.size() / 4); img.insert(img.end(), parts[k]->begin(), parts[k]->end()); }
        P->sc_bvh_lds = P->sc_bvh;
        P->sc_bvh_lds.exit_lanes = !resumable ? 0u : pack_exit_rule(exitEnv >= 0 ? (uint32_t)exitEnv : kExitLanes, exitMul > 0 ? (uint32_t)exitMul : kExitMul);
        P->sc_bvh_lds.refill_min = refillLds;
        if (!getenv("PT_LEAF_BATCH")) P->sc_bvh_lds.leaf_batch = kLeafBatchLds;
        P->sc_bvh_lds.carry_max = carryLds;
        P->sc_bvh_lds.blob = nullptr; P->sc_bvh_lds.blob_f4 = (uint32_t)(img.size() / 4);
        for (int k = 0; k < 4; ++k) P->sc_bvh_lds.bvh_off[k] = off[k];
        P->sc_bvh_lds.bvh = BvhView{nullptr, nullptr, nullptr, nullptr, ldsRoot, bv.n_spheres, {0, 0, 0}, {1, 1, 1}};
        P->sc_bvh_lds.stack_entries = entries;
    }
    else
        P->mat_lds.clear();
    // (4) a handful of spheres: no tree at all -- planes + every sphere in leaf blocks, every ray meets every block
    P->few_ok = bv.n_spheres != 0 && bv.n_spheres <= kFewMax && !getenv("PT_NO_FEW");
    if (P->few_ok)
    {
        const uint32_t nb = (bv.n_spheres + 3u) / 4u;
        std::vector<float> fb(16 * (size_t)nb), fv(bv.moving ? 12 * (size_t)nb : 0, 0.0f);
        P->mat_few.assign(32 * (size_t)nb, 0.0f);
        for (uint32_t sl = 0; sl < 4u * nb; ++sl)
        {
            const bool real = sl < bv.n_spheres;
            const float* sp = bv.sph.data() + 4 * (size_t)(real ? sl : 0u);
            float* blk = fb.data() + 16 * (size_t)(sl >> 2); const uint32_t k = sl & 3u;
            blk[k] = sp[0]; blk[4 + k] = sp[1]; blk[8 + k] = sp[2]; blk[12 + k] = real ? sp[3] : -1.0e30f;
            float* m = P->mat_few.data() + 8 * (size_t)sl;
            if (real) memcpy(m, bv.mat.data() + 8 * (size_t)sl, 32); else m[3] = pth::u2f(0xFFFFFFFFu);
            if (bv.moving && real) { const float* v = bv.vel.data() + 4 * (size_t)sl; float* vb = fv.data() + 12 * (size_t)(sl >> 2); vb[k] = v[0]; vb[4 + k] = v[1]; vb[8 + k] = v[2]; }
        }
        std::vector<float>& img = P->few_img;
        img.resize(P->blob_planes.f4.size() * 4);
        if (!img.empty()) memcpy(img.data(), P->blob_planes.f4.data(), img.size() * 4);
        P->sc_few = P->sc_bvh;
        P->sc_few.refill_min = refillLds; P->sc_few.carry_max = carryLds; P->sc_few.exit_lanes = 0; P->sc_few.stack_entries = 0;
        P->sc_few.bvh_off[0] = 0; P->sc_few.bvh_off[3] = 0xFFFFFFFFu;
        P->sc_few.bvh_off[1] = (uint32_t)(img.size() / 4); img.insert(img.end(), fb.begin(), fb.end());
        P->sc_few.bvh_off[2] = (uint32_t)(img.size() / 4); img.insert(img.end(), fv.begin(), fv.end());
        P->sc_few.blob = nullptr; P->sc_few.blob_f4 = (uint32_t)(img.size() / 4);
        P->sc_few.bvh = BvhView{nullptr, nullptr, nullptr, nullptr, 0u, 4u * nb, {0, 0, 0}, {1, 1, 1}};
    }
    if (!P->bvh_wide && !P->bvh_q) P->nodes_f.swap(bv.nodes);
    P->sph.swap(bv.sph); P->vel.swap(bv.vel); P->mat.swap(bv.mat);
    return PT_OK;
}

// Device half: allocations and copies on the context's GPU, then the kernel arguments get their pointers.
int commit_scene(pt_ctx* ctx, const pt_prepared& P)
{
    Bind b(ctx);
    if (!b.ok) return PT_ERR_NO_DEVICE;
    free_scene(ctx);
    ctx->err.clear();
    ctx->cam = P.cam;
    ctx->lds_ok = P.lds_ok; ctx->bvh_lds_ok = P.bvh_lds_ok; ctx->bvh_q = P.bvh_q; ctx->bvh_wide = P.bvh_wide; ctx->moving = P.moving;
    ctx->bvh_nodes = P.bvh_nodes; ctx->bvh_depth = P.bvh_depth;
    // host-side copies for pt_hip_build_reference_tree (the prepared scene may be freed right after the commit)
    ctx->h_pos_r = P.h_pos_r; ctx->h_vel = P.h_vel; ctx->slot_of_sphere = P.slot_of_sphere;
    ctx->desc = P.desc;
    ctx->desc.sphere_pos_r = ctx->h_pos_r.data(); ctx->desc.sphere_vel = ctx->h_vel.empty() ? nullptr : ctx->h_vel.data();
    int st;
    if (P.lds_ok)
    {
        st = upload_f4(ctx, P.blob.f4.data(), P.blob.f4.size() * 16, &ctx->blob_dev); if (st != PT_OK) return st;
        ctx->sc_lds = P.sc_lds; ctx->sc_lds.blob = ctx->blob_dev;
    }
    st = upload_f4(ctx, P.blob_planes.f4.data(), P.blob_planes.f4.size() * 16, &ctx->blob_planes_dev); if (st != PT_OK) return st;
    if (P.bvh_wide || P.bvh_q) st = upload_f4(ctx, P.nodes_u.data(), P.nodes_u.size() * 4, &ctx->nodes_dev);
    else st = upload_f4(ctx, P.nodes_f.data(), P.nodes_f.size() * 4, &ctx->nodes_dev);
    if (st != PT_OK) return st;
    st = upload_f4(ctx, P.sph.data(), P.sph.size() * 4, &ctx->sph_dev); if (st != PT_OK) return st;
    st = upload_f4(ctx, P.vel.data(), P.vel.size() * 4, &ctx->vel_dev); if (st != PT_OK) return st;
    st = upload_f4(ctx, P.mat.data(), P.mat.size() * 4, &ctx->mat_dev); if (st != PT_OK) return st;
    ctx->sc_bvh = P.sc_bvh;
    ctx->sc_bvh.blob = ctx->blob_planes_dev;
    ctx->sc_bvh.bvh.nodes = ctx->nodes_dev; ctx->sc_bvh.bvh.sph = ctx->sph_dev; ctx->sc_bvh.bvh.vel = ctx->vel_dev; ctx->sc_bvh.bvh.mat = ctx->mat_dev;
    if (P.bvh_lds_ok)
    {
        st = upload_f4(ctx, P.bvh_lds_img.data(), P.bvh_lds_img.size() * 4, &ctx->blob_bvh_dev); if (st != PT_OK) return st;
        st = upload_f4(ctx, P.mat_lds.data(), P.mat_lds.size() * 4, &ctx->mat_lds_dev); if (st != PT_OK) return st;
        ctx->sc_bvh_lds = P.sc_bvh_lds;
        ctx->sc_bvh_lds.blob = ctx->blob_bvh_dev;
        ctx->sc_bvh_lds.bvh.mat = ctx->mat_lds_dev;
    }
    ctx->few_ok = P.few_ok;
    if (P.few_ok)
    {
        st = upload_f4(ctx, P.few_img.data(), P.few_img.size() * 4, &ctx->blob_few_dev); if (st != PT_OK) return st;
        st = upload_f4(ctx, P.mat_few.data(), P.mat_few.size() * 4, &ctx->mat_few_dev); if (st != PT_OK) return st;
        ctx->sc_few = P.sc_few;
        ctx->sc_few.blob = ctx->blob_few_dev;
        ctx->sc_few.bvh.mat = ctx->mat_few_dev;
    }
    ctx->has_scene = true;
    return PT_OK;
}

} // namespace

extern "C" {

int pt_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* pt_hip_strerror(int status)
{
    switch (status)
    {
    case PT_OK: return "ok";
    case PT_ERR_INVALID: return "invalid argument";
    case PT_ERR_NO_DEVICE: return "no such HIP device";
    case PT_ERR_HIP: return "HIP call failed (see pt_hip_last_error)";
    case PT_ERR_NO_SCENE: return "no scene uploaded";
    case PT_ERR_UNSUPPORTED: return "unsupported scene or flag combination";
    case PT_ERR_IO: return "I/O error";
    case PT_ERR_NO_MEMORY: return "out of host memory";
    case PT_ERR_INTERNAL: return "internal error (exception caught at the library boundary)";
    default: return "unknown status";
    }
}

const char* pt_hip_last_error(pt_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int pt_hip_create(int device, pt_ctx** out)
{
    PT_ENTER(nullptr, "pt_hip_create")
    if (!out) return PT_ERR_INVALID;
    *out = nullptr;
    int n = pt_hip_device_count();
    if (device < 0 || device >= n) return PT_ERR_NO_DEVICE;
    pt_ctx* c = new (std::nothrow) pt_ctx();
    if (!c) return PT_ERR_INVALID;
    c->device = device;
    c->occ.reserve(64); c->err.reserve(512); // (a frame loop's few shapes and its error texts then fit without a host allocation)
    Bind b(c);
    if (!b.ok) { delete c; return PT_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return PT_ERR_NO_DEVICE; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    {
        // Written for CDNA4 only; refuse loudly rather than run an untested configuration.
        fprintf(stderr, "pt_hip: device %d is %s, this library is built for gfx950 only\n", device, prop.gcnArchName);
        delete c;
        return PT_ERR_NO_DEVICE;
    }
    c->lds_limit = kMaxLdsBytes;
    c->num_cus = prop.multiProcessorCount;
    c->stat_slots = (uint32_t)c->num_cus * 32u; // no launch holds more waves than the chip's 32 per CU
    // (this device is current: the address is that of ITS copy of the array)
    if (hipGetSymbolAddress((void**)&c->static_scratch, HIP_SYMBOL(g_static_scratch)) != hipSuccess) c->static_scratch = nullptr; // pt_hip_render_device_static then fails
    if (hipMalloc((void**)&c->stats_dev, (size_t)kStatWords * (1u + c->stat_slots) * sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess)
    {
        pt_hip_destroy(c);
        return PT_ERR_HIP;
    }
    *out = c;
    return PT_OK;
    PT_LEAVE
}

void pt_hip_destroy(pt_ctx* c)
{
    if (!c) return;
    Bind b(c);
    free_scene(c);
    if (c->accum_dev) (void)hipFree(c->accum_dev);
    if (c->partial_dev) (void)hipFree(c->partial_dev);
    if (c->stats_dev) (void)hipFree(c->stats_dev);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

// ---- scene upload in two steps: host PREPARE (layouts + tree, no device touched) and per-device COMMIT (copies) ----------
// The reference builds its tree once per process (src/main.cpp:402) and every worker reads it; a node of 8 GPUs does
// the same here: pt_node_upload_scene prepares once and commits the result to each device.
int pt_scene_prepare(const pt_scene_desc* d, pt_prepared** out)
{
    if (!out) return PT_ERR_INVALID;
    *out = nullptr;
    if (!d) return PT_ERR_INVALID;
    pt_prepared* P = nullptr;
    const int st = ptg::guard_status(nullptr, "pt_scene_prepare", [&]() {
        P = new pt_prepared();
        return prepare_scene(d, P);
    });
    if (st != PT_OK) { delete P; return st; }
    *out = P;
    return PT_OK;
}

void pt_prepared_free(pt_prepared* P) { delete P; }

uint64_t pt_tree_build_count(void) { return g_tree_builds.load(); }

int pt_hip_commit_scene(pt_ctx* ctx, const pt_prepared* P)
{
    if (!ctx || !P) return PT_ERR_INVALID;
    return ptg::guard_status(&ctx->err, "pt_hip_commit_scene", [&]() { return commit_scene(ctx, *P); });
}

int pt_hip_upload_scene(pt_ctx* ctx, const pt_scene_desc* d)
{
    if (!ctx || !d) return PT_ERR_INVALID;
    return ptg::guard_status(&ctx->err, "pt_hip_upload_scene", [&]() {
        pt_prepared P;
        int st = prepare_scene(d, &P);
        if (st != PT_OK) { ctx->err = P.err; return st; }
        return commit_scene(ctx, P);
    });
}

int pt_hip_build_reference_tree(pt_ctx* ctx, uint32_t rng_skip)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_build_reference_tree")
    int st = check_scene(ctx);
    if (st != PT_OK) return st;
    const uint32_t n = ctx->desc.n_spheres;
    if (n > PT_REFERENCE_MAX_SPHERES) { ctx->err = "PT_FLAG_REFERENCE_QUERY: more spheres than the reference's World holds (4192)"; return PT_ERR_UNSUPPORTED; }
    Bind b(ctx);
    if (ctx->ref_nodes_dev) { (void)hipFree(ctx->ref_nodes_dev); ctx->ref_nodes_dev = nullptr; }
    ctx->has_ref = false;
    ctx->sc_ref = ctx->sc_bvh;                       // planes-only LDS image; spheres, velocities, materials in HBM (slot order)
    ctx->sc_ref.exit_lanes = 0;
    ctx->sc_ref.ref = RefView{nullptr, 0u, 0u};
    uint32_t depth = 0;
    if (n != 0)
    {
        const uint32_t nn = 2u * n - 1u;
        std::vector<float> boxes(6 * (size_t)nn); std::vector<int32_t> links(3 * (size_t)nn);
        if (pt_reference_bvh(&ctx->desc, rng_skip, nn, boxes.data(), links.data(), nullptr, &depth) != nn) return PT_ERR_INVALID;
        std::vector<float> img(8 * (size_t)nn);
        for (uint32_t i = 0; i < nn; ++i)
        {
            const float* bx = boxes.data() + 6 * (size_t)i; const int32_t* lk = links.data() + 3 * (size_t)i;
            float* o = img.data() + 8 * (size_t)i;
            o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3]; o[4] = bx[4]; o[5] = bx[5];
            const bool leaf = lk[2] >= 0;
            o[6] = pth::u2f(leaf ? ctx->slot_of_sphere[(size_t)lk[2]] : (uint32_t)lk[0]);
            o[7] = pth::u2f(leaf ? kRefLeaf : (uint32_t)lk[1]);
        }
        st = upload_f4(ctx, img.data(), img.size() * 4, &ctx->ref_nodes_dev); if (st != PT_OK) return st;
        ctx->sc_ref.ref = RefView{ctx->ref_nodes_dev, nn - 1u, nn};  // the root is allocated last (main.cpp:152)
    }
    ctx->sc_ref.stack_entries = 3u * (depth + 1u);   // ref_query: three dwords per waiting call
    ctx->has_ref = true;
    return PT_OK;
    PT_LEAVE
}

// One frame: the body of pt_hip_render_device and of pt_hip_render_device_static.  `fixed` = the scratch is the
// per-device static array (nothing is allocated, bigger needs are refused); else the context's own, grown on demand.
static int render_launch(pt_ctx* ctx, const pt_params* p, float* accum_dev, hipStream_t stream, bool fixed)
{
    int st = check_scene(ctx);
    if (st != PT_OK) return st;
    ctx->err.clear();
    if (!p || !accum_dev || p->width == 0 || p->height == 0 || p->spp_end <= p->spp_begin) return PT_ERR_INVALID;
    // (a carried path keeps its pixel's index in 31 bits; an item index = 64 x samples of a chunk in 32)
    if ((uint64_t)p->width * p->height >= 0x80000000ull || (uint64_t)(p->spp_end - p->spp_begin) * 64ull > 0xFFFFFFFFull) return PT_ERR_INVALID;
    Bind b(ctx);
    unsigned long long* const stats_dev = fixed ? reinterpret_cast<unsigned long long*>(ctx->static_scratch) : ctx->stats_dev;
    const uint32_t stat_slots = fixed ? kStaticSlots : ctx->stat_slots;
    if (fixed && !ctx->static_scratch) { ctx->err = "the static scratch array has no address on this device"; return PT_ERR_HIP; }
    // (PT_FLAG_REFERENCE_QUERY builds its tree on first use -- an allocation: the entry point that allocates nothing wants it built before)
    if (fixed && (p->flags & PT_FLAG_REFERENCE_QUERY) && !ctx->has_ref)
    {
        ctx->err = "pt_hip_render_device_static: PT_FLAG_REFERENCE_QUERY needs pt_hip_build_reference_tree to have been called (this entry point does not allocate)";
        return PT_ERR_UNSUPPORTED;
    }
    int mode; const SceneArgs* sc; size_t lds = 0;
    // Workgroup shape: the waves of a workgroup share one staged copy of the scene, so bigger workgroups spend less LDS
    // per wave; take the shape that keeps the most waves resident on a CU (ties: the smaller workgroup).  With the tree
    // in HBM there is nothing to share and one wave per workgroup packs best.
    uint32_t waves = 0; int wgPerCu = 0;
    uint32_t forced = 0;
    if (const char* e = getenv("PT_WAVES_PER_WG")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) forced = (uint32_t)v; } // tuning experiments only
    for (uint32_t flags = p->flags;;)
    {
        st = select_mode(ctx, flags, 1, true, &mode, &sc, &lds);
        if (st != PT_OK) return st;
        const uint32_t lo = forced ? forced : (mode_in_hbm(mode) ? kWavesPerBlockBvh : kWavesPerBlock), hi = forced ? forced : (mode_in_hbm(mode) ? kWavesPerBlockBvh : 8u);
        waves = 0; wgPerCu = 0;
        for (uint32_t wv = lo; wv <= hi; wv *= 2)
        {
            int m2; const SceneArgs* sc2; size_t lds2; int occ = 0;
            if (select_mode(ctx, flags, wv, true, &m2, &sc2, &lds2) != PT_OK) continue;
            st = render_occupancy(ctx, ctx->moving, m2, sc2->exit_lanes != 0, !(p->flags & PT_FLAG_REPRODUCIBLE), (int)(64 * wv), lds2, &occ);
            if (st != PT_OK) return st;
            if ((uint32_t)occ * wv > (uint32_t)wgPerCu * waves) { waves = wv; wgPerCu = occ; lds = lds2; }
        }
        // A tree that fits LDS only by crowding out the waves is better left in HBM (L1/L2-resident at that size).
        if (mode == MODE_BVH_LDS && (uint32_t)wgPerCu * waves < kMinResidentWaves && !(p->flags & PT_FLAG_FORCE_LDS) && flags == p->flags) { flags |= PT_FLAG_FORCE_BVH; continue; }
        break;
    }
    if (waves == 0) { ctx->err = "render kernel does not fit a compute unit"; return PT_ERR_UNSUPPORTED; }
    RenderArgs a{};
    a.sc = *sc;
    a.width = p->width; a.height = p->height;
    a.tiles_x = (p->width + 7) / 8;
    a.ntiles = a.tiles_x * ((p->height + 7) / 8);
    a.tile_stride = pick_stride(a.ntiles);
    a.spp_begin = p->spp_begin; a.nspp = p->spp_end - p->spp_begin; a.max_depth = p->max_depth; a.seed = p->seed;
    a.out_sum = accum_dev; a.stats = stats_dev;
    // Persistent launch: as many workgroups as the chip holds at once; waves pull (tile, sample chunk) units.
    const dim3 block(64 * waves);
    const uint64_t slots = (uint64_t)wgPerCu * (uint64_t)ctx->num_cus * waves; // waves resident at once
    // The guided schedule (chunk_size above): a unit of the current chunk should be no more than 1/kGuide of an even
    // share of the remaining work per resident wave -- ratio = ntiles / (kGuide x slots), capped at 1/2 per chunk.
    // (scenes of a handful of spheres run a sample in a quarter of the time: a unit's fixed costs weigh as much more, and
    // 8 stays their optimum: C2 22.6 against 20.7, C4 25.2 against 24.1 Gsamples/s)
    const bool fewSpheres = ctx->desc.n_spheres <= kFewSpheres;
    uint32_t minSpp = fewSpheres ? 2u * kMinChunkSpp : (uint64_t)a.ntiles * a.nspp < kTinyFrame * slots ? 2u : kMinChunkSpp;
    uint64_t guide = fewSpheres ? 2u * kGuide : kGuide; // (C2, 3 spheres at 1080p: 22.6 against 20.8 Gsamples/s)
    if (const char* e = getenv("PT_CHUNK_MIN")) { int v = atoi(e); if (v >= 1) minSpp = (uint32_t)v; }  // tuning experiments only
    if (const char* e = getenv("PT_GUIDE")) { int v = atoi(e); if (v >= 1) guide = (uint64_t)v; }        // tuning experiments only
    uint64_t ratio = ((uint64_t)a.ntiles << 16) / (guide * slots);
    if (ratio > 0x8000u) ratio = 0x8000u;
    if (getenv("PT_UNIFORM_CHUNKS")) ratio = 0;                                                          // tuning experiments only
    // Several chunks per tile meet in one image.  Default: every unit adds its tile to the output (float atomics; the
    // order of the additions into a pixel is the run's, like the carried paths').  PT_FLAG_REPRODUCIBLE: one partial
    // image per chunk in scratch, folded in chunk order by k_fold_chunks.
    const bool reproducible = (p->flags & PT_FLAG_REPRODUCIBLE) != 0;
    // PT_FLAG_ACCUMULATE: the frame's sums are ADDED to what accum_dev holds (a progressive render's running sums): the
    // tiles go in with the same float atomics (no zeroing), a reproducible frame is folded first and added once.
    const bool addTo = (p->flags & PT_FLAG_ACCUMULATE) != 0;
    const uint64_t frameBytes = (uint64_t)p->width * p->height * 16ull;
    uint32_t nchunks = 0;
    for (;;)
    {
        nchunks = 0;
        for (uint32_t s0 = 0; s0 < a.nspp; ++nchunks) s0 += chunk_size(a.nspp - s0, (uint32_t)ratio, minSpp);
        if (nchunks == 1) break;                     // one piece: no partial images, nothing to bound
        if (nchunks <= kMaxChunks && (!reproducible || fixed || (uint64_t)nchunks * frameBytes <= kMaxPartialBytes)) break;
        if (minSpp > 0x40000000u) { ratio = 0; minSpp = a.nspp; continue; } // (cannot get coarser: one chunk)
        minSpp *= 2; // (tiny images with huge sample counts) coarser chunks: fewer partial images
    }
    a.chunk_ratio_q16 = (uint32_t)ratio; a.chunk_min = minSpp;
    a.nchunks = nchunks;
    if ((uint64_t)a.ntiles * nchunks > 0xFFFFFF00ull) return PT_ERR_INVALID;
    a.nunits = a.ntiles * nchunks;
    a.partial = nullptr; a.chunk_stride = 0; a.accumulate = 0;
    if ((nchunks > 1 || addTo) && reproducible)
    {
        const uint64_t bytes = (uint64_t)nchunks * frameBytes;
        if (fixed)
        {
            // same schedule as pt_hip_render_device (the frame must come out bit-identical), so the need is what it is
            if (bytes > kStaticPartialBytes)
            {
                char msg[256];
                snprintf(msg, sizeof msg, "pt_hip_render_device_static: a PT_FLAG_REPRODUCIBLE frame of %ux%u in %u sample chunks needs %llu bytes of scratch, the static maximum is %llu",
                         p->width, p->height, nchunks, (unsigned long long)bytes, (unsigned long long)kStaticPartialBytes);
                ctx->err.assign(msg);
                return PT_ERR_UNSUPPORTED;
            }
            a.partial = reinterpret_cast<float*>(ctx->static_scratch + kStaticStatsBytes);
        }
        else
        {
            const size_t floats = (size_t)(bytes / 4);
            if (floats > ctx->partial_floats)
            {
                if (ctx->partial_dev) { (void)hipFree(ctx->partial_dev); ctx->partial_dev = nullptr; ctx->partial_floats = 0; }
                HIPCHK(ctx, hipMalloc((void**)&ctx->partial_dev, floats * sizeof(float)));
                ctx->partial_floats = floats;
            }
            a.partial = ctx->partial_dev;
        }
        a.chunk_stride = p->width * p->height;
    }
    else if (nchunks > 1 || addTo)
        a.accumulate = 1;
    uint64_t wgs = ((uint64_t)a.nunits + waves - 1) / waves;
    if (wgs > (uint64_t)wgPerCu * ctx->num_cus) wgs = (uint64_t)wgPerCu * ctx->num_cus;
    const dim3 grid((uint32_t)wgs);
    if (getenv("PT_VERBOSE"))
        fprintf(stderr, "pt_hip: render mode %d, %u waves/WG, LDS %zu B/WG, %d WG/CU, grid %u, %u tiles x %u chunks\n", mode, waves, lds, wgPerCu, grid.x, a.ntiles, nchunks);
    const uint32_t nslots = grid.x * waves;
    if (nslots > stat_slots) { ctx->err = "launch larger than the statistics buffer"; return PT_ERR_UNSUPPORTED; }
    // One launch of a context in flight at a time: the statistics words (the unit cursor among them), the partial images
    // and the two events are per context.  A launch on another stream waits for the previous one here.
    // Under stream capture (a HIP graph being recorded) the context's events stay out of the graph: no wait on the previous
    // launch, no timing records; pt_hip_last_stats then reports nothing for the captured launch.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (stream) (void)hipStreamIsCapturing(stream, &capture);
    const bool capturing = capture != hipStreamCaptureStatusNone;
    if (ctx->timed && !capturing) HIPCHK(ctx, hipStreamWaitEvent(stream, ctx->ev1, 0));
    HIPCHK(ctx, hipMemsetAsync(stats_dev, 0, (size_t)kStatWords * (1u + nslots) * sizeof(unsigned long long), stream));
    if (a.accumulate && !addTo) HIPCHK(ctx, hipMemsetAsync(accum_dev, 0, (size_t)frameBytes, stream));
    if (!capturing) HIPCHK(ctx, hipEventRecord(ctx->ev0, stream));
    {
        const RenderKernel kern = render_kernel(ctx->moving, mode, a.sc.exit_lanes != 0, !(p->flags & PT_FLAG_REPRODUCIBLE));
        st = set_lds(ctx, kern, lds);
        if (st != PT_OK) return st;
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    }
    HIPCHK(ctx, hipGetLastError());
    if (a.partial)
    {
        const uint32_t n = p->width * p->height;
        uint32_t blocks = (n + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_fold_chunks, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(a.partial), nchunks, n, reinterpret_cast<float4*>(accum_dev), addTo ? 1u : 0u);
    }
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_sum_stats, dim3((nslots + 255u) / 256u), dim3(256), 0, stream, stats_dev, nslots);
    HIPCHK(ctx, hipGetLastError());
    if (!capturing) HIPCHK(ctx, hipEventRecord(ctx->ev1, stream));
    ctx->timed = !capturing;
    ctx->last_stats_dev = stats_dev;
    ctx->last_mode = mode;
    ctx->last = pt_stats{};
    ctx->last.samples = (uint64_t)p->width * p->height * a.nspp;
    ctx->last.kernel = mode_in_hbm(mode) ? 1u : mode == MODE_BVH_LDS ? 3u : mode == MODE_LDS_STRICT ? 2u : mode == MODE_REF ? 4u : mode == MODE_FEW ? 5u : 0u;
    ctx->last.node_bytes = mode == MODE_BVH_Q ? 32u : mode_is_bvh(mode) ? 64u : 0u;
    ctx->last.node_width = mode == MODE_BVH_Q4 ? 4u : mode_is_bvh(mode) ? 2u : 0u;
    return PT_OK;
}

int pt_hip_render_device(pt_ctx* ctx, const pt_params* p, float* accum_dev, void* stream_)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_render_device")
    return render_launch(ctx, p, accum_dev, (hipStream_t)stream_, false);
    PT_LEAVE
}

int pt_hip_render_device_static(pt_ctx* ctx, const pt_params* p, float* accum_dev, void* stream_)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_render_device_static")
    return render_launch(ctx, p, accum_dev, (hipStream_t)stream_, true);
    PT_LEAVE
}

int pt_hip_render(pt_ctx* ctx, const pt_params* p, float* accum_host)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_render")
    int st = check_scene(ctx);
    if (st != PT_OK) return st;
    if (!p || !accum_host || p->width == 0 || p->height == 0) return PT_ERR_INVALID;
    Bind b(ctx);
    const size_t floats = (size_t)p->width * p->height * 4;
    if (floats > ctx->accum_floats)
    {
        if (ctx->accum_dev) { (void)hipFree(ctx->accum_dev); ctx->accum_dev = nullptr; ctx->accum_floats = 0; }
        HIPCHK(ctx, hipMalloc((void**)&ctx->accum_dev, floats * sizeof(float)));
        ctx->accum_floats = floats;
    }
    st = pt_hip_render_device(ctx, p, ctx->accum_dev, nullptr);
    if (st != PT_OK) return st;
    HIPCHK(ctx, hipMemcpy(accum_host, ctx->accum_dev, floats * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
    PT_LEAVE
}

int pt_hip_last_stats(pt_ctx* ctx, pt_stats* out)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_last_stats")
    if (!ctx || !out) return PT_ERR_INVALID;
    if (!ctx->timed) return PT_ERR_NO_SCENE;
    Bind b(ctx);
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.0f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    unsigned long long st[8];
    HIPCHK(ctx, hipMemcpy(st, ctx->last_stats_dev, sizeof(st), hipMemcpyDeviceToHost));
    ctx->last.kernel_ms = ms;
    ctx->last.segments = st[0];
    ctx->last.wave_iterations = st[1];
    ctx->last.phase_cycles[0] = st[2]; ctx->last.phase_cycles[1] = st[3]; ctx->last.phase_cycles[2] = st[4];
    if (mode_is_bvh(ctx->last_mode) || ctx->last_mode == MODE_REF)
    {
        ctx->last.node_visits = st[6];   // inner nodes fetched (two box tests each; reference tree: one hit_test each)
        ctx->last.sphere_tests = st[7];  // exact leaf tests
    }
    else if (ctx->last_mode == MODE_FEW)
    {
        ctx->last.sphere_tests = st[0] * (uint64_t)ctx->desc.n_spheres; // every segment meets every sphere (filter; exact test for the candidates)
        ctx->last.node_visits = 0;
    }
    else
    {
        ctx->last.sphere_tests = st[5] * 8ull * 64ull;                          // wave-level group searches x 8 slots x 64 lanes
        ctx->last.node_visits = st[1] * 64ull * (uint64_t)ctx->sc_lds.ngroups;   // bound tests: every lane, every group, every trip
    }
    *out = ctx->last;
    return PT_OK;
    PT_LEAVE
}

uint32_t pt_chunk_schedule(uint32_t nspp, uint32_t ratio_q16, uint32_t min_spp, uint32_t* out_sizes, uint32_t cap)
{
    if (min_spp == 0) min_spp = 1;
    uint32_t n = 0;
    for (uint32_t s0 = 0; s0 < nspp; ++n)
    {
        const uint32_t u = chunk_size(nspp - s0, ratio_q16, min_spp);
        if (out_sizes && n < cap) out_sizes[n] = u;
        s0 += u;
    }
    return n;
}

int pt_bvh_selfcheck(const pt_scene_desc* d, uint64_t out8[8])
{
    PT_ENTER(nullptr, "pt_bvh_selfcheck")
    if (!d || !out8) return PT_ERR_INVALID;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    pth::BvhData bv;
    pth::build_bvh(d, &bv);
    pth::NodeGrid grid{};
    std::vector<uint32_t> qn;
    pth::quantise_nodes(bv, &grid, &qn);
    out8[0] = bv.n_nodes; out8[1] = bv.depth;
    const bool moving = bv.moving;
    // a sphere (slot order) against a box: its swept extent t0..t1, unpadded -- the builder's padding only adds room
    auto sphere_inside = [&](uint32_t slot, const float* lo, const float* hi) {
        const float* sp = bv.sph.data() + 4 * (size_t)slot;
        const float r = sqrtf(sp[3]);
        for (int k = 0; k < 3; ++k)
        {
            float a = sp[k], b2 = sp[k];
            if (moving) { const float v = bv.vel[4 * (size_t)slot + k]; a = sp[k] + v * d->t0; b2 = sp[k] + v * d->t1; }
            if (std::min(a, b2) - r < lo[k] || std::max(a, b2) + r > hi[k]) return false;
        }
        return true;
    };
    auto visit_leaf = [&](uint32_t ref, const float* lo, const float* hi) {
        const uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 7u) + 1u;
        ++out8[2];
        if (count > out8[3]) out8[3] = count;
        if ((uint64_t)first + count > bv.n_spheres) { ++out8[7]; return; }
        for (uint32_t i = 0; i < count; ++i) { ++out8[4]; if (lo && !sphere_inside(first + i, lo, hi)) ++out8[5]; }
    };
    if (bv.n_spheres != 0)
    {
        if (bv.n_nodes == 0) visit_leaf(bv.root_ref, nullptr, nullptr);
        else
        {
            std::vector<uint32_t> stack{bv.root_ref};
            std::vector<char> seen(bv.n_nodes, 0);
            while (!stack.empty())
            {
                const uint32_t node = stack.back(); stack.pop_back();
                if (node >= bv.n_nodes || seen[node]) { ++out8[7]; continue; }
                seen[node] = 1;
                const float* nd = bv.nodes.data() + 16 * (size_t)node;
                const uint32_t* q = qn.data() + 8 * (size_t)node;
                for (int side = 0; side < 2; ++side)
                {
                    const float* lo = nd + 6 * side; const float* hi = nd + 6 * side + 3;
                    for (int k = 0; k < 3; ++k) // the grid box, decoded exactly, must enclose the float box
                    {
                        const uint32_t w = q[3 * side + k];
                        const double glo = (double)grid.org[k] + (double)(w & 0xFFFFu) * (double)grid.step[k];
                        const double ghi = (double)grid.org[k] + (double)(w >> 16) * (double)grid.step[k];
                        if (glo > (double)lo[k] || ghi < (double)hi[k]) ++out8[6];
                    }
                    const uint32_t ref = pth::f2u(nd[12 + side]);
                    if (ref != q[6 + side]) ++out8[7];
                    if (ref & pth::kLeafFlag) visit_leaf(ref, lo, hi); else stack.push_back(ref);
                }
            }
        }
    }
    // the 4-wide grid nodes, both flavours (leaves of the binary tree / single spheres): every sphere under exactly one
    // leaf reference and inside the decoded box of the child that leads to it, references in range
    for (int single = 0; single < 2 && bv.n_spheres != 0; ++single)
    {
        pth::NodeGrid wg{}; pth::WideData wd;
        pth::widen_nodes(bv, d, single != 0, &wg, &wd);
        if (getenv("PT_VERBOSE")) fprintf(stderr, "pt_bvh_selfcheck: binary %u nodes depth %u; wide(%d) %u nodes depth %u\n", bv.n_nodes, bv.depth, single, wd.n_nodes, wd.depth);
        uint64_t found = 0;
        auto wide_leaf = [&](uint32_t ref, const float* lo, const float* hi) {
            const uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 7u) + 1u;
            if ((uint64_t)first + count > bv.n_spheres || (single && count != 1 && wd.n_nodes != 0)) { ++out8[7]; return; }
            for (uint32_t i = 0; i < count; ++i) { ++found; if (lo && !sphere_inside(first + i, lo, hi)) ++out8[5]; }
        };
        if (wd.n_nodes == 0) wide_leaf(wd.root_ref, nullptr, nullptr);
        else
        {
            std::vector<uint32_t> stack{wd.root_ref};
            std::vector<char> seen(wd.n_nodes, 0);
            uint64_t visited = 0;
            while (!stack.empty())
            {
                const uint32_t node = stack.back(); stack.pop_back();
                if (node >= wd.n_nodes || seen[node]) { ++out8[7]; continue; }
                seen[node] = 1; ++visited;
                const uint32_t* q = wd.nodes.data() + 16 * (size_t)node;
                for (int j = 0; j < 4; ++j)
                {
                    const uint32_t ref = q[12 + j];
                    if (j > 0 && ref == q[12] && q[3 * j] == 0u && q[3 * j + 1] == 0u && q[3 * j + 2] == 0u) continue; // an empty slot (pt_bvh.h)
                    float lo[3], hi[3];
                    for (int k = 0; k < 3; ++k)
                    {
                        // what the device's arithmetic makes of the cell can differ from this by rounding; the builder's pad covers it
                        lo[k] = (float)((double)wg.org[k] + (double)(q[3 * j + k] & 0xFFFFu) * (double)wg.step[k]);
                        hi[k] = (float)((double)wg.org[k] + (double)(q[3 * j + k] >> 16) * (double)wg.step[k]);
                    }
                    if (ref & pth::kLeafFlag) wide_leaf(ref, lo, hi); else stack.push_back(ref);
                }
            }
            if (visited != wd.n_nodes) ++out8[7];
        }
        if (found != bv.n_spheres) ++out8[7];
    }
    return (out8[4] == bv.n_spheres && out8[5] == 0 && out8[6] == 0 && out8[7] == 0) ? PT_OK : PT_ERR_INVALID;
    PT_LEAVE
}

int pt_hip_debug_counters(pt_ctx* ctx, uint64_t out16[16])
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_debug_counters")
    if (!ctx || !out16) return PT_ERR_INVALID;
    if (!ctx->timed) return PT_ERR_NO_SCENE;
    Bind b(ctx);
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipMemcpy(out16, ctx->last_stats_dev, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return PT_OK;
    PT_LEAVE
}

int pt_hip_resolve_device(pt_ctx* ctx, const float* accum_dev, uint32_t n_pixels, uint32_t spp_total, float* pixels_dev, uint32_t* bgra_dev, void* stream_)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_resolve_device")
    if (!ctx || !accum_dev || n_pixels == 0 || spp_total == 0) return PT_ERR_INVALID;
    Bind b(ctx);
    const float inv = 1.0f / (float)spp_total; // vectors.cpp:100-103
    uint32_t blocks = (n_pixels + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, reinterpret_cast<const float4*>(accum_dev), n_pixels, inv,
                       reinterpret_cast<float4*>(pixels_dev), bgra_dev);
    HIPCHK(ctx, hipGetLastError());
    return PT_OK;
    PT_LEAVE
}

// ------------------------------------------------------------------------------------------------ probes
int pt_hip_probe_primitives(pt_ctx* ctx, uint32_t n, const float* rays, const float* spheres, const float* planes, float* out_ts, float* out_tp)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_probe_primitives")
    if (!ctx || !rays || !spheres || !planes || !out_ts || !out_tp) return PT_ERR_INVALID;
    if (n == 0) return PT_OK;
    Bind b(ctx);
    DevBuf dr, ds, dp, ots, otp;
    HIPCHK(ctx, dr.alloc(n * 24ull)); HIPCHK(ctx, ds.alloc(n * 16ull)); HIPCHK(ctx, dp.alloc(n * 16ull)); HIPCHK(ctx, ots.alloc(n * 4ull)); HIPCHK(ctx, otp.alloc(n * 4ull));
    HIPCHK(ctx, hipMemcpy(dr.p, rays, n * 24ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ds.p, spheres, n * 16ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(dp.p, planes, n * 16ull, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_primitives, dim3((n + 255) / 256), dim3(256), 0, 0, n, dr.as<float>(), ds.as<float>(), dp.as<float>(), ots.as<float>(), otp.as<float>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemcpy(out_ts, ots.p, n * 4ull, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(out_tp, otp.p, n * 4ull, hipMemcpyDeviceToHost));
    return PT_OK;
    PT_LEAVE
}

int pt_hip_probe_hits(pt_ctx* ctx, uint32_t n, const float* rays, const float* times, uint32_t flags,
                      float* out_t, int32_t* out_prim, float* out_point, float* out_normal)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_probe_hits")
    int st = check_scene(ctx);
    if (st != PT_OK) return st;
    if (!rays || !times || !out_t || !out_prim || !out_point || !out_normal) return PT_ERR_INVALID;
    if (n == 0) return PT_OK;
    Bind b(ctx);
    int mode; const SceneArgs* sc; size_t lds;
    st = select_mode(ctx, flags, 4, false, &mode, &sc, &lds);
    if (st != PT_OK) return st;
    DevBuf dr, dt, ot, op, opt, on;
    HIPCHK(ctx, dr.alloc(n * 24ull)); HIPCHK(ctx, dt.alloc(n * 4ull)); HIPCHK(ctx, ot.alloc(n * 4ull));
    HIPCHK(ctx, op.alloc(n * 4ull)); HIPCHK(ctx, opt.alloc(n * 12ull)); HIPCHK(ctx, on.alloc(n * 12ull));
    HIPCHK(ctx, hipMemcpy(dr.p, rays, n * 24ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(dt.p, times, n * 4ull, hipMemcpyHostToDevice));
    ProbeHitsArgs a{*sc, n, dr.as<float>(), dt.as<float>(), ot.as<float>(), op.as<int32_t>(), opt.as<float>(), on.as<float>()};
    const dim3 grid((n + 255) / 256), block(256);
    PT_LAUNCH(k_probe_hits, ctx->moving, mode, grid, block, lds, 0, a);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemcpy(out_t, ot.p, n * 4ull, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(out_prim, op.p, n * 4ull, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(out_point, opt.p, n * 12ull, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(out_normal, on.p, n * 12ull, hipMemcpyDeviceToHost));
    return PT_OK;
    PT_LEAVE
}

int pt_hip_probe_get_rays(pt_ctx* ctx, uint32_t n, const float* uv, const float* tape, const uint64_t* ofs, float* out_rays)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_probe_get_rays")
    int st = check_scene(ctx);
    if (st != PT_OK) return st;
    if (!uv || !tape || !ofs || !out_rays) return PT_ERR_INVALID;
    if (n == 0) return PT_OK;
    Bind b(ctx);
    const uint64_t ntape = ofs[n];
    DevBuf duv, dtape, dofs, dout;
    HIPCHK(ctx, duv.alloc(n * 8ull)); HIPCHK(ctx, dtape.alloc(ntape * 4ull)); HIPCHK(ctx, dofs.alloc((n + 1ull) * 8)); HIPCHK(ctx, dout.alloc(n * 24ull));
    HIPCHK(ctx, hipMemcpy(duv.p, uv, n * 8ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(dtape.p, tape, ntape * 4ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(dofs.p, ofs, (n + 1ull) * 8, hipMemcpyHostToDevice));
    ProbeRaysArgs a{ctx->cam, n, duv.as<float>(), dtape.as<float>(), dofs.as<unsigned long long>(), dout.as<float>()};
    hipLaunchKernelGGL(k_probe_get_rays, dim3((n + 255) / 256), dim3(256), 0, 0, a);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemcpy(out_rays, dout.p, n * 24ull, hipMemcpyDeviceToHost));
    return PT_OK;
    PT_LEAVE
}

int pt_hip_probe_scatter(pt_ctx* ctx, uint32_t n, const float* dirs, const float* points, const float* normals,
                         const uint32_t* mat_type, const float* mat_rgba, const float* mat_param,
                         const float* tape32, const uint64_t* ofs32, const double* tape64, const uint64_t* ofs64,
                         float* out_dir, uint32_t* out_event)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_probe_scatter")
    if (!ctx) return PT_ERR_INVALID;
    if (!dirs || !points || !normals || !mat_type || !mat_rgba || !mat_param || !tape32 || !ofs32 || !tape64 || !ofs64 || !out_dir || !out_event) return PT_ERR_INVALID;
    if (n == 0) return PT_OK;
    Bind b(ctx);
    const uint64_t n32 = ofs32[n], n64 = ofs64[n];
    DevBuf dd, dp, dn, dty, drg, dpa, dt32, do32, dt64, do64, od, oe;
    HIPCHK(ctx, dd.alloc(n * 12ull)); HIPCHK(ctx, dp.alloc(n * 12ull)); HIPCHK(ctx, dn.alloc(n * 12ull));
    HIPCHK(ctx, dty.alloc(n * 4ull)); HIPCHK(ctx, drg.alloc(n * 16ull)); HIPCHK(ctx, dpa.alloc(n * 8ull));
    HIPCHK(ctx, dt32.alloc(n32 * 4)); HIPCHK(ctx, do32.alloc((n + 1ull) * 8)); HIPCHK(ctx, dt64.alloc(n64 * 8)); HIPCHK(ctx, do64.alloc((n + 1ull) * 8));
    HIPCHK(ctx, od.alloc(n * 12ull)); HIPCHK(ctx, oe.alloc(n * 4ull));
    HIPCHK(ctx, hipMemcpy(dd.p, dirs, n * 12ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(dp.p, points, n * 12ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(dn.p, normals, n * 12ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(dty.p, mat_type, n * 4ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(drg.p, mat_rgba, n * 16ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(dpa.p, mat_param, n * 8ull, hipMemcpyHostToDevice));
    if (n32) HIPCHK(ctx, hipMemcpy(dt32.p, tape32, n32 * 4, hipMemcpyHostToDevice));
    if (n64) HIPCHK(ctx, hipMemcpy(dt64.p, tape64, n64 * 8, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(do32.p, ofs32, (n + 1ull) * 8, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(do64.p, ofs64, (n + 1ull) * 8, hipMemcpyHostToDevice));
    ProbeScatterArgs a{n, dd.as<float>(), dp.as<float>(), dn.as<float>(), dty.as<uint32_t>(), drg.as<float>(), dpa.as<float>(),
                       dt32.as<float>(), do32.as<unsigned long long>(), dt64.as<double>(), do64.as<unsigned long long>(), od.as<float>(), oe.as<uint32_t>()};
    hipLaunchKernelGGL(k_probe_scatter, dim3((n + 255) / 256), dim3(256), 0, 0, a);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemcpy(out_dir, od.p, n * 12ull, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(out_event, oe.p, n * 4ull, hipMemcpyDeviceToHost));
    return PT_OK;
    PT_LEAVE
}

int pt_hip_probe_paths(pt_ctx* ctx, uint32_t n, const float* rays, const float* times, uint32_t max_depth, uint32_t flags,
                       const float* tape32, const uint64_t* ofs32, const double* tape64, const uint64_t* ofs64,
                       float* out_rgba, uint32_t* out_segments)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_probe_paths")
    int st = check_scene(ctx);
    if (st != PT_OK) return st;
    if (!rays || !times || !tape32 || !ofs32 || !tape64 || !ofs64 || !out_rgba) return PT_ERR_INVALID;
    if (n == 0) return PT_OK;
    Bind b(ctx);
    int mode; const SceneArgs* sc; size_t lds;
    st = select_mode(ctx, flags, 4, false, &mode, &sc, &lds);
    if (st != PT_OK) return st;
    const uint64_t n32 = ofs32[n], n64 = ofs64[n];
    DevBuf dr, dt, dt32, do32, dt64, do64, oc, os;
    HIPCHK(ctx, dr.alloc(n * 24ull)); HIPCHK(ctx, dt.alloc(n * 4ull));
    HIPCHK(ctx, dt32.alloc(n32 * 4)); HIPCHK(ctx, do32.alloc((n + 1ull) * 8)); HIPCHK(ctx, dt64.alloc(n64 * 8)); HIPCHK(ctx, do64.alloc((n + 1ull) * 8));
    HIPCHK(ctx, oc.alloc(n * 16ull)); HIPCHK(ctx, os.alloc(n * 4ull));
    HIPCHK(ctx, hipMemcpy(dr.p, rays, n * 24ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(dt.p, times, n * 4ull, hipMemcpyHostToDevice));
    if (n32) HIPCHK(ctx, hipMemcpy(dt32.p, tape32, n32 * 4, hipMemcpyHostToDevice));
    if (n64) HIPCHK(ctx, hipMemcpy(dt64.p, tape64, n64 * 8, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(do32.p, ofs32, (n + 1ull) * 8, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(do64.p, ofs64, (n + 1ull) * 8, hipMemcpyHostToDevice));
    ProbePathsArgs a{*sc, n, max_depth, dr.as<float>(), dt.as<float>(), dt32.as<float>(), do32.as<unsigned long long>(),
                     dt64.as<double>(), do64.as<unsigned long long>(), oc.as<float>(), os.as<uint32_t>()};
    const dim3 grid((n + 255) / 256), block(256);
    PT_LAUNCH(k_probe_paths, ctx->moving, mode, grid, block, lds, 0, a);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemcpy(out_rgba, oc.p, n * 16ull, hipMemcpyDeviceToHost));
    if (out_segments) HIPCHK(ctx, hipMemcpy(out_segments, os.p, n * 4ull, hipMemcpyDeviceToHost));
    return PT_OK;
    PT_LEAVE
}

int pt_hip_probe_samples(pt_ctx* ctx, const pt_params* p, uint32_t n, const uint32_t* pixels, const uint32_t* samples,
                         float* out_rgba, uint32_t* out_segments)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_probe_samples")
    int st = check_scene(ctx);
    if (st != PT_OK) return st;
    if (!p || !pixels || !samples || !out_rgba || p->width == 0 || p->height == 0) return PT_ERR_INVALID;
    if (n == 0) return PT_OK;
    for (uint32_t i = 0; i < n; ++i) if (pixels[i] >= (uint64_t)p->width * p->height) return PT_ERR_INVALID;
    Bind b(ctx);
    int mode; const SceneArgs* sc; size_t lds;
    st = select_mode(ctx, p->flags, 4, false, &mode, &sc, &lds);
    if (st != PT_OK) return st;
    DevBuf dp, ds, oc, os;
    HIPCHK(ctx, dp.alloc(n * 4ull)); HIPCHK(ctx, ds.alloc(n * 4ull)); HIPCHK(ctx, oc.alloc(n * 16ull)); HIPCHK(ctx, os.alloc(n * 4ull));
    HIPCHK(ctx, hipMemcpy(dp.p, pixels, n * 4ull, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ds.p, samples, n * 4ull, hipMemcpyHostToDevice));
    ProbeSamplesArgs a{*sc, n, p->width, p->height, p->max_depth, p->seed, dp.as<uint32_t>(), ds.as<uint32_t>(), oc.as<float>(), os.as<uint32_t>()};
    const dim3 grid((n + 255) / 256), block(256);
    PT_LAUNCH(k_probe_samples, ctx->moving, mode, grid, block, lds, 0, a);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemcpy(out_rgba, oc.p, n * 16ull, hipMemcpyDeviceToHost));
    if (out_segments) HIPCHK(ctx, hipMemcpy(out_segments, os.p, n * 4ull, hipMemcpyDeviceToHost));
    return PT_OK;
    PT_LEAVE
}

int pt_hip_probe_philox(pt_ctx* ctx, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t* out_words)
{
    PT_ENTER(ctx ? &ctx->err : nullptr, "pt_hip_probe_philox")
    if (!ctx || !out_words) return PT_ERR_INVALID;
    if (n == 0) return PT_OK;
    Bind b(ctx);
    DevBuf o;
    HIPCHK(ctx, o.alloc(n * 4ull));
    hipLaunchKernelGGL(k_probe_philox, dim3(1), dim3(64), 0, 0, seed, pixel, sample, n, o.as<uint32_t>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemcpy(out_words, o.p, n * 4ull, hipMemcpyDeviceToHost));
    return PT_OK;
    PT_LEAVE
}

} // extern "C"
