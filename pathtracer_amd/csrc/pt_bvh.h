// pt_bvh.h -- host-side BVH builder for scenes that do not fit in LDS (C5: 10^6 spheres; the reference's scene 2).
//
// Takes the place of build_bvh_tree / sort_render_objects (reference src/main.cpp:93-159) -- a recursive random-axis
// median split over an in-place quicksort with first-element pivot, one VirtualAlloc per node -- with a design of our
// own: binned-SAH top-down build over swept boxes, iterative (explicit stack), O(N log N), leaves of up to 4 spheres,
// nodes linearised depth-first into one array for HBM.  Topology is free: the build defines the sphere query as the
// true closest hit, so any tree that encloses every sphere gives the same answer.
//
// Node = 4 x float4 = 64 B, the two CHILD boxes live in the parent so one fetch decides both children:
//   n0 = {L.min.x, L.min.y, L.min.z, L.max.x}   n1 = {L.max.y, L.max.z, R.min.x, R.min.y}
//   n2 = {R.min.z, R.max.x, R.max.y, R.max.z}   n3 = {bits(leftRef), bits(rightRef), 0, 0}
//   ref: inner node -> node index; leaf -> 0x80000000 | (count-1) << 28 | first   (count <= 8, first < 2^28)
// Spheres are re-ordered so that a leaf's spheres are contiguous: sph[i] = {cx,cy,cz,r2}, vel[i] = {vx,vy,vz,0},
// mat[2i] = {type, roughness, n, bits(original index)}, mat[2i+1] = rgba.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/pt_hip.h"
#include "pt_guard.h"

namespace pth {

struct BvhData
{
    std::vector<float> nodes;  // 16 floats per node
    std::vector<float> sph;    // 4 per sphere (sorted order)
    std::vector<float> vel;    // 4 per sphere (empty if static)
    std::vector<float> mat;    // 8 per sphere
    uint32_t n_nodes = 0, n_spheres = 0, depth = 0, root_ref = 0;
    bool moving = false;
};

struct Aabb { float lo[3], hi[3]; };
inline void aabb_reset(Aabb& b) { for (int k = 0; k < 3; ++k) { b.lo[k] = 3.0e38f; b.hi[k] = -3.0e38f; } }
inline void aabb_grow(Aabb& b, const Aabb& o) { for (int k = 0; k < 3; ++k) { if (o.lo[k] < b.lo[k]) b.lo[k] = o.lo[k]; if (o.hi[k] > b.hi[k]) b.hi[k] = o.hi[k]; } }
inline float aabb_area(const Aabb& b)
{
    float e[3] = {b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]};
    if (e[0] < 0) return 0.0f;
    return 2.0f * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
}
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

constexpr uint32_t kLeafMaxDefault = 4; // spheres per leaf (<= 8: three bits of the leaf reference)
constexpr uint32_t kLeafFlag = 0x80000000u;

inline void build_bvh(const pt_scene_desc* d, BvhData* out)
{
    const uint32_t n = d->n_spheres;
    uint32_t kLeafMax = kLeafMaxDefault;
    if (const char* e = getenv("PT_LEAF_MAX")) { int v = atoi(e); if (v >= 1 && v <= 8) kLeafMax = (uint32_t)v; } // tuning experiments only
    out->n_spheres = n;
    out->moving = false;
    if (d->sphere_vel)
        for (size_t i = 0; i < 3 * (size_t)n; ++i)
            if (d->sphere_vel[i] != 0.0f) { out->moving = true; break; }
    // swept, padded boxes: the device's slab test is ordinary f32 arithmetic whose rounding error in position space is
    // a few ulp of the largest coordinate involved (ray origins lie on scene surfaces or at the camera); the leaf test
    // is the reference's exact one.  The pad -- 16 ulp of the scene extent plus 1e-5 of the radius -- keeps every sphere
    // the exact test could hit inside its box.
    float extent = 1.0f;
    for (int k = 0; k < 3; ++k) if (fabsf(d->camera[k]) > extent) extent = fabsf(d->camera[k]);
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
        {
            float a = fabsf(d->sphere_pos_r[4 * (size_t)i + k]) + fabsf(d->sphere_pos_r[4 * (size_t)i + 3]);
            if (out->moving) a += fabsf(d->sphere_vel[3 * (size_t)i + k]) * (fabsf(d->t0) > fabsf(d->t1) ? fabsf(d->t0) : fabsf(d->t1));
            if (a > extent) extent = a;
        }
    const float extentPad = extent * 1.0e-6f;
    std::vector<Aabb> box(n);
    std::vector<float> cen(3 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i)
    {
        const float* s = d->sphere_pos_r + 4 * (size_t)i;
        const float r = fabsf(s[3]);
        for (int k = 0; k < 3; ++k)
        {
            float a = s[k], b2 = s[k];
            if (out->moving) { float v = d->sphere_vel[3 * (size_t)i + k]; a = s[k] + v * d->t0; b2 = s[k] + v * d->t1; }
            float lo = (a < b2 ? a : b2) - r, hi = (a > b2 ? a : b2) + r;
            float pad = extentPad + 1e-5f * r;
            box[i].lo[k] = lo - pad; box[i].hi[k] = hi + pad;
            cen[3 * (size_t)i + k] = 0.5f * (lo + hi);
        }
    }
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;

    // ---- top-down binned-SAH build.  One split = centroid bounds -> 16 bins x 3 axes -> best plane -> partition of idx.
    // Parallel in two ways (worker count = PT_BUILD_THREADS, else min(hardware threads, 16)): the passes over a BIG range
    // (the top of the tree) are cut into per-thread slices; once the frontier holds enough subtrees they are built
    // independently, each into its own node array, and spliced into the final array afterwards.
    struct Task { uint32_t begin, end, parent, side, depth; }; // parent = node index in the SAME array (or kNoParent), side 0/1
    constexpr uint32_t kNoParent = 0xFFFFFFFFu;
    constexpr int NB = 16;
    unsigned nthreads = ptg::host_threads(16); // (affinity mask and cgroup quota respected; a thread that cannot start runs inline)
    if (n < 65536) nthreads = 1;
    // ranges of up to sweepMax spheres get the exact SAH sweep instead of the bins (below) -- in scenes small enough that
    // the three sorts per node do not show in the upload (book-1: 7.06 -> 6.82 node visits per segment, +0.7 %; a 10^6-sphere
    // build keeps its bins all the way down)
    uint32_t sweepMax = n <= 8192u ? 512u : 0u; bool sweepQuant = false;
    if (const char* e = getenv("PT_SAH_SWEEP")) { int v = atoi(e); if (v >= 0) sweepMax = (uint32_t)v; }   // tuning experiments only
    if (const char* e = getenv("PT_SAH_QUANT")) sweepQuant = atoi(e) != 0;                                  // (leaves priced per block: fewer nodes, more leaf visits, -3.5 %)
    out->nodes.clear();
    out->depth = 0;
    auto make_leaf_ref = [](uint32_t begin, uint32_t count) { return kLeafFlag | ((count - 1u) << 28) | begin; };
    auto range_box = [&](uint32_t b, uint32_t e) { Aabb r; aabb_reset(r); for (uint32_t i = b; i < e; ++i) aabb_grow(r, box[idx[i]]); return r; };
    auto set_child = [](std::vector<float>& nodes, uint32_t parent, uint32_t side, uint32_t ref, const Aabb& bx) {
        float* nd = nodes.data() + 16 * (size_t)parent;
        if (side == 0) { nd[0] = bx.lo[0]; nd[1] = bx.lo[1]; nd[2] = bx.lo[2]; nd[3] = bx.hi[0]; nd[4] = bx.hi[1]; nd[5] = bx.hi[2]; nd[12] = u2f(ref); }
        else { nd[6] = bx.lo[0]; nd[7] = bx.lo[1]; nd[8] = bx.lo[2]; nd[9] = bx.hi[0]; nd[10] = bx.hi[1]; nd[11] = bx.hi[2]; nd[13] = u2f(ref); }
    };
    // run f(slice_begin, slice_end, slice_index) over [b, e) on `workers` threads (1 = inline)
    auto parallel_slices = [&](uint32_t b, uint32_t e, unsigned workers, const std::function<void(uint32_t, uint32_t, unsigned)>& f) {
        if (workers <= 1 || e - b < 32768) { f(b, e, 0); return 1u; }
        const uint64_t len = e - b;
        ptg::run_workers(workers, [&](unsigned w) { f(b + (uint32_t)(len * w / workers), b + (uint32_t)(len * (w + 1) / workers), w); });
        return workers;
    };
    struct Bins { Aabb bb[3][NB]; uint32_t bc[3][NB]; float clo[3], chi[3]; };
    // One split of task t inside node array `nodes`: allocates the inner node, links it to its parent, partitions idx and
    // appends the (up to two) child tasks to `todo`.  `workers` > 1 slices the big passes.
    auto split = [&](const Task& t, std::vector<float>& nodes, std::vector<Task>& todo, uint32_t* maxDepth, unsigned workers) {
        if (t.depth > *maxDepth) *maxDepth = t.depth;
        const uint32_t count = t.end - t.begin;
        const uint32_t self = (uint32_t)(nodes.size() / 16);
        nodes.resize(nodes.size() + 16, 0.0f);
        if (t.parent != kNoParent) set_child(nodes, t.parent, t.side, self, range_box(t.begin, t.end));
        // centroid bounds
        std::vector<Bins> part(workers > 1 ? workers : 1);
        unsigned used = parallel_slices(t.begin, t.end, workers, [&](uint32_t b, uint32_t e, unsigned w) {
            Bins& P = part[w];
            for (int k = 0; k < 3; ++k) { P.clo[k] = 3e38f; P.chi[k] = -3e38f; }
            for (uint32_t i = b; i < e; ++i)
                for (int k = 0; k < 3; ++k) { float c = cen[3 * (size_t)idx[i] + k]; if (c < P.clo[k]) P.clo[k] = c; if (c > P.chi[k]) P.chi[k] = c; }
        });
        float clo[3] = {3e38f, 3e38f, 3e38f}, chi[3] = {-3e38f, -3e38f, -3e38f};
        for (unsigned w = 0; w < used; ++w) for (int k = 0; k < 3; ++k) { if (part[w].clo[k] < clo[k]) clo[k] = part[w].clo[k]; if (part[w].chi[k] > chi[k]) chi[k] = part[w].chi[k]; }
        float scale[3];
        for (int k = 0; k < 3; ++k) scale[k] = chi[k] - clo[k] > 0 ? NB / (chi[k] - clo[k]) : 0.0f;
        auto bin_of = [&](uint32_t i, int k) { int b2 = (int)((cen[3 * (size_t)i + k] - clo[k]) * scale[k]); if (b2 >= NB) b2 = NB - 1; if (b2 < 0) b2 = 0; return b2; };
        // binning, all three axes in one pass
        used = parallel_slices(t.begin, t.end, workers, [&](uint32_t b, uint32_t e, unsigned w) {
            Bins& P = part[w];
            for (int k = 0; k < 3; ++k) for (int b2 = 0; b2 < NB; ++b2) { aabb_reset(P.bb[k][b2]); P.bc[k][b2] = 0; }
            for (uint32_t i = b; i < e; ++i)
                for (int k = 0; k < 3; ++k) { if (scale[k] == 0.0f) continue; const int b2 = bin_of(idx[i], k); aabb_grow(P.bb[k][b2], box[idx[i]]); ++P.bc[k][b2]; }
        });
        for (unsigned w = 1; w < used; ++w)
            for (int k = 0; k < 3; ++k) for (int b2 = 0; b2 < NB; ++b2) { aabb_grow(part[0].bb[k][b2], part[w].bb[k][b2]); part[0].bc[k][b2] += part[w].bc[k][b2]; }
        int bestAxis = -1, bestBin = -1; float bestCost = 3e38f;
        for (int k = 0; k < 3; ++k)
        {
            if (scale[k] == 0.0f) continue;
            const Aabb* bb = part[0].bb[k]; const uint32_t* bc = part[0].bc[k];
            float rightArea[NB]; uint32_t rightCount[NB];
            Aabb acc; aabb_reset(acc); uint32_t cnt = 0;
            for (int b2 = NB - 1; b2 > 0; --b2) { aabb_grow(acc, bb[b2]); cnt += bc[b2]; rightArea[b2] = aabb_area(acc); rightCount[b2] = cnt; }
            aabb_reset(acc); cnt = 0;
            for (int b2 = 0; b2 < NB - 1; ++b2)
            {
                aabb_grow(acc, bb[b2]); cnt += bc[b2];
                if (cnt == 0 || rightCount[b2 + 1] == 0) continue;
                float cost = aabb_area(acc) * cnt + rightArea[b2 + 1] * rightCount[b2 + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = k; bestBin = b2; }
            }
        }
        uint32_t mid = t.begin; // (stays there when all centroids coincide)
        // Small ranges of small scenes: the exact SAH sweep instead of the bins -- every split position of the range sorted
        // by centroid, on each axis.  (PT_SAH_QUANT prices a child by the leaf blocks it will need instead of its spheres.)
        bool swept = false;
        if (count <= sweepMax && workers <= 1)
        {
            std::vector<uint32_t> order(idx.begin() + t.begin, idx.begin() + t.end), bestOrder;
            std::vector<float> rightArea(count + 1);
            float sweepCost = 3e38f; uint32_t sweepAt = 0;
            auto price = [&](uint32_t c) { return sweepQuant ? (float)((c + kLeafMax - 1) / kLeafMax) : (float)c; };
            for (int k = 0; k < 3; ++k)
            {
                std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b2) { return cen[3 * (size_t)a + k] < cen[3 * (size_t)b2 + k]; });
                Aabb acc; aabb_reset(acc);
                for (uint32_t i = count; i-- > 0;) { aabb_grow(acc, box[order[i]]); rightArea[i] = aabb_area(acc); }
                aabb_reset(acc);
                for (uint32_t i = 1; i < count; ++i)
                {
                    aabb_grow(acc, box[order[i - 1]]);
                    const float cost = aabb_area(acc) * price(i) + rightArea[i] * price(count - i);
                    if (cost < sweepCost) { sweepCost = cost; sweepAt = i; bestOrder = order; }
                }
            }
            if (sweepAt != 0)
            {
                std::copy(bestOrder.begin(), bestOrder.end(), idx.begin() + t.begin);
                mid = t.begin + sweepAt;
                swept = true;
            }
        }
        if (!swept && bestAxis >= 0)
        {
            const int k = bestAxis;
            uint32_t* m = std::partition(idx.data() + t.begin, idx.data() + t.end, [&](uint32_t i) { return bin_of(i, k) <= bestBin; });
            mid = (uint32_t)(m - idx.data());
        }
        if (mid == t.begin || mid == t.end) // no useful plane: median split along the widest axis
        {
            mid = t.begin + count / 2;
            int k = 0; float e = -1;
            for (int a = 0; a < 3; ++a) if (chi[a] - clo[a] > e) { e = chi[a] - clo[a]; k = a; }
            std::nth_element(idx.begin() + t.begin, idx.begin() + mid, idx.begin() + t.end,
                             [&](uint32_t a, uint32_t b2) { return cen[3 * (size_t)a + k] < cen[3 * (size_t)b2 + k]; });
        }
        const uint32_t rng[2][2] = {{t.begin, mid}, {mid, t.end}};
        for (int side = 1; side >= 0; --side) // right first: the left subtree is laid out next (depth-first)
        {
            const uint32_t b2 = rng[side][0], e = rng[side][1];
            if (e - b2 <= kLeafMax) set_child(nodes, self, (uint32_t)side, make_leaf_ref(b2, e - b2), range_box(b2, e));
            else todo.push_back({b2, e, self, (uint32_t)side, t.depth + 1});
        }
    };
    if (n == 0) { out->root_ref = kLeafFlag; out->n_nodes = 0; return; } // leaf of "1 sphere at 0" is never used: n_spheres == 0 short-cuts
    if (n <= kLeafMax) { out->root_ref = make_leaf_ref(0, n); }
    else
    {
        out->root_ref = 0; // root inner node = node 0
        std::vector<Task> frontier;
        frontier.push_back({0, n, kNoParent, 0, 1});
        // phase A: split the biggest frontier task (sliced passes) until there are enough subtrees for everyone
        while (nthreads > 1 && frontier.size() < 4 * (size_t)nthreads)
        {
            size_t big = 0;
            for (size_t i = 1; i < frontier.size(); ++i) if (frontier[i].end - frontier[i].begin > frontier[big].end - frontier[big].begin) big = i;
            if (frontier[big].end - frontier[big].begin < 4096) break;
            const Task t = frontier[big];
            frontier.erase(frontier.begin() + (long)big);
            split(t, out->nodes, frontier, &out->depth, nthreads);
        }
        // phase B: every frontier task is an independent subtree
        struct Sub { std::vector<float> nodes; uint32_t depth = 0; };
        std::vector<Sub> subs(frontier.size());
        auto build_sub = [&](size_t j) {
            std::vector<Task> stack;
            Task root = frontier[j]; root.parent = kNoParent;
            stack.push_back(root);
            while (!stack.empty()) { const Task t = stack.back(); stack.pop_back(); split(t, subs[j].nodes, stack, &subs[j].depth, 1); }
        };
        if (nthreads > 1)
        {
            std::atomic<size_t> next{0};
            ptg::run_workers(nthreads, [&](unsigned) { for (size_t j; (j = next.fetch_add(1)) < frontier.size();) build_sub(j); });
        }
        else
            for (size_t j = 0; j < frontier.size(); ++j) build_sub(j);
        // splice: a subtree's inner references move by its base index; its root is hooked into the phase-A node
        for (size_t j = 0; j < frontier.size(); ++j)
        {
            const uint32_t base = (uint32_t)(out->nodes.size() / 16);
            std::vector<float>& sn = subs[j].nodes;
            for (size_t k = 0; k < sn.size(); k += 16)
                for (int c = 12; c <= 13; ++c) { const uint32_t ref = f2u(sn[k + c]); if (!(ref & kLeafFlag)) sn[k + c] = u2f(ref + base); }
            out->nodes.insert(out->nodes.end(), sn.begin(), sn.end());
            if (frontier[j].parent != kNoParent) set_child(out->nodes, frontier[j].parent, frontier[j].side, base, range_box(frontier[j].begin, frontier[j].end));
            if (subs[j].depth > out->depth) out->depth = subs[j].depth;
            std::vector<float>().swap(sn);
        }
    }
    out->n_nodes = (uint32_t)(out->nodes.size() / 16);
    // sphere payload in leaf order
    out->sph.resize(4 * (size_t)n); out->mat.resize(8 * (size_t)n);
    if (out->moving) out->vel.assign(4 * (size_t)n, 0.0f); else out->vel.clear();
    for (uint32_t sl = 0; sl < n; ++sl)
    {
        const uint32_t i = idx[sl];
        const float* s = d->sphere_pos_r + 4 * (size_t)i;
        float* o = out->sph.data() + 4 * (size_t)sl;
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3] * s[3]; // r2 as the reference forms it (geometry.cpp:47)
        if (out->moving) { float* v = out->vel.data() + 4 * (size_t)sl; v[0] = d->sphere_vel[3 * (size_t)i]; v[1] = d->sphere_vel[3 * (size_t)i + 1]; v[2] = d->sphere_vel[3 * (size_t)i + 2]; }
        float* m = out->mat.data() + 8 * (size_t)sl;
        m[0] = u2f(d->sphere_mat_type[i]); m[1] = d->sphere_mat_param[2 * (size_t)i]; m[2] = d->sphere_mat_param[2 * (size_t)i + 1]; m[3] = u2f(i);
        memcpy(m + 4, d->sphere_mat_rgba + 4 * (size_t)i, 16);
    }
}

// ---- 32-byte nodes for big trees in HBM -------------------------------------------------------------------------------
// Every child-box coordinate on a 16-bit grid over the scene's box (lo rounded down, hi rounded up: the decoded box
// encloses the padded float box), two boxes + two references = 32 bytes, same node order and references as the float
// nodes:   q0 = {L.x, L.y, L.z, R.x}  q1 = {R.y, R.z, refL, refR}   with  .x = lo.x | hi.x << 16  etc.
// The device never decodes a plane: with  plane = org + g * step  the slab value is
//     (plane - o) * r  =  g * (step * r) + (org * r - o * r),
// two per-ray constants per plane set up once per query, then one conversion and one FMA per plane (pt_device.h).
// Half the bytes per node halves the vector loads and the lines a wave step pulls through the L1.
struct NodeGrid { float org[3], step[3]; };
inline void quantise_nodes(const BvhData& in, NodeGrid* grid, std::vector<uint32_t>* out)
{
    out->clear();
    for (int k = 0; k < 3; ++k) { grid->org[k] = 0.0f; grid->step[k] = 1.0f; }
    if (in.n_nodes == 0) return;
    double lo[3] = {3e38, 3e38, 3e38}, hi[3] = {-3e38, -3e38, -3e38};
    for (uint32_t i = 0; i < in.n_nodes; ++i)
    {
        const float* nd = in.nodes.data() + 16 * (size_t)i;
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = std::min(lo[k], (double)std::min(nd[k], nd[6 + k]));
            hi[k] = std::max(hi[k], (double)std::max(nd[3 + k], nd[9 + k]));
        }
    }
    for (int k = 0; k < 3; ++k)
    {
        grid->org[k] = (float)lo[k];
        if ((double)grid->org[k] > lo[k]) grid->org[k] = nextafterf(grid->org[k], -INFINITY);
        double st = (hi[k] - (double)grid->org[k]) / 65535.0 * (1.0 + 1e-6);
        if (!(st > 1e-30)) st = 1e-30;
        grid->step[k] = (float)st;
        if ((double)grid->step[k] < st) grid->step[k] = nextafterf(grid->step[k], INFINITY);
    }
    out->resize(8 * (size_t)in.n_nodes);
    auto cell = [&](float v, int k, bool up) -> uint32_t {
        const double org = grid->org[k], st = grid->step[k];
        double g = up ? ceil(((double)v - org) / st) : floor(((double)v - org) / st);
        if (g < 0) g = 0;
        if (g > 65535) g = 65535;
        // the division rounds: make sure of the side
        if (up) { while (g < 65535 && org + g * st < (double)v) g += 1; }
        else { while (g > 0 && org + g * st > (double)v) g -= 1; }
        return (uint32_t)g;
    };
    for (uint32_t i = 0; i < in.n_nodes; ++i)
    {
        const float* nd = in.nodes.data() + 16 * (size_t)i;
        uint32_t* q = out->data() + 8 * (size_t)i;
        for (int k = 0; k < 3; ++k)
        {
            q[k] = cell(nd[k], k, false) | cell(nd[3 + k], k, true) << 16;          // left child
            q[3 + k] = cell(nd[6 + k], k, false) | cell(nd[9 + k], k, true) << 16;  // right child
        }
        q[6] = f2u(nd[12]); q[7] = f2u(nd[13]);
    }
}

// ---- 64-byte 4-wide nodes for big trees in HBM (MODE_BVH_Q4) ------------------------------------------------------------
// What bounds the traversal of a 10^6-sphere tree is the number of DEPENDENT fetches per ray (56 node visits + 16 sphere
// loads per segment with 2-wide nodes, each as slow as the slowest of a wave's 64 lanes).  A wide node holds up to FOUR
// child boxes on the same 16-bit grid (12 bytes each) + four references = 64 bytes = one fetch for two levels of the
// binary tree; and the bottom of the tree names SINGLE spheres, each with its own box in its parent, so a sphere is
// fetched only when the ray's cone touches its box (the 2-wide leaves fetch up to four spheres one after the other).
//   node = 16 dwords:  child k (0..3): dwords 3k .. 3k+2 = {lo.x | hi.x << 16, lo.y | hi.y << 16, lo.z | hi.z << 16}
//                      dwords 12..15 = references: inner -> wide node index; leaf -> kLeafFlag | (count-1) << 28 | first;
//                      an empty slot (fewer than four children) repeats child 0's reference behind the box {cell 0 .. cell 0}
//                      on every axis: a point in the grid's corner that a ray's cone all but never touches, and a
//                      reference that is harmless when it does (the same sphere or subtree once more: same t, no change)
//                      -- so the traversal tests four boxes and no "is there a child" flag
// Built by collapsing the binary SAH tree: a node's two children are replaced, largest surface first, by THEIR children
// until four are held (inner children only; leaves stay whole).  With `single` every multi-sphere leaf then becomes a
// wide node of its own whose children are its spheres (or is inlined into its parent when there is room).
struct WideData { std::vector<uint32_t> nodes; uint32_t n_nodes = 0, depth = 0, root_ref = 0; };
inline void widen_nodes(const BvhData& in, const pt_scene_desc* d, bool single, NodeGrid* grid, WideData* out)
{
    out->nodes.clear(); out->n_nodes = 0; out->depth = 0; out->root_ref = in.root_ref;
    for (int k = 0; k < 3; ++k) { grid->org[k] = 0.0f; grid->step[k] = 1.0f; }
    if (in.n_spheres == 0) return;
    struct Child { Aabb box; uint32_t ref; bool made = false; }; // made: ref already names a WIDE node (an over-full leaf's remainder)
    // the padded, swept box of one sphere slot, as build_bvh forms it (same pad: the leaf test is the exact one)
    float extent = 1.0f;
    for (int k = 0; k < 3; ++k) if (fabsf(d->camera[k]) > extent) extent = fabsf(d->camera[k]);
    for (uint32_t sl = 0; sl < in.n_spheres; ++sl)
        for (int k = 0; k < 3; ++k)
        {
            float a = fabsf(in.sph[4 * (size_t)sl + k]) + sqrtf(in.sph[4 * (size_t)sl + 3]);
            if (in.moving) a += fabsf(in.vel[4 * (size_t)sl + k]) * (fabsf(d->t0) > fabsf(d->t1) ? fabsf(d->t0) : fabsf(d->t1));
            if (a > extent) extent = a;
        }
    const float extentPad = extent * 1.0e-6f;
    auto slot_box = [&](uint32_t sl) {
        Aabb bx;
        const float* s = in.sph.data() + 4 * (size_t)sl;
        const float r = sqrtf(s[3]) * (1.0f + 1e-6f); // (r2 -> r rounds: a hair more)
        for (int k = 0; k < 3; ++k)
        {
            float a = s[k], b2 = s[k];
            if (in.moving) { const float v = in.vel[4 * (size_t)sl + k]; a = s[k] + v * d->t0; b2 = s[k] + v * d->t1; }
            const float pad = extentPad + 1e-5f * r;
            bx.lo[k] = (a < b2 ? a : b2) - r - pad; bx.hi[k] = (a > b2 ? a : b2) + r + pad;
        }
        return bx;
    };
    auto bin_child = [&](uint32_t node, int side) {
        const float* nd = in.nodes.data() + 16 * (size_t)node;
        Child c{};
        for (int k = 0; k < 3; ++k) { c.box.lo[k] = nd[6 * side + k]; c.box.hi[k] = nd[6 * side + 3 + k]; }
        c.ref = f2u(nd[12 + side]);
        return c;
    };
    auto leaf_first = [](uint32_t ref) { return ref & 0x0FFFFFFFu; };
    auto leaf_count = [](uint32_t ref) { return ((ref >> 28) & 7u) + 1u; };
    auto single_ref = [](uint32_t slot) { return kLeafFlag | slot; };
    struct Wide { Child c[4]; uint32_t n = 0; };
    std::vector<Wide> wide;
    // work list: (children to place, index of the wide node to fill, its level)
    struct Job { std::vector<Child> kids; uint32_t self, level; };
    std::vector<Job> todo;
    auto new_node = [&]() { wide.push_back(Wide{}); return (uint32_t)wide.size() - 1; };
    auto leaf_kids = [&](uint32_t ref) { std::vector<Child> k; for (uint32_t i = 0; i < leaf_count(ref); ++i) k.push_back(Child{slot_box(leaf_first(ref) + i), single_ref(leaf_first(ref) + i), false}); return k; };
    if (in.n_nodes == 0)
    {
        // the whole scene is one leaf of the binary tree
        if (!single || leaf_count(in.root_ref) == 1) return; // root_ref stays the leaf reference
        todo.push_back(Job{leaf_kids(in.root_ref), new_node(), 1});
        out->root_ref = 0;
    }
    else
    {
        todo.push_back(Job{{bin_child(0, 0), bin_child(0, 1)}, new_node(), 1});
        out->root_ref = 0;
    }
    while (!todo.empty())
    {
        Job job = std::move(todo.back()); todo.pop_back();
        if (job.level > out->depth) out->depth = job.level;
        std::vector<Child>& kids = job.kids;
        // a leaf of more than four spheres (PT_LEAF_MAX 5..8): three of them here, the rest behind a node one level down
        if (kids.size() > 4)
        {
            std::vector<Child> rest(kids.begin() + 3, kids.end());
            kids.resize(3);
            Child c{}; aabb_reset(c.box);
            for (const Child& r : rest) aabb_grow(c.box, r.box);
            c.ref = new_node(); c.made = true;
            todo.push_back(Job{std::move(rest), c.ref, job.level + 1});
            kids.push_back(c);
        }
        // pull grandchildren up, largest box first, while there is room
        for (;;)
        {
            if (kids.size() >= 4) break;
            int pick = -1; float area = -1.0f;
            for (size_t j = 0; j < kids.size(); ++j)
                if (!(kids[j].ref & kLeafFlag) && !kids[j].made && aabb_area(kids[j].box) > area) { area = aabb_area(kids[j].box); pick = (int)j; }
            if (pick < 0) break;
            const uint32_t node = kids[(size_t)pick].ref;
            kids[(size_t)pick] = bin_child(node, 0);
            kids.push_back(bin_child(node, 1));
        }
        if (single) // leaves of several spheres: inline them where there is room (smallest first), else give them a node
            for (;;)
            {
                int pick = -1; uint32_t cnt = 9;
                for (size_t j = 0; j < kids.size(); ++j)
                    if ((kids[j].ref & kLeafFlag) && leaf_count(kids[j].ref) > 1 && leaf_count(kids[j].ref) < cnt && kids.size() - 1 + leaf_count(kids[j].ref) <= 4)
                    { cnt = leaf_count(kids[j].ref); pick = (int)j; }
                if (pick < 0) break;
                const std::vector<Child> sp = leaf_kids(kids[(size_t)pick].ref);
                kids[(size_t)pick] = sp[0];
                kids.insert(kids.end(), sp.begin() + 1, sp.end());
            }
        Wide w; w.n = (uint32_t)kids.size();
        for (uint32_t j = 0; j < w.n; ++j)
        {
            w.c[j] = kids[j];
            if (kids[j].made) continue;                           // (its node and job exist already)
            if (!(kids[j].ref & kLeafFlag))                       // an inner node of the binary tree: a wide node of its own
            {
                const uint32_t self = new_node();
                w.c[j].ref = self;
                todo.push_back(Job{{bin_child(kids[j].ref, 0), bin_child(kids[j].ref, 1)}, self, job.level + 1});
            }
            else if (single && leaf_count(kids[j].ref) > 1)       // a leaf that did not fit: its spheres one level down
            {
                const uint32_t self = new_node();
                w.c[j].ref = self;
                todo.push_back(Job{leaf_kids(kids[j].ref), self, job.level + 1});
            }
        }
        wide[job.self] = w;
    }
    out->n_nodes = (uint32_t)wide.size();
    // one grid over everything
    double lo[3] = {3e38, 3e38, 3e38}, hi[3] = {-3e38, -3e38, -3e38};
    for (const Wide& w : wide)
        for (uint32_t j = 0; j < w.n; ++j)
            for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], (double)w.c[j].box.lo[k]); hi[k] = std::max(hi[k], (double)w.c[j].box.hi[k]); }
    for (int k = 0; k < 3; ++k)
    {
        grid->org[k] = (float)lo[k];
        if ((double)grid->org[k] > lo[k]) grid->org[k] = nextafterf(grid->org[k], -INFINITY);
        double st = (hi[k] - (double)grid->org[k]) / 65535.0 * (1.0 + 1e-6);
        if (!(st > 1e-30)) st = 1e-30;
        grid->step[k] = (float)st;
        if ((double)grid->step[k] < st) grid->step[k] = nextafterf(grid->step[k], INFINITY);
    }
    auto cell = [&](float v, int k, bool up) -> uint32_t {
        const double org = grid->org[k], st = grid->step[k];
        double g = up ? ceil(((double)v - org) / st) : floor(((double)v - org) / st);
        if (g < 0) g = 0;
        if (g > 65535) g = 65535;
        if (up) { while (g < 65535 && org + g * st < (double)v) g += 1; }
        else { while (g > 0 && org + g * st > (double)v) g -= 1; }
        return (uint32_t)g;
    };
    out->nodes.assign(16 * (size_t)wide.size(), 0u);
    auto quantise = [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i)
        {
            uint32_t* q = out->nodes.data() + 16 * i;
            for (uint32_t j = 0; j < 4; ++j)
            {
                if (j < wide[i].n)
                {
                    for (int k = 0; k < 3; ++k) q[3 * j + k] = cell(wide[i].c[j].box.lo[k], k, false) | cell(wide[i].c[j].box.hi[k], k, true) << 16;
                    q[12 + j] = wide[i].c[j].ref;
                }
                else
                    q[12 + j] = wide[i].c[0].ref;   // (cells stay 0: see the layout comment)
            }
        }
    };
    // (ten million cell computations at 10^6 spheres: sliced over the builder's worker threads)
    const unsigned workers = ptg::host_threads(16);
    if (wide.size() < 65536 || workers == 1) quantise(0, wide.size());
    else ptg::run_workers(workers, [&](unsigned w) { quantise(wide.size() * w / workers, wide.size() * (w + 1) / workers); });
}

} // namespace pth
