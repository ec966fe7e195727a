// oracle/pt_oracle.cpp -- TEST INFRASTRUCTURE (the checker).  CPU restatement of the reference's hot path.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
// (pathtracer_amd/, include/) never links, imports or calls it.
//
// PARITY STATUS: PINNED.  The reference holds no tests or golden vectors, so this
// restatement is pinned against the reference ITSELF, compiled here from its own sources into
// oracle/_ref/libptref.so (oracle/ref_harness.cpp, oracle/Makefile): tests/test_oracle_vs_reference.py checks
// every function below bit-for-bit against it on identical inputs and identical MT19937 draws, and
// tests/golden/ holds vectors generated from the reference by tests/golden/make_golden.py for boxes where
// /root/reference is absent.  Not compiled from the reference (needs Windows.h): build_bvh_tree's allocator,
// run_thread_batch's signature and the BMP writer -- see DESIGN.md "Oracle".
//
// All citations are file:line under /root/reference/src.  Arithmetic is IEEE binary32 with the reference's
// exact operation order (no contraction: built with -ffp-contract=off on FMA-less baseline x86-64), except
// the f64 island in reflectance().
//
// Two knobs separate "what the reference does" from "what the MI355X build defines" (decisions):
//   query   = QUERY_BVH_QUIRK : the reference's recursive BVH query incl. its pairwise combine (main.cpp:53-91)
//             QUERY_CLOSEST   : true closest hit over all spheres (t > 0.001, smallest t, lowest index on ties)
//   product = PRODUCT_RECURSIVE : colour = c1*(c2*(...*terminal)), the reference's recursion order (main.cpp:233)
//             PRODUCT_ITERATIVE : colour = ((c1*c2)*...)*terminal, the order a loop with a throughput register gives
// The reference-pinning tests use (BVH_QUIRK, RECURSIVE); the GPU parity tests use (CLOSEST, ITERATIVE).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <cmath>
#include <random>
#include <thread>
#include <vector>
#include <atomic>
#include <chrono>

namespace {

const float F32_MAX_ = 3.402823466e+38F; // types.h:9
const float MIN_T = 0.001f;              // main.cpp:56,178

enum { MAT_NONE = 0, MAT_DIFFUSE = 1, MAT_METAL = 2, MAT_DIALECTRIC = 3 }; // render_world.h:32-38
enum { QUERY_CLOSEST = 0, QUERY_BVH_QUIRK = 1 };
enum { PRODUCT_ITERATIVE = 0, PRODUCT_RECURSIVE = 1 };

// ---------------------------------------------------------------- vectors.h / vectors.cpp
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }          // vectors.cpp:13-20
inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }          // vectors.cpp:22-29
inline V3 mul(V3 v, float s) { return v3(v.x * s, v.y * s, v.z * s); }             // vectors.cpp:31-42
inline V3 neg(V3 v) { return v3(-v.x, -v.y, -v.z); }                               // vectors.cpp:49-56
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // vectors.cpp:132-135
inline float norm_squared(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }       // vectors.cpp:112-115
inline float norm(V3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }        // vectors.cpp:107-110 (float sqrt overload)
inline V3 divs(V3 v, float s) { return mul(v, 1.0f / s); }                         // vectors.cpp:44-47
inline V3 normalize(V3 v) { return divs(v, norm(v)); }                             // vectors.cpp:127-130
inline V3 cross(V3 a, V3 b)                                                        // vectors.cpp:137-142
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline bool near_zero(V3 v)                                                        // vectors.cpp:144-148
{
    const float e = 1e-8f;
    return fabsf(v.x) < e && fabsf(v.y) < e && fabsf(v.z) < e;
}
inline V4 v4(float x, float y, float z, float w) { V4 r = {x, y, z, w}; return r; }
inline V4 hadamard(V4 a, V4 b) { return v4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); } // vectors.cpp:150-158
inline V4 add4(V4 a, V4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }     // vectors.cpp:66-74
inline V4 mul4(V4 v, float s) { return v4(v.x * s, v.y * s, v.z * s, v.w * s); }          // vectors.cpp:86-94

// ---------------------------------------------------------------- RNG (utils.h:27-51)
struct Rng
{
    virtual float f32() = 0;  // random_f32()  utils.h:38-45
    virtual double f64() = 0; // random_f64()  utils.h:28-35
    virtual ~Rng() {}
    virtual void align() {}    // Philox only: skip to the next 4-word block (see PhiloxRng)
    virtual bool pinhole_skips_lens() const { return false; } // Philox only: get_ray draws no lens point when lensRadius == 0
    virtual bool still_skips_time() const { return false; }   // Philox only: no time draw when t0 == t1
    virtual bool unit_sphere_point(V3*) { return false; }     // Philox only: its own form of utils.h:80-94 (see PhiloxRng)
    uint64_t n32 = 0, n64 = 0; // draws taken, for tests that build tapes
    float f32(float lo, float hi) { return f32() * (hi - lo) + lo; } // utils.h:47-51
};

// What the reference gets on a fresh thread: default-seeded engines, a fresh distribution object per call.
struct MtRng : Rng
{
    std::mt19937 g32;
    std::mt19937_64 g64;
    MtRng(uint64_t skip32, uint64_t skip64)
    {
        for (uint64_t i = 0; i < skip32; ++i) { std::uniform_real_distribution<float> d(0.0f, 1.0f); (void)d(g32); }
        for (uint64_t i = 0; i < skip64; ++i) { std::uniform_real_distribution<double> d(0.0, 1.0); (void)d(g64); }
    }
    float f32() override { ++n32; std::uniform_real_distribution<float> d(0.0f, 1.0f); return d(g32); }
    double f64() override { ++n64; std::uniform_real_distribution<double> d(0.0, 1.0); return d(g64); }
};

// Explicit uniforms (what the GPU probe kernels are fed).
struct TapeRng : Rng
{
    const float* t32; uint64_t c32; const double* t64; uint64_t c64; bool overrun = false;
    TapeRng(const float* a, uint64_t na, const double* b, uint64_t nb) : t32(a), c32(na), t64(b), c64(nb) {}
    float f32() override { if (n32 >= c32) { overrun = true; ++n32; return 0.5f; } return t32[n32++]; }
    double f64() override { if (n64 >= c64) { overrun = true; ++n64; return 0.5; } return t64[n64++]; }
};

// Philox4x32-10 (Salmon et al., SC'11; the Random123 constants).  The MI355X build's RNG; the reference has no
// counterpart (it uses MT19937), so this is DEFINED here and in DESIGN.md, pinned by the Random123 known-answer
// vectors (tests/test_oracle_golden.py):
//   word k of the stream of (seed, pixel, sample) = philox4x32_10(ctr = {k>>2, pixel, sample, 0}, key = {seed_lo, seed_hi})[k&3]
//   f32 uniform  = (word >> 8) * 2^-24                         in [0,1), one word
//   f64 uniform  = (((u64)w0 << 32 | w1) >> 11) * 2^-53        in [0,1), two consecutive words
//   align()      = k <- (k+3) & ~3, at the start of every scatter event
inline void philox4x32_10(const uint32_t c[4], const uint32_t k[2], uint32_t out[4])
{
    uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], k0 = k[0], k1 = k[1];
    for (int r = 0; r < 10; ++r)
    {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
struct PhiloxRng : Rng
{
    uint32_t key[2], pixel, sample, k = 0, buf[4];
    PhiloxRng(uint64_t seed, uint32_t px, uint32_t smp) : pixel(px), sample(smp) { key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32); }
    uint32_t word()
    {
        if ((k & 3) == 0) { uint32_t c[4] = {k >> 2, pixel, sample, 0}; philox4x32_10(c, key, buf); }
        return buf[(k++) & 3];
    }
    // Every scatter starts on a block boundary (a SIMT efficiency device of the MI355X build; part of the stream's
    // definition).  The reference's MT19937 stream has no such notion: MtRng / TapeRng ignore it.
    void align() override { k = (k + 3u) & ~3u; }
    // A pinhole camera (lensRadius == 0) draws no lens point on this stream: camera.cpp:74 multiplies the point by zero,
    // so the ray is the same and the sample starts with one Philox block instead of two (definition of the MI355X build's
    // stream; the reference's own streams -- MtRng, TapeRng -- keep the loop).
    bool pinhole_skips_lens() const override { return true; }
    bool still_skips_time() const override { return true; }
    float f32() override { ++n32; return (float)(word() >> 8) * 5.9604644775390625e-8f; }
    double f64() override
    {
        ++n64;
        uint64_t hi = word(), lo = word();
        return (double)(((hi << 32) | lo) >> 11) * 1.1102230246251565404e-16;
    }
    // random_point_in_unit_sphere (utils.h:80-94) on this stream, one Philox block per trip of the loop (definition of
    // the MI355X build's stream).  The reference draws z, y, x uniform on the 24-bit grid
    // k * 2^-23 - 1 and keeps the point if norm(p) <= 1.  Here the top 9 bits of each coordinate (a cell of side 2^-8)
    // and its low 15 bits are drawn separately: words 0..2 of a block are three attempts at a cell (z = bits 31..23,
    // y = 22..14, x = 13..5), rejected outright when the cell's nearest corner is outside the sphere (no point of it can
    // pass); the first surviving cell is completed with the block's one set of low bits (z: word 3 bits 31..17,
    // y: 16..2, x: word 3 bits 1..0 above 13 of the 15 spare low bits of words 0..2) and takes the reference's test.
    // A block without a surviving cell, or whose point fails, is spent; the next block starts afresh.  The accepted
    // point is uniform over the grid points that pass the reference's test, as in the reference.
    static bool cell_outside(uint32_t w)
    {
        auto a = [](uint32_t q) { return (q & 256u) ? (q & 255u) : (255u - q); }; // the cell's distance from the coordinate plane, in cells
        uint32_t az = a(w >> 23), ay = a((w >> 14) & 511u), ax = a((w >> 5) & 511u);
        return ax * ax + ay * ay + az * az > 65536u;                              // nearest corner: |p|^2 = sum / 2^16 > 1
    }
    bool unit_sphere_point(V3* out) override
    {
        for (;;)
        {
            align();
            uint32_t w[4];
            for (int i = 0; i < 4; ++i) w[i] = word();
            int j = 0;
            while (j < 3 && cell_outside(w[j])) ++j;
            if (j == 3) continue;
            const uint32_t spare = (w[0] & 31u) | ((w[1] & 31u) << 5) | ((w[2] & 31u) << 10);
            const uint32_t kz = ((w[j] >> 23) << 15) | (w[3] >> 17);
            const uint32_t ky = (((w[j] >> 14) & 511u) << 15) | ((w[3] >> 2) & 0x7FFFu);
            const uint32_t kx = (((w[j] >> 5) & 511u) << 15) | ((w[3] & 3u) << 13) | (spare & 0x1FFFu);
            // random_f32(-1, 1) = r * 2 - 1 with r = k * 2^-24 (utils.h:47-51): both scalings exact
            V3 p = v3((float)kx * 5.9604644775390625e-8f * 2.0f + -1.0f, (float)ky * 5.9604644775390625e-8f * 2.0f + -1.0f,
                      (float)kz * 5.9604644775390625e-8f * 2.0f + -1.0f);
            if (norm(p) <= 1.0f) { *out = p; return true; }                        // utils.h:88
        }
    }
};

// ---------------------------------------------------------------- samplers (utils.h:71-138)
// v3f(random_f32(a,b), random_f32(a,b), random_f32(a,b)) has unspecified argument order; the reference as
// compiled by g++ evaluates right-to-left: first draw -> z, second -> y, third -> x (verified
// against libptref.so by tests/test_oracle_vs_reference.py).
V3 random_point_in_unit_sphere(Rng& r)                                             // utils.h:80-94
{
    { V3 p; if (r.unit_sphere_point(&p)) return p; }                              // the Philox stream's own form (PhiloxRng)
    for (;;)
    {
        float z = r.f32(-1.0f, 1.0f), y = r.f32(-1.0f, 1.0f), x = r.f32(-1.0f, 1.0f); // utils.h:75-78,87
        V3 p = v3(x, y, z);
        if (norm(p) <= 1.0f) return p;                                             // utils.h:89
    }
}
V3 random_unit_vector(Rng& r) { return normalize(random_point_in_unit_sphere(r)); } // utils.h:102-108
V3 random_point_in_unit_circle(Rng& r)                                             // utils.h:124-138
{
    for (;;)
    {
        float y = r.f32(-1.0f, 1.0f), x = r.f32(-1.0f, 1.0f);                      // utils.h:131, right-to-left
        V3 p = v3(x, y, 0.0f);
        if (norm(p) <= 1.0f) return p;                                             // utils.h:133
    }
}

// ---------------------------------------------------------------- geometry (geometry.h, geometry.cpp)
struct Ray { V3 origin, dir; };
inline V3 ray_at(const Ray& r, float t) { return add(r.origin, mul(r.dir, t)); }   // geometry.h:10
inline Ray make_ray(V3 o, V3 d, bool normalized)                                   // geometry.cpp:3-7
{
    Ray r; r.origin = o; r.dir = normalized ? d : normalize(d); return r;
}
inline V3 reflect_direction(V3 dir, V3 n)                                          // geometry.h:62-67
{
    float projectedDistance = -dot(dir, n);
    return add(dir, mul(mul(n, 2.0f), projectedDistance));
}
inline float intersect_sphere(const Ray& ray, V3 centre, float radius)             // geometry.cpp:39-68
{
    float tResult = F32_MAX_;
    float a = 1;
    V3 oc = sub(ray.origin, centre);
    float b = 2 * dot(ray.dir, oc);
    float c = dot(oc, oc) - radius * radius;
    float discriminant = b * b - 4 * a * c;
    if (discriminant > 0)
    {
        float rootValue = sqrtf(discriminant);
        tResult = (-b - rootValue) / (2 * a); // near root only, may be negative (geometry.cpp:56-59)
    }
    else if (discriminant == 0)
        tResult = -b / (2 * a);
    return tResult;
}
inline float intersect_plane(const Ray& ray, V3 n, float offset)                   // geometry.cpp:70-81
{
    float tResult = F32_MAX_;
    float denominator = dot(ray.dir, n);
    if (denominator != 0.0f) tResult = (offset - dot(n, ray.origin)) / denominator;
    return tResult;
}
struct Box { V3 pos; float hw, hh, hl; };                                          // geometry.h:31-52
#define MAXV(a, b) ((a) < (b) ? (b) : (a)) /* types.h:12 */
#define MINV(a, b) ((a) < (b) ? (a) : (b)) /* types.h:13 */
inline bool hit_box(const Ray& ray, const Box& rect)                               // geometry.cpp:83-115
{
    float inverseDirX = 1.0f / ray.dir.x;
    float tx0 = ((rect.pos.x - rect.hw) - ray.origin.x) * inverseDirX;
    float tx1 = ((rect.pos.x + rect.hw) - ray.origin.x) * inverseDirX;
    if (tx0 > tx1) { float c = tx0; tx0 = tx1; tx1 = c; }
    float inverseDirY = 1.0f / ray.dir.y;
    float ty0 = ((rect.pos.y - rect.hh) - ray.origin.y) * inverseDirY;
    float ty1 = ((rect.pos.y + rect.hh) - ray.origin.y) * inverseDirY;
    if (ty0 > ty1) { float c = ty0; ty0 = ty1; ty1 = c; }
    float inverseDirZ = 1.0f / ray.dir.z;
    float tz0 = ((rect.pos.z - rect.hl) - ray.origin.z) * inverseDirZ;
    float tz1 = ((rect.pos.z + rect.hl) - ray.origin.z) * inverseDirZ;
    if (tz0 > tz1) { float c = tz0; tz0 = tz1; tz1 = c; }
    float tMin = MAXV(MAXV(tx0, ty0), tz0);
    float tMax = MINV(MINV(tx1, ty1), tz1);
    if (tMin >= tMax) return false;
    return true;
}

// ---------------------------------------------------------------- scene (render_world.h) and camera (camera.h)
struct Material { uint32_t type; V4 colour; float roughness, n; };                 // render_world.h:28-53
struct SphereObj { Material m; V3 pos; float radius; V3 vel; };                     // render_world.h:57-71
struct PlaneObj { Material m; V3 normal; float offset; };                          // render_world.h:73-79
struct BvhNode { Box box; int32_t left, right, object; };                          // main.cpp:44-51 (pointers -> indices)

struct Camera { V3 pos, dir, up; float fov, aspect, focus, lensRadius; };          // camera.h:9-38

struct Scene
{
    std::vector<SphereObj> spheres;
    std::vector<PlaneObj> planes;
    float t0 = 0, t1 = 0;
    Camera cam;
    std::vector<BvhNode> nodes; // optional: reference-topology tree for QUERY_BVH_QUIRK
    int32_t root = -1;
    bool moving = false;
};

Ray get_ray(const Camera& c, float u, float v, Rng& rng)                           // camera.cpp:57-79
{
    V3 planePos = add(c.pos, mul(c.dir, c.focus));                                 // camera.cpp:40-44
    float height = 2.0f * std::tan(c.fov / 2.0f);                                  // camera.cpp:34 (float overload of tan)
    float width = height * c.aspect;                                               // camera.cpp:35
    V3 horizontal = normalize(cross(c.dir, c.up));                                 // camera.cpp:64
    V3 vertical = c.up;                                                            // camera.cpp:65 (NOT re-orthogonalised)
    V3 topLeft = add(sub(planePos, mul(mul(horizontal, width / 2.0f), c.focus)), mul(mul(vertical, height / 2.0f), c.focus)); // :69
    V3 rayTarget = sub(add(topLeft, mul(mul(mul(horizontal, u), width), c.focus)), mul(mul(mul(vertical, v), height), c.focus)); // :70
    V3 disc = v3(0, 0, 0);
    if (!(rng.pinhole_skips_lens() && c.lensRadius == 0.0f)) disc = random_point_in_unit_circle(rng); // camera.cpp:74 (runs even if lensRadius == 0)
    V3 pointOnLens = mul(disc, c.lensRadius);
    V3 lensOffset = add(mul(horizontal, pointOnLens.x), mul(vertical, pointOnLens.y)); // camera.cpp:75
    V3 origin = add(c.pos, lensOffset);
    return make_ray(origin, normalize(sub(rayTarget, add(c.pos, lensOffset))), true); // camera.cpp:77
}

inline V3 sphere_centre(const SphereObj& s, float time) { return add(s.pos, mul(s.vel, time)); } // main.cpp:66,210

// main.cpp:53-91 -- the reference's recursive query, pairwise combine and all.
float bvh_query_quirk(const Scene& sc, int32_t node, const Ray& ray, float time, int32_t* outObject)
{
    float tResult = F32_MAX_;
    if (node < 0) return tResult;
    const BvhNode& n = sc.nodes[node];
    if (n.object >= 0) // leaf: raw near-root t, NO box test, may be negative (main.cpp:61-69)
    {
        *outObject = n.object;
        const SphereObj& s = sc.spheres[n.object];
        tResult = intersect_sphere(ray, sphere_centre(s, time), s.radius);
    }
    else if (hit_box(ray, n.box))
    {
        int32_t lo = -1, ro = -1;
        float tLeft = bvh_query_quirk(sc, n.left, ray, time, &lo);
        float tRight = bvh_query_quirk(sc, n.right, ray, time, &ro);
        if (tLeft < tRight && tLeft > MIN_T) { tResult = tLeft; *outObject = lo; }
        else if (tRight < tLeft && tRight > MIN_T) { tResult = tRight; *outObject = ro; }
    }
    return tResult;
}

// The MI355X build's definition (DESIGN.md decision D1): smallest t with t > MIN_T over ALL spheres.
float query_closest(const Scene& sc, const Ray& ray, float time, int32_t* outObject)
{
    float best = F32_MAX_;
    for (size_t i = 0; i < sc.spheres.size(); ++i)
    {
        const SphereObj& s = sc.spheres[i];
        float t = intersect_sphere(ray, sphere_centre(s, time), s.radius);
        if (t > MIN_T && t < best) { best = t; *outObject = (int32_t)i; }
    }
    return best;
}

double reflectance(double cosine, double refractRatio)                             // main.cpp:161-168
{
    double r0 = (1.0 - refractRatio) / (1.0 + refractRatio);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * pow((1.0 - cosine), 5);
}

// One bounce record, for per-ray parity fixtures.
struct Segment
{
    float t; int32_t prim; // prim: sphere index >= 0, plane index = -1 - i, miss = INT32_MIN
    V3 point, normal, newDir; uint32_t event;
};
enum { EV_SCATTER = 0, EV_SKY = 1, EV_DEPTH = 2, EV_ABSORB = 3, EV_NONE_MAT = 4 };
const int32_t PRIM_MISS = INT32_MIN;

struct TraceOpts { int query; int product; };

// Closest hit of planes then spheres, as cast_ray orders it (main.cpp:180-216).
struct Hit { float t; int32_t prim; V3 point, normal; const Material* m; };
Hit closest_hit(const Scene& sc, const Ray& ray, float time, int query)
{
    Hit h; h.t = F32_MAX_; h.prim = PRIM_MISS; h.point = v3(0, 0, 0); h.normal = v3(0, 0, 0); h.m = 0;
    for (size_t i = 0; i < sc.planes.size(); ++i)                                  // main.cpp:186-200
    {
        const PlaneObj& p = sc.planes[i];
        float t = intersect_plane(ray, p.normal, p.offset);
        if (t > MIN_T && t < h.t) { h.t = t; h.point = ray_at(ray, t); h.normal = p.normal; h.m = &p.m; h.prim = -1 - (int32_t)i; }
    }
    int32_t obj = -1;
    float t = (query == QUERY_BVH_QUIRK) ? bvh_query_quirk(sc, sc.root, ray, time, &obj) : query_closest(sc, ray, time, &obj);
    if (t > MIN_T && t < h.t)                                                      // main.cpp:204-216
    {
        const SphereObj& s = sc.spheres[obj];
        h.t = t;
        V3 centre = sphere_centre(s, time);
        h.point = ray_at(ray, t);
        h.normal = normalize(sub(h.point, centre));
        h.m = &s.m; h.prim = obj;
    }
    return h;
}

// Scatter at a hit (main.cpp:219-298).  Returns false when the path ends here (metal absorbed / NONE material).
bool scatter(const Ray& ray, const Hit& h, Rng& rng, Ray* out, V4* attenuation, uint32_t* event)
{
    const Material& m = *h.m;
    *event = EV_SCATTER;
    rng.align();
    if (m.type == MAT_DIFFUSE)                                                     // main.cpp:223-234
    {
        V3 scatterDirection = add(random_unit_vector(rng), h.normal);
        if (near_zero(add(scatterDirection, h.normal))) scatterDirection = h.normal; // sic: tests scatter+normal (main.cpp:226)
        *out = make_ray(h.point, scatterDirection, false);
        *attenuation = m.colour;
        return true;
    }
    if (m.type == MAT_METAL)                                                       // main.cpp:235-259
    {
        V3 reflectedDir = reflect_direction(ray.dir, h.normal);
        if (m.roughness > 0.0f)
        {
            V3 centre = add(h.point, reflectedDir);                                // main.cpp:244
            V3 randomPoint = add(mul(random_point_in_unit_sphere(rng), m.roughness), centre); // utils.h:96-100
            reflectedDir = sub(randomPoint, h.point);                              // main.cpp:247
        }
        *out = make_ray(h.point, reflectedDir, false);
        *attenuation = m.colour;
        if (dot(reflectedDir, h.normal) > 0) return true;                          // main.cpp:252
        *event = EV_ABSORB;                                                        // -> Colour::BLACK (main.cpp:258)
        return false;
    }
    if (m.type == MAT_DIALECTRIC)                                                  // main.cpp:260-297
    {
        float worldIndex = 1.0f;
        float refractRatio = worldIndex / m.n;
        if (dot(ray.dir, h.normal) > 0.0f) refractRatio = 1.0f / refractRatio;
        float cosTheta = dot(neg(ray.dir), h.normal);                              // not clamped, not flipped
        float sinTheta = sqrtf(1 - cosTheta * cosTheta);
        bool internalReflection = refractRatio * sinTheta > 1.0f;
        bool shouldReflect = reflectance(cosTheta, refractRatio) > rng.f64();      // one f64 draw ALWAYS (main.cpp:276)
        V3 newRayDir;
        if (internalReflection || shouldReflect)
            newRayDir = reflect_direction(ray.dir, h.normal);
        else
        {
            V3 rayPerpendicular = mul(add(ray.dir, mul(h.normal, cosTheta)), refractRatio); // main.cpp:287
            float d = 1 - norm_squared(rayPerpendicular);
            float ad = d < 0 ? -d : d;                                             // ABS_VALUE types.h:15
            V3 rayParallel = mul(h.normal, -sqrtf(ad));                            // main.cpp:288
            newRayDir = add(rayPerpendicular, rayParallel);
        }
        *out = make_ray(h.point, newRayDir, false);
        *attenuation = m.colour;
        return true;
    }
    *event = EV_NONE_MAT; // Material::NONE: resultColour stays v4f() = (0,0,0,0) (main.cpp:173)
    return false;
}

inline V4 sky(const Ray& ray)                                                      // main.cpp:299-304
{
    float ratio = 0.5f * (ray.dir.y + 1.0f);
    V4 white = v4(1, 1, 1, 1), blue = v4(0.7f, 0.8f, 0.9f, 1.0f);
    return add4(mul4(white, 1.0f - ratio), mul4(blue, ratio));
}

// cast_ray (main.cpp:171-307) as a loop.  `log`, if given, receives one Segment per loop iteration.
V4 cast_ray(const Scene& sc, Ray ray, uint32_t maxDepth, float time, Rng& rng, TraceOpts opt,
            std::vector<Segment>* log, uint32_t* segments)
{
    std::vector<V4> stack; // attenuations, outermost first (PRODUCT_RECURSIVE only)
    V4 throughput = v4(1, 1, 1, 1);
    bool first = true;
    V4 terminal;
    uint32_t nseg = 0;
    for (uint32_t depth = maxDepth;; --depth)
    {
        if (depth == 0) // maxDepth <= 0 on a u32 (main.cpp:175-176)
        {
            terminal = v4(0, 0, 0, 1);
            if (log) { Segment s; memset(&s, 0, sizeof(s)); s.prim = PRIM_MISS; s.event = EV_DEPTH; log->push_back(s); }
            break;
        }
        ++nseg;
        Hit h = closest_hit(sc, ray, time, opt.query);
        Segment s; memset(&s, 0, sizeof(s));
        s.t = h.t; s.prim = h.prim; s.point = h.point; s.normal = h.normal;
        if (h.t != F32_MAX_ && h.t > 0)                                            // main.cpp:219
        {
            Ray next; V4 att; uint32_t ev;
            bool alive = scatter(ray, h, rng, &next, &att, &ev);
            s.event = ev;
            if (ev != EV_NONE_MAT) s.newDir = next.dir;
            if (log) log->push_back(s);
            if (!alive)
            {
                terminal = (ev == EV_ABSORB) ? v4(0, 0, 0, 1) : v4(0, 0, 0, 0);
                break;
            }
            if (opt.product == PRODUCT_RECURSIVE) stack.push_back(att);
            else { throughput = first ? att : hadamard(throughput, att); first = false; }
            ray = next;
        }
        else
        {
            terminal = sky(ray);
            s.event = EV_SKY;
            if (log) log->push_back(s);
            break;
        }
    }
    if (segments) *segments = nseg;
    if (opt.product == PRODUCT_RECURSIVE)
    {
        V4 c = terminal;
        for (size_t i = stack.size(); i-- > 0;) c = hadamard(stack[i], c);         // main.cpp:233,255,296
        return c;
    }
    return first ? terminal : hadamard(throughput, terminal);
}

// One sample of run_thread_batch's inner loop (main.cpp:349-355).
V4 pixel_sample(const Scene& sc, uint32_t px, uint32_t py, uint32_t w, uint32_t h, uint32_t depth, Rng& rng,
                TraceOpts opt, uint32_t* segments)
{
    // main.cpp:349 (always one draw; the Philox stream of the build skips it for a still scene: r * 0 + t0 = t0)
    float rayTime = (rng.still_skips_time() && sc.t0 == sc.t1) ? sc.t0 : rng.f32(sc.t0, sc.t1);
    float u = ((float)px + rng.f32()) / (float)w;                                  // main.cpp:351
    float v = ((float)py - rng.f32()) / (float)h;                                  // main.cpp:352 (MINUS)
    Ray ray = get_ray(sc.cam, u, v, rng);                                          // main.cpp:354
    return cast_ray(sc, ray, depth, rayTime, rng, opt, 0, segments);               // main.cpp:355
}

inline float clampf(float v, float lo, float hi) { if (v < lo) return lo; else if (v > hi) return hi; return v; } // utils.h:8-15

Scene* S(void* h) { return (Scene*)h; }

} // namespace

extern "C" {

// ================================================================ scene handles
void* oracle_scene_new(uint32_t ns, const float* pos_r, const float* vel3 /*or NULL*/, const uint32_t* type,
                       const float* rgba, const float* param, uint32_t np, const float* pl_n_d,
                       const uint32_t* pl_type, const float* pl_rgba, const float* pl_param, float t0, float t1,
                       const float* camera13)
{
    Scene* s = new Scene();
    s->spheres.resize(ns);
    for (uint32_t i = 0; i < ns; ++i)
    {
        SphereObj& o = s->spheres[i];
        o.pos = v3(pos_r[4*i], pos_r[4*i+1], pos_r[4*i+2]); o.radius = pos_r[4*i+3];
        o.vel = vel3 ? v3(vel3[3*i], vel3[3*i+1], vel3[3*i+2]) : v3(0, 0, 0);
        if (o.vel.x != 0 || o.vel.y != 0 || o.vel.z != 0) s->moving = true;
        o.m.type = type[i]; o.m.colour = v4(rgba[4*i], rgba[4*i+1], rgba[4*i+2], rgba[4*i+3]);
        o.m.roughness = param[2*i]; o.m.n = param[2*i+1];
    }
    s->planes.resize(np);
    for (uint32_t i = 0; i < np; ++i)
    {
        PlaneObj& o = s->planes[i];
        o.normal = v3(pl_n_d[4*i], pl_n_d[4*i+1], pl_n_d[4*i+2]); o.offset = pl_n_d[4*i+3];
        o.m.type = pl_type[i]; o.m.colour = v4(pl_rgba[4*i], pl_rgba[4*i+1], pl_rgba[4*i+2], pl_rgba[4*i+3]);
        o.m.roughness = pl_param[2*i]; o.m.n = pl_param[2*i+1];
    }
    s->t0 = t0; s->t1 = t1;
    const float* c = camera13;
    s->cam.pos = v3(c[0], c[1], c[2]); s->cam.dir = v3(c[3], c[4], c[5]); s->cam.up = v3(c[6], c[7], c[8]);
    s->cam.fov = c[9]; s->cam.aspect = c[10]; s->cam.focus = c[11]; s->cam.lensRadius = c[12];
    return s;
}
void oracle_scene_free(void* h) { delete S(h); }
// Reference-topology tree (from ref_scene_dump_bvh) for QUERY_BVH_QUIRK; root = last node (post-order dump).
void oracle_scene_set_bvh(void* h, uint32_t nn, const float* boxes, const int32_t* links)
{
    Scene* s = S(h);
    s->nodes.resize(nn);
    for (uint32_t i = 0; i < nn; ++i)
    {
        BvhNode& n = s->nodes[i];
        n.box.pos = v3(boxes[6*i], boxes[6*i+1], boxes[6*i+2]); n.box.hw = boxes[6*i+3]; n.box.hh = boxes[6*i+4]; n.box.hl = boxes[6*i+5];
        n.left = links[3*i]; n.right = links[3*i+1]; n.object = links[3*i+2];
    }
    s->root = nn ? (int32_t)nn - 1 : -1;
}

// ================================================================ RNG
void oracle_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }
void oracle_philox_words(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t* out)
{
    PhiloxRng r(seed, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.word();
}
// n points of PhiloxRng::unit_sphere_point from the start of one stream; *words_used = stream position afterwards
void oracle_philox_unit_sphere_points(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out, uint32_t* words_used)
{
    PhiloxRng r(seed, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) { V3 p = random_point_in_unit_sphere(r); out[3*i] = p.x; out[3*i+1] = p.y; out[3*i+2] = p.z; }
    *words_used = r.k;
}
void oracle_philox_f32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out)
{
    PhiloxRng r(seed, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.f32();
}
void oracle_philox_f64(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double* out)
{
    PhiloxRng r(seed, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.f64();
}
void oracle_mt_f32(uint32_t skip, uint32_t n, float* out) { MtRng r(skip, 0); for (uint32_t i = 0; i < n; ++i) out[i] = r.f32(); }
void oracle_mt_f64(uint32_t skip, uint32_t n, double* out) { MtRng r(0, skip); for (uint32_t i = 0; i < n; ++i) out[i] = r.f64(); }

// ================================================================ primitives (same signatures as ref_*)
void oracle_intersect_spheres(uint32_t n, const float* rays, const float* spheres, float* out_t)
{
    for (uint32_t i = 0; i < n; ++i)
    {
        const float* r = rays + 6*i; const float* s = spheres + 4*i;
        Ray ray = make_ray(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), true);
        out_t[i] = intersect_sphere(ray, v3(s[0], s[1], s[2]), s[3]);
    }
}
void oracle_intersect_planes(uint32_t n, const float* rays, const float* planes, float* out_t)
{
    for (uint32_t i = 0; i < n; ++i)
    {
        const float* r = rays + 6*i; const float* p = planes + 4*i;
        Ray ray = make_ray(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), true);
        out_t[i] = intersect_plane(ray, v3(p[0], p[1], p[2]), p[3]);
    }
}
void oracle_hit_boxes(uint32_t n, const float* rays, const float* boxes, uint8_t* out_hit)
{
    for (uint32_t i = 0; i < n; ++i)
    {
        const float* r = rays + 6*i; const float* b = boxes + 6*i;
        Ray ray = make_ray(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), true);
        Box box; box.pos = v3(b[0], b[1], b[2]); box.hw = b[3]; box.hh = b[4]; box.hl = b[5];
        out_hit[i] = hit_box(ray, box) ? 1 : 0;
    }
}
void oracle_reflect(uint32_t n, const float* dirs, const float* normals, float* out)
{
    for (uint32_t i = 0; i < n; ++i)
    {
        V3 v = reflect_direction(v3(dirs[3*i], dirs[3*i+1], dirs[3*i+2]), v3(normals[3*i], normals[3*i+1], normals[3*i+2]));
        out[3*i] = v.x; out[3*i+1] = v.y; out[3*i+2] = v.z;
    }
}
void oracle_normalize(uint32_t n, const float* v, float* out)
{
    for (uint32_t i = 0; i < n; ++i)
    {
        V3 r = normalize(v3(v[3*i], v[3*i+1], v[3*i+2]));
        out[3*i] = r.x; out[3*i+1] = r.y; out[3*i+2] = r.z;
    }
}
double oracle_reflectance(double cosine, double ratio) { return reflectance(cosine, ratio); }

// ================================================================ samplers on the MT stream (as ref_*)
void oracle_unit_sphere_points(uint32_t skip, uint32_t n, float* out, float* next_draw)
{
    MtRng r(skip, 0);
    for (uint32_t i = 0; i < n; ++i) { V3 p = random_point_in_unit_sphere(r); out[3*i] = p.x; out[3*i+1] = p.y; out[3*i+2] = p.z; }
    *next_draw = r.f32();
}
void oracle_unit_vectors(uint32_t skip, uint32_t n, float* out, float* next_draw)
{
    MtRng r(skip, 0);
    for (uint32_t i = 0; i < n; ++i) { V3 p = random_unit_vector(r); out[3*i] = p.x; out[3*i+1] = p.y; out[3*i+2] = p.z; }
    *next_draw = r.f32();
}
void oracle_unit_circle_points(uint32_t skip, uint32_t n, float* out, float* next_draw)
{
    MtRng r(skip, 0);
    for (uint32_t i = 0; i < n; ++i) { V3 p = random_point_in_unit_circle(r); out[2*i] = p.x; out[2*i+1] = p.y; }
    *next_draw = r.f32();
}
// get_ray, n consecutive calls.  draw_ofs (n+1 entries, optional) = f32 draws consumed before ray i.
void oracle_get_rays(const float* c, uint32_t skip, uint32_t n, const float* uv, float* out_rays, float* next_draw, uint64_t* draw_ofs)
{
    Camera cam; cam.pos = v3(c[0], c[1], c[2]); cam.dir = v3(c[3], c[4], c[5]); cam.up = v3(c[6], c[7], c[8]);
    cam.fov = c[9]; cam.aspect = c[10]; cam.focus = c[11]; cam.lensRadius = c[12];
    MtRng r(skip, 0);
    for (uint32_t i = 0; i < n; ++i)
    {
        if (draw_ofs) draw_ofs[i] = r.n32;
        Ray ray = get_ray(cam, uv[2*i], uv[2*i+1], r);
        float* o = out_rays + 6*i;
        o[0] = ray.origin.x; o[1] = ray.origin.y; o[2] = ray.origin.z; o[3] = ray.dir.x; o[4] = ray.dir.y; o[5] = ray.dir.z;
    }
    if (draw_ofs) draw_ofs[n] = r.n32;
    *next_draw = r.f32();
}
// get_ray on explicit tapes: ray i reads tape[tape_ofs[i] ..).
void oracle_get_rays_tape(const float* c, uint32_t n, const float* uv, const float* tape, const uint64_t* tape_ofs, float* out_rays)
{
    Camera cam; cam.pos = v3(c[0], c[1], c[2]); cam.dir = v3(c[3], c[4], c[5]); cam.up = v3(c[6], c[7], c[8]);
    cam.fov = c[9]; cam.aspect = c[10]; cam.focus = c[11]; cam.lensRadius = c[12];
    for (uint32_t i = 0; i < n; ++i)
    {
        TapeRng r(tape + tape_ofs[i], tape_ofs[i+1] - tape_ofs[i], 0, 0);
        Ray ray = get_ray(cam, uv[2*i], uv[2*i+1], r);
        float* o = out_rays + 6*i;
        o[0] = ray.origin.x; o[1] = ray.origin.y; o[2] = ray.origin.z; o[3] = ray.dir.x; o[4] = ray.dir.y; o[5] = ray.dir.z;
    }
}

// ================================================================ queries and paths
void oracle_bvh_query(void* h, uint32_t n, const float* rays, const float* times, int query, float* out_t, int32_t* out_obj)
{
    Scene* s = S(h);
    for (uint32_t i = 0; i < n; ++i)
    {
        const float* r = rays + 6*i;
        Ray ray = make_ray(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), true);
        int32_t obj = -1;
        out_t[i] = (query == QUERY_BVH_QUIRK) ? bvh_query_quirk(*s, s->root, ray, times[i], &obj) : query_closest(*s, ray, times[i], &obj);
        out_obj[i] = obj;
    }
}

// Closest hit of planes+spheres for n rays (cast_ray's first half): t, prim, point, normal.
void oracle_closest_hits(void* h, uint32_t n, const float* rays, const float* times, int query,
                         float* out_t, int32_t* out_prim, float* out_point, float* out_normal)
{
    Scene* s = S(h);
    for (uint32_t i = 0; i < n; ++i)
    {
        const float* r = rays + 6*i;
        Ray ray = make_ray(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), true);
        Hit hit = closest_hit(*s, ray, times[i], query);
        out_t[i] = hit.t; out_prim[i] = hit.prim;
        out_point[3*i] = hit.point.x; out_point[3*i+1] = hit.point.y; out_point[3*i+2] = hit.point.z;
        out_normal[3*i] = hit.normal.x; out_normal[3*i+1] = hit.normal.y; out_normal[3*i+2] = hit.normal.z;
    }
}

// One scatter event per record on explicit tapes: incident dir, hit point, normal, material -> new dir, alive flag.
void oracle_scatter_tape(uint32_t n, const float* dirs, const float* points, const float* normals, const uint32_t* type,
                         const float* rgba, const float* param, const float* tape32, const uint64_t* ofs32,
                         const double* tape64, const uint64_t* ofs64, float* out_dir, uint32_t* out_event, uint32_t* out_used32)
{
    for (uint32_t i = 0; i < n; ++i)
    {
        Ray ray = make_ray(v3(0, 0, 0), v3(dirs[3*i], dirs[3*i+1], dirs[3*i+2]), true);
        Material m; m.type = type[i]; m.colour = v4(rgba[4*i], rgba[4*i+1], rgba[4*i+2], rgba[4*i+3]); m.roughness = param[2*i]; m.n = param[2*i+1];
        Hit hit; hit.t = 1; hit.prim = 0; hit.m = &m;
        hit.point = v3(points[3*i], points[3*i+1], points[3*i+2]); hit.normal = v3(normals[3*i], normals[3*i+1], normals[3*i+2]);
        TapeRng r(tape32 + ofs32[i], ofs32[i+1] - ofs32[i], tape64 + ofs64[i], ofs64[i+1] - ofs64[i]);
        Ray next; next.origin = v3(0, 0, 0); next.dir = v3(0, 0, 0); V4 att; uint32_t ev;
        scatter(ray, hit, r, &next, &att, &ev);
        out_dir[3*i] = next.dir.x; out_dir[3*i+1] = next.dir.y; out_dir[3*i+2] = next.dir.z;
        out_event[i] = ev | (r.overrun ? 0x80000000u : 0);
        if (out_used32) out_used32[i] = (uint32_t)r.n32;
    }
}

// cast_ray for n rays consecutively on ONE MT stream (as ref_cast_rays).  Optional outputs (may be NULL):
//   ofs32/ofs64 (n+1): draws consumed before ray i  -> slices of the MT stream = per-ray tapes
//   segs (n): loop iterations of ray i
void oracle_cast_rays_mt(void* h, uint32_t n, const float* rays, const float* times, uint32_t max_depth,
                         uint32_t skip32, uint32_t skip64, int query, int product, float* out_rgba,
                         float* next_f32, double* next_f64, uint64_t* ofs32, uint64_t* ofs64, uint32_t* segs)
{
    Scene* s = S(h);
    MtRng r(skip32, skip64);
    TraceOpts opt = {query, product};
    for (uint32_t i = 0; i < n; ++i)
    {
        const float* p = rays + 6*i;
        Ray ray = make_ray(v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5]), true);
        if (ofs32) ofs32[i] = r.n32;
        if (ofs64) ofs64[i] = r.n64;
        uint32_t ns = 0;
        V4 c = cast_ray(*s, ray, max_depth, times[i], r, opt, 0, &ns);
        if (segs) segs[i] = ns;
        out_rgba[4*i] = c.x; out_rgba[4*i+1] = c.y; out_rgba[4*i+2] = c.z; out_rgba[4*i+3] = c.w;
    }
    if (ofs32) ofs32[n] = r.n32;
    if (ofs64) ofs64[n] = r.n64;
    *next_f32 = r.f32();
    *next_f64 = r.f64();
}

// cast_ray on explicit per-ray tapes (what the GPU path probe is fed).  seg_log (optional): up to max_log
// Segments per ray as 12 floats {t, prim(bits), point[3], normal[3], newDir[3], event(bits)}; seg_count (n).
void oracle_cast_rays_tape(void* h, uint32_t n, const float* rays, const float* times, uint32_t max_depth,
                           const float* tape32, const uint64_t* ofs32, const double* tape64, const uint64_t* ofs64,
                           int query, int product, float* out_rgba, uint32_t* out_flags,
                           float* seg_log, uint32_t max_log, uint32_t* seg_count)
{
    Scene* s = S(h);
    TraceOpts opt = {query, product};
    for (uint32_t i = 0; i < n; ++i)
    {
        const float* p = rays + 6*i;
        Ray ray = make_ray(v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5]), true);
        TapeRng r(tape32 + ofs32[i], ofs32[i+1] - ofs32[i], tape64 + ofs64[i], ofs64[i+1] - ofs64[i]);
        std::vector<Segment> log;
        uint32_t ns = 0;
        V4 c = cast_ray(*s, ray, max_depth, times[i], r, opt, seg_log ? &log : 0, &ns);
        out_rgba[4*i] = c.x; out_rgba[4*i+1] = c.y; out_rgba[4*i+2] = c.z; out_rgba[4*i+3] = c.w;
        if (out_flags) out_flags[i] = r.overrun ? 1u : 0u;
        if (seg_count) seg_count[i] = (uint32_t)log.size();
        if (seg_log)
            for (uint32_t k = 0; k < log.size() && k < max_log; ++k)
            {
                float* o = seg_log + ((size_t)i * max_log + k) * 12;
                const Segment& g = log[k];
                o[0] = g.t; memcpy(o + 1, &g.prim, 4);
                o[2] = g.point.x; o[3] = g.point.y; o[4] = g.point.z;
                o[5] = g.normal.x; o[6] = g.normal.y; o[7] = g.normal.z;
                o[8] = g.newDir.x; o[9] = g.newDir.y; o[10] = g.newDir.z; memcpy(o + 11, &g.event, 4);
            }
    }
}

// ================================================================ the per-pixel sample loop
// (a) On the MT stream, one thread, rect [x0,x1)x[y0,y1): bit-comparable with ref_render_rect
//     (run_thread_batch, main.cpp:341-361) when (query, product) = (BVH_QUIRK, RECURSIVE).
void oracle_render_rect_mt(void* h, uint32_t w, uint32_t hgt, uint32_t spp, uint32_t depth, uint32_t x0, uint32_t y0,
                           uint32_t x1, uint32_t y1, uint32_t skip32, uint32_t skip64, int query, int product, float* out_pixels)
{
    Scene* s = S(h);
    MtRng r(skip32, skip64);
    TraceOpts opt = {query, product};
    for (uint32_t py = y0; py < y1; ++py)
        for (uint32_t px = x0; px < x1; ++px)
        {
            V4 acc = v4(0, 0, 0, 0);                                               // main.cpp:345
            for (uint32_t k = 0; k < spp; ++k) acc = add4(acc, pixel_sample(*s, px, py, w, hgt, depth, r, opt, 0));
            acc = mul4(acc, 1.0f / (float)spp);                                    // main.cpp:358, vectors.cpp:100-103
            float* o = out_pixels + 4 * ((size_t)py * w + px);
            o[0] = clampf(acc.x, 0, 1); o[1] = clampf(acc.y, 0, 1); o[2] = clampf(acc.z, 0, 1); o[3] = clampf(acc.w, 0, 1);
        }
}

// (b) On the Philox stream keyed (seed, pixel = y*w+x, sample): the MI355X build's definition of a render.
//     out_sum = per-pixel SUM over samples [s0,s1) accumulated in sample order (rgba, w*h*4 floats).
//     Returns total segments (loop iterations of cast_ray) for the S-bar of SURVEY.md 8(d).
uint64_t oracle_render_philox(void* h, uint32_t w, uint32_t hgt, uint32_t s0, uint32_t s1, uint32_t depth, uint64_t seed,
                              int query, int product, uint32_t nthreads, float* out_sum)
{
    Scene* s = S(h);
    TraceOpts opt = {query, product};
    std::atomic<uint32_t> nextRow(0);
    std::atomic<uint64_t> total(0);
    if (nthreads == 0) nthreads = 1;
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < nthreads; ++t)
        pool.emplace_back([&] {
            uint64_t mine = 0;
            for (;;)
            {
                uint32_t py = nextRow.fetch_add(1);
                if (py >= hgt) break;
                for (uint32_t px = 0; px < w; ++px)
                {
                    V4 acc = v4(0, 0, 0, 0);
                    for (uint32_t k = s0; k < s1; ++k)
                    {
                        PhiloxRng r(seed, py * w + px, k);
                        uint32_t ns = 0;
                        acc = add4(acc, pixel_sample(*s, px, py, w, hgt, depth, r, opt, &ns));
                        mine += ns;
                    }
                    float* o = out_sum + 4 * ((size_t)py * w + px);
                    o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
                }
            }
            total += mine;
        });
    for (auto& t : pool) t.join();
    return total.load();
}

// (c) Individual samples on the Philox stream: out_rgba[i] = radiance of (pixels[i], samples[i]); also the
//     primary ray and the number of stream words consumed, for per-sample bit-parity with the GPU.
void oracle_samples_philox(void* h, uint32_t w, uint32_t hgt, uint32_t n, const uint32_t* pixels, const uint32_t* samples,
                           uint32_t depth, uint64_t seed, int query, int product, float* out_rgba, uint32_t* out_segments)
{
    Scene* s = S(h);
    TraceOpts opt = {query, product};
    for (uint32_t i = 0; i < n; ++i)
    {
        PhiloxRng r(seed, pixels[i], samples[i]);
        uint32_t ns = 0;
        V4 c = pixel_sample(*s, pixels[i] % w, pixels[i] / w, w, hgt, depth, r, opt, &ns);
        out_rgba[4*i] = c.x; out_rgba[4*i+1] = c.y; out_rgba[4*i+2] = c.z; out_rgba[4*i+3] = c.w;
        if (out_segments) out_segments[i] = ns;
    }
}

// ================================================================ resolve + BMP (main.cpp:358; file_io.h:4-52)
// mean = sum * (1.0f/spp) (vectors.cpp:100-103), clamp to [0,1] (utils.h:17-20).
void oracle_resolve(uint32_t npix, const float* sum_rgba, uint32_t spp, float* out_pixels)
{
    float inv = 1.0f / (float)spp;
    for (size_t i = 0; i < (size_t)npix * 4; ++i) out_pixels[i] = clampf(sum_rgba[i] * inv, 0.0f, 1.0f);
}
// gamma-2 sqrt, (u8)(255.0f*c) truncation, 0xAARRGGBB words (file_io.h:29-45).
void oracle_pack_bgra(uint32_t npix, const float* pixels, uint32_t* out)
{
    for (uint32_t i = 0; i < npix; ++i)
    {
        const float* c = pixels + 4 * (size_t)i;
        uint8_t red = (uint8_t)(255.0f * sqrtf(c[0])), green = (uint8_t)(255.0f * sqrtf(c[1]));
        uint8_t blue = (uint8_t)(255.0f * sqrtf(c[2])), alpha = (uint8_t)(255.0f * sqrtf(c[3]));
        out[i] = (uint32_t)((alpha << 24) | (red << 16) | (green << 8) | blue);
    }
}
// 58-byte header: BITMAPFILEHEADER (14, packed) + BITMAPINFO (40 + one RGBQUAD) (file_io.h:6-23).  Every field
// the reference leaves untouched is zero (VirtualAlloc memory).  Pinned by the headers of /root/reference/output/*.bmp.
void oracle_bmp_header(uint32_t w, uint32_t hgt, uint8_t* out58)
{
    memset(out58, 0, 58);
    uint32_t off = 58, size = off + 4u * w * hgt;
    int32_t negh = -(int32_t)hgt;
    uint32_t bisize = 40; uint16_t planes = 1, bits = 32;
    out58[0] = 'B'; out58[1] = 'M';          // bfType = 'MB' little-endian (file_io.h:13)
    memcpy(out58 + 2, &size, 4);             // bfSize
    memcpy(out58 + 10, &off, 4);             // bfOffBits
    memcpy(out58 + 14, &bisize, 4);          // biSize
    memcpy(out58 + 18, &w, 4);               // biWidth
    memcpy(out58 + 22, &negh, 4);            // biHeight = -h: top-down (file_io.h:20)
    memcpy(out58 + 26, &planes, 2);          // biPlanes
    memcpy(out58 + 28, &bits, 2);            // biBitCount; biCompression = BI_RGB = 0
}

// ================================================================ timing helper for bench.py's "port" baseline
double oracle_time_render_philox(void* h, uint32_t w, uint32_t hgt, uint32_t s0, uint32_t s1, uint32_t depth, uint64_t seed,
                                 uint32_t nthreads, float* out_sum, uint64_t* out_segments)
{
    auto t0 = std::chrono::steady_clock::now();
    uint64_t seg = oracle_render_philox(h, w, hgt, s0, s1, depth, seed, QUERY_CLOSEST, PRODUCT_ITERATIVE, nthreads, out_sum);
    auto t1 = std::chrono::steady_clock::now();
    if (out_segments) *out_segments = seg;
    return std::chrono::duration<double>(t1 - t0).count();
}

} // extern "C"
