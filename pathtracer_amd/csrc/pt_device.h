// pt_device.h -- device functions of the MI355X path tracer (gfx950, wave64).
//
// Two arithmetic regimes live here, on purpose (DESIGN.md "Arithmetic"):
//   * everything a result VALUE flows through (ray generation, the chosen hit's t, hit point, normal, scatter
//     directions, colours) is evaluated in the reference's own operation order with contraction OFF, so it is
//     bit-identical to the reference on identical inputs;
//   * the closest-sphere SEARCH over the LDS-resident scene -- >80 % of all instructions -- only has to pick the
//     right sphere index, so it runs contracted, packed (v_pk_fma_f32) and with the hardware sqrt.
// Reference citations are file:line in the reference tree (src/...).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#pragma clang fp contract(off)

namespace ptd {

#define PT_DEV __device__ __forceinline__

constexpr float kF32Max = 3.402823466e+38F; // types.h:9
constexpr float kMinT = 0.001f;             // main.cpp:56,178
constexpr int32_t kPrimMiss = INT32_MIN;

enum : uint32_t { MAT_NONE = 0, MAT_DIFFUSE = 1, MAT_METAL = 2, MAT_DIALECTRIC = 3 }; // render_world.h:32-38
enum : uint32_t { EV_SCATTER = 0, EV_SKY = 1, EV_DEPTH = 2, EV_ABSORB = 3, EV_NONE_MAT = 4 };

// ------------------------------------------------------------------ vectors.h / vectors.cpp (strict)
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
PT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV V4 v4(float x, float y, float z, float w) { V4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
PT_DEV V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   // vectors.cpp:13-20
PT_DEV V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   // vectors.cpp:22-29
PT_DEV V3 mul(V3 v, float s) { return v3(v.x * s, v.y * s, v.z * s); }      // vectors.cpp:31-42
PT_DEV V3 neg(V3 v) { return v3(-v.x, -v.y, -v.z); }                        // vectors.cpp:49-56
// Lane mask of a predicate.  (HIP's __ballot(int) widens the bool and compares it with zero again -- a v_cndmask and a
// v_cmp per call that the compiler does not fold; the builtin takes the i1 as it is.)
PT_DEV unsigned long long pt_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
PT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // vectors.cpp:132-135
PT_DEV float norm_squared(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; } // vectors.cpp:112-115
PT_DEV float norm(V3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); } // vectors.cpp:107-110 (correctly rounded sqrt)
PT_DEV V3 normalize(V3 v) { return mul(v, 1.0f / norm(v)); }                // vectors.cpp:44-47,127-130
PT_DEV bool near_zero(V3 v)                                                 // vectors.cpp:144-148
{
    const float e = 1e-8f;
    return fabsf(v.x) < e && fabsf(v.y) < e && fabsf(v.z) < e;
}
PT_DEV V4 hadamard(V4 a, V4 b) { return v4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); } // vectors.cpp:150-158

struct Ray { V3 o, d; };
PT_DEV V3 ray_at(const Ray& r, float t) { return add(r.o, mul(r.d, t)); }   // geometry.h:10
PT_DEV V3 reflect_direction(V3 dir, V3 n)                                   // geometry.h:62-67
{
    float projectedDistance = -dot(dir, n);
    return add(dir, mul(mul(n, 2.0f), projectedDistance));
}

// ------------------------------------------------------------------ RNG back-ends
// Philox4x32-10 stream of (seed, pixel, sample): word k = philox(ctr={k>>2,pixel,sample,0}, key=seed)[k&3]
// (defined in DESIGN.md "RNG"; pinned by the Random123 known-answer vectors and against the oracle).
PT_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* o)
{
#ifndef PT_PHILOX_HOIST
    // The key is wave-uniform (the seed) and so are the ten round keys: left alone, the compiler hoists all twenty out of
    // the render loop, runs out of scalar registers and parks them in lanes of a vector register (a v_readlane and a
    // hazard nop per use).  Opaque to the optimiser here, each round key is one scalar add where it is used.
    asm volatile("" : "+s"(k0), "+s"(k1));
#endif
#pragma unroll
    for (int r = 0; r < 10; ++r)
    {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

struct PhiloxRng
{
    static constexpr bool kPinholeSkipsLens = true;  // get_ray: no lens draws when lensRadius == 0
    static constexpr bool kStillSkipsTime = true;    // begin_sample: no time draw when t0 == t1
    uint32_t k0, k1, pixel, sample, k;
    uint32_t b0, b1, b2, b3; // words k, k+1, .. of the current block, shifted down as they are consumed
    bool primed;             // the block of word k (k on a block boundary) is already in b0..b3
    PT_DEV void init(uint64_t seed, uint32_t px, uint32_t smp)
    {
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32); pixel = px; sample = smp; k = 0;
        b0 = b1 = b2 = b3 = 0; primed = false;
    }
    PT_DEV void fill()
    {
        uint32_t o[4];
        philox4x32_10(k >> 2, pixel, sample, 0u, k0, k1, o);
        b0 = o[0]; b1 = o[1]; b2 = o[2]; b3 = o[3];
    }
    PT_DEV uint32_t word()
    {
        if ((k & 3u) == 0u)
        {
            if (!primed) fill();
            primed = false;
        }
        uint32_t w = b0;
        b0 = b1; b1 = b2; b2 = b3;
        ++k;
        return w;
    }
    PT_DEV float f32() { return (float)(word() >> 8) * 5.9604644775390625e-8f; }
    PT_DEV double f64()
    {
        uint64_t hi = word(), lo = word();
        return (double)(((hi << 32) | lo) >> 11) * 1.1102230246251565404e-16;
    }
    // Skip to the next 4-word block.  Called at the start of every scatter: lanes of a wave arrive with different
    // draw counts, and without this each of the 3 draws of a rejection attempt refills for SOME lane (three Philox
    // evaluations per attempt instead of 3/4).  Part of the stream's definition (DESIGN.md "RNG"); the oracle does the same.
    PT_DEV void align() { k = (k + 3u) & ~3u; }
    // Behind align(): evaluate the block now, for every lane that is here.  The draws that follow come from different
    // branches -- the glass lanes' f64, the first attempt of the other lanes' rejection loop -- and each would evaluate
    // the same block of its own lanes' streams in its own copy of the ten rounds, one after the other.
    PT_DEV void prime() { fill(); primed = true; }
    // align() + forget the rest of the block: the next draw refills anyway, and the four words stop being live state
    // (registers) across whatever comes in between -- the whole sphere query, in the render loop.
    PT_DEV void park() { align(); b0 = b1 = b2 = b3 = 0; primed = false; }
    // The whole block at the (aligned) position, four words at once.
    PT_DEV void take_block(uint32_t& w0, uint32_t& w1, uint32_t& w2, uint32_t& w3)
    {
        align();
        if (!primed) fill();
        primed = false;
        w0 = b0; w1 = b1; w2 = b2; w3 = b3;
        k += 4u;
    }
    PT_DEV bool overrun() const { return false; }
};

// Explicit uniforms for the parity probes.
struct TapeRng
{
    static constexpr bool kPinholeSkipsLens = false; // the reference's draw sequence, loop and all
    static constexpr bool kStillSkipsTime = false;
    const float* t32; const double* t64; uint32_t n32, n64, i32, i64; bool over;
    PT_DEV void init(const float* a, uint32_t na, const double* b, uint32_t nb) { t32 = a; n32 = na; t64 = b; n64 = nb; i32 = i64 = 0; over = false; }
    PT_DEV float f32() { if (i32 >= n32) { over = true; ++i32; return 0.5f; } return t32[i32++]; }
    PT_DEV double f64() { if (i64 >= n64) { over = true; ++i64; return 0.5; } return t64[i64++]; }
    PT_DEV void align() {}
    PT_DEV void prime() {}
    PT_DEV void park() {}
    PT_DEV bool overrun() const { return over; }
};

template <class R> PT_DEV float rand_range(R& r, float lo, float hi) { return r.f32() * (hi - lo) + lo; } // utils.h:47-51
// rand_range(r, -1, 1) = u * 2 - 1 with u = k * 2^-24: both scalings are by powers of two, hence exact, so k * 2^-23 - 1
// is the same float -- and as the product is exact, one FMA rounds exactly like the multiply-then-add (the rejection
// loops draw nothing else: two instructions less per draw).
template <class R> PT_DEV float rand_pm1(R& r) { return rand_range(r, -1.0f, 1.0f); }
PT_DEV float rand_pm1(PhiloxRng& r) { return __builtin_fmaf((float)(r.word() >> 8), 1.1920928955078125e-7f, -1.0f); }

// ------------------------------------------------------------------ samplers (utils.h:71-138)
// Draw order z,y,x / y,x: what g++ makes of the reference's unspecified argument order.
// The reference accepts a point when norm(p) <= 1 with norm = sqrt(x*x + y*y + z*z).  For a correctly rounded square
// root, sqrt(s) <= 1 holds exactly when s <= 1 + 2^-23 (sqrt(1 + 2^-23) = 1 + 2^-24 - ... rounds down to 1, sqrt of the
// next float does not; sqrt is monotone; NaN fails both) -- the same decisions without the ~15-instruction exact root
// in each trip of the two rejection loops.
constexpr float kOnePlusUlp = 1.00000011920928955078125f; // 0x3F800001
template <class R> PT_DEV V3 random_point_in_unit_sphere(R& r)               // utils.h:80-94
{
    V3 p;
    for (;;)
    {
        float z = rand_pm1(r), y = rand_pm1(r), x = rand_pm1(r);
        p = v3(x, y, z);
        if (norm_squared(p) <= kOnePlusUlp) break;                       // == (norm(p) <= 1.0f), see above
    }
    return p;
}
template <class R> PT_DEV V3 random_unit_vector(R& r) { return normalize(random_point_in_unit_sphere(r)); } // utils.h:102-108
// The same loop, left after `maxAttempts` rejected points (*found = false): the render loop lets such a lane try again
// in the wave's next trip instead of keeping the whole wave in the loop (a wave of ~45 scattering lanes needs ~5.7
// attempts for a mean of 1.9 per lane).  The draws are the uninterrupted loop's: the caller resumes at the same
// position of the same stream.  maxAttempts = 0: never left early.
template <class R> PT_DEV V3 random_point_in_unit_sphere_bounded(R& r, uint32_t maxAttempts, bool* found)
{
    V3 p = v3(0, 0, 0);
    *found = true;
    for (uint32_t attempt = 1;; ++attempt)
    {
        float z = rand_pm1(r), y = rand_pm1(r), x = rand_pm1(r);
        p = v3(x, y, z);
        if (norm_squared(p) <= kOnePlusUlp) break;                       // == (norm(p) <= 1.0f), see above
        if (attempt == maxAttempts) { *found = false; break; }
    }
    return p;
}
// The same sampler on the Philox stream: ONE block per trip of the loop instead of 3/4 block per point.
// A uniform point of the cube [-1,1)^3 with 24-bit coordinates k * 2^-23 - 1 (what three draws of rand_pm1 give) is
// drawn in two steps: the top 9 bits of each coordinate pick a cell of side 2^-8, the low 15 bits a point in the cell.
// A cell whose nearest corner is outside the unit sphere holds no acceptable point -- it is rejected on its 27 bits,
// and a 128-bit block holds three such attempts (words 0..2: z = bits 31..23, y = 22..14, x = 13..5) plus ONE set of
// low bits (z: word 3 bits 31..17, y: 16..2, x: word 3 bits 1..0 and 13 of the 15 spare low bits of words 0..2) for
// the first cell that survives.  The completed point takes the reference's test (norm(p) <= 1, utils.h:88); if it
// fails (a boundary cell), or all three cells were outside, the block is spent and the next one starts afresh.  The
// accepted point is uniform over the 24-bit points that pass the reference's test -- the reference loop's distribution
// exactly; the lanes of a wave evaluate one Philox block per trip where twelve draws cost three.
// (Cell test in integers: a = distance of the cell from the coordinate plane in cells, 0..255; the nearest corner has
// |p|^2 = (ax^2 + ay^2 + az^2) / 2^16; above 1 + 2^-16 the float test norm_squared(p) <= 1 + 2^-23 fails for the whole cell.)
PT_DEV bool philox_cell_outside(uint32_t w)
{
    const uint32_t f = w ^ 0x80402000u;       // top bit of each 9-bit field flipped: the field is the signed cell number
    const int32_t tz = (int32_t)f >> 23, ty = (int32_t)(f << 9) >> 23, tx = (int32_t)(f << 18) >> 23;
    const uint32_t az = (uint32_t)(tz ^ (tz >> 31)), ay = (uint32_t)(ty ^ (ty >> 31)), ax = (uint32_t)(tx ^ (tx >> 31));
    return ax * ax + ay * ay + az * az > 65536u;
}
PT_DEV bool philox_sphere_block(PhiloxRng& r, V3* p)
{
    uint32_t w0, w1, w2, w3;
    r.take_block(w0, w1, w2, w3);
    const bool o0 = philox_cell_outside(w0), o1 = philox_cell_outside(w1), o2 = philox_cell_outside(w2);
    const uint32_t w = !o0 ? w0 : (!o1 ? w1 : w2);
    const uint32_t spare = (w0 & 31u) | ((w1 & 31u) << 5) | ((w2 & 31u) << 10);
    const uint32_t kz = ((w >> 23) << 15) | (w3 >> 17);
    const uint32_t ky = (((w >> 14) & 511u) << 15) | ((w3 >> 2) & 0x7FFFu);
    const uint32_t kx = (((w >> 5) & 511u) << 15) | ((w3 & 3u) << 13) | (spare & 0x1FFFu);
    // k * 2^-23 - 1: the product is exact, so one FMA rounds like the reference's multiply-then-add (see rand_pm1)
    *p = v3(__builtin_fmaf((float)kx, 1.1920928955078125e-7f, -1.0f), __builtin_fmaf((float)ky, 1.1920928955078125e-7f, -1.0f),
            __builtin_fmaf((float)kz, 1.1920928955078125e-7f, -1.0f));
    return !(o0 && o1 && o2) && norm_squared(*p) <= kOnePlusUlp;          // == (norm(p) <= 1.0f), see above
}
// maxBlocks = 0: until a point is found; else *found = false after that many spent blocks (the caller tries again later).
PT_DEV V3 random_point_in_unit_sphere_bounded(PhiloxRng& r, uint32_t maxBlocks, bool* found)
{
    V3 p = v3(0, 0, 0);
    *found = true;
    for (uint32_t n = 1;; ++n)
    {
        if (philox_sphere_block(r, &p)) break;
        if (n == maxBlocks) { *found = false; break; }
    }
    return p;
}
PT_DEV V3 random_point_in_unit_sphere(PhiloxRng& r) { bool found; return random_point_in_unit_sphere_bounded(r, 0u, &found); }
PT_DEV V3 random_unit_vector(PhiloxRng& r) { return normalize(random_point_in_unit_sphere(r)); }
template <class R> PT_DEV V3 random_point_in_unit_circle(R& r)               // utils.h:124-138
{
    V3 p;
    for (;;)
    {
        float y = rand_pm1(r), x = rand_pm1(r);
        p = v3(x, y, 0.0f);
        if (norm_squared(p) <= kOnePlusUlp) break;                       // == (norm(p) <= 1.0f), see above
    }
    return p;
}

// ------------------------------------------------------------------ camera (camera.cpp:57-79)
// Everything that does not depend on (u,v) is folded on the host in the same f32 operation order
// (pt_host.cpp: make_device_camera), so the per-ray part below continues the reference's expression trees.
struct DeviceCamera
{
    float pos[3], horizontal[3], vertical[3], topLeft[3];
    float planeW, planeH, focus, lensRadius;
};
template <class R> PT_DEV Ray get_ray(const DeviceCamera& c, float u, float v, R& rng)
{
    V3 pos = v3(c.pos[0], c.pos[1], c.pos[2]);
    V3 horizontal = v3(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
    V3 vertical = v3(c.vertical[0], c.vertical[1], c.vertical[2]);
    V3 topLeft = v3(c.topLeft[0], c.topLeft[1], c.topLeft[2]);
    V3 rayTarget = sub(add(topLeft, mul(mul(mul(horizontal, u), c.planeW), c.focus)),
                       mul(mul(mul(vertical, v), c.planeH), c.focus));                        // camera.cpp:70
    // camera.cpp:74 runs its rejection loop even for a pinhole camera and multiplies the point by lensRadius = 0.  On the
    // Philox stream (this build's own) a pinhole camera draws nothing here: the point is (0,0,0), the
    // ray is the same (p * 0 = 0 whatever p) and a sample of a pinhole scene needs ONE Philox block instead of two.  The
    // tape-driven probes, which replay the reference's draws, keep the loop.
    V3 disc = v3(0.0f, 0.0f, 0.0f);
    if (!(R::kPinholeSkipsLens && c.lensRadius == 0.0f)) disc = random_point_in_unit_circle(rng);
    V3 pointOnLens = mul(disc, c.lensRadius);                                                  // camera.cpp:74
    V3 lensOffset = add(mul(horizontal, pointOnLens.x), mul(vertical, pointOnLens.y));         // camera.cpp:75
    Ray r;
    r.o = add(pos, lensOffset);
    r.d = normalize(sub(rayTarget, add(pos, lensOffset)));                                     // camera.cpp:77
    return r;
}

// ------------------------------------------------------------------ scene view (LDS-resident blob)
// Blob = array of float4, identical in HBM and LDS (pt_host.cpp: build_blob):
//   pairs   [2*npairs] : A = {cx0,cx1,cy0,cy1}  B = {cz0,cz1,r2_0,r2_1}     (two spheres per entry, SoA-of-pairs)
//   velp    [2*npairs] : A = {vx0,vx1,vy0,vy1}  B = {vz0,vz1,0,0}           (only if the scene moves)
//   groups  [ngroups]  : bounding spheres of pairs 4g..4g+3 (8 spheres, swept over [t0,t1]), SoA-of-4:
//                        {cx0..3},{cy0..3},{cz0..3},{R2_0..3} per four groups; ngroups is a multiple of 4
//   smat    [2*slots]  : {type bits, roughness, n, original index bits}, {r,g,b,a}
//   planes  [np]       : {nx,ny,nz,offset}
//   pmat    [2*np]     : as smat
// Spheres are stored in a spatially sorted order (slots); npairs is a multiple of 4; unused slots are spheres with
// r2 = -1 (negative discriminant for every ray) and original index 0xFFFFFFFF.
struct SceneView
{
    const float4* pairs; const float4* velp; const float4* groups; const float4* smat; const float4* planes; const float4* pmat;
    uint32_t ns, npairs, ngroups, np; uint32_t idx_mask; // ns = 2*npairs slots; idx_mask = 2^bits-1 >= ns-1
};

struct Material { uint32_t type; V4 colour; float roughness, n; };
PT_DEV Material load_material(const float4* m, uint32_t i)
{
    float4 a = m[2 * i], b = m[2 * i + 1];
    Material r; r.type = __float_as_uint(a.x); r.roughness = a.y; r.n = a.z; r.colour = v4(b.x, b.y, b.z, b.w);
    return r;
}
PT_DEV uint32_t original_index(const SceneView& s, uint32_t slot) { return __float_as_uint(s.smat[2 * slot].w); }

PT_DEV void load_sphere(const SceneView& s, uint32_t i, V3* c, float* r2)
{
    const float* p = reinterpret_cast<const float*>(s.pairs + 2 * (i >> 1));
    uint32_t l = i & 1u;
    *c = v3(p[l], p[2 + l], p[4 + l]);
    *r2 = p[6 + l];
}
PT_DEV V3 load_velocity(const SceneView& s, uint32_t i)
{
    const float* p = reinterpret_cast<const float*>(s.velp + 2 * (i >> 1));
    uint32_t l = i & 1u;
    return v3(p[l], p[2 + l], p[4 + l]);
}

// intersection_test(Ray, Sphere) in the reference's arithmetic (geometry.cpp:39-68); r2 = radius*radius.
PT_DEV float intersect_sphere_exact(const Ray& ray, V3 centre, float r2)
{
    float tResult = kF32Max;
    V3 oc = sub(ray.o, centre);
    float b = 2 * dot(ray.d, oc);
    float c = dot(oc, oc) - r2;
    float discriminant = b * b - 4 * c;
    if (discriminant > 0)
    {
        float rootValue = sqrtf(discriminant);
        tResult = (-b - rootValue) / 2;
    }
    else if (discriminant == 0)
        tResult = -b / 2;
    return tResult;
}
// intersection_test(Ray, Plane) (geometry.cpp:70-81)
PT_DEV float intersect_plane_exact(const Ray& ray, V3 n, float offset)
{
    float tResult = kF32Max;
    float denominator = dot(ray.d, n);
    if (denominator != 0.0f) tResult = (offset - dot(n, ray.o)) / denominator;
    return tResult;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The search: which sphere SLOT has the smallest t > kMinT.  Returns a slot to evaluate exactly, or -1.
//
// Two levels.  (1) Spheres are stored spatially sorted, 8 to a group with a bounding sphere; a group is searched only
// if the ray of at least ONE live lane of the wave can touch its bound (a wave-uniform branch on a ballot: no
// divergence, and all lanes read the same LDS address -> broadcast).  (2) Inside a group, brute force over 4 sphere
// pairs in packed math.
// Key trick of (2): w = t - kMinT as raw bits; valid hits are positive floats, which order like unsigned integers; a
// negative w (sphere behind the ray / the sphere just left) has the sign bit set and a NaN (no real root) is above
// +inf, so ONE unsigned min does validity test and closest test.  The index rides in the low bits of the key:
//   WIDE = false (<= 256 pairs): key = {w[31:8], pair index} by one v_perm_b32 on min(w0,w1); the two spheres of
//                                the winning pair are then both evaluated exactly by the caller.  Ordering
//                                resolution between DIFFERENT pairs: 2^-15 relative.
//   WIDE = true                : key = (w & ~mask) | slot (v_and + v_or per sphere), resolution 2^-(23-bits).
// The resolution only matters for two spheres hit at distances that close, i.e. the seam of interpenetrating spheres.
template <bool MOVING, bool WIDE> PT_DEV int32_t search_spheres_fast(const SceneView& s, const Ray& ray, float time, bool live, uint32_t* searched)
{
#pragma clang fp contract(fast)
    // Origin advanced by kMinT along the (unit) direction: the near root measured from there IS t - kMinT.
    const float sx = ray.o.x + kMinT * ray.d.x, sy = ray.o.y + kMinT * ray.d.y, sz = ray.o.z + kMinT * ray.d.z;
    const f32x2 ox = {sx, sx}, oy = {sy, sy}, oz = {sz, sz};
    const f32x2 dx = {ray.d.x, ray.d.x}, dy = {ray.d.y, ray.d.y}, dz = {ray.d.z, ray.d.z};
    const f32x2 tt = {time, time};
    const uint32_t keep = ~s.idx_mask;
    uint32_t best = 0x7F800000u; // +inf
    // Bounds are tested four groups at a time (SoA-of-4: {cx0..3},{cy0..3},{cz0..3},{R2_0..3}) in packed math, with
    // the four LDS reads in flight together; then only the groups some live lane can touch are searched.
    const uint32_t nb = s.ngroups >> 2; // ngroups is a multiple of 4 (unused bounds have R2 = -1: never touched)
    for (uint32_t q = 0; q < nb; ++q)
    {
        const float4 X = s.groups[4 * q], Y = s.groups[4 * q + 1], Z = s.groups[4 * q + 2], R = s.groups[4 * q + 3];
        uint32_t todo = 0;
        {
            const f32x2 gx0 = ox - f32x2{X.x, X.y}, gy0 = oy - f32x2{Y.x, Y.y}, gz0 = oz - f32x2{Z.x, Z.y};
            const f32x2 gx1 = ox - f32x2{X.z, X.w}, gy1 = oy - f32x2{Y.z, Y.w}, gz1 = oz - f32x2{Z.z, Z.w};
            const f32x2 hb0 = gx0 * dx + gy0 * dy + gz0 * dz, hb1 = gx1 * dx + gy1 * dy + gz1 * dz;
            const f32x2 cc0 = gx0 * gx0 + (gy0 * gy0 + (gz0 * gz0 - f32x2{R.x, R.y}));
            const f32x2 cc1 = gx1 * gx1 + (gy1 * gy1 + (gz1 * gz1 - f32x2{R.z, R.w}));
            const f32x2 ds0 = hb0 * hb0 - cc0, ds1 = hb1 * hb1 - cc1;
            // the line meets the bound, and the bound is not wholly behind the origin
            const bool t0 = live && ds0.x > 0.0f && !(hb0.x > 0.0f && cc0.x > 0.0f);
            const bool t1 = live && ds0.y > 0.0f && !(hb0.y > 0.0f && cc0.y > 0.0f);
            const bool t2 = live && ds1.x > 0.0f && !(hb1.x > 0.0f && cc1.x > 0.0f);
            const bool t3 = live && ds1.y > 0.0f && !(hb1.y > 0.0f && cc1.y > 0.0f);
            todo = (pt_ballot(t0) ? 1u : 0u) | (pt_ballot(t1) ? 2u : 0u) | (pt_ballot(t2) ? 4u : 0u) | (pt_ballot(t3) ? 8u : 0u);
        }
        while (todo) // wave-uniform
        {
            const uint32_t g = 4 * q + (uint32_t)__builtin_ctz(todo);
            todo &= todo - 1u;
            ++*searched; // groups this wave had to search
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
            {
                const uint32_t j = 4 * g + u;
                float4 A = s.pairs[2 * j], B = s.pairs[2 * j + 1];
                f32x2 cx = {A.x, A.y}, cy = {A.z, A.w}, cz = {B.x, B.y}, r2 = {B.z, B.w};
                if (MOVING)
                {
                    float4 VA = s.velp[2 * j], VB = s.velp[2 * j + 1];
                    f32x2 vx = {VA.x, VA.y}, vy = {VA.z, VA.w}, vz = {VB.x, VB.y};
                    cx = vx * tt + cx; cy = vy * tt + cy; cz = vz * tt + cz; // main.cpp:66
                }
                f32x2 ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
                f32x2 hb = ocx * dx + ocy * dy + ocz * dz;                      // b/2
                f32x2 c = ocx * ocx + (ocy * ocy + (ocz * ocz - r2));
                f32x2 disc = hb * hb - c;                                       // discriminant/4
                float w0 = -hb.x - __builtin_amdgcn_sqrtf(disc.x);              // t - kMinT; NaN when disc < 0
                float w1 = -hb.y - __builtin_amdgcn_sqrtf(disc.y);
                if (WIDE)
                {
                    uint32_t k0 = (__float_as_uint(w0) & keep) | (2 * j);
                    uint32_t k1 = (__float_as_uint(w1) & keep) | (2 * j + 1);
                    best = min(best, min(k0, k1));
                }
                else
                {
                    uint32_t m = min(__float_as_uint(w0), __float_as_uint(w1));
                    best = min(best, __builtin_amdgcn_perm(m, j, 0x07060500u)); // {m.b3, m.b2, m.b1, j.b0}
                }
            }
        }
    }
    if (best >= 0x7F800000u) return -1;
    return WIDE ? (int32_t)(best & s.idx_mask) : (int32_t)(2u * (best & 0xFFu));
}

// The same query in reference arithmetic for EVERY sphere; an exact tie goes to the lowest ORIGINAL index, as in a
// loop over the caller's sphere order.
template <bool MOVING> PT_DEV int32_t search_spheres_strict(const SceneView& s, const Ray& ray, float time, float* tOut)
{
    float best = kF32Max;
    int32_t obj = -1;
    uint32_t bestOrig = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < s.ns; ++i)
    {
        V3 c; float r2;
        load_sphere(s, i, &c, &r2);
        if (MOVING) c = add(c, mul(load_velocity(s, i), time));
        const float t = intersect_sphere_exact(ray, c, r2);
        const uint32_t orig = original_index(s, i);
        if (t > kMinT && (t < best || (t == best && orig < bestOrig)) && orig != 0xFFFFFFFFu) { best = t; obj = (int32_t)i; bestOrig = orig; }
    }
    *tOut = best;
    return obj;
}

// ------------------------------------------------------------------ BVH in HBM (scenes that overflow LDS)
// Node / payload layout: pt_bvh.h.  Per-lane traversal with a 32-entry stack in LDS (lane-strided, conflict-free),
// near child first, boxes culled against the current closest t.  Leaf spheres are tested in the REFERENCE's arithmetic,
// so the result is the exact closest hit (smallest t > kMinT, lowest original index on an exact tie) whatever the
// tree looks like -- the same definition the LDS search implements.
constexpr uint32_t kBvhLeafFlag = 0x80000000u;
constexpr float kCullSlack = 1.002f;         // box culling distance = closest t so far x this (see bvh_closest)
constexpr uint32_t kBvhDone = 0xFFFFFFFFu;   // stack sentinel: "nothing left" (a real leaf reference has first < 2^28, so it is never all ones)
constexpr uint32_t kBvhStack = 64; // deepest tree the traversal accepts (the per-lane LDS stack holds depth + 1 entries; host checks)
struct BvhView
{
    const float4* nodes; const float4* sph; const float4* vel; const float4* mat;
    uint32_t root_ref, n_spheres;
    float gorg[3], gstep[3]; // MODE_BVH_Q: the grid of the 32-byte nodes (pt_bvh.h, quantise_nodes)
};

PT_DEV bool sphere_clearly_missed(const Ray& ray, V3 c, float r2)
{
#pragma clang fp contract(fast)
    const float ox = ray.o.x - c.x, oy = ray.o.y - c.y, oz = ray.o.z - c.z;
    const float hb = ox * ray.d.x + oy * ray.d.y + oz * ray.d.z;
    const float oo = ox * ox + oy * oy + oz * oz;
    // discriminant hb^2 - (oo - r2) clearly negative: below -1e-5 (hb^2 + oo) -- the rounding of either evaluation, the
    // reference's and this one, is bounded by a few ulp of the terms that cancel (hb^2 and |oc|^2).  As two FMAs:
    return (hb * hb) * 1.00001f + r2 < oo * 0.99999f;
}

// Slab test of one child box against the ray segment [0, best]; contracted arithmetic (boxes are padded by the builder).
PT_DEV bool box_hit(float lox, float loy, float loz, float hix, float hiy, float hiz, float ix, float iy, float iz,
                    float nox, float noy, float noz, float best, float* tnear)
{
#pragma clang fp contract(fast)
    const float a0 = lox * ix + nox, a1 = hix * ix + nox; // (lo - o) * inv as lo*inv - o*inv
    const float b0 = loy * iy + noy, b1 = hiy * iy + noy;
    const float c0 = loz * iz + noz, c1 = hiz * iz + noz;
    const float tn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1));
    const float tf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
    *tnear = tn;
    return tf >= fmaxf(tn, 0.0f) && tn <= best;
}
// The same test for a CONE around the ray (half-angle kConeEps): the box grown by kConeEps * t at distance t.  Per axis
//     o + d t - eps t <= hi   and   o + d t + eps t >= lo,
// i.e. the usual slab test with one reciprocal per PLANE: r1 = 1/(d - eps) for hi, r2 = 1/(d + eps) for lo (folded per
// ray together with -o r1, -o r2).  Same instruction count per box as box_hit.  |d| <= eps (the cone outgrows the axis):
// d - eps < 0 < d + eps, both inequalities bound t from BELOW and nothing from above (a side branch, ~3e-3 of rays).
// Used where the reference's sphere formula works far from its origin (trees in HBM = big scenes): there it reports
// hits on spheres the ray misses by up to sqrt(45 * 2^-24) / 2 * |o-c| = 8.2e-4 |o-c| (cancellation in b*b - 4c), and a
// brute-force search in that arithmetic -- the definition of the query -- finds them; the cone makes the tree visit them.
#ifndef PT_CONE_EPS
#define PT_CONE_EPS 1.0e-3f // (timing experiments only: a smaller cone loses the reference formula's phantom hits)
#endif
constexpr float kConeEps = PT_CONE_EPS;
PT_DEV bool cone_flat(float d) { return !(fabsf(d) > kConeEps * 1.001f); } // the cone outgrows this axis
// lo planes go with (r2, n2), hi planes with (r1, n1); the coordinates are floats (64-byte nodes) or grid cells (grid
// nodes, whose grid is folded into the pairs).
// A flat axis (|d| <= eps: d - eps < 0 < d + eps) bounds t from below with BOTH planes and not at all from above:
//   FLAT_EXACT (trees in HBM): near = max of the two, far = +big.  Branch-free and for free: with a per-ray, per-axis
//     selector s = -big (ordinary axis) / +big (flat axis), near = med3(a0, a1, s) and far = max3(a0, a1, s) -- the same
//     two instructions as min / max.  (Round 2 first had a side branch here: its seven selects per box were issued for
//     every wave step, and half of all wave steps have a flat lane.)
//     Exactness matters there: a ray that runs parallel to a slab of 10^6 spheres, outside of it, is kept out of every
//     box by that second bound; without it such a ray visits everything along its way, and a handful of those rays is
//     the launch's tail (measured: C5 29 -> 146 ms).
//   else (tree in LDS, a few hundred nodes, no registers to spare for the selectors): the hi-plane pair of a flat axis
//     is (0, +big) (set up once per ray), i.e. that plane's value is +big -- its lower bound is dropped, a superset of
//     the cone/box overlap.
PT_DEV float cone_flat_select(float d) { return cone_flat(d) ? kF32Max : -kF32Max; }
template <bool FLAT_EXACT>
PT_DEV bool cone_planes(float lox, float hix, float loy, float hiy, float loz, float hiz, float r1x, float r1y, float r1z, float n1x, float n1y, float n1z,
                        float r2x, float r2y, float r2z, float n2x, float n2y, float n2z, float sx, float sy, float sz, float best, float* tnear)
{
#pragma clang fp contract(fast)
    const float a0 = lox * r2x + n2x, a1 = hix * r1x + n1x;
    const float b0 = loy * r2y + n2y, b1 = hiy * r1y + n1y;
    const float c0 = loz * r2z + n2z, c1 = hiz * r1z + n1z;
    float nx, ny, nz, fx, fy, fz;
    if (FLAT_EXACT)
    {
        nx = __builtin_amdgcn_fmed3f(a0, a1, sx); ny = __builtin_amdgcn_fmed3f(b0, b1, sy); nz = __builtin_amdgcn_fmed3f(c0, c1, sz);
        fx = fmaxf(fmaxf(a0, a1), sx); fy = fmaxf(fmaxf(b0, b1), sy); fz = fmaxf(fmaxf(c0, c1), sz);
    }
    else
    {
        nx = fminf(a0, a1); ny = fminf(b0, b1); nz = fminf(c0, c1);
        fx = fmaxf(a0, a1); fy = fmaxf(b0, b1); fz = fmaxf(c0, c1);
    }
    const float tn = fmaxf(fmaxf(nx, ny), nz);
    const float tf = fminf(fminf(fx, fy), fz);
    *tnear = tn;
    return tf >= fmaxf(tn, 0.0f) && tn <= best;
}
// the per-plane pairs of one axis (hi plane: t (d - eps) <= hi - o -> (r1, n1); lo plane: t (d + eps) >= lo - o -> (r2, n2));
// on a flat axis the divisors are kept away from zero so that the reciprocals stay finite
template <bool FLAT_EXACT>
PT_DEV void cone_axis(float o, float d, float& r1, float& n1, float& r2, float& n2)
{
    const bool flat = cone_flat(d);
    const float s2 = flat ? fmaxf(d + kConeEps, kConeEps * 1.0e-3f) : d + kConeEps;
    r2 = __builtin_amdgcn_rcpf(s2); n2 = -o * r2;
    if (FLAT_EXACT)
    {
        const float s1 = flat ? fminf(d - kConeEps, -kConeEps * 1.0e-3f) : d - kConeEps;
        r1 = __builtin_amdgcn_rcpf(s1); n1 = -o * r1;
    }
    else
    {
        r1 = __builtin_amdgcn_rcpf(d - kConeEps); n1 = -o * r1;
        if (flat) { r1 = 0.0f; n1 = 1.0e30f; }
    }
}

struct RefView { const float4* nodes; uint32_t root, n_nodes; };
constexpr uint32_t kRefLeaf = 0xFFFFFFFFu;

struct QueryCtx // per-lane scratch and counters of the sphere query
{
    BvhView bvh; RefView ref; uint32_t* stack; uint32_t searched, nodeVisits, sphereTests, leafBatch;
    int16_t* stack16;                            // MODE_BVH_LDS: the same per-lane column with 16-bit entries (see "tree in LDS" below)
    uint32_t laneTests;                          // MODE_BVH_LDS: this LANE's leaf sphere tests (summed over the wave when the kernel ends)
    uint32_t* ovf; uint32_t stackCap;            // MODE_BVH_Q4: levels >= stackCap of the traversal stack live in this per-lane (scratch) array
#ifdef PT_STAMPS
    uint32_t innerIters, leafIters;              // wave-level trips through the two kinds of traversal step
    uint32_t leafSlotIters, leafExactIters;      // ... through the per-sphere loop of a leaf step, and how many of those ran the exact test
    unsigned long long cycInner, cycLeaf;        // s_memtime spent in them
#endif
};
#ifdef PT_STAMPS
#define PT_QSTAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PT_QSTAMP(var) do { } while (0)
#endif

// A wave-uniform value the compiler has lost track of: back into a scalar register.
PT_DEV uint32_t pt_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The traversal of one lane between calls of bvh_run: where it stands (node / leaf reference, or kBvhDone), its stack
// pointer (the stack itself lives in LDS), the closest hit so far.  A query is trav_begin -> bvh_run (once, or several
// times when the run is left early) -> the hit record.
// (pk = stack level | (index of the closest plane + 1) << 8: the state is live across the shading of other lanes, where
// registers are short.  Parking it in LDS instead was tried: three more stack entries per lane cost the seventh
// workgroup of a CU.)
struct TravState { uint32_t cur; float best; int32_t bestSlot; uint32_t pk; };
PT_DEV int32_t trav_plane(uint32_t pk) { return (int32_t)((pk >> 8) & 0xFFu) - 1; } // (<= 64 planes; the bits above are the render loop's flags)
// ---- the tree in LDS (MODE_BVH_LDS; BLK = true below) differs from the trees in HBM in two ways, both to fit more into the
// LDS that bounds its resident waves:
//  * 16-bit stack entries.  References are kept sign-extended: an inner node is its index (< 0x8000), a leaf is
//    0xFFFF8000 | (count - 1) << 12 | leaf index (the sign extension of the 16-bit entry 0x8000 | ...; bit 14 stays 0), the
//    sentinel is all ones in both widths -- so a push is a 16-bit store of the low half and a pop a sign-extending load, no
//    conversion, and `cur & kBvhLeafFlag` / `cur == kBvhDone` mean what they mean for the 32-bit references.
//  * leaf blocks.  A leaf's (up to four) spheres sit in one structure-of-arrays block {cx[4], cy[4], cz[4], r2[4]} (+
//    {vx[4], vy[4], vz[4]} when spheres move): the conservative filter runs over all four at once in packed FP32, and only
//    the candidates it leaves get the exact test, one per trip of a short loop.  An unused slot has r2 = -1e30 (never a
//    candidate).  Sphere "slots" are 4 x leaf + position; the materials in HBM are laid out to match.
constexpr uint32_t kLdsLeafBase = 0xFFFF8000u;
template <bool BLK = false>
PT_DEV void trav_begin(const BvhView& b, float tLimit, int32_t planeIdx, bool live, const QueryCtx& q, TravState& ts)
{
    ts.best = tLimit;                  // a plane hit found earlier wins ties against spheres (main.cpp:204)
    ts.bestSlot = -1;
    // The stack pointer is the LDS address itself (this lane's column, 64 dwords per level); level 0 holds a sentinel,
    // so "pop" needs no emptiness test: popping the sentinel ends the lane's traversal.
    ts.pk = (uint32_t)(planeIdx + 1) << 8;
    if (BLK) *q.stack16 = (int16_t)-1; else *q.stack = kBvhDone;
    // A lane without a live path holds a stale ray: it starts on the sentinel, so it cannot keep the wave in the loop.
    ts.cur = (live && b.n_spheres != 0) ? b.root_ref : kBvhDone;
}

// One leaf block against one ray (lanes with !active ride along): the conservative filter of sphere_clearly_missed over
// the block's four slots, then the exact (reference-arithmetic) test for the candidates it leaves, one per lane and trip
// of a wave-uniform loop.  Updates the lane's closest hit (slot = 4 x block + position).
// The filter is plain FP32, one slot after the other.  Packed FP32 (two slots per v_pk_* instruction) was built and
// measured twice: a packed FMA issues at half the plain rate on this chip and wants its operands in register pairs -- the
// tree kernel lost 3.3 % to the spills, and once the build stopped the compiler from pairing scalar work on its own
// (-fno-slp-vectorize, Makefile) the plain filter also won in the few-sphere loop (+6.5 / +10.5 % on the C2 / C4 scenes).
template <bool MOVING>
PT_DEV void block_test(const BvhView& b, uint32_t leaf, bool active, const Ray& ray, float time, QueryCtx& q, float& best, float& cull, int32_t& bestSlot)
{
    uint32_t cand = 0;
    if (active)
    {
#pragma clang fp contract(fast)
        const float4 X = b.sph[4 * leaf], Y = b.sph[4 * leaf + 1], Z = b.sph[4 * leaf + 2], R = b.sph[4 * leaf + 3];
        float4 VX = X, VY = Y, VZ = Z;
        if (MOVING) { VX = b.vel[3 * leaf]; VY = b.vel[3 * leaf + 1]; VZ = b.vel[3 * leaf + 2]; }
        auto slot = [&](float cx, float cy, float cz, float r2, float vx, float vy, float vz) {
            if (MOVING) { cx = vx * time + cx; cy = vy * time + cy; cz = vz * time + cz; }
            const float gx = ray.o.x - cx, gy = ray.o.y - cy, gz = ray.o.z - cz;
            const float hb = gx * ray.d.x + gy * ray.d.y + gz * ray.d.z;
            const float oo = gx * gx + gy * gy + gz * gz;
            const float h2 = hb * hb;
            // clearly missed:  h2 - (oo - r2) < -1e-5 (h2 + oo),  as two FMAs:  (1 + 1e-5) h2 + r2 < (1 - 1e-5) oo
            // (a NaN -- the stale ray of a lane without a path never gets here -- would make the slot a candidate: safe)
            return !(h2 * 1.00001f + r2 < oo * 0.99999f);
        };
        cand = (slot(X.x, Y.x, Z.x, R.x, VX.x, VY.x, VZ.x) ? 1u : 0u) | (slot(X.y, Y.y, Z.y, R.y, VX.y, VY.y, VZ.y) ? 2u : 0u) |
               (slot(X.z, Y.z, Z.z, R.z, VX.z, VY.z, VZ.z) ? 4u : 0u) | (slot(X.w, Y.w, Z.w, R.w, VX.w, VY.w, VZ.w) ? 8u : 0u);
    }
    for (;;)
    {
        if (pt_ballot(cand != 0u) == 0ull) break;
#ifdef PT_STAMPS
        ++q.leafSlotIters; ++q.leafExactIters;
#endif
        if (cand != 0u)
        {
            const uint32_t k = (uint32_t)__builtin_ctz(cand);
            cand &= cand - 1u;
            const float* blk = reinterpret_cast<const float*>(b.sph + 4 * leaf) + k;
            V3 c = v3(blk[0], blk[4], blk[8]);
            const float r2 = blk[12];
            if (MOVING) { const float* vb = reinterpret_cast<const float*>(b.vel + 3 * leaf) + k; c = add(c, mul(v3(vb[0], vb[4], vb[8]), time)); } // main.cpp:66
            const float t = intersect_sphere_exact(ray, c, r2);
            // an exact tie between two spheres goes to the lower original index; the indices are fetched only then
            // (a tie with the plane hit, bestSlot < 0, leaves the plane in place).  The update itself is three selects:
            // as a branch the compiler carried (best, cull, bestSlot) through the loop in copies, six more moves per trip
            const uint32_t slot = 4u * leaf + k;
            bool take = t > kMinT && t < best;
            if (t > kMinT && t == best && bestSlot >= 0)
                take = __float_as_uint(b.mat[2 * slot].w) < __float_as_uint(b.mat[2 * (uint32_t)bestSlot].w);
            best = take ? t : best; cull = take ? t * kCullSlack : cull; bestSlot = take ? (int32_t)slot : bestSlot;
        }
    }
}
// Scenes of a handful of spheres (MODE_FEW): no tree, no traversal state -- every ray meets every block.
template <bool MOVING>
PT_DEV void few_query(const BvhView& b, uint32_t nblocks, const Ray& ray, float time, bool live, QueryCtx& q, float& best, int32_t& bestSlot)
{
    float cull = best;
#pragma nounroll
    for (uint32_t blk = 0; blk < nblocks; ++blk) block_test<MOVING>(b, blk, live, ray, time, q, best, cull, bestSlot);
}

// Advances the traversals of the wave.  exitRule = 0: until every lane is done.  Else (pack_exit_rule: minActive |
// doneMul << 8) the run is also left as soon as fewer than minActive lanes are still traversing AND the live lanes that
// have finished number more than 4 / doneMul of them (doneMul 4: more finished than traversing) -- the
// stragglers keep their state (ts, LDS stack) and carry on in the next call, together with the rays the finished lanes
// will have produced by then.  (Measured on book-1: 33 % of the inner steps and 44 % of the leaf steps of a
// run-to-completion traversal are executed for fewer than 8 lanes.)
constexpr uint32_t pack_exit_rule(uint32_t minActive, uint32_t doneMul) { return minActive ? (minActive | (doneMul << 8)) : 0u; }
template <bool MOVING, bool CONE, bool QN, bool FLAT_EXACT, bool BLK = false>
PT_DEV void bvh_run(const BvhView& b, const Ray& ray, float time, bool live, QueryCtx& q, TravState& ts, uint32_t exitRule)
{
    // stack entries: 32-bit references, or (BLK) their low halves -- a pop sign-extends (see above)
    using StackT = typename std::conditional<BLK, int16_t, int32_t>::type;
    StackT* const sbase = BLK ? reinterpret_cast<StackT*>(q.stack16) : reinterpret_cast<StackT*>(q.stack);
    const uint32_t minActive = exitRule & 255u, doneMul = exitRule >> 8; // (see pack_exit_rule)
    const uint32_t leafBatch = q.leafBatch;
#ifdef PT_STAMPS
    unsigned long long qs0 = 0, qs1 = 0, qs2 = 0;
#endif
    float best = ts.best;
    // Boxes are culled against `cull`, not `best`: the reference's sphere formula (geometry.cpp:45-66) forms b*b - 4c,
    // which cancels down from |o-c|^2 -- its t is off the geometric one by up to ~sqrt(45 * 2^-24) * |o-c| = 8e-4 |o-c|
    // (0.25 units a thousand units out).  A brute-force search in that arithmetic -- the definition of the query -- can
    // therefore prefer a sphere that is geometrically just BEHIND the current best; its box must not be culled.
    float cull = best * kCullSlack;
    int32_t bestSlot = ts.bestSlot;
    // finite reciprocals: a zero direction component must not turn (lo - o) * inv into NaN
    const float big = 1.0e18f;
    // (v_rcp_f32, 1 ulp: the same relative error on both planes of an axis, far inside the builder's padding)
    float ix = fabsf(ray.d.x) > 1e-18f ? __builtin_amdgcn_rcpf(ray.d.x) : (ray.d.x < 0 ? -big : big);
    float iy = fabsf(ray.d.y) > 1e-18f ? __builtin_amdgcn_rcpf(ray.d.y) : (ray.d.y < 0 ? -big : big);
    float iz = fabsf(ray.d.z) > 1e-18f ? __builtin_amdgcn_rcpf(ray.d.z) : (ray.d.z < 0 ? -big : big);
    // The slab test as lo*inv + (-o*inv): one fma per plane.  Its rounding error, mapped back to position space, is a
    // few ulp of max(|o|, |lo|) -- the builder pads every box by 16 ulp of the scene extent (camera included).
    float nox = -ray.o.x * ix, noy = -ray.o.y * iy, noz = -ray.o.z * iz;
    // cone form (box_hit_cone): (ix, nox) become the hi-plane pair (r1, n1), (r2, n2) is the lo-plane pair
    float r2x = 0, r2y = 0, r2z = 0, n2x = 0, n2y = 0, n2z = 0;
    if (CONE)
    {
        cone_axis<FLAT_EXACT>(ray.o.x, ray.d.x, ix, nox, r2x, n2x); cone_axis<FLAT_EXACT>(ray.o.y, ray.d.y, iy, noy, r2y, n2y); cone_axis<FLAT_EXACT>(ray.o.z, ray.d.z, iz, noz, r2z, n2z);
    }
    float sx = 0, sy = 0, sz = 0; // (FLAT_EXACT: the flat-axis selectors of cone_planes)
    if (CONE && FLAT_EXACT) { sx = cone_flat_select(ray.d.x); sy = cone_flat_select(ray.d.y); sz = cone_flat_select(ray.d.z); }
    if constexpr (QN) // fold the node grid into the per-plane pairs: (r, n) -> (step * r, org * r + n)
    {
        nox = b.gorg[0] * ix + nox; noy = b.gorg[1] * iy + noy; noz = b.gorg[2] * iz + noz;
        n2x = b.gorg[0] * r2x + n2x; n2y = b.gorg[1] * r2y + n2y; n2z = b.gorg[2] * r2z + n2z;
        ix *= b.gstep[0]; iy *= b.gstep[1]; iz *= b.gstep[2];
        r2x *= b.gstep[0]; r2y *= b.gstep[1]; r2z *= b.gstep[2];
    }
    // All lanes run the loops below together (their exits are wave-uniform), which keeps the two tallies in scalar registers.
    StackT* sp = sbase + (ts.pk & 255u) * 64u;
    uint32_t cur = ts.cur;
    bool leave = false;
    // Two kinds of step, and a wave pays for both whenever its lanes disagree.  Leaf steps are the dear ones (up to
    // four sphere tests, the exact evaluation among them), so lanes that reach a leaf WAIT there while the others keep
    // descending; the leaf code runs once enough lanes have gathered (or nobody is left on an inner node).
    for (;;)
    {
        // ---- inner steps
        PT_QSTAMP(qs0);
        for (;;)
        {
            const bool atInner = !(cur & kBvhLeafFlag);                  // (the sentinel carries the leaf flag)
            const uint32_t nInner = (uint32_t)__popcll(pt_ballot(atInner));
            const uint32_t nLeaf = (uint32_t)__popcll(pt_ballot(!atInner && cur != kBvhDone));
            if (nInner == 0u || nLeaf >= leafBatch) break;
            if (nInner + nLeaf < minActive && (uint32_t)__popcll(pt_ballot(live && cur == kBvhDone)) * doneMul > (nInner + nLeaf) * 4u) { leave = true; break; }
#ifdef PT_STAMPS
            ++q.innerIters;
#endif
            q.nodeVisits += nInner;                                      // wave-level tally
            if (atInner)
            {
                bool hitL, hitR; float nearL, nearR; uint32_t refL, refR;
                if constexpr (QN)
                {
                    const float4 qa = b.nodes[2 * cur], qb = b.nodes[2 * cur + 1];
                    const uint32_t lx = __float_as_uint(qa.x), ly = __float_as_uint(qa.y), lz = __float_as_uint(qa.z);
                    const uint32_t rx = __float_as_uint(qa.w), ry = __float_as_uint(qb.x), rz = __float_as_uint(qb.y);
                    hitL = cone_planes<FLAT_EXACT>((float)(lx & 0xFFFFu), (float)(lx >> 16), (float)(ly & 0xFFFFu), (float)(ly >> 16), (float)(lz & 0xFFFFu), (float)(lz >> 16),
                                                   ix, iy, iz, nox, noy, noz, r2x, r2y, r2z, n2x, n2y, n2z, sx, sy, sz, cull, &nearL);
                    hitR = cone_planes<FLAT_EXACT>((float)(rx & 0xFFFFu), (float)(rx >> 16), (float)(ry & 0xFFFFu), (float)(ry >> 16), (float)(rz & 0xFFFFu), (float)(rz >> 16),
                                                   ix, iy, iz, nox, noy, noz, r2x, r2y, r2z, n2x, n2y, n2z, sx, sy, sz, cull, &nearR);
                    refL = __float_as_uint(qb.z); refR = __float_as_uint(qb.w);
                }
                else
                {
                    const float4 n0 = b.nodes[4 * cur], n1 = b.nodes[4 * cur + 1], n2 = b.nodes[4 * cur + 2];
                    // the two references are half of the node's last 16 bytes: an 8-byte load (from HBM the compiler would fetch all 16)
                    const float2 n3 = *reinterpret_cast<const float2*>(b.nodes + 4 * cur + 3);
                    // left box {n0.xyz .. n0.w, n1.x, n1.y}, right box {n1.z, n1.w, n2.x .. n2.yzw}
                    hitL = CONE ? cone_planes<FLAT_EXACT>(n0.x, n0.w, n0.y, n1.x, n0.z, n1.y, ix, iy, iz, nox, noy, noz, r2x, r2y, r2z, n2x, n2y, n2z, sx, sy, sz, cull, &nearL)
                                : box_hit(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, ix, iy, iz, nox, noy, noz, cull, &nearL);
                    hitR = CONE ? cone_planes<FLAT_EXACT>(n1.z, n2.y, n1.w, n2.z, n2.x, n2.w, ix, iy, iz, nox, noy, noz, r2x, r2y, r2z, n2x, n2y, n2z, sx, sy, sz, cull, &nearR)
                                : box_hit(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, ix, iy, iz, nox, noy, noz, cull, &nearR);
                    refL = __float_as_uint(n3.x); refR = __float_as_uint(n3.y);
                }
                if (hitL && hitR)
                {
                    const bool leftFirst = nearL <= nearR;
                    sp += 64; *sp = (StackT)(leftFirst ? refR : refL);
                    cur = leftFirst ? refL : refR;
                }
                else if (hitL) cur = refL;
                else if (hitR) cur = refR;
                else { cur = (uint32_t)(int32_t)*sp; if (cur != kBvhDone) sp -= 64; }
            }
        }
        PT_QSTAMP(qs1);
#ifdef PT_STAMPS
        q.cycInner += qs1 - qs0;
#endif
        if (leave) break;
        {
            const unsigned long long act = pt_ballot(cur != kBvhDone);
            if (act == 0ull) break;
            if ((uint32_t)__popcll(act) < minActive && (uint32_t)__popcll(pt_ballot(live && cur == kBvhDone)) * doneMul > (uint32_t)__popcll(act) * 4u) break;
        }
#ifdef PT_STAMPS
        ++q.leafIters;
#endif
        // ---- leaf steps: every lane that waits at a leaf
        const bool atLeaf = (cur & kBvhLeafFlag) && cur != kBvhDone;
        if constexpr (BLK)
        {
            if (atLeaf) q.laneTests += ((cur >> 12) & 3u) + 1u;
            block_test<MOVING>(b, cur & 0x0FFFu, atLeaf, ray, time, q, best, cull, bestSlot);
        }
        else
        {
        const uint32_t first = cur & 0x0FFFFFFFu, count = atLeaf ? ((cur >> 28) & 7u) + 1u : 0u;
        for (uint32_t k = 0;; ++k)
        {
            const unsigned long long todo = pt_ballot(k < count);
            if (todo == 0ull) break;
            q.sphereTests += (uint32_t)__popcll(todo);                   // wave-level tally
#ifdef PT_STAMPS
            ++q.leafSlotIters;
            {
                bool ex = false;
                if (k < count) { const float4 sd0 = b.sph[first + k]; V3 c0 = v3(sd0.x, sd0.y, sd0.z); if (MOVING) { const float4 vd0 = b.vel[first + k]; c0 = add(c0, mul(v3(vd0.x, vd0.y, vd0.z), time)); } ex = !sphere_clearly_missed(ray, c0, sd0.w); }
                if (pt_ballot(ex) != 0ull) ++q.leafExactIters;
            }
#endif
            if (k < count)
            {
                const uint32_t slot = first + k;
                const float4 sd = b.sph[slot];
                V3 c = v3(sd.x, sd.y, sd.z);
                if (MOVING) { const float4 vd = b.vel[slot]; c = add(c, mul(v3(vd.x, vd.y, vd.z), time)); } // main.cpp:66
                // cheap conservative filter: most spheres of a visited leaf are missed; only a ray whose quarter
                // discriminant is not clearly negative gets the exact (correctly rounded sqrt / divide) evaluation
                if (!sphere_clearly_missed(ray, c, sd.w))
                {
                    const float t = intersect_sphere_exact(ray, c, sd.w);
                    if (t > kMinT && t <= best)
                    {
                        // an exact tie between two spheres goes to the lower original index; the indices are fetched
                        // only then (a tie with the plane hit, bestSlot < 0, leaves the plane in place)
                        bool take = t < best;
                        if (!take && bestSlot >= 0)
                            take = __float_as_uint(b.mat[2 * slot].w) < __float_as_uint(b.mat[2 * (uint32_t)bestSlot].w);
                        if (take) { best = t; cull = t * kCullSlack; bestSlot = (int32_t)slot; }
                    }
                }
            }
        }
        } // !BLK
        if (atLeaf) { cur = (uint32_t)(int32_t)*sp; if (cur != kBvhDone) sp -= 64; }
        PT_QSTAMP(qs2);
#ifdef PT_STAMPS
        q.cycLeaf += qs2 - qs1;
#endif
    }
    ts.cur = cur; ts.best = best; ts.bestSlot = bestSlot;
    ts.pk = (ts.pk & ~255u) | ((uint32_t)(sp - sbase) >> 6);
}
// ---- the same traversal over 4-wide grid nodes (MODE_BVH_Q4; host: pt_bvh.h widen_nodes) ------------------------------
// One 64-byte fetch decides up to four subtrees; the hit children are ordered near to far by one sorting network over
// keys {bits of max(near, 0) with the child number in the two low bits} (positive floats order like unsigned integers,
// a miss is all ones), the nearest is descended into, the others are pushed far first.  Up to three pushes per level
// would make a worst-case stack of 3 x depth + 1 entries -- LDS that most rays never touch and that costs resident
// waves -- so only the first `stackCap` levels live in LDS and deeper ones in a per-lane scratch array (q.ovf); the
// branch around that path is wave-uniform and all but never taken.
constexpr uint32_t kOvfEntries = 72;   // + the LDS levels: trees of up to (stackCap + 72 - 1) / 3 wide levels (host checks)
// (The LDS part is addressed as LDS and the overflow as private memory, explicitly: written through one generic pointer
// the two stores merge into a flat_store of a selected address, which waits on both memory pipes.)
typedef __attribute__((address_space(3))) uint32_t pt_lds_u32;
PT_DEV void wide_push(const QueryCtx& q, uint32_t& level, uint32_t ref)
{
    ++level;
    if (level < q.stackCap) ((pt_lds_u32*)q.stack)[level * 64u] = ref;
    else q.ovf[level - q.stackCap] = ref;
}
PT_DEV uint32_t wide_pop(const QueryCtx& q, uint32_t& level)
{
    uint32_t r;
    if (level < q.stackCap) r = ((pt_lds_u32*)q.stack)[level * 64u];
    else r = q.ovf[level - q.stackCap];
    if (r != kBvhDone) --level;           // (level 0 holds the sentinel and stays)
    return r;
}
template <bool MOVING>
PT_DEV void bvh_run_wide(const BvhView& b, const Ray& ray, float time, bool live, QueryCtx& q, TravState& ts, uint32_t exitRule)
{
    const uint32_t minActive = exitRule & 255u, doneMul = exitRule >> 8; // (see pack_exit_rule)
    const uint32_t leafBatch = q.leafBatch;
    float best = ts.best;
    float cull = best * kCullSlack;      // (see bvh_run: boxes are culled with slack, the leaf test decides)
    int32_t bestSlot = ts.bestSlot;
    // cone test per plane: hi planes with (r1, n1), lo planes with (r2, n2), the node grid folded in (see bvh_run)
    float r1x, r1y, r1z, n1x, n1y, n1z, r2x, r2y, r2z, n2x, n2y, n2z;
    {
        auto axis = [](float o, float d, float org, float step, float& r1, float& n1, float& r2, float& n2)
        {
            cone_axis<true>(o, d, r1, n1, r2, n2);
            n1 = org * r1 + n1; n2 = org * r2 + n2; r1 *= step; r2 *= step;   // plane = org + cell * step
        };
        axis(ray.o.x, ray.d.x, b.gorg[0], b.gstep[0], r1x, n1x, r2x, n2x);
        axis(ray.o.y, ray.d.y, b.gorg[1], b.gstep[1], r1y, n1y, r2y, n2y);
        axis(ray.o.z, ray.d.z, b.gorg[2], b.gstep[2], r1z, n1z, r2z, n2z);
    }
    const float sx = cone_flat_select(ray.d.x), sy = cone_flat_select(ray.d.y), sz = cone_flat_select(ray.d.z);
    uint32_t level = ts.pk & 255u;
    uint32_t cur = ts.cur;
    bool leave = false;
#ifdef PT_STAMPS
    unsigned long long qs0 = 0, qs1 = 0, qs2 = 0;
#endif
    for (;;)
    {
        // ---- inner steps
        PT_QSTAMP(qs0);
        for (;;)
        {
            const bool atInner = !(cur & kBvhLeafFlag);                  // (the sentinel carries the leaf flag)
            const uint32_t nInner = (uint32_t)__popcll(pt_ballot(atInner));
            const uint32_t nLeaf = (uint32_t)__popcll(pt_ballot(!atInner && cur != kBvhDone));
            if (nInner == 0u || nLeaf >= leafBatch) break;
            if (nInner + nLeaf < minActive && (uint32_t)__popcll(pt_ballot(live && cur == kBvhDone)) * doneMul > (nInner + nLeaf) * 4u) { leave = true; break; }
            q.nodeVisits += nInner;                                      // wave-level tally
#ifdef PT_STAMPS
            ++q.innerIters;
#endif
            if (atInner)
            {
                const float4 q0 = b.nodes[4 * cur], q1 = b.nodes[4 * cur + 1], q2 = b.nodes[4 * cur + 2], q3 = b.nodes[4 * cur + 3];
                const uint32_t cx[4] = {__float_as_uint(q0.x), __float_as_uint(q0.w), __float_as_uint(q1.z), __float_as_uint(q2.y)};
                const uint32_t cy[4] = {__float_as_uint(q0.y), __float_as_uint(q1.x), __float_as_uint(q1.w), __float_as_uint(q2.z)};
                const uint32_t cz[4] = {__float_as_uint(q0.z), __float_as_uint(q1.y), __float_as_uint(q2.x), __float_as_uint(q2.w)};
                const uint32_t ref[4] = {__float_as_uint(q3.x), __float_as_uint(q3.y), __float_as_uint(q3.z), __float_as_uint(q3.w)};
                uint32_t key[4];
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k)
                {
                    float tnear;
                    const bool hit = cone_planes<true>((float)(cx[k] & 0xFFFFu), (float)(cx[k] >> 16), (float)(cy[k] & 0xFFFFu), (float)(cy[k] >> 16),
                                                       (float)(cz[k] & 0xFFFFu), (float)(cz[k] >> 16), r1x, r1y, r1z, n1x, n1y, n1z, r2x, r2y, r2z, n2x, n2y, n2z, sx, sy, sz, cull, &tnear);
                    key[k] = hit ? ((__float_as_uint(fmaxf(tnear, 0.0f)) & ~3u) | k) : 0xFFFFFFFFu; // (an empty slot is a point box in the grid's corner, pt_bvh.h)
                }
                const uint32_t a0 = min(key[0], key[1]), a1 = max(key[0], key[1]), c0 = min(key[2], key[3]), c1 = max(key[2], key[3]);
                const uint32_t s0 = min(a0, c0), m0 = max(a0, c0), m1 = min(a1, c1), s3 = max(a1, c1);
                const uint32_t s1 = min(m0, m1), s2 = max(m0, m1);
                auto pick = [&](uint32_t k2) { const uint32_t lo = (k2 & 1u) ? ref[1] : ref[0], hi = (k2 & 1u) ? ref[3] : ref[2]; return (k2 & 2u) ? hi : lo; };
                if (s0 == 0xFFFFFFFFu) cur = wide_pop(q, level);
                else
                {
                    if (s3 != 0xFFFFFFFFu) wide_push(q, level, pick(s3));
                    if (s2 != 0xFFFFFFFFu) wide_push(q, level, pick(s2));
                    if (s1 != 0xFFFFFFFFu) wide_push(q, level, pick(s1));
                    cur = pick(s0);
                }
            }
        }
        PT_QSTAMP(qs1);
#ifdef PT_STAMPS
        q.cycInner += qs1 - qs0;
#endif
        if (leave) break;
        {
            const unsigned long long act = pt_ballot(cur != kBvhDone);
            if (act == 0ull) break;
            if ((uint32_t)__popcll(act) < minActive && (uint32_t)__popcll(pt_ballot(live && cur == kBvhDone)) * doneMul > (uint32_t)__popcll(act) * 4u) break;
        }
#ifdef PT_STAMPS
        ++q.leafIters;
#endif
        // ---- leaf steps: every lane that waits at a leaf (as bvh_run)
        const bool atLeaf = (cur & kBvhLeafFlag) && cur != kBvhDone;
        const uint32_t first = cur & 0x0FFFFFFFu, count = atLeaf ? ((cur >> 28) & 7u) + 1u : 0u;
        for (uint32_t k = 0;; ++k)
        {
            const unsigned long long todo = pt_ballot(k < count);
            if (todo == 0ull) break;
            q.sphereTests += (uint32_t)__popcll(todo);                   // wave-level tally
            if (k < count)
            {
                const uint32_t slot = first + k;
                const float4 sd = b.sph[slot];
                V3 c = v3(sd.x, sd.y, sd.z);
                if (MOVING) { const float4 vd = b.vel[slot]; c = add(c, mul(v3(vd.x, vd.y, vd.z), time)); } // main.cpp:66
                if (!sphere_clearly_missed(ray, c, sd.w))
                {
                    const float t = intersect_sphere_exact(ray, c, sd.w);
                    if (t > kMinT && t <= best)
                    {
                        bool take = t < best;
                        if (!take && bestSlot >= 0)
                            take = __float_as_uint(b.mat[2 * slot].w) < __float_as_uint(b.mat[2 * (uint32_t)bestSlot].w);
                        if (take) { best = t; cull = t * kCullSlack; bestSlot = (int32_t)slot; }
                    }
                }
            }
        }
        if (atLeaf) cur = wide_pop(q, level);
        PT_QSTAMP(qs2);
#ifdef PT_STAMPS
        q.cycLeaf += qs2 - qs1;
#endif
    }
    ts.cur = cur; ts.best = best; ts.bestSlot = bestSlot;
    ts.pk = (ts.pk & ~255u) | level;
}
// The winner's centre, as the leaf test saw it.
template <bool MOVING, bool BLK = false> PT_DEV V3 trav_centre(const BvhView& b, int32_t slot, float time)
{
    V3 centre;   // (slot < 0: undefined, and make_hit does not read it)
    if (BLK)
    {
        if (slot >= 0)   // slot = 4 x leaf + position: component `position` of the leaf's block
        {
            const float* blk = reinterpret_cast<const float*>(b.sph + ((uint32_t)slot & ~3u)) + ((uint32_t)slot & 3u);
            centre = v3(blk[0], blk[4], blk[8]);
            if (MOVING) { const float* vb = reinterpret_cast<const float*>(b.vel + 3u * ((uint32_t)slot >> 2)) + ((uint32_t)slot & 3u); centre = add(centre, mul(v3(vb[0], vb[4], vb[8]), time)); }
        }
        return centre;
    }
    if (slot >= 0)
    {
        const float4 sd = b.sph[(uint32_t)slot];
        centre = v3(sd.x, sd.y, sd.z);
        if (MOVING) { const float4 vd = b.vel[(uint32_t)slot]; centre = add(centre, mul(v3(vd.x, vd.y, vd.z), time)); }
    }
    return centre;
}

struct Hit { float t; int32_t prim; V3 point, normal; Material m; };

// How the sphere query is answered.
enum : int { MODE_LDS_FAST = 0, MODE_LDS_STRICT = 1, MODE_BVH = 2, MODE_BVH_LDS = 3, MODE_BVH_Q = 4, MODE_REF = 5, MODE_BVH_Q4 = 6, MODE_FEW = 7 };
constexpr bool mode_is_bvh(int m) { return m == MODE_BVH || m == MODE_BVH_LDS || m == MODE_BVH_Q || m == MODE_BVH_Q4; }
constexpr bool mode_in_hbm(int m) { return m == MODE_BVH || m == MODE_BVH_Q || m == MODE_BVH_Q4; } // tree read from HBM; BVH_LDS: tree staged in LDS
// Which traversals test boxes against a cone around the ray (box_hit_cone): all of them.  The cone is what makes the
// tree agree with the brute-force definition of the query wherever the reference's sphere formula reports phantom hits
// (far from the ray origin: big scenes, and rays that come back from a distant point of an infinite plane).
#ifdef PT_LDS_PLAIN_BOXES // experiments only: the tree-in-LDS kernel with plain boxes (round 1's state)
constexpr bool mode_uses_cone(int m) { return mode_in_hbm(m); }
#else
constexpr bool mode_uses_cone(int m) { return mode_is_bvh(m); }
#endif

// Which traversal answers MODE.
template <bool MOVING, int MODE>
PT_DEV void run_traversal(const BvhView& b, const Ray& ray, float time, bool live, QueryCtx& q, TravState& ts, uint32_t minActive)
{
    if constexpr (MODE == MODE_BVH_Q4) bvh_run_wide<MOVING>(b, ray, time, live, q, ts, minActive);
    else bvh_run<MOVING, mode_uses_cone(MODE), MODE == MODE_BVH_Q, mode_in_hbm(MODE), MODE == MODE_BVH_LDS>(b, ray, time, live, q, ts, minActive);
}


// ------------------------------------------------------------------ the reference's own query (PT_FLAG_REFERENCE_QUERY)
// hit_test(Ray, Rect3f) (geometry.cpp:83-115) in the reference's arithmetic: here the box test decides the RESULT (a
// subtree whose box is missed contributes "no hit", and the pairwise combine below turns that into discarded hits), so
// it is restated operation by operation, contraction off, correctly rounded divides, `<`-ternary MIN / MAX (types.h:12-13)
// with their NaN behaviour (a zero direction component gives 0 * inf).
PT_DEV bool ref_hit_test(const Ray& ray, float px, float py, float pz, float hw, float hh, float hl)
{
    float inverseDirX = 1.0f / ray.d.x;
    float tx0 = ((px - hw) - ray.o.x) * inverseDirX;
    float tx1 = ((px + hw) - ray.o.x) * inverseDirX;
    if (tx0 > tx1) { float c = tx0; tx0 = tx1; tx1 = c; }
    float inverseDirY = 1.0f / ray.d.y;
    float ty0 = ((py - hh) - ray.o.y) * inverseDirY;
    float ty1 = ((py + hh) - ray.o.y) * inverseDirY;
    if (ty0 > ty1) { float c = ty0; ty0 = ty1; ty1 = c; }
    float inverseDirZ = 1.0f / ray.d.z;
    float tz0 = ((pz - hl) - ray.o.z) * inverseDirZ;
    float tz1 = ((pz + hl) - ray.o.z) * inverseDirZ;
    if (tz0 > tz1) { float c = tz0; tz0 = tz1; tz1 = c; }
    const float mxy = tx0 < ty0 ? ty0 : tx0, tMin = mxy < tz0 ? tz0 : mxy; // MAX_VALUE(MAX_VALUE(tx0, ty0), tz0)
    const float nxy = tx1 < ty1 ? tx1 : ty1, tMax = nxy < tz1 ? nxy : tz1; // MIN_VALUE(MIN_VALUE(tx1, ty1), tz1)
    return !(tMin >= tMax);
}
// intersection_test(Ray, BVH*, time, &obj) (main.cpp:53-91) without the recursion: every lane walks the reference's
// call tree with an explicit stack of the calls that wait for a child (two dwords per level in LDS, lane-strided:
// {node | second child under way << 31, t of the left child} and the left child's sphere slot + 1).  A leaf returns the
// RAW near-root t of its sphere (no box test, may be <= 0.001 or negative); an inner node whose box is hit asks BOTH
// children, always, and combines: left iff tLeft < tRight && tLeft > MIN_T, else right iff tRight < tLeft && tRight >
// MIN_T, else no hit -- so a child that returns a small invalid t hides its sibling's valid one (the quirk).
// Returns t and the winner's sphere slot (-1: none).  Stack: 3 dwords x (tree depth + 1) per lane.
template <bool MOVING> PT_DEV float ref_query(const QueryCtx& qc, const Ray& ray, float time, bool live, int32_t* slotOut, uint32_t* boxTests, uint32_t* sphereTests)
{
    const RefView& rv = qc.ref;
    uint32_t* const stack = qc.stack;
    uint32_t level = 0;                       // calls waiting for a child
    uint32_t enter = rv.root;                 // node to call next; kRefLeaf = a call has just returned (tRet, sRet)
    float tRet = kF32Max; int32_t sRet = -1;
    bool done = !live || rv.n_nodes == 0;
    for (;;)
    {
        if (pt_ballot(!done) == 0ull) break;
        if (!done)
        {
            if (enter != kRefLeaf)
            {
                const float4 n1 = rv.nodes[2 * enter + 1];
                const uint32_t left = __float_as_uint(n1.z), right = __float_as_uint(n1.w);
                if (right == kRefLeaf)                                    // main.cpp:61-69
                {
                    const float4 sd = qc.bvh.sph[left];
                    V3 c = v3(sd.x, sd.y, sd.z);
                    if (MOVING) { const float4 vd = qc.bvh.vel[left]; c = add(c, mul(v3(vd.x, vd.y, vd.z), time)); } // main.cpp:66
                    tRet = intersect_sphere_exact(ray, c, sd.w);
                    sRet = (int32_t)left;
                    enter = kRefLeaf;
                    *sphereTests += 1u;
                }
                else
                {
                    const float4 n0 = rv.nodes[2 * enter];
                    *boxTests += 1u;
                    if (ref_hit_test(ray, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y))   // main.cpp:70
                    {
                        stack[(3u * level) * 64u] = enter;                 // first child under way
                        ++level;
                        enter = left;                                       // main.cpp:75
                    }
                    else { tRet = kF32Max; sRet = -1; enter = kRefLeaf; }
                }
            }
            else if (level == 0u)
                done = true;
            else
            {
                uint32_t* f = stack + (3u * (level - 1u)) * 64u;
                const uint32_t head = f[0];
                if (!(head & 0x80000000u))                                 // the left child has returned: ask the right one (main.cpp:76)
                {
                    f[0] = head | 0x80000000u; f[64] = __float_as_uint(tRet); f[128] = (uint32_t)(sRet + 1);
                    enter = __float_as_uint(rv.nodes[2 * head + 1].w);
                }
                else                                                        // both are back: main.cpp:78-87
                {
                    const float tLeft = __uint_as_float(f[64]), tRight = tRet;
                    const int32_t sLeft = (int32_t)f[128] - 1;
                    if (tLeft < tRight && tLeft > kMinT) { tRet = tLeft; sRet = sLeft; }
                    else if (tRight < tLeft && tRight > kMinT) { /* tRet, sRet stay the right child's */ }
                    else { tRet = kF32Max; sRet = -1; }
                    --level;
                }
            }
        }
    }
    *slotOut = sRet;
    return tRet;
}

// First half of cast_ray (main.cpp:180-216): planes linearly, then the sphere query; closest wins, planes win ties.
// plane loop (main.cpp:186-200): closest plane hit, or kF32Max / -1
PT_DEV float closest_plane(const SceneView& s, const Ray& ray, int32_t* planeIdxOut)
{
    float tBest = kF32Max;
    int32_t planeIdx = -1;
    for (uint32_t i = 0; i < s.np; ++i)
    {
        float4 p = s.planes[i];
        float t = intersect_plane_exact(ray, v3(p.x, p.y, p.z), p.w);
        if (t > kMinT && t < tBest) { tBest = t; planeIdx = (int32_t)i; }
    }
    *planeIdxOut = planeIdx;
    return tBest;
}
// the hit record (main.cpp:204-216) from the winners of the two searches: sphere `obj` at t (-1: none) and plane planeIdx at planeT
// On a miss (prim == kPrimMiss; t = planeT, the caller's limit) point, normal and material are left as they come:
// shade_step reads none of them (main.cpp:219 looks at t alone), and the probe kernel blanks them for its records.
// (Zeroed "just in case" they were thirteen moves per shading phase.)
PT_DEV Hit make_hit(const SceneView& s, const Ray& ray, int32_t obj, float t, V3 centre, const Material& sphereMat, uint32_t origIndex, int32_t planeIdx, float planeT)
{
    Hit h;
    const bool sphere = obj >= 0 && t > kMinT && t < planeT;
    h.t = sphere ? t : planeT;
    h.prim = kPrimMiss;
    h.point = ray_at(ray, h.t);
    if (sphere)
    {
        h.prim = (int32_t)origIndex;
        h.normal = normalize(sub(h.point, centre));
        h.m = sphereMat;
    }
    else if (planeIdx >= 0)
    {
        float4 p = s.planes[planeIdx];
        h.prim = -1 - planeIdx;
        h.normal = v3(p.x, p.y, p.z);
        h.m = load_material(s.pmat, (uint32_t)planeIdx);
    }
    return h;
}
template <bool MOVING, bool BLK = false> PT_DEV Hit make_hit_bvh(const SceneView& s, const Ray& ray, float time, const QueryCtx& q, const TravState& ts)
{
    const int32_t planeIdx = trav_plane(ts.pk);
    const int32_t obj = ts.bestSlot;
    Material sphereMat; uint32_t origIndex = 0;   // (read by make_hit only with obj >= 0)
    if (obj >= 0) { sphereMat = load_material(q.bvh.mat, (uint32_t)obj); origIndex = __float_as_uint(q.bvh.mat[2 * (uint32_t)obj].w); }
    // a sphere is recorded only when its t is below the plane's (ties stay with the plane), so with obj >= 0 the plane's
    // t is anything above ts.best, and without it ts.best still is the plane's t
    return make_hit(s, ray, obj, ts.best, trav_centre<MOVING, BLK>(q.bvh, obj, time), sphereMat, origIndex, planeIdx, obj >= 0 ? kF32Max : ts.best);
}
template <bool MOVING, int MODE> PT_DEV Hit closest_hit(const SceneView& s, const Ray& ray, float time, bool live, QueryCtx& q)
{
    constexpr bool STRICT = MODE == MODE_LDS_STRICT;
    uint32_t* searched = &q.searched;
    int32_t planeIdx = -1;
    const float planeT = closest_plane(s, ray, &planeIdx);
    if (mode_is_bvh(MODE))
    {
        TravState ts;
        trav_begin<MODE == MODE_BVH_LDS>(q.bvh, planeT, planeIdx, live, q, ts);
        run_traversal<MOVING, MODE>(q.bvh, ray, time, live, q, ts, 0u);
        return make_hit_bvh<MOVING, MODE == MODE_BVH_LDS>(s, ray, time, q, ts);
    }
    if (MODE == MODE_FEW)
    {
        // a handful of spheres as leaf blocks in LDS (n_spheres = 4 x blocks here): every ray meets every block; a sphere
        // is recorded only when its t is below the plane's (ties stay with the plane), as in the traversals
        float best = planeT; int32_t slot = -1;
        few_query<MOVING>(q.bvh, q.bvh.n_spheres >> 2, ray, time, live, q, best, slot);
        Material m; m.type = MAT_NONE; m.colour = v4(0, 0, 0, 0); m.roughness = 0; m.n = 0;
        uint32_t orig = 0;
        if (slot >= 0) { m = load_material(q.bvh.mat, (uint32_t)slot); orig = __float_as_uint(q.bvh.mat[2 * (uint32_t)slot].w); }
        return make_hit(s, ray, slot, best, trav_centre<MOVING, true>(q.bvh, slot, time), m, orig, planeIdx, slot >= 0 ? kF32Max : best);
    }
    if (MODE == MODE_REF)
    {
        // main.cpp:203-216 on the reference's tree: t must still beat the closest plane, strictly
        int32_t slot = -1;
        uint32_t boxTests = 0, sphereTests = 0;
        const float t = ref_query<MOVING>(q, ray, time, live, &slot, &boxTests, &sphereTests);
        for (int o = 32; o > 0; o >>= 1) { boxTests += __shfl_xor(boxTests, o); sphereTests += __shfl_xor(sphereTests, o); }
        q.nodeVisits += boxTests; q.sphereTests += sphereTests;
        Material m; m.type = MAT_NONE; m.colour = v4(0, 0, 0, 0); m.roughness = 0; m.n = 0;
        uint32_t orig = 0;
        if (slot >= 0) { m = load_material(q.bvh.mat, (uint32_t)slot); orig = __float_as_uint(q.bvh.mat[2 * (uint32_t)slot].w); }
        return make_hit(s, ray, slot, t, trav_centre<MOVING>(q.bvh, slot, time), m, orig, planeIdx, planeT);
    }
    int32_t obj;
    float t = kF32Max;
    V3 centre = v3(0, 0, 0);
    Material sphereMat; sphereMat.type = MAT_NONE; sphereMat.colour = v4(0, 0, 0, 0); sphereMat.roughness = 0; sphereMat.n = 0;
    uint32_t origIndex = 0;
    if (STRICT)
    {
        obj = search_spheres_strict<MOVING>(s, ray, time, &t);
        if (obj >= 0)
        {
            float r2; load_sphere(s, (uint32_t)obj, &centre, &r2);
            if (MOVING) centre = add(centre, mul(load_velocity(s, (uint32_t)obj), time));
        }
    }
    else
    {
        // The winner (or the winning pair) is re-evaluated in reference arithmetic; lower index wins an exact tie.
        if (s.npairs <= 256u)
        {
            obj = search_spheres_fast<MOVING, false>(s, ray, time, live, searched);
            if (obj >= 0)
            {
                V3 c0, c1; float r20, r21;
                load_sphere(s, (uint32_t)obj, &c0, &r20);
                load_sphere(s, (uint32_t)obj + 1u, &c1, &r21);       // an unused slot has r2 = -1: never hit
                if (MOVING)
                {
                    c0 = add(c0, mul(load_velocity(s, (uint32_t)obj), time));
                    c1 = add(c1, mul(load_velocity(s, (uint32_t)obj + 1u), time));
                }
                const float t0 = intersect_sphere_exact(ray, c0, r20), t1 = intersect_sphere_exact(ray, c1, r21);
                const bool v0 = t0 > kMinT, v1 = t1 > kMinT;
                const bool second = v1 && (!v0 || t1 < t0 ||
                                           (t1 == t0 && original_index(s, (uint32_t)obj + 1u) < original_index(s, (uint32_t)obj)));
                t = second ? t1 : t0;
                centre = second ? c1 : c0;
                obj = second ? obj + 1 : obj;
            }
        }
        else
        {
            obj = search_spheres_fast<MOVING, true>(s, ray, time, live, searched);
            if (obj >= 0)
            {
                float r2; load_sphere(s, (uint32_t)obj, &centre, &r2);
                if (MOVING) centre = add(centre, mul(load_velocity(s, (uint32_t)obj), time));
                t = intersect_sphere_exact(ray, centre, r2);
            }
        }
    }
    if (obj >= 0) { sphereMat = load_material(s.smat, (uint32_t)obj); origIndex = original_index(s, (uint32_t)obj); }
    return make_hit(s, ray, obj, t, centre, sphereMat, origIndex, planeIdx, planeT);
}

// reflectance (main.cpp:161-168): the reference's f64 island.  pow(x,5) is evaluated as x*x*x*x*x here; it feeds
// only the comparison against a 53-bit uniform, so a last-bit difference flips a decision with probability ~1e-16.
PT_DEV double reflectance(double cosine, double refractRatio)
{
    double r0 = (1.0 - refractRatio) / (1.0 + refractRatio);
    r0 = r0 * r0;
    double x = 1.0 - cosine, x2 = x * x;
    return r0 + (1.0 - r0) * (x2 * x2 * x);
}

// Second half of cast_ray (main.cpp:219-298).  Returns false when the path ends at this hit.
// maxAttempts > 0: the unit-sphere rejection loop gives up after that many Philox blocks (philox_sphere_block: up to three
// points each) and the event is EV_PENDING (nothing but the RNG has changed; the caller calls again for the same hit and
// the loop goes on where it stopped, on a block boundary, where every scatter starts anyway).
enum : uint32_t { EV_PENDING = 5 };
template <class R> PT_DEV bool scatter(const Ray& ray, const Hit& h, R& rng, Ray* out, uint32_t* event, uint32_t maxAttempts = 0)
{
    *event = EV_SCATTER;
    out->o = h.point;
    rng.align();
    rng.prime();
    // The dielectric's single f64 draw (main.cpp:276) is taken here, right behind align(): its two words are then
    // words 0,1 of a fresh block, one Philox evaluation, known at compile time.
    const double xi = (h.m.type == MAT_DIALECTRIC) ? rng.f64() : 0.0;
    // Diffuse and rough metal both start from random_point_in_unit_sphere (utils.h:80-94): one rejection loop for
    // both, so a wave pays max(attempts) once, not once per material branch.  Per lane the draws are unchanged.
    const bool roughMetal = h.m.type == MAT_METAL && h.m.roughness > 0.0f;
    V3 unitPoint;   // (set and read for diffuse and rough metal only)
    bool found = true;
    if (h.m.type == MAT_DIFFUSE || roughMetal) unitPoint = random_point_in_unit_sphere_bounded(rng, maxAttempts, &found);
    // Every material ends in `normalize(new direction)` (main.cpp:231, 251, 293): the branches below only pick the raw
    // direction, the normalisation -- an exact sqrt and three exact divides -- is issued once for the whole wave.
    V3 raw;         // (every material that returns below sets it)
    bool lives = true;
    if (h.m.type == MAT_DIFFUSE)                                         // main.cpp:223-234
    {
        raw = add(normalize(unitPoint), h.normal);                       // random_unit_vector, utils.h:102-108
        if (near_zero(add(raw, h.normal))) raw = h.normal;               // sic (main.cpp:226)
    }
    else if (h.m.type == MAT_METAL)                                      // main.cpp:235-259
    {
        raw = reflect_direction(ray.d, h.normal);
        if (roughMetal)
        {
            V3 centre = add(h.point, raw);
            V3 randomPoint = add(mul(unitPoint, h.m.roughness), centre); // utils.h:96-100
            raw = sub(randomPoint, h.point);
        }
        if (!(dot(raw, h.normal) > 0)) { *event = EV_ABSORB; lives = false; }
    }
    else if (h.m.type == MAT_DIALECTRIC)                                 // main.cpp:260-297
    {
        float refractRatio = 1.0f / h.m.n;
        if (dot(ray.d, h.normal) > 0.0f) refractRatio = 1.0f / refractRatio;
        float cosTheta = dot(neg(ray.d), h.normal);
        // refractRatio * sqrtf(1 - cos^2) > 1.0f (main.cpp:272-273) can only hold for a ratio above 1: the root of a value
        // <= 1 is <= 1 (a negative argument gives NaN, which compares false), and the rounded product of two floats <= 1
        // is <= 1 -- so the correctly rounded root (~17 instructions) is only formed for lanes that leave the denser
        // medium (in the reference's scenes: hits on the glass floor of scene 2 from below)
        bool internalReflection = false;
        if (refractRatio > 1.0f) internalReflection = refractRatio * sqrtf(1 - cosTheta * cosTheta) > 1.0f;
        bool shouldReflect = reflectance((double)cosTheta, (double)refractRatio) > xi;
        if (internalReflection || shouldReflect)
            raw = reflect_direction(ray.d, h.normal);
        else
        {
            V3 rayPerpendicular = mul(add(ray.d, mul(h.normal, cosTheta)), refractRatio);
            float d = 1 - norm_squared(rayPerpendicular);
            float ad = d < 0 ? -d : d;
            V3 rayParallel = mul(h.normal, -sqrtf(ad));
            raw = add(rayPerpendicular, rayParallel);
        }
    }
    else
    {
        *event = EV_NONE_MAT;
        out->d = v3(0, 0, 0);
        return false;
    }
    out->d = normalize(raw);
    if (!found) { *event = EV_PENDING; return true; }
    return lives;
}

PT_DEV V4 sky(const Ray& ray)                                            // main.cpp:299-304
{
    float ratio = 0.5f * (ray.d.y + 1.0f);
    float a = 1.0f - ratio;
    return v4(1.0f * a + 0.7f * ratio, 1.0f * a + 0.8f * ratio, 1.0f * a + 0.9f * ratio, 1.0f * a + 1.0f * ratio);
}

// One loop iteration of cast_ray.  Returns true while the path lives; when it ends, *terminal is the colour the
// throughput must be multiplied with (sky, BLACK (0,0,0,1), or (0,0,0,0) for a NONE material).
// pending != nullptr: bounded rejection loop (scatter's maxAttempts = kScatterAttempts); *pending = true means the hit
// is not shaded yet -- ray, throughput and *terminal untouched, call again with the same hit.
#ifndef PT_SCATTER_ATTEMPTS
#define PT_SCATTER_ATTEMPTS 2 // Philox blocks per trip: a block fails for 11 % of the lanes, two for 1.2 % (one block: -1 %)
#endif
constexpr uint32_t kScatterAttempts = PT_SCATTER_ATTEMPTS;
template <class R>
PT_DEV bool shade_step(Ray& ray, const Hit& h, V4& throughput, R& rng, V4* terminal, bool* pending = nullptr)
{
    if (pending) *pending = false;
    if (h.t != kF32Max && h.t > 0)                                       // main.cpp:219
    {
        Ray next; uint32_t ev;
        bool alive = scatter(ray, h, rng, &next, &ev, pending ? kScatterAttempts : 0u);
        if (ev == EV_PENDING) { *pending = true; return true; }
        if (!alive)
        {
            *terminal = (ev == EV_ABSORB) ? v4(0, 0, 0, 1) : v4(0, 0, 0, 0);
            return false;
        }
        throughput = hadamard(throughput, h.m.colour);                   // main.cpp:233,255,296 (loop form)
        ray = next;
        return true;
    }
    *terminal = sky(ray);
    return false;
}
template <bool MOVING, int MODE, class R>
PT_DEV bool path_step(const SceneView& s, QueryCtx& q, Ray& ray, float time, V4& throughput, R& rng, V4* terminal)
{
    const Hit h = closest_hit<MOVING, MODE>(s, ray, time, true, q);
    return shade_step(ray, h, throughput, rng, terminal);
}

// cast_ray (main.cpp:171-307) as a loop, for the probes.
template <bool MOVING, int MODE, class R>
PT_DEV V4 cast_ray(const SceneView& s, QueryCtx& q, Ray ray, uint32_t maxDepth, float time, R& rng, uint32_t* segments)
{
    V4 T = v4(1, 1, 1, 1), terminal = v4(0, 0, 0, 1);
    uint32_t n = 0;
    for (uint32_t depth = maxDepth; depth != 0; --depth)                 // depth 0 -> BLACK (main.cpp:175-176)
    {
        ++n;
        if (!path_step<MOVING, MODE>(s, q, ray, time, T, rng, &terminal)) break;
        terminal = v4(0, 0, 0, 1);
    }
    *segments = n;
    return hadamard(T, terminal);
}

// The head of one sample (main.cpp:349-354): time, jittered (u,v), primary ray.
template <class R>
PT_DEV Ray begin_sample(const DeviceCamera& cam, uint32_t px, uint32_t py, uint32_t w, uint32_t h, float t0, float t1, R& rng, float* time)
{
    // main.cpp:349 draws the time even for a still scene and scales the draw by t1 - t0 = 0.  On the Philox stream (this
    // build's own) a still scene draws nothing here: r * 0 + t0 = t0 whatever r, and (u, v, first lens point) fill one
    // Philox block exactly.  The tape-driven probes keep the draw.
    if (R::kStillSkipsTime && t0 == t1) *time = t0;
    else *time = rand_range(rng, t0, t1);                                // main.cpp:349
    float u = ((float)px + rng.f32()) / (float)w;                        // main.cpp:351
    float v = ((float)py - rng.f32()) / (float)h;                        // main.cpp:352
    return get_ray(cam, u, v, rng);                                      // main.cpp:354
}

} // namespace ptd
