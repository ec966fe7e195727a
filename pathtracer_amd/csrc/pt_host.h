// pt_host.h -- host-side preparation of what the kernels read: the LDS image of the scene and the folded camera.
// Host arithmetic here continues into device arithmetic (pt_device.h), so it keeps the reference's f32 operation
// order and is compiled without contraction (-ffp-contract=off; baseline x86-64 has no FMA anyway).
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/pt_hip.h"
#include "pt_device.h"

namespace pth {

struct H3 { float x, y, z; };
inline H3 h3(float x, float y, float z) { H3 r = {x, y, z}; return r; }
inline H3 add(H3 a, H3 b) { return h3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline H3 sub(H3 a, H3 b) { return h3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline H3 mul(H3 v, float s) { return h3(v.x * s, v.y * s, v.z * s); }
inline H3 cross(H3 a, H3 b) { return h3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); } // vectors.cpp:137-142
inline float norm(H3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }                                     // vectors.cpp:107-110
inline H3 normalize(H3 v) { return mul(v, 1.0f / norm(v)); }                                                     // vectors.cpp:127-130

// The (u,v)-independent part of Camera::get_ray (camera.cpp:31-44, 59-69), evaluated once instead of per ray.
inline void make_device_camera(const float c[13], ptd::DeviceCamera* out)
{
    H3 pos = h3(c[0], c[1], c[2]), dir = h3(c[3], c[4], c[5]), up = h3(c[6], c[7], c[8]);
    float fov = c[9], aspect = c[10], focus = c[11], lensRadius = c[12];
    H3 planePos = add(pos, mul(dir, focus));                                   // camera.cpp:40-44
    float height = 2.0f * std::tan(fov / 2.0f);                                // camera.cpp:34 (float tan)
    float width = height * aspect;                                             // camera.cpp:35
    H3 horizontal = normalize(cross(dir, up));                                 // camera.cpp:64
    H3 vertical = up;                                                          // camera.cpp:65
    H3 topLeft = add(sub(planePos, mul(mul(horizontal, width / 2.0f), focus)), mul(mul(vertical, height / 2.0f), focus)); // camera.cpp:69
    out->pos[0] = pos.x; out->pos[1] = pos.y; out->pos[2] = pos.z;
    out->horizontal[0] = horizontal.x; out->horizontal[1] = horizontal.y; out->horizontal[2] = horizontal.z;
    out->vertical[0] = vertical.x; out->vertical[1] = vertical.y; out->vertical[2] = vertical.z;
    out->topLeft[0] = topLeft.x; out->topLeft[1] = topLeft.y; out->topLeft[2] = topLeft.z;
    out->planeW = width; out->planeH = height; out->focus = focus; out->lensRadius = lensRadius;
}

inline int validate_desc(const pt_scene_desc* d)
{
    if (d->n_spheres && (!d->sphere_pos_r || !d->sphere_mat_type || !d->sphere_mat_rgba || !d->sphere_mat_param)) return PT_ERR_INVALID;
    if (d->n_planes && (!d->plane_n_d || !d->plane_mat_type || !d->plane_mat_rgba || !d->plane_mat_param)) return PT_ERR_INVALID;
    if (d->n_planes > 64) return PT_ERR_UNSUPPORTED;   // World::planes[64], render_world.h:88
    if (d->n_spheres > (1u << 28)) return PT_ERR_UNSUPPORTED;
    if (!(d->t0 <= d->t1)) return PT_ERR_INVALID;      // random_f32(min,max) asserts min <= max, utils.h:49
    return PT_OK;
}

struct Blob
{
    std::vector<float4> f4;
    std::vector<uint32_t> order; // slot -> original sphere index (0xFFFFFFFF = unused slot)
    uint32_t off_velp = 0, off_groups = 0, off_smat = 0, off_planes = 0, off_pmat = 0, off_cam = 0, npairs = 0, ngroups = 0, idx_mask = 1;
    bool moving = false;
};

inline float4 f4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// Spatial order for the groups of 8: spheres of similar size first (a few huge spheres would otherwise blow up the
// bounds of the small ones they are mixed with), Morton order of the mid-time centre inside a size class.
inline void spatial_order(const pt_scene_desc* d, std::vector<uint32_t>* order)
{
    const uint32_t ns = d->n_spheres;
    order->resize(ns);
    if (!ns) return;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    const float tm = 0.5f * (d->t0 + d->t1);
    std::vector<float> mid(3 * (size_t)ns);
    for (uint32_t i = 0; i < ns; ++i)
        for (int k = 0; k < 3; ++k)
        {
            float c = d->sphere_pos_r[4 * (size_t)i + k] + (d->sphere_vel ? d->sphere_vel[3 * (size_t)i + k] * tm : 0.0f);
            mid[3 * (size_t)i + k] = c;
            if (c < lo[k]) lo[k] = c;
            if (c > hi[k]) hi[k] = c;
        }
    std::vector<uint64_t> key(ns);
    auto spread = [](uint64_t v) { // 21 bits -> every third bit
        v &= 0x1FFFFF; v = (v | v << 32) & 0x1F00000000FFFFull; v = (v | v << 16) & 0x1F0000FF0000FFull;
        v = (v | v << 8) & 0x100F00F00F00F00Full; v = (v | v << 4) & 0x10C30C30C30C30C3ull; v = (v | v << 2) & 0x1249249249249249ull;
        return v;
    };
    for (uint32_t i = 0; i < ns; ++i)
    {
        uint64_t m = 0;
        for (int k = 0; k < 3; ++k)
        {
            float ext = hi[k] - lo[k];
            float u = ext > 0 ? (mid[3 * (size_t)i + k] - lo[k]) / ext : 0.0f;
            uint64_t q = (uint64_t)(u * 2097151.0f);
            m |= spread(q) << k;
        }
        float r = d->sphere_pos_r[4 * (size_t)i + 3];
        int cls = 0;
        if (r > 0) { int e; frexpf(r, &e); cls = 64 - (e / 2); } // bigger spheres first, classes a factor 4 apart
        key[i] = ((uint64_t)(cls & 0x7F) << 57) ^ (m >> 7);
        (*order)[i] = i;
    }
    std::stable_sort(order->begin(), order->end(), [&key](uint32_t a, uint32_t b) { return key[a] < key[b]; });
}

// Layout: see ptd::SceneView.
inline void build_blob(const pt_scene_desc* d, Blob* b)
{
    const uint32_t ns = d->n_spheres, np = d->n_planes;
    b->moving = false;
    if (d->sphere_vel)
        for (uint32_t i = 0; i < 3 * ns; ++i)
            if (d->sphere_vel[i] != 0.0f) { b->moving = true; break; }
    b->ngroups = ((ns + 7) / 8 + 3) / 4 * 4; // whole batches of four bounds
    b->npairs = 4 * b->ngroups;
    const uint32_t slots = 2 * b->npairs;
    uint32_t bits = 1;
    while ((1u << bits) < slots) ++bits;
    b->idx_mask = (1u << bits) - 1;
    std::vector<uint32_t> sorted;
    spatial_order(d, &sorted);
    b->order.assign(slots, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < ns; ++i) b->order[i] = sorted[i];
    b->f4.clear();
    // spheres, two per entry.  r2 = radius*radius as the reference forms it (geometry.cpp:47).  An unused slot is a
    // sphere with r2 = -1: its discriminant is negative for every ray.
    for (uint32_t j = 0; j < b->npairs; ++j)
    {
        float cx[2] = {0, 0}, cy[2] = {0, 0}, cz[2] = {0, 0}, r2[2] = {-1.0f, -1.0f};
        for (uint32_t l = 0; l < 2; ++l)
        {
            uint32_t i = b->order[2 * j + l];
            if (i == 0xFFFFFFFFu) continue;
            const float* s = d->sphere_pos_r + 4 * (size_t)i;
            cx[l] = s[0]; cy[l] = s[1]; cz[l] = s[2]; r2[l] = s[3] * s[3];
        }
        b->f4.push_back(f4(cx[0], cx[1], cy[0], cy[1]));
        b->f4.push_back(f4(cz[0], cz[1], r2[0], r2[1]));
    }
    b->off_velp = (uint32_t)b->f4.size();
    if (b->moving)
        for (uint32_t j = 0; j < b->npairs; ++j)
        {
            float vx[2] = {0, 0}, vy[2] = {0, 0}, vz[2] = {0, 0};
            for (uint32_t l = 0; l < 2; ++l)
            {
                uint32_t i = b->order[2 * j + l];
                if (i == 0xFFFFFFFFu) continue;
                const float* v = d->sphere_vel + 3 * (size_t)i;
                vx[l] = v[0]; vy[l] = v[1]; vz[l] = v[2];
            }
            b->f4.push_back(f4(vx[0], vx[1], vy[0], vy[1]));
            b->f4.push_back(f4(vz[0], vz[1], 0, 0));
        }
    // group bounds: sphere around the box of the members' centres at t0 and t1, radius = farthest member surface, padded
    b->off_groups = (uint32_t)b->f4.size();
    {
        std::vector<float> gb(4 * (size_t)b->ngroups);
        for (uint32_t g = 0; g < b->ngroups; ++g)
        {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, c[3] = {0, 0, 0}, R = -1;
            for (int pass = 0; pass < 2; ++pass)
            {
                if (pass == 1) { if (lo[0] > hi[0]) break; for (int k = 0; k < 3; ++k) c[k] = 0.5 * (lo[k] + hi[k]); R = 0; }
                for (uint32_t l = 0; l < 8; ++l)
                {
                    uint32_t i = b->order[8 * g + l];
                    if (i == 0xFFFFFFFFu) continue;
                    for (int e2 = 0; e2 < 2; ++e2)
                    {
                        double t = e2 ? d->t1 : d->t0, p[3], dist2 = 0;
                        for (int k = 0; k < 3; ++k)
                        {
                            p[k] = (double)d->sphere_pos_r[4 * (size_t)i + k] + (b->moving ? (double)d->sphere_vel[3 * (size_t)i + k] * t : 0.0);
                            if (pass == 0) { if (p[k] < lo[k]) lo[k] = p[k]; if (p[k] > hi[k]) hi[k] = p[k]; }
                            else dist2 += (p[k] - c[k]) * (p[k] - c[k]);
                        }
                        if (pass == 1) { double far = sqrt(dist2) + fabs((double)d->sphere_pos_r[4 * (size_t)i + 3]); if (far > R) R = far; }
                    }
                }
            }
            // the search is approximate arithmetic: pad, so that no sphere the exact test could hit is culled;
            // an empty group keeps R2 = -1 (never touched)
            double R2 = R < 0 ? -1.0 : (R * (1.0 + 1e-4) + 1e-4) * (R * (1.0 + 1e-4) + 1e-4);
            gb[4 * (size_t)g] = (float)c[0]; gb[4 * (size_t)g + 1] = (float)c[1]; gb[4 * (size_t)g + 2] = (float)c[2]; gb[4 * (size_t)g + 3] = (float)R2;
        }
        for (uint32_t q = 0; q < b->ngroups / 4; ++q)
            for (int k = 0; k < 4; ++k)
                b->f4.push_back(f4(gb[16 * (size_t)q + k], gb[16 * (size_t)q + 4 + k], gb[16 * (size_t)q + 8 + k], gb[16 * (size_t)q + 12 + k]));
    }
    b->off_smat = (uint32_t)b->f4.size();
    for (uint32_t sl = 0; sl < slots; ++sl)
    {
        uint32_t i = b->order[sl];
        if (i == 0xFFFFFFFFu) { b->f4.push_back(f4(0, 0, 0, bits_to_float(0xFFFFFFFFu))); b->f4.push_back(f4(0, 0, 0, 0)); continue; }
        b->f4.push_back(f4(bits_to_float(d->sphere_mat_type[i]), d->sphere_mat_param[2 * (size_t)i], d->sphere_mat_param[2 * (size_t)i + 1], bits_to_float(i)));
        const float* c = d->sphere_mat_rgba + 4 * (size_t)i;
        b->f4.push_back(f4(c[0], c[1], c[2], c[3]));
    }
    b->off_planes = (uint32_t)b->f4.size();
    for (uint32_t i = 0; i < np; ++i)
    {
        const float* p = d->plane_n_d + 4 * (size_t)i;
        b->f4.push_back(f4(p[0], p[1], p[2], p[3]));
    }
    b->off_pmat = (uint32_t)b->f4.size();
    for (uint32_t i = 0; i < np; ++i)
    {
        b->f4.push_back(f4(bits_to_float(d->plane_mat_type[i]), d->plane_mat_param[2 * i], d->plane_mat_param[2 * i + 1], 0));
        const float* c = d->plane_mat_rgba + 4 * (size_t)i;
        b->f4.push_back(f4(c[0], c[1], c[2], c[3]));
    }
    // the folded camera (16 floats) rides in the same image: the kernels read it from LDS where they generate rays,
    // instead of holding it in 16 scalar registers for the whole launch
    b->off_cam = (uint32_t)b->f4.size();
    ptd::DeviceCamera cam;
    make_device_camera(d->camera, &cam);
    static_assert(sizeof(ptd::DeviceCamera) == 64, "camera = 4 float4");
    const float* cf = reinterpret_cast<const float*>(&cam);
    for (int k = 0; k < 4; ++k) b->f4.push_back(f4(cf[4 * k], cf[4 * k + 1], cf[4 * k + 2], cf[4 * k + 3]));
}

} // namespace pth
