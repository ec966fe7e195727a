// pt_host.h -- host-side preparation of what the kernels read: the LDS image of the scene and the folded camera.
// Host arithmetic here continues into device arithmetic (pt_device.h), so it keeps the reference's f32 operation
// order and is compiled without contraction (-ffp-contract=off; baseline x86-64 has no FMA anyway).
#pragma once
#include <stdint.h>
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
    uint32_t off_velp = 0, off_smat = 0, off_planes = 0, off_pmat = 0, npairs = 0, idx_mask = 1;
    bool moving = false;
};

inline float4 f4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// Layout: see ptd::SceneView.
inline void build_blob(const pt_scene_desc* d, Blob* b)
{
    const uint32_t ns = d->n_spheres, np = d->n_planes;
    b->moving = false;
    if (d->sphere_vel)
        for (uint32_t i = 0; i < 3 * ns; ++i)
            if (d->sphere_vel[i] != 0.0f) { b->moving = true; break; }
    b->npairs = (ns + 1) / 2;
    uint32_t bits = 1;
    while ((1u << bits) < 2 * b->npairs) ++bits;
    b->idx_mask = (1u << bits) - 1;
    b->f4.clear();
    // spheres, two per entry.  r2 = radius*radius as the reference forms it (geometry.cpp:47).  The odd slot of an
    // odd-sized scene is a sphere with r2 = -1: its discriminant is negative for every ray.
    for (uint32_t j = 0; j < b->npairs; ++j)
    {
        float cx[2] = {0, 0}, cy[2] = {0, 0}, cz[2] = {0, 0}, r2[2] = {-1.0f, -1.0f};
        for (uint32_t l = 0; l < 2; ++l)
        {
            uint32_t i = 2 * j + l;
            if (i >= ns) continue;
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
                uint32_t i = 2 * j + l;
                if (i >= ns) continue;
                const float* v = d->sphere_vel + 3 * (size_t)i;
                vx[l] = v[0]; vy[l] = v[1]; vz[l] = v[2];
            }
            b->f4.push_back(f4(vx[0], vx[1], vy[0], vy[1]));
            b->f4.push_back(f4(vz[0], vz[1], 0, 0));
        }
    b->off_smat = (uint32_t)b->f4.size();
    for (uint32_t i = 0; i < ns; ++i)
    {
        b->f4.push_back(f4(bits_to_float(d->sphere_mat_type[i]), d->sphere_mat_param[2 * i], d->sphere_mat_param[2 * i + 1], 0));
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
}

} // namespace pth
