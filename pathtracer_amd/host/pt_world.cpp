// pt_world.cpp -- host scene model, scene builders, resolve and BMP output (all host C++, no GPU code).
// Compiled with -ffp-contract=off: positions, radii and camera vectors must come out bit-identical to what the
// reference computes (tests/test_scenes_and_abi.py compares against tests/golden/scenes.npz, dumped from the compiled
// reference).  Citations are file:line in the reference tree (src/...).
#include "pt_world.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <memory>
#include <new>

#include "../csrc/pt_guard.h"

namespace pt {

// ------------------------------------------------------------------------------------------------ vector helpers
static inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline Vec3 scaled(Vec3 v, float s) { return Vec3(v.x * s, v.y * s, v.z * s); }
static inline Vec3 unit(Vec3 v)
{
    float len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); // vectors.cpp:107-110
    return scaled(v, 1.0f / len);                         // vectors.cpp:44-47,127-130
}

// ------------------------------------------------------------------------------------------------ palette
// 8-bit sRGB-ish triples divided by 255 in f32 (render_world.cpp:3-21); 255 -> exactly 1.0f.
static inline Colour rgb8(int r, int g, int b) { return Colour((float)r / 255, (float)g / 255, (float)b / 255); }
namespace colours {
Colour BLACK() { return Colour(0, 0, 0); }
Colour GREY() { return Colour(.5f, .5f, .5f); }
Colour SILVER() { return rgb8(212, 212, 212); }
Colour WHITE() { return Colour(1, 1, 1); }
Colour RED() { return rgb8(209, 42, 42); }
Colour BROWN() { return rgb8(130, 108, 78); }
Colour ORANGE() { return rgb8(255, 156, 56); }
Colour YELLOW() { return rgb8(255, 243, 107); }
Colour GREEN() { return rgb8(50, 161, 68); }
Colour DARK_GREEN() { return rgb8(58, 89, 65); }
Colour TEAL() { return rgb8(54, 158, 132); }
Colour BLUE() { return rgb8(41, 85, 196); }
Colour INDIGO() { return rgb8(45, 40, 168); }
Colour VIOLET() { return rgb8(164, 86, 219); }
Colour PINK() { return rgb8(235, 131, 231); }
Colour MAROON() { return rgb8(117, 39, 53); }
Colour LAVENDER() { return rgb8(219, 196, 245); }
Colour CYAN() { return rgb8(179, 247, 255); }
Colour GOLD() { return rgb8(255, 210, 8); }
} // namespace colours

// ------------------------------------------------------------------------------------------------ materials
Material Material::diffuse(Colour c) { Material m; m.type = DIFFUSE; m.colour = c; return m; }                       // render_world.cpp:27-33
Material Material::metal(Colour c, float rough) { Material m; m.type = METAL; m.colour = c; m.roughness = rough; return m; } // :35-42
Material Material::dialectric(float index) { Material m; m.type = DIALECTRIC; m.colour = colours::WHITE(); m.n = index; return m; } // :44-51

// ------------------------------------------------------------------------------------------------ camera
Camera::Camera(Vec3 p, float fovDegrees, float aspect)                                                               // camera.cpp:18-29
{
    pos = p;
    fov = fovDegrees * (3.141592653589793f / 180.0f);                                                                // types.h:6,19
    aspectRatio = aspect;
}
void Camera::set_target(Vec3 target) { dir = unit(target - pos); }                                                   // camera.cpp:46-49
void Camera::set_lens(float aperture, float focus) { lensRadius = aperture / 2; focusDistance = focus; }             // camera.cpp:51-55
void Camera::write13(float* o) const
{
    o[0] = pos.x; o[1] = pos.y; o[2] = pos.z; o[3] = dir.x; o[4] = dir.y; o[5] = dir.z; o[6] = up.x; o[7] = up.y; o[8] = up.z;
    o[9] = fov; o[10] = aspectRatio; o[11] = focusDistance; o[12] = lensRadius;
}

// ------------------------------------------------------------------------------------------------ world
SphereObject* World::add_sphere(Vec3 pos, float radius, const Material* material, Vec3 velocity)                     // render_world.cpp:80-99
{
    if (!material) return nullptr;
    SphereObject o;
    o.material = *material; o.pos = pos; o.radius = radius; o.velocity = velocity;
    objects.push_back(o);
    return &objects.back();
}
PlaneObject* World::add_plane(Vec3 normal, float d, const Material* material)                                        // render_world.cpp:101-118
{
    if (!material || planes.size() >= 64) return nullptr; // the device keeps the reference's 64-plane bound
    PlaneObject o;
    o.material = *material; o.normal = normal; o.offset = d;
    planes.push_back(o);
    return &planes.back();
}

// ------------------------------------------------------------------------------------------------ scene builders
// One layer of the sphere cube of scene 2 (scene_init.cpp:3-33).  Draw order per cell: material choice, size, then
// the colour triple -- which g++ evaluates right-to-left (blue first, then green, then red; blue is overwritten).
static void sphere_layer(World* world, HostRng& rng, uint32_t rows, uint32_t cols, float yLevel, float minRadius, float maxRadius)
{
    const float cell = maxRadius * 2.0f;
    const Material glass = Material::dialectric(1.42f);
    for (uint32_t row = 0; row < rows; ++row)
        for (uint32_t col = 0; col < cols; ++col)
        {
            const uint32_t choice = rng.u32(0, 100);
            const float size = rng.f32(minRadius, maxRadius);
            const Vec3 centre((float)row * cell, yLevel + size, (float)col * cell);
            float blue = rng.f32(), green = rng.f32(), red = rng.f32();
            (void)blue;
            const Colour tint(red, green, 1.0f);
            Material m;
            if (choice < 50) m = Material::diffuse(tint);
            else if (choice < 90) m = Material::metal(tint, rng.f32());
            else m = glass;
            world->add_sphere(centre, size, &m);
        }
}

void init_test_scene_1(World* world, Camera* camera, float aspectRatio)                                              // scene_init.cpp:35-107
{
    if (!world || !camera) return;
    HostRng rng;
    const Material palette[] = {
        Material::dialectric(1.5f), Material::metal(colours::GOLD(), 0.2f), Material::metal(colours::SILVER(), 0.01f),
        Material::diffuse(colours::WHITE()), Material::diffuse(colours::RED()), Material::diffuse(colours::ORANGE()),
        Material::diffuse(colours::YELLOW()), Material::diffuse(colours::GREEN()), Material::diffuse(colours::BLUE()),
        Material::diffuse(colours::INDIGO()), Material::diffuse(colours::VIOLET()), Material::diffuse(colours::PINK()),
        Material::diffuse(colours::MAROON()), Material::diffuse(colours::LAVENDER()), Material::diffuse(colours::CYAN()),
        Material::diffuse(colours::TEAL()), Material::diffuse(colours::DARK_GREEN()), Material::diffuse(colours::BROWN())};
    const uint32_t paletteSize = sizeof(palette) / sizeof(palette[0]);
    world->add_plane(Vec3(0.0f, 1.0f, 0.0f), 0.0f, &palette[3]);

    const uint32_t grid = 16;
    const float cell = 3.5f, rMin = 0.5f, rMax = 0.7f;
    const float origin = -(float)grid / 2 * cell;
    for (uint32_t gz = 0; gz < grid; ++gz)
        for (uint32_t gx = 0; gx < grid; ++gx)
        {
            const float cx = origin + (float)gx * cell + cell * 0.5f;
            const float cz = origin + (float)gz * cell + cell * 0.5f;
            const float x = cx + rng.f32(-0.5f, 0.5f) * cell * .7f;
            const float z = cz + rng.f32(-0.5f, 0.5f) * cell * .7f;
            const float radius = rng.f32(rMin, rMax);
            uint32_t pick = rng.u32(4, paletteSize);
            if (rng.f32() > 0.9f) pick = 0; // one sphere in ten is glass
            world->add_sphere(Vec3(x, 0.55f, z), radius, &palette[pick]);
        }
    world->add_sphere(Vec3(1.0f, 4.0f, 0.5f), 4.0f, &palette[0]);
    world->add_sphere(Vec3(-11.0f, 4.0f, -5.0f), 4.0f, &palette[1]);
    world->add_sphere(Vec3(5.5f, 4.0f, 15.0f), 4.0f, &palette[2]);

    *camera = Camera(Vec3(-3.5f, 2.5f, 35), 35, aspectRatio);
    camera->set_target(Vec3(0, .5f, 0));
    camera->up = unit(Vec3(.2f, 10, .8f)); // deliberately not orthogonal to dir: the image plane is sheared (camera.cpp:64-65)
    camera->set_lens(0.3f, 35.0f);
}

void init_test_scene_2(World* world, Camera* camera, float aspectRatio)                                              // scene_init.cpp:109-143
{
    if (!world || !camera) return;
    HostRng rng;
    const Material wall = Material::diffuse(colours::LAVENDER());
    const Material glass = Material::dialectric(1.42f);
    const uint32_t rows = 20, cols = 20;
    const float rMin = 0.8f, rMax = 2.0f, cell = rMax * 2.0f, gap = 2.5f;
    world->add_plane(Vec3(0.0f, 0.0f, 1.0f), -3.0f, &wall);                             // back wall
    world->add_plane(Vec3(-1.0f, 0.0, 0.0f), -((float)rows * (float)cell) - 3.0f, &wall); // right wall
    world->add_plane(Vec3(0.0f, 1.0f, 0.0f), -0.1f, &glass);                            // glass floor
    const float levels[6] = {0.0f, cell + gap, cell * 2.0f + gap * 2.0f, cell * 3.0f + gap * 3.0f, cell * 4.0f + gap * 4.0f, cell * 5.0f + gap * 5.0f};
    for (float y : levels) sphere_layer(world, rng, rows, cols, y, rMin, rMax);
    *camera = Camera(Vec3(-5.0f, 3.0f, rows * cell * 1.2f), 50.0f, aspectRatio);
    camera->set_target(Vec3((float)rows * 0.5f * cell, 30.0f, (float)cols * 0.5f * cell));
    camera->set_lens(1.0f, 40.0f);
}

void init_test_scene_3(World* world, Camera* camera, float aspectRatio)                                              // scene_init.cpp:145-177
{
    if (!world || !camera) return;
    const Material floor = Material::diffuse(colours::RED());
    world->add_plane(Vec3(0.0f, 1.0f, 0.0f), 0.0f, &floor);
    world->startTime = 0.0f;
    world->endTime = 1.0f;
    struct Item { Vec3 pos; float radius; Material m; Vec3 vel; };
    const Item items[] = {
        {Vec3(-1.0f, 2.0f, -2.0f), 0.5f, Material::diffuse(colours::PINK()), Vec3(0.5f, 0.0f, 0.0f)},
        {Vec3(2.0f, 1.0f, -3.5f), 1.0f, Material::diffuse(colours::YELLOW()), Vec3(0.0f, 0.1f, 0.0f)},
        {Vec3(-1.5f, 3.5f, -0.5f), 0.75f, Material::diffuse(colours::BROWN()), Vec3()},
        {Vec3(-0.5f, 1.2f, -0.6f), 0.3f, Material::diffuse(colours::MAROON()), Vec3()},
        {Vec3(-1.9f, 1.5f, -3.0f), 1.2f, Material::dialectric(1.42f), Vec3()}};
    for (const Item& it : items) world->add_sphere(it.pos, it.radius, &it.m, it.vel);
    *camera = Camera(Vec3(0.0f, 2.0f, 3.0f), 55.0f, aspectRatio);
}

// C2 of SURVEY.md 8(d): ground plane + diffuse / glass / rough-metal unit spheres, thin-lens camera. No RNG.
void init_materials_scene(World* world, Camera* camera, float aspectRatio)
{
    if (!world || !camera) return;
    const Material ground = Material::diffuse(colours::GREY());
    const Material glass = Material::dialectric(1.5f), red = Material::diffuse(colours::RED()), gold = Material::metal(colours::GOLD(), 0.3f);
    world->add_plane(Vec3(0.0f, 1.0f, 0.0f), 0.0f, &ground);
    world->add_sphere(Vec3(0.0f, 1.0f, 0.0f), 1.0f, &glass);
    world->add_sphere(Vec3(-2.2f, 1.0f, 0.0f), 1.0f, &red);
    world->add_sphere(Vec3(2.2f, 1.0f, 0.0f), 1.0f, &gold);
    *camera = Camera(Vec3(0.0f, 2.0f, 8.0f), 35.0f, aspectRatio);
    camera->set_target(Vec3(0.0f, 1.0f, 0.0f));
    camera->set_lens(0.2f, 8.06f);
}

// C5 of SURVEY.md 8(d): `count` spheres in a 1000 x 20 x 1000 slab over a diffuse ground plane; splitmix64 stream,
// not <random>, so the scene is the same on every platform.
void init_random_spheres_scene(World* world, Camera* camera, float aspectRatio, uint32_t count, uint64_t seed)
{
    if (!world || !camera) return;
    uint64_t state = seed;
    auto next01 = [&state]() {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (float)(z >> 40) * 5.9604644775390625e-8f; // 24 bits -> [0,1)
    };
    const Material ground = Material::diffuse(colours::GREY());
    world->add_plane(Vec3(0.0f, 1.0f, 0.0f), 0.0f, &ground);
    world->objects.reserve(count);
    const Material glass = Material::dialectric(1.5f);
    for (uint32_t i = 0; i < count; ++i)
    {
        const float radius = 0.05f + 0.45f * next01();
        const Vec3 centre(-500.0f + 1000.0f * next01(), radius + 20.0f * next01(), -500.0f + 1000.0f * next01());
        const float choice = next01();
        const Colour tint(next01(), next01(), next01());
        Material m;
        if (choice < 0.5f) m = Material::diffuse(tint);
        else if (choice < 0.9f) m = Material::metal(tint, next01());
        else m = glass;
        world->add_sphere(centre, radius, &m);
    }
    *camera = Camera(Vec3(0.0f, 15.0f, -60.0f), 50.0f, aspectRatio);
    camera->set_target(Vec3(0.0f, 0.0f, 0.0f));
}

// ------------------------------------------------------------------------------------------------ flatten
void flatten(const World& w, const Camera& c, FlatScene* f)
{
    const size_t ns = w.objects.size(), np = w.planes.size();
    f->pos_r.resize(4 * ns); f->vel.resize(3 * ns); f->type.resize(ns); f->rgba.resize(4 * ns); f->param.resize(2 * ns);
    f->pl_n_d.resize(4 * np); f->pl_type.resize(np); f->pl_rgba.resize(4 * np); f->pl_param.resize(2 * np);
    bool moving = false;
    for (size_t i = 0; i < ns; ++i)
    {
        const SphereObject& o = w.objects[i];
        f->pos_r[4*i] = o.pos.x; f->pos_r[4*i+1] = o.pos.y; f->pos_r[4*i+2] = o.pos.z; f->pos_r[4*i+3] = o.radius;
        f->vel[3*i] = o.velocity.x; f->vel[3*i+1] = o.velocity.y; f->vel[3*i+2] = o.velocity.z;
        moving = moving || o.velocity.x != 0 || o.velocity.y != 0 || o.velocity.z != 0;
        f->type[i] = o.material.type;
        f->rgba[4*i] = o.material.colour.r; f->rgba[4*i+1] = o.material.colour.g; f->rgba[4*i+2] = o.material.colour.b; f->rgba[4*i+3] = o.material.colour.a;
        f->param[2*i] = o.material.roughness; f->param[2*i+1] = o.material.n;
    }
    for (size_t i = 0; i < np; ++i)
    {
        const PlaneObject& o = w.planes[i];
        f->pl_n_d[4*i] = o.normal.x; f->pl_n_d[4*i+1] = o.normal.y; f->pl_n_d[4*i+2] = o.normal.z; f->pl_n_d[4*i+3] = o.offset;
        f->pl_type[i] = o.material.type;
        f->pl_rgba[4*i] = o.material.colour.r; f->pl_rgba[4*i+1] = o.material.colour.g; f->pl_rgba[4*i+2] = o.material.colour.b; f->pl_rgba[4*i+3] = o.material.colour.a;
        f->pl_param[2*i] = o.material.roughness; f->pl_param[2*i+1] = o.material.n;
    }
    pt_scene_desc& d = f->desc;
    d.n_spheres = (uint32_t)ns;
    d.sphere_pos_r = f->pos_r.data(); d.sphere_vel = moving ? f->vel.data() : nullptr; d.sphere_mat_type = f->type.data();
    d.sphere_mat_rgba = f->rgba.data(); d.sphere_mat_param = f->param.data();
    d.n_planes = (uint32_t)np;
    d.plane_n_d = f->pl_n_d.data(); d.plane_mat_type = f->pl_type.data(); d.plane_mat_rgba = f->pl_rgba.data(); d.plane_mat_param = f->pl_param.data();
    d.t0 = w.startTime; d.t1 = w.endTime;
    c.write13(d.camera);
}

// ------------------------------------------------------------------------------------------------ reference-topology tree
// build_bvh_tree / sort_render_objects (src/main.cpp:93-159) in the reference's own recipe, for the compatibility mode
// PT_FLAG_REFERENCE_QUERY: the reference's pairwise-combine query (src/main.cpp:78-87) gives results that depend on
// the tree's topology, so reproducing its images to the noise floor needs ITS tree -- random axis per node (one
// random_u32(0,3) for every node, leaves included), in-place sort of the sub-range by centre coordinate at t = 0,
// split at the midpoint, leaf boxes swept over [startTime, endTime], inner box = union of the children.
// The reference permutes World::objects in place; here `order` is permuted instead (order[i] = original index of the
// sphere that ends up at position i), so leaves name spheres of the caller's arrays.
namespace {

struct RefBox { float px, py, pz, hw, hh, hl; }; // Rect3f (src/geometry.h:31-52): centre + half extents

// Rect3f::from_bounds over the corners of two boxes (src/geometry.cpp:9-31; MIN_VALUE / MAX_VALUE are `<` ternaries, src/types.h:12-13)
RefBox ref_union(const RefBox& a, const RefBox& b)
{
    auto mn = [](float x, float y) { return x < y ? x : y; };
    auto mx = [](float x, float y) { return x < y ? y : x; };
    const float minX = mn(a.px - a.hw, b.px - b.hw), maxX = mx(a.px + a.hw, b.px + b.hw);
    const float minY = mn(a.py - a.hh, b.py - b.hh), maxY = mx(a.py + a.hh, b.py + b.hh);
    const float minZ = mn(a.pz - a.hl, b.pz - b.hl), maxZ = mx(a.pz + a.hl, b.pz + b.hl);
    RefBox r;
    r.px = (minX + maxX) * 0.5f; r.py = (minY + maxY) * 0.5f; r.pz = (minZ + maxZ) * 0.5f;
    r.hw = (maxX - minX) * 0.5f; r.hh = (maxY - minY) * 0.5f; r.hl = (maxZ - minZ) * 0.5f;
    return r;
}
// bounding_box(Sphere) (src/geometry.cpp:33-37): Rect3f(pos, 2r, 2r, 2r), whose ctor halves the extents again
RefBox ref_sphere_box(float x, float y, float z, float radius)
{
    const float e = radius * 2.0f;
    return RefBox{x, y, z, e * 0.5f, e * 0.5f, e * 0.5f};
}

struct RefTreeBuilder
{
    const pt_scene_desc* d;
    pt::HostRng rng;
    std::vector<uint32_t> order;
    std::vector<RefBox> boxes;
    std::vector<int32_t> links; // left, right, object per node
    uint32_t depth = 0;

    float key(uint32_t pos, uint32_t axis) const { return d->sphere_pos_r[4 * (size_t)order[pos] + axis]; } // SphereObject::pos() at time 0
    // sort_render_objects (src/main.cpp:93-129): quicksort, first element as pivot, strict `<` partition
    void sort(uint32_t start, uint32_t end, uint32_t axis)
    {
        if (end - start == 1) return;
        if (end - start == 2)
        {
            if (key(start, axis) > key(start + 1, axis)) std::swap(order[start], order[start + 1]);
            return;
        }
        const uint32_t pivot = start;
        uint32_t split = start + 1;
        for (uint32_t i = start + 1; i < end; ++i)
            if (key(i, axis) < key(pivot, axis)) { std::swap(order[i], order[split]); ++split; }
        std::swap(order[pivot], order[split - 1]);
        sort(start, split, axis);
        if (split != end) sort(split, end, axis);
    }
    // SphereObject::get_bounding_box (src/render_world.cpp:62-74)
    RefBox leaf_box(uint32_t orig) const
    {
        const float* s = d->sphere_pos_r + 4 * (size_t)orig;
        const float zero[3] = {0, 0, 0};
        const float* v = d->sphere_vel ? d->sphere_vel + 3 * (size_t)orig : zero;
        const RefBox a = ref_sphere_box(s[0] + v[0] * d->t0, s[1] + v[1] * d->t0, s[2] + v[2] * d->t0, s[3]);
        const RefBox b = ref_sphere_box(s[0] + v[0] * d->t1, s[1] + v[1] * d->t1, s[2] + v[2] * d->t1, s[3]);
        return ref_union(a, b);
    }
    // build_bvh_tree (src/main.cpp:131-159); nodes are appended in the order the reference allocates them (children
    // first, the root last).  Returns the node's index.
    int32_t build(uint32_t start, uint32_t end, uint32_t level)
    {
        const uint32_t axis = rng.u32(0, 3);
        if (level > depth) depth = level;
        if (end - start == 1)
        {
            boxes.push_back(leaf_box(order[start]));
            links.insert(links.end(), {-1, -1, (int32_t)order[start]});
            return (int32_t)boxes.size() - 1;
        }
        sort(start, end, axis);
        const uint32_t mid = (start + end) / 2;
        const int32_t l = build(start, mid, level + 1);
        const int32_t r = build(mid, end, level + 1);
        boxes.push_back(ref_union(boxes[(size_t)l], boxes[(size_t)r]));
        links.insert(links.end(), {l, r, -1});
        return (int32_t)boxes.size() - 1;
    }
};

} // namespace

} // namespace pt

// ================================================================================================ C-ABI, host-only part
static inline float clamp01(float v) { if (v < 0.0f) return 0.0f; else if (v > 1.0f) return 1.0f; return v; } // utils.h:8-15

extern "C" {

pt_scene* pt_scene_builtin(int id, float aspect, uint32_t n_for_id5)
{
    return ptg::guard_ptr<pt_scene>([&]() -> pt_scene* {
        std::unique_ptr<pt_scene> s(new pt_scene());
        switch (id)
        {
        case 1: pt::init_test_scene_1(&s->world, &s->camera, aspect); break;
        case 2: pt::init_test_scene_2(&s->world, &s->camera, aspect); break;
        case 3: pt::init_test_scene_3(&s->world, &s->camera, aspect); break;
        case 4: pt::init_materials_scene(&s->world, &s->camera, aspect); break;
        case 5: pt::init_random_spheres_scene(&s->world, &s->camera, aspect, n_for_id5 ? n_for_id5 : 1000000u); break;
        default: return nullptr;
        }
        pt::flatten(s->world, s->camera, &s->flat);
        return s.release();
    });
}

uint32_t pt_reference_bvh(const pt_scene_desc* d, uint32_t rng_skip, uint32_t cap_nodes, float* boxes6, int32_t* links3, uint32_t* order, uint32_t* depth_out)
{
    if (!d || d->n_spheres == 0 || d->n_spheres > PT_REFERENCE_MAX_SPHERES || !d->sphere_pos_r) return 0;
    try {
    pt::RefTreeBuilder b;
    b.d = d;
    for (uint32_t i = 0; i < rng_skip; ++i) (void)b.rng.f32();
    b.order.resize(d->n_spheres);
    for (uint32_t i = 0; i < d->n_spheres; ++i) b.order[i] = i;
    b.boxes.reserve(2 * (size_t)d->n_spheres); b.links.reserve(6 * (size_t)d->n_spheres);
    b.build(0, d->n_spheres, 0);
    const uint32_t nn = (uint32_t)b.boxes.size();
    if (depth_out) *depth_out = b.depth;
    if (nn <= cap_nodes)
    {
        if (boxes6) memcpy(boxes6, b.boxes.data(), (size_t)nn * sizeof(pt::RefBox));
        if (links3) memcpy(links3, b.links.data(), (size_t)nn * 3 * sizeof(int32_t));
    }
    if (order) memcpy(order, b.order.data(), (size_t)d->n_spheres * sizeof(uint32_t));
    return nn;
    } catch (...) { return 0; } // (0 nodes = failure, as for a scene the reference cannot hold)
}

int pt_scene_get_desc(const pt_scene* s, pt_scene_desc* out)
{
    if (!s || !out) return PT_ERR_INVALID;
    *out = s->flat.desc;
    return PT_OK;
}

void pt_scene_free(pt_scene* s) { delete s; }

// ---- scene files (the reference hard-codes its scene call, main.cpp:396) --------------------
// Text, one record per line, '#' comments; floats are written with 9 significant digits (round-trips f32 exactly):
//   pt_scene 1
//   time   t0 t1
//   camera px py pz  dx dy dz  ux uy uz  fov aspect focus lensRadius
//   sphere x y z r  vx vy vz  type  r g b a  roughness n
//   plane  nx ny nz d  type  r g b a  roughness n
int pt_scene_save(const pt_scene_desc* d, const char* path)
{
    if (!d || !path) return PT_ERR_INVALID;
    FILE* f = fopen(path, "w");
    if (!f) return PT_ERR_IO;
    fprintf(f, "pt_scene 1\n# %u spheres, %u planes\ntime %.9g %.9g\ncamera", d->n_spheres, d->n_planes, d->t0, d->t1);
    for (int k = 0; k < 13; ++k) fprintf(f, " %.9g", d->camera[k]);
    fprintf(f, "\n");
    for (uint32_t i = 0; i < d->n_planes; ++i)
    {
        const float* p = d->plane_n_d + 4 * (size_t)i; const float* c = d->plane_mat_rgba + 4 * (size_t)i; const float* q = d->plane_mat_param + 2 * (size_t)i;
        fprintf(f, "plane %.9g %.9g %.9g %.9g %u %.9g %.9g %.9g %.9g %.9g %.9g\n", p[0], p[1], p[2], p[3], d->plane_mat_type[i], c[0], c[1], c[2], c[3], q[0], q[1]);
    }
    for (uint32_t i = 0; i < d->n_spheres; ++i)
    {
        const float* p = d->sphere_pos_r + 4 * (size_t)i; const float* c = d->sphere_mat_rgba + 4 * (size_t)i; const float* q = d->sphere_mat_param + 2 * (size_t)i;
        const float zero[3] = {0, 0, 0};
        const float* v = d->sphere_vel ? d->sphere_vel + 3 * (size_t)i : zero;
        fprintf(f, "sphere %.9g %.9g %.9g %.9g %.9g %.9g %.9g %u %.9g %.9g %.9g %.9g %.9g %.9g\n", p[0], p[1], p[2], p[3], v[0], v[1], v[2],
                d->sphere_mat_type[i], c[0], c[1], c[2], c[3], q[0], q[1]);
    }
    const bool ok = !ferror(f);
    return (fclose(f) == 0 && ok) ? PT_OK : PT_ERR_IO;
}

pt_scene* pt_scene_load(const char* path)
{
    if (!path) return nullptr;
    struct File { FILE* f; ~File() { if (f) fclose(f); } } file{fopen(path, "r")};
    FILE* f = file.f;
    if (!f) return nullptr;
    return ptg::guard_ptr<pt_scene>([&]() -> pt_scene* {
    std::unique_ptr<pt_scene> s(new pt_scene());
    pt::FlatScene& fl = s->flat;
    pt_scene_desc& d = fl.desc;
    bool ok = true, header = false, haveCamera = false, moving = false;
    char line[1024];
    while (ok && fgets(line, sizeof line, f))
    {
        char* p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (*p == '#' || *p == '\n' || *p == '\r' || *p == 0) continue;
        char word[16]; int used = 0;
        if (sscanf(p, "%15s%n", word, &used) != 1) { ok = false; break; }
        p += used;
        float v[14]; unsigned type = 0;
        auto floats = [&](int n, float* out) { for (int k = 0; k < n; ++k) { char* e; out[k] = strtof(p, &e); if (e == p) return false; p = e; } return true; };
        auto uint = [&](unsigned* out) { char* e; unsigned long u = strtoul(p, &e, 10); if (e == p) return false; p = e; *out = (unsigned)u; return true; };
        if (!strcmp(word, "pt_scene")) { unsigned ver = 0; ok = uint(&ver) && ver == 1; header = ok; }
        else if (!header) ok = false;
        else if (!strcmp(word, "time")) { ok = floats(2, v); d.t0 = v[0]; d.t1 = v[1]; }
        else if (!strcmp(word, "camera")) { ok = floats(13, d.camera); haveCamera = ok; }
        else if (!strcmp(word, "plane"))
        {
            ok = floats(4, v) && uint(&type) && floats(6, v + 4) && type <= PT_MAT_DIALECTRIC;
            if (ok) { fl.pl_n_d.insert(fl.pl_n_d.end(), v, v + 4); fl.pl_type.push_back(type); fl.pl_rgba.insert(fl.pl_rgba.end(), v + 4, v + 8); fl.pl_param.insert(fl.pl_param.end(), v + 8, v + 10); }
        }
        else if (!strcmp(word, "sphere"))
        {
            ok = floats(7, v) && uint(&type) && floats(6, v + 7) && type <= PT_MAT_DIALECTRIC;
            if (ok)
            {
                fl.pos_r.insert(fl.pos_r.end(), v, v + 4); fl.vel.insert(fl.vel.end(), v + 4, v + 7); fl.type.push_back(type);
                fl.rgba.insert(fl.rgba.end(), v + 7, v + 11); fl.param.insert(fl.param.end(), v + 11, v + 13);
                moving = moving || v[4] != 0 || v[5] != 0 || v[6] != 0;
            }
        }
        else ok = false;
    }
    if (!ok || !header || !haveCamera) return nullptr;
    d.n_spheres = (uint32_t)fl.type.size(); d.n_planes = (uint32_t)fl.pl_type.size();
    d.sphere_pos_r = fl.pos_r.data(); d.sphere_vel = moving ? fl.vel.data() : nullptr; d.sphere_mat_type = fl.type.data();
    d.sphere_mat_rgba = fl.rgba.data(); d.sphere_mat_param = fl.param.data();
    d.plane_n_d = fl.pl_n_d.data(); d.plane_mat_type = fl.pl_type.data(); d.plane_mat_rgba = fl.pl_rgba.data(); d.plane_mat_param = fl.pl_param.data();
    return s.release();
    });
}

int pt_resolve(const float* sum, uint32_t n, uint32_t spp, float* out)                       // main.cpp:358
{
    if (!sum || !out || spp == 0) return PT_ERR_INVALID;
    const float inv = 1.0f / (float)spp;                                                    // vectors.cpp:100-103
    for (size_t i = 0; i < (size_t)n * 4; ++i) out[i] = clamp01(sum[i] * inv);
    return PT_OK;
}

int pt_pack_bgra(const float* px, uint32_t n, uint32_t* out)                                 // file_io.h:29-45
{
    if (!px || !out) return PT_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i)
    {
        const float* c = px + 4 * (size_t)i;
        uint32_t r = (uint8_t)(255.0f * sqrtf(c[0])), g = (uint8_t)(255.0f * sqrtf(c[1]));
        uint32_t b = (uint8_t)(255.0f * sqrtf(c[2])), a = (uint8_t)(255.0f * sqrtf(c[3]));
        out[i] = (a << 24) | (r << 16) | (g << 8) | b;
    }
    return PT_OK;
}

int pt_bmp_header(uint32_t w, uint32_t h, uint8_t* o)                                        // file_io.h:6-23
{
    if (!o || w == 0 || h == 0) return PT_ERR_INVALID;
    memset(o, 0, 58);
    auto put32 = [o](int at, uint32_t v) { o[at] = (uint8_t)v; o[at+1] = (uint8_t)(v >> 8); o[at+2] = (uint8_t)(v >> 16); o[at+3] = (uint8_t)(v >> 24); };
    auto put16 = [o](int at, uint32_t v) { o[at] = (uint8_t)v; o[at+1] = (uint8_t)(v >> 8); };
    const uint32_t headerBytes = 14 + 44; // BITMAPFILEHEADER + BITMAPINFO (40-byte header and one RGBQUAD)
    o[0] = 'B'; o[1] = 'M';
    put32(2, headerBytes + 4u * w * h);
    put32(10, headerBytes);
    put32(14, 40);
    put32(18, w);
    put32(22, (uint32_t)(-(int32_t)h)); // negative height: rows stored top-down
    put16(26, 1);
    put16(28, 32);
    return PT_OK;
}

int pt_bmp_write_bgra(const char* path, const uint32_t* bgra, uint32_t w, uint32_t h)
{
    if (!path || !bgra) return PT_ERR_INVALID;
    uint8_t header[58];
    int st = pt_bmp_header(w, h, header);
    if (st != PT_OK) return st;
    FILE* f = fopen(path, "wb");
    if (!f) return PT_ERR_IO;
    bool ok = fwrite(header, 1, 58, f) == 58 && fwrite(bgra, 4, (size_t)w * h, f) == (size_t)w * h;
    ok = (fclose(f) == 0) && ok;
    return ok ? PT_OK : PT_ERR_IO;
}

int pt_bmp_write(const char* path, const float* px, uint32_t w, uint32_t h)                  // file_io.h:4-52
{
    if (!path || !px || w == 0 || h == 0) return PT_ERR_INVALID;
    uint32_t* packed = new (std::nothrow) uint32_t[(size_t)w * h];
    if (!packed) return PT_ERR_NO_MEMORY;
    pt_pack_bgra(px, w * h, packed);
    int st = pt_bmp_write_bgra(path, packed, w, h);
    delete[] packed;
    return st;
}

} // extern "C"
