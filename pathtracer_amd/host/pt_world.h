// pt_world.h -- host-side scene model: the C++ face a user of the reference's World / Camera / Material
// (src/render_world.h, src/camera.h) finds in this build.  Same names, same argument meaning, same failure
// behaviour (add_* return nullptr on a null material); storage is std::vector instead of the reference's fixed
// World::objects[4192] / planes[64] (src/render_world.h:85,88) because the 10^6-sphere configuration does not fit
// those.  Everything here is plain host C++; flatten() produces the pt_scene_desc the C-ABI takes.
#pragma once
#include <stdint.h>
#include <random>
#include <vector>

#include "../../include/pt_hip.h"

namespace pt {

struct Vec3 { float x = 0, y = 0, z = 0; Vec3() {} Vec3(float a, float b, float c) : x(a), y(b), z(c) {} };
struct Colour { float r = 0, g = 0, b = 0, a = 0; Colour() {} Colour(float R, float G, float B, float A = 1.0f) : r(R), g(G), b(B), a(A) {} };

// The reference's palette (src/render_world.cpp:3-21) by name.
namespace colours {
Colour BLACK(); Colour GREY(); Colour SILVER(); Colour WHITE(); Colour RED(); Colour BROWN(); Colour ORANGE(); Colour YELLOW();
Colour GREEN(); Colour DARK_GREEN(); Colour TEAL(); Colour BLUE(); Colour INDIGO(); Colour VIOLET(); Colour PINK(); Colour MAROON();
Colour LAVENDER(); Colour CYAN(); Colour GOLD();
}

struct Material // src/render_world.h:28-53
{
    enum Type : uint32_t { NONE = PT_MAT_NONE, DIFFUSE = PT_MAT_DIFFUSE, METAL = PT_MAT_METAL, DIALECTRIC = PT_MAT_DIALECTRIC };
    Type type = NONE;
    Colour colour;
    float roughness = 0.0f;
    float n = 0.0f;
    static Material diffuse(Colour c);
    static Material metal(Colour c, float roughness = 0.0f);
    static Material dialectric(float refractiveIndex);
};

struct SphereObject { Material material; Vec3 pos; float radius = 1.0f; Vec3 velocity; }; // src/render_world.h:57-71
struct PlaneObject { Material material; Vec3 normal; float offset = 0.0f; };              // src/render_world.h:73-79

struct Camera // src/camera.h:9-38, src/camera.cpp:5-55
{
    Vec3 pos, dir{0.0f, 0.0f, -1.0f}, up{0.0f, 1.0f, 0.0f};
    float fov = 3.141592653589793f / 2, aspectRatio = 1.0f, focusDistance = 1.0f;
    Camera() {}
    Camera(Vec3 pos, float fovDegrees, float aspectRatio);
    void set_target(Vec3 target);
    void set_lens(float aperture, float focusDistance);
    float lens_radius() const { return lensRadius; }
    void write13(float* out) const; // pos dir up fov aspect focus lensRadius
private:
    float lensRadius = 0.0f;
};

struct World // src/render_world.h:82-96
{
    std::vector<SphereObject> objects;
    std::vector<PlaneObject> planes;
    float startTime = 0.0f, endTime = 0.0f;
    SphereObject* add_sphere(Vec3 pos, float radius, const Material* material, Vec3 velocity = Vec3());
    PlaneObject* add_plane(Vec3 normal, float d, const Material* material);
};

// The reference's random_f32 family (src/utils.h:38-63) on an explicit engine: std::mt19937, default seed, a fresh
// uniform_real_distribution per draw -- i.e. what the reference's main thread sees, so the builders below lay out
// the same scenes the reference does when it is built with libstdc++ (pinned by tests/golden/scenes.npz).
struct HostRng
{
    std::mt19937 engine;
    float f32() { std::uniform_real_distribution<float> d(0.0f, 1.0f); return d(engine); }
    float f32(float lo, float hi) { return f32() * (hi - lo) + lo; }
    uint32_t u32(uint32_t lo, uint32_t hi) { return (uint32_t)f32((float)lo, (float)hi); }
};

// src/scene_init.h:7-14
void init_test_scene_1(World* world, Camera* camera, float aspectRatio);
void init_test_scene_2(World* world, Camera* camera, float aspectRatio);
void init_test_scene_3(World* world, Camera* camera, float aspectRatio);
// Build-defined scenes of SURVEY.md 8(d): C2 (materials + depth of field) and C5 (n random spheres, BVH-sized).
void init_materials_scene(World* world, Camera* camera, float aspectRatio);
void init_random_spheres_scene(World* world, Camera* camera, float aspectRatio, uint32_t count, uint64_t seed = 1);

// Owns the flat arrays a pt_scene_desc points into.
struct FlatScene
{
    std::vector<float> pos_r, vel, rgba, param, pl_n_d, pl_rgba, pl_param;
    std::vector<uint32_t> type, pl_type;
    pt_scene_desc desc{};
};
void flatten(const World& world, const Camera& camera, FlatScene* out);

} // namespace pt

struct pt_scene { pt::World world; pt::Camera camera; pt::FlatScene flat; };
