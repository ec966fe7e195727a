// pt_main.cpp -- command-line renderer: the counterpart of the reference's main() (src/main.cpp:386-495) on top of
// libpt_hip.so.  What the reference fixes at compile time -- image size, SAMPLES_PER_PIXEL, MAX_RAY_DEPTH
// (src/main.cpp:29-42), the scene call (src/main.cpp:396), the output name (src/main.cpp:487) -- are options here;
// the flow is the same: build the scene, time the "PathTracing" span (src/main.cpp:412-481), resolve, write the BMP.
// Several GPUs: one host thread and one pt_ctx per device, each renders a contiguous slice of the sample range
// (SURVEY.md 8e); the slices are summed on the host in device order before the resolve.  (The RCCL all-reduce variant
// of the same split is pathtracer_amd/distributed.py; this tool stays free of a communicator so that it links against
// nothing but libpt_hip.so.)
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pt_hip.h"

namespace {

struct Options
{
    int scene = 1; uint32_t n5 = 100000; std::string sceneFile, saveScene, out = "output.bmp";
    uint32_t width = 800, height = 450, spp = 32, depth = 5, flags = 0; // the reference's defaults (src/main.cpp:29-42)
    uint64_t seed = 0; int gpus = 1;
};

void usage()
{
    fprintf(stderr,
            "usage: pt_render [--scene 1|2|3|4|5] [--spheres N (scene 5)] [--scene-file FILE] [--save-scene FILE]\n"
            "                 [--width W] [--height H] [--spp S] [--depth D] [--seed N] [--gpus G] [--flags F] [--out FILE.bmp]\n"
            "  scenes 1-3: the reference's init_test_scene_1/2/3; 4: materials + depth of field; 5: N random spheres\n");
}

bool parse(int argc, char** argv, Options* o)
{
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        auto next = [&](const char** v) { if (i + 1 >= argc) return false; *v = argv[++i]; return true; };
        const char* v = nullptr;
        if (a == "--help" || a == "-h") return false;
        else if (a == "--scene" && next(&v)) o->scene = atoi(v);
        else if (a == "--spheres" && next(&v)) o->n5 = (uint32_t)strtoul(v, nullptr, 10);
        else if (a == "--scene-file" && next(&v)) o->sceneFile = v;
        else if (a == "--save-scene" && next(&v)) o->saveScene = v;
        else if (a == "--width" && next(&v)) o->width = (uint32_t)strtoul(v, nullptr, 10);
        else if (a == "--height" && next(&v)) o->height = (uint32_t)strtoul(v, nullptr, 10);
        else if (a == "--spp" && next(&v)) o->spp = (uint32_t)strtoul(v, nullptr, 10);
        else if (a == "--depth" && next(&v)) o->depth = (uint32_t)strtoul(v, nullptr, 10);
        else if (a == "--seed" && next(&v)) o->seed = strtoull(v, nullptr, 10);
        else if (a == "--gpus" && next(&v)) o->gpus = atoi(v);
        else if (a == "--flags" && next(&v)) o->flags = (uint32_t)strtoul(v, nullptr, 0);
        else if (a == "--out" && next(&v)) o->out = v;
        else { fprintf(stderr, "pt_render: bad argument '%s'\n", a.c_str()); return false; }
    }
    return o->width > 0 && o->height > 0 && o->spp > 0 && o->gpus > 0;
}

double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

} // namespace

int main(int argc, char** argv)
{
    Options o;
    if (!parse(argc, argv, &o)) { usage(); return 2; }

    // ---- scene (src/main.cpp:388-402)
    const auto tScene = std::chrono::steady_clock::now();
    pt_scene* scene = o.sceneFile.empty() ? pt_scene_builtin(o.scene, (float)o.width / (float)o.height, o.n5) : pt_scene_load(o.sceneFile.c_str());
    if (!scene) { fprintf(stderr, "pt_render: cannot build/load the scene\n"); return 1; }
    pt_scene_desc d;
    if (pt_scene_get_desc(scene, &d) != PT_OK) return 1;
    printf("Scene: %u spheres, %u planes (%.1f ms)\n", d.n_spheres, d.n_planes, ms_since(tScene));
    if (!o.saveScene.empty() && pt_scene_save(&d, o.saveScene.c_str()) != PT_OK) { fprintf(stderr, "pt_render: cannot write %s\n", o.saveScene.c_str()); return 1; }

    const int have = pt_hip_device_count();
    if (have < o.gpus) { fprintf(stderr, "pt_render: %d HIP device(s) visible, %d asked for -- there is no CPU fallback\n", have, o.gpus); return 3; }
    if ((uint32_t)o.gpus > o.spp) o.gpus = (int)o.spp;

    // ---- contexts + upload (BVH build; src/main.cpp:402)
    const auto tUpload = std::chrono::steady_clock::now();
    std::vector<pt_ctx*> ctx((size_t)o.gpus, nullptr);
    for (int g = 0; g < o.gpus; ++g)
    {
        int st = pt_hip_create(g, &ctx[(size_t)g]);
        if (st == PT_OK) st = pt_hip_upload_scene(ctx[(size_t)g], &d);
        if (st != PT_OK) { fprintf(stderr, "pt_render: device %d: %s (%s)\n", g, pt_hip_strerror(st), ctx[(size_t)g] ? pt_hip_last_error(ctx[(size_t)g]) : ""); return 1; }
    }
    printf("Upload + BVH build on %d GPU(s): %.1f ms\n", o.gpus, ms_since(tUpload));

    // ---- the "PathTracing" span (src/main.cpp:412-481): every device renders its slice of the samples
    const size_t floats = (size_t)o.width * o.height * 4;
    std::vector<std::vector<float>> sums((size_t)o.gpus, std::vector<float>(floats));
    std::vector<int> status((size_t)o.gpus, PT_OK);
    std::vector<pt_stats> stats((size_t)o.gpus);
    const auto tRender = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> workers;
        for (int g = 0; g < o.gpus; ++g)
            workers.emplace_back([&, g]() {
                pt_params p{};
                p.width = o.width; p.height = o.height; p.max_depth = o.depth; p.flags = o.flags; p.seed = o.seed;
                p.spp_begin = (uint32_t)((uint64_t)o.spp * (uint64_t)g / (uint64_t)o.gpus);
                p.spp_end = (uint32_t)((uint64_t)o.spp * (uint64_t)(g + 1) / (uint64_t)o.gpus);
                status[(size_t)g] = pt_hip_render(ctx[(size_t)g], &p, sums[(size_t)g].data());
                if (status[(size_t)g] == PT_OK) status[(size_t)g] = pt_hip_last_stats(ctx[(size_t)g], &stats[(size_t)g]);
            });
        for (auto& w : workers) w.join();
    }
    for (int g = 0; g < o.gpus; ++g)
        if (status[(size_t)g] != PT_OK) { fprintf(stderr, "pt_render: device %d: %s (%s)\n", g, pt_hip_strerror(status[(size_t)g]), pt_hip_last_error(ctx[(size_t)g])); return 1; }
    for (int g = 1; g < o.gpus; ++g)
        for (size_t i = 0; i < floats; ++i) sums[0][i] += sums[(size_t)g][i];
    std::vector<float> pixels(floats);
    if (pt_resolve(sums[0].data(), o.width * o.height, o.spp, pixels.data()) != PT_OK) return 1; // src/main.cpp:358
    const double renderMs = ms_since(tRender);
    uint64_t samples = 0, segments = 0; double kernelMs = 0;
    for (int g = 0; g < o.gpus; ++g) { samples += stats[(size_t)g].samples; segments += stats[(size_t)g].segments; if (stats[(size_t)g].kernel_ms > kernelMs) kernelMs = stats[(size_t)g].kernel_ms; }
    printf("PathTracing: Time elapsed: %.3f ms  (%ux%u, %u spp, depth %u: %.1f Msamples/s; kernel %.3f ms, %.3f segments/sample)\n",
           renderMs, o.width, o.height, o.spp, o.depth, (double)samples / renderMs / 1e3, kernelMs, samples ? (double)segments / (double)samples : 0.0);

    // ---- output (src/main.cpp:487)
    const auto tWrite = std::chrono::steady_clock::now();
    const int st = pt_bmp_write(o.out.c_str(), pixels.data(), o.width, o.height);
    if (st != PT_OK) { fprintf(stderr, "pt_render: cannot write %s: %s\n", o.out.c_str(), pt_hip_strerror(st)); return 1; }
    printf("Wrote %s (%.1f ms)\n", o.out.c_str(), ms_since(tWrite));
    for (pt_ctx* c : ctx) pt_hip_destroy(c);
    pt_scene_free(scene);
    return 0;
}
