// pt_main.cpp -- command-line renderer: the counterpart of the reference's main() (src/main.cpp:386-495) on top of
// libpt_hip.so.  What the reference fixes at compile time -- image size, SAMPLES_PER_PIXEL, MAX_RAY_DEPTH
// (src/main.cpp:29-42), the scene call (src/main.cpp:396), the output name (src/main.cpp:487) -- are options here;
// the flow is the same: build the scene, time the "PathTracing" span (src/main.cpp:412-481), resolve, write the BMP.
// Several GPUs: pt_node_* -- one pt_ctx, stream and RCCL communicator per device; each renders a contiguous slice of the
// sample range, one ncclAllReduce(sum) joins the f32 sums on the devices, device 0 resolves and packs.
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/pt_hip.h"

namespace {

struct Options
{
    int scene = 1; uint32_t n5 = 100000; std::string sceneFile, saveScene, out = "output.bmp";
    uint32_t width = 800, height = 450, spp = 32, depth = 5, flags = 0; // the reference's defaults (src/main.cpp:29-42)
    uint64_t seed = 0; int gpus = 1; uint32_t rngSkip = 0;
    uint32_t pieces = 1;   // > 1: the samples in this many pieces added to the GPUs' resident sums (PT_FLAG_ACCUMULATE), a progress line each
};

void usage()
{
    fprintf(stderr,
            "usage: pt_render [--scene 1|2|3|4|5] [--spheres N (scene 5)] [--scene-file FILE] [--save-scene FILE]\n"
            "                 [--width W] [--height H] [--spp S] [--depth D] [--seed N] [--gpus G] [--flags F] [--out FILE.bmp]\n"
            "                 [--reference-query [--rng-skip N]] [--pieces N]\n"
            "  scenes 1-3: the reference's init_test_scene_1/2/3; 4: materials + depth of field; 5: N random spheres\n"
            "  --reference-query: the reference's own BVH (random-axis median split) and pairwise-combine traversal\n"
            "  --pieces N: render the samples in N pieces into sums that stay on the GPUs (progress line per piece)\n");
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
        else if (a == "--reference-query") o->flags |= PT_FLAG_REFERENCE_QUERY;
        else if (a == "--rng-skip" && next(&v)) o->rngSkip = (uint32_t)strtoul(v, nullptr, 10);
        else if (a == "--pieces" && next(&v)) o->pieces = (uint32_t)strtoul(v, nullptr, 10);
        else if (a == "--out" && next(&v)) o->out = v;
        else { fprintf(stderr, "pt_render: bad argument '%s'\n", a.c_str()); return false; }
    }
    return o->width > 0 && o->height > 0 && o->spp > 0 && o->gpus > 0 && o->pieces > 0;
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

    // ---- the node: one context + stream (+ RCCL communicator) per GPU; upload = BVH build (src/main.cpp:402)
    const auto tUpload = std::chrono::steady_clock::now();
    pt_node* node = nullptr;
    int st = pt_node_create(o.gpus, &node);
    if (st != PT_OK) { fprintf(stderr, "pt_render: cannot set up %d GPU(s): %s\n", o.gpus, pt_hip_strerror(st)); return 1; }
    st = pt_node_upload_scene(node, &d);
    if (st != PT_OK) { fprintf(stderr, "pt_render: %s (%s)\n", pt_hip_strerror(st), pt_node_last_error(node)); return 1; }
    if (o.flags & PT_FLAG_REFERENCE_QUERY)
        for (int g = 0; g < o.gpus; ++g)
        {
            st = pt_hip_build_reference_tree(pt_node_ctx(node, g), o.rngSkip);   // (its error text lives in the context, not the node)
            if (st != PT_OK) { fprintf(stderr, "pt_render: device %d: %s (%s)\n", g, pt_hip_strerror(st), pt_hip_last_error(pt_node_ctx(node, g))); return 1; }
        }
    printf("Upload + BVH build on %d GPU(s): %.1f ms\n", o.gpus, ms_since(tUpload));

    // ---- the "PathTracing" span (src/main.cpp:412-481): every device renders its slice of the samples, one RCCL
    // all-reduce joins the f32 sums, device 0 resolves and packs; 4 bytes per pixel come back
    pt_params p{};
    p.width = o.width; p.height = o.height; p.max_depth = o.depth; p.flags = o.flags; p.seed = o.seed; p.spp_begin = 0; p.spp_end = o.spp;
    std::vector<uint32_t> bgra((size_t)o.width * o.height);
    st = pt_node_reserve(node, o.width, o.height);   // the image is allocated before the span, as in the reference (src/main.cpp:385)
    if (st != PT_OK) { fprintf(stderr, "pt_render: %s (%s)\n", pt_hip_strerror(st), pt_node_last_error(node)); return 1; }
    const auto tRender = std::chrono::steady_clock::now();
    if (o.pieces > 1)
    {
        // the reference prints "Progress" from its pool callback as blocks finish (src/main.cpp:321-332); here a piece of
        // the samples is what finishes: each is added to the sums resident on the GPUs, the last snapshot is the frame
        if (o.pieces > o.spp) o.pieces = o.spp;
        for (uint32_t k = 0; k < o.pieces && st == PT_OK; ++k)
        {
            pt_params q = p;
            q.flags |= PT_FLAG_ACCUMULATE;
            pt_spp_slice(0, o.spp, o.pieces, k, &q.spp_begin, &q.spp_end);
            st = pt_node_render(node, &q, nullptr, nullptr);
            uint64_t done = 0;
            if (st == PT_OK) st = pt_node_snapshot(node, nullptr, bgra.data(), &done);
            if (st == PT_OK) printf("Progress: %.2f%% (%llu of %u spp)\n", 100.0 * (double)done / (double)o.spp, (unsigned long long)done, o.spp);
        }
    }
    else
        st = pt_node_render(node, &p, nullptr, bgra.data());
    const double renderMs = ms_since(tRender);
    if (st != PT_OK) { fprintf(stderr, "pt_render: %s (%s)\n", pt_hip_strerror(st), pt_node_last_error(node)); return 1; }
    uint64_t samples = 0, segments = 0; double kernelMs = 0;
    for (int g = 0; g < o.gpus; ++g)
    {
        pt_stats s{};
        if (pt_hip_last_stats(pt_node_ctx(node, g), &s) != PT_OK) continue;
        samples += s.samples; segments += s.segments; if (s.kernel_ms > kernelMs) kernelMs = s.kernel_ms;
    }
    // (with --pieces the statistics are the last piece's; the rate is over the whole frame)
    printf("PathTracing: Time elapsed: %.3f ms  (%ux%u, %u spp, depth %u: %.1f Msamples/s; kernel %.3f ms%s, %.3f segments/sample)\n",
           renderMs, o.width, o.height, o.spp, o.depth, (double)o.width * o.height * o.spp / renderMs / 1e3, kernelMs, o.pieces > 1 ? " (last piece)" : "",
           samples ? (double)segments / (double)samples : 0.0);

    // ---- output (src/main.cpp:487)
    const auto tWrite = std::chrono::steady_clock::now();
    st = pt_bmp_write_bgra(o.out.c_str(), bgra.data(), o.width, o.height);
    if (st != PT_OK) { fprintf(stderr, "pt_render: cannot write %s: %s\n", o.out.c_str(), pt_hip_strerror(st)); return 1; }
    printf("Wrote %s (%.1f ms)\n", o.out.c_str(), ms_since(tWrite));
    pt_node_destroy(node);
    pt_scene_free(scene);
    return 0;
}
