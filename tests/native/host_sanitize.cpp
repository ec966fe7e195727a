// host_sanitize.cpp -- the host-side passes of libpt_hip.so under AddressSanitizer + UBSan (the
// reference has asserts on these paths, src/main.cpp:95-97; here they run instrumented).  Built by `make asan` from
// pt_world.cpp + pt_bvh.h with g++ -fsanitize=address,undefined (host code only, no HIP); run by
// tests/test_host_hardening.py.  Exit status 0 = every check held and the sanitizers saw nothing.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/pt_hip.h"
#include "../../pathtracer_amd/csrc/pt_bvh.h"

static int g_failed = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_failed; } } while (0)

// every sphere under exactly one leaf reference of the binary tree; every wide flavour names each sphere once
static void check_tree(const pt_scene_desc& d, const char* what)
{
    pth::BvhData bv;
    pth::build_bvh(&d, &bv);
    CHECK(bv.n_spheres == d.n_spheres);
    CHECK(bv.sph.size() == 4 * (size_t)d.n_spheres);
    pth::NodeGrid grid{};
    std::vector<uint32_t> qn;
    pth::quantise_nodes(bv, &grid, &qn);
    CHECK(qn.size() == 8 * (size_t)bv.n_nodes);
    uint64_t out8[8];
    for (int single = 0; single < 2; ++single)
    {
        pth::NodeGrid wg{}; pth::WideData wd;
        pth::widen_nodes(bv, &d, single != 0, &wg, &wd);
        CHECK(wd.nodes.size() == 16 * (size_t)wd.n_nodes);
        std::vector<uint32_t> seen(d.n_spheres, 0u);
        std::vector<uint32_t> stack;
        auto leaf = [&](uint32_t ref) { const uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 7u) + 1u; for (uint32_t i = 0; i < count; ++i) { CHECK(first + i < d.n_spheres); if (first + i < d.n_spheres) ++seen[first + i]; } };
        if (d.n_spheres == 0) continue;
        if (wd.n_nodes == 0) leaf(wd.root_ref); else stack.push_back(wd.root_ref);
        uint64_t visited = 0;
        while (!stack.empty())
        {
            const uint32_t node = stack.back(); stack.pop_back();
            CHECK(node < wd.n_nodes);
            if (node >= wd.n_nodes) break;
            ++visited;
            const uint32_t* q = wd.nodes.data() + 16 * (size_t)node;
            for (int j = 0; j < 4; ++j)
            {
                const uint32_t ref = q[12 + j];
                if (j > 0 && ref == q[12] && q[3 * j] == 0u && q[3 * j + 1] == 0u && q[3 * j + 2] == 0u) continue; // empty slot
                if (ref & pth::kLeafFlag) leaf(ref); else stack.push_back(ref);
            }
        }
        CHECK(wd.n_nodes == 0 || visited == wd.n_nodes);
        uint64_t once = 0;
        for (uint32_t s : seen) once += s == 1u;
        CHECK(once == d.n_spheres);
    }
    (void)out8;
    fprintf(stderr, "  tree ok: %s (%u spheres, %u nodes, depth %u)\n", what, d.n_spheres, bv.n_nodes, bv.depth);
}

static void write_file(const char* path, const std::string& text) { FILE* f = fopen(path, "w"); if (f) { fwrite(text.data(), 1, text.size(), f); fclose(f); } }

int main(int argc, char** argv)
{
    const char* tmp = argc > 1 ? argv[1] : "/tmp";
    const std::string base = std::string(tmp) + "/pt_sanitize_";
    // ---- builders, flatten, trees (single-threaded sizes, then the multi-threaded builder at 70 000 spheres)
    for (int id = 1; id <= 5; ++id)
    {
        pt_scene* s = pt_scene_builtin(id, 16.0f / 9.0f, id == 5 ? 70000u : 0u);
        CHECK(s != nullptr);
        if (!s) continue;
        pt_scene_desc d{};
        CHECK(pt_scene_get_desc(s, &d) == PT_OK);
        check_tree(d, ("builtin scene " + std::to_string(id)).c_str());
        if (id <= 3)
        {
            // the reference-topology tree (src/main.cpp:131-159): node count, every sphere once in `order`
            const uint32_t nn = 2 * d.n_spheres - 1;
            std::vector<float> boxes(6 * (size_t)nn); std::vector<int32_t> links(3 * (size_t)nn); std::vector<uint32_t> order(d.n_spheres);
            uint32_t depth = 0;
            CHECK(pt_reference_bvh(&d, 0, nn, boxes.data(), links.data(), order.data(), &depth) == nn);
            std::vector<uint32_t> seen(d.n_spheres, 0u);
            for (uint32_t o : order) { CHECK(o < d.n_spheres); if (o < d.n_spheres) ++seen[o]; }
            for (uint32_t c : seen) CHECK(c == 1u);
            CHECK(pt_reference_bvh(&d, 0, 1, boxes.data(), links.data(), nullptr, nullptr) == nn); // (too small a cap: count only, nothing written)
        }
        // scene file round trip, bit for bit
        const std::string path = base + std::to_string(id) + ".scene";
        if (id != 5)
        {
            CHECK(pt_scene_save(&d, path.c_str()) == PT_OK);
            pt_scene* l = pt_scene_load(path.c_str());
            CHECK(l != nullptr);
            if (l)
            {
                pt_scene_desc e{};
                CHECK(pt_scene_get_desc(l, &e) == PT_OK);
                CHECK(e.n_spheres == d.n_spheres && e.n_planes == d.n_planes);
                if (e.n_spheres == d.n_spheres) CHECK(memcmp(e.sphere_pos_r, d.sphere_pos_r, 16 * (size_t)d.n_spheres) == 0);
                CHECK(memcmp(e.camera, d.camera, sizeof d.camera) == 0);
                pt_scene_free(l);
            }
            remove(path.c_str());
        }
        pt_scene_free(s);
    }
    // ---- leaves of up to 8 spheres (PT_LEAF_MAX 5..8): the single-sphere widening splits them over several nodes
    for (const char* lm : {"5", "8"})
    {
        setenv("PT_LEAF_MAX", lm, 1);
        for (int id : {1, 2, 3})
        {
            pt_scene* s = pt_scene_builtin(id, 16.0f / 9.0f, 0);
            pt_scene_desc d{};
            if (s && pt_scene_get_desc(s, &d) == PT_OK) check_tree(d, (std::string("PT_LEAF_MAX=") + lm + " scene " + std::to_string(id)).c_str());
            pt_scene_free(s);
        }
        unsetenv("PT_LEAF_MAX");
    }
    // ---- degenerate scenes: none, one, all spheres in one place
    {
        pt_scene* s = pt_scene_builtin(4, 1.0f, 0);
        pt_scene_desc d{};
        pt_scene_get_desc(s, &d);
        pt_scene_desc e = d; e.n_spheres = 0; check_tree(e, "no spheres");
        e.n_spheres = 1; check_tree(e, "one sphere");
        std::vector<float> same(4 * 100, 1.0f);
        std::vector<uint32_t> ty(100, PT_MAT_DIFFUSE); std::vector<float> rgba(400, 0.5f), prm(200, 0.0f);
        e.n_spheres = 100; e.sphere_pos_r = same.data(); e.sphere_vel = nullptr; e.sphere_mat_type = ty.data(); e.sphere_mat_rgba = rgba.data(); e.sphere_mat_param = prm.data();
        check_tree(e, "100 coincident spheres");
        pt_scene_free(s);
    }
    // ---- the scene-file parser on malformed input: NULL, never a fault
    {
        const std::string good = "pt_scene 1\ntime 0 1\ncamera 0 0 0 0 0 -1 0 1 0 1 1 1 0\nplane 0 1 0 0 1 .5 .5 .5 1 0 0\nsphere 0 1 0 1 0 0 0 1 .5 .5 .5 1 0 0\n";
        const std::string path = base + "bad.scene";
        write_file(path.c_str(), good);
        pt_scene* ok = pt_scene_load(path.c_str());
        CHECK(ok != nullptr);
        pt_scene_free(ok);
        std::vector<std::string> bad;
        bad.push_back("");                                                   // empty
        bad.push_back("sphere 0 1 0 1 0 0 0 1 .5 .5 .5 1 0 0\n");            // no header
        bad.push_back("pt_scene 2\n");                                       // unknown version
        bad.push_back("pt_scene 1\ntime 0 1\n");                             // no camera
        bad.push_back(good + "sphere 0 1 0\n");                              // truncated record
        bad.push_back(good + "sphere 0 1 0 1 0 0 0 9 .5 .5 .5 1 0 0\n");     // material type out of range
        bad.push_back(good + "sphere 0 1 0 1 0 0 0 x .5 .5 .5 1 0 0\n");     // not a number
        bad.push_back(good + "wobble 1 2 3\n");                              // unknown record
        bad.push_back(good.substr(0, good.size() / 2));                      // cut in the middle of a line
        bad.push_back(good + "sphere " + std::string(5000, '1') + " 1 0 1 0 0 0 1 .5 .5 .5 1 0 0\n"); // a field longer than the line buffer
        bad.push_back(good + std::string(3000, ' ') + "sphere 0 1\n");       // over-long line whose tail is a short record
        bad.push_back("pt_scene 1\ncamera 0 0 0\n");                         // short camera
        bad.push_back(std::string("pt_scene 1\0time", 15));                  // embedded NUL
        for (size_t i = 0; i < bad.size(); ++i)
        {
            write_file(path.c_str(), bad[i]);
            pt_scene* s = pt_scene_load(path.c_str());
            if (s) { fprintf(stderr, "malformed file %zu was accepted\n", i); ++g_failed; pt_scene_free(s); }
        }
        // over-long lines that ARE well-formed up to the buffer: a comment of 5000 characters is skipped piecewise -- its tail
        // is then read as a record and must be refused, not crash
        write_file(path.c_str(), "pt_scene 1\n#" + std::string(5000, 'c') + "\n" + good.substr(11));
        pt_scene* s = pt_scene_load(path.c_str());
        pt_scene_free(s); // (either answer is acceptable; the sanitizers judge the run)
        remove(path.c_str());
        CHECK(pt_scene_load((base + "does_not_exist").c_str()) == nullptr);
        CHECK(pt_scene_load(nullptr) == nullptr);
    }
    // ---- resolve / pack / header
    {
        float sum[8] = {4, 2, 0, 4, -1, 9, 2, 4}, px[8]; uint32_t bg[2]; uint8_t hd[58];
        CHECK(pt_resolve(sum, 2, 4, px) == PT_OK && px[0] == 1.0f && px[1] == 0.5f && px[4] == 0.0f && px[5] == 1.0f);
        CHECK(pt_pack_bgra(px, 2, bg) == PT_OK && (bg[0] >> 24) == 255u);
        CHECK(pt_bmp_header(1920, 1080, hd) == PT_OK && hd[0] == 'B' && hd[1] == 'M');
        CHECK(pt_resolve(nullptr, 2, 4, px) == PT_ERR_INVALID);
    }
    if (g_failed) { fprintf(stderr, "host_sanitize: %d check(s) failed\n", g_failed); return 1; }
    fprintf(stderr, "host_sanitize: all checks held\n");
    return 0;
}
