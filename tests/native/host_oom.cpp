// host_oom.cpp -- fault injection at the C-ABI (include/pt_hip.h: "nothing aborts or throws across the boundary"; the
// reference is built /EHa-, src/build.bat:7, and reports trouble through return values, src/render_world.cpp:85-86).
// Links libpt_hip.so and replaces the global operator new: allocation number k of a call fails with std::bad_alloc, for
// k = 0, 1, 2, ... until the call succeeds.  Every failed run must come back with a status (PT_ERR_NO_MEMORY) or a NULL
// handle -- never std::terminate.  Also: PT_TEST_SPAWN_LIMIT (pt_guard.h) refuses worker threads; the multi-threaded
// builder must finish on the calling thread with the same tree.  Host-only: no device is touched (pt_scene_prepare).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <new>

#include "../../include/pt_hip.h"

static std::atomic<long> g_countdown{-1}; // < 0: allocations succeed; else the allocation that finds 0 here fails (and all after it)
static std::atomic<long> g_failures{0};

static void* counted_alloc(size_t n)
{
    long c = g_countdown.load();
    if (c >= 0)
    {
        if (c == 0) { g_failures.fetch_add(1); throw std::bad_alloc(); }
        g_countdown.store(c - 1);
    }
    void* p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new(size_t n) { return counted_alloc(n); }
void* operator new[](size_t n) { return counted_alloc(n); }
void* operator new(size_t n, const std::nothrow_t&) noexcept { try { return counted_alloc(n); } catch (...) { return nullptr; } }
void* operator new[](size_t n, const std::nothrow_t&) noexcept { try { return counted_alloc(n); } catch (...) { return nullptr; } }
void operator delete(void* p) noexcept { free(p); }
void operator delete[](void* p) noexcept { free(p); }
void operator delete(void* p, size_t) noexcept { free(p); }
void operator delete[](void* p, size_t) noexcept { free(p); }

int main()
{
    int bad = 0;
    // ---- pt_scene_builtin: every failing allocation -> NULL
    long needed = -1;
    for (long k = 0; k < 200000; ++k)
    {
        g_countdown.store(k);
        pt_scene* s = pt_scene_builtin(1, 16.0f / 9.0f, 0);
        g_countdown.store(-1);
        if (s) { needed = k; pt_scene_free(s); break; }
    }
    printf("pt_scene_builtin(1): succeeded once %ld allocations were allowed; %ld injected failures came back as NULL\n", needed, g_failures.load());
    if (needed <= 0) ++bad;

    pt_scene* s = pt_scene_builtin(2, 16.0f / 9.0f, 0); // 2400 spheres + 3 planes
    pt_scene_desc d{};
    if (!s || pt_scene_get_desc(s, &d) != PT_OK) { printf("no scene\n"); return 1; }
    // ---- pt_scene_prepare (tree build, widening, LDS images): status, never terminate
    long prepNeeded = -1, noMem = 0, other = 0;
    for (long k = 0; k < 200000; ++k)
    {
        pt_prepared* p = nullptr;
        g_countdown.store(k);
        const int st = pt_scene_prepare(&d, &p);
        g_countdown.store(-1);
        if (st == PT_OK) { prepNeeded = k; pt_prepared_free(p); break; }
        if (p) { printf("a failed pt_scene_prepare returned a handle\n"); ++bad; }
        if (st == PT_ERR_NO_MEMORY) ++noMem; else ++other;
    }
    printf("pt_scene_prepare(scene 2): succeeded once %ld allocations were allowed; %ld failures returned PT_ERR_NO_MEMORY, %ld another status\n", prepNeeded, noMem, other);
    if (prepNeeded <= 0 || noMem == 0 || other != 0) ++bad;
    // ---- pt_reference_bvh: 0 nodes on failure
    {
        g_countdown.store(0);
        const uint32_t nn = pt_reference_bvh(&d, 0, 0, nullptr, nullptr, nullptr, nullptr);
        g_countdown.store(-1);
        if (nn != 0) { printf("pt_reference_bvh ignored a failed allocation\n"); ++bad; }
        if (pt_reference_bvh(&d, 0, 0, nullptr, nullptr, nullptr, nullptr) != 2 * d.n_spheres - 1) ++bad;
    }
    // ---- pt_bvh_selfcheck under failure: a status
    {
        uint64_t out8[8];
        g_countdown.store(3);
        const int st = pt_bvh_selfcheck(&d, out8);
        g_countdown.store(-1);
        if (st != PT_ERR_NO_MEMORY) { printf("pt_bvh_selfcheck under a failed allocation: status %d\n", st); ++bad; }
        if (pt_bvh_selfcheck(&d, out8) != PT_OK) ++bad;
    }
    pt_scene_free(s);
    // ---- worker threads that cannot start (PT_TEST_SPAWN_LIMIT, read once per process: set by the test before exec):
    // the 70 000-sphere build asks for several workers and must get through with what it is given
    {
        pt_scene* big = pt_scene_builtin(5, 16.0f / 9.0f, 70000);
        pt_scene_desc e{};
        uint64_t out8[8] = {0};
        if (!big || pt_scene_get_desc(big, &e) != PT_OK) { printf("no big scene\n"); return 1; }
        const int st = pt_bvh_selfcheck(&e, out8);
        printf("pt_bvh_selfcheck(70000 spheres, PT_TEST_SPAWN_LIMIT=%s, PT_BUILD_THREADS=%s): status %d, %llu nodes, %llu spheres in leaves\n",
               getenv("PT_TEST_SPAWN_LIMIT") ? getenv("PT_TEST_SPAWN_LIMIT") : "-", getenv("PT_BUILD_THREADS") ? getenv("PT_BUILD_THREADS") : "-", st,
               (unsigned long long)out8[0], (unsigned long long)out8[4]);
        if (st != PT_OK || out8[4] != 70000) ++bad;
        const uint64_t before = pt_tree_build_count();
        pt_prepared* p = nullptr;
        if (pt_scene_prepare(&e, &p) != PT_OK) ++bad;
        if (pt_tree_build_count() != before + 1) { printf("pt_tree_build_count did not advance by one\n"); ++bad; }
        pt_prepared_free(p);
        pt_scene_free(big);
    }
    printf(bad ? "host_oom: FAILED (%d)\n" : "host_oom: ok\n", bad);
    return bad ? 1 : 0;
}
