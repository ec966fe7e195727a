// pt_guard.h -- what keeps C++ failures inside the library (include/pt_hip.h: "nothing aborts or throws across the
// boundary"; the reference is built /EHa- /GR-, src/build.bat:7, and reports trouble through return values only,
// src/render_world.cpp:85-86).
//
//   guard_status / guard_ptr : every extern "C" entry point runs its body through one of these -- a std::bad_alloc from a
//                              vector that cannot grow becomes PT_ERR_NO_MEMORY, any other exception PT_ERR_INTERNAL
//                              (NULL for the entry points that return a handle); nothing unwinds into the caller's frames.
//   run_workers              : fn(w) for w in [0, n) on n threads.  A thread that cannot be started (std::system_error:
//                              the process limit of a GPU box, RLIMIT_NPROC) does not end the process: its share runs on
//                              the calling thread, down to n = 1 = a serial loop.
// Host-only.
#pragma once
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <exception>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/pt_hip.h"

namespace ptg {

inline void set_error(std::string* err, const char* what) noexcept
{
    if (!err) return;
    try { *err = what; } catch (...) { } // (the message itself may not fit: the status still says what happened)
}

template <class F> inline int guard_status(std::string* err, const char* where, F&& body) noexcept
{
    try { return body(); }
    catch (const std::bad_alloc&) { set_error(err, where); return PT_ERR_NO_MEMORY; }
    catch (const std::exception& e)
    {
        try { if (err) *err = std::string(where) + ": " + e.what(); } catch (...) { }
        return PT_ERR_INTERNAL;
    }
    catch (...) { set_error(err, where); return PT_ERR_INTERNAL; }
}

template <class T, class F> inline T* guard_ptr(F&& body) noexcept
{
    try { return body(); }
    catch (...) { return nullptr; }
}

// Fault injection for the CPU tests: PT_TEST_SPAWN_LIMIT=k lets only k more threads start in this
// process, every later start fails the way a refused pthread_create does.  Unset (always, outside tests/): no limit.
inline bool spawn_allowed()
{
    static long budget = [] { const char* e = getenv("PT_TEST_SPAWN_LIMIT"); return e ? atol(e) : -1L; }();
    if (budget < 0) return true;
    if (budget == 0) return false;
    --budget;
    return true;
}

template <class F> inline unsigned run_workers(unsigned n, F&& fn)
{
    if (n <= 1) { fn(0u); return 1u; }
    // an exception inside a share (a vector that cannot grow) is carried back to the caller, whichever thread met it
    std::vector<std::exception_ptr> failed(n);
    auto share = [&fn, &failed](unsigned w) noexcept { try { fn(w); } catch (...) { failed[w] = std::current_exception(); } };
    std::vector<std::thread> th;
    unsigned started = 0;
    try
    {
        th.reserve(n - 1);
        for (unsigned w = 1; w < n; ++w)
        {
            if (!spawn_allowed()) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
            th.emplace_back(share, w);
            ++started;
        }
    }
    catch (const std::system_error&) { }
    catch (const std::bad_alloc&) { }
    // the calling thread takes share 0 and every share whose thread did not start
    share(0u);
    for (unsigned w = started + 1; w < n; ++w) share(w);
    for (std::thread& t : th) t.join();
    for (const std::exception_ptr& e : failed) if (e) std::rethrow_exception(e);
    return started + 1;
}

// Worker threads a host-side pass may use: the CPUs this process may run on (its affinity mask), further capped by the
// cgroup CPU quota when one is set (a GPU box hands one GPU's job a share of the machine: threads beyond it are only
// throttled) and by `cap`; PT_BUILD_THREADS overrides (tests / tuning).
inline unsigned host_threads(unsigned cap)
{
    if (const char* e = getenv("PT_BUILD_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) return (unsigned)v; }
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (n == 0 || (unsigned)c < n)) n = (unsigned)c; }
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r"))
    {
        char q[32] = {0}; double period = 0;
        if (fscanf(f, "%31s %lf", q, &period) == 2 && q[0] != 'm' && period > 0)
        {
            const unsigned quota = (unsigned)(atof(q) / period + 0.5);
            if (quota >= 1 && quota < n) n = quota;
        }
        fclose(f);
    }
    if (n == 0) n = 1;
    return n > cap ? cap : n;
}

} // namespace ptg
