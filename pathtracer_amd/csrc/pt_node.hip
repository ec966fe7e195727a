// pt_node.hip -- the multi-GPU entry points of include/pt_hip.h (pt_node_*, pt_hip_render_node): one process, one
// pt_ctx + stream + RCCL communicator per GPU of the node.
//
// Takes the place of the reference's thread-pool submit + join (src/main.cpp:448-477) at node scale: the
// samples of every pixel are split into contiguous slices, GPU g renders slice g of ALL pixels into its own f32 sum
// buffer (pt_hip_render_device: asynchronous, so one host thread enqueues on every GPU in turn), ONE
// ncclAllReduce(sum, f32) of that buffer over xGMI joins them (33 MB at 1080p, 133 MB at 4K: a ring moves 2 (G-1)/G of
// it per GPU, ~0.4 / 1.5 ms at ~153 GB/s per link against >= 100 ms of rendering), then device 0 resolves (clamp(mean),
// gamma, BGRA pack) and only what the caller asked for crosses PCIe.  No data-path exchange anywhere else.
//
// RCCL is bound at run time (dlopen) on the first pt_node_create with more than one GPU, so libpt_hip.so keeps linking
// against the HIP runtime alone and a process that already holds an RCCL (PyTorch-ROCm ships one) shares it.
// Host-only code: no kernels here.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pt_hip.h"
#include "pt_guard.h"

namespace {

struct Rccl
{
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
    bool load()
    {
        if (lib) return true;
        // a copy that is already in the process first (same communicator library as the rest of the process)
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names) if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { err = std::string("RCCL not found: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) { err = "RCCL lacks a required symbol"; lib = nullptr; return false; }
        return true;
    }
};
Rccl g_rccl;

} // namespace

struct pt_node
{
    int n = 0;
    std::vector<pt_ctx*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<float*> accum;      // per GPU: width*height*4 sums (grown on demand, kept)
    size_t accum_floats = 0;
    // Progressive rendering (PT_FLAG_ACCUMULATE): accum[g] then holds GPU g's RUNNING sums over the pieces rendered so
    // far; the join happens in pt_node_snapshot, out of place into snap[g], so the running sums stay per-GPU partials.
    std::vector<float*> snap; size_t snap_floats = 0;
    uint32_t res_w = 0, res_h = 0; uint64_t res_spp = 0;   // frame shape and samples per pixel of the resident sums (0 = none)
    // The packed image: PINNED host memory that device 0 writes straight over PCIe (k_resolve), copied on to the caller's
    // buffer by the CPU.  A hipMemcpyAsync into the caller's (pageable) memory took its place until round 3: the runtime
    // sets that copy path up on first use -- 9 ms inside the one frame a command-line render times (profiles/README.md).
    uint32_t* bgra_pin = nullptr; uint32_t* bgra_pin_dev = nullptr; size_t bgra_pixels = 0;
    std::vector<ncclComm_t> comm;   // empty with one GPU
    bool force_collective = false;  // PT_NODE_FORCE_COLLECTIVE=1: take the all-reduce path with one GPU too (rehearsal)
    std::string err;
};

namespace {

#define NODE_HIP(node, call)                                                                                   \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) { (node)->err = std::string(#call) + ": " + hipGetErrorString(e_); return PT_ERR_HIP; } \
    } while (0)
#define NODE_NCCL(node, call)                                                                                  \
    do {                                                                                                        \
        ncclResult_t r_ = (call);                                                                               \
        if (r_ != ncclSuccess) { (node)->err = std::string(#call) + ": " + g_rccl.GetErrorString(r_); return PT_ERR_HIP; } \
    } while (0)

#define NODE_ENTER(nd, name) return ptg::guard_status((nd) ? &(nd)->err : nullptr, name, [&]() -> int {
#define NODE_LEAVE });

struct DeviceScope // make a device current for a few calls
{
    int prev = -1;
    explicit DeviceScope(int d) { (void)hipGetDevice(&prev); (void)hipSetDevice(d); }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int node_buffers(pt_node* nd, uint32_t w, uint32_t h)
{
    const size_t floats = (size_t)w * h * 4;
    if (floats > nd->accum_floats)
    {
        for (int g = 0; g < nd->n; ++g)
        {
            DeviceScope s(g);
            if (nd->accum[(size_t)g]) { (void)hipFree(nd->accum[(size_t)g]); nd->accum[(size_t)g] = nullptr; }
            NODE_HIP(nd, hipMalloc((void**)&nd->accum[(size_t)g], floats * sizeof(float)));
        }
        nd->accum_floats = floats;
    }
    if (nd->snap_floats != 0 && floats > nd->snap_floats) // (re-grown by pt_node_snapshot when it is next needed)
    {
        for (int g = 0; g < nd->n; ++g) { DeviceScope s(g); if (nd->snap[(size_t)g]) { (void)hipFree(nd->snap[(size_t)g]); nd->snap[(size_t)g] = nullptr; } }
        nd->snap_floats = 0;
    }
    if ((size_t)w * h > nd->bgra_pixels)
    {
        DeviceScope s(0);
        if (nd->bgra_pin) { (void)hipHostFree(nd->bgra_pin); nd->bgra_pin = nullptr; nd->bgra_pin_dev = nullptr; nd->bgra_pixels = 0; }
        NODE_HIP(nd, hipHostMalloc((void**)&nd->bgra_pin, (size_t)w * h * sizeof(uint32_t), hipHostMallocDefault));
        NODE_HIP(nd, hipHostGetDevicePointer((void**)&nd->bgra_pin_dev, nd->bgra_pin, 0));
        nd->bgra_pixels = (size_t)w * h;
    }
    return PT_OK;
}

} // namespace

extern "C" {

void pt_spp_slice(uint32_t spp_begin, uint32_t spp_end, uint32_t n_parts, uint32_t part, uint32_t* out_begin, uint32_t* out_end)
{
    const uint64_t total = spp_end > spp_begin ? (uint64_t)(spp_end - spp_begin) : 0ull;
    const uint64_t n = n_parts ? n_parts : 1u, g = part < n ? part : n - 1;
    if (out_begin) *out_begin = spp_begin + (uint32_t)(total * g / n);
    if (out_end) *out_end = spp_begin + (uint32_t)(total * (g + 1) / n);
}

const char* pt_node_last_error(pt_node* nd) { return nd ? nd->err.c_str() : "null node"; }
int pt_node_gpu_count(const pt_node* nd) { return nd ? nd->n : 0; }
pt_ctx* pt_node_ctx(pt_node* nd, int gpu) { return (nd && gpu >= 0 && gpu < nd->n) ? nd->ctx[(size_t)gpu] : nullptr; }

void pt_node_destroy(pt_node* nd)
{
    if (!nd) return;
    for (ncclComm_t c : nd->comm) if (c) (void)g_rccl.CommDestroy(c);
    for (int g = 0; g < nd->n; ++g)
    {
        DeviceScope s(g);
        if ((size_t)g < nd->accum.size() && nd->accum[(size_t)g]) (void)hipFree(nd->accum[(size_t)g]);
        if ((size_t)g < nd->snap.size() && nd->snap[(size_t)g]) (void)hipFree(nd->snap[(size_t)g]);
        if (g == 0 && nd->bgra_pin) (void)hipHostFree(nd->bgra_pin);
        if ((size_t)g < nd->stream.size() && nd->stream[(size_t)g]) (void)hipStreamDestroy(nd->stream[(size_t)g]);
        if ((size_t)g < nd->ctx.size() && nd->ctx[(size_t)g]) pt_hip_destroy(nd->ctx[(size_t)g]);
    }
    delete nd;
}

int pt_node_create(int n_gpus, pt_node** out)
{
    if (!out) return PT_ERR_INVALID;
    *out = nullptr;
    if (n_gpus < 1) return PT_ERR_INVALID;
    if (n_gpus > pt_hip_device_count()) return PT_ERR_NO_DEVICE;
    pt_node* nd = new (std::nothrow) pt_node();
    if (!nd) return PT_ERR_NO_MEMORY;
    const int created = ptg::guard_status(nullptr, "pt_node_create", [&]() -> int {
    nd->n = n_gpus;
    nd->ctx.assign((size_t)n_gpus, nullptr); nd->stream.assign((size_t)n_gpus, nullptr); nd->accum.assign((size_t)n_gpus, nullptr);
    if (const char* e = getenv("PT_NODE_FORCE_COLLECTIVE")) nd->force_collective = e[0] == '1';
    int st = PT_OK;
    for (int g = 0; g < n_gpus && st == PT_OK; ++g)
    {
        st = pt_hip_create(g, &nd->ctx[(size_t)g]);
        if (st != PT_OK) break;
        DeviceScope s(g);
        if (hipStreamCreateWithFlags(&nd->stream[(size_t)g], hipStreamNonBlocking) != hipSuccess) st = PT_ERR_HIP;
    }
    if (st == PT_OK && (n_gpus > 1 || nd->force_collective))
    {
        if (!g_rccl.load()) { fprintf(stderr, "pt_node_create: %s\n", g_rccl.err.c_str()); st = PT_ERR_UNSUPPORTED; }
        else
        {
            nd->comm.assign((size_t)n_gpus, nullptr);
            std::vector<int> devs((size_t)n_gpus);
            for (int g = 0; g < n_gpus; ++g) devs[(size_t)g] = g;
            const ncclResult_t r = g_rccl.CommInitAll(nd->comm.data(), n_gpus, devs.data());
            if (r != ncclSuccess) { fprintf(stderr, "pt_node_create: ncclCommInitAll: %s\n", g_rccl.GetErrorString(r)); nd->comm.clear(); st = PT_ERR_HIP; }
        }
    }
    return st;
    });
    if (created != PT_OK) { pt_node_destroy(nd); return created; }
    *out = nd;
    return PT_OK;
}

int pt_node_upload_scene(pt_node* nd, const pt_scene_desc* scene)
{
    if (!nd || !scene) return PT_ERR_INVALID;
    NODE_ENTER(nd, "pt_node_upload_scene")
    // ONE host-side build (the reference builds its tree once per process, src/main.cpp:402), then one commit per GPU:
    // allocations + copies, each on a thread of its own (a thread that cannot start runs on this one, pt_guard.h).
    nd->res_spp = 0;
    pt_prepared* prep = nullptr;
    int st = pt_scene_prepare(scene, &prep);
    if (st != PT_OK) { nd->err = std::string("pt_scene_prepare: ") + pt_hip_strerror(st); return st; }
    struct Free { pt_prepared* p; ~Free() { pt_prepared_free(p); } } guard{prep};
    std::vector<int> status((size_t)nd->n, PT_OK);
    ptg::run_workers((unsigned)nd->n, [&](unsigned g) { status[g] = pt_hip_commit_scene(nd->ctx[g], prep); });
    for (int g = 0; g < nd->n; ++g)
        if (status[(size_t)g] != PT_OK) { nd->err = std::string("device ") + std::to_string(g) + ": " + pt_hip_last_error(nd->ctx[(size_t)g]); return status[(size_t)g]; }
    return PT_OK;
    NODE_LEAVE
}

namespace {

// An error path's last step: nothing of the node stays in flight behind a failed call.
int node_fail(pt_node* nd, int st)
{
    for (int g = 0; g < nd->n; ++g) { DeviceScope s(g); (void)hipStreamSynchronize(nd->stream[(size_t)g]); }
    nd->res_spp = 0; // (whatever the sum buffers hold now is not a frame)
    return st;
}
#define NODE_TRY_HIP(node, call)                                                                               \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) { (node)->err = std::string(#call) + ": " + hipGetErrorString(e_); return node_fail(node, PT_ERR_HIP); } \
    } while (0)

// One all-reduce(sum, f32) of `floats` words over the node's communicators, rank g from src[g] to dst[g] on its stream.
int node_all_reduce(pt_node* nd, const std::vector<float*>& src, const std::vector<float*>& dst, size_t floats)
{
    ncclResult_t r = g_rccl.GroupStart();
    if (r != ncclSuccess) { nd->err = std::string("ncclGroupStart: ") + g_rccl.GetErrorString(r); return PT_ERR_HIP; }
    ncclResult_t bad = ncclSuccess;
    for (int g = 0; g < nd->n && bad == ncclSuccess; ++g)
        bad = g_rccl.AllReduce(src[(size_t)g], dst[(size_t)g], floats, ncclFloat32, ncclSum, nd->comm[(size_t)g], nd->stream[(size_t)g]);
    r = g_rccl.GroupEnd();
    if (bad != ncclSuccess) { nd->err = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(bad); return PT_ERR_HIP; }
    if (r != ncclSuccess) { nd->err = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r); return PT_ERR_HIP; }
    return PT_OK;
}

// Device 0: resolve + pack `sums` by `spp`, then only what was asked for crosses PCIe; every stream drained on return.
int node_read_out(pt_node* nd, const float* sums0, uint32_t w, uint32_t h, uint64_t spp, float* accum_rgba_host, uint32_t* bgra_host)
{
    const size_t floats = (size_t)w * h * 4;
    {
        DeviceScope s(0);
        if (bgra_host)
        {
            const int st = pt_hip_resolve_device(nd->ctx[0], sums0, w * h, (uint32_t)spp, nullptr, nd->bgra_pin_dev, nd->stream[0]);
            if (st != PT_OK) { nd->err = std::string("device 0: ") + pt_hip_last_error(nd->ctx[0]); return node_fail(nd, st); }
        }
        if (accum_rgba_host)
            NODE_TRY_HIP(nd, hipMemcpyAsync(accum_rgba_host, sums0, floats * sizeof(float), hipMemcpyDeviceToHost, nd->stream[0]));
    }
    for (int g = 0; g < nd->n; ++g)
    {
        DeviceScope s(g);
        NODE_TRY_HIP(nd, hipStreamSynchronize(nd->stream[(size_t)g]));
    }
    if (bgra_host) memcpy(bgra_host, nd->bgra_pin, (size_t)w * h * sizeof(uint32_t));
    return PT_OK;
}

} // namespace

int pt_node_render(pt_node* nd, const pt_params* p, float* accum_rgba_host, uint32_t* bgra_host)
{
    if (!nd || !p || p->width == 0 || p->height == 0 || p->spp_end <= p->spp_begin) return PT_ERR_INVALID;
    NODE_ENTER(nd, "pt_node_render")
    nd->err.clear();
    const bool addTo = (p->flags & PT_FLAG_ACCUMULATE) != 0;
    // A piece of a progressive render stays on the GPUs (pt_node_snapshot reads out): no host outputs, one frame shape.
    if (addTo && (accum_rgba_host || bgra_host)) { nd->err = "PT_FLAG_ACCUMULATE: the sums stay resident, read them with pt_node_snapshot (pass NULL outputs)"; return PT_ERR_INVALID; }
    if (addTo && nd->res_spp != 0 && (nd->res_w != p->width || nd->res_h != p->height)) { nd->err = "PT_FLAG_ACCUMULATE: frame shape differs from the resident sums' (pt_node_reset first)"; return PT_ERR_INVALID; }
    int st = node_buffers(nd, p->width, p->height);
    if (st != PT_OK) return st;
    const uint32_t spp = p->spp_end - p->spp_begin;
    const int active = (uint32_t)nd->n <= spp ? nd->n : (int)spp;      // a GPU without a sample contributes a zero buffer
    const size_t floats = (size_t)p->width * p->height * 4;
    const bool first = addTo && nd->res_spp == 0;                        // the first piece starts from zero sums
    if (!addTo) nd->res_spp = 0;                                         // (a plain render overwrites the buffers)
    // every GPU renders its slice of the samples (asynchronous launches: one host thread serves them all)
    for (int g = 0; g < nd->n; ++g)
    {
        DeviceScope s(g);
        if (first || (!addTo && g >= active))
            NODE_TRY_HIP(nd, hipMemsetAsync(nd->accum[(size_t)g], 0, floats * sizeof(float), nd->stream[(size_t)g]));
        if (g < active)
        {
            pt_params q = *p;
            pt_spp_slice(p->spp_begin, p->spp_end, (uint32_t)active, (uint32_t)g, &q.spp_begin, &q.spp_end);
            st = pt_hip_render_device(nd->ctx[(size_t)g], &q, nd->accum[(size_t)g], nd->stream[(size_t)g]);
            if (st != PT_OK) { nd->err = std::string("device ") + std::to_string(g) + ": " + pt_hip_last_error(nd->ctx[(size_t)g]); return node_fail(nd, st); }
        }
    }
    if (addTo)
    {
        nd->res_w = p->width; nd->res_h = p->height; nd->res_spp += spp;
        return PT_OK;   // asynchronous: the next piece queues behind this one on every GPU's stream
    }
    // the join: one all-reduce of the f32 sums, in place, each rank on its own stream behind its render
    if (!nd->comm.empty())
    {
        st = node_all_reduce(nd, nd->accum, nd->accum, floats);
        if (st != PT_OK) return node_fail(nd, st);
    }
    return node_read_out(nd, nd->accum[0], p->width, p->height, spp, accum_rgba_host, bgra_host);
    NODE_LEAVE
}

int pt_node_snapshot(pt_node* nd, float* accum_rgba_host, uint32_t* bgra_host, uint64_t* spp_out)
{
    if (!nd) return PT_ERR_INVALID;
    NODE_ENTER(nd, "pt_node_snapshot")
    nd->err.clear();
    if (spp_out) *spp_out = nd->res_spp;
    if (nd->res_spp == 0) { nd->err = "pt_node_snapshot: no resident sums (render with PT_FLAG_ACCUMULATE first)"; return PT_ERR_NO_SCENE; }
    if (nd->res_spp > 0xFFFFFFFFull) return PT_ERR_INVALID;
    const uint32_t w = nd->res_w, h = nd->res_h;
    const size_t floats = (size_t)w * h * 4;
    const float* sums0 = nd->accum[0];
    if (!nd->comm.empty())
    {
        if (floats > nd->snap_floats)
        {
            if (nd->snap.empty()) nd->snap.assign((size_t)nd->n, nullptr);
            for (int g = 0; g < nd->n; ++g)
            {
                DeviceScope s(g);
                if (nd->snap[(size_t)g]) { (void)hipFree(nd->snap[(size_t)g]); nd->snap[(size_t)g] = nullptr; }
                NODE_HIP(nd, hipMalloc((void**)&nd->snap[(size_t)g], floats * sizeof(float)));
            }
            nd->snap_floats = floats;
        }
        const uint64_t keep = nd->res_spp;
        const int st = node_all_reduce(nd, nd->accum, nd->snap, floats);
        if (st != PT_OK) return node_fail(nd, st);
        nd->res_spp = keep;
        sums0 = nd->snap[0];
    }
    const uint64_t keep = nd->res_spp;
    const int st = node_read_out(nd, sums0, w, h, nd->res_spp, accum_rgba_host, bgra_host);
    if (st == PT_OK) nd->res_spp = keep;
    return st;
    NODE_LEAVE
}

int pt_node_reserve(pt_node* nd, uint32_t width, uint32_t height)
{
    if (!nd || width == 0 || height == 0) return PT_ERR_INVALID;
    NODE_ENTER(nd, "pt_node_reserve")
    nd->err.clear();
    return node_buffers(nd, width, height);
    NODE_LEAVE
}

int pt_node_reset(pt_node* nd)
{
    if (!nd) return PT_ERR_INVALID;
    for (int g = 0; g < nd->n; ++g) { DeviceScope s(g); (void)hipStreamSynchronize(nd->stream[(size_t)g]); }
    nd->res_spp = 0; nd->res_w = 0; nd->res_h = 0;
    return PT_OK;
}

int pt_hip_render_node(const pt_scene_desc* scene, const pt_params* params, int n_gpus, float* accum_rgba_host)
{
    if (!scene || !params || !accum_rgba_host) return PT_ERR_INVALID;
    if (params->flags & PT_FLAG_ACCUMULATE) return PT_ERR_INVALID; // (the node does not outlive this call: nothing to add to)
    pt_node* nd = nullptr;
    int st = pt_node_create(n_gpus, &nd);
    if (st != PT_OK) return st;
    st = pt_node_upload_scene(nd, scene);
    if (st == PT_OK) st = pt_node_render(nd, params, accum_rgba_host, nullptr);
    if (st != PT_OK) fprintf(stderr, "pt_hip_render_node: %s\n", pt_node_last_error(nd));
    pt_node_destroy(nd);
    return st;
}

} // extern "C"
