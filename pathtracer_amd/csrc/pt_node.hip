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
    uint32_t* bgra_dev = nullptr; size_t bgra_pixels = 0;   // device 0
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
    if ((size_t)w * h > nd->bgra_pixels)
    {
        DeviceScope s(0);
        if (nd->bgra_dev) { (void)hipFree(nd->bgra_dev); nd->bgra_dev = nullptr; }
        NODE_HIP(nd, hipMalloc((void**)&nd->bgra_dev, (size_t)w * h * sizeof(uint32_t)));
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
        if (g == 0 && nd->bgra_dev) (void)hipFree(nd->bgra_dev);
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
    if (!nd) return PT_ERR_INVALID;
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
    if (st != PT_OK) { pt_node_destroy(nd); return st; }
    *out = nd;
    return PT_OK;
}

int pt_node_upload_scene(pt_node* nd, const pt_scene_desc* scene)
{
    if (!nd || !scene) return PT_ERR_INVALID;
    // (each upload builds the tree on the host and copies it: one thread per GPU, as the reference builds once per process)
    std::vector<int> status((size_t)nd->n, PT_OK);
    std::vector<std::thread> workers;
    for (int g = 0; g < nd->n; ++g) workers.emplace_back([&, g]() { status[(size_t)g] = pt_hip_upload_scene(nd->ctx[(size_t)g], scene); });
    for (std::thread& t : workers) t.join();
    for (int g = 0; g < nd->n; ++g)
        if (status[(size_t)g] != PT_OK) { nd->err = std::string("device ") + std::to_string(g) + ": " + pt_hip_last_error(nd->ctx[(size_t)g]); return status[(size_t)g]; }
    return PT_OK;
}

int pt_node_render(pt_node* nd, const pt_params* p, float* accum_rgba_host, uint32_t* bgra_host)
{
    if (!nd || !p || p->width == 0 || p->height == 0 || p->spp_end <= p->spp_begin) return PT_ERR_INVALID;
    nd->err.clear();
    int st = node_buffers(nd, p->width, p->height);
    if (st != PT_OK) return st;
    const uint32_t spp = p->spp_end - p->spp_begin;
    const int active = (uint32_t)nd->n <= spp ? nd->n : (int)spp;      // a GPU without a sample contributes a zero buffer
    const size_t floats = (size_t)p->width * p->height * 4;
    // every GPU renders its slice of the samples (asynchronous launches: one host thread serves them all)
    for (int g = 0; g < nd->n; ++g)
    {
        DeviceScope s(g);
        if (g < active)
        {
            pt_params q = *p;
            pt_spp_slice(p->spp_begin, p->spp_end, (uint32_t)active, (uint32_t)g, &q.spp_begin, &q.spp_end);
            st = pt_hip_render_device(nd->ctx[(size_t)g], &q, nd->accum[(size_t)g], nd->stream[(size_t)g]);
            if (st != PT_OK) { nd->err = std::string("device ") + std::to_string(g) + ": " + pt_hip_last_error(nd->ctx[(size_t)g]); return st; }
        }
        else
            NODE_HIP(nd, hipMemsetAsync(nd->accum[(size_t)g], 0, floats * sizeof(float), nd->stream[(size_t)g]));
    }
    // the join: one all-reduce of the f32 sums, in place, each rank on its own stream behind its render
    if (!nd->comm.empty())
    {
        NODE_NCCL(nd, g_rccl.GroupStart());
        for (int g = 0; g < nd->n; ++g)
        {
            const ncclResult_t r = g_rccl.AllReduce(nd->accum[(size_t)g], nd->accum[(size_t)g], floats, ncclFloat32, ncclSum, nd->comm[(size_t)g], nd->stream[(size_t)g]);
            if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); nd->err = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r); return PT_ERR_HIP; }
        }
        NODE_NCCL(nd, g_rccl.GroupEnd());
    }
    // device 0: resolve + pack, then only what was asked for crosses PCIe
    {
        DeviceScope s(0);
        if (bgra_host)
        {
            st = pt_hip_resolve_device(nd->ctx[0], nd->accum[0], p->width * p->height, spp, nullptr, nd->bgra_dev, nd->stream[0]);
            if (st != PT_OK) return st;
            NODE_HIP(nd, hipMemcpyAsync(bgra_host, nd->bgra_dev, (size_t)p->width * p->height * sizeof(uint32_t), hipMemcpyDeviceToHost, nd->stream[0]));
        }
        if (accum_rgba_host)
            NODE_HIP(nd, hipMemcpyAsync(accum_rgba_host, nd->accum[0], floats * sizeof(float), hipMemcpyDeviceToHost, nd->stream[0]));
    }
    for (int g = 0; g < nd->n; ++g)
    {
        DeviceScope s(g);
        NODE_HIP(nd, hipStreamSynchronize(nd->stream[(size_t)g]));
    }
    return PT_OK;
}

int pt_hip_render_node(const pt_scene_desc* scene, const pt_params* params, int n_gpus, float* accum_rgba_host)
{
    if (!scene || !params || !accum_rgba_host) return PT_ERR_INVALID;
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
