// tools/microbench.hip -- per-SIMD issue rates of the instructions the sphere search is made of (gfx950).
// Not part of the library; run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/mb && /tmp/mb
// Each wave runs ITER iterations of 16 independent copies of one instruction and stamps s_memtime around the loop;
// rate = waves_per_SIMD * 16 * ITER / cycles  (wave-instructions per cycle per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define ITER 2048

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND> __global__ void __launch_bounds__(256) k(unsigned long long* cyc, float* sink, float seed)
{
    __shared__ float4 lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(seed, seed + 1, seed + 2, seed + 3);
    __syncthreads();
    float a[16]; f2 p[16]; unsigned u[16]; float4 q[4];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i; p[i] = f2{seed + i, seed - i}; u[i] = (unsigned)(seed * 1000) + i; }
    for (int i = 0; i < 4; ++i) q[i] = make_float4(0, 0, 0, 0);
    float b = seed * 0.5f, c = seed * 0.25f; f2 pb = {b, c}, pc = {c, b};
    unsigned ub = 0x9E3779B9u ^ (unsigned)seed;
    unsigned long long uw[16]; for (int i = 0; i < 16; ++i) uw[i] = i + (unsigned)seed;
    unsigned addr = (threadIdx.x & 0) * 16; // all lanes the same LDS address: broadcast read
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < ITER; ++it)
    {
        if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X)
#undef X
        } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
            REP16(X)
#undef X
        } else if (KIND == 2) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (KIND == 3) {
#define X(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(u[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if (KIND == 4) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "s"(ub), "v"(u[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if (KIND == 5) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            REP16(X)
#undef X
        } else if (KIND == 6) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            REP16(X)
#undef X
        } else if (KIND == 7) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
            REP16(X)
#undef X
        } else if (KIND == 8) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(uw[i]) : "v"(ub), "v"(u[i]) : "vcc");
            REP16(X)
#undef X
        } else if (KIND == 9) { // broadcast ds_read_b128, 4 in flight
#define X(i) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(addr));
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 10) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X)
#undef X
        } else if (KIND == 11) { // the search's per-pair mix: 10 packed + 2 sqrt + 2 sub + min + perm = 16
            asm volatile(
                "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n"
                "v_pk_mul_f32 %3, %0, %5\n v_pk_fma_f32 %3, %1, %5, %3\n v_pk_fma_f32 %3, %2, %5, %3\n"
                "v_pk_fma_f32 %0, %0, %0, %4\n v_pk_fma_f32 %0, %1, %1, %0\n v_pk_fma_f32 %0, %2, %2, %0\n"
                "v_pk_fma_f32 %0, %3, %3, %0\n"
                : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(pb), "v"(pc));
            asm volatile("v_sqrt_f32 %0, %2\n v_sqrt_f32 %1, %3\n v_sub_f32 %0, %2, %0\n v_sub_f32 %1, %3, %1\n"
                         "v_min_u32 %0, %0, %1\n v_perm_b32 %0, %0, %4, %5"
                         : "+v"(a[0]), "+v"(a[1]) : "v"(a[2]), "v"(a[3]), "s"(ub), "v"(u[0]));
        } else if (KIND == 12) {
#define X(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (KIND == 13) { // sqrt interleaved 1:3 with fma (does the transcendental unit overlap?)
#define X(i) asm volatile("v_sqrt_f32 %0, %0\n v_fma_f32 %1, %1, %3, %4\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %5, %5, %3, %4" \
                          : "+v"(a[(i & 3)]), "+v"(a[4 + (i & 3)]), "+v"(a[8 + (i & 3)]) : "v"(b), "v"(c), "v"(a[12 + (i & 3)]));
            X(0) X(1) X(2) X(3)
#undef X
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i] + (float)uw[i];
    for (int i = 0; i < 4; ++i) s += q[i].x + q[i].w;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> void run(const char* name, int instr_per_iter)
{
    int ncu = 256;
    for (int wps : {1, 2, 4, 8})
    {
        int blocks = ncu * wps;
        unsigned long long* d; float* sink;
        hipMalloc(&d, blocks * 4 * 8); hipMalloc(&sink, 4);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, sink, 1.5f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, sink, 1.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), d, blocks * 4 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        double med = (double)h[h.size() / 2];
        double rate = (double)wps * instr_per_iter * ITER / med; // wave-instr / cycle / SIMD
        double wall_rate = (double)blocks * 4 * instr_per_iter * ITER / (ms * 1e-3) / (256.0 * 4) / 1e9; // G wave-instr/s/SIMD
        printf("%-28s waves/SIMD %d  cycles/wave %10.0f  -> %.3f instr/clk/SIMD (%.2f clk/instr)   wall %.3f Ginstr/s/SIMD  kernel %.3f ms\n",
               name, wps, med, rate, 1.0 / rate, wall_rate, ms);
        hipFree(d); hipFree(sink);
    }
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<0>("v_fma_f32", 16);
    run<1>("v_pk_fma_f32", 16);
    run<5>("v_pk_add_f32", 16);
    run<6>("v_pk_mul_f32", 16);
    run<10>("v_sub_f32", 16);
    run<2>("v_sqrt_f32", 16);
    run<12>("v_rsq_f32", 16);
    run<13>("sqrt:fma 1:3 (16 instr)", 16);
    run<3>("v_min3_u32", 16);
    run<4>("v_perm_b32 (sgpr src)", 16);
    run<7>("v_mul_hi_u32", 16);
    run<8>("v_mad_u64_u32", 16);
    run<9>("ds_read_b128 broadcast", 16);
    run<11>("search pair mix (16 instr)", 16);
    return 0;
}
