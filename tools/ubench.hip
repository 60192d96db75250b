// VALU issue-rate microbenchmark for gfx950 (MI355X).
//
// Purpose: pin the numbers the force kernel's roofline is built from:
// cycles per wave64 instruction per SIMD for v_fma_f32, v_pk_fma_f32, v_mul_f32, v_sub_f32,
// v_rsq_f32 and for the 12:1 FMA:rsq mix one body-body interaction needs, at 1/2/4/8 waves
// per SIMD. Cycles come from s_memtime (shader clock); the shader clock itself from the
// s_memtime / s_memrealtime (100 MHz) ratio.
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

enum Mode { M_FMA = 0, M_PKFMA, M_MUL, M_SUB, M_RSQ, M_MIX12_1, M_FMA_SGPR, M_PKMUL, M_MIX_PK, M_COUNT };
static const char* mode_name[] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_sub_f32", "v_rsq_f32",
                                  "mix 12 fma : 1 rsq", "v_fma_f32 (sgpr src)", "v_pk_mul_f32",
                                  "mix 6 pk_fma : 2 rsq"};
// wave-instructions issued per loop iteration, per mode
static const int mode_instr[] = {32, 32, 32, 32, 32, 52, 32, 32, 32};

struct Stamp { unsigned long long cyc, real; };

template <int MODE>
__global__ __launch_bounds__(1024) void rate_kernel(float* sink, Stamp* stamps, int iters, float seed)
{
    float a[16];
    f2 p[8];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = seed + 0.001f * (float)(threadIdx.x + k);
#pragma unroll
    for (int k = 0; k < 8; ++k) { p[k].x = a[2 * k]; p[k].y = a[2 * k + 1]; }
    float b = 0.999f + seed * 1e-6f, c = 1e-3f;
    f2 pb = {b, b}, pc = {c, c};
    const float sb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b)));

    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_FMA) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        } else if (MODE == M_FMA_SGPR) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "s"(sb), "v"(c));
        } else if (MODE == M_PKFMA) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
        } else if (MODE == M_PKMUL) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
        } else if (MODE == M_MUL) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        } else if (MODE == M_SUB) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        } else if (MODE == M_RSQ) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
        } else if (MODE == M_MIX12_1) {
            // 4 groups of (12 independent FMAs + 1 rsq): the instruction mix of 4 interactions
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int k = 0; k < 12; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(a[12 + g]));
            }
        } else if (MODE == M_MIX_PK) {
            // 4 groups of (6 pk_fma + 2 rsq): same flops as two interactions per group
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int k = 0; k < 6; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(a[12 + (g & 1) * 2]));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(a[13 + (g & 1) * 2]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();

    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += p[k].x + p[k].y;
    if (s == 123.456f) sink[0] = s;   // keep everything live
    if ((threadIdx.x & 63) == 0) {
        int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        stamps[w].cyc = t1 - t0;
        stamps[w].real = r1 - r0;
    }
}

template <int MODE>
static void run_mode(int cus, int waves_per_simd, int iters, float* sink, Stamp* d_stamps, std::vector<Stamp>& h)
{
    const int threads = 64 * 4 * waves_per_simd;       // one block per CU, 4*w waves
    const int blocks = cus;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<MODE><<<blocks, threads>>>(sink, d_stamps, 64, 1.0f);   // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    rate_kernel<MODE><<<blocks, threads>>>(sink, d_stamps, iters, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = blocks * threads / 64;
    CK(hipMemcpy(h.data(), d_stamps, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
    std::vector<double> cyc(nw), mhz(nw);
    for (int i = 0; i < nw; ++i) { cyc[i] = (double)h[i].cyc; mhz[i] = h[i].real ? 100.0 * h[i].cyc / (double)h[i].real : 0; }
    std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
    const double instr = (double)mode_instr[MODE] * iters;
    const double cpi_simd = cyc[nw / 2] / (instr * waves_per_simd);   // cycles per wave-instruction per SIMD
    const double wall_cpi = (ms * 1e-3) * (mhz[nw / 2] * 1e6) / (instr * waves_per_simd);
    printf("%-24s w/SIMD=%d  cyc/instr/SIMD=%6.3f (wall-derived %6.3f)  shader_clk=%7.1f MHz  kernel=%8.3f ms\n",
           mode_name[MODE], waves_per_simd, cpi_simd, wall_cpi, mhz[nw / 2], ms);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  arch=%s  CUs=%d  clockRate=%d kHz  memClock=%d kHz  L2=%d B  regsPerBlock=%d  smemPerBlock=%zu\n",
           prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, prop.memoryClockRate,
           prop.l2CacheSize, prop.regsPerBlock, prop.sharedMemPerBlock);
    const int cus = prop.multiProcessorCount;
    float* sink; CK(hipMalloc(&sink, 4));
    const int maxw = cus * 32;
    Stamp* d_stamps; CK(hipMalloc(&d_stamps, sizeof(Stamp) * maxw));
    std::vector<Stamp> h(maxw);
    const int ws[] = {1, 2, 4};
    for (int w : ws) {
        run_mode<M_FMA>(cus, w, iters, sink, d_stamps, h);
        run_mode<M_FMA_SGPR>(cus, w, iters, sink, d_stamps, h);
        run_mode<M_PKFMA>(cus, w, iters, sink, d_stamps, h);
        run_mode<M_PKMUL>(cus, w, iters, sink, d_stamps, h);
        run_mode<M_MUL>(cus, w, iters, sink, d_stamps, h);
        run_mode<M_SUB>(cus, w, iters, sink, d_stamps, h);
        run_mode<M_RSQ>(cus, w, iters, sink, d_stamps, h);
        run_mode<M_MIX12_1>(cus, w, iters, sink, d_stamps, h);
        run_mode<M_MIX_PK>(cus, w, iters, sink, d_stamps, h);
        printf("\n");
    }
    return 0;
}
