// Lab tool (not a product path): murb_force_sym_kernel: i bodies by vector load + readfirstlane against scalar loads, interleaved.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Inbody-eurohpc_amd/csrc tools/sload_lab.hip -o tools/sload_lab
//   tools/waves_lab N split4 split8
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "murb_kernels_sym.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void lab_sum(const float* part, int nrows, unsigned int row_stride, float* out)
{
    const unsigned int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= row_stride) return;
    for (int c = 0; c < 3; ++c) {
        double acc = 0;
        for (int r = 0; r < nrows; ++r) acc += part[((unsigned long)c * nrows + r) * row_stride + s];
        out[(unsigned long)c * row_stride + s] = (float)acc;
    }
}

struct Side {
    int split, nrows; float* part; int2* items; size_t nitems; MurbSymArgs sa; float* sum;
};

int main(int argc, char** argv)
{
    const unsigned long n = argc > 1 ? strtoul(argv[1], nullptr, 10) : 30000;
    const int split4 = argc > 2 ? atoi(argv[2]) : 4, split8 = argc > 3 ? atoi(argv[3]) : 4;
    const unsigned long slots = ((n + MURB_SYM_BLOCK - 1) / MURB_SYM_BLOCK) * MURB_SYM_BLOCK;
    const int T = (int)(slots / MURB_SYM_BLOCK);
    std::vector<float4> rec(slots, make_float4(0, 0, 0, 0));
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-1.f, 1.f), m(0.f, 5e20f);
    for (unsigned long s = 0; s < n; ++s) {
        const unsigned long ra = murb_rec_a(s >> 1);
        float* A = reinterpret_cast<float*>(&rec[ra]);
        float* B = reinterpret_cast<float*>(&rec[ra + MURB_TILE_PAIRS]);
        const int h = (int)(s & 1);
        A[h] = 2e8f * u(rng); A[2 + h] = 2e8f * u(rng); B[h] = 2e8f * u(rng);
        B[2 + h] = 6.67384e-11f * (s == 0 ? 2e24f : m(rng));
    }
    float4* d_rec; CK(hipMalloc(&d_rec, slots * sizeof(float4)));
    CK(hipMemcpy(d_rec, rec.data(), slots * sizeof(float4), hipMemcpyHostToDevice));
    auto make = [&](int split) {
        Side sd{}; sd.split = split; sd.nrows = T * split;
        CK(hipMalloc(&sd.part, (size_t)3 * sd.nrows * slots * sizeof(float)));
        CK(hipMemset(sd.part, 0, (size_t)3 * sd.nrows * slots * sizeof(float)));
        std::vector<int2> items;
        for (int j = 0; j < T; ++j)
            for (int i = 0; i < (j + 1) * split; ++i) items.push_back(make_int2(i, j));
        sd.nitems = items.size();
        CK(hipMalloc(&sd.items, items.size() * sizeof(int2)));
        CK(hipMemcpy(sd.items, items.data(), items.size() * sizeof(int2), hipMemcpyHostToDevice));
        sd.sa.rec = d_rec; sd.sa.part = sd.part; sd.sa.items = sd.items; sd.sa.split = split; sd.sa.nrows = sd.nrows;
        sd.sa.row_stride = (unsigned)slots; sd.sa.soft2 = 4e16f;
        CK(hipMalloc(&sd.sum, 3 * slots * 4));
        return sd;
    };
    Side a = make(split4), b = make(split8);
    auto run4 = [&]() { hipLaunchKernelGGL((murb_force_sym_kernel<4, 4>), dim3((unsigned)a.nitems), dim3(256), 0, 0, a.sa); };
    auto run8 = [&]() { hipLaunchKernelGGL((murb_force_sym_kernel<4, 4, 1>), dim3((unsigned)b.nitems), dim3(256), 0, 0, b.sa); };
    run4(); run8();
    hipLaunchKernelGGL(lab_sum, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, 0, a.part, a.nrows, (unsigned)slots, a.sum);
    hipLaunchKernelGGL(lab_sum, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, 0, b.part, b.nrows, (unsigned)slots, b.sum);
    CK(hipDeviceSynchronize());
    std::vector<float> ha(3 * slots), hb(3 * slots);
    CK(hipMemcpy(ha.data(), a.sum, ha.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b.sum, hb.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (unsigned long i = 0; i < n; ++i) {
        double num = 0, den = 0;
        for (int c = 0; c < 3; ++c) { const double x = ha[c * slots + i], y = hb[c * slots + i]; num += (x - y) * (x - y); den += x * x; }
        worst = std::max(worst, std::sqrt(num / std::max(den, 1e-300)));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](auto&& fn) {
        float best = 1e30f;
        for (int rep = 0; rep < 9; ++rep) {
            CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        return best;
    };
    for (int k = 0; k < 30; ++k) { run4(); run8(); }   // clock ramp
    float t4 = 1e30f, t8 = 1e30f;
    for (int rep = 0; rep < 5; ++rep) { t4 = std::min(t4, time_it(run4)); t8 = std::min(t8, time_it(run8)); }
    const double pe = (double)T * (T + 1) / 2 * 1024.0 * 1024.0;
    printf("N=%lu T=%d: vector i loads (split %d, %zu items) %.4f ms %.3f T pe/s | scalar i loads (split %d, %zu items) %.4f ms %.3f T pe/s | x%.3f | max rel diff %.2e\n",
           n, T, split4, a.nitems, t4, pe / (t4 * 1e-3) / 1e12, split8, b.nitems, t8, pe / (t8 * 1e-3) / 1e12, t4 / t8, worst);
    return 0;
}
