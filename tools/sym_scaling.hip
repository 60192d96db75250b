// Lab tool (not a product path): how does the pair-symmetric kernel's time grow with the number of items?
// Launches the first K items of the N-body triangular schedule for a range of K and prints ms and ms per
// 1024 items ("round").  Build:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Inbody-eurohpc_amd/csrc tools/sym_scaling.hip -o tools/sym_scaling
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "murb_kernels_sym.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

int main(int argc, char** argv)
{
    const unsigned long n = argc > 1 ? strtoul(argv[1], nullptr, 10) : 200000;
    const int split = argc > 2 ? atoi(argv[2]) : 1;
    const int order = argc > 3 ? atoi(argv[3]) : 0;   // 0: J-major (library), 1: shuffled
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
        B[2 + h] = 6.67384e-11f * m(rng);
    }
    float4* d_rec; CK(hipMalloc(&d_rec, slots * sizeof(float4)));
    CK(hipMemcpy(d_rec, rec.data(), slots * sizeof(float4), hipMemcpyHostToDevice));
    const int nrows = T * split;
    float* d_part; CK(hipMalloc(&d_part, (size_t)3 * nrows * slots * sizeof(float)));
    CK(hipMemset(d_part, 0, (size_t)3 * nrows * slots * sizeof(float)));
    std::vector<int2> items;
    for (int j = 0; j < T; ++j)
        for (int i = 0; i < (j + 1) * split; ++i) items.push_back(make_int2(i, j));
    if (order == 1) std::shuffle(items.begin(), items.end(), rng);
    int2* d_items; CK(hipMalloc(&d_items, items.size() * sizeof(int2)));
    CK(hipMemcpy(d_items, items.data(), items.size() * sizeof(int2), hipMemcpyHostToDevice));
    MurbSymArgs sa{};
    sa.rec = d_rec; sa.part = d_part; sa.items = d_items; sa.item_first = 0; sa.split = split; sa.nrows = nrows;
    sa.row_stride = (unsigned)slots; sa.soft2 = 4e16f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("N=%lu T=%d split=%d items=%zu order=%d\n", n, T, split, items.size(), order);
    std::vector<long> ks = {256, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 32768, 65536, (long)items.size()};
    for (long k : ks) {
        if (k > (long)items.size()) continue;
        float best = 1e30f;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(murb_force_sym_kernel<4>, dim3((unsigned)k), dim3(256), 0, 0, sa);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        const double pair_evals = (double)k * 1024.0 * 1024.0 / split;
        printf("  items %7ld (%6.2f rounds of 1024): %8.4f ms   %7.4f ms/round   %.3f T pair-evals/s\n", k, k / 1024.0, best,
               best / (k / 1024.0), pair_evals / (best * 1e-3) / 1e12);
    }
    return 0;
}
