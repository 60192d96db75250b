// Lab tool (not a product path): when does each workgroup of murb_force_sym_kernel start and end, and where?
// Per workgroup: s_memrealtime (100 MHz) at entry and exit of wave 0 + XCC / SE / CU ids.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Inbody-eurohpc_amd/csrc tools/sym_stamps.hip -o tools/sym_stamps
//   tools/sym_stamps N split K
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

__device__ unsigned long long* g_stamps;   // [wg][4]: t0, t1, hw_id, xcc_id
#define MURB_LAB_BEGIN()                                                                          \
    if (threadIdx.x == 0) {                                                                       \
        g_stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();                          \
        g_stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                 \
        g_stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                \
    }
#define MURB_LAB_END() \
    if (threadIdx.x == 0) g_stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#include "murb_kernels_sym.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

int main(int argc, char** argv)
{
    const unsigned long n = argc > 1 ? strtoul(argv[1], nullptr, 10) : 200000;
    const int split = argc > 2 ? atoi(argv[2]) : 1;
    long K = argc > 3 ? atol(argv[3]) : 1024;
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
    K = std::min<long>(K, (long)items.size());
    int2* d_items; CK(hipMalloc(&d_items, items.size() * sizeof(int2)));
    CK(hipMemcpy(d_items, items.data(), items.size() * sizeof(int2), hipMemcpyHostToDevice));
    unsigned long long* d_st; CK(hipMalloc(&d_st, (size_t)K * 4 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof d_st));
    MurbSymArgs sa{};
    sa.rec = d_rec; sa.part = d_part; sa.items = d_items; sa.split = split; sa.nrows = nrows; sa.row_stride = (unsigned)slots;
    sa.soft2 = 4e16f;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(murb_force_sym_kernel<4>, dim3((unsigned)K), dim3(256), 0, 0, sa);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> st((size_t)K * 4);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t_min = ~0ull, t_max = 0;
    for (long w = 0; w < K; ++w) { t_min = std::min(t_min, st[4 * w]); t_max = std::max(t_max, st[4 * w + 1]); }
    std::vector<double> dur(K), start(K), end(K);
    std::map<unsigned, int> per_cu;
    for (long w = 0; w < K; ++w) {
        start[w] = (st[4 * w] - t_min) * 0.01; end[w] = (st[4 * w + 1] - t_min) * 0.01; dur[w] = end[w] - start[w];   // us
        const unsigned hw = (unsigned)st[4 * w + 2], xcc = (unsigned)st[4 * w + 3] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    printf("N=%lu split=%d K=%ld: kernel span %.1f us; distinct CUs %zu\n", n, split, K, (t_max - t_min) * 0.01, per_cu.size());
    printf("  start  us: min %.1f p50 %.1f p90 %.1f max %.1f\n", pct(start, 0), pct(start, .5), pct(start, .9), pct(start, 1));
    printf("  end    us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", pct(end, 0), pct(end, .1), pct(end, .5), pct(end, .9), pct(end, 1));
    printf("  durat. us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", pct(dur, 0), pct(dur, .1), pct(dur, .5), pct(dur, .9), pct(dur, 1));
    // first-round vs later workgroups
    if (K > 2048) {
        std::vector<double> d1(dur.begin(), dur.begin() + 1024), d2(dur.begin() + 2048, dur.end());
        printf("  duration of WGs 0..1023: p50 %.1f ; of WGs 2048..: p50 %.1f p10 %.1f p90 %.1f\n", pct(d1, .5), pct(d2, .5), pct(d2, .1), pct(d2, .9));
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("  workgroups per CU over the launch:");
    for (auto& kv : hist) printf("  %d WGs x %d CUs", kv.first, kv.second);
    printf("\n");
    // concurrency over time: how many workgroups are alive at 10 sample points
    for (int k = 1; k <= 9; ++k) {
        const double t = (t_max - t_min) * 0.01 * k / 10.0; int alive = 0;
        for (long w = 0; w < K; ++w) alive += (start[w] <= t && end[w] > t);
        printf("  t=%.0f us: %d alive;", t, alive);
    }
    printf("\n");
    return 0;
}
