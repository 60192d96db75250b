// Lab tool (not a product path): when does each workgroup of murb_force_sym_kernel start and end, and where?
// Per workgroup: s_memrealtime (100 MHz) at entry and exit of wave 0 + XCC / SE / CU ids.  The item table and the layout
// of the partial rows come from the library's own planner (murbhip_schedule_layout), so what is stamped is exactly a
// launch of the product kernel under the given plan.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Inbody-eurohpc_amd/csrc -Iinclude tools/sym_stamps.hip -o tools/sym_stamps \
//         -Lnbody-eurohpc_amd/lib -lmurbhip -Wl,-rpath,'$ORIGIN/../nbody-eurohpc_amd/lib'
//   tools/sym_stamps N split waves taper diag_tri red [K]        (K: only the first K items, 0 = all)
// Prints the launch's span, the distribution of workgroup start / end / duration, how many workgroups are alive over
// time, and the fill/drain account: item time at full occupancy, the time an ideal dealer would need for the same
// items (sum of item work / slots), and what is lost before the chip is full and after the queue is empty.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

#include "murbhip.h"

__device__ unsigned long long* g_stamps;   // [wg][4]: t0, t1, hw_id, xcc_id
#define MURB_LAB_BEGIN()                                                                          \
    if (threadIdx.x == 0) {                                                                       \
        g_stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();                          \
        g_stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                 \
        g_stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                \
    }
#define MURB_LAB_END() \
    if (threadIdx.x == 0) g_stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
// The library this tool links (for its planner) contains the same kernel templates: a namespace of our own keeps the
// instrumented instantiations from sharing a symbol with the library's, or the launch may pick up the uninstrumented one.
namespace lab {
#include "murb_kernels_sym.h"
}
using namespace lab;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int WAVES, int RED> void launch(const MurbSymArgs& sa, long K)
{
    hipLaunchKernelGGL((murb_force_sym_kernel<4, WAVES, 1, 0, RED>), dim3((unsigned)K), dim3(64 * WAVES), 0, 0, sa);
}

int main(int argc, char** argv)
{
    const unsigned long n = argc > 1 ? strtoul(argv[1], nullptr, 10) : 200000;
    const int split = argc > 2 ? atoi(argv[2]) : 1, waves = argc > 3 ? atoi(argv[3]) : 4, taper = argc > 4 ? atoi(argv[4]) : 0;
    const int diag_tri = argc > 5 ? atoi(argv[5]) : 0, red = argc > 6 ? atoi(argv[6]) : 0;
    long K = argc > 7 ? atol(argv[7]) : 0;
    const unsigned long slots = murbhip_slice_slots(n, 1);
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

    unsigned long ni = 0, nr = 0, fm = 0, ft = 0;
    const int tp = taper + (diag_tri ? 256 : 0);
    if (murbhip_schedule_layout(n, 1, 0, split, waves, tp, 50, 0, nullptr, 0, &ni, nullptr, 0, &nr, &fm, &ft) != 0) { fprintf(stderr, "bad plan\n"); return 1; }
    std::vector<long> flat(8 * ni), rows(7 * nr);
    murbhip_schedule_layout(n, 1, 0, split, waves, tp, 50, 0, flat.data(), ni, &ni, rows.data(), nr, &nr, &fm, &ft);
    std::vector<MurbSymItem> items(ni);
    double work_total = 0;
    std::vector<double> work(ni);
    for (unsigned long k = 0; k < ni; ++k) {
        const long* o = &flat[8 * k];
        items[k] = MurbSymItem{(int)o[0], (int)(o[1] / (waves * MURB_SYM_R)), (int)o[2], (int)o[3], (unsigned long)o[5], (unsigned long)o[6]};
        const int p_first = (o[3] & 2) ? (int)((o[3] >> 8) & 15) : 0;
        work[k] = (double)o[1] * 128.0 * (8 - p_first);   // pair evaluations of the item
    }
    if (K <= 0 || K > (long)ni) K = (long)ni;
    for (long k = 0; k < K; ++k) work_total += work[k];
    float* d_part; CK(hipMalloc(&d_part, (size_t)3 * fm * sizeof(float)));
    CK(hipMemset(d_part, 0, (size_t)3 * fm * sizeof(float)));
    MurbSymItem* d_items; CK(hipMalloc(&d_items, items.size() * sizeof(MurbSymItem)));
    CK(hipMemcpy(d_items, items.data(), items.size() * sizeof(MurbSymItem), hipMemcpyHostToDevice));
    unsigned long long* d_st; CK(hipMalloc(&d_st, (size_t)K * 4 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof d_st));
    MurbSymArgs sa{};
    sa.rec = d_rec; sa.part = d_part; sa.comp_stride = fm; sa.items = d_items; sa.item_first = 0; sa.soft2 = 4e16f;
    for (int rep = 0; rep < 40; ++rep) {   // the clock needs ~40 ms of work to settle; the last launch is the one read
        if (waves == 8) { if (red) launch<8, 1>(sa, K); else launch<8, 0>(sa, K); }
        else { if (red) launch<4, 1>(sa, K); else launch<4, 0>(sa, K); }
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
    const double span = (t_max - t_min) * 0.01;
    const int slots_wg = (int)per_cu.size() * (waves == 8 ? 2 : 4);
    printf("N=%lu split=%d waves=%d taper=%d diag_tri=%d red=%d items=%ld of %lu: kernel span %.1f us; distinct CUs %zu; %d workgroup slots\n",
           n, split, waves, taper, diag_tri, red, K, ni, span, per_cu.size(), slots_wg);
    printf("  start  us: min %.1f p50 %.1f p90 %.1f max %.1f\n", pct(start, 0), pct(start, .5), pct(start, .9), pct(start, 1));
    printf("  end    us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", pct(end, 0), pct(end, .1), pct(end, .5), pct(end, .9), pct(end, 1));
    printf("  durat. us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", pct(dur, 0), pct(dur, .1), pct(dur, .5), pct(dur, .9), pct(dur, 1));
    // occupancy over time
    printf("  workgroups alive:");
    for (int k = 1; k <= 19; ++k) {
        const double t = span * k / 20.0; int alive = 0;
        for (long w = 0; w < K; ++w) alive += (start[w] <= t && end[w] > t);
        printf(" %d", alive);
    }
    printf("   (at 5 %%, 10 %%, ... 95 %% of the span)\n");
    // the fill/drain account
    double t_queue_empty = 0;                       // when the last workgroup starts
    for (long w = 0; w < K; ++w) t_queue_empty = std::max(t_queue_empty, start[w]);
    double wg_time = 0;
    for (long w = 0; w < K; ++w) wg_time += dur[w];
    // steady-state rate: work completed per us while the chip is full = work of the items that start after the first round
    // and end before the queue runs empty, over their share of slot time
    double w_mid = 0, t_mid = 0;
    for (long w = 0; w < K; ++w)
        if (start[w] > pct(dur, .5) && end[w] < t_queue_empty) { w_mid += work[w]; t_mid += dur[w]; }
    const double rate_slot = t_mid > 0 ? w_mid / t_mid : 0;             // pair evaluations per us and slot, chip full
    const double ideal = rate_slot > 0 ? work_total / (rate_slot * slots_wg) : 0;
    printf("  account: queue empty at %.1f us (%.1f %% of the span); mean occupancy %.1f %% of %d slots; chip-full rate %.3g pair evaluations/us/slot\n",
           t_queue_empty, 100 * t_queue_empty / span, 100 * wg_time / (span * slots_wg), slots_wg, rate_slot);
    if (ideal > 0)
        printf("           the same items at the chip-full rate with every slot busy to the end: %.1f us -> fill + drain cost %.1f us = %.1f %% of the span\n",
               ideal, span - ideal, 100 * (span - ideal) / span);
    return 0;
}
