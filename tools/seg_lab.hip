// Lab tool (not a product path): murb_force_seg_kernel (balanced runs) against murb_force_sym_kernel
// (one workgroup per item) on the same bodies: agreement of the summed accelerations, then interleaved
// timing for several workgroup counts.  Build:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Inbody-eurohpc_amd/csrc -Itools tools/seg_lab.hip -o tools/seg_lab
// Run: tools/seg_lab [N] [split of the item kernel]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "lab_kernels_seg.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void lab_sum_items(const float* part, int split, int nrows, unsigned int row_stride, float* out)
{
    const unsigned int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= row_stride) return;
    for (int c = 0; c < 3; ++c) {
        double acc = 0;
        for (int r = 0; r < nrows; ++r) acc += part[((unsigned long)c * nrows + r) * row_stride + s];
        out[(unsigned long)c * row_stride + s] = (float)acc;
    }
}

// slot s of block B: i-plane rows J >= B, j-plane rows [jfirst[B], jfirst[B+1])
__global__ __launch_bounds__(256) void lab_sum_seg(const float* iplane, const float* jplane, const int* jfirst, int T,
                                                   unsigned int slots, float* out)
{
    const unsigned int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots) return;
    const int B = s / MURB_SYM_BLOCK;
    for (int c = 0; c < 3; ++c) {
        double acc = 0;
        for (int J = B; J < T; ++J) acc += iplane[((unsigned long)c * T + J) * slots + s];
        for (int r = jfirst[B]; r < jfirst[B + 1]; ++r) acc += jplane[((unsigned long)r * 3 + c) * MURB_SYM_BLOCK + (s % MURB_SYM_BLOCK)];
        out[(unsigned long)c * slots + s] = (float)acc;
    }
}

int main(int argc, char** argv)
{
    const unsigned long n = argc > 1 ? strtoul(argv[1], nullptr, 10) : 30000;
    const int split = argc > 2 ? atoi(argv[2]) : 4;
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

    // item kernel
    const int nrows = T * split;
    float* d_part; CK(hipMalloc(&d_part, (size_t)3 * nrows * slots * sizeof(float)));
    CK(hipMemset(d_part, 0, (size_t)3 * nrows * slots * sizeof(float)));
    std::vector<int2> items;
    for (int j = 0; j < T; ++j)
        for (int i = 0; i < (j + 1) * split; ++i) items.push_back(make_int2(i, j));
    int2* d_items; CK(hipMalloc(&d_items, items.size() * sizeof(int2)));
    CK(hipMemcpy(d_items, items.data(), items.size() * sizeof(int2), hipMemcpyHostToDevice));
    MurbSymArgs sa{};
    sa.rec = d_rec; sa.part = d_part; sa.items = d_items; sa.split = split; sa.nrows = nrows; sa.row_stride = (unsigned)slots;
    sa.soft2 = 4e16f;
    float *d_a, *d_b; CK(hipMalloc(&d_a, 3 * slots * 4)); CK(hipMalloc(&d_b, 3 * slots * 4));

    // segment kernel: strips (J, units [0, 64 (J+1))), cut into nwg equal runs
    float* d_ipl; CK(hipMalloc(&d_ipl, (size_t)3 * T * slots * sizeof(float)));
    CK(hipMemset(d_ipl, 0, (size_t)3 * T * slots * sizeof(float)));
    const long total_units = 64L * T * (T + 1) / 2;
    struct Sched { std::vector<MurbSegEntry> ent; std::vector<int> wg_first, jfirst; };
    auto build_runs = [&](const std::vector<long>& run_end_of) {   // run w = units [run_end_of[w-1], run_end_of[w])
        Sched sc;
        const int nwg = (int)run_end_of.size();
        sc.jfirst.assign(T + 1, 0);
        std::vector<std::vector<MurbSegEntry>> per_wg(nwg);
        long pos = 0;
        int w = 0;
        for (int J = 0; J < T; ++J) {
            const long strip = 64L * (J + 1);
            long done = 0;
            while (done < strip) {
                while (w < nwg - 1 && pos >= run_end_of[w]) ++w;
                const long run_end = run_end_of[w];
                const long take = std::min(strip - done, std::max<long>(1, run_end - pos));
                MurbSegEntry e{}; e.J = J; e.irow = J; e.u0 = (int)done; e.u1 = (int)(done + take); e.jrow = 0;
                per_wg[w].push_back(e);
                done += take; pos += take;
            }
        }
        // j rows: contiguous per J -> number the pieces in J-major order
        std::vector<int> count(T, 0);
        for (auto& v : per_wg) for (auto& e : v) count[e.J]++;
        for (int J = 0; J < T; ++J) sc.jfirst[J + 1] = sc.jfirst[J] + count[J];
        std::vector<int> next(sc.jfirst.begin(), sc.jfirst.end() - 1);
        sc.wg_first.push_back(0);
        for (auto& v : per_wg) {
            for (auto& e : v) { e.jrow = next[e.J]++; sc.ent.push_back(e); }
            sc.wg_first.push_back((int)sc.ent.size());
        }
        return sc;
    };
    auto build = [&](int nwg) {
        std::vector<long> ends(nwg);
        for (int w = 0; w < nwg; ++w) ends[w] = ((long)(w + 1) * total_units + nwg - 1) / nwg;
        return build_runs(ends);
    };
    // guided taper: run = max(minlen, remaining / (f * 1024)) units, in launch (= age) order
    auto build_taper = [&](double f, long minlen, long cap = 1L << 40) {
        std::vector<long> ends;
        long pos = 0;
        while (pos < total_units) {
            long len = std::max<long>(minlen, (long)((total_units - pos) / (f * 1024.0)));
            len = std::min(std::min(len, cap), total_units - pos);
            pos += len; ends.push_back(pos);
        }
        return build_runs(ends);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](auto&& fn) {
        float best = 1e30f;
        for (int rep = 0; rep < 9; ++rep) {
            CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        return best;
    };
    auto run_items = [&]() { hipLaunchKernelGGL(murb_force_sym_kernel<4>, dim3((unsigned)items.size()), dim3(256), 0, 0, sa); };
    run_items();
    hipLaunchKernelGGL(lab_sum_items, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, 0, d_part, split, nrows, (unsigned)slots, d_a);
    CK(hipDeviceSynchronize());
    std::vector<float> ha(3 * slots), hb(3 * slots);
    CK(hipMemcpy(ha.data(), d_a, ha.size() * 4, hipMemcpyDeviceToHost));
    float t_items = time_it(run_items);
    const double pe = (double)T * (T + 1) / 2 * 1024.0 * 1024.0;
    printf("N=%lu T=%d: item kernel (split %d, %zu items) %.4f ms  %.3f T pair-evals/s\n", n, T, split, items.size(), t_items,
           pe / (t_items * 1e-3) / 1e12);

    struct Cfg { int nwg; double f; long minlen; long cap; };
    std::vector<Cfg> cfgs = {{0, 2.0, 8, 1L << 40}, {0, 2.0, 8, 256}, {0, 2.0, 8, 128}, {0, 2.0, 8, 64}, {0, 2.0, 8, 32},
                             {0, 2.0, 6, 64}, {0, 2.0, 12, 64}, {0, 1.5, 8, 64}, {0, 3.0, 8, 64}, {0, 2.0, 8, 48}, {0, 2.5, 10, 96}};
    for (const Cfg& cf : cfgs) {
        if ((long)cf.nwg > total_units) continue;
        Sched sc = cf.nwg ? build(cf.nwg) : build_taper(cf.f, cf.minlen, cf.cap);
        const int nwg = (int)sc.wg_first.size() - 1;
        MurbSegEntry* d_ent; int *d_wg, *d_jf; float* d_jpl;
        CK(hipMalloc(&d_ent, sc.ent.size() * sizeof(MurbSegEntry)));
        CK(hipMemcpy(d_ent, sc.ent.data(), sc.ent.size() * sizeof(MurbSegEntry), hipMemcpyHostToDevice));
        CK(hipMalloc(&d_wg, sc.wg_first.size() * 4)); CK(hipMemcpy(d_wg, sc.wg_first.data(), sc.wg_first.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_jf, sc.jfirst.size() * 4)); CK(hipMemcpy(d_jf, sc.jfirst.data(), sc.jfirst.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_jpl, sc.ent.size() * 3 * MURB_SYM_BLOCK * sizeof(float)));
        MurbSegArgs ga{};
        ga.rec = d_rec; ga.iplane = d_ipl; ga.jplane = d_jpl; ga.entries = d_ent; ga.wg_first = d_wg; ga.irows = T;
        ga.i_stride = (unsigned)slots; ga.i_slot0 = 0; ga.soft2 = 4e16f;
        auto run_seg = [&]() { hipLaunchKernelGGL(murb_force_seg_kernel<4>, dim3((unsigned)nwg), dim3(256), 0, 0, ga); };
        run_seg();
        hipLaunchKernelGGL(lab_sum_seg, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, 0, d_ipl, d_jpl, d_jf, T, (unsigned)slots, d_b);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hb.data(), d_b, hb.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (unsigned long i = 0; i < n; ++i) {
            double num = 0, den = 0;
            for (int c = 0; c < 3; ++c) { const double x = ha[c * slots + i], y = hb[c * slots + i]; num += (x - y) * (x - y); den += x * x; }
            worst = std::max(worst, std::sqrt(num / std::max(den, 1e-300)));
        }
        // interleaved A/B (the clock drifts over a process's life): alternate the two kernels, keep the best of each
        float t = 1e30f, t_items = 1e30f;
        for (int rep = 0; rep < 4; ++rep) { t_items = std::min(t_items, time_it(run_items)); t = std::min(t, time_it(run_seg)); }
        printf("  seg kernel %s f=%.1f min=%2ld cap=%4ld nwg=%5d pieces=%6zu: items %.4f seg %.4f ms  %.3f T pair-evals/s  (x%.3f vs items)  max rel diff %.2e\n", cf.nwg ? "equal" : "taper", cf.f, cf.minlen, std::min(cf.cap, 9999L), nwg,
               sc.ent.size(), t_items, t, pe / (t * 1e-3) / 1e12, t_items / t, worst);
        hipFree(d_ent); hipFree(d_wg); hipFree(d_jf); hipFree(d_jpl);
    }
    return 0;
}
