// Kernel lab (not a product path): runs force-kernel candidates on the same random bodies in one
// process, checks them against each other, and times them interleaved (guide rule: perf deltas come
// from interleaved rounds in ONE process).
//
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Inbody-eurohpc_amd/csrc tools/kernel_lab.hip -o tools/kernel_lab
// Run:   tools/kernel_lab [N] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "murb_kernels_sym.h"

__global__ __launch_bounds__(256) void murb_sym_sum_rows(const float* part, int nrows_used, int nrows,
                                                         unsigned int row_stride, float* out)
{
    const unsigned int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= row_stride) return;
    for (int c = 0; c < 3; ++c) {
        float acc = 0.f;
        for (int r = 0; r < nrows_used; ++r) acc += part[((unsigned long)c * nrows + r) * row_stride + s];
        out[(unsigned long)c * row_stride + s] = acc;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

int main(int argc, char** argv)
{
    const unsigned long n = argc > 1 ? strtoul(argv[1], nullptr, 10) : 200000;
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    const unsigned long slots = ((n + MURB_SYM_BLOCK - 1) / MURB_SYM_BLOCK) * MURB_SYM_BLOCK;
    const int T = (int)(slots / MURB_SYM_BLOCK);
    const float soft = 2e8f, G = 6.67384e-11f;
    printf("N=%lu slots=%lu blocks(T)=%d items=%ld\n", n, slots, T, (long)T * (T + 1) / 2);

    // galaxy-like random bodies, packed in the pair layout
    std::vector<float4> rec(slots, make_float4(0, 0, 0, 0));
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-1.f, 1.f), m(0.f, 5e20f);
    for (unsigned long s = 0; s < n; ++s) {
        const unsigned long ra = murb_rec_a(s >> 1);
        float* A = reinterpret_cast<float*>(&rec[ra]);
        float* B = reinterpret_cast<float*>(&rec[ra + MURB_TILE_PAIRS]);
        const int h = (int)(s & 1);
        A[h] = 2e8f * u(rng); A[2 + h] = 2e8f * u(rng); B[h] = 2e8f * u(rng);
        B[2 + h] = G * (s == 0 ? 2e24f : m(rng));
    }
    float4* d_rec; CK(hipMalloc(&d_rec, slots * sizeof(float4)));
    CK(hipMemcpy(d_rec, rec.data(), slots * sizeof(float4), hipMemcpyHostToDevice));

    // one-sided reference kernel (library default): 6 chunks + integrate-style row sum
    const int nch = 6;
    float4* d_accp; CK(hipMalloc(&d_accp, (size_t)nch * slots * sizeof(float4)));
    float* d_ref; CK(hipMalloc(&d_ref, 3 * slots * sizeof(float)));
    MurbForceArgs fa{};
    fa.rec = d_rec; fa.accp = d_accp; fa.tiles = MurbTileRange{0, (int)(slots / MURB_TILE_BODIES), (int)(slots / MURB_TILE_BODIES), 0};
    fa.i_first_slot = 0; fa.chunk_first = 0; fa.nchunks = nch; fa.acc_stride = (unsigned)slots; fa.soft2 = soft * soft;
    MurbIntegrateArgs ia{};
    ia.rec_in = d_rec; ia.rec_out = d_rec; ia.vel = nullptr; ia.accp = d_accp; ia.acc_out = d_ref; ia.i_first_slot = 0;
    ia.count = (int)n; ia.nparts = nch; ia.acc_stride = (unsigned)slots; ia.dt = 0; ia.update_state = 0; ia.nsched = 0;

    // symmetric kernel
    const int nrows = T;
    float* d_part; CK(hipMalloc(&d_part, (size_t)3 * nrows * slots * sizeof(float)));
    CK(hipMemset(d_part, 0xff, (size_t)3 * nrows * slots * sizeof(float)));   // NaN: any unwritten cell shows up
    float* d_sym; CK(hipMalloc(&d_sym, 3 * slots * sizeof(float)));
    MurbSymArgs sa{};
    std::vector<int2> h_items;
    for (int j = 0; j < T; ++j) for (int i = 0; i <= j; ++i) h_items.push_back(make_int2(i, j));
    int2* d_items; CK(hipMalloc(&d_items, h_items.size() * sizeof(int2)));
    CK(hipMemcpy(d_items, h_items.data(), h_items.size() * sizeof(int2), hipMemcpyHostToDevice));
    sa.rec = d_rec; sa.part = d_part; sa.items = d_items; sa.item_first = 0; sa.split = 1; sa.nrows = nrows; sa.row_stride = (unsigned)slots;
    sa.soft2 = soft * soft;
    const long items = (long)T * (T + 1) / 2;

    auto run_ref = [&]() {
        hipLaunchKernelGGL((murb_force_kernel<MURB_MODE_PK_LDS, 8, 4, 4>), dim3((unsigned)((n + 31) / 32), nch), dim3(256), 0, 0, fa);
        hipLaunchKernelGGL(murb_integrate_kernel, dim3((unsigned)((slots / 2 + 255) / 256)), dim3(256), 0, 0, ia);
    };
    int minw = argc > 3 ? atoi(argv[3]) : 3;
    auto run_sym = [&]() {
        if (minw == 2) hipLaunchKernelGGL(murb_force_sym_kernel<2>, dim3((unsigned)items), dim3(256), 0, 0, sa);
        else if (minw == 3) hipLaunchKernelGGL(murb_force_sym_kernel<3>, dim3((unsigned)items), dim3(256), 0, 0, sa);
        else hipLaunchKernelGGL(murb_force_sym_kernel<4>, dim3((unsigned)items), dim3(256), 0, 0, sa);
        hipLaunchKernelGGL(murb_sym_sum_rows, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, 0, d_part, T, nrows,
                           (unsigned)slots, d_sym);
    };
    run_ref(); run_sym();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());

    std::vector<float> a(3 * slots), b(3 * slots);
    CK(hipMemcpy(a.data(), d_ref, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_sym, b.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, sumsq = 0; unsigned long bad = 0, worst_i = 0;
    for (unsigned long i = 0; i < n; ++i) {
        double num = 0, den = 0;
        for (int c = 0; c < 3; ++c) {
            const double x = a[c * slots + i], y = b[c * slots + i];
            if (!(y == y)) { ++bad; }
            num += (x - y) * (x - y); den += x * x;
        }
        const double e = std::sqrt(num / std::max(den, 1e-300));
        if (e > worst) { worst = e; worst_i = i; }
        sumsq += e * e;
    }
    printf("symmetric vs one-sided: max rel %.3e (body %lu)  rms %.3e  NaN entries %lu\n", worst, worst_i,
           std::sqrt(sumsq / n), bad);

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = n <= 50000 ? 50 : 8;
    for (int r = 0; r < rounds; ++r) {
        float ms_ref, ms_sym;
        CK(hipEventRecord(e0)); for (int k = 0; k < reps; ++k) run_ref(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_ref, e0, e1));
        CK(hipEventRecord(e0)); for (int k = 0; k < reps; ++k) run_sym(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_sym, e0, e1));
        const double inter = (double)n * (double)n;
        printf("round %d: one-sided %8.3f ms (%.3f T inter/s)   symmetric %8.3f ms (%.3f T inter/s)   speedup %.3f\n", r,
               ms_ref / reps, inter / (ms_ref / reps * 1e-3) / 1e12, ms_sym / reps, inter / (ms_sym / reps * 1e-3) / 1e12,
               ms_ref / ms_sym);
    }
    return 0;
}
