// LAB KERNEL (not part of the product; measured and dropped).
// Pair-symmetric force kernel, balanced-run form ("segments").
//
// Same arithmetic as murb_force_sym_kernel (murb_kernels_sym.h): every unordered body pair once, both
// directions (reference SimulationNBodyOptim.cpp:60-82), 16 packed + 2 rsq wave instructions per
// (i, j-pair).  What changes is how the work is cut:
//
//   murb_force_sym_kernel : one workgroup per item (i sub-block x j block); the hardware deals items to
//                           free slots.  A launch therefore costs a whole number of item times per slot
//                           plus a fill/drain phase in which the four workgroups of a CU load, compute
//                           and reduce in lockstep (measured with tools/sym_scaling: 0.10 ms on top of
//                           0.33 ms per 1024 items; 1860 items at N=30k take as long as 2048).
//   murb_force_seg_kernel : the host lays all (j block, i strip) work of a launch end to end in units of 16
//                           i bodies (one group of R = 4 per wave) and cuts the sequence into one equal
//                           run per workgroup; a workgroup walks its run, keeping the j tile in LDS and the
//                           j-side sums in registers until the j block changes.  Every workgroup gets the
//                           same number of units (+-1), so a launch of exactly one workgroup per resident
//                           slot has no tail, and the j-side prologue/epilogue (tile load, cross-wave
//                           combine, 12 KiB of partial sums) happens once per (run, j block) instead of
//                           once per item.
//
// Partial sums (no atomics, every cell has exactly one writer per launch, bit-reproducible):
//   i side: iplane[c][irow][slot - i_slot0]   one row per j block the launch touches (irow from the entry)
//   j side: jplane[jrow][c][0..1023]          one compact row per (run, j block) piece
// A strip's i bodies may include the j block itself (the diagonal block of a triangular schedule): those
// groups are evaluated in full and contribute to the i side only (their G*m enters the j side as 0).
#ifndef MURB_KERNELS_SEG_H_
#define MURB_KERNELS_SEG_H_

#include "murb_kernels_sym.h"

#define MURB_SEG_UNIT 16   /* i bodies per unit: one group of MURB_SYM_R per wave */

struct MurbSegEntry {
    int J;       // j block (1024 slots of the record buffer)
    int irow;    // row of the i plane that collects the i-side sums against block J
    int u0, u1;  // units [u0, u1): i slots [16 u0, 16 u1) of the record buffer
    int jrow;    // row of the j plane for this piece's j-side sums
    int pad[3];
};

struct MurbSegArgs {
    const float4* rec;
    float* iplane;
    float* jplane;
    const MurbSegEntry* entries;
    const int* wg_first;     // entries of workgroup w: [wg_first[w], wg_first[w + 1])
    int irows;               // rows of the i plane
    unsigned int i_stride;   // floats per i-plane row
    int i_slot0;             // record-buffer slot of i-plane column 0
    float soft2;
};

// as murb_interact_sym, with the i body's G*m for the j side passed separately (0 on diagonal groups)
__device__ __forceinline__ void murb_interact_seg(const murb_f2 xj, const murb_f2 yj, const murb_f2 zj, const murb_f2 gj,
                                                  const float xi, const float yi, const float zi, const float neg_gi_j,
                                                  const float soft2, murb_f2& aix, murb_f2& aiy, murb_f2& aiz,
                                                  murb_f2& ajx, murb_f2& ajy, murb_f2& ajz)
{
    const murb_f2 dx = xj - xi;
    const murb_f2 dy = yj - yi;
    const murb_f2 dz = zj - zi;
    murb_f2 r2 = __builtin_elementwise_fma(dx, dx, (murb_f2)(soft2));
    r2 = __builtin_elementwise_fma(dy, dy, r2);
    r2 = __builtin_elementwise_fma(dz, dz, r2);
    murb_f2 inv;
    inv.x = __builtin_amdgcn_rsqf(r2.x);
    inv.y = __builtin_amdgcn_rsqf(r2.y);
    const murb_f2 inv3 = (inv * inv) * inv;
    const murb_f2 fi = gj * inv3;
    const murb_f2 fj = inv3 * neg_gi_j;
    aix = __builtin_elementwise_fma(fi, dx, aix);
    aiy = __builtin_elementwise_fma(fi, dy, aiy);
    aiz = __builtin_elementwise_fma(fi, dz, aiz);
    ajx = __builtin_elementwise_fma(fj, dx, ajx);
    ajy = __builtin_elementwise_fma(fj, dy, ajy);
    ajz = __builtin_elementwise_fma(fj, dz, ajz);
}

template <int MINW>
__global__ __launch_bounds__(256, MINW) void murb_force_seg_kernel(const MurbSegArgs a)
{
    constexpr int R = MURB_SYM_R;
    __shared__ float4 tileA[MURB_SYM_PAIRS];
    __shared__ float4 tileB[MURB_SYM_PAIRS];
    __shared__ murb_f2 scratch[2][3][MURB_SYM_PAIRS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float soft2 = a.soft2;
    const int e_begin = __builtin_amdgcn_readfirstlane(a.wg_first[blockIdx.x]);
    const int e_end = __builtin_amdgcn_readfirstlane(a.wg_first[blockIdx.x + 1]);

    // which of the 12 i-side totals this lane ends up with (murb_reduce12): value 3 * body + component
    int out_r, out_c;
    {
        const int b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
        const int idx = b2 ? 8 + 2 * b4 + b5 : 4 * b3 + 2 * b4 + b5;
        out_r = idx / 3;
        out_c = idx - 3 * out_r;
    }

#pragma unroll 1
    for (int e = e_begin; e < e_end; ++e) {
        const MurbSegEntry* ent = a.entries + e;
        const int J = __builtin_amdgcn_readfirstlane(ent->J);
        const int irow = __builtin_amdgcn_readfirstlane(ent->irow);
        const int u0 = __builtin_amdgcn_readfirstlane(ent->u0);
        const int u1 = __builtin_amdgcn_readfirstlane(ent->u1);
        const int jrow = __builtin_amdgcn_readfirstlane(ent->jrow);

        __syncthreads();   // the previous piece's epilogue has finished with the tiles' neighbours in LDS
        {
            const float4* src = a.rec + (unsigned long)J * (MURB_SYM_BLOCK / MURB_TILE_BODIES) * MURB_TILE_F4;
#pragma unroll
            for (int k = threadIdx.x; k < 2 * MURB_TILE_F4; k += 256) {
                const int tile = k / MURB_TILE_F4, in = k % MURB_TILE_F4;
                const float4 v = src[k];
                if (in < MURB_TILE_PAIRS) tileA[tile * MURB_TILE_PAIRS + in] = v;
                else tileB[tile * MURB_TILE_PAIRS + in - MURB_TILE_PAIRS] = v;
            }
        }
        __syncthreads();

        murb_f2 ajx[MURB_SYM_STEPS], ajy[MURB_SYM_STEPS], ajz[MURB_SYM_STEPS];
#pragma unroll
        for (int p = 0; p < MURB_SYM_STEPS; ++p) { ajx[p] = (murb_f2)(0.f); ajy[p] = (murb_f2)(0.f); ajz[p] = (murb_f2)(0.f); }

        const unsigned long out_base = ((unsigned long)out_c * a.irows + (unsigned long)irow) * a.i_stride + out_r;
#pragma unroll 1
        for (int u = u0; u < u1; ++u) {
            asm volatile("" ::: "memory");   // keep the tile reads inside the loop (see murb_force_sym_kernel)
            const unsigned int i_slot = (unsigned int)(u * 4 + wave) * R;   // wave-uniform
            const bool diagonal = (int)(i_slot / MURB_SYM_BLOCK) == J;
            float xi[R], yi[R], zi[R], gi[R];
            {
                const unsigned long ra = murb_rec_a((unsigned long)(i_slot >> 1));
#pragma unroll
                for (int h = 0; h < R / 2; ++h) {
                    const float4 A = a.rec[ra + h];
                    const float4 B = a.rec[ra + h + MURB_TILE_PAIRS];
                    xi[2 * h] = A.x; xi[2 * h + 1] = A.y;
                    yi[2 * h] = A.z; yi[2 * h + 1] = A.w;
                    zi[2 * h] = B.x; zi[2 * h + 1] = B.y;
                    gi[2 * h] = B.z; gi[2 * h + 1] = B.w;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    xi[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xi[r])));
                    yi[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, yi[r])));
                    zi[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, zi[r])));
                    gi[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, gi[r])));
                    gi[r] = diagonal ? 0.f : -gi[r];   // scalar select: the j side of a diagonal group receives nothing
                }
            }
            murb_f2 aix[R], aiy[R], aiz[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { aix[r] = (murb_f2)(0.f); aiy[r] = (murb_f2)(0.f); aiz[r] = (murb_f2)(0.f); }

#pragma unroll
            for (int p = 0; p < MURB_SYM_STEPS; ++p) {
                const float4 A = tileA[p * 64 + lane];
                const float4 B = tileB[p * 64 + lane];
                const murb_f2 xj = {A.x, A.y}, yj = {A.z, A.w}, zj = {B.x, B.y}, gj = {B.z, B.w};
#pragma unroll
                for (int r = 0; r < R; ++r)
                    murb_interact_seg(xj, yj, zj, gj, xi[r], yi[r], zi[r], gi[r], soft2, aix[r], aiy[r], aiz[r], ajx[p],
                                      ajy[p], ajz[p]);
                __builtin_amdgcn_sched_barrier(0);
            }

            float v[12];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                v[3 * r + 0] = aix[r].x + aix[r].y;
                v[3 * r + 1] = aiy[r].x + aiy[r].y;
                v[3 * r + 2] = aiz[r].x + aiz[r].y;
            }
            const float total = murb_reduce12(v, lane);
            a.iplane[out_base + (i_slot - (unsigned int)a.i_slot0)] = total;   // every lane stores (no branch in the loop)
        }

        // j side: combine the four waves in a fixed order (3+2 -> 1+0 -> 0), wave 0 writes the piece's row
        if (wave >= 2) {
#pragma unroll
            for (int p = 0; p < MURB_SYM_STEPS; ++p) {
                scratch[wave - 2][0][p * 64 + lane] = ajx[p];
                scratch[wave - 2][1][p * 64 + lane] = ajy[p];
                scratch[wave - 2][2][p * 64 + lane] = ajz[p];
            }
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int p = 0; p < MURB_SYM_STEPS; ++p) {
                ajx[p] += scratch[wave][0][p * 64 + lane];
                ajy[p] += scratch[wave][1][p * 64 + lane];
                ajz[p] += scratch[wave][2][p * 64 + lane];
            }
        }
        __syncthreads();
        if (wave == 1) {
#pragma unroll
            for (int p = 0; p < MURB_SYM_STEPS; ++p) {
                scratch[0][0][p * 64 + lane] = ajx[p];
                scratch[0][1][p * 64 + lane] = ajy[p];
                scratch[0][2][p * 64 + lane] = ajz[p];
            }
        }
        __syncthreads();
        if (wave == 0) {
            murb_f2* row = reinterpret_cast<murb_f2*>(a.jplane + (unsigned long)jrow * 3 * MURB_SYM_BLOCK);
#pragma unroll
            for (int p = 0; p < MURB_SYM_STEPS; ++p) {
                row[0 * MURB_SYM_PAIRS + p * 64 + lane] = ajx[p] + scratch[0][0][p * 64 + lane];
                row[1 * MURB_SYM_PAIRS + p * 64 + lane] = ajy[p] + scratch[0][1][p * 64 + lane];
                row[2 * MURB_SYM_PAIRS + p * 64 + lane] = ajz[p] + scratch[0][2][p * 64 + lane];
            }
        }
    }
}

#endif
