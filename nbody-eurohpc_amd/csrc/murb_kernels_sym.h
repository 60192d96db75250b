// Pair-symmetric force kernel: every unordered body pair is evaluated ONCE and applied to both
// bodies (Newton's third law) — the idea of the reference's cpu+optim loop
// (reference SimulationNBodyOptim.cpp:60-82: `a_i += s*m_j*d ; a_j -= s*m_i*d`), mapped to gfx950.
//
// Why: the one-sided kernel (murb_kernels.h) keeps the vector ALUs 97 % busy; the only way up is
// fewer instructions.  One-sided: 12 packed + 2 rsq wave instructions per (i, j-pair) = 2
// interactions.  Symmetric: 16 packed + 2 rsq per (i, j-pair) = 4 interactions (both directions):
// 20.6 instead of 32.9 issue cycles per interaction.
//
// Decomposition: bodies are cut into blocks of MURB_SYM_BLOCK = 1024 slots (2 layout tiles).  A
// workgroup takes one block pair (I <= J):
//   * the J block (512 pairs) is staged in LDS once; lane l of every wave owns j-pairs l, l+64, ...
//     (8 per lane) and keeps their 24 packed accumulators in VGPRs for the whole item;
//   * the 4 waves split the I (sub-)block: a wave walks its 64/split i groups of R = 4 bodies
//     (coordinates and G*m in SGPRs), 8 fully unrolled steps per group;
//   * after a group the 12 i-side sums (4 bodies x 3) are spread over 64 lanes x 2 halves; they are
//     folded with a transposing reduction (v_permlane32_swap / v_permlane16_swap, then DPP row
//     mirrors with a select): 39 instructions instead of 12 x 11, and written to partial row J;
//   * at the end the four waves' j-side sums are combined through LDS in a fixed order and written
//     to partial row I.  Diagonal items (I == J) evaluate the full square and only keep the i side.
// Partial sums: for every block B the launch owns two small matrices of 1024-slot rows, "i rows" (one per j block
// that some item of block B's bodies walked against: the i-side sums) and "j rows" (one per item that had B as its j
// block: the j-side sums), laid end to end in one buffer, once per component (x, y, z).  The host builds the layout
// together with the item table (murbhip.hip: plan_sym_layout) and hands every item the offsets of its two outputs, so the
// kernel knows nothing about it; every cell has exactly one writer per launch, rows exist only where somebody writes
// (a rank of a multi-GPU run holds just the rows of its own items), and a row sum in fp64 in a fixed order
// (murb_sym_rowsum_*) turns them into accelerations.  No atomics: bit-reproducible.
// The i range of an item is `ngroups` groups per wave (any multiple of 16 bodies with 4 waves): whole blocks for big
// problems, fractions of a block for small ones and for the last items of a launch (shorter drain phase).
#ifndef MURB_KERNELS_SYM_H_
#define MURB_KERNELS_SYM_H_

#include "murb_kernels.h"
#include "murb_sym_types.h"

#ifndef MURB_SYM_STEP_BARRIER
#define MURB_SYM_STEP_BARRIER 1
#endif

struct MurbSymArgs {
    const float4* rec;          // body records (murb_layout.h)
    float* part;                // partial sums, 3 components of comp_stride floats each
    unsigned long comp_stride;
    const MurbSymItem* items;   // one per workgroup
    int item_first;             // first entry of `items` this launch evaluates
    float soft2;
};

__device__ __forceinline__ void murb_swap32(float& a, float& b)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void murb_swap16(float& a, float& b)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
template <int CTRL> __device__ __forceinline__ float murb_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Transposing wave reduction of 12 values.  On return lane L holds the 64-lane total of value
//   idx(L) = (L & 4) ? 8 + 2*b4 + b5 : 4*b3 + 2*b4 + b5      (b3 = bit 3 of L, ...)
// (lanes of one quad hold the same total).
__device__ __forceinline__ float murb_reduce12(float (&v)[12], int lane)
{
    // distance 32 and 16: swap halves / rows between two registers, then one add folds both
    float w[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { murb_swap32(v[2 * k], v[2 * k + 1]); w[k] = v[2 * k] + v[2 * k + 1]; }
    float x[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) { murb_swap16(w[2 * m], w[2 * m + 1]); x[m] = w[2 * m] + w[2 * m + 1]; }
    // distance 8 (row mirror): lanes with bit 3 clear keep x0, the others x1; x2 is folded on all lanes
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
    const float keep3 = b3 ? x[1] : x[0], send3 = b3 ? x[0] : x[1];
    const float y0 = keep3 + murb_dpp<0x140>(send3);          // row_mirror
    const float y1 = x[2] + murb_dpp<0x140>(x[2]);
    // distance 4 (half-row mirror): bit 2 clear keeps y0, set keeps y1
    const float keep2 = b2 ? y1 : y0, send2 = b2 ? y0 : y1;
    float z = keep2 + murb_dpp<0x141>(send2);                 // row_half_mirror
    z += murb_dpp<0x4E>(z);                                   // quad_perm [2,3,0,1]
    z += murb_dpp<0xB1>(z);                                   // quad_perm [1,0,3,2]
    return z;
}

// Scalar helpers kept in inline asm: written in C, LLVM turns "uniform value AND uniform mask" into a vector select
// (v_mov + v_cndmask per use), which costs VALU issue slots in the interaction loop.
__device__ __forceinline__ float murb_masked(float v, unsigned mask)   // SGPR operands: one s_and_b32
{
    float out;
    asm("s_and_b32 %0, %1, %2" : "=s"(out) : "s"(v), "s"(mask) : "scc");
    return out;
}
__device__ __forceinline__ unsigned murb_ones_if_ge(int p, int threshold)   // all ones iff p >= threshold (SGPRs)
{
    unsigned out;
    asm("s_sub_i32 %0, %1, %2\n\ts_ashr_i32 %0, %0, 31" : "=s"(out) : "s"(threshold), "s"(p + 1) : "scc");
    return out;
}

// one i body against a j pair, both directions
__device__ __forceinline__ void murb_interact_sym(const murb_f2 xj, const murb_f2 yj, const murb_f2 zj, const murb_f2 gj,
                                                  const float xi, const float yi, const float zi, const float gi,
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
    const murb_f2 fi = gj * inv3;          // pull of j on i
    const murb_f2 fj = inv3 * (-gi);       // pull of i on j (opposite direction)
    aix = __builtin_elementwise_fma(fi, dx, aix);
    aiy = __builtin_elementwise_fma(fi, dy, aiy);
    aiz = __builtin_elementwise_fma(fi, dz, aiz);
    ajx = __builtin_elementwise_fma(fj, dx, ajx);
    ajy = __builtin_elementwise_fma(fj, dy, ajy);
    ajz = __builtin_elementwise_fma(fj, dz, ajz);
}

// force + pair potential (PHI = 2): the same operations in the same order as murb_interact_sym (the forces of a tracked
// evaluation are bit-identical to an untracked one) plus G m_i G m_j / r of the pair summed per lane: two more packed
// instructions.  What the tracked metrics need is the TOTAL potential energy, not a potential per body: one float per group
// of 4 i bodies instead of a second N^2 sweep.  Items on the DIAGONAL (i block = j block) do not count here: they meet pairs
// from both sides and every body itself, and that self term — (G m)^2 / soft, for the galaxy's central body 10^4 times
// everything around it — would swallow the low bits of its neighbours in an fp32 chain; murb_sym_pe_diag_kernel sums the
// diagonal blocks' pairs separately, in fp64 and without the self terms.
__device__ __forceinline__ void murb_interact_sym_pe(const murb_f2 xj, const murb_f2 yj, const murb_f2 zj, const murb_f2 gj,
                                                     const float xi, const float yi, const float zi, const float gi, const float gi_pe,
                                                     const float soft2, murb_f2& aix, murb_f2& aiy, murb_f2& aiz,
                                                     murb_f2& ajx, murb_f2& ajy, murb_f2& ajz, murb_f2& pe)
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
    const murb_f2 fi = gj * inv3;          // pull of j on i
    const murb_f2 fj = inv3 * (-gi);       // pull of i on j (opposite direction)
    const murb_f2 gg = gj * inv;           // G m_j / r
    pe = __builtin_elementwise_fma(gg, (murb_f2)(gi_pe), pe);
    aix = __builtin_elementwise_fma(fi, dx, aix);
    aiy = __builtin_elementwise_fma(fi, dy, aiy);
    aiz = __builtin_elementwise_fma(fi, dz, aiz);
    ajx = __builtin_elementwise_fma(fj, dx, ajx);
    ajy = __builtin_elementwise_fma(fj, dy, ajy);
    ajz = __builtin_elementwise_fma(fj, dz, ajz);
}

// potential form: phi_i += G m_j / r, phi_j += G m_i / r (8 packed + 2 rsq per i x j-pair)
__device__ __forceinline__ void murb_interact_sym_phi(const murb_f2 xj, const murb_f2 yj, const murb_f2 zj, const murb_f2 gj,
                                                      const float xi, const float yi, const float zi, const float gi,
                                                      const float soft2, murb_f2& phi_i, murb_f2& phi_j)
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
    phi_i = __builtin_elementwise_fma(gj, inv, phi_i);
    phi_j = __builtin_elementwise_fma(inv, (murb_f2)(gi), phi_j);
}

// The walk of one wave over its share of an item's i range: groups of R bodies (coordinates and G*m in SGPRs) against
// the j block in LDS, i-side sums reduced and stored per group, j-side sums left in the caller's registers.
// DYN = 0: every step applies both sides (items off the diagonal; also the plain form of a diagonal item, whose j side
// is then simply not written).  DYN = 1: a diagonal item in its triangular form — the i range covers the j bodies of
// steps [p_first, p_sym): steps before p_first are skipped (those pairs belong to the items of the earlier i ranges,
// which apply both sides), steps in [p_first, p_sym) keep the i side only (their j bodies include the i bodies
// themselves), steps from p_sym on apply both sides.
template <int WAVES, int ILOAD, int PHI, int RED, int DYN>
__device__ __forceinline__ void murb_sym_walk(const MurbSymArgs& a, const float4* tileA, const float4* tileB, float* stage,
                                              const int lane, const int wave, const int groups_per_wave,
                                              const unsigned int i_block_slot, const unsigned long out_off, const int out_stride,
                                              const float pe_scale, const int p_first,
                                              const int p_sym, murb_f2 (&ajx)[MURB_SYM_STEPS], murb_f2 (&ajy)[MURB_SYM_STEPS],
                                              murb_f2 (&ajz)[MURB_SYM_STEPS])
{
    constexpr int R = MURB_SYM_R;
    // PHI = 2 (needs RED = 1): a 13th value per group, the group's pair potential, takes the same way out as the 12 i-side
    // sums — staged, summed over the wave by a team of four lanes, stored (out_off / out_stride point the team at the
    // potential's own area behind the three components: one float per group)
    constexpr int NV = PHI == 2 ? 13 : 12;
    const float soft2 = a.soft2;
    // staging rows of 64 floats at a stride of 80: lane L writes entry L (64 consecutive floats: no bank conflict); team lane
    // q of value k reads the four float4 at entries 16 m + 4 q (m = 0..3): the eight lanes served together (two values x
    // four q) start 4 floats apart within a row and the odd row is 80 = 16 (mod 32) floats further: 32 distinct banks
    const int stage_wr = lane;
    const int stage_rd = ((lane >> 2) < NV ? (lane >> 2) : NV - 1) * 80 + (lane & 3) * 4;
    // RED = 1: four lanes per value add up its 64 staged entries
    const auto team_sum = [&](int g_of) {
        const float4* src = reinterpret_cast<const float4*>(stage + stage_rd);
        const float4 t0 = src[0], t1 = src[4], t2 = src[8], t3 = src[12];
        murb_f2 s2 = (murb_f2){t0.x, t0.y} + (murb_f2){t0.z, t0.w};
        s2 += (murb_f2){t1.x, t1.y}; s2 += (murb_f2){t1.z, t1.w};
        s2 += (murb_f2){t2.x, t2.y}; s2 += (murb_f2){t2.z, t2.w};
        s2 += (murb_f2){t3.x, t3.y}; s2 += (murb_f2){t3.z, t3.w};
        float z = s2.x + s2.y;
        z += murb_dpp<0xB1>(z);                               // quad_perm [1,0,3,2]
        z += murb_dpp<0x4E>(z);                               // quad_perm [2,3,0,1]
        a.part[(unsigned long)out_off + (unsigned long)(g_of * out_stride)] = z;
    };
#pragma unroll 1
    for (int gk = 0; gk < groups_per_wave; ++gk) {
        asm volatile("" ::: "memory");   // keep the tile reads inside the loop: 64 VGPRs of hoisted j data spill
        const int g = gk * WAVES + wave;                        // interleave the waves over the range
        const unsigned int i_slot = i_block_slot + g * R;       // wave-uniform
        float xi[R], yi[R], zi[R], gi[R];
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (ILOAD == 1) {
            typedef const float4 __attribute__((address_space(4))) * murb_cf4p;
            const murb_cf4p crec = (murb_cf4p)a.rec;
            const unsigned long ra = murb_rec_a((unsigned long)(i_slot >> 1));
#pragma unroll
            for (int h = 0; h < R / 2; ++h) {
                const float4 A = crec[ra + h];
                const float4 B = crec[ra + h + MURB_TILE_PAIRS];
                xi[2 * h] = A.x; xi[2 * h + 1] = A.y;
                yi[2 * h] = A.z; yi[2 * h + 1] = A.w;
                zi[2 * h] = B.x; zi[2 * h + 1] = B.y;
                gi[2 * h] = B.z; gi[2 * h + 1] = B.w;
            }
        } else
#endif
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
            }
        }
        murb_f2 aix[R], aiy[R], aiz[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { aix[r] = (murb_f2)(0.f); aiy[r] = (murb_f2)(0.f); aiz[r] = (murb_f2)(0.f); }
        murb_f2 pe = (murb_f2)(0.f);   // PHI = 2: this group's pair potential (chains of 32 terms)

#pragma unroll
        for (int p = 0; p < MURB_SYM_STEPS; ++p) {
            if (DYN == 0 || p >= p_first) {   // wave-uniform
                const float4 A = tileA[p * 64 + lane];
                const float4 B = tileB[p * 64 + lane];
                const murb_f2 xj = {A.x, A.y}, yj = {A.z, A.w}, zj = {B.x, B.y}, gj = {B.z, B.w};
                // scalar mask: 0 where this step's j bodies include the i bodies themselves (their j side is dropped)
                const unsigned both = DYN ? murb_ones_if_ge(p, p_sym) : ~0u;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float gie = DYN ? murb_masked(gi[r], both) : gi[r];
                    if constexpr (PHI == 1)
                        murb_interact_sym_phi(xj, yj, zj, gj, xi[r], yi[r], zi[r], gie, soft2, aix[r], ajx[p]);
                    else if constexpr (PHI == 2)
                        murb_interact_sym_pe(xj, yj, zj, gj, xi[r], yi[r], zi[r], gie, gi[r], soft2, aix[r], aiy[r], aiz[r], ajx[p],
                                             ajy[p], ajz[p], pe);
                    else
                        murb_interact_sym(xj, yj, zj, gj, xi[r], yi[r], zi[r], gie, soft2, aix[r], aiy[r], aiz[r], ajx[p], ajy[p],
                                          ajz[p]);
                }
            }
            if (MURB_SYM_STEP_BARRIER) __builtin_amdgcn_sched_barrier(0);   // one step at a time: bounds the live temporaries
        }

        // i side: 12 sums -> lanes, one float per (body, component) -> the item's i row
        float v[12];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            v[3 * r + 0] = aix[r].x + aix[r].y;
            v[3 * r + 1] = aiy[r].x + aiy[r].y;
            v[3 * r + 2] = aiz[r].x + aiz[r].y;
        }
        // Every lane stores (no branch in the loop body: a conditional store here makes LLVM sink the
        // j-side FMAs into the loop latch and spill 32 x 8 registers).  Lanes that hold the same total
        // write the same value to the same address.
        if constexpr (RED == 0) {
            a.part[(unsigned long)out_off + g * R] = murb_reduce12(v, lane);
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) stage[k * 80 + stage_wr] = v[k];
            if constexpr (PHI == 2) stage[12 * 80 + stage_wr] = (pe.x + pe.y) * pe_scale;   // pe_scale = 0 on the diagonal
            team_sum(g);
        }
    }
}

// grid.x = items; 64 * WAVES threads.  MINW = waves per SIMD the register allocator must allow.
// WAVES = 4: one wave per SIMD and workgroup, four workgroups per CU.  WAVES = 8: two waves per SIMD and
// workgroup, two workgroups per CU — an item takes half as long and a CU's last workgroup still has two
// waves per SIMD to interleave (a lone wave reaches 61 % of the issue rate, tools/sym_stamps.hip), which
// shortens the drain phase of short launches; per item it pays one more combine stage.
// ILOAD = 1 (what the library launches): the i bodies come through scalar loads (s_load_dwordx4 from the
// constant address space, 4 per group) instead of 4 vector loads + 16 v_readfirstlane: 16 VALU issue slots
// less per group of 576, +2.1 % at N=200k and +2.7 % at 30k, bit-identical results.  ILOAD = 0 keeps the vector-load form.
// PHI = 2: force + pair potential in one pass (murb_interact_sym_pe; RED = 1 only): what murbhip_energy runs on a
// pair-symmetric plan.  The buffer then has room for one more float per group behind its three components.
// PHI = 1: the same sweep for the potential (murbhip_energy): phi instead of the three acceleration components,
// written to component 0 only (the cells of components 1 and 2 keep whatever they held; their row sums are not used).
// RED = 0: the 12 i-side sums of a group are folded in registers (murb_reduce12: 39 VALU instructions).  RED = 1: through
// LDS — each lane adds the two halves of its 12 sums and stores them (12 ds_write_b32 into a per-wave area aliased with
// the end-of-item combine scratch, rows of 64 floats at a stride of 80 so that neither the writes nor the reads conflict),
// then teams of four lanes sum one value's 64 entries (4 ds_read_b128 and 7 packed adds per lane, one add, two DPP adds):
// 22 VALU instructions per group.
// (Taking the team sums one group later, in the middle of the next group's sweep, to cover the LDS round trip, measured
// 2 % SLOWER than RED = 1 at N = 200 000 and was dropped.)
template <int MINW, int WAVES = 4, int ILOAD = 0, int PHI = 0, int RED = 0>
__global__ __launch_bounds__(64 * WAVES, MINW) void murb_force_sym_kernel(const MurbSymArgs a)
{
    constexpr int THREADS = 64 * WAVES;
    __shared__ float4 tileA[MURB_SYM_PAIRS];                    // {x0,x1,y0,y1} of the J block
    __shared__ float4 tileB[MURB_SYM_PAIRS];                    // {z0,z1,gm0,gm1}
    __shared__ murb_f2 scratch[WAVES / 2][3][MURB_SYM_PAIRS];   // cross-wave combine of the j-side sums

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // this workgroup's item from the host-built table (uniform address: scalar loads)
    const MurbSymItem it = a.items[a.item_first + blockIdx.x];
    const int J = __builtin_amdgcn_readfirstlane(it.J);
    const int i_item_slot = __builtin_amdgcn_readfirstlane(it.i_slot0);
    const int groups_per_wave = __builtin_amdgcn_readfirstlane(it.ngroups);
    const int flags = __builtin_amdgcn_readfirstlane(it.flags);
    const bool no_j_side = (flags & 1) != 0;            // nothing to write on the j side
    const bool triangular = (flags & 2) != 0;           // diagonal item in its triangular form (murb_sym_walk, DYN = 1)
    const int p_first = (flags >> 8) & 15, p_sym = (flags >> 12) & 15;
#ifdef MURB_LAB_BEGIN     /* tools/sym_stamps.hip: per-workgroup time stamps (never defined in the product build) */
    MURB_LAB_BEGIN();
#endif

    // stage the J block: 2 layout tiles, A records to tileA, B records to tileB
    {
        const float4* src = a.rec + (unsigned long)J * (MURB_SYM_BLOCK / MURB_TILE_BODIES) * MURB_TILE_F4;
#pragma unroll
        for (int k = threadIdx.x; k < 2 * MURB_TILE_F4; k += THREADS) {
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

    // where this lane's i-side total goes: value idx(lane) = 3 * body + component (see murb_reduce12)
    unsigned long out_off;
    int out_stride = MURB_SYM_R;   // floats between two groups' totals
    {
        const int b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
        int idx = b2 ? 8 + 2 * b4 + b5 : 4 * b3 + 2 * b4 + b5;
        if constexpr (RED != 0) idx = (lane >> 2) < 12 ? (lane >> 2) : 11;   // team of four lanes per value; lanes 48-63 repeat value 11
        const int r = idx / 3, c = idx - 3 * r;
        out_off = (unsigned long)c * a.comp_stride + it.ioff + r;
        if constexpr (PHI == 2) {   // lanes 48-63: the group's pair potential, one float per group behind the three components
            static_assert(RED == 1, "the pair potential leaves through the LDS team reduction");
            if ((lane >> 2) >= 12) { out_off = 3ul * a.comp_stride + it.ioff / MURB_SYM_R; out_stride = 1; }
        }
    }
    // RED = 1: this wave's staging area (12 rows of 64 floats at a stride of 80)
    float* const stage = reinterpret_cast<float*>(&scratch[0][0][0]) + wave * ((PHI == 2 ? 13 : 12) * 80);
    if (triangular)
        murb_sym_walk<WAVES, ILOAD, PHI, RED, 1>(a, tileA, tileB, stage, lane, wave, groups_per_wave, (unsigned int)i_item_slot, out_off,
                                                 out_stride, 0.f, p_first, p_sym, ajx, ajy, ajz);
    else
        murb_sym_walk<WAVES, ILOAD, PHI, RED, 0>(a, tileA, tileB, stage, lane, wave, groups_per_wave, (unsigned int)i_item_slot, out_off,
                                                 out_stride, (i_item_slot / MURB_SYM_BLOCK == J) ? 0.f : 1.f, 0, 0, ajx, ajy, ajz);

    // j side: fold the waves pairwise in a fixed order (WAVES = 4: 3+2 -> 1+0 -> 0), wave 0 writes the item's j row
    if (!no_j_side) {
        if constexpr (RED != 0) __syncthreads();   // the scratch doubles as the waves' staging areas
#pragma unroll
        for (int half = WAVES / 2; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half) {
#pragma unroll
                for (int p = 0; p < MURB_SYM_STEPS; ++p) {
                    scratch[wave - half][0][p * 64 + lane] = ajx[p];
                    scratch[wave - half][1][p * 64 + lane] = ajy[p];
                    scratch[wave - half][2][p * 64 + lane] = ajz[p];
                }
            }
            __syncthreads();
            if (wave < half) {
#pragma unroll
                for (int p = 0; p < MURB_SYM_STEPS; ++p) {
                    ajx[p] += scratch[wave][0][p * 64 + lane];
                    ajy[p] += scratch[wave][1][p * 64 + lane];
                    ajz[p] += scratch[wave][2][p * 64 + lane];
                }
            }
            if (half > 1) __syncthreads();
        }
        if (wave == 0) {
            murb_f2* px = reinterpret_cast<murb_f2*>(a.part + it.joff);
            murb_f2* py = reinterpret_cast<murb_f2*>(a.part + a.comp_stride + it.joff);
            murb_f2* pz = reinterpret_cast<murb_f2*>(a.part + 2ul * a.comp_stride + it.joff);
#pragma unroll
            for (int p = 0; p < MURB_SYM_STEPS; ++p) {
                px[p * 64 + lane] = ajx[p];
                if constexpr (PHI != 1) {
                    py[p * 64 + lane] = ajy[p];
                    pz[p * 64 + lane] = ajz[p];
                }
            }
        }
    }
#ifdef MURB_LAB_END
    MURB_LAB_END();
#endif
}

// ---- row sums -------------------------------------------------------------------------------------------
// Row sum of one slot: 64 * MURB_ROWSUM_GROUPS threads = 64 consecutive slots x 16 row groups.  Row group g adds the
// j rows g, g+16, ... and then the i rows g, g+16, ... of the slot's block in fp64; the partial sums are combined in
// a fixed order through LDS.  Returns true on the threads of row group 0, which then hold the totals.
// (16 groups, not 4: at N = 30 000 a block has 30-270 rows of only 1024 slots, and 4 groups left the kernel
// latency-bound at 1.7 TB/s.)
#define MURB_ROWSUM_GROUPS 16
#define MURB_ROWSUM_THREADS (64 * MURB_ROWSUM_GROUPS)
__device__ __forceinline__ bool murb_sym_rowsum_slot(const float* part, unsigned long comp_stride, const MurbSymBlockRows& br,
                                                     unsigned int in_block, int g, int lane,
                                                     double (&red)[MURB_ROWSUM_GROUPS - 1][3][64], double (&total)[3])
{
    double acc[3] = {0.0, 0.0, 0.0};
    for (int idx = g; idx < br.nj; idx += MURB_ROWSUM_GROUPS) {
        const unsigned long o = br.base_j + (unsigned long)idx * MURB_SYM_BLOCK + in_block;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += (double)part[(unsigned long)c * comp_stride + o];
    }
    for (int idx = g; idx < br.ni; idx += MURB_ROWSUM_GROUPS) {
        const unsigned long o = br.base_i + (unsigned long)idx * MURB_SYM_BLOCK + in_block;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += (double)part[(unsigned long)c * comp_stride + o];
    }
    if (g > 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) red[g - 1][c][lane] = acc[c];
    }
    __syncthreads();
    if (g != 0) return false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double t = acc[c];
#pragma unroll
        for (int k = 0; k < MURB_ROWSUM_GROUPS - 1; ++k) t += red[k][c][lane];
        total[c] = t;
    }
    return true;
}

// grid.x = 16 workgroups per table entry (64 slots each).  The table is walked from its END: under the j-major item
// order the last blocks own the most rows (the tail of a launch is cut into finer items, each with a j row of its own), and
// their workgroups should start first.
__global__ __launch_bounds__(MURB_ROWSUM_THREADS) void murb_sym_rowsum_kernel(const float* part, unsigned long comp_stride,
                                                                              const MurbSymBlockRows* rows, float* out,
                                                                              unsigned int out_slice_slots)
{
    __shared__ double red[MURB_ROWSUM_GROUPS - 1][3][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const unsigned int wg = gridDim.x - 1 - blockIdx.x;
    const MurbSymBlockRows br = rows[wg / (MURB_SYM_BLOCK / 64)];
    const unsigned int in_block = (wg % (MURB_SYM_BLOCK / 64)) * 64 + lane;
    double total[3];
    if (!murb_sym_rowsum_slot(part, comp_stride, br, in_block, g, lane, red, total)) return;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        out[((unsigned long)br.out_slice * 3 + c) * out_slice_slots + (unsigned long)br.out_block * MURB_SYM_BLOCK + in_block] =
            (float)total[c];
}

// Multi-pass evaluation (one GPU, rows larger than the budget): the row sums of a pass are ADDED to an fp64 accumulator
// acc64[c * slots + slot]; blocks a pass has no rows for are not in its table and stay untouched.
__global__ __launch_bounds__(MURB_ROWSUM_THREADS) void murb_sym_rowsum_acc_kernel(const float* part, unsigned long comp_stride,
                                                                                  const MurbSymBlockRows* rows, double* acc64,
                                                                                  unsigned int slots)
{
    __shared__ double red[MURB_ROWSUM_GROUPS - 1][3][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const unsigned int wg = gridDim.x - 1 - blockIdx.x;
    const MurbSymBlockRows br = rows[wg / (MURB_SYM_BLOCK / 64)];
    const unsigned int in_block = (wg % (MURB_SYM_BLOCK / 64)) * 64 + lane;
    double total[3];
    if (!murb_sym_rowsum_slot(part, comp_stride, br, in_block, g, lane, red, total)) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) acc64[(unsigned long)c * slots + (unsigned long)br.out_block * MURB_SYM_BLOCK + in_block] += total[c];
}

// The row sum and the state update in ONE launch (a dependent launch costs ~6 us, 3 % of an N = 30 000 step): one
// GPU, everything; a rank of several, the own-slice triangle's rows plus the reduce-scatter's result (a.acc_planes).
// One thread per SLOT here (the stand-alone murb_integrate_kernel has one per pair): the two lanes of a pair read the
// same records and write disjoint halves.  Same arithmetic, same rounding as murb_sym_rowsum_kernel followed by
// murb_integrate_kernel.  The table has one entry per block, in block order.
__global__ __launch_bounds__(MURB_ROWSUM_THREADS) void murb_sym_rowsum_integrate_kernel(const float* part, unsigned long comp_stride,
                                                                                        const MurbSymBlockRows* rows,
                                                                                        const MurbIntegrateArgs a)
{
#pragma clang fp contract(off)
    __shared__ double red[MURB_ROWSUM_GROUPS - 1][3][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const unsigned int wg = gridDim.x - 1 - blockIdx.x;   // last blocks first: they own the most rows
    const unsigned int s = wg * 64 + lane;
    const MurbSymBlockRows br = rows[wg / (MURB_SYM_BLOCK / 64)];
    double total[3];
    if (!murb_sym_rowsum_slot(part, comp_stride, br, s % MURB_SYM_BLOCK, g, lane, red, total)) return;
    float ax = (float)total[0], ay = (float)total[1], az = (float)total[2];
    if (a.acc_planes) {   // a rank of several: the other ranks' (and the own rectangles') share, as reduced and scattered
        ax += a.acc_planes[s]; ay += a.acc_planes[a.acc_stride + s]; az += a.acc_planes[2u * a.acc_stride + s];
    }
    a.acc_out[s] = ax;
    a.acc_out[a.acc_stride + s] = ay;
    a.acc_out[2u * a.acc_stride + s] = az;
    if (!a.update_state || (int)s >= a.count) return;   // padding slots never change (both record buffers hold them)

    const unsigned long ra = murb_rec_a((unsigned long)((unsigned int)a.i_first_slot + s) >> 1);
    const unsigned long va = murb_rec_a((unsigned long)(s >> 1));
    const int h = (int)(s & 1u);
    const float4 A = a.rec_in[ra], B = a.rec_in[ra + MURB_TILE_PAIRS];
    const float4 VA = a.vel[va], VB = a.vel[va + MURB_TILE_PAIRS];
    float x = h ? A.y : A.x, y = h ? A.w : A.z, z = h ? B.y : B.x;
    const float gm = h ? B.w : B.z;
    float vx = h ? VA.y : VA.x, vy = h ? VA.w : VA.z, vz = h ? VB.y : VB.x;
    const float dt = a.dt;
    if (a.scheme == 1) {   // leapfrog kick-drift, see murb_integrate_kernel
        const float k = a.kick_dt;
        vx = murb_add_rounded(vx, murb_kick(ax, k)); vy = murb_add_rounded(vy, murb_kick(ay, k)); vz = murb_add_rounded(vz, murb_kick(az, k));
        x = murb_drift(x, vx, 0.f, dt); y = murb_drift(y, vy, 0.f, dt); z = murb_drift(z, vz, 0.f, dt);
    } else {
        const float kx = murb_kick(ax, dt), ky = murb_kick(ay, dt), kz = murb_kick(az, dt);
        x = murb_drift(x, vx, kx, dt); y = murb_drift(y, vy, ky, dt); z = murb_drift(z, vz, kz, dt);
        vx = murb_add_rounded(vx, kx); vy = murb_add_rounded(vy, ky); vz = murb_add_rounded(vz, kz);
    }
    float* oa = reinterpret_cast<float*>(a.rec_out + ra);
    float* ob = reinterpret_cast<float*>(a.rec_out + ra + MURB_TILE_PAIRS);
    oa[h] = x; oa[2 + h] = y; ob[h] = z; ob[2 + h] = gm;
    float* wa = reinterpret_cast<float*>(a.vel + va);
    float* wb = reinterpret_cast<float*>(a.vel + va + MURB_TILE_PAIRS);
    wa[h] = vx; wa[2 + h] = vy; wb[h] = vz;
}

// The pair potential of the DIAGONAL blocks: sum over the unordered pairs {i, j}, i != j, of a block's bodies of
// G m_i G m_j / sqrt(r^2 + soft^2) — in fp64 from the first addition on (the diagonal is 1/T of the work: N x 512 pair terms).
// Every pair once: body i meets the 512 bodies that follow it round the block, j = i + 1 ... i + 512 (mod 1024); the pair
// {i, i + 512} is met from both ends and counts half.  16 workgroups of 256 threads per block (one would leave a small
// problem on T of the 256 CUs for 100 us): workgroup y takes the bodies 64 y ... 64 y + 63, its wave w the offsets
// 128 w + 1 ... 128 w + 128.  out[16 * block + y]; the host adds them in index order.
#define MURB_PE_DIAG_SPLIT 16
__global__ __launch_bounds__(256) void murb_sym_pe_diag_kernel(const float4* rec, int first_block, float soft2, double* out)
{
    __shared__ float4 body[MURB_SYM_BLOCK];   // x, y, z, G m
    __shared__ double red[4];
    const int t = threadIdx.x, block = first_block + blockIdx.x / MURB_PE_DIAG_SPLIT, y = blockIdx.x % MURB_PE_DIAG_SPLIT;
    for (int k = t; k < MURB_SYM_BLOCK; k += 256) {
        const unsigned long slot = (unsigned long)block * MURB_SYM_BLOCK + k;
        const unsigned long ra = murb_rec_a(slot >> 1);
        const float4 A = rec[ra], B = rec[ra + MURB_TILE_PAIRS];
        body[k] = (k & 1) ? make_float4(A.y, A.w, B.y, B.w) : make_float4(A.x, A.z, B.x, B.z);
    }
    __syncthreads();
    const int i = 64 * y + (t & 63), wave = t >> 6;
    const float4 me = body[i];
    double acc = 0.0;
#pragma unroll 4
    for (int off = 128 * wave + 1; off <= 128 * wave + 128; ++off) {
        const float4 o = body[(i + off) & (MURB_SYM_BLOCK - 1)];
        const float dx = o.x - me.x, dy = o.y - me.y, dz = o.z - me.z;
        const float inv = __builtin_amdgcn_rsqf(fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, soft2))));
        const float term = o.w * inv;
        acc += (double)(off == MURB_SYM_BLOCK / 2 ? 0.5f * term : term);
    }
    acc *= (double)me.w;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((t & 63) == 0) red[wave] = acc;
    __syncthreads();
    if (t == 0) out[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// Sum of the groups' pair potentials of a set of launches (PHI = 2: `count` floats behind the three components of the partial
// rows, zero where no item has groups) in fp64, in a fixed order: thread t adds entries t, t + 1024 x blocks, ...; every
// workgroup folds its 1024 partial sums through LDS and writes one double; the host adds the few hundred of them.
// accumulate: add to what `out` holds (the later passes of a multi-pass evaluation, whose launches share the buffer).
__global__ __launch_bounds__(1024) void murb_sym_pe_sum_kernel(const float* pe, unsigned long count, double* out, int accumulate)
{
    __shared__ double red[1024];
    double acc = 0.0;
    for (unsigned long k = (unsigned long)blockIdx.x * 1024 + threadIdx.x; k < count; k += 1024ul * gridDim.x) acc += (double)pe[k];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int half = 512; half >= 1; half >>= 1) {
        if ((int)threadIdx.x < half) red[threadIdx.x] += red[threadIdx.x + half];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (accumulate ? out[blockIdx.x] : 0.0) + red[0];
}

// Point-to-point form of the reduce-scatter ("exchange_p2p"): out = this rank's own contribution to its slice + the chunks
// received from the ranks that evaluated pairs with it, added in the order of their distance along the ring.
__global__ __launch_bounds__(256) void murb_sym_chunk_sum_kernel(const float* own, const float* received, int nreceived,
                                                                 unsigned int count, float* out)
{
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    float acc = own[k];
    for (int d = 0; d < nreceived; ++d) acc += received[(unsigned long)d * count + k];
    out[k] = acc;
}

// Reduce-scatter by peer reads (one process, several shards): out = sum over shards of their chunk.
struct MurbPeerPtrs {
    const float* p[MURB_SYM_MAX_RANKS];
    int n;
};
__global__ __launch_bounds__(256) void murb_sym_peer_sum_kernel(const MurbPeerPtrs peers, unsigned long chunk_offset,
                                                                unsigned int count, float* out)
{
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    float acc = 0.f;
    for (int r = 0; r < peers.n; ++r) acc += peers.p[r][chunk_offset + k];   // rank order: reproducible
    out[k] = acc;
}

#endif
