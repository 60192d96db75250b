// gfx950 (CDNA4) kernels of the all-pairs force + integrate path.  Device code only; included by
// murbhip.hip (the C-ABI translation unit); variants are A/B-timed through the C ABI (tools/ab.py, tools/sweep.py).
//
// What is computed (reference SimulationNBodyOptim.cpp:34-94; device twin
// SimulationNBodyCUDATileFullDevice.cu:110-137):
//     a_i = sum_j  GM_j * (q_j - q_i) * (|q_j - q_i|^2 + soft^2)^(-3/2)
// in fp32, full N^2 form (the j == i term is exactly 0 because q_j - q_i = 0 and soft > 0).
//
// How it is mapped to the machine (this is NOT the reference's one-thread-per-body tiling):
//   * the i bodies of a wavefront are wave-uniform: R of them live in SGPRs (loaded once per sweep,
//     moved with v_readfirstlane), so every VALU instruction reads its i operand from the scalar
//     file for free;
//   * the j bodies are spread over the 64 lanes, two per lane, in the pair layout of murb_layout.h,
//     so the twelve arithmetic instructions of an interaction issue as packed fp32
//     (v_pk_add/v_pk_mul/v_pk_fma_f32) on two interactions at once; only v_rsq_f32 is per element;
//   * j tiles are staged global -> LDS once per workgroup and read back with conflict-free
//     ds_read_b128, one 16-byte read feeding R*2 interactions per lane;
//   * a body's sum is therefore spread over 64 lanes x 2 halves [x jsplit chunks]; it is folded
//     with a wavefront-wide DPP/shuffle reduction at the end of the j sweep and the per-chunk
//     partial sums are added in fixed order by the integrate kernel (bit-reproducible).
// The work per wavefront is R x (chunk length) interactions, so N = 30 000 already yields thousands
// of wavefronts for 256 CUs x 4 SIMDs (the reference's 1024-bodies-per-block tiling gives 30 blocks).
#ifndef MURB_KERNELS_H_
#define MURB_KERNELS_H_

#include <hip/hip_runtime.h>
#include "murb_layout.h"

typedef float murb_f2 __attribute__((ext_vector_type(2)));

// Which tiles of the record buffer a launch sweeps as j: `count` virtual tiles starting at
// `offset`, with a hole of `skip_len` tiles at virtual index `skip_at` (sharded mode sweeps "own
// slice" and "everything but own slice" in two launches so the exchange can overlap the first).
struct MurbTileRange {
    int offset, count, skip_at, skip_len;
};

struct MurbForceArgs {
    const float4* rec;   // body records, all slots (murb_layout.h)
    float4* accp;        // partial sums: accp[chunk * acc_stride + local_slot] = {ax, ay, az, 0}
    MurbTileRange tiles; // j range of this launch
    int i_first_slot;    // first slot of the i slice (tile aligned)
    int chunk_first;     // partial-sum row of this launch's chunk 0
    int nchunks;         // gridDim.y
    unsigned int acc_stride;   // slots per partial-sum row
    float soft2;
};

__device__ __forceinline__ int murb_actual_tile(const MurbTileRange& tr, int v)
{
    return tr.offset + v + (v >= tr.skip_at ? tr.skip_len : 0);
}

// ---- wavefront-wide sum: every lane ends up with the total ------------------------------------
// Rows of 16 lanes are folded with DPP (no LDS traffic), the four row totals are combined through
// v_readlane: 4 DPP adds + 4 readlanes + 3 adds per value.
__device__ __forceinline__ float murb_wave_sum(float v)
{
    // quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// ---- one i body against two j bodies (packed) ---------------------------------------------------
// 3 pk_add + 3 pk_fma + 2 rsq + 3 pk_mul + 3 pk_fma: the instruction mix DESIGN.md's roofline counts.
__device__ __forceinline__ void murb_interact_pk(const murb_f2 xj, const murb_f2 yj, const murb_f2 zj,
                                                 const murb_f2 gj, const float xi, const float yi, const float zi,
                                                 const float soft2, murb_f2& ax, murb_f2& ay, murb_f2& az)
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
    const murb_f2 inv2 = inv * inv;
    const murb_f2 gi = gj * inv;
    const murb_f2 s = gi * inv2;   // GM_j * inv^3, never G*inv^3 alone (fp32 range)
    ax = __builtin_elementwise_fma(s, dx, ax);
    ay = __builtin_elementwise_fma(s, dy, ay);
    az = __builtin_elementwise_fma(s, dz, az);
}

// Same arithmetic, one j at a time (A/B reference for the packed form).
__device__ __forceinline__ void murb_interact_sc(const float xj, const float yj, const float zj, const float gj,
                                                 const float xi, const float yi, const float zi, const float soft2,
                                                 float& ax, float& ay, float& az)
{
    const float dx = xj - xi, dy = yj - yi, dz = zj - zi;
    float r2 = __builtin_fmaf(dx, dx, soft2);
    r2 = __builtin_fmaf(dy, dy, r2);
    r2 = __builtin_fmaf(dz, dz, r2);
    const float inv = __builtin_amdgcn_rsqf(r2);
    const float s = (gj * inv) * (inv * inv);
    ax = __builtin_fmaf(s, dx, ax);
    ay = __builtin_fmaf(s, dy, ay);
    az = __builtin_fmaf(s, dz, az);
}

// Gravitational potential of i due to a j pair: phi_i += GM_j / sqrt(|d|^2 + soft^2) (sign and the
// 1/2 m_i factor are applied by the caller).  Energy diagnostic of the reference's gpu+tracking
// implementation (SimulationNBodyCUDAPropertyTracking.cu:217-304), not on the force path.
__device__ __forceinline__ void murb_interact_phi(const murb_f2 xj, const murb_f2 yj, const murb_f2 zj, const murb_f2 gj,
                                                  const float xi, const float yi, const float zi, const float soft2,
                                                  murb_f2& phi)
{
    const murb_f2 dx = xj - xi, dy = yj - yi, dz = zj - zi;
    murb_f2 r2 = __builtin_elementwise_fma(dx, dx, (murb_f2)(soft2));
    r2 = __builtin_elementwise_fma(dy, dy, r2);
    r2 = __builtin_elementwise_fma(dz, dz, r2);
    murb_f2 inv;
    inv.x = __builtin_amdgcn_rsqf(r2.x);
    inv.y = __builtin_amdgcn_rsqf(r2.y);
    phi = __builtin_elementwise_fma(gj, inv, phi);
}

// Variant tags (template parameter MODE)
#define MURB_MODE_PK_LDS 1      // packed math, j tiles staged in LDS
#define MURB_MODE_PK_DIRECT 2   // packed math, j read straight from L2/HBM per wave
#define MURB_MODE_SC_LDS 3      // scalar math, j tiles staged in LDS
#define MURB_MODE_PHI 4         // potential instead of acceleration (x component of the output), LDS tiles

// ---- force kernel --------------------------------------------------------------------------------
// grid.x = i groups of WAVES*R bodies, grid.y = j chunks.  LDS: STAGE layout tiles (8 KiB each).
// (the body is shared with murb_force_integrate_kernel below; `lds` = the workgroup's STAGE tiles)
template <int MODE, int R, int WAVES, int STAGE>
__device__ __forceinline__ void murb_force_body(const MurbForceArgs& a, float4* lds)
{
    static_assert(R % 2 == 0 && MURB_TILE_BODIES % (WAVES * R) == 0, "i groups must tile the layout");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i_slot = a.i_first_slot + (blockIdx.x * WAVES + wave) * R;   // wave-uniform

    // the wave's R i bodies -> scalar registers
    float xi[R], yi[R], zi[R];
    {
        const unsigned long ra = murb_rec_a((unsigned long)(i_slot >> 1));
#pragma unroll
        for (int h = 0; h < R / 2; ++h) {
            const float4 A = a.rec[ra + h];
            const float4 B = a.rec[ra + h + MURB_TILE_PAIRS];
            xi[2 * h] = A.x; xi[2 * h + 1] = A.y;
            yi[2 * h] = A.z; yi[2 * h + 1] = A.w;
            zi[2 * h] = B.x; zi[2 * h + 1] = B.y;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            xi[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xi[r])));
            yi[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, yi[r])));
            zi[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, zi[r])));
        }
    }

    // this block's j chunk: virtual tiles [vt0, vt1)
    const int chunk = blockIdx.y;
    const int vt0 = (int)(((long)a.tiles.count * chunk) / a.nchunks);
    const int vt1 = (int)(((long)a.tiles.count * (chunk + 1)) / a.nchunks);
    const float soft2 = a.soft2;

    murb_f2 ax[R], ay[R], az[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { ax[r] = (murb_f2)(0.f); ay[r] = (murb_f2)(0.f); az[r] = (murb_f2)(0.f); }

    if (MODE == MURB_MODE_PK_DIRECT) {
        for (int vt = vt0; vt < vt1; ++vt) {
            const float4* tile = a.rec + (unsigned long)murb_actual_tile(a.tiles, vt) * MURB_TILE_F4;
#pragma unroll
            for (int q = 0; q < MURB_TILE_PAIRS; q += 64) {
                const float4 A = tile[q + lane];
                const float4 B = tile[q + lane + MURB_TILE_PAIRS];
                const murb_f2 xj = {A.x, A.y}, yj = {A.z, A.w}, zj = {B.x, B.y}, gj = {B.z, B.w};
#pragma unroll
                for (int r = 0; r < R; ++r) murb_interact_pk(xj, yj, zj, gj, xi[r], yi[r], zi[r], soft2, ax[r], ay[r], az[r]);
            }
        }
    } else {
        for (int vs = vt0; vs < vt1; vs += STAGE) {
            const int nt = (vt1 - vs) < STAGE ? (vt1 - vs) : STAGE;
            __syncthreads();   // previous stage fully consumed
            for (int t = 0; t < nt; ++t) {
                const float4* src = a.rec + (unsigned long)murb_actual_tile(a.tiles, vs + t) * MURB_TILE_F4;
#pragma unroll
                for (int k = threadIdx.x; k < MURB_TILE_F4; k += WAVES * 64) lds[t * MURB_TILE_F4 + k] = src[k];
            }
            __syncthreads();
            for (int t = 0; t < nt; ++t) {
                const float4* tile = lds + t * MURB_TILE_F4;
#pragma unroll
                for (int q = 0; q < MURB_TILE_PAIRS; q += 64) {
                    const float4 A = tile[q + lane];
                    const float4 B = tile[q + lane + MURB_TILE_PAIRS];
                    if (MODE == MURB_MODE_PK_LDS) {
                        const murb_f2 xj = {A.x, A.y}, yj = {A.z, A.w}, zj = {B.x, B.y}, gj = {B.z, B.w};
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            murb_interact_pk(xj, yj, zj, gj, xi[r], yi[r], zi[r], soft2, ax[r], ay[r], az[r]);
                    } else if (MODE == MURB_MODE_PHI) {
                        const murb_f2 xj = {A.x, A.y}, yj = {A.z, A.w}, zj = {B.x, B.y}, gj = {B.z, B.w};
#pragma unroll
                        for (int r = 0; r < R; ++r) murb_interact_phi(xj, yj, zj, gj, xi[r], yi[r], zi[r], soft2, ax[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            float tx = ax[r].x, ty = ay[r].x, tz = az[r].x;
                            float ux = ax[r].y, uy = ay[r].y, uz = az[r].y;
                            murb_interact_sc(A.x, A.z, B.x, B.z, xi[r], yi[r], zi[r], soft2, tx, ty, tz);
                            murb_interact_sc(A.y, A.w, B.y, B.w, xi[r], yi[r], zi[r], soft2, ux, uy, uz);
                            ax[r] = (murb_f2){tx, ux}; ay[r] = (murb_f2){ty, uy}; az[r] = (murb_f2){tz, uz};
                        }
                    }
                }
            }
        }
    }

    // fold the 64 lanes x 2 halves of every accumulator; lane r keeps body r's total
    float ox = 0.f, oy = 0.f, oz = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float sx = murb_wave_sum(ax[r].x + ax[r].y);
        const float sy = murb_wave_sum(ay[r].x + ay[r].y);
        const float sz = murb_wave_sum(az[r].x + az[r].y);
        if (lane == r) { ox = sx; oy = sy; oz = sz; }
    }
    if (lane < R) {
        const unsigned long row = (unsigned long)(a.chunk_first + chunk) * a.acc_stride;
        a.accp[row + (unsigned long)(i_slot - a.i_first_slot) + lane] = make_float4(ox, oy, oz, 0.f);
    }
}

template <int MODE, int R, int WAVES, int STAGE>
__global__ __launch_bounds__(WAVES * 64) void murb_force_kernel(const MurbForceArgs a)
{
    __shared__ float4 lds[(MODE == MURB_MODE_PK_DIRECT) ? 1 : STAGE * MURB_TILE_F4];
    murb_force_body<MODE, R, WAVES, STAGE>(a, lds);
}

// ---- balanced persistent schedule -----------------------------------------------------------------
// The (i group, j tile) units of a launch, linearised group-major (u = g * tiles + t), are cut into
// `nblocks` contiguous, equally long runs: workgroup b owns [begin(b), begin(b+1)).  Every workgroup
// does the same amount of work whatever N is (the 2-D grid above leaves the last scheduling round
// partly empty: N = 30 000 gives 938 i groups for 1024 SIMDs).  A group whose tiles are split over
// several workgroups gets one partial-sum row per workgroup: row = b - block_of(first unit of g).
struct MurbSchedule {
    int groups, tiles, nblocks, row_base;
};

__host__ __device__ __forceinline__ long murb_sched_begin(const MurbSchedule& s, long b)
{
    return ((long)s.groups * (long)s.tiles * b) / (long)s.nblocks;
}

__host__ __device__ __forceinline__ int murb_sched_block_of(const MurbSchedule& s, long u)
{
    const long total = (long)s.groups * (long)s.tiles;
    long b = (u * (long)s.nblocks) / total;
    while (b + 1 < s.nblocks && murb_sched_begin(s, b + 1) <= u) ++b;
    while (b > 0 && murb_sched_begin(s, b) > u) --b;
    return (int)b;
}

// rows a group's partial sums occupy (>= 1)
__host__ __device__ __forceinline__ int murb_sched_rows_of_group(const MurbSchedule& s, int g)
{
    return murb_sched_block_of(s, (long)(g + 1) * s.tiles - 1) - murb_sched_block_of(s, (long)g * s.tiles) + 1;
}

template <int R, int WAVES, int STAGE>
__global__ __launch_bounds__(WAVES * 64) void murb_force_persistent(const MurbForceArgs a, const MurbSchedule s)
{
    static_assert(R % 2 == 0 && MURB_TILE_BODIES % (WAVES * R) == 0, "i groups must tile the layout");
    __shared__ float4 lds[STAGE * MURB_TILE_F4];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float soft2 = a.soft2;
    long u = murb_sched_begin(s, blockIdx.x);
    const long u_end = murb_sched_begin(s, blockIdx.x + 1);

    while (u < u_end) {
        const int g = (int)(u / s.tiles);
        const int t0 = (int)(u - (long)g * s.tiles);
        const int run = (int)((u_end - u) < (long)(s.tiles - t0) ? (u_end - u) : (long)(s.tiles - t0));
        const int i_slot = a.i_first_slot + (g * WAVES + wave) * R;   // wave-uniform

        float xi[R], yi[R], zi[R];
        {
            const unsigned long ra = murb_rec_a((unsigned long)(i_slot >> 1));
#pragma unroll
            for (int h = 0; h < R / 2; ++h) {
                const float4 A = a.rec[ra + h];
                const float4 B = a.rec[ra + h + MURB_TILE_PAIRS];
                xi[2 * h] = A.x; xi[2 * h + 1] = A.y;
                yi[2 * h] = A.z; yi[2 * h + 1] = A.w;
                zi[2 * h] = B.x; zi[2 * h + 1] = B.y;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                xi[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xi[r])));
                yi[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, yi[r])));
                zi[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, zi[r])));
            }
        }
        murb_f2 ax[R], ay[R], az[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { ax[r] = (murb_f2)(0.f); ay[r] = (murb_f2)(0.f); az[r] = (murb_f2)(0.f); }

        for (int vs = t0; vs < t0 + run; vs += STAGE) {
            const int nt = (t0 + run - vs) < STAGE ? (t0 + run - vs) : STAGE;
            __syncthreads();   // previous stage fully consumed
            for (int t = 0; t < nt; ++t) {
                const float4* src = a.rec + (unsigned long)murb_actual_tile(a.tiles, vs + t) * MURB_TILE_F4;
#pragma unroll
                for (int k = threadIdx.x; k < MURB_TILE_F4; k += WAVES * 64) lds[t * MURB_TILE_F4 + k] = src[k];
            }
            __syncthreads();
            for (int t = 0; t < nt; ++t) {
                const float4* tile = lds + t * MURB_TILE_F4;
#pragma unroll
                for (int q = 0; q < MURB_TILE_PAIRS; q += 64) {
                    const float4 A = tile[q + lane];
                    const float4 B = tile[q + lane + MURB_TILE_PAIRS];
                    const murb_f2 xj = {A.x, A.y}, yj = {A.z, A.w}, zj = {B.x, B.y}, gj = {B.z, B.w};
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        murb_interact_pk(xj, yj, zj, gj, xi[r], yi[r], zi[r], soft2, ax[r], ay[r], az[r]);
                }
            }
        }

        float ox = 0.f, oy = 0.f, oz = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float sx = murb_wave_sum(ax[r].x + ax[r].y);
            const float sy = murb_wave_sum(ay[r].x + ay[r].y);
            const float sz = murb_wave_sum(az[r].x + az[r].y);
            if (lane == r) { ox = sx; oy = sy; oz = sz; }
        }
        if (lane < R) {
            const int row = s.row_base + (int)blockIdx.x - murb_sched_block_of(s, (long)g * s.tiles);
            a.accp[(unsigned long)row * a.acc_stride + (unsigned long)(i_slot - a.i_first_slot) + lane] =
                make_float4(ox, oy, oz, 0.f);
        }
        u += run;
    }
}

// ---- integrate ------------------------------------------------------------------------------------
// Reference semantics, Bodies.cpp:260-278 (= CUDABodies.cu:126-153):
//     aDt = a*dt (fp32);  q' = q + (v + aDt*0.5)*dt with the parenthesis and the add evaluated in
//     fp64 (the literal 0.5 is a double) and rounded once to fp32;  v' = v + aDt (fp32).
struct MurbIntegrateArgs {
    const float4* rec_in;    // all slots
    float4* rec_out;         // all slots (other buffer); only the local slice is written
    float4* vel;             // local slice, pair layout
    const float4* accp;      // partial sums [nparts][acc_stride]
    float* acc_out;          // ax | ay | az, acc_stride entries each (sum of the partials)
    int i_first_slot;        // first slot of the local slice
    int count;               // real bodies in the local slice
    int nparts;              // rows to add when nsched == 0
    unsigned int acc_stride;
    float dt;
    int update_state;        // 0: only reduce partial sums into acc_out
    int scheme;              // 0: the reference's update ; 1: leapfrog kick-drift (see murb_integrate_kernel)
    float kick_dt;           // scheme 1: length of the velocity kick ending at this step's mid point
    const float* acc_planes; // accelerations ax | ay | az (acc_stride each): row sums, or a reduce-scatter's output
    const float* acc_planes2;// optional second addend of the same shape (own-slice part that skipped the reduce-scatter)
    const double* acc64;     // accelerations as fp64 sums ax | ay | az (acc64_stride each): multi-pass pair-symmetric evaluation
    unsigned int acc64_stride;
    int nsched;              // persistent launches that produced accp (0, 1 or 2)
    int group_bodies;        // bodies per i group of those launches
    MurbSchedule sched[2];
};

// `#pragma clang fp contract(off)`: hipcc contracts a*b+c into one FMA by default (also through the
// __fmul_rn/__fadd_rn wrappers, which are plain operators in HIP); the reference's x86 build has no
// FMA, so every product here must be rounded on its own or v' = v + a*dt is off by up to 15 ulp.
__device__ __forceinline__ float murb_drift(float q, float v, float a_dt, float dt)
{
#pragma clang fp contract(off)
    const double half_kick = (double)a_dt * 0.5;
    const double vel_mid = (double)v + half_kick;
    const double moved = vel_mid * (double)dt;
    return (float)((double)q + moved);
}

__device__ __forceinline__ float murb_kick(float a, float dt)
{
#pragma clang fp contract(off)
    return a * dt;
}

__device__ __forceinline__ float murb_add_rounded(float v, float a_dt)
{
#pragma clang fp contract(off)
    return v + a_dt;
}

// One pair of local slots (2 lp, 2 lp + 1): partial sums -> accelerations -> state.
__device__ __forceinline__ void murb_integrate_pair(const MurbIntegrateArgs& a, const int lp)
{
#pragma clang fp contract(off)
    const int s0 = 2 * lp;                                    // local slots s0, s0+1
    if (s0 >= (int)a.acc_stride) return;

    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
    if (a.acc64) {
        const unsigned int g0 = (unsigned int)a.i_first_slot + (unsigned int)s0;
        acc0.x = (float)a.acc64[g0]; acc1.x = (float)a.acc64[g0 + 1];
        acc0.y = (float)a.acc64[a.acc64_stride + g0]; acc1.y = (float)a.acc64[a.acc64_stride + g0 + 1];
        acc0.z = (float)a.acc64[2ul * a.acc64_stride + g0]; acc1.z = (float)a.acc64[2ul * a.acc64_stride + g0 + 1];
    } else if (a.acc_planes) {
        acc0.x = a.acc_planes[s0]; acc1.x = a.acc_planes[s0 + 1];
        acc0.y = a.acc_planes[a.acc_stride + s0]; acc1.y = a.acc_planes[a.acc_stride + s0 + 1];
        acc0.z = a.acc_planes[2u * a.acc_stride + s0]; acc1.z = a.acc_planes[2u * a.acc_stride + s0 + 1];
        if (a.acc_planes2) {
            acc0.x += a.acc_planes2[s0]; acc1.x += a.acc_planes2[s0 + 1];
            acc0.y += a.acc_planes2[a.acc_stride + s0]; acc1.y += a.acc_planes2[a.acc_stride + s0 + 1];
            acc0.z += a.acc_planes2[2u * a.acc_stride + s0]; acc1.z += a.acc_planes2[2u * a.acc_stride + s0 + 1];
        }
    } else if (a.nsched == 0) {
        for (int p = 0; p < a.nparts; ++p) {
            const float4 u = a.accp[(unsigned long)p * a.acc_stride + s0];
            const float4 w = a.accp[(unsigned long)p * a.acc_stride + s0 + 1];
            acc0.x += u.x; acc0.y += u.y; acc0.z += u.z;
            acc1.x += w.x; acc1.y += w.y; acc1.z += w.z;
        }
    } else {
        const int g = s0 / a.group_bodies;   // both slots of a pair are in the same group
        for (int k = 0; k < a.nsched; ++k) {
            if (g >= a.sched[k].groups) continue;
            const int rows = murb_sched_rows_of_group(a.sched[k], g);
            for (int p = a.sched[k].row_base; p < a.sched[k].row_base + rows; ++p) {
                const float4 u = a.accp[(unsigned long)p * a.acc_stride + s0];
                const float4 w = a.accp[(unsigned long)p * a.acc_stride + s0 + 1];
                acc0.x += u.x; acc0.y += u.y; acc0.z += u.z;
                acc1.x += w.x; acc1.y += w.y; acc1.z += w.z;
            }
        }
    }
    a.acc_out[s0] = acc0.x; a.acc_out[s0 + 1] = acc1.x;
    a.acc_out[a.acc_stride + s0] = acc0.y; a.acc_out[a.acc_stride + s0 + 1] = acc1.y;
    a.acc_out[2u * a.acc_stride + s0] = acc0.z; a.acc_out[2u * a.acc_stride + s0 + 1] = acc1.z;
    if (!a.update_state) return;

    const unsigned long gp = (unsigned long)(a.i_first_slot >> 1) + lp;   // global pair
    const unsigned long ra = murb_rec_a(gp);
    const unsigned long va = murb_rec_a((unsigned long)lp);
    float4 A = a.rec_in[ra], B = a.rec_in[ra + MURB_TILE_PAIRS];
    float4 VA = a.vel[va], VB = a.vel[va + MURB_TILE_PAIRS];
    const float dt = a.dt;
    if (a.scheme == 1) {
        // Kick-drift-kick leapfrog in its one-force-per-step form: the stored velocity lags the positions
        // by half a step, v_{n+1/2} = v_{n-1/2} + a_n*kick_dt, q_{n+1} = q_n + v_{n+1/2}*dt (the drift keeps
        // the reference integrator's fp64 intermediate).  The closing half kick is applied on read-out
        // (murbhip_download_state).  The formulation is the one the reference states but does not
        // implement (CUDABodies.cu:172-178: its kernels take a_n at the positions of step n-1).
        const float h = a.kick_dt;
        if (s0 < a.count) {
            VA.x = murb_add_rounded(VA.x, murb_kick(acc0.x, h)); VA.z = murb_add_rounded(VA.z, murb_kick(acc0.y, h));
            VB.x = murb_add_rounded(VB.x, murb_kick(acc0.z, h));
            A.x = murb_drift(A.x, VA.x, 0.f, dt); A.z = murb_drift(A.z, VA.z, 0.f, dt); B.x = murb_drift(B.x, VB.x, 0.f, dt);
        }
        if (s0 + 1 < a.count) {
            VA.y = murb_add_rounded(VA.y, murb_kick(acc1.x, h)); VA.w = murb_add_rounded(VA.w, murb_kick(acc1.y, h));
            VB.y = murb_add_rounded(VB.y, murb_kick(acc1.z, h));
            A.y = murb_drift(A.y, VA.y, 0.f, dt); A.w = murb_drift(A.w, VA.w, 0.f, dt); B.y = murb_drift(B.y, VB.y, 0.f, dt);
        }
        a.rec_out[ra] = A; a.rec_out[ra + MURB_TILE_PAIRS] = B;
        a.vel[va] = VA; a.vel[va + MURB_TILE_PAIRS] = VB;
        return;
    }
    if (s0 < a.count) {
        const float kx = murb_kick(acc0.x, dt), ky = murb_kick(acc0.y, dt), kz = murb_kick(acc0.z, dt);
        A.x = murb_drift(A.x, VA.x, kx, dt); A.z = murb_drift(A.z, VA.z, ky, dt); B.x = murb_drift(B.x, VB.x, kz, dt);
        VA.x = murb_add_rounded(VA.x, kx); VA.z = murb_add_rounded(VA.z, ky); VB.x = murb_add_rounded(VB.x, kz);
    }
    if (s0 + 1 < a.count) {
        const float kx = murb_kick(acc1.x, dt), ky = murb_kick(acc1.y, dt), kz = murb_kick(acc1.z, dt);
        A.y = murb_drift(A.y, VA.y, kx, dt); A.w = murb_drift(A.w, VA.w, ky, dt); B.y = murb_drift(B.y, VB.y, kz, dt);
        VA.y = murb_add_rounded(VA.y, kx); VA.w = murb_add_rounded(VA.w, ky); VB.y = murb_add_rounded(VB.y, kz);
    }
    a.rec_out[ra] = A; a.rec_out[ra + MURB_TILE_PAIRS] = B;
    a.vel[va] = VA; a.vel[va + MURB_TILE_PAIRS] = VB;
}

__global__ __launch_bounds__(256) void murb_integrate_kernel(const MurbIntegrateArgs a)
{
    murb_integrate_pair(a, blockIdx.x * blockDim.x + threadIdx.x);
}

// One-sided force launch with the state update in its tail: no second launch.  A step of few bodies on one GPU is two
// DEPENDENT launches of a few microseconds each, and the gap between them is as long as the kernels (N = 2 048: 13 us per
// step for 3 us of arithmetic).  Only where ONE workgroup holds an i group's complete sums (one j chunk: gridDim.y = 1, one
// partial row): it then needs nothing from any other workgroup — a workgroup barrier, and its first threads add the (single)
// row exactly as murb_integrate_kernel does and move the group's bodies (bit-identical).  It writes the OTHER position buffer,
// which nobody reads in this launch.  (MEASURED and dropped: the same for several j chunks or ranks, the last of an i group's
// workgroups — by a ticket counter — doing the update: the device-scope fences this needs on a chip of 8 L2s cost more than
// the launch they save, N = 2 048 13.3 -> 20.2 us, a rank of 8 at N = 30 000 69 -> 118 us.)
template <int R, int WAVES, int STAGE>
__global__ __launch_bounds__(WAVES * 64) void murb_force_integrate_kernel(const MurbForceArgs a, const MurbIntegrateArgs ia)
{
    __shared__ float4 lds[STAGE * MURB_TILE_F4];
    murb_force_body<MURB_MODE_PK_LDS, R, WAVES, STAGE>(a, lds);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();     // the four waves' rows of this i group are written (same CU: visible after the barrier)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (threadIdx.x < WAVES * R / 2) murb_integrate_pair(ia, (int)blockIdx.x * (WAVES * R / 2) + (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// Tracked metrics: the O(N) sums of murbhip_energy / murbhip_moments on the device (the reference reduces
// its per-body energy with cub::DeviceReduce::Sum, SimulationNBodyCUDAPropertyTracking.cu:330-356).
// Per body, in fp64:  kinetic 1/2 m v^2 ; potential -1/2 m (phi - G m / soft) (self term of the sweep
// removed, .cu:287-294) ; m v ; m q x v ; m q ; m.  One block = 256 consecutive slots of the rank's slice;
// the 12 block sums are written in a fixed order (wave shuffle tree, then the 4 waves through LDS) and the
// host adds the few hundred block rows in index order: bit-reproducible.
#define MURB_METRIC_VALUES 12
struct MurbMetricsArgs {
    const float4* rec;      // positions + G*m, all slots
    const float4* vel;      // local slice
    const float* mass;      // local slice, as uploaded
    const float* phi;       // potential sweep output (x plane), or null: potential = 0
    const float* acc;       // ax | ay | az (acc_stride each), used for the closing half kick when half_dt != 0
    double* out;            // [blocks][MURB_METRIC_VALUES]
    int i_first_slot, count;
    unsigned int acc_stride;
    float half_dt;          // leapfrog read-out: v_n = v_{n-1/2} + a * half_dt (0: velocities are current)
    double g_over_soft;     // G / soft
};

__global__ __launch_bounds__(256) void murb_metrics_kernel(const MurbMetricsArgs a)
{
#pragma clang fp contract(off)
    __shared__ double red[3][MURB_METRIC_VALUES];
    const int s = blockIdx.x * 256 + threadIdx.x;   // local slot
    double v[MURB_METRIC_VALUES];
#pragma unroll
    for (int k = 0; k < MURB_METRIC_VALUES; ++k) v[k] = 0.0;
    if (s < a.count) {
        const unsigned long ra = murb_rec_a((unsigned long)((unsigned int)a.i_first_slot + s) >> 1);
        const unsigned long va = murb_rec_a((unsigned long)(s >> 1));
        const int h = s & 1;
        const float4 A = a.rec[ra], B = a.rec[ra + MURB_TILE_PAIRS];
        const float4 VA = a.vel[va], VB = a.vel[va + MURB_TILE_PAIRS];
        float ux = h ? VA.y : VA.x, uy = h ? VA.w : VA.z, uz = h ? VB.y : VB.x;
        if (a.half_dt != 0.f) {   // same rounding as the device kicks
            ux = murb_add_rounded(ux, murb_kick(a.acc[s], a.half_dt));
            uy = murb_add_rounded(uy, murb_kick(a.acc[a.acc_stride + s], a.half_dt));
            uz = murb_add_rounded(uz, murb_kick(a.acc[2u * a.acc_stride + s], a.half_dt));
        }
        const double x = h ? A.y : A.x, y = h ? A.w : A.z, z = h ? B.y : B.x;
        const double m = a.mass[s], wx = ux, wy = uy, wz = uz;
        v[0] = 0.5 * m * (wx * wx + wy * wy + wz * wz);
        v[1] = a.phi ? -0.5 * m * ((double)a.phi[s] - a.g_over_soft * m) : 0.0;
        v[2] = m * wx; v[3] = m * wy; v[4] = m * wz;
        v[5] = m * (y * wz - z * wy); v[6] = m * (z * wx - x * wz); v[7] = m * (x * wy - y * wx);
        v[8] = m * x; v[9] = m * y; v[10] = m * z;
        v[11] = m;
    }
#pragma unroll
    for (int k = 0; k < MURB_METRIC_VALUES; ++k)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0 && wave > 0) {
#pragma unroll
        for (int k = 0; k < MURB_METRIC_VALUES; ++k) red[wave - 1][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < MURB_METRIC_VALUES; ++k)
            a.out[(unsigned long)blockIdx.x * MURB_METRIC_VALUES + k] = ((v[k] + red[0][k]) + red[1][k]) + red[2][k];
    }
}

#endif
