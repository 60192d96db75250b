// libmurbhip.so — the C ABI of include/murbhip.h: context, residency, launches, exchange.
// The only translation unit of the product that needs hipcc.
//
// Design notes (full text in DESIGN.md):
//   * body state stays resident in HBM for the whole simulation (reference twin:
//     CUDABodies, src/common/core/CUDABodies.cu:12-49); the only per-iteration host work is
//     enqueueing 2-3 kernels;
//   * positions are double-buffered: a step reads rec[cur] and the integrate kernel writes
//     rec[cur^1], so no kernel ever reads a buffer another kernel (or a peer GPU) is writing;
//   * force kernels: the pair-symmetric kernel (murb_kernels_sym.h; every body pair once, both
//     directions) wherever a GPU gets enough block pairs, the one-sided kernel (murb_kernels.h)
//     otherwise; make_plan() decides, "variant"/"jsplit" override;
//   * multi-GPU: bodies are block-partitioned (murbhip_partition) into equal block-aligned slot
//     ranges of one replicated record buffer.  Half-ring schedule (sym_schedule_items): every pair is
//     evaluated by exactly one rank, ONE reduce-scatter returns each rank the accelerations of its own
//     bodies, ONE in-place all-gather publishes the integrated slice; both run on a second,
//     high-priority stream (RCCL, bound lazily with dlopen, or peer copies/peer reads inside one
//     process) under the two halves of the own-slice triangle (shard_iteration_sym_multi);
//   * one process driving several GPUs (murbhip_create_sharded, `--im hip+tile+multi`): the caller stays single-threaded
//     (reference contract, main.cpp:348-354), but every shard has a host thread of its own inside this library
//     (ShardCrew) that enqueues its device's share of a step — 100-140 us of HIP calls per shard and step, which one
//     thread would serialise to more than the 0.9 ms a rank of 8 computes at N = 200 000
//     (profiles/r03_host_enqueue.txt).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/murbhip.h"
#include "murb_crew.h"
#include "murb_init.h"
#include "murb_kernels_sym.h"
#include "murb_plan.h"
#include "murb_rccl.h"
#include "murb_schedule.h"

namespace {

// ------------------------------------------------------------------------------------ error codes
inline int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : -(int)e; }
inline int nccl_rc(int r) { return r == 0 ? 0 : -(3000 + r); }   // disjoint from -(hipError_t), which reaches past 1000

// Teardown and error paths: a release that fails cannot be acted on (the context is going away either way).
template <typename... P> inline void release(P*... p) { ((void)hipFree((void*)p), ...); }
inline void release_event(hipEvent_t& e) { if (e) (void)hipEventDestroy(e); e = nullptr; }
inline void release_stream(hipStream_t& s) { if (s) (void)hipStreamDestroy(s); s = nullptr; }
inline void drain(hipStream_t s) { if (s) (void)hipStreamSynchronize(s); }

#define HIP_TRY(expr)                        \
    do {                                     \
        const int rc_ = hip_rc((expr));      \
        if (rc_ != 0) return rc_;            \
    } while (0)
#define RC_TRY(expr)                \
    do {                            \
        const int rc_ = (expr);     \
        if (rc_ != 0) return rc_;   \
    } while (0)

// ------------------------------------------------------------------------------------ context
// Partial rows of one group of launches (murb_kernels_sym.h): the buffer, and per block the row table its row sum reads
// (SymPass, the layout and the planner: murb_plan.h).
struct SymSet {
    float* part = nullptr;
    size_t comp_stride = 0;           // floats per component (single pass)
    MurbSymBlockRows* rows = nullptr; // device copy of the table
    int nblocks = 0;                  // entries (all passes)
    std::vector<SymPass> passes;      // more than one entry: multi-pass evaluation
};

struct Shard {
    int device = 0;
    int rank = 0;
    unsigned long first = 0, count = 0;   // global body range owned
    hipStream_t compute = nullptr, comm = nullptr;
    hipStream_t compute_low = nullptr;   // lowest priority: the own-slice triangle in "overlap" mode 2
    hipEvent_t ev_integrated = nullptr, ev_gathered = nullptr, ev_tri = nullptr;
    float4* rec[2] = {nullptr, nullptr};
    float4* vel = nullptr;
    float4* accp = nullptr;      // one-sided kernels: partial-sum rows, allocated on first use (ensure_accp)
    float* acc_out = nullptr;
    float* phi_out = nullptr;    // murbhip_energy's potential sweep (same shape as acc_out), allocated on first use
    float* mass = nullptr;       // masses of the local slice as uploaded (metrics)
    float* radius = nullptr;     // radii of the local slice: only after murbhip_init_bodies (the host never sent them)
    double* metrics = nullptr;   // block sums of murb_metrics_kernel, then murbhip_energy's pair potentials (metrics_doubles)
    double* metrics_host = nullptr;   // its pinned host copy: the read-out is one asynchronous copy behind the kernels
    // pair-symmetric kernel: item table and partial-row layouts (built by build_sym_schedule for one plan)
    MurbSymItem* sym_items = nullptr;
    int sym_items_own = 0, sym_items_total = 0;   // [0, own) = own-slice triangle, the rest need the gathered positions
    int sym_split = 0, sym_waves = 0, sym_taper = -1, sym_diag_tri = -1;   // what the table was built for
    int sym_red = 0;                              // i-side reduction of the plan (kernel template parameter)
    long sym_pass_mb = -1;                        // "sym_pass_mb" the passes were cut for
    int sym_pad_aware = -1;                       // ... and "pad_aware"
    int sym_tri_div = -1;                         // ... and the triangle launches' extra division
    int sym_xcd_order = -1;                       // ... and the item order ("xcd_order")
    int sym_tri_first = -1, sym_overlap = -1;     // ... and the launch boundaries inside the own-slice triangle ("tri_first_pct", "overlap")
    int sym_t1 = 0;                               // items of the triangle's first launch (exchange pipeline, overlap 1)
    bool sym_exchange_mode = false;               // ... and whether it was built for the exchange pipeline
    SymSet sym_main;             // one GPU: every item; exchange pipeline: the rectangles (-> reduce-scatter send chunks)
    SymSet sym_tri;              // exchange pipeline: the own-slice triangle (never enters the reduce-scatter)
    float* sym_send = nullptr;   // [world][3][slice]
    float* sym_recv = nullptr;   // [3][slice]
    float* sym_p2p = nullptr;    // "exchange_p2p": chunks received from the floor(W/2) ranks behind this one [floor(W/2)][3][slice]
    float* sym_tri_acc = nullptr;// row sums of sym_tri [3][slice]
    double* sym_acc64 = nullptr; // multi-pass evaluation: fp64 row sums accumulated over the passes [3][slots]
    size_t sym_bytes = 0;        // device bytes of all of the above
    hipEvent_t ev_rowsum = nullptr, ev_reduced = nullptr;
    rccl_comm_t comm_rccl = nullptr;
    std::vector<hipEvent_t> prof;   // pool of timing events ("profile"): two per recorded span
    size_t prof_used = 0;
    std::vector<int> prof_kind;     // what span k (events 2k, 2k+1) brackets: ProfKind
    unsigned long sym_launches = 0; // pair-symmetric launches of any form since "profile" was last set (force, potential sweep)
    size_t bytes = 0;
};

constexpr int kMaxParts = 64;          // rows of the partial-sum buffer
constexpr int kPeSumBlocks = 256;      // workgroups of murb_sym_pe_sum_kernel
constexpr size_t kProfPairs = 4096;

// What a pair of timing events brackets.  "profile" 1: the force launches only (two event records per launch);
// 2: also the collectives on the exchange stream, the compute stream's waits for them (= the EXPOSED part of the
// exchange) and the compute stream's whole step.
enum ProfKind {
    kProfForce = 0,       // a force launch outside the exchange pipeline (one GPU; the one-sided kernels)
    kProfTri1,            // exchange pipeline: first part of the own-slice triangle (runs under the position gather)
    kProfRect,            // ... rectangles against the other slices
    kProfTri2,            // ... rest of the own-slice triangle (runs under the reduce-scatter)
    kProfReduceScatter,   // exchange stream: from "my send chunks are ready" to "my reduced share has arrived"
    kProfAllGather,       // exchange stream: from "my slice is integrated" to "all slices have arrived"
    kProfWaitGather,      // compute stream: idle in front of the rectangles, waiting for the gathered positions
    kProfWaitReduce,      // compute stream: idle in front of the state update, waiting for the reduced share
    kProfStep,            // compute stream: first launch of a step to the end of its state update
    kProfKinds
};

}  // namespace

struct murbhip_ctx {
    unsigned long n = 0;
    int world = 1;
    unsigned long slice = 0;   // slots per rank
    unsigned long slots = 0;   // world * slice
    float soft2 = 0.f, g = 0.f;
    int exchange = 0;          // 0 peer copies, 1 RCCL
    bool rank_mode = false;    // one shard here, the others live in other processes
    std::vector<Shard> shards;
    ShardCrew* crew = nullptr; // one host thread per shard when this process drives several
    int cur = 0;               // record buffer holding the current positions
    bool uploaded = false;
    bool gather_pending = false;   // an exchange into rec[cur] is in flight on the comm streams
    bool reduce_pending = false;   // peers may still be reading this context's reduce-scatter send buffers
    // options
    int variant = 0, jsplit = 0, profile = 0, overlap = 1;
    int sym_waves = 0;        // pair-symmetric kernel: waves per workgroup, 0 = auto, 4 or 8
    int tri_first_pct = 50;   // overlap 1: share of the own-slice triangle launched BEFORE the rectangles (under the
                              // position gather); the rest runs under the reduce-scatter
    int xcd_order = 0;        // pair-symmetric kernel: 1 = item table interleaved into one run per XCD (measured worse)
    int integrator = 0;       // 0 the reference's update (Bodies.cpp:260-278), 1 kick-drift-kick leapfrog
    bool lf_half = false;     // leapfrog: device velocities lag the positions by half a step of lf_last_dt
    // acceleration cache: murbhip_compute_acc / a leapfrog read-out evaluated the forces at the CURRENT positions
    bool acc_current = false;        // acc_out holds them (a second evaluation would be bit-identical: skip it)
    unsigned long state_serial = 1;  // counts the changes of the body state (upload, device initialisation, every update)
    unsigned long metrics_serial = 0;// the state the cached metric sums below belong to (murbhip_energy and murbhip_moments of one
    bool metrics_with_phi = false;   // tracked iteration share one pass of the metrics kernel and one read-back)
    double metrics_sums[MURB_METRIC_VALUES] = {0};
    bool want_pe = false;            // the force launches being enqueued also sum the pair potential (murbhip_energy)
    bool pe_current = false;         // ... and the partial-row buffers hold it for the current positions
    float lf_last_dt = 0.f;
    int force_exchange = 0;   // run the exchange even with one rank (self-test of the RCCL binding)
    int taper = -1;           // pair-symmetric kernel: % of each launch cut into finer items (-1 = the plan's default)
    int diag_tri = -1;        // ... diagonal blocks as triangular pieces (-1 = the plan's default)
    long sym_pass_mb = 0;     // ... one GPU: budget (MiB) for the partial rows of one pass; 0 = a quarter of the device memory
    int sym_red = -1;         // ... i-side reduction in registers (0) or through LDS (1) (-1 = the plan's default)
    int init_libm_fma = -1;   // murbhip_init_bodies: which build of glibc's sincosf to reproduce (-1 = what this host's libm picks)
    int energy_sweep = 0;
    int fuse_integrate = 1;   // "fuse_integrate": one-sided plan, the state update in the tail of the step's last force launch     // murbhip_energy on a pair-symmetric plan: 1 = the separate potential sweep of rounds 1-2 (kept for the A/B)
    int exchange_p2p = 0;     // RCCL exchange by grouped ncclSend/ncclRecv instead of ncclReduceScatter / ncclAllGather
    int tri_div = 0;          // ... exchange pipeline: the own-slice triangle's items cut into this many parts more (0 = the plan's choice)
    int pad_aware = 1;        // ... 1: padding slots are not walked (murb_schedule.h, sym_orient); 0: every block as if full (A/B)
    int cu_reserve = 0;       // CUs masked out of the compute streams (left free for the collectives' kernels)
    int solo_shard = -1;      // >= 0: only this shard computes (timing aid: one rank's isolated timeline
                              // when W shards share one GPU; results are then meaningless)
    // facts
    int cu_count = 0, clock_mhz = 0;
    size_t device_mem = 0;     // bytes of HBM on the first device
    int last_parts = 0;
    int plan_waves = 4;        // waves per workgroup of the current plan's pair-symmetric launches
    double interactions_per_launch = 0;
    int async_error = 0;
};

namespace {

struct Plan {
    int variant;   // resolved
    int parts_local, parts_remote;   // 2-D grid variants: j chunks of the own-slice launch and of the rest
    bool persistent;                 // balanced persistent schedule (murb_force_persistent)
    bool symmetric;                  // pair-symmetric kernel (murb_force_sym_kernel)
    int split;                       // its i-side sub-blocks per block (1, 2, 4, 8, 16)
    int waves;                       // ... and its waves per workgroup (4 or 8)
    int taper;                       // ... and the share (%) of each launch whose items are cut finer ("taper")
    bool diag_tri;                   // ... diagonal blocks in triangular pieces ("diag_tri")
    int red;                         // ... i-side reduction: 0 registers, 1 LDS teams ("sym_red")
    MurbSchedule sched[2];           // [0] own slice (or everything), [1] the rest
};

constexpr int kPotentialKernel = 100;   // not a selectable variant: murbhip_energy's potential sweep

// "solo_shard" timing aid: every shard but one stays completely idle
inline bool is_idle(const murbhip_ctx* c, const Shard& sh) { return c->solo_shard >= 0 && sh.rank != c->solo_shard; }

template <int MODE, int R, int WAVES, int STAGE>
int launch_force_t(const MurbForceArgs& a, int i_slots, hipStream_t s)
{
    const dim3 grid((unsigned)((i_slots + WAVES * R - 1) / (WAVES * R)), (unsigned)a.nchunks, 1);
    hipLaunchKernelGGL((murb_force_kernel<MODE, R, WAVES, STAGE>), grid, dim3(WAVES * 64), 0, s, a);
    return hip_rc(hipGetLastError());
}

// variant table: id -> (mode, R).  Keep in sync with DESIGN.md 4.1 (tools/ab.py and tools/sweep.py select them by id).
int launch_force(int variant, const MurbForceArgs& a, int i_slots, hipStream_t s)
{
    switch (variant) {
        case 1: return launch_force_t<MURB_MODE_PK_LDS, 8, 4, 4>(a, i_slots, s);
        case 2: return launch_force_t<MURB_MODE_PK_LDS, 4, 4, 4>(a, i_slots, s);
        case 3: return launch_force_t<MURB_MODE_PK_DIRECT, 8, 4, 1>(a, i_slots, s);
        case 4: return launch_force_t<MURB_MODE_SC_LDS, 8, 4, 4>(a, i_slots, s);
        case 5: return launch_force_t<MURB_MODE_PK_LDS, 16, 4, 4>(a, i_slots, s);
        case 6: return launch_force_t<MURB_MODE_PK_DIRECT, 4, 4, 1>(a, i_slots, s);
        case kPotentialKernel:   // few bodies: 2 per wave and 8 waves per workgroup, like murb_force_integrate_kernel (same sums per body)
            return i_slots <= 8192 ? launch_force_t<MURB_MODE_PHI, 2, 8, 4>(a, i_slots, s) : launch_force_t<MURB_MODE_PHI, 8, 4, 4>(a, i_slots, s);
        default: return MURBHIP_E_INVALID;
    }
}
// variant 1 with the state update in its tail (murb_force_integrate_kernel).  Few bodies: fewer i bodies per wave, i.e. more
// and shorter workgroups (N = 2 048 at 8 per wave is 64 workgroups on 256 CUs, each wave walking all j for 8 bodies); the
// sums of a body do not depend on how many others share its wave, so the results stay those of variant 1, bit for bit.
template <int R, int WAVES = 4>
int launch_force_integrate_t(const MurbForceArgs& a, const MurbIntegrateArgs& ia, int i_slots, hipStream_t s)
{
    const dim3 grid((unsigned)((i_slots + WAVES * R - 1) / (WAVES * R)), 1, 1);
    hipLaunchKernelGGL((murb_force_integrate_kernel<R, WAVES, 4>), grid, dim3(64 * WAVES), 0, s, a, ia);
    return hip_rc(hipGetLastError());
}
int launch_force_integrate(const MurbForceArgs& a, const MurbIntegrateArgs& ia, int i_slots, hipStream_t s)
{
    // 8 waves share a workgroup's staged j tiles up to 4 blocks (N = 2 048: 6.9 vs 7.3 us per step, 3 000: 8.6 vs 9.4,
    // 4 096: 10.2 vs 11.4; 5 000: 15.5 vs 14.9, 6 000: 18.2 vs 16.2 — profiles/r03_fused_small_steps.txt)
    if (i_slots <= 4096) return launch_force_integrate_t<2, 8>(a, ia, i_slots, s);
    if (i_slots <= 6144) return launch_force_integrate_t<2>(a, ia, i_slots, s);   // tools/rate_curve.py: 2 per wave wins up to 6 000,
    if (i_slots <= 8192) return launch_force_integrate_t<4>(a, ia, i_slots, s);   // 4 at 7 000 (21.9 vs 22.4 us), all equal from 8 193
    return launch_force_integrate_t<8>(a, ia, i_slots, s);
}
constexpr int kNumVariants = 8;
constexpr int kOneSidedVariant = 1;     // the persistent schedule (7) measured no faster: DESIGN.md §4.1
constexpr int kOneSidedFewBodies = 2;   // 4 i bodies per wave instead of 8: twice the workgroups for a rank's small slice (tools/solo_rank.py,
                                        // round 3, one rank alone: N = 30 000 W = 2/4/8 148 -> 138, 92 -> 86, 69 -> 63 us per step; N = 16 000
                                        // W = 4: 54 -> 43; N = 45 000 W = 8: 108 -> 99; equal from ~20 000 bodies per rank)
constexpr int kPersistentVariant = 7;   // murb_force_persistent<8, 4, 4>
constexpr int kSymmetricVariant = 8;    // murb_force_sym_kernel<4, 4 or 8>
constexpr unsigned long kSymmetricMinBodies = 2049;    // below this (one or two blocks) the one-sided kernel wins; round 2: 10 240 —
                                                       // with items of 64-128 bodies the pair-symmetric kernel is 1.2-1.4x faster
                                                       // from 3 blocks up (tools/small_plan_table.py: N = 3584 14.4 vs 20.5 us per step)
constexpr int kRowsPerLaunch = kMaxParts / 2;

int launch_persistent(const MurbForceArgs& a, const MurbSchedule& sc, hipStream_t s)
{
    hipLaunchKernelGGL((murb_force_persistent<8, 4, 4>), dim3((unsigned)sc.nblocks), dim3(256), 0, s, a, sc);
    return hip_rc(hipGetLastError());
}

// Workgroups of the persistent kernel that fit on the chip at once.
int resident_blocks(const murbhip_ctx* c)
{
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, murb_force_persistent<8, 4, 4>, 256, 0) != hipSuccess || n < 1)
            n = 4;
        per_cu = n;
    }
    return per_cu * std::max(c->cu_count, 1);
}

// Cut groups x tiles units into equal runs: `rounds` runs per resident slot (so that a slot lost to
// another process or to the profiler costs 1/rounds, not 2x), at least ~8 tiles per run (the
// end-of-run reduction is ~1 % of that), and few enough runs that a group spans < kRowsPerLaunch rows.
MurbSchedule make_schedule(const murbhip_ctx* c, long groups, long tiles, int row_base)
{
    MurbSchedule sc{(int)groups, (int)tiles, 1, row_base};
    const long units = groups * tiles;
    if (units <= 0) { sc.nblocks = 0; return sc; }
    const long slots = resident_blocks(c);
    long rounds = c->jsplit > 0 ? c->jsplit : std::min<long>(8, std::max<long>(1, units / (slots * 8)));
    long nb = std::min(units, slots * rounds);
    nb = std::min(nb, std::max<long>(1, (kRowsPerLaunch - 2) * groups));
    sc.nblocks = (int)std::max<long>(nb, 1);
    return sc;
}

int variant_group(int variant)   // bodies per workgroup = waves * R
{
    switch (variant) {
        case kPersistentVariant: return 32;
        case kSymmetricVariant: return 32;
        case 2: case 6: return 16;
        case 5: return 64;
        default: return 32;
    }
}

// How many j chunks: enough workgroups for ~24 scheduling rounds of the chip, but chunks of at
// least 8 tiles (4096 bodies) so the end-of-sweep reduction stays well under 1 % of the sweep.
int auto_parts(const murbhip_ctx* c, int variant, unsigned long i_slots, unsigned long tiles)
{
    if (tiles == 0) return 0;
    const unsigned long groups = (i_slots + variant_group(variant) - 1) / variant_group(variant);
    const unsigned long want_blocks = (unsigned long)std::max(c->cu_count, 1) * 6ul * 24ul;
    unsigned long parts = (want_blocks + groups - 1) / groups;
    parts = std::min(parts, std::max(tiles / 8ul, 1ul));
    parts = std::max(parts, 1ul);
    return (int)std::min<unsigned long>(parts, kMaxParts / 2);
}

// Block pairs a rank evaluates under the (half-ring) pair-symmetric schedule with whole-block items.
long sym_items_per_rank(const murbhip_ctx* c)
{
    const long tb = (long)(c->slice / MURB_SYM_BLOCK), w = c->world;
    return tb * (tb + 1) / 2 + ((w - 1) / 2) * tb * tb + (w > 1 && w % 2 == 0 ? tb * ((tb + 1) / 2) : 0);
}

// One GPU: what the partial rows of one pass may take.  Problems whose rows exceed it are evaluated in several passes
// over ranges of j columns (the rows of N = 1M take 12 GB, of 3.5M 144 GB: half of the HBM).
size_t sym_pass_budget(const murbhip_ctx* c)
{
    if (c->sym_pass_mb > 0) return (size_t)c->sym_pass_mb << 20;
    return c->device_mem ? c->device_mem / 4 : 0;
}

// Bytes of the partial rows of the pair-symmetric kernel on one rank for uniform items of 1024/split bodies
// (12 B per row slot: three components).  One GPU: block b has split*b j rows and T-b i rows.  A rank of W: its
// triangle (tb blocks) plus the rectangles against floor(W/2) slices: tb i rows per own block and split*tb j rows per
// far block at most.
size_t sym_row_bytes(const murbhip_ctx* c, int split)
{
    const size_t tb = c->slice / MURB_SYM_BLOCK, w = (size_t)c->world, far = w / 2;
    size_t rows = tb * tb + (size_t)(split - 1) * tb * (tb - 1) / 2;
    if (w > 1) rows += tb * far * tb + far * tb * (size_t)split * tb;
    return rows * MURB_SYM_BLOCK * 3 * sizeof(float);
}

// One GPU, few bodies: the one-sided kernel with the state update in its tail (murb_force_integrate_kernel: ONE launch per
// step, 2 i bodies per wave) against the pair-symmetric plan's two launches (force, row sum + update), by block count —
// tools/rate_curve.py, us per step, round 3: N = 2 049: 8.9 vs 14.0; 3 000: 9.4 vs 13.9; 3 584: 11.2 vs 14.4; 4 097 (5 blocks):
// 14.2 vs 14.9, 5 000: 15.0 vs 15.1 (a tie: the pair-symmetric plan stays); 6 000 (6 blocks, where the pair-symmetric items
// fall badly on the workgroup slots): 16.2 vs 19.8; 7 000: 21.9 vs 20.2; 8 193: 37 vs 27.
inline bool fused_one_sided_wins(const murbhip_ctx* c)
{
    const unsigned long T = c->slots / MURB_SYM_BLOCK;
    return c->fuse_integrate && c->jsplit == 0 && (T <= 4 || T == 6);
}

Plan make_plan(const murbhip_ctx* c)
{
    Plan p{};
    // variant 0 = auto: pair-symmetric when a GPU gets enough block pairs and its partial rows fit comfortably
    // (they grow as N^2/1024 on one GPU: 0.5 GB at 200k, 12 GB at 1M; a rank of W holds ~1/W of that), else one-sided
    // (one GPU: rows beyond the budget are handled in passes, so only a rank of several has to fit them whole)
    const auto fits = [&](int split) { return c->world == 1 || c->device_mem == 0 || sym_row_bytes(c, split) < c->device_mem / 2; };
    if (c->variant >= 1 && c->variant <= kNumVariants) p.variant = c->variant;
    else if (c->world == 1) p.variant = (c->n >= kSymmetricMinBodies && !fused_one_sided_wins(c) && fits(1)) ? kSymmetricVariant : kOneSidedVariant;
    else p.variant = (sym_items_per_rank(c) >= 400 && fits(1)) ? kSymmetricVariant : (c->slice <= 16384 ? kOneSidedFewBodies : kOneSidedVariant);
    p.symmetric = p.variant == kSymmetricVariant;
    if (p.symmetric) {
        // finer items (i side cut in 2 or 4) until a GPU has ~8 scheduling rounds of them; ~16 in the
        // multi-rank pipeline, whose three force launches per step each end in a tail (measured with
        // tools/solo_profile.py: N=200k, W=2/4/8 -> split 2/4/4 is best)
        const long items = sym_items_per_rank(c);
        const long want = (c->world > 1 ? 16L : 8L) * 4 * std::max(c->cu_count, 1);
        // ... and with few block pairs per rank finer still, so that each of the three launches of a step gets its round of
        // workgroups (tools/solo_rank.py, round 3: N=100k W=8, 689 block pairs: split 8 beats 4 by 3 %; N=60k W=4, 465: by 4.6 %;
        // N=30k W=2, 240: split 16 beats 4 by 5.5 %; from ~1000 block pairs up 4 is best: N=100k W=4, N=200k W=8)
        p.split = (c->jsplit == 1 || c->jsplit == 2 || c->jsplit == 4 || c->jsplit == 8 || c->jsplit == 16)
                      ? c->jsplit
                      : (items >= want ? 1 : (2 * items >= want ? 2 : (c->world == 1 || items >= 1000 ? 4 : (items >= 400 ? 8 : 16))));
        // Rounds 1-2, one GPU below 45 000 bodies (BASELINE's N = 30 000: 465 block pairs for 1024 workgroup slots): 8-wave
        // workgroups (2 per SIMD, 2 workgroups per CU: a CU's last workgroup still has two waves per SIMD to interleave),
        // quarter-block items with the last 30 % of the launch cut finer, diagonal blocks as triangular pieces.
        // tools/ab.py, interleaved, N = 30 000, wall per step: 8 waves / split 8 (round 1) 174.5 us, 8 / 4 / taper 30 /
        // triangular diagonal 170.5, 8 / 2 / taper 60 171.4; 4 waves never better.
        // Round 3 (tools/small_plan_table.py: five plans interleaved for every block count T = 10 ... 44; padding-aware
        // items, measurement without the profiling events): from T = 28 blocks up (N > 27 648) the plan of the larger
        // problems — 4 waves, quarter blocks, 5 % taper, plain diagonal — is the fastest or within 1 % of it (N = 30 000:
        // +3.6 % over the 8-wave plan, interleaved).  Below, the winner follows how the item count falls on the 1024
        // (4 waves) or 512 (8 waves) workgroup slots of the chip, block count by block count, with up to 27 % between the
        // plans at T = 10-16: a table (measured on the 256 CUs of an MI355X; any other CU count keeps the 8-wave plan).
        struct SmallPlan { int waves, split, taper; bool diag_tri; };
        static const SmallPlan kSmallPlans[5] = {{8, 4, 30, true}, {4, 4, 5, false}, {4, 8, 5, false}, {8, 8, 30, true}, {4, 16, 5, false}};
        static const signed char kSmallPlanOfBlocks[25] = {4, 4, 4, 3, 3, 3, 4,                                      // T = 3 ... 9
                                                           3, 3, 0, 2, 2, 1, 3, 0, 2, 1, 2, 0, 1, 2, 0, 1, 0, 2};   // T = 10 ... 27
        const int T = (int)(c->slots / MURB_SYM_BLOCK);
        const bool small = c->world == 1 && T <= 27;
        const SmallPlan sp = kSmallPlans[(small && c->cu_count == 256 && T >= 3) ? kSmallPlanOfBlocks[T - 3] : (T < 10 ? 3 : 0)];
        p.waves = (c->sym_waves == 4 || c->sym_waves == 8) ? c->sym_waves : (small ? sp.waves : 4);
        if (c->jsplit == 0 && c->sym_waves == 0 && small) p.split = sp.split;
        while (p.split > 1 && MURB_SYM_BLOCK / p.split < 16 * p.waves) p.split /= 2;   // an item is at least one group per wave
        while (p.split > 1 && !fits(p.split)) p.split /= 2;   // the rows of the split actually used must fit, too
        // the tail of a launch in finer items (murb_schedule.h): +1.2-1.4 % on the force launch at N = 200 000 with 5 %,
        // nothing at 1M (the tail is 0.3 % of the launch there), and nothing on the wall clock of a rank of 8, whose three
        // short launches gain what their row sums lose to the extra rows
        p.taper = c->taper >= 0 ? c->taper : (c->world > 1 ? 0 : (small ? sp.taper : (c->n <= 600000 ? 5 : 0)));
        p.diag_tri = c->diag_tri >= 0 ? c->diag_tri != 0 : (small && sp.diag_tri);
        // i-side sums through LDS: 599 instead of 616 VALU instructions per group; +0.8-1.3 % at N = 200 000, +1.7 % for
        // a rank of 8 (tools/ab.py)
        p.red = c->sym_red >= 0 ? c->sym_red : 1;
        p.persistent = false;
        p.parts_local = p.parts_remote = 0;
        return p;
    }
    const unsigned long tiles_local = c->slice / MURB_TILE_BODIES;
    const unsigned long tiles_remote = (c->slots - c->slice) / MURB_TILE_BODIES;
    p.persistent = p.variant == kPersistentVariant;
    if (p.persistent) {
        // every shard sweeps the same number of i groups: the largest slice count decides
        unsigned long first, count;
        partition(c->n, c->world, 0, &first, &count);
        const long groups = (long)((count + 31) / 32);
        p.sched[0] = make_schedule(c, groups, (long)tiles_local, 0);
        p.sched[1] = make_schedule(c, groups, (long)tiles_remote, kRowsPerLaunch);
        p.parts_local = p.parts_remote = 0;
        return p;
    }
    if (c->world == 1) {
        p.parts_local = c->jsplit > 0 ? std::min<int>(c->jsplit, (int)std::min<unsigned long>(tiles_local, kMaxParts))
                                      : auto_parts(c, p.variant, c->slice, tiles_local);
        // up to 6 blocks the default one-sided launch keeps all j in one chunk: its workgroups then need nothing from each
        // other and take the state update along (murb_force_integrate_kernel)
        if (c->jsplit == 0 && c->fuse_integrate && p.variant == kOneSidedVariant && c->slots / MURB_SYM_BLOCK <= 6) p.parts_local = 1;
        p.parts_remote = 0;
    } else {
        // split the requested/auto chunk count between the two launches in proportion to their tiles
        const unsigned long tiles_all = tiles_local + tiles_remote;
        int total = c->jsplit > 0 ? c->jsplit : auto_parts(c, p.variant, c->slice, tiles_all);
        total = std::max(total, 2);
        int loc = (int)std::max<unsigned long>(1ul, (unsigned long)total * tiles_local / tiles_all);
        int rem = std::max(1, total - loc);
        p.parts_local = (int)std::min<unsigned long>((unsigned long)loc, std::min<unsigned long>(tiles_local, kMaxParts / 2));
        p.parts_remote = (int)std::min<unsigned long>((unsigned long)rem, std::min<unsigned long>(tiles_remote, kMaxParts / 2));
    }
    return p;
}

// ---- the shards' host threads: ShardCrew (murb_crew.h), one member per shard, bound to its device for life ---------------
// run() hands every member the same job for ITS shard and returns when all have finished enqueueing (never waits for the GPU);
// with one shard there is no thread: the caller runs the job.
int crew_run(murbhip_ctx* c, const std::function<int(Shard&)>& job)
{
    return c->crew->run([&](int i) -> int {
        Shard& sh = c->shards[(size_t)i];
        if (c->crew->threads() == 0) RC_TRY(hip_rc(hipSetDevice(sh.device)));   // a member thread set its device when it started
        return job(sh);
    });
}

// ---- timing spans ("profile") --------------------------------------------------------------------------------------------
// A span = two events recorded on `stream` around something; -1 = not recording (profiling off, level too low, pool empty:
// sampling just stops).  Each shard's spans are recorded by that shard's thread only.
int span_begin(murbhip_ctx* c, Shard& sh, int kind, hipStream_t stream, int* rc)
{
    const int level = (kind <= kProfTri2) ? 1 : 2;
    if (c->profile < level || sh.prof_used + 2 > sh.prof.size()) return -1;
    const int k = (int)(sh.prof_used / 2);
    sh.prof_used += 2;
    sh.prof_kind[(size_t)k] = kind;
    const int r = hip_rc(hipEventRecord(sh.prof[(size_t)2 * k], stream));
    if (r && !*rc) *rc = r;
    return k;
}
int span_end(Shard& sh, int span, hipStream_t stream)
{
    if (span < 0) return 0;
    return hip_rc(hipEventRecord(sh.prof[(size_t)2 * span + 1], stream));
}
// hipStreamWaitEvent as a span: the time the stream actually sat waiting (the exposed part of what it waits for)
int timed_wait(murbhip_ctx* c, Shard& sh, int kind, hipStream_t stream, hipEvent_t ev)
{
    int rc = 0;
    const int sp = span_begin(c, sh, kind, stream, &rc);
    RC_TRY(rc);
    HIP_TRY(hipStreamWaitEvent(stream, ev, 0));
    return span_end(sh, sp, stream);
}

int build_sym_schedule(murbhip_ctx* c, Shard& sh, const Plan& p);
int ensure_accp(murbhip_ctx* c, Shard& sh);
// What a shard's read-out buffer (Shard::metrics) holds, in doubles: the block rows of murb_metrics_kernel, then the pair
// potentials of murbhip_energy — the partial sums of the two sets' groups (murb_sym_pe_sum_kernel) and the own slice's
// diagonal blocks (murb_sym_pe_diag_kernel).  One buffer so that one copy brings a tracked iteration's numbers to the host.
struct MetricsLayout {
    size_t blocks, rows, pe_main, pe_tri, pe_diag, own_blocks, total;
};
MetricsLayout metrics_layout(const murbhip_ctx* c)
{
    MetricsLayout l{};
    l.blocks = (c->slice + 255) / 256;
    l.rows = 0;
    l.pe_main = l.blocks * MURB_METRIC_VALUES;
    l.pe_tri = l.pe_main + kPeSumBlocks;
    l.pe_diag = l.pe_tri + kPeSumBlocks;
    l.own_blocks = c->slice / MURB_SYM_BLOCK;
    l.total = l.pe_diag + l.own_blocks * MURB_PE_DIAG_SPLIT;
    return l;
}

int enqueue_sym_passes(murbhip_ctx* c, Shard& sh, bool potential);
int enqueue_sym_launch(murbhip_ctx* c, Shard& sh, int first, int count, bool own_triangle_rows = false,
                       hipStream_t stream = nullptr, bool potential = false, size_t comp_stride = 0, int kind = kProfForce);

// a fact for murbhip_get_info, noted by the first shard's thread only
inline void note_interactions(murbhip_ctx* c, const Shard& sh, double v) { if (&sh == &c->shards[0]) c->interactions_per_launch = v; }

// Force over the tiles of `which` (0 = own slice / everything when world == 1, 1 = all but own slice).
// `then`: the step's state update, to run in the tail of this launch (the one-sided kernel's default variant only;
// *fused says whether it did — otherwise the caller launches the integrate kernel as usual).
int enqueue_force(murbhip_ctx* c, Shard& sh, const Plan& p, int which, const MurbIntegrateArgs* then = nullptr, bool* fused = nullptr)
{
    MurbForceArgs a{};
    if (!p.symmetric) RC_TRY(ensure_accp(c, sh));
    a.rec = sh.rec[c->cur];
    a.accp = sh.accp;
    a.i_first_slot = (int)((unsigned long)sh.rank * c->slice);
    a.acc_stride = (unsigned int)c->slice;
    a.soft2 = c->soft2;
    const int tiles_local = (int)(c->slice / MURB_TILE_BODIES);
    const int tiles_all = (int)(c->slots / MURB_TILE_BODIES);
    if (which == 0) {
        a.tiles = MurbTileRange{sh.rank * tiles_local, tiles_local, tiles_local, 0};
        a.chunk_first = 0;
        a.nchunks = p.parts_local;
    } else {
        a.tiles = MurbTileRange{0, tiles_all - tiles_local, sh.rank * tiles_local, tiles_local};
        a.chunk_first = p.parts_local;
        a.nchunks = p.parts_remote;
    }
    const int i_slots = (int)sh.count;   // the grid rounds up to whole i groups; the extra slots hold mass 0
    if (p.symmetric) {   // one shard, no exchange: the whole triangle in one launch
        if (which != 0) return 0;
        RC_TRY(build_sym_schedule(c, sh, p));
        if (sh.sym_main.passes.size() > 1) {
            RC_TRY(enqueue_sym_passes(c, sh, false));
            note_interactions(c, sh, (double)c->n * (double)c->n / (double)sh.sym_main.passes.size());
            return 0;
        }
        RC_TRY(enqueue_sym_launch(c, sh, 0, sh.sym_items_total));   // its row sum is fused into the integrate launch
        note_interactions(c, sh, (double)c->n * (double)c->n);
        return 0;
    }
    if (p.persistent) {
        const MurbSchedule& sc = p.sched[which];
        if (sc.nblocks <= 0 || a.tiles.count <= 0) return 0;
        int rc = 0;
        const int sp = span_begin(c, sh, kProfForce, sh.compute, &rc);
        RC_TRY(rc);
        RC_TRY(launch_persistent(a, sc, sh.compute));
        RC_TRY(span_end(sh, sp, sh.compute));
    } else {
        if (a.nchunks <= 0 || a.tiles.count <= 0) return 0;
        int rc = 0;
        const int sp = span_begin(c, sh, kProfForce, sh.compute, &rc);
        RC_TRY(rc);
        if (then && fused && p.variant == kOneSidedVariant && a.nchunks == 1 && then->nparts == 1) {
            RC_TRY(launch_force_integrate(a, *then, i_slots, sh.compute));
            *fused = true;
        } else {
            RC_TRY(launch_force(p.variant, a, i_slots, sh.compute));
        }
        RC_TRY(span_end(sh, sp, sh.compute));
    }
    note_interactions(c, sh, (double)i_slots * (double)a.tiles.count * MURB_TILE_BODIES);
    return 0;
}

// Leapfrog kick length for a step of `dt`: half of it on the first step after an upload, otherwise
// the second half of the previous step's kick plus the first half of this one's.
inline float leapfrog_kick(const murbhip_ctx* c, float dt) { return c->lf_half ? 0.5f * (c->lf_last_dt + dt) : 0.5f * dt; }

// What the state update of a step works on (everything but the source of the accelerations, which the launch sites add).
MurbIntegrateArgs integrate_args(const murbhip_ctx* c, const Shard& sh, int nparts, float dt, int update_state, int scheme = -1,
                                 float* acc_out = nullptr)
{
    MurbIntegrateArgs a{};
    a.scheme = scheme >= 0 ? scheme : c->integrator;
    a.kick_dt = leapfrog_kick(c, dt);
    a.rec_in = sh.rec[c->cur];
    a.rec_out = sh.rec[c->cur ^ 1];
    a.vel = sh.vel;
    a.accp = sh.accp;
    a.acc_out = acc_out ? acc_out : sh.acc_out;
    a.i_first_slot = (int)((unsigned long)sh.rank * c->slice);
    a.count = (int)sh.count;
    a.nparts = nparts;
    a.acc_stride = (unsigned int)c->slice;
    a.dt = dt;
    a.update_state = update_state;
    return a;
}

int enqueue_integrate(murbhip_ctx* c, Shard& sh, int nparts, float dt, int update_state, const Plan* plan = nullptr,
                      int scheme = -1, float* acc_out = nullptr, bool acc_from_out = false)
{
    MurbIntegrateArgs a = integrate_args(c, sh, nparts, dt, update_state, scheme, acc_out);
    if (plan && plan->persistent) {
        a.group_bodies = 32;
        a.sched[0] = plan->sched[0];
        a.nsched = 1;
        if (c->world > 1 && plan->sched[1].nblocks > 0) { a.sched[1] = plan->sched[1]; a.nsched = 2; }
    }
    if (acc_from_out) a.acc_planes = sh.acc_out;   // remembered forces: nothing to sum
    if (plan && plan->symmetric && sh.sym_main.passes.size() > 1) {   // several passes: the sums are in the fp64 accumulator
        a.acc64 = sh.sym_acc64;
        a.acc64_stride = (unsigned int)c->slots;
    } else if (plan && plan->symmetric) {   // one shard, triangular schedule: row sum of the partial rows + update in one launch
        hipLaunchKernelGGL(murb_sym_rowsum_integrate_kernel, dim3((unsigned)(c->slots / 64)), dim3(MURB_ROWSUM_THREADS), 0, sh.compute,
                           sh.sym_main.part, sh.sym_main.comp_stride, sh.sym_main.rows, a);
        return hip_rc(hipGetLastError());
    }
    if (!acc_from_out && !a.acc64) { RC_TRY(ensure_accp(c, sh)); a.accp = sh.accp; }
    const unsigned pairs = (unsigned)(c->slice / 2);
    hipLaunchKernelGGL(murb_integrate_kernel, dim3((pairs + 255) / 256), dim3(256), 0, sh.compute, a);
    return hip_rc(hipGetLastError());
}

// Publish this shard's freshly integrated slice of rec[buf] to all shards and collect theirs (exchange stream).  Runs on
// the shard's own thread; every shard's job calls it (it contains a meet()).  `failed`: an earlier phase of this shard's
// job failed — keep meeting the others, enqueue nothing.
int shard_exchange(murbhip_ctx* c, Shard& sh, int buf, int failed)
{
    const size_t slice_f4 = c->slice;                  // float4 records per slice (1 per body slot)
    const size_t slice_bytes = slice_f4 * sizeof(float4);
    const bool idle = is_idle(c, sh);
    int rc = failed;
    if (!rc && !idle) rc = hip_rc(hipEventRecord(sh.ev_integrated, sh.compute));
    if (c->exchange == 0) c->crew->meet();             // the peers' ev_integrated are recorded
    if (rc) return rc;
    int span = -1;
    if (c->exchange == 1) {
        // one communicator per shard, each driven by its own thread: no ncclGroupStart/End around the calls
        HIP_TRY(hipStreamWaitEvent(sh.comm, sh.ev_integrated, 0));
        span = span_begin(c, sh, kProfAllGather, sh.comm, &rc);
        RC_TRY(rc);
        float4* base = sh.rec[buf];
        if (c->exchange_p2p && c->world > 1) {   // every slice straight to every peer: W - 1 sends and receives, one hop each
            Rccl& r = rccl();
            RC_TRY(nccl_rc(r.GroupStart()));
            for (int d = 1; d < c->world; ++d) {
                const int to = (sh.rank + d) % c->world, from = (sh.rank - d + c->world) % c->world;
                RC_TRY(nccl_rc(r.Send(base + (size_t)sh.rank * slice_f4, slice_f4 * 4, kRcclFloat, to, sh.comm_rccl, sh.comm)));
                RC_TRY(nccl_rc(r.Recv(base + (size_t)from * slice_f4, slice_f4 * 4, kRcclFloat, from, sh.comm_rccl, sh.comm)));
            }
            RC_TRY(nccl_rc(r.GroupEnd()));
        } else {
            RC_TRY(nccl_rc(rccl().AllGather(base + (size_t)sh.rank * slice_f4, base, slice_f4 * 4, kRcclFloat, sh.comm_rccl, sh.comm)));
        }
    } else {
        if (idle) return 0;
        // pull model: each shard copies every peer's slice out of the peer's buffer
        for (Shard& peer : c->shards) {
            if (&peer == &sh || is_idle(c, peer)) continue;
            HIP_TRY(hipStreamWaitEvent(sh.comm, peer.ev_integrated, 0));
        }
        HIP_TRY(hipStreamWaitEvent(sh.comm, sh.ev_integrated, 0));
        span = span_begin(c, sh, kProfAllGather, sh.comm, &rc);
        RC_TRY(rc);
        for (Shard& peer : c->shards) {
            if (&peer == &sh) continue;
            const size_t off = (size_t)peer.rank * slice_f4;
            if (peer.device == sh.device)
                HIP_TRY(hipMemcpyAsync(sh.rec[buf] + off, peer.rec[buf] + off, slice_bytes, hipMemcpyDeviceToDevice, sh.comm));
            else
                HIP_TRY(hipMemcpyPeerAsync(sh.rec[buf] + off, sh.device, peer.rec[buf] + off, peer.device, slice_bytes, sh.comm));
        }
    }
    RC_TRY(span_end(sh, span, sh.comm));
    if (!idle) HIP_TRY(hipEventRecord(sh.ev_gathered, sh.comm));
    return 0;
}

// The one-sided kernels' partial-sum rows: kMaxParts rows of float4 per local slot, zeroed once (rows a launch does
// not write must read as 0).  Not needed by the pair-symmetric plan, so only allocated when a one-sided launch, the
// one-sided potential sweep or murbhip_integrate_host_acc asks for it.
int ensure_accp(murbhip_ctx* c, Shard& sh)
{
    if (sh.accp) return 0;
    const size_t bytes = (size_t)kMaxParts * c->slice * sizeof(float4);
    HIP_TRY(hipMalloc((void**)&sh.accp, bytes));
    HIP_TRY(hipMemsetAsync(sh.accp, 0, bytes, sh.compute));
    sh.bytes += bytes;
    return 0;
}

void free_sym_set(SymSet& st)
{
    release(st.part, st.rows);
    st = SymSet{};
}

int upload_sym_set(Shard& sh, SymSet& st, const std::vector<MurbSymBlockRows>& table, size_t floats)
{
    st.comp_stride = floats;
    st.nblocks = (int)table.size();
    if (floats == 0) return 0;
    // three components, and behind them one float per group of 4 i bodies for the pair potential of a tracked evaluation
    // (murb_kernels_sym.h, PHI = 2: entry ioff / 4 + group; zero wherever no item has groups)
    const size_t bytes = (3 * floats + floats / MURB_SYM_R + 1) * sizeof(float);
    HIP_TRY(hipMalloc((void**)&st.part, bytes));
    HIP_TRY(hipMemsetAsync(st.part, 0, bytes, sh.compute));   // every cell has a writer; zero anyway (on OUR stream: non-blocking w.r.t. stream 0)
    HIP_TRY(hipMalloc((void**)&st.rows, table.size() * sizeof(MurbSymBlockRows)));
    HIP_TRY(hipMemcpy(st.rows, table.data(), table.size() * sizeof(MurbSymBlockRows), hipMemcpyHostToDevice));
    sh.sym_bytes += bytes + table.size() * sizeof(MurbSymBlockRows);
    return 0;
}

// ---- pair-symmetric schedule over several ranks ("half ring") --------------------------------------
// Rank r evaluates, once each, the block pairs of (own slice x own slice) and of (own slice x slice
// r+d) for d = 1 .. floor(W/2); for even W the pair of slices half a ring apart is shared: the lower
// rank takes the first half of ITS blocks against all of the other's, the higher rank the rest.  Every
// unordered body pair is evaluated by exactly one rank.  A rank's partial sums for ALL slices it
// touched are then row-summed into one chunk per slice and combined with ONE reduce-scatter (each
// rank receives the complete accelerations of its own bodies); positions travel as before.
// Exchange pipeline: the own-slice triangle is T_s (T_s + 1) / 2 block pairs in TWO launches (one under each collective); with
// few blocks per slice neither fills the chip's 4 x CUs workgroup slots and both run at a fraction of the issue rate.
// Their items (and only theirs) are cut finer until each launch has ~2 rounds of them ("tri_div" overrides).
int plan_tri_div(const murbhip_ctx* c, const Plan& p)
{
    if (c->tri_div > 0) return c->tri_div;
    if (c->world == 1) return 1;
    const long tb = (long)(c->slice / MURB_SYM_BLOCK);
    const long items = tb * (tb + 1) / 2 * p.split / 2;             // per triangle launch
    const long slots = 4L * std::max(c->cu_count, 1) * 4 / p.waves;  // resident workgroups
    int div = 1;
    while (div < 4 && items * div < 2 * slots && MURB_SYM_BLOCK / (p.split * div * 2) >= 16 * p.waves) div *= 2;
    return div;
}

// false when the shard's tables were built for exactly this plan and these options.  A rebuild of EXISTING tables needs
// every shard of the process drained first (the peer-read sums of the previous step may still be reading this shard's
// send buffer): enqueue_iteration and murbhip_energy do that on the caller's thread before the shards' threads start.
bool sym_schedule_stale(const murbhip_ctx* c, const Shard& sh, const Plan& p)
{
    const bool exchange_mode = c->world > 1 || c->force_exchange;
    return !(sh.sym_items && sh.sym_split == p.split && sh.sym_waves == p.waves && sh.sym_taper == p.taper && sh.sym_diag_tri == (int)p.diag_tri &&
             sh.sym_exchange_mode == exchange_mode && sh.sym_xcd_order == c->xcd_order && sh.sym_pass_mb == c->sym_pass_mb &&
             sh.sym_pad_aware == c->pad_aware && sh.sym_tri_div == plan_tri_div(c, p) &&
             (!exchange_mode || (sh.sym_tri_first == c->tri_first_pct && sh.sym_overlap == c->overlap)));
}

int build_sym_schedule(murbhip_ctx* c, Shard& sh, const Plan& p)
{
    const bool exchange_mode = c->world > 1 || c->force_exchange;
    sh.sym_red = p.red;
    if (!sym_schedule_stale(c, sh, p)) return 0;
    release(sh.sym_items); sh.sym_items = nullptr;
    free_sym_set(sh.sym_main);
    free_sym_set(sh.sym_tri);
    sh.bytes -= sh.sym_bytes;
    sh.sym_bytes = 0;

    SymHostLayout L;
    plan_sym_layout(c->world, sh.rank, sym_fill(c->n, c->world, c->pad_aware != 0), p.split, p.waves, p.taper, p.diag_tri, exchange_mode,
                    c->overlap, c->tri_first_pct, c->xcd_order != 0, sym_pass_budget(c) / (3 * sizeof(float)), L, plan_tri_div(c, p));
    if (!exchange_mode && L.passes.size() == 1 && (int)L.table_main.size() != (int)(c->slots / MURB_SYM_BLOCK))
        return MURBHIP_E_STATE;   // the fused row sum + integrate walks every block
    RC_TRY(upload_sym_set(sh, sh.sym_tri, L.table_tri, L.floats_tri));
    RC_TRY(upload_sym_set(sh, sh.sym_main, L.table_main, L.floats_main));
    sh.sym_main.passes = L.passes;
    sh.sym_pass_mb = c->sym_pass_mb;
    sh.sym_pad_aware = c->pad_aware;
    sh.sym_tri_div = plan_tri_div(c, p);
    if (L.passes.size() > 1 && !sh.sym_acc64) {
        HIP_TRY(hipMalloc((void**)&sh.sym_acc64, 3 * c->slots * sizeof(double)));
        sh.bytes += 3 * c->slots * sizeof(double);
    }
    const std::vector<MurbSymItem>& items = L.items;
    const size_t own_pieces = (size_t)L.own;
    const int W = c->world;
    sh.sym_items_own = (int)own_pieces;
    sh.sym_items_total = (int)items.size();
    sh.sym_split = p.split;
    sh.sym_waves = p.waves;
    sh.sym_taper = p.taper;
    sh.sym_diag_tri = (int)p.diag_tri;
    sh.sym_xcd_order = c->xcd_order;
    sh.sym_exchange_mode = exchange_mode;
    sh.sym_tri_first = c->tri_first_pct;
    sh.sym_overlap = c->overlap;
    sh.sym_t1 = L.t1;
    HIP_TRY(hipMalloc((void**)&sh.sym_items, items.size() * sizeof(MurbSymItem)));
    HIP_TRY(hipMemcpy(sh.sym_items, items.data(), items.size() * sizeof(MurbSymItem), hipMemcpyHostToDevice));
    sh.sym_bytes += items.size() * sizeof(MurbSymItem);
    if (exchange_mode) {
        const size_t chunk = (size_t)3 * c->slice * sizeof(float);
        if (!sh.sym_send) {
            HIP_TRY(hipMalloc((void**)&sh.sym_send, chunk * W));
            HIP_TRY(hipMalloc((void**)&sh.sym_recv, chunk));
            HIP_TRY(hipMalloc((void**)&sh.sym_tri_acc, chunk));
            if (W > 1) {   // receive area of the point-to-point exchange (read as 0 where nothing has arrived yet)
                HIP_TRY(hipMalloc((void**)&sh.sym_p2p, chunk * (size_t)(W / 2)));
                HIP_TRY(hipMemsetAsync(sh.sym_p2p, 0, chunk * (size_t)(W / 2), sh.compute));
                sh.bytes += chunk * (size_t)(W / 2);
            }
            HIP_TRY(hipEventCreateWithFlags(&sh.ev_rowsum, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&sh.ev_reduced, hipEventDisableTiming));
            sh.bytes += chunk * (W + 2);
        }
        // chunks of slices this rank has no rows for are never written by the row sum: they must read as 0
        HIP_TRY(hipMemsetAsync(sh.sym_send, 0, chunk * W, sh.compute));
    }
    sh.bytes += sh.sym_bytes;
    return 0;
}

int enqueue_sym_launch(murbhip_ctx* c, Shard& sh, int first, int count, bool own_triangle_rows, hipStream_t stream,
                       bool potential, size_t comp_stride, int kind)
{
    if (count <= 0) return 0;
    if (!stream) stream = sh.compute;
    ++sh.sym_launches;
    const SymSet& st = own_triangle_rows ? sh.sym_tri : sh.sym_main;
    MurbSymArgs sa{};
    sa.rec = sh.rec[c->cur];
    sa.part = st.part;
    sa.comp_stride = comp_stride ? comp_stride : st.comp_stride;
    sa.items = sh.sym_items;
    sa.item_first = first;
    sa.soft2 = c->soft2;
    const bool with_pe = !potential && c->want_pe;           // force + pair potential in one pass (murb_kernels_sym.h, PHI = 2)
    const bool timed = stream == sh.compute && !potential;   // the profiling events live on the main compute stream
    int rc_span = 0;
    const int sp = timed ? span_begin(c, sh, kind, stream, &rc_span) : -1;
    RC_TRY(rc_span);
    const dim3 grid((unsigned)count);
    if (sh.sym_waves == 8) {
        if (potential) hipLaunchKernelGGL((murb_force_sym_kernel<4, 8, 1, 1>), grid, dim3(512), 0, stream, sa);
        else if (with_pe) hipLaunchKernelGGL((murb_force_sym_kernel<4, 8, 1, 2, 1>), grid, dim3(512), 0, stream, sa);
        else if (sh.sym_red == 1) hipLaunchKernelGGL((murb_force_sym_kernel<4, 8, 1, 0, 1>), grid, dim3(512), 0, stream, sa);
        else hipLaunchKernelGGL((murb_force_sym_kernel<4, 8, 1>), grid, dim3(512), 0, stream, sa);
    } else {
        if (potential) hipLaunchKernelGGL((murb_force_sym_kernel<4, 4, 1, 1>), grid, dim3(256), 0, stream, sa);
        else if (with_pe) hipLaunchKernelGGL((murb_force_sym_kernel<4, 4, 1, 2, 1>), grid, dim3(256), 0, stream, sa);
        else if (sh.sym_red == 1) hipLaunchKernelGGL((murb_force_sym_kernel<4, 4, 1, 0, 1>), grid, dim3(256), 0, stream, sa);
        else hipLaunchKernelGGL((murb_force_sym_kernel<4, 4, 1>), grid, dim3(256), 0, stream, sa);
    }
    RC_TRY(hip_rc(hipGetLastError()));
    RC_TRY(span_end(sh, sp, stream));
    return 0;
}

// row sum of a set's partial rows into `out` (chunks of [3][out_slice_slots])
int enqueue_sym_rowsum(const SymSet& st, float* out, unsigned int out_slice_slots, hipStream_t stream)
{
    if (st.nblocks <= 0) return 0;
    hipLaunchKernelGGL(murb_sym_rowsum_kernel, dim3((unsigned)st.nblocks * (MURB_SYM_BLOCK / 64)), dim3(MURB_ROWSUM_THREADS), 0, stream,
                       st.part, st.comp_stride, st.rows, out, out_slice_slots);
    return hip_rc(hipGetLastError());
}

// The reduce-scatter of the send chunks on the exchange stream: every rank ends up with the other ranks' (and its own
// rectangles') contributions to its own bodies in sym_recv.  Called by every shard's thread after its row sum has been
// enqueued (ev_rowsum recorded; with peer copies: after the meet() that follows).
int shard_reduce_scatter(murbhip_ctx* c, Shard& sh)
{
    const unsigned int chunk_floats = (unsigned int)(3 * c->slice);
    const bool idle = is_idle(c, sh);
    int rc = 0, span = -1;
    if (c->exchange == 1 && c->exchange_p2p && c->world > 1) {
        // Point-to-point form.  Under the half-ring schedule a rank only has contributions for the floor(W/2) slices ahead of
        // it (and its own): the reduce-scatter moves and adds zeros for the rest.  Here every rank sends those chunks straight
        // to their owners (one xGMI hop each, all links at once) and adds up what the floor(W/2) ranks behind it sent.
        Rccl& r = rccl();
        const int W = c->world, D = W / 2;
        HIP_TRY(hipStreamWaitEvent(sh.comm, sh.ev_rowsum, 0));
        span = span_begin(c, sh, kProfReduceScatter, sh.comm, &rc);
        RC_TRY(rc);
        RC_TRY(nccl_rc(r.GroupStart()));
        for (int d = 1; d <= D; ++d) {
            const int to = (sh.rank + d) % W, from = (sh.rank - d + W) % W;
            RC_TRY(nccl_rc(r.Send(sh.sym_send + (size_t)to * chunk_floats, chunk_floats, kRcclFloat, to, sh.comm_rccl, sh.comm)));
            RC_TRY(nccl_rc(r.Recv(sh.sym_p2p + (size_t)(d - 1) * chunk_floats, chunk_floats, kRcclFloat, from, sh.comm_rccl, sh.comm)));
        }
        RC_TRY(nccl_rc(r.GroupEnd()));
        hipLaunchKernelGGL(murb_sym_chunk_sum_kernel, dim3((chunk_floats + 255) / 256), dim3(256), 0, sh.comm,
                           sh.sym_send + (size_t)sh.rank * chunk_floats, sh.sym_p2p, D, chunk_floats, sh.sym_recv);
        RC_TRY(hip_rc(hipGetLastError()));
    } else if (c->exchange == 1) {
        HIP_TRY(hipStreamWaitEvent(sh.comm, sh.ev_rowsum, 0));
        span = span_begin(c, sh, kProfReduceScatter, sh.comm, &rc);
        RC_TRY(rc);
        RC_TRY(nccl_rc(rccl().ReduceScatter(sh.sym_send, sh.sym_recv, chunk_floats, kRcclFloat, kRcclSum, sh.comm_rccl, sh.comm)));
    } else {
        if (idle) return 0;
        // Under the half-ring schedule only the floor(W/2) ranks BEHIND this one (and the rank itself) hold contributions to its
        // slice: the chunks the others keep for it are zero and are not read (no xGMI traffic for zeros).  Fixed order: own,
        // then by distance along the ring.
        const int W = c->world, D = W / 2;
        MurbPeerPtrs peers{};
        peers.n = 0;
        for (int d = 0; d <= D; ++d) {
            const int from = (sh.rank - d + W) % W;
            if (d > 0 && from == sh.rank) break;
            for (Shard& peer : c->shards)
                if (peer.rank == from) { peers.p[peers.n++] = peer.sym_send; HIP_TRY(hipStreamWaitEvent(sh.comm, peer.ev_rowsum, 0)); }
        }
        span = span_begin(c, sh, kProfReduceScatter, sh.comm, &rc);
        RC_TRY(rc);
        hipLaunchKernelGGL(murb_sym_peer_sum_kernel, dim3((chunk_floats + 255) / 256), dim3(256), 0, sh.comm, peers,
                           (unsigned long)sh.rank * chunk_floats, chunk_floats, sh.sym_recv);
        RC_TRY(hip_rc(hipGetLastError()));
    }
    RC_TRY(span_end(sh, span, sh.comm));
    if (!idle) HIP_TRY(hipEventRecord(sh.ev_reduced, sh.comm));
    return 0;
}

// nobody may still be reading our send buffer: the peer-read sums of the previous step (one process), or our own
// previous reduce-scatter (RCCL reads it on the exchange stream)
int wait_send_buffer_free(murbhip_ctx* c, Shard& sh)
{
    if (!c->reduce_pending) return 0;
    if (c->exchange == 0) {   // the readers of this shard's send buffer: itself and the floor(W/2) ranks AHEAD of it (shard_reduce_scatter)
        const int W = c->world, D = W / 2;
        for (Shard& peer : c->shards)
            if ((peer.rank - sh.rank + W) % W <= D) HIP_TRY(hipStreamWaitEvent(sh.compute, peer.ev_reduced, 0));
    } else HIP_TRY(hipStreamWaitEvent(sh.compute, sh.ev_reduced, 0));
    return 0;
}

// One iteration under the half-ring schedule: ONE shard's share, enqueued by that shard's own thread (ShardCrew).  On the
// compute stream unless noted:
//   T1  first part of the own-slice triangle          (needs no remote data: overlaps the position gather)
//       wait: positions of the previous step gathered
//   R   rectangles against the other slices            -> the rectangles' rows (sym_main)
//   SR  row sum of those rows -> send chunks           (own-slice chunk = i-side sums of the rectangles)
//       [exchange stream] reduce-scatter of the chunks -> recv          (overlaps T2)
//   T2  rest of the own-slice triangle                 -> the triangle's rows (sym_tri)
//       wait: reduce-scatter done
//   I   row sum of the triangle's rows + recv, state update: one launch ; then [exchange stream] all-gather of the new positions
// The triangle never enters the reduce-scatter (it only touches the rank's own bodies), which is what
// lets half of it hide the collective's latency.
// The context's fields (cur, gather_pending, ...) are read-only while the shards' threads run; the caller updates them
// afterwards.  meet() only where a stream must wait for an event ANOTHER shard's thread records (peer-copy exchange): the
// shards otherwise never wait for each other on the host.
int shard_iteration_sym_multi(murbhip_ctx* c, Shard& sh, const Plan& p, float dt, int update_state)
{
    const bool idle = is_idle(c, sh);   // "solo_shard" timing aid: idle shards enqueue no work of their own
    const bool copies = c->exchange == 0;
    int rc = build_sym_schedule(c, sh, p);
    const int own = sh.sym_items_own, t1 = sh.sym_t1;
    int step_span = -1;
    if (!rc && !idle) rc = [&]() -> int {
        int r = 0;
        step_span = span_begin(c, sh, kProfStep, sh.compute, &r);
        RC_TRY(r);
        if (c->overlap == 2) {
            // the whole own-slice triangle on a second, lowest-priority compute stream: it runs alone while
            // the positions are still being gathered, then fills the gaps and the tail of the rectangles
            if (c->gather_pending || c->reduce_pending) HIP_TRY(hipStreamWaitEvent(sh.compute_low, sh.ev_integrated, 0));
            RC_TRY(enqueue_sym_launch(c, sh, 0, own, true, sh.compute_low));
            RC_TRY(enqueue_sym_rowsum(sh.sym_tri, sh.sym_tri_acc, (unsigned int)c->slice, sh.compute_low));
            HIP_TRY(hipEventRecord(sh.ev_tri, sh.compute_low));
        }
        RC_TRY(enqueue_sym_launch(c, sh, 0, t1, true, nullptr, false, 0, kProfTri1));
        if (c->gather_pending) RC_TRY(timed_wait(c, sh, kProfWaitGather, sh.compute, sh.ev_gathered));
        RC_TRY(enqueue_sym_launch(c, sh, own, sh.sym_items_total - own, false, nullptr, false, 0, kProfRect));
        RC_TRY(wait_send_buffer_free(c, sh));
        RC_TRY(enqueue_sym_rowsum(sh.sym_main, sh.sym_send, (unsigned int)c->slice, sh.compute));
        HIP_TRY(hipEventRecord(sh.ev_rowsum, sh.compute));
        // what ONE launch covers on average: the rank's share of the step over its non-empty force launches
        const int launches = (t1 > 0) + (own - t1 > 0 || c->overlap == 2) + (sh.sym_items_total - own > 0);
        note_interactions(c, sh, (double)sh.count * (double)c->n / (double)std::max(launches, 1));
        return 0;
    }();
    if (copies) c->crew->meet();   // every shard's ev_rowsum is recorded
    if (!rc) rc = shard_reduce_scatter(c, sh);
    if (!rc && !idle) rc = [&]() -> int {
        // meanwhile: the rest of the own-slice triangle and its row sum
        if (c->overlap == 2) {
            HIP_TRY(hipStreamWaitEvent(sh.compute, sh.ev_tri, 0));
        } else {
            RC_TRY(enqueue_sym_launch(c, sh, t1, own - t1, true, nullptr, false, 0, kProfTri2));
        }
        RC_TRY(timed_wait(c, sh, kProfWaitReduce, sh.compute, sh.ev_reduced));
        MurbIntegrateArgs a{};
        a.rec_in = sh.rec[c->cur];
        a.rec_out = sh.rec[c->cur ^ 1];
        a.vel = sh.vel;
        a.accp = sh.accp;
        a.acc_out = sh.acc_out;
        a.acc_planes = sh.sym_recv;
        a.scheme = c->integrator;
        a.kick_dt = leapfrog_kick(c, dt);
        a.i_first_slot = (int)((unsigned long)sh.rank * c->slice);
        a.count = (int)sh.count;
        a.acc_stride = (unsigned int)c->slice;
        a.dt = dt;
        a.update_state = update_state;
        if (c->overlap == 2) {   // the triangle's row sums were taken on the other stream
            a.acc_planes2 = sh.sym_tri_acc;
            hipLaunchKernelGGL(murb_integrate_kernel, dim3((unsigned)((c->slice / 2 + 255) / 256)), dim3(256), 0, sh.compute, a);
        } else {                 // row sum of the triangle's rows + the reduced share + state update in one launch
            hipLaunchKernelGGL(murb_sym_rowsum_integrate_kernel, dim3((unsigned)(c->slice / 64)), dim3(MURB_ROWSUM_THREADS), 0, sh.compute,
                               sh.sym_tri.part, sh.sym_tri.comp_stride, sh.sym_tri.rows, a);
        }
        RC_TRY(hip_rc(hipGetLastError()));
        RC_TRY(span_end(sh, step_span, sh.compute));
        if (!update_state) HIP_TRY(hipEventRecord(sh.ev_integrated, sh.compute));   // else shard_exchange records it
        return 0;
    }();
    if (update_state) rc = shard_exchange(c, sh, c->cur ^ 1, rc);
    return rc;
}

// One GPU, several passes: every pass's items into the shared row buffer, its row sums added to the fp64 accumulator.
int enqueue_sym_passes(murbhip_ctx* c, Shard& sh, bool potential)
{
    HIP_TRY(hipMemsetAsync(sh.sym_acc64, 0, 3 * c->slots * sizeof(double), sh.compute));
    // force + pair potential (murbhip_energy): every pass has a layout of its own in the shared buffer, so the plane of the
    // groups' potentials — one float per group of the pass's i rows, zero elsewhere — is cleared before the pass and summed
    // right after it, into the same doubles of the read-out buffer pass after pass
    const bool with_pe = !potential && c->want_pe && sh.metrics;
    bool first = true;
    for (const SymPass& ps : sh.sym_main.passes) {
        float* const pe_plane = sh.sym_main.part + 3 * ps.floats;
        const size_t pe_count = ps.floats / MURB_SYM_R + 1;
        if (with_pe) HIP_TRY(hipMemsetAsync(pe_plane, 0, pe_count * sizeof(float), sh.compute));
        RC_TRY(enqueue_sym_launch(c, sh, ps.item_first, ps.item_count, false, nullptr, potential, ps.floats));
        if (with_pe) {
            hipLaunchKernelGGL(murb_sym_pe_sum_kernel, dim3(kPeSumBlocks), dim3(1024), 0, sh.compute, pe_plane, (unsigned long)pe_count,
                               sh.metrics + metrics_layout(c).pe_main, first ? 0 : 1);
            RC_TRY(hip_rc(hipGetLastError()));
            first = false;
        }
        hipLaunchKernelGGL(murb_sym_rowsum_acc_kernel, dim3((unsigned)ps.table_count * (MURB_SYM_BLOCK / 64)), dim3(MURB_ROWSUM_THREADS), 0,
                           sh.compute, sh.sym_main.part, ps.floats, sh.sym_main.rows + ps.table_first, sh.sym_acc64, (unsigned int)c->slots);
        RC_TRY(hip_rc(hipGetLastError()));
    }
    return 0;
}

// The potential sweep of murbhip_energy under the half-ring schedule: the same items as a force evaluation in the
// kernel's PHI form (phi_i += G m_j / r and phi_j += G m_i / r per pair, once), the same reduce-scatter — half the
// pair terms of a one-sided sweep per rank.  No overlap games here: triangle, rectangles, row sums, reduce-scatter,
// phi = received + own triangle.  A collective in one-process-per-GPU mode, like a step.  One shard's share, on its thread.
int shard_potential_sym_multi(murbhip_ctx* c, Shard& sh, const Plan& p)
{
    const bool idle = is_idle(c, sh);
    int rc = build_sym_schedule(c, sh, p);
    if (!rc && !idle) rc = [&]() -> int {
        if (c->gather_pending) HIP_TRY(hipStreamWaitEvent(sh.compute, sh.ev_gathered, 0));
        RC_TRY(enqueue_sym_launch(c, sh, 0, sh.sym_items_own, true, nullptr, true));
        RC_TRY(enqueue_sym_launch(c, sh, sh.sym_items_own, sh.sym_items_total - sh.sym_items_own, false, nullptr, true));
        RC_TRY(enqueue_sym_rowsum(sh.sym_tri, sh.sym_tri_acc, (unsigned int)c->slice, sh.compute));
        RC_TRY(wait_send_buffer_free(c, sh));
        RC_TRY(enqueue_sym_rowsum(sh.sym_main, sh.sym_send, (unsigned int)c->slice, sh.compute));
        HIP_TRY(hipEventRecord(sh.ev_rowsum, sh.compute));
        return 0;
    }();
    if (c->exchange == 0) c->crew->meet();
    if (!rc) rc = shard_reduce_scatter(c, sh);
    if (rc || idle) return rc;
    HIP_TRY(hipStreamWaitEvent(sh.compute, sh.ev_reduced, 0));
    MurbIntegrateArgs a{};   // no state update: phi_out = received + own triangle (component 0 is the potential)
    a.rec_in = sh.rec[c->cur];
    a.rec_out = sh.rec[c->cur ^ 1];
    a.vel = sh.vel;
    a.acc_out = sh.phi_out;
    a.acc_planes = sh.sym_recv;
    a.acc_planes2 = sh.sym_tri_acc;
    a.i_first_slot = (int)((unsigned long)sh.rank * c->slice);
    a.count = (int)sh.count;
    a.acc_stride = (unsigned int)c->slice;
    a.update_state = 0;
    hipLaunchKernelGGL(murb_integrate_kernel, dim3((unsigned)((c->slice / 2 + 255) / 256)), dim3(256), 0, sh.compute, a);
    return hip_rc(hipGetLastError());
}

// One iteration with the one-sided kernels (or with one shard and no exchange): one shard's share, on its thread.
int shard_iteration_plain(murbhip_ctx* c, Shard& sh, const Plan& p, float dt, int update_state, bool reuse)
{
    const bool exchange = update_state && (c->world > 1 || c->force_exchange);
    int rc = 0;
    if (!is_idle(c, sh)) rc = [&]() -> int {   // timing aid: see "solo_shard"
        // one-sided kernel, one GPU, one j chunk: the state update rides in the tail of the force launch (murb_force_integrate_kernel)
        bool fused = false;
        MurbIntegrateArgs then{};
        const bool may_fuse = c->fuse_integrate && !reuse && !p.symmetric && !p.persistent && p.variant == kOneSidedVariant &&
                              c->world == 1 && p.parts_local + p.parts_remote == 1;
        if (may_fuse) {
            RC_TRY(ensure_accp(c, sh));
            then = integrate_args(c, sh, p.parts_local + p.parts_remote, dt, update_state);
        }
        const MurbIntegrateArgs* const tail = may_fuse ? &then : nullptr;
        if (c->world == 1 || reuse) {   // reuse: the forces at these positions are in acc_out, only the update is left
            if (c->gather_pending) HIP_TRY(hipStreamWaitEvent(sh.compute, sh.ev_gathered, 0));
            if (!reuse) RC_TRY(enqueue_force(c, sh, p, 0, tail, &fused));
        } else if (c->overlap) {
            RC_TRY(enqueue_force(c, sh, p, 0));   // own slice: written by our own integrate, already ordered
            if (c->gather_pending) RC_TRY(timed_wait(c, sh, kProfWaitGather, sh.compute, sh.ev_gathered));
            RC_TRY(enqueue_force(c, sh, p, 1, tail, &fused));
        } else {
            if (c->gather_pending) RC_TRY(timed_wait(c, sh, kProfWaitGather, sh.compute, sh.ev_gathered));
            RC_TRY(enqueue_force(c, sh, p, 0));
            RC_TRY(enqueue_force(c, sh, p, 1, tail, &fused));
        }
        if (fused) return 0;
        return enqueue_integrate(c, sh, p.parts_local + p.parts_remote, dt, update_state, reuse ? nullptr : &p, -1, nullptr, reuse);
    }();
    if (exchange) rc = shard_exchange(c, sh, c->cur ^ 1, rc);
    return rc;
}

int enqueue_iteration(murbhip_ctx* c, float dt, int update_state)
{
    const Plan p = make_plan(c);
    c->last_parts = p.parts_local + p.parts_remote;
    c->plan_waves = p.symmetric ? p.waves : 4;
    // forces at the current positions are already in acc_out (compute_acc, or a leapfrog read-out, just ran): an
    // evaluation needs nothing at all, a state update (one shard, no exchange) only the integrate launch
    const bool have_acc = c->acc_current;
    const bool have_pe = c->pe_current;
    c->acc_current = false;
    c->pe_current = false;
    if (have_acc && !update_state && (!c->want_pe || have_pe)) { c->acc_current = true; c->pe_current = have_pe; return 0; }
    const bool exchanging = c->world > 1 || c->force_exchange;
    // ... with several shards: the integrate launch from the remembered forces and the position exchange
    const bool reuse = have_acc && update_state && c->solo_shard < 0;
    if (p.symmetric) {
        bool stale = false;
        for (const Shard& sh : c->shards) stale = stale || (sh.sym_items && sym_schedule_stale(c, sh, p));
        if (stale) RC_TRY(murbhip_sync(c));   // tables are rebuilt below: nothing may be in flight
    }
    if (p.symmetric && exchanging && !reuse)
        RC_TRY(crew_run(c, [&](Shard& sh) { return shard_iteration_sym_multi(c, sh, p, dt, update_state); }));
    else
        RC_TRY(crew_run(c, [&](Shard& sh) { return shard_iteration_plain(c, sh, p, dt, update_state, reuse); }));
    if (p.symmetric && exchanging && !reuse) c->reduce_pending = true;
    if (update_state) {
        if (exchanging) c->gather_pending = true;
        c->cur ^= 1;
        ++c->state_serial;
    } else {
        c->acc_current = true;
        c->pe_current = c->want_pe && p.symmetric;
    }
    return 0;
}

// The compute streams of a shard.  reserve > 0: created with a CU mask that leaves out the `reserve` highest-numbered
// CUs.  The force kernels fill every CU they may use (4 waves per SIMD, 120 VGPRs each), and stream priority does
// not pre-empt resident workgroups: a collective's kernel would otherwise wait for a workgroup to retire before it can
// start.  Bit b of the mask is CU b / 8 of XCD b % 8 (the driver deals the mask's bits to the XCDs round-robin), so
// 8 reserved CUs are one per XCD, 16 two per XCD.  hipExtStreamCreateWithCUMask has no flags argument: such a stream
// has default priority and the default (blocking with respect to stream 0) flag — the library never uses stream 0, but a
// host application that does (torch's default stream) then synchronises with the force kernels implicitly
// (include/murbhip.h, "cu_reserve").  The low-priority stream of "overlap" 2 stays unmasked so that it keeps its priority.
int create_compute_streams(const murbhip_ctx* c, Shard& sh, int reserve)
{
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    HIP_TRY(hipStreamCreateWithPriority(&sh.compute_low, hipStreamNonBlocking, least));
    if (reserve > 0 && c->cu_count > reserve) {
        std::vector<uint32_t> mask((size_t)(c->cu_count + 31) / 32, 0u);
        for (int b = 0; b < c->cu_count - reserve; ++b) mask[(size_t)b / 32] |= 1u << (b % 32);
        HIP_TRY(hipExtStreamCreateWithCUMask(&sh.compute, (uint32_t)mask.size(), mask.data()));
        return 0;
    }
    HIP_TRY(hipStreamCreateWithFlags(&sh.compute, hipStreamNonBlocking));
    return 0;
}

int create_common(murbhip_ctx** out, unsigned long n, float soft, float g, int world, int nlocal, const int* devices,
                  const int* ranks, int exchange, bool rank_mode)
{
    if (!out || n == 0 || world < 1 || world > MURB_SYM_MAX_RANKS || nlocal < 1 || !(soft == soft)) return MURBHIP_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MURBHIP_E_NO_DEVICE;
    for (int i = 0; i < nlocal; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) return MURBHIP_E_INVALID;
    if (slice_slots(n, world) * (unsigned long)world > 0x7fffffffUL) return MURBHIP_E_INVALID;

    murbhip_ctx* c = new (std::nothrow) murbhip_ctx;
    if (!c) return MURBHIP_E_NOMEM;
    c->n = n;
    c->world = world;
    c->slice = slice_slots(n, world);
    c->slots = c->slice * (unsigned long)world;
    c->soft2 = soft * soft;
    c->g = g;
    c->exchange = exchange;
    c->rank_mode = rank_mode;
    c->shards.resize(nlocal);

    hipDeviceProp_t prop;
    int rc = hip_rc(hipGetDeviceProperties(&prop, devices[0]));
    if (rc == 0) { c->cu_count = prop.multiProcessorCount; c->clock_mhz = prop.clockRate / 1000; c->device_mem = prop.totalGlobalMem; }

    for (int i = 0; rc == 0 && i < nlocal; ++i) {
        Shard& sh = c->shards[i];
        sh.device = devices[i];
        sh.rank = ranks[i];
        partition(n, world, sh.rank, &sh.first, &sh.count);
        if ((rc = hip_rc(hipSetDevice(sh.device)))) break;
        if ((rc = create_compute_streams(c, sh, 0))) break;
        // the exchange stream gets the highest priority: its (few, small) collective kernels must be
        // dispatched as soon as they are ready although the force kernel keeps every CU full
        {
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            if ((rc = hip_rc(hipStreamCreateWithPriority(&sh.comm, hipStreamNonBlocking, greatest)))) break;
        }
        if ((rc = hip_rc(hipEventCreateWithFlags(&sh.ev_tri, hipEventDisableTiming)))) break;
        if ((rc = hip_rc(hipEventCreateWithFlags(&sh.ev_integrated, hipEventDisableTiming)))) break;
        if ((rc = hip_rc(hipEventCreateWithFlags(&sh.ev_gathered, hipEventDisableTiming)))) break;
        const size_t rec_bytes = c->slots * sizeof(float4);
        const size_t vel_bytes = c->slice * sizeof(float4);
        const size_t acco_bytes = 3 * c->slice * sizeof(float);
        if ((rc = hip_rc(hipMalloc((void**)&sh.rec[0], rec_bytes)))) break;
        if ((rc = hip_rc(hipMalloc((void**)&sh.rec[1], rec_bytes)))) break;
        if ((rc = hip_rc(hipMalloc((void**)&sh.vel, vel_bytes)))) break;
        if ((rc = hip_rc(hipMalloc((void**)&sh.acc_out, acco_bytes)))) break;
        if ((rc = hip_rc(hipMemset(sh.acc_out, 0, acco_bytes)))) break;
        sh.bytes = 2 * rec_bytes + vel_bytes + acco_bytes;
    }
    // peer access for the copy exchange between distinct devices
    if (rc == 0 && world > 1 && !rank_mode && exchange == 0) {
        for (Shard& a : c->shards)
            for (Shard& b : c->shards)
                if (a.device != b.device) {
                    int can = 0;
                    if ((rc = hip_rc(hipSetDevice(a.device)))) break;
                    if (hipDeviceCanAccessPeer(&can, a.device, b.device) == hipSuccess && can) {
                        hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
                        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) rc = hip_rc(e);
                        (void)hipGetLastError();
                    }
                }
    }
    if (rc != 0) { murbhip_destroy(c); return rc; }
    {   // after the shards exist: the members bind to their devices at once
        std::vector<int> devs;
        for (const Shard& sh : c->shards) devs.push_back(sh.device);
        c->crew = new (std::nothrow) ShardCrew((int)devs.size(), [devs](int i) { (void)hipSetDevice(devs[(size_t)i]); });
    }
    if (!c->crew) { murbhip_destroy(c); return MURBHIP_E_NOMEM; }
    *out = c;
    return 0;
}

// host SoA -> pair records for all slots
void pack_records(const murbhip_ctx* c, const float* x, const float* y, const float* z, const float* w, float scale_w,
                  bool w_present, unsigned long first_body, unsigned long nbodies, unsigned long first_slot,
                  std::vector<float4>& out)
{
    // writes bodies [first_body, first_body + nbodies) at slots first_slot..; caller zero-fills `out`
    for (unsigned long k = 0; k < nbodies; ++k) {
        const unsigned long slot = first_slot + k, body = first_body + k;
        const unsigned long ra = murb_rec_a(slot >> 1);
        float* A = reinterpret_cast<float*>(&out[ra]);
        float* B = reinterpret_cast<float*>(&out[ra + MURB_TILE_PAIRS]);
        const int h = (int)(slot & 1);
        A[h] = x[body];
        A[2 + h] = y[body];
        B[h] = z[body];
        B[2 + h] = w_present ? scale_w * w[body] : 0.f;
    }
    (void)c;
}

}  // namespace

// ===================================================================================== C ABI
extern "C" {

int murbhip_version(void) { return 103; }   // 1.03: murbhip_warmup, murbhip_init_bodies, the potential out of the force evaluation

const char* murbhip_error_string(int code)
{
    static thread_local char buf[160];
    if (code == 0) return "success";
    if (code == MURBHIP_E_INVALID) return "murbhip: invalid argument";
    if (code == MURBHIP_E_STATE) return "murbhip: call made in the wrong state (upload first?)";
    if (code == MURBHIP_E_NO_DEVICE) return "murbhip: no usable HIP device";
    if (code == MURBHIP_E_NO_RCCL) return "murbhip: librccl could not be loaded";
    if (code == MURBHIP_E_NOMEM) return "murbhip: host allocation failed";
    if (code <= -3000 && code > -4000) {
        Rccl& r = rccl();
        snprintf(buf, sizeof buf, "RCCL error %d: %s", -code - 3000,
                 (r.ok && r.GetErrorString) ? r.GetErrorString(-code - 3000) : "?");
        return buf;
    }
    if (code < 0 && code > -2000) {
        snprintf(buf, sizeof buf, "HIP error %d: %s", -code, hipGetErrorString((hipError_t)(-code)));
        return buf;
    }
    snprintf(buf, sizeof buf, "murbhip: unknown code %d", code);
    return buf;
}

int murbhip_partition(unsigned long n, int world, int rank, unsigned long* first, unsigned long* count)
{
    if (world < 1 || rank < 0 || rank >= world || !first || !count) return MURBHIP_E_INVALID;
    partition(n, world, rank, first, count);
    return 0;
}

unsigned long murbhip_slice_slots(unsigned long n, int world) { return world < 1 ? 0 : slice_slots(n, world); }

unsigned long murbhip_slot_of_body(unsigned long n, int world, unsigned long i)
{
    if (world < 1 || i >= n) return ~0ul;
    const unsigned long base = n / (unsigned long)world, rem = n % (unsigned long)world;
    // ranks < rem own base+1 bodies, the rest own base
    unsigned long r, first;
    if (i < rem * (base + 1)) { r = i / (base + 1); first = r * (base + 1); }
    else { r = rem + (base ? (i - rem * (base + 1)) / base : 0); first = rem * (base + 1) + (r - rem) * base; }
    return r * slice_slots(n, world) + (i - first);
}

int murbhip_schedule_items(unsigned long n, int world, int rank, int split, int* pairs, unsigned long capacity,
                           unsigned long* count, unsigned long* own_count)
{
    if (world < 1 || world > MURB_SYM_MAX_RANKS || rank < 0 || rank >= world || !count || !own_count) return MURBHIP_E_INVALID;
    if (split != 1 && split != 2 && split != 4 && split != 8 && split != 16) return MURBHIP_E_INVALID;
    std::vector<int> flat;
    int own = 0;
    sym_schedule_items(world, rank, (int)(slice_slots(n, world) / MURB_SYM_BLOCK), split, sym_fill(n, world), flat, &own);
    *count = flat.size() / 2;
    *own_count = (unsigned long)own;
    if (pairs) {
        if (capacity < flat.size() / 2) return MURBHIP_E_INVALID;
        std::memcpy(pairs, flat.data(), flat.size() * sizeof(int));
    }
    return 0;
}

int murbhip_schedule_layout(unsigned long n, int world, int rank, int split, int waves, int taper_pct, int tri_first_pct,
                            int exchange_mode, long* items, unsigned long item_capacity, unsigned long* item_count, long* rows,
                            unsigned long row_capacity, unsigned long* row_count, unsigned long* floats_main,
                            unsigned long* floats_tri)
{
    if (world < 1 || world > MURB_SYM_MAX_RANKS || rank < 0 || rank >= world || !item_count || !row_count) return MURBHIP_E_INVALID;
    if (split != 1 && split != 2 && split != 4 && split != 8 && split != 16) return MURBHIP_E_INVALID;
    if ((waves != 4 && waves != 8) || taper_pct < 0 || (taper_pct & 0xff) > 100 || taper_pct > 0x7ff || tri_first_pct < 0 || tri_first_pct > 100)
        return MURBHIP_E_INVALID;
    if (MURB_SYM_BLOCK / split < 16 * waves) return MURBHIP_E_INVALID;
    SymHostLayout L;
    plan_sym_layout(world, rank, sym_fill(n, world), split, waves, taper_pct & 0xff, (taper_pct & 0x100) != 0,
                    exchange_mode != 0 || world > 1, 1, tri_first_pct, false, 0, L, 1 << ((taper_pct >> 9) & 3));
    *item_count = L.items.size();
    *row_count = L.table_main.size() + L.table_tri.size();
    if (floats_main) *floats_main = L.floats_main;
    if (floats_tri) *floats_tri = L.floats_tri;
    if (items) {
        if (item_capacity < L.items.size()) return MURBHIP_E_INVALID;
        const bool ex = exchange_mode != 0 || world > 1;
        for (size_t k = 0; k < L.items.size(); ++k) {
            const MurbSymItem& it = L.items[k];
            long* o = items + 8 * k;
            o[0] = it.i_slot0; o[1] = (long)it.ngroups * waves * MURB_SYM_R; o[2] = it.J; o[3] = it.flags;
            o[4] = (ex && (int)k < L.own) ? 1 : 0;
            o[5] = (long)it.ioff; o[6] = (long)it.joff;
            o[7] = !ex ? 0 : ((int)k < L.t1 ? 0 : ((int)k < L.own ? 1 : 2));
        }
    }
    if (rows) {
        if (row_capacity < *row_count) return MURBHIP_E_INVALID;
        size_t e = 0;
        for (int set = 0; set < 2; ++set)
            for (const MurbSymBlockRows& br : (set == 0 ? L.table_main : L.table_tri)) {
                long* o = rows + 7 * e++;
                o[0] = set; o[1] = br.out_slice; o[2] = br.out_block; o[3] = (long)br.base_i; o[4] = br.ni; o[5] = (long)br.base_j; o[6] = br.nj;
            }
    }
    return 0;
}

int murbhip_device_count(int* count)
{
    if (!count) return MURBHIP_E_INVALID;
    *count = 0;
    return hip_rc(hipGetDeviceCount(count));
}

int murbhip_create(murbhip_ctx** out, unsigned long n, float soft, float g, int device)
{
    const int rank0 = 0;
    return create_common(out, n, soft, g, 1, 1, &device, &rank0, 0, false);
}

int murbhip_create_sharded(murbhip_ctx** out, unsigned long n, float soft, float g, int ndev, const int* devices,
                           int exchange)
{
    if (ndev < 1 || ndev > 64 || !devices || (exchange != 0 && exchange != 1)) return MURBHIP_E_INVALID;
    if ((unsigned long)ndev > n) return MURBHIP_E_INVALID;
    std::vector<int> ranks(ndev);
    for (int i = 0; i < ndev; ++i) ranks[i] = i;
    if (exchange == 1) {
        if (!rccl().ok) return MURBHIP_E_NO_RCCL;
    }
    RC_TRY(create_common(out, n, soft, g, ndev, ndev, devices, ranks.data(), exchange, false));
    if (exchange == 1 && ndev > 1) {
        std::vector<rccl_comm_t> comms(ndev);
        const int rc = nccl_rc(rccl().CommInitAll(comms.data(), ndev, devices));
        if (rc != 0) { murbhip_destroy(*out); *out = nullptr; return rc; }
        for (int i = 0; i < ndev; ++i) (*out)->shards[i].comm_rccl = comms[i];
    }
    return 0;
}

int murbhip_unique_id(void* id_out)
{
    if (!id_out) return MURBHIP_E_INVALID;
    Rccl& r = rccl();
    if (!r.ok) return MURBHIP_E_NO_RCCL;
    rccl_id_t id;
    RC_TRY(nccl_rc(r.GetUniqueId(&id)));
    std::memcpy(id_out, &id, sizeof id);
    return 0;
}

int murbhip_create_rank(murbhip_ctx** out, unsigned long n, float soft, float g, int device, int rank, int world,
                        const void* unique_id)
{
    if (world < 1 || rank < 0 || rank >= world || (unsigned long)world > n) return MURBHIP_E_INVALID;
    if (world > MURB_SYM_MAX_RANKS) return MURBHIP_E_INVALID;   // fixed-size per-rank tables (MurbPeerPtrs)
    if (world > 1 && !unique_id) return MURBHIP_E_INVALID;
    if ((world > 1 || unique_id) && !rccl().ok) return MURBHIP_E_NO_RCCL;
    RC_TRY(create_common(out, n, soft, g, world, 1, &device, &rank, 1, true));
    if (world > 1 || unique_id) {   // a one-rank communicator is legal and exercises the whole RCCL binding
        rccl_id_t id;
        std::memcpy(&id, unique_id, sizeof id);
        int rc = hip_rc(hipSetDevice(device));
        if (rc == 0) rc = nccl_rc(rccl().CommInitRank(&(*out)->shards[0].comm_rccl, world, id, rank));
        if (rc != 0) { murbhip_destroy(*out); *out = nullptr; return rc; }
    }
    return 0;
}

int murbhip_destroy(murbhip_ctx* c)
{
    if (!c) return 0;
    delete c->crew;   // joins the shards' threads (idle: every entry point returns only when they have finished enqueueing)
    c->crew = nullptr;
    for (Shard& sh : c->shards) {
        (void)hipSetDevice(sh.device);
        drain(sh.compute); drain(sh.comm); drain(sh.compute_low);
        if (sh.comm_rccl && rccl().ok) rccl().CommDestroy(sh.comm_rccl);
        for (hipEvent_t& e : sh.prof) release_event(e);
        for (hipEvent_t* e : {&sh.ev_tri, &sh.ev_integrated, &sh.ev_gathered, &sh.ev_rowsum, &sh.ev_reduced}) release_event(*e);
        for (hipStream_t* q : {&sh.compute_low, &sh.compute, &sh.comm}) release_stream(*q);
        release(sh.rec[0], sh.rec[1], sh.vel, sh.accp, sh.acc_out, sh.phi_out, sh.mass, sh.radius, sh.metrics);
        if (sh.metrics_host) (void)hipHostFree(sh.metrics_host);
        release(sh.sym_items, sh.sym_send, sh.sym_recv, sh.sym_p2p, sh.sym_tri_acc, sh.sym_acc64);
        free_sym_set(sh.sym_main); free_sym_set(sh.sym_tri);
    }
    delete c;
    return 0;
}

int murbhip_upload(murbhip_ctx* c, const float* qx, const float* qy, const float* qz, const float* vx, const float* vy,
                   const float* vz, const float* m)
{
    if (!c || !qx || !qy || !qz || !vx || !vy || !vz || !m) return MURBHIP_E_INVALID;
    RC_TRY(murbhip_sync(c));
    // positions + GM for every slot (replicated), velocities for each local slice
    std::vector<float4> rec(c->slots, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int r = 0; r < c->world; ++r) {
        unsigned long first, count;
        partition(c->n, c->world, r, &first, &count);
        pack_records(c, qx, qy, qz, m, c->g, true, first, count, (unsigned long)r * c->slice, rec);
    }
    for (Shard& sh : c->shards) {
        HIP_TRY(hipSetDevice(sh.device));
        std::vector<float4> vel(c->slice, make_float4(0.f, 0.f, 0.f, 0.f));
        pack_records(c, vx, vy, vz, nullptr, 0.f, false, sh.first, sh.count, 0, vel);
        HIP_TRY(hipMemcpy(sh.rec[0], rec.data(), rec.size() * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(sh.rec[1], rec.data(), rec.size() * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(sh.vel, vel.data(), vel.size() * sizeof(float4), hipMemcpyHostToDevice));
        if (!sh.mass) {
            HIP_TRY(hipMalloc((void**)&sh.mass, c->slice * sizeof(float)));
            sh.bytes += c->slice * sizeof(float);
        }
        std::vector<float> mass(c->slice, 0.f);
        std::copy(m + sh.first, m + sh.first + sh.count, mass.begin());
        HIP_TRY(hipMemcpy(sh.mass, mass.data(), mass.size() * sizeof(float), hipMemcpyHostToDevice));
        sh.prof_used = 0;
    }
    c->cur = 0;
    c->gather_pending = false;
    c->uploaded = true;
    c->lf_half = false;
    c->acc_current = false;
    ++c->state_serial;
    return 0;
}

namespace {
// The 61 generator words around the first value rand() returns after srand(seed): glibc's srandom_r / random_r for the
// default TYPE_3 state (stdlib/random_r.c: 31 words from a 16807 Lehmer sequence, taps 3 apart, 310 outputs discarded).
MurbRandBase rand_base_words(unsigned int seed)
{
    int32_t st[MURB_RAND_DEG];
    int32_t word = seed == 0 ? 1 : (int32_t)seed;
    st[0] = word;
    for (int i = 1; i < MURB_RAND_DEG; ++i) {
        const long hi = word / 127773, lo = word % 127773;
        word = (int32_t)(16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        st[i] = word;
    }
    const int discard = 10 * MURB_RAND_DEG, first = discard - MURB_RAND_DEG, count = 2 * MURB_RAND_DEG - 1;
    MurbRandBase b{};
    int f = 3, r = 0;
    for (int t = 0; t < first + count; ++t) {
        const uint32_t val = (uint32_t)st[f] + (uint32_t)st[r];
        st[f] = (int32_t)val;
        if (t >= first) b.u[t - first] = val;
        f = (f + 1) % MURB_RAND_DEG;
        r = (r + 1) % MURB_RAND_DEG;
    }
    return b;
}
}  // namespace

int murbhip_init_bodies(murbhip_ctx* c, const char* scheme, unsigned long seed)
{
    if (!c || !scheme) return MURBHIP_E_INVALID;
    const std::string sc(scheme);
    const bool galaxy = sc == "galaxy";
    if (!galaxy && sc != "random") return MURBHIP_E_INVALID;
    RC_TRY(murbhip_sync(c));
    const unsigned long draws = galaxy ? 4ul * (c->n - 1) : 7ul * c->n;
    const MurbRandBase base = rand_base_words((unsigned int)seed);
    // glibc picks its FMA build of sincosf on CPUs with FMA and AVX2 (sysdeps/x86_64/fpu/multiarch/ifunc-fma.h)
    const bool fma = c->init_libm_fma >= 0 ? c->init_libm_fma != 0 : (__builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2"));
    for (Shard& sh : c->shards) {
        HIP_TRY(hipSetDevice(sh.device));
        if (!sh.mass) { HIP_TRY(hipMalloc((void**)&sh.mass, c->slice * sizeof(float))); sh.bytes += c->slice * sizeof(float); }
        if (!sh.radius) { HIP_TRY(hipMalloc((void**)&sh.radius, c->slice * sizeof(float))); sh.bytes += c->slice * sizeof(float); }
        unsigned int* d_draws = nullptr;
        HIP_TRY(hipMalloc((void**)&d_draws, std::max(draws, 1ul) * sizeof(unsigned int)));
        int rc = 0;
        const unsigned long chunks = (draws + MURB_RAND_CHUNK - 1) / MURB_RAND_CHUNK;
        if (chunks) hipLaunchKernelGGL(murb_rand_fill_kernel, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, sh.compute, base, draws, d_draws);
        rc = hip_rc(hipGetLastError());
        // padding slots: position 0, mass 0, like the records murbhip_upload packs
        if (!rc) rc = hip_rc(hipMemsetAsync(sh.rec[0], 0, c->slots * sizeof(float4), sh.compute));
        if (!rc) rc = hip_rc(hipMemsetAsync(sh.rec[1], 0, c->slots * sizeof(float4), sh.compute));
        if (!rc) rc = hip_rc(hipMemsetAsync(sh.vel, 0, c->slice * sizeof(float4), sh.compute));
        if (!rc) rc = hip_rc(hipMemsetAsync(sh.mass, 0, c->slice * sizeof(float), sh.compute));
        if (!rc) rc = hip_rc(hipMemsetAsync(sh.radius, 0, c->slice * sizeof(float), sh.compute));
        if (!rc) {
            MurbInitArgs a{};
            a.draws = d_draws; a.rec0 = sh.rec[0]; a.rec1 = sh.rec[1]; a.vel = sh.vel; a.mass = sh.mass; a.radius = sh.radius;
            a.n = c->n; a.world = (unsigned int)c->world; a.rank = (unsigned int)sh.rank; a.slice = (unsigned int)c->slice; a.g = c->g;
            const dim3 grid((unsigned)((c->n + 255) / 256));
            if (!galaxy) hipLaunchKernelGGL(murb_init_random_kernel, grid, dim3(256), 0, sh.compute, a);
            else if (fma) hipLaunchKernelGGL((murb_init_galaxy_kernel<true>), grid, dim3(256), 0, sh.compute, a);
            else hipLaunchKernelGGL((murb_init_galaxy_kernel<false>), grid, dim3(256), 0, sh.compute, a);
            rc = hip_rc(hipGetLastError());
        }
        const int rs = hip_rc(hipStreamSynchronize(sh.compute));
        release(d_draws);
        if (rc || rs) return rc ? rc : rs;
        sh.prof_used = 0;
    }
    c->cur = 0;
    c->gather_pending = false;
    c->uploaded = true;
    c->lf_half = false;
    c->acc_current = false;
    ++c->state_serial;
    return 0;
}

int murbhip_download_mass(murbhip_ctx* c, float* m, float* r)
{
    if (!c || !m) return MURBHIP_E_INVALID;
    if (!c->uploaded) return MURBHIP_E_STATE;
    RC_TRY(murbhip_sync(c));
    std::vector<float> buf(c->slice);
    for (Shard& sh : c->shards) {
        HIP_TRY(hipSetDevice(sh.device));
        HIP_TRY(hipMemcpy(buf.data(), sh.mass, buf.size() * sizeof(float), hipMemcpyDeviceToHost));
        std::memcpy(m + sh.first, buf.data(), sh.count * sizeof(float));
        if (r) {
            if (!sh.radius) return MURBHIP_E_STATE;   // radii exist on the device only after murbhip_init_bodies
            HIP_TRY(hipMemcpy(buf.data(), sh.radius, buf.size() * sizeof(float), hipMemcpyDeviceToHost));
            std::memcpy(r + sh.first, buf.data(), sh.count * sizeof(float));
        }
    }
    return 0;
}

namespace {
// v + a*h with the product rounded on its own, like the device kicks (murb_kernels.h)
float half_kick(float v, float a, float h)
{
#pragma clang fp contract(off)
    const float k = a * h;
    return v + k;
}
}  // namespace

int murbhip_sync(murbhip_ctx* c)
{
    if (!c) return MURBHIP_E_INVALID;
    int rc = 0;
    for (Shard& sh : c->shards) {
        int r1 = hip_rc(hipSetDevice(sh.device));
        if (!r1) r1 = hip_rc(hipStreamSynchronize(sh.compute));
        if (!r1) r1 = hip_rc(hipStreamSynchronize(sh.compute_low));
        if (!r1) r1 = hip_rc(hipStreamSynchronize(sh.comm));
        if (r1 && !rc) rc = r1;
    }
    if (rc && !c->async_error) c->async_error = rc;
    return c->async_error ? c->async_error : rc;
}

int murbhip_download_state(murbhip_ctx* c, float* qx, float* qy, float* qz, float* vx, float* vy, float* vz)
{
    if (!c) return MURBHIP_E_INVALID;
    if (!c->uploaded) return MURBHIP_E_STATE;
    // leapfrog: the device holds v_{n-1/2}; what the caller gets is v_n = v_{n-1/2} + a(q_n)*dt/2, which
    // costs one force evaluation (in one-process-per-GPU mode that makes this call a collective)
    const bool closing_kick = c->lf_half && (vx || vy || vz);
    if (closing_kick) RC_TRY(enqueue_iteration(c, 0.f, 0));
    RC_TRY(murbhip_sync(c));
    std::vector<float4> rec(c->slots), vel(c->slice);
    std::vector<float> acc(closing_kick ? 3 * c->slice : 0);
    // positions: any shard holds all of them once its exchange has landed (sync above)
    {
        Shard& sh = c->shards[0];
        HIP_TRY(hipSetDevice(sh.device));
        HIP_TRY(hipMemcpy(rec.data(), sh.rec[c->cur], rec.size() * sizeof(float4), hipMemcpyDeviceToHost));
        for (int r = 0; r < c->world; ++r) {
            unsigned long first, count;
            partition(c->n, c->world, r, &first, &count);
            for (unsigned long k = 0; k < count; ++k) {
                const unsigned long slot = (unsigned long)r * c->slice + k;
                const unsigned long ra = murb_rec_a(slot >> 1);
                const float* A = reinterpret_cast<const float*>(&rec[ra]);
                const float* B = reinterpret_cast<const float*>(&rec[ra + MURB_TILE_PAIRS]);
                const int h = (int)(slot & 1);
                if (qx) qx[first + k] = A[h];
                if (qy) qy[first + k] = A[2 + h];
                if (qz) qz[first + k] = B[h];
            }
        }
    }
    if (vx || vy || vz) {
        for (Shard& sh : c->shards) {
            HIP_TRY(hipSetDevice(sh.device));
            HIP_TRY(hipMemcpy(vel.data(), sh.vel, vel.size() * sizeof(float4), hipMemcpyDeviceToHost));
            if (closing_kick) HIP_TRY(hipMemcpy(acc.data(), sh.acc_out, acc.size() * sizeof(float), hipMemcpyDeviceToHost));
            const float half = 0.5f * c->lf_last_dt;
            for (unsigned long k = 0; k < sh.count; ++k) {
                const unsigned long ra = murb_rec_a(k >> 1);
                const float* A = reinterpret_cast<const float*>(&vel[ra]);
                const float* B = reinterpret_cast<const float*>(&vel[ra + MURB_TILE_PAIRS]);
                const int h = (int)(k & 1);
                float ox = A[h], oy = A[2 + h], oz = B[h];
                if (closing_kick) {
                    ox = half_kick(ox, acc[k], half);
                    oy = half_kick(oy, acc[c->slice + k], half);
                    oz = half_kick(oz, acc[2 * c->slice + k], half);
                }
                if (vx) vx[sh.first + k] = ox;
                if (vy) vy[sh.first + k] = oy;
                if (vz) vz[sh.first + k] = oz;
            }
        }
    }
    return 0;
}

int murbhip_download_acc(murbhip_ctx* c, float* ax, float* ay, float* az)
{
    if (!c || !ax || !ay || !az) return MURBHIP_E_INVALID;
    if (!c->uploaded) return MURBHIP_E_STATE;
    RC_TRY(murbhip_sync(c));
    std::vector<float> a(3 * c->slice);
    for (Shard& sh : c->shards) {
        HIP_TRY(hipSetDevice(sh.device));
        HIP_TRY(hipMemcpy(a.data(), sh.acc_out, a.size() * sizeof(float), hipMemcpyDeviceToHost));
        std::memcpy(ax + sh.first, a.data(), sh.count * sizeof(float));
        std::memcpy(ay + sh.first, a.data() + c->slice, sh.count * sizeof(float));
        std::memcpy(az + sh.first, a.data() + 2 * c->slice, sh.count * sizeof(float));
    }
    return 0;
}

int murbhip_compute_acc(murbhip_ctx* c)
{
    if (!c) return MURBHIP_E_INVALID;
    if (!c->uploaded) return MURBHIP_E_STATE;
    return enqueue_iteration(c, 0.f, 0);
}

int murbhip_warmup(murbhip_ctx* c, double milliseconds)
{
    if (!c || !(milliseconds >= 0.0) || milliseconds > 10000.0) return MURBHIP_E_INVALID;
    if (!c->uploaded) return MURBHIP_E_STATE;
    // a COUNT fixed by n and the number of ranks, not a clock: in rank mode every evaluation carries collectives, and
    // all ranks must enqueue the same number of them
    const double per_eval_s = (double)c->n * (double)c->n / (5e12 * (double)c->world) + 10e-6;
    const int evaluations = (int)std::min(5000.0, std::ceil(milliseconds * 1e-3 / per_eval_s));
    for (int k = 0; k < evaluations; ++k) {
        c->acc_current = false;   // evaluate again, whatever is remembered
        RC_TRY(enqueue_iteration(c, 0.f, 0));
    }
    c->acc_current = false;       // and nothing of it is kept: the first timed step does all of its own work
    c->pe_current = false;
    return murbhip_sync(c);
}

int murbhip_step(murbhip_ctx* c, float dt)
{
    if (!c) return MURBHIP_E_INVALID;
    if (!c->uploaded) return MURBHIP_E_STATE;
    RC_TRY(enqueue_iteration(c, dt, 1));
    if (c->integrator == 1) { c->lf_half = true; c->lf_last_dt = dt; }
    return 0;
}

int murbhip_steps(murbhip_ctx* c, float dt, int iterations)
{
    if (!c || iterations < 0) return MURBHIP_E_INVALID;
    if (!c->uploaded) return MURBHIP_E_STATE;
    for (int i = 0; i < iterations; ++i) {
        RC_TRY(enqueue_iteration(c, dt, 1));
        if (c->integrator == 1) { c->lf_half = true; c->lf_last_dt = dt; }
    }
    return 0;
}

int murbhip_integrate_host_acc(murbhip_ctx* c, const float* ax, const float* ay, const float* az, float dt)
{
    if (!c || !ax || !ay || !az) return MURBHIP_E_INVALID;
    if (!c->uploaded || c->lf_half) return MURBHIP_E_STATE;   // a leapfrog run in flight has half-step velocities
    std::vector<float4> part(c->slice);
    for (Shard& sh : c->shards) {
        HIP_TRY(hipSetDevice(sh.device));
        if (c->gather_pending) HIP_TRY(hipStreamWaitEvent(sh.compute, sh.ev_gathered, 0));
        std::fill(part.begin(), part.end(), make_float4(0.f, 0.f, 0.f, 0.f));
        for (unsigned long k = 0; k < sh.count; ++k)
            part[k] = make_float4(ax[sh.first + k], ay[sh.first + k], az[sh.first + k], 0.f);
        RC_TRY(ensure_accp(c, sh));
        HIP_TRY(hipMemcpyAsync(sh.accp, part.data(), part.size() * sizeof(float4), hipMemcpyHostToDevice, sh.compute));
        HIP_TRY(hipStreamSynchronize(sh.compute));   // `part` is reused for the next shard
        RC_TRY(enqueue_integrate(c, sh, 1, dt, 1, nullptr, 0));
    }
    if (c->world > 1) {
        RC_TRY(crew_run(c, [&](Shard& sh) { return shard_exchange(c, sh, c->cur ^ 1, 0); }));
        c->gather_pending = true;
    }
    c->cur ^= 1;
    c->acc_current = false;
    ++c->state_serial;
    return 0;
}

namespace {
// Leapfrog: the device holds v_{n-1/2}; anything that reports v_n needs a(q_n) (one force evaluation, remembered;
// a collective in one-process-per-GPU mode).
int ensure_acc_for_readout(murbhip_ctx* c)
{
    if (c->lf_half && !c->acc_current) RC_TRY(enqueue_iteration(c, 0.f, 0));
    return 0;
}

// The O(N) sums of the tracked metrics over this process's bodies (murb_metrics_kernel + the block rows added
// in index order on the host).  want_phi: the potential sweep has just been written to phi_out.  pair_sum: also add up
// the pair potentials murbhip_energy has just enqueued into the buffers' tails (then the kept sums are not enough).
int device_metrics(murbhip_ctx* c, bool want_phi, double (&sums)[MURB_METRIC_VALUES], double* pair_sum = nullptr)
{
    if (!pair_sum && c->metrics_serial == c->state_serial && (c->metrics_with_phi || !want_phi)) {   // same state, sums already here
        RC_TRY(murbhip_sync(c));
        for (int k = 0; k < MURB_METRIC_VALUES; ++k) sums[k] = c->metrics_sums[k];
        return 0;
    }
    const MetricsLayout l = metrics_layout(c);
    for (Shard& sh : c->shards) {
        HIP_TRY(hipSetDevice(sh.device));
        if (!sh.metrics) {
            HIP_TRY(hipMalloc((void**)&sh.metrics, l.total * sizeof(double)));
            sh.bytes += l.total * sizeof(double);
        }
        if (c->gather_pending) HIP_TRY(hipStreamWaitEvent(sh.compute, sh.ev_gathered, 0));
        MurbMetricsArgs a{};
        a.rec = sh.rec[c->cur];
        a.vel = sh.vel;
        a.mass = sh.mass;
        a.phi = want_phi ? sh.phi_out : nullptr;
        a.acc = sh.acc_out;
        a.out = sh.metrics;
        a.i_first_slot = (int)((unsigned long)sh.rank * c->slice);
        a.count = (int)sh.count;
        a.acc_stride = (unsigned int)c->slice;
        a.half_dt = c->lf_half ? 0.5f * c->lf_last_dt : 0.f;
        a.g_over_soft = (double)c->g / std::sqrt((double)c->soft2);
        hipLaunchKernelGGL(murb_metrics_kernel, dim3((unsigned)l.blocks), dim3(256), 0, sh.compute, a);
        RC_TRY(hip_rc(hipGetLastError()));
        // the read-out rides behind the kernels on the same stream, into pinned memory: one wait for everything
        if (!sh.metrics_host) HIP_TRY(hipHostMalloc((void**)&sh.metrics_host, l.total * sizeof(double), hipHostMallocDefault));
        HIP_TRY(hipMemcpyAsync(sh.metrics_host, sh.metrics, (pair_sum ? l.total : l.pe_main) * sizeof(double), hipMemcpyDeviceToHost, sh.compute));
    }
    RC_TRY(murbhip_sync(c));
    for (double& v : sums) v = 0.0;
    if (pair_sum) *pair_sum = 0.0;
    for (Shard& sh : c->shards) {
        const double* const host = sh.metrics_host;
        for (size_t b = 0; b < l.blocks; ++b)
            for (int k = 0; k < MURB_METRIC_VALUES; ++k) sums[k] += host[b * MURB_METRIC_VALUES + k];
        if (!pair_sum) continue;
        if (sh.sym_main.part) for (int k = 0; k < kPeSumBlocks; ++k) *pair_sum += host[l.pe_main + k];
        if (sh.sym_tri.part) for (int k = 0; k < kPeSumBlocks; ++k) *pair_sum += host[l.pe_tri + k];
        for (size_t b = 0; b < l.own_blocks * MURB_PE_DIAG_SPLIT; ++b) *pair_sum += host[l.pe_diag + b];
    }
    for (int k = 0; k < MURB_METRIC_VALUES; ++k) c->metrics_sums[k] = sums[k];
    c->metrics_serial = c->state_serial;
    c->metrics_with_phi = want_phi;
    return 0;
}
}  // namespace

int murbhip_energy(murbhip_ctx* c, double* kinetic, double* potential)
{
    if (!c || !kinetic || !potential) return MURBHIP_E_INVALID;
    if (!c->uploaded) return MURBHIP_E_STATE;
    const Plan main_plan = make_plan(c);
    if (main_plan.symmetric && !c->energy_sweep) {
        // Pair-symmetric plan: the potential energy comes out of a FORCE evaluation (murb_kernels_sym.h, PHI = 2: two more
        // packed instructions per 18 sum G m_i G m_j / r of every pair a wave meets, one float per group of 4 i bodies behind
        // the partial rows) — no second N^2 sweep.  The forces of that evaluation are remembered, so a step that follows only
        // launches the state update (also with several shards): a tracked iteration costs one force evaluation.  Every shard
        // sums the pairs IT evaluated off the diagonal, plus the pairs inside its own blocks (murb_sym_pe_diag_kernel: fp64,
        // no self terms).  Several passes over one shared buffer (one GPU, N > 2.4 M): the groups' sums are added up pass by
        // pass while the evaluation runs (enqueue_sym_passes).
        const MetricsLayout l = metrics_layout(c);
        for (Shard& sh : c->shards) {
            if (sh.metrics) continue;
            HIP_TRY(hipSetDevice(sh.device));
            HIP_TRY(hipMalloc((void**)&sh.metrics, l.total * sizeof(double)));
            sh.bytes += l.total * sizeof(double);
        }
        if (!(c->acc_current && c->pe_current)) {
            c->acc_current = false;   // forces alone are not enough: evaluate again, this time with the pair potential
            c->want_pe = true;
            const int rc = enqueue_iteration(c, 0.f, 0);
            c->want_pe = false;
            RC_TRY(rc);
        }
        for (Shard& sh : c->shards) {
            HIP_TRY(hipSetDevice(sh.device));
            const size_t at[2] = {l.pe_main, l.pe_tri};
            int k = 0;
            for (SymSet* st : {&sh.sym_main, &sh.sym_tri}) {
                const size_t off = at[k++];
                if (!st->part || st->passes.size() > 1) continue;   // several passes: summed during the evaluation
                hipLaunchKernelGGL(murb_sym_pe_sum_kernel, dim3(kPeSumBlocks), dim3(1024), 0, sh.compute, st->part + 3 * st->comp_stride,
                                   (unsigned long)(st->comp_stride / MURB_SYM_R + 1), sh.metrics + off, 0);
                RC_TRY(hip_rc(hipGetLastError()));
            }
            // the diagonal blocks (the own slice's) separately, in fp64 and without the bodies' own terms
            hipLaunchKernelGGL(murb_sym_pe_diag_kernel, dim3((unsigned)(l.own_blocks * MURB_PE_DIAG_SPLIT)), dim3(256), 0, sh.compute, sh.rec[c->cur],
                               (int)(sh.rank * l.own_blocks), c->soft2, sh.metrics + l.pe_diag);
            RC_TRY(hip_rc(hipGetLastError()));
        }
        double sums[MURB_METRIC_VALUES], pair_sum = 0.0;
        RC_TRY(device_metrics(c, false, sums, &pair_sum));   // syncs; one copy per shard
        *kinetic = sums[0];
        *potential = -pair_sum / (double)c->g;
        return 0;
    }
    // phi_i = sum_j GM_j / sqrt(r_ij^2 + soft^2) over ALL j (self term included), written to the x plane of phi_out:
    // pair-symmetric sweep where the force plan is pair-symmetric (one GPU: one launch; several ranks: the half-ring
    // schedule with its reduce-scatter), the one-sided sweep otherwise
    RC_TRY(ensure_acc_for_readout(c));   // before the sweep: it reuses the one-sided partial rows
    Plan p{};
    p.variant = kPotentialKernel;
    {
        const unsigned long tiles_local = c->slice / MURB_TILE_BODIES, tiles_remote = (c->slots - c->slice) / MURB_TILE_BODIES;
        p.parts_local = std::max(1, auto_parts(c, 1, c->slice, tiles_local));
        p.parts_remote = c->world > 1 ? std::max(1, std::min<int>(auto_parts(c, 1, c->slice, tiles_remote), kMaxParts / 2)) : 0;
    }
    // one shard on the pair-symmetric plan: the sweep is pair-symmetric too (phi_i += G m_j / r, phi_j += G m_i / r:
    // 8 packed + 2 rsq per 4 pair terms instead of 7 + 2 per 2), through the force kernel's partial rows
    const bool symmetric_sweep = main_plan.symmetric && c->world == 1 && !c->force_exchange;
    const bool symmetric_multi = main_plan.symmetric && !symmetric_sweep;   // several ranks: the half-ring form
    for (Shard& sh : c->shards) {
        HIP_TRY(hipSetDevice(sh.device));
        if (!sh.phi_out) {
            HIP_TRY(hipMalloc((void**)&sh.phi_out, 3 * c->slice * sizeof(float)));
            sh.bytes += 3 * c->slice * sizeof(float);
        }
    }
    if (main_plan.symmetric) {   // a rebuild of existing tables needs everything drained (see sym_schedule_stale)
        bool stale = false;
        for (const Shard& sh : c->shards) stale = stale || (sh.sym_items && sym_schedule_stale(c, sh, main_plan));
        if (stale) RC_TRY(murbhip_sync(c));
    }
    if (symmetric_multi) {
        RC_TRY(crew_run(c, [&](Shard& sh) { return shard_potential_sym_multi(c, sh, main_plan); }));
        c->reduce_pending = true;
    }
    for (Shard& sh : c->shards) {
        if (symmetric_multi) break;
        HIP_TRY(hipSetDevice(sh.device));
        if (c->gather_pending) HIP_TRY(hipStreamWaitEvent(sh.compute, sh.ev_gathered, 0));
        if (symmetric_sweep) {
            RC_TRY(build_sym_schedule(c, sh, main_plan));
            if (sh.sym_main.passes.size() > 1) {
                RC_TRY(enqueue_sym_passes(c, sh, true));
                MurbIntegrateArgs a{};   // no state update: phi_out = the accumulated sums
                a.rec_in = sh.rec[c->cur]; a.rec_out = sh.rec[c->cur ^ 1]; a.vel = sh.vel;
                a.acc_out = sh.phi_out;
                a.acc64 = sh.sym_acc64; a.acc64_stride = (unsigned int)c->slots;
                a.count = (int)sh.count; a.acc_stride = (unsigned int)c->slice;
                hipLaunchKernelGGL(murb_integrate_kernel, dim3((unsigned)((c->slice / 2 + 255) / 256)), dim3(256), 0, sh.compute, a);
                RC_TRY(hip_rc(hipGetLastError()));
                continue;
            }
            RC_TRY(enqueue_sym_launch(c, sh, 0, sh.sym_items_total, false, nullptr, true));
            RC_TRY(enqueue_sym_rowsum(sh.sym_main, sh.phi_out, (unsigned int)c->slots, sh.compute));
            continue;
        }
        RC_TRY(enqueue_force(c, sh, p, 0));
        if (c->world > 1) RC_TRY(enqueue_force(c, sh, p, 1));
        RC_TRY(enqueue_integrate(c, sh, p.parts_local + p.parts_remote, 0.f, 0, nullptr, -1, sh.phi_out));
    }
    double sums[MURB_METRIC_VALUES];
    RC_TRY(device_metrics(c, true, sums));
    *kinetic = sums[0];
    *potential = sums[1];
    return 0;
}

int murbhip_moments(murbhip_ctx* c, double* out10)
{
    if (!c || !out10) return MURBHIP_E_INVALID;
    if (!c->uploaded) return MURBHIP_E_STATE;
    RC_TRY(ensure_acc_for_readout(c));
    double sums[MURB_METRIC_VALUES];
    RC_TRY(device_metrics(c, false, sums));
    std::memcpy(out10, sums + 2, 10 * sizeof(double));
    return 0;
}

int murbhip_set_option(murbhip_ctx* c, const char* key, long value)
{
    if (!c || !key) return MURBHIP_E_INVALID;
    const std::string k(key);
    if (k == "solo_shard" || k == "force_exchange") c->acc_current = false;   // what acc_out covers changes
    if (k == "variant") { if (value < 0 || value > kNumVariants) return MURBHIP_E_INVALID; c->variant = (int)value; }
    else if (k == "jsplit") { if (value < 0 || value > kMaxParts / 2) return MURBHIP_E_INVALID; c->jsplit = (int)value; }
    else if (k == "xcd_order") c->xcd_order = value ? 1 : 0;
    else if (k == "pad_aware") c->pad_aware = value ? 1 : 0;
    else if (k == "energy_sweep") c->energy_sweep = value ? 1 : 0;
    else if (k == "fuse_integrate") c->fuse_integrate = value ? 1 : 0;
    else if (k == "exchange_p2p") {
        if (value && (c->exchange != 1 || !rccl().Send || !rccl().Recv)) return MURBHIP_E_STATE;   // needs the RCCL exchange and ncclSend/ncclRecv
        RC_TRY(murbhip_sync(c));
        c->exchange_p2p = value ? 1 : 0;
    }
    else if (k == "tri_div") { if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return MURBHIP_E_INVALID; c->tri_div = (int)value; }
    else if (k == "init_libm_fma") { if (value < -1 || value > 1) return MURBHIP_E_INVALID; c->init_libm_fma = (int)value; }
    else if (k == "tri_first_pct") { if (value < 0 || value > 100) return MURBHIP_E_INVALID; c->tri_first_pct = (int)value; }
    else if (k == "taper") { if (value < -1 || value > 100) return MURBHIP_E_INVALID; c->taper = (int)value; }
    else if (k == "sym_pass_mb") { if (value < 0) return MURBHIP_E_INVALID; c->sym_pass_mb = value; }
    else if (k == "diag_tri") { if (value < -1 || value > 1) return MURBHIP_E_INVALID; c->diag_tri = (int)value; }
    else if (k == "sym_red") { if (value < -1 || value > 1) return MURBHIP_E_INVALID; c->sym_red = (int)value; }
    else if (k == "sym_waves") { if (value != 0 && value != 4 && value != 8) return MURBHIP_E_INVALID; c->sym_waves = (int)value; }
    else if (k == "overlap") { if (value < 0 || value > 2) return MURBHIP_E_INVALID; c->overlap = (int)value; }
    else if (k == "integrator") {
        if (value < 0 || value > 1) return MURBHIP_E_INVALID;
        if (c->lf_half && value != c->integrator) return MURBHIP_E_STATE;   // half-step velocities on the device: upload first
        c->integrator = (int)value;
    }
    else if (k == "solo_shard") c->solo_shard = (int)value;
    else if (k == "cu_reserve") {
        if (value < 0 || value > c->cu_count / 2) return MURBHIP_E_INVALID;
        if ((int)value != c->cu_reserve) {
            RC_TRY(murbhip_sync(c));
            for (Shard& sh : c->shards) {
                HIP_TRY(hipSetDevice(sh.device));
                HIP_TRY(hipStreamDestroy(sh.compute)); sh.compute = nullptr;
                HIP_TRY(hipStreamDestroy(sh.compute_low)); sh.compute_low = nullptr;
                RC_TRY(create_compute_streams(c, sh, (int)value));
            }
            c->cu_reserve = (int)value;
        }
    }
    else if (k == "force_exchange") {
        if (value && c->exchange == 1 && !c->shards[0].comm_rccl) return MURBHIP_E_STATE;
        c->force_exchange = value ? 1 : 0;
    }
    else if (k == "profile") {
        if (value < 0 || value > 2) return MURBHIP_E_INVALID;
        RC_TRY(murbhip_sync(c));   // spans of the previous setting may still be in flight
        c->profile = (int)value;
        for (Shard& sh : c->shards) {
            HIP_TRY(hipSetDevice(sh.device));
            if (c->profile && sh.prof.empty()) {
                sh.prof.resize(2 * kProfPairs);
                sh.prof_kind.assign(kProfPairs, 0);
                for (hipEvent_t& e : sh.prof) HIP_TRY(hipEventCreate(&e));
            }
            sh.prof_used = 0;
            sh.sym_launches = 0;
        }
    } else return MURBHIP_E_INVALID;
    return 0;
}

int murbhip_get_info(murbhip_ctx* c, const char* key, double* value)
{
    if (!c || !key || !value) return MURBHIP_E_INVALID;
    const std::string k(key);
    const Plan p = make_plan(c);
    if (k == "cu_count") *value = c->cu_count;
    else if (k == "clock_mhz") *value = c->clock_mhz;
    else if (k == "n") *value = (double)c->n;
    else if (k == "slots") *value = (double)c->slots;
    else if (k == "world") *value = c->world;
    else if (k == "cu_reserve") *value = c->cu_reserve;
    else if (k == "sym_passes") *value = c->shards[0].sym_main.passes.empty() ? 0.0 : (double)c->shards[0].sym_main.passes.size();
    else if (k == "rank") *value = c->shards[0].rank;
    else if (k == "jsplit") *value = p.symmetric ? (double)p.split
                                     : p.persistent ? (double)p.sched[0].nblocks / std::max(resident_blocks(c), 1)
                                                    : (double)(p.parts_local + p.parts_remote);
    else if (k == "sym_waves") *value = p.symmetric ? p.waves : 0;
    else if (k == "taper") *value = p.symmetric ? p.taper : 0;
    else if (k == "workgroups") *value = p.persistent ? p.sched[0].nblocks + (c->world > 1 ? p.sched[1].nblocks : 0) : 0;
    else if (k == "variant") *value = p.variant;
    else if (k == "interactions_per_launch") *value = c->interactions_per_launch;
    else if (k == "device_bytes") { double b = 0; for (Shard& sh : c->shards) b += (double)sh.bytes; *value = b; }
    else if (k == "sym_launches") { double v = 0; for (Shard& sh : c->shards) v += (double)sh.sym_launches; *value = v; }
    else if (k == "spans_dropped") {   // 1: the event pool ran out during the profiled steps (averages cover the first part only)
        *value = 0;
        for (Shard& sh : c->shards) if (!sh.prof.empty() && sh.prof_used + 2 > sh.prof.size()) *value = 1;
    }
    else if (k == "force_launches" || k == "force_ms_avg" || k == "force_ms_total" || k.rfind("span_", 0) == 0 || k == "compute_wait_ms_per_step") {
        // Timing spans of the profiled steps ("profile"), over all shards of this process:
        //   force_*                       every force launch (one GPU: the launch; exchange pipeline: T1, R and T2 together)
        //   span_<kind>_ms_avg / _count   kind in tri1, rect, tri2, reduce_scatter, all_gather, wait_gather, wait_reduce, step
        //   compute_wait_ms_per_step      (wait_gather + wait_reduce) per profiled step: what the exchange costs the compute stream
        static const char* const names[kProfKinds] = {"force", "tri1", "rect", "tri2", "reduce_scatter", "all_gather", "wait_gather",
                                                      "wait_reduce", "step"};
        RC_TRY(murbhip_sync(c));
        double total[kProfKinds] = {0};
        size_t count[kProfKinds] = {0};
        for (Shard& sh : c->shards) {
            HIP_TRY(hipSetDevice(sh.device));
            for (size_t i = 0; i + 1 < sh.prof_used; i += 2) {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, sh.prof[i], sh.prof[i + 1]));
                const int kind = sh.prof_kind[i / 2];
                total[kind] += ms; ++count[kind];
            }
        }
        const double f_total = total[kProfForce] + total[kProfTri1] + total[kProfRect] + total[kProfTri2];
        const size_t f_count = count[kProfForce] + count[kProfTri1] + count[kProfRect] + count[kProfTri2];
        if (k == "force_launches") *value = (double)f_count;
        else if (k == "force_ms_total") *value = f_total;
        else if (k == "force_ms_avg") *value = f_count ? f_total / (double)f_count : 0.0;
        else if (k == "compute_wait_ms_per_step")
            *value = count[kProfStep] ? (total[kProfWaitGather] + total[kProfWaitReduce]) / (double)count[kProfStep] : 0.0;
        else {
            const bool avg = k.size() > 7 && k.compare(k.size() - 7, 7, "_ms_avg") == 0;
            const bool cnt = k.size() > 6 && k.compare(k.size() - 6, 6, "_count") == 0;
            if (!avg && !cnt) return MURBHIP_E_INVALID;
            const std::string name = k.substr(5, k.size() - 5 - (avg ? 7 : 6));
            int kind = -1;
            for (int q = 0; q < kProfKinds; ++q) if (name == names[q]) kind = q;
            if (kind < 0) return MURBHIP_E_INVALID;
            *value = avg ? (count[kind] ? total[kind] / (double)count[kind] : 0.0) : (double)count[kind];
        }
    } else return MURBHIP_E_INVALID;
    return 0;
}

}  // extern "C"
