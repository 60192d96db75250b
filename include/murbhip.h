/*
 * murbhip.h — C ABI of the MI355X-native all-pairs force + integrate path for MUrB.
 *
 * This is the drop-in boundary: everything the reference's `--im` implementations do on the
 * device side of SimulationNBodyInterface<T>::computeOneIteration()
 * (reference src/common/core/SimulationNBodyInterface.hpp:45) is reachable through these
 * entry points with plain pointers and sizes.  The only translation unit behind it that needs
 * hipcc is nbody-eurohpc_amd/csrc/murbhip.hip; host code (C++, or ctypes/cgo/JNI) links
 * libmurbhip.so and never sees a HIP header.
 *
 * Conventions
 *   - every function returns 0 on success, a negative value on failure:
 *       -1 … -1999    : -(hipError_t)
 *       -2000 … -2999 : MURBHIP_E_* argument / state errors
 *       -3000 … -3999 : -(3000 + ncclResult_t)
 *     murbhip_error_string() turns any of them into text.  The C++ wrapper maps non-zero to the
 *     reference's print-to-stderr + exit(code) convention
 *     (reference src/murb/implem/SimulationNBodyCUDATileFullDevice.cu:10-17).
 *   - a context is driven by ONE host thread (reference driver contract, src/murb/main.cpp:348-354).
 *   - step functions only enqueue work; murbhip_sync() is the per-iteration device sync the
 *     reference driver performs itself (src/murb/main.cpp:356-368).
 *   - arrays are fp32 SoA of n entries in the reference's body order (dataSoA_t,
 *     src/common/core/Bodies.hpp:15-24).  SIMD padding bodies (Bodies.cpp:201-213) are NOT passed
 *     in: they carry no mass and no implementation of the reference reads them in the j loop.
 */
#ifndef MURBHIP_H_
#define MURBHIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct murbhip_ctx murbhip_ctx;

#define MURBHIP_UNIQUE_ID_BYTES 128

#define MURBHIP_E_INVALID (-2000)   /* bad argument                                   */
#define MURBHIP_E_STATE (-2001)     /* call made in the wrong state (e.g. no upload)  */
#define MURBHIP_E_NO_DEVICE (-2002) /* no usable HIP device                           */
#define MURBHIP_E_NO_RCCL (-2003)   /* librccl could not be loaded                    */
#define MURBHIP_E_NOMEM (-2004)     /* host allocation failed                         */

/* ------------------------------------------------------------------ host-only helpers (no GPU needed) */

/* ABI version of this header (major*100 + minor). */
int murbhip_version(void);

/* Text for any return code of this library (static storage, never NULL). */
const char* murbhip_error_string(int code);

/* Block partition of n bodies over `world` ranks: first index and count of rank `rank`.
 * Same rule as the reference's MPI path: counts[r] = n/world + (r < n%world), displs = prefix sums
 * (reference src/murb/implem/SimulationNBodyMultiNode.cpp:76-91). */
int murbhip_partition(unsigned long n, int world, int rank, unsigned long* first, unsigned long* count);

/* Device slots each rank owns in the replicated position buffer: the largest slice rounded up to
 * the slice unit (a multiple of 1024 body slots).  Slots past a rank's count hold mass 0 and contribute
 * exactly 0 to every sum, so one equal-count all-gather replaces the reference's MPI_Allgatherv
 * (SimulationNBodyMultiNode.cpp:104-114). */
unsigned long murbhip_slice_slots(unsigned long n, int world);

/* Slot of body i in the replicated buffer (rank(i) * slice_slots + offset inside its slice). */
unsigned long murbhip_slot_of_body(unsigned long n, int world, unsigned long i);

/* The work list of rank `rank` under the multi-GPU pair-symmetric ("half ring") schedule: `*count`
 * items (i-side sub-block, j-side block), the first `*own_count` of which lie inside the rank's own slice.
 * A block is 1024 slots, a sub-block 1024/split.  Over all ranks every unordered pair of bodies is
 * covered exactly once (own-slice items cover both orders inside their diagonal blocks).  `pairs`
 * (2 ints per item, may be NULL to query the count) must hold `capacity` items.  Replaces the "who
 * computes what" of the reference's MPI path, which has every rank sweep all j for its i range
 * (SimulationNBodyMultiNode.cpp:151-170). */
int murbhip_schedule_items(unsigned long n, int world, int rank, int split, int* pairs, unsigned long capacity,
                           unsigned long* count, unsigned long* own_count);

/* Host only: the same work list as the pair-symmetric kernel consumes it, with the layout of its partial sums —
 * what murbhip_step builds for (n, world, rank) under the given plan (`split` i-side sub-blocks per block, `waves` 4 or 8
 * per workgroup, `taper_pct` % of each launch cut into finer items (+ 256: diagonal blocks as triangular pieces,
 * option "diag_tri"; + 512 x k, k = 0..3: the own-slice triangle's launches cut 2^k times finer, option "tri_div"), `tri_first_pct` % of the own-slice triangle in its first launch; exchange_mode != 0 or world > 1: the three-launch pipeline with separate rows for the own-slice
 * triangle).  Per item 8 longs: first i slot, number of i bodies, j block, flags, row set (0 main,
 * 1 own-slice triangle), float offset of its i-side output, of its j-side output, launch (0, 1, 2).
 * flags, exactly as the kernel reads them (MurbSymItem::flags, csrc/murb_kernels_sym.h; SymPiece, csrc/murb_schedule.h):
 *   bit 0      nothing is written on the j side (a diagonal item in its plain form - the full square with only the
 *              i side kept - or the LAST triangular piece of a diagonal block, which has no later step to apply)
 *   bit 1      diagonal item in its triangular form ("diag_tri"): the piece skips the j steps before its own
 *   bits 8-11  triangular form: first j step (of 128 bodies) the piece evaluates
 *   bits 12-15 triangular form: first j step whose terms are applied to BOTH sides (the steps before it, i.e. the
 *              piece's own, keep the i side only)  Per row-table
 * entry 7 longs: row set, destination slice chunk, block inside it, offset and count of its i rows, offset and count of
 * its j rows (rows are 1024 slots).  NULL arrays query the counts.  Exists so that "every cell of every row has exactly
 * one writer" and "every pair is evaluated once" can be checked without a GPU. */
int murbhip_schedule_layout(unsigned long n, int world, int rank, int split, int waves, int taper_pct, int tri_first_pct,
                            int exchange_mode, long* items, unsigned long item_capacity, unsigned long* item_count, long* rows,
                            unsigned long row_capacity, unsigned long* row_count, unsigned long* floats_main,
                            unsigned long* floats_tri);

/* ------------------------------------------------------------------ life cycle */

/* Number of visible HIP devices. */
int murbhip_device_count(int* count);

/* One GPU, whole problem.  Replaces what the reference does at construction of a device
 * implementation: CUDABodies allocation (src/common/core/CUDABodies.cu:12-31), acceleration and
 * GM buffers (SimulationNBodyCUDATileFullDevice.cu:181-199).  `g` is the gravitational constant
 * (SimulationNBodyInterface.hpp:18), `soft` the softening length (squared inside). */
int murbhip_create(murbhip_ctx** out, unsigned long n, float soft, float g, int device);

/* One process driving `ndev` GPUs (bodies block-partitioned over them, positions exchanged every
 * step).  `devices` lists HIP device ordinals; the same ordinal may appear more than once (the
 * shards then share that GPU — used to exercise the sharded path on a one-GPU machine).
 * exchange: 0 = device-to-device copies and peer reads issued by this library, 1 = RCCL (all-gather of
 * positions, reduce-scatter of accelerations under the pair-symmetric schedule; ncclCommInitAll).
 * Threading: the CALLER stays single-threaded (the reference's driver contract, main.cpp:348-354), but the context owns
 * one host thread per shard that enqueues that shard's share of every step (murbhip_step returns when all of them have
 * finished enqueueing, never waits for the GPU); each thread drives its own communicator, no ncclGroupStart/End. */
int murbhip_create_sharded(murbhip_ctx** out, unsigned long n, float soft, float g, int ndev, const int* devices,
                           int exchange);

/* One process per GPU (torchrun / mpirun style).  Rank 0 calls murbhip_unique_id() and ships the
 * 128 bytes to every rank out of band; every rank then calls murbhip_create_rank().  Takes the place
 * of the reference's lazy MPI_Init/Comm_rank/Comm_size (SimulationNBodyMultiNode.cpp:62-73).
 * At most 64 ranks (MURBHIP_E_INVALID beyond: the per-slice tables of the half-ring schedule are fixed-size).
 * RCCL is bound at run time (librccl.so.1 by soname, so a host that already loaded RCCL shares it);
 * the environment variable MURBHIP_RCCL_LIBRARY names a specific library file to bind instead ("none": behave as
 * on a machine without RCCL: MURBHIP_E_NO_RCCL). */
int murbhip_unique_id(void* id_out /* MURBHIP_UNIQUE_ID_BYTES */);
int murbhip_create_rank(murbhip_ctx** out, unsigned long n, float soft, float g, int device, int rank, int world,
                        const void* unique_id);

int murbhip_destroy(murbhip_ctx* ctx);

/* ------------------------------------------------------------------ state in / out */

/* Host SoA -> device (all n bodies; every rank passes the full arrays, as every reference MPI rank
 * builds the full Bodies).  Replaces CUDABodies::memcpyBuffersOnDevice (CUDABodies.cu:34-49) and
 * devInitializeDevGM (SimulationNBodyCUDATileFullDevice.cu:41-45): G*m is folded in here. */
int murbhip_upload(murbhip_ctx* ctx, const float* qx, const float* qy, const float* qz, const float* vx,
                   const float* vy, const float* vz, const float* m);

/* Initial conditions generated ON THE DEVICE, instead of murbhip_upload: the n bodies of the reference's scheme "galaxy"
 * (Bodies::initGalaxy, src/common/core/Bodies.cpp:158-214) or "random" (initRandomly, :217-257) for srand(seed), bit-identical
 * to what the reference's host code computes on this machine — glibc's rand() sequence (jump-ahead on the linear TYPE_3
 * generator), the float/double mix of the reference's expressions as compiled with its flags, and glibc's sincosf
 * (csrc/murb_init.h).  Every shard fills its own copy of the replicated records; nothing crosses PCIe.  The host-side SIMD
 * padding bodies of the reference (Bodies.cpp:201-213) draw from rand() AFTER the n bodies and never reach the device.
 * Option "init_libm_fma": which build of glibc's sincosf to reproduce (1 = the -mfma one glibc selects on CPUs with FMA and
 * AVX2, 0 = the SSE2 one; -1, default = what this host's libm would pick). */
int murbhip_init_bodies(murbhip_ctx* ctx, const char* scheme, unsigned long seed);

/* Masses (and, after murbhip_init_bodies, radii; `r` may be NULL) of all n bodies, device -> host: what a host mirror
 * needs to complete its dataSoA when the bodies were created on the device (rank mode: own slice only). */
int murbhip_download_mass(murbhip_ctx* ctx, float* m, float* r);

/* Device -> host SoA of all n bodies; waits for enqueued steps first.  This is the lazy D2H behind
 * CUDABodies::getDataSoA() (CUDABodies.cu:64-93).  Any pointer may be NULL.  In rank mode velocities
 * are only known for the caller's own slice: entries of other ranks are left untouched. */
int murbhip_download_state(murbhip_ctx* ctx, float* qx, float* qy, float* qz, float* vx, float* vy, float* vz);

/* Accelerations used by the most recent step (or murbhip_compute_acc), n entries each; other ranks'
 * entries untouched in rank mode.  Test hook, like getAccSoA()
 * (SimulationNBodyCUDATileFullDevice200k.cu:179-189). */
int murbhip_download_acc(murbhip_ctx* ctx, float* ax, float* ay, float* az);

/* ------------------------------------------------------------------ compute */

/* a_i = sum_j G m_j (q_j - q_i) / (|q_j - q_i|^2 + soft^2)^(3/2) for the current positions, no
 * integration.  Enqueue only.  (computeBodiesAcceleration: SimulationNBodyOptim.cpp:34-94,
 * device twin SimulationNBodyCUDATileFullDevice.cu:53-153.)  The result is remembered: a second call
 * without a state change in between costs nothing, and a murbhip_step() that follows directly reuses
 * the forces instead of evaluating them again (one shard; bit-identical either way). */
int murbhip_compute_acc(murbhip_ctx* ctx);

/* Untimed device warm-up for about `milliseconds` (0 ... 10 000) of force evaluations on the current state, then a sync.
 * An MI355X needs ~40 ms of work to reach its steady clock after an idle spell (the first 12 ms run 25 % slow, DESIGN.md
 * §4.5) — as long as the reference's whole 200-iteration run at N = 30 000.  Construction is outside the reference's timing
 * window (main.cpp:353-371 times computeOneIteration() + the driver's sync only; the upload and the GM precompute of
 * SimulationNBodyCUDATileFullDevice.cu:191-215 are not in it), and this belongs there.  The state does not change and
 * nothing is remembered: the step that follows evaluates its own forces.  The number of evaluations is a function of n and
 * the number of ranks only (rank mode: the same count, hence the same collectives, on every rank). */
int murbhip_warmup(murbhip_ctx* ctx, double milliseconds);

/* One iteration = force + position/velocity update [+ position exchange].  Enqueue only.
 * (computeOneIteration: SimulationNBodyCUDATileFullDevice.cu:203-236; integrator semantics
 * Bodies.cpp:260-278 / CUDABodies.cu:126-153, including the fp64 intermediates.) */
int murbhip_step(murbhip_ctx* ctx, float dt);

/* `iterations` calls of murbhip_step in one go. */
int murbhip_steps(murbhip_ctx* ctx, float dt, int iterations);

/* Integrator alone with caller-supplied accelerations (host SoA, n entries): the overload
 * CUDABodies::updatePositionsAndVelocities(const accSoA_t&, T&) (CUDABodies.cu:355-370) that the
 * reference's test_CUDABodies.cpp:42-75 drives. */
int murbhip_integrate_host_acc(murbhip_ctx* ctx, const float* ax, const float* ay, const float* az, float dt);

/* Wait for everything enqueued on this context; returns the first asynchronous error, if any. */
int murbhip_sync(murbhip_ctx* ctx);

/* Mechanical energy of the current state: kinetic = sum 1/2 m v^2, potential = -1/2 sum_i sum_{j != i}
 * G m_i m_j / sqrt(r_ij^2 + soft^2) — the per-iteration metric of the reference's gpu+tracking
 * implementation (SimulationNBodyCUDAPropertyTracking.cu:217-304, summed there with cub).
 * Pair-symmetric plan (round 3): the potential comes out of a FORCE evaluation — two more packed instructions per 18 sum
 * G m_i G m_j / r of every pair a wave meets, one float per group of 4 i bodies behind the partial rows, fp64 from there on —
 * so there is no second N^2 sweep; the forces of that evaluation are remembered (bit-identical to a plain evaluation's),
 * and a step that follows directly only launches the state update (with several shards: state update + position
 * exchange).  A tracked iteration (energy, then step: `--im hip+tracking`) therefore costs ONE force evaluation: 6.9 ms
 * instead of 10.2 at N = 200 000 (6.1 untracked).  The pairs INSIDE a block of 1024 bodies are summed by a small kernel of their
 * own, in fp64 and without the bodies' own terms: the result is within 3e-8 of an fp64 evaluation from a few dozen bodies up.
 * Under the multi-pass evaluation (N > 2.4 M) the groups' sums are added up pass by pass.  In rank mode the potential covers the PAIRS this rank evaluated under the half-ring
 * schedule, the kinetic energy its own bodies: sum both over the ranks; the call is a collective (the force evaluation
 * contains the reduce-scatter) unless the forces of the current positions are already remembered.
 * One-sided plan (below 2 049 bodies; few bodies per rank) or option "energy_sweep" 1: one N^2 potential sweep on the device
 * (phi_i = sum_j G m_j / r), then -1/2 sum m_i phi_i; values cover the caller's own bodies.
 * The per-body terms are summed in fp64 on the device (256-body block sums in a fixed order; the host adds the few hundred
 * block rows); waits for enqueued steps. */
int murbhip_energy(murbhip_ctx* ctx, double* kinetic, double* potential);

/* First moments of the current state, fp64 sums (on the device, like murbhip_energy) over the caller's own bodies:
 *   out10 = { Px, Py, Pz,  Lx, Ly, Lz,  Mx, My, Mz,  M }
 * linear momentum sum m v, angular momentum sum m (q x v), mass-weighted position sum m q and total
 * mass (centre of mass = M{x,y,z} / M).  These fill the ang_momentum / density_center columns the
 * reference's SimulationHistory reserves but never computes (SimulationHistory.hpp:13-15,
 * SimulationNBodyCUDAPropertyTracking.cu:5-8: only COMPUTE_ENERGY_METRIC is enabled).  In rank mode
 * sum the ten values over ranks. */
int murbhip_moments(murbhip_ctx* ctx, double* out10);

/* ------------------------------------------------------------------ tuning and measurement */

/* Integer options.  Keys:
 *   "variant"        force kernel variant.  0 = auto: the pair-symmetric kernel (8) on one
 *                    GPU from 4 097 bodies (5 blocks of 1024) up, except at 6 blocks, and in multi-GPU runs when a rank gets
 *                    >= 400 block pairs, the one-sided kernel otherwise (1; 2 = four i bodies per wave for a rank's slice of up
 *                    to 16 384 bodies; one GPU: with the state update in the tail of its launch, see "fuse_integrate").  1-6: one-sided variants, 7: persistent schedule
 *   "jsplit"         one-sided variants: number of j-chunks a body's sum is split into; variant 7:
 *                    scheduling rounds; variant 8: i-side sub-blocks per item (1, 2, 4, 8, 16).  0 = auto
 *   "taper"          variant 8: percentage (0..100) of each launch's work whose items are cut finer (the last
 *                    taper % in halves, the last taper/2 % in quarters): a shorter drain phase at the end of a launch.
 *                    -1 (default) = the plan's own choice
 *   "sym_pass_mb"    variant 8, one GPU: budget in MiB for the partial sums of one pass (0 = default: a quarter of the
 *                    device memory).  A problem whose partial sums exceed it (N > ~2.4 M bodies by default) is
 *                    evaluated in several passes over ranges of j columns that share one buffer, their row sums
 *                    accumulated in fp64 ("sym_passes" of murbhip_get_info says how many)
 *   "diag_tri"       variant 8: 1 = a diagonal block (i block = j block) is cut into pieces of 128 i bodies that only
 *                    evaluate the j bodies from their own position on (36 instead of 64 units of work per diagonal
 *                    block); 0 = the full square with the i side kept.  -1 (default) = the plan's own choice
 *   "sym_red"        variant 8: how the i-side sums of a group are folded over the wave: 0 = in registers (permlane
 *                    swaps + DPP), 1 = through LDS (fewer VALU instructions).  -1 (default) = the plan's own choice
 *   "sym_waves"      variant 8: waves per workgroup, 4 or 8; 0 = auto (one GPU up to 27 blocks: 8 or 4 by a measured table per
 *                    block count together with the item length, profiles/r03_small_plan_table.txt; 4 otherwise)
 *   "pad_aware"      variant 8: 1 (default) = the zero-mass padding slots that fill a slice up to whole blocks of 1024 are
 *                    not walked: the emptier block of a pair goes on the walked (i) side and its items end at its last
 *                    real body; 0 = every block as if full (kept for the A/B: -3 % at N = 30 000, -4 % for a rank of 8
 *                    at N = 200 000)
 *   "tri_div"        variant 8, several ranks: the items of the own-slice triangle's two launches (which run under the two
 *                    collectives and, with few blocks per slice, do not fill the chip) cut into 1, 2, 4 or 8 parts more
 *                    than the rectangles' items; 0 (default) = the plan's choice (~2 rounds of workgroups per launch)
 *   "energy_sweep"   murbhip_energy on a pair-symmetric plan: 1 = the separate potential sweep of rounds 1-2 instead of the
 *                    pair potential summed inside a force evaluation (default 0); kept for the A/B and as a cross-check
 *   "xcd_order"      variant 8: 0 (default) = j-major item order (round-robin dispatch then gives XCD x the i
 *                    blocks x mod 8 of every j block); 1 = one contiguous run of items per XCD (measured:
 *                    more L2 misses, same time; kept for the comparison)
 *   "profile"        1: bracket every force kernel with HIP events (read with murbhip_get_info); 2: also both collectives
 *                    on the exchange stream, the compute stream's waits for them (the EXPOSED part of the exchange) and
 *                    the compute stream's whole step - ~16 more event records per step, meant for a short diagnostic
 *                    run next to the timed one.  Setting it (to any value) drains the device and clears the samples
 *   "overlap"        sharded/rank mode: 0 = no overlap; 1 (default) = the own-slice work brackets the
 *                    exchanges on the compute stream; 2 = the own-slice triangle runs on a second,
 *                    lowest-priority compute stream next to the rectangle launch (pair-symmetric only)
 *   "integrator"     0 (default) = the reference's update, Bodies.cpp:260-278; 1 = kick-drift-kick
 *                    leapfrog, the scheme the reference's gpu+leapfrog states (CUDABodies.cu:172-178) with
 *                    the force taken at the positions it belongs to: one force evaluation per step, the
 *                    device keeps v_{n-1/2}, murbhip_download_state applies the closing half kick (one
 *                    extra force evaluation; a collective in rank mode).  Cannot be changed between a
 *                    leapfrog step and the next murbhip_upload (MURBHIP_E_STATE)
 *   "tri_first_pct"  "overlap" 1, pair-symmetric schedule: percentage (0..100, default 50) of the own-slice
 *                    triangle that is launched before the rectangles, i.e. under the all-gather of positions;
 *                    the rest runs under the reduce-scatter of accelerations.  A tuning knob for real
 *                    interconnect latencies (bench.py picks it per run, untimed)
 *   "exchange_p2p"   RCCL exchange only (MURBHIP_E_STATE otherwise): 1 = both exchanges of a step as grouped ncclSend / ncclRecv
 *                    instead of collectives.  Accelerations: under the half-ring schedule a rank only has contributions for
 *                    the floor(W/2) slices ahead of it, so it sends those chunks straight to their owners and adds up the
 *                    floor(W/2) it receives (ncclReduceScatter moves and adds zeros for the other half); positions: every
 *                    slice straight to every peer.  One hop per message on a fully connected xGMI node.  0 (default) =
 *                    ncclReduceScatter + ncclAllGather.  UNMEASURED on hardware: bench.py times it beside the default for
 *                    N > 1 ("p2p_plan") so that the first multi-GPU run shows which to prefer
 *   "cu_reserve"     k >= 0 (default 0): the compute streams are re-created with a CU mask that leaves the k highest
 *                    CUs (8 = one per XCD, 16 = two per XCD) to the exchange stream.  The force kernels otherwise fill
 *                    every CU, and a collective's kernel (RCCL) has to wait ~0.1 ms for one of their workgroups to retire
 *                    (7 us with k = 8); but dispatch on a masked queue is slower: 6-8 % on long force launches, much more
 *                    on short ones.  bench.py times it per run for N > 1, like "tri_first_pct".
 *                    Two side effects of hipExtStreamCreateWithCUMask, which has no flags argument: (i) a masked stream
 *                    is BLOCKING with respect to the device's null stream, so a host application that works on stream 0
 *                    of the same device (torch's default stream) synchronises with the force kernels implicitly;
 *                    (ii) it has default priority.  The low-priority stream of "overlap" 2 therefore stays unmasked
 *                    (it keeps its priority and may use the reserved CUs)
 *   "fuse_integrate" 1 (default): one GPU, one-sided kernel with all j in one chunk (the default up to 6 blocks): the state
 *                    update runs in the tail of the force launch — one launch per step instead of two, bit-identical results
 *                    (N = 2 048: 7.2 instead of 13.4 us per step).  0 = two launches, and the round-2 rule for "variant" 0
 *                    (pair-symmetric from 3 blocks up)
 *   "solo_shard"     r >= 0: in a sharded context only shard r launches force work (timing aid: the
 *                    isolated per-step timeline of one rank of W; results are meaningless).  -1 = off
 *   "force_exchange" 1: run the position exchange even with a single rank/shard (self-test of the
 *                    RCCL binding on a one-GPU machine; rank mode needs a unique id at creation)
 */
int murbhip_set_option(murbhip_ctx* ctx, const char* key, long value);

/* Numeric facts.  Keys: "cu_count", "clock_mhz", "n", "slots", "world", "rank", "jsplit", "variant", "cu_reserve", "sym_passes", "sym_waves", "taper",
 * "workgroups", "interactions_per_launch", "device_bytes", and the timing spans of the steps since "profile" was set (HIP
 * events on the library's own streams, all shards of this process; the call drains the device):
 *   "force_launches", "force_ms_avg", "force_ms_total"      every force launch
 *   "span_<kind>_ms_avg", "span_<kind>_count"               kind = tri1 | rect | tri2 (the three force launches of the exchange
 *       pipeline), reduce_scatter | all_gather (exchange stream: from "my input is ready" to "my output has arrived"),
 *       wait_gather | wait_reduce (compute stream idle, waiting for that collective), step (compute stream, first launch of
 *       a step to the end of its state update); the last five need "profile" 2
 *   "compute_wait_ms_per_step"                              (wait_gather + wait_reduce) per profiled step
 *   "spans_dropped"                                         1 when the event pool (4096 spans per shard) ran out
 *   "sym_launches"                                          pair-symmetric launches of any form (force, force + pair potential,
 *                                                           potential sweep) since "profile" was last set */
int murbhip_get_info(murbhip_ctx* ctx, const char* key, double* value);

#ifdef __cplusplus
}
#endif
#endif /* MURBHIP_H_ */
