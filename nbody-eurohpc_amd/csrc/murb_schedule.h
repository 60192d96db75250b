// Host-only scheduling rules (no HIP call): the reference's body partition, and which (i sub-block, j block)
// items a rank evaluates under the half-ring pair-symmetric schedule.  "Every unordered body pair is
// evaluated by exactly one rank" lives here and is unit-tested on the CPU through murbhip_schedule_items().
// Included by murbhip.hip only.
#ifndef MURB_SCHEDULE_H_
#define MURB_SCHEDULE_H_

#include <algorithm>
#include <vector>

#include "murb_layout.h"

namespace {

// Reference rule counts[r] = n/R + (r < n%R), displs = prefix sums (SimulationNBodyMultiNode.cpp:76-91).
void partition(unsigned long n, int world, int rank, unsigned long* first, unsigned long* count)
{
    const unsigned long base = n / (unsigned long)world, rem = n % (unsigned long)world;
    const unsigned long r = (unsigned long)rank;
    *count = base + (r < rem ? 1 : 0);
    *first = r * base + std::min(r, rem);
}

unsigned long slice_slots(unsigned long n, int world)
{
    unsigned long first, count;
    partition(n, world, 0, &first, &count);   // rank 0 always holds the largest slice
    return murb_round_up_tile(std::max(count, 1ul));
}

// Items (i-side sub-block, j-side block) of rank r under the half-ring pair-symmetric schedule, in launch
// order; the first `*own` entries are the own-slice triangle.  Host only (no HIP call): the rule that
// every unordered body pair is evaluated by exactly one rank lives here and is unit-tested on the CPU.
//   own slice x own slice : sub-block i against block j of the same slice, block(i) <= j
//   own slice x slice r+d : d = 1 .. floor(W/2), all (sub-block, block) combinations; for even W the slice
//                           pair half a ring apart is shared and cut at a block boundary of the LOWER
//                           rank's slice: the lower rank walks its first ceil(tb/2) blocks against all of
//                           the other slice, the higher rank walks all of its own against the rest
// Workgroups are dealt to the 8 XCDs round-robin (workgroup b of a launch runs on XCD b % 8) and every XCD
// has its own L2.  The plain j-major table order is already XCD-friendly: the items of one j block are
// consecutive, so XCD x gets the i blocks I = x mod 8 of EVERY j block — its L2 keeps one eighth of the
// bodies (0.4 MB at N=200k) for the whole launch and only the 16 KiB j tile is fetched by all eight.
// Option "xcd_order" = 1 instead cuts a launch's items into 8 contiguous runs, one per XCD (each XCD then
// walks few j blocks but ALL i blocks): measured 35 % MORE L2 misses (FETCH_SIZE 216 vs 160 MiB per launch
// at N=200k) and no time difference (tools/xcd_ab.py) — kept only for that comparison.
constexpr int kXcds = 8;
void xcd_interleave(std::vector<int>& flat, size_t first_item, size_t end_item)
{
    const size_t n = end_item - first_item;
    if (n < 2 * kXcds) return;
    std::vector<int> src(flat.begin() + 2 * first_item, flat.begin() + 2 * end_item);
    // run x = source items [start(x), start(x + 1)); position p of the launch takes item p / 8 of run p % 8
    auto start = [&](size_t x) { return x * n / kXcds; };
    size_t p = 0;
    for (size_t k = 0; p < n; ++k)
        for (size_t x = 0; x < (size_t)kXcds && p < n; ++x) {
            if (start(x) + k >= start(x + 1)) continue;   // this run is one item shorter
            const size_t it = start(x) + k;
            flat[2 * (first_item + p)] = src[2 * it];
            flat[2 * (first_item + p) + 1] = src[2 * it + 1];
            ++p;
        }
}

// How many REAL bodies each block holds: slice s keeps its counts[s] bodies in its first slots, the rest of its tb blocks
// is zero-mass padding (at most the last block of a slice is partly filled; a whole block is empty only when a slice
// holds one body fewer than the largest and that body was alone in its block — such a block counts as holding one body,
// so that every block keeps its items and its entry in the row tables, which the fused row sum + update indexes by block).
struct SymFill {
    int tb = 0;
    std::vector<int> count;   // bodies per slice
    int of(int block) const
    {
        const int left = count[(size_t)(block / tb)] - (block % tb) * MURB_SLICE_ALIGN;
        return left <= 1 ? 1 : (left >= MURB_SLICE_ALIGN ? MURB_SLICE_ALIGN : left);
    }
};
inline SymFill sym_fill(unsigned long n, int world, bool aware = true)
{
    SymFill f;
    f.tb = (int)(slice_slots(n, world) / MURB_SLICE_ALIGN);
    for (int s = 0; s < world; ++s) {
        unsigned long first, count;
        partition(n, world, s, &first, &count);
        f.count.push_back(aware ? (int)count : f.tb * MURB_SLICE_ALIGN);   // not aware ("pad_aware" 0): every block counts as full
    }
    return f;
}

// Which side of a block pair is walked (i side) and which is staged in LDS (j side).  The kernel's cost is (i bodies
// walked) x 1024: the j block is always staged whole, the i range can be any multiple of 16 x waves bodies.  So the block
// with FEWER real bodies goes on the i side and only its real part is walked — padding bodies have zero mass and
// contribute exactly 0 to every sum, and nobody reads theirs.  (N = 30 000: the last of 30 blocks holds 304 bodies and
// takes part in 30 of the 465 block pairs; a rank of 8 at N = 200 000: 424 of 1024 in the last block of every slice,
// 7.6 % of a rank's block pairs.)  `def_i`, `def_j`: the orientation the schedule would otherwise use.
inline void sym_orient(const SymFill& fill, int def_i, int def_j, int* i_block, int* j_block)
{
    const bool swap = def_i != def_j && fill.of(def_j) < fill.of(def_i);
    *i_block = swap ? def_j : def_i;
    *j_block = swap ? def_i : def_j;
}
// the sub-blocks of block `i_block` that hold real bodies, against block `j_block`
inline void sym_push_block_pair(const SymFill& fill, int split, int i_block, int j_block, std::vector<int>& flat)
{
    const int len = MURB_SLICE_ALIGN / split, real = fill.of(i_block);
    for (int q = 0; q < split && q * len < real; ++q) { flat.push_back(i_block * split + q); flat.push_back(j_block); }
}

void sym_schedule_items(int world, int rank, int tb, int split, const SymFill& fill, std::vector<int>& flat, int* own, bool xcd_order = false)
{
    const int W = world, r = rank;
    flat.clear();
    for (int j = 0; j < tb; ++j)
        for (int a = 0; a <= j; ++a) {
            int ib, jb;
            sym_orient(fill, r * tb + a, r * tb + j, &ib, &jb);
            sym_push_block_pair(fill, split, ib, jb, flat);
        }
    *own = (int)flat.size() / 2;
    for (int d = 1; d <= W / 2; ++d) {
        const int s = (r + d) % W;
        if (s == r) continue;
        const bool shared = (W % 2 == 0) && d == W / 2;
        const int lo = std::min(r, s);
        // the shared slice pair is cut at a block boundary of the LOWER rank's slice so that both halves hold the same
        // number of real bodies as nearly as possible (the last block of a slice is partly empty and cheap to walk);
        // both ranks compute the same cut
        int hb = (tb + 1) / 2;
        if (shared) {
            long total = 0, acc = 0, best = -1;
            for (int a = 0; a < tb; ++a) total += fill.of(lo * tb + a);
            for (int h = 0; h <= tb; ++h) {
                const long off = 2 * acc > total ? 2 * acc - total : total - 2 * acc;
                if (best < 0 || off <= best) { best = off; hb = h; }
                if (h < tb) acc += fill.of(lo * tb + h);
            }
        }
        std::vector<int> swapped;
        for (int a = 0; a < tb; ++a) {
            const size_t row_first = flat.size();
            for (int j = 0; j < tb; ++j) {
                // a: OWN block (by default walked, i side); j: the other slice's block (LDS resident, j side)
                if (shared && !((r == lo) ? (a < hb) : (j >= hb))) continue;
                int ib, jb;
                sym_orient(fill, r * tb + a, s * tb + j, &ib, &jb);
                if (ib == r * tb + a) { flat.push_back(ib); flat.push_back(jb); }   // expanded into sub-blocks below
                else sym_push_block_pair(fill, split, ib, jb, swapped);             // the other slice's emptier block is walked
            }
            // i-major order as before: sub-block q of block a against every j of the row, then q + 1
            const std::vector<int> row(flat.begin() + (long)row_first, flat.end());
            flat.resize(row_first);
            const int len = MURB_SLICE_ALIGN / split, real = fill.of(r * tb + a);
            for (int q = 0; q < split && q * len < real; ++q)
                for (size_t k = 0; k < row.size(); k += 2) { flat.push_back(row[k] * split + q); flat.push_back(row[k + 1]); }
        }
        flat.insert(flat.end(), swapped.begin(), swapped.end());
    }
    // the three launches of a step: first half of the own-slice triangle, second half, rectangles
    if (!xcd_order) return;
    const size_t n_own = (size_t)*own, n_all = flat.size() / 2;
    if (W == 1) { xcd_interleave(flat, 0, n_all); return; }   // one launch
    xcd_interleave(flat, 0, n_own / 2);
    xcd_interleave(flat, n_own / 2, n_own);
    xcd_interleave(flat, n_own, n_all);
}

// A piece of work for one workgroup of the pair-symmetric kernel: the i bodies [i_slot0, i_slot0 + len) against j
// block J.  flags (= MurbSymItem::flags): bit 0 = nothing is written on the j side; bit 1 = diagonal item in its
// triangular form, with bits 8-11 = first step evaluated, bits 12-15 = first step that applies both sides (a step = the
// 128 bodies a lane's p-th pair vector covers).
struct SymPiece {
    int i_slot0, len, J, flags;
    bool diag() const { return i_slot0 / MURB_SLICE_ALIGN == J; }
    bool j_side() const { return (flags & 1) == 0; }
};
constexpr int kSymStepBodies = 128;

// (i sub-block, j block) pairs -> pieces.
//   taper_pct > 0 cuts the i side of the LAST items of each launch finer, geometrically: the last taper_pct % of a
//     launch's work in halves, the last taper_pct / 2 % in quarters, the last taper_pct / 4 % in eighths, ... (never
//     below `min_len` bodies).  The hardware deals workgroups in table order and a CU works through ~75 items of 0.33 ms
//     four at a time (N = 200 000): at the end of a launch the CUs run dry up to one item time apart, and the last
//     workgroups of a CU run alone at 61 % of the issue rate.  With the tail cut fine both effects shrink
//     with the item length.  `launch_ends` lists the item indices (of `flat`) at which a launch ends (the last one =
//     number of items).
//   diag_tri: a diagonal block (i block == j block) is cut into pieces of at most 128 bodies, each evaluating only the
//     j steps from its own on: its own step one-sided, the later ones both ways — 36 instead of 64 step units per
//     diagonal block (plain form: the full square, i side kept).
//   fill: a piece that lies entirely in a block's padding is dropped, one that straddles the end of the real bodies is cut
//     to the next multiple of `min_len` (sym_orient put the emptier block of a pair on this, the i side).
//   launch_div (may be empty): every item of launch l is cut into launch_div[l] (a power of two) equal parts at least —
//     for launches that would otherwise not fill the chip (the own-slice triangle parts of a rank of 8).
inline void sym_pieces(const std::vector<int>& flat, int split, int taper_pct, int min_len, bool diag_tri, const SymFill& fill,
                       const std::vector<size_t>& launch_ends, std::vector<SymPiece>& out, std::vector<size_t>& piece_launch_ends,
                       const std::vector<int>& launch_div = std::vector<int>())
{
    const int len = MURB_SLICE_ALIGN / split;
    out.clear();
    piece_launch_ends.clear();
    size_t first = 0;
    for (size_t l = 0; l < launch_ends.size(); ++l) {
        const size_t end = launch_ends[l];
        const int base_div = l < launch_div.size() ? launch_div[l] : 1;
        const double total = (double)(end - first);
        for (size_t k = first; k < end; ++k) {
            const int isub = flat[2 * k], J = flat[2 * k + 1];
            const bool diag = isub / split == J;
            const double before = (double)(k - first) / (total > 0 ? total : 1.0);
            int div = 1;
            while (div < base_div && len / (2 * div) >= min_len) div *= 2;
            if (taper_pct > 0) {
                const double left = 1.0 - before;   // share of the launch still to be dealt, this item included
                for (double f = taper_pct / 100.0; left <= f && len / (2 * div) >= min_len; f *= 0.5) div *= 2;
            }
            if (diag && diag_tri)
                while (len / div > kSymStepBodies && len / (2 * div) >= min_len) div *= 2;
            const int real_end = (isub / split) * MURB_SLICE_ALIGN + fill.of(isub / split);   // first padding slot of the i block
            for (int q = 0; q < div; ++q) {
                SymPiece pc{isub * len + q * (len / div), len / div, J, diag ? 1 : 0};
                if (pc.i_slot0 >= real_end) break;
                if (pc.i_slot0 + pc.len > real_end) pc.len = (real_end - pc.i_slot0 + min_len - 1) / min_len * min_len;
                if (diag && diag_tri && pc.len <= kSymStepBodies) {
                    const int in_block = pc.i_slot0 % MURB_SLICE_ALIGN;
                    const int a = in_block / kSymStepBodies, b = (in_block + pc.len + kSymStepBodies - 1) / kSymStepBodies;
                    pc.flags = 2 | (a << 8) | (b << 12) | (b >= MURB_SLICE_ALIGN / kSymStepBodies ? 1 : 0);
                }
                out.push_back(pc);
            }
        }
        piece_launch_ends.push_back(out.size());
        first = end;
    }
}

}  // namespace

#endif
