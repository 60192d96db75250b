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

void sym_schedule_items(int world, int rank, int tb, int split, std::vector<int>& flat, int* own, bool xcd_order = false)
{
    const int W = world, r = rank, ts = tb * split;
    flat.clear();
    for (int j = 0; j < tb; ++j)
        for (int i = 0; i < (j + 1) * split; ++i) { flat.push_back(r * ts + i); flat.push_back(r * tb + j); }
    *own = (int)flat.size() / 2;
    for (int d = 1; d <= W / 2; ++d) {
        const int s = (r + d) % W;
        if (s == r) continue;
        const bool shared = (W % 2 == 0) && d == W / 2;
        const int lo = std::min(r, s), hb = (tb + 1) / 2;
        for (int i = 0; i < ts; ++i)
            for (int j = 0; j < tb; ++j) {
                // i: OWN sub-blocks (walked, i side); j: the other slice's blocks (LDS resident, j side)
                if (shared && !((r == lo) ? (i / split < hb) : (j >= hb))) continue;
                flat.push_back(r * ts + i);
                flat.push_back(s * tb + j);
            }
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
// block J; `diag` = the i range lies inside block J (full square evaluated, i side kept).
struct SymPiece {
    int i_slot0, len, J;
    bool diag;
};

// (i sub-block, j block) pairs -> pieces.  `taper_pct` > 0 cuts the i side of the LAST items of each launch finer:
// the last taper_pct % of a launch's work in halves, the last taper_pct / 2 % in quarters (never below `min_len`
// bodies).  The hardware deals workgroups in table order, so the drain phase of a launch — when the last workgroups
// of each CU run alone at 61 % of the issue rate — then consists of short items.  `launch_ends` lists
// the item indices (of `flat`) at which a launch ends (the last one = number of items).
inline void sym_pieces(const std::vector<int>& flat, int split, int taper_pct, int min_len, const std::vector<size_t>& launch_ends,
                       std::vector<SymPiece>& out, std::vector<size_t>& piece_launch_ends)
{
    const int len = MURB_SLICE_ALIGN / split;
    out.clear();
    piece_launch_ends.clear();
    size_t first = 0;
    for (size_t end : launch_ends) {
        const double total = (double)(end - first);
        for (size_t k = first; k < end; ++k) {
            const int isub = flat[2 * k], J = flat[2 * k + 1];
            const double before = (double)(k - first) / (total > 0 ? total : 1.0);
            int div = 1;
            if (taper_pct > 0) {
                if (before >= 1.0 - taper_pct / 200.0) div = 4;
                else if (before >= 1.0 - taper_pct / 100.0) div = 2;
                while (div > 1 && len / div < min_len) div /= 2;
            }
            for (int q = 0; q < div; ++q)
                out.push_back(SymPiece{isub * len + q * (len / div), len / div, J, isub / split == J});
        }
        piece_launch_ends.push_back(out.size());
        first = end;
    }
}

}  // namespace

#endif
