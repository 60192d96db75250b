// The host-side planner of the pair-symmetric launches: item table + partial-row layout of one rank, computed without touching
// a device (no HIP in here).  Unit-tested on the CPU through murbhip_schedule_layout (tests/test_abi_and_host.py) and, under
// AddressSanitizer / UBSan, by tests/helpers/plan_selftest.cpp.  Included by murbhip.hip only.
#ifndef MURB_PLAN_H_
#define MURB_PLAN_H_

#include <cstddef>
#include <map>
#include <utility>
#include <vector>

#include "murb_schedule.h"
#include "murb_sym_types.h"

namespace {

// One GPU, very large N: the items are evaluated in several passes (ranges of j columns) that reuse ONE buffer of partial
// rows; the row sums of the passes are accumulated in fp64.  Everything else has a single pass.
struct SymPass {
    int item_first = 0, item_count = 0;    // items of the pass
    int table_first = 0, table_count = 0;  // its entries of the row table
    size_t floats = 0;                     // floats per component of its layout (= comp_stride of its launches)
};

// Cut the pieces (j-major order) into passes at column boundaries so that no pass needs more than budget_floats of
// partial rows per component; 0 = no limit (one pass).  Conservative estimate per column: every piece owns a j row, every
// distinct (i block, column) pair an i row.
std::vector<std::pair<size_t, size_t>> cut_passes(const std::vector<SymPiece>& pieces, size_t budget_floats)
{
    std::vector<std::pair<size_t, size_t>> out;
    size_t first = 0, used = 0, k = 0;
    while (k < pieces.size()) {
        size_t e = k, rows = 0;
        int last_block = -1;
        while (e < pieces.size() && pieces[e].J == pieces[k].J) {   // one column
            rows += pieces[e].j_side() ? 1 : 0;
            if (pieces[e].i_slot0 / MURB_SYM_BLOCK != last_block) { ++rows; last_block = pieces[e].i_slot0 / MURB_SYM_BLOCK; }
            ++e;
        }
        const size_t need = rows * MURB_SYM_BLOCK;
        if (budget_floats && used > 0 && used + need > budget_floats) { out.emplace_back(first, k); first = k; used = 0; }
        used += need;
        k = e;
    }
    out.emplace_back(first, pieces.size());
    return out;
}

// Lay out the partial rows of pieces [first, end): per block touched, an "i rows" matrix (one 1024-slot row per j block
// its bodies were walked against) and a "j rows" matrix (one row per piece that had it as j block), end to end.  Fills
// the pieces' output offsets into `items` and returns the per-block table; `out_of(block)` says where a block's row
// sums go (slice chunk, block inside it).
template <class OutOf>
size_t layout_sym_set(const std::vector<SymPiece>& pieces, size_t first, size_t end, int waves, std::vector<MurbSymItem>& items,
                      std::vector<MurbSymBlockRows>& table, OutOf out_of)
{
    std::map<int, int> index;                       // global block -> table entry, ascending block order
    for (size_t k = first; k < end; ++k) { index[pieces[k].i_slot0 / MURB_SYM_BLOCK] = 0; if (pieces[k].j_side()) index[pieces[k].J] = 0; }
    table.assign(index.size(), MurbSymBlockRows{});
    { int e = 0; for (auto& kv : index) { kv.second = e; const auto o = out_of(kv.first); table[e].out_slice = o.first; table[e].out_block = o.second; ++e; } }
    std::vector<std::map<int, int>> irow(index.size());   // per block: j block -> i row
    std::vector<int> item_irow(end - first), item_jrow(end - first);
    for (size_t k = first; k < end; ++k) {
        const SymPiece& pc = pieces[k];
        const int bi = index[pc.i_slot0 / MURB_SYM_BLOCK];
        auto f = irow[bi].find(pc.J);
        if (f == irow[bi].end()) f = irow[bi].emplace(pc.J, table[bi].ni++).first;
        item_irow[k - first] = f->second;
        item_jrow[k - first] = pc.j_side() ? table[index[pc.J]].nj++ : -1;
    }
    size_t floats = 0;
    for (MurbSymBlockRows& br : table) {
        br.base_j = floats; floats += (size_t)br.nj * MURB_SYM_BLOCK;
        br.base_i = floats; floats += (size_t)br.ni * MURB_SYM_BLOCK;
    }
    for (size_t k = first; k < end; ++k) {
        const SymPiece& pc = pieces[k];
        MurbSymItem& it = items[k];
        it.i_slot0 = pc.i_slot0;
        it.ngroups = pc.len / (waves * MURB_SYM_R);
        it.J = pc.J;
        it.flags = pc.flags;
        const MurbSymBlockRows& bi = table[index[pc.i_slot0 / MURB_SYM_BLOCK]];
        it.ioff = bi.base_i + (size_t)item_irow[k - first] * MURB_SYM_BLOCK + (size_t)(pc.i_slot0 % MURB_SYM_BLOCK);
        it.joff = pc.j_side() ? table[index[pc.J]].base_j + (size_t)item_jrow[k - first] * MURB_SYM_BLOCK : 0;
    }
    return floats;
}

// Everything the pair-symmetric launches of one rank need, computed on the host without touching a device (unit-tested
// on the CPU through murbhip_schedule_layout): the item table in launch order ([0, own) = own-slice triangle, its first
// t1 items forming the launch that runs under the position gather), and the partial-row layout of the two sets.
struct SymHostLayout {
    std::vector<MurbSymItem> items;
    int own = 0, t1 = 0;
    std::vector<MurbSymBlockRows> table_main, table_tri;
    size_t floats_main = 0, floats_tri = 0;   // floats per component of the buffers (main: the largest pass)
    std::vector<SymPass> passes;              // of the main set
};

void plan_sym_layout(int W, int r, const SymFill& fill, int split, int waves, int taper, bool diag_tri, bool exchange_mode, int overlap,
                     int tri_first_pct, bool xcd_order, size_t budget_floats, SymHostLayout& L, int tri_div = 1)
{
    const int tb = fill.tb;
    std::vector<int> flat;
    int own = 0;
    size_t t1 = 0;
    std::vector<SymPiece> pieces;
    std::vector<size_t> piece_ends;
    const auto build_pieces = [&](bool interleaved) {
        sym_schedule_items(W, r, tb, split, fill, flat, &own, interleaved);
        // the launches of a step: one GPU = everything; exchange pipeline = triangle part 1, part 2, rectangles
        const size_t n_all = flat.size() / 2;
        t1 = (exchange_mode && overlap == 1) ? (size_t)((long)own * tri_first_pct / 100) : 0;
        std::vector<size_t> launch_ends;
        std::vector<int> launch_div;   // the own-slice triangle's launches in finer items ("tri_div")
        if (exchange_mode) {
            if (t1 > 0) { launch_ends.push_back(t1); launch_div.push_back(tri_div); }
            if ((size_t)own > t1) { launch_ends.push_back((size_t)own); launch_div.push_back(tri_div); }
            if (n_all > (size_t)own) { launch_ends.push_back(n_all); launch_div.push_back(1); }
        } else {
            launch_ends.push_back(n_all);
        }
        sym_pieces(flat, split, taper, 16 * waves, diag_tri, fill, launch_ends, pieces, piece_ends, launch_div);
    };
    build_pieces(xcd_order);
    // Several passes share ONE row buffer and are cut at COLUMN boundaries (cut_passes takes a column to be a contiguous
    // run of pieces with the same J): under the XCD-interleaved order a column is scattered over the table, a cut would
    // fall inside it and an i row would keep an earlier pass's sums in the cells this pass does not write.  A problem
    // that needs several passes is therefore always laid out in the plain j-major order.
    if (!exchange_mode && xcd_order && cut_passes(pieces, budget_floats).size() > 1) build_pieces(false);
    // pieces of the own-slice triangle = those of the launches before the rectangles' (a rectangle item may have an own
    // block on its j side: sym_orient puts the emptier block of a pair on the i side)
    size_t own_pieces = pieces.size();
    if (exchange_mode && flat.size() / 2 > (size_t)own) own_pieces = piece_ends.size() >= 2 ? piece_ends[piece_ends.size() - 2] : 0;
    L.items.assign(pieces.size(), MurbSymItem{});
    L.table_main.clear(); L.table_tri.clear();
    L.floats_main = L.floats_tri = 0;
    if (exchange_mode) {
        L.floats_tri = layout_sym_set(pieces, 0, own_pieces, waves, L.items, L.table_tri, [&](int b) { return std::make_pair(0, b - r * tb); });
        L.floats_main = layout_sym_set(pieces, own_pieces, pieces.size(), waves, L.items, L.table_main,
                                       [&](int b) { return std::make_pair(b / tb, b % tb); });
        L.passes.assign(1, SymPass{(int)own_pieces, (int)(pieces.size() - own_pieces), 0, (int)L.table_main.size(), L.floats_main});
    } else {
        L.passes.clear();
        std::vector<MurbSymBlockRows> table;
        for (const auto& range : cut_passes(pieces, budget_floats)) {
            const size_t floats = layout_sym_set(pieces, range.first, range.second, waves, L.items, table, [&](int b) { return std::make_pair(0, b); });
            L.passes.push_back(SymPass{(int)range.first, (int)(range.second - range.first), (int)L.table_main.size(), (int)table.size(), floats});
            L.table_main.insert(L.table_main.end(), table.begin(), table.end());
            L.floats_main = std::max(L.floats_main, floats);
        }
    }
    L.own = (int)own_pieces;
    L.t1 = t1 > 0 ? (int)piece_ends[0] : 0;
}

}  // namespace

#endif
