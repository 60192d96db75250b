// TEST INFRASTRUCTURE: the host-side planner of the pair-symmetric launches (csrc/murb_plan.h + murb_schedule.h, no HIP in
// them) compiled with g++ under AddressSanitizer and UBSan and swept over many (n, ranks, split, waves, taper, ...) plans:
// every item's two outputs lie inside the buffer the plan asks for, no partial-row cell has two writers, the row tables cover
// exactly the rows the items write, the passes partition the items — and the sanitizers see every index the planner computes.
//   plan_selftest        prints "ok <plans checked>" and exits 0
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "murb_plan.h"

#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "n=%lu W=%d r=%d split=%d waves=%d taper=%d tri=%d ex=%d div=%d xcd=%d budget=%zu: ", n, W, r, split, waves, taper, (int)diag_tri, (int)exchange, tri_div, (int)xcd, budget); \
    std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); return 1; } } while (0)

static int check_set(const std::vector<MurbSymItem>& items, size_t first, size_t end, const std::vector<MurbSymBlockRows>& table, size_t floats,
                     int waves, const char** what)
{
    std::vector<unsigned char> writers(floats, 0), in_table(floats, 0);
    for (const MurbSymBlockRows& br : table) {
        if (br.base_j + (size_t)br.nj * MURB_SYM_BLOCK > floats || br.base_i + (size_t)br.ni * MURB_SYM_BLOCK > floats) { *what = "row table outside the buffer"; return 1; }
        for (size_t k = br.base_j; k < br.base_j + (size_t)br.nj * MURB_SYM_BLOCK; ++k) if (in_table[k]++) { *what = "two table entries share a row"; return 1; }
        for (size_t k = br.base_i; k < br.base_i + (size_t)br.ni * MURB_SYM_BLOCK; ++k) if (in_table[k]++) { *what = "two table entries share a row"; return 1; }
    }
    for (size_t k = first; k < end; ++k) {
        const MurbSymItem& it = items[k];
        const size_t len = (size_t)it.ngroups * waves * MURB_SYM_R;
        if (it.ngroups < 1 || it.i_slot0 % (waves * MURB_SYM_R) != 0 || it.i_slot0 / MURB_SYM_BLOCK != (int)((it.i_slot0 + len - 1) / MURB_SYM_BLOCK)) { *what = "i range leaves its block"; return 1; }
        if (it.ioff + len > floats) { *what = "i-side output outside the buffer"; return 1; }
        for (size_t c = it.ioff; c < it.ioff + len; ++c) { if (writers[c]++) { *what = "two writers for an i-row cell"; return 1; } if (!in_table[c]) { *what = "an item writes outside the row tables"; return 1; } }
        if (!(it.flags & 1)) {
            if (it.joff + MURB_SYM_BLOCK > floats) { *what = "j-side output outside the buffer"; return 1; }
            for (size_t c = it.joff; c < it.joff + MURB_SYM_BLOCK; ++c) { if (writers[c]++) { *what = "two writers for a j-row cell"; return 1; } if (!in_table[c]) { *what = "an item writes outside the row tables"; return 1; } }
        }
    }
    return 0;
}

int main()
{
    long plans = 0;
    const unsigned long sizes[] = {1, 250, 1025, 2049, 9001, 30000, 60001};
    for (unsigned long n : sizes)
        for (int W : {1, 2, 3, 4, 8}) {
            if ((unsigned long)W > n) continue;
            const SymFill fill = sym_fill(n, W);
            for (int split : {1, 4, 16})
                for (int waves : {4, 8}) {
                    if (MURB_SYM_BLOCK / split < 16 * waves) continue;
                    for (int taper : {0, 40})
                        for (int variant = 0; variant < 6; ++variant) {
                            const bool diag_tri = variant & 1, exchange = W > 1 || variant >= 4, xcd = variant == 2;
                            const int tri_div = (exchange && variant == 5) ? 4 : 1;
                            const size_t budget = (!exchange && variant == 3) ? (size_t)40 * MURB_SYM_BLOCK : 0;
                            for (int r = 0; r < W; r += (W > 4 ? 3 : 1)) {
                                SymHostLayout L;
                                plan_sym_layout(W, r, fill, split, waves, taper, diag_tri, exchange, 1, 50, xcd, budget, L, tri_div);
                                ++plans;
                                CHECK(!L.items.empty() && L.own >= 0 && (size_t)L.own <= L.items.size() && L.t1 >= 0 && L.t1 <= L.own, "item counts");
                                const char* what = "";
                                if (exchange) {
                                    CHECK(L.passes.size() == 1, "the exchange pipeline has one pass");
                                    CHECK(!check_set(L.items, 0, (size_t)L.own, L.table_tri, L.floats_tri, waves, &what), "triangle set: %s", what);
                                    CHECK(!check_set(L.items, (size_t)L.own, L.items.size(), L.table_main, L.floats_main, waves, &what), "main set: %s", what);
                                    CHECK((int)L.table_tri.size() == fill.tb, "the triangle's table has %zu entries for %d blocks", L.table_tri.size(), fill.tb);
                                } else {
                                    size_t next = 0;
                                    CHECK(budget || L.passes.size() == 1, "passes without a budget");
                                    for (const SymPass& ps : L.passes) {
                                        CHECK((size_t)ps.item_first == next && ps.item_count > 0 && ps.floats <= L.floats_main, "passes do not partition the items");
                                        next += (size_t)ps.item_count;
                                        const std::vector<MurbSymBlockRows> table(L.table_main.begin() + ps.table_first, L.table_main.begin() + ps.table_first + ps.table_count);
                                        CHECK(!check_set(L.items, (size_t)ps.item_first, next, table, ps.floats, waves, &what), "pass: %s", what);
                                    }
                                    CHECK(next == L.items.size(), "passes do not partition the items");
                                    if (L.passes.size() == 1) CHECK((int)L.table_main.size() == fill.tb, "one GPU: a table entry per block");
                                }
                            }
                        }
                }
        }
    std::printf("ok %ld\n", plans);
    return 0;
}
