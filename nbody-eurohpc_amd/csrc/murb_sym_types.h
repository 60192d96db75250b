// Plain types shared by the host-side planner (murb_plan.h, murb_schedule.h: no HIP in them) and the pair-symmetric kernels
// (murb_kernels_sym.h).  Nothing here needs hipcc.
#ifndef MURB_SYM_TYPES_H_
#define MURB_SYM_TYPES_H_

#define MURB_SYM_BLOCK 1024                       /* body slots per block             */
#define MURB_SYM_PAIRS (MURB_SYM_BLOCK / 2)       /* 512 pairs                        */
#define MURB_SYM_STEPS (MURB_SYM_PAIRS / 64)      /* 8 pair-vectors per lane          */
#define MURB_SYM_R 4                              /* i bodies per group               */

// One workgroup's work: the bodies [i_slot0, i_slot0 + ngroups * WAVES * R) against j block J.
struct MurbSymItem {
    int i_slot0;          // first slot (global, multiple of WAVES * R) of the i range
    int ngroups;          // i groups of R bodies per WAVE
    int J;                // j block (staged in LDS)
    int flags;            // bit 0: nothing is written on the j side (plain diagonal item: full square, i side kept only; or
                          // the last triangular piece of a diagonal block); bit 1: diagonal item in its triangular form,
                          // bits 8-11 its first j step, bits 12-15 its first j step applied both ways (SymPiece, murb_schedule.h)
    unsigned long ioff;   // float offset (component 0) where the i-side sums of slot i_slot0 go
    unsigned long joff;   // float offset (component 0) where the j-side sums of block J's first slot go
};


// One entry per block a launch produced partial rows for: where its two matrices start (component 0), how many rows
// each has, and where the block's totals go in the output (out[(out_slice * 3 + c) * out_slice_slots + out_block *
// 1024 + slot]): the acceleration planes ax | ay | az of one GPU (out_slice = 0), or the reduce-scatter send buffer of
// a multi-GPU rank, one chunk of [3][slice] per destination slice.
#define MURB_SYM_MAX_RANKS 64
struct MurbSymBlockRows {
    unsigned long base_i, base_j;
    int ni, nj;
    int out_slice, out_block;
};

#endif
