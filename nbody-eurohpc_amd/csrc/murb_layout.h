// Device data layout of the body records (shared by host packing code and the kernels).
//
// The reference keeps eight separate fp32 arrays (dataSoA_t, reference
// src/common/core/Bodies.hpp:15-24) and its device twin stages qx/qy/qz/GM tiles through shared
// memory (SimulationNBodyCUDATileFullDevice.cu:93-106).  Here the four values the force loop
// needs are laid out for gfx950's packed-fp32 VALU (v_pk_fma_f32 works on an even/odd VGPR pair):
//
//   bodies are taken two at a time (slot 2p and 2p+1 form pair p); per pair two float4 records
//       A[p] = { x0, x1, y0, y1 }      B[p] = { z0, z1, gm0, gm1 }         (gm = G * m)
//   so one 16-byte load lands (x0,x1) and (y0,y1) in adjacent registers, ready for packed math.
//   Pairs are grouped in tiles of MURB_TILE_PAIRS; inside a tile all A records come first, then all
//   B records:   tile t = [ A[t*TP .. t*TP+TP) | B[t*TP .. t*TP+TP) ]   (TP*32 bytes = 8 KiB)
//   A tile is a linear 8 KiB copy into LDS, and a wave reading A[q..q+63] / B[q..q+63] touches
//   1 KiB of consecutive addresses (coalesced in HBM, bank-conflict free in LDS for ds_read_b128).
//
// Velocities use the same shape (A = {vx0,vx1,vy0,vy1}, B = {vz0,vz1,0,0}), local slice only.
#ifndef MURB_LAYOUT_H_
#define MURB_LAYOUT_H_

#define MURB_TILE_PAIRS 256                       /* pairs per layout tile                     */
#define MURB_TILE_BODIES (2 * MURB_TILE_PAIRS)    /* 512 body slots per tile                   */
#define MURB_TILE_F4 (2 * MURB_TILE_PAIRS)        /* float4 records per tile (A block+B block) */

// hipcc sees these helpers from device code too; a plain host compiler (g++) only needs `inline`
#if defined(__HIPCC__)
#define MURB_HD __host__ __device__ __forceinline__
#else
#define MURB_HD inline
#endif

/* float4 index of record A of pair p; record B sits MURB_TILE_PAIRS further. */
MURB_HD unsigned long murb_rec_a(unsigned long pair)
{
    return (pair / MURB_TILE_PAIRS) * MURB_TILE_F4 + (pair % MURB_TILE_PAIRS);
}

#define MURB_SLICE_ALIGN 1024   /* slots: one block of the pair-symmetric kernel = 2 layout tiles */

/* Round a body count up to whole slice units (also whole layout tiles). */
MURB_HD unsigned long murb_round_up_tile(unsigned long bodies)
{
    return ((bodies + MURB_SLICE_ALIGN - 1) / MURB_SLICE_ALIGN) * MURB_SLICE_ALIGN;
}

#endif
