"""Worker of tests/test_sharding_gloo.py::test_half_ring_...: the multi-GPU pair-symmetric schedule on CPU
over gloo.  WHO evaluates WHICH block pair, WHERE a body sits and which exchange steps run come from the
product library's host entry points (murbhip_schedule_items / partition / slice_slots / slot_of_body);
the arithmetic of one block pair is numpy fp64 (this is a test), both directions applied — like the HIP
kernel.  One step = evaluate own items -> all-reduce of the partial accelerations (gloo has no
reduce-scatter; a rank keeps only its own slice) -> integrate own slice -> all-gather positions."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip  # noqa: E402

G, SOFT, DT = 6.67384e-11, 2e8, 3600.0


def pair_block(pos, gm, ia, ib, ja, jb, acc, both):
    d = pos[None, ja:jb, :] - pos[ia:ib, None, :]                       # (ni, nj, 3) = q_j - q_i
    inv3 = (np.einsum("ijk,ijk->ij", d, d) + SOFT * SOFT) ** -1.5
    acc[ia:ib] += np.einsum("ij,ijk->ik", inv3 * gm[None, ja:jb], d)
    if both:
        acc[ja:jb] -= np.einsum("ij,ijk->jk", inv3 * gm[ia:ib, None], d)


def main():
    n, steps, split, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    s = murbhip.init_bodies(n, "galaxy")
    slots = murbhip.slice_slots(n, world)
    first, count = murbhip.partition(n, world, rank)
    pos = np.zeros((world * slots, 3))
    gm = np.zeros(world * slots)
    vel = np.zeros((world * slots, 3))
    slot = np.array([murbhip.slot_of_body(n, world, i) for i in range(n)])
    pos[slot] = np.stack([s["qx"], s["qy"], s["qz"]], 1)
    vel[slot] = np.stack([s["vx"], s["vy"], s["vz"]], 1)
    gm[slot] = G * s["m"].astype(np.float64)
    items, own = murbhip.schedule_items(n, world, rank, split)
    sub = 1024 // split
    lo, hi = rank * slots, rank * slots + count
    for _ in range(steps):
        acc = np.zeros_like(pos)
        for i, j in items.tolist():
            diagonal = (i // split) == j
            pair_block(pos, gm, i * sub, (i + 1) * sub, j * 1024, (j + 1) * 1024, acc, both=not diagonal)
        t = torch.from_numpy(acc)
        dist.all_reduce(t)                                   # reduce-scatter in the product: only [lo, hi) is used
        a = t.numpy()[lo:hi]
        pos[lo:hi] += (vel[lo:hi] + a * DT * 0.5) * DT
        vel[lo:hi] += a * DT
        send = torch.from_numpy(pos[rank * slots:(rank + 1) * slots].copy())
        gathered = torch.zeros(world * slots, 3, dtype=torch.float64)
        dist.all_gather_into_tensor(gathered, send)
        pos = gathered.numpy().copy()
    if rank == 0:
        np.save(out, pos[slot])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
