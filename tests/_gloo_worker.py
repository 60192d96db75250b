"""Worker of tests/test_sharding_gloo.py: one rank of the body-range partition + per-step position
all-gather, on CPU over gloo.  The force on the rank's own slice comes from the oracle (this is a
test), everything that defines WHO owns WHAT and WHERE it sits in the replicated buffer comes from
the product library's host entry points (murbhip_partition / slice_slots / slot_of_body), i.e. the
same layout contract libmurbhip.so applies on the GPUs with RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import murbhip  # noqa: E402
import oracle as O  # noqa: E402


def main():
    n, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    soft, dt = np.float32(2e8), np.float32(3600.0)
    s = murbhip.init_bodies(n, "galaxy")              # every rank builds the full state, like the MPI path
    first, count = murbhip.partition(n, world, rank)
    slots = murbhip.slice_slots(n, world)

    # replicated buffer in slot order: 4 floats per slot (x, y, z, m); padding slots stay 0
    rec = np.zeros((world * slots, 4), np.float32)
    for i in range(n):
        rec[murbhip.slot_of_body(n, world, i)] = (s["qx"][i], s["qy"][i], s["qz"][i], s["m"][i])
    mine = {k: s[k][first:first + count].copy() for k in ("qx", "qy", "qz", "vx", "vy", "vz")}
    lo = rank * slots

    for _ in range(steps):
        view = {"qx": np.ascontiguousarray(rec[:, 0]), "qy": np.ascontiguousarray(rec[:, 1]),
                "qz": np.ascontiguousarray(rec[:, 2]), "m": np.ascontiguousarray(rec[:, 3])}
        acc = O.accel_slice_f32(view, lo, lo + count, soft)      # own i slice against ALL slots
        O.integrate(mine, acc, dt)
        send = np.zeros((slots, 4), np.float32)
        send[:count, 0], send[:count, 1], send[:count, 2] = mine["qx"], mine["qy"], mine["qz"]
        send[:count, 3] = s["m"][first:first + count]
        gathered = torch.zeros(world * slots, 4)
        dist.all_gather_into_tensor(gathered, torch.from_numpy(send))   # equal counts: plain all-gather
        rec = gathered.numpy().copy()

    # rank 0 collects velocities too and writes the global state
    vel = torch.zeros(slots, 3)
    vel[:count] = torch.from_numpy(np.stack([mine["vx"], mine["vy"], mine["vz"]], 1))
    allv = [torch.zeros(slots, 3) for _ in range(world)]
    dist.all_gather(allv, vel)
    if rank == 0:
        res = {k: np.zeros(n, np.float32) for k in ("qx", "qy", "qz", "vx", "vy", "vz")}
        for r in range(world):
            f, c = murbhip.partition(n, world, r)
            res["qx"][f:f + c], res["qy"][f:f + c], res["qz"][f:f + c] = (rec[r * slots:r * slots + c, j] for j in range(3))
            v = allv[r].numpy()
            res["vx"][f:f + c], res["vy"][f:f + c], res["vz"][f:f + c] = v[:c, 0], v[:c, 1], v[:c, 2]
        np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
