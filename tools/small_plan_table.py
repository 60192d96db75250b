"""Lab: which plan of the pair-symmetric kernel is fastest for every block count T = ceil(N / 1024) below the 45 000-body switch, one
GPU.  Interleaved rounds of the candidates in one process (like tools/ab.py); prints one line per T and, at the end, the table
csrc/murbhip.hip keeps (kSmallPlanOfBlocks).    python tools/small_plan_table.py [--tmin 10] [--tmax 44]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip  # noqa: E402

CANDIDATES = [("8w s4 t30 tri", dict(sym_waves=8, jsplit=4, taper=30, diag_tri=1)),
              ("4w s4 t5", dict(sym_waves=4, jsplit=4, taper=5, diag_tri=0)),
              ("4w s8 t5", dict(sym_waves=4, jsplit=8, taper=5, diag_tri=0)),
              ("8w s8 t30 tri", dict(sym_waves=8, jsplit=8, taper=30, diag_tri=1)),
              ("4w s2 t5", dict(sym_waves=4, jsplit=2, taper=5, diag_tri=0)),
              ("4w s16 t5", dict(sym_waves=4, jsplit=16, taper=5, diag_tri=0)),
              ("one-sided", dict(variant=1))]
ap = argparse.ArgumentParser()
ap.add_argument("--tmin", type=int, default=10)
ap.add_argument("--tmax", type=int, default=44)
ap.add_argument("--fill", type=int, default=512, help="bodies in the last block")
ap.add_argument("--rounds", type=int, default=4)
args = ap.parse_args()
table = {}
for T in range(args.tmin, args.tmax + 1):
    n = (T - 1) * 1024 + args.fill
    s = murbhip.init_bodies(n, "galaxy")
    sims = []
    for _, opts in CANDIDATES:
        sim = murbhip.Simulation(n, soft=2e8)
        sim.set_option("variant", 8)
        for k, v in opts.items():   # ("variant" in opts overrides the line above)
            sim.set_option(k, v)
        sim.upload(s)
        sims.append(sim)
    steps = min(1000, max(100, int(0.04 / (n * n / 5e12))))   # capped: below ~5000 bodies a step is launch-bound (~15 us), not N^2
    for sim in sims:
        sim.steps(3600.0, steps); sim.sync()
    wall = np.zeros((args.rounds, len(sims)))
    for r in range(args.rounds):
        for i, sim in enumerate(sims):
            sim.steps(3600.0, 5); sim.sync()
            t0 = time.perf_counter(); sim.steps(3600.0, steps); sim.sync()
            wall[r, i] = (time.perf_counter() - t0) * 1e6 / steps
    best = wall.min(0)
    win = int(np.argmin(best))
    table[T] = win
    print(f"T={T:2d} N={n}: " + "  ".join(f"{name} {b:7.2f}" for (name, _), b in zip(CANDIDATES, best)) + f"  -> {CANDIDATES[win][0]}  ({best[0] / best[win]:.3f}x the first)", flush=True)
    for sim in sims:
        sim.close()
print("table:", ",".join(str(table[T]) for T in sorted(table)))
