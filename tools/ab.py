"""Lab: interleaved A/B of library options through the C ABI, in ONE process (the clock ramps for ~100 ms after an idle
spell and boxes differ by +-4 %, so only interleaved rounds in one process compare anything).

    python tools/ab.py --bodies 30000 --steps 200 --rounds 6  base:  tri:diag_tri=1  red:sym_red=1  "both:diag_tri=1,sym_red=1,taper=40"
    python tools/ab.py --bodies 200000 --shards 8 --solo 0 ...      # one rank of 8, isolated ("solo_shard")

Each configuration is a context of its own with the given options (name:key=value,key=value); a round runs `steps`
steps of every configuration in turn.  Reported per configuration: force-kernel time per step (HIP events on the
library's stream, all force launches of a step added up), wall time per step, both as mean over rounds and relative to
the first configuration; plus the largest relative difference of the accelerations against the first configuration.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=200000)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--shards", type=int, default=1)
ap.add_argument("--solo", type=int, default=-1)
ap.add_argument("configs", nargs="+")
args = ap.parse_args()

n = args.bodies
s = murbhip.init_bodies(n, "galaxy")
sims, names = [], []
for spec in args.configs:
    name, _, opts = spec.partition(":")
    sim = murbhip.Simulation(n, soft=2e8, devices=[0] * args.shards) if args.shards > 1 else murbhip.Simulation(n, soft=2e8)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        sim.set_option(k, int(v))
    if args.solo >= 0:
        sim.set_option("solo_shard", args.solo)
    sim.upload(s)
    sims.append(sim)
    names.append(name)

acc0 = None
for name, sim in zip(names, sims):
    if args.solo < 0:
        sim.compute_acc()
        sim.sync()
        a = np.stack(sim.acc()).astype(np.float64)
        if acc0 is None:
            acc0 = a
        else:
            err = np.sqrt(((a - acc0) ** 2).sum(0)) / np.maximum(np.sqrt((acc0 ** 2).sum(0)), 1e-300)
            print(f"  {name}: max rel acceleration difference vs {names[0]}: {err.max():.3e}")
    sim.steps(3600.0, max(10, int(0.2 / (n * n / 6e12 / max(args.shards if args.solo >= 0 else 1, 1)))))   # clock ramp + first-use allocations
    sim.sync()

force = np.zeros((args.rounds, len(sims)))
wall = np.zeros((args.rounds, len(sims)))
for r in range(args.rounds):
    for i, sim in enumerate(sims):
        sim.steps(3600.0, 3)
        sim.sync()
        sim.set_option("profile", 1)
        t0 = time.perf_counter()
        sim.steps(3600.0, args.steps)
        sim.sync()
        wall[r, i] = (time.perf_counter() - t0) * 1e3 / args.steps
        force[r, i] = sim.info("force_ms_total") / args.steps
        sim.set_option("profile", 0)

print(f"N={n} shards={args.shards} solo={args.solo} steps={args.steps} rounds={args.rounds}")
print(f"{'config':>14} {'force ms/step':>14} {'(min)':>9} {'vs first':>9} {'wall ms/step':>13} {'(min)':>9} {'vs first':>9}   info")
for i, (name, sim) in enumerate(zip(names, sims)):
    f, w = force[:, i].mean(), wall[:, i].mean()
    info = f"variant {int(sim.info('variant'))} jsplit {int(sim.info('jsplit'))} bytes {sim.info('device_bytes') / 1e6:.0f} MB"
    print(f"{name:>14} {f:14.4f} {force[:, i].min():9.4f} {force[:, 0].mean() / f:9.4f} {w:13.4f} {wall[:, i].min():9.4f} {wall[:, 0].mean() / w:9.4f}   {info}")
for sim in sims:
    sim.close()
