"""Lab: how long does a small kernel on a second, high-priority stream wait while the force kernels keep every CU busy —
the situation of a collective's kernel during a multi-GPU step — and what does "cu_reserve" change?
    python tools/comm_latency.py [--bodies 200000] [--shards 1]
For cu_reserve in (0, 8, 16): queue ~0.2 s of steps, then time 40 launches of a one-workgroup and of a 64-workgroup
torch kernel on a priority stream (submit -> complete on the host clock, and start event -> end event on the GPU), and the
force rate of the same configuration.  An idle-GPU line gives the floor."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=200000)
ap.add_argument("--shards", type=int, default=1)
ap.add_argument("--solo", type=int, default=-1)
args = ap.parse_args()
n = args.bodies
s = murbhip.init_bodies(n, "galaxy")
sim = murbhip.Simulation(n, soft=2e8, devices=[0] * args.shards) if args.shards > 1 else murbhip.Simulation(n, soft=2e8)
if args.solo >= 0:
    sim.set_option("solo_shard", args.solo)
sim.upload(s)
hp = torch.cuda.Stream(priority=-1)
small = torch.zeros(256, device="cuda")
big = torch.zeros(64 * 256 * 4, device="cuda")          # ~64 workgroups of an elementwise kernel


def probe(x, busy):
    host, dev = [], []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        with torch.cuda.stream(hp):
            e0.record()
            x.add_(1.0)
            e1.record()
        e1.synchronize()
        host.append((time.perf_counter() - t0) * 1e6)
        dev.append(e0.elapsed_time(e1) * 1e3)
        time.sleep(0.001)
    return np.median(host), np.percentile(host, 90), np.median(dev), np.percentile(dev, 90)


probe(small, False); probe(big, False)
print(f"N={n} shards={args.shards} solo={args.solo}")
print("idle GPU:        1 WG host %.0f / %.0f us (p50 / p90), device %.0f / %.0f us;  64 WGs host %.0f / %.0f, device %.0f / %.0f"
      % (probe(small, False) + probe(big, False)))
per_step = n * n / 6.5e12 / (args.shards if args.solo >= 0 else 1)
for reserve in (0, 8, 16):
    sim.set_option("cu_reserve", reserve)
    sim.steps(3600.0, 20); sim.sync()
    sim.set_option("profile", 1)
    k = max(20, int(0.6 / per_step))
    t0 = time.perf_counter()
    sim.steps(3600.0, k)                 # queued: the GPU is busy for ~0.6 s from here on
    time.sleep(0.05)
    a = probe(small, True)
    b = probe(big, True)
    sim.sync()
    wall = (time.perf_counter() - t0) * 1e3 / k
    print("cu_reserve %2d:   1 WG host %.0f / %.0f us, device %.0f / %.0f us;  64 WGs host %.0f / %.0f, device %.0f / %.0f;  force %.4f ms/step (wall %.4f incl. the probes)"
          % ((reserve,) + a + b + (sim.info("force_ms_total") / k, wall)))
    sim.set_option("profile", 0)
sim.close()
