"""Lab: the step time of ONE real rank process of a W-rank job without its peers and without the wire — the one-process-
per-GPU entry point (murbhip_create_rank) with its own three streams, collectives from the stand-in library in its
"solo" mode (tests/helpers/rccl_mock.cpp: nobody is waited for, data stays local).  What a rank computes per step
against the single-GPU step = the scaling a perfect interconnect would give.
    python tools/solo_rank.py [--bodies 200000] [--worlds 2,4,8]
(Several shards of ONE process on one GPU — tools/rank_timeline.py — share the process's few hardware queues, which makes
their timing depend on how the streams happen to be mapped; a rank process of its own does not have that problem.)"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "helpers", "_build", "librccl_mock.so")

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=200000)
ap.add_argument("--worlds", default="2,4,8")
ap.add_argument("--child", type=int, default=0)
ap.add_argument("--opts", default="")
args = ap.parse_args()

if not args.child:
    env = dict(os.environ, MURBHIP_RCCL_LIBRARY=MOCK, MURB_MOCK_SOLO="1")
    for w in [1] + [int(x) for x in args.worlds.split(",")]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--bodies", str(args.bodies), "--child", str(w), "--opts", args.opts],
                       env=env, check=True)
    sys.exit(0)

sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip  # noqa: E402

n, w = args.bodies, args.child
s = murbhip.init_bodies(n, "galaxy")
sim = murbhip.Simulation(n, soft=2e8) if w == 1 else murbhip.Simulation(n, soft=2e8, device=0, rank=0, world=w, uid=murbhip.unique_id())
for kv in filter(None, args.opts.split(",")):
    k, v = kv.split("=")
    sim.set_option(k, int(v))
sim.upload(s)
per = n * n / 6e12 / w
sim.steps(3600.0, max(10, int(0.3 / per))); sim.sync()
best = 1e9
for _ in range(3):
    k = max(10, int(0.2 / per))
    t0 = time.perf_counter(); sim.steps(3600.0, k); sim.sync()
    best = min(best, (time.perf_counter() - t0) * 1e3 / k)
print(f"N={n} W={w}: {'single GPU' if w == 1 else 'rank 0 alone'} {best:.4f} ms/step  (variant {int(sim.info('variant'))}, split {int(sim.info('jsplit'))}, "
      f"{sim.info('device_bytes') / 1e6:.0f} MB)", flush=True)
sim.close()
