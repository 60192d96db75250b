"""Lab: host cost of ENQUEUEING one step of the one-process-several-shards path (murbhip_create_sharded, what
`--im hip+tile+multi` drives from the reference's single host thread, main.cpp:348-354).

W shards of one process time-share device 0; murbhip_steps(k) returns when everything is enqueued, so the wall time until
it returns is host work only as long as the GPU has not fallen so far behind that a queue fills (k is kept small).
Reported per W: host us per step (and per shard), the wall time per step of the free-running loop and of the murb loop
(one device sync per iteration: enqueue and execution are then serial for the first shard's first kernel at least).

    python tools/host_enqueue.py [--bodies 32768,200000] [--worlds 1,2,4,8] [--exchange copy,rccl]
`rccl` here is the stand-in library (tests/helpers/rccl_mock.cpp): its host cost is not RCCL's, the call pattern is.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "helpers", "_build", "librccl_mock.so")
os.environ.setdefault("MURBHIP_RCCL_LIBRARY", MOCK)
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", default="32768,200000")
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--exchange", default="copy,rccl")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--opts", default="")
args = ap.parse_args()

DT = 3600.0
for n in [int(x) for x in args.bodies.split(",")]:
    s = murbhip.init_bodies(n, "galaxy")
    for ex in args.exchange.split(","):
        for w in [int(x) for x in args.worlds.split(",")]:
            if w == 1 and ex != "copy":
                continue
            sim = murbhip.Simulation(n, soft=2e8) if w == 1 else murbhip.Simulation(n, soft=2e8, devices=[0] * w, exchange=ex)
            if n < 100000:
                sim.set_option("variant", 8)      # the plan of the benchmark sizes, on a problem the GPU finishes quickly
            for kv in filter(None, args.opts.split(",")):
                k_, v_ = kv.split("=")
                sim.set_option(k_, int(v_))
            sim.upload(s)
            sim.steps(DT, 30); sim.sync()
            host, free, murb = [], [], []
            for _ in range(5):
                t0 = time.perf_counter(); sim.steps(DT, args.steps); t1 = time.perf_counter(); sim.sync(); t2 = time.perf_counter()
                host.append((t1 - t0) / args.steps * 1e6); free.append((t2 - t0) / args.steps * 1e6)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    sim.step(DT); sim.sync()
                murb.append((time.perf_counter() - t0) / args.steps * 1e6)
            h, f, m = min(host), min(free), min(murb)
            print(f"N={n} W={w} exchange={ex if w > 1 else '-'} variant={int(sim.info('variant'))} split={int(sim.info('jsplit'))}: "
                  f"host enqueue {h:8.1f} us/step = {h / w:6.1f} us/step/shard | free-running {f:8.1f} us/step | "
                  f"sync each iteration {m:8.1f} us/step", flush=True)
            sim.close()
