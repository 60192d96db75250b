"""Proxy for multi-GPU efficiency on ONE GPU: W shards time-share the device, so the step time of the
sharded context divided by the single-context step time is the work inflation of the partition
(padding, per-rank row sums, tail of each rank's item list) — everything except wire latency."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import numpy as np, murbhip

def step_ms(n, shards, steps, variant=0, overlap=1):
    s = murbhip.init_bodies(n, "galaxy")
    kw = {} if shards == 1 else {"devices": [0] * shards}
    with murbhip.Simulation(n, **kw) as sim:
        sim.set_option("variant", variant); sim.set_option("overlap", overlap)
        sim.upload(s); sim.steps(3600.0, 2); sim.sync()
        t0 = time.perf_counter(); sim.steps(3600.0, steps); sim.sync()
        return (time.perf_counter() - t0) / steps * 1e3, sim.info("variant")

if __name__ == "__main__":
    for n, steps in ((200000, 20), (1000000, 3)):
        base, _ = step_ms(n, 1, steps)
        print(f"N={n}: single {base:.3f} ms/step")
        for w in (2, 4, 8):
            for variant in (0, 1):
                ms, v = step_ms(n, w, steps, variant)
                print(f"   W={w} variant={int(v)}: {ms:.3f} ms/step for all shards on one GPU  -> inflation x{ms/base:.3f}"
                      f"  (ideal per-rank time {ms/w:.3f} ms)", flush=True)
