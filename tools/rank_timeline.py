"""Isolated per-step time of ONE rank of a W-rank job, on one GPU: a sharded context in which only shard r
launches force work ("solo_shard"); the exchange steps still run (peer copies on the same device, so wire
latency is NOT included).  Shows what the launch tails and row sums of the multi-rank pipeline cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip

def solo_ms(n, w, steps, split=0, overlap=1, variant=0, r=0, waves=0):
    s = murbhip.init_bodies(n, "galaxy")
    with murbhip.Simulation(n, devices=[0] * w) as sim:
        sim.set_option("variant", variant); sim.set_option("jsplit", split); sim.set_option("overlap", overlap)
        sim.set_option("solo_shard", r); sim.set_option("sym_waves", waves)
        sim.upload(s); sim.steps(3600.0, max(3, int(0.15 * w / (n * n / 6e12)))); sim.sync()   # clock ramp
        t0 = time.perf_counter(); sim.steps(3600.0, steps); sim.sync()
        return (time.perf_counter() - t0) / steps * 1e3

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    with murbhip.Simulation(n) as one:
        one.upload(murbhip.init_bodies(n, "galaxy")); one.steps(3600.0, max(3, int(0.15 / (n * n / 6e12)))); one.sync()   # clock ramp
        t0 = time.perf_counter(); one.steps(3600.0, 20); one.sync(); base = (time.perf_counter() - t0) / 20 * 1e3
    print(f"N={n}: single GPU {base:.3f} ms/step")
    splits = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    overlaps = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2]
    waves_list = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
    for w in (2, 4, 8):
      for waves in waves_list:
        for split in splits:
            for overlap in overlaps:
                ms = solo_ms(n, w, 60 if n <= 300000 else 10, split, overlap, waves=waves)
                print(f"  W={w} waves={waves} split={split} overlap={overlap}: rank 0 alone {ms:.3f} ms/step -> speedup x{base/ms:.2f} of ideal x{w}", flush=True)
