"""Lab: XCD-interleaved item order against plain j-major order of the pair-symmetric kernel.
   python tools/xcd_ab.py ab [N]      interleaved timing in one process (force kernel ms from HIP events)
   python tools/xcd_ab.py 0|1 [N]     20 steps with that order (run under rocprofv3 --pmc FETCH_SIZE)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip
mode = sys.argv[1] if len(sys.argv) > 1 else "ab"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
s = murbhip.init_bodies(n, "galaxy")
with murbhip.Simulation(n) as sim:
    sim.upload(s)
    if mode in ("0", "1"):
        sim.set_option("xcd_order", int(mode)); sim.steps(3600.0, 20); sim.sync()
    else:
        sim.steps(3600.0, 30); sim.sync()
        for rnd in range(4):
            for order in (1, 0):
                sim.set_option("xcd_order", order); sim.steps(3600.0, 3); sim.sync()
                sim.set_option("profile", 0); sim.set_option("profile", 1)
                t0 = time.perf_counter(); sim.steps(3600.0, 20); sim.sync(); wall = (time.perf_counter() - t0) / 20 * 1e3
                print(f"N={n} xcd_order={order}: force {sim.info('force_ms_avg'):.4f} ms  step {wall:.4f} ms", flush=True)
