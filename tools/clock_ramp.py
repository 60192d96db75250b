"""Lab: per-chunk step time from a cold process start (does the GPU clock ramp, and for how long?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 50
s = murbhip.init_bodies(n, "galaxy")
with murbhip.Simulation(n) as sim:
    sim.upload(s); sim.sync()
    t_start = time.perf_counter()
    out = []
    for k in range(40):
        t0 = time.perf_counter(); sim.steps(3600.0, chunk); sim.sync(); t1 = time.perf_counter()
        out.append((t1 - t_start, (t1 - t0) / chunk * 1e3))
    print(f"N={n} chunk={chunk} steps: " + "  ".join(f"{t*1e3:.0f}ms:{ms:.4f}" for t, ms in out))
    time.sleep(1.0)   # idle, then again: does it fall back?
    t_start = time.perf_counter(); out = []
    for k in range(10):
        t0 = time.perf_counter(); sim.steps(3600.0, chunk); sim.sync(); t1 = time.perf_counter()
        out.append((t1 - t_start, (t1 - t0) / chunk * 1e3))
    print("after 1 s idle: " + "  ".join(f"{t*1e3:.0f}ms:{ms:.4f}" for t, ms in out))
