import sys, time
sys.path.insert(0, "nbody-eurohpc_amd")
import murbhip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
with murbhip.Simulation(n, soft=2e8) as sim:
    sim.init_bodies("galaxy", 0)
    sim.sync()
    t_start = time.perf_counter()
    out = []
    while time.perf_counter() - t_start < 8.0:
        t0 = time.perf_counter(); sim.steps(3600.0, 10); sim.sync(); t1 = time.perf_counter()
        out.append((t0 - t_start, (t1 - t0) / 10 * 1e3))
    for i, (t, ms) in enumerate(out):
        if i < 10 or i % 10 == 0: print(f"t={t:6.3f} s  {ms:.4f} ms/step")
