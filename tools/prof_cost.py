import sys, time
sys.path.insert(0, "nbody-eurohpc_amd")
import murbhip
for n, k in ((30000, 2000), (200000, 100)):
    s = murbhip.init_bodies(n, "galaxy")
    sim = murbhip.Simulation(n, soft=2e8); sim.upload(s); sim.steps(3600.0, k); sim.sync()
    for rnd in range(3):
        out = []
        for prof in (0, 1, 2):
            sim.set_option("profile", prof); sim.steps(3600.0, 20); sim.sync()
            sim.set_option("profile", prof)
            t0 = time.perf_counter(); sim.steps(3600.0, k); sim.sync(); out.append((time.perf_counter() - t0) * 1e6 / k)
        print(f"N={n} round {rnd}: us/step with profile 0/1/2: {out[0]:.2f} {out[1]:.2f} {out[2]:.2f}", flush=True)
    sim.close()
