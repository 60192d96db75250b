import sys, time
sys.path.insert(0, "nbody-eurohpc_amd")
import murbhip
def rate(n, opts):
    s = murbhip.init_bodies(n, "galaxy")
    with murbhip.Simulation(n, soft=2e8) as sim:
        for k, v in opts.items(): sim.set_option(k, v)
        sim.upload(s)
        k = max(20, int(0.3 / (n * n / 6e12)))
        sim.steps(3600.0, k); sim.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); sim.steps(3600.0, k); sim.sync(); best = min(best, (time.perf_counter() - t0) / k)
        return n * n / best / 1e12, int(sim.info("sym_waves")), int(sim.info("jsplit")), int(sim.info("taper"))
small = dict(sym_waves=8, jsplit=4, taper=30, diag_tri=1)
large = dict(sym_waves=4, jsplit=0, taper=5, diag_tri=0)
for n in (16000, 24000, 30000, 36000, 40000, 44000, 46000, 50000, 56000, 64000, 80000, 100000, 140000):
    a = rate(n, {}); b = rate(n, small); c = rate(n, dict(large, jsplit=4)); d = rate(n, dict(large, jsplit=2)); e = rate(n, dict(small, jsplit=2))
    print(f"N={n}: default {a[0]:.3f} T/s (waves {a[1]} split {a[2]} taper {a[3]}) | small-plan {b[0]:.3f} | 4w split4 t5 {c[0]:.3f} | 4w split2 t5 {d[0]:.3f} | 8w split2 t30 tri {e[0]:.3f}", flush=True)
