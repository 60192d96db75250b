"""Lab: a tracked iteration (energy + moments + step, what `--im hip+tracking` runs) of small problems on one GPU under the two plan
rules — "fuse_integrate" 1 (default: one-sided kernel with the state update in its tail up to 4 blocks and at 6) and 0 (the
pair-symmetric plan from 3 blocks up, potential out of the force evaluation).    python tools/tracked_small.py"""
import sys, time
sys.path.insert(0, "nbody-eurohpc_amd")
import murbhip
for n in (1000, 2048, 2049, 3000, 4096, 6000):
    s = murbhip.init_bodies(n, "galaxy")
    out = []
    for fuse in (0, 1, 0, 1):
        with murbhip.Simulation(n, soft=2e8) as sim:
            sim.set_option("fuse_integrate", fuse)
            sim.upload(s)
            sim.steps(3600.0, 200); sim.sync()
            k = 2000
            t0 = time.perf_counter()
            for _ in range(k):
                sim.energy(); sim.moments(); sim.step(3600.0)
            sim.sync()
            out.append(f"fuse={fuse} variant {int(sim.info('variant'))}: {(time.perf_counter() - t0) * 1e6 / k:.1f} us")
    print(f"N={n} tracked iteration: " + " | ".join(out), flush=True)
