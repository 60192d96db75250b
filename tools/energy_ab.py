"""Lab: murbhip_energy from the force evaluation's own pair potential (default) against the separate potential sweep of rounds 1-2
("energy_sweep" 1) and an fp64 evaluation; and what a TRACKED iteration (energy + moments + step, `--im hip+tracking`) costs
either way.    python tools/energy_ab.py [--bodies 30000,200000] [--shards 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import murbhip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", default="30000,200000")
ap.add_argument("--shards", type=int, default=1)
ap.add_argument("--truth", type=int, default=60000, help="largest N for which the fp64 energy is evaluated on the host")
args = ap.parse_args()
for n in [int(x) for x in args.bodies.split(",")]:
    s = murbhip.init_bodies(n, "galaxy")
    truth = None
    if n <= args.truth:
        import oracle as O
        truth = O.energy_f64(s, np.float32(2e8))
    out = {}
    for name, sweep in (("fused", 0), ("sweep", 1)):
        kw = {"devices": [0] * args.shards} if args.shards > 1 else {}
        with murbhip.Simulation(n, soft=2e8, **kw) as sim:
            sim.set_option("energy_sweep", sweep)
            sim.upload(s)
            ke, pe = sim.energy()
            sim.steps(3600.0, 5); sim.sync()
            k = max(5, int(0.5 / (n * n / 6e12)))
            t0 = time.perf_counter()
            for _ in range(k):
                sim.energy(); sim.moments(); sim.step(3600.0)
            sim.sync()
            out[name] = (ke, pe, (time.perf_counter() - t0) * 1e3 / k)
            t0 = time.perf_counter(); sim.steps(3600.0, k); sim.sync(); plain = (time.perf_counter() - t0) * 1e3 / k
    (kf, pf, tf), (ks, ps, ts) = out["fused"], out["sweep"]
    line = f"N={n} shards={args.shards}: PE fused {pf:.9e} sweep {ps:.9e} rel diff {abs(pf - ps) / abs(ps):.2e}"
    if truth:
        line += f" | vs fp64: fused {abs(pf - truth[1]) / abs(truth[1]):.2e} sweep {abs(ps - truth[1]) / abs(truth[1]):.2e} (KE {abs(kf - truth[0]) / truth[0]:.1e})"
    print(line + f" | tracked iteration {tf:.3f} ms fused, {ts:.3f} ms with the sweep, plain step {plain:.3f} ms", flush=True)
