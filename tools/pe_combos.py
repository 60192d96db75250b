import sys, numpy as np
sys.path.insert(0, "nbody-eurohpc_amd"); sys.path.insert(0, "oracle")
import murbhip, oracle as O
for scheme, n in (("galaxy", 12001), ("random", 6151)):
    s = O.init_bodies(n, scheme)
    ke, pe = O.energy_f64(s, np.float32(2e8))
    for (t, d, r, w, j) in [(0, 0, 0, 4, 4), (50, 0, 0, 4, 2), (0, 1, 0, 4, 1), (0, 1, 0, 8, 8), (0, 0, 1, 4, 2), (0, 0, 1, 8, 8), (100, 1, 1, 4, 1), (40, 1, 1, 8, 4), (30, 1, 1, 4, 16), (5, 1, 1, 4, 1), (60, 1, 0, 8, 2), (0, 0, 1, 4, 1)]:
        out = []
        for sweep in (0, 1):
            with murbhip.Simulation(n, soft=2e8) as sim:
                sim.set_option("variant", 8)
                for k, v in dict(taper=t, diag_tri=d, sym_red=r, sym_waves=w, jsplit=j, energy_sweep=sweep).items():
                    sim.set_option(k, v)
                sim.upload(s)
                k1, p1 = sim.energy()
                out.append((p1 - pe) / abs(pe))
        print(scheme, n, dict(taper=t, diag_tri=d, red=r, waves=w, split=j), "fused %.2e sweep %.2e" % tuple(out), flush=True)
