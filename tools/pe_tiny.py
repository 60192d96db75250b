import sys, numpy as np
sys.path.insert(0, "nbody-eurohpc_amd"); sys.path.insert(0, "oracle")
import murbhip, oracle as O
for n, shards, opts in ((132, 8, dict(variant=8, jsplit=8, taper=0)), (158, 5, dict(variant=8, sym_waves=8, taper=100, tri_div=8, tri_first_pct=75, sym_pass_mb=1)), (300, 8, dict(variant=0)), (300, 8, dict(variant=8)), (300, 1, dict(variant=8)), (2500, 3, dict(variant=8))):
    s = O.init_bodies(n, "galaxy")
    ke, pe = O.energy_f64(s, np.float32(2e8))
    for sweep in (0, 1):
        with murbhip.Simulation(n, soft=2e8, devices=[0] * shards) if shards > 1 else murbhip.Simulation(n, soft=2e8) as sim:
            for k, v in opts.items(): sim.set_option(k, v)
            sim.set_option("energy_sweep", sweep)
            sim.upload(s)
            k1, p1 = sim.energy()
            print(n, shards, opts, "sweep" if sweep else "fused", "variant", int(sim.info("variant")), "PE rel err %.3e  KE rel err %.1e" % ((p1 - pe) / abs(pe), (k1 - ke) / max(abs(ke), 1e-300)), flush=True)
