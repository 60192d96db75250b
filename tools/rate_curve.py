"""Lab: whole-step rate of the DEFAULT plan over the problem size, one GPU — where the curve dips is where a plan rule is off.
Per N: interactions/s, fraction of the fp32 vector peak at 20 flop per interaction (bench.py's whole-step figure), the plan.
    python tools/rate_curve.py [--sizes 1000,2049,...]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip  # noqa: E402

DEFAULT = ("500,1000,2048,2049,3000,4097,6000,8193,10000,12289,16000,20481,24000,28673,30000,33000,36865,40000,48000,56000,"
           "65537,80000,100000,120000,150000,200000,262145,300000,400000,524289,700000,1000000,1500000,2000000")
ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default=DEFAULT)
ap.add_argument("--scheme", default="galaxy")
ap.add_argument("--opts", default="", help="library options, key=value,key=value")
args = ap.parse_args()
PEAK = 256 * 256 * 2.4e9   # flop/s: 256 CUs x 256 fp32 flop per clock x 2.4 GHz
for n in [int(x) for x in args.sizes.split(",")]:
    with murbhip.Simulation(n, soft=2e8) as sim:
        for kv in filter(None, args.opts.split(",")):
            sim.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        sim.init_bodies(args.scheme, 0)
        k = min(1000, max(5, int(0.25 / (n * n / 6e12))))
        sim.steps(3600.0, k); sim.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); sim.steps(3600.0, k); sim.sync(); best = min(best, (time.perf_counter() - t0) / k)
        rate = float(n) * n / best
        print(f"N={n:8d}: {best * 1e3:9.4f} ms/step  {rate:.3e} inter/s  {20 * rate / PEAK * 100:5.1f} % of the fp32 peak | variant {int(sim.info('variant'))} "
              f"waves {int(sim.info('sym_waves'))} split {int(sim.info('jsplit'))} taper {int(sim.info('taper'))} passes {int(sim.info('sym_passes'))}", flush=True)
