"""Times every force-kernel variant / jsplit on the GPU through the C ABI (not a product path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, murbhip, oracle as O

def run(n, variant, jsplit, steps=10, waves=0):
    s = O.init_bodies(n, "galaxy")
    with murbhip.Simulation(n) as sim:
        sim.set_option("variant", variant); sim.set_option("jsplit", jsplit); sim.set_option("sym_waves", waves)
        sim.upload(s); sim.steps(3600.0, max(5, int(0.1 / (n * n / 6e12)))); sim.sync()   # clock ramp
        sim.upload(s); sim.steps(3600.0, 2); sim.sync()
        sim.set_option("profile", 1)
        t0 = time.perf_counter(); sim.steps(3600.0, steps); sim.sync(); t1 = time.perf_counter()
        kms = sim.info("force_ms_avg"); js = sim.info("jsplit"); ipl = sim.info("interactions_per_launch")
    wall = (t1 - t0) / steps
    print(f"N={n:7d} variant={variant} waves={waves} jsplit={int(js):2d}  force {kms:8.3f} ms  step(wall) {wall*1e3:8.3f} ms  "
          f"{ipl/(kms*1e-3)/1e12:6.3f} T inter/s (kernel)  {n*n/wall/1e12:6.3f} T inter/s (wall)  "
          f"{20*ipl/(kms*1e-3)/157.3e12*100:5.1f}% of 157.3 TF", flush=True)

if __name__ == "__main__":
    ns = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [30000, 200000]
    for n in ns:
        for v in (1, 8):
            for w in ((4, 8) if v == 8 else (0,)):
                for js in ([0, 2, 4, 8, 16] if v == 8 else [0]):
                    run(n, v, js, steps=200 if n <= 50000 else 20, waves=w)
        run(n, 0, 0, steps=200 if n <= 50000 else 20)   # what auto picks
