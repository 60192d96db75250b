"""Lab: how long the initial conditions take — on the host (host/core/Bodies.cpp, the reference's way) and upload, against
murbhip_init_bodies on the device.    python tools/init_time.py [--bodies 30000,200000,1000000]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", default="30000,200000,1000000")
args = ap.parse_args()
for n in [int(x) for x in args.bodies.split(",")]:
    for scheme in ("galaxy", "random"):
        with murbhip.Simulation(n, soft=2e8) as sim:
            t0 = time.perf_counter(); s = murbhip.init_bodies(n, scheme); t1 = time.perf_counter()
            sim.upload(s); sim.sync(); t2 = time.perf_counter()
            sim.init_bodies(scheme, 0); sim.sync()          # first call: allocations
            t3 = time.perf_counter(); sim.init_bodies(scheme, 0); sim.sync(); t4 = time.perf_counter()
        print(f"N={n} {scheme}: host init {1e3 * (t1 - t0):8.2f} ms + upload {1e3 * (t2 - t1):7.2f} ms | on the device {1e3 * (t4 - t3):7.2f} ms", flush=True)
