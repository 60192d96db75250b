"""Lab: where one rank's step goes — the timing spans ("profile" 2) of ONE real rank process of W without peers and wire
(stand-in collectives in solo mode, like tools/solo_rank.py).    python tools/solo_spans.py [--bodies 200000] [--worlds 8] [--opts k=v,..]"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "helpers", "_build", "librccl_mock.so")
ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=200000)
ap.add_argument("--worlds", default="8")
ap.add_argument("--child", type=int, default=0)
ap.add_argument("--opts", default="")
ap.add_argument("--steps", type=int, default=200)
args = ap.parse_args()
if not args.child:
    env = dict(os.environ, MURBHIP_RCCL_LIBRARY=MOCK, MURB_MOCK_SOLO="1")
    for w in [int(x) for x in args.worlds.split(",")]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--bodies", str(args.bodies), "--child", str(w), "--opts", args.opts,
                        "--steps", str(args.steps)], env=env, check=True)
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip  # noqa: E402
n, w = args.bodies, args.child
s = murbhip.init_bodies(n, "galaxy")
sim = murbhip.Simulation(n, soft=2e8, device=0, rank=0, world=w, uid=murbhip.unique_id())
for kv in filter(None, args.opts.split(",")):
    k, v = kv.split("=")
    sim.set_option(k, int(v))
sim.upload(s)
sim.steps(3600.0, 300); sim.sync()
t0 = time.perf_counter(); sim.steps(3600.0, args.steps); sim.sync(); plain = (time.perf_counter() - t0) * 1e3 / args.steps
sim.set_option("profile", 2)
t0 = time.perf_counter(); sim.steps(3600.0, args.steps); sim.sync(); prof = (time.perf_counter() - t0) * 1e3 / args.steps
g = sim.info
print(f"N={n} W={w} opts[{args.opts}] variant {int(g('variant'))} split {int(g('jsplit'))}: {plain:.4f} ms/step ({prof:.4f} with spans) | "
      f"T1 {g('span_tri1_ms_avg'):.4f} R {g('span_rect_ms_avg'):.4f} T2 {g('span_tri2_ms_avg'):.4f} | step on the compute stream {g('span_step_ms_avg'):.4f} | "
      f"RS {g('span_reduce_scatter_ms_avg'):.4f} AG {g('span_all_gather_ms_avg'):.4f} wait_g {g('span_wait_gather_ms_avg'):.4f} wait_r {g('span_wait_reduce_ms_avg'):.4f}", flush=True)
sim.close()
