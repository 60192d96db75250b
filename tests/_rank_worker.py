"""Worker of tests/test_rank_mode_mock.py: ONE rank of a one-process-per-GPU run (murbhip_create_rank), all
ranks on GPU 0, collectives through tests/helpers/rccl_mock.cpp (MURBHIP_RCCL_LIBRARY in the environment).
    python _rank_worker.py RANK WORLD UIDHEX N STEPS VARIANT OVERLAP JSPLIT INTEGRATOR OUT.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip  # noqa: E402

rank, world = int(sys.argv[1]), int(sys.argv[2])
uid = bytes.fromhex(sys.argv[3])
n, steps, variant, overlap, jsplit, integrator = (int(x) for x in sys.argv[4:10])
out = sys.argv[10]
s = murbhip.init_bodies(n, "galaxy")
with murbhip.Simulation(n, soft=2e8, device=0, rank=rank, world=world, uid=uid) as sim:
    sim.set_option("variant", variant)
    sim.set_option("overlap", overlap)
    sim.set_option("jsplit", jsplit)
    sim.set_option("integrator", integrator)
    warmup_ms = 0.0
    for kv in filter(None, os.environ.get("MURB_TEST_OPTIONS", "").split(",")):   # further library options: "key=value,key=value"
        key, value = kv.split("=")
        if key == "warmup":           # not an option: murbhip_warmup for that many ms after the upload (collectives inside)
            warmup_ms = float(value)
        else:
            sim.set_option(key, int(value))
    sim.upload(s)
    if warmup_ms:
        sim.warmup(warmup_ms)
    sim.compute_acc()
    sim.sync()
    acc0 = sim.acc()                 # own bodies only; zeros elsewhere
    sim.steps(3600.0, steps)
    sim.sync()
    st = sim.state()                 # all positions, own velocities
    ke, pe = sim.energy()            # own bodies' share
    first, count = murbhip.partition(n, world, rank)
    np.savez(out, first=first, count=count, ax=acc0[0], ay=acc0[1], az=acc0[2], ke=ke, pe=pe,
             used_variant=int(sim.info("variant")), **st)
