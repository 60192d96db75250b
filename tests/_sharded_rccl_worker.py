"""Worker of tests/test_rank_mode_mock.py: ONE process driving several shards with exchange = RCCL
(murbhip_create_sharded(..., exchange = 1): ncclCommInitAll, then every shard's own host thread inside the library
drives its communicator — no grouped calls), all shards on GPU 0, collectives from the stand-in library
(MURBHIP_RCCL_LIBRARY), which makes the threads meet inside every call.  Prints "ok" when the sharded run matches the
single-GPU run.    python _sharded_rccl_worker.py SHARDS N VARIANT OVERLAP"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip  # noqa: E402

shards, n, variant, overlap = (int(x) for x in sys.argv[1:5])
s = murbhip.init_bodies(n, "galaxy")
with murbhip.Simulation(n, soft=2e8) as one, murbhip.Simulation(n, soft=2e8, devices=[0] * shards, exchange="rccl") as many:
    many.set_option("variant", variant)
    many.set_option("overlap", overlap)
    for kv in filter(None, os.environ.get("MURB_TEST_OPTIONS", "").split(",")):   # further library options: "key=value,key=value"
        key, value = kv.split("=")
        many.set_option(key, int(value))
    for sim in (one, many):
        sim.upload(s)
        sim.steps(3600.0, 4)
        sim.sync()
    a, b = one.state(), many.state()
    assert int(many.info("variant")) == variant
scale = max(np.abs(a[k]).max() for k in ("qx", "qy", "qz"))
vscale = max(np.abs(a[k]).max() for k in ("vx", "vy", "vz"))
for k in ("qx", "qy", "qz"):
    assert np.abs(a[k] - b[k]).max() <= 2e-6 * scale, k
for k in ("vx", "vy", "vz"):
    assert np.abs(a[k] - b[k]).max() <= 2e-5 * vscale, k
print("ok")
