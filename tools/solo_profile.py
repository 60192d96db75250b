import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))
import murbhip
n, w, split = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
s = murbhip.init_bodies(n, "galaxy")
with murbhip.Simulation(n, devices=[0] * w) as sim:
    sim.set_option("variant", variant); sim.set_option("jsplit", split); sim.set_option("solo_shard", 0)
    sim.upload(s); sim.steps(3600.0, 30); sim.sync()
