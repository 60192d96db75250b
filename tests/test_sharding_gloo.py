"""CPU, world_size 2 (and 3) over gloo: the N>1 path's partition, padded-slot layout and per-step
equal-count all-gather reproduce the single-rank result.  (Not bit for bit: the oracle's j loop is
vectorised with -ffast-math, and the zero-mass padding slots between the slices shift which SIMD lane
sums which body — an fp32 reassociation of ~1e-7 on the accelerations.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n", [(2, 1031), (3, 700)])
def test_partitioned_steps_match_single_rank(tmp_path, O, world, n):
    steps = 3
    out = str(tmp_path / "sharded.npz")
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_gloo_worker.py"), str(n), str(steps), out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)

    import murbhip
    s = murbhip.init_bodies(n, "galaxy")
    for _ in range(steps):
        acc = O.accel_slice_f32(s, 0, n, np.float32(2e8))
        O.integrate(s, acc, np.float32(3600.0))
    for k in ("qx", "qy", "qz"):
        np.testing.assert_allclose(got[k], s[k], rtol=1e-6, atol=1.0)
    for k in ("vx", "vy", "vz"):
        np.testing.assert_allclose(got[k], s[k], rtol=1e-5, atol=1e-4)
    # and the slices really moved: a rank that never received its peer's positions would be far off
    assert np.abs(got["qx"] - murbhip.init_bodies(n, "galaxy")["qx"]).max() > 1e5


@pytest.mark.parametrize("world,n,split", [(2, 3000, 1), (3, 4500, 2)])
def test_half_ring_schedule_matches_direct_sum(tmp_path, world, n, split):
    """The multi-GPU pair-symmetric schedule end to end on CPU ranks (gloo): positions after 2 steps equal
    a single-process fp64 direct sum to rounding."""
    steps = 2
    out = str(tmp_path / "ring.npy")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_gloo_ring_worker.py"), str(n), str(steps), str(split), out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)

    import murbhip
    s = murbhip.init_bodies(n, "galaxy")
    pos = np.stack([s["qx"], s["qy"], s["qz"]], 1).astype(np.float64)
    vel = np.stack([s["vx"], s["vy"], s["vz"]], 1).astype(np.float64)
    gm = 6.67384e-11 * s["m"].astype(np.float64)
    for _ in range(steps):
        acc = np.zeros_like(pos)
        for a0 in range(0, n, 500):                      # direct N^2 sum in chunks
            d = pos[None, :, :] - pos[a0:a0 + 500, None, :]
            inv3 = (np.einsum("ijk,ijk->ij", d, d) + 2e8 * 2e8) ** -1.5
            acc[a0:a0 + 500] = np.einsum("ij,ijk->ik", inv3 * gm[None, :], d)
        pos += (vel + acc * 3600.0 * 0.5) * 3600.0
        vel += acc * 3600.0
    np.testing.assert_allclose(got, pos, rtol=1e-11, atol=1e-3)
