"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libmurbref.so, compiled from the
sources under /root/reference by `make -C oracle ref`).  Run in the build container only:

    python tests/golden/make_golden.py

The fixtures are data (inputs + expected outputs of the reference's own classes), not source.
Cases follow the reference's own hot-path test, src/test/implem/test_SimulationNBody.cpp:73-82
(n=2048/2049, random/galaxy, soft=2e8, dt=3600, 1/3/4/3 iterations) and the benchmark
configuration N=30000 galaxy (README.md:54-70), stored as checksums + samples to stay small.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as O  # noqa: E402

POS = ("qx", "qy", "qz")
DYN = ("qx", "qy", "qz", "vx", "vy", "vz")


def checksums(a):
    a = np.ascontiguousarray(a, np.float32)
    return np.array([a.astype(np.float64).sum(), (a.astype(np.float64) ** 2).sum(),
                     float(np.bitwise_xor.reduce(a.view(np.uint32)))], np.float64)


def small_case(n, scheme, iters):
    out = {}
    opt = O.RefSim("cpu+optim", n, scheme)
    init = opt.state(with_padding=True)
    out["padding"] = np.array([opt.padding])
    out["flops_per_ite"] = np.array([opt.flops_per_ite()], np.float32)
    out["allocated_bytes"] = np.array([opt.allocated_bytes()], np.float32)
    for k in O.FIELDS:
        out["init_" + k] = init[k]
    opt.step(1)
    for c, a in zip("xyz", opt.acc()):
        out["optim_acc1_a" + c] = a
    s = opt.state()
    for k in DYN:
        out["optim_step1_" + k] = s[k]
    opt.step(iters - 1) if iters > 1 else None
    s = opt.state()
    for k in DYN:
        out["optim_final_" + k] = s[k]
    opt.close()
    nv = O.RefSim("cpu+naive", n, scheme)
    nv.step(iters)
    s = nv.state()
    for k in POS:
        out["naive_final_" + k] = s[k]
    nv.close()
    out["iters"] = np.array([iters])
    np.savez_compressed(os.path.join(HERE, f"ref_{scheme}_{n}.npz"), **out)
    print("wrote", f"ref_{scheme}_{n}.npz")


def bench_case(n=30000, scheme="galaxy"):
    out = {}
    opt = O.RefSim("cpu+optim", n, scheme)
    sub = np.arange(0, n, 59)
    edge = np.r_[0:64, n - 64:n]
    out["sub"], out["edge"] = sub, edge
    s = opt.state()
    for k in O.FIELDS:
        out["init_sum_" + k] = checksums(s[k])
        out["init_edge_" + k] = s[k][edge]
    f64 = O.accel_f64_subset(s, sub)
    for c, a in zip("xyz", f64):
        out["f64_acc1_sub_a" + c] = a
    for it in (1, 5):
        opt.step(1 if it == 1 else 4)
        s = opt.state()
        for k in DYN:
            out[f"optim_step{it}_sum_{k}"] = checksums(s[k])
            out[f"optim_step{it}_edge_{k}"] = s[k][edge]
            out[f"optim_step{it}_sub_{k}"] = s[k][sub]
        if it == 1:
            for c, a in zip("xyz", opt.acc()):
                out["optim_acc1_sub_a" + c] = a[sub]
                out["optim_acc1_sum_a" + c] = checksums(a)
    opt.close()
    np.savez_compressed(os.path.join(HERE, f"ref_{scheme}_{n}_summary.npz"), **out)
    print("wrote", f"ref_{scheme}_{n}_summary.npz")


def integrator_case(n=4000):
    # the shape of src/test/implem/test_CUDABodies.cpp:42-75: synthetic accelerations, dt=0.01, 4 steps
    for scheme in ("random", "galaxy"):
        acc = (np.arange(1, n + 1, dtype=np.float32), np.full(n, 3.0, np.float32),
               (n - np.arange(n)).astype(np.float32))
        out = {}
        for steps in (1, 4):
            r = O.ref_integrate(n, scheme, acc, np.float32(0.01), steps)
            for k in DYN:
                out[f"steps{steps}_{k}"] = r[k]
        np.savez_compressed(os.path.join(HERE, f"ref_integrator_{scheme}_{n}.npz"), **out)
        print("wrote", f"ref_integrator_{scheme}_{n}.npz")


if __name__ == "__main__":
    if not O.have_ref():
        O.build(ref=True)
    small_case(2048, "random", 1)
    small_case(2049, "random", 3)
    small_case(2048, "galaxy", 4)
    small_case(2049, "galaxy", 3)
    integrator_case()
    bench_case()
