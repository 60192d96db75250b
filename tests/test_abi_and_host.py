"""CPU: the C-ABI library loads and exports everything include/murbhip.h declares; host-only entry
points behave; the C++ host mirror (Bodies) reproduces the reference's initial conditions and
integrator bit for bit (fixtures from the compiled reference).  No compute call needs a GPU here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def mh():
    import murbhip
    murbhip.lib()
    return murbhip


def test_every_declared_symbol_is_exported(mh):
    header = open(os.path.join(ROOT, "include", "murbhip.h")).read()
    declared = sorted(set(re.findall(r"\b(murbhip_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(mh.EXPORTS), "python binding list out of date with the header"
    L = mh.lib()
    for name in declared:
        assert hasattr(L, name), name + " declared in include/murbhip.h but not exported"
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.normpath(mh.LIB_PATH)], capture_output=True, text=True)
    exported = set(re.findall(r" T (murbhip_[a-z_0-9]+)", nm.stdout))
    assert exported == set(declared), exported ^ set(declared)
    assert L.murbhip_version() == 100


def test_no_oracle_in_product(mh):
    """The shipped libraries must not link or reference anything under oracle/."""
    for lib in ("libmurbhip.so", "libmurbhost.so"):
        path = os.path.join(ROOT, "nbody-eurohpc_amd", "lib", lib)
        out = subprocess.run(["ldd", path], capture_output=True, text=True).stdout
        assert "oracle" not in out and "murbref" not in out
        assert b"liboracle" not in open(path, "rb").read()


@pytest.mark.parametrize("n,world", [(30000, 4), (200000, 8), (2049, 2), (2049, 3), (7, 8), (1000000, 8), (5, 1)])
def test_partition_is_the_reference_rule(mh, n, world):
    """counts[r] = n/world + (r < n%world), displs = prefix sums (SimulationNBodyMultiNode.cpp:76-91)."""
    first = 0
    for r in range(world):
        f, c = mh.partition(n, world, r)
        assert c == n // world + (1 if r < n % world else 0)
        assert f == first
        first += c
    assert first == n
    slots = mh.slice_slots(n, world)
    assert slots % 1024 == 0 and slots >= mh.partition(n, world, 0)[1] and slots - 1024 < max(mh.partition(n, world, 0)[1], 1)
    for i in {0, n // 2, n - 1, min(n - 1, n // world), min(n - 1, n // world + 1)}:
        s = mh.slot_of_body(n, world, i)
        r = s // slots
        f, c = mh.partition(n, world, r)
        assert f <= i < f + c and s - r * slots == i - f


def test_host_entry_points_reject_bad_arguments(mh):
    L = mh.lib()
    f, c = C.c_ulong(), C.c_ulong()
    assert L.murbhip_partition(10, 0, 0, C.byref(f), C.byref(c)) == -2000
    assert L.murbhip_partition(10, 2, 2, C.byref(f), C.byref(c)) == -2000
    assert "invalid" in mh.error_string(-2000)
    assert mh.error_string(0) == "success"
    assert "HIP error" in mh.error_string(-1)
    assert L.murbhip_sync(None) == -2000
    assert L.murbhip_step(None, 1.0) == -2000
    assert L.murbhip_destroy(None) == 0


def test_fails_loudly_without_a_gpu(mh):
    if mh.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(mh.MurbHipError) as e:
        mh.Simulation(1000)
    assert e.value.code == -2002     # MURBHIP_E_NO_DEVICE: no CPU fallback exists


@pytest.mark.parametrize("scheme,n", [("random", 2048), ("random", 2049), ("galaxy", 2048), ("galaxy", 2049)])
def test_product_initial_conditions_bit_exact(mh, scheme, n):
    g = np.load(os.path.join(GOLDEN, f"ref_{scheme}_{n}.npz"))
    assert mh.host_padding(n, scheme) == int(g["padding"][0])
    s = mh.init_bodies(n, scheme, with_padding=True)
    for k in mh.FIELDS:
        assert np.array_equal(bits(s[k]), bits(g["init_" + k])), k


def test_product_initial_conditions_benchmark_size(mh):
    g = np.load(os.path.join(GOLDEN, "ref_galaxy_30000_summary.npz"))
    s = mh.init_bodies(30000, "galaxy")
    for k in mh.FIELDS:
        a = s[k]
        got = np.array([a.astype(np.float64).sum(), (a.astype(np.float64) ** 2).sum(),
                        float(np.bitwise_xor.reduce(bits(a)))])
        assert np.array_equal(got, g["init_sum_" + k]), k
    # first bodies quoted in SURVEY.md §8c
    assert s["m"][0] == np.float32(2e24) and s["m"][1] == np.float32(4.20093873e20)
    assert s["qy"][1] == np.float32(117566928.0)


@pytest.mark.parametrize("scheme", ["random", "galaxy"])
def test_product_host_integrator_bit_exact(mh, scheme):
    n = 4000
    g = np.load(os.path.join(GOLDEN, f"ref_integrator_{scheme}_{n}.npz"))
    acc = (np.arange(1, n + 1, dtype=np.float32), np.full(n, 3.0, np.float32), (n - np.arange(n)).astype(np.float32))
    for steps in (1, 4):
        out = mh.host_integrate(n, scheme, acc, np.float32(0.01), steps)
        for k in mh.FIELDS[:6]:
            assert np.array_equal(bits(out[k]), bits(g[f"steps{steps}_{k}"])), k


def test_driver_cli_contract():
    """murb-hip mirrors the reference CLI (main.cpp:61-165): missing -n/-i -> usage + exit(-1);
    unknown --im tag -> message + exit(-1); --soft 0 rejected."""
    exe = os.path.join(ROOT, "nbody-eurohpc_amd", "bin", "murb-hip")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 255 and "Usage" in r.stdout
    r = subprocess.run([exe, "-n", "100", "-i", "1", "--im", "cpu+nope"], capture_output=True, text=True)
    assert r.returncode == 255 and "Implementation 'cpu+nope' does not exist... Exiting." in r.stdout
    r = subprocess.run([exe, "-n", "100", "-i", "1", "--soft", "0"], capture_output=True, text=True)
    assert r.returncode == 255 and "Softening factor can't be equal to 0" in r.stdout


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 8])
@pytest.mark.parametrize("n,split", [(9000, 1), (9000, 4), (30000, 2), (200000, 1)])
def test_half_ring_schedule_covers_every_pair_once(mh, n, world, split):
    """Multi-GPU pair-symmetric schedule (murbhip_schedule_items, host only): over all ranks every
    unordered pair of (sub-block, block) cells is listed exactly once, each rank's share is balanced,
    and a rank only ever walks sub-blocks of its own slice on the i side."""
    if n // world < 1024 and world > 4:
        pytest.skip("slices of one block: nothing to balance")
    tb = mh.slice_slots(n, world) // 1024
    ts = tb * split
    seen = {}
    counts = []
    for r in range(world):
        items, own = mh.schedule_items(n, world, r, split)
        counts.append(len(items))
        assert own == split * tb * (tb + 1) // 2
        for k, (i, j) in enumerate(items.tolist()):
            assert r * ts <= i < (r + 1) * ts, "i side must be an own sub-block"
            if k < own:
                assert r * tb <= j < (r + 1) * tb and i // split <= j
            else:
                assert not (r * tb <= j < (r + 1) * tb)
            # canonical key of the unordered cell {sub-block i, block j}: own-slice items are ordered
            # (block(i) <= j) and unique by construction; a rectangle cell {i, j} could also be listed by
            # the rank owning j as (some sub-block of j, block(i)): express both in sub-block pairs
            for jj in range(j * split, (j + 1) * split):
                if i // split == j:            # diagonal block: both orders evaluated inside the item
                    key = ("diag", i, jj)
                else:
                    key = ("off", min(i, jj), max(i, jj))
                seen[key] = seen.get(key, 0) + 1
    assert all(v == 1 for v in seen.values()), "a cell is evaluated twice"
    total_sub = world * ts
    off = sum(1 for k in seen if k[0] == "off")
    diag = sum(1 for k in seen if k[0] == "diag")
    assert diag == world * tb * split * split                      # every (sub-block, sub-block) cell of a diagonal block
    assert off == (total_sub * total_sub - world * tb * split * split) // 2   # every other unordered pair exactly once
    if world > 1 and tb >= 2:
        assert max(counts) - min(counts) <= split * tb, counts     # balanced up to one block row


def test_simulation_history_csv(mh, tmp_path):
    """SimulationHistory<double>::saveMetricsToCSV: the reference's column names and max_digits10
    precision (SimulationHistory.cpp:103-121), so the doubles survive the round trip exactly."""
    rng = np.random.default_rng(3)
    e, a, c = rng.normal(size=5) * 1e29, rng.uniform(size=5) * 1e34, rng.normal(size=(5, 3)) * 1e7
    path = tmp_path / "metrics.csv"
    assert mh.history_csv(path, e, a, c)
    lines = path.read_text().splitlines()
    assert lines[0] == "iteration,energy,ang_momentum,density_center_x,density_center_y,density_center_z"
    assert len(lines) == 6
    for i, line in enumerate(lines[1:]):
        f = line.split(",")
        assert int(f[0]) == i
        assert [float(x) for x in f[1:]] == [e[i], a[i], c[i, 0], c[i, 1], c[i, 2]]
    # an unwritable path is an error, not a silent no-op (the reference throws std::runtime_error)
    assert not mh.history_csv(tmp_path / "no_such_dir" / "metrics.csv", e, a, c)


def test_simulation_history_container_semantics():
    """The host cases of the reference's test_SimulationHistory.cu:12-77 restated for host/core/SimulationHistory.hpp
    (a small C++ program, tests/helpers/history_selftest.cpp)."""
    exe = os.path.join(ROOT, "tests", "helpers", "_build", "history_selftest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "helpers"), "_build/history_selftest"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout
