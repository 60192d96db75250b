"""CPU, build container only: the oracle restatement against the REAL reference compiled from
/root/reference (oracle/_ref/libmurbref.so).  Skipped where that library does not exist or the
reference tree is absent (the GPU box) — there tests/test_oracle_golden.py carries the pin."""
import numpy as np
import pytest

SOFT, DT = np.float32(2e8), np.float32(3600.0)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ref(O):
    if not O.have_ref():
        pytest.skip("oracle/_ref/libmurbref.so not built (needs /root/reference)")
    O.ref_lib()
    return O


@pytest.mark.parametrize("scheme", ["galaxy", "random"])
@pytest.mark.parametrize("n", [1, 2, 63, 1000, 2049])
def test_init_bit_exact(ref, scheme, n):
    r = ref.RefSim("cpu+optim", n, scheme)
    want = r.state(with_padding=True)
    got = ref.init_bodies(n, scheme, with_padding=True)
    assert r.padding == ref.padding(n)
    for k in ref.FIELDS:
        assert np.array_equal(bits(want[k]), bits(got[k])), k
    r.close()


@pytest.mark.parametrize("scheme", ["galaxy", "random"])
def test_optim_bit_exact_over_steps(ref, scheme):
    n = 1500
    r = ref.RefSim("cpu+optim", n, scheme, SOFT, DT)
    s = ref.init_bodies(n, scheme)
    for _ in range(3):
        r.step(1)
        acc = ref.simulate(s, 1, "cpu+optim", SOFT, DT)
        for a, b in zip(acc, r.acc()):
            assert np.array_equal(bits(a), bits(b))
        want = r.state()
        for k in ref.FIELDS[:6]:
            assert np.array_equal(bits(s[k]), bits(want[k])), k
    r.close()


def test_naive_close(ref):
    n = 700
    r = ref.RefSim("cpu+naive", n, "random", SOFT, DT)
    s = ref.init_bodies(n, "random")
    r.step(2)
    ref.simulate(s, 2, "cpu+naive", SOFT, DT)
    want = r.state()
    for k in ("qx", "qy", "qz"):
        np.testing.assert_allclose(s[k], want[k], rtol=1e-5)
    r.close()


def test_integrator_bit_exact(ref):
    n = 1234
    acc = (np.linspace(-3, 7, n, dtype=np.float32), np.full(n, 1e-3, np.float32), np.linspace(5, -5, n, dtype=np.float32))
    for scheme, dt in (("galaxy", np.float32(3600.0)), ("random", np.float32(0.01))):
        want = ref.ref_integrate(n, scheme, acc, dt, 3)
        s = ref.init_bodies(n, scheme)
        for _ in range(3):
            ref.integrate(s, acc, dt)
        for k in ref.FIELDS[:6]:
            assert np.array_equal(bits(s[k]), bits(want[k])), k


def test_simd_and_omp_are_speed_baselines_only(ref):
    """cpu+simd / cpu+omp use a 12-bit rsqrt (SimulationNBodySIMD.cpp:12-32): ~8e-4 from cpu+optim."""
    n = 1024
    opt = ref.RefSim("cpu+optim", n, "galaxy")
    opt.step(1)
    for tag in ("cpu+simd", "cpu+omp"):
        r = ref.RefSim(tag, n, "galaxy")
        r.step(1)
        e = ref.rel_err(r.acc(), opt.acc())
        assert 1e-5 < e.max() < 5e-3
        r.close()
    opt.close()


def test_file_scheme_matches_reference(ref, tmp_path, monkeypatch):
    """Any scheme other than galaxy/random reads `milkyway_andromeda.tab` from the working directory
    (Bodies.cpp:14-25, 83-153; the file itself is not in the reference repository): one body per non-empty
    line, rescaled by component galaxy.  A synthetic 70 000-line file reaches five of the six index ranges;
    the product's host mirror must produce the reference's arrays bit for bit, and both must refuse a
    missing file."""
    import murbhip
    rng = np.random.default_rng(11)
    n = 70000
    rows = rng.normal(size=(n, 7)).astype(np.float32)
    rows[:, 0] = np.abs(rows[:, 0]) * 1e-5
    with open(tmp_path / "milkyway_andromeda.tab", "w") as f:
        for k, r in enumerate(rows):
            f.write(" ".join(f"{x:.9g}" for x in r) + "\n")
            if k % 1000 == 0:
                f.write("\n")                      # empty lines are skipped
    monkeypatch.chdir(tmp_path)
    r = ref.RefSim("cpu+naive", 5, "collision")    # n is replaced by the number of lines
    assert r.n == n and r.padding == 0
    want = r.state()
    r.close()
    got = murbhip.init_bodies(n, "collision")
    for k in ref.FIELDS:
        assert np.array_equal(bits(want[k]), bits(got[k])), k
    assert np.all(got["r"] == np.float32(1e5))
    # Milky Way ranges use 4.5e10 / 4.0 / 220, Andromeda ranges 9.4e10 / 6.0 / 260
    assert got["m"][0] == np.float32(np.float64(rows[0, 0]) * 4.5e10) and got["qx"][20000] == np.float32(np.float64(rows[20000, 1]) * 6.0)
    assert got["vz"][66000] == np.float32(np.float64(rows[66000, 6]) * 260)
