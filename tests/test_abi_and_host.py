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
    assert L.murbhip_version() == 103


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


def real_bodies_per_block(mh, n, world):
    """Real (non-padding) bodies of every global block: slice s keeps its bodies in its first slots."""
    tb = mh.slice_slots(n, world) // 1024
    fill = np.zeros(world * tb, np.int64)
    for sl in range(world):
        cnt = mh.partition(n, world, sl)[1]
        for b in range(tb):
            fill[sl * tb + b] = min(1024, max(0, cnt - b * 1024))
    return fill


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 8])
@pytest.mark.parametrize("n,split", [(9000, 1), (9000, 4), (30000, 2), (200000, 1), (2049, 2)])
def test_half_ring_schedule_covers_every_pair_once(mh, n, world, split):
    """Multi-GPU pair-symmetric schedule (murbhip_schedule_items, host only): over all ranks every unordered pair of
    (sub-block, sub-block) cells that hold REAL bodies is covered exactly once (padding never twice), each rank's share is
    balanced, a rank's triangle stays inside its own slice and every rectangle item has an own block on one side.  The
    emptier block of a pair is the walked (i) side and its padding sub-blocks are not walked at all."""
    if n // world < 1024 and world > 4:
        pytest.skip("slices of one block: nothing to balance")
    tb = mh.slice_slots(n, world) // 1024
    ts = tb * split
    sub = 1024 // split
    fill = real_bodies_per_block(mh, n, world)
    real_sub = np.array([fill[u // split] > (u % split) * sub for u in range(world * ts)])   # sub-block holds a real body
    seen = {}
    counts = []
    for r in range(world):
        items, own = mh.schedule_items(n, world, r, split)
        # a rank's work: bodies it walks (each against a staged block of 1024)
        counts.append(sum(min(sub, max(1, int(fill[i // split])) - (i % split) * sub) for i, _ in items.tolist()))
        assert own <= split * tb * (tb + 1) // 2
        mine = lambda blk: r * tb <= blk < (r + 1) * tb   # noqa: E731
        for k, (i, j) in enumerate(items.tolist()):
            bi = i // split
            if k < own:
                assert mine(bi) and mine(j), "the own-slice triangle needs no remote data"
            else:
                assert mine(bi) != mine(j), "a rectangle item pairs an own block with another slice's"
            assert fill[bi] <= fill[j] or bi == j, "the emptier block of a pair is the walked one"
            assert real_sub[i] or (i % split == 0), "a padding sub-block is never walked (a block's first one always is)"
            # canonical key of the unordered cell {sub-block i, sub-block jj}; diagonal blocks evaluate both orders
            # inside the item
            for jj in range(j * split, (j + 1) * split):
                key = ("diag", i, jj) if bi == j else ("off", min(i, jj), max(i, jj))
                seen[key] = seen.get(key, 0) + 1
    assert all(v == 1 for v in seen.values()), "a cell is evaluated twice"
    # every cell with real bodies on both sides is there
    real_ids = np.flatnonzero(real_sub)
    for u in real_ids:
        for v in real_ids:
            if u // split == v // split:
                assert ("diag", int(u), int(v)) in seen
            elif u < v:
                assert ("off", int(u), int(v)) in seen
    if world > 1 and tb >= 2:
        assert max(counts) - min(counts) <= 1024 * tb, counts      # balanced up to one block row (walked bodies)


@pytest.mark.parametrize("diag_tri", [False, True])
@pytest.mark.parametrize("n,world,split,waves,taper,exchange,tri_div", [p_ if len(p_) == 7 else p_ + (1,) for p_ in [
    (5000, 1, 1, 4, 0, False), (5000, 1, 4, 4, 0, False), (9000, 1, 8, 8, 0, False), (9000, 1, 2, 4, 50, False),
    (9000, 1, 8, 8, 40, False), (9000, 1, 16, 4, 100, False), (3000, 1, 1, 4, 0, True),      # one rank with the exchange pipeline (RCCL self-test)
    (9000, 2, 1, 4, 0, True), (9001, 3, 2, 4, 30, True), (20000, 4, 4, 4, 0, True), (20000, 4, 2, 8, 60, True),
    (30000, 5, 1, 4, 0, True), (60000, 8, 4, 4, 25, True), (30000, 1, 4, 8, 30, False), (2049, 2, 2, 4, 0, True),
    (60000, 8, 4, 4, 0, True, 2), (100000, 8, 4, 4, 0, True, 4), (40000, 4, 2, 8, 20, True, 2),     # the triangle's launches cut finer ("tri_div")
]])
def test_pair_symmetric_layout_is_complete_and_collision_free(mh, n, world, split, waves, taper, exchange, diag_tri, tri_div):
    """What the pair-symmetric kernel is handed (murbhip_schedule_layout, host only), for every rank of a run:
      * every ordered (i, j) interaction between REAL bodies is applied exactly once over all items of all ranks (at the
        granularity of 16 slots), with any item size mix the taper produces; nothing is applied twice;
      * every cell of every partial row has at most one writer, and exactly one where the slot holds a real body;
      * pushing "how many REAL bodies did this cell's writer sum over" through the row tables, the reduce-scatter chunk
        layout and the own-triangle addend gives every real slot exactly n interactions — i.e. the data flow of a step,
        launch by launch, loses and duplicates nothing;
      * padding is walked only up to the next multiple of 16 x waves bodies behind a block's last real body."""
    slice_ = mh.slice_slots(n, world)
    slots, tb = slice_ * world, slice_ // 1024
    G = 16
    fill = real_bodies_per_block(mh, n, world)
    real = np.zeros(slots, bool)
    for b, f in enumerate(fill):
        real[b * 1024:b * 1024 + f] = True
    assert real.sum() == n
    nreal = lambda lo, hi: int(real[lo:hi].sum())   # noqa: E731
    count = np.zeros((slots // G, slots // G), np.int32)
    recv = np.zeros((world, slice_), np.int64)     # what the reduce-scatter delivers: sum over ranks of their chunk for a slice
    own = np.zeros((world, slice_), np.int64)      # own-triangle row sums (exchange pipeline) / everything (one GPU)
    for r in range(world):
        items, rows, fm, ft = mh.schedule_layout(n, world, r, split, waves, taper, 50, exchange, diag_tri, tri_div)
        assert len(items) > 0
        if tri_div > 1:   # the own-slice triangle's items (row set 1) are finer than the rectangles' whole sub-blocks
            assert items[items[:, 4] == 1][:, 1].max() <= max(1024 // split // tri_div, 16 * waves) and items[items[:, 4] == 0][:, 1].max() == 1024 // split
        sets = {0: np.zeros(fm, np.int32), 1: np.zeros(ft, np.int32)}      # writers per cell
        vals = {0: np.zeros(fm, np.int64), 1: np.zeros(ft, np.int64)}      # real bodies summed into the cell
        needs = {0: np.zeros(fm, bool), 1: np.zeros(ft, bool)}             # cells that belong to a real slot
        launches = items[:, 7]
        assert (np.diff(launches) >= 0).all()                              # launch order
        for i0, ln, J, flags, st, ioff, joff, launch in items:
            assert ln % (4 * waves) == 0 and ln >= 16 * waves and i0 % (4 * waves) == 0 and i0 // 1024 == (i0 + ln - 1) // 1024
            bi = i0 // 1024
            assert i0 % 1024 < max(fill[bi], 1), "an item that walks nothing but padding"
            assert i0 % 1024 + ln < max(fill[bi], 1) + 16 * waves, "padding is walked only up to the item granule"
            diag = bi == J
            j_side, tri = not (flags & 1), bool(flags & 2)
            assert diag or (j_side and not tri)
            assert tri == (diag and diag_tri and ln <= 128)
            if exchange or world > 1:
                assert st == (1 if launch < 2 else 0)
                if st == 1:
                    assert bi // tb == r and J // tb == r                    # the triangle needs no remote data
                else:
                    assert (bi // tb == r) != (J // tb == r)                 # a rectangle: one own block, one of another slice
            assert fill[bi] <= fill[J]                                       # the emptier block is the walked one
            # which j steps (of 128 bodies) the item evaluates, and from which one on it applies both sides
            p_first, p_sym = ((flags >> 8) & 15, (flags >> 12) & 15) if tri else (0, 8 if diag else 0)
            if tri:
                assert ln <= 128 and p_first == (i0 % 1024) // 128 and p_sym == p_first + 1
            assert j_side == (p_sym < 8) or (tri and not j_side)             # the last REAL triangular piece may have later (padding) steps
            j0 = J * 1024
            count[i0 // G:(i0 + ln) // G, (j0 + 128 * p_first) // G:(j0 + 1024) // G] += 1
            sets[st][ioff:ioff + ln] += 1
            needs[st][ioff:ioff + ln] |= real[i0:i0 + ln]
            vals[st][ioff:ioff + ln] += nreal(j0 + 128 * p_first, j0 + 1024)
            if j_side:
                count[(j0 + 128 * p_sym) // G:(j0 + 1024) // G, i0 // G:(i0 + ln) // G] += 1
                sets[st][joff:joff + 1024] += 1
                needs[st][joff:joff + 1024] |= real[j0:j0 + 1024]
                vals[st][joff + 128 * p_sym:joff + 1024] += nreal(i0, i0 + ln)
        for st in (0, 1):
            assert (sets[st] <= 1).all(), (r, st, np.unique(sets[st]))       # never two writers
        seen = {0: np.zeros(fm, bool), 1: np.zeros(ft, bool)}
        for st, out_slice, out_block, base_i, ni, base_j, nj in rows:
            assert not seen[st][base_j:base_j + nj * 1024].any() and not seen[st][base_i:base_i + ni * 1024].any()
            seen[st][base_j:base_j + nj * 1024] = True
            seen[st][base_i:base_i + ni * 1024] = True
            blk = (r * tb + out_block) if (st == 1 or not (exchange or world > 1)) else out_slice * tb + out_block
            blk_real = real[blk * 1024:(blk + 1) * 1024]
            for base, cnt in ((base_j, nj), (base_i, ni)):                   # a row's cells of real slots all have their writer
                w = sets[st][base:base + cnt * 1024].reshape(cnt, 1024)
                nd = needs[st][base:base + cnt * 1024].reshape(cnt, 1024)
                assert (w[:, blk_real] == 1).all() and not nd[:, ~blk_real].any()
            tot = vals[st][base_j:base_j + nj * 1024].reshape(nj, 1024).sum(0) + vals[st][base_i:base_i + ni * 1024].reshape(ni, 1024).sum(0)
            if exchange or world > 1:
                if st == 1:
                    assert out_slice == 0
                    own[r, out_block * 1024:(out_block + 1) * 1024] += tot
                else:
                    recv[out_slice, out_block * 1024:(out_block + 1) * 1024] += tot
            else:
                own[0, out_block * 1024:(out_block + 1) * 1024] += tot
        for st in (0, 1):
            assert seen[st].all()                                          # the tables account for every row
    assert (count <= 1).all(), np.unique(count)
    real_cells = real.reshape(-1, G).any(1)
    assert (count[np.ix_(real_cells, real_cells)] == 1).all()
    assert ((own + recv).reshape(-1)[real] == n).all()


@pytest.mark.parametrize("binary,members,rounds", [("crew_selftest", 8, 2000), ("crew_selftest", 1, 50), ("crew_selftest", 3, 2000),
                                                   ("crew_selftest_tsan", 6, 300)])
def test_shard_crew_selftest(binary, members, rounds):
    """The per-shard host threads of the library (csrc/murb_crew.h: run / meet, no HIP in it) driven on the CPU: every member
    runs every job once on its own thread, meet() and run() are barriers (members of different speed), a failing member's
    code comes back while the barriers still pair up, sleeping members wake up, destruction joins them.  Once more under
    ThreadSanitizer (host code only): no data race in the hand-over of jobs and results."""
    exe = os.path.join(ROOT, "tests", "helpers", "_build", binary)
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "helpers"), "_build/crew_selftest"], check=True, timeout=600)
    r = subprocess.run([exe, str(members), str(rounds)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok" and "ThreadSanitizer" not in r.stderr, (r.stdout, r.stderr[-2000:])


@pytest.mark.parametrize("ending", ["abort", "kill", "normal"])
def test_bench_last_line_keeper(ending):
    """bench.py's LastLineKeeper without a GPU: whatever ends the writing process — abort(), SIGKILL, or a normal end — stdout
    carries exactly ONE line, the last one that was sent (rank 0 of an N > 1 run sends its line once when the main measurement
    is complete and again at the very end)."""
    import sys
    end = {"abort": "os.abort()", "kill": "os.kill(os.getpid(), 9)", "normal": "k.close_and_wait()"}[ending]
    code = (
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "k = bench.LastLineKeeper(os.fdopen(os.dup(1), 'w'))\n"
        "os.dup2(2, 1)\n"
        "k.write('{\"value\": 1, \"incomplete\": true}\\n'); k.flush()\n"
        "print('library chatter on the C-level stdout')\n"
        "k.write('{\"value\": 1, \"pad\": \"' + 'x' * 200000 + '\"}\\n'); k.flush()\n"
        "%s\n"
    ) % (ROOT, end)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, (r.stdout[:300], r.stderr[-1000:])
    import json
    d = json.loads(lines[0])
    assert d["value"] == 1 and len(d["pad"]) == 200000 and "incomplete" not in d
    assert (r.returncode == 0) == (ending == "normal")


def test_bench_host_facts_and_roofline_arithmetic():
    """bench.py without a GPU: the host's share is read at import, BEFORE OpenMP can narrow the main thread's affinity mask
    (usable cores = physical cores in the mask, capped by the cgroup CPU quota) and becomes the CPU baseline's thread count; and
    the roofline arithmetic of a measurement (interactions per launch from the library: N^2 / passes under the multi-pass
    evaluation; (N / ranks) x N over the launches of a step for several ranks)."""
    code = (
        "import json, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "m = {'info': {'variant': 8}, 'force_launches': 30, 'steps': 10, 'interactions_per_launch': 3.6e9 / 3, 'force_ms_avg': 0.2, 'events': 'x'}\n"
        "one = bench.roofline_of(m, 60000, 1, 157.2864)\n"
        "m2 = dict(m, force_launches=60, steps=20, force_ms_avg=0.25)\n"
        "many = bench.roofline_of(m2, 200000, 8, 157.2864)\n"
        "print(json.dumps({'host': bench.HOST, 'omp': os.environ['OMP_NUM_THREADS'], 'ipc': os.environ['HSA_ENABLE_IPC_MODE_LEGACY'], 'one': one, 'many': many}))\n"
    ) % ROOT
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "HSA_ENABLE_IPC_MODE_LEGACY")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    d = json.loads(r.stdout.strip().splitlines()[-1])
    h = d["host"]
    assert 1 <= h["cpus_usable"] <= h["physical_cores_in_mask"] <= h["cpus_in_affinity_mask"] <= h["logical_cpus"] and h["model"]
    assert h["cpus_in_affinity_mask"] == len(os.sched_getaffinity(0))
    if h["cgroup_cpu_quota"]:
        assert h["cpus_usable"] <= max(1, int(h["cgroup_cpu_quota"] + 1e-9))
    assert d["omp"] == str(h["cpus_usable"]) and d["ipc"] == "0"
    one, many = d["one"], d["many"]
    assert one["bound"] == "valu" and one["launches_per_step"] == 3 and abs(one["interactions_per_launch"] - 1.2e9) < 1
    assert abs(one["achieved"] - 20 * 1.2e9 / 0.2e-3 / 1e12) < 1e-9 and abs(one["frac"] - one["achieved"] / 157.2864) < 1e-12
    assert abs(one["frac_executed"] - one["frac"] * 13 / 20) < 1e-12
    assert abs(many["interactions_per_launch"] - 25000.0 * 200000.0 / 3) < 1 and many["launches_per_step"] == 3


def test_planner_selftest_under_sanitizers():
    """The host-side planner (csrc/murb_plan.h, murb_schedule.h: item tables and partial-row layouts, no HIP in them)
    compiled with g++ under AddressSanitizer + UBSan and swept over a few thousand plans (sizes 1 … 60 001, 1-8 ranks, splits,
    waves, tapers, triangular diagonals, the exchange pipeline, several passes, finer triangle launches): every index the
    planner computes is in bounds, no partial-row cell has two writers, tables and passes account for every item."""
    exe = os.path.join(ROOT, "tests", "helpers", "_build", "plan_selftest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "helpers"), "_build/plan_selftest"], check=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok ") and int(r.stdout.split()[1]) > 2000, (r.stdout, r.stderr[-3000:])


def test_layout_query_rejects_bad_arguments(mh):
    lib = mh.lib()
    cnt = C.c_ulong()
    bad = [(1000, 0, 0, 1, 4, 0, 50, 0), (1000, 2, 2, 1, 4, 0, 50, 0), (1000, 1, 0, 3, 4, 0, 50, 0), (1000, 1, 0, 1, 5, 0, 50, 0),
           (1000, 1, 0, 1, 4, 101, 50, 0), (1000, 1, 0, 16, 8, 0, 50, 0), (1000, 65, 0, 1, 4, 0, 50, 0)]
    for a in bad:
        assert lib.murbhip_schedule_layout(*a, None, 0, C.byref(cnt), None, 0, C.byref(cnt), None, None) == -2000, a


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
