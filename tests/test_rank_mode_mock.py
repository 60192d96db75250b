"""GPU: the one-process-per-GPU path (murbhip_create_rank with rank > 0 and world > 1) with SEVERAL PROCESSES
ON ONE GPU.  Real RCCL refuses two ranks on one device, so the collectives come from a stand-in library
(tests/helpers/rccl_mock.cpp, bound through MURBHIP_RCCL_LIBRARY) that has the same entry points and
argument meaning and moves the data through shared host memory.  What this exercises that nothing else on
a one-GPU box can: every rank-mode code path of csrc/murbhip.hip for rank != 0 — slice offsets of the
in-place all-gather, the reduce-scatter chunk layout, the row ranges of the half-ring schedule, and that
all ranks issue their collectives in the same order (a mismatch deadlocks here exactly as it would on
eight GPUs; the test then fails on its timeout)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, free_port

pytestmark = pytest.mark.gpu

MOCK = os.path.join(ROOT, "tests", "helpers", "_build", "librccl_mock.so")
WORKER = os.path.join(ROOT, "tests", "_rank_worker.py")
SOFT, DT = np.float32(2e8), np.float32(3600.0)


def run_ranks(tmp_path, world, n, steps, variant, overlap=1, jsplit=0, integrator=0, mode="async", options=""):
    """mode: "async" = the stand-in only enqueues its copies and host functions on the stream it is given (what real
    RCCL does; a missing cross-stream wait in the library then shows as a wrong result), "sync" = it drains the stream."""
    if not os.path.exists(MOCK):   # normally built by __graft_entry__.build()
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "helpers")], check=True, timeout=600)
    env = dict(os.environ, MURBHIP_RCCL_LIBRARY=MOCK, MURB_MOCK_MODE=mode, MURB_TEST_OPTIONS=options)
    # the unique id comes from the same library the ranks will bind: ask a throw-away process for it
    uid = subprocess.run([sys.executable, "-c",
                          f"import sys; sys.path.insert(0, {os.path.join(ROOT, 'nbody-eurohpc_amd')!r}); import murbhip; "
                          "print(murbhip.unique_id().hex())"], env=env, capture_output=True, text=True, timeout=120)
    assert uid.returncode == 0, uid.stderr
    uidhex = uid.stdout.strip()
    assert bytes.fromhex(uidhex)[:8] == b"MOCKRCCL"
    procs = []
    for r in range(world):
        out = tmp_path / f"rank{r}.npz"
        procs.append((out, subprocess.Popen([sys.executable, WORKER, str(r), str(world), uidhex, str(n), str(steps), str(variant),
                                             str(overlap), str(jsplit), str(integrator), str(out)], env=env,
                                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    results, failed = [], []
    for out, p in procs:
        try:
            _, err = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for _, q in procs:
                q.kill()                      # exactly the processes started here
            pytest.fail("ranks did not finish: collective order mismatch or a dead rank")
        if p.returncode != 0:
            failed.append(err[-1500:])
    assert not failed, failed
    for out, _ in procs:
        results.append(dict(np.load(out)))
    return results


@pytest.mark.parametrize("world,n,variant,overlap,jsplit,mode", [
    (2, 9000, 1, 1, 0, "async"),      # one-sided kernel, all-gather only
    (2, 9000, 8, 1, 0, "async"),      # half-ring schedule, even world (shared slice pair)
    (3, 9001, 8, 1, 2, "async"),      # odd world, bodies not divisible by it (ranks own 3001 / 3000 / 3000)
    (4, 10003, 1, 0, 0, "async"),     # one-sided, ragged partition
    (4, 20000, 8, 0, 0, "async"),     # no overlap
    (4, 20000, 8, 1, 4, "async"),
    (3, 12000, 8, 2, 1, "async"),     # overlap mode 2 (at most 4 ranks here: the GPU box allows 6 processes on the card, pytest is one)
    (4, 40000, 8, 1, 0, "async"),     # longer force launches: the collectives really do run beside them
    (2, 9000, 8, 1, 0, "sync"),       # the draining stand-in of round 1, for comparison
    (4, 20000, 8, 1, 4, "sync"),
    (4, 30000, 0, 1, 0, "async"),     # BASELINE's N = 30 000 on 4 ranks with the plan the library picks itself (variant 0)
    # the point-to-point form of both exchanges ("exchange_p2p": grouped ncclSend / ncclRecv instead of the collectives)
    (2, 9000, 8, 1, 0, "async+p2p"), (3, 9001, 8, 1, 2, "async+p2p"), (4, 20000, 8, 1, 4, "async+p2p"), (4, 40000, 8, 2, 0, "async+p2p"),
    (3, 10003, 1, 1, 0, "async+p2p"), (4, 20000, 8, 0, 0, "sync+p2p"),
])
def test_ranks_match_single_gpu(gpu, O, tmp_path, world, n, variant, overlap, jsplit, mode):
    steps = 3 if n < 40000 else 6
    mode, _, p2p = mode.partition("+")
    ranks = run_ranks(tmp_path, world, n, steps, variant, overlap, jsplit, mode=mode, options="exchange_p2p=1" if p2p else "")
    s = O.init_bodies(n, "galaxy")
    with gpu.Simulation(n, soft=SOFT) as one:
        one.upload(s)
        one.compute_acc(); one.sync()
        acc = one.acc()
        one.steps(DT, steps); one.sync()
        ref = one.state()
        ke, pe = one.energy()
    scale = max(np.abs(ref[k]).max() for k in ("qx", "qy", "qz"))
    vscale = max(np.abs(ref[k]).max() for k in ("vx", "vy", "vz"))
    covered = np.zeros(n, bool)
    for r, d in enumerate(ranks):
        assert int(d["used_variant"]) == (variant or int(ranks[0]["used_variant"])) and int(d["used_variant"]) in (1, 2, 8)
        f, c = int(d["first"]), int(d["count"])
        covered[f:f + c] = True
        # every rank holds ALL gathered positions
        for k in ("qx", "qy", "qz"):
            assert np.abs(d[k] - ref[k]).max() <= 2e-6 * scale, (r, k)
        # ... and the velocities and accelerations of its own bodies
        for k in ("vx", "vy", "vz"):
            assert np.abs(d[k][f:f + c] - ref[k][f:f + c]).max() <= 2e-5 * vscale, (r, k)
        own = tuple(d[k][f:f + c] for k in ("ax", "ay", "az"))
        assert O.rel_err(own, tuple(a[f:f + c] for a in acc)).max() <= 2e-6, r
    assert covered.all()
    # positions are bit-identical on all ranks (they all received the same bytes)
    for d in ranks[1:]:
        for k in ("qx", "qy", "qz"):
            assert np.array_equal(d[k].view(np.uint32), ranks[0][k].view(np.uint32))
    # energy: each rank reports its own bodies' share
    assert abs(sum(float(d["ke"]) for d in ranks) - ke) <= 1e-5 * abs(ke)
    assert abs(sum(float(d["pe"]) for d in ranks) - pe) <= 1e-5 * abs(pe)


@pytest.mark.parametrize("world,n,overlap,options", [(4, 20000, 1, ""), (3, 15000, 2, ""), (4, 20000, 1, "exchange_p2p=1"),
                                                     (3, 15000, 1, "warmup=3")])   # murbhip_warmup: the same count of collectives on every rank
def test_long_rank_mode_run_with_asynchronous_collectives(gpu, O, tmp_path, world, n, overlap, options):
    """300 steps, every one with a reduce-scatter and an all-gather that only ENQUEUE work on the library's exchange stream
    (the stand-in's async mode): a missing dependency between the compute and exchange streams — reading positions before
    they are gathered, overwriting the send buffer before it is reduced, integrating before the sums arrive — has 600
    chances to show as a divergence from the single-GPU trajectory.  All ranks must end with bit-identical positions."""
    steps = 300
    ranks = run_ranks(tmp_path, world, n, steps, 8, overlap=overlap, mode="async", options=options)
    s = O.init_bodies(n, "galaxy")
    with gpu.Simulation(n, soft=SOFT) as one:
        one.upload(s)
        one.steps(DT, steps); one.sync()
        ref = one.state()
    scale = max(np.abs(ref[k]).max() for k in ("qx", "qy", "qz"))
    for d in ranks:
        for k in ("qx", "qy", "qz"):
            assert np.isfinite(d[k]).all()
            assert np.abs(d[k] - ref[k]).max() <= 2e-5 * scale, k      # two fp32 summation orders, 300 steps apart
    for d in ranks[1:]:
        for k in ("qx", "qy", "qz"):
            assert np.array_equal(d[k].view(np.uint32), ranks[0][k].view(np.uint32))


def test_ranks_leapfrog(gpu, O, tmp_path):
    """The closing half kick on read-out is a collective in rank mode (one more force evaluation on every rank)."""
    world, n, steps = 3, 9000, 4
    ranks = run_ranks(tmp_path, world, n, steps, 8, integrator=1)
    ref = O.init_bodies(n, "galaxy")
    O.leapfrog(ref, steps, SOFT, DT)
    scale = max(np.abs(ref[k]).max() for k in ("qx", "qy", "qz"))
    vscale = max(np.abs(ref[k]).max() for k in ("vx", "vy", "vz"))
    for d in ranks:
        f, c = int(d["first"]), int(d["count"])
        for k in ("qx", "qy", "qz"):
            assert np.abs(d[k] - ref[k]).max() <= 2e-6 * scale
        for k in ("vx", "vy", "vz"):
            assert np.abs(d[k][f:f + c] - ref[k][f:f + c]).max() <= 2e-5 * vscale


@pytest.mark.parametrize("world,n", [(2, 40000), (4, 120000)])
def test_bench_py_with_several_ranks(gpu, world, n):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one process per rank), with
    the ranks sharing GPU 0: its own process group on gloo and the library's collectives on the stand-in
    (real RCCL refuses both on one device).  Checks the N > 1 flow the driver depends on: unique-id broadcast,
    rank-mode context per rank, max-over-ranks timing, ONE JSON line from rank 0, and the run's own check of
    the multi-rank result against a single-GPU run."""
    import json
    if not os.path.exists(MOCK):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "helpers")], check=True, timeout=600)
    env = dict(os.environ, MURBHIP_RCCL_LIBRARY=MOCK, MURB_BENCH_BACKEND="gloo", MURB_BENCH_SHARE_GPU="1",
               MURB_BENCH_OTHER_CONFIGS="30000:10,60000:5",   # stand-ins for BASELINE's other sizes (the ranks share one GPU here)
               MURB_BENCH_UNTIMED_SCALE="0.1",                # ... and every step pays a host-staged collective: fewer untimed steps
               MURB_BENCH_KEEP_PLAN="1")                      # ... which makes the reduce-scatter dear: the tuner would (rightly) switch plans
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)               # bench.py must set it itself in this launch path
    if world > 2:   # the one-process diagnostic runs in a child of rank 0: with 4 ranks, the launcher and this test that is one
        env["MURB_BENCH_NO_ONE_PROCESS"] = "1"                # process more on GPU 0 than the test box allows (6)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--bodies",
           str(n), "--steps", "10", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["config"]["kernel_variant"] == 8
    assert abs(d["value"] - float(n) ** 2 * 10 / (d["ms_per_step"] * 10e-3)) / d["value"] < 1e-6
    chk = d["rank_mode_check"]
    assert chk["positions_identical_on_all_ranks"] and chk["finite"] and chk["max_position_diff_rel"] < 1e-5, chk
    assert "cpu_baseline" not in d      # rank 0 at N = 1 only
    assert d["tuned"]["tri_first_pct"] in (0, 25, 50, 75, 100) and len(d["tuned"]["ms_per_step_by_candidate"]) == 5
    assert d["tuned"]["cu_reserve"] in (0, 8, 16) and len(d["tuned"]["ms_per_step_by_cu_reserve"]) == 3
    assert "half-ring" in d["config"]["parallelism"] and "plan" not in d["tuned"]     # MURB_BENCH_KEEP_PLAN
    # a first real multi-GPU run must explain itself: both collectives and the compute stream's waits for them, event-timed
    # on rank 0's streams; the one-sided (all-gather only) plan timed beside the half-ring one; the other sizes
    ex = d["exchange"]
    # three force launches per step, two when the tuned split leaves one part of the own-slice triangle empty
    assert ex["steps_profiled"] == 10 and abs(ex["launches_per_step"] - (2 if d["tuned"]["tri_first_pct"] in (0, 100) else 3)) < 1e-9
    for k in ("reduce_scatter_ms_avg", "all_gather_ms_avg", "compute_stream_step_ms_avg", "ms_per_step_with_profiling"):
        assert ex[k] > 0, (k, ex)
    for k in ("compute_wait_gather_ms_avg", "compute_wait_reduce_ms_avg"):
        assert 0 <= ex[k] <= ex["compute_stream_step_ms_avg"], (k, ex)
    assert abs(ex["compute_wait_ms_avg"] - ex["compute_wait_gather_ms_avg"] - ex["compute_wait_reduce_ms_avg"]) < 1e-6
    assert sum(v > 0 for v in ex["force_ms_avg"].values()) >= 2 and ex["force_ms_avg"]["rectangles"] > 0
    assert ex["payload_bytes_per_rank"]["all_gather_out"] == world * ex["payload_bytes_per_rank"]["all_gather_in"]
    alt = d["one_sided_plan"]
    assert alt["ms_per_step"] > 0 and abs(alt["half_ring_speedup"] - alt["ms_per_step"] / d["ms_per_step"]) < 1e-9
    p2p = d["p2p_plan"]                # both exchanges as grouped sends / receives, timed beside the collectives
    assert p2p["ms_per_step"] > 0 and abs(p2p["speedup_over_collectives"] - p2p["collectives_ms_per_step_same_context"] / p2p["ms_per_step"]) < 1e-9
    if world <= 2:
        one = d["one_process_plan"]     # the drop-in form: ONE process (a child of rank 0) drives all shards, both exchanges
        for ex in ("copy", "rccl"):
            assert one[ex]["ms_per_step"] > 0 and one[ex]["ms_per_step_sync_each_iteration"] > 0 and one[ex]["kernel_variant"] == 8, one
            assert one["vs_one_process_per_gpu"][ex] > 0
    else:
        assert "one_process_plan" not in d
    oc = {e["n_bodies"]: e for e in d["other_configs"]}
    assert set(oc) == {30000, 60000}
    for e in oc.values():
        assert (e["other_plan"]["kernel_variant"] in (1, 2) if e["plan"]["kernel_variant"] == 8 else e["other_plan"]["kernel_variant"] == 8) and e["other_plan"]["ms_per_step"] > 0
        assert e["exchange"]["all_gather_ms_avg"] > 0 and (e["exchange"]["reduce_scatter_ms_avg"] > 0) == (e["plan"]["kernel_variant"] == 8)
        assert e["value"] > 0 and e["roofline"]["kernel_ms_avg"] > 0 and abs(e["value"] - float(e["n_bodies"]) ** 2 * e["steps"] / (e["ms_per_step"] * e["steps"] * 1e-3)) / e["value"] < 1e-6
    assert oc[60000]["plan"]["kernel_variant"] == 8 or world == 4     # a rank of 4 at 60 000 bodies falls back to the one-sided plan
    assert "environment" not in d


def test_bench_py_launches_its_own_ranks(gpu):
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run around it: the script must start the launcher itself (as a
    child process, before touching the GPU) and relay the one JSON line — the other launch convention a driver may use."""
    import json
    if not os.path.exists(MOCK):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "helpers")], check=True, timeout=600)
    env = dict(os.environ, MURBHIP_RCCL_LIBRARY=MOCK, MURB_BENCH_BACKEND="gloo", MURB_BENCH_SHARE_GPU="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bodies", "40000", "--steps", "6",
                        "--warmup", "1", "--cu-reserve", "8", "--no-other-configs"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["tuned"]["cu_reserve"] == 8
    chk = d["rank_mode_check"]
    assert chk["positions_identical_on_all_ranks"] and chk["finite"] and chk["max_position_diff_rel"] < 1e-5, chk


def test_bench_py_watchdog_reports_a_stuck_phase(gpu):
    """N > 1: a phase of bench.py that makes no progress within its limit (here: a limit no phase can meet) must end the
    run with ONE JSON line carrying an "error" key from rank 0 and a non-zero exit code — not a silent hang."""
    import json
    env = dict(os.environ, MURBHIP_RCCL_LIBRARY=MOCK, MURB_BENCH_BACKEND="gloo", MURB_BENCH_SHARE_GPU="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", MURB_BENCH_PHASE_LIMIT_S="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bodies", "40000", "--steps", "5",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["value"] is None and "no progress in phase" in d["error"] and d["phase"]
    assert "giving up" in r.stderr
    # the line names the settings a failed multi-GPU start depends on
    assert d["environment"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and d["environment"]["MURBHIP_RCCL_LIBRARY"] == MOCK


def test_bench_py_keeps_the_measurement_when_an_extra_fails(gpu):
    """N > 1: a failure AFTER the main measurement (here injected into the first of the other sizes) must not cost the
    measurement: rank 0 prints the line it has, with the failure under "incomplete", and the run exits 0."""
    import json
    env = dict(os.environ, MURBHIP_RCCL_LIBRARY=MOCK, MURB_BENCH_BACKEND="gloo", MURB_BENCH_SHARE_GPU="1", MURB_BENCH_UNTIMED_SCALE="0.1",
               MURB_BENCH_OTHER_CONFIGS="30000:10", MURB_BENCH_FAIL_IN="other config")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bodies", "40000", "--steps", "6", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_gpus"] == 2 and d["rank_mode_check"]["positions_identical_on_all_ranks"] and "exchange" in d
    assert "injected failure" in d["incomplete"]["error"] and d["incomplete"]["phase"].startswith("other config")
    assert r.returncode == 0, r.stderr[-1500:]


def test_bench_py_takes_the_one_sided_plan_when_it_wins(gpu):
    """N > 1: where the all-gather-only plan beats the half-ring schedule in the untimed tuning steps (a node whose
    reduce-scatter costs more than it saves) the timed region runs it.  Forced here: the line must then describe THAT plan
    (one-sided kernel, no reduce-scatter span), carry the forced half-ring schedule as the other plan, and still pass its check."""
    import json
    env = dict(os.environ, MURBHIP_RCCL_LIBRARY=MOCK, MURB_BENCH_BACKEND="gloo", MURB_BENCH_SHARE_GPU="1", MURB_BENCH_UNTIMED_SCALE="0.1",
               MURB_BENCH_OTHER_CONFIGS="30000:10", MURB_BENCH_TAKE_ONE_SIDED="1", MURB_BENCH_NO_ONE_PROCESS="1", MURB_BENCH_NO_P2P="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bodies", "40000", "--steps", "6", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["config"]["kernel_variant"] == 1 and "one-sided" in d["config"]["parallelism"]   # 20 000 bodies per rank: 8 per wave
    assert d["tuned"]["plan"].startswith("one-sided") and d["tuned"]["ms_per_step_one_sided_plan"] > 0 and d["tuned"]["ms_per_step_half_ring_plan"] > 0
    assert d["rank_mode_check"]["positions_identical_on_all_ranks"] and d["rank_mode_check"]["max_position_diff_rel"] < 1e-5
    assert d["exchange"]["all_gather_ms_avg"] > 0 and d["exchange"]["plan"].startswith("one-sided")
    assert d["one_sided_plan"]["kernel_variant"] == 8 and d["one_sided_plan"]["ms_per_step"] > 0
    assert d["roofline"]["frac"] < 1.0 and "incomplete" not in d


def test_bench_py_keeps_the_measurement_when_rank_0_is_killed(gpu):
    """N > 1: rank 0 ended the hard way (abort() inside a library, a GPU fault: nothing Python can catch) during a diagnostic
    AFTER the main measurement.  The line it had sent to its keeper process by then still reaches stdout — ONE line, with the
    measurement and an "incomplete" note; the exit code says that the run did not end well."""
    import json
    env = dict(os.environ, MURBHIP_RCCL_LIBRARY=MOCK, MURB_BENCH_BACKEND="gloo", MURB_BENCH_SHARE_GPU="1", MURB_BENCH_UNTIMED_SCALE="0.1",
               MURB_BENCH_OTHER_CONFIGS="30000:10", MURB_BENCH_ABORT_IN="other config", MURB_BENCH_PHASE_LIMIT_S="60")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bodies", "40000", "--steps", "6", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_gpus"] == 2 and d["rank_mode_check"]["positions_identical_on_all_ranks"] and "exchange" in d
    assert "process ended before the diagnostics" in d["incomplete"]["error"]
    assert r.returncode != 0


def test_bench_py_reports_a_failed_start(gpu):
    """N > 1 with a collective library that cannot be loaded: the run must end at once with ONE JSON line from rank 0 whose
    "error" carries the library's message and the phase, and a non-zero exit code."""
    import json
    env = dict(os.environ, MURBHIP_RCCL_LIBRARY="none", MURB_BENCH_BACKEND="gloo", MURB_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bodies", "40000", "--steps", "5",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["value"] is None and "librccl could not be loaded" in d["error"] and d["phase"] == "context + upload"


@pytest.mark.parametrize("shards,n,variant,overlap,options", [(2, 9000, 8, 1, ""), (3, 9001, 8, 0, ""), (4, 20000, 8, 1, ""), (3, 9000, 1, 1, ""),
                                                              (4, 20000, 8, 1, "exchange_p2p=1"), (5, 30000, 8, 1, "exchange_p2p=1"),
                                                              (3, 9000, 1, 1, "exchange_p2p=1")])
def test_one_process_several_shards_over_rccl_calls(gpu, shards, n, variant, overlap, options):
    """`--im hip+tile+multi` on a multi-GPU node = murbhip_create_sharded(..., exchange = RCCL): ncclCommInitAll, then
    every shard's own host thread (the library's ShardCrew) issues that shard's all-gather and reduce-scatter on its own
    communicator — no ncclGroupStart/End.  With all shards on GPU 0 the calls go to the stand-in library, whose local mode
    makes the callers' threads meet inside every collective and moves the data with copies and host functions on the
    callers' streams (asynchronous towards the GPU, like RCCL)."""
    if not os.path.exists(MOCK):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "helpers")], check=True, timeout=600)
    env = dict(os.environ, MURBHIP_RCCL_LIBRARY=MOCK, MURB_TEST_OPTIONS=options)   # "exchange_p2p=1": grouped sends / receives per shard thread
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_sharded_rccl_worker.py"), str(shards), str(n), str(variant),
                        str(overlap)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-1500:])
