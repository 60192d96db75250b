"""The drop-in claim of INTEGRATION.md §A, guarded: the product's plugin files compile UNCHANGED against the
reference's own headers (Bodies.hpp:79-225, SimulationNBodyInterface.hpp:16-88, BodiesAllocator.hpp:11-15), link with
the reference's own core sources, and (on the GPU) pass the reference's own hot-path test with the HIP target.

CPU part: needs the reference tree (build container only; skipped elsewhere).  GPU part: runs the binary the CPU part's
recipe (oracle/Makefile, target _ref/murb_dropin_test) left under oracle/_ref/, which travels to the GPU box."""
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"
PKG = os.path.join(ROOT, "nbody-eurohpc_amd")
BIN = os.path.join(ROOT, "oracle", "_ref", "murb_dropin_test")
PLUGIN_TUS = ["host/core/HIPBodies.cpp", "host/implem/SimulationNBodyHIP.cpp", "host/implem/SimulationNBodyHIPTracking.cpp"]

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")


@needs_ref
@pytest.mark.parametrize("tu", PLUGIN_TUS)
def test_plugin_tu_compiles_against_reference_headers(tu):
    """g++ with the reference's include directories FIRST: every "core/…" header the reference owns must come from
    /root/reference, the product's only from host/ (HIPBodies, SimulationNBodyHIP, SimulationHistory mirror)."""
    r = subprocess.run(["g++", "-std=c++20", "-O1", "-fsyntax-only", "-H", f"-I{REF}/src/common", f"-I{REF}/lib/MIPP/src",
                        f"-I{PKG}/host", f"-I{ROOT}/include", os.path.join(PKG, tu)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    used = [l.strip(". ").strip() for l in r.stderr.splitlines() if l.startswith(".")]
    for name, always in (("core/Bodies.hpp", True), ("core/SimulationNBodyInterface.hpp", "implem/" in tu)):
        hits = [u for u in used if u.endswith(name)]
        assert (hits or not always) and all(u.startswith(REF) for u in hits), (name, hits)
    assert not any(u.startswith(PKG) and u.endswith(("core/Bodies.hpp", "core/SimulationNBodyInterface.hpp", "core/BodiesAllocator.hpp"))
                   for u in used)


@needs_ref
def test_dropin_binary_builds_and_refuses_to_run_without_a_gpu():
    """Reference core + cpu+naive sources, plugin files, harness -> one executable.  Without a device it must fail
    loudly (there is no CPU fallback behind SimulationNBodyHIP)."""
    import murbhip
    murbhip.lib()   # libmurbhip.so must exist for the link line
    r = subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "oracle"), "_ref/murb_dropin_test"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and os.path.exists(BIN), r.stderr[-3000:]
    if murbhip.device_count() == 0:
        run = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
        assert run.returncode != 0 and "no usable HIP device" in run.stderr, (run.returncode, run.stderr[-500:])


@pytest.mark.gpu
def test_reference_test_sections_inside_the_reference_tree(gpu):
    """The four sections of test_SimulationNBody.cpp:73-82, reference cpu+naive (compiled from the reference's sources)
    against SimulationNBodyHIP behind the reference's own SimulationNBodyInterface / Bodies."""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/murb_dropin_test was not built (no reference tree in the build container)")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("dropin ok"), (r.stdout[-1500:], r.stderr[-1500:])
    assert r.stdout.count("0 outside") == 4, r.stdout
