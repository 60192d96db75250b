"""TEST INFRASTRUCTURE — ctypes front end of the CPU oracle (oracle/_build/liboracle.so) and, when it
has been built, of the real reference CPU path (oracle/_ref/libmurbref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product (nbody-eurohpc_amd/) never does: it fails loudly when its HIP library is missing instead of
falling back to anything in here.

Reference citations (relative to /root/reference) are in murb_oracle.cpp / oracle_f64.cpp / ref_harness.cpp.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libmurbref.so")
REF_AVX2_SO = os.path.join(HERE, "_ref", "libmurbref_avx2.so")   # same sources + -mavx2 -mfma: speed baseline only
REF_AVX512_SO = os.path.join(HERE, "_ref", "libmurbref_avx512.so")   # ... + the AVX-512 subsets: speed baseline only
AVX512_FLAGS = ("avx512f", "avx512dq", "avx512bw", "avx512vl")

G = np.float32(6.67384e-11)      # SimulationNBodyInterface.hpp:18
SOFT = np.float32(2e8)           # main.cpp:47
DT = np.float32(3600.0)          # main.cpp:45
REF_SIMD_WIDTH = 4               # mipp::N<float>() of the reference build (no -march => SSE2)

_f = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_d = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ul = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
FIELDS = ("qx", "qy", "qz", "vx", "vy", "vz", "m", "r")


def build(ref=True):
    """Compile the oracle (and, if /root/reference is present, the reference checker)."""
    subprocess.run(["make", "-C", HERE, "all"] + (["ref"] if ref else []), check=True,
                   stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=os.path.isdir("/root/reference/src"))
        L = C.CDLL(ORACLE_SO)
        L.oracle_padding.restype = C.c_ulong
        L.oracle_padding.argtypes = [C.c_ulong, C.c_int]
        L.oracle_init.restype = None
        L.oracle_init.argtypes = [C.c_ulong, C.c_char_p, C.c_ulong, C.c_int] + [_f] * 8
        for name in ("oracle_accel_optim", "oracle_accel_naive"):
            fn = getattr(L, name)
            fn.restype = None
            fn.argtypes = [C.c_ulong, _f, _f, _f, _f, C.c_float, _f, _f, _f]
        L.oracle_integrate.restype = None
        L.oracle_integrate.argtypes = [C.c_ulong] + [_f] * 9 + [C.c_float]
        L.oracle_simulate.restype = None
        L.oracle_simulate.argtypes = [C.c_int, C.c_ulong, C.c_int, C.c_float, C.c_float] + [_f] * 7 + [C.c_void_p]
        L.oracle_leapfrog.restype = None
        L.oracle_leapfrog.argtypes = [C.c_ulong, C.c_int, C.c_float, C.c_float] + [_f] * 7
        L.oracle_accel_slice_f32.restype = None
        L.oracle_accel_slice_f32.argtypes = [C.c_ulong, C.c_ulong, C.c_ulong, _f, _f, _f, _f, C.c_float, _f, _f, _f]
        L.oracle_accel_f64.restype = None
        L.oracle_accel_f64.argtypes = [C.c_ulong, C.c_ulong, C.c_ulong, _f, _f, _f, _f, C.c_float, _d, _d, _d]
        L.oracle_accel_f64_subset.restype = None
        L.oracle_accel_f64_subset.argtypes = [C.c_ulong, C.c_ulong, _ul, _f, _f, _f, _f, C.c_float, _d, _d, _d]
        L.oracle_energy_f64.restype = None
        L.oracle_energy_f64.argtypes = [C.c_ulong] + [_f] * 7 + [C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def energy_f64(s, soft=SOFT):
    """(kinetic, potential) in fp64, definitions of the reference's gpu+tracking metric."""
    ke, pe = C.c_double(), C.c_double()
    lib().oracle_energy_f64(len(s["qx"]), s["qx"], s["qy"], s["qz"], s["vx"], s["vy"], s["vz"], s["m"], soft,
                            C.byref(ke), C.byref(pe))
    return ke.value, pe.value


# ----------------------------------------------------------------------------- restated reference path
def padding(n, simd_width=REF_SIMD_WIDTH):
    return int(lib().oracle_padding(n, simd_width))


def init_bodies(n, scheme="galaxy", seed=0, simd_width=REF_SIMD_WIDTH, with_padding=False):
    """Initial conditions (Bodies.cpp:158-257).  Returns a dict of fp32 arrays of n entries
    (n + padding with with_padding=True)."""
    tot = n + padding(n, simd_width)
    a = {k: np.zeros(tot, np.float32) for k in FIELDS}
    lib().oracle_init(n, scheme.encode(), seed, simd_width, *[a[k] for k in FIELDS])
    return a if with_padding else {k: v[:n].copy() for k, v in a.items()}


def accel_optim(s, soft=SOFT):
    n = len(s["qx"])
    ax, ay, az = (np.zeros(n, np.float32) for _ in range(3))
    lib().oracle_accel_optim(n, s["qx"], s["qy"], s["qz"], s["m"], soft, ax, ay, az)
    return ax, ay, az


def accel_naive(s, soft=SOFT):
    n = len(s["qx"])
    ax, ay, az = (np.zeros(n, np.float32) for _ in range(3))
    lib().oracle_accel_naive(n, s["qx"], s["qy"], s["qz"], s["m"], soft, ax, ay, az)
    return ax, ay, az


def accel_slice_f32(s, i0, i1, soft=SOFT):
    n = len(s["qx"])
    ax, ay, az = (np.zeros(i1 - i0, np.float32) for _ in range(3))
    lib().oracle_accel_slice_f32(n, i0, i1, s["qx"], s["qy"], s["qz"], s["m"], soft, ax, ay, az)
    return ax, ay, az


def accel_f64(s, soft=SOFT, i0=0, i1=None):
    n = len(s["qx"])
    i1 = n if i1 is None else i1
    ax, ay, az = (np.zeros(i1 - i0, np.float64) for _ in range(3))
    lib().oracle_accel_f64(n, i0, i1, s["qx"], s["qy"], s["qz"], s["m"], soft, ax, ay, az)
    return ax, ay, az


def accel_f64_subset(s, idx, soft=SOFT):
    n = len(s["qx"])
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    ax, ay, az = (np.zeros(len(idx), np.float64) for _ in range(3))
    lib().oracle_accel_f64_subset(n, len(idx), idx, s["qx"], s["qy"], s["qz"], s["m"], soft, ax, ay, az)
    return ax, ay, az


def integrate(s, acc, dt=DT):
    """In-place position/velocity update (Bodies.cpp:260-278)."""
    ax, ay, az = (np.ascontiguousarray(a, np.float32) for a in acc)
    lib().oracle_integrate(len(s["qx"]), s["qx"], s["qy"], s["qz"], s["vx"], s["vy"], s["vz"], ax, ay, az, dt)


def simulate(s, iterations, variant="cpu+optim", soft=SOFT, dt=DT):
    """`iterations` whole steps in place; returns the last step's accelerations."""
    n = len(s["qx"])
    acc = np.zeros(3 * n, np.float32)
    lib().oracle_simulate({"cpu+optim": 0, "cpu+naive": 1}[variant], n, iterations, soft, dt, s["qx"], s["qy"],
                          s["qz"], s["vx"], s["vy"], s["vz"], s["m"], acc.ctypes.data_as(C.c_void_p))
    return acc[:n], acc[n:2 * n], acc[2 * n:]


def leapfrog(s, iterations, soft=SOFT, dt=DT):
    """`iterations` kick-drift-kick steps in place, velocities synchronised at the end (parity unpinned:
    see murb_oracle.cpp)."""
    lib().oracle_leapfrog(len(s["qx"]), iterations, soft, dt, s["qx"], s["qy"], s["qz"], s["vx"], s["vy"], s["vz"], s["m"])


def moments_f64(s):
    """P = sum m v, L = sum m q x v, Mq = sum m q, M: fp64 (what murbhip_moments returns)."""
    m = s["m"].astype(np.float64)
    q = np.stack([s[k].astype(np.float64) for k in ("qx", "qy", "qz")])
    v = np.stack([s[k].astype(np.float64) for k in ("vx", "vy", "vz")])
    return {"P": (m * v).sum(1), "L": (m * np.cross(q.T, v.T).T).sum(1), "Mq": (m * q).sum(1), "M": float(m.sum())}


def rel_err(test, ref):
    """Per-body relative vector error |a_test - a_ref| / |a_ref| (the metric of SURVEY.md §8c)."""
    t = np.stack([np.asarray(c, np.float64) for c in test])
    r = np.stack([np.asarray(c, np.float64) for c in ref])
    num = np.sqrt(((t - r) ** 2).sum(0))
    den = np.sqrt((r ** 2).sum(0))
    return num / np.maximum(den, np.finfo(np.float64).tiny)


# ----------------------------------------------------------------------------- the real reference (checker only)
_ref = {}


def _ref_path(avx2=False, isa=None):
    isa = isa or ("avx2" if avx2 else None)
    return {None: REF_SO, "sse2": REF_SO, "avx2": REF_AVX2_SO, "avx512": REF_AVX512_SO}[isa]


def cpu_has(flags):
    """True when /proc/cpuinfo lists every one of `flags` (an AVX-512 build must not be loaded on a CPU without them)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                have = set(line.split(":", 1)[1].split())
                return all(f in have for f in flags)
    except OSError:
        pass
    return False


def have_ref(avx2=False, isa=None):
    """The reference CPU build of that instruction set exists AND this host can run it."""
    isa = isa or ("avx2" if avx2 else None)
    if isa == "avx512" and not cpu_has(AVX512_FLAGS):
        return False
    if isa == "avx2" and not cpu_has(("avx2", "fma")):
        return False
    return os.path.exists(_ref_path(isa=isa))


def ref_lib(avx2=False, isa=None):
    path = _ref_path(avx2, isa)
    if path not in _ref:
        if not os.path.exists(path):
            raise FileNotFoundError(path + " not built (make -C oracle ref needs /root/reference)")
        L = C.CDLL(path)
        L.murbref_create.restype = C.c_void_p
        L.murbref_create.argtypes = [C.c_char_p, C.c_ulong, C.c_char_p, C.c_float, C.c_float]
        L.murbref_destroy.argtypes = [C.c_void_p]
        L.murbref_n.restype = C.c_ulong
        L.murbref_n.argtypes = [C.c_void_p]
        L.murbref_padding.restype = C.c_ulong
        L.murbref_padding.argtypes = [C.c_void_p]
        L.murbref_flops_per_ite.restype = C.c_float
        L.murbref_flops_per_ite.argtypes = [C.c_void_p]
        L.murbref_allocated_bytes.restype = C.c_float
        L.murbref_allocated_bytes.argtypes = [C.c_void_p]
        L.murbref_step.argtypes = [C.c_void_p, C.c_int]
        L.murbref_get_state.argtypes = [C.c_void_p] + [_f] * 8
        L.murbref_get_acc.restype = C.c_int
        L.murbref_get_acc.argtypes = [C.c_void_p, _f, _f, _f]
        L.murbref_integrate.argtypes = [C.c_ulong, C.c_char_p, _f, _f, _f, C.c_float, C.c_int] + [_f] * 6
        _ref[path] = L
    return _ref[path]


class RefSim:
    """One of the reference's own CPU implementations (--im cpu+naive|cpu+optim|cpu+simd|cpu+omp)."""

    def __init__(self, tag, n, scheme="galaxy", soft=SOFT, dt=DT, avx2=False, isa=None):
        self.L = ref_lib(avx2, isa)
        self.h = self.L.murbref_create(tag.encode(), n, scheme.encode(), soft, dt)
        if not self.h:
            raise ValueError("unknown reference implementation tag " + tag)
        self.n = int(self.L.murbref_n(self.h))
        self.padding = int(self.L.murbref_padding(self.h))

    def step(self, iterations=1):
        self.L.murbref_step(self.h, iterations)

    def state(self, with_padding=False):
        tot = self.n + self.padding
        a = {k: np.zeros(tot, np.float32) for k in FIELDS}
        self.L.murbref_get_state(self.h, *[a[k] for k in FIELDS])
        return a if with_padding else {k: v[:self.n].copy() for k, v in a.items()}

    def acc(self):
        ax, ay, az = (np.zeros(self.n, np.float32) for _ in range(3))
        if self.L.murbref_get_acc(self.h, ax, ay, az) != 0:
            raise RuntimeError("implementation exposes no accelerations")
        return ax, ay, az

    def flops_per_ite(self):
        return float(self.L.murbref_flops_per_ite(self.h))

    def allocated_bytes(self):
        return float(self.L.murbref_allocated_bytes(self.h))

    def close(self):
        if self.h:
            self.L.murbref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ref_integrate(n, scheme, acc, dt, steps):
    """Reference Bodies::updatePositionsAndVelocities driven with given accelerations."""
    out = {k: np.zeros(n, np.float32) for k in FIELDS[:6]}
    ax, ay, az = (np.ascontiguousarray(a, np.float32) for a in acc)
    ref_lib().murbref_integrate(n, scheme.encode(), ax, ay, az, dt, steps, *[out[k] for k in FIELDS[:6]])
    return out


# ----------------------------------------------------------------------------------------------------------------------
# Initial conditions restated in numpy with every rounding spelled out (checker for csrc/murb_init.h, the on-device
# initialisation).  rand() comes from this host's libc; sincosf is glibc 2.35's algorithm (sysdeps/ieee754/flt-32/
# s_sincosf.{c,h}, the ARM optimized-routines sincosf) in its NON-FMA (SSE2) build: numpy rounds every product and sum on
# its own, which is exactly that build.  (The -mfma build glibc selects on CPUs with FMA and AVX2 differs from it in the
# last bit of a few results per million; the device's rendering of that one is checked against the host's libm itself.)
def glibc_rand_draws(seed, count):
    """The first `count` values of rand() after srand(seed), from this process's libc."""
    libc = C.CDLL(None)
    libc.rand.restype = C.c_int
    libc.srand(C.c_uint(seed))
    return np.fromiter((libc.rand() for _ in range(count)), dtype=np.int64, count=count)


def sincosf_glibc_sse2(y):
    """(sin, cos) of float32 values |y| < 120 by glibc's sincosf, SSE2 build: float64 arithmetic, one rounding per operation."""
    y = np.asarray(y, np.float32)
    x = y.astype(np.float64)
    top = (y.view(np.uint32) >> 20) & 0x7FF
    assert (top < 0x42F).all()
    hpi_inv, hpi = float.fromhex("0x1.45f306dc9c883p+23"), float.fromhex("0x1.921fb54442d18p+0")
    r = x * hpi_inv
    n = ((r.astype(np.int32).astype(np.int64) + 0x800000) >> 24).astype(np.int64)
    medium = top >= 0x3F4
    n = np.where(medium, n, 0)
    xr = np.where(medium, x - n.astype(np.float64) * hpi, x)
    s = np.where(medium & (((n & 3) == 1) | ((n & 3) == 2)), -1.0, 1.0)
    sg = np.where(medium & ((n & 2) != 0), -1.0, 1.0)
    c0, c1, c2 = sg * 1.0, sg * float.fromhex("-0x1.ffffffd0c621cp-2"), sg * float.fromhex("0x1.55553e1068f19p-5")
    c3, c4 = sg * float.fromhex("-0x1.6c087e89a359dp-10"), sg * float.fromhex("0x1.99343027bf8c3p-16")
    s1, s2, s3 = float.fromhex("-0x1.555545995a603p-3"), float.fromhex("0x1.1107605230bc4p-7"), float.fromhex("-0x1.994eb3774cf24p-13")
    xx, x2 = xr * s, xr * xr
    x3, x4 = x2 * xx, x2 * x2
    c2p, s1p, c1p = x2 * c4 + c3, x2 * s3 + s2, x2 * c1 + c0
    x5, x6 = x2 * x3, x2 * x4
    sv = ((x3 * s1 + xx) + s1p * x5).astype(np.float32)
    cv = ((x4 * c2 + c1p) + c2p * x6).astype(np.float32)
    odd = (n & 1) == 1
    sin, cos = np.where(odd, cv, sv), np.where(odd, sv, cv)
    tiny = top < 0x398
    return np.where(tiny, y, sin).astype(np.float32), np.where(tiny, np.float32(1.0), cos).astype(np.float32)


def init_galaxy_spelled_out(n, seed=0):
    """Bodies::initGalaxy (Bodies.cpp:158-214) as the reference's flags compile it (operation by operation, read off the
    object code), with sincosf_glibc_sse2: what the device produces under "init_libm_fma" = 0."""
    f32, f64 = np.float32, np.float64
    y = glibc_rand_draws(seed, 4 * (n - 1)).reshape(n - 1, 4)
    k31 = f32(2.0 ** -31)
    frac_up = y[:, 0].astype(f32) * k31
    down = lambda col: (2147483647 - y[:, col]).astype(f32) * k31   # noqa: E731
    m = (frac_up.astype(f64) * 5e20).astype(f32)
    r = (m.astype(f64) * 2.5e-15).astype(f32)
    two_pi = float.fromhex("0x1.921fb54442d18p+2")
    sh, ch = sincosf_glibc_sse2((down(1).astype(f64) * two_pi).astype(f32))
    sv, cv = sincosf_glibc_sse2((down(2).astype(f64) * two_pi).astype(f32))
    dist = ((down(3).astype(f64) + 1.0) * 1.0e8).astype(f32)
    qx, qy, qz = (sh * cv) * dist, sv * dist, (cv * ch) * dist
    vx, vy = (qy.astype(f64) * 4.0e-6).astype(f32), ((-qx).astype(f64) * 4.0e-6).astype(f32)
    z = np.zeros(1, f32)
    cat = lambda first, rest: np.concatenate([np.asarray([first], f32), rest.astype(f32)])   # noqa: E731
    return {"m": cat(2.0e24, m), "r": cat(0, r), "qx": cat(0, qx), "qy": cat(0, qy), "qz": cat(0, qz), "vx": cat(0, vx), "vy": cat(0, vy),
            "vz": np.zeros(n, f32) + z}


def init_random_spelled_out(n, seed=0):
    """Bodies::initRandomly (Bodies.cpp:217-257) as compiled: mass draw, then the six draws of the box (three of them
    folded by -ffast-math into a single float multiply)."""
    f32, f64 = np.float32, np.float64
    y = glibc_rand_draws(seed, 7 * n).reshape(n, 7)
    m = ((y[:, 0].astype(f32) * f32(2.0 ** -31)).astype(f64) * 5.0e21).astype(f32)
    r = (m.astype(f64) * 0.5e-14).astype(f32)
    fc = lambda col: (y[:, col] - 0x3FFFFFFF).astype(f32)   # noqa: E731
    k = lambda bits: np.array([bits], np.uint32).view(f32)[0]   # noqa: E731
    kv = k(0x33C80000)
    return {"m": m, "r": r, "qx": fc(1) * k(0x3F1E8C61), "qy": fc(2) * k(0x3EEE6B28),
            "qz": ((fc(3) * f32(2.0 ** -30)).astype(f64) * 5.0e8 - 1.0e9).astype(f32),
            "vx": fc(4) * kv, "vy": fc(5) * kv, "vz": fc(6) * kv}
