"""ctypes binding of libmurbhip.so (include/murbhip.h) for tests/ and bench.py.

This is plumbing only: the product's host side is the C++ mirror of the reference's plugin
interface in nbody-eurohpc_amd/host/ (the reference is compiled C++).  There is no CPU fallback:
importing works without a GPU (the library loads and the host-only helpers run), but every compute
entry point raises MurbHipError when no MI355X is present or the library is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MURBHIP_LIBRARY") or os.path.join(_HERE, "..", "lib", "libmurbhip.so")   # override: lab builds
G = np.float32(6.67384e-11)   # reference SimulationNBodyInterface.hpp:18

_fp = C.POINTER(C.c_float)


class MurbHipError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(f"{what}: {error_string(code)} (code {code})")


_lib = None


def lib():
    """The loaded library; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        path = os.path.normpath(LIB_PATH)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: build it with `make -C nbody-eurohpc_amd` "
                                    "(there is no CPU fallback for the HIP path)")
        L = C.CDLL(path)
        L.murbhip_version.restype = C.c_int
        L.murbhip_error_string.restype = C.c_char_p
        L.murbhip_error_string.argtypes = [C.c_int]
        L.murbhip_partition.argtypes = [C.c_ulong, C.c_int, C.c_int, C.POINTER(C.c_ulong), C.POINTER(C.c_ulong)]
        L.murbhip_slice_slots.restype = C.c_ulong
        L.murbhip_slice_slots.argtypes = [C.c_ulong, C.c_int]
        L.murbhip_slot_of_body.restype = C.c_ulong
        L.murbhip_slot_of_body.argtypes = [C.c_ulong, C.c_int, C.c_ulong]
        L.murbhip_schedule_items.argtypes = [C.c_ulong, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_ulong,
                                             C.POINTER(C.c_ulong), C.POINTER(C.c_ulong)]
        L.murbhip_schedule_layout.argtypes = [C.c_ulong] + [C.c_int] * 7 + [C.POINTER(C.c_long), C.c_ulong, C.POINTER(C.c_ulong),
                                                                            C.POINTER(C.c_long), C.c_ulong, C.POINTER(C.c_ulong),
                                                                            C.POINTER(C.c_ulong), C.POINTER(C.c_ulong)]
        L.murbhip_device_count.argtypes = [C.POINTER(C.c_int)]
        L.murbhip_create.argtypes = [C.POINTER(C.c_void_p), C.c_ulong, C.c_float, C.c_float, C.c_int]
        L.murbhip_create_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_ulong, C.c_float, C.c_float, C.c_int,
                                             C.POINTER(C.c_int), C.c_int]
        L.murbhip_unique_id.argtypes = [C.c_void_p]
        L.murbhip_create_rank.argtypes = [C.POINTER(C.c_void_p), C.c_ulong, C.c_float, C.c_float, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p]
        L.murbhip_destroy.argtypes = [C.c_void_p]
        L.murbhip_upload.argtypes = [C.c_void_p] + [_fp] * 7
        L.murbhip_init_bodies.argtypes = [C.c_void_p, C.c_char_p, C.c_ulong]
        L.murbhip_download_mass.argtypes = [C.c_void_p, _fp, _fp]
        L.murbhip_download_state.argtypes = [C.c_void_p] + [_fp] * 6
        L.murbhip_download_acc.argtypes = [C.c_void_p] + [_fp] * 3
        L.murbhip_compute_acc.argtypes = [C.c_void_p]
        L.murbhip_warmup.argtypes = [C.c_void_p, C.c_double]
        L.murbhip_step.argtypes = [C.c_void_p, C.c_float]
        L.murbhip_steps.argtypes = [C.c_void_p, C.c_float, C.c_int]
        L.murbhip_integrate_host_acc.argtypes = [C.c_void_p] + [_fp] * 3 + [C.c_float]
        L.murbhip_sync.argtypes = [C.c_void_p]
        L.murbhip_energy.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.murbhip_moments.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.murbhip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
        L.murbhip_get_info.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
        _lib = L
    return _lib


EXPORTS = ("murbhip_version murbhip_error_string murbhip_partition murbhip_slice_slots murbhip_slot_of_body "
           "murbhip_schedule_items murbhip_schedule_layout "
           "murbhip_device_count murbhip_create murbhip_create_sharded murbhip_unique_id murbhip_create_rank "
           "murbhip_destroy murbhip_upload murbhip_init_bodies murbhip_download_mass murbhip_download_state murbhip_download_acc murbhip_compute_acc "
           "murbhip_warmup murbhip_step murbhip_steps murbhip_integrate_host_acc murbhip_sync murbhip_energy murbhip_moments murbhip_set_option "
           "murbhip_get_info").split()


def error_string(code):
    return lib().murbhip_error_string(code).decode()


def _check(code, what):
    if code != 0:
        raise MurbHipError(code, what)


def _ptr(a):
    return a.ctypes.data_as(_fp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ host-only helpers
def partition(n, world, rank):
    first, count = C.c_ulong(), C.c_ulong()
    _check(lib().murbhip_partition(n, world, rank, C.byref(first), C.byref(count)), "murbhip_partition")
    return first.value, count.value


def slice_slots(n, world):
    return lib().murbhip_slice_slots(n, world)


def slot_of_body(n, world, i):
    return lib().murbhip_slot_of_body(n, world, i)


def schedule_items(n, world, rank, split=1):
    """(items, own_count): the half-ring work list of `rank`; items is an (count, 2) int array of
    (i-side sub-block, j-side block)."""
    count, own = C.c_ulong(), C.c_ulong()
    _check(lib().murbhip_schedule_items(n, world, rank, split, None, 0, C.byref(count), C.byref(own)),
           "murbhip_schedule_items")
    items = np.zeros((count.value, 2), np.int32)
    _check(lib().murbhip_schedule_items(n, world, rank, split, items.ctypes.data_as(C.POINTER(C.c_int)), count.value,
                                        C.byref(count), C.byref(own)), "murbhip_schedule_items")
    return items, own.value


def schedule_layout(n, world, rank, split=1, waves=4, taper=0, tri_first_pct=50, exchange_mode=False, diag_tri=False, tri_div=1):
    """(items, rows, floats_main, floats_tri): the pair-symmetric work list with its partial-row layout, as arrays of
    8 longs per item and 7 per row-table entry (include/murbhip.h: murbhip_schedule_layout)."""
    ni, nr, fm, ft = C.c_ulong(), C.c_ulong(), C.c_ulong(), C.c_ulong()
    args = (n, world, rank, split, waves, taper + (256 if diag_tri else 0) + 512 * {1: 0, 2: 1, 4: 2, 8: 3}[tri_div], tri_first_pct,
            int(exchange_mode))
    _check(lib().murbhip_schedule_layout(*args, None, 0, C.byref(ni), None, 0, C.byref(nr), C.byref(fm), C.byref(ft)),
           "murbhip_schedule_layout")
    items = np.zeros((ni.value, 8), np.int64)
    rows = np.zeros((nr.value, 7), np.int64)
    lp = C.POINTER(C.c_long)
    _check(lib().murbhip_schedule_layout(*args, items.ctypes.data_as(lp), ni.value, C.byref(ni), rows.ctypes.data_as(lp), nr.value,
                                         C.byref(nr), C.byref(fm), C.byref(ft)), "murbhip_schedule_layout")
    return items, rows, fm.value, ft.value


def device_count():
    c = C.c_int(0)
    rc = lib().murbhip_device_count(C.byref(c))
    return c.value if rc == 0 else 0


def unique_id():
    buf = C.create_string_buffer(128)
    _check(lib().murbhip_unique_id(buf), "murbhip_unique_id")
    return buf.raw


class Simulation:
    """One device-resident n-body state.  mode: single GPU (default), `devices=[...]` for one process
    driving several shards, or `rank/world/uid` for one process per GPU."""

    def __init__(self, n, soft=2e8, g=G, device=0, devices=None, exchange="copy", rank=None, world=None, uid=None):
        self.n = int(n)
        self._h = C.c_void_p()
        L = lib()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            _check(L.murbhip_create_sharded(C.byref(self._h), self.n, soft, g, len(devices), arr,
                                            {"copy": 0, "rccl": 1}[exchange]), "murbhip_create_sharded")
        elif rank is not None:
            _check(L.murbhip_create_rank(C.byref(self._h), self.n, soft, g, device, rank, world, uid),
                   "murbhip_create_rank")
        else:
            _check(L.murbhip_create(C.byref(self._h), self.n, soft, g, device), "murbhip_create")

    # -- state
    def upload(self, s):
        a = [_f32(s[k]) for k in ("qx", "qy", "qz", "vx", "vy", "vz", "m")]
        for x in a:
            if x.shape[0] < self.n:
                raise ValueError("state arrays shorter than n")
        _check(lib().murbhip_upload(self._h, *[_ptr(x) for x in a]), "murbhip_upload")

    def init_bodies(self, scheme="galaxy", seed=0):
        """Initial conditions generated on the device (include/murbhip.h: murbhip_init_bodies)."""
        _check(lib().murbhip_init_bodies(self._h, scheme.encode(), seed), "murbhip_init_bodies")

    def masses(self, with_radii=False):
        m = np.zeros(self.n, np.float32)
        r = np.zeros(self.n, np.float32) if with_radii else None
        _check(lib().murbhip_download_mass(self._h, _ptr(m), _ptr(r) if with_radii else None), "murbhip_download_mass")
        return (m, r) if with_radii else m

    def state(self):
        out = {k: np.zeros(self.n, np.float32) for k in ("qx", "qy", "qz", "vx", "vy", "vz")}
        _check(lib().murbhip_download_state(self._h, *[_ptr(out[k]) for k in ("qx", "qy", "qz", "vx", "vy", "vz")]),
               "murbhip_download_state")
        return out

    def acc(self):
        a = [np.zeros(self.n, np.float32) for _ in range(3)]
        _check(lib().murbhip_download_acc(self._h, *[_ptr(x) for x in a]), "murbhip_download_acc")
        return tuple(a)

    # -- compute (enqueue only; sync() waits)
    def compute_acc(self):
        _check(lib().murbhip_compute_acc(self._h), "murbhip_compute_acc")

    def warmup(self, milliseconds=50.0):
        """Untimed force evaluations on the current state (include/murbhip.h: murbhip_warmup); syncs."""
        _check(lib().murbhip_warmup(self._h, milliseconds), "murbhip_warmup")

    def step(self, dt=3600.0):
        _check(lib().murbhip_step(self._h, dt), "murbhip_step")

    def steps(self, dt, iterations):
        _check(lib().murbhip_steps(self._h, dt, iterations), "murbhip_steps")

    def integrate_host_acc(self, acc, dt):
        a = [_f32(x) for x in acc]
        _check(lib().murbhip_integrate_host_acc(self._h, *[_ptr(x) for x in a], dt), "murbhip_integrate_host_acc")

    def sync(self):
        _check(lib().murbhip_sync(self._h), "murbhip_sync")

    def energy(self):
        """(kinetic, potential) of the current state, reference definitions (gpu+tracking)."""
        ke, pe = C.c_double(), C.c_double()
        _check(lib().murbhip_energy(self._h, C.byref(ke), C.byref(pe)), "murbhip_energy")
        return ke.value, pe.value

    def moments(self):
        """dict: linear momentum P, angular momentum L, mass-weighted position Mq (3 each) and mass M of
        the caller's own bodies (fp64 host sums)."""
        out = (C.c_double * 10)()
        _check(lib().murbhip_moments(self._h, out), "murbhip_moments")
        v = np.array(out[:])
        return {"P": v[0:3], "L": v[3:6], "Mq": v[6:9], "M": float(v[9])}

    # -- tuning / facts
    def set_option(self, key, value):
        _check(lib().murbhip_set_option(self._h, key.encode(), int(value)), f"murbhip_set_option({key})")

    def info(self, key):
        v = C.c_double()
        _check(lib().murbhip_get_info(self._h, key.encode(), C.byref(v)), f"murbhip_get_info({key})")
        return v.value

    def close(self):
        if self._h:
            lib().murbhip_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ====================================================================== host mirror (libmurbhost.so)
# The C++ mirror of the reference's plugin interface (nbody-eurohpc_amd/host/), reached through
# host/capi.cpp.  It provides the product's own initial conditions and lets tests drive
# SimulationNBodyHIP / HIPBodies the way the reference's Catch2 tests drive their CUDA twins.
HOST_LIB_PATH = os.path.join(_HERE, "..", "lib", "libmurbhost.so")
FIELDS = ("qx", "qy", "qz", "vx", "vy", "vz", "m", "r")
_host = None


def host_lib():
    global _host
    if _host is None:
        lib()   # libmurbhost.so depends on libmurbhip.so: load it first (same directory, via rpath too)
        path = os.path.normpath(HOST_LIB_PATH)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: build it with `make -C nbody-eurohpc_amd`")
        H = C.CDLL(path)
        H.murbhost_padding.restype = C.c_ulong
        H.murbhost_padding.argtypes = [C.c_ulong, C.c_char_p]
        H.murbhost_init_bodies.argtypes = [C.c_ulong, C.c_char_p, C.c_ulong] + [_fp] * 8
        H.murbhost_integrate.argtypes = [C.c_ulong, C.c_char_p, _fp, _fp, _fp, C.c_float, C.c_int, C.c_int] + [_fp] * 6
        H.murbhost_sim_create.restype = C.c_void_p
        H.murbhost_sim_create.argtypes = [C.c_ulong, C.c_char_p, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_int),
                                          C.c_int]
        H.murbhost_sim_destroy.argtypes = [C.c_void_p]
        H.murbhost_sim_step.argtypes = [C.c_void_p, C.c_int]
        H.murbhost_sim_init_on_device.argtypes = [C.c_void_p, C.c_ulong]
        H.murbhost_sim_n.restype = C.c_ulong
        H.murbhost_sim_n.argtypes = [C.c_void_p]
        H.murbhost_sim_flops_per_ite.restype = C.c_float
        H.murbhost_sim_flops_per_ite.argtypes = [C.c_void_p]
        H.murbhost_sim_allocated_bytes.restype = C.c_float
        H.murbhost_sim_allocated_bytes.argtypes = [C.c_void_p]
        H.murbhost_sim_state.argtypes = [C.c_void_p] + [_fp] * 8
        H.murbhost_sim_acc.argtypes = [C.c_void_p] + [_fp] * 3
        _dp = C.POINTER(C.c_double)
        H.murbhost_tracking_create.restype = C.c_void_p
        H.murbhost_tracking_create.argtypes = [C.c_ulong, C.c_char_p, C.c_float, C.c_float, C.c_int, C.c_int,
                                               C.POINTER(C.c_int), C.c_int]
        H.murbhost_history_rows.argtypes = [C.c_void_p]
        H.murbhost_history_get.argtypes = [C.c_void_p, _dp, _dp, _dp]
        H.murbhost_history_csv.argtypes = [C.c_char_p, C.c_int, _dp, _dp, _dp]
        H.murbhost_sim_history_csv.argtypes = [C.c_void_p, C.c_char_p]
        _host = H
    return _host


def host_padding(n, scheme="galaxy"):
    return int(host_lib().murbhost_padding(n, scheme.encode()))


def init_bodies(n, scheme="galaxy", seed=0, with_padding=False):
    """The product's initial conditions: host/core/Bodies.cpp (mirror of reference Bodies.cpp:158-257)."""
    tot = n + host_padding(n, scheme)
    a = {k: np.zeros(tot, np.float32) for k in FIELDS}
    host_lib().murbhost_init_bodies(n, scheme.encode(), seed, *[_ptr(a[k]) for k in FIELDS])
    return a if with_padding else {k: v[:n].copy() for k, v in a.items()}


def host_integrate(n, scheme, acc, dt, steps, on_device=False):
    """Bodies / HIPBodies ::updatePositionsAndVelocities(accSoA, dt) applied `steps` times."""
    out = {k: np.zeros(n, np.float32) for k in FIELDS[:6]}
    a = [_f32(x) for x in acc]
    host_lib().murbhost_integrate(n, scheme.encode(), *[_ptr(x) for x in a], dt, steps, int(on_device),
                                  *[_ptr(out[k]) for k in FIELDS[:6]])
    return out


def history_csv(path, energy, ang_momentum, centers):
    """SimulationHistory<double>::saveMetricsToCSV on the given rows (host only; False if the file cannot be opened)."""
    e, a = (np.ascontiguousarray(x, np.float64) for x in (energy, ang_momentum))
    c = np.ascontiguousarray(centers, np.float64).reshape(-1)
    dp = C.POINTER(C.c_double)
    return host_lib().murbhost_history_csv(str(path).encode(), len(e), e.ctypes.data_as(dp), a.ctypes.data_as(dp),
                                           c.ctypes.data_as(dp)) == 0


class HostSim:
    """SimulationNBodyHIP<float> behind HIPBodiesAllocator<float> — the `--im hip+tile[+multi]` plugin."""

    def __init__(self, n, scheme="galaxy", soft=2e8, dt=3600.0, devices=(0,), exchange="rccl", tracking=False,
                 leapfrog=False):
        """tracking=True: SimulationNBodyHIPTracking (`--im hip+tracking`; with leapfrog=True `hip+leapfrog`)."""
        arr = (C.c_int * len(devices))(*devices)
        self.H = host_lib()
        ex = {"copy": 0, "rccl": 1}[exchange]
        if tracking or leapfrog:
            self.h = self.H.murbhost_tracking_create(n, scheme.encode(), soft, dt, int(leapfrog), len(devices), arr, ex)
        else:
            self.h = self.H.murbhost_sim_create(n, scheme.encode(), soft, dt, len(devices), arr, ex)
        self.n = int(self.H.murbhost_sim_n(self.h))

    def history(self):
        """dict of the tracked metrics, one entry per computed iteration (tracking sims only)."""
        rows = int(self.H.murbhost_history_rows(self.h))
        if rows < 0:
            raise RuntimeError("not a tracking simulation")
        e, a, c = np.zeros(rows), np.zeros(rows), np.zeros(3 * rows)
        dp = C.POINTER(C.c_double)
        self.H.murbhost_history_get(self.h, e.ctypes.data_as(dp), a.ctypes.data_as(dp), c.ctypes.data_as(dp))
        return {"energy": e, "ang_momentum": a, "density_center": c.reshape(rows, 3)}

    def save_history_csv(self, path):
        if self.H.murbhost_sim_history_csv(self.h, str(path).encode()) != 0:
            raise RuntimeError(f"cannot open {path}")

    def step(self, iterations=1):
        self.H.murbhost_sim_step(self.h, iterations)

    def init_on_device(self, seed=0):
        """HIPBodies::initOnDevice: the same initial conditions, generated on the device."""
        self.H.murbhost_sim_init_on_device(self.h, seed)

    def state(self):
        pad = 0
        a = {k: np.zeros(self.n + 64, np.float32) for k in FIELDS}   # room for SIMD padding bodies
        self.H.murbhost_sim_state(self.h, *[_ptr(a[k]) for k in FIELDS])
        return {k: v[:self.n].copy() for k, v in a.items()}

    def acc(self):
        a = [np.zeros(self.n, np.float32) for _ in range(3)]
        self.H.murbhost_sim_acc(self.h, *[_ptr(x) for x in a])
        return tuple(a)

    def flops_per_ite(self):
        return float(self.H.murbhost_sim_flops_per_ite(self.h))

    def allocated_bytes(self):
        return float(self.H.murbhost_sim_allocated_bytes(self.h))

    def close(self):
        if self.h:
            self.H.murbhost_sim_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
