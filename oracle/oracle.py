"""oracle/oracle.py -- TEST INFRASTRUCTURE: ctypes doorway onto oracle/liboracle*.so and oracle/_ref/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (cuda-nbody_amd/) never does.

All reference citations are relative to /root/reference/.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# enum class NBodyConfig, src/nbody/nbody_config.hpp:3
NBODY_CONFIG_RANDOM, NBODY_CONFIG_SHELL, NBODY_CONFIG_EXPAND = 0, 1, 2

# Compute::demo_params[0], src/nbody/compute.hpp:91 : {dt, cluster, velocity, softening, damping}
DEMO0 = dict(time_step=np.float32(0.016), cluster_scale=np.float32(1.54), velocity_scale=np.float32(8.0),
             softening=np.float32(0.1), damping=np.float32(1.0))


def scales_for(nb_bodies: int) -> tuple[np.float32, np.float32]:
    """N-dependent (cluster_scale, velocity_scale), src/nbody/compute.cpp:74-92."""
    table = [(1024, 1.52, 2.0), (2048, 1.56, 2.64), (4096, 1.68, 2.98), (8192, 1.98, 2.9),
             (16384, 1.54, 8.0), (32768, 1.44, 11.0)]
    for limit, c, v in table:
        if nb_bodies <= limit:
            return np.float32(c), np.float32(v)
    return DEMO0["cluster_scale"], DEMO0["velocity_scale"]


def build(with_ref: bool | None = None) -> None:
    """Compile the checker (and, when the reference tree is present, oracle/_ref)."""
    targets = ["all"]
    if with_ref is None:
        with_ref = os.path.isdir("/root/reference/src/nbody")
    if with_ref:
        targets.append("ref")
    subprocess.run(["make", "-s", "-C", HERE] + targets, check=True)


def _load(name: str) -> ctypes.CDLL:
    override = os.environ.get("NBODY_ORACLE_LIB")  # `make test-sanitize`: the ASan/UBSan build of the single-thread checker
    if override and name == "liboracle.so":
        return ctypes.CDLL(override)
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        build()
    return ctypes.CDLL(path)


class Oracle:
    """CPU restatement of BodySystemCPU<T>::update + randomise_bodies<T> (see nbody_oracle.c)."""

    def __init__(self, openmp: bool = False):
        lib = _load("liboracle_omp.so" if openmp else "liboracle.so")
        self.lib = lib
        f32p, f64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
        sz, ci, cf, cd = ctypes.c_size_t, ctypes.c_int, ctypes.c_float, ctypes.c_double
        lib.oracle_srand.argtypes = [ctypes.c_uint]
        lib.oracle_rand.restype = ci
        lib.oracle_num_threads.restype = ci
        lib.oracle_set_num_threads.argtypes = [ci]
        for name, fp, ft in (("f32", f32p, cf), ("f64", f64p, cd)):
            getattr(lib, f"oracle_update_{name}").argtypes = [fp, fp, sz, ft, ft, ft, ci]
            getattr(lib, f"oracle_update_{name}").restype = ci
            getattr(lib, f"oracle_benchmark_{name}").argtypes = [fp, fp, sz, ft, ft, ft, ci]
            getattr(lib, f"oracle_benchmark_{name}").restype = cd
            getattr(lib, f"oracle_randomise_{name}").argtypes = [ci, fp, fp, sz, cf, cf]
            getattr(lib, f"oracle_randomise_{name}").restype = None
        lib.oracle_update_f32_avx.argtypes = [f32p, f32p, sz, cf, cf, cf, ci]
        lib.oracle_update_f32_avx.restype = ci
        lib.oracle_softening_sq_f32.argtypes = [cf]
        lib.oracle_softening_sq_f32.restype = cf
        lib.oracle_softening_sq_f64.argtypes = [cf]
        lib.oracle_softening_sq_f64.restype = cd
        lib.oracle_benchmark_partial_f32.argtypes = [f32p, sz, sz, cf, f64p]
        lib.oracle_benchmark_partial_f32.restype = cd
        lib.oracle_benchmark_partial_f64.argtypes = [f64p, sz, sz, cd, f64p]
        lib.oracle_benchmark_partial_f64.restype = cd
        lib.oracle_update_subset_f32.argtypes = [f32p, f32p, sz, sz, sz, cf, cf, cf, f32p, f32p]
        lib.oracle_update_subset_f32.restype = None
        lib.oracle_update_subset_f64.argtypes = [f64p, f64p, sz, sz, sz, cd, cd, cd, f64p, f64p]
        lib.oracle_update_subset_f64.restype = None
        lib.oracle_accel_f64_from_f32.argtypes = [f32p, sz, sz, sz, cd, f64p]
        lib.oracle_accel_f64_from_f32.restype = None

    # -- helpers ---------------------------------------------------------------------------
    @staticmethod
    def _suffix(dtype) -> str:
        dtype = np.dtype(dtype)
        if dtype == np.float32:
            return "f32"
        if dtype == np.float64:
            return "f64"
        raise TypeError(dtype)

    @staticmethod
    def _ptr(a: np.ndarray):
        assert a.flags.c_contiguous and a.flags.writeable
        return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float if a.dtype == np.float32 else ctypes.c_double))

    def num_threads(self) -> int:
        return int(self.lib.oracle_num_threads())

    def set_num_threads(self, n: int) -> None:
        self.lib.oracle_set_num_threads(int(n))

    def srand(self, seed: int = 1) -> None:
        self.lib.oracle_srand(seed)

    def softening_sq(self, softening, dtype):
        s = np.float32(softening)
        if np.dtype(dtype) == np.float32:
            return np.float32(self.lib.oracle_softening_sq_f32(s))
        return np.float64(self.lib.oracle_softening_sq_f64(s))

    # -- randomise_bodies ------------------------------------------------------------------
    def randomise(self, config: int, nb_bodies: int, cluster_scale, velocity_scale, dtype=np.float32):
        """randomise_bodies<T> on the CURRENT libc rand() state.  Returns (pos[4N], vel[4N])."""
        pos = np.zeros(4 * nb_bodies, dtype=dtype)
        vel = np.zeros(4 * nb_bodies, dtype=dtype)
        fn = getattr(self.lib, f"oracle_randomise_{self._suffix(dtype)}")
        fn(config, self._ptr(pos), self._ptr(vel), nb_bodies, np.float32(cluster_scale), np.float32(velocity_scale))
        return pos, vel

    def startup_state(self, nb_bodies: int, dtype=np.float32, seed: int = 1, config: int = NBODY_CONFIG_SHELL):
        """The bodies a fresh `cuda-nbody --numbodies=N [--fp64] [--cpu]` process starts from.

        rand() is consumed three times before the first step: the fp32 system's
        ctor reset, the fp64 system's ctor reset (both with demo_params[0] scales, compute_cuda.cpp:129-133 /
        compute_cpu.cpp:43-44), then Compute's ctor re-resets the ACTIVE precision with the N-scaled
        params (compute.cpp:74-100).  The third segment is what the simulation runs on.
        """
        self.srand(seed)
        self.randomise(NBODY_CONFIG_SHELL, nb_bodies, DEMO0["cluster_scale"], DEMO0["velocity_scale"], np.float32)
        self.randomise(NBODY_CONFIG_SHELL, nb_bodies, DEMO0["cluster_scale"], DEMO0["velocity_scale"], np.float64)
        c, v = scales_for(nb_bodies)
        return self.randomise(config, nb_bodies, c, v, dtype)

    # -- BodySystemCPU<T>::update ----------------------------------------------------------
    def update(self, pos: np.ndarray, vel: np.ndarray, dt, steps: int = 1, softening=DEMO0["softening"],
               damping=DEMO0["damping"], avx: bool = False):
        """In place: `steps` x BodySystemCPU<T>::update(dt) on interleaved pos[4N]/vel[4N]."""
        assert pos.dtype == vel.dtype and pos.size == vel.size and pos.size % 4 == 0
        n = pos.size // 4
        sfx = self._suffix(pos.dtype)
        eps2 = self.softening_sq(softening, pos.dtype)
        T = np.float32 if sfx == "f32" else np.float64
        name = "oracle_update_f32_avx" if (avx and sfx == "f32") else f"oracle_update_{sfx}"
        # dt and damping originate as `float` in the reference (NBodyParams, params.hpp:8-16; update(float dt),
        # compute_cpu.cpp:117) and are widened to T, so they pass through float32 here too
        rc = getattr(self.lib, name)(self._ptr(pos), self._ptr(vel), n, T(eps2), T(np.float32(damping)), T(np.float32(dt)), steps)
        if rc:
            raise RuntimeError(f"{name} failed rc={rc}")
        return pos, vel

    def benchmark(self, pos: np.ndarray, vel: np.ndarray, dt, steps: int, softening=DEMO0["softening"],
                  damping=DEMO0["damping"]) -> float:
        """ComputeCPU::run_benchmark semantics (compute_cpu.cpp:72-80): ms for `steps` updates, no warm-up."""
        n = pos.size // 4
        sfx = self._suffix(pos.dtype)
        T = np.float32 if sfx == "f32" else np.float64
        eps2 = self.softening_sq(softening, pos.dtype)
        ms = getattr(self.lib, f"oracle_benchmark_{sfx}")(self._ptr(pos), self._ptr(vel), n, T(eps2), T(np.float32(damping)), T(np.float32(dt)), steps)
        if ms < 0:
            raise RuntimeError("oracle benchmark failed")
        return float(ms)

    def update_subset(self, pos: np.ndarray, vel: np.ndarray, i0: int, ni: int, dt, softening=DEMO0["softening"],
                      damping=DEMO0["damping"]):
        """One BodySystemCPU<T>::update step for bodies [i0, i0+ni) only (bit-identical rows of a full update)."""
        n = pos.size // 4
        sfx = self._suffix(pos.dtype)
        T = np.float32 if sfx == "f32" else np.float64
        eps2 = self.softening_sq(softening, pos.dtype)
        out_pos, out_vel = np.zeros(4 * ni, dtype=pos.dtype), np.zeros(4 * ni, dtype=pos.dtype)
        getattr(self.lib, f"oracle_update_subset_{sfx}")(self._ptr(pos), self._ptr(vel), n, i0, ni, T(eps2), T(np.float32(damping)),
                                                         T(np.float32(dt)), self._ptr(out_pos), self._ptr(out_vel))
        return out_pos, out_vel

    def benchmark_partial(self, pos: np.ndarray, sample_i: int, softening=DEMO0["softening"]) -> float:
        """ms for the force pass of bodies i in [0, sample_i) against all bodies j (bounded cpu_baseline sample)."""
        n = pos.size // 4
        sfx = self._suffix(pos.dtype)
        eps2 = self.softening_sq(softening, pos.dtype)
        chk = ctypes.c_double(0)
        ms = getattr(self.lib, f"oracle_benchmark_partial_{sfx}")(self._ptr(pos), n, sample_i, eps2, ctypes.byref(chk))
        if ms < 0 or not np.isfinite(chk.value):
            raise RuntimeError(f"oracle partial benchmark failed ({ms}, {chk.value})")
        return float(ms)

    def accel_f64(self, pos32: np.ndarray, i0: int, ni: int, softening=DEMO0["softening"]) -> np.ndarray:
        """fp64 accelerations of bodies [i0,i0+ni) from fp32 positions (yardstick, not a reference function)."""
        n = pos32.size // 4
        out = np.zeros(3 * ni, dtype=np.float64)
        eps2 = float(np.float64(np.float32(softening)) ** 2)
        self.lib.oracle_accel_f64_from_f32(self._ptr(pos32), n, i0, ni, eps2, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        return out.reshape(ni, 3)


class ReferenceRandomise:
    """The reference's own randomise_bodies<T>, compiled unmodified into oracle/_ref (build container only)."""

    PATH = os.path.join(HERE, "_ref", "librandomise_ref.so")

    @classmethod
    def available(cls) -> bool:
        return os.path.exists(cls.PATH)

    def __init__(self):
        lib = ctypes.CDLL(self.PATH)
        self.lib = lib
        f32p, f64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
        lib.ref_srand.argtypes = [ctypes.c_uint]
        lib.ref_randomise_f32.argtypes = [ctypes.c_int, f32p, f32p, ctypes.c_size_t, ctypes.c_float, ctypes.c_float]
        lib.ref_randomise_f64.argtypes = [ctypes.c_int, f64p, f64p, ctypes.c_size_t, ctypes.c_float, ctypes.c_float]

    def srand(self, seed: int = 1) -> None:
        self.lib.ref_srand(seed)

    def randomise(self, config: int, nb_bodies: int, cluster_scale, velocity_scale, dtype=np.float32):
        pos = np.zeros(4 * nb_bodies, dtype=dtype)
        vel = np.zeros(4 * nb_bodies, dtype=dtype)
        if np.dtype(dtype) == np.float32:
            fn, ct = self.lib.ref_randomise_f32, ctypes.c_float
        else:
            fn, ct = self.lib.ref_randomise_f64, ctypes.c_double
        fn(config, pos.ctypes.data_as(ctypes.POINTER(ct)), vel.ctypes.data_as(ctypes.POINTER(ct)), nb_bodies,
           np.float32(cluster_scale), np.float32(velocity_scale))
        return pos, vel
