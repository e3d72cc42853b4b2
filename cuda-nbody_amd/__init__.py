"""cuda-nbody_amd -- MI355X-native all-pairs N-body hot path behind the reference's own seams.

The product is the C-ABI shared library ``libnbody_hip.so`` (include/nbody_hip.h) built from the hand-written
gfx950 HIP kernels in ``csrc/``, plus the C++23 host mirror of the reference's ``BodySystemCUDA`` /
``ComputeCUDA`` / ``Compute`` / CLI in ``host/``.  This Python module is the thin ctypes binding used by
tests/, bench.py and __graft_entry__.py: it mirrors the reference's ``BodySystemCUDADefault<T>`` interface
(/root/reference/src/nbody/bodysystemcuda.hpp:38-72, bodysystemcuda_default.cu:19-55) call for call.

There is NO CPU fallback: if libnbody_hip.so is missing or a HIP call fails, this module raises.

The directory name carries a hyphen, so import it with ``__graft_entry__.load_package()``.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBODY_HIP_LIB", os.path.join(HERE, "libnbody_hip.so"))
# The lab bench (include/nbody_hip_lab.h: real-RCCL self-test, loopback rank, in-process world, allocation-failure hook) is a second
# library made of the SAME object files plus csrc/nbody_comm_lab.hip.  A process uses ONE of the two: use_lab() (or NBODY_HIP_LAB=1 in
# the environment) before the first lib() call makes lib() load libnbody_hip_lab.so instead -- tests/ and tools/ that need the lab
# do that in a process of their own; the product library never exports the lab's symbols.
LAB_LIB_PATH = os.environ.get("NBODY_HIP_LAB_LIB", os.path.join(HERE, "libnbody_hip_lab.so"))

NB_MODE_STRICT, NB_MODE_FAST = 0, 1
NB_SHARD_ACC_IN, NB_SHARD_FINALIZE = 1, 2
NB_ERR_INVALID_ARGUMENT, NB_ERR_UNSUPPORTED, NB_ERR_RCCL_BASE = 10001, 10002, 20000
NB_ERR_OUT_OF_MEMORY = 2  # = hipErrorOutOfMemory: what nb_alloc answers when the device has no room

# enum class NBodyConfig, src/nbody/nbody_config.hpp:3
NBODY_CONFIG_RANDOM, NBODY_CONFIG_SHELL, NBODY_CONFIG_EXPAND = 0, 1, 2


@dataclass
class NBodyParams:
    """struct NBodyParams, src/nbody/params.hpp:8-16 (camera_origin dropped: display only)."""
    time_step: float = 0.016
    cluster_scale: float = 1.54
    velocity_scale: float = 8.0
    softening: float = 0.1
    damping: float = 1.0


# Compute::demo_params, src/nbody/compute.hpp:90-97
DEMO_PARAMS = (
    NBodyParams(0.016, 1.54, 8.0, 0.1, 1.0),
    NBodyParams(0.016, 0.68, 20.0, 0.1, 1.0),
    NBodyParams(0.0006, 0.16, 1000.0, 1.0, 1.0),
    NBodyParams(0.0006, 0.16, 1000.0, 1.0, 1.0),
    NBodyParams(0.0019, 0.32, 276.0, 1.0, 1.0),
    NBodyParams(0.0016, 0.32, 272.0, 0.145, 1.0),
    NBodyParams(0.016, 6.04, 0.0, 1.0, 1.0),
)


class NBodyHipError(RuntimeError):
    def __init__(self, code: int, what: str):
        super().__init__(f"{what}: {code}")
        self.code = code


class DeviceInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 256), ("arch", ctypes.c_char * 64), ("compute_units", ctypes.c_int),
                ("wavefront_size", ctypes.c_int), ("clock_khz", ctypes.c_int), ("can_map_host_memory", ctypes.c_int),
                ("lds_bytes_per_cu", ctypes.c_int), ("total_memory", ctypes.c_size_t)]


class PairPlan(ctypes.Structure):
    _fields_ = [("applies", ctypes.c_int), ("bodies_per_lane", ctypes.c_int), ("waves_per_block", ctypes.c_int), ("splits", ctypes.c_uint),
                ("blocks", ctypes.c_uint), ("block_bodies", ctypes.c_uint), ("reaction_slots", ctypes.c_uint), ("grid_blocks", ctypes.c_uint),
                ("lds_bytes", ctypes.c_uint), ("workspace_bytes", ctypes.c_size_t), ("slices", ctypes.c_uint)]


class CommSelftest(ctypes.Structure):
    """nb_comm_selftest_t (include/nbody_hip_tuning.h): what the self-loop through the real RCCL reported"""
    _fields_ = [("rccl_version", ctypes.c_int), ("send_recv_status", ctypes.c_int), ("all_gather_status", ctypes.c_int),
                ("send_recv_ms", ctypes.c_float), ("all_gather_ms", ctypes.c_float),
                ("send_recv_wrong_bytes", ctypes.c_size_t), ("all_gather_wrong_bytes", ctypes.c_size_t),
                ("refused_call", ctypes.c_char * 64), ("library_path", ctypes.c_char * 256)]


class LaunchPlan(ctypes.Structure):
    _fields_ = [("bodies_per_lane", ctypes.c_int), ("lanes_per_body", ctypes.c_int), ("tile_bodies", ctypes.c_int),
                ("block_threads", ctypes.c_int), ("grid_blocks", ctypes.c_uint), ("lds_bytes", ctypes.c_uint)]


# name -> (restype, argtypes); this table is also what tests/test_capi_symbols.py checks against the header
_vp, _ci, _cu, _cf, _cd, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_float, ctypes.c_double, ctypes.c_size_t
_P = ctypes.POINTER
SIGNATURES = {
    "nb_error_string": (ctypes.c_char_p, [_ci]),
    "nb_version": (ctypes.c_char_p, []),
    "nb_device_count": (_ci, [_P(_ci)]),
    "nb_set_device": (_ci, [_ci]),
    "nb_get_device": (_ci, [_P(_ci)]),
    "nb_device_info": (_ci, [_ci, _P(DeviceInfo)]),
    "nb_alloc": (_ci, [_P(_vp), _sz]),
    "nb_free": (_ci, [_vp]),
    "nb_memset": (_ci, [_vp, _ci, _sz, _vp]),
    "nb_h2d": (_ci, [_vp, _vp, _sz, _vp]),
    "nb_d2h": (_ci, [_vp, _vp, _sz, _vp]),
    "nb_d2d": (_ci, [_vp, _vp, _sz, _vp]),
    "nb_host_alloc_mapped": (_ci, [_P(_vp), _P(_vp), _sz]),
    "nb_host_free": (_ci, [_vp]),
    "nb_stream_create": (_ci, [_P(_vp)]),
    "nb_stream_destroy": (_ci, [_vp]),
    "nb_stream_synchronize": (_ci, [_vp]),
    "nb_stream_wait_event": (_ci, [_vp, _vp]),
    "nb_event_create": (_ci, [_P(_vp)]),
    "nb_event_destroy": (_ci, [_vp]),
    "nb_event_record": (_ci, [_vp, _vp]),
    "nb_event_synchronize": (_ci, [_vp]),
    "nb_event_elapsed_ms": (_ci, [_P(_cf), _vp, _vp]),
    "nb_device_synchronize": (_ci, []),
    "nb_set_softening_sq_f32": (_ci, [_cf]),
    "nb_set_softening_sq_f64": (_ci, [_cd]),
    "nb_get_softening_sq_f32": (_ci, [_P(_cf)]),
    "nb_get_softening_sq_f64": (_ci, [_P(_cd)]),
    "nb_integrate_f32": (_ci, [_vp, _vp, _vp, _cf, _cf, _cu, _ci, _ci, _vp]),
    "nb_integrate_f64": (_ci, [_vp, _vp, _vp, _cd, _cd, _cu, _ci, _ci, _vp]),
    "nb_integrate_shard_f32": (_ci, [_vp, _vp, _vp, _vp, _cu, _cu, _cu, _cu, _cu, _cf, _cf, _ci, _ci, _vp]),
    "nb_integrate_shard_f64": (_ci, [_vp, _vp, _vp, _vp, _cu, _cu, _cu, _cu, _cu, _cd, _cd, _ci, _ci, _vp]),
    "nb_graph_create_f32": (_ci, [_P(_vp), _vp, _vp, _vp, _cf, _cf, _cu, _ci, _ci, _cu]),
    "nb_graph_create_f64": (_ci, [_P(_vp), _vp, _vp, _vp, _cd, _cd, _cu, _ci, _ci, _cu]),
    "nb_graph_create_ws_f32": (_ci, [_P(_vp), _vp, _vp, _vp, _cf, _cf, _cu, _ci, _ci, _cu, _vp, _sz]),
    "nb_graph_create_ws_f64": (_ci, [_P(_vp), _vp, _vp, _vp, _cd, _cd, _cu, _ci, _ci, _cu, _vp, _sz]),
    "nb_workspace_bytes_f32": (_ci, [_cu, _ci, _P(_sz)]),
    "nb_workspace_bytes_f64": (_ci, [_cu, _ci, _P(_sz)]),
    "nb_workspace_bytes_capped_f32": (_ci, [_cu, _ci, _sz, _P(_sz)]),
    "nb_workspace_bytes_capped_f64": (_ci, [_cu, _ci, _sz, _P(_sz)]),
    "nb_integrate_ws_f32": (_ci, [_vp, _vp, _vp, _cf, _cf, _cu, _ci, _ci, _vp, _sz, _vp]),
    "nb_integrate_ws_f64": (_ci, [_vp, _vp, _vp, _cd, _cd, _cu, _ci, _ci, _vp, _sz, _vp]),
    "nb_pair_plan_f32": (_ci, [_cu, _P(PairPlan)]),
    "nb_pair_plan_f64": (_ci, [_cu, _P(PairPlan)]),
    "nb_graph_launch": (_ci, [_vp, _vp]),
    "nb_graph_destroy": (_ci, [_vp]),
    "nb_comm_unique_id": (_ci, [_vp]),
    "nb_comm_init_rank": (_ci, [_P(_vp), _vp, _ci, _ci]),
    "nb_comm_init_all": (_ci, [_P(_vp), _ci, _P(_ci)]),
    "nb_comm_destroy": (_ci, [_vp]),
    "nb_comm_info": (_ci, [_vp, _P(_ci), _P(_ci), _P(_ci)]),
    "nb_comm_stream_create": (_ci, [_vp, _P(_vp)]),
    "nb_stream_create_placed": (_ci, [_P(_vp)]),
    "nb_comm_set_workspace": (_ci, [_vp, _vp, _sz]),
    "nb_comm_layout_f32": (_ci, [_vp, _cu, _ci, _P(_ci)]),
    "nb_comm_layout_f64": (_ci, [_vp, _cu, _ci, _P(_ci)]),
    "nb_comm_set_exchange_grouping": (_ci, [_vp, _ci]),
    "nb_comm_get_exchange_grouping": (_ci, [_vp, _P(_ci)]),
    "nb_comm_workspace_bytes_f32": (_ci, [_vp, _cu, _ci, _P(_sz)]),
    "nb_comm_workspace_bytes_f64": (_ci, [_vp, _cu, _ci, _P(_sz)]),
    "nb_sharded_step_f32": (_ci, [_vp, _vp, _vp, _vp, _vp, _cu, _cf, _cf, _ci, _ci, _vp]),
    "nb_sharded_step_f64": (_ci, [_vp, _vp, _vp, _vp, _vp, _cu, _cd, _cd, _ci, _ci, _vp]),
    "nb_sharded_step_all_f32": (_ci, [_P(_vp), _ci, _P(_vp), _P(_vp), _P(_vp), _P(_vp), _cu, _cf, _cf, _ci, _ci, _P(_vp)]),
    "nb_sharded_step_all_f64": (_ci, [_P(_vp), _ci, _P(_vp), _P(_vp), _P(_vp), _P(_vp), _cu, _cd, _cd, _ci, _ci, _P(_vp)]),
    "nb_exchange_tiles_f32": (_ci, [_vp, _vp, _cu, _vp]),
    "nb_exchange_tiles_f64": (_ci, [_vp, _vp, _cu, _vp]),
    "nb_allgather_f32": (_ci, [_vp, _vp, _cu, _vp]),
    "nb_allgather_f64": (_ci, [_vp, _vp, _cu, _vp]),
    "nb_exchange_wait_tile": (_ci, [_vp, _ci, _vp]),
    "nb_exchange_wait_all": (_ci, [_vp, _vp]),
    "nb_plan_f32": (_ci, [_cu, _cu, _P(LaunchPlan)]),
    "nb_plan_f64": (_ci, [_cu, _cu, _P(LaunchPlan)]),
}

# include/nbody_hip_tuning.h: process-global tuning / test hooks, not part of the drop-in boundary
TUNING_SIGNATURES = {
    "nb_set_plan_override": (_ci, [_ci, _ci, _ci]),
    "nb_set_pair_plan_override": (_ci, [_ci, _ci, _ci, _ci]),
    "nb_set_pair_slices_override": (_ci, [_ci]),
    "nb_comm_set_pair_min_slice": (_ci, [_ci]),
    "nb_set_late_diagonal": (_ci, [_ci]),
    "nb_emulate_pair_rank_f32": (_ci, [_vp, _vp, _vp, _vp, _P(_sz), _cu, _ci, _ci, _cf, _cf, _vp]),
    "nb_emulate_pair_rank_f64": (_ci, [_vp, _vp, _vp, _vp, _P(_sz), _cu, _ci, _ci, _cd, _cd, _vp]),
    "nb_comm_reaction_exchange_f32": (_ci, [_vp, _cu, _vp]),
    "nb_comm_reaction_exchange_f64": (_ci, [_vp, _cu, _vp]),
    "nb_set_pair_probe_event": (_ci, [_vp]),
    "nb_set_memory_budget": (_ci, [_sz]),
    "nb_lds_optin_count": (_ci, [_P(_ci)]),
    "nb_comm_transport_info": (_ci, [_vp, _P(_ci), ctypes.c_char_p, _sz]),
    "nb_comm_last_step_trace": (_ci, [_vp, ctypes.c_char_p, _sz]),
    "nb_comm_side_stream_collisions": (_ci, [_vp, _P(_ci)]),
    "nb_comm_settle_side_stream": (_ci, [_vp, _vp]),
    "nb_comm_caller_stream_placement": (_ci, [_vp, _P(_ci)]),
    "nb_comm_pair_work_f32": (_ci, [_vp, _cu, _P(ctypes.c_ulonglong), _P(_ci)]),
    "nb_comm_pair_work_f64": (_ci, [_vp, _cu, _P(ctypes.c_ulonglong), _P(_ci)]),
    "nb_comm_last_enqueue_ms": (_ci, [_vp, _P(_cd)]),
    "nb_set_pair_clock_words": (_ci, [_vp, _sz]),
}

# include/nbody_hip_lab.h: exported by libnbody_hip_lab.so only
LAB_SIGNATURES = {
    "nb_clock_probe_launch": (_ci, [_vp, _ci, _cu, _vp]),
    "nb_set_alloc_limit": (_ci, [_sz]),
    "nb_comm_selftest_open": (_ci, [_P(_vp), _vp]),
    "nb_comm_loopback_open": (_ci, [_P(_vp), _vp, _ci, _ci]),
    "nb_comm_inprocess_open_all": (_ci, [_P(_vp), _ci, _vp]),
    "nb_comm_selftest_f32": (_ci, [_vp, _sz, _vp, _P(CommSelftest)]),
    "nb_comm_self_transfer_f32": (_ci, [_vp, _vp, _vp, _sz, _ci, _ci, _vp, _vp, _vp]),
    "nb_comm_replace_side_stream": (_ci, [_vp]),
}

_lib = None
_lab = os.environ.get("NBODY_HIP_LAB") == "1"


def use_lab() -> None:
    """This process works with the lab library (libnbody_hip_lab.so: everything libnbody_hip.so exports + include/nbody_hip_lab.h).
    Must come before the first lib() call: communicators, streams and the process-global settings belong to ONE loaded library."""
    global _lab
    if _lib is not None and not _lab:
        raise RuntimeError("use_lab() after libnbody_hip.so was loaded: a process uses one of the two libraries (set NBODY_HIP_LAB=1 or call it first)")
    _lab = True


def is_lab() -> bool:
    return _lab


def lib() -> ctypes.CDLL:
    """Load libnbody_hip.so -- or, after use_lab(), libnbody_hip_lab.so (fails loudly when the HIP extension has not been built)."""
    global _lib
    if _lib is None:
        path = LAB_LIB_PATH if _lab else LIB_PATH
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: build it with `make -C {os.path.join(HERE, 'csrc')}` "
                                    "(or __graft_entry__.build()); there is no CPU fallback")
        handle = ctypes.CDLL(path)
        for name, (restype, argtypes) in {**SIGNATURES, **TUNING_SIGNATURES, **(LAB_SIGNATURES if _lab else {})}.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = restype, argtypes
        _lib = handle
    return _lib


def check(code: int, what: str = "nbody_hip") -> None:
    if code != 0:
        name = lib().nb_error_string(code)
        raise NBodyHipError(code, f"{what}: {name.decode() if name else '?'}")


def device_count() -> int:
    n = _ci(0)
    check(lib().nb_device_count(ctypes.byref(n)), "nb_device_count")
    return n.value


def device_info(device: int = 0) -> DeviceInfo:
    info = DeviceInfo()
    check(lib().nb_device_info(device, ctypes.byref(info)), "nb_device_info")
    return info


def plan(i_count: int, j_count: int, dtype=np.float32) -> LaunchPlan:
    p = LaunchPlan()
    fn = lib().nb_plan_f32 if np.dtype(dtype) == np.float32 else lib().nb_plan_f64
    check(fn(i_count, j_count, ctypes.byref(p)), "nb_plan")
    return p


def pair_plan(num_bodies: int, dtype=np.float32) -> PairPlan:
    p = PairPlan()
    fn = lib().nb_pair_plan_f32 if np.dtype(dtype) == np.float32 else lib().nb_pair_plan_f64
    check(fn(num_bodies, ctypes.byref(p)), "nb_pair_plan")
    return p


def set_pair_plan_override(vectors_per_lane: int = 0, waves_per_block: int = 0, splits: int = 0, min_bodies: int = 0) -> None:
    check(lib().nb_set_pair_plan_override(vectors_per_lane, waves_per_block, splits, min_bodies), "nb_set_pair_plan_override")


def set_pair_slices_override(slices: int = 0) -> None:
    check(lib().nb_set_pair_slices_override(slices), "nb_set_pair_slices_override")


def set_plan_override(bodies_per_lane: int = 0, lanes_per_body: int = 0, tile_bodies: int = 0) -> None:
    check(lib().nb_set_plan_override(bodies_per_lane, lanes_per_body, tile_bodies), "nb_set_plan_override")


def set_softening_squared(value) -> None:
    """set_softening_squared(float|double), src/nbody/bodysystemcuda.cu:46-60 (overload chosen by dtype)."""
    if isinstance(value, np.float64) or type(value) is float:
        check(lib().nb_set_softening_sq_f64(float(value)), "nb_set_softening_sq_f64")
    else:
        check(lib().nb_set_softening_sq_f32(np.float32(value)), "nb_set_softening_sq_f32")


class DeviceBuffer:
    """A caller-owned device array (what thrust::device_vector<T> is to the reference)."""

    def __init__(self, nbytes: int):
        self.ptr = _vp()
        self.nbytes = nbytes
        check(lib().nb_alloc(ctypes.byref(self.ptr), nbytes), "nb_alloc")
        check(lib().nb_memset(self.ptr, 0, nbytes, None), "nb_memset")

    def upload(self, host: np.ndarray) -> None:
        assert host.flags.c_contiguous and host.nbytes <= self.nbytes
        check(lib().nb_h2d(self.ptr, host.ctypes.data_as(_vp), host.nbytes, None), "nb_h2d")

    def download(self, host: np.ndarray) -> np.ndarray:
        assert host.flags.c_contiguous and host.nbytes <= self.nbytes
        check(lib().nb_d2h(host.ctypes.data_as(_vp), self.ptr, host.nbytes, None), "nb_d2h")
        return host

    def free(self) -> None:
        if self.ptr:
            lib().nb_free(self.ptr)
            self.ptr = _vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def integrate_nbody_system(new_positions, old_positions, velocities, current_read: int, delta_time, damping,
                           num_bodies: int, block_size: int, dtype=np.float32, mode: int = NB_MODE_FAST, stream=None) -> None:
    """integrateNbodySystem<T>, src/nbody/integrate_nbody_cuda.hpp:5 (same argument order; `current_read` is
    unused there too).  Pointers are raw device addresses (int / c_void_p)."""
    del current_read
    if np.dtype(dtype) == np.float32:
        rc = lib().nb_integrate_f32(new_positions, old_positions, velocities, np.float32(delta_time), np.float32(damping),
                                    num_bodies, block_size, mode, stream)
    else:
        rc = lib().nb_integrate_f64(new_positions, old_positions, velocities, float(delta_time), float(damping),
                                    num_bodies, block_size, mode, stream)
    check(rc, "integrateNbodySystem")


class BodySystemHIP:
    """Mirror of BodySystemCUDADefault<T> (src/nbody/bodysystemcuda_default.hpp:28-36, .cu:8-55) on HIP.

    Two ping-pong position arrays + one velocity array of 4N T on the device; ``update`` writes
    pos[1-read] from pos[read] and swaps; ``set_*`` resets read=0/write=1; ``get_*`` are blocking D2H copies.
    ``reset`` needs initial conditions: pass ``randomise`` (a callable (config, n, cluster, velocity, dtype) ->
    (pos, vel)); the C++ host mirror owns the real randomise_bodies restatement.
    """

    def __init__(self, nb_bodies: int, block_size: int = 256, params: NBodyParams | None = None, dtype=np.float32,
                 positions: np.ndarray | None = None, velocities: np.ndarray | None = None, mode: int = NB_MODE_FAST,
                 workspace: bool = False, workspace_cap: int | None = None):
        """`workspace=True`: own the scratch memory nb_workspace_bytes_* asks for and step through nb_integrate_ws_* (FAST mode
        then takes the pairwise layout where it applies), as BodySystemHIPStored does in the C++ host."""
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError("float32 or float64")
        self.nb_bodies = int(nb_bodies)
        self.block_size = int(block_size)
        self.mode = mode
        params = params or NBodyParams()
        self.damping = self.dtype.type(np.float32(params.damping))  # float -> T, bodysystemcuda.cpp:56
        self.current_read, self.current_write = 0, 1
        nbytes = 4 * self.nb_bodies * self.dtype.itemsize
        self._pos = [DeviceBuffer(nbytes), DeviceBuffer(nbytes)]
        self._vel = DeviceBuffer(nbytes)
        self._host_pos = np.zeros(4 * self.nb_bodies, dtype=self.dtype)
        self._host_vel = np.zeros(4 * self.nb_bodies, dtype=self.dtype)
        self._workspace, self._workspace_bytes = None, 0
        if workspace:
            need = workspace_bytes(self.nb_bodies, self.dtype, self.mode, workspace_cap)  # (`workspace_cap`: spend at most that many bytes)
            if need:
                self._workspace, self._workspace_bytes = _vp(), need
                check(lib().nb_alloc(ctypes.byref(self._workspace), need), "nb_alloc(workspace)")
        self._set_softening(params.softening)
        if positions is not None:
            self.set_position(positions)
            self.set_velocity(velocities)

    def _set_softening(self, softening) -> None:
        # bodysystemcuda.cpp:42-46: softening2 = T(softening) * T(softening)
        s = self.dtype.type(np.float32(softening))
        self._softening_sq = s * s

    def _apply_softening(self) -> None:
        if self.dtype == np.float32:
            check(lib().nb_set_softening_sq_f32(self._softening_sq), "nb_set_softening_sq_f32")
        else:
            check(lib().nb_set_softening_sq_f64(float(self._softening_sq)), "nb_set_softening_sq_f64")

    def update_params(self, params: NBodyParams) -> None:
        self._set_softening(params.softening)
        self.damping = self.dtype.type(np.float32(params.damping))

    def reset(self, params: NBodyParams, config: int, randomise) -> None:
        pos, vel = randomise(config, self.nb_bodies, params.cluster_scale, params.velocity_scale, self.dtype)
        self.set_position(pos)
        self.set_velocity(vel)

    def update(self, delta_time, stream=None) -> None:
        self._apply_softening()
        if self._workspace is not None:
            f32 = self.dtype == np.float32
            fn, scalar = (lib().nb_integrate_ws_f32, np.float32) if f32 else (lib().nb_integrate_ws_f64, float)
            check(fn(self._pos[1 - self.current_read].ptr, self._pos[self.current_read].ptr, self._vel.ptr, scalar(delta_time), scalar(self.damping),
                     self.nb_bodies, self.block_size, self.mode, self._workspace, self._workspace_bytes, stream), "nb_integrate_ws")
        else:
            integrate_nbody_system(self._pos[1 - self.current_read].ptr, self._pos[self.current_read].ptr, self._vel.ptr,
                                   self.current_read, delta_time, self.damping, self.nb_bodies, self.block_size,
                                   self.dtype, self.mode, stream)
        self.current_read, self.current_write = self.current_write, self.current_read

    def update_many(self, delta_time, steps: int, stream=None) -> None:
        """`steps` (even) updates as ONE hipGraph launch (nb_graph_*): same kernels, same results as `steps` x update()."""
        # everything nb_graph_create_* bakes into the capture (damping and softening^2 are kernel arguments too)
        key = (float(delta_time), steps, self.current_read, self.mode, float(self.damping), float(self._softening_sq))
        if getattr(self, "_graph_key", None) != key:
            self._free_graph()
            self._apply_softening()
            g = _vp()
            a, b = self._pos[self.current_read].ptr, self._pos[1 - self.current_read].ptr
            if self.dtype == np.float32:
                rc = lib().nb_graph_create_ws_f32(ctypes.byref(g), a, b, self._vel.ptr, np.float32(delta_time), self.damping,
                                                  self.nb_bodies, self.block_size, self.mode, steps, self._workspace, self._workspace_bytes)
            else:
                rc = lib().nb_graph_create_ws_f64(ctypes.byref(g), a, b, self._vel.ptr, float(delta_time), float(self.damping),
                                                  self.nb_bodies, self.block_size, self.mode, steps, self._workspace, self._workspace_bytes)
            check(rc, "nb_graph_create")
            self._graph, self._graph_key = g, key
        check(lib().nb_graph_launch(self._graph, stream), "nb_graph_launch")  # even step count: read index unchanged

    def _free_graph(self) -> None:
        if getattr(self, "_graph", None):
            lib().nb_graph_destroy(self._graph)
        self._graph, self._graph_key = None, None

    def get_position(self) -> np.ndarray:
        return self._pos[self.current_read].download(self._host_pos)

    def get_velocity(self) -> np.ndarray:
        return self._vel.download(self._host_vel)

    def set_position(self, data: np.ndarray) -> None:
        data = np.ascontiguousarray(data, dtype=self.dtype)
        assert data.size == 4 * self.nb_bodies
        self.current_read, self.current_write = 0, 1
        self._pos[0].upload(data)

    def set_velocity(self, data: np.ndarray) -> None:
        data = np.ascontiguousarray(data, dtype=self.dtype)
        assert data.size == 4 * self.nb_bodies
        self.current_read, self.current_write = 0, 1
        self._vel.upload(data)

    def synchronize(self) -> None:
        check(lib().nb_device_synchronize(), "nb_device_synchronize")

    def free(self) -> None:
        self._free_graph()
        for b in self._pos + [self._vel]:
            b.free()
        if self._workspace is not None:
            lib().nb_free(self._workspace)
            self._workspace = None


class Event:
    def __init__(self):
        self.h = _vp()
        check(lib().nb_event_create(ctypes.byref(self.h)), "nb_event_create")

    def record(self, stream=None) -> None:
        check(lib().nb_event_record(self.h, stream), "nb_event_record")

    def synchronize(self) -> None:
        check(lib().nb_event_synchronize(self.h), "nb_event_synchronize")

    def elapsed_ms(self, stop: "Event") -> float:
        ms = _cf(0)
        check(lib().nb_event_elapsed_ms(ctypes.byref(ms), self.h, stop.h), "nb_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if self.h:
                lib().nb_event_destroy(self.h)
        except Exception:
            pass


class ShardedRank:
    """One rank of the body-sharded system THROUGH THE C-ABI: nb_comm_init_rank + nb_sharded_step_* (csrc/nbody_comm.hip),
    i.e. the product's own multi-GPU path -- RCCL send/recv rounds of position tiles on the communicator's side stream,
    the kernel of tile k waiting on tile k's event.  One process (or thread) per GPU; `unique_id` is rank 0's
    ``comm_unique_id()`` shipped to the other ranks by any means (bench.py: the torch.distributed store over gloo).

    The caller owns the arrays, exactly as with nb_integrate_*: `positions` = the two full-size ping-pong arrays (device
    addresses), `velocities`, `acc` (partial accelerations), all 4N T.  Python mirror of host/bodysystemhip_sharded.cpp's
    use of the same entry points (that class drives all local devices from one thread through the *_all form)."""

    def __init__(self, unique_id, world: int, rank: int, positions, velocities, acc, num_bodies: int, dtype=np.float32,
                 mode: int = NB_MODE_FAST, block_size: int = 256, stream=None, comm=None):
        """`comm`: an existing communicator to step ANOTHER system with (a communicator is not tied to a system size; the rank that
        created it keeps the ownership) -- otherwise nb_comm_init_rank makes one from `unique_id`."""
        self.dtype = np.dtype(dtype)
        self.world, self.rank, self.n = int(world), int(rank), int(num_bodies)
        if self.n % self.world:
            raise ValueError(f"{self.n} bodies do not shard evenly over {self.world} ranks; pad with zero-mass bodies")
        self.pos, self.vel, self.acc = [int(positions[0]), int(positions[1])], int(velocities), int(acc)
        self.mode, self.block_size, self.stream = mode, block_size, stream
        self.read = 0
        self.owns_comm = comm is None
        if comm is None:
            self.comm = _vp()
            check(lib().nb_comm_init_rank(ctypes.byref(self.comm), unique_id, self.world, self.rank), "nb_comm_init_rank")
        else:
            self.comm = comm
        f32 = self.dtype == np.float32
        self._step = lib().nb_sharded_step_f32 if f32 else lib().nb_sharded_step_f64
        self._tiles = lib().nb_exchange_tiles_f32 if f32 else lib().nb_exchange_tiles_f64
        self._scalar = np.float32 if f32 else float

    def workspace_bytes(self) -> int:
        """nb_comm_workspace_bytes_*: the scratch memory this rank can use in its mode (one rank: the single-GPU pairwise step;
        several: pairs once across the ranks, reaction sums sent to their owners); 0 = none."""
        need = _sz(0)
        fn = lib().nb_comm_workspace_bytes_f32 if self.dtype == np.float32 else lib().nb_comm_workspace_bytes_f64
        check(fn(self.comm, self.n, self.mode, ctypes.byref(need)), "nb_comm_workspace_bytes")
        return need.value

    def set_workspace(self, workspace, nbytes: int) -> None:
        """nb_comm_set_workspace: lend this rank the scratch memory of workspace_bytes().  With several ranks this is a COLLECTIVE
        over the communicator (every rank calls it; a rank without memory passes None, 0): the ranks learn the smallest amount
        lent anywhere, and the step is pairwise only if that suffices -- on every rank or on none."""
        check(lib().nb_comm_set_workspace(self.comm, workspace, nbytes), "nb_comm_set_workspace")

    def pairwise(self) -> bool:
        """nb_comm_layout_*: does nb_sharded_step_* of this communicator evaluate every pair once (the same answer on every rank)?"""
        flag = _ci(0)
        fn = lib().nb_comm_layout_f32 if self.dtype == np.float32 else lib().nb_comm_layout_f64
        check(fn(self.comm, self.n, self.mode, ctypes.byref(flag)), "nb_comm_layout")
        return bool(flag.value)

    def set_exchange_grouping(self, one_group: bool) -> None:
        """nb_comm_set_exchange_grouping: all G-1 position rounds of a step in one RCCL group (True) or a group and
        an event per round (False, the default since round 5); every rank must choose the same."""
        check(lib().nb_comm_set_exchange_grouping(self.comm, 1 if one_group else 0), "nb_comm_set_exchange_grouping")

    def exchange_grouping(self) -> bool:
        flag = _ci(0)
        check(lib().nb_comm_get_exchange_grouping(self.comm, ctypes.byref(flag)), "nb_comm_get_exchange_grouping")
        return bool(flag.value)

    def info(self) -> dict:
        """nb_comm_info + nb_comm_transport_info: what the communicator itself says about this rank and the RCCL it is bound to"""
        r, w, d = _ci(-1), _ci(-1), _ci(-1)
        check(lib().nb_comm_info(self.comm, ctypes.byref(r), ctypes.byref(w), ctypes.byref(d)), "nb_comm_info")
        out = {"rank": r.value, "world": w.value, "device": d.value}
        out.update(comm_transport_info(self.comm))
        hits = _ci(-1)
        check(lib().nb_comm_side_stream_collisions(self.comm, ctypes.byref(hits)), "nb_comm_side_stream_collisions")
        out["side_stream_collisions"] = hits.value  # (-1: never probed -- no pairwise step with two partners yet)
        bad = _ci(-1)
        check(lib().nb_comm_caller_stream_placement(self.comm, ctypes.byref(bad)), "nb_comm_caller_stream_placement")
        out["caller_stream_badly_placed"] = bad.value  # 1: stepping on the null stream or on its hardware queue (~40 % slower with RCCL active)
        return out

    def pair_work(self):
        """nb_comm_pair_work_* (tuning header): (pair evaluations, force launches) of this rank per pairwise step; None when the
        communicator steps one-sidedly"""
        evals, launches = ctypes.c_ulonglong(0), _ci(0)
        fn = lib().nb_comm_pair_work_f32 if self.dtype == np.float32 else lib().nb_comm_pair_work_f64
        rc = fn(self.comm, self.n, ctypes.byref(evals), ctypes.byref(launches))
        if rc == NB_ERR_UNSUPPORTED:
            return None
        check(rc, "nb_comm_pair_work")
        return evals.value, launches.value

    def make_step_stream(self):
        """nb_comm_stream_create: a well-placed non-blocking stream to step on, which becomes this rank's stream (the caller destroys
        it with nb_stream_destroy); returns it as a ctypes.c_void_p"""
        made = _vp()
        check(lib().nb_comm_stream_create(self.comm, ctypes.byref(made)), "nb_comm_stream_create")
        self.stream = made
        return made

    def update(self, delta_time, damping) -> None:
        """pos[1-read][own slice], vel[own slice] <- one step from pos[read]; then the tiles of pos[1-read] start moving."""
        check(self._step(self.comm, self.pos[1 - self.read], self.pos[self.read], self.vel, self.acc, self.n, self._scalar(delta_time),
                         self._scalar(damping), self.block_size, self.mode, self.stream), "nb_sharded_step")
        self.read = 1 - self.read

    def exchange_once(self, which: int | None = None) -> None:
        """One exchange of a position array outside any step (bring-up / diagnostics); the stream waits for all its tiles."""
        array = self.pos[self.read if which is None else which]
        check(self._tiles(self.comm, array, self.n, self.stream), "nb_exchange_tiles")
        self.finish()

    def finish(self) -> None:
        """Make the compute stream wait for every tile still in flight (asynchronous; synchronise the stream to block)."""
        check(lib().nb_exchange_wait_all(self.comm, self.stream), "nb_exchange_wait_all")

    def reaction_exchange_once(self) -> None:
        """nb_comm_reaction_exchange_* (tuning header): the reaction leg of a pairwise step alone; the stream waits for it."""
        fn = lib().nb_comm_reaction_exchange_f32 if self.dtype == np.float32 else lib().nb_comm_reaction_exchange_f64
        check(fn(self.comm, self.n, self.stream), "nb_comm_reaction_exchange")

    def destroy(self) -> None:
        if self.comm and self.owns_comm:
            lib().nb_comm_destroy(self.comm)
        self.comm = _vp()


def workspace_bytes(num_bodies: int, dtype=np.float32, mode: int = NB_MODE_FAST, max_bytes: int | None = None) -> int:
    """nb_workspace_bytes_* (max_bytes: nb_workspace_bytes_capped_*): scratch memory nb_integrate_ws_* wants for this system (0 = none)."""
    need = _sz(0)
    f32 = np.dtype(dtype) == np.float32
    if max_bytes is None:
        check((lib().nb_workspace_bytes_f32 if f32 else lib().nb_workspace_bytes_f64)(num_bodies, mode, ctypes.byref(need)), "nb_workspace_bytes")
    else:
        check((lib().nb_workspace_bytes_capped_f32 if f32 else lib().nb_workspace_bytes_capped_f64)(num_bodies, mode, max_bytes, ctypes.byref(need)), "nb_workspace_bytes_capped")
    return need.value


def comm_transport_info(comm) -> dict:
    """nb_comm_transport_info (tuning header): {"rccl_version": 22204, "rccl_library": ".../librccl.so.1"}; 0 / "" when the
    communicator has no transport bound (a world of one)."""
    version, path = _ci(0), ctypes.create_string_buffer(512)
    check(lib().nb_comm_transport_info(comm, ctypes.byref(version), path, len(path)), "nb_comm_transport_info")
    return {"rccl_version": version.value, "rccl_library": path.value.decode()}


def comm_unique_id() -> bytes:
    """nb_comm_unique_id: the 128 bytes rank 0 hands to every other rank."""
    buf = ctypes.create_string_buffer(128)
    check(lib().nb_comm_unique_id(buf), "nb_comm_unique_id")
    return buf.raw
