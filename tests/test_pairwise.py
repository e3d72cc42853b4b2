"""The pairwise FAST layout behind nb_integrate_ws_* (csrc/nbody_pair.hip): every pair of bodies evaluated once and applied to
both, the reaction sums through a caller-owned workspace.  Everything goes through the C-ABI; the yardstick is an fp64 direct
sum (oracle.accel_f64 / numpy long double), the same one the one-sided FAST kernel is held to -- the reference has no pairwise
kernel to compare with (bodysystemcuda.cu:125-146 evaluates every directed interaction), so parity here means: the same
accelerations as bodyBodyInteraction summed over all j, to fp32/fp64 summation accuracy.

A step with zero velocities, dt = 1 and damping = 1 leaves v = a: the accelerations are read straight from the velocity array.
"""
import ctypes

import numpy as np
import pytest

from conftest import xyz
from test_gpu_parity import direct_sum_f64

pytestmark = pytest.mark.gpu


def accel_ws(pkg, pos, dtype, workspace=True, softening=0.1):
    """accelerations of all bodies through nb_integrate_ws_* (FAST): one step from rest with dt = 1"""
    n = pos.size // 4
    params = pkg.NBodyParams(softening=softening)
    system = pkg.BodySystemHIP(n, 256, params, dtype, pos.astype(dtype), np.zeros(4 * n, dtype), mode=pkg.NB_MODE_FAST, workspace=workspace)
    system.update(dtype(1))
    acc = system.get_velocity().copy()
    new_pos = system.get_position().copy()
    system.free()
    return acc, new_pos


def random_bodies(oracle, n, seed=3, masses="unit"):
    oracle.srand(seed)
    pos, _ = oracle.randomise(0, n, 1.54, 8.0, np.float32)
    m = pos.reshape(n, 4)[:, 3]
    if masses == "ramp":
        m[:] = np.linspace(0.5, 2.0, n).astype(np.float32)
    elif masses == "species":  # three species in contiguous blocks whose borders fall inside tiles and blocks
        m[n // 3 + 5:] = 2.0
        m[2 * n // 3 + 11:] = 0.25
    elif masses == "odd":  # zero, negative and huge masses sprinkled in; the first body (the reference mass) stays 1
        m[1::7] = 0.0
        m[3::11] = -1.5
        m[5::13] = 1.0e6
    elif masses == "first_zero":  # a reference mass the kernel must not divide by
        m[0] = 0.0
    return pos


@pytest.mark.parametrize("plan", [(8, 8, 1), (8, 12, 2), (8, 4, 3), (4, 8, 1), (4, 8, 2), (4, 16, 1), (4, 4, 3), (2, 8, 1), (2, 16, 2), (1, 8, 1), (1, 4, 5)])
@pytest.mark.parametrize("n", [3000, 4096 + 64, 517])
def test_pair_force_error_every_geometry_fp32(gpu, oracle, plan, n):
    """Every (vectors per lane, waves, workgroups per block) instantiation, ragged body counts (a last block and a last tile
    that are partly empty, odd and even block counts, a single block), masses from 0.5 to 2: against the fp64 direct sum."""
    pos = random_bodies(oracle, n, masses="ramp")
    gpu.set_pair_plan_override(*plan, 1)
    try:
        assert gpu.pair_plan(n, np.float32).applies == 1
        acc, new_pos = accel_ws(gpu, pos, np.float32)
    finally:
        gpu.set_pair_plan_override(0, 0, 0, 0)
    ref = oracle.accel_f64(pos, 0, n)
    err = np.linalg.norm(xyz(acc) - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert err.max() < 5e-6, err.max()
    assert np.all(acc.reshape(n, 4)[:, 3] == 0)  # velocity .w is preserved (tipsy keeps eps there)
    assert np.all(new_pos.reshape(n, 4)[:, 3] == pos.reshape(n, 4)[:, 3])  # and so is the mass


@pytest.mark.parametrize("plan", [(0, 0, 0), (8, 8, 1), (8, 12, 2), (4, 8, 1)])
@pytest.mark.parametrize("masses", ["unit", "species", "odd", "first_zero"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_pair_mass_forms(gpu, oracle, dtype, masses, plan):
    """Unit-mass tiles (no mass multiply), tiles of mixed masses, zero / negative / huge masses, a first body of mass zero -- the four
    compiled loops (neither side multiplies / mixed bodies j / mixed bodies i / both) of the automatic plan at this size (R = 2) and of
    the large-system kernels (R = 8 with 8 and 12 waves, R = 4)."""
    n = 64 * 37 + 41
    pos = random_bodies(oracle, n, seed=5, masses=masses)
    gpu.set_pair_plan_override(*plan, 1)
    try:
        acc, _ = accel_ws(gpu, pos, dtype)
    finally:
        gpu.set_pair_plan_override(0, 0, 0, 0)
    eps2 = float(np.float64(np.float32(0.1)) ** 2) if dtype == np.float64 else float(np.float32(0.1) * np.float32(0.1))
    ref, size = direct_sum_f64(pos.reshape(n, 4).astype(np.float64), 0, n, 0, n, eps2)
    # error relative to the sum of the magnitudes of a body's terms (what rounding scales with: with +-1e6 masses the terms cancel)
    err = np.linalg.norm(xyz(acc).astype(np.float64) - ref, axis=1) / size
    assert err.max() < (2e-6 if dtype == np.float32 else 1e-14), err.max()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_pair_agrees_with_one_sided_kernel_and_is_reproducible(gpu, oracle, dtype):
    n = 20000
    pos0, vel0 = oracle.startup_state(n, np.float32)
    runs = []
    for ws in (True, True, False):
        s = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), dtype, pos0.astype(dtype), vel0.astype(dtype), mode=gpu.NB_MODE_FAST, workspace=ws)
        for _ in range(3):
            s.update(dtype(np.float32(0.016)))
        runs.append((s.get_position().copy(), s.get_velocity().copy()))
        s.free()
    assert runs[0][0].tobytes() == runs[1][0].tobytes() and runs[0][1].tobytes() == runs[1][1].tobytes()  # static work map: same bits every run
    tol = 2e-5 if dtype == np.float32 else 1e-12
    np.testing.assert_allclose(runs[0][0], runs[2][0], rtol=tol, atol=tol)
    assert runs[0][0].tobytes() != runs[2][0].tobytes() or dtype == np.float64  # (another summation order: not the same kernel)


def test_workspace_rules(gpu, oracle):
    """nb_workspace_bytes_* says 0 where a workspace is of no use; too small a workspace, none at all, or STRICT mode make
    nb_integrate_ws_* exactly nb_integrate_*; a workspace that overlaps the bodies is refused."""
    lib = gpu.lib()
    need = ctypes.c_size_t(1)
    gpu.check(lib.nb_workspace_bytes_f32(1024, gpu.NB_MODE_FAST, ctypes.byref(need)))
    assert need.value == 0  # too small a system to gain
    gpu.check(lib.nb_workspace_bytes_f32(65536, gpu.NB_MODE_STRICT, ctypes.byref(need)))
    assert need.value == 0  # STRICT keeps the CPU path's summation order
    gpu.check(lib.nb_workspace_bytes_f32(65536, gpu.NB_MODE_FAST, ctypes.byref(need)))
    plan = gpu.pair_plan(65536, np.float32)
    assert need.value == plan.workspace_bytes > 0 and plan.applies == 1
    assert plan.workspace_bytes == (plan.splits + plan.reaction_slots) * 3 * plan.blocks * plan.block_bodies * 4

    n = 32768
    pos0, vel0 = oracle.startup_state(n, np.float32)
    gpu.set_softening_squared(np.float32(0.1) * np.float32(0.1))
    bufs = [gpu.DeviceBuffer(pos0.nbytes) for _ in range(3)]
    gpu.check(lib.nb_workspace_bytes_f32(n, gpu.NB_MODE_FAST, ctypes.byref(need)))
    work = gpu.DeviceBuffer(need.value)

    def step(mode, workspace, nbytes):
        bufs[0].upload(pos0), bufs[2].upload(vel0)
        gpu.check(lib.nb_integrate_ws_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, np.float32(0.016), np.float32(1), n, 256, mode, workspace, nbytes, None), "nb_integrate_ws_f32")
        return bufs[1].download(np.zeros_like(pos0)).copy()

    def plain(mode):
        bufs[0].upload(pos0), bufs[2].upload(vel0)
        gpu.check(lib.nb_integrate_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, np.float32(0.016), np.float32(1), n, 256, mode, None))
        return bufs[1].download(np.zeros_like(pos0)).copy()

    one_sided = plain(gpu.NB_MODE_FAST)
    assert step(gpu.NB_MODE_FAST, None, 0).tobytes() == one_sided.tobytes()
    assert step(gpu.NB_MODE_FAST, work.ptr, 4096).tobytes() == one_sided.tobytes()  # far too few bytes for any form of the layout
    # a few bytes short of the request: the tournament cut into the fewest slices that still fit (round 4) -- or, if no sliced form
    # is smaller than the request at this size, the one-sided kernel; nb_workspace_bytes_capped_* says which
    short = step(gpu.NB_MODE_FAST, work.ptr, need.value - 4)
    if gpu.workspace_bytes(n, max_bytes=need.value - 4) == 0:
        assert short.tobytes() == one_sided.tobytes()
    else:
        assert short.tobytes() != one_sided.tobytes()
        np.testing.assert_allclose(short, one_sided, rtol=1e-5, atol=1e-5)
    assert step(gpu.NB_MODE_STRICT, work.ptr, need.value).tobytes() == plain(gpu.NB_MODE_STRICT).tobytes()
    pairwise = step(gpu.NB_MODE_FAST, work.ptr, need.value)
    assert pairwise.tobytes() != one_sided.tobytes()
    np.testing.assert_allclose(pairwise, one_sided, rtol=1e-5, atol=1e-5)
    # the workspace may not overlap the bodies the launch reads
    assert lib.nb_integrate_ws_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, np.float32(0.016), np.float32(1), n, 256, gpu.NB_MODE_FAST, bufs[0].ptr, need.value, None) == 10001
    assert lib.nb_integrate_ws_f32(bufs[0].ptr, bufs[0].ptr, bufs[2].ptr, np.float32(0.016), np.float32(1), n, 256, gpu.NB_MODE_FAST, work.ptr, need.value, None) == 10001
    # ... nor anything else the launch writes (pair_forces' reaction planes on the arrays pair_finish reads and writes):
    # a workspace that starts inside the velocities or the new positions, a workspace that ENDS inside them, velocities on the new positions
    big = gpu.DeviceBuffer(need.value + 2 * pos0.nbytes)  # [workspace-sized gap][one body array] : lets a workspace end inside an array
    tail_array = ctypes.c_void_p(big.ptr.value + need.value - 4096)
    args = (np.float32(0.016), np.float32(1), n, 256, gpu.NB_MODE_FAST)
    assert lib.nb_integrate_ws_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, *args, bufs[2].ptr, need.value, None) == 10001
    assert lib.nb_integrate_ws_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, *args, ctypes.c_void_p(bufs[1].ptr.value + 64), need.value, None) == 10001
    assert lib.nb_integrate_ws_f32(bufs[1].ptr, bufs[0].ptr, tail_array, *args, big.ptr, need.value, None) == 10001
    assert lib.nb_integrate_ws_f32(tail_array, bufs[0].ptr, bufs[2].ptr, *args, big.ptr, need.value, None) == 10001
    assert lib.nb_integrate_ws_f32(bufs[1].ptr, bufs[0].ptr, bufs[1].ptr, *args, work.ptr, need.value, None) == 10001
    # the same arrays with the workspace next to them (not inside) are fine
    after = ctypes.c_void_p(big.ptr.value + need.value)
    gpu.check(lib.nb_integrate_ws_f32(bufs[1].ptr, bufs[0].ptr, after, *args, big.ptr, need.value, None), "disjoint parts of one allocation")
    gpu.check(lib.nb_device_synchronize())
    big.free()
    for b in bufs + [work]:
        b.free()


def test_pair_graph_replay_equals_the_step_loop(gpu, oracle):
    n = 16384
    pos0, vel0 = oracle.startup_state(n, np.float32)
    a = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), np.float32, pos0, vel0, mode=gpu.NB_MODE_FAST, workspace=True)
    b = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), np.float32, pos0, vel0, mode=gpu.NB_MODE_FAST, workspace=True)
    assert a._workspace is not None
    for _ in range(6):
        a.update(np.float32(0.016))
    b.update_many(np.float32(0.016), 6)
    assert a.get_position().tobytes() == b.get_position().tobytes() and a.get_velocity().tobytes() == b.get_velocity().tobytes()
    a.free(), b.free()


@pytest.mark.parametrize("n,dtype", [(65536, np.float32), (262144, np.float64), (16384, np.float32)])
def test_clocked_variant_of_the_forces_kernel_changes_no_bit(gpu, oracle, n, dtype):
    """Round 6: while nb_set_pair_clock_words lends device memory, the one-GPU pairwise step launches pair_forces_clocked in pair_forces'
    place -- the same kernel text (csrc/nbody_pair_forces.inc, included twice) + four scalar instructions: every workgroup notes its
    lifetime on the shader-cycle counter and on the constant 100 MHz counter.  It is what bench.py reads the delivered clock from, so
    it must compute what pair_forces computes, bit for bit; the words must be plausible (a clock between 1.0 and 2.6 GHz, a lifetime
    that matches the kernel's duration); geometries without the variant (R < 8: 16 384 bodies) ignore the words and leave them alone;
    too few bytes for the launch's workgroups: ignored as well."""
    lib = gpu.lib()
    pos0, vel0 = oracle.startup_state(n, dtype)
    plan = gpu.pair_plan(n, dtype)
    has_variant = plan.waves_per_block == 8 and plan.bodies_per_lane == (16 if dtype == np.float32 else 8) and plan.slices == 1
    assert has_variant == (n >= 65536)
    dt = dtype(np.float32(0.016))
    plain = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), dtype, pos0, vel0, mode=gpu.NB_MODE_FAST, workspace=True)
    clocked = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), dtype, pos0, vel0, mode=gpu.NB_MODE_FAST, workspace=True)
    words = gpu.DeviceBuffer(plan.grid_blocks * 16)
    gpu.check(lib.nb_memset(words.ptr, 0, plan.grid_blocks * 16, None))
    try:
        for _ in range(3):
            plain.update(dt)
        gpu.check(lib.nb_set_pair_clock_words(words.ptr, plan.grid_blocks * 16 - 16), "nb_set_pair_clock_words")  # one workgroup short: not used
        clocked.update(dt)
        gpu.check(lib.nb_device_synchronize())
        assert not words.download(np.zeros(plan.grid_blocks * 2, np.uint64)).any()
        gpu.check(lib.nb_set_pair_clock_words(words.ptr, plan.grid_blocks * 16), "nb_set_pair_clock_words")
        e0, e1 = gpu.Event(), gpu.Event()
        gpu.check(lib.nb_set_pair_probe_event(e1.h))
        clocked.update(dt)
        e0.record(None)
        clocked.update(dt)
        gpu.check(lib.nb_device_synchronize())
        forces_ms = e0.elapsed_ms(e1)
    finally:
        gpu.check(lib.nb_set_pair_probe_event(None))
        gpu.check(lib.nb_set_pair_clock_words(None, 0), "nb_set_pair_clock_words")
    assert clocked.get_position().tobytes() == plain.get_position().tobytes() and clocked.get_velocity().tobytes() == plain.get_velocity().tobytes()
    got = words.download(np.zeros(plan.grid_blocks * 2, np.uint64)).reshape(-1, 2)
    if has_variant:
        cycles, ticks = got[:, 0].astype(np.float64), got[:, 1].astype(np.float64)
        assert (ticks > 0).all() and (cycles > 0).all()
        mhz = 100.0 * cycles / ticks
        assert 1000 < np.median(mhz) < 2600, np.median(mhz)
        rounds = -(-plan.grid_blocks // 256)  # (eight-wave workgroups: one per CU at a time -- 512 workgroups of the fp64 plan take two rounds)
        assert 0.5 * forces_ms < rounds * np.median(ticks) / 1e5 <= 1.05 * forces_ms  # a workgroup's lifetime in ms (100 MHz ticks) x rounds against the kernel's duration by events
    else:
        assert not got.any()
    plain.free(), clocked.free(), words.free()


@pytest.mark.parametrize("n,dtype", [(262144, np.float32), (1048576, np.float32), (262144, np.float64)])
def test_pair_full_size_sampled_forces_and_momentum(gpu, oracle, n, dtype):
    """BASELINE sizes: accelerations of 512 sampled bodies against the fp64 direct sum, and a size-independent property the
    pairwise evaluation has by construction: what body i feels from j is, term by term, the opposite of what j feels from i,
    so the total momentum change sum(m a) vanishes to summation accuracy."""
    pos0, _ = oracle.startup_state(n, np.float32)
    acc, _ = accel_ws(gpu, pos0, dtype)
    a = xyz(acc).astype(np.float64)
    rng = np.random.default_rng(7)
    sample = np.sort(rng.choice(n, 512, replace=False))
    ref = np.stack([oracle.accel_f64(pos0, int(i), 1)[0] for i in sample])
    err = np.linalg.norm(a[sample] - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert err.max() < (1e-5 if dtype == np.float32 else 1e-10), err.max()
    total = np.abs(a.sum(axis=0)).max() / np.abs(a).sum(axis=0).max()
    assert total < (1e-6 if dtype == np.float32 else 1e-14), total


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,plan", [(600, (1, 4, 5)), (8256, (2, 8, 4)), (20000, (0, 0, 0)), (20000, (4, 4, 3)), (20000, (8, 12, 2)), (65536, (0, 0, 0)), (100032, (0, 0, 0)), (100032, (2, 16, 2)),
                                    (262144, (0, 0, 0))])
def test_pair_reads_nothing_it_did_not_write(gpu, oracle, dtype, n, plan):
    """The workspace carries nothing from launch to launch: a step over a workspace filled with NaN bit patterns gives the bits
    of a step over a zeroed one (every slot the second kernel adds was written by the first in the same step, whatever the
    plan, ragged sizes included) -- so a caller may hand over uninitialised memory, and share one workspace between systems."""
    lib = gpu.lib()
    f32 = dtype == np.float32
    pos0, vel0 = oracle.startup_state(n, np.float32)
    pos0, vel0 = pos0.astype(dtype), vel0.astype(dtype)
    gpu.set_softening_squared(dtype(np.float32(0.1)) * dtype(np.float32(0.1)))
    gpu.set_pair_plan_override(plan[0], plan[1], plan[2], 256 if any(plan) else 0)
    try:
        need = ctypes.c_size_t(0)
        gpu.check((lib.nb_workspace_bytes_f32 if f32 else lib.nb_workspace_bytes_f64)(n, gpu.NB_MODE_FAST, ctypes.byref(need)))
        assert need.value > 0
        bufs = [gpu.DeviceBuffer(pos0.nbytes) for _ in range(3)]
        work = gpu.DeviceBuffer(need.value)
        step = lib.nb_integrate_ws_f32 if f32 else lib.nb_integrate_ws_f64
        out = []
        for fill in (0x00, 0xFF, 0x7F):  # zeros; NaN; NaN again (0x7f7f7f7f is a large finite fp32 -- as wrong as any when added)
            bufs[0].upload(pos0), bufs[2].upload(vel0)
            gpu.check(lib.nb_memset(work.ptr, fill, need.value, None), "nb_memset")
            for k in range(2):
                gpu.check(step(bufs[1 - k].ptr, bufs[k].ptr, bufs[2].ptr, dtype(np.float32(0.016)), dtype(1), n, 256, gpu.NB_MODE_FAST, work.ptr, need.value, None), "nb_integrate_ws")
            out.append((bufs[0].download(np.zeros_like(pos0)).copy(), bufs[2].download(np.zeros_like(vel0)).copy()))
        assert np.isfinite(out[1][0]).all() and np.isfinite(out[1][1]).all()
        for p, v in out[1:]:
            assert p.tobytes() == out[0][0].tobytes() and v.tobytes() == out[0][1].tobytes()
        for b in bufs + [work]:
            b.free()
    finally:
        gpu.set_pair_plan_override(0, 0, 0, 0)


# ------------------------------------------------------------------------------------------------ bounded workspace: K slices
class sliced:
    """nb_set_pair_slices_override(K) + the pairwise layout forced at any size, for the duration of a block"""

    def __init__(self, gpu, slices, plan=(0, 0, 0)):
        self.gpu, self.slices, self.plan = gpu, slices, plan

    def __enter__(self):
        self.gpu.set_pair_plan_override(*self.plan, 1)
        self.gpu.set_pair_slices_override(self.slices)

    def __exit__(self, *exc):
        self.gpu.set_pair_slices_override(0)
        self.gpu.set_pair_plan_override(0, 0, 0, 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("slices,n,plan", [(2, 3000, (2, 8, 1)), (3, 3000, (1, 4, 2)), (4, 4096 + 64, (2, 8, 2)), (5, 20000, (4, 8, 1)), (8, 20000, (2, 16, 1)), (15, 9000, (1, 8, 1)),
                                           (2, 700, (4, 8, 1)), (3, 130, (1, 4, 1))])
def test_sliced_pairwise_force_error(gpu, oracle, dtype, slices, n, plan):
    """The tournament cut into K slices that share one region of reaction planes (csrc/nbody_pair.hip launch_pair_sliced): the
    diagonal of every slice, the rectangles against the next K/2 slices -- for an even K the rectangle at distance K/2 split
    between its two partners --, every launch folded into the receiving slice's arrays, K finish launches.  Even and odd K, ragged
    last slices, slices of a single block (no diagonal reaction slots, an empty half of a split rectangle), masses from 0.5 to 2:
    the accelerations of ALL bodies against the fp64 direct sum, to the tolerance of the single tournament."""
    pos = random_bodies(oracle, n, masses="ramp")
    with sliced(gpu, slices, plan):
        p = gpu.pair_plan(n, dtype)
        assert p.applies == 1 and 2 <= p.slices <= slices
        acc, new_pos = accel_ws(gpu, pos, dtype)
    # (the bodies are fp32 values in both precisions, so the oracle's fp64 sum over fp32 positions is the yardstick for both; its
    # softening^2 is double(0.1f)^2, the fp64 kernels' -- the fp32 kernels use float(0.1f * 0.1f), 3e-9 away: far inside 5e-6)
    ref = oracle.accel_f64(pos, 0, n)
    err = np.linalg.norm(xyz(acc).astype(np.float64) - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert err.max() < (5e-6 if dtype == np.float32 else 1e-12), err.max()
    assert np.all(acc.reshape(n, 4)[:, 3] == 0) and np.all(new_pos.reshape(n, 4)[:, 3] == pos.reshape(n, 4)[:, 3].astype(dtype))


@pytest.mark.parametrize("slices", [2, 3, 4, 7])
def test_sliced_pairwise_reads_nothing_it_did_not_write_and_is_reproducible(gpu, oracle, slices):
    """A step over a NaN-filled workspace gives the bits of a step over a zeroed one, for K > 1 too (the reusable region, the
    received arrays of slices whose diagonal has no slots, the windows of split rectangles); two runs give the same bits; and the
    result agrees with the single tournament to summation order."""
    lib = gpu.lib()
    n = 20000
    pos0, vel0 = oracle.startup_state(n, np.float32)
    gpu.set_softening_squared(np.float32(0.1) * np.float32(0.1))
    out = []
    with sliced(gpu, slices):
        need = gpu.workspace_bytes(n)
        assert need > 0 and gpu.pair_plan(n).slices == slices
        bufs = [gpu.DeviceBuffer(pos0.nbytes) for _ in range(3)]
        work = gpu.DeviceBuffer(need)
        for fill in (0x00, 0xFF, 0x7F):
            bufs[0].upload(pos0), bufs[2].upload(vel0)
            gpu.check(lib.nb_memset(work.ptr, fill, need, None), "nb_memset")
            for k in range(2):
                gpu.check(lib.nb_integrate_ws_f32(bufs[1 - k].ptr, bufs[k].ptr, bufs[2].ptr, np.float32(0.016), np.float32(1), n, 256, gpu.NB_MODE_FAST, work.ptr, need, None), "nb_integrate_ws")
            out.append((bufs[0].download(np.zeros_like(pos0)).copy(), bufs[2].download(np.zeros_like(vel0)).copy()))
        for b in bufs + [work]:
            b.free()
    assert np.isfinite(out[1][0]).all()
    for p, v in out[1:]:
        assert p.tobytes() == out[0][0].tobytes() and v.tobytes() == out[0][1].tobytes()
    single = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), np.float32, pos0, vel0, mode=gpu.NB_MODE_FAST, workspace=True)
    assert gpu.pair_plan(n).slices == 1
    for _ in range(2):
        single.update(np.float32(0.016))
    one = single.get_position().copy()
    single.free()
    assert one.tobytes() != out[0][0].tobytes()
    np.testing.assert_allclose(out[0][0], one, rtol=2e-5, atol=2e-5)


def test_sliced_pairwise_by_the_bytes_on_offer(gpu, oracle):
    """nb_integrate_ws_* takes the fewest slices that fit the bytes it is handed: the full request -> one tournament; what
    nb_workspace_bytes_capped_* names for a smaller budget -> that sliced form; one byte less than any form -> the one-sided kernel.
    And BodySystemHIP(workspace_cap=...) / the hipGraph form run the sliced step."""
    lib = gpu.lib()
    n = 262144  # (slicing pays where a block needs no help to fill the chip: at small sizes the sliced forms want MORE memory than one tournament)
    pos0, vel0 = oracle.startup_state(n, np.float32)
    gpu.set_softening_squared(np.float32(0.1) * np.float32(0.1))
    full = gpu.workspace_bytes(n)
    assert gpu.pair_plan(n).slices == 1 and full > 0
    # the sizes the library names for smaller and smaller budgets: each within its budget, never growing, at least two sliced forms
    named = [gpu.workspace_bytes(n, max_bytes=int(full * f)) for f in (0.9, 0.6, 0.4, 0.3, 0.25)]
    assert all(0 <= b <= int(full * f) for b, f in zip(named, (0.9, 0.6, 0.4, 0.3, 0.25))) and all(a >= b for a, b in zip(named, named[1:]))
    forms = sorted({b for b in named if b}, reverse=True)
    assert len(forms) >= 2, named
    third, smallest = forms[0], forms[-1]
    bufs = [gpu.DeviceBuffer(pos0.nbytes) for _ in range(3)]
    work = gpu.DeviceBuffer(full)

    def step(nbytes):
        bufs[0].upload(pos0), bufs[2].upload(vel0)
        gpu.check(lib.nb_integrate_ws_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, np.float32(0.016), np.float32(1), n, 256, gpu.NB_MODE_FAST, work.ptr, nbytes, None), "nb_integrate_ws_f32")
        return bufs[1].download(np.zeros_like(pos0)).copy()

    bufs[0].upload(pos0), bufs[2].upload(vel0)
    gpu.check(lib.nb_integrate_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, np.float32(0.016), np.float32(1), n, 256, gpu.NB_MODE_FAST, None))
    one_sided = bufs[1].download(np.zeros_like(pos0)).copy()
    results = {nbytes: step(nbytes) for nbytes in (full, full - 4, third, third - 4, smallest, smallest - 4)}
    assert results[full].tobytes() != results[full - 4].tobytes()          # one tournament / a sliced form
    assert results[smallest].tobytes() != results[third].tobytes()         # more slices
    assert len({r.tobytes() for r in results.values()}) >= 4
    for r in results.values():
        np.testing.assert_allclose(r, one_sided, rtol=2e-5, atol=2e-5)
    tiny = gpu.workspace_bytes(n, max_bytes=1 << 16)
    assert tiny == 0 and step(1 << 16).tobytes() == one_sided.tobytes()     # nothing fits: exactly nb_integrate_*
    for b in bufs + [work]:
        b.free()
    a = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), np.float32, pos0, vel0, mode=gpu.NB_MODE_FAST, workspace=True, workspace_cap=third)
    b = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), np.float32, pos0, vel0, mode=gpu.NB_MODE_FAST, workspace=True, workspace_cap=third)
    assert a._workspace_bytes == third
    for _ in range(4):
        a.update(np.float32(0.016))
    b.update_many(np.float32(0.016), 4)
    assert a.get_position().tobytes() == b.get_position().tobytes() and a.get_velocity().tobytes() == b.get_velocity().tobytes()
    a.free(), b.free()


def test_sliced_pairwise_4mi_bodies_in_16_gb(gpu, O):
    """4 194 304 bodies -- one tournament would want 103 GB of reaction slots -- step pairwise inside 16 GB:
    sampled accelerations against the fp64 direct sum, total momentum change zero to summation accuracy."""
    import os

    n = 4 * 1048576
    omp = O.Oracle(openmp=True)
    omp.set_num_threads(min(16, os.cpu_count() or 1))
    pos0, _ = omp.startup_state(n, np.float32)
    need = gpu.workspace_bytes(n, max_bytes=16 << 30)
    plan = gpu.pair_plan(n)
    assert 0 < need <= (16 << 30) and plan.applies == 1 and plan.slices >= 2
    system = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), np.float32, pos0, np.zeros_like(pos0), mode=gpu.NB_MODE_FAST, workspace=True, workspace_cap=16 << 30)
    assert system._workspace_bytes == need
    system.update(np.float32(1))
    acc = xyz(system.get_velocity().copy()).astype(np.float64)
    system.free()
    rng = np.random.default_rng(11)
    sample = np.sort(rng.choice(n, 96, replace=False))
    ref = np.stack([omp.accel_f64(pos0, int(i), 1)[0] for i in sample])
    err = np.linalg.norm(acc[sample] - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert err.max() < 1e-5, err.max()
    total = np.abs(acc.sum(axis=0)).max() / np.abs(acc).sum(axis=0).max()
    assert total < 1e-6, total


def test_pair_default_plan_at_awkward_body_counts(gpu, O):
    """The automatic plan at body counts that are not powers of two (round 4: R and any C up to 16 chosen so that the launch fills
    whole rounds of 256 workgroups -- 5, 7, 11, 13 and 15 workgroups per block among these sizes --, ragged last blocks and tiles,
    odd and even block counts): sampled
    accelerations against the fp64 direct sum and the pairwise layout's built-in property, total momentum change = 0; and the same
    sizes through the bounded-workspace form with the fewest slices that fit a quarter of the full request."""
    import os

    omp = O.Oracle(openmp=True)
    omp.set_num_threads(min(16, os.cpu_count() or 1))
    rng = np.random.default_rng(2024)
    sizes = [9001, 12345, 20000, 33333, 65537, 100000, 131071, 196609, 300000] + [int(x) for x in rng.integers(9000, 400000, 6)]
    seen = set()
    for n in sizes:
        pos0, _ = omp.startup_state((n + 7) // 8 * 8, np.float32)
        pos0 = pos0[:4 * n].copy()
        plan = gpu.pair_plan(n)
        assert plan.applies == 1 and plan.slices == 1
        seen.add(plan.splits)
        full = gpu.workspace_bytes(n)
        for cap in (None, full // 4):
            if cap is not None and gpu.workspace_bytes(n, max_bytes=cap) == 0:
                continue  # (small systems: no sliced form is that much smaller)
            system = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), np.float32, pos0, np.zeros_like(pos0), mode=gpu.NB_MODE_FAST, workspace=True, workspace_cap=cap)
            system.update(np.float32(1))
            acc = xyz(system.get_velocity().copy()).astype(np.float64)
            system.free()
            sample = np.sort(rng.choice(n, 48, replace=False))
            sample[0], sample[-1] = 0, n - 1
            ref = np.stack([omp.accel_f64(pos0, int(i), 1)[0] for i in sample])
            err = np.linalg.norm(acc[sample] - ref, axis=1) / np.linalg.norm(ref, axis=1)
            assert err.max() < 1e-5, (n, cap, err.max())
            total = np.abs(acc.sum(axis=0)).max() / np.abs(acc).sum(axis=0).max()
            assert total < 1e-6, (n, cap, total)
    assert len(seen) >= 6 and any(c > 1 and c % 2 for c in seen), seen  # many values of C were exercised, odd ones among them (any C up to 16 is a plan)


@pytest.mark.parametrize("seed", [11, 12])
def test_pair_random_geometries_against_the_one_sided_kernel(gpu, seed):
    """Seeded random cases over everything a plan may be -- R in {1, 2, 4, 8}, 4 / 8 / 12 / 16 waves, ANY number of workgroups per
    block up to 16 (the automatic plan takes odd ones since round 4), one tournament or 2 ... 5 slices, fp32 and fp64, 1 ... 9 000 bodies,
    equal / scaled / ramped / two-species / partly massless bodies: the accelerations of nb_integrate_ws_* against those of the
    one-sided kernel (nb_integrate_*) on the same bodies, and no NaN anywhere.  (The one-sided kernel is held to the fp64 direct sum
    in test_gpu_parity.py; this test is about indexing: every unit lands on a wave, every sum in its plane.)"""
    rng = np.random.default_rng(seed)

    def accel(pos, dtype, workspace):
        n = pos.size // 4
        s = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), dtype, pos.astype(dtype), np.zeros(4 * n, dtype), mode=gpu.NB_MODE_FAST, workspace=workspace)
        s.update(dtype(1))
        a = s.get_velocity().copy()
        s.free()
        return xyz(a).astype(np.float64)

    ran_sliced = 0
    try:
        for case in range(70):
            n = int(rng.integers(1, 9000)) if case % 3 else int(rng.integers(1, 400))
            R, S, C = int(rng.choice([1, 2, 4, 8])), int(rng.choice([4, 8, 12, 16])), int(rng.integers(1, 17))
            if R == 8 and S == 16:
                S = 8  # (196 KB of LDS: not a launch)
            slices = int(rng.integers(2, 6)) if case % 4 == 3 else 0
            dtype = np.float32 if case % 5 else np.float64
            pos = rng.uniform(-5, 5, size=(n, 4)).astype(np.float32)
            kind = case % 6
            pos[:, 3] = 2.5 if kind == 1 else 1.0
            if kind == 2:
                pos[:, 3] = rng.uniform(0.5, 2.0, n)
            if kind == 3:
                pos[n // 2:, 3] = 3.0
            if kind == 4:
                pos[rng.integers(0, n, max(1, n // 10)), 3] = 0.0
            gpu.set_pair_plan_override(R, S, C, 1)
            gpu.set_pair_slices_override(slices)
            plan = gpu.pair_plan(n, dtype)
            ran_sliced += plan.slices > 1
            pairwise = accel(pos.ravel(), dtype, True)
            gpu.set_pair_plan_override(0, 0, 0, 0)
            gpu.set_pair_slices_override(0)
            one_sided = accel(pos.ravel(), dtype, False)
            assert np.isfinite(pairwise).all(), (case, n, (R, S, C), slices)
            err = np.abs(pairwise - one_sided).max() / (np.abs(one_sided).max() + 1e-30)
            assert err < (2e-5 if dtype == np.float32 else 1e-12), (case, n, (R, S, C), slices, np.dtype(dtype).name, kind, err, plan.applies)
    finally:
        gpu.set_pair_plan_override(0, 0, 0, 0)
        gpu.set_pair_slices_override(0)
    assert ran_sliced >= 5
