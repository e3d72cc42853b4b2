"""The HEADLINE kernel -- FAST with a workspace, the pairwise layout (csrc/nbody_pair.hip: pair_forces + pair_finish) -- held
to the CPU path's trajectories on one GPU, with exactly the envelope the one-sided FAST kernel is held to in
tests/test_gpu_parity.py.  The reference's own acceptance check compares the GPU system's positions with the CPU system's
after stepping both from one state (/root/reference/src/nbody/compute_cuda.cpp:294-333); the arithmetic being compared with is
BodySystemCPU<T>::update (bodysystemcpu.cpp:149-243 fp32, :245-299 fp64) as restated in oracle/nbody_oracle.c.
PARITY UNPINNED: every golden trajectory used here (tests/golden/*.npz, the 16 384-body compact fixtures included) is that
restatement's own output -- the reference holds no fixture for `update` and its translation unit cannot be built unmodified in
this image (oracle section); only the initial states are pinned to reference code (randomise_bodies).

The layout applies by default above 8 192 bodies (fp32) / above 6 144 (fp64), and the committed full trajectories stop at
4 096 bodies, so: (a) the layout is FORCED at the golden sizes, in every compiled geometry (nb_set_pair_plan_override with
min_bodies = 1); (b) a fixture at 16 384 bodies, where the layout applies by itself, pins the default plan; (c) the
fp64-truth and conservation checks of the one-sided kernel run here through nb_integrate_ws_* too.  (BASELINE sizes: the FAST half
of test_gpu_parity.py::test_full_size_strict_bitwise_on_a_sample runs both layouts against STRICT on all bodies.)
"""
import numpy as np
import pytest

from conftest import golden_steps, load_golden, load_golden_compact, xyz
from test_gpu_parity import DT, rel_err

pytestmark = pytest.mark.gpu

# (vectors per lane R, waves per workgroup S, workgroups per block C): every instantiation launch_pair_tile can pick
PLANS_F32 = [(8, 8, 1), (8, 12, 2), (8, 4, 1), (4, 8, 1), (4, 8, 2), (4, 16, 1), (4, 4, 3), (2, 8, 1), (2, 16, 2), (1, 8, 1), (1, 4, 5)]
PLANS_F64 = [(8, 8, 1), (8, 12, 1), (8, 4, 2), (4, 8, 1), (4, 16, 2), (2, 8, 1), (2, 4, 3), (1, 8, 2), (1, 16, 1)]


class forced_pairwise:
    """nb_set_pair_plan_override(R, S, C, min_bodies = 1) for the duration of a block"""

    def __init__(self, gpu, plan=(0, 0, 0)):
        self.gpu, self.plan = gpu, plan

    def __enter__(self):
        self.gpu.set_pair_plan_override(*self.plan, 1)

    def __exit__(self, *exc):
        self.gpu.set_pair_plan_override(0, 0, 0, 0)


def pairwise_system(gpu, n, dtype, pos0, vel0):
    system = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), dtype, pos0, vel0, mode=gpu.NB_MODE_FAST, workspace=True)
    assert gpu.pair_plan(n, dtype).applies == 1
    assert system._workspace is not None or gpu.pair_plan(n, dtype).workspace_bytes == 0
    return system


def trajectory_errors(gpu, n, dtype, g, steps_wanted):
    system = pairwise_system(gpu, n, dtype, g["pos_0"], g["vel_0"])
    done, errs = 0, {}
    for s in steps_wanted:
        for _ in range(s - done):
            system.update(dtype(DT))
        done = s
        errs[s] = rel_err(system.get_position(), g[f"pos_{s}"])
    vel = system.get_velocity().copy()
    system.free()
    return errs, vel


def assert_fp32_envelope(errs):
    """the one-sided kernel's bars (test_fast_fp32_vs_golden), unchanged"""
    assert errs[1].max() <= 2e-6, errs[1].max()
    assert errs[10].max() <= 2e-5, errs[10].max()
    if 100 in errs:
        assert np.median(errs[100]) <= 5e-5, np.median(errs[100])
        assert np.percentile(errs[100], 99) <= 2e-3, np.percentile(errs[100], 99)
        assert errs[100].max() <= 1e-2, errs[100].max()


def assert_fp64_envelope(errs):
    """the one-sided kernel's bars (test_fast_fp64_vs_golden), unchanged"""
    assert errs[1].max() <= 1e-14, errs[1].max()
    assert errs[10].max() <= 1e-12, errs[10].max()
    if 100 in errs:
        assert errs[100].max() <= 1e-8, errs[100].max()


# ------------------------------------------------------------------------------- (a) forced at the golden sizes
@pytest.mark.parametrize("plan", PLANS_F32)
@pytest.mark.parametrize("n", [256, 1024, 4096])
def test_pairwise_fp32_vs_golden_every_geometry(gpu, n, plan):
    """pair_forces<float, R, S> with C workgroups per block + pair_finish against the CPU path's trajectory at 1 / 10 / 100
    steps (4 096 bodies: 1 / 10).  256 bodies with R = 4 is a single block (no reaction slots at all), 1 024 with R = 4 two
    blocks (an even tournament: q = NB/2 keeps only the i side)."""
    g = load_golden(n, "f32")
    with forced_pairwise(gpu, plan):
        errs, _ = trajectory_errors(gpu, n, np.float32, g, golden_steps(g))
    assert_fp32_envelope(errs)


@pytest.mark.parametrize("plan", PLANS_F64)
@pytest.mark.parametrize("n", [256, 1024, 4096])
def test_pairwise_fp64_vs_golden_every_geometry(gpu, n, plan):
    g = load_golden(n, "f64")
    with forced_pairwise(gpu, plan):
        errs, _ = trajectory_errors(gpu, n, np.float64, g, golden_steps(g))
    assert_fp64_envelope(errs)


def _dealing(n, R, S, C):
    """launch_pair_tile's rule for the diagonal launch of an fp32 system (csrc/nbody_pair.hip, nbody_kernels.h), restated: 2 = equal
    whole units + QUARTERS of the units left over (needs LDS for the quarters' sums), 1 / 0 = every unit whole, slots interleaved /
    blocked -- what is left when 3 072 B per tail unit no longer fit under the 160 KiB"""
    block = 64 * R * 2
    blocks = -(-n // block)
    units = (blocks // 2 + 1) * R * 2
    slots = C * S
    tail = -(-(units % slots) // C)
    lds = S * 3 * R * 2 * 64 * 4 + 256 + tail * 4 * 3 * 64 * 4
    if lds <= 160 * 1024:
        return 2, tail
    return (1 if (C > 1 and units % slots != 0 and units // slots < 34) else 0), tail


@pytest.mark.parametrize("n,plan,deal", [(2048, (8, 12, 1), 0), (3072, (8, 12, 1), 0), (1024, (8, 12, 2), 1), (7168, (8, 12, 2), 1), (16384, (8, 12, 7), 1), (16384, (8, 12, 5), 2)])
def test_pairwise_whole_unit_dealing_when_the_lds_has_no_room_for_quarters(gpu, oracle, n, plan, deal):
    """twelve waves of R = 8 leave 16 128 B of the 160 KiB for the quarters' sums -- five tail units; with more, the
    PRODUCT build deals every unit whole (PairArgs::deal 0: blocked, one workgroup per block; 1: interleaved), the branch that
    otherwise only the NB_PAIR_NO_QUARTERS build takes.  Each case is checked to take the dealing it is here for, then held to the
    CPU path over 1 and 10 steps like every other geometry."""
    assert _dealing(n, *plan)[0] == deal, _dealing(n, *plan)
    assert (_dealing(n, *plan)[1] > 5) == (deal != 2)
    pos0, vel0 = oracle.startup_state(n, np.float32)
    ref = {0: (pos0.copy(), vel0.copy())}
    p, v = pos0.copy(), vel0.copy()
    for s in (1, 10):
        oracle.update(p, v, DT, steps=s - max(k for k in ref if k < s))
        ref[s] = (p.copy(), v.copy())
    g = {"pos_0": pos0, "vel_0": vel0, "pos_1": ref[1][0], "pos_10": ref[10][0]}
    with forced_pairwise(gpu, plan):
        errs, _ = trajectory_errors(gpu, n, np.float32, g, [1, 10])
        again, _ = trajectory_errors(gpu, n, np.float32, g, [1, 10])
    assert_fp32_envelope(errs)
    assert errs[10].tobytes() == again[10].tobytes()  # (bit-reproducible: the dealing is a fixed function of the geometry)


@pytest.mark.parametrize("n", [8, 63, 200])
def test_pairwise_tiny_systems_forced(gpu, oracle, n):
    """fewer bodies than one tile: the single ragged block, zero-mass stand-ins beyond N"""
    oracle.srand(n)
    pos0, vel0 = oracle.randomise(1, n, 1.54, 8.0, np.float32)
    ref_pos, ref_vel = pos0.copy(), vel0.copy()
    oracle.update(ref_pos, ref_vel, DT, steps=2)
    with forced_pairwise(gpu):
        system = pairwise_system(gpu, n, np.float32, pos0, vel0)
        for _ in range(2):
            system.update(DT)
        pos, vel = system.get_position().copy(), system.get_velocity().copy()
        system.free()
    assert rel_err(pos, ref_pos).max() < 2e-6
    assert np.all(pos.reshape(n, 4)[:, 3] == 1) and np.all(vel.reshape(n, 4)[:, 3] == 0)


@pytest.mark.parametrize("n", [256, 1024])
def test_pairwise_is_as_close_to_the_fp64_trajectory_as_the_cpu_fp32_path(gpu, oracle, n):
    """test_fast_is_as_close_to_the_fp64_trajectory_as_the_cpu_fp32_path, through the pairwise layout: after 100 steps the
    kernel must be at least as close to the fp64 CPU path started from the same fp32 bodies as the CPU path's own fp32
    arithmetic (the golden trajectory) is."""
    g = load_golden(n, "f32")
    truth_p, truth_v = g["pos_0"].astype(np.float64), g["vel_0"].astype(np.float64)
    oracle.update(truth_p, truth_v, np.float64(DT), steps=100)
    with forced_pairwise(gpu):
        system = pairwise_system(gpu, n, np.float32, g["pos_0"], g["vel_0"])
        for _ in range(100):
            system.update(DT)
        fast = system.get_position().copy()
        system.free()
    e_fast = rel_err(fast.astype(np.float64), truth_p)
    e_cpu = rel_err(g["pos_100"].astype(np.float64), truth_p)
    print(f"N={n}, 100 steps, rel. error against the fp64 trajectory  pairwise FAST: max {e_fast.max():.2e} p99 {np.percentile(e_fast, 99):.2e} "
          f"median {np.median(e_fast):.2e}   CPU fp32 path: max {e_cpu.max():.2e} p99 {np.percentile(e_cpu, 99):.2e} median {np.median(e_cpu):.2e}")
    assert np.median(e_fast) <= 1.5 * np.median(e_cpu)
    assert np.percentile(e_fast, 99) <= 3 * np.percentile(e_cpu, 99)
    assert e_fast.max() <= 5 * e_cpu.max()


def test_pairwise_conserves_what_the_cpu_path_conserves(gpu):
    """100 steps at 1 024 bodies (BASELINE configs[0]'s system) through the pairwise layout: momentum and energy drift no
    worse than the CPU path's own.  (Momentum: the pairwise kernel applies each term to both bodies, so the sum of m a
    vanishes to summation accuracy by construction.)"""
    n, steps = 1024, 100
    g = load_golden(n, "f32")
    pos0, vel0 = g["pos_0"], g["vel_0"]
    with forced_pairwise(gpu):
        system = pairwise_system(gpu, n, np.float32, pos0, vel0)
        for _ in range(steps):
            system.update(DT)
        fast_pos, fast_vel = system.get_position().copy(), system.get_velocity().copy()
        system.free()
    cpu_pos, cpu_vel = g["pos_100"], g["vel_100"]

    def momentum(pos, vel):
        return (pos.reshape(n, 4)[:, 3:4].astype(np.float64) * xyz(vel).astype(np.float64)).sum(axis=0)

    def energy(pos, vel):
        p, m = xyz(pos).astype(np.float64), pos.reshape(n, 4)[:, 3].astype(np.float64)
        kin = 0.5 * (m * (xyz(vel).astype(np.float64) ** 2).sum(axis=1)).sum()
        d = p[:, None, :] - p[None, :, :]
        r = np.sqrt((d * d).sum(axis=2) + 0.1 ** 2)
        return kin - 0.5 * ((m[:, None] * m[None, :]) / r).sum()

    p0, e0 = momentum(pos0, vel0), energy(pos0, vel0)
    scale_p = (np.abs(xyz(vel0)).astype(np.float64) * pos0.reshape(n, 4)[:, 3:4]).sum()
    drift_fast = np.abs(momentum(fast_pos, fast_vel) - p0).max() / scale_p
    drift_cpu = np.abs(momentum(cpu_pos, cpu_vel) - p0).max() / scale_p
    assert drift_fast <= max(2 * drift_cpu, 1e-6), (drift_fast, drift_cpu)
    de_fast = abs(energy(fast_pos, fast_vel) - e0) / abs(e0)
    de_cpu = abs(energy(cpu_pos, cpu_vel) - e0) / abs(e0)
    assert de_fast <= max(1.5 * de_cpu, 1e-4), (de_fast, de_cpu)


# ------------------------------------------------------------------------------- (b) where the layout applies by default
def test_pairwise_default_plan_fp32_vs_golden_16384(gpu, oracle):
    """16 384 bodies: nb_integrate_ws_f32 takes the pairwise layout with its own plan (no override).  1 step <= 2e-6 and
    10 steps <= 2e-5 per body against the CPU path's trajectory -- the one-sided kernel's envelope -- and the same for the
    velocities after 10 steps (relative to the body's own speed)."""
    n = 16384
    g = load_golden_compact(n, "f32", oracle)
    plan = gpu.pair_plan(n, np.float32)
    assert plan.applies == 1 and plan.reaction_slots > 0
    errs, vel = trajectory_errors(gpu, n, np.float32, g, [1, 10])
    assert_fp32_envelope(errs)
    assert rel_err(vel, g["vel_10"]).max() <= 2e-4  # (|v| ~ 0.1 |p| / dt-scale here: the same absolute error, ten times the relative one)
    # and the one-sided kernel on the same system, for the record: the two FAST layouts sit in the same envelope
    system = gpu.BodySystemHIP(n, 256, gpu.NBodyParams(), np.float32, g["pos_0"], g["vel_0"], mode=gpu.NB_MODE_FAST)
    for _ in range(10):
        system.update(DT)
    one_sided = rel_err(system.get_position(), g["pos_10"])
    system.free()
    print(f"16 384 bodies, 10 steps, max rel. error vs the CPU path: pairwise {errs[10].max():.2e}, one-sided {one_sided.max():.2e}")
    assert one_sided.max() <= 2e-5


def test_pairwise_default_plan_fp64_vs_golden_16384(gpu, oracle):
    n = 16384
    g = load_golden_compact(n, "f64", oracle)
    assert gpu.pair_plan(n, np.float64).applies == 1
    errs, _ = trajectory_errors(gpu, n, np.float64, g, [1, 10])
    assert_fp64_envelope(errs)


def test_pairwise_graph_replay_vs_golden_16384(gpu, oracle):
    """the hipGraph form of the pairwise step loop (nb_graph_create_ws_f32), 10 captured steps, against the same fixture"""
    n = 16384
    g = load_golden_compact(n, "f32", oracle)
    system = pairwise_system(gpu, n, np.float32, g["pos_0"], g["vel_0"])
    system.update_many(DT, 10)
    err = rel_err(system.get_position(), g["pos_10"])
    system.free()
    assert err.max() <= 2e-5, err.max()


# ------------------------------------------------------------------------------- the bounded-workspace form (K slices)
@pytest.mark.parametrize("slices", [2, 3, 5])
@pytest.mark.parametrize("n", [1024, 4096])
def test_sliced_pairwise_fp32_vs_golden(gpu, n, slices):
    """The tournament cut into K slices (what a system too large for one tournament's workspace steps through) against the CPU
    path's trajectory at the golden sizes: the same envelope again."""
    g = load_golden(n, "f32")
    gpu.set_pair_slices_override(slices)
    try:
        with forced_pairwise(gpu, (1, 8, 1)):  # blocks of 128 bodies: 8 / 32 blocks to cut
            assert 2 <= gpu.pair_plan(n, np.float32).slices <= slices  # (rounding to whole blocks can leave fewer slices than asked for)
            errs, _ = trajectory_errors(gpu, n, np.float32, g, golden_steps(g))
    finally:
        gpu.set_pair_slices_override(0)
    assert_fp32_envelope(errs)


def test_sliced_pairwise_fp64_vs_golden(gpu):
    g = load_golden(4096, "f64")
    gpu.set_pair_slices_override(4)
    try:
        with forced_pairwise(gpu, (2, 8, 1)):
            errs, _ = trajectory_errors(gpu, 4096, np.float64, g, golden_steps(g))
    finally:
        gpu.set_pair_slices_override(0)
    assert_fp64_envelope(errs)
