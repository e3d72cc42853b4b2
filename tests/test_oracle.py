"""CPU tests of the checker itself: the oracle against the reference's own randomise_bodies (oracle/_ref),
against the committed golden vectors, against an independent numpy restatement, and scalar vs AVX vs OpenMP."""
import numpy as np
import pytest

from conftest import golden_steps, load_golden, load_golden_compact

DT = np.float32(0.016)


def test_glibc_rand_stream(oracle):
    # glibc TYPE_3 additive generator, seed 1 (the reference never calls srand): first draws are fixed
    oracle.srand(1)
    draws = [oracle.lib.oracle_rand() for _ in range(3)]
    assert draws == [1804289383, 846930886, 1681692777]
    assert oracle.lib.oracle_rand_max() == 2147483647


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("config", [0, 1, 2])
def test_randomise_matches_reference_build(O, oracle, dtype, config):
    """oracle_randomise_* == the reference's randomise_bodies<T> compiled unmodified (randomise_bodies.cpp:47-189)."""
    if not O.ReferenceRandomise.available():
        pytest.skip("oracle/_ref not built (needs /root/reference; the build container builds it)")
    ref = O.ReferenceRandomise()
    for n in (1, 8, 255, 1024, 5000):
        for cluster, velocity in ((1.54, 8.0), (1.52, 2.0), (0.16, 1000.0), (6.04, 0.0)):
            for seed in (1, 12345):
                oracle.srand(seed)
                a = oracle.randomise(config, n, cluster, velocity, dtype)
                ref.srand(seed)
                b = ref.randomise(config, n, cluster, velocity, dtype)
                assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes(), (n, cluster, velocity, seed)


@pytest.mark.parametrize("tag,dtype", [("f32", np.float32), ("f64", np.float64)])
@pytest.mark.parametrize("n", [8, 256, 1024, 4096])
def test_oracle_reproduces_golden(oracle, n, tag, dtype):
    g = load_golden(n, tag)
    pos, vel = oracle.startup_state(n, dtype)
    assert pos.tobytes() == g["pos_0"].tobytes() and vel.tobytes() == g["vel_0"].tobytes()
    done = 0
    for s in golden_steps(g):
        oracle.update(pos, vel, DT, steps=s - done)
        done = s
        assert pos.tobytes() == g[f"pos_{s}"].tobytes(), f"positions differ at step {s}"
        assert vel.tobytes() == g[f"vel_{s}"].tobytes(), f"velocities differ at step {s}"


@pytest.mark.parametrize("tag,dtype", [("f32", np.float32), ("f64", np.float64)])
def test_oracle_reproduces_compact_golden_16384(oracle, tag, dtype):
    """The fixture of a system large enough for the pairwise layout to apply by default (x, y, z of later states only;
    start-up state by digest)."""
    n = 16384
    g = load_golden_compact(n, tag, oracle)
    pos, vel = g["pos_0"].copy(), g["vel_0"].copy()
    oracle.update(pos, vel, DT, steps=1)
    assert pos.tobytes() == g["pos_1"].tobytes()
    oracle.update(pos, vel, DT, steps=9)
    assert pos.tobytes() == g["pos_10"].tobytes()
    if "vel_10" in g:
        assert vel.tobytes() == g["vel_10"].tobytes()


def test_shell_geometry(oracle):
    # SHELL: |p| between inner=2.5*scale and outer=4*scale per axis factor, mass 1, v = (p x z)*vscale  (randomise_bodies.cpp:101-147)
    n = 1024
    pos, vel = oracle.startup_state(n, np.float32)
    p, v = pos.reshape(n, 4), vel.reshape(n, 4)
    assert np.all(p[:, 3] == 1.0) and np.all(v[:, 3] == 0.0)
    r = np.linalg.norm(p[:, :3], axis=1)
    assert r.min() >= 2.5 * 1.52 * 0.999 and r.max() <= 4 * 1.52 * 1.001
    vs = np.float32(1.52) * np.float32(2.0)
    np.testing.assert_allclose(v[:, 0], p[:, 1] * vs, rtol=1e-6)
    np.testing.assert_allclose(v[:, 1], -p[:, 0] * vs, rtol=1e-6)
    np.testing.assert_array_equal(v[:, 2], 0)


def numpy_update_f32(pos, vel, eps2, damping, dt):
    """Independent restatement of bodysystemcpu.cpp:149-243 in numpy float32 (vectorised over i, j sequential)."""
    n = pos.size // 4
    p = pos.reshape(n, 4)
    v = vel.reshape(n, 4)
    f = np.float32
    dv = np.zeros((n, 3), dtype=f)
    for j in range(n):
        d = p[j, :3][None, :] - p[:, :3]
        d2 = d * d
        r2 = ((f(eps2) + d2[:, 0]) + d2[:, 1]) + d2[:, 2]
        r = np.sqrt(r2)
        m_r3 = (p[j, 3] / (r2 * r2)) * r
        dv += m_r3[:, None] * d
    dv = dv * f(dt)
    v[:, :3] = (v[:, :3] + dv) * f(damping)
    p[:, :3] = p[:, :3] + v[:, :3] * f(dt)


def numpy_update_f64(pos, vel, eps2, damping, dt):
    """Independent restatement of bodysystemcpu.cpp:245-299 (vectorised over i, j sequential => same per-i order)."""
    n = pos.size // 4
    p = pos.reshape(n, 4)
    v = vel.reshape(n, 4)
    acc = np.zeros((n, 3))
    for j in range(n):
        d = p[j, :3][None, :] - p[:, :3]
        d2 = d * d
        r2 = (d2[:, 0] + d2[:, 1]) + (d2[:, 2] + eps2)
        r = np.sqrt(r2)
        s = (p[j, 3] / (r2 * r2)) * r
        acc += d * s[:, None]
    dv = acc * dt
    v[:, :3] = (v[:, :3] + dv) * damping
    p[:, :3] = p[:, :3] + v[:, :3] * dt


@pytest.mark.parametrize("dtype,fn", [(np.float32, numpy_update_f32), (np.float64, numpy_update_f64)])
def test_oracle_matches_numpy_restatement(oracle, dtype, fn):
    n = 200  # ragged on purpose (not a multiple of 8)
    oracle.srand(7)
    pos, vel = oracle.randomise(0, n, 1.54, 8.0, dtype)
    pos.reshape(n, 4)[:, 3] = np.linspace(0.5, 2.0, n).astype(dtype)  # variable masses
    a_pos, a_vel, b_pos, b_vel = pos.copy(), vel.copy(), pos.copy(), vel.copy()
    eps2 = oracle.softening_sq(0.1, dtype)
    for _ in range(3):
        oracle.update(a_pos, a_vel, DT, steps=1, damping=0.995)
        fn(b_pos, b_vel, eps2, dtype(np.float32(0.995)), dtype(DT))
    assert a_pos.tobytes() == b_pos.tobytes() and a_vel.tobytes() == b_vel.tobytes()


def test_scalar_avx_openmp_agree(O, oracle):
    omp = O.Oracle(openmp=True)
    for n in (8, 264, 1024):
        pos, vel = oracle.startup_state(n, np.float32)
        outs = []
        for orc, avx in ((oracle, False), (oracle, True), (omp, True)):
            p, v = pos.copy(), vel.copy()
            orc.update(p, v, DT, steps=5, avx=avx)
            outs.append(p.tobytes() + v.tobytes())
        assert outs[0] == outs[1] == outs[2]
    posd, veld = oracle.startup_state(512, np.float64)
    a, b = (posd.copy(), veld.copy()), (posd.copy(), veld.copy())
    oracle.update(*a, DT, steps=5)
    omp.update(*b, DT, steps=5)
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


def test_edge_cases(oracle):
    # N=1: only the self-interaction, which contributes exactly zero for eps > 0
    pos = np.array([1, 2, 3, 5], dtype=np.float32)
    vel = np.array([0.5, 0, -1, 0], dtype=np.float32)
    oracle.update(pos, vel, DT, steps=1)
    np.testing.assert_array_equal(vel, [0.5, 0, -1, 0])
    np.testing.assert_array_equal(pos, np.float32([1, 2, 3, 5]) + np.float32([0.5, 0, -1, 0]) * DT * np.float32([1, 1, 1, 0]))
    # zero-mass bodies exert no force but are moved (tipsy padding, tipsy.cpp:111-119)
    p, v = oracle.startup_state(16, np.float32)
    p2, v2 = np.concatenate([p, np.zeros(32, np.float32)]), np.concatenate([v, np.zeros(32, np.float32)])
    oracle.update(p, v, DT, steps=3)
    oracle.update(p2, v2, DT, steps=3)
    assert p.tobytes() == p2[:64].tobytes()
    # .w of velocity and position are never written
    assert np.all(p.reshape(-1, 4)[:, 3] == 1) and np.all(v.reshape(-1, 4)[:, 3] == 0)
    # AVX form keeps the reference's N % 8 restriction
    with pytest.raises(RuntimeError):
        oracle.update(np.zeros(4 * 12, np.float32), np.zeros(4 * 12, np.float32), DT, avx=True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_update_subset_rows_equal_full_update(O, oracle, dtype):
    """oracle.update_subset (used by the full-size GPU tests) == the matching rows of a full update, bitwise."""
    n = 1000
    pos, vel = oracle.startup_state(n, dtype)
    full_p, full_v = pos.copy(), vel.copy()
    oracle.update(full_p, full_v, DT, steps=1, damping=0.995)
    for orc in (oracle, O.Oracle(openmp=True)):
        for i0, ni in ((0, 8), (123, 77), (992, 8)):
            sp, sv = orc.update_subset(pos, vel, i0, ni, DT, damping=0.995)
            assert sp.tobytes() == full_p[4 * i0:4 * (i0 + ni)].tobytes()
            assert sv.tobytes() == full_v[4 * i0:4 * (i0 + ni)].tobytes()
