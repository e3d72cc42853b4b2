"""The C++23 host mirror (cuda-nbody_amd/host): randomise_bodies, parameter tables, tipsy I/O and the `nbody`
command line.  CPU tests cover host logic; `gpu`-marked tests drive the CLI end to end on the MI355X and compare
its dumps with the golden vectors (so the whole reference-shaped stack -- CLI -> Compute -> ComputeHIP ->
BodySystemHIPDefault -> integrateNbodySystem -> C-ABI -> HIP kernel -- is what is checked)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

PKG = os.path.join(ROOT, "cuda-nbody_amd")
CLI = os.environ.get("NBODY_CLI", os.path.join(PKG, "nbody"))  # `make test-sanitize` points these at the ASan/UBSan builds


@pytest.fixture(scope="module")
def host():
    path = os.environ.get("NBODY_HOST_LIB", os.path.join(PKG, "libnbody_host.so"))
    if not os.path.exists(path):
        subprocess.run(["make", "-s", "-C", os.path.join(PKG, "csrc")], check=True)
        subprocess.run(["make", "-s", "-C", os.path.join(PKG, "host")], check=True)
    lib = ctypes.CDLL(path)
    f32p, f64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
    lib.nbh_srand.argtypes = [ctypes.c_uint]
    lib.nbh_randomise_f32.argtypes = [ctypes.c_int, f32p, f32p, ctypes.c_size_t, ctypes.c_float, ctypes.c_float]
    lib.nbh_randomise_f64.argtypes = [ctypes.c_int, f64p, f64p, ctypes.c_size_t, ctypes.c_float, ctypes.c_float]
    lib.nbh_scale_params_for.argtypes = [ctypes.c_size_t, f32p, f32p]
    lib.nbh_demo_params.argtypes = [ctypes.c_size_t, f32p]
    lib.nbh_read_tipsy.argtypes = [ctypes.c_char_p, f64p, f64p, ctypes.c_size_t]
    lib.nbh_read_tipsy.restype = ctypes.c_long
    lib.nbh_write_tipsy.argtypes = [ctypes.c_char_p, f64p, f64p, ctypes.c_size_t, ctypes.c_int]
    lib.nbh_precision_switch_roundtrip.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p, f64p, f32p]
    lib.nbh_compare_results.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lib.nbh_run_demo.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, f32p]
    return lib


def host_randomise(host, config, n, cluster, velocity, dtype):
    pos, vel = np.zeros(4 * n, dtype), np.zeros(4 * n, dtype)
    if dtype == np.float32:
        host.nbh_randomise_f32(config, pos.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), vel.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n, cluster, velocity)
    else:
        host.nbh_randomise_f64(config, pos.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), vel.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n, cluster, velocity)
    return pos, vel


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("config", [0, 1, 2])
def test_host_randomise_bodies_bitwise(O, oracle, host, dtype, config):
    """The product's randomise_bodies == the oracle's == (when oracle/_ref is built) the reference's own code."""
    ref = O.ReferenceRandomise() if O.ReferenceRandomise.available() else None
    for n in (1, 8, 255, 1024, 5000):
        for cluster, velocity in ((1.54, 8.0), (1.52, 2.0), (0.16, 1000.0)):
            host.nbh_srand(1)
            a = host_randomise(host, config, n, cluster, velocity, dtype)
            oracle.srand(1)
            b = oracle.randomise(config, n, cluster, velocity, dtype)
            assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
            if ref is not None:
                ref.srand(1)
                c = ref.randomise(config, n, cluster, velocity, dtype)
                assert a[0].tobytes() == c[0].tobytes() and a[1].tobytes() == c[1].tobytes()


def test_parameter_tables(O, host):
    # N-dependent scales, compute.cpp:74-92
    for n in (1, 1024, 1025, 2048, 4096, 4097, 8192, 16384, 16385, 32768, 32769, 262144):
        c, v = ctypes.c_float(), ctypes.c_float()
        host.nbh_scale_params_for(n, ctypes.byref(c), ctypes.byref(v))
        want = O.scales_for(n)
        assert (np.float32(c.value), np.float32(v.value)) == want, n
    # demo table, compute.hpp:90-97
    want = [(0.016, 1.54, 8.0, 0.1, 1.0), (0.016, 0.68, 20.0, 0.1, 1.0), (0.0006, 0.16, 1000.0, 1.0, 1.0), (0.0006, 0.16, 1000.0, 1.0, 1.0),
            (0.0019, 0.32, 276.0, 1.0, 1.0), (0.0016, 0.32, 272.0, 0.145, 1.0), (0.016, 6.04, 0.0, 1.0, 1.0)]
    for i, row in enumerate(want):
        out = (ctypes.c_float * 5)()
        assert host.nbh_demo_params(i, out) == 0
        assert tuple(np.float32(x) for x in out) == tuple(np.float32(x) for x in row)
    assert host.nbh_demo_params(7, (ctypes.c_float * 5)()) == -1


def test_tipsy_round_trip_and_padding(host, tmp_path):
    n, ndark = 300, 120
    rng = np.random.default_rng(5)
    pos = rng.standard_normal(4 * n).astype(np.float32).astype(np.float64)
    vel = rng.standard_normal(4 * n).astype(np.float32).astype(np.float64)
    pos[3::4] = (np.abs(pos[3::4]) + 0.1).astype(np.float32)  # masses (float on disk)
    path = str(tmp_path / "model.tipsy").encode()
    dp = ctypes.POINTER(ctypes.c_double)
    assert host.nbh_write_tipsy(path, pos.ctypes.data_as(dp), vel.ctypes.data_as(dp), n, ndark) == 0
    # on-disk layout: 32-byte header, 36-byte dark records, 44-byte star records (tipsy.cpp:25-50)
    assert os.path.getsize(path) == 32 + 36 * ndark + 44 * (n - ndark)
    raw = open(path, "rb").read()
    assert np.frombuffer(raw[8:28], dtype=np.int32).tolist() == [n, 3, 0, ndark, n - ndark]
    first = np.frombuffer(raw[32:32 + 36], dtype=np.float32)
    assert first[0] == np.float32(pos[3]) and first[1] == np.float32(pos[0]) and first[4] == np.float32(vel[0]) and first[7] == np.float32(vel[3])
    rpos, rvel = np.zeros(4 * 512), np.zeros(4 * 512)
    got = host.nbh_read_tipsy(path, rpos.ctypes.data_as(dp), rvel.ctypes.data_as(dp), 512)
    assert got == 512  # 300 padded to a multiple of 256 (tipsy.cpp:111-119)
    np.testing.assert_array_equal(rpos[:4 * n], pos)
    np.testing.assert_array_equal(rvel[:4 * n], vel)
    assert not rpos[4 * n:].any() and not rvel[4 * n:].any()  # zero-mass padding
    assert host.nbh_read_tipsy(b"/nonexistent/file", rpos.ctypes.data_as(dp), rvel.ctypes.data_as(dp), 512) == -1


def run_cli(*args, timeout=300):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=timeout)


def test_cli_argument_handling_without_gpu():
    # exit codes: help 0, bad CLI 1, invalid_argument 1   (nbody.cpp:332-338,396-408)
    r = run_cli("--help")
    assert r.returncode == 0 and "--numbodies" in r.stdout and "--blockSize" in r.stdout
    for bad in (["--bogus"], ["--numbodies=0"], ["--numbodies"], ["--mode=turbo"], ["--tipsy=/no/such/file"], ["--benchmark=1"], ["stray"], ["--demo=7"], ["--demo"],
                ["--inject-error=abc"]):
        r = run_cli(*bad)
        assert r.returncode == 1, bad
        assert "CRITICAL ERROR" in r.stderr
    r = run_cli("--cpu", "--benchmark")  # no CPU path in the product, stated loudly
    assert r.returncode == 1 and "no CPU BodySystem path" in r.stderr
    r = run_cli()  # no viewer
    assert r.returncode == 1 and "viewer" in r.stderr
    # single-dash spellings of the NVIDIA sample parse too (BASELINE.json writes "-cpu -benchmark")
    r = run_cli("-cpu", "-benchmark", "-numbodies=1024")
    assert r.returncode == 1 and "no CPU BodySystem path" in r.stderr


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_cli_benchmark_output_format():
    r = run_cli("--benchmark", "--numbodies=4096", "-i", "5")
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "number of bodies = 4096" in out
    assert "> Simulation data stored in video memory" in out and "> Single precision floating point simulation" in out
    # the reference's three benchmark lines, compute.cpp:108-111
    assert re.search(r"^4096 bodies, total time for 5 iterations: +[\d.e+-]+ ms$", out, re.M)
    assert re.search(r"^= +[\d.e+-]+ billion interactions per second$", out, re.M)
    assert re.search(r"^= +[\d.e+-]+ single-precision GFLOP/s at 20 flops per interaction$", out, re.M)
    r = run_cli("--benchmark", "--numbodies=1000", "--fp64", "--iterations=2")
    assert r.returncode == 0, r.stderr
    assert 'Warning: "number of bodies" specified 1000 is not a multiple of 256.' in r.stdout
    assert "Rounding up to the nearest multiple: 1024." in r.stdout
    assert re.search(r"double-precision GFLOP/s at 30 flops per interaction$", r.stdout, re.M)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--fp64"], ["--hostmem"]])
def test_cli_compare_passes(extra):
    r = run_cli("--compare", "--numbodies=2048", *extra)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr
    assert "  OK" in r.stdout and "Error:" not in r.stdout
    r = run_cli("--qatest", "--numbodies=1024", *extra)
    assert r.returncode == 0


@pytest.mark.gpu
@pytest.mark.parametrize("fp64", [0, 1])
def test_compare_can_fail(host, fp64):
    """--compare is a check, so it must be able to fail (compute_cuda.cpp:310-323, exit code nbody.cpp:375-379): the same
    state passes untouched, and fails -- with the reference's "Error:" line and exit code 1 -- once the FAST result is
    perturbed by more than the 5e-4 tolerance; a perturbation inside the tolerance still passes."""
    assert host.nbh_compare_results(2048, fp64, 0, 0.0) == 1
    assert host.nbh_compare_results(2048, fp64, 0, 4.0e-4) == 1
    assert host.nbh_compare_results(2048, fp64, 0, 6.0e-4) == 0
    assert host.nbh_compare_results(2048, fp64, 1, -1.0e-3) == 0  # mapped host memory variant
    flags = ["--fp64"] if fp64 else []
    r = run_cli("--compare", "--numbodies=1024", "--inject-error=0.001", *flags)
    assert r.returncode == 1 and "Error: (strict)" in r.stdout and "  OK" not in r.stdout
    r = run_cli("--qatest", "--numbodies=1024", "--inject-error=0.0001", *flags)
    assert r.returncode == 0 and "  OK" in r.stdout
    # --compare --steps=K also reports where the fast trajectory stands against the strict one after K steps
    r = run_cli("--compare", "--numbodies=1024", "--steps=100", *flags)
    assert r.returncode == 0 and "  OK" in r.stdout
    m = re.search(r"> after 100 steps of dt = 0.016: \|fast - strict\| / \|strict\| per body: max ([0-9.e+-]+), 99th percentile ([0-9.e+-]+), median ([0-9.e+-]+)", r.stdout)
    assert m, r.stdout
    worst, p99, median = (float(x) for x in m.groups())
    assert median <= p99 <= worst and (median < 1e-10 if fp64 else 1e-6 < median < 1e-4) and worst < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("demo", range(7))
def test_demo_sets_strict_bitwise(host, oracle, O, demo):
    """Compute::select_demo (compute.cpp:156-187, table compute.hpp:90-97) through the reference-shaped stack, STRICT, 5
    steps: bit-identical to the CPU path run with THAT row's dt / softening / damping from the bodies the row's scales
    draw.  rand() stream: three start-up resets, then the reset select_demo itself does (fourth segment)."""
    n, steps = 1024, 5
    rows = [(0.016, 1.54, 8.0, 0.1, 1.0), (0.016, 0.68, 20.0, 0.1, 1.0), (0.0006, 0.16, 1000.0, 1.0, 1.0), (0.0006, 0.16, 1000.0, 1.0, 1.0),
            (0.0019, 0.32, 276.0, 1.0, 1.0), (0.0016, 0.32, 272.0, 0.145, 1.0), (0.016, 6.04, 0.0, 1.0, 1.0)]
    dt, cluster, velocity, softening, damping = rows[demo]
    out = np.zeros(8 * n, np.float32)
    host.nbh_srand(1)
    assert host.nbh_run_demo(n, demo, 0, steps, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))) == 0
    oracle.startup_state(n, np.float32)  # consumes the three start-up segments
    pos, vel = oracle.randomise(O.NBODY_CONFIG_SHELL, n, cluster, velocity, np.float32)
    oracle.update(pos, vel, np.float32(dt), steps=steps, softening=softening, damping=damping)
    assert out[:4 * n].tobytes() == pos.tobytes()
    assert out[4 * n:].tobytes() == vel.tobytes()


@pytest.mark.gpu
def test_cli_demo_and_config_runs_bitwise(tmp_path, oracle, O):
    """`--demo=k` and `--config=random|expand` through the command line, STRICT: the dumps equal the CPU path's."""
    n = 512
    dump = tmp_path / "demo.bin"
    r = run_cli(f"--numbodies={n}", "--mode=strict", "--demo=5", "--steps=4", f"--dump={dump}")
    assert r.returncode == 0, r.stderr
    oracle.startup_state(n, np.float32)
    pos, vel = oracle.randomise(O.NBODY_CONFIG_SHELL, n, 0.32, 272.0, np.float32)
    oracle.update(pos, vel, np.float32(0.0016), steps=4, softening=0.145, damping=1.0)
    raw = np.fromfile(dump, dtype=np.float32)
    assert raw[:4 * n].tobytes() == pos.tobytes() and raw[4 * n:].tobytes() == vel.tobytes()
    for cfg, code in (("random", O.NBODY_CONFIG_RANDOM), ("expand", O.NBODY_CONFIG_EXPAND)):
        for flags, dtype in (([], np.float32), (["--fp64"], np.float64)):
            dump = tmp_path / f"{cfg}{len(flags)}.bin"
            r = run_cli(f"--numbodies={n}", "--mode=strict", f"--config={cfg}", "--steps=3", f"--dump={dump}", *flags)
            assert r.returncode == 0, r.stderr
            pos, vel = oracle.startup_state(n, dtype, config=code)
            oracle.update(pos, vel, np.float32(0.016), steps=3)
            raw = np.fromfile(dump, dtype=dtype)
            assert raw[:4 * n].tobytes() == pos.tobytes() and raw[4 * n:].tobytes() == vel.tobytes(), (cfg, flags)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,dtype,flags", [("f32", np.float32, []), ("f64", np.float64, ["--fp64"]), ("f32", np.float32, ["--hostmem"])])
@pytest.mark.parametrize("n", [256, 1024, 4096])
def test_cli_strict_run_reproduces_golden(tmp_path, n, tag, dtype, flags):
    """A fresh `nbody --numbodies=N --mode=strict --steps=10` process == the CPU path's 10-step trajectory, bitwise:
    exercises the rand() start-up sequence (three resets), the N-scaled params and the strict kernels through the CLI."""
    dump = tmp_path / "state.bin"
    r = run_cli(f"--numbodies={n}", "--mode=strict", "--steps=10", f"--dump={dump}", *flags)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(dump, dtype=dtype)
    g = load_golden(n, tag)
    assert raw[:4 * n].tobytes() == g["pos_10"].tobytes()
    assert raw[4 * n:].tobytes() == g["vel_10"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [[], ["--fp64"]])
def test_cli_fast_hostmem_equals_device_memory_bitwise(tmp_path, flags):
    """FAST at a size that takes the wave-stream kernel (bodies j through scalar loads): with the positions in mapped host
    memory (--hostmem) the run must produce the same bits as with device arrays -- same kernel, same data, and FAST is
    deterministic from run to run."""
    dumps = []
    for extra in (["--no-workspace"], ["--hostmem"]):  # (device arrays would own a workspace and take the pairwise layout)
        dump = tmp_path / ("state" + "_".join(extra) + ".bin")
        r = run_cli("--numbodies=32768", "--steps=3", f"--dump={dump}", *flags, *extra)
        assert r.returncode == 0, r.stderr
        dumps.append(np.fromfile(dump, dtype=np.uint8))
    assert dumps[0].size == 32768 * 8 * (8 if flags else 4)
    assert dumps[0].tobytes() == dumps[1].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [[], ["--fp64"]])
def test_cli_owns_a_workspace_by_default(tmp_path, flags):
    """BodySystemHIPDefault owns the scratch memory nb_workspace_bytes_* asks for and steps through nb_integrate_ws_*: at
    32 768 bodies that is the pairwise layout.  Same trajectory as --no-workspace (the one-sided kernel) up to summation order,
    not the same bits; --compare passes; --benchmark (plain loop and captured as a hipGraph) prints the reference's lines."""
    dumps = {}
    for name, extra in (("default", []), ("again", []), ("one_sided", ["--no-workspace"])):
        dump = tmp_path / f"state_{name}.bin"
        r = run_cli("--numbodies=32768", "--steps=4", f"--dump={dump}", *flags, *extra)
        assert r.returncode == 0, r.stderr
        dumps[name] = np.fromfile(dump, dtype=np.float64 if flags else np.float32)
    assert dumps["default"].tobytes() == dumps["again"].tobytes()  # reproducible from run to run
    assert dumps["default"].tobytes() != dumps["one_sided"].tobytes()
    tol = 1e-11 if flags else 5e-5
    np.testing.assert_allclose(dumps["default"], dumps["one_sided"], rtol=tol, atol=tol)
    r = run_cli("--compare", "--numbodies=32768", *flags)
    assert r.returncode == 0 and "  OK" in r.stdout, r.stdout + r.stderr
    for extra in ([], ["--graph"]):
        r = run_cli("--benchmark", "--numbodies=32768", "-i", "4", *flags, *extra)
        assert r.returncode == 0 and "billion interactions per second" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cli_fast_on_a_three_species_galaxy_file(host, tmp_path, oracle):
    """A tipsy model with three species of different masses in contiguous blocks (what galaxy files look like), large enough
    for the wave-stream kernel: FAST (chunks of one species take the loop without the mass multiply, chunks straddling a
    boundary the mixed one) against STRICT -- the CPU path's bits -- after 3 steps."""
    n, nd = 40960, 21111
    oracle.srand(21)
    pos, vel = oracle.randomise(0, n, 1.54, 8.0, np.float64)
    mass = np.empty(n, np.float32)
    mass[:nd], mass[nd:nd + 12345], mass[nd + 12345:] = 2.5, 0.75, 1.0  # (dark first, then stars: the writer's order)
    pos[3::4] = mass
    vel[3::4] = np.float32(0.01)
    path = tmp_path / "galaxy.tipsy"
    dp = ctypes.POINTER(ctypes.c_double)
    assert host.nbh_write_tipsy(str(path).encode(), pos.ctypes.data_as(dp), vel.ctypes.data_as(dp), n, nd) == 0
    out = {}
    for mode in ("fast", "strict"):
        dump = tmp_path / f"{mode}.bin"
        r = run_cli(f"--tipsy={path}", f"--mode={mode}", "--steps=3", f"--dump={dump}")
        assert r.returncode == 0, r.stderr
        out[mode] = np.fromfile(dump, dtype=np.float32)
    assert out["fast"].size == out["strict"].size == 8 * n
    p_fast, p_strict = out["fast"][:4 * n].reshape(n, 4), out["strict"][:4 * n].reshape(n, 4)
    assert np.array_equal(p_fast[:, 3], p_strict[:, 3])  # masses untouched
    scale = np.abs(p_strict[:, :3]).max()
    assert np.abs(p_fast[:, :3] - p_strict[:, :3]).max() / scale < 2e-6
    v_fast, v_strict = out["fast"][4 * n:].reshape(n, 4), out["strict"][4 * n:].reshape(n, 4)
    assert np.abs(v_fast[:, :3] - v_strict[:, :3]).max() / np.abs(v_strict[:, :3]).max() < 2e-5


@pytest.mark.gpu
def test_cli_workspace_cap_steps_through_the_sliced_tournament(tmp_path):
    """`nbody --workspace-mib=<n>`: the body system spends at most that much on its workspace, the library cuts the pair tournament
    into slices that share one region of reaction planes: the same trajectory up to summation order as the default (one
    tournament, 403 MB at 262 144 bodies) and as --no-workspace, different bits; --compare passes; the benchmark lines print."""
    n = 262144
    dumps = {}
    for name, extra in (("one", []), ("capped", ["--workspace-mib=200"]), ("tiny", ["--workspace-mib=1"]), ("none", ["--no-workspace"])):
        dump = tmp_path / f"{name}.bin"
        r = run_cli(f"--numbodies={n}", "--steps=2", f"--dump={dump}", *extra)
        assert r.returncode == 0, r.stderr
        dumps[name] = np.fromfile(dump, dtype=np.float32)
    assert dumps["capped"].tobytes() != dumps["one"].tobytes() and dumps["capped"].tobytes() != dumps["none"].tobytes()
    assert dumps["tiny"].tobytes() == dumps["none"].tobytes()  # nothing fits one MiB: the one-sided kernel
    p = lambda a: a[:4 * n].reshape(n, 4)[:, :3]  # noqa: E731
    scale = np.abs(p(dumps["one"])).max()
    assert np.abs(p(dumps["capped"]) - p(dumps["one"])).max() / scale < 2e-5
    assert np.abs(p(dumps["capped"]) - p(dumps["none"])).max() / scale < 2e-5
    r = run_cli("--compare", "--numbodies=65536", "--workspace-mib=64")
    assert r.returncode == 0 and "  OK" in r.stdout, r.stdout + r.stderr
    r = run_cli("--benchmark", f"--numbodies={n}", "--workspace-mib=200", "-i", "4")
    assert r.returncode == 0 and "billion interactions per second" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cli_tipsy_and_other_configs(host, tmp_path, oracle):
    n = 700
    oracle.srand(9)
    pos, vel = oracle.randomise(0, n, 1.54, 8.0, np.float64)
    pos[3::4] = np.linspace(0.5, 1.5, n).astype(np.float32)
    vel[3::4] = np.float32(0.05)
    path = tmp_path / "m.tipsy"
    dp = ctypes.POINTER(ctypes.c_double)
    assert host.nbh_write_tipsy(str(path).encode(), pos.ctypes.data_as(dp), vel.ctypes.data_as(dp), n, 300) == 0
    dump = tmp_path / "t.bin"
    r = run_cli(f"--tipsy={path}", "--mode=strict", "--steps=3", f"--dump={dump}")
    assert r.returncode == 0, r.stderr
    assert "Read 768 bodies" in r.stdout
    raw = np.fromfile(dump, dtype=np.float32)
    p32 = np.concatenate([pos.astype(np.float32), np.zeros(4 * 68, np.float32)])
    v32 = np.concatenate([vel.astype(np.float32), np.zeros(4 * 68, np.float32)])
    v_in = v32.copy()
    oracle.update(p32, v32, np.float32(0.016), steps=3)
    assert raw[:4 * 768].tobytes() == p32.tobytes()
    # velocity.w carries the tipsy eps and is never touched by the integrator
    v32.reshape(-1, 4)[:, 3] = v_in.reshape(-1, 4)[:, 3]
    assert raw[4 * 768:].tobytes() == v32.tobytes()
    # --numbodies that contradicts the file is an invalid_argument
    assert run_cli(f"--tipsy={path}", "--numbodies=1024", "--benchmark").returncode == 1
    # RANDOM / EXPAND start-up configurations run
    for cfg in ("random", "expand"):
        assert run_cli("--numbodies=512", f"--config={cfg}", "--steps=2", f"--dump={tmp_path / cfg}").returncode == 0


@pytest.mark.gpu
@pytest.mark.parametrize("hostmem", [0, 1])
def test_precision_switch_through_compute(host, oracle, hostmem):
    """Compute::switch_precision (compute.cpp:123-131, compute_cuda.cpp:152-181): the state crosses fp32 -> fp64 -> fp32
    through the host exactly (widening is exact; narrowing rounds once), for both storage variants."""
    n, s32, s64 = 1024, 3, 2
    a = np.zeros(8 * n, np.float32)
    d = np.zeros(8 * n, np.float64)
    b = np.zeros(8 * n, np.float32)
    f32p, f64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
    host.nbh_srand(1)
    rc = host.nbh_precision_switch_roundtrip(n, hostmem, s32, s64, a.ctypes.data_as(f32p), d.ctypes.data_as(f64p), b.ctypes.data_as(f32p))
    assert rc == 0
    # fp32 leg == the FAST kernel from the reference's start-up state (tolerance as in test_gpu_parity)
    ref_p, ref_v = oracle.startup_state(n, np.float32)
    oracle.update(ref_p, ref_v, np.float32(0.016), steps=s32)
    np.testing.assert_allclose(a[:4 * n], ref_p, rtol=1e-5, atol=1e-5)
    # fp64 leg started from exactly the widened fp32 state: re-run it on the oracle
    p64, v64 = a[:4 * n].astype(np.float64), a[4 * n:].astype(np.float64)
    oracle.update(p64, v64, np.float32(0.016), steps=s64)
    np.testing.assert_allclose(d[:4 * n], p64, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(d[4 * n:], v64, rtol=1e-11, atol=1e-11)
    # switching back narrows the fp64 state once
    assert b[:4 * n].tobytes() == d[:4 * n].astype(np.float32).tobytes()
    assert b[4 * n:].tobytes() == d[4 * n:].astype(np.float32).tobytes()
