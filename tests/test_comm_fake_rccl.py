"""The product's multi-GPU C-ABI (csrc/nbody_comm.hip) with MORE THAN ONE rank, on the one GPU of the test box.

RCCL refuses two ranks on one device, so until round 3 every G > 1 line of nbody_comm.hip -- the grouped send/recv rounds of
exchange_tiles, the per-round events, the `arrived[peer]` waits and STRICT's rank order in sharded_step, the in-flight
bookkeeping across steps -- had never executed anywhere.  These tests run exactly that code: the library resolves RCCL
with dlopen and honours NBODY_RCCL_LIB, which here points at a TEST DOUBLE for RCCL (tests/fake_rccl/fake_rccl.cpp: the
ten nccl* entry points as device-to-device copies with RCCL's stream semantics; ranks may share a device).  The double
stands in for a third-party transport, not for anything of the reference (which is single-GPU).

The library binds RCCL once per process, so every case runs in a worker process (tests/fake_rccl/worker.py) that gets
the environment variable; the parent compares what the worker's ranks hold with the CPU oracle / golden fixtures.
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")
FAKE_LIB = os.path.join(FAKE_DIR, "libfake_rccl.so")
NCCL_SYMBOLS = ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclAllGather", "ncclGroupStart",
                "ncclGroupEnd", "ncclGetErrorString")


def _env(**extra):
    if not os.path.exists(FAKE_LIB):  # (normally built by __graft_entry__.build() and shipped with the tree)
        subprocess.run(["make", "-s", "-C", FAKE_DIR], check=True)
    env = dict(os.environ)
    env["NBODY_RCCL_LIB"] = FAKE_LIB
    env["FAKE_RCCL_TIMEOUT_S"] = "120"
    env.update(extra)
    return env


def _run(tmp_path, case, pos0, vel0, world, steps, mode, streams="streams", workspace=False, real_rccl=False, **env):
    src, dst = tmp_path / f"in_{case}.npz", tmp_path / f"out_{case}.npz"
    np.savez(src, pos=pos0, vel=vel0)
    environment = _env(**env)
    if real_rccl:  # the `all` case as an in-process world over the REAL library (nb_comm_inprocess_open_all)
        environment.pop("NBODY_RCCL_LIB")
        environment.pop("NCCL_DEBUG", None)
        environment["WORKER_REAL_RCCL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(FAKE_DIR, "worker.py"), case, str(src), str(dst), str(world), str(steps), mode, streams, "ws" if workspace else "-"],
                       env=environment, capture_output=True, text=True, timeout=300 if real_rccl else 900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    with np.load(dst, allow_pickle=False) as data:  # read everything now: the next run of the same case rewrites the file
        return {k: data[k] for k in data.files}


def test_fake_rccl_builds_and_exports_what_the_library_resolves():
    """CPU check: the test double compiles and exports the ten entry points nbody_comm.hip asks dlsym for."""
    subprocess.run(["make", "-s", "-C", FAKE_DIR], check=True)
    out = subprocess.run(["nm", "-D", "--defined-only", FAKE_LIB], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert set(NCCL_SYMBOLS) <= exported
    with open(os.path.join(ROOT, "cuda-nbody_amd", "csrc", "nbody_comm.hip")) as fh:
        source = fh.read()
    for name in NCCL_SYMBOLS:  # the list above is the list the product resolves
        assert f'sym("{name}")' in source


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tag", [(np.float32, "f32"), (np.float64, "f64")])
@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_init_all_four_ranks_on_one_gpu(tmp_path, oracle, dtype, tag, mode):
    """nb_comm_init_all(4, {0,0,0,0}) + nb_sharded_step_all_*, 4 096 bodies, 5 steps, one compute stream per rank: every
    rank ends with the same positions; STRICT == the CPU path bit for bit (and == nb_integrate_*), FAST within the
    tolerance of the single-GPU tests.  3 rounds x 4 ranks x 5 steps of send/recv pairs went through the transport."""
    n, steps, world = 4096, 5, 4
    pos0, vel0 = oracle.startup_state(n, dtype)
    got = _run(tmp_path, "all", pos0, vel0, world, steps, mode)
    assert list(got["rejected"]) == [10001, 10001, 10001]  # subset of the group / a rank twice / per-rank form on a multi-rank group
    sends, recvs, gathers, groups, copies = got["counters"]
    # a group per round (the default since round 5) -- and, since round 6, per RANK: the ranks own a communicator each, so a crew of
    # threads steps them independently, every thread issuing its own rank's groups (RCCL's thread-per-device model)
    assert sends == recvs == copies == (world - 1) * world * steps and groups == (world - 1) * world * steps and gathers == 0
    # NBODY_STEP_THREADS=0: the calling thread enqueues every rank, a round is ONE group over the local ranks -- the same bits
    alone = _run(tmp_path, "all", pos0, vel0, world, steps, mode, NBODY_STEP_THREADS="0")
    assert alone["counters"][3] == (world - 1) * steps and alone["counters"][0] == sends
    for k in range(world):
        assert alone[f"pos_{k}"].tobytes() == got[f"pos_{k}"].tobytes() and alone[f"vel_{k}"].tobytes() == got[f"vel_{k}"].tobytes()
    ref_p, ref_v = pos0.copy(), vel0.copy()
    oracle.update(ref_p, ref_v, dtype(np.float32(0.016)), steps=steps)
    pos = [got[f"pos_{k}"] for k in range(world)]
    vel = np.concatenate([got[f"vel_{k}"] for k in range(world)])
    for k in range(1, world):
        assert pos[k].tobytes() == pos[0].tobytes(), f"rank {k} holds other positions than rank 0"
    if mode == "strict":
        assert pos[0].tobytes() == ref_p.tobytes() and vel.tobytes() == ref_v.tobytes()
        assert pos[0].tobytes() == got["single_pos"].tobytes() and vel.tobytes() == got["single_vel"].tobytes()
    else:
        tol = 1e-5 if dtype == np.float32 else 1e-12
        np.testing.assert_allclose(pos[0], ref_p, rtol=tol, atol=tol)
        np.testing.assert_allclose(vel, ref_v, rtol=10 * tol, atol=10 * tol)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_init_all_default_stream_and_odd_world(tmp_path, oracle, world):
    """Every rank on the device's DEFAULT stream (what BodySystemHIPSharded passes), worlds of 2 and 3 (a tile count that is
    not a power of two), STRICT bitwise against the CPU path."""
    n, steps = 3072, 4
    pos0, vel0 = oracle.startup_state(n, np.float32)
    got = _run(tmp_path, "all", pos0, vel0, world, steps, "strict", streams="default")
    ref_p, ref_v = pos0.copy(), vel0.copy()
    oracle.update(ref_p, ref_v, np.float32(0.016), steps=steps)
    for k in range(world):
        assert got[f"pos_{k}"].tobytes() == ref_p.tobytes()
    assert np.concatenate([got[f"vel_{k}"] for k in range(world)]).tobytes() == ref_v.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["strict", "fast"])
def test_init_rank_one_thread_per_rank(tmp_path, oracle, mode):
    """The one-process-per-GPU model in miniature: nb_comm_unique_id on one thread, nb_comm_init_rank + nb_sharded_step_f32
    on a thread per rank (groups of ONE local rank: the rounds of different ranks meet inside the transport)."""
    n, steps, world = 4096, 5, 4
    pos0, vel0 = oracle.startup_state(n, np.float32)
    got = _run(tmp_path, "threads", pos0, vel0, world, steps, mode)
    sends, recvs, gathers, groups, copies = got["counters"]
    assert sends == recvs == copies == (world - 1) * world * steps and groups == (world - 1) * world * steps  # (a thread per rank: every rank issues its own group per round)
    ref_p, ref_v = pos0.copy(), vel0.copy()
    oracle.update(ref_p, ref_v, np.float32(0.016), steps=steps)
    vel = np.concatenate([got[f"vel_{k}"] for k in range(world)])
    for k in range(world):
        if mode == "strict":
            assert got[f"pos_{k}"].tobytes() == ref_p.tobytes()
        else:
            assert got[f"pos_{k}"].tobytes() == got["pos_0"].tobytes()
            np.testing.assert_allclose(got[f"pos_{k}"], ref_p, rtol=1e-5, atol=1e-5)
    if mode == "strict":
        assert vel.tobytes() == ref_v.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_exchange_entry_points_on_their_own(tmp_path, oracle, dtype):
    """nb_exchange_tiles_* / nb_allgather_* / nb_exchange_wait_tile with 4 ranks: each rank starts with only its own slice
    (the rest poisoned) and ends with the whole array."""
    n, world = 2048, 4
    pos0, vel0 = oracle.startup_state(n, dtype)
    got = _run(tmp_path, "exchange", pos0, vel0, world, 0, "strict")
    for k in range(world):
        assert got[f"tiles_{k}"].tobytes() == pos0.tobytes()
        assert got[f"gather_{k}"].tobytes() == (pos0 + dtype(1)).tobytes()
    sends, recvs, gathers, groups, copies = got["counters"]
    assert sends == recvs == (world - 1) * world and gathers == world


@pytest.mark.gpu
def test_full_size_four_ranks_bitwise_equal_to_one_gpu(tmp_path, oracle):
    """262 144 bodies (BASELINE configs[2]) over 4 ranks, 5 steps: STRICT sharded == nb_integrate_f32 on one rank, all bits of
    all bodies (the single-rank STRICT step is itself held to the CPU path on sampled bodies in test_gpu_parity.py); FAST
    sharded agrees with FAST on one rank to summation-order accuracy."""
    n, steps, world = 262144, 5, 4
    pos0, vel0 = oracle.startup_state(n, np.float32)
    got = _run(tmp_path, "all", pos0, vel0, world, steps, "strict")
    for k in range(world):
        assert got[f"pos_{k}"].tobytes() == got["single_pos"].tobytes()
    assert np.concatenate([got[f"vel_{k}"] for k in range(world)]).tobytes() == got["single_vel"].tobytes()
    # FAST: another summation order than one rank, so not bitwise.  This system collapses violently (|v| 75 -> 1 200 in five
    # steps) and amplifies any difference ~10x per step (profiles/round3_fast_strict_vs_fp64_truth.txt), hence one step held
    # tightly and three steps loosely; a stale or missing tile would be off by ~0.2 per step (bodies move 1.2 per step).
    for fast_steps, tol in ((1, 2e-5), (3, 5e-3)):
        fast = _run(tmp_path, "all", pos0, vel0, world, fast_steps, "fast")
        for k in range(1, world):
            assert fast[f"pos_{k}"].tobytes() == fast["pos_0"].tobytes()
        np.testing.assert_allclose(fast["pos_0"], fast["single_pos"], rtol=0, atol=tol)
        vel = np.concatenate([fast[f"vel_{k}"] for k in range(world)])
        np.testing.assert_allclose(vel, fast["single_vel"], rtol=0, atol=tol / 0.016 * 4)


@pytest.mark.gpu
def test_full_size_fp64_four_ranks_bitwise_equal_to_one_gpu(tmp_path, oracle):
    n, steps, world = 65536, 3, 4
    pos0, vel0 = oracle.startup_state(n, np.float64)
    got = _run(tmp_path, "all", pos0, vel0, world, steps, "strict")
    for k in range(world):
        assert got[f"pos_{k}"].tobytes() == got["single_pos"].tobytes()
    assert np.concatenate([got[f"vel_{k}"] for k in range(world)]).tobytes() == got["single_vel"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0"])
def test_cli_devices_sharing_one_gpu_match_golden(tmp_path, devices):
    """`nbody --devices=0,0 --mode=strict --steps=10`: BodySystemHIPSharded -> nb_comm_init_all -> nb_sharded_step_all_* with 2
    and 4 shards, bit-identical to the golden CPU-path trajectory, fp32 and fp64."""
    cli = os.path.join(ROOT, "cuda-nbody_amd", "nbody")
    n = 1024
    shards = devices.count(",") + 1
    for flags, dtype, tag in (([], np.float32, "f32"), (["--fp64"], np.float64, "f64")):
        dump = tmp_path / f"sharded_{tag}.bin"
        r = subprocess.run([cli, f"--numbodies={n}", "--mode=strict", "--steps=10", f"--devices={devices}", f"--dump={dump}", *flags],
                           env=_env(), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"> {shards} Devices used for simulation" in r.stdout
        raw = np.fromfile(dump, dtype=dtype)
        g = load_golden(n, tag)
        assert raw[:4 * n].tobytes() == g["pos_10"].tobytes() and raw[4 * n:].tobytes() == g["vel_10"].tobytes()
    r = subprocess.run([cli, "--compare", "--numbodies=4096", f"--devices={devices}"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "  OK" in r.stdout, r.stdout + r.stderr


# ------------------------------------------------------------------------------------------------ pairs once, across the ranks
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("world", [2, 3, 4, 5])
def test_pairwise_step_across_ranks(tmp_path, oracle, world, dtype):
    """Every rank lent a workspace: FAST evaluates each pair of bodies once across the ranks -- the diagonal pairwise, the
    rectangles against ranks r+1 .. r+G/2 pairwise with the reaction sums SENT to their owners (one more send/recv round per
    partner), for an even world the rectangle at distance G/2 split between the two partners.  Worlds of 2 .. 5 (odd and even,
    with and without a split rectangle), a slice that is not a multiple of a block: all ranks hold the same positions, within
    the FAST tolerance of the CPU path, and the same bits in a second run; the transport saw the extra rounds."""
    n, steps = world * 1000, 4
    n -= n % 8
    n -= n % world
    pos0, vel0 = oracle.startup_state((n + 7) // 8 * 8, np.float32)
    pos0, vel0 = pos0[:4 * n].astype(dtype), vel0[:4 * n].astype(dtype)
    got = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True)
    assert np.all(got["workspace_bytes"] > 0)
    sends, recvs, gathers, groups, copies = got["counters"]
    # a group per round for the world // 2 tiles the next step's kernels wait for, ONE group for the tiles nobody waits for (round 5),
    # a group per reaction round
    H = world // 2
    # (each by the thread of its own rank: x world)
    assert sends == recvs == (world - 1 + H) * world * steps and groups == (H + (1 if world - 1 > H else 0) + H) * steps * world
    alone = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True, NBODY_STEP_THREADS="0")  # the calling thread alone: a group per round over all ranks, the same bits
    assert alone["counters"][3] == (H + (1 if world - 1 > H else 0) + H) * steps and alone["pos_0"].tobytes() == got["pos_0"].tobytes()
    ref_p, ref_v = pos0.copy(), vel0.copy()
    oracle.update(ref_p, ref_v, dtype(np.float32(0.016)), steps=steps)
    for k in range(1, world):
        assert got[f"pos_{k}"].tobytes() == got["pos_0"].tobytes()
    tol = 1e-5 if dtype == np.float32 else 1e-12
    np.testing.assert_allclose(got["pos_0"], ref_p, rtol=tol, atol=tol)
    vel = np.concatenate([got[f"vel_{k}"] for k in range(world)])
    np.testing.assert_allclose(vel, ref_v, rtol=10 * tol, atol=10 * tol)
    again = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True)
    assert again["pos_0"].tobytes() == got["pos_0"].tobytes()
    one_sided = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=False)
    assert one_sided["pos_0"].tobytes() != got["pos_0"].tobytes()  # (it really is another schedule)
    # STRICT ignores the workspace: still the CPU path's bits
    strict = _run(tmp_path, "all", pos0, vel0, world, steps, "strict", workspace=True)
    assert strict["pos_0"].tobytes() == ref_p.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 5, 8])
def test_reaction_sends_are_enqueued_before_the_last_force_kernel(tmp_path, oracle, world):
    """VERDICT r4 item 2.  The finish kernel of a rank waits for the reaction sums its partners send, and a partner used to produce
    the last of them with its last kernel: that hop sat bare at the end of every step.  Round 5: a rank's diagonal -- work that
    needs nothing from anybody -- goes out as two launches, the second one LAST, and every reaction round is enqueued before it
    (nb_comm_last_step_trace, tuning header: the host order of rank 0's last step).  nb_set_late_diagonal(0) is round 4's order,
    kept for A/B timings: another summation order (other last bits), the same physics."""
    n, steps = world * 1024, 2
    pos0, vel0 = oracle.startup_state(n, np.float32)
    got = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True)
    trace = bytes(got["trace_0"]).decode().split("\n")[:-1]
    H = world // 2
    folds = [f"fold {s}" for s in range(1, H + 1)]
    # the rounds leave in the order their sums are expected to be ready (the exchange stream is a FIFO): with 4 ranks the half rectangle
    # (partner 2, on the step's own stream behind a quarter of the diagonal) is done before partner 1's full one on the second stream;
    # with 8 ranks partner 4's half rectangle before partner 3's
    sends = [f"send reaction {s}" for s in {2: [1], 4: [2, 1], 5: [1, 2], 8: [1, 2, 4, 3]}[world]]
    assert trace[0] == "forces diagonal-early" and trace[-2:] == ["forces diagonal-late", "finish"], trace
    assert [t for t in trace if t.startswith("fold")] == folds and [t for t in trace if t.startswith("send")] == sends
    assert max(trace.index(t) for t in folds) < min(trace.index(t) for t in sends)  # a round is enqueued once its rectangle's fold is
    assert max(trace.index(t) for t in sends) < trace.index("forces diagonal-late")
    assert sum(t.startswith("forces") for t in trace) == 2 + H
    before = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True, WORKER_LATE_DIAGONAL="0")
    old = bytes(before["trace_0"]).decode().split("\n")[:-1]
    assert old[0] == "forces diagonal" and "forces diagonal-late" not in old and old[-1] == "finish" and old[-2].startswith("send reaction")
    assert before["pos_0"].tobytes() != got["pos_0"].tobytes()
    np.testing.assert_allclose(before["pos_0"], got["pos_0"], rtol=2e-6, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [4, 8, 6, 5])
def test_both_compute_streams_can_end_on_local_work(tmp_path, oracle, world):
    """Round 6: nb_set_late_diagonal(2) -- the second compute stream takes half of the late diagonal offsets as ITS last
    kernel and hands the same amount of work, the first tiles of bodies j of its last rectangle, to the step's own stream (own reaction
    planes, own fold into the first part of the send array, the send waiting for both folds).  Measured slower on one GPU
    (profiles/round6_cut_rectangle_ab.txt) and NOT the default; selectable for A/B timings on real links, so it must be right: every rank
    the same positions, as close to the CPU path as the shipping order, the crew and the calling thread alone the same bits, every send
    enqueued before the late kernels; worlds it does not apply to (6 ranks: the split rectangle is the second stream's last; odd worlds:
    the second stream ends early anyway) keep the shipping deal bit for bit."""
    n, steps = world * 4096, 3
    pos0, vel0 = oracle.startup_state(n, np.float32)
    ref_p, ref_v = pos0.copy(), vel0.copy()
    oracle.update(ref_p, ref_v, np.float32(0.016), steps=steps)
    ships = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True)
    cut = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True, WORKER_LATE_DIAGONAL="2")
    assert list(cut["layout"]) == [1] * world
    for k in range(world):
        assert cut[f"pos_{k}"].tobytes() == cut["pos_0"].tobytes()
    trace = bytes(cut["trace_0"]).decode().split("\n")[:-1]
    if world in (6, 5):
        assert cut["pos_0"].tobytes() == ships["pos_0"].tobytes() and not any("cut-off" in t or "second stream" in t for t in trace)
        return
    H, s_cut = world // 2, world // 2 - 1
    assert cut["pos_0"].tobytes() != ships["pos_0"].tobytes() and np.all(cut["workspace_bytes"] > ships["workspace_bytes"])
    assert np.abs(cut["pos_0"] - ref_p).max() <= 1.5 * np.abs(ships["pos_0"] - ref_p).max() + 1e-7 and np.abs(cut["pos_0"] - ref_p).max() < 2e-5
    alone = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True, WORKER_LATE_DIAGONAL="2", NBODY_STEP_THREADS="0")
    assert alone["pos_0"].tobytes() == cut["pos_0"].tobytes()
    assert trace[0] == "forces diagonal-early" and trace[-3:] == ["forces diagonal-late", "forces diagonal-late second stream", "finish"], trace
    assert f"forces rectangle {s_cut} cut-off part" in trace and f"fold {s_cut} cut-off part" in trace
    assert trace.index(f"fold {s_cut} cut-off part") < trace.index(f"send reaction {s_cut}") and trace.index(f"fold {s_cut}") < trace.index(f"send reaction {s_cut}")
    assert max(trace.index(t) for t in trace if t.startswith("send")) < trace.index("forces diagonal-late")
    assert sum(t.startswith("forces") for t in trace) == 2 + H + 2
    sends, recvs = cut["counters"][0], cut["counters"][1]
    assert sends == recvs == (world - 1 + H) * world * steps  # (the same transfers: the cut is inside a rank)


@pytest.mark.gpu
def test_pairwise_step_across_ranks_one_thread_per_rank(tmp_path, oracle):
    n, steps, world = 4096, 4, 4
    pos0, vel0 = oracle.startup_state(n, np.float32)
    got = _run(tmp_path, "threads", pos0, vel0, world, steps, "fast", workspace=True)
    sends, recvs, gathers, groups, copies = got["counters"]
    assert sends == recvs == (world - 1 + world // 2) * world * steps
    # nb_comm_set_workspace was a collective here: every rank told every other what it was lent (one grouped round of notes each)
    assert list(got["setup_counters"][:2]) == [(world - 1) * world] * 2 and got["setup_counters"][3] == world
    assert list(got["layout"]) == [1] * world
    ref_p, ref_v = pos0.copy(), vel0.copy()
    oracle.update(ref_p, ref_v, np.float32(0.016), steps=steps)
    for k in range(world):
        assert got[f"pos_{k}"].tobytes() == got["pos_0"].tobytes()
    np.testing.assert_allclose(got["pos_0"], ref_p, rtol=1e-5, atol=1e-5)
    same = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True)
    assert same["pos_0"].tobytes() == got["pos_0"].tobytes()  # one thread for all ranks or one per rank: the same sums in the same order


@pytest.mark.gpu
def test_pairwise_step_across_ranks_full_size(tmp_path, oracle):
    """262 144 bodies over 4 and 8 ranks (BASELINE configs[2] as the strong-scaling series shards it), one step held tightly and
    three loosely against one rank (this system amplifies any difference ~10x per step, see above)."""
    n = 262144
    pos0, vel0 = oracle.startup_state(n, np.float32)
    for world in (4, 8):
        for steps, tol in ((1, 2e-5), (3, 5e-3)):
            got = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True)
            for k in range(1, world):
                assert got[f"pos_{k}"].tobytes() == got["pos_0"].tobytes()
            np.testing.assert_allclose(got["pos_0"], got["single_pos"], rtol=0, atol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("world,workspace", [(4, False), (4, True), (8, True)])
def test_full_size_calm_system_five_steps_tightly(tmp_path, oracle, world, workspace):
    """the SHELL start-up system collapses and amplifies any difference ~10x per step, so the full-size FAST
    checks above can hold only ONE step tightly.  The EXPAND configuration (bodies moving outward from rest positions,
    randomise_bodies.cpp:149-187) amplifies nothing: 262 144 bodies over 4 and 8 ranks, one-sided tiles and pairwise across the ranks,
    FIVE steps against one rank at 1e-5 of the system's size -- a stale or missing tile at ANY of the five steps would be off by
    orders of magnitude more (a body moves ~1e-2 of the system's size per step)."""
    n, steps = 262144, 5
    oracle.srand(3)
    pos0, vel0 = oracle.randomise(2, n, 1.54, 8.0, np.float32)  # NBODY_CONFIG_EXPAND
    got = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=workspace)
    assert list(got["layout"]) == [1 if workspace else 0] * world
    for k in range(1, world):
        assert got[f"pos_{k}"].tobytes() == got["pos_0"].tobytes()
    size = np.abs(got["single_pos"].reshape(n, 4)[:, :3]).max()
    moved = np.abs(got["single_pos"] - pos0).reshape(n, 4)[:, :3].max()
    assert moved > 1e-2 * size  # (the bodies did move: five steps are not a no-op)
    np.testing.assert_allclose(got["pos_0"], got["single_pos"], rtol=0, atol=1e-5 * size)
    vel = np.concatenate([got[f"vel_{k}"] for k in range(world)])
    np.testing.assert_allclose(vel, got["single_vel"], rtol=0, atol=1e-5 * np.abs(got["single_vel"]).max() + 1e-4 * size)


@pytest.mark.gpu
def test_cli_sharded_fast_owns_workspaces(tmp_path):
    """`nbody --devices=0,0,0,0` in FAST mode: BodySystemHIPSharded lends every shard the workspace the library asks for, so the
    step is the pairwise one across the shards; `--no-workspace` is the one-sided tile schedule.  Same trajectory up to summation
    order as each other and as the single-GPU run, different bits; --compare passes; the benchmark lines print."""
    cli = os.path.join(ROOT, "cuda-nbody_amd", "nbody")
    n = 32768
    dumps = {}
    for name, extra in (("pairwise", ["--devices=0,0,0,0"]), ("one_sided", ["--devices=0,0,0,0", "--no-workspace"]), ("single", [])):
        dump = tmp_path / f"{name}.bin"
        r = subprocess.run([cli, f"--numbodies={n}", "--steps=3", f"--dump={dump}", *extra], env=_env(), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        dumps[name] = np.fromfile(dump, dtype=np.float32)
    assert dumps["pairwise"].tobytes() != dumps["one_sided"].tobytes()
    np.testing.assert_allclose(dumps["pairwise"], dumps["one_sided"], rtol=5e-5, atol=5e-5)
    np.testing.assert_allclose(dumps["pairwise"], dumps["single"], rtol=5e-5, atol=5e-5)
    r = subprocess.run([cli, "--compare", f"--numbodies={n}", "--devices=0,0,0"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "multiple of the number of devices" in r.stderr  # 32 768 bodies do not shard over 3 devices
    r = subprocess.run([cli, "--compare", f"--numbodies={n}", "--devices=0,0,0,0"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "  OK" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([cli, "--benchmark", f"--numbodies={n}", "--devices=0,0", "-i", "4"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "billion interactions per second" in r.stdout, r.stdout + r.stderr
    # round 6: a run over several devices says what the HOST needed to enqueue a step (after the reference's three lines)
    m = re.search(r"^= ([\d.]+) ms of host time to enqueue a step \(host_enqueue_ms_per_step; ([\d.]+) ms per step on the devices\)$", r.stdout, re.M)
    assert m and 0 < float(m.group(1)) < 50 and float(m.group(2)) > 0, r.stdout
    # ... with the crew of threads (default) and with the calling thread alone (NBODY_STEP_THREADS=0): the very same bits
    for name, threads in (("crew", "1"), ("alone", "0")):
        dump = tmp_path / f"threads_{name}.bin"
        r = subprocess.run([cli, f"--numbodies={n}", "--steps=3", f"--dump={dump}", "--devices=0,0,0,0"], env={**_env(), "NBODY_STEP_THREADS": threads}, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.fromfile(dump, dtype=np.float32).tobytes() == dumps["pairwise"].tobytes(), name


@pytest.mark.gpu
def test_config4_shape_eight_ranks_at_one_mi_bodies(tmp_path, oracle):
    """BASELINE configs[3]: 1 048 576 bodies sharded over 8 ranks (131 072 each), FAST with workspaces -- the pairwise step across
    the ranks, every position tile and reaction array through the transport -- one step, all 8 ranks on the one GPU: every rank
    ends with the positions one rank computes (pairwise single-GPU step), to summation accuracy."""
    n, world = 1048576, 8
    pos0, vel0 = oracle.startup_state(n, np.float32)
    got = _run(tmp_path, "all", pos0, vel0, world, 1, "fast", workspace=True)
    assert np.all(got["workspace_bytes"] > 0)
    for k in range(1, world):
        assert got[f"pos_{k}"].tobytes() == got["pos_0"].tobytes()
    np.testing.assert_allclose(got["pos_0"], got["single_pos"], rtol=0, atol=5e-5)
    vel = np.concatenate([got[f"vel_{k}"] for k in range(world)])
    np.testing.assert_allclose(vel, got["single_vel"], rtol=0, atol=1e-2)
    # round 5: the same step as an in-process world over the REAL RCCL (2 MiB position tiles, 1.5 MiB reaction arrays; slices of
    # 131 072 bodies: the diagonal is one launch at this size) -- the very bits of the double, on every rank
    real = _run(tmp_path, "all", pos0, vel0, world, 1, "fast", workspace=True, real_rccl=True)
    assert list(real["layout"]) == [1] * world
    for k in range(world):
        assert real[f"pos_{k}"].tobytes() == got["pos_0"].tobytes()
    assert np.concatenate([real[f"vel_{k}"] for k in range(world)]).tobytes() == vel.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("workspace", [False, True])
def test_position_exchange_grouping_is_a_setting_not_a_result(tmp_path, oracle, workspace):
    """nb_comm_set_exchange_grouping(comm, 1): all G-1 position rounds of a step in ONE RCCL group (round 4's default) instead of a
    group and an event per round (the default since round 5: measured, profiles/round5_exchange_contention.jsonl) -- a
    per-communicator setting, so that one multi-GPU job can time both: the same bits -- STRICT == the CPU path, FAST (one-sided
    tiles and pairwise across ranks) == the default.  The environment variable only sets the default."""
    n, steps, world = 4096, 4, 4
    pos0, vel0 = oracle.startup_state(n, np.float32)
    ref_p, ref_v = pos0.copy(), vel0.copy()
    oracle.update(ref_p, ref_v, np.float32(0.016), steps=steps)
    reaction = (world // 2) * steps if workspace else 0  # (the reaction rounds of the pairwise step: always a group each)
    strict = _run(tmp_path, "all", pos0, vel0, world, steps, "strict", workspace=workspace, WORKER_ONE_GROUP="1")
    assert strict["pos_0"].tobytes() == ref_p.tobytes()
    sends, recvs, gathers, groups, copies = strict["counters"]
    assert sends == recvs == (world - 1) * world * steps and groups == steps * world  # one group per step (and rank: each from its own thread)
    fast = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=workspace, WORKER_ONE_GROUP="1")
    default = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=workspace)
    assert default["counters"][3] == ((world - 1) * steps + reaction) * world and fast["counters"][3] == (steps + reaction) * world
    explicit = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=workspace, WORKER_ONE_GROUP="0", NBODY_EXCHANGE_ONE_GROUP="1")  # the API outranks the variable
    by_env = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=workspace, NBODY_EXCHANGE_ONE_GROUP="1")  # ... which is the default only
    assert explicit["counters"][3] == default["counters"][3] and by_env["counters"][3] == fast["counters"][3] != default["counters"][3]
    for k in range(world):
        assert fast[f"pos_{k}"].tobytes() == default["pos_0"].tobytes()
        assert explicit[f"pos_{k}"].tobytes() == default["pos_0"].tobytes() and by_env[f"pos_{k}"].tobytes() == default["pos_0"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["threads", "all"])
def test_layout_is_agreed_by_the_whole_communicator(tmp_path, oracle, case):
    """a rank that was lent no workspace must not step one-sidedly while its peers step pairwise (they would sit in
    unmatched reaction rounds).  Rank 2 of 4 passes (NULL, 0) to nb_comm_set_workspace: with a thread per rank the call is a
    collective and every rank learns the smallest amount; with one thread for all ranks the library sees them all.  Either way
    nb_comm_layout_* says one-sided on EVERY rank, the transport sees no reaction round, and the result is the bits of a run in
    which nobody lent anything."""
    n, steps, world = 4096, 3, 4
    pos0, vel0 = oracle.startup_state(n, np.float32)
    got = _run(tmp_path, case, pos0, vel0, world, steps, "fast", workspace=True, WORKER_NO_WORKSPACE_RANK="2")
    assert list(got["layout"]) == [0] * world
    sends, recvs, gathers, groups, copies = got["counters"]
    assert sends == recvs == (world - 1) * world * steps  # position rounds only
    plain = _run(tmp_path, case, pos0, vel0, world, steps, "fast", workspace=False)
    for k in range(world):
        assert got[f"pos_{k}"].tobytes() == plain["pos_0"].tobytes()
    everyone = _run(tmp_path, case, pos0, vel0, world, steps, "fast", workspace=True)
    assert list(everyone["layout"]) == [1] * world and everyone["pos_0"].tobytes() != plain["pos_0"].tobytes()


# ------------------------------------------------------------------------------------------------ ... and through the REAL library
@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_even_worlds_through_the_real_rccl_in_one_process(tmp_path, oracle, world):
    """RCCL refuses two ranks on one device, so every test above runs on the transport double.  An IN-PROCESS world
    (nb_comm_inprocess_open_all, tuning header) closes the gap for the world sizes the benchmark runs at: all G ranks in one process
    on the one GPU share ONE real one-rank ncclComm, every transfer is a self-transfer, and the library routes rank a's send to
    rank b's receive by the order in which it issues them (RCCL matches the sends and receives of one peer first in, first out).
    nb_sharded_step_all_f32 is then the FULL G-rank step -- split rectangle, reaction leg in ready order, late diagonal, the group of
    tiles nobody waits for -- through the product's own calls into the real ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd:
    STRICT == the CPU path bit for bit on every rank, FAST pairwise across the ranks within the single-GPU tolerance and the same
    bits as the same step over the transport double."""
    n, steps = world * 1024, 3
    pos0, vel0 = oracle.startup_state(n, np.float32)
    ref_p, ref_v = pos0.copy(), vel0.copy()
    oracle.update(ref_p, ref_v, np.float32(0.016), steps=steps)
    strict = _run(tmp_path, "all", pos0, vel0, world, steps, "strict", real_rccl=True)
    for k in range(world):
        assert strict[f"pos_{k}"].tobytes() == ref_p.tobytes()
    assert np.concatenate([strict[f"vel_{k}"] for k in range(world)]).tobytes() == ref_v.tobytes()
    fast = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True, real_rccl=True)
    assert list(fast["layout"]) == [1] * world
    for k in range(1, world):
        assert fast[f"pos_{k}"].tobytes() == fast["pos_0"].tobytes()
    np.testing.assert_allclose(fast["pos_0"], ref_p, rtol=1e-5, atol=1e-5)
    double = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True)
    assert double["pos_0"].tobytes() == fast["pos_0"].tobytes()  # the transport changes nothing: the same sums in the same order
    one_sided = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=False, real_rccl=True)
    np.testing.assert_allclose(one_sided["pos_0"], ref_p, rtol=1e-5, atol=1e-5)
    assert one_sided["pos_0"].tobytes() != fast["pos_0"].tobytes()


@pytest.mark.gpu
def test_full_size_eight_ranks_through_the_real_rccl(tmp_path, oracle):
    """BASELINE configs[2] as the strong-scaling series shards it over 8 ranks -- 262 144 bodies, slices of 32 768 -- as an in-process
    world over the REAL RCCL: STRICT two steps == one rank's nb_integrate_f32 on ALL bodies, bit for bit (512 KiB position tiles);
    FAST pairwise across the ranks (384 KiB reaction arrays, the late diagonal, rounds in ready order) one step against one rank at the
    tolerance of the transport-double test above, and the very bits the double gives."""
    n, world = 262144, 8
    pos0, vel0 = oracle.startup_state(n, np.float32)
    strict = _run(tmp_path, "all", pos0, vel0, world, 2, "strict", real_rccl=True)
    for k in range(world):
        assert strict[f"pos_{k}"].tobytes() == strict["single_pos"].tobytes()
    assert np.concatenate([strict[f"vel_{k}"] for k in range(world)]).tobytes() == strict["single_vel"].tobytes()
    fast = _run(tmp_path, "all", pos0, vel0, world, 1, "fast", workspace=True, real_rccl=True)
    assert list(fast["layout"]) == [1] * world
    for k in range(1, world):
        assert fast[f"pos_{k}"].tobytes() == fast["pos_0"].tobytes()
    np.testing.assert_allclose(fast["pos_0"], fast["single_pos"], rtol=0, atol=2e-5)
    double = _run(tmp_path, "all", pos0, vel0, world, 1, "fast", workspace=True)
    assert double["pos_0"].tobytes() == fast["pos_0"].tobytes()


@pytest.mark.gpu
def test_fp64_four_ranks_through_the_real_rccl_in_one_process(tmp_path, oracle):
    """... and in double precision (ncclFloat64 tiles of 32 B per body, reaction arrays of 24 B): STRICT == the CPU path bit for bit, FAST
    pairwise across the ranks at the fp64 tolerance and the bits of the transport double."""
    n, steps, world = 4096, 3, 4
    pos0, vel0 = oracle.startup_state(n, np.float64)
    ref_p, ref_v = pos0.copy(), vel0.copy()
    oracle.update(ref_p, ref_v, np.float64(np.float32(0.016)), steps=steps)
    strict = _run(tmp_path, "all", pos0, vel0, world, steps, "strict", real_rccl=True)
    for k in range(world):
        assert strict[f"pos_{k}"].tobytes() == ref_p.tobytes()
    fast = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True, real_rccl=True)
    assert list(fast["layout"]) == [1] * world
    np.testing.assert_allclose(fast["pos_0"], ref_p, rtol=1e-12, atol=1e-12)
    double = _run(tmp_path, "all", pos0, vel0, world, steps, "fast", workspace=True)
    assert double["pos_0"].tobytes() == fast["pos_0"].tobytes()
