"""`python3 bench.py --gpus N` from a plain shell starts its own ranks and carries one
fallback attempt (bench.py self_launch)."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, timeout=600, extra_env=None):
    env = dict(os.environ)
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(name, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=env, timeout=timeout)


def test_launcher_falls_back_and_reports_failure():
    """No GPU here: every attempt must fail; the retries must be, in this order, the tile schedule over torch.distributed, the
    all-gather form and the RCCL-free one, and the status must say so."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU: on a GPU box the first attempt would go on to RCCL")
    out = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--bodies", "1024", "--no-cpu-baseline", "--launch-timeout", "120", "--launch-budget", "480")
    assert out.returncode != 0
    order = [out.stderr.find(f"one more attempt with --exchange {name}") for name in ("torch", "allgather", "staged")]
    assert all(at >= 0 for at in order) and order == sorted(order), out.stderr[-2000:]
    assert out.stderr.count("one more attempt with --exchange") == 3
    assert '"metric"' not in out.stdout


def test_launcher_does_not_retry_an_exchange_the_caller_chose():
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU")
    out = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--bodies", "1024", "--no-cpu-baseline", "--exchange", "allgather", "--launch-timeout", "240")
    assert out.returncode != 0
    assert "one more attempt" not in out.stderr


def test_launcher_ends_hung_attempts_inside_its_budget(tmp_path):
    """the launcher's fallbacks must be reachable inside what the caller allows the whole command.  Stand-in
    ranks (self_launch's make_cmd): the first attempt imports and then never brings its exchange up (ended `bringup_s` after the
    import mark), `--exchange torch` comes up and then hangs (ended at the attempt's limit), `--exchange allgather` reports.
    The launcher must end both hung process groups, reach the third attempt, relay its line, return 0 -- all inside the budget."""
    import time

    rank = tmp_path / "stand_in_rank.py"
    rank.write_text(
        "import sys, time\n"
        "print('[bench rank 0] torch imported', file=sys.stderr, flush=True)\n"
        "how = sys.argv[sys.argv.index('--exchange') + 1] if '--exchange' in sys.argv else 'rccl'\n"
        "if how == 'rccl':\n"
        "    time.sleep(600)\n"
        "print(f'[bench rank 0] up: exchange {how}', file=sys.stderr, flush=True)\n"
        "if how == 'torch':\n"
        "    time.sleep(600)\n"
        "print('{\"metric\": \"stand-in\", \"exchange\": \"%s\"}' % how, flush=True)\n")
    driver = tmp_path / "driver.py"
    driver.write_text(
        f"import sys, time\nsys.path.insert(0, {ROOT!r})\nimport bench\n"
        f"t0 = time.monotonic()\n"
        f"rc = bench.self_launch(2, False, attempt_s=8.0, budget_s=60.0, bringup_s=2.0, import_s=5.0, make_cmd=lambda extra, port: [sys.executable, {str(rank)!r}] + extra)\n"
        f"print('LAUNCHER', rc, round(time.monotonic() - t0, 1), flush=True)\n")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=120)
    took = time.monotonic() - t0
    assert out.returncode == 0, out.stderr[-3000:]
    assert "LAUNCHER 0 " in out.stdout and out.stdout.count('"metric"') == 1 and '"exchange": "allgather"' in out.stdout
    at = [out.stderr.find(text) for text in ("no rank has its exchange up", "one more attempt with --exchange torch", "ranks still running after 8 s", "one more attempt with --exchange allgather")]
    assert all(a >= 0 for a in at) and at == sorted(at), out.stderr[-3000:]
    assert "--exchange staged" not in out.stderr  # (the third attempt reported: nothing plainer was needed)
    assert took < 40, took  # 2-3 s (no bring-up) + 8 s (hung after bring-up) + the attempt that reports: well inside the 60 s budget


def test_launcher_does_not_charge_a_slow_import_to_the_attempt(tmp_path):
    """A fresh box pages torch in for a minute or two.  An attempt's limit counts from the first rank's import mark, and once the line
    is out the attempt may finish its diagnostics inside what is left of the whole budget: a healthy run that imports for longer than
    `attempt_s` and keeps working after its line is relayed whole and returns 0, without a retry."""
    rank = tmp_path / "stand_in_rank.py"
    rank.write_text(
        "import sys, time\n"
        "time.sleep(6)\n"
        "print('[bench rank 0] torch imported', file=sys.stderr, flush=True)\n"
        "print('[bench rank 0] up: exchange rccl', file=sys.stderr, flush=True)\n"
        "time.sleep(2)\n"
        "print('{\"metric\": \"stand-in\"}', flush=True)\n"
        "time.sleep(5)\n"
        "print('[bench rank 0] diagnostics done', file=sys.stderr, flush=True)\n")
    driver = tmp_path / "driver.py"
    driver.write_text(
        f"import sys\nsys.path.insert(0, {ROOT!r})\nimport bench\n"
        f"rc = bench.self_launch(2, False, attempt_s=4.0, budget_s=60.0, bringup_s=2.0, import_s=20.0, make_cmd=lambda extra, port: [sys.executable, {str(rank)!r}] + extra)\n"
        f"print('LAUNCHER', rc, flush=True)\n")
    out = subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "LAUNCHER 0" in out.stdout and out.stdout.count('"metric"') == 1, (out.stdout, out.stderr[-2000:])
    assert "diagnostics done" in out.stderr and "ending process group" not in out.stderr and "one more attempt" not in out.stderr


def test_launcher_lets_ranks_that_are_up_and_slow_finish(tmp_path):
    """Round 6 (advisor): a healthy but slow configuration (large --bodies / --steps, fp64, strict) used to be ended at the attempt's
    limit and retried three times with exchanges that are slower still.  Ranks that are up say how long they expect to need (two probe
    steps); the launcher leaves them alone for twice that + 30 s, inside the whole budget -- and still ends ranks that said nothing."""
    rank = tmp_path / "stand_in_rank.py"
    rank.write_text(
        "import sys, time\n"
        "print('[bench rank 0] torch imported', file=sys.stderr, flush=True)\n"
        "print('[bench rank 0] up: exchange rccl', file=sys.stderr, flush=True)\n"
        "print('[bench rank 0] expects 6.0 s until the headline (300.000 ms per step by two probe steps)', file=sys.stderr, flush=True)\n"
        "time.sleep(9)\n"
        "print('{\"metric\": \"stand-in\"}', flush=True)\n")
    driver = tmp_path / "driver.py"
    driver.write_text(
        f"import sys\nsys.path.insert(0, {ROOT!r})\nimport bench\n"
        f"rc = bench.self_launch(2, False, attempt_s=3.0, budget_s=90.0, bringup_s=2.0, import_s=5.0, make_cmd=lambda extra, port: [sys.executable, {str(rank)!r}] + extra)\n"
        f"print('LAUNCHER', rc, flush=True)\n")
    out = subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "LAUNCHER 0" in out.stdout and out.stdout.count('"metric"') == 1, (out.stdout, out.stderr[-2000:])
    assert "ending process group" not in out.stderr and "one more attempt" not in out.stderr


def test_launcher_stops_when_its_budget_is_spent(tmp_path):
    """Every attempt hangs after bring-up: the launcher shares what is left of the budget among the attempts that may follow, ends
    each, and returns a failure INSIDE the budget instead of running into the caller's own limit."""
    import time

    rank = tmp_path / "stand_in_rank.py"
    rank.write_text("import sys, time\nprint('[bench rank 0] torch imported', file=sys.stderr, flush=True)\nprint('[bench rank 0] up: x', file=sys.stderr, flush=True)\ntime.sleep(600)\n")
    driver = tmp_path / "driver.py"
    driver.write_text(
        f"import sys\nsys.path.insert(0, {ROOT!r})\nimport bench\n"
        f"rc = bench.self_launch(2, False, attempt_s=100.0, budget_s=16.0, bringup_s=2.0, import_s=5.0, make_cmd=lambda extra, port: [sys.executable, {str(rank)!r}] + extra)\n"
        f"print('LAUNCHER', rc, flush=True)\n")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=120)
    took = time.monotonic() - t0
    assert out.returncode == 0 and "LAUNCHER 0" not in out.stdout and "LAUNCHER" in out.stdout, out.stderr[-3000:]
    assert out.stderr.count("ranks still running after") >= 2 and '"metric"' not in out.stdout
    assert took < 16.0 + 12.0, took  # the budget, plus ending the last group and the interpreter start-ups


def test_chip_watch_reads_the_card_with_the_devices_pci_address(tmp_path):
    """bench_support.ChipWatch on a made-up sysfs tree: the card is found by PCI address (a box shows the cards of the whole host),
    the samples are the hwmon files' values in MHz and W, and a missing or unreadable tree gives None, never an error."""
    import time

    sys.path.insert(0, ROOT)
    from bench_support import ChipWatch

    for card, address, clock, power in ((0, "0000:05:00.0", 101000000, 244000000), (40, "0000:c5:00.0", 2256000000, 1354000000)):
        real = tmp_path / "devices" / address
        (real / "hwmon" / "hwmon7").mkdir(parents=True)
        (real / "hwmon" / "hwmon7" / "freq1_input").write_text(f"{clock}\n")
        (real / "hwmon" / "hwmon7" / "power1_input").write_text(f"{power}\n")
        (real / "hwmon" / "hwmon7" / "power1_cap").write_text("1400000000\n")
        (tmp_path / "drm" / f"card{card}").mkdir(parents=True)
        os.symlink(real, tmp_path / "drm" / f"card{card}" / "device")
    watch = ChipWatch("0000:C5:00.0", sysfs=str(tmp_path / "drm"))
    time.sleep(0.5)  # (the child process starts up: samples before start() must not count)
    before = time.time()
    watch.start()
    time.sleep(0.2)
    watch.stop()
    got = watch.summary()
    assert not os.path.exists(watch.path) and watch.child is None
    assert got["sclk_mhz"] == 2256.0 and got["socket_power_w"] == 1354.0 and got["power_cap_w"] == 1400.0 and 2 <= got["samples"] <= 12
    assert time.time() - before < 1.0
    for nothing in (ChipWatch(None, sysfs=str(tmp_path / "drm")), ChipWatch("0000:aa:00.0", sysfs=str(tmp_path / "drm")), ChipWatch("0000:c5:00.0", sysfs=str(tmp_path / "none"))):
        nothing.start(), nothing.stop()
        assert nothing.summary() is None and nothing.child is None


@pytest.mark.gpu
def test_plain_shell_bench_starts_two_ranks_on_one_gpu():
    """The rehearsal form (gloo, ranks sharing the one GPU): the launcher, the rank set-up, the sharded step and the
    JSON line of the N-rank path all run; the number is not a performance figure."""
    out = run_bench("--gpus", "2", "--exchange", "host", "--steps", "3", "--warmup", "1", "--bodies", "16384", "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0
    assert "REHEARSAL" in line["config"]["exchange"]


def _metric_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_one_rank_under_torchrun_runs_the_capi_sharded_step_and_matches_the_plain_run(tmp_path):
    """What the driver's scaling series does for N=1: `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`.  The step must be
    the product's nb_sharded_step_* (communicator of one rank), no fallback, and the final positions must be the same bytes as
    those of the plain `python bench.py` run (nb_integrate_shard_* directly)."""
    import socket

    import numpy as np

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    common = ["--gpus", "1", "--steps", "6", "--warmup", "2", "--bodies", "16384", "--no-cpu-baseline", "--no-configs"]
    env = dict(os.environ)
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(name, None)
    a, b = tmp_path / "torchrun.npz", tmp_path / "plain.npz"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), *common, "--dump-state", str(a)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _metric_line(out.stdout)
    assert line["config"]["step_entry_point"] == "nb_sharded_step_*" and line["exchange_fallback"] is False and line["n_gpus"] == 1
    plain = run_bench(*common, "--dump-state", str(b))
    assert plain.returncode == 0, plain.stderr[-3000:]
    pl = _metric_line(plain.stdout)
    assert pl["config"]["step_entry_point"] == "nb_integrate_ws_*" and pl["exchange_fallback"] is False
    with np.load(a) as ta, np.load(b) as pb:
        assert ta["initial"].tobytes() == pb["initial"].tobytes(), "the two runs did not start from the same bodies (the libc rand() stream was disturbed)"
        assert ta["final"].tobytes() == pb["final"].tobytes(), float(np.abs(ta["final"] - pb["final"]).max())


@pytest.mark.gpu
def test_default_line_carries_every_baseline_config():
    """`python bench.py` (what the driver runs): the one JSON line holds the headline AND, timed after it, the other BASELINE
    configs and STRICT."""
    out = run_bench("--steps", "5", "--warmup", "1", "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-3000:]
    line = _metric_line(out.stdout)
    got = {(c["bodies"], c["dtype"], c["mode"], c["layout"]) for c in line["configs"]}
    assert {(65536, "f32", "fast", "pairwise"), (65536, "f32", "fast", "one-sided"), (262144, "f64", "fast", "pairwise"), (262144, "f64", "fast", "one-sided"),
            (1048576, "f32", "fast", "pairwise"), (1048576, "f32", "fast", "one-sided"), (262144, "f32", "strict", "strict"), (262144, "f64", "strict", "strict"),
            (1024, "f32", "fast", "one-sided"), (1024, "f32", "strict", "strict"), (262144, "f32", "fast", "one-sided"),
            (50000, "f32", "fast", "pairwise"), (100000, "f32", "fast", "pairwise"), (300000, "f32", "fast", "pairwise")} <= got
    assert (262144, "f32", "fast", "pairwise") not in got  # that one IS the headline
    assert line["config"]["step_entry_point"] == "nb_integrate_ws_*" and line["config"]["kernel_plan"]["layout"] == "pairwise"
    # the protocol: untimed steps for ~0.3 s first (the card's clock ramp from idle; the count is stated), then W warm-up and exactly K timed steps
    assert line["warmup"] == 1 and line["steps"] == 5 and 10 <= line["settle_steps"] <= 80
    roof = line["roofline"]
    assert roof["executed"]["frac"] < roof["frac"] and roof["executed"]["frac"] < 1
    # the dominant kernel is timed on its own (an event between the two launches of a step); the two kernels add up to the step
    assert roof["kernel"] == "pair_forces" and roof["kernel_ms"] == roof["pair_forces_ms"] > 10 * roof["pair_finish_ms"] > 0
    assert abs(roof["pair_forces_ms"] + roof["pair_finish_ms"] - roof["stream_ms_per_step"]) < 0.05 * roof["stream_ms_per_step"]
    assert roof["step_frac"] <= 1.03 * roof["frac"]  # (the kernel is timed apart, AFTER the region the step comes from: the clock may have drifted a per cent or two)
    if roof["chip"] is not None and roof["chip"].get("sclk_mhz"):  # (sysfs readable: the power management's figure for the clock)
        assert 500 < roof["chip"]["sclk_mhz"] <= 2500 and roof["chip"]["samples"] >= 1
        assert 0.98 * roof["step_frac"] <= roof["frac_at_hwmon_clock"] < 1.25
    # round 6: the clock pair_forces REALLY ran at, read inside the kernel (pair_forces_clocked: every workgroup's lifetime on the shader-cycle
    # counter and on the constant 100 MHz counter) -- frac_at_delivered_clock and the cycle counts are computed from THIS one
    clock = roof["chip"]["delivered_clock"]
    assert roof["chip"]["delivered_mhz_by_kernel"] == clock["mhz"] and 1200 < clock["mhz"] < 2600 and clock["workgroups"] == line["config"]["kernel_plan"]["grid"]
    assert clock["mhz_p10"] <= clock["mhz"] <= clock["mhz_p90"]
    assert abs(roof["frac_at_delivered_clock"] - roof["step_frac"] * 2400.0 / clock["mhz"]) < 1e-9 and roof["frac_at_delivered_clock"] < 1.25  # (algorithmic flop: may pass 1, see frac_counts)
    assert abs(roof["pair_forces_mcycles"] - roof["pair_forces_ms"] * clock["mhz"] * 1e-3) < 1e-6 and 15 < roof["pair_forces_mcycles"] < 30
    # one workgroup per CU: the median workgroup lives 20.71 Mcycles on every box met; the launch as events time it (ramp, tail, the slowest
    # XCD's workgroups) is 2-7 % longer
    assert 20.3 < clock["workgroup_mcycles_median"] < 21.2 and clock["workgroup_mcycles_median"] <= roof["pair_forces_mcycles"] < 1.12 * clock["workgroup_mcycles_median"]
    assert line["exchange_path"] == "none" and 0 < line["host_enqueue_ms_per_step"] < line["ms_per_step"] + 1.0
    # algorithmic HBM bytes are the bodies in and out; the workspace traffic is stated next to them
    assert roof["algorithmic_hbm_bytes_per_launch"] == 64 * 262144
    assert roof["workspace_rw_bytes_per_step"] == 2 * line["config"]["kernel_plan"]["workspace_bytes"]
    projection = line["multi_gpu_kernel_projection"]
    assert "PROJECTION" in projection["what"] and set(projection["ranks"]) == {"2", "4", "8"}
    assert projection["ranks"]["8"]["kernel_ms"] < projection["ranks"]["2"]["kernel_ms"] < line["ms_per_step"]
    # round 5: the same rank's whole step with the REAL RCCL's kernels on the chip (a loopback rank, in a child process)
    loop = projection["loopback"]
    assert "error" not in loop and loop["rccl_version"] >= 20000 and "fake" not in loop["rccl_library"], loop
    assert set(loop["ranks"]) == {"2", "4", "8"} and loop["ranks"]["8"]["step_ms"] < loop["ranks"]["4"]["step_ms"] < loop["ranks"]["2"]["step_ms"] < line["ms_per_step"]
    assert all(abs(v["exposed_exchange_ms"]) < 0.25 * v["step_ms"] for v in loop["ranks"].values())
    # round 6: what the host needs to enqueue a step over several ranks, from the same child processes (one rank = what one thread of the
    # crew enqueues; the in-process world = eight ranks that share ONE communicator: its RCCL groups cannot be issued in parallel)
    enq = projection["host_enqueue"]
    assert "error" not in enq and 0 < enq["one_rank_of_8_plain_process_ms"] < 1.0 and 0 < enq["one_rank_of_8_torch_process_ms"] < 3.0 and 0 < enq["in_process_world_8_ranks"]["crew_ms"] < 5.0, enq
    assert enq["in_process_world_8_ranks"]["crew_ms"] <= 1.2 * enq["in_process_world_8_ranks"]["calling_thread_alone_ms"]
    for c in line["configs"]:
        assert "frac" not in c  # (round 5: no bare "frac" that reads as utilisation)
        if c["dtype"] == "f64":  # round 6: every fp64 entry says where its peak comes from, and what the issue rate measured on the box allows
            assert "NOT confirmed" in c["fp64_peak_source"] and 60 < c["fp64_issue_peak_tflops"] < c["fp64_peak_tflops"] == 78.6
        if c["layout"] == "pairwise":
            assert "half" in c["frac_algorithmic_counts"]
        if c["workload"] == "configs[1]":
            assert c["lds_tile"] is None and "round2_scalar_stream_ab" in c["why"]
        assert c["ms_per_step"] > 0 and 0 < c["frac_algorithmic"] < 1.3  # (pairwise: the algorithmic count may pass the one-sided peak)
        assert 0 < c["executed_frac"] < 1  # ... which is why the flop really issued stand next to it, and never pass 1
        assert (c["executed_frac"] < c["frac_algorithmic"]) == (c["layout"] == "pairwise")
        assert c["valu_busy"] is None or 0.1 < c["valu_busy"] <= 1.02  # the hardware's own figure (a ratio of two sampled counters: the fp64 kernel reads 1.007), where a committed PMC pass matches the plan
    by_size = {(c["bodies"], c["layout"]): c for c in line["configs"] if c["dtype"] == "f32" and c["mode"] == "fast"}
    for n in (16384, 65536):  # the hipGraph form timed next to the eager steps
        assert by_size[(n, "pairwise")]["hipgraph_ms_per_step"] > 0 and -0.5 < by_size[(n, "pairwise")]["hipgraph_gain"] < 1.0
    # the headline's own counters and the bytes the step really moves against SURVEY 8(d)'s 64 N
    if roof["traffic"] is not None:
        assert 0.5 < roof["valu_busy"] <= 1.0 and roof["wasted_traffic_ratio"]["ratio"] > 10 and roof["wasted_traffic_ratio"]["kernels_counted"] == ["pair_forces", "pair_finish"]
    f64 = next(c for c in line["configs"] if (c["bodies"], c["dtype"], c["layout"]) == (262144, "f64", "pairwise"))
    assert f64["executed_frac"] < 0.8


@pytest.mark.gpu
def test_emulated_rank_of_a_pairwise_multi_gpu_step():
    """`tools/kernel_sweeps.py --emulate-gpus 8`: one rank's kernels of the pairwise step across 8 ranks (nb_emulate_pair_rank_*),
    timed on the one GPU -- the compute side of the strong-scaling projection; and the one-sided tile schedule for comparison."""
    def sweep(*flags):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_sweeps.py"), "--emulate-gpus", "8", "--steps", "5", "--warmup", "1", *flags],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])

    line = sweep()
    assert line["emulated_gpus"] == 8 and "pairwise" in line["schedule"] and len(line["ranks"]) == 3
    pair_ms = max(r["ms_per_step_kernels_only"] for r in line["ranks"])
    one_ms = max(r["ms_per_step_kernels_only"] for r in sweep("--layout", "one-sided")["ranks"])
    assert 0 < pair_ms < one_ms


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 4])
def test_n_rank_line_rehearsed_on_one_gpu_through_the_capi(ranks):
    """`bench.py --gpus N --rehearse-one-gpu`: N PROCESSES (as the driver starts them), each through the product's own multi-GPU
    entry points -- nb_comm_unique_id on rank 0 and its broadcast, nb_comm_init_rank, the collective nb_comm_set_workspace,
    nb_sharded_step_* -- with RCCL replaced by the cross-process test double, all on the one GPU.  Never a performance number;
    what it shows is that the N > 1 line comes out whole: the layout the communicator agreed on, what every rank's communicator
    says about itself, the A/B timings of the diagnostics (both exchange groupings x both layouts, the exchange legs alone, the
    kernels alone) and BASELINE configs[3]'s shape through the same communicator."""
    out = run_bench("--gpus", str(ranks), "--rehearse-one-gpu", "--steps", "3", "--warmup", "1", "--bodies", "16384", "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-4000:]
    line = _metric_line(out.stdout)
    assert line["n_gpus"] == ranks and line["exchange_fallback"] is False and line["value"] > 0
    assert line["config"]["step_entry_point"] == "nb_sharded_step_*" and "REHEARSAL" in line["config"]["exchange"] and "PAIRWISE" in line["config"]["exchange"]
    # round 6: a reader of the parsed line alone can tell the native path from a fall-back -- the path in one word, how many
    # ranks stepped through a REAL RCCL (here none: the double), its version -- and has the host's enqueue time per step
    assert line["exchange_path"] == "rccl-tiles" and line["rccl_ranks_seen"] == 0 and line["rccl_version"] is None
    assert 0 < line["host_enqueue_ms_per_step"] and line["host_enqueue"]["last_step_by_the_library_ms"] > 0
    assert "ONE GPU" in line["defaults_provisional"]
    # ... and the N = 1 figure of this very process tree (rank 0 alone on its GPU after the timed region, the others waiting at the barrier)
    single = line["single_gpu_same_process"]
    assert single["value"] > 0 and single["ms_per_step"] > 0 and abs(single["speedup_of_this_line"] - line["value"] / single["value"]) < 0.01
    assert line["config"]["exchange_grouping"].startswith("one RCCL group per position round")  # (round 5's default: measured, profiles/round5_exchange_contention.jsonl)
    seen = line["ranks_seen"]
    assert [r["rank"] for r in seen] == list(range(ranks)) and all(r["world"] == ranks and r["pairwise"] and not r["one_group"] and r["workspace_bytes"] > 0 for r in seen)
    assert all("chip" in r for r in seen)  # (every rank's own card: clock and power while the headline was timed; None where sysfs says nothing)
    # the record alone says which RCCL carried the bytes (here: the double, version 0), which card a rank ran on, what
    # the rank executed per step and how long its own stream took
    assert all(r["rccl_version"] == 0 and "fake_rccl" in r["rccl_library"] for r in seen)
    assert all(r["pci"] == seen[0]["pci"] and r["stream_ms_per_step"] > 0 for r in seen)  # (the rehearsal: one card)
    assert all(0 < r["host_enqueue_ms_per_step"] <= line["host_enqueue_ms_per_step"] * 1.001 for r in seen)  # (round 6: every rank's own enqueue time; the line's is their maximum)
    # ... and on what kind of stream it stepped: one from nb_comm_stream_create, never the null stream (-1 / 0 / 1; with several processes
    # on ONE card the spin-kernel probe behind the flag is noise, so only its presence is held here -- the flag itself: test_sharded_gpu.py)
    assert all(r["caller_stream_badly_placed"] in (-1, 0, 1) and r["side_stream_collisions"] >= -1 for r in seen)
    assert all(r["pair_work"]["pair_evaluations_per_step"] > 0 and r["pair_work"]["force_launches_per_step"] == 2 + ranks // 2 for r in seen)
    total = sum(r["pair_work"]["pair_evaluations_per_step"] for r in seen)
    assert 0.5 * 16384 ** 2 <= total <= 0.75 * 16384 ** 2  # every pair once, plus the half-kept diagonal blocks
    executed = line["roofline"]["executed"]
    assert executed["pair_evaluations_per_step_this_rank"] == seen[0]["pair_work"]["pair_evaluations_per_step"] and 0 < executed["frac"] < 1
    diag = line["diagnostics"]
    assert set(diag["step_ms"]) == {"pairwise_one_group", "pairwise_group_per_round", "one_sided_one_group", "one_sided_group_per_round", "pairwise_diagonal_first_group_per_round",
                                    "pairwise_both_streams_end_on_local_work_group_per_round"}  # (round 6: the cut rectangle, timed beside the shipping order)
    assert diag["headline_was"] == "pairwise_group_per_round"
    assert len(diag["by_rank"]["pairwise_kernels_alone_ms"]) == ranks and all(v > 0 for v in diag["by_rank"]["pairwise_kernels_alone_ms"])
    assert all(v > 0 for v in diag["step_ms"].values())
    assert set(diag["position_exchange_alone_ms"]) == {"one_group", "group_per_round"} and diag["reaction_exchange_alone_ms"] > 0
    assert 0 < diag["pairwise_kernels_alone_ms"] and 0 < diag["one_sided_kernels_alone_ms"]
    (big,) = line["configs"]
    assert big["bodies"] == 65536 and big["n_gpus"] == ranks and big["layout"] == "pairwise across ranks" and big["ms_per_step"] > 0 and big["workspace_bytes_per_rank"] > 0
    assert "diagnostics_incomplete" not in line and "diagnostics_error" not in line


def test_line_takes_counters_only_from_a_profile_of_the_same_launch_plan(pkg):
    """bench_support.pmc_summary / plan_dict (host logic): `valu_busy`, `traffic` and `wasted_traffic_ratio` in the bench line come from
    the committed rocprofv3 --pmc passes -- the NEWEST round's file for the configuration, and only if it was taken with the launch
    plan that runs now; a configuration without a pass says None, never a neighbour's figure."""
    import numpy as np

    sys.path.insert(0, ROOT)
    from bench_support import plan_dict, pmc_summary

    plan = plan_dict(pkg, 262144, np.float32, "pairwise")
    assert plan["layout"] == "pairwise" and plan["bodies_per_lane"] == 16 and plan["grid"] == 256 and plan["workspace_bytes"] == 402653184
    got = pmc_summary(262144, False, "fast", "pairwise", plan)
    assert got["source"].startswith("profiles/round6_n262144_f32_pairwise_pmc_summary.json") and 0.9 < got["valu_busy"] <= 1.0 and got["hbm_bytes_per_launch"] > 8e8
    finish = pmc_summary(262144, False, "fast", "pairwise", plan, kernel="_finish")
    assert "finish" in finish["source"] and finish["hbm_bytes_per_launch"] > 4e8 and finish["valu_busy"] < 0.5  # the HBM-bound kernel of the path
    other = dict(plan, workgroups_per_block=2)
    stale = pmc_summary(262144, False, "fast", "pairwise", other)
    assert set(stale) == {"source"} and "another launch plan" in stale["source"]
    assert pmc_summary(123456, False, "fast", "pairwise", None) is None  # no pass for this size
    strict = pmc_summary(262144, False, "strict", "strict", None)
    assert "strict" in strict["source"] and strict["valu_busy"] > 0.9
    one_sided = plan_dict(pkg, 262144, np.float32, "one-sided")
    assert one_sided == {"bodies_per_lane": 4, "lane_groups": 8, "lds_tile_bodies": 2048, "grid": 1024, "lds_bytes": one_sided["lds_bytes"]}


@pytest.mark.gpu
def test_a_hung_rccl_bring_up_ends_in_a_line_without_rccl():
    """Under the driver's own launch (torch.distributed.run, no launcher of ours around it) a rank that never comes back from
    ncclCommInitRank used to end the job with the headline watchdog's status 5 and no line.  nb_comm_init_rank runs in a thread with a
    limit (--rccl-init-timeout): here rank 1's call never returns (the double's FAKE_RCCL_HANG_INIT), rank 0's does; the ranks decide
    TOGETHER, and all of them step with the exchange that needs no RCCL at all -- a line marked exchange_fallback, never a hang."""
    out = run_bench("--gpus", "2", "--rehearse-one-gpu", "--steps", "3", "--warmup", "1", "--bodies", "16384", "--no-cpu-baseline", "--rccl-init-timeout", "5",
                    extra_env={"FAKE_RCCL_HANG_INIT": "1"}, timeout=300)
    assert out.returncode == 0, out.stderr[-4000:]
    line = _metric_line(out.stdout)
    assert line["n_gpus"] == 2 and line["exchange_fallback"] is True and line["value"] > 0
    assert "RCCL's bring-up hung on some rank" in out.stderr and "nb_comm_init_rank has not returned after 5 s" in out.stderr
    assert line["config"]["step_entry_point"] != "nb_sharded_step_*" and "gloo" in line["config"]["exchange"].lower()
    assert line["exchange_path"] == "staged" and line["rccl_ranks_seen"] == 0  # (round 6: nobody can read this line as one of the native path)


@pytest.mark.gpu
def test_a_rank_stuck_in_an_exchange_says_where_and_leaves():
    """What cannot be fallen back from -- a rank that never returns from an RCCL group once the communicators are up (its peers'
    kernels would wait on the GPU for ever) -- must at least end, and say where: the headline watchdog (--headline-timeout) names the
    stage the rank is stuck in and leaves with status 5, so that the job fails inside the caller's limit with its reason on stderr."""
    import time

    t0 = time.monotonic()
    out = run_bench("--gpus", "2", "--rehearse-one-gpu", "--steps", "3", "--warmup", "1", "--bodies", "16384", "--no-cpu-baseline", "--headline-timeout", "8",
                    extra_env={"FAKE_RCCL_HANG_GROUP_END": "1", "FAKE_RCCL_TIMEOUT_S": "100"}, timeout=300)
    took = time.monotonic() - t0
    assert out.returncode != 0 and '"metric"' not in out.stdout, out.stdout[-2000:]
    assert re.search(r"no headline after \d+ s: stuck in 'C-ABI communicator bring-up", out.stderr) and "leaving with status 5" in out.stderr, out.stderr[-3000:]
    assert took < 90, took


@pytest.mark.gpu
def test_stalled_diagnostics_never_cost_the_headline(tmp_path):
    """Everything after the timed region of an N > 1 run is collective work under a watchdog (--diagnostics-timeout): when it stalls --
    here every RCCL group hangs from the moment the ranks say the headline is timed -- rank 0 prints the line with what it has
    ("diagnostics_incomplete"), first, and every rank leaves with status 0: the measured headline is never lost to what explains it."""
    import threading

    env = dict(os.environ)
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(name, None)
    trigger = tmp_path / "hang_now"
    env.update({"FAKE_RCCL_HANG_WHEN_EXISTS": str(trigger), "FAKE_RCCL_TIMEOUT_S": "100"})
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--steps", "3", "--warmup", "1", "--bodies", "16384", "--no-cpu-baseline",
                              "--diagnostics-timeout", "6"], stdout=open(tmp_path / "out.txt", "w"), stderr=subprocess.PIPE, text=True, env=env)
    err = []

    def watch():
        for text in child.stderr:
            err.append(text)
            if "post-headline measurements start" in text and not trigger.exists():
                trigger.write_text("now\n")

    watcher = threading.Thread(target=watch, daemon=True)
    watcher.start()
    try:
        child.wait(timeout=240)
    except subprocess.TimeoutExpired:
        child.kill()
        raise
    watcher.join(timeout=10)
    out = (tmp_path / "out.txt").read_text()
    stderr = "".join(err)
    assert child.returncode == 0, stderr[-3000:]
    line = _metric_line(out)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["exchange_fallback"] is False
    assert "did not finish within 6 s" in line["diagnostics_incomplete"] and "diagnostics" not in line
    assert stderr.count("post-headline measurements stalled: leaving") == 2

