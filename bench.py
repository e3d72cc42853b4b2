#!/usr/bin/env python3
"""bench.py -- headline benchmark: body-body interactions/s of the all-pairs N-body step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bodies B] [--fp64] [--mode fast|strict]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (integrateNbodySystem: N^2 interactions + leapfrog update) over the
synthetic SHELL system the reference itself starts from.  Workload at any N GPUs: BASELINE.json configs[2],
262 144 bodies fp32 (the configuration the metric is quoted on), total size fixed => "scaling": "strong";
bodies shard across ranks with one RCCL all-gather of the new positions per step, issued as position tiles by the product's own
multi-GPU entry points (nb_comm_init_rank + nb_sharded_step_*, csrc/nbody_comm.hip); torch.distributed (gloo) only does the
rendezvous, the barriers and the time reduction.  cuda-nbody_amd/sharded.py, the same schedule over torch.distributed, stays as
`--exchange torch` for A/B and as the fallback.  At N=1 the line also carries "configs": the other BASELINE configs and STRICT,
timed after the headline measurement; at N>1 it carries BASELINE configs[3] (1 048 576 bodies over the N ranks), "ranks_seen"
(what every rank's communicator says about itself) and "diagnostics" (the same job timed with the other exchange grouping,
one-sided, the exchange legs alone, the kernels alone) -- all taken after the timed region, under a watchdog that prints the
line without them should they stall.  Tuning sweeps and one-rank projections live in tools/kernel_sweeps.py.

Metric conventions are the reference's (src/nbody/compute.cpp:16-18,105-121): interactions/step = N^2
(self-interaction counted), 20 flop per fp32 interaction, 30 per fp64.  Timing protocol: W untimed warm-up
steps, then exactly K steps between barrier+synchronize pairs, max over ranks (the reference's GPU protocol,
compute_cuda.cpp:183-195, is 1 warm-up + K between two events).

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel against the fp32 vector-FMA peak
(157.3 TFLOP/s, MI355X_MICROARCH.md chip table) from HIP-event timing of the kernel's own stream;
`cpu_baseline` is the CPU oracle (oracle/, a port of the reference's CPU path) timed on this host.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md:41
FP64_VECTOR_PEAK_TFLOPS = 78.6   # public spec
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bodies", type=int, default=262144)
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--mode", choices=["fast", "strict"], default="fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-bodies", type=int, default=0, help="bodies i in the CPU sample (0 = auto, ~10 s)")
    ap.add_argument("--plan", type=str, default="", help="I,S,TILE override for the one-sided fast kernel, e.g. 2,8,1024 (tuning; implies --layout one-sided)")
    ap.add_argument("--exchange", choices=["rccl", "torch", "allgather", "staged", "host", "host-tiles"], default="rccl",
                    help="rccl: the PRODUCT's multi-GPU path -- nb_comm_init_rank + nb_sharded_step_* of the C-ABI (csrc/nbody_comm.hip): the "
                         "position all-gather issued as its G-1 tiles, RCCL send/recv pairs on the communicator's side "
                         "stream, the kernel of tile k waiting only on tile k's event; torch.distributed then only does rendezvous, barrier and the "
                         "time reduction (gloo).  torch: the same tile schedule re-implemented over torch.distributed "
                         "(cuda-nbody_amd/sharded.py, batch_isend_irecv) -- A/B and first fallback; allgather: one all_gather_into_tensor per "
                         "step (sharded.py); staged: no RCCL at all -- gloo, the slices gathered through host memory, each rank on its OWN GPU "
                         "(the last resort: a real N-GPU number of the kernels with a slow exchange); "
                         "host: gloo + host-staged all-gather, so that several ranks can share ONE GPU (functional rehearsal of the "
                         "N-rank code path on a one-GPU box; RCCL refuses two ranks per device); host-tiles: the same rehearsal with the TILE "
                         "schedule (gloo send/recv rounds staged through host memory).  host* is never a performance number.")
    ap.add_argument("--layout", choices=["pairwise", "one-sided"], default="pairwise",
                    help="FAST on one GPU: pairwise = nb_integrate_ws_* with a caller-owned workspace (every pair of bodies evaluated once and "
                         "applied to both, csrc/nbody_pair.hip); one-sided = nb_integrate_* (every directed interaction, as the reference kernel)")
    ap.add_argument("--dump-state", type=str, default="", help="rank 0 writes its initial and final positions (.npz) here (tests)")
    ap.add_argument("--no-configs", action="store_true", help="skip the extra BASELINE configs timed after the headline measurement (N=1)")
    ap.add_argument("--launch-timeout", type=float, default=600.0,
                    help="plain `bench.py --gpus N`: seconds the launcher waits for the N ranks before ending them (see self_launch)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="REHEARSAL of the N-rank C-ABI path on a one-GPU box: every rank uses device 0 and RCCL is replaced by the test double "
                         "(NBODY_RCCL_LIB=tests/fake_rccl/libfake_rccl.so with FAKE_RCCL_IPC=1, set here when absent).  Never a performance number.")
    ap.add_argument("--no-diagnostics", action="store_true", help="N>1: skip the A/B timings and BASELINE configs[3] after the timed region")
    ap.add_argument("--diagnostics-timeout", type=float, default=240.0, help="N>1: seconds the post-headline measurements may take before the line is printed without them")
    return ap.parse_args()


def self_launch(n_ranks: int, explicit_exchange: bool, limit_s: float) -> int:
    """Run this same command line as `n_ranks` ranks under torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1) as a CHILD process group and relay its output.  The >1-GPU path has not run on hardware yet, so the launcher
    carries one safety net: when the ranks fail or go `limit_s` seconds without finishing before rank 0 printed its line,
    and the exchange was not chosen on the command line, their process group is ended (by its exact id) and the job is
    started again with a plainer exchange: `--exchange torch` (the tile schedule over torch.distributed), `--exchange allgather`
    (one all-gather per step), then `--exchange staged` (gloo through host memory, no RCCL) -- at most those three further
    attempts, each marked "exchange_fallback": true in its JSON line."""
    import signal
    import socket
    import subprocess
    import threading

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")

    def attempt(extra):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + extra
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
        seen = {"metric": False}

        def relay():
            for line in child.stdout:
                if line.startswith("{") and '"metric"' in line:
                    seen["metric"] = True
                sys.stdout.write(line)
                sys.stdout.flush()

        reader = threading.Thread(target=relay, daemon=True)
        reader.start()
        try:
            rc = child.wait(timeout=limit_s)
        except subprocess.TimeoutExpired:
            print(f"[bench] ranks still running after {limit_s:.0f} s: ending process group {child.pid}", file=sys.stderr, flush=True)
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(child.pid, sig)  # the group this call created (start_new_session), nothing else
                except ProcessLookupError:
                    break
                try:
                    child.wait(timeout=20)
                    break
                except subprocess.TimeoutExpired:
                    continue
            rc = child.returncode if child.returncode is not None else -9
        reader.join(timeout=10)
        return rc, seen["metric"]

    rc, reported = attempt([])
    # plainer and plainer: the tile schedule over torch.distributed, one all-gather per step, then no RCCL at all.  A line
    # produced by a retry says so at its top level ("exchange_fallback": true), not only in config.exchange.
    env["NBODY_BENCH_EXCHANGE_FALLBACK"] = "1"
    for fallback in ("torch", "allgather", "staged"):
        if rc == 0 or reported or explicit_exchange:
            break
        print(f"[bench] the {n_ranks}-rank run ended with status {rc} before reporting; one more attempt with --exchange {fallback}", file=sys.stderr, flush=True)
        rc, reported = attempt(["--exchange", fallback])
    return rc


def make_bodies(n: int, dtype):
    """The bodies a fresh `nbody --numbodies=N [--fp64]` process starts from, drawn by the PRODUCT's randomise_bodies
    (libnbody_host.so, pinned bit-for-bit to the reference's own code in tests/test_host_cpp.py): SHELL configuration,
    third segment of the unseeded rand() stream (fp32 reset, fp64 reset with demo_params[0] scales, then the active
    precision with the N-scaled params).  "synthetic random bodies" of BASELINE.json."""
    host = ctypes.CDLL(os.path.join(ROOT, "cuda-nbody_amd", "libnbody_host.so"))
    f32p, f64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
    host.nbh_srand.argtypes = [ctypes.c_uint]
    host.nbh_randomise_f32.argtypes = [ctypes.c_int, f32p, f32p, ctypes.c_size_t, ctypes.c_float, ctypes.c_float]
    host.nbh_randomise_f64.argtypes = [ctypes.c_int, f64p, f64p, ctypes.c_size_t, ctypes.c_float, ctypes.c_float]
    host.nbh_scale_params_for.argtypes = [ctypes.c_size_t, f32p, f32p]
    shell = 1  # NBodyConfig::NBODY_CONFIG_SHELL

    def draw(T, cluster, velocity):
        pos, vel = np.zeros(4 * n, T), np.zeros(4 * n, T)
        if T == np.float32:
            host.nbh_randomise_f32(shell, pos.ctypes.data_as(f32p), vel.ctypes.data_as(f32p), n, cluster, velocity)
        else:
            host.nbh_randomise_f64(shell, pos.ctypes.data_as(f64p), vel.ctypes.data_as(f64p), n, cluster, velocity)
        return pos, vel

    host.nbh_srand(1)
    draw(np.float32, 1.54, 8.0)
    draw(np.float64, 1.54, 8.0)
    c, v = ctypes.c_float(1.54), ctypes.c_float(8.0)
    host.nbh_scale_params_for(n, ctypes.byref(c), ctypes.byref(v))
    return draw(dtype, c.value, v.value)


# What the instruction mix of the production loop can do at best on this chip: the unit-mass inner loop run in isolation
# (tools/loop_microbench_gen.py, profiles/round2_loop_microbench.txt) sustains one packed interaction pair per 61.5 SIMD cycles
# with 3-4 runnable waves (11 v_pk_* at ~4 cycles + 2 v_rsq_f32 at ~8.3); at the nominal 2.4 GHz that is
# 1024 SIMDs x 128 interactions / 61.5 cycles = 5.115e12 interactions/s = 65.0 % of the 157.3 TFLOP/s "20 flop" roofline.
FP32_ISSUE_CEILING_INTERACTIONS_PER_S = 1024 * 128 * 2.4e9 / 61.5
# fp64 (profiles/round2_fp64_issue_probes.txt): 14 add/mul/fma at 4.41 cycles + one v_rsq_f64 at 16.3 = 78 cycles per interaction and wave
FP64_ISSUE_CEILING_INTERACTIONS_PER_S = 1024 * 64 * 2.4e9 / 78.0


def pair_evaluations(pair) -> float:
    """pair evaluations per step of the pairwise layout: NB x (NB/2 + 1) block pairs of (64 I)^2"""
    return float(pair.blocks) * (pair.blocks // 2 + 1) * pair.block_bodies * pair.block_bodies


def fractions(n, fp64, layout, ms, pair=None):
    """(frac, executed_frac) of the vector-FMA peak.  `frac`: ALGORITHMIC flop of the reference convention -- 20 (30) per directed
    interaction, N^2 interactions (compute.cpp:16-18) -- over the time; it can pass 1 for the pairwise layout, which
    evaluates each pair once.  `executed_frac`: the flop the kernels really issue -- 24 (36) per pair evaluation for the pairwise
    layout, the algorithmic count for the one-sided and STRICT kernels (they do evaluate every directed interaction) -- i.e. how
    busy the FMA pipes are; never above 1."""
    flops, peak = (30, FP64_VECTOR_PEAK_TFLOPS) if fp64 else (20, FP32_VECTOR_PEAK_TFLOPS)
    frac = flops * float(n) * n / (ms * 1e-3) / (peak * 1e12)
    if layout != "pairwise" or pair is None:
        return round(frac, 4), round(frac, 4)
    return round(frac, 4), round((36 if fp64 else 24) * pair_evaluations(pair) / (ms * 1e-3) / (peak * 1e12), 4)


def other_configs(pkg, lib, headline):
    """BASELINE.json configs besides the headline one, plus STRICT (the parity-exact mode) and, for FAST, both layouts
    (pairwise = nb_integrate_ws_* with a workspace, one-sided = nb_integrate_*), each as
    {workload, bodies, dtype, mode, layout, steps, ms_per_step, frac, executed_frac}: 1 warm-up step, then K steps between two
    HIP events on the launch stream (the reference's GPU protocol, compute_cuda.cpp:183-195); the fractions: see fractions()."""
    cases = [
        ("configs[1]", 65536, False, "fast", 200),
        ("configs[2]", 262144, False, "fast", 20),
        ("configs[4]", 262144, True, "fast", 5),
        ("configs[3]'s system on ONE GPU", 1048576, False, "fast", 3),
        ("STRICT = the CPU path's bits", 262144, False, "strict", 5),
        ("STRICT", 262144, True, "strict", 3),
        ("configs[0]'s system on the GPU", 1024, False, "fast", 100),
        ("configs[0]'s system on the GPU", 1024, False, "strict", 100),
        ("small system", 16384, False, "fast", 200),
        # beyond BASELINE's sizes: one tournament would want 206 GB of reaction slots; the tournament cut into slices inside 16 GB
        ("4 Mi bodies, workspace capped at 16 GB", 4194304, False, "fast", 2),
    ]
    out = []
    for what, n, fp64, mode_name, steps in cases:
        dtype = np.float64 if fp64 else np.float32
        mode = pkg.NB_MODE_FAST if mode_name == "fast" else pkg.NB_MODE_STRICT
        layouts = ["one-sided"] if mode_name == "fast" else ["strict"]
        cap = (16 << 30) if n > 1048576 else None
        if mode_name == "fast" and pkg.workspace_bytes(n, dtype, mode, cap):
            layouts.insert(0, "pairwise")
        if cap is not None:
            layouts = layouts[:1]  # (the one-sided kernel at this size: 3.6 s per step, nothing new)
        pos0 = vel0 = None
        for layout in layouts:
            if (n, fp64, mode_name, layout) == headline:
                continue
            if pos0 is None:
                pos0, vel0 = make_bodies(n, dtype)
            system = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0, vel0, mode=mode, workspace=(layout == "pairwise"), workspace_cap=cap)
            dt = dtype(np.float32(0.016))
            system_bytes = system._workspace_bytes
            system.update(dt)
            e0, e1 = pkg.Event(), pkg.Event()
            system.synchronize()
            e0.record(None)
            for _ in range(steps):
                system.update(dt)
            e1.record(None)
            e1.synchronize()
            ms = e0.elapsed_ms(e1) / steps
            system.free()
            frac, executed = fractions(n, fp64, layout, ms, pkg.pair_plan(n, dtype) if layout == "pairwise" else None)
            # (kept short: the whole line should stay well under what a log tail holds; interactions/s = bodies^2 / ms_per_step)
            out.append({"workload": what, "bodies": n, "dtype": "f64" if fp64 else "f32", "mode": mode_name, "layout": layout, "steps": steps,
                        "ms_per_step": float(f"{ms:.5g}"), "frac": frac, "executed_frac": executed})
            if cap is not None:
                out[-1]["workspace_bytes"] = system_bytes
    return out


def rank_projection(pkg, lib, n, dtype, dt, damping, single_ms):
    """ms of ONE rank's kernels of a 2 / 4 / 8-rank pairwise step on this GPU (rank G/2; diagonal + G/2 rectangles + folds + finish),
    no exchange -- tools/pair_rank_probe.py has the longer version."""
    f32 = np.dtype(dtype) == np.float32
    emulate = lib.nb_emulate_pair_rank_f32 if f32 else lib.nb_emulate_pair_rank_f64
    pos0, vel0 = make_bodies(n, dtype)
    bufs = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(3)]
    bufs[0].upload(pos0), bufs[2].upload(vel0)
    out = {"what": "PROJECTION, not a multi-GPU measurement: kernel ms of ONE rank of a G-rank pairwise step, alone on one GPU, no exchange",
           "single_gpu_ms_per_step": float(f"{single_ms:.5g}"), "ranks": {}}
    for G in (2, 4, 8):
        need = ctypes.c_size_t(0)
        if emulate(None, None, None, None, ctypes.byref(need), n, G, 0, dt, damping, None) != 0:
            continue
        work = pkg.DeviceBuffer(need.value)

        def one_step():
            pkg.check(emulate(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, work.ptr, ctypes.byref(need), n, G, G // 2, dt, damping, None), "nb_emulate_pair_rank")

        one_step()
        pkg.check(lib.nb_device_synchronize())
        e0, e1 = pkg.Event(), pkg.Event()
        e0.record(None)
        for _ in range(10):
            one_step()
        e1.record(None)
        e1.synchronize()
        ms = e0.elapsed_ms(e1) / 10
        out["ranks"][str(G)] = {"kernel_ms": float(f"{ms:.5g}"), "speedup_excl_exchange": round(single_ms / ms, 2)}
        work.free()
    for b in bufs:
        b.free()
    return out


def pair_kernel_split(pkg, lib, step, stream, reps=10):
    """Average duration of the two kernels of the pairwise step, each on its own: pair_forces (the dominant kernel) and pair_finish,
    from HIP events on the launch stream -- one before the step, one the library records BETWEEN the two launches
    (nb_set_pair_probe_event, tuning header), one after.  Taken after the timed region."""
    before, between, after = pkg.Event(), pkg.Event(), pkg.Event()
    forces = finish = 0.0
    pkg.check(lib.nb_set_pair_probe_event(between.h), "nb_set_pair_probe_event")
    try:
        for _ in range(reps):
            before.record(stream)
            step()
            after.record(stream)
            after.synchronize()
            forces += before.elapsed_ms(between)
            finish += between.elapsed_ms(after)
    finally:
        pkg.check(lib.nb_set_pair_probe_event(None), "nb_set_pair_probe_event")
    return forces / reps, finish / reps


def cpu_baseline(n, dtype, pos0, vel0, sample_bodies):
    """The CPU path (oracle/: a port of BodySystemCPU<T>::update; test infrastructure, loaded here only) timed on this host:
    a bounded sample of the headline workload, and BASELINE configs[0] exactly as stated -- 1 024 bodies, fp32, 100 steps, no
    warm-up, steady clock around the loop (compute_cpu.cpp:72-88)."""
    O = entry.load_oracle()
    orc1 = O.Oracle()
    pos_h, vel_h = orc1.startup_state(n, dtype)
    # the workload above came from the product's randomise_bodies; the checker's must be the same bytes
    assert pos_h.tobytes() == pos0.tobytes() and vel_h.tobytes() == vel0.tobytes(), "product and oracle start-up bodies differ"
    sample = sample_bodies or max(8, min(n, int(2.0e10 // n) // 8 * 8))
    base = {}
    # OpenMP leg: the reference's fp32 loop forks INSIDE the j loop (bodysystemcpu.cpp:156-168), i.e. one fork/join per body j --
    # it is slow by construction, so it gets a smaller sample and at most the box's CPU share (16 threads per GPU).
    for key, omp, smp in (("one_thread", False, sample), ("openmp", True, max(8, sample // 16 // 8 * 8))):
        orc = O.Oracle(openmp=omp)
        if omp:
            orc.set_num_threads(min(16, os.cpu_count() or 1))
        ms = orc.benchmark_partial(pos_h, smp)
        base[key] = {"value": smp * float(n) / (ms * 1e-3), "cores": orc.num_threads() if omp else 1, "ms": ms, "sample_bodies_i": smp}
    p0, v0 = orc1.startup_state(1024, np.float32)
    ms0 = orc1.benchmark(p0, v0, np.float32(0.016), 100)
    return {
        "value": base["one_thread"]["value"],
        "unit": "interactions/s",
        "cores": 1,
        "kind": "port",
        "sample": f"force pass of BodySystemCPU::update (oracle/ port) for the first {sample} bodies i against all {n} bodies j = {sample * n:.3g} "
                  f"interactions; 1 thread is how the reference ships",
        "openmp": base["openmp"],
        "config0": {"what": "BASELINE configs[0]: 1024 bodies, fp32, 100 steps of the CPU path, no warm-up (compute_cpu.cpp:72-88), 1 thread",
                    "ms_total": float(f"{ms0:.5g}"), "interactions_per_s": 1024.0 * 1024.0 * 100 / (ms0 * 1e-3), "gflops": 20 * 1024.0 * 1024.0 * 100 / (ms0 * 1e-3) * 1e-9},
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_one_gpu:
        # every rank on device 0, RCCL replaced by the cross-process test double: set before anything resolves RCCL (once per process)
        os.environ.setdefault("NBODY_RCCL_LIB", os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so"))
        os.environ.setdefault("FAKE_RCCL_IPC", "1")
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # plain `python3 bench.py --gpus N`: start the N ranks ourselves, BEFORE anything in this process touches the GPU
        # (no torch import, no HIP call so far), relay their output and exit with their status.  Children, never exec.
        raise SystemExit(self_launch(args.gpus, any(a.startswith("--exchange") for a in sys.argv[1:]) or args.rehearse_one_gpu, args.launch_timeout))
    if args.gpus != world:
        args.gpus = world

    dtype = np.float64 if args.fp64 else np.float32
    n = args.bodies
    flops_per = 30 if args.fp64 else 20
    peak = FP64_VECTOR_PEAK_TFLOPS if args.fp64 else FP32_VECTOR_PEAK_TFLOPS

    # torch first: it bundles its own libamdhip64.so.7; loading libnbody_hip.so afterwards binds to that one
    # copy by SONAME.  The other order puts two HIP runtimes in the process and the second sees no device.
    import torch

    # The bodies come from the process-global libc rand() stream (the reference's randomise_bodies does), so they are drawn NOW:
    # while this process has a single thread.  Later, torch.distributed's store and gloo threads are running and may draw from
    # the same stream in between (seen once in round 3: two runs of one command that differed in a few bodies).
    pos0, vel0 = make_bodies(n, dtype)
    big = None  # BASELINE configs[3]'s system, for the N > 1 line (drawn now for the same reason)
    if world > 1 and not args.no_diagnostics and args.exchange == "rccl" and args.mode == "fast" and not args.fp64 and CONFIG3_BODIES % world == 0:
        big = make_bodies(CONFIG3_BODIES if not args.rehearse_one_gpu else 65536, np.float32)

    pkg = entry.load_package()
    lib = pkg.lib()
    mode = pkg.NB_MODE_FAST if args.mode == "fast" else pkg.NB_MODE_STRICT
    if args.plan:
        pkg.set_plan_override(*[int(x) for x in args.plan.split(",")])

    if args.exchange.startswith("host") or args.rehearse_one_gpu:
        local_rank = 0  # every rank on the one GPU
    torch.cuda.set_device(local_rank)
    pkg.check(lib.nb_set_device(local_rank), "nb_set_device")
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the distributed path is taken even for one rank, so the 1-GPU line of a
    # scaling series is produced by the same code as the N-GPU lines
    distributed = world > 1 or "RANK" in os.environ
    exchange_fallback = os.environ.get("NBODY_BENCH_EXCHANGE_FALLBACK") == "1"  # set by self_launch for its retries
    rccl_group = None
    if distributed:
        import torch.distributed as dist

        # The default process group is gloo, always: rendezvous, barriers, the unique-id broadcast, collective decisions and
        # the time reduction.  The data path is either the product's own RCCL communicator behind the C-ABI (--exchange rccl)
        # or, for the torch.distributed re-implementation (--exchange torch|allgather), a separate "nccl" group.
        dist.init_process_group("gloo")

        def torch_rccl_group():
            # RCCL's kernels compete with the force kernel for CUs (a 1024-thread workgroup fills a CU's VGPRs): a
            # high-priority stream lets their few workgroups dispatch first, so the exchange overlaps the own-slice chunk
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                return dist.new_group(backend="nccl", pg_options=opts)
            except (AttributeError, TypeError):
                return dist.new_group(backend="nccl")

        if args.exchange in ("torch", "allgather"):
            rccl_group = torch_rccl_group()
    info = pkg.device_info(local_rank)

    params = pkg.NBodyParams()
    dt = dtype(np.float32(params.time_step))
    damping = dtype(np.float32(params.damping))
    soft = dtype(np.float32(params.softening))
    if args.fp64:
        pkg.check(lib.nb_set_softening_sq_f64(float(soft * soft)))
    else:
        pkg.check(lib.nb_set_softening_sq_f32(np.float32(soft * soft)))
    shard_fn = lib.nb_integrate_shard_f64 if args.fp64 else lib.nb_integrate_shard_f32

    tdtype = torch.float64 if args.fp64 else torch.float32
    pos_t = torch.from_numpy(pos0.reshape(n, 4)).to(dev, tdtype)
    vel_t = torch.from_numpy(vel0.reshape(n, 4)).to(dev, tdtype)
    stream = torch.cuda.current_stream()
    stream_ptr = ctypes.c_void_p(stream.cuda_stream)

    def lend(nbytes):
        """scratch memory of the pairwise layout: caller-owned (a torch tensor here), sized by the library, contents irrelevant.
        An allocation failure is not an error: the rank then lends nothing (and, several ranks: nb_comm_set_workspace lets every
        rank know, so that all of them step one-sidedly)."""
        if not nbytes:
            return None
        try:
            return torch.empty(nbytes, dtype=torch.uint8, device=dev)
        except RuntimeError as exc:  # (torch.OutOfMemoryError is one)
            print(f"[bench rank {rank}] no memory for a {nbytes}-byte workspace ({exc!r}): stepping without one", file=sys.stderr, flush=True)
            return None

    want_pairwise = args.layout == "pairwise" and not args.plan
    work_t = lend(pkg.workspace_bytes(n, dtype, mode)) if (want_pairwise and world == 1) else None
    pairwise = work_t is not None  # (several ranks: decided below by the communicator)
    work_bytes = work_t.numel() if work_t is not None else 0
    ws_fn = lib.nb_integrate_ws_f64 if args.fp64 else lib.nb_integrate_ws_f32

    def launch(new_pos, old_pos, vel, acc, i0, ni, j0, nj, flags):
        pkg.check(shard_fn(new_pos.data_ptr(), old_pos.data_ptr(), vel.data_ptr(), acc.data_ptr(), i0, ni, j0, nj, flags,
                           dt, damping, 256, mode, stream_ptr), "nb_integrate_shard")

    capi_rank = None  # --exchange rccl: this rank of the product's sharded system (nb_comm_init_rank + nb_sharded_step_*)
    system = None     # every other exchange: cuda-nbody_amd/sharded.py over torch.distributed
    sharded = entry.load_package_module("sharded") if distributed else None
    if distributed:
        def everyone(ok: bool) -> bool:
            """collective decision over gloo: true only if `ok` on EVERY rank (a rank deciding on its own would leave the
            others inside mismatched collectives)"""
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        if args.exchange == "rccl":
            bufs = [pos_t, pos_t.clone()]
            acc_t = torch.zeros_like(pos_t)
            problem = None
            try:
                ids = [pkg.comm_unique_id() if (rank == 0 and world > 1) else None]
            except pkg.NBodyHipError as exc:
                ids, problem = [None], exc
            dist.broadcast_object_list(ids, src=0)
            if world > 1 and ids[0] is None:
                problem = problem or RuntimeError("rank 0 could not create the RCCL unique id")
            else:
                try:
                    capi_rank = pkg.ShardedRank(ids[0], world, rank, [b.data_ptr() for b in bufs], vel_t.data_ptr(), acc_t.data_ptr(), n, dtype, mode, 256, stream_ptr)
                except pkg.NBodyHipError as exc:
                    problem = exc
            if everyone(problem is None):
                # The communicator is up on every rank.  Lend it the workspace: with several ranks nb_comm_set_workspace is a
                # COLLECTIVE (every rank calls it, with nothing if it has nothing), after which the layout of a step -- pairwise
                # across the ranks or one-sided tiles -- is the communicator's, the same on every rank.
                try:
                    if want_pairwise:
                        work_t = lend(capi_rank.workspace_bytes())
                        work_bytes = work_t.numel() if work_t is not None else 0
                    capi_rank.set_workspace(work_t.data_ptr() if work_t is not None else None, work_bytes)
                    pairwise = capi_rank.pairwise()
                    # bring the communicator up (channels, first-call set-up) outside any timed step, whatever --warmup says;
                    # every rank holds identical positions at this point, so exchanging them changes nothing
                    capi_rank.exchange_once(0)
                    torch.cuda.synchronize()
                except pkg.NBodyHipError as exc:
                    problem = exc
            if not everyone(problem is None):
                print(f"[bench rank {rank}] the C-ABI RCCL path could not be brought up ({problem!r} on this rank); "
                      "ALL ranks fall back to the tile schedule over torch.distributed", file=sys.stderr, flush=True)
                if capi_rank is not None:
                    capi_rank.destroy()
                capi_rank, exchange_fallback, pairwise = None, True, False
                args.exchange = "torch"
                rccl_group = torch_rccl_group()

        if capi_rank is not None:
            def step():
                capi_rank.update(dt, damping)

            finish = capi_rank.finish
        else:
            host_gather = None
            if args.exchange in ("host", "staged"):
                class _Done:
                    def wait(self):
                        pass

                def host_gather(full, own):
                    staged = torch.empty(full.shape, dtype=full.dtype)
                    dist.all_gather_into_tensor(staged, own.cpu())
                    full.copy_(staged)
                    return _Done()

            form = "tiles" if args.exchange in ("torch", "host-tiles") else "allgather"
            system = sharded.ShardedBodySystem(pos_t, vel_t, launch, ordered=(mode == pkg.NB_MODE_STRICT), group=rccl_group, gather=host_gather, exchange=form)
            step = system.update
            finish = system.finish
            if world > 1:
                # bring the exchange up outside any timed step; if the tile form fails on ANY rank, ALL ranks take the single collective
                problem = None
                try:
                    system.exchange_once(system.pos[0])
                    torch.cuda.synchronize()
                except RuntimeError as exc:
                    problem = exc
                if not everyone(problem is None):
                    if system.exchange != "tiles":
                        raise problem or RuntimeError("another rank failed to bring the exchange up")
                    print(f"[bench rank {rank}] tile exchange failed at bring-up ({problem!r} on this rank); ALL ranks fall back to one all-gather per step",
                          file=sys.stderr, flush=True)
                    exchange_fallback = True
                    system = sharded.ShardedBodySystem(pos_t, vel_t, launch, ordered=(mode == pkg.NB_MODE_STRICT), group=rccl_group, exchange="allgather")
                    step, finish = system.update, system.finish
                    system.exchange_once(system.pos[0])
                    torch.cuda.synchronize()
    else:
        bufs = [pos_t, pos_t.clone()]
        acc_t = torch.zeros_like(pos_t)
        state = {"read": 0}

        def step():
            r = state["read"]
            if pairwise:  # (one call = the forces kernel + the kernel that adds the reaction slots and integrates)
                pkg.check(ws_fn(bufs[1 - r].data_ptr(), bufs[r].data_ptr(), vel_t.data_ptr(), dt, damping, n, 256, mode, work_t.data_ptr(), work_bytes, stream_ptr), "nb_integrate_ws")
            else:
                launch(bufs[1 - r], bufs[r], vel_t, acc_t, 0, n, 0, n, pkg.NB_SHARD_FINALIZE)
            state["read"] = 1 - r

        def finish():
            pass

    def fence():
        finish()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()  # gloo: a host barrier between two device synchronisations
            torch.cuda.synchronize()

    # ------------------------------------------------------------------------------------------------ the timed region
    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = pkg.Event(), pkg.Event()
    t0 = time.perf_counter()
    ev0.record(stream_ptr)
    for _ in range(args.steps):
        step()
    finish()
    ev1.record(stream_ptr)
    fence()
    elapsed = time.perf_counter() - t0
    ev1.synchronize()
    stream_ms_per_step = ev0.elapsed_ms(ev1) / args.steps  # HIP events on the launch stream: this rank's step, kernels only at N = 1

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64)  # gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if args.dump_state and rank == 0:  # (the state after exactly warmup + steps steps: before anything below steps the system further)
        torch.cuda.synchronize()
        final = capi_rank.pos[capi_rank.read] if capi_rank is not None else (system.positions().data_ptr() if system is not None else bufs[state["read"]].data_ptr())
        host = np.zeros(4 * n, dtype)
        pkg.check(lib.nb_d2h(host.ctypes.data_as(ctypes.c_void_p), final, host.nbytes, None), "nb_d2h")
        np.savez(args.dump_state, final=host, initial=pos0)  # (the bodies the run started from, too: a test can tell a different start from a different step)

    # ------------------------------------------------------------------------------------------------ after it: the line
    line = None
    if rank == 0:
        value = float(n) * float(n) * args.steps / elapsed
        plan = pkg.plan(n // world, n, dtype)
        pair = pkg.pair_plan(n, dtype) if (pairwise and world == 1) else None
        layout_name = "pairwise" if pairwise else ("one-sided" if args.mode == "fast" else "strict")
        # the dominant kernel's own duration: the pairwise step is two kernels (pair_forces, pair_finish), timed separately AFTER
        # the timed region with an event the library records between them; every other single-GPU step is one kernel
        forces_ms = finish_ms = None
        if pair is not None and not distributed:
            forces_ms, finish_ms = pair_kernel_split(pkg, lib, step, stream_ptr)
        dominant_ms = forces_ms if forces_ms is not None else stream_ms_per_step
        algorithmic_flops = flops_per * float(n) * float(n) / world  # per launch of the dominant kernel(s) of one rank's step
        achieved_tflops = algorithmic_flops / (dominant_ms * 1e-3) / 1e12
        # HBM traffic cannot be counted from inside this process: it comes from the separate rocprofv3 --pmc passes
        # of this same command (tools/profile.sh -> tools/summarize_prof.py), committed under profiles/.
        traffic, traffic_src = None, None
        plan_now = {"bodies_per_lane": plan.bodies_per_lane, "lane_groups": plan.lanes_per_body, "lds_tile_bodies": plan.tile_bodies,
                    "grid": plan.grid_blocks, "lds_bytes": plan.lds_bytes}
        if pair is not None:
            plan_now = {"layout": "pairwise", "bodies_per_lane": pair.bodies_per_lane, "waves_per_block": pair.waves_per_block, "workgroups_per_block": pair.splits,
                        "blocks": pair.blocks, "block_bodies": pair.block_bodies, "reaction_slots": pair.reaction_slots, "grid": pair.grid_blocks,
                        "lds_bytes": pair.lds_bytes, "workspace_bytes": pair.workspace_bytes}
        if world == 1 and args.mode == "fast":
            import glob

            tag = f"n{n}_{'f64' if args.fp64 else 'f32'}" + ("_pairwise" if pairwise else "")
            found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"round*_{tag}_pmc_summary.json")))
            if found:
                with open(found[-1]) as fh:
                    summary = json.load(fh)
                if summary.get("kernel_plan") == plan_now:  # counters of another geometry say nothing about this run
                    traffic = summary["derived"].get("hbm_bytes_per_launch")
                    traffic_src = os.path.relpath(found[-1], ROOT)
                else:
                    traffic_src = f"{os.path.relpath(found[-1], ROOT)} was taken with another launch plan: re-run tools/profile.sh"
        executed = None
        if pair is not None:
            evals = pair_evaluations(pair)
            per = 36 if args.fp64 else 24
            executed = {"pair_evaluations_per_launch": evals, "flops_per_pair_evaluation": per,
                        "tflops": per * evals / (dominant_ms * 1e-3) / 1e12, "frac": per * evals / (dominant_ms * 1e-3) / 1e12 / peak}
        roofline = {
            "bound": "valu_fp32_fma" if not args.fp64 else "valu_fp64_fma",
            "kernel": "pair_forces" if pair is not None else ("one rank's step (all its kernels and waits)" if distributed and world > 1 else "the step's one kernel"),
            # ALGORITHMIC flop per launch -- 20 (30) x N^2, the reference's convention (compute.cpp:16-18) -- over the
            # dominant kernel's average duration.  For the pairwise layout this is NOT a utilisation figure (each pair is evaluated
            # once, so it can pass 1): `executed` holds the flop the kernel really issues and their fraction of the same peak.
            "achieved": achieved_tflops,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": achieved_tflops / peak,
            "frac_counts": "algorithmic flop (reference convention) / peak; executed.frac = flop issued / peak",
            "executed": executed,
            "step_frac": flops_per * value / world / 1e12 / peak,  # the same count over the WHOLE step as the driver times it (value)
            "traffic": traffic,  # HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes (tools/profile.sh)
            "traffic_source": traffic_src,
            "kernel_ms": dominant_ms,
            "pair_forces_ms": forces_ms,
            "pair_finish_ms": finish_ms,
            "stream_ms_per_step": stream_ms_per_step,
            # achieved / what the one-sided instruction mix can issue at best on the chip (see FP32_ISSUE_CEILING_...); STRICT and the
            # pairwise layout execute other instruction mixes
            "issue_ceiling_frac": None if (args.mode != "fast" or pairwise) else value / world / (FP64_ISSUE_CEILING_INTERACTIONS_PER_S if args.fp64 else FP32_ISSUE_CEILING_INTERACTIONS_PER_S),
            "algorithmic_flops_per_launch": algorithmic_flops,
            # positions + velocities in and out = 64 (128) bytes per body; what the pairwise layout moves through its
            # workspace on top of that -- written once by pair_forces, read once by pair_finish -- is stated separately
            "algorithmic_hbm_bytes_per_launch": (128 if args.fp64 else 64) * (n // world),
            "workspace_rw_bytes_per_step": (2 * pair.workspace_bytes) if pair is not None else None,
        }
        line = {
            "metric": "body-body interactions/s, all-pairs N-body step (reference convention N^2 per step)",
            "value": value,
            "unit": "interactions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            # true when this line was NOT produced by the exchange asked for (the launcher's retries, or a collective in-rank
            # fallback at bring-up): such a value must not pass for a number of the C-ABI RCCL path
            "exchange_fallback": bool(exchange_fallback),
            "vs_baseline": None,
            "dtype": "f64" if args.fp64 else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n} bodies, SHELL start-up configuration (reference rand() stream), "
                            f"{'fp64' if args.fp64 else 'fp32'}, dt 0.016, softening 0.1, damping 1.0, mode {args.mode}",
                "bodies": n,
                "bodies_per_gpu": n // world,
                "exchange": "none" if world == 1 else (
                    "REHEARSAL: gloo, host-staged gather, ranks share one GPU" if args.exchange == "host" else
                    "FALLBACK (no RCCL): gloo all-gather of the slices through host memory, one GPU per rank" if args.exchange == "staged" else
                    "REHEARSAL: gloo send/recv rounds (tile schedule) staged through host memory, ranks share one GPU" if args.exchange == "host-tiles" else
                    (("REHEARSAL on ONE GPU with the RCCL test double (never a performance number): " if args.rehearse_one_gpu else "") +
                     ("C-ABI (nb_comm_init_rank + nb_sharded_step_*, csrc/nbody_comm.hip), PAIRWISE across the ranks: each rank evaluates its own slice and "
                      "the rectangles against ranks r+1 .. r+G/2 once per pair and sends the reaction sums (N/G x 12 B per partner) to their owners; "
                      "positions all-gathered as G-1 RCCL send/recv tiles on the communicator's high-priority stream" if pairwise else
                      "C-ABI (nb_comm_init_rank + nb_sharded_step_*, csrc/nbody_comm.hip): RCCL all-gather of the new positions per step, issued as "
                      "G-1 position tiles (grouped ncclSend/ncclRecv rounds on the communicator's high-priority stream); the kernel of tile k waits "
                      "only on round k, the own-slice chunk runs first")) if capi_rank is not None else
                    "torch.distributed re-implementation (sharded.py) of the tile schedule: batch_isend_irecv rounds on RCCL's stream" if system.exchange == "tiles" else
                    "torch.distributed (sharded.py): RCCL all_gather_into_tensor of the new positions per step, overlapped with the own-slice j chunk"),
                "exchange_grouping": None if (capi_rank is None or world == 1) else ("one RCCL group for all G-1 position rounds" if capi_rank.exchange_grouping() else "one RCCL group per position round"),
                "layout": "pairwise (every pair of bodies evaluated once, reaction sums through a caller-owned workspace)" if pairwise else
                          "one-sided (every directed interaction evaluated, as bodysystemcuda.cu:125-146 does)",
                "step_entry_point": ("nb_integrate_ws_*" if pairwise else "nb_integrate_shard_*") if not distributed else ("nb_sharded_step_*" if capi_rank is not None else "sharded.py -> nb_integrate_shard_*"),
                "kernel_plan": plan_now,
                "device": info.name.decode(),
                "arch": info.arch.decode(),
            },
            "gflops": value * flops_per * 1e-9,
            "flops_per_interaction": flops_per,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(n, dtype, pos0, vel0, args.cpu_sample_bodies)
        if world == 1 and not args.no_configs and not args.plan:
            # The other BASELINE configs and the parity-exact mode, timed AFTER the headline measurement (never inside it)
            # so that one driver-run line carries them all.  A failure here costs only this list, never the headline.
            try:
                line["configs"] = other_configs(pkg, lib, (n, args.fp64, args.mode, layout_name))
            except Exception as exc:  # noqa: BLE001
                line["configs"] = [{"error": repr(exc)}]
            if pairwise:
                # No multi-GPU node has run this yet.  What CAN be measured on one GPU is the compute side: exactly the kernels one rank
                # of a G-rank pairwise step launches (nb_emulate_pair_rank_*), alone on the chip, no exchange.  A projection, labelled so.
                try:
                    line["multi_gpu_kernel_projection"] = rank_projection(pkg, lib, n, dtype, dt, damping, elapsed / args.steps * 1e3)
                except Exception as exc:  # noqa: BLE001
                    line["multi_gpu_kernel_projection"] = {"error": repr(exc)}

    # ------------------------------------------------------------------------------------------------ N > 1: what explains the number
    # Everything below runs AFTER the timed region and never touches `value`.  It is collective work on a path that has not met
    # real multi-GPU hardware yet, so it runs under a watchdog: if it stalls, rank 0 prints the line with what it has and every
    # rank leaves.  One JSON line either way.
    printed = threading.Lock()

    def emit(extra=None):
        if rank == 0 and line is not None and printed.acquire(blocking=False):
            if extra:
                line.update(extra)
            print(json.dumps(line), flush=True)

    if world > 1:
        def stalled():
            emit({"diagnostics_incomplete": f"the post-headline measurements did not finish within {args.diagnostics_timeout:.0f} s"})
            sys.stderr.write(f"[bench rank {rank}] post-headline measurements stalled: leaving\n")
            sys.stderr.flush()
            os._exit(0 if rank == 0 else 3)

        watchdog = threading.Timer(args.diagnostics_timeout, stalled)
        watchdog.daemon = True
        watchdog.start()
        extra = {}
        try:
            seen = [None] * world
            dist.all_gather_object(seen, dict(capi_rank.info(), pairwise=capi_rank.pairwise(), one_group=capi_rank.exchange_grouping(), workspace_bytes=work_bytes,
                                              cuda_device=torch.cuda.current_device()) if capi_rank is not None else {"rank": rank, "path": "sharded.py"})
            extra["ranks_seen"] = seen
            if not args.no_diagnostics:
                extra["diagnostics"] = multi_gpu_diagnostics(pkg, lib, dist, torch, args, capi_rank, system, sharded, launch, fence, step, finish, lend, stream_ptr,
                                                             rank, world, n, dtype, mode, dt, damping, bufs if capi_rank is not None else None,
                                                             vel_t, acc_t if capi_rank is not None else None, work_t, work_bytes, big, dev)
        except Exception as exc:  # noqa: BLE001 -- diagnostics must never cost the headline line
            extra["diagnostics_error"] = repr(exc)
        watchdog.cancel()
        configs = extra.get("diagnostics", {}).pop("configs", None) if isinstance(extra.get("diagnostics"), dict) else None
        if configs:
            extra["configs"] = configs
        emit(extra)
    else:
        emit()

    if capi_rank is not None:
        torch.cuda.synchronize()
        capi_rank.destroy()
    if distributed:
        dist.destroy_process_group()


CONFIG3_BODIES = 1048576  # BASELINE.json configs[3]: 1 048 576 bodies over the GPUs of one node


def multi_gpu_diagnostics(pkg, lib, dist, torch, args, capi_rank, system, sharded, launch, fence, step, finish, lend, stream_ptr, rank, world, n, dtype, mode, dt, damping,
                          bufs, vel_t, acc_t, work_t, work_bytes, big, dev):
    """The same job timed other ways, by every rank together (max over ranks, ms per step): the other exchange grouping, the
    one-sided tile schedule, the exchange legs alone, the kernels alone; and BASELINE configs[3] through the same entry points.
    Returns a dict for the JSON line."""
    def timed(fn, reps):
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        finish()
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(f"{float(t.item()) / reps * 1e3:.5g}")

    def stream_timed(fn, reps):
        """this rank's stream time (HIP events), max over ranks: for work that involves no other rank"""
        fence()
        e0, e1 = pkg.Event(), pkg.Event()
        e0.record(stream_ptr)
        for _ in range(reps):
            fn()
        e1.record(stream_ptr)
        e1.synchronize()
        t = torch.tensor([e0.elapsed_ms(e1) / reps], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        fence()
        return float(f"{float(t.item()):.5g}")

    reps = max(2, min(args.steps, 10))
    out = {"what": "ms per step, max over ranks, taken after the timed region; not part of `value`", "reps": reps}
    if capi_rank is None:
        out["exchange_alone_ms"] = timed(lambda: system.exchange_once(system.positions()), reps)
        i0, ni, schedule = system.i0, system.ni, system.schedule

        def kernels():
            for k, (j0, nj, _) in enumerate(schedule):
                launch(system.pos[1 - system.read], system.pos[system.read], system.vel, system.acc, i0, ni, j0, nj, (pkg.NB_SHARD_ACC_IN if k else 0) | (pkg.NB_SHARD_FINALIZE if k == len(schedule) - 1 else 0))

        out["one_sided_kernels_alone_ms"] = stream_timed(kernels, reps)
        return out

    was_one_group, was_pairwise = capi_rank.exchange_grouping(), capi_rank.pairwise()
    label = lambda pw, og: ("pairwise" if pw else "one_sided") + ("_one_group" if og else "_group_per_round")  # noqa: E731
    steps = {}
    # (1) the step as timed in the headline, then with the other grouping of the position rounds
    for og in (was_one_group, not was_one_group):
        capi_rank.set_exchange_grouping(og)
        steps[label(was_pairwise, og)] = timed(step, reps)
    capi_rank.set_exchange_grouping(was_one_group)
    # (2) the exchange legs on their own
    out["position_exchange_alone_ms"] = {}
    for og in (True, False):
        capi_rank.set_exchange_grouping(og)
        out["position_exchange_alone_ms"]["one_group" if og else "group_per_round"] = timed(lambda: capi_rank.exchange_once(), reps)
    capi_rank.set_exchange_grouping(was_one_group)
    if was_pairwise:
        out["reaction_exchange_alone_ms"] = timed(capi_rank.reaction_exchange_once, reps)
        # (3) this rank's kernels alone: exactly what it launches in a pairwise step, no exchange, no waits
        emulate = lib.nb_emulate_pair_rank_f32 if np.dtype(dtype) == np.float32 else lib.nb_emulate_pair_rank_f64
        need = ctypes.c_size_t(work_bytes)
        r = capi_rank.read
        out["pairwise_kernels_alone_ms"] = stream_timed(lambda: pkg.check(emulate(bufs[1 - r].data_ptr(), bufs[r].data_ptr(), vel_t.data_ptr(), work_t.data_ptr(), ctypes.byref(need), n, world, rank, dt, damping,
                                                                                 stream_ptr), "nb_emulate_pair_rank"), reps)
    i0, ni = sharded.slice_of(rank, world, n)
    schedule = sharded.tile_schedule(rank, world, n, mode == pkg.NB_MODE_STRICT)

    def tile_kernels():
        r = capi_rank.read
        for k, (j0, nj, _) in enumerate(schedule):
            launch(bufs[1 - r], bufs[r], vel_t, acc_t, i0, ni, j0, nj, (pkg.NB_SHARD_ACC_IN if k else 0) | (pkg.NB_SHARD_FINALIZE if k == len(schedule) - 1 else 0))

    out["one_sided_kernels_alone_ms"] = stream_timed(tile_kernels, reps)
    # (4) the other layout: every rank takes its workspace back (the call is collective) -> the one-sided tile schedule
    if was_pairwise:
        capi_rank.set_workspace(None, 0)
        assert not capi_rank.pairwise()
        for og in (True, False):
            capi_rank.set_exchange_grouping(og)
            steps[label(False, og)] = timed(step, reps)
        capi_rank.set_exchange_grouping(was_one_group)
        capi_rank.set_workspace(work_t.data_ptr(), work_bytes)
        assert capi_rank.pairwise()
    out["step_ms"] = steps
    out["headline_was"] = label(was_pairwise, was_one_group)
    # (5) BASELINE configs[3]: 1 048 576 bodies over the ranks, same communicator, same entry points
    if big is not None:
        pos_b, vel_b = big
        nb = pos_b.size // 4
        p0 = torch.from_numpy(pos_b.reshape(nb, 4)).to(dev)
        b_bufs, b_vel, b_acc = [p0, p0.clone()], torch.from_numpy(vel_b.reshape(nb, 4)).to(dev), torch.zeros_like(p0)
        job = pkg.ShardedRank(None, world, rank, [b.data_ptr() for b in b_bufs], b_vel.data_ptr(), b_acc.data_ptr(), nb, np.float32, mode, 256, stream_ptr, comm=capi_rank.comm)
        b_work = lend(job.workspace_bytes()) if was_pairwise else None
        job.set_workspace(b_work.data_ptr() if b_work is not None else None, b_work.numel() if b_work is not None else 0)
        job.exchange_once(0)

        def big_fence():
            job.finish()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

        big_step = lambda: job.update(np.float32(dt), np.float32(damping))  # noqa: E731
        big_step()
        big_fence()
        k = 3
        t0 = time.perf_counter()
        for _ in range(k):
            big_step()
        big_fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item()) / k * 1e3
        out["configs"] = [{"workload": "configs[3]" if nb == CONFIG3_BODIES else f"configs[3]'s shape at {nb} bodies (rehearsal)", "bodies": nb, "n_gpus": world, "dtype": "f32", "mode": "fast",
                           "layout": "pairwise across ranks" if job.pairwise() else "one-sided tiles", "steps": k, "ms_per_step": float(f"{ms:.5g}"),
                           "interactions_per_s": float(nb) * nb / (ms * 1e-3), "frac": round(20 * float(nb) * nb / (ms * 1e-3) / world / (FP32_VECTOR_PEAK_TFLOPS * 1e12), 4),
                           "workspace_bytes_per_rank": b_work.numel() if b_work is not None else 0}]
        # hand the communicator back to the headline system
        capi_rank.set_workspace(work_t.data_ptr() if work_t is not None else None, work_bytes)
        job.destroy()
    return out


if __name__ == "__main__":
    main()
