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
timed after the headline measurement.

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
    ap.add_argument("--sweep", action="store_true", help="time every fast-kernel geometry (tuning aid), N=1 only")
    ap.add_argument("--plan", type=str, default="", help="I,S,TILE override for the fast kernel, e.g. 2,1,1024")
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
    ap.add_argument("--emulate-gpus", type=int, default=0,
                    help="on ONE GPU, run rank 0's kernel schedule of a G-rank strong-scaling job (no collective): "
                         "projection aid, prints its own JSON and never the headline metric")
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


def other_configs(pkg, lib, headline):
    """BASELINE.json configs besides the headline one, plus STRICT (the parity-exact mode) and, for FAST, both layouts
    (pairwise = nb_integrate_ws_* with a workspace, one-sided = nb_integrate_*), each as
    {workload, bodies, dtype, mode, layout, steps, ms_per_step, frac}: 1 warm-up step, then K steps between two
    HIP events on the launch stream (the reference's GPU protocol, compute_cuda.cpp:183-195); frac against the same
    vector-FMA peaks as the headline (20 flop per fp32 interaction, 30 per fp64: compute.cpp:16-18)."""
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
    ]
    out = []
    for what, n, fp64, mode_name, steps in cases:
        dtype = np.float64 if fp64 else np.float32
        mode = pkg.NB_MODE_FAST if mode_name == "fast" else pkg.NB_MODE_STRICT
        layouts = ["one-sided"] if mode_name == "fast" else ["strict"]
        if mode_name == "fast" and pkg.workspace_bytes(n, dtype, mode):
            layouts.insert(0, "pairwise")
        pos0 = vel0 = None
        for layout in layouts:
            if (n, fp64, mode_name, layout) == headline:
                continue
            if pos0 is None:
                pos0, vel0 = make_bodies(n, dtype)
            system = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0, vel0, mode=mode, workspace=(layout == "pairwise"))
            dt = dtype(np.float32(0.016))
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
            flops, peak = (30, FP64_VECTOR_PEAK_TFLOPS) if fp64 else (20, FP32_VECTOR_PEAK_TFLOPS)
            # (kept short: the whole line should stay well under what a log tail holds; interactions/s = bodies^2 / ms_per_step)
            out.append({"workload": what, "bodies": n, "dtype": "f64" if fp64 else "f32", "mode": mode_name, "layout": layout, "steps": steps,
                        "ms_per_step": float(f"{ms:.5g}"), "frac": round(flops * float(n) * n / (ms * 1e-3) / (peak * 1e12), 4)})
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


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # plain `python3 bench.py --gpus N`: start the N ranks ourselves, BEFORE anything in this process touches the GPU
        # (no torch import, no HIP call so far), relay their output and exit with their status.  Children, never exec.
        raise SystemExit(self_launch(args.gpus, any(a.startswith("--exchange") for a in sys.argv[1:]), args.launch_timeout))
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

    pkg = entry.load_package()
    lib = pkg.lib()
    mode = pkg.NB_MODE_FAST if args.mode == "fast" else pkg.NB_MODE_STRICT
    if args.plan:
        pkg.set_plan_override(*[int(x) for x in args.plan.split(",")])

    if args.exchange.startswith("host"):
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

    # scratch memory of the pairwise layout: caller-owned (a torch tensor here), sized by the library, contents irrelevant
    work_t, work_bytes = None, 0
    if args.layout == "pairwise" and world == 1 and not args.sweep and not args.emulate_gpus and not args.plan:
        work_bytes = pkg.workspace_bytes(n, dtype, mode)
        if work_bytes:
            work_t = torch.empty(work_bytes, dtype=torch.uint8, device=dev)
    pairwise = work_t is not None  # (several ranks: decided below, once the communicator says what it can use)
    ws_fn = lib.nb_integrate_ws_f64 if args.fp64 else lib.nb_integrate_ws_f32

    kernel_launches = [0]

    def launch(new_pos, old_pos, vel, acc, i0, ni, j0, nj, flags):
        kernel_launches[0] += 1
        pkg.check(shard_fn(new_pos.data_ptr(), old_pos.data_ptr(), vel.data_ptr(), acc.data_ptr(), i0, ni, j0, nj, flags,
                           dt, damping, 256, mode, ctypes.c_void_p(stream.cuda_stream)), "nb_integrate_shard")

    capi_rank = None  # --exchange rccl: this rank of the product's sharded system (nb_comm_init_rank + nb_sharded_step_*)
    system = None     # every other exchange: cuda-nbody_amd/sharded.py over torch.distributed
    if distributed:
        sharded = entry.load_package_module("sharded")

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
                    capi_rank = pkg.ShardedRank(ids[0], world, rank, [b.data_ptr() for b in bufs], vel_t.data_ptr(), acc_t.data_ptr(), n, dtype, mode, 256,
                                                ctypes.c_void_p(stream.cuda_stream))
                    # bring the communicator up (channels, first-call set-up) outside any timed step, whatever --warmup says;
                    # every rank holds identical positions at this point, so exchanging them changes nothing
                    if args.layout == "pairwise" and world > 1:  # pairs once across the ranks: reaction sums travel to their owners
                        work_bytes = capi_rank.workspace_bytes()
                        if work_bytes:
                            work_t = torch.empty(work_bytes, dtype=torch.uint8, device=dev)
                            pairwise = True
                    if pairwise:
                        capi_rank.set_workspace(work_t.data_ptr(), work_bytes)  # (a world of one: the single-GPU step with its workspace)
                    capi_rank.exchange_once(0)
                    torch.cuda.synchronize()
                except pkg.NBodyHipError as exc:
                    problem = exc
            if not everyone(problem is None):
                print(f"[bench rank {rank}] the C-ABI RCCL path could not be brought up ({problem!r} on this rank); "
                      "ALL ranks fall back to the tile schedule over torch.distributed", file=sys.stderr, flush=True)
                if capi_rank is not None:
                    capi_rank.destroy()
                capi_rank, exchange_fallback = None, True
                args.exchange = "torch"
                rccl_group = torch_rccl_group()

        if capi_rank is not None:
            def step():
                kernel_launches[0] += 1 if (pairwise and world == 1) else (world // 2 + 1 if pairwise else world)  # one per position tile / per diagonal + partner
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
            if pairwise:
                kernel_launches[0] += 1  # (one call = the forces kernel + the kernel that adds the reaction slots and integrates)
                pkg.check(ws_fn(bufs[1 - r].data_ptr(), bufs[r].data_ptr(), vel_t.data_ptr(), dt, damping, n, 256, mode, work_t.data_ptr(), work_bytes,
                                ctypes.c_void_p(stream.cuda_stream)), "nb_integrate_ws")
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

    if args.emulate_gpus > 1 and world == 1 and args.sweep:
        sharded = entry.load_package_module("sharded")
        G = args.emulate_gpus
        i0, ni = sharded.slice_of(G // 2, G, n)
        sched = sharded.tile_schedule(G // 2, G, n, False) if args.exchange == "rccl" else sharded.chunk_schedule(i0, ni, n, False)
        acc_t = torch.zeros_like(pos_t)
        nxt = pos_t.clone()
        rows = []
        for I in ((1, 2, 4) if args.fp64 else (2, 4)):
            for S in (4, 8, 16, 64):
                for tile in (256, 512, 1024, 2048):
                    if S == 64:
                        if tile not in (512, 1024):
                            continue
                    elif tile < 64 * S or tile // (64 * S) not in (1, 2, 4):
                        continue
                    if lib.nb_set_plan_override(I, S, tile) != 0:
                        continue
                    def one_step():
                        for k, (j0, nj, _) in enumerate(sched):
                            flags = (pkg.NB_SHARD_ACC_IN if k else 0) | (pkg.NB_SHARD_FINALIZE if k == len(sched) - 1 else 0)
                            launch(nxt, pos_t, vel_t, acc_t, i0, ni, j0, nj, flags)
                    for _ in range(2):
                        one_step()
                    e0, e1 = pkg.Event(), pkg.Event()
                    torch.cuda.synchronize()
                    e0.record(ctypes.c_void_p(stream.cuda_stream))
                    for _ in range(10):
                        one_step()
                    e1.record(ctypes.c_void_p(stream.cuda_stream))
                    e1.synchronize()
                    ms = e0.elapsed_ms(e1) / 10
                    rows.append(dict(I=I, S=S, tile=tile, ms=round(ms, 4), speedup_vs_ideal=round((float(n) * n / G) / (ms * 1e-3) * 1e-12, 3)))
                    print(json.dumps(rows[-1]), flush=True)
        pkg.set_plan_override(0, 0, 0)
        print("best:", json.dumps(min(rows, key=lambda r: r["ms"])))
        return

    if args.emulate_gpus > 1 and world == 1 and args.layout == "pairwise" and mode == pkg.NB_MODE_FAST:
        # one rank's kernels of the pairwise step across G ranks (nb_emulate_pair_rank_*: diagonal, G/2 rectangles, folds, finish)
        G = args.emulate_gpus
        emulate = lib.nb_emulate_pair_rank_f64 if args.fp64 else lib.nb_emulate_pair_rank_f32
        need = ctypes.c_size_t(0)
        pkg.check(emulate(None, None, None, None, ctypes.byref(need), n, G, 0, dt, damping, None), "nb_emulate_pair_rank (size)")
        work = torch.empty(need.value, dtype=torch.uint8, device=dev)
        nxt, out = pos_t.clone(), []
        for r in sorted({0, G // 2, G - 1}):
            def one_step():
                pkg.check(emulate(nxt.data_ptr(), pos_t.data_ptr(), vel_t.data_ptr(), work.data_ptr(), ctypes.byref(need), n, G, r, dt, damping,
                                  ctypes.c_void_p(stream.cuda_stream)), "nb_emulate_pair_rank")
            for _ in range(args.warmup):
                one_step()
            e0, e1 = pkg.Event(), pkg.Event()
            torch.cuda.synchronize()
            e0.record(ctypes.c_void_p(stream.cuda_stream))
            for _ in range(args.steps):
                one_step()
            e1.record(ctypes.c_void_p(stream.cuda_stream))
            e1.synchronize()
            out.append({"rank": r, "ms_per_step_kernels_only": e0.elapsed_ms(e1) / args.steps, "launches_per_step": 2 * (G // 2) + 2})
        worst = max(o["ms_per_step_kernels_only"] for o in out)
        print(json.dumps({"emulated_gpus": G, "bodies": n, "schedule": "pairwise across ranks: diagonal + G/2 rectangles, reaction sums to their owners",
                          "workspace_bytes_per_rank": need.value, "ranks": out, "projected_interactions_per_s_excluding_exchange": float(n) * n / (worst * 1e-3)}), flush=True)
        return

    if args.emulate_gpus > 1 and world == 1:
        sharded = entry.load_package_module("sharded")
        G = args.emulate_gpus
        out = []
        for r in sorted({0, G // 2, G - 1}):
            i0, ni = sharded.slice_of(r, G, n)
            if args.exchange == "rccl":  # the tile form: one kernel per position tile
                sched = sharded.tile_schedule(r, G, n, mode == pkg.NB_MODE_STRICT)
            else:
                sched = sharded.chunk_schedule(i0, ni, n, mode == pkg.NB_MODE_STRICT)
            acc_t = torch.zeros_like(pos_t)
            nxt = pos_t.clone()

            def one_step():
                for k, (j0, nj, _) in enumerate(sched):
                    flags = (pkg.NB_SHARD_ACC_IN if k else 0) | (pkg.NB_SHARD_FINALIZE if k == len(sched) - 1 else 0)
                    launch(nxt, pos_t, vel_t, acc_t, i0, ni, j0, nj, flags)

            for _ in range(args.warmup):
                one_step()
            e0, e1 = pkg.Event(), pkg.Event()
            torch.cuda.synchronize()
            e0.record(ctypes.c_void_p(stream.cuda_stream))
            for _ in range(args.steps):
                one_step()
            e1.record(ctypes.c_void_p(stream.cuda_stream))
            e1.synchronize()
            ms = e0.elapsed_ms(e1) / args.steps
            pl = pkg.plan(ni, n, dtype)
            out.append({"rank": r, "ms_per_step_kernels_only": ms, "launches_per_step": len(sched),
                        "plan": [pl.bodies_per_lane, pl.lanes_per_body, pl.tile_bodies, pl.grid_blocks]})
        worst = max(o["ms_per_step_kernels_only"] for o in out)
        print(json.dumps({"emulated_gpus": G, "bodies": n, "schedule": "tiles" if args.exchange == "rccl" else "own/below/above", "ranks": out,
                          "projected_interactions_per_s_excluding_exchange": float(n) * n / (worst * 1e-3)}), flush=True)
        return

    if args.sweep and world == 1:
        results = []
        for I in ((1, 2, 4) if args.fp64 else (2, 4)):
            for S in (4, 8, 16, 64):
                for tile in (256, 512, 1024, 2048):
                    blk = 256 if S == 64 else 64 * S
                    if tile < blk or tile // blk not in (1, 2, 4):
                        continue
                    if S == 64 and (tile not in (512, 1024) or I > (2 if args.fp64 else 4)):
                        continue
                    pkg.set_plan_override(I, S, tile)
                    for _ in range(2):
                        step()
                    e0, e1 = pkg.Event(), pkg.Event()
                    torch.cuda.synchronize()
                    e0.record(ctypes.c_void_p(stream.cuda_stream))
                    for _ in range(5):
                        step()
                    e1.record(ctypes.c_void_p(stream.cuda_stream))
                    e1.synchronize()
                    ms = e0.elapsed_ms(e1) / 5
                    results.append(dict(I=I, S=S, tile=tile, ms=round(ms, 4), ginter=round(n * n / ms * 1e-6, 1),
                                        frac=round(flops_per * n * n / (ms * 1e-3) / (peak * 1e12), 4)))
                    print(json.dumps(results[-1]), flush=True)
        pkg.set_plan_override(0, 0, 0)
        best = max(results, key=lambda r: r["ginter"])
        print("best:", json.dumps(best))
        return

    for _ in range(args.warmup):
        step()
    fence()
    launches_before = kernel_launches[0]
    ev0, ev1 = pkg.Event(), pkg.Event()
    t0 = time.perf_counter()
    ev0.record(ctypes.c_void_p(stream.cuda_stream))
    for _ in range(args.steps):
        step()
    finish()
    ev1.record(ctypes.c_void_p(stream.cuda_stream))
    fence()
    elapsed = time.perf_counter() - t0
    ev1.synchronize()
    kernel_ms_total = ev0.elapsed_ms(ev1)
    launches = kernel_launches[0] - launches_before

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64)  # gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        interactions = float(n) * float(n) * args.steps
        value = interactions / elapsed
        # dominant kernel: per-launch duration from HIP events on the launch stream.  At N=1 the stream holds
        # nothing but the K back-to-back launches, so events/K is the kernel's average duration (the rocprofv3
        # --kernel-trace --stats average under profiles/ agrees).  For N>1 it is the rank-0 step time.
        ms_per_launch = kernel_ms_total / max(launches, 1)
        per_launch_interactions = float(n) * float(n) * args.steps / max(launches, 1) / world
        achieved_tflops = flops_per * per_launch_interactions / (ms_per_launch * 1e-3) / 1e12
        plan = pkg.plan(n // world, n, dtype)
        pair = pkg.pair_plan(n, dtype) if (pairwise and world == 1) else None
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
                    ("C-ABI (nb_comm_init_rank + nb_sharded_step_*, csrc/nbody_comm.hip), PAIRWISE across the ranks: each rank evaluates its own slice and "
                     "the rectangles against ranks r+1 .. r+G/2 once per pair and sends the reaction sums (N/G x 12 B per partner) to their owners; "
                     "positions all-gathered as G-1 RCCL send/recv tiles on the communicator's high-priority stream" if pairwise else
                     "C-ABI (nb_comm_init_rank + nb_sharded_step_*, csrc/nbody_comm.hip): RCCL all-gather of the new positions per step, issued as "
                     "G-1 position tiles (grouped ncclSend/ncclRecv rounds on the communicator's high-priority stream); the kernel of tile k waits "
                     "only on round k, the own-slice chunk runs first") if capi_rank is not None else
                    "torch.distributed re-implementation (sharded.py) of the tile schedule: batch_isend_irecv rounds on RCCL's stream" if system.exchange == "tiles" else
                    "torch.distributed (sharded.py): RCCL all_gather_into_tensor of the new positions per step, overlapped with the own-slice j chunk"),
                "layout": "pairwise (every pair of bodies evaluated once, reaction sums through a caller-owned workspace)" if pairwise else
                          "one-sided (every directed interaction evaluated, as bodysystemcuda.cu:125-146 does)",
                "step_entry_point": ("nb_integrate_ws_*" if pairwise else "nb_integrate_shard_*") if not distributed else ("nb_sharded_step_*" if capi_rank is not None else "sharded.py -> nb_integrate_shard_*"),
                "kernel_plan": plan_now,
                "device": info.name.decode(),
                "arch": info.arch.decode(),
            },
            "gflops": value * flops_per * 1e-9,
            "flops_per_interaction": flops_per,
            "roofline": {
                "bound": "valu_fp32_fma" if not args.fp64 else "valu_fp64_fma",
                "achieved": achieved_tflops,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved_tflops / peak,
                "traffic": traffic,  # HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes (tools/profile.sh)
                "traffic_source": traffic_src,
                # achieved / what this instruction mix can issue at best on the chip (see FP32_ISSUE_CEILING_...): how close
                # the kernel is to ITS ceiling; `frac` above is against the nominal "20 flop" peak
                # (the ceilings are those of the FAST instruction mix; STRICT executes other, exactly rounded, operations)
                "issue_ceiling_frac": None if (args.mode != "fast" or pairwise) else value / world / (FP64_ISSUE_CEILING_INTERACTIONS_PER_S if args.fp64 else FP32_ISSUE_CEILING_INTERACTIONS_PER_S),
                "issue_ceiling_interactions_per_s": None if (args.mode != "fast" or pairwise) else (FP64_ISSUE_CEILING_INTERACTIONS_PER_S if args.fp64 else FP32_ISSUE_CEILING_INTERACTIONS_PER_S),
                "kernel_ms": ms_per_launch,
                # pairwise layout: the step is two kernels (pair_forces, pair_finish) and evaluates each pair of bodies once; `achieved` and
                # `frac` above count the ALGORITHMIC 20 (30) flop per directed interaction of the reference convention (compute.cpp:16-18),
                # `executed` counts what the kernels really issue: 24 (36) flop per pair evaluation
                "executed": None if pair is None else {
                    "pair_evaluations_per_launch": float(pair.blocks) * (pair.blocks // 2 + 1) * pair.block_bodies * pair.block_bodies,
                    "flops_per_pair_evaluation": 36 if args.fp64 else 24,
                    "tflops": (36 if args.fp64 else 24) * float(pair.blocks) * (pair.blocks // 2 + 1) * pair.block_bodies * pair.block_bodies / (ms_per_launch * 1e-3) / 1e12,
                    "frac": (36 if args.fp64 else 24) * float(pair.blocks) * (pair.blocks // 2 + 1) * pair.block_bodies * pair.block_bodies / (ms_per_launch * 1e-3) / 1e12 / peak},
                "algorithmic_flops_per_launch": flops_per * per_launch_interactions,
                # positions + velocities in and out; the pairwise layout also writes and reads its reaction slots once
                "algorithmic_hbm_bytes_per_launch": (128 if args.fp64 else 64) * (n // world) + (2 * pair.workspace_bytes if pair is not None else 0),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            O = entry.load_oracle()
            orc1 = O.Oracle()
            pos_h, vel_h = orc1.startup_state(n, dtype)
            # the workload above came from the product's randomise_bodies; the checker's must be the same bytes
            assert pos_h.tobytes() == pos0.tobytes() and vel_h.tobytes() == vel0.tobytes(), "product and oracle start-up bodies differ"
            sample = args.cpu_sample_bodies or max(8, min(n, int(2.0e10 // n) // 8 * 8))
            base = {}
            # OpenMP leg: the reference's fp32 loop forks INSIDE the j loop (bodysystemcpu.cpp:156-168), i.e. one
            # fork/join per body j -- it is slow by construction, so it gets a smaller sample and at most the
            # box's CPU share (16 threads per GPU).
            for key, omp, smp in (("one_thread", False, sample), ("openmp", True, max(8, sample // 16 // 8 * 8))):
                orc = O.Oracle(openmp=omp)
                if omp:
                    orc.set_num_threads(min(16, os.cpu_count() or 1))
                ms = orc.benchmark_partial(pos_h, smp)
                base[key] = {"value": smp * float(n) / (ms * 1e-3), "cores": orc.num_threads() if omp else 1, "ms": ms,
                             "sample_bodies_i": smp}
            line["cpu_baseline"] = {
                "value": base["one_thread"]["value"],
                "unit": "interactions/s",
                "cores": 1,
                "kind": "port",
                "sample": f"force pass of BodySystemCPU::update (oracle/ port) for the first {sample} bodies i against all {n} bodies j = {sample * n:.3g} "
                          f"interactions; 1 thread is how the reference ships",
                "openmp": base["openmp"],
            }
        if world == 1 and not args.no_configs and not args.plan:
            # The other BASELINE configs and the parity-exact mode, timed AFTER the headline measurement (never inside it)
            # so that one driver-run line carries them all.  A failure here costs only this list, never the headline.
            try:
                line["configs"] = other_configs(pkg, lib, (n, args.fp64, args.mode, "pairwise" if pairwise else ("one-sided" if args.mode == "fast" else "strict")))
            except Exception as exc:  # noqa: BLE001
                line["configs"] = [{"error": repr(exc)}]
            if pairwise:
                # No multi-GPU node has run this yet.  What CAN be measured on one GPU is the compute side: exactly the kernels one rank
                # of a G-rank pairwise step launches (nb_emulate_pair_rank_*), alone on the chip, no exchange.  A projection, labelled so.
                try:
                    line["multi_gpu_kernel_projection"] = rank_projection(pkg, lib, n, dtype, dt, damping, elapsed / args.steps * 1e3)
                except Exception as exc:  # noqa: BLE001
                    line["multi_gpu_kernel_projection"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)
        if args.dump_state:
            torch.cuda.synchronize()
            final = capi_rank.pos[capi_rank.read] if capi_rank is not None else (system.positions().data_ptr() if system is not None else bufs[state["read"]].data_ptr())
            host = np.zeros(4 * n, dtype)
            pkg.check(lib.nb_d2h(host.ctypes.data_as(ctypes.c_void_p), final, host.nbytes, None), "nb_d2h")
            np.savez(args.dump_state, final=host, initial=pos0)  # (the bodies the run started from, too: a test can tell a different start from a different step)

    # Diagnostics for N > 1: the exchange alone and the kernels of one step alone (exposed exchange = step - kernels).
    # Taken after the timed region, never part of `value`, and printed to STDERR after the JSON line is already out,
    # so nothing here can cost the result.
    diagnostics = None
    if world > 1:
        try:
            fence()
            t1 = time.perf_counter()
            for _ in range(10):
                if capi_rank is not None:
                    capi_rank.exchange_once()
                else:
                    system.exchange_once(system.positions())
            torch.cuda.synchronize()
            exchange_ms = (time.perf_counter() - t1) / 10 * 1e3
            fence()
            if capi_rank is not None:
                cur, nxt, d_vel, d_acc = bufs[capi_rank.read], bufs[1 - capi_rank.read], vel_t, acc_t
                i0, ni = sharded.slice_of(rank, world, n)
                schedule = sharded.tile_schedule(rank, world, n, mode == pkg.NB_MODE_STRICT)
            else:
                cur, nxt, d_vel, d_acc = system.pos[system.read], system.pos[1 - system.read], system.vel, system.acc
                i0, ni, schedule = system.i0, system.ni, system.schedule
            e0, e1 = pkg.Event(), pkg.Event()
            e0.record(ctypes.c_void_p(stream.cuda_stream))
            reps = max(2, min(args.steps, 10))
            for _ in range(reps):
                for k, (j0, nj, _) in enumerate(schedule):
                    flags = (pkg.NB_SHARD_ACC_IN if k else 0) | (pkg.NB_SHARD_FINALIZE if k == len(schedule) - 1 else 0)
                    launch(nxt, cur, d_vel, d_acc, i0, ni, j0, nj, flags)
            e1.record(ctypes.c_void_p(stream.cuda_stream))
            e1.synchronize()
            diagnostics = {"exchange_alone_ms": exchange_ms, "kernels_alone_ms_per_step_rank0": e0.elapsed_ms(e1) / reps,
                           "launches_per_step_rank0": len(schedule)}
            fence()
        except Exception as exc:  # diagnostics must never cost the headline line
            diagnostics = {"error": repr(exc)}

    if diagnostics is not None and rank == 0:
        print("diagnostics: " + json.dumps(diagnostics), file=sys.stderr, flush=True)

    if capi_rank is not None:
        torch.cuda.synchronize()
        capi_rank.destroy()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
