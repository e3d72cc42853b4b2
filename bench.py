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

from bench_support import (CONFIG3_BODIES, ChipWatch, FP32_ISSUE_CEILING_INTERACTIONS_PER_S, FP32_VECTOR_PEAK_TFLOPS, FP64_ISSUE_CEILING_INTERACTIONS_PER_S,  # noqa: E402
                           FP64_PEAK_NOTE, FP64_VECTOR_PEAK_TFLOPS, single_gpu_reference, cpu_baseline, make_bodies, multi_gpu_diagnostics, other_configs, pair_evaluations, pair_kernel_split,
                           plan_dict, pmc_summary, rank_projection)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-seconds", type=float, default=0.3,
                    help="untimed steps BEFORE the --warmup steps, for about this long: from idle the card needs ~0.2 s under load to reach the clock it then holds "
                         "(profiles/round4_clock_and_power.txt), and K steps of an N-GPU job can be over in 25 ms.  The count is in the line (settle_steps); "
                         "0 = none (also with --dump-state: the state after exactly warmup + steps steps)")
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
    ap.add_argument("--launch-timeout", type=float, default=150.0,
                    help="plain `bench.py --gpus N`: seconds the launcher gives ONE attempt of the N ranks before ending them (see self_launch)")
    ap.add_argument("--launch-budget", type=float, default=560.0,
                    help="plain `bench.py --gpus N`: seconds for the whole command -- the first attempt and up to three plainer exchanges (under the driver's 600 s)")
    ap.add_argument("--rccl-init-timeout", type=float, default=90.0,
                    help="N>1, --exchange rccl: seconds a rank gives nb_comm_init_rank (ncclCommInitRank) before it calls it hung; ALL ranks then take "
                         "the RCCL-free exchange (gloo through host memory, 'exchange_fallback': true) instead of waiting for the headline watchdog")
    ap.add_argument("--bringup-timeout", type=float, default=60.0,
                    help="plain `bench.py --gpus N`: an attempt is ended when no rank has its exchange up this long after the first rank imported torch")
    ap.add_argument("--headline-timeout", type=float, default=240.0,
                    help="N>1: seconds a rank allows bring-up + warm-up + the timed steps before it says where it is stuck and leaves (status 5) -- "
                         "under torch.distributed.run nothing else would end a rank that sits inside a collective")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="REHEARSAL of the N-rank C-ABI path on a one-GPU box: every rank uses device 0 and RCCL is replaced by the test double "
                         "(NBODY_RCCL_LIB=tests/fake_rccl/libfake_rccl.so with FAKE_RCCL_IPC=1, set here when absent).  Never a performance number.")
    ap.add_argument("--no-chip-watch", action="store_true", help="do not sample the card's clock and power (sysfs) during the timed region")
    ap.add_argument("--no-diagnostics", action="store_true", help="N>1: skip the A/B timings and BASELINE configs[3] after the timed region")
    ap.add_argument("--diagnostics-timeout", type=float, default=150.0, help="N>1: seconds the post-headline measurements may take before the line is printed without them")
    return ap.parse_args()


BRINGUP_MARK = "] up:"          # every rank writes "[bench rank R] up: ..." to stderr once its exchange is up, before the warm-up
IMPORTED_MARK = "] torch imported"  # ... and "[bench rank R] torch imported" right after the import (a fresh box pages torch in for a minute or two)
EXPECTS_MARK = "] expects "        # ... and "[bench rank R] expects S s until the headline" once two probe steps have said how long a step takes


def self_launch(n_ranks: int, explicit_exchange: bool, attempt_s: float, budget_s: float = 560.0, bringup_s: float = 60.0, import_s: float = 150.0, make_cmd=None) -> int:
    """Run this same command line as `n_ranks` ranks under torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1) as a CHILD process group and relay its output.  The >1-GPU path has not run on hardware yet, so the launcher
    carries one safety net: when the ranks fail, or stall, before rank 0 printed its line, and the exchange was not chosen on
    the command line, their process group is ended (by its exact id) and the job is started again with a plainer exchange:
    `--exchange torch` (the tile schedule over torch.distributed), `--exchange allgather` (one all-gather per step), then
    `--exchange staged` (gloo through host memory, no RCCL) -- at most those three further attempts, each marked
    "exchange_fallback": true in its JSON line.

    The net has to close INSIDE what the caller allows the whole command (the driver: 600 s), so time is rationed: the command
    as a whole gets `budget_s`; an attempt gets `attempt_s` of it at most, and is ended early when no rank has reported its
    exchange up (BRINGUP_MARK on stderr) `bringup_s` after the first rank imported torch (IMPORTED_MARK; `import_s` at the
    latest after the start) -- a run that will finish has printed that line within seconds, one that sits in a rendezvous or in
    RCCL's bring-up never does.  An attempt's own time counts from that import mark too; once its line is out, an attempt may use
    what is left of the whole budget (the diagnostics run after the line).  Round 6 (advisor): ranks that are UP and have said how long
    they expect to need (EXPECTS_MARK, from two probe steps: a large --bodies / --steps, fp64 or strict run is slow, not hung) are
    left alone until twice that time + 30 s has passed -- inside the whole budget --, instead of being ended at the attempt's limit and
    retried with exchanges that are slower still.  `make_cmd(extra_flags, port) -> argv`: the command of an attempt (tests pass stand-in ranks)."""
    import signal
    import socket
    import subprocess
    import threading

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    started = time.monotonic()
    if make_cmd is None:
        def make_cmd(extra, port):
            return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                    os.path.abspath(__file__)] + sys.argv[1:] + extra

    def end_group(child, why):
        print(f"[bench] {why}: ending process group {child.pid}", file=sys.stderr, flush=True)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)  # the group this call created (start_new_session), nothing else
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        return child.returncode if child.returncode is not None else -9

    def attempt(extra, attempts_left):
        # what is left of the budget, shared fairly with the attempts that may still have to follow
        left = budget_s - (time.monotonic() - started)
        limit = max(5.0, min(attempt_s, left / max(1, attempts_left) if explicit_exchange is False else left))
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        child = subprocess.Popen(make_cmd(extra, port), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        seen = {"metric": False, "up": False, "imported_at": None, "expected_by": None}

        def relay_out():
            for line in child.stdout:
                if line.startswith("{") and '"metric"' in line:
                    seen["metric"] = True
                sys.stdout.write(line)
                sys.stdout.flush()

        def relay_err():
            for line in child.stderr:
                if BRINGUP_MARK in line:
                    seen["up"] = True
                elif IMPORTED_MARK in line and seen["imported_at"] is None:
                    seen["imported_at"] = time.monotonic()
                elif EXPECTS_MARK in line:
                    try:  # "[bench rank R] expects 123.4 s until the headline": alive and slow is not hung
                        seen["expected_by"] = time.monotonic() + 2.0 * float(line.split(EXPECTS_MARK, 1)[1].split()[0]) + 30.0
                    except (ValueError, IndexError):
                        pass
                sys.stderr.write(line)
                sys.stderr.flush()

        readers = [threading.Thread(target=relay_out, daemon=True), threading.Thread(target=relay_err, daemon=True)]
        for r in readers:
            r.start()
        begun, rc = time.monotonic(), None
        while rc is None:
            try:
                rc = child.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                now = time.monotonic()
                # an attempt's time counts from the first rank's import mark (a fresh box pages torch in for a minute or two: not the
                # run's fault), from `import_s` after its start at the latest
                imported = seen["imported_at"]
                ran = now - (imported if imported is not None else begun + import_s)
                if seen["metric"]:
                    # the line is out: what follows (diagnostics, tear-down) may take what is left of the whole budget, no more
                    if now - started > budget_s:
                        rc = end_group(child, f"the line is out and the launcher's budget of {budget_s:.0f} s is spent")
                elif ran > limit and seen["up"] and seen["expected_by"] is not None and now < seen["expected_by"] and now - started < budget_s - 5.0:
                    pass  # up, stepping, and within what the ranks themselves said they need: let them work
                elif ran > limit:
                    rc = end_group(child, f"ranks still running after {limit:.0f} s" + (" (and past twice the time they expected to need)" if seen["expected_by"] is not None else ""))
                elif not seen["up"] and (ran > bringup_s or (imported is None and ran > 0)):  # (no rank even imported torch in `import_s`)
                    rc = end_group(child, f"no rank has its exchange up {now - begun:.0f} s after the start")
        for r in readers:
            r.join(timeout=10)
        return rc, seen["metric"]

    plainer = [] if explicit_exchange else ["torch", "allgather", "staged"]
    rc, reported = attempt([], 1 + len(plainer))
    # plainer and plainer: the tile schedule over torch.distributed, one all-gather per step, then no RCCL at all.  A line
    # produced by a retry says so at its top level ("exchange_fallback": true), not only in config.exchange.
    env["NBODY_BENCH_EXCHANGE_FALLBACK"] = "1"
    for k, fallback in enumerate(plainer):
        if rc == 0 or reported:
            break
        if budget_s - (time.monotonic() - started) < 10.0:
            print(f"[bench] the {n_ranks}-rank run ended with status {rc} before reporting and the launcher's budget of {budget_s:.0f} s is spent", file=sys.stderr, flush=True)
            break
        print(f"[bench] the {n_ranks}-rank run ended with status {rc} before reporting; one more attempt with --exchange {fallback}", file=sys.stderr, flush=True)
        rc, reported = attempt(["--exchange", fallback], len(plainer) - k)
    return rc


def pci_address(torch, device_index):
    """'0000:c5:00.0' of a HIP device, as sysfs spells it (None when torch does not say)."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:  # noqa: BLE001
        return None


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
        raise SystemExit(self_launch(args.gpus, any(a.startswith("--exchange") for a in sys.argv[1:]) or args.rehearse_one_gpu, args.launch_timeout, args.launch_budget,
                                     args.bringup_timeout))
    if args.gpus != world:
        args.gpus = world

    dtype = np.float64 if args.fp64 else np.float32
    n = args.bodies
    flops_per = 30 if args.fp64 else 20
    peak = FP64_VECTOR_PEAK_TFLOPS if args.fp64 else FP32_VECTOR_PEAK_TFLOPS

    # torch first: it bundles its own libamdhip64.so.7; loading libnbody_hip.so afterwards binds to that one
    # copy by SONAME.  The other order puts two HIP runtimes in the process and the second sees no device.
    import torch

    distributed_env = world > 1 or "RANK" in os.environ
    started_at = time.monotonic()
    stage = {"name": "start-up", "since": time.monotonic()}

    def enter(name):
        """where this rank is (for the headline watchdog and the launcher: both read stderr)"""
        stage["name"], stage["since"] = name, time.monotonic()

    headline_watchdog = None
    if distributed_env:
        print(f"[bench rank {rank}] torch imported", file=sys.stderr, flush=True)  # (IMPORTED_MARK: the launcher's bring-up clock starts here)

        def stuck():
            sys.stderr.write(f"[bench rank {rank}] no headline after {time.monotonic() - started_at:.0f} s: stuck in '{stage['name']}' for {time.monotonic() - stage['since']:.0f} s; leaving with status 5\n")
            sys.stderr.flush()
            os._exit(5)

        headline_watchdog = threading.Timer(args.headline_timeout, stuck)
        headline_watchdog.daemon = True
        headline_watchdog.start()

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
    rccl_init_hung = False  # a thread of this process is still inside ncclCommInitRank (or ncclCommDestroy): see the end of main()
    if distributed:
        import torch.distributed as dist

        # The default process group is gloo, always: rendezvous, barriers, the unique-id broadcast, collective decisions and
        # the time reduction.  The data path is either the product's own RCCL communicator behind the C-ABI (--exchange rccl)
        # or, for the torch.distributed re-implementation (--exchange torch|allgather), a separate "nccl" group.
        enter("torch.distributed rendezvous (gloo)")
        dist.init_process_group("gloo")

        def torch_rccl_group():
            # RCCL's kernels compete with the force kernel for CUs (a 1024-thread workgroup fills a CU's VGPRs): a
            # high-priority stream lets their few workgroups dispatch first, so the exchange overlaps the own-slice chunk
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                return dist.new_group(backend="nccl", pg_options=opts)
            except (AttributeError, TypeError):
                return dist.new_group(backend="nccl")

        def step_on_a_placed_stream():
            """torch.distributed's RCCL next to our kernels: not on the null stream (torch's current stream), where RCCL works -- a rank
            computing there steps ~40 % slower (profiles/round5_hw_queue_collision.txt).  A stream probed to be clear of the null stream's
            hardware queue becomes torch's current stream (sharded.py orders its kernels against the collectives on the current stream)."""
            placed = ctypes.c_void_p()
            pkg.check(lib.nb_stream_create_placed(ctypes.byref(placed)), "nb_stream_create_placed")
            torch.cuda.synchronize()
            torch.cuda.set_stream(torch.cuda.ExternalStream(placed.value, device=dev))
            return placed

        if args.exchange in ("torch", "allgather"):
            rccl_group = torch_rccl_group()
            if world > 1:
                step_on_a_placed_stream()  # (torch's current stream from here on: what `stream` / `stream_ptr` below pick up)
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

    own_stream = None  # (several ranks through the C-ABI: the well-placed stream the steps run on)
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
            enter("C-ABI communicator bring-up (nb_comm_unique_id / nb_comm_init_rank / nb_comm_set_workspace / first exchange)")
            bufs = [pos_t, pos_t.clone()]
            acc_t = torch.zeros_like(pos_t)
            problem, init_hung = None, False
            try:
                ids = [pkg.comm_unique_id() if (rank == 0 and world > 1) else None]
            except pkg.NBodyHipError as exc:
                ids, problem = [None], exc
            dist.broadcast_object_list(ids, src=0)
            if world > 1 and ids[0] is None:
                problem = problem or RuntimeError("rank 0 could not create the RCCL unique id")
            else:
                # ncclCommInitRank is the one call of the bring-up that can wait for a peer for ever with nothing on the GPU yet, so it
                # runs in a thread of its own with a limit: a rank whose call has not returned says so, and ALL ranks then take the
                # RCCL-free exchange below (the stuck thread is left behind; whatever it may still make is never used)
                made = {}

                def init_rank():
                    try:
                        pkg.check(lib.nb_set_device(local_rank), "nb_set_device")  # (HIP's current device is per thread)
                        made["rank"] = pkg.ShardedRank(ids[0], world, rank, [b.data_ptr() for b in bufs], vel_t.data_ptr(), acc_t.data_ptr(), n, dtype, mode, 256, stream_ptr)
                    except pkg.NBodyHipError as exc:
                        made["problem"] = exc

                worker = threading.Thread(target=init_rank, daemon=True)
                worker.start()
                worker.join(args.rccl_init_timeout if world > 1 else None)
                if worker.is_alive():
                    init_hung = True
                    problem = TimeoutError(f"nb_comm_init_rank has not returned after {args.rccl_init_timeout:.0f} s")
                else:
                    capi_rank, problem = made.get("rank"), made.get("problem")
            if world > 1 and not everyone(not init_hung):  # (collective: a hang on ANY rank sends every rank the RCCL-free way)
                if capi_rank is not None:  # (this rank's call did return: its communicator goes, in a thread -- the peers' may never answer)
                    threading.Thread(target=capi_rank.destroy, daemon=True).start()
                print(f"[bench rank {rank}] RCCL's bring-up hung on some rank ({problem!r} on this one); ALL ranks fall back to the exchange without RCCL (gloo through host memory)",
                      file=sys.stderr, flush=True)
                capi_rank, exchange_fallback, pairwise, init_hung = None, True, False, True
                rccl_init_hung = True
                args.exchange = "staged"
            if capi_rank is not None and world > 1 and problem is None:
                # Never step on the null stream (torch's current stream) next to RCCL: a rank that computes there -- or on a stream
                # that shares its hardware queue, one created stream in three -- steps ~40 % slower (measured with the real RCCL,
                # profiles/round5_hw_queue_collision.txt).  The communicator hands out a stream that is probed to be clear of it.
                try:
                    torch.cuda.synchronize()  # (what torch did to the tensors so far ran on ITS stream)
                    own_stream = capi_rank.make_step_stream()
                    stream_ptr = own_stream
                except pkg.NBodyHipError as exc:
                    problem = exc
            if not init_hung and everyone(problem is None):
                # The communicator is up on every rank.  Lend it the workspace: with several ranks nb_comm_set_workspace is a
                # COLLECTIVE (every rank calls it, with nothing if it has nothing), after which the layout of a step -- pairwise
                # across the ranks or one-sided tiles -- is the communicator's, the same on every rank.
                try:
                    if want_pairwise:
                        work_t = lend(capi_rank.workspace_bytes())
                        work_bytes = work_t.numel() if work_t is not None else 0
                    capi_rank.set_workspace(work_t.data_ptr() if work_t is not None else None, work_bytes)
                    pairwise = capi_rank.pairwise()
                    # bring the communicator up (channels, first-call set-up, the transport's registration of every buffer it will
                    # ever be handed) outside any timed step, whatever --warmup says: BOTH position arrays -- every rank holds
                    # identical positions in both at this point, so exchanging them changes nothing -- and, pairwise, the send and
                    # receive arrays of the reaction leg (whatever they hold: every step overwrites them before it reads them)
                    capi_rank.exchange_once(0)
                    capi_rank.exchange_once(1)
                    if pairwise and world > 1:
                        capi_rank.reaction_exchange_once()
                        # ... and the one-off probe of the rank's second compute stream against this one (a shared hardware queue
                        # would serialise the two) happens here, not inside the first step
                        pkg.check(lib.nb_comm_settle_side_stream(capi_rank.comm, stream_ptr), "nb_comm_settle_side_stream")
                    torch.cuda.synchronize()
                except pkg.NBodyHipError as exc:
                    problem = exc
            if not init_hung and not everyone(problem is None):
                print(f"[bench rank {rank}] the C-ABI RCCL path could not be brought up ({problem!r} on this rank); "
                      "ALL ranks fall back to the tile schedule over torch.distributed", file=sys.stderr, flush=True)
                if capi_rank is not None:
                    capi_rank.destroy()
                capi_rank, exchange_fallback, pairwise = None, True, False
                stream_ptr = step_on_a_placed_stream()  # (sharded.py orders its kernels against torch's collectives on torch's CURRENT stream: this one now)
                args.exchange = "torch"
                rccl_group = torch_rccl_group()

        if capi_rank is not None:
            def step():
                capi_rank.update(dt, damping)

            finish = capi_rank.finish
        else:
            enter(f"bring-up of the torch.distributed exchange ({args.exchange})")
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
    # (a child process per rank reading two sysfs files of the rank's own card; see bench_support.ChipWatch.  Rank 0's goes into `roofline`,
    # every rank's into `ranks_seen`: on a multi-GPU node the cards need not be granted the same clock)
    chip = ChipWatch(None if args.no_chip_watch else pci_address(torch, local_rank))
    if distributed:  # (BRINGUP_MARK: what the launcher waits for before it trusts an attempt with its full time)
        print(f"[bench rank {rank}] up: exchange {args.exchange}, {'pairwise across the ranks' if pairwise else 'one-sided'}, {world} rank(s)", file=sys.stderr, flush=True)
    # From idle the card ramps its clock for ~0.2 s under load (2.10 -> 2.29 GHz measured, profiles/round4_clock_and_power.txt); W warm-up
    # steps are 48 ms at N = 1 and 6 ms at N = 8, so the K timed steps would measure the ramp, not the integrator.  Untimed steps for
    # ~settle_seconds come first -- the SAME number on every rank (a step is collective): two steps are timed, the slowest rank's time
    # decides the count -- and the line says how many ("settle_steps").  Then W warm-up steps and exactly K timed steps, as before.
    settle_steps = 0
    if args.settle_seconds > 0 and not args.dump_state:
        enter("clock settle (untimed steps)")
        fence()
        t_probe = time.perf_counter()
        step()
        step()
        finish()
        torch.cuda.synchronize()
        probe = torch.tensor([(time.perf_counter() - t_probe) / 2.0], dtype=torch.float64)
        if distributed:
            dist.all_reduce(probe, op=dist.ReduceOp.MAX)  # gloo: every rank derives the same count
        settle_steps = int(min(4000, max(0, round(args.settle_seconds / max(float(probe.item()), 1e-6)))))
        settle_steps += settle_steps & 1  # (even: the ping-pong ends where it began)
        if distributed_env:
            # alive and slow is not hung: say how long the rest will take (the launcher reads it: EXPECTS_MARK) and give the watchdog that long
            expected = float(probe.item()) * (settle_steps + args.warmup + args.steps) * 1.25 + 5.0
            print(f"[bench rank {rank}] expects {expected:.1f} s until the headline ({float(probe.item()) * 1e3:.3f} ms per step by two probe steps)", file=sys.stderr, flush=True)
            if headline_watchdog is not None and 2.0 * expected + 30.0 > args.headline_timeout - (time.monotonic() - started_at):
                headline_watchdog.cancel()
                headline_watchdog = threading.Timer(2.0 * expected + 30.0, stuck)
                headline_watchdog.daemon = True
                headline_watchdog.start()
        for k in range(settle_steps):
            step()
            if k % 16 == 15:  # (keep the host a bounded distance ahead)
                finish()
                torch.cuda.synchronize()
        settle_steps += 2
    enter("warm-up steps")
    for _ in range(args.warmup):
        step()
    fence()
    enter("timed steps")
    chip.start()
    ev0, ev1 = pkg.Event(), pkg.Event()
    t0 = time.perf_counter()
    ev0.record(stream_ptr)
    head = max(1, min(args.steps, 8))  # the host's own enqueue time is read over the first few steps: later the loop may run into a full device queue
    enqueued = t0
    for k in range(args.steps):
        step()
        if k == head - 1:
            enqueued = time.perf_counter()
    finish()
    ev1.record(stream_ptr)
    # The closing bracket: this rank's device work done (synchronize), its clock stopped, THEN the barrier.  The figure reported is the
    # MAX over the ranks of these times -- the moment the last rank finished -- so the host barrier that closes the bracket (gloo: a
    # few hundred microseconds, as much as a step at 8 GPUs) is not part of anybody's K steps.
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    chip.stop()
    ev1.synchronize()
    if headline_watchdog is not None:
        headline_watchdog.cancel()
    enter("after the timed region")
    stream_ms_per_step = ev0.elapsed_ms(ev1) / args.steps  # HIP events on the launch stream: this rank's step, kernels only at N = 1
    # what the HOST needed to enqueue a timed step (this rank's loop, its first 8 steps: 40 steps of an 8-rank step in a torch process run
    # into a full device queue and read 0.63 ms per step where 8 read 0.13; a step whose enqueue takes longer than its kernels is bound by
    # the host).  The library's own figure for the last step stands next to it.
    host_enqueue_ms = (enqueued - t0) / head * 1e3
    host_enqueue_ms_this_rank = host_enqueue_ms  # (the line's figure is the MAX over the ranks; ranks_seen[] keeps each rank's own)
    lib_enqueue_ms = None
    if capi_rank is not None:
        ms = ctypes.c_double(0)
        if lib.nb_comm_last_enqueue_ms(capi_rank.comm, ctypes.byref(ms)) == 0:
            lib_enqueue_ms = ms.value

    if distributed:
        t = torch.tensor([elapsed, host_enqueue_ms], dtype=torch.float64)  # gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, host_enqueue_ms = float(t[0].item()), float(t[1].item())

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
        forces_ms = finish_ms = clock_by_kernel = None
        if pair is not None and not distributed:
            clocked = pair.slices == 1 and pair.waves_per_block == 8 and pair.bodies_per_lane == (16 if not args.fp64 else 8)  # (the geometry pair_forces_clocked exists for)
            forces_ms, finish_ms, clock_by_kernel = pair_kernel_split(pkg, lib, step, stream_ptr, grid_blocks=pair.grid_blocks if clocked else 0)
        dominant_ms = forces_ms if forces_ms is not None else stream_ms_per_step
        algorithmic_flops = flops_per * float(n) * float(n) / world  # per launch of the dominant kernel(s) of one rank's step
        achieved_tflops = algorithmic_flops / (dominant_ms * 1e-3) / 1e12
        # HBM traffic and VALU busy cannot be counted from inside this process: they come from the separate rocprofv3 --pmc passes
        # of this same command (tools/profile.sh -> tools/summarize_prof.py), committed under profiles/ -- used only where the
        # summary was taken with THIS launch plan (counters of another geometry say nothing about this run).
        plan_now = plan_dict(pkg, n, dtype, "pairwise" if pair is not None else "one-sided", world)
        traffic = traffic_src = valu_busy = wasted = None
        if world == 1:
            pmc = pmc_summary(n, args.fp64, args.mode, "pairwise" if pairwise else "one-sided", plan_now)
            if pmc is not None:
                traffic, traffic_src, valu_busy = pmc.get("hbm_bytes_per_launch"), pmc["source"], pmc.get("valu_busy")
            if traffic is not None:
                # every kernel of the step: the pairwise layout's second kernel reads back what the first one stored
                finish_pmc = pmc_summary(n, args.fp64, args.mode, "pairwise", plan_now, kernel="_finish") if pair is not None else None
                step_bytes = traffic + ((finish_pmc or {}).get("hbm_bytes_per_launch") or 0.0)
                wasted = {"ratio": round(step_bytes / ((128 if args.fp64 else 64) * n), 1), "hbm_bytes_per_step": step_bytes,
                          "kernels_counted": ["pair_forces", "pair_finish"] if (finish_pmc or {}).get("hbm_bytes_per_launch") else ["the dominant kernel"],
                          "what": "HBM bytes the step's kernels really move (PMC passes) / SURVEY 8(d)'s algorithmic 64 N (128 N fp64): the pairwise layout "
                                  "writes its reaction slots once and reads them once; ~0.1-0.2 ms of the step, not its limiter"}
        executed = None
        per = 36 if args.fp64 else 24
        if pair is not None:
            evals = pair_evaluations(pair)
            executed = {"pair_evaluations_per_launch": evals, "flops_per_pair_evaluation": per,
                        "tflops": per * evals / (dominant_ms * 1e-3) / 1e12, "frac": per * evals / (dominant_ms * 1e-3) / 1e12 / peak}
        elif capi_rank is not None and pairwise and world > 1:
            # pairwise ACROSS the ranks: what THIS rank (rank 0) evaluates per step -- its diagonal and its rectangles against ranks
            # r+1 .. r+G/2 (nb_comm_pair_work_*, from the plan the step runs) -- over its own step time, all kernels and waits included
            work = capi_rank.pair_work()
            if work is not None:
                evals, launches = work
                executed = {"pair_evaluations_per_step_this_rank": evals, "force_launches_per_step": launches, "flops_per_pair_evaluation": per,
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
            "traffic": traffic,  # HBM bytes per launch of the dominant kernel, rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes (tools/profile.sh)
            "traffic_source": traffic_src,
            "wasted_traffic_ratio": wasted,
            "valu_busy": valu_busy,  # SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES of the same passes: the hardware's own utilisation figure
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
        # The clock the chip ran at while it was timed (rank 0's card): the peak above assumes 2.4 GHz; under this load the socket
        # sits near its power cap and the power management grants 2.0-2.3 GHz by box.  frac_at_delivered_clock = step_frac x 2400 /
        # sclk: the timed steps' algorithmic flop over what the vector ALUs could issue AT THE CLOCK SAMPLED WHILE THEY RAN (step_frac,
        # not frac: the dominant kernel is timed apart, after the region the samples come from).  Never the headline fraction.
        if args.fp64:
            roofline.update(FP64_PEAK_NOTE)
        roofline["chip"] = chip.summary()
        if roofline["chip"] and roofline["chip"]["sclk_mhz"]:
            roofline["frac_at_hwmon_clock"] = roofline["step_frac"] * ChipWatch.PEAK_CLOCK_MHZ / roofline["chip"]["sclk_mhz"]
            if world == 1 and roofline["chip"]["socket_power_w"]:
                roofline["chip"]["interactions_per_joule"] = value / roofline["chip"]["socket_power_w"]
        chip_summary = roofline["chip"]
        # Round 6: the clock read INSIDE the kernel (pair_forces_clocked, after the timed region), not the power management's table: hwmon's
        # figure reads higher than what a dense vector kernel gets and lags behind a changing load, and cycles per step computed from it
        # differed by 4 % between boxes for one binary (round-5 review).  frac_at_delivered_clock and the cycle counts use THIS clock.
        if clock_by_kernel is not None:
            roofline["chip"] = dict(roofline["chip"] or {}, delivered_mhz_by_kernel=clock_by_kernel["mhz"], delivered_clock=clock_by_kernel)
            roofline["frac_at_delivered_clock"] = roofline["step_frac"] * ChipWatch.PEAK_CLOCK_MHZ / clock_by_kernel["mhz"]
            roofline["mcycles_per_step"] = stream_ms_per_step * clock_by_kernel["mhz"] * 1e-3  # (stream time of a timed step x the delivered clock)
            if forces_ms is not None:
                roofline["pair_forces_mcycles"] = forces_ms * clock_by_kernel["mhz"] * 1e-3
        elif roofline.get("frac_at_hwmon_clock") is not None:
            roofline["frac_at_delivered_clock"] = roofline["frac_at_hwmon_clock"]  # (no probe in this run: the power management's figure stands in)
        line = {
            "metric": "body-body interactions/s, all-pairs N-body step (reference convention N^2 per step)",
            "value": value,
            "unit": "interactions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_steps,  # untimed steps before the warm-up steps (--settle-seconds: the card's clock ramp from idle)
            "ms_per_step": elapsed / args.steps * 1e3,
            # host wall clock to ENQUEUE a timed step (max over the ranks; last_step_by_the_library: nb_comm_last_enqueue_ms of rank 0's last step)
            "host_enqueue_ms_per_step": host_enqueue_ms,
            "host_enqueue": {"ms_per_step": host_enqueue_ms, "last_step_by_the_library_ms": lib_enqueue_ms,
                             "what": "perf_counter around the first 8 step calls of the timed loop, before anything is waited for (further on a launch may block on a full device queue); "
                                     "one process per GPU: each rank enqueues its own step"},
            "higher_is_better": True,
            "scaling": "strong",
            # true when this line was NOT produced by the exchange asked for (the launcher's retries, or a collective in-rank
            # fallback at bring-up): such a value must not pass for a number of the C-ABI RCCL path
            "exchange_fallback": bool(exchange_fallback),
            # WHICH exchange produced `value`, in one word a reader of the parsed line cannot miss: "rccl-tiles" is the product's own path
            # (nb_comm_init_rank + nb_sharded_step_*); everything else is a fall-back or an A/B and must not be credited as that path
            "exchange_path": "none" if world == 1 else ("rccl-tiles" if capi_rank is not None else
                                                         "staged" if args.exchange in ("staged", "host", "host-tiles") else ("torch" if system.exchange == "tiles" else "allgather")),
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
        print(f"[bench rank {rank}] headline timed: post-headline measurements start", file=sys.stderr, flush=True)

        def stalled():
            emit({"diagnostics_incomplete": f"the post-headline measurements did not finish within {args.diagnostics_timeout:.0f} s"})
            sys.stderr.write(f"[bench rank {rank}] post-headline measurements stalled: leaving\n")
            sys.stderr.flush()
            os._exit(0)  # (the headline is valid and, on rank 0, out: not a failure of the job)

        # rank 0 first: a launcher that sees another rank leave may end the ones that are left -- the line must be out by then
        watchdog = threading.Timer(args.diagnostics_timeout + (0.0 if rank == 0 else 5.0), stalled)
        watchdog.daemon = True
        watchdog.start()
        extra = {}
        try:
            seen = [None] * world
            clock = chip_summary if rank == 0 else chip.summary()  # (this rank's card while the headline was timed)
            clock = None if clock is None else {k: clock[k] for k in ("sclk_mhz", "sclk_mhz_min", "socket_power_w", "samples")}
            # what every rank's communicator says about itself, the RCCL it is bound to (version, file), the card it runs on (PCI address) and
            # -- pairwise across the ranks -- the pair evaluations and force launches of its step, and its step time by HIP events
            mine = {"rank": rank, "path": "sharded.py", "chip": clock}
            if capi_rank is not None:
                mine = dict(capi_rank.info(), pairwise=capi_rank.pairwise(), one_group=capi_rank.exchange_grouping(), workspace_bytes=work_bytes,
                            cuda_device=torch.cuda.current_device(), chip=clock)
                work = capi_rank.pair_work() if (capi_rank.pairwise() and world > 1) else None
                mine["pair_work"] = None if work is None else {"pair_evaluations_per_step": work[0], "force_launches_per_step": work[1]}
            mine["pci"] = pci_address(torch, local_rank)
            mine["stream_ms_per_step"] = float(f"{stream_ms_per_step:.5g}")
            mine["host_enqueue_ms_per_step"] = float(f"{host_enqueue_ms_this_rank:.5g}")
            dist.all_gather_object(seen, mine)
            extra["ranks_seen"] = seen
            # how many ranks stepped through a communicator bound to a real RCCL (== n_gpus for a line of the native path), and which RCCL
            with_rccl = [r for r in seen if isinstance(r, dict) and (r.get("rccl_version") or 0) > 0 and "fake" not in (r.get("rccl_library") or "")]
            extra["rccl_ranks_seen"] = len(with_rccl)
            extra["rccl_version"] = with_rccl[0]["rccl_version"] if with_rccl else None
            extra["defaults_provisional"] = ("the multi-GPU defaults (an RCCL group per position round + one tail group; the diagonal as two launches, the second one last; reaction "
                                             "rounds in ready order) were chosen on ONE GPU with a loopback rank -- no link latency, no peer to be late; diagnostics.step_ms times "
                                             "the alternatives in this very job: choose from them")
            if not args.no_diagnostics and args.mode == "fast":
                # the N = 1 figure of THIS process tree: rank 0 steps the whole system alone on its GPU while the others wait at the
                # barrier -- "does N = 1 of a scaling series agree with the single-GPU bench" is then answerable from this one record
                fence()
                try:
                    extra["single_gpu_same_process"] = single_gpu_reference(pkg, n, dtype, mode, pos0, vel0) if rank == 0 else None
                except Exception as exc:  # noqa: BLE001
                    extra["single_gpu_same_process"] = {"error": repr(exc)}
                fence()
                if rank == 0 and isinstance(extra["single_gpu_same_process"], dict) and extra["single_gpu_same_process"].get("value"):
                    extra["single_gpu_same_process"]["speedup_of_this_line"] = round(value / extra["single_gpu_same_process"]["value"], 3)
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

    if world > 1:
        # the line is out; taking the communicators down is collective work too (ncclCommDestroy, gloo's tear-down): a rank that does not
        # come back from it leaves after 30 s with status 0 rather than keep the whole job until the caller's limit
        def leave():
            sys.stderr.write(f"[bench rank {rank}] tear-down did not finish within 30 s: leaving\n")
            sys.stderr.flush()
            os._exit(0)

        last = threading.Timer(30.0, leave)
        last.daemon = True
        last.start()
    if capi_rank is not None:
        torch.cuda.synchronize()
        capi_rank.destroy()
        if own_stream is not None:
            lib.nb_stream_destroy(own_stream)
    if distributed:
        dist.destroy_process_group()
    if rccl_init_hung:
        # A daemon thread is still blocked inside ncclCommInitRank (or ncclCommDestroy).  A normal interpreter shutdown would run the
        # static destructors of HIP and RCCL under it -- a hang or an abort AFTER a valid line, and the 30 s `leave` timer is a daemon thread
        # that no longer fires once finalisation has begun.  The line is out and the process group is gone: leave now.
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)



if __name__ == "__main__":
    main()
