"""What the exchange of a multi-GPU step costs NEXT TO the kernels that ship, measured on one GPU with the real RCCL (round 5).

A LOOPBACK rank (nb_comm_loopback_open, lab header) is rank r of a nominal G-rank communicator whose RCCL communicator has one
rank: nb_sharded_step_f32 on it launches exactly the kernels, RCCL groups, events and waits of that rank of a real G-GPU step --
the position tiles and reaction arrays go through ncclSend / ncclRecv to the rank itself, so RCCL's kernels compete with
pair_forces for the chip as they would on a node; only the xGMI transfer time is missing (a local copy instead).  Per system:

    kernels_alone_*      nb_emulate_pair_rank_f32 (pairwise) / the one-sided tile kernels: the same kernels, no communicator, no exchange
    step_*               the loopback step, all G-1 position rounds in ONE RCCL group / a group per round;
                         late2 / late1 / late0 (nb_set_late_diagonal): the diagonal's late half dealt to BOTH streams and the second stream's last rectangle cut (round 6) /
                         the diagonal as two launches, the second one last / one launch, first
    *_one_sided          no workspace lent (the one-sided tile schedule)
    *_exchange_alone     the position exchange / the reaction leg on an otherwise idle chip

    python3 tools/exchange_contention.py [--bodies 262144,1048576] [--world 8] [--rank 4] [--steps 60] [--rounds 5]

One JSON line per world size and system.  Positions are meaningless after the first loopback step; only the time counts."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", default="262144,1048576")
    ap.add_argument("--world", default="8", help="nominal world size(s), e.g. 2,4,8 (a communicator each).  PREFER ONE PER PROCESS: after one communicator has been "
                                                  "destroyed and the next made, the kernels-alone phase has been seen at twice its time (its two streams no longer overlapping)")
    ap.add_argument("--phases", default="", help="comma-separated label prefixes: time only these phases (default: all)")
    ap.add_argument("--rank", type=int, default=-1, help="-1: world / 2")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=5, help="how many times the phases take turns (the median over the rounds is reported)")
    ap.add_argument("--torch", action="store_true", help="import torch first (binds torch's RCCL, as bench.py does)")
    ap.add_argument("--fp64", action="store_true", help="double precision: ncclFloat64 tiles of 32 B per body, reaction arrays of 24 B per body")
    args = ap.parse_args()
    if args.torch:
        import torch  # noqa: F401
    import __graft_entry__ as entry
    from bench_support import make_bodies

    pkg = entry.load_package()
    pkg.use_lab()  # (the lab library: include/nbody_hip_lab.h)
    lib = pkg.lib()
    pkg.check(lib.nb_set_device(0), "nb_set_device")
    dtype = np.float64 if args.fp64 else np.float32
    if args.fp64:
        pkg.check(lib.nb_set_softening_sq_f64(0.01))
        dt, damping = 0.016, 1.0
        emulate_fn, shard_fn = lib.nb_emulate_pair_rank_f64, lib.nb_integrate_shard_f64
    else:
        pkg.check(lib.nb_set_softening_sq_f32(np.float32(0.01)))
        dt, damping = np.float32(0.016), np.float32(1.0)
        emulate_fn, shard_fn = lib.nb_emulate_pair_rank_f32, lib.nb_integrate_shard_f32
    stream = ctypes.c_void_p()  # (made per communicator below: nb_comm_stream_create -- a created stream in three is badly placed)
    wanted = [w for w in args.phases.split(",") if w]

    host_ms = {}

    def timed(fn, reps, after=None, label=None):
        """ms per repetition on the stream (events); host_ms[label]: what the HOST needs to enqueue one repetition (the loop's wall
        clock before anything is waited for) -- a step whose enqueue takes longer than its kernels is bound by the host"""
        import time

        fn()
        (after or (lambda: None))()
        pkg.check(lib.nb_device_synchronize())
        e0, e1 = pkg.Event(), pkg.Event()
        e0.record(stream)
        t0 = time.perf_counter()
        head = min(reps, 8)  # the host's own time is read over the FIRST few repetitions: further on the loop runs into a full device queue
        for k in range(reps):  # (torch's HIP runtime: 40 steps of an 8-rank step read 0.63 ms per step where 8 steps read 0.13 -- round 6)
            fn()
            if k == head - 1:
                t1 = time.perf_counter()
        (after or (lambda: None))()
        e1.record(stream)
        e1.synchronize()
        pkg.check(lib.nb_device_synchronize())
        if label is not None:
            host_ms.setdefault(label, []).append((t1 - t0) / head * 1e3)
        return round(e0.elapsed_ms(e1) / reps, 4)

    for G in [int(x) for x in args.world.split(",")]:
        r = G // 2 if args.rank < 0 else args.rank
        comm = ctypes.c_void_p()
        pkg.check(lib.nb_comm_loopback_open(ctypes.byref(comm), pkg.comm_unique_id(), G, r), "nb_comm_loopback_open")
        if stream:
            pkg.check(lib.nb_stream_destroy(stream))
        pkg.check(lib.nb_comm_stream_create(comm, ctypes.byref(stream)), "nb_comm_stream_create")
        for n in [int(x) for x in args.bodies.split(",")]:
            pos0, vel0 = make_bodies(n, dtype)
            bufs = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(4)]  # pos a, pos b, vel, acc
            bufs[0].upload(pos0), bufs[1].upload(pos0), bufs[2].upload(vel0)
            job = pkg.ShardedRank(None, G, r, [bufs[0].ptr.value, bufs[1].ptr.value], bufs[2].ptr.value, bufs[3].ptr.value, n, dtype, pkg.NB_MODE_FAST, 256, stream, comm=comm)
            need = 0
            for late in (2, 1, 0):  # (the forms of the diagonal want different numbers of planes: lend the largest amount to all)
                pkg.check(lib.nb_set_late_diagonal(late))
                need = max(need, job.workspace_bytes())
            pkg.check(lib.nb_set_late_diagonal(1))
            work = pkg.DeviceBuffer(need) if need else None
            ni = n // G
            size = ctypes.c_size_t(need)

            def emulate():
                pkg.check(emulate_fn(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, work.ptr, ctypes.byref(size), n, G, r, dt, damping, stream), "nb_emulate_pair_rank")

            def tiles():
                for t in range(G):
                    peer = (r + t) % G
                    flags = (pkg.NB_SHARD_ACC_IN if t else 0) | (pkg.NB_SHARD_FINALIZE if t == G - 1 else 0)
                    pkg.check(shard_fn(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, bufs[3].ptr, r * ni, ni, peer * ni, ni, flags, dt, damping, 256, pkg.NB_MODE_FAST, stream), "nb_integrate_shard")

            def configure(layout, one_group, late=1):
                pkg.check(lib.nb_set_late_diagonal(late))
                job.set_workspace(work.ptr if layout == "pairwise" else None, need if layout == "pairwise" else 0)
                assert job.pairwise() == (layout == "pairwise")
                job.set_exchange_grouping(one_group)

            step = lambda: job.update(dt, damping)  # noqa: E731
            # phase: label -> (configure arguments, what one repetition does, what ends the timed stretch)
            phases = {}
            if work is not None:
                for late in (2, 1, 0):
                    for og in (True, False):
                        phases[f"step_pairwise_late{late}_{'one_group' if og else 'group_per_round'}"] = (("pairwise", og, late), step, job.finish)
                    phases[f"kernels_alone_pairwise_late{late}"] = (("pairwise", True, late), emulate, None)
            for og in (True, False):
                phases[f"step_one_sided_{'one_group' if og else 'group_per_round'}"] = (("one_sided", og), step, job.finish)
            phases["kernels_alone_one_sided"] = (("one_sided", True), tiles, None)
            for og in (True, False):
                phases[f"position_exchange_alone_{'one_group' if og else 'group_per_round'}"] = (("one_sided", og), lambda: job.exchange_once(0), None)
            if work is not None:
                phases["reaction_exchange_alone"] = (("pairwise", True), job.reaction_exchange_once, None)
            # the phases take turns, `--rounds` times over: the clock the power management grants drifts over a run, so back-to-back
            # stretches of ONE phase each would compare clocks, not schedules
            if wanted:
                phases = {label: what for label, what in phases.items() if any(label.startswith(w) for w in wanted)}
            samples = {label: [] for label in phases}
            for _ in range(args.rounds):
                for label, (config, fn, after) in phases.items():
                    configure(*config)
                    samples[label].append(timed(fn, args.steps, after, label))
            row = {"dtype": "f64" if args.fp64 else "f32", "bodies": n, "nominal_world": G, "nominal_rank": r, "steps_per_stretch": args.steps, "rounds": args.rounds, "position_tile_bytes": ni * 4 * np.dtype(dtype).itemsize,
                   "reaction_array_bytes": ni * 3 * np.dtype(dtype).itemsize, "workspace_bytes": need, **pkg.comm_transport_info(comm),
                   "what": "ms per repetition: median over the rounds (phases interleaved); *_min: the fastest stretch"}
            row["side_stream_collisions"] = job.info()["side_stream_collisions"]  # candidates for the second compute stream that shared a hardware queue with the first
            for label, got in samples.items():
                got = sorted(got)
                row[label] = got[len(got) // 2]
                row[label + "_min"] = got[0]
                if label.startswith("step_"):  # (the enqueue loop may run into a full queue: the MINIMUM over the stretches is the host's own time)
                    row[label.replace("step_", "host_enqueue_ms_", 1)] = round(min(host_ms[label]), 4)
            print(json.dumps(row), flush=True)
            host_ms.clear()
            configure("one_sided", True)
            pkg.check(lib.nb_device_synchronize())
            for b in bufs + ([work] if work is not None else []):
                b.free()
        pkg.check(lib.nb_comm_destroy(comm))
    pkg.check(lib.nb_stream_destroy(stream))


if __name__ == "__main__":
    main()
