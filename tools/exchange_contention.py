"""What the exchange of a multi-GPU step costs NEXT TO the kernels that ship, measured on one GPU with the real RCCL (round 5).

A LOOPBACK rank (nb_comm_loopback_open, tuning header) is rank r of a nominal G-rank communicator whose RCCL communicator has one
rank: nb_sharded_step_f32 on it launches exactly the kernels, RCCL groups, events and waits of that rank of a real G-GPU step --
the position tiles and reaction arrays go through ncclSend / ncclRecv to the rank itself, so RCCL's kernels compete with
pair_forces for the chip as they would on a node; only the xGMI transfer time is missing (a local copy instead).  Per system:

    kernels_alone_ms     nb_emulate_pair_rank_f32: the same kernels, no communicator, no exchange
    step_ms              the loopback step, all G-1 position rounds in ONE RCCL group / a group per round
    *_one_sided          the same with no workspace lent (the one-sided tile schedule)
    exchange_alone_ms    the position exchange / the reaction leg on an otherwise idle chip

    python3 tools/exchange_contention.py [--bodies 262144,1048576] [--world 8] [--rank 4] [--steps 20] [--reserve 0,8,16]

--reserve k: nb_comm_set_reserved_cus(comm, k) where the library has it (the pairwise launches of a rank leave k CUs to the
exchange).  One JSON line per system and setting.  Positions are meaningless after the first loopback step; only the time counts."""
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
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=-1, help="-1: world / 2")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reserve", default="0")
    ap.add_argument("--torch", action="store_true", help="import torch first (binds torch's RCCL, as bench.py does)")
    args = ap.parse_args()
    if args.torch:
        import torch  # noqa: F401
    import __graft_entry__ as entry
    from bench_support import make_bodies

    pkg = entry.load_package()
    lib = pkg.lib()
    pkg.check(lib.nb_set_device(0), "nb_set_device")
    G, r = args.world, (args.world // 2 if args.rank < 0 else args.rank)
    pkg.check(lib.nb_set_softening_sq_f32(np.float32(0.01)))
    dt, damping = np.float32(0.016), np.float32(1.0)
    comm = ctypes.c_void_p()
    pkg.check(lib.nb_comm_loopback_open(ctypes.byref(comm), pkg.comm_unique_id(), G, r), "nb_comm_loopback_open")
    stream = ctypes.c_void_p()
    pkg.check(lib.nb_stream_create(ctypes.byref(stream)), "nb_stream_create")
    has_reserve = hasattr(lib, "nb_comm_set_reserved_cus")

    def timed(fn, reps, after=None):
        fn()
        (after or (lambda: None))()
        pkg.check(lib.nb_device_synchronize())
        e0, e1 = pkg.Event(), pkg.Event()
        e0.record(stream)
        for _ in range(reps):
            fn()
        (after or (lambda: None))()
        e1.record(stream)
        e1.synchronize()
        pkg.check(lib.nb_device_synchronize())
        return round(e0.elapsed_ms(e1) / reps, 4)

    for n in [int(x) for x in args.bodies.split(",")]:
        pos0, vel0 = make_bodies(n, np.float32)
        bufs = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(4)]  # pos a, pos b, vel, acc
        bufs[0].upload(pos0), bufs[1].upload(pos0), bufs[2].upload(vel0)
        job = pkg.ShardedRank(None, G, r, [bufs[0].ptr.value, bufs[1].ptr.value], bufs[2].ptr.value, bufs[3].ptr.value, n, np.float32, pkg.NB_MODE_FAST, 256, stream, comm=comm)
        need = job.workspace_bytes()
        work = pkg.DeviceBuffer(need) if need else None
        for reserve in [int(x) for x in args.reserve.split(",")]:
            if reserve and not has_reserve:
                continue
            if has_reserve:
                pkg.check(lib.nb_comm_set_reserved_cus(comm, reserve), "nb_comm_set_reserved_cus")
            row = {"bodies": n, "nominal_world": G, "nominal_rank": r, "steps": args.steps, "reserved_cus": reserve, "position_tile_bytes": n // G * 16,
                   "reaction_array_bytes": n // G * 12, "workspace_bytes": need, **pkg.comm_transport_info(comm)}
            step = lambda: job.update(dt, damping)  # noqa: E731
            for layout in (("pairwise", "one_sided") if work is not None else ("one_sided",)):
                job.set_workspace(work.ptr if layout == "pairwise" else None, need if layout == "pairwise" else 0)
                assert job.pairwise() == (layout == "pairwise")
                for one_group in (True, False):
                    job.set_exchange_grouping(one_group)
                    row[f"step_ms_{layout}_{'one_group' if one_group else 'group_per_round'}"] = timed(step, args.steps, job.finish)
                job.set_exchange_grouping(True)
                if layout == "pairwise":
                    row["reaction_exchange_alone_ms"] = timed(job.reaction_exchange_once, args.steps)
                    size = ctypes.c_size_t(need)
                    emulate = lambda: pkg.check(lib.nb_emulate_pair_rank_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, work.ptr, ctypes.byref(size), n, G, r, dt, damping, stream), "nb_emulate_pair_rank")  # noqa: E731
                    row["kernels_alone_ms_pairwise"] = timed(emulate, args.steps)
                else:
                    ni = n // G

                    def tiles():
                        for t in range(G):
                            peer = (r + t) % G
                            flags = (pkg.NB_SHARD_ACC_IN if t else 0) | (pkg.NB_SHARD_FINALIZE if t == G - 1 else 0)
                            pkg.check(lib.nb_integrate_shard_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, bufs[3].ptr, r * ni, ni, peer * ni, ni, flags, dt, damping, 256, pkg.NB_MODE_FAST, stream), "nb_integrate_shard")

                    row["kernels_alone_ms_one_sided"] = timed(tiles, args.steps)
            for one_group in (True, False):
                job.set_exchange_grouping(one_group)
                row[f"position_exchange_alone_ms_{'one_group' if one_group else 'group_per_round'}"] = timed(lambda: job.exchange_once(0), args.steps)
            job.set_exchange_grouping(True)
            for layout in ("pairwise", "one_sided"):
                k = row.get(f"kernels_alone_ms_{layout}")
                if k:
                    for grouping in ("one_group", "group_per_round"):
                        row[f"exposed_ms_{layout}_{grouping}"] = round(row[f"step_ms_{layout}_{grouping}"] - k, 4)
            print(json.dumps(row), flush=True)
        if has_reserve:
            pkg.check(lib.nb_comm_set_reserved_cus(comm, 0))
        job.set_workspace(None, 0)
        pkg.check(lib.nb_device_synchronize())
        for b in bufs + ([work] if work is not None else []):
            b.free()
    pkg.check(lib.nb_stream_destroy(stream))
    pkg.check(lib.nb_comm_destroy(comm))


if __name__ == "__main__":
    main()
