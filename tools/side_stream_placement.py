"""How much does the PLACEMENT of a rank's second compute stream matter?  (round 5)

The HIP runtime maps streams onto hardware queues, the driver maps those onto the pipes of the command processor.  A loopback rank
(rank 4 of a nominal 8, 262 144 bodies, real RCCL) steps with a dozen second streams in turn (nb_comm_replace_side_stream), with the
caller computing on a created stream and on the null stream: ms per step for each pairing.  One JSON line per pairing."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--candidates", type=int, default=10)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--torch", action="store_true")
    ap.add_argument("--null-follows", action="store_true", help="after every step the NULL stream waits for an event of the stream stepped on (what a caller does who "
                                                                  "times or copies on the null stream): does a marker on the null stream's queue slow the step?")
    ap.add_argument("--placed", action="store_true", help="one more caller: a stream from nb_comm_stream_create")
    args = ap.parse_args()
    if args.torch:
        import torch  # noqa: F401
    import __graft_entry__ as entry
    from bench_support import make_bodies

    pkg = entry.load_package()
    pkg.use_lab()  # (the lab library: include/nbody_hip_lab.h)
    lib = pkg.lib()
    pkg.check(lib.nb_set_device(0))
    n, G, r = 262144, 8, 4
    pkg.check(lib.nb_set_softening_sq_f32(np.float32(0.01)))
    dt, damping = np.float32(0.016), np.float32(1.0)
    comm = ctypes.c_void_p()
    pkg.check(lib.nb_comm_loopback_open(ctypes.byref(comm), pkg.comm_unique_id(), G, r), "nb_comm_loopback_open")
    created = ctypes.c_void_p()
    pkg.check(lib.nb_stream_create(ctypes.byref(created)))
    pos0, vel0 = make_bodies(n, np.float32)
    bufs = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(4)]
    bufs[0].upload(pos0), bufs[1].upload(pos0), bufs[2].upload(vel0)
    others = []
    for _ in range(3):  # (more callers' streams: the CALLER's stream has a placement too)
        s = ctypes.c_void_p()
        pkg.check(lib.nb_stream_create(ctypes.byref(s)))
        others.append(s)
    callers = [("created stream", created)] + [(f"created stream {k + 2}", s) for k, s in enumerate(others)] + [("null stream", None)]
    if args.placed:
        placed = ctypes.c_void_p()
        pkg.check(lib.nb_comm_stream_create(comm, ctypes.byref(placed)), "nb_comm_stream_create")
        callers.append(("nb_comm_stream_create", placed))
    follow = pkg.Event()

    def one_step(job, stream):
        job.update(dt, damping)
        if args.null_follows and stream is not None:
            follow.record(stream)
            pkg.check(lib.nb_stream_wait_event(None, follow.h), "nb_stream_wait_event")

    for name, stream in callers:
        job = pkg.ShardedRank(None, G, r, [bufs[0].ptr.value, bufs[1].ptr.value], bufs[2].ptr.value, bufs[3].ptr.value, n, np.float32, pkg.NB_MODE_FAST, 256, stream, comm=comm)
        need = job.workspace_bytes()
        work = pkg.DeviceBuffer(need)
        job.set_workspace(work.ptr, need)
        for k in range(args.candidates):
            if k:
                pkg.check(lib.nb_comm_replace_side_stream(comm), "nb_comm_replace_side_stream")
            times = []
            for _ in range(3):
                one_step(job, stream)
                job.finish()
                pkg.check(lib.nb_device_synchronize())
                e0, e1 = pkg.Event(), pkg.Event()
                e0.record(stream)
                for _ in range(args.steps):
                    one_step(job, stream)
                job.finish()
                e1.record(stream)
                e1.synchronize()
                times.append(round(e0.elapsed_ms(e1) / args.steps, 4))
            print(json.dumps({"caller_computes_on": name, "candidate": k, "ms_per_step": sorted(times)[1], "all": times, "collisions_replaced_by_the_probe": job.info()["side_stream_collisions"], "caller_stream_badly_placed": job.info()["caller_stream_badly_placed"]}), flush=True)
        job.set_workspace(None, 0)
        pkg.check(lib.nb_device_synchronize())
        work.free()
    pkg.check(lib.nb_comm_destroy(comm))


if __name__ == "__main__":
    main()
