"""Can one rank's whole multi-GPU step -- kernels, events, RCCL send/recv -- be captured into a hipGraph and replayed? (round 6)

Feasibility probe against the REAL RCCL on one GPU, before the library grew nb_sharded_graph_*: the capture is opened from here
(hipStreamBeginCapture on the stream the rank steps on), the step is the product's own nb_sharded_step_* / nb_sharded_step_all_*
followed by nb_exchange_wait_all (which joins the exchange stream back into the capturing stream), then hipStreamEndCapture,
hipGraphInstantiate and replays.  Two worlds:

    loopback   rank G/2 of a nominal G-rank communicator (nb_comm_loopback_open): one rank's step
    inprocess  all G ranks in this process on one GPU sharing one real ncclComm (nb_comm_inprocess_open_all): the whole world's step
               as ONE graph (the other ranks' streams fork from rank 0's)

Per world one JSON line: what the host needs to enqueue one eager step (perf_counter around the calls, nothing waited for), whether
the capture was accepted (the refusing call and its error VERBATIM if not), nodes in the graph, host time per replay, stream time
per step both ways.

    python3 tools/graph_capture_probe.py [--bodies 262144] [--world 8] [--steps 40] [--mode thread_local|relaxed|global]
"""
import argparse
import ctypes
import faulthandler
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CAPTURE_MODES = {"global": 0, "thread_local": 1, "relaxed": 2}


def stage(text):
    print(f"[stage] {text}", file=sys.stderr, flush=True)


def main():
    faulthandler.enable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=262144)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--mode", default="thread_local")
    ap.add_argument("--worlds", default="loopback,inprocess")
    ap.add_argument("--layout", default="pairwise", help="pairwise | one_sided")
    ap.add_argument("--what", default="steps", help="what goes into the capture: steps (two steps) | step1 (one step, no wait for events recorded outside) | exchange (two position exchanges, no kernels)")
    ap.add_argument("--leave", action="store_true", help="os._exit after the line: no tear-down of communicators")
    args = ap.parse_args()
    import __graft_entry__ as entry
    from bench_support import make_bodies

    pkg = entry.load_package()
    pkg.use_lab()  # (the lab library: include/nbody_hip_lab.h)
    lib = pkg.lib()
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipGetErrorString.restype = ctypes.c_char_p
    hip.hipGetErrorString.argtypes = [ctypes.c_int]

    def hip_text(code):
        return f"{code} {hip.hipGetErrorString(code).decode()}"

    pkg.check(lib.nb_set_device(0), "nb_set_device")
    pkg.check(lib.nb_set_softening_sq_f32(np.float32(0.01)))
    n, G = args.bodies, args.world
    dt, damping = np.float32(0.016), np.float32(1.0)
    pos0, vel0 = make_bodies(n, np.float32)
    vp = ctypes.c_void_p

    for world in args.worlds.split(","):
        row = {"what": args.what, "world_kind": world, "bodies": n, "ranks": G, "layout": args.layout, "capture_mode": args.mode}
        uid = ctypes.create_string_buffer(128)
        if world != "double":
            pkg.check(lib.nb_comm_unique_id(uid), "nb_comm_unique_id")
        if world == "loopback":
            comm = vp()
            pkg.check(lib.nb_comm_loopback_open(ctypes.byref(comm), uid, G, G // 2), "nb_comm_loopback_open")
            comms = [comm]
        elif world == "double":
            # G ranks on this one device through the TRANSPORT DOUBLE (NBODY_RCCL_LIB=tests/fake_rccl/libfake_rccl.so): every rank owns a
            # communicator, so the crew steps the ranks independently, a thread each -- the form a real node takes.  The double's
            # ncclGroupEnd blocks the host until the peer has posted, which the real library does not: an upper bound.
            assert os.environ.get("NBODY_RCCL_LIB", "").endswith("libfake_rccl.so"), "world kind `double` needs NBODY_RCCL_LIB=.../libfake_rccl.so"
            arr = (vp * G)()
            pkg.check(lib.nb_comm_init_all(arr, G, (ctypes.c_int * G)(*([0] * G))), "nb_comm_init_all")
            comms = [vp(arr[k]) for k in range(G)]
        else:
            arr = (vp * G)()
            pkg.check(lib.nb_comm_inprocess_open_all(arr, G, uid), "nb_comm_inprocess_open_all")
            comms = [vp(arr[k]) for k in range(G)]
        L = len(comms)
        streams, bufs, works = [], [], []
        for k, c in enumerate(comms):
            s = vp()
            pkg.check(lib.nb_comm_stream_create(c, ctypes.byref(s)), "nb_comm_stream_create")
            streams.append(s)
            b = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(4)]
            b[0].upload(pos0), b[1].upload(pos0), b[2].upload(vel0)
            bufs.append(b)
            need = ctypes.c_size_t(0)
            pkg.check(lib.nb_comm_workspace_bytes_f32(c, n, pkg.NB_MODE_FAST, ctypes.byref(need)))
            if args.layout == "pairwise" and need.value:
                w = pkg.DeviceBuffer(need.value)
                works.append(w)
                pkg.check(lib.nb_comm_set_workspace(c, w.ptr, need.value), "nb_comm_set_workspace")
            else:
                pkg.check(lib.nb_comm_set_workspace(c, None, 0), "nb_comm_set_workspace")
        flag = ctypes.c_int(-1)
        pkg.check(lib.nb_comm_layout_f32(comms[0], n, pkg.NB_MODE_FAST, ctypes.byref(flag)))
        row["pairwise"] = flag.value
        arr_of = lambda xs: (vp * L)(*xs)  # noqa: E731
        comm_arr, stream_arr = arr_of(comms), arr_of(streams)
        read = [0]

        def step():
            rd = read[0]
            if world == "loopback":
                pkg.check(lib.nb_sharded_step_f32(comms[0], bufs[0][1 - rd].ptr, bufs[0][rd].ptr, bufs[0][2].ptr, bufs[0][3].ptr, n, dt, damping, 256, pkg.NB_MODE_FAST, streams[0]), "nb_sharded_step")
            else:
                pkg.check(lib.nb_sharded_step_all_f32(comm_arr, L, arr_of([b[1 - rd].ptr for b in bufs]), arr_of([b[rd].ptr for b in bufs]), arr_of([b[2].ptr for b in bufs]),
                                                      arr_of([b[3].ptr for b in bufs]), n, dt, damping, 256, pkg.NB_MODE_FAST, stream_arr), "nb_sharded_step_all")
            read[0] = 1 - rd

        def wait_all():
            for c, s in zip(comms, streams):
                pkg.check(lib.nb_exchange_wait_all(c, s), "nb_exchange_wait_all")

        def timed(fn, reps):
            """(stream ms per repetition on rank 0's stream, host ms per repetition)"""
            pkg.check(lib.nb_device_synchronize())
            e0, e1 = pkg.Event(), pkg.Event()
            e0.record(streams[0])
            t0 = time.perf_counter()
            head = min(reps, 8)  # (the host's own time: over the first repetitions, before the loop can run into a full device queue)
            for k in range(reps):
                fn()
                if k == head - 1:
                    t1 = time.perf_counter()
            wait_all()
            e1.record(streams[0])
            e1.synchronize()
            pkg.check(lib.nb_device_synchronize())
            return round(e0.elapsed_ms(e1) / reps, 4), round((t1 - t0) / head * 1e3, 4)

        stage(f"{world}: set up, pairwise={flag.value}")
        for _ in range(4):  # warm-up: first-use set-up (LDS opt-in, the stream probes) happens here, outside any capture
            step()
        wait_all()
        pkg.check(lib.nb_device_synchronize())
        eager = [timed(step, args.steps) for _ in range(3)]
        row["eager_stream_ms_per_step"] = min(e[0] for e in eager)
        row["eager_host_enqueue_ms_per_step"] = min(e[1] for e in eager)

        stage(f"eager timed: {eager}")
        ms = ctypes.c_double(0)
        pkg.check(lib.nb_comm_last_enqueue_ms(comms[0], ctypes.byref(ms)))
        row["last_enqueue_ms_by_the_library"] = round(ms.value, 4)
        row["step_threads"] = os.environ.get("NBODY_STEP_THREADS", "1")
        if os.environ.get("NBODY_ENQUEUE_TRACE") == "1":
            text = ctypes.create_string_buffer(8192)
            pkg.check(lib.nb_comm_last_step_trace(comms[0], text, len(text)))
            row["host_phases_ms_last_step"] = {" ".join(line.split()[2:]): round(float(line.split()[1]), 4) for line in text.value.decode().splitlines() if line.startswith("host ")}
        if args.what == "none":  # the eager step only: what the host needs to enqueue it (NBODY_STEP_THREADS=0 / 1)
            print(json.dumps(row), flush=True)
            sys.stdout.flush()
            os._exit(0)
        # --- capture two steps (a -> b, b -> a) + the join, from rank 0's stream
        graph, execg = vp(), vp()
        refusal = None
        pkg.check(lib.nb_device_synchronize())
        if args.what == "step1" and world == "loopback":
            # the tiles in flight become those of the array the captured step WRITES: the step then waits for no event recorded outside the capture
            pkg.check(lib.nb_exchange_tiles_f32(comms[0], bufs[0][1 - read[0]].ptr, n, streams[0]), "nb_exchange_tiles")
            wait_all()
            pkg.check(lib.nb_device_synchronize())
        rc = hip.hipStreamBeginCapture(streams[0], CAPTURE_MODES[args.mode])
        stage(f"hipStreamBeginCapture -> {rc}")
        if rc != 0:
            refusal = f"hipStreamBeginCapture: {hip_text(rc)}"
        else:
            fork = pkg.Event()
            try:
                if L > 1:  # the other ranks' streams join the capture
                    fork.record(streams[0])
                    for s in streams[1:]:
                        pkg.check(lib.nb_stream_wait_event(s, fork.h))
                if args.what == "exchange":
                    for which in (0, 1):
                        for c, s, b in zip(comms, streams, bufs):
                            pkg.check(lib.nb_exchange_tiles_f32(c, b[which].ptr, n, s), "nb_exchange_tiles")
                        wait_all()
                    stage("two exchanges captured")
                else:
                    step()
                    stage("first step captured")
                    wait_all()
                    stage("first join captured")
                    if args.what == "steps":
                        step()
                        wait_all()
                        stage("second step captured")
                if L > 1:
                    joins = [pkg.Event() for _ in streams[1:]]
                    for e, s in zip(joins, streams[1:]):
                        e.record(s)
                        pkg.check(lib.nb_stream_wait_event(streams[0], e.h))
            except Exception as exc:  # noqa: BLE001 -- the refusal is the result
                refusal = f"inside the capture: {exc}"
            rc = hip.hipStreamEndCapture(streams[0], ctypes.byref(graph))
            stage(f"hipStreamEndCapture -> {rc}")
            if rc != 0 and refusal is None:
                refusal = f"hipStreamEndCapture: {hip_text(rc)}"
        if refusal is None:
            count = ctypes.c_size_t(0)
            hip.hipGraphGetNodes(graph, None, ctypes.byref(count))
            row["graph_nodes_two_steps"] = count.value
            rc = hip.hipGraphInstantiate(ctypes.byref(execg), graph, None, None, ctypes.c_size_t(0))
            if rc != 0:
                refusal = f"hipGraphInstantiate: {hip_text(rc)}"
        row["capture_accepted"] = refusal is None
        row["refusal"] = refusal
        if refusal is None:
            def replay():
                rc = hip.hipGraphLaunch(execg, streams[0])
                if rc != 0:
                    raise RuntimeError(f"hipGraphLaunch: {hip_text(rc)}")

            replay()
            pkg.check(lib.nb_device_synchronize())
            got = [timed(replay, args.steps // 2) for _ in range(3)]
            per = 1 if args.what == "step1" else 2
            row["graph_stream_ms_per_step"] = round(min(g[0] for g in got) / per, 4)
            row["graph_host_enqueue_ms_per_step"] = round(min(g[1] for g in got) / per, 4)
        else:
            pass
        print(json.dumps(row), flush=True)
        if refusal is None:
            pkg.check(lib.nb_device_synchronize())
            stage(f"hipGraphExecDestroy -> {hip.hipGraphExecDestroy(execg)}, hipGraphDestroy -> {hip.hipGraphDestroy(graph)}")
        if args.leave:
            sys.stdout.flush()
            os._exit(0)
        if refusal is not None:
            # a refused capture may leave streams in an invalidated capture: leave the process to the OS rather than tearing RCCL down
            sys.stdout.flush()
            os._exit(0)
        pkg.check(lib.nb_device_synchronize())
        for c in reversed(comms):
            pkg.check(lib.nb_comm_destroy(c))
        for s in streams:
            pkg.check(lib.nb_stream_destroy(s))
        for b in bufs:
            for x in b:
                x.free()
        for w in works:
            w.free()


if __name__ == "__main__":
    main()
