"""The REAL RCCL under the product's multi-GPU layer, on one GPU (round 5): nb_comm_unique_id + nb_comm_selftest_open make a
communicator of one rank that owns a real ncclComm; nb_comm_selftest_f32 sends a known pattern to itself through
GroupStart / ncclSend / ncclRecv / GroupEnd on the communicator's exchange stream with the ready / arrived events of
exchange_tiles, then ncclAllGather out of place and in place, and compares every byte (csrc/nbody_comm.hip).

    python3 tools/rccl_selfloop.py [--torch] [--bytes 393216,2097152,16777216]

--torch imports torch first (the RCCL that gets bound is then torch's own copy, as in bench.py).  Run it in a child process under
a timeout: RCCL is a third-party library this code had never called before round 5.  Prints one JSON line per size."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--torch", action="store_true")
    ap.add_argument("--bytes", default="393216,2097152,16777216")
    args = ap.parse_args()
    if args.torch:
        import torch  # noqa: F401  (first: its HIP runtime and RCCL are the ones the library then binds)
    import __graft_entry__ as entry

    pkg = entry.load_package()
    pkg.use_lab()  # (the lab library: include/nbody_hip_lab.h)
    lib = pkg.lib()
    pkg.check(lib.nb_set_device(0), "nb_set_device")
    t0 = time.perf_counter()
    uid = pkg.comm_unique_id()
    comm = ctypes.c_void_p()
    pkg.check(lib.nb_comm_selftest_open(ctypes.byref(comm), uid), "nb_comm_selftest_open")
    print(json.dumps({"what": "communicator of one rank with a real ncclComm (nb_comm_unique_id + nb_comm_selftest_open)", "seconds": round(time.perf_counter() - t0, 3),
                      "torch_imported_first": args.torch, **pkg.comm_transport_info(comm)}), flush=True)
    stream = ctypes.c_void_p()
    pkg.check(lib.nb_stream_create(ctypes.byref(stream)), "nb_stream_create")
    failed = 0
    for nbytes in [int(x) for x in args.bytes.split(",")]:
        for rep in range(2):  # (the first call of a size pays RCCL's lazy set-up)
            report = pkg.CommSelftest()
            rc = lib.nb_comm_selftest_f32(comm, nbytes, stream, ctypes.byref(report))
            print(json.dumps({"bytes": nbytes, "call": rep, "rc": rc, "rc_text": lib.nb_error_string(rc).decode(), "rccl_version": report.rccl_version,
                              "send_recv_status": report.send_recv_status, "all_gather_status": report.all_gather_status,
                              "send_recv_ms": round(report.send_recv_ms, 4), "all_gather_ms": round(report.all_gather_ms, 4),
                              "send_recv_wrong_bytes": report.send_recv_wrong_bytes, "all_gather_wrong_bytes": report.all_gather_wrong_bytes,
                              "refused_call": report.refused_call.decode(), "library": report.library_path.decode()}), flush=True)
            failed += rc != 0
    pkg.check(lib.nb_stream_destroy(stream), "nb_stream_destroy")
    pkg.check(lib.nb_comm_destroy(comm), "nb_comm_destroy")
    print(json.dumps({"what": "communicator destroyed", "failed_calls": failed}), flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    raise SystemExit(main())
