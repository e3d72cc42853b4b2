"""The REAL RCCL carrying REAL step data, checked against one GPU (round 5).

A loopback rank (nb_comm_loopback_open: rank r of a nominal G-rank communicator, every transfer to itself) normally computes
garbage -- what "arrives" is its own data.  Unless the system is PERIODIC IN THE SLICES: G identical copies of one slice of bodies
(copies of a body sit on top of each other: zero distance, zero force between them, the softening keeps it finite).  Then by
symmetry every rank of a real G-rank run holds, at every step, exactly the slice this rank holds, so the tile that "arrives from
rank r+s" IS this rank's own slice, and -- for an ODD G, where no rectangle is split between two partners -- the reaction sums
that "arrive from rank r-s" ARE the ones this rank computed for rank r+s.  The loopback step is then the true step of rank r,
its bytes really travelling through ncclSend / ncclRecv, and its slice can be held to a single-GPU run of the whole system:

    STRICT (one-sided tiles, ascending j)    == nb_integrate_f32 STRICT on one GPU, bit for bit
    FAST one-sided tiles / FAST pairwise     == nb_integrate_ws_f32 on one GPU to summation-order accuracy

    python3 tools/rccl_loopback_parity.py [--world 5] [--slice 4096] [--steps 3] [--torch]

Prints one JSON line; exit status 1 when a comparison fails."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=5)
    ap.add_argument("--slice", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--torch", action="store_true")
    args = ap.parse_args()
    assert args.world % 2 == 1 and args.world >= 3, "an odd world: no rectangle is split between two partners"
    if args.torch:
        import torch  # noqa: F401
    import __graft_entry__ as entry

    pkg = entry.load_package()
    pkg.use_lab()  # (the lab library: include/nbody_hip_lab.h)
    lib = pkg.lib()
    oracle = entry.load_oracle().Oracle()
    pkg.check(lib.nb_set_device(0), "nb_set_device")
    G, ni, r = args.world, args.slice, args.world // 2
    n = G * ni
    p1, v1 = oracle.startup_state(ni, np.float32)
    pos0, vel0 = np.tile(p1.reshape(ni, 4), (G, 1)).ravel().copy(), np.tile(v1.reshape(ni, 4), (G, 1)).ravel().copy()
    dt, damping = np.float32(0.016), np.float32(1.0)
    pkg.check(lib.nb_set_softening_sq_f32(np.float32(0.1) * np.float32(0.1)))
    pkg.check(lib.nb_comm_set_pair_min_slice(64))

    def one_gpu(mode, workspace):
        system = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float32, pos0, vel0, mode=mode, workspace=workspace)
        for _ in range(args.steps):
            system.update(dt)
        p, v = system.get_position().copy(), system.get_velocity().copy()
        system.free()
        return p.reshape(n, 4)[r * ni:(r + 1) * ni], v.reshape(n, 4)[r * ni:(r + 1) * ni]

    comm = ctypes.c_void_p()
    pkg.check(lib.nb_comm_loopback_open(ctypes.byref(comm), pkg.comm_unique_id(), G, r), "nb_comm_loopback_open")
    stream = ctypes.c_void_p()
    pkg.check(lib.nb_comm_stream_create(comm, ctypes.byref(stream)), "nb_comm_stream_create")

    def loopback(mode, workspace):
        bufs = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(4)]
        bufs[0].upload(pos0), bufs[1].upload(pos0), bufs[2].upload(vel0)
        job = pkg.ShardedRank(None, G, r, [bufs[0].ptr.value, bufs[1].ptr.value], bufs[2].ptr.value, bufs[3].ptr.value, n, np.float32, mode, 256, stream, comm=comm)
        need = job.workspace_bytes() if workspace else 0
        work = pkg.DeviceBuffer(need) if need else None
        if work is not None:
            pkg.check(lib.nb_memset(work.ptr, 0xFF, need, None))  # NaN patterns: whatever is read must have been written (or received)
        job.set_workspace(work.ptr if work is not None else None, need)
        assert job.pairwise() == bool(need)
        for _ in range(args.steps):
            job.update(dt, damping)
        job.finish()
        pkg.check(lib.nb_stream_synchronize(stream))
        p = bufs[job.read].download(np.zeros_like(pos0)).reshape(n, 4)
        v = bufs[2].download(np.zeros_like(vel0)).reshape(n, 4)[r * ni:(r + 1) * ni].copy()
        # every slot of the position array holds the slice (the tiles that "arrived" are this rank's own)
        whole = bool(all(p[k * ni:(k + 1) * ni].tobytes() == p[r * ni:(r + 1) * ni].tobytes() for k in range(G)))
        own = p[r * ni:(r + 1) * ni].copy()
        job.set_workspace(None, 0)
        pkg.check(lib.nb_device_synchronize())
        for b in bufs + ([work] if work is not None else []):
            b.free()
        return own, v, whole

    out = {"nominal_world": G, "nominal_rank": r, "bodies": n, "steps": args.steps, **pkg.comm_transport_info(comm)}
    ok = True
    sp, sv = one_gpu(pkg.NB_MODE_STRICT, False)
    lp, lv, whole = loopback(pkg.NB_MODE_STRICT, False)
    out["strict_bitwise"] = bool(lp.tobytes() == sp.tobytes() and lv.tobytes() == sv.tobytes())
    out["strict_every_tile_arrived"] = whole
    ok = ok and out["strict_bitwise"] and whole
    fp, fv = one_gpu(pkg.NB_MODE_FAST, True)
    scale = float(np.abs(fp[:, :3]).max())
    for name, workspace in (("fast_one_sided", False), ("fast_pairwise", True)):
        lp, lv, whole = loopback(pkg.NB_MODE_FAST, workspace)
        err = float(np.abs(lp[:, :3].astype(np.float64) - fp[:, :3]).max() / scale)
        out[name + "_max_err_rel_to_size"] = err
        out[name + "_every_tile_arrived"] = whole
        out[name + "_finite"] = bool(np.isfinite(lp).all() and np.isfinite(lv).all())
        # (two FAST summation orders on a system that collapses: any difference grows ~10x per step at BASELINE sizes, docs/history.md
        # section 5 "FAST accuracy" -- the bar is 2e-5 of the system's size up to 65 536 bodies, 5e-4 beyond)
        ok = ok and err < (2e-5 if n <= 65536 else 5e-4) and whole and out[name + "_finite"]
        ok = ok and lp.tobytes() != sp.tobytes()  # (it is not the STRICT path again)
    out["ok"] = bool(ok)
    print(json.dumps(out), flush=True)
    pkg.check(lib.nb_stream_destroy(stream))
    pkg.check(lib.nb_comm_destroy(comm))
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
