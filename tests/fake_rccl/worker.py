"""Worker process of tests/test_comm_fake_rccl.py: drives the product's multi-GPU C-ABI (nb_comm_*, nb_sharded_step_*,
nb_exchange_*, nb_allgather_*) with G logical ranks that all live on device 0, RCCL replaced by the test double
tests/fake_rccl/libfake_rccl.so through NBODY_RCCL_LIB (set by the parent; it must be in the environment before the
library resolves RCCL, which happens once per process -- hence a process of its own).

    python worker.py <case> <in.npz> <out.npz> [G] [steps] [mode] [streams] [ws]

`ws`: every rank is lent the workspace nb_comm_workspace_bytes_* asks for (FAST then takes the pairwise step across the ranks).
Environment: WORKER_ONE_GROUP=0|1 -> nb_comm_set_exchange_grouping on every communicator (unset: the library's default);
WORKER_LATE_DIAGONAL=0|1|2 -> nb_set_late_diagonal; WORKER_REAL_RCCL=1 -> the `all` case over the REAL RCCL (an in-process world: libnbody_hip_lab.so);
WORKER_NO_WORKSPACE_RANK=k -> rank k lends nothing (nb_comm_set_workspace(comm, NULL, 0)): the layout must then be one-sided on
EVERY rank.  The output holds `layout` (nb_comm_layout_* per rank) and `setup_counters` (what the transport had seen before the
first step: nb_comm_set_workspace's exchange with one process -- here: thread -- per rank).

Everything here goes through ctypes into libnbody_hip.so; nothing is computed in Python.  The parent compares the arrays
written to <out.npz> with the CPU oracle / the golden fixtures.
"""
from __future__ import annotations

import ctypes
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


REAL_RCCL = os.environ.get("WORKER_REAL_RCCL") == "1"  # the `all` case as an IN-PROCESS world over the REAL library (nb_comm_inprocess_open_all)


def counters():
    if REAL_RCCL:  # (the real library keeps no such counters)
        return dict.fromkeys(("sends", "recvs", "allgathers", "groups", "copies"), 0)
    fake = ctypes.CDLL(os.environ["NBODY_RCCL_LIB"])
    vals = [ctypes.c_long(0) for _ in range(5)]
    fake.fake_rccl_counters(*[ctypes.byref(v) for v in vals])
    return dict(zip(("sends", "recvs", "allgathers", "groups", "copies"), (v.value for v in vals)))


class Rank:
    """Full-size arrays of one logical rank (positions a/b, velocities, partial accelerations) + its compute stream."""

    def __init__(self, pkg, pos0, vel0, own_stream):
        self.pkg, self.lib = pkg, pkg.lib()
        self.bufs = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(4)]
        self.bufs[0].upload(pos0), self.bufs[2].upload(vel0)
        self.stream = ctypes.c_void_p()
        if own_stream:
            pkg.check(self.lib.nb_stream_create(ctypes.byref(self.stream)), "nb_stream_create")
        self.read = 0

    def results(self, comm, like_pos, like_vel):
        self.pkg.check(self.lib.nb_exchange_wait_all(comm, self.stream), "nb_exchange_wait_all")
        self.pkg.check(self.lib.nb_stream_synchronize(self.stream), "nb_stream_synchronize")
        return self.bufs[self.read].download(np.zeros_like(like_pos)).copy(), self.bufs[2].download(np.zeros_like(like_vel)).copy()


def main():
    case, src, dst = sys.argv[1:4]
    G = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    mode_name = sys.argv[6] if len(sys.argv) > 6 else "strict"
    own_streams = (sys.argv[7] if len(sys.argv) > 7 else "streams") == "streams"
    with_workspace = len(sys.argv) > 8 and sys.argv[8] == "ws"
    assert REAL_RCCL or os.environ.get("NBODY_RCCL_LIB", "").endswith("libfake_rccl.so"), "the parent must point NBODY_RCCL_LIB at the test double"
    assert not (REAL_RCCL and "NBODY_RCCL_LIB" in os.environ), "WORKER_REAL_RCCL=1 binds the real library"

    pkg = entry.load_package()
    if REAL_RCCL:  # nb_comm_inprocess_open_all is the lab library's (include/nbody_hip_lab.h); every other case runs on libnbody_hip.so
        pkg.use_lab()
    lib = pkg.lib()
    pkg.check(lib.nb_set_device(0), "nb_set_device")
    data = np.load(src, allow_pickle=False)
    pos0, vel0 = data["pos"], data["vel"]
    dtype = pos0.dtype
    f32 = dtype == np.float32
    n = pos0.size // 4
    ni = n // G
    dt = dtype.type(np.float32(0.016))
    one = dtype.type(1)
    mode = pkg.NB_MODE_STRICT if mode_name == "strict" else pkg.NB_MODE_FAST
    soft = dtype.type(np.float32(0.1))
    if f32:
        pkg.check(lib.nb_set_softening_sq_f32(np.float32(soft * soft)))
    else:
        pkg.check(lib.nb_set_softening_sq_f64(float(soft * soft)))
    if with_workspace:
        pkg.check(lib.nb_comm_set_pair_min_slice(64), "nb_comm_set_pair_min_slice")  # (the test systems are small)
    if os.environ.get("WORKER_LATE_DIAGONAL") is not None:
        pkg.check(lib.nb_set_late_diagonal(int(os.environ["WORKER_LATE_DIAGONAL"])), "nb_set_late_diagonal")
    ws_bytes_fn = lib.nb_comm_workspace_bytes_f32 if f32 else lib.nb_comm_workspace_bytes_f64
    layout_fn = lib.nb_comm_layout_f32 if f32 else lib.nb_comm_layout_f64
    workspaces = []
    one_group = os.environ.get("WORKER_ONE_GROUP")
    no_workspace_rank = int(os.environ.get("WORKER_NO_WORKSPACE_RANK", "-1"))

    def lend_workspace(comm, rank):
        """every rank makes the call when workspaces are in play (with one process / thread per rank it is a collective)"""
        if one_group is not None:
            pkg.check(lib.nb_comm_set_exchange_grouping(comm, int(one_group)), "nb_comm_set_exchange_grouping")
            now = ctypes.c_int(-1)
            pkg.check(lib.nb_comm_get_exchange_grouping(comm, ctypes.byref(now)))
            assert now.value == int(one_group)
        if not with_workspace:
            return 0
        need = ctypes.c_size_t(0)
        pkg.check(ws_bytes_fn(comm, n, mode, ctypes.byref(need)), "nb_comm_workspace_bytes")
        if need.value and rank != no_workspace_rank:
            buf = pkg.DeviceBuffer(need.value)
            pkg.check(lib.nb_memset(buf.ptr, 0xFF, need.value, None), "nb_memset")  # NaN patterns: whatever is read must have been written
            workspaces.append(buf)
            pkg.check(lib.nb_comm_set_workspace(comm, buf.ptr, need.value), "nb_comm_set_workspace")
        else:
            pkg.check(lib.nb_comm_set_workspace(comm, None, 0), "nb_comm_set_workspace")
        return need.value

    def layout_of(comm):
        flag = ctypes.c_int(-1)
        pkg.check(layout_fn(comm, n, mode, ctypes.byref(flag)), "nb_comm_layout")
        return flag.value

    def counter_row():
        c = counters()
        return np.array([c["sends"], c["recvs"], c["allgathers"], c["groups"], c["copies"]])

    step_all = lib.nb_sharded_step_all_f32 if f32 else lib.nb_sharded_step_all_f64
    step_one = lib.nb_sharded_step_f32 if f32 else lib.nb_sharded_step_f64
    out = {}

    if case == "all":
        # ONE thread drives every rank: nb_comm_init_all + nb_sharded_step_all_* (every RCCL round is one group over the ranks)
        comms = (ctypes.c_void_p * G)()
        if REAL_RCCL:  # RCCL refuses two ranks per device: the ranks share ONE real one-rank communicator, transfers routed by order
            uid = ctypes.create_string_buffer(128)
            pkg.check(lib.nb_comm_unique_id(uid), "nb_comm_unique_id")
            pkg.check(lib.nb_comm_inprocess_open_all(comms, G, uid), "nb_comm_inprocess_open_all")
        else:
            pkg.check(lib.nb_comm_init_all(comms, G, (ctypes.c_int * G)(*([0] * G))), "nb_comm_init_all")
        ranks = [Rank(pkg, pos0, vel0, own_streams) for _ in range(G)]
        arr = lambda xs: (ctypes.c_void_p * G)(*xs)  # noqa: E731
        # argument checks first (nothing is launched by a rejected call): a subset of the group, a rank twice, the per-rank form
        sub = (ctypes.c_void_p * (G - 1))(*list(comms)[:G - 1])
        pick = lambda xs: (ctypes.c_void_p * (G - 1))(*xs[:G - 1])  # noqa: E731
        rc_subset = step_all(sub, G - 1, pick([r.bufs[1].ptr for r in ranks]), pick([r.bufs[0].ptr for r in ranks]), pick([r.bufs[2].ptr for r in ranks]),
                             pick([r.bufs[3].ptr for r in ranks]), n, dt, one, 256, mode, pick([r.stream for r in ranks]))
        twice = arr([comms[0]] * G)
        rc_twice = step_all(twice, G, arr([r.bufs[1].ptr for r in ranks]), arr([r.bufs[0].ptr for r in ranks]), arr([r.bufs[2].ptr for r in ranks]),
                            arr([r.bufs[3].ptr for r in ranks]), n, dt, one, 256, mode, arr([r.stream for r in ranks]))
        rc_single = step_one(comms[0], ranks[0].bufs[1].ptr, ranks[0].bufs[0].ptr, ranks[0].bufs[2].ptr, ranks[0].bufs[3].ptr, n, dt, one, 256, mode, ranks[0].stream)
        out["rejected"] = np.array([rc_subset, rc_twice, rc_single])
        out["workspace_bytes"] = np.array([lend_workspace(c, k) for k, c in enumerate(comms)])
        out["layout"] = np.array([layout_of(c) for c in comms])
        out["setup_counters"] = counter_row()
        for _ in range(steps):
            rd = ranks[0].read
            pkg.check(step_all(comms, G, arr([r.bufs[1 - rd].ptr for r in ranks]), arr([r.bufs[rd].ptr for r in ranks]), arr([r.bufs[2].ptr for r in ranks]),
                               arr([r.bufs[3].ptr for r in ranks]), n, dt, one, 256, mode, arr([r.stream for r in ranks])), "nb_sharded_step_all")
            for r in ranks:
                r.read = 1 - rd
        if steps:  # what the last pairwise step of rank 0 enqueued, in host order (empty for the one-sided schedule)
            text = ctypes.create_string_buffer(4096)
            pkg.check(lib.nb_comm_last_step_trace(comms[0], text, len(text)), "nb_comm_last_step_trace")
            out["trace_0"] = np.frombuffer(text.value, dtype=np.uint8)
        for k, r in enumerate(ranks):
            p, v = r.results(comms[k], pos0, vel0)
            out[f"pos_{k}"] = p
            out[f"vel_{k}"] = v.reshape(n, 4)[k * ni:(k + 1) * ni].ravel()  # velocities live with their owner
        for c in (reversed(list(comms)) if REAL_RCCL else comms):  # (an in-process world shares its ncclComm and exchange stream: any order must do)
            pkg.check(lib.nb_comm_destroy(c), "nb_comm_destroy")

    elif case in ("threads", "exchange"):
        # one THREAD per rank: nb_comm_unique_id + nb_comm_init_rank + the per-rank entry points; the send/recv rounds of the
        # ranks meet inside the transport, as they would with one process per GPU
        uid = ctypes.create_string_buffer(128)
        pkg.check(lib.nb_comm_unique_id(uid), "nb_comm_unique_id")
        results, errors, layouts, setup = {}, [], {}, {}
        ready = threading.Barrier(G)

        def rank_main(k):
            try:
                pkg.check(lib.nb_set_device(0), "nb_set_device")
                comm = ctypes.c_void_p()
                pkg.check(lib.nb_comm_init_rank(ctypes.byref(comm), uid, G, k), "nb_comm_init_rank")
                r, w, d = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
                pkg.check(lib.nb_comm_info(comm, ctypes.byref(r), ctypes.byref(w), ctypes.byref(d)))
                assert (r.value, w.value, d.value) == (k, G, 0)
                if case == "threads":
                    lend_workspace(comm, k)
                    layouts[k] = layout_of(comm)
                    ready.wait(timeout=120)  # every rank is through the set-up: what the transport has seen so far is set-up traffic
                    if k == 0:
                        setup["counters"] = counter_row()
                    ready.wait(timeout=120)
                    me = Rank(pkg, pos0, vel0, own_streams)
                    for _ in range(steps):
                        pkg.check(step_one(comm, me.bufs[1 - me.read].ptr, me.bufs[me.read].ptr, me.bufs[2].ptr, me.bufs[3].ptr, n, dt, one, 256, mode, me.stream), "nb_sharded_step")
                        me.read = 1 - me.read
                    p, v = me.results(comm, pos0, vel0)
                    results[k] = (p, v.reshape(n, 4)[k * ni:(k + 1) * ni].ravel())
                else:
                    # the exchange on its own: every rank holds only its own slice (the rest poisoned); after the exchange all
                    # hold the whole array.  Tiles first (waiting tile by tile), then the single-collective form.
                    exch = (lib.nb_exchange_tiles_f32, lib.nb_allgather_f32) if f32 else (lib.nb_exchange_tiles_f64, lib.nb_allgather_f64)
                    got = []
                    for form, fn in enumerate(exch):
                        mine = np.full_like(pos0, -7.0)
                        mine.reshape(n, 4)[k * ni:(k + 1) * ni] = pos0.reshape(n, 4)[k * ni:(k + 1) * ni] + dtype.type(form)
                        buf = pkg.DeviceBuffer(mine.nbytes)
                        buf.upload(mine)
                        stream = ctypes.c_void_p()
                        pkg.check(lib.nb_stream_create(ctypes.byref(stream)))
                        pkg.check(fn(comm, buf.ptr, n, stream), "exchange")
                        for peer in range(G):
                            pkg.check(lib.nb_exchange_wait_tile(comm, peer, stream), "nb_exchange_wait_tile")
                        pkg.check(lib.nb_stream_synchronize(stream))
                        got.append(buf.download(np.zeros_like(pos0)).copy())
                        pkg.check(lib.nb_stream_destroy(stream))
                        buf.free()
                    results[k] = tuple(got)
                pkg.check(lib.nb_comm_destroy(comm), "nb_comm_destroy")
            except BaseException as exc:  # noqa: BLE001 -- reported by the main thread
                errors.append((k, repr(exc)))
                ready.abort()

        threads = [threading.Thread(target=rank_main, args=(k,)) for k in range(G)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise SystemExit(f"rank threads failed: {errors}")
        if case == "threads":
            out["layout"] = np.array([layouts[k] for k in range(G)])
            out["setup_counters"] = setup["counters"]
        for k in range(G):
            if case == "threads":
                out[f"pos_{k}"], out[f"vel_{k}"] = results[k]
            else:
                out[f"tiles_{k}"], out[f"gather_{k}"] = results[k]
    else:
        raise SystemExit(f"unknown case {case}")

    # the same system on ONE rank through nb_integrate_* (what the sharded runs must reproduce)
    if case != "exchange":
        single = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0, vel0, mode=mode)
        for _ in range(steps):
            single.update(dt)
        out["single_pos"], out["single_vel"] = single.get_position().copy(), single.get_velocity().copy()
        single.free()
    out["counters"] = counter_row() - out.get("setup_counters", 0)  # the steps' own traffic
    np.savez(dst, **out)


if __name__ == "__main__":
    main()
