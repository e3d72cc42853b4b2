"""tools/shard_plan_times.py [f64] -- one rank's kernels of a G-rank pairwise step (nb_emulate_pair_rank_*: diagonal, rectangles, folds,
finish; rank G/2, alone on one GPU, no exchange) under plan overrides (R, S, C; 0 = automatic): ms per step and MiB of workspace."""
import ctypes, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import bench
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
def run(n, dtype, G, plan, steps=20):
    f32 = dtype == np.float32
    emulate = lib.nb_emulate_pair_rank_f32 if f32 else lib.nb_emulate_pair_rank_f64
    pos0, vel0 = bench.make_bodies(n, dtype)
    bufs = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(3)]
    bufs[0].upload(pos0), bufs[2].upload(vel0)
    dt, damping = dtype(0.016), dtype(1.0)
    pkg.set_pair_plan_override(*plan, 1 if any(plan) else 0)
    need = ctypes.c_size_t(0)
    rc = emulate(None, None, None, None, ctypes.byref(need), n, G, 0, dt, damping, None)
    if rc != 0 or need.value == 0:
        pkg.set_pair_plan_override(0,0,0,0); [b.free() for b in bufs]; return None
    work = pkg.DeviceBuffer(need.value)
    out = []
    for r in (G // 2,):
        def one(): pkg.check(emulate(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, work.ptr, ctypes.byref(need), n, G, r, dt, damping, None))
        for _ in range(3): one()
        pkg.check(lib.nb_device_synchronize())
        best = 1e9
        for rep in range(3):
            e0, e1 = pkg.Event(), pkg.Event(); e0.record(None)
            for _ in range(steps): one()
            e1.record(None); e1.synchronize(); best = min(best, e0.elapsed_ms(e1) / steps)
        out.append(best)
    pkg.set_pair_plan_override(0,0,0,0)
    work.free(); [b.free() for b in bufs]
    return out[0], need.value
if __name__ == "__main__":
    dtype = np.float64 if "f64" in sys.argv[1:] else np.float32
    for n in (262144, 1048576):
        for G in (2, 3, 4, 8):
            if n % G or (n == 1048576 and G < 4):
                continue
            row = {}
            for plan in ((0, 0, 0), (4, 8, 0), (8, 8, 0), (8, 8, 1), (8, 8, 2), (8, 8, 4), (8, 8, 8), (4, 8, 4), (4, 8, 8)):
                r = run(n, dtype, G, plan, steps=20 if n == 262144 else 4)
                row[str(plan)] = r and (round(r[0], 4), r[1] >> 20)
            print(json.dumps({"bodies": n, "dtype": dtype.__name__, "G": G, "ms_and_MiB": row}), flush=True)
