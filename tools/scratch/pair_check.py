"""first light for the pairwise layout: one step against the one-sided FAST kernel and an fp64 run; timing."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
O = entry.load_oracle(); oracle = O.Oracle()
def run(n, dtype, ws, steps, pos0, vel0):
    s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0.astype(dtype), vel0.astype(dtype), mode=pkg.NB_MODE_FAST, workspace=ws)
    for _ in range(steps): s.update(dtype(np.float32(0.016)))
    p, v = s.get_position().copy(), s.get_velocity().copy()
    # timing
    s.synchronize(); e0, e1 = pkg.Event(), pkg.Event(); reps = 10 if n >= 65536 else 100
    e0.record(None)
    for _ in range(reps): s.update(dtype(np.float32(0.016)))
    e1.record(None); e1.synchronize(); ms = e0.elapsed_ms(e1) / reps
    s.free()
    return p, v, ms
for n, masses in ((16384, "unit"), (65536, "unit"), (262144, "unit"), (20000, "unit"), (65536 + 77, "mixed"), (262144, "species")):
    pos0, vel0 = oracle.startup_state((n + 7) // 8 * 8, np.float32)
    pos0, vel0 = pos0[:4 * n].copy(), vel0[:4 * n].copy()
    if masses == "mixed":
        pos0.reshape(n, 4)[:, 3] = np.linspace(0.5, 2.0, n).astype(np.float32)
    if masses == "species":
        pos0.reshape(n, 4)[n // 3:, 3] = 2.0; pos0.reshape(n, 4)[2 * n // 3:, 3] = 0.25
    for dtype in (np.float32, np.float64):
        pl = pkg.pair_plan(n, dtype)
        t_p, t_v, _ = run(n, np.float64, False, 1, pos0, vel0)
        a_p, a_v, a_ms = run(n, dtype, False, 1, pos0, vel0)
        b_p, b_v, b_ms = run(n, dtype, True, 1, pos0, vel0)
        ea = np.abs(a_v.astype(np.float64) - t_v).reshape(n, 4)[:, :3].max(); eb = np.abs(b_v.astype(np.float64) - t_v).reshape(n, 4)[:, :3].max()
        scale = np.abs(t_v - vel0.astype(np.float64)).max()
        print(f"n={n} {masses} {np.dtype(dtype).name}: plan I={pl.bodies_per_lane} S={pl.waves_per_block} C={pl.splits} NB={pl.blocks} slots={pl.reaction_slots} ws={pl.workspace_bytes/2**20:.0f} MiB applies={pl.applies} | "
              f"dv err vs fp64 (rel to max dv): one-sided {ea/scale:.2e} pairwise {eb/scale:.2e} | ms one-sided {a_ms:.3f} pairwise {b_ms:.3f} (x{a_ms/b_ms:.2f})", flush=True)
