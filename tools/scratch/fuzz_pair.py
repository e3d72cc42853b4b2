"""one-off fuzz: the pairwise layout at random sizes / geometries against the one-sided FAST kernel (accelerations, one step from rest)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
def accel(pos, dtype, ws):
    n = pos.size // 4
    s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos.astype(dtype), np.zeros(4 * n, dtype), mode=pkg.NB_MODE_FAST, workspace=ws)
    s.update(dtype(1)); a = s.get_velocity().copy(); s.free(); return a.reshape(n, 4)[:, :3].astype(np.float64)
worst = 0.0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    n = int(rng.integers(1, 6000)) if case % 3 else int(rng.integers(1, 300))
    R, S, C = int(rng.choice([1, 2, 4])), int(rng.choice([4, 8, 16])), int(rng.integers(1, 9))
    dtype = np.float32 if case % 4 else np.float64
    pos = rng.uniform(-5, 5, size=(n, 4)).astype(np.float32)
    kind = case % 5
    pos[:, 3] = {0: 1.0, 1: 2.5}.get(kind, 1.0)
    if kind == 2: pos[:, 3] = rng.uniform(0.5, 2.0, n)
    if kind == 3: pos[n // 2:, 3] = 3.0
    if kind == 4: pos[rng.integers(0, n, max(1, n // 10)), 3] = 0.0
    pkg.set_pair_plan_override(R, S, C, 1)
    pl = pkg.pair_plan(n, dtype)
    b = accel(pos.ravel(), dtype, True)
    pkg.set_pair_plan_override(0, 0, 0, 0)
    a = accel(pos.ravel(), dtype, False)
    scale = np.abs(a).max() + 1e-30
    err = np.abs(a - b).max() / scale
    worst = max(worst, err if dtype == np.float32 else err * 1e7)
    ok = err < (2e-5 if dtype == np.float32 else 1e-12)
    if not ok or not np.isfinite(b).all():
        print("FAIL", case, n, (R, S, C), np.dtype(dtype).name, kind, err, "applies", pl.applies, flush=True); sys.exit(1)
print("fuzz ok, worst normalised difference", worst)
