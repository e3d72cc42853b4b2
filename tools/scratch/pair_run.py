"""N steps of the pairwise layout (for rocprofv3): python pair_run.py [n] [steps] [f32|f64] [R S C]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dtype = np.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else np.float32
if len(sys.argv) > 6: pkg.set_pair_plan_override(int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), 1)
host = entry.load_oracle().Oracle()
pos0, vel0 = host.startup_state(n, np.float32)
s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0.astype(dtype), vel0.astype(dtype), mode=pkg.NB_MODE_FAST, workspace=True)
dt = dtype(np.float32(0.016))
for _ in range(2): s.update(dt)
s.synchronize(); e0, e1 = pkg.Event(), pkg.Event(); e0.record(None)
for _ in range(steps): s.update(dt)
e1.record(None); e1.synchronize()
pl = pkg.pair_plan(n, dtype)
print(f"n={n} {np.dtype(dtype).name} I={pl.bodies_per_lane} S={pl.waves_per_block} C={pl.splits}: {e0.elapsed_ms(e1)/steps:.4f} ms/step")
