import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
host = entry.load_oracle().Oracle()
n = 4 * 1048576
pos0, vel0 = host.startup_state(n, np.float32)
for cap_gib in (16, 8, 32, 64):
    cap = cap_gib << 30
    s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float32, pos0, vel0, mode=pkg.NB_MODE_FAST, workspace=True, workspace_cap=cap)
    dt = np.float32(0.016)
    s.update(dt); s.synchronize()
    e0, e1 = pkg.Event(), pkg.Event()
    e0.record(None)
    for _ in range(2): s.update(dt)
    e1.record(None); e1.synchronize()
    ms = e0.elapsed_ms(e1) / 2
    print(json.dumps({"cap_gib": cap_gib, "workspace_gib": round(s._workspace_bytes / 2**30, 2), "ms": round(ms, 1), "frac": round(20.0 * n * n / (ms * 1e-3) / 157.3e12, 4)}), flush=True)
    s.free()
