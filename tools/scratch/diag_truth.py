import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); pkg.check(pkg.lib().nb_set_device(0))
O = entry.load_oracle(); oracle = O.Oracle()
for n in (16384, 65536, 262144):
    pos0, vel0 = oracle.startup_state(n, np.float32)
    print(n, "pos range", np.abs(pos0.reshape(n,4)[:,:3]).max(), "vel range", np.abs(vel0.reshape(n,4)[:,:3]).max())
    systems = {
        "f64": pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float64, pos0.astype(np.float64), vel0.astype(np.float64), mode=pkg.NB_MODE_FAST),
        "fast": pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float32, pos0, vel0, mode=pkg.NB_MODE_FAST),
        "strict": pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float32, pos0, vel0, mode=pkg.NB_MODE_STRICT),
    }
    for step in range(1, 11):
        systems["f64"].update(np.float64(np.float32(0.016)))
        systems["fast"].update(np.float32(0.016)); systems["strict"].update(np.float32(0.016))
        if step in (1, 2, 5, 10):
            t = systems["f64"].get_position().reshape(n, 4)[:, :3]
            out = []
            for k in ("fast", "strict"):
                d = np.abs(systems[k].get_position().reshape(n, 4)[:, :3].astype(np.float64) - t).max(axis=1)
                out.append(f"{k}: max {d.max():.2e} p99 {np.percentile(d,99):.2e} med {np.median(d):.2e}")
            v = systems["f64"].get_velocity().reshape(n,4)[:,:3]
            print(n, "step", step, " | ".join(out), "| |v|max", np.abs(v).max(), flush=True)
    for s in systems.values(): s.free()
