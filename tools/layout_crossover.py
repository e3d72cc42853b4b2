"""Where does the wave-split layout stop paying?  Times both layouts of the FAST kernel over a range of sizes."""
import ctypes, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
O = entry.load_oracle(); orc = O.Oracle()
for dtype in (np.float32, np.float64):
    for n in (2048, 8192, 12288, 16384, 20480, 24576, 28672, 32768, 40960, 49152):
        pos0, vel0 = orc.startup_state(n, dtype)
        row = {"dtype": np.dtype(dtype).name, "n": n}
        for name, ovr in (("tile_S16", (0, 16, 0)), ("wave_split", (0, 64, 0))):
            pkg.set_plan_override(*ovr)
            sysm = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0, vel0)
            for _ in range(5): sysm.update(dtype(0.016))
            e0, e1 = pkg.Event(), pkg.Event()
            sysm.synchronize(); e0.record()
            K = 100
            for _ in range(K): sysm.update(dtype(0.016))
            e1.record(); e1.synchronize()
            ms = e0.elapsed_ms(e1) / K
            row[name] = round(n * n / ms * 1e-6, 1)
            sysm.free()
        pkg.set_plan_override(0, 0, 0)
        row["auto"] = pkg.plan(n, n, dtype).lanes_per_body
        print(json.dumps(row), flush=True)
