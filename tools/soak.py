"""soak: long runs of the automatic plans -- finite, bit-reproducible from run to run, momentum kept -- at BASELINE sizes and between the powers of two"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
host = entry.load_oracle().Oracle()
for n, steps, dtype, ws in ((9000, 20000, np.float32, True), (16384, 20000, np.float32, True), (50000, 3000, np.float32, True), (18000, 5000, np.float32, False), (262144, 400, np.float32, True), (100000, 300, np.float64, True), (300000, 100, np.float32, True)):
    p32, v32 = host.startup_state(n, np.float32); pos0, vel0 = p32.astype(dtype), v32.astype(dtype)
    outs = []
    for rep in range(2):
        s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0, vel0, mode=pkg.NB_MODE_FAST, workspace=ws)
        for _ in range(steps): s.update(dtype(0.016))
        outs.append((s.get_position().copy(), s.get_velocity().copy())); s.free()
    m = pos0.reshape(n, 4)[:, 3:4].astype(np.float64)
    p_before = (m * vel0.reshape(n, 4)[:, :3]).sum(axis=0); p_after = (m * outs[0][1].reshape(n, 4)[:, :3].astype(np.float64)).sum(axis=0)
    scale = (m * np.abs(outs[0][1].reshape(n, 4)[:, :3])).sum()
    print(np.dtype(dtype).name, n, steps, "workspace" if ws else "one-sided", "finite:", bool(np.isfinite(outs[0][0]).all()), "reproducible:", outs[0][0].tobytes() == outs[1][0].tobytes(),
          "|x|max %.3g" % float(np.abs(outs[0][0].reshape(n, 4)[:, :3]).max()), "momentum drift / sum|p| %.2e" % (np.abs(p_after - p_before).max() / scale), flush=True)
