import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
host = entry.load_oracle().Oracle()
for n in (4096, 8192, 10240, 16384, 32768):
    pos0, vel0 = host.startup_state(n, np.float32)
    for ws in (False, True):
        s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float32, pos0, vel0, mode=pkg.NB_MODE_FAST, workspace=ws)
        dt = np.float32(0.016)
        for _ in range(5): s.update(dt)
        s.synchronize(); e0, e1 = pkg.Event(), pkg.Event(); e0.record(None)
        for _ in range(400): s.update(dt)
        e1.record(None); e1.synchronize(); loop = e0.elapsed_ms(e1)/400
        s.update_many(dt, 400); s.synchronize()
        e0.record(None); s.update_many(dt, 400); e1.record(None); e1.synchronize(); graph = e0.elapsed_ms(e1)/400
        print(f"n={n} {'pairwise ws' if ws and s._workspace is not None else 'one-sided'}: loop {loop*1e3:.1f} us/step, one hipGraph of 400 steps {graph*1e3:.1f} us/step", flush=True)
        s.free()
