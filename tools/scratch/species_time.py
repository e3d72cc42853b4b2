import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
host = entry.load_oracle().Oracle()
n = 262144
pos0, vel0 = host.startup_state(n, np.float32)
def t(name, masses):
    p = pos0.copy(); p.reshape(n,4)[:,3] = masses
    s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float32, p, vel0, mode=pkg.NB_MODE_FAST, workspace=True)
    dt = np.float32(0.016)
    for _ in range(2): s.update(dt)
    s.synchronize(); e0, e1 = pkg.Event(), pkg.Event(); e0.record(None)
    for _ in range(10): s.update(dt)
    e1.record(None); e1.synchronize(); print(f"{name}: {e0.elapsed_ms(e1)/10:.3f} ms", flush=True); s.free()
m = np.ones(n, np.float32)
t("all 1.0", m)
t("all 2.0", m*2)
m2 = m.copy(); m2[n//2:] = 2.0
t("two species, boundary on a block", m2)
m3 = m.copy(); m3[n//3:] = 2.0; m3[2*n//3:] = 0.25
t("three species, boundaries inside blocks", m3)
m4 = m.copy(); m4[512*170:] = 2.0; m4[512*340:] = 0.25
t("three species, boundaries on blocks", m4)
t("all 1.0 again", m)
