import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
host = entry.load_oracle().Oracle()
for n, steps in ((16384, 20000), (262144, 400)):
    pos0, vel0 = host.startup_state(n, np.float32)
    outs = []
    for rep in range(2):
        s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float32, pos0, vel0, mode=pkg.NB_MODE_FAST, workspace=True)
        for _ in range(steps): s.update(np.float32(0.016))
        outs.append(s.get_position().copy()); s.free()
    print(n, steps, "finite:", bool(np.isfinite(outs[0]).all()), "reproducible:", outs[0].tobytes() == outs[1].tobytes(), "|x|max", float(np.abs(outs[0]).max()), flush=True)
