"""cross-process determinism of the pairwise step: prints a hash of the positions after 8 steps (bodies drawn before HIP starts)"""
import hashlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
host = entry.load_oracle().Oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
pos0, vel0 = host.startup_state(n, np.float32)
pkg = entry.load_package(); pkg.check(pkg.lib().nb_set_device(0))
junk = [pkg.DeviceBuffer(int(s)) for s in np.random.default_rng(os.getpid()).integers(1 << 20, 1 << 26, 4)]  # move the allocations around
s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float32, pos0, vel0, mode=pkg.NB_MODE_FAST, workspace=True)
for _ in range(8): s.update(np.float32(0.016))
print(hashlib.sha1(s.get_position().tobytes()).hexdigest()[:16], hashlib.sha1(pos0.tobytes()).hexdigest()[:8])
