"""Force accuracy of the STRICT (= CPU path arithmetic) and FAST kernels against an fp64 direct sum."""
import ctypes, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
O = entry.load_oracle(); orc = O.Oracle(openmp=True); orc.set_num_threads(16)
pkg.set_softening_squared(np.float32(0.1) * np.float32(0.1))
for n in (1024, 16384, 262144):
    pos, _ = orc.startup_state(n, np.float32)
    d_pos, d_acc = pkg.DeviceBuffer(pos.nbytes), pkg.DeviceBuffer(pos.nbytes)
    d_pos.upload(pos)
    sample = np.unique(np.linspace(0, n - 1, 512).astype(int))
    ref = np.concatenate([orc.accel_f64(pos, int(i), 1) for i in sample])
    row = {"n": n}
    for name, mode in (("strict", pkg.NB_MODE_STRICT), ("fast", pkg.NB_MODE_FAST)):
        pkg.check(lib.nb_integrate_shard_f32(None, d_pos.ptr, None, d_acc.ptr, 0, n, 0, n, 0, np.float32(0.016), np.float32(1), 256, mode, None))
        acc = d_acc.download(np.zeros(4 * n, np.float32)).reshape(n, 4)[sample, :3].astype(np.float64)
        err = np.linalg.norm(acc - ref, axis=1) / np.linalg.norm(ref, axis=1)
        row[name] = {"max": float(err.max()), "median": float(np.median(err))}
    print(json.dumps(row), flush=True)
