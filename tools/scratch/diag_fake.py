import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
O = entry.load_oracle(); oracle = O.Oracle()
n, world = 262144, 4
pos0, vel0 = oracle.startup_state(n, np.float32)
env = dict(os.environ); env["NBODY_RCCL_LIB"] = os.path.join(ROOT, "tests/fake_rccl/libfake_rccl.so")
np.savez("/tmp/in.npz", pos=pos0, vel=vel0)
def run(steps, mode):
    subprocess.run([sys.executable, os.path.join(ROOT, "tests/fake_rccl/worker.py"), "all", "/tmp/in.npz", "/tmp/out.npz", str(world), str(steps), mode, "streams"], env=env, check=True)
    return dict(np.load("/tmp/out.npz"))
for steps in (1, 2, 5):
    f = run(steps, "fast"); s = run(steps, "strict")
    def stat(a, b):
        d = np.abs(a - b).reshape(n, 4)[:, :3].max(axis=1)
        return f"max {d.max():.3e} p99 {np.percentile(d, 99):.3e} median {np.median(d):.3e} per-slice-max {[float(f'{d[k*n//world:(k+1)*n//world].max():.2e}') for k in range(world)]}"
    print(steps, "sharded FAST vs single FAST :", stat(f["pos_0"], f["single_pos"]))
    print(steps, "single FAST  vs single STRICT:", stat(f["single_pos"], s["single_pos"]))
    print(steps, "sharded FAST vs single STRICT:", stat(f["pos_0"], s["single_pos"]), flush=True)
