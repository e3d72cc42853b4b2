"""one-off fuzz: the pairwise step across ranks (RCCL test double) at random world sizes / body counts against the CPU oracle"""
import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
oracle = entry.load_oracle().Oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
env = dict(os.environ, NBODY_RCCL_LIB=os.path.join(ROOT, "tests/fake_rccl/libfake_rccl.so"))
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    world = int(rng.integers(2, 9))
    ni = int(rng.integers(70, 1500))
    n = world * ni
    steps = int(rng.integers(1, 4))
    dtype = np.float32 if case % 3 else np.float64
    p32, v32 = oracle.startup_state((n + 7) // 8 * 8, np.float32)
    pos0, vel0 = p32[:4 * n].astype(dtype), v32[:4 * n].astype(dtype)
    if case % 2:
        pos0.reshape(n, 4)[:, 3] = rng.uniform(0.5, 2.0, n).astype(dtype)
    np.savez("/tmp/fz_in.npz", pos=pos0, vel=vel0)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests/fake_rccl/worker.py"), "all", "/tmp/fz_in.npz", "/tmp/fz_out.npz", str(world), str(steps), "fast", "streams", "ws"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    got = dict(np.load("/tmp/fz_out.npz"))
    ref_p, ref_v = pos0.copy(), vel0.copy()
    oracle.update(ref_p, ref_v, dtype(np.float32(0.016)), steps=steps)
    tol = 2e-5 if dtype == np.float32 else 1e-11
    for k in range(world):
        assert got[f"pos_{k}"].tobytes() == got["pos_0"].tobytes(), (case, world, n)
    err = np.abs(got["pos_0"] - ref_p).max()
    print(f"case {case}: world {world} bodies {n} ({ni}/rank) steps {steps} {np.dtype(dtype).name} ws_bytes {int(got['workspace_bytes'][0])} max |dp| {err:.2e}", flush=True)
    assert err < tol * max(1.0, np.abs(ref_p).max()), (case, err)
print("shard fuzz ok")
