"""How much does a small high-priority kernel (a stand-in for RCCL's all-gather) hurt the own-slice force kernel of an
8-GPU shard, for different launch geometries?  One GPU; the spin kernel holds `blocks` x 512 threads for 100 us at the
start of every step on a high-priority stream."""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libspin.so"))
spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sharded = entry.load_package_module("sharded")
SPIN_US = int(os.environ.get("SPIN_US", "100"))
n, G = 262144, 8
orc = entry.load_oracle().Oracle()
pos0, vel0 = orc.startup_state(n, np.float32)
dev = torch.device("cuda", 0)
pos_t = torch.from_numpy(pos0.reshape(n, 4)).to(dev); vel_t = torch.from_numpy(vel0.reshape(n, 4)).to(dev)
nxt = pos_t.clone(); acc_t = torch.zeros_like(pos_t)
pkg.check(lib.nb_set_softening_sq_f32(np.float32(0.01)))
main = torch.cuda.current_stream()
hi = torch.cuda.Stream(priority=-1)
i0, ni = sharded.slice_of(G // 2, G, n)
sched = sharded.chunk_schedule(i0, ni, n, False)
def launch(k, j0, nj):
    flags = (pkg.NB_SHARD_ACC_IN if k else 0) | (pkg.NB_SHARD_FINALIZE if k == len(sched) - 1 else 0)
    pkg.check(lib.nb_integrate_shard_f32(nxt.data_ptr(), pos_t.data_ptr(), vel_t.data_ptr(), acc_t.data_ptr(), i0, ni, j0, nj, flags,
                                         np.float32(0.016), np.float32(1.0), 256, pkg.NB_MODE_FAST, ctypes.c_void_p(main.cuda_stream)))
for plan in ((2, 16, 2048), (2, 8, 2048), (4, 64, 1024)):
    pkg.set_plan_override(*plan)
    row = {"plan": plan}
    for spin_blocks in (0, 8, 16, 32):
        def step():
            if spin_blocks:
                hi.wait_stream(main)                      # the "gather" starts when the previous step's kernels are done
                spin.spin_launch(spin_blocks, 512, SPIN_US, ctypes.c_void_p(hi.cuda_stream))
            launch(0, *sched[0][:2])                      # own-slice chunk: overlaps the "gather"
            if spin_blocks:
                main.wait_stream(hi)                      # remote chunks wait for it
            for k in range(1, len(sched)):
                launch(k, *sched[k][:2])
        for _ in range(3): step()
        e0, e1 = pkg.Event(), pkg.Event()
        torch.cuda.synchronize(); e0.record(ctypes.c_void_p(main.cuda_stream))
        K = 30
        for _ in range(K): step()
        e1.record(ctypes.c_void_p(main.cuda_stream)); e1.synchronize()
        row[f"ms_spin{spin_blocks}"] = round(e0.elapsed_ms(e1) / K, 4)
    print(json.dumps(row), flush=True)
pkg.set_plan_override(0, 0, 0)
