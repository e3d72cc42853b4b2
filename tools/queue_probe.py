"""Does bringing communicators up and down change how the two streams of nb_emulate_pair_rank_f32 overlap?"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
from bench_support import make_bodies
pkg = entry.load_package(); pkg.use_lab(); lib = pkg.lib(); pkg.check(lib.nb_set_device(0))
n, G, r = 262144, 8, 4
pkg.check(lib.nb_set_softening_sq_f32(np.float32(0.01)))
dt, damping = np.float32(0.016), np.float32(1.0)
pos0, vel0 = make_bodies(n, np.float32)
bufs = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(3)]
bufs[0].upload(pos0); bufs[1].upload(pos0); bufs[2].upload(vel0)
need = ctypes.c_size_t(0)
pkg.check(lib.nb_emulate_pair_rank_f32(None, None, None, None, ctypes.byref(need), n, G, r, dt, damping, None))
work = pkg.DeviceBuffer(need.value)
def emulate_ms(stream, reps=40):
    f = lambda: pkg.check(lib.nb_emulate_pair_rank_f32(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, work.ptr, ctypes.byref(need), n, G, r, dt, damping, stream))
    f(); pkg.check(lib.nb_device_synchronize())
    e0, e1 = pkg.Event(), pkg.Event(); e0.record(stream)
    for _ in range(reps): f()
    e1.record(stream); e1.synchronize()
    return round(e0.elapsed_ms(e1) / reps, 4)
out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
s1 = ctypes.c_void_p(); pkg.check(lib.nb_stream_create(ctypes.byref(s1)))
out["fresh_process_created_stream"] = emulate_ms(s1)
out["fresh_process_null_stream"] = emulate_ms(None)
for k in range(3):
    comm = ctypes.c_void_p()
    pkg.check(lib.nb_comm_loopback_open(ctypes.byref(comm), pkg.comm_unique_id(), G, r))
    out[f"comm{k}_alive_created_stream"] = emulate_ms(s1)
    pkg.check(lib.nb_comm_destroy(comm))
    out[f"comm{k}_destroyed_created_stream"] = emulate_ms(s1)
    out[f"comm{k}_destroyed_null_stream"] = emulate_ms(None)
s2 = ctypes.c_void_p(); pkg.check(lib.nb_stream_create(ctypes.byref(s2)))
out["new_stream_after"] = emulate_ms(s2)
print(json.dumps(out))
