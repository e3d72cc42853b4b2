import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libspin.so"))
spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
s = torch.cuda.current_stream()
for blocks in (1, 16, 256):
    for us in (100, 1000):
        spin.spin_launch(blocks, 512, us, ctypes.c_void_p(s.cuda_stream)); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): spin.spin_launch(blocks, 512, us, ctypes.c_void_p(s.cuda_stream))
        e1.record(); torch.cuda.synchronize()
        print(blocks, us, "-> measured us per launch:", e0.elapsed_time(e1) * 100)
