import subprocess, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
g = np.load("tests/golden/shell_n256_f32.npz")
for extra in ([], ["--seed=1"]):
    r = subprocess.run(["cuda-nbody_amd/nbody", "--numbodies=256", "--mode=strict", "--steps=0", "--dump=/tmp/d0.bin"] + extra, capture_output=True, text=True)
    raw = np.fromfile("/tmp/d0.bin", dtype=np.float32)
    print(extra, "rc", r.returncode, "initial pos equal:", raw[:1024].tobytes() == g["pos_0"].tobytes(), raw[:4], g["pos_0"][:4])
r = subprocess.run(["cuda-nbody_amd/nbody", "--numbodies=256", "--mode=strict", "--steps=1", "--dump=/tmp/d1.bin"], capture_output=True, text=True)
raw = np.fromfile("/tmp/d1.bin", dtype=np.float32)
print("step1 equal:", raw[:1024].tobytes() == g["pos_1"].tobytes(), np.abs(raw[:1024]-g["pos_1"]).max())
