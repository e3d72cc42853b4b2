#!/bin/bash
C="--gpus 1 --steps 6 --warmup 2 --bodies 16384 --no-cpu-baseline --no-configs"
python3 bench.py $C --dump-state /tmp/p.npz > /dev/null 2>&1
for k in 1 2 3 4 5 6 7 8; do
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $((29620+k)) bench.py $C --dump-state /tmp/t$k.npz > /dev/null 2>&1
done
python3 - <<'PY'
import numpy as np
p=np.load("/tmp/p.npz")
print([bool(np.load(f"/tmp/t{k}.npy").tobytes()==p.tobytes()) for k in range(1,9)])
PY
