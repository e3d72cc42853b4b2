"""Many consecutive steps of the full G-rank step through the REAL RCCL on one GPU (an in-process world: nb_comm_inprocess_open_all),
held against one GPU (round 5).  STRICT arithmetic is the same whoever runs it, so after any number of steps every rank's positions
must equal a single rank's nb_integrate_* BIT FOR BIT: one tile read before it arrived, one reaction array added a round early, and
the chaotic system shows it.  FAST pairwise across the ranks: the same bits through the real library as through the transport double.

    python3 tools/inprocess_world_soak.py [--world 8] [--bodies 32768] [--steps 3000]

Uses the tests' worker (tests/fake_rccl/worker.py).  Prints one JSON line; exit status 1 on a mismatch."""
import argparse
import json
import os
import pathlib
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--bodies", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--fast-steps", type=int, default=1000)
    args = ap.parse_args()
    import __graft_entry__ as entry
    from test_comm_fake_rccl import _run

    oracle = entry.load_oracle().Oracle()
    pos0, vel0 = oracle.startup_state(args.bodies, np.float32)
    out = {"world": args.world, "bodies": args.bodies, "strict_steps": args.steps, "fast_steps": args.fast_steps}
    with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None) as tmp:
        tmp = pathlib.Path(tmp)
        t0 = time.monotonic()
        strict = _run(tmp, "all", pos0, vel0, args.world, args.steps, "strict", real_rccl=True)
        out["strict_s"] = round(time.monotonic() - t0, 1)
        out["strict_every_rank_equals_one_gpu_bitwise"] = bool(all(strict[f"pos_{k}"].tobytes() == strict["single_pos"].tobytes() for k in range(args.world)))
        out["strict_velocities_bitwise"] = bool(np.concatenate([strict[f"vel_{k}"] for k in range(args.world)]).tobytes() == strict["single_vel"].tobytes())
        out["strict_finite"] = bool(np.isfinite(strict["pos_0"]).all())
        t0 = time.monotonic()
        real = _run(tmp, "all", pos0, vel0, args.world, args.fast_steps, "fast", workspace=True, real_rccl=True)
        double = _run(tmp, "all", pos0, vel0, args.world, args.fast_steps, "fast", workspace=True)
        out["fast_s"] = round(time.monotonic() - t0, 1)
        out["fast_pairwise_layout"] = [int(v) for v in real["layout"]]
        out["fast_real_rccl_equals_transport_double_bitwise"] = bool(all(real[f"pos_{k}"].tobytes() == double["pos_0"].tobytes() for k in range(args.world)))
        out["fast_finite"] = bool(np.isfinite(real["pos_0"]).all())
    ok = all(v for k, v in out.items() if k.endswith("bitwise") or k.endswith("finite")) and out["fast_pairwise_layout"] == [1] * args.world
    out["ok"] = bool(ok)
    print(json.dumps(out), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
