"""What the step costs when the boundary is handed HOST buffers every step (round 5).

`value` in the bench line is measured with the bodies resident in HBM.  A caller that keeps its bodies on the host pays, per step,
positions + velocities up (nb_h2d) and down (nb_d2h): 32 bytes per body each way in fp32.  This times that at BASELINE's sizes,
from pageable memory (what numpy / std::vector hand over) and from pinned memory (nb_host_alloc_mapped), beside the resident step.

    python3 tools/pcie_inclusive.py [--bodies 262144] [--steps 20]

Prints one JSON line."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=262144)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import __graft_entry__ as entry

    pkg = entry.load_package()
    lib = pkg.lib()
    oracle = entry.load_oracle().Oracle()
    pkg.check(lib.nb_set_device(0), "nb_set_device")
    n, dt = args.bodies, np.float32(0.016)
    pos0, vel0 = oracle.startup_state(n, np.float32)
    system = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float32, pos0, vel0, mode=pkg.NB_MODE_FAST, workspace=True)
    sync = lambda: pkg.check(lib.nb_device_synchronize())

    def timed(step):
        for _ in range(3):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        return (time.perf_counter() - t0) * 1e3 / args.steps

    resident = timed(lambda: system.update(dt))

    def host_step(hp, hv):
        system.set_position(hp)  # blocking nb_h2d, read index back to 0
        system.set_velocity(hv)
        system.update(dt)
        system._pos[system.current_read].download(hp)  # blocking nb_d2h
        system._vel.download(hv)

    hp, hv = pos0.copy(), vel0.copy()
    pageable = timed(lambda: host_step(hp, hv))

    nbytes = pos0.nbytes
    pinned = []
    for _ in range(2):
        h, d = ctypes.c_void_p(), ctypes.c_void_p()
        pkg.check(lib.nb_host_alloc_mapped(ctypes.byref(h), ctypes.byref(d), nbytes), "nb_host_alloc_mapped")
        pinned.append((h, np.ctypeslib.as_array(ctypes.cast(h, ctypes.POINTER(ctypes.c_float)), shape=(4 * n,))))
    pinned[0][1][:], pinned[1][1][:] = pos0, vel0
    from_pinned = timed(lambda: host_step(pinned[0][1], pinned[1][1]))
    finite = bool(np.isfinite(pinned[0][1]).all())
    for h, _ in pinned:
        pkg.check(lib.nb_host_free(h))
    system.free()
    rate = lambda ms: 1e-9 * n * float(n) / (ms * 1e-3)
    moved = 4 * nbytes  # positions and velocities, up and down
    print(json.dumps({
        "bodies": n, "steps": args.steps, "bytes_over_pcie_per_step": moved,
        "resident_ms_per_step": round(resident, 4), "resident_ginteractions_per_s": round(rate(resident), 1),
        "pageable_host_buffers_ms_per_step": round(pageable, 4), "pageable_ginteractions_per_s": round(rate(pageable), 1),
        "pageable_copy_GBps": round(moved / ((pageable - resident) * 1e-3) / 1e9, 1),
        "pinned_host_buffers_ms_per_step": round(from_pinned, 4), "pinned_ginteractions_per_s": round(rate(from_pinned), 1),
        "pinned_copy_GBps": round(moved / ((from_pinned - resident) * 1e-3) / 1e9, 1),
        "finite": finite,
    }), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
