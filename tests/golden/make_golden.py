"""Regenerates tests/golden/*.npz.  Run from the repo root in the build container:

    python tests/golden/make_golden.py

What the vectors are (and are not):
  * initial conditions: produced by the REFERENCE's own randomise_bodies<T> (compiled unmodified into
    oracle/_ref/librandomise_ref.so by oracle/Makefile) on the process-start rand() stream, third segment
     -- and asserted bit-identical to the oracle's restatement while generating;
  * trajectories after 1/10/100 steps: produced by oracle/nbody_oracle.c (the CPU restatement of
    BodySystemCPU<T>::update).  The reference's bodysystemcpu.cpp cannot be compiled in this image without
    stand-in headers, so these trajectories are the restatement's, not the reference binary's ("parity
    unpinned").  They pin the oracle against silent change and travel to the GPU box.
  * shell_n16384_{f32,f64}_compact.npz (round 4): a system large enough for nb_integrate_ws_* to take the PAIRWISE layout by
    default (it applies above 8 192 bodies fp32 / from 6 144 fp64).  To stay small the fixture holds x, y, z only (mass 1 and
    velocity .w 0 are the start-up values and never change) of the positions after 1 and 10 steps (fp32: the velocities after
    10 steps too), and the SHA-256 of the initial arrays instead of the arrays: the test draws them with the oracle's
    randomise_bodies restatement (pinned to the reference's code, see above) and checks the digest.
    PARITY UNPINNED like the others: its later states are the restatement's own output -- only the initial state, through the
    digest of randomise_bodies' arrays, goes back to reference code.
Parameters: SHELL config, demo_params[0] (dt 0.016, softening 0.1, damping 1.0), cluster/velocity scale by N
(src/nbody/compute.cpp:74-92).  Sizes: N = 8, 256, 1024 (steps 0/1/10/100) and 4096 (steps 0/1/10 -- the fixtures stay small),
the four sizes SURVEY 8(c) names.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
STEPS = {8: (1, 10, 100), 256: (1, 10, 100), 1024: (1, 10, 100), 4096: (1, 10)}
COMPACT = {16384: (1, 10)}


def ref_startup_state(ref, n, dtype):
    ref.srand(1)
    ref.randomise(O.NBODY_CONFIG_SHELL, n, O.DEMO0["cluster_scale"], O.DEMO0["velocity_scale"], np.float32)
    ref.randomise(O.NBODY_CONFIG_SHELL, n, O.DEMO0["cluster_scale"], O.DEMO0["velocity_scale"], np.float64)
    c, v = O.scales_for(n)
    return ref.randomise(O.NBODY_CONFIG_SHELL, n, c, v, dtype)


def main():
    O.build(with_ref=True)
    orc = O.Oracle()
    ref = O.ReferenceRandomise()
    for n in sorted(STEPS):
        for dtype, tag in ((np.float32, "f32"), (np.float64, "f64")):
            pos0, vel0 = ref_startup_state(ref, n, dtype)
            opos0, ovel0 = orc.startup_state(n, dtype)
            assert pos0.tobytes() == opos0.tobytes() and vel0.tobytes() == ovel0.tobytes()
            data = {"pos_0": pos0, "vel_0": vel0}
            pos, vel, done = pos0.copy(), vel0.copy(), 0
            for s in STEPS[n]:
                orc.update(pos, vel, O.DEMO0["time_step"], steps=s - done)
                done = s
                data[f"pos_{s}"], data[f"vel_{s}"] = pos.copy(), vel.copy()
            path = os.path.join(OUT, f"shell_n{n}_{tag}.npz")
            np.savez_compressed(path, **data)
            print(path, os.path.getsize(path))
    for n in sorted(COMPACT):
        for dtype, tag in ((np.float32, "f32"), (np.float64, "f64")):
            pos0, vel0 = ref_startup_state(ref, n, dtype)
            opos0, ovel0 = orc.startup_state(n, dtype)
            assert pos0.tobytes() == opos0.tobytes() and vel0.tobytes() == ovel0.tobytes()
            assert np.all(pos0.reshape(n, 4)[:, 3] == 1) and np.all(vel0.reshape(n, 4)[:, 3] == 0)
            data = {"sha256_pos_0": np.frombuffer(hashlib.sha256(pos0.tobytes()).digest(), np.uint8),
                    "sha256_vel_0": np.frombuffer(hashlib.sha256(vel0.tobytes()).digest(), np.uint8)}
            pos, vel, done = pos0.copy(), vel0.copy(), 0
            for s in COMPACT[n]:
                orc.update(pos, vel, O.DEMO0["time_step"], steps=s - done)
                done = s
                assert np.all(pos.reshape(n, 4)[:, 3] == 1) and np.all(vel.reshape(n, 4)[:, 3] == 0)
                data[f"pos_{s}_xyz"] = pos.reshape(n, 4)[:, :3].copy()
            if tag == "f32":
                data[f"vel_{done}_xyz"] = vel.reshape(n, 4)[:, :3].copy()
            path = os.path.join(OUT, f"shell_n{n}_{tag}_compact.npz")
            np.savez_compressed(path, **data)
            print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
