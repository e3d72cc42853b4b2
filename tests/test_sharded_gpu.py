"""GPU test of the body-sharded step with the REAL kernels: two ranks (two processes) share the one MI355X of the
test box.  RCCL refuses two ranks on one device, so the position all-gather is staged through host memory over
gloo here (ShardedBodySystem's `gather` hook); everything else -- slices, chunk schedule, flag chaining, the
nb_integrate_shard_* launches on each rank's stream, the ping-pong -- is the code path bench.py --gpus N runs."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, steps, mode_name, out_dir):
    sys.path.insert(0, ROOT)
    import ctypes

    import torch  # before libnbody_hip.so: one HIP runtime per process
    import torch.distributed as dist

    import __graft_entry__ as entry

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = entry.load_package()
        lib = pkg.lib()
        torch.cuda.set_device(0)
        pkg.check(lib.nb_set_device(0))
        O = entry.load_oracle()
        pos0, vel0 = O.Oracle().startup_state(n, np.float32)
        dev = torch.device("cuda", 0)
        pos_t = torch.from_numpy(pos0.reshape(n, 4)).to(dev)
        vel_t = torch.from_numpy(vel0.reshape(n, 4)).to(dev)
        mode = pkg.NB_MODE_STRICT if mode_name == "strict" else pkg.NB_MODE_FAST
        pkg.check(lib.nb_set_softening_sq_f32(np.float32(0.1) * np.float32(0.1)))
        stream = torch.cuda.current_stream()

        def launch(new_pos, old_pos, vel, acc, i0, ni, j0, nj, flags):
            pkg.check(lib.nb_integrate_shard_f32(new_pos.data_ptr(), old_pos.data_ptr(), vel.data_ptr(), acc.data_ptr(), i0, ni, j0, nj, flags,
                                                 np.float32(0.016), np.float32(1.0), 256, mode, ctypes.c_void_p(stream.cuda_stream)))

        class Done:
            def wait(self):
                pass

        def gather(full, own):
            host_own = own.cpu()
            host_full = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(host_full, host_own)
            full.copy_(host_full)
            return Done()

        sharded = entry.load_package_module("sharded")
        system = sharded.ShardedBodySystem(pos_t, vel_t, launch, ordered=(mode == pkg.NB_MODE_STRICT), gather=gather)
        for _ in range(steps):
            system.update()
        pos = system.positions().cpu().numpy()
        i0, ni = system.i0, system.ni
        vel = system.vel[i0:i0 + ni].cpu().numpy()
        np.save(os.path.join(out_dir, f"pos_{rank}.npy"), pos)
        np.save(os.path.join(out_dir, f"vel_{rank}.npy"), vel)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode_name", ["strict", "fast"])
def test_two_ranks_on_one_gpu_match_the_oracle(tmp_path, oracle, mode_name):
    import torch.multiprocessing as mp

    n, steps, world = 4096, 3, 2
    mp.spawn(_worker, args=(world, _free_port(), n, steps, mode_name, str(tmp_path)), nprocs=world, join=True)
    ref_pos, ref_vel = oracle.startup_state(n, np.float32)
    oracle.update(ref_pos, ref_vel, np.float32(0.016), steps=steps)
    ref_pos, ref_vel = ref_pos.reshape(n, 4), ref_vel.reshape(n, 4)
    pos = [np.load(tmp_path / f"pos_{r}.npy") for r in range(world)]
    assert pos[0].tobytes() == pos[1].tobytes()  # both ranks hold the same gathered positions
    vel = np.concatenate([np.load(tmp_path / f"vel_{r}.npy") for r in range(world)])
    if mode_name == "strict":
        assert pos[0].tobytes() == ref_pos.tobytes()  # ordered schedule: bit-identical to one process / the CPU path
        assert vel.tobytes() == ref_vel.tobytes()
    else:
        np.testing.assert_allclose(pos[0], ref_pos, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(vel, ref_vel, rtol=1e-4, atol=1e-4)
