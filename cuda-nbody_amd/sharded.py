"""Body sharding across the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI) for the one exchange step the path has -- an all-gather of the new positions per step.

New design (the reference is single-GPU).  Rank r owns bodies i in [i0, i0+ni): their velocities
and their slice of every new position array.  Forces need all positions, so each step ends with an in-place
all-gather of the ranks' slices into the next read buffer (N x 16 B in total: 4 MiB at 262 144 bodies).

Overlap (FAST mode): force accumulation is additive over j chunks, so the next step starts with the chunk
that is already local -- j in the rank's OWN slice -- while RCCL moves the remote slices; the kernel(s) for the
remote chunks wait on the collective, the last one integrates.  STRICT mode keeps the CPU path's per-body
summation order (j ascending), so it waits for the gather and walks the chunks in order: bit-identical to
one GPU, no overlap.

`launch` is the per-rank compute callable with the signature of nb_integrate_shard_* minus the buffers:
    launch(new_pos, old_pos, vel, acc, i_begin, i_count, j_begin, j_count, flags)
bench.py binds it to libnbody_hip.so on the rank's GPU; tests/test_sharded_gloo.py binds it to the CPU oracle
to check the sharding logic with world_size 2 on gloo.
"""
from __future__ import annotations

NB_SHARD_ACC_IN, NB_SHARD_FINALIZE = 1, 2


def slice_of(rank: int, world: int, n: int) -> tuple[int, int]:
    """Contiguous equal slices; n must divide evenly (pad with zero-mass bodies otherwise, as tipsy.cpp:111-119 does)."""
    if n % world:
        raise ValueError(f"{n} bodies do not shard evenly over {world} ranks; pad with zero-mass bodies")
    ni = n // world
    return rank * ni, ni


def chunk_schedule(i0: int, ni: int, n: int, ordered: bool) -> list[tuple[int, int, bool]]:
    """(j_begin, j_count, needs_remote_data) in issue order.
    ordered=False: own chunk first (local data), then the chunks below and above it.
    ordered=True : ascending j (STRICT summation order); everything waits for the gather."""
    below, own, above = (0, i0), (i0, ni), (i0 + ni, n - i0 - ni)
    if ordered:
        seq = [(*below, True), (*own, True), (*above, True)]
    else:
        seq = [(*own, False), (*below, True), (*above, True)]
    return [c for c in seq if c[1] > 0]


class ShardedBodySystem:
    """Ping-pong positions + velocities + partial-acceleration scratch, all full-size torch tensors on this
    rank's device (N x 4 T each: 16 MiB at 1 Mi bodies -- nothing next to 288 GB), indexed by global body id."""

    def __init__(self, pos0, vel0, launch, ordered: bool = False, group=None, gather=None):
        """`gather(full, own_slice)` -> object with .wait() replaces the in-place RCCL all-gather (tests stage it
        through host memory with gloo to run two ranks on one GPU)."""
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n = pos0.shape[0]
        self.i0, self.ni = slice_of(self.rank, self.world, self.n)
        self.pos = [pos0.clone(), pos0.clone()]
        self.vel = vel0.clone()
        self.acc = pos0.new_zeros(pos0.shape)
        self.launch = launch
        self.ordered = ordered
        self.read = 0
        self.pending = None  # the in-flight all-gather of self.pos[self.read]
        self.schedule = chunk_schedule(self.i0, self.ni, self.n, ordered)
        self._gather = gather or self._rccl_gather
        self._in_place = True

    def _rccl_gather(self, full, own):
        """In-place all-gather (own slice already sits at its offset in `full`; RCCL then moves only remote slices).
        If the backend rejects the aliasing form, fall back once and for all to gathering into a staging buffer."""
        if self._in_place:
            try:
                return self.dist.all_gather_into_tensor(full, own, group=self.group, async_op=True)
            except (RuntimeError, ValueError):
                self._in_place = False
                self._staging = full.new_empty(full.shape)
        work = self.dist.all_gather_into_tensor(self._staging, own.contiguous(), group=self.group, async_op=True)

        class _CopyBack:
            def __init__(self, work, dst, src):
                self.work, self.dst, self.src = work, dst, src

            def wait(self):
                self.work.wait()
                self.dst.copy_(self.src)

        return _CopyBack(work, full, self._staging)

    def update(self) -> None:
        """One step: pos[1-read][own], vel[own] <- integrate(pos[read]); then start gathering pos[1-read]."""
        cur, nxt = self.pos[self.read], self.pos[1 - self.read]
        last = len(self.schedule) - 1
        for k, (j0, nj, remote) in enumerate(self.schedule):
            if remote and self.pending is not None:
                self.pending.wait()  # compute stream waits for the collective; the host does not block
                self.pending = None
            flags = (NB_SHARD_ACC_IN if k > 0 else 0) | (NB_SHARD_FINALIZE if k == last else 0)
            self.launch(nxt, cur, self.vel, self.acc, self.i0, self.ni, j0, nj, flags)
        if self.pending is not None:  # world == 1 with nothing remote
            self.pending.wait()
            self.pending = None
        if self.world > 1:
            own = nxt[self.i0:self.i0 + self.ni]
            self.pending = self._gather(nxt, own)
        self.read = 1 - self.read

    def finish(self) -> None:
        if self.pending is not None:
            self.pending.wait()
            self.pending = None

    def positions(self):
        self.finish()
        return self.pos[self.read]

    def velocities(self):
        """Full velocity array (gathers the ranks' slices; not on the timed path)."""
        self.finish()
        if self.world == 1:
            return self.vel
        out = self.vel.clone()
        self.dist.all_gather_into_tensor(out, self.vel[self.i0:self.i0 + self.ni].contiguous(), group=self.group)
        return out
