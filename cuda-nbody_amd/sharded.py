"""Body sharding across the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI) for the one exchange step the path has -- an all-gather of the new positions per step.

New design (the reference is single-GPU).  Rank r owns bodies i in [i0, i0+ni): their velocities
and their slice of every new position array.  Forces need all positions, so each step ends with an in-place
all-gather of the ranks' slices into the next read buffer (N x 16 B in total: 4 MiB at 262 144 bodies).

Two forms of that all-gather:
  exchange="tiles" (default)  the gather is issued as its G-1 position TILES: in round s = 1..G-1 every rank sends
      its slice to rank r-s and receives the slice of rank r+s (one grouped RCCL send/recv pair per round; xGMI is a
      full mesh, so every round runs on direct links).  Force accumulation is additive over j chunks, so the next
      step starts with the chunk that is already local -- j in the rank's OWN slice -- and then takes the tiles in
      the order they arrive, the kernel for tile k waiting only on round k: the exchange of tile k+1 runs under the
      force compute of tile k.  STRICT mode keeps the CPU path's per-body summation order (j ascending), so it takes
      the tiles in rank order, each waiting on its own round: bit-identical to one GPU.
  exchange="allgather"        ONE in-place all_gather_into_tensor per step; the own-slice chunk overlaps it, the two
      remote chunks (below / above the own slice) wait for the whole collective.

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


def tile_schedule(rank: int, world: int, n: int, ordered: bool) -> list[tuple[int, int, int | None]]:
    """(j_begin, j_count, peer) in issue order for the tile exchange; peer is None for the rank's own slice.
    ordered=False: own slice, then the peers in the order their tiles arrive (rank+1, rank+2, ... mod world).
    ordered=True : ascending rank = ascending j (STRICT summation order)."""
    ni = n // world
    peers = list(range(world)) if ordered else [(rank + s) % world for s in range(world)]
    return [(p * ni, ni, None if p == rank else p) for p in peers]


class ShardedBodySystem:
    """Ping-pong positions + velocities + partial-acceleration scratch, all full-size torch tensors on this
    rank's device (N x 4 T each: 16 MiB at 1 Mi bodies -- nothing next to 288 GB), indexed by global body id."""

    def __init__(self, pos0, vel0, launch, ordered: bool = False, group=None, gather=None, exchange: str = "tiles"):
        """`gather(full, own_slice)` -> object with .wait() replaces the RCCL exchange altogether (tests stage it
        through host memory with gloo to run two ranks on one GPU); it implies the allgather form."""
        import torch.distributed as dist

        if exchange not in ("tiles", "allgather"):
            raise ValueError(exchange)
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
        self.exchange = "allgather" if gather is not None else exchange
        self.pending = None   # allgather form: the in-flight all-gather of self.pos[self.read]
        self.arrivals = {}    # tiles form: peer -> in-flight work objects of the round that brings its tile
        if self.exchange == "tiles":
            self.schedule = tile_schedule(self.rank, self.world, self.n, ordered)
        else:
            self.schedule = chunk_schedule(self.i0, self.ni, self.n, ordered)
        self._gather = gather or self._rccl_gather
        self._in_place = True

    # ---- allgather form ----------------------------------------------------------------------------------------
    def _rccl_gather(self, full, own):
        """In-place all-gather (own slice already sits at its offset in `full`; RCCL then moves only remote slices).
        If the backend rejects the aliasing form, fall back once and for all to gathering into a staging buffer."""
        if self._in_place:
            try:
                return self.dist.all_gather_into_tensor(full, own, group=self.group, async_op=True)
            except (RuntimeError, ValueError) as exc:
                import sys

                print(f"[sharded rank {self.rank}] in-place all_gather_into_tensor rejected ({exc!r}); using a staging buffer from now on",
                      file=sys.stderr, flush=True)
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

    # ---- tiles form --------------------------------------------------------------------------------------------
    def _start_tiles(self, full) -> None:
        """Round s = 1..G-1: send the own slice to rank-s, receive the slice of rank+s into its place in `full`.
        One grouped send/recv pair per round (batch_isend_irecv = ncclGroupStart/End on RCCL): the rounds complete in
        order on RCCL's stream, and the kernel of a tile waits only for its own round.
        gloo with device tensors (the one-GPU rehearsal): gloo is not stream-aware, so the slices are staged through host
        memory -- own slice copied out after a stream synchronize, tiles copied in when their round is waited for."""
        dist, G, r, ni = self.dist, self.world, self.rank, self.ni
        own = full[self.i0:self.i0 + ni]
        staged = full.is_cuda and dist.get_backend(self.group) == "gloo"
        if staged:
            import torch

            torch.cuda.current_stream().synchronize()
            own = own.cpu()
        for s in range(1, G):
            dst, src = (r - s) % G, (r + s) % G
            target = full[src * ni:(src + 1) * ni]
            landing = own.new_empty(own.shape) if staged else target
            ops = [dist.P2POp(dist.isend, own, dst, group=self.group), dist.P2POp(dist.irecv, landing, src, group=self.group)]
            self.arrivals[src] = (dist.batch_isend_irecv(ops), landing if staged else None, target)

    def _wait_tile(self, peer) -> None:
        works, landing, target = self.arrivals.pop(peer, ((), None, None))
        for work in works:
            work.wait()  # RCCL: the compute stream waits for the round's event, the host does not block
        if landing is not None:
            target.copy_(landing)

    def exchange_once(self, full) -> None:
        """One exchange of `full` outside any step (communicator bring-up / diagnostics)."""
        if self.world == 1:
            return
        if self.exchange == "tiles":
            self._start_tiles(full)
            for peer in list(self.arrivals):
                self._wait_tile(peer)
        else:
            self._gather(full, full[self.i0:self.i0 + self.ni]).wait()

    # ---- the step ----------------------------------------------------------------------------------------------
    def update(self) -> None:
        """One step: pos[1-read][own], vel[own] <- integrate(pos[read]); then start exchanging pos[1-read]."""
        cur, nxt = self.pos[self.read], self.pos[1 - self.read]
        last = len(self.schedule) - 1
        for k, (j0, nj, remote) in enumerate(self.schedule):
            if self.exchange == "tiles":
                if remote is not None:
                    self._wait_tile(remote)
            elif remote and self.pending is not None:
                self.pending.wait()  # compute stream waits for the collective; the host does not block
                self.pending = None
            flags = (NB_SHARD_ACC_IN if k > 0 else 0) | (NB_SHARD_FINALIZE if k == last else 0)
            self.launch(nxt, cur, self.vel, self.acc, self.i0, self.ni, j0, nj, flags)
        self.finish()  # nothing left in flight (only ever something when world == 1 or a tile was not consumed)
        if self.world > 1:
            if self.exchange == "tiles":
                self._start_tiles(nxt)
            else:
                self.pending = self._gather(nxt, nxt[self.i0:self.i0 + self.ni])
        self.read = 1 - self.read

    def finish(self) -> None:
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
        for peer in list(self.arrivals):
            self._wait_tile(peer)

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
