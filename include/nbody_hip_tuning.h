/*
 * nbody_hip_tuning.h -- the lab bench of libnbody_hip.so: plan overrides for tuning sweeps, the kernel-time projection of one
 * rank of a multi-GPU step, a probe event between the two kernels of a pairwise step, counters for tests; round 5: the REAL RCCL on
 * a one-GPU box (a self-loop with every byte checked, a loopback rank that steps as rank r of a nominal G-rank communicator) and
 * what a communicator can say about its last step (host-order trace, executed work, hardware-queue collisions of its second stream).
 *
 * NOT part of the drop-in boundary (that is nbody_hip.h: what a maintainer of the reference binds).  Everything declared here
 * is PROCESS-GLOBAL state and NOT THREAD-SAFE against steps running concurrently: an override set while another thread is
 * inside nb_integrate_* / nb_sharded_step_* changes that thread's launch geometry (results stay correct -- every geometry is
 * tested -- but the step is no longer the one that was planned), and with one process per GPU the ranks of a communicator must
 * set the same overrides (nb_comm_set_workspace checks it).  Used by tests/, bench.py's diagnostics and tools/.
 */
#ifndef NBODY_HIP_TUNING_H
#define NBODY_HIP_TUNING_H

#include "nbody_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Override the automatic plan of the one-sided FAST kernels (0 = automatic) -- tools/kernel_sweeps.py. */
NB_API int nb_set_plan_override(int bodies_per_lane, int lanes_per_body, int tile_bodies);

/* ... and of the pairwise layout (0 = automatic); min_bodies: smallest system that takes the pairwise layout (tests run it at
 * the golden sizes with min_bodies = 1). */
NB_API int nb_set_pair_plan_override(int vectors_per_lane, int waves_per_block, int splits, int min_bodies);

/* Force the number of slices of the one-GPU pairwise step (0 = automatic, 1 = one tournament or nothing, 2 .. 15): tests run the
 * sliced form at sizes where one tournament would fit. */
NB_API int nb_set_pair_slices_override(int slices);

/* Smallest slice (bodies per rank) for which a multi-GPU step goes pairwise across the ranks; 0 = automatic (2 048). */
NB_API int nb_comm_set_pair_min_slice(int min_bodies_per_rank);

/* A rank's diagonal (its own slice against itself) in a pairwise multi-GPU step: 1 (default) = two launches (for slices up to 65 536
 * bodies: beyond, the hop it hides is under 0.3 % of a step and the second launch costs as much), the first half of the
 * block offsets first and the rest as the rank's LAST force kernel, so that the last reaction sums travel under local work;
 * 0 = one launch, first (the order up to round 4) -- for A/B timings.  Process-global like the plan overrides; with one process
 * per GPU every rank must set the same (nb_comm_set_workspace checks it).  Summation order, hence the last bits, differ. */
NB_API int nb_set_late_diagonal(int on);

/* Tuning / projection hook (bench.py --emulate-gpus): exactly the kernels that rank `rank` of a `world_size`-rank pairwise step
 * launches, on the current device, with no communicator and no exchange (what the other ranks would send is whatever the
 * workspace holds: the positions written are meaningless, the kernel time is the point).  workspace == NULL: *workspace_bytes
 * is set to what the rank needs. */
NB_API int nb_emulate_pair_rank_f32(float* new_positions, const float* old_positions, float* velocities, void* workspace, size_t* workspace_bytes,
                                    unsigned num_bodies, int world_size, int rank, float delta_time, float damping, nb_stream_t stream);
NB_API int nb_emulate_pair_rank_f64(double* new_positions, const double* old_positions, double* velocities, void* workspace, size_t* workspace_bytes,
                                    unsigned num_bodies, int world_size, int rank, double delta_time, double damping, nb_stream_t stream);

/* The reaction leg of a pairwise multi-GPU step on its own (one process per rank; the communicator's layout must be pairwise for
 * this system): the G/2 send/recv rounds of N/G x 12 B that carry reaction sums to their owners, on whatever the workspace holds
 * -- bench.py's diagnostics time it to say what the second exchange leg costs.  A collective like the step itself. */
NB_API int nb_comm_reaction_exchange_f32(nb_comm_t comm, unsigned num_bodies, nb_stream_t stream);
NB_API int nb_comm_reaction_exchange_f64(nb_comm_t comm, unsigned num_bodies, nb_stream_t stream);

/* ---- the REAL transport on a one-GPU box.  RCCL refuses two ranks on one device, so everything with more than one rank is
 * tested against a transport double (tests/fake_rccl); what one GPU can still prove against the real library is the binding
 * itself -- the hand-resolved entry points, the dlopen inside whatever process this is, the stream and event semantics.
 * nb_comm_selftest_open: a communicator of ONE rank that does own an RCCL communicator (nb_comm_init_rank binds no transport
 * for a world of one); `id` from nb_comm_unique_id.  nb_comm_selftest_f32: on that communicator, `bytes` of a known pattern
 * produced on `stream`; on the communicator's exchange stream, after the `ready` event: GroupStart, Send(to self),
 * Recv(from self), GroupEnd, the tile's event, which `stream` waits for before it reads the bytes back; then ncclAllGather out
 * of place and in place the same way.  Returns 0 when every call was accepted and every byte arrived; the report says which
 * call refused (status = NB_ERR_RCCL_BASE + ncclResult_t) or how many bytes differ.  Blocking.
 * nb_comm_self_transfer_f32: the measuring form (tools/exchange_contention.py): `rounds` self send/recv pairs of `count` floats
 * each in one group or a group per round, asynchronous, `begin` / `end` recorded on the exchange stream around them.
 * nb_comm_transport_info: version (ncclGetVersion) and file of the RCCL a communicator is bound to (0 / "" for a world of one
 * made by nb_comm_init_rank: none bound). */
typedef struct nb_comm_selftest {
    int    rccl_version;
    int    send_recv_status, all_gather_status;
    float  send_recv_ms, all_gather_ms;            /* on the exchange stream, events around the calls */
    size_t send_recv_wrong_bytes, all_gather_wrong_bytes;
    char   refused_call[64];
    char   library_path[256];
} nb_comm_selftest_t;
NB_API int nb_comm_selftest_open(nb_comm_t* comm, const void* id /* NB_COMM_ID_BYTES */);
/* A LOOPBACK rank: rank `nominal_rank` of a `nominal_world`-rank communicator whose RCCL communicator has one rank -- every
 * send goes to, every receive comes from, the rank itself.  nb_sharded_step_* on it launches exactly the kernels, RCCL calls,
 * events and waits of that rank of a real multi-GPU step, on one GPU, with the real RCCL kernels competing for the chip; what
 * "arrives" is the rank's own data, so the positions are meaningless after the first step and only the TIME means anything:
 * a real step minus what the xGMI links would add (tools/exchange_contention.py, bench.py's one-GPU projection).
 * nb_comm_set_workspace stays the collective it is (the notes travel to the rank itself). */
NB_API int nb_comm_loopback_open(nb_comm_t* comm, const void* id /* NB_COMM_ID_BYTES */, int nominal_world, int nominal_rank);
/* An IN-PROCESS world: `world` ranks in this process, all on the current device, sharing ONE real one-rank RCCL communicator -- every
 * transfer a self-transfer, rank a's send routed to rank b's receive by the order in which the library issues them (RCCL matches the
 * sends and receives of one peer first in, first out).  nb_sharded_step_all_* on these comms is the full G-rank step -- even G, the
 * split rectangle and all -- through the product's own calls into the REAL library, comparable with the CPU path
 * (tests/test_comm_fake_rccl.py runs its `all` cases this way too).  Destroy every rank with nb_comm_destroy. */
NB_API int nb_comm_inprocess_open_all(nb_comm_t* comms /* [world] */, int world, const void* id /* NB_COMM_ID_BYTES */);
NB_API int nb_comm_selftest_f32(nb_comm_t comm, size_t bytes, nb_stream_t stream, nb_comm_selftest_t* report);
NB_API int nb_comm_self_transfer_f32(nb_comm_t comm, const float* src, float* dst, size_t count, int rounds, int one_group, nb_stream_t after,
                                     nb_event_t begin, nb_event_t end);
NB_API int nb_comm_transport_info(nb_comm_t comm, int* rccl_version, char* library_path, size_t path_bytes);

/* What this rank EXECUTES in one pairwise multi-GPU step (the communicator's layout must be pairwise for this system: NB_ERR_UNSUPPORTED
 * otherwise): pair evaluations -- whole tiles against whole blocks, as the kernel loops, 24 flop each in fp32 / 36 in fp64 -- and
 * force-kernel launches.  bench.py's N > 1 line states the flop a rank really issues next to the algorithmic count with it. */
NB_API int nb_comm_pair_work_f32(nb_comm_t comm, unsigned num_bodies, unsigned long long* pair_evaluations, int* force_launches);
NB_API int nb_comm_pair_work_f64(nb_comm_t comm, unsigned num_bodies, unsigned long long* pair_evaluations, int* force_launches);

/* The second compute stream of a pairwise multi-GPU step must run BESIDE the caller's stream; the HIP runtime lets streams share a
 * hardware queue once a process has more than a few, and two streams on one queue run one after the other.  A communicator
 * probes its side stream against the caller's the first time the two meet and replaces it while they collide (csrc/nbody_comm.hip,
 * settle_side_stream; NBODY_AUX_PROBE=0 switches that off).  *collisions: how many candidates were replaced; -1: not probed yet. */
NB_API int nb_comm_side_stream_collisions(nb_comm_t comm, int* collisions);
/* ... and the same probe NOW, against the stream the caller is going to step on (else it runs inside the first pairwise step with two or
 * more partners: two stream synchronisations and ~0.2 ms, once). */
NB_API int nb_comm_settle_side_stream(nb_comm_t comm, nb_stream_t beside);
/* *badly_placed: 1 = the stream the caller stepped on last is the null stream or shares its hardware queue -- with RCCL active such a
 * rank steps ~40 % slower (profiles/round5_hw_queue_collision.txt): step on a stream from nb_comm_stream_create instead; 0 = fine;
 * -1 = no step with more than one rank yet. */
NB_API int nb_comm_caller_stream_placement(nb_comm_t comm, int* badly_placed);
/* Retire the second compute stream and make another (experiments on how much its placement matters: tools/side_stream_placement.py). */
NB_API int nb_comm_replace_side_stream(nb_comm_t comm);

/* What this rank's LAST pairwise multi-GPU step enqueued, in host order, one item per line: "forces diagonal-early", "forces
 * rectangle s", "fold s", "send reaction s", "forces diagonal-late", "finish" (empty before the first such step).  Tests read the
 * ORDER from it: every reaction send is enqueued before the rank's last force kernel. */
NB_API int nb_comm_last_step_trace(nb_comm_t comm, char* text, size_t bytes);

/* An event recorded between the forces kernel and the finish kernel of every ONE-GPU pairwise step from now on (NULL = none):
 * bench.py times the two kernels of the headline step separately with it, after the timed region. */
NB_API int nb_set_pair_probe_event(nb_event_t event);

/* The device memory the library assumes when it decides whether a workspace is affordable (at most a third of it is ever asked
 * for): 0 = the device's own total; tests of the guard set a small figure. */
NB_API int nb_set_memory_budget(size_t bytes);

/* Tests of the out-of-memory fall-backs (halve the workspace and ask again; step without one): every nb_alloc request above
 * `bytes` is refused BY THE RUNTIME (the request is replaced by one no device can serve), 0 = no limit.  The CLI's
 * --alloc-limit-mib sets it. */
NB_API int nb_set_alloc_limit(size_t bytes);

/* How many (kernel, device) pairs have been granted more than 64 KiB of dynamic LDS so far (the opt-in is made once per
 * kernel instantiation and device, on first use, and before any graph capture). */
NB_API int nb_lds_optin_count(int* count);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_HIP_TUNING_H */
