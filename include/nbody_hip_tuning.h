/*
 * nbody_hip_tuning.h -- what libnbody_hip.so exports beyond the drop-in boundary: plan overrides for tuning sweeps and tests, a probe
 * event between the two kernels of a pairwise step (bench.py times the headline kernel with it), counters for tests, and what a
 * communicator can say about its last step (host-order trace, host enqueue time, executed work, the RCCL it is bound to,
 * hardware-queue collisions of its second stream), and the hooks bench.py's own lines use on the
 * product's communicators (a rank's kernels alone, the reaction leg alone, the order of the diagonal).  The lab bench proper -- the real RCCL on a one-GPU box (self-test, loopback rank,
 * in-process world), the stream-placement A/B hook, the allocation-failure hook -- moved to nbody_hip_lab.h in
 * round 6 and is exported by libnbody_hip_lab.so only.
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
 * 0 = one launch, first (the order up to round 4); 2 (round 6) = as 1, and the SECOND compute stream ends on local work too: it takes
 * half of the late offsets as its last kernel and hands the same amount of work -- the first tiles of bodies j of its last rectangle --
 * to the step's own stream (worlds of 4, 8, 12, 16 ranks; measured slower on one GPU, profiles/round6_cut_rectangle_ab.txt: for A/B
 * timings on real links).  Process-global like the plan overrides; with one process per GPU every rank must set the same
 * (nb_comm_set_workspace checks it).  Summation order, hence the last bits, differ between the three. */
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

/* Version (ncclGetVersion) and file of the RCCL a communicator is bound to (0 / "" for a world of one made by nb_comm_init_rank: none
 * bound) -- bench.py's N > 1 line reports it per rank. */
NB_API int nb_comm_transport_info(nb_comm_t comm, int* rccl_version, char* library_path, size_t path_bytes);

/* What this rank EXECUTES in one pairwise multi-GPU step (the communicator's layout must be pairwise for this system: NB_ERR_UNSUPPORTED
 * otherwise): pair evaluations -- whole tiles against whole blocks, as the kernel loops, 24 flop each in fp32 / 36 in fp64 -- and
 * force-kernel launches.  bench.py's N > 1 line states the flop a rank really issues next to the algorithmic count with it. */
NB_API int nb_comm_pair_work_f32(nb_comm_t comm, unsigned num_bodies, unsigned long long* pair_evaluations, int* force_launches);
NB_API int nb_comm_pair_work_f64(nb_comm_t comm, unsigned num_bodies, unsigned long long* pair_evaluations, int* force_launches);

/* The second compute stream of a pairwise multi-GPU step must run BESIDE the caller's stream; the HIP runtime lets streams share a
 * hardware queue once a process has more than a few, and two streams on one queue run one after the other.  A communicator
 * probes its side stream against the caller's the first time the two meet and replaces it while they collide (csrc/nbody_comm.hip,
 * note_stream / settle_side_stream; NBODY_AUX_PROBE=0 switches every probe off).  *collisions: how many candidates were replaced; -1: not probed yet. */
NB_API int nb_comm_side_stream_collisions(nb_comm_t comm, int* collisions);
/* ... and the same probe NOW, against the stream the caller is going to step on (else it runs inside the first step that meets that
 * stream -- once per stream and rank, never while the stream is capturing, never for a stream made by nb_comm_stream_create: two stream
 * synchronisations and ~0.2 ms).  Asked again for a stream already settled, it looks afresh (a recycled stream handle). */
NB_API int nb_comm_settle_side_stream(nb_comm_t comm, nb_stream_t beside);
/* *badly_placed: 1 = the stream the caller stepped on last is the null stream or shares its hardware queue -- with RCCL active such a
 * rank steps ~40 % slower (profiles/round5_hw_queue_collision.txt): step on a stream from nb_comm_stream_create instead; 0 = fine;
 * -1 = not looked at (no step with more than one rank yet, the stream was capturing, or NBODY_AUX_PROBE=0). */
NB_API int nb_comm_caller_stream_placement(nb_comm_t comm, int* badly_placed);
/* What this rank's LAST pairwise multi-GPU step enqueued, in host order, one item per line: "forces diagonal-early", "forces
 * rectangle s", "fold s", "send reaction s", "forces diagonal-late", "finish" (empty before the first such step).  Tests read the
 * ORDER from it: every reaction send is enqueued before the rank's last force kernel. */
NB_API int nb_comm_last_step_trace(nb_comm_t comm, char* text, size_t bytes);

/* What the HOST needed to enqueue the last nb_sharded_step_* call this rank took part in (wall clock of the whole call, all its local
 * ranks: with several local ranks their kernels are enqueued by one thread each, the RCCL groups by the caller).  A step whose enqueue
 * takes longer than its kernels is bound by the host: the CLI's --numdevices run and bench.py report it as host_enqueue_ms_per_step. */
NB_API int nb_comm_last_enqueue_ms(nb_comm_t comm, double* milliseconds);

/* An event recorded between the forces kernel and the finish kernel of every ONE-GPU pairwise step from now on (NULL = none):
 * bench.py times the two kernels of the headline step separately with it, after the timed region. */
NB_API int nb_set_pair_probe_event(nb_event_t event);

/* The clock the headline kernel REALLY runs at.  While device memory is lent here (16 bytes per workgroup of the launch: nb_pair_plan_t.grid_blocks;
 * NULL, 0 takes it back), the one-GPU pairwise step launches pair_forces_clocked instead of pair_forces -- the same kernel with four
 * scalar instructions more: every workgroup notes its lifetime in shader cycles (s_memtime) and in ticks of the constant 100 MHz counter
 * (s_memrealtime) as two 64-bit words at device_words[2 * workgroup].  Clock in MHz = 100 * cycles / ticks (median over the
 * workgroups).  Only the geometry of the headline sizes has the variant (8 vectors per lane, 8 waves: 65 536 bodies and more in fp32);
 * other launches ignore the words.  bench.py reads it after its timed region: the power management's own figure (hwmon) reads higher
 * and lags, and nothing launched BEHIND the kernel can tell -- the chip raises its clock within microseconds of the kernel's exit
 * (profiles/round6_delivered_clock.txt). */
NB_API int nb_set_pair_clock_words(void* device_words, size_t bytes);

/* The device memory the library assumes when it decides whether a workspace is affordable (at most a third of it is ever asked
 * for): 0 = the device's own total; tests of the guard set a small figure. */
NB_API int nb_set_memory_budget(size_t bytes);

/* How many (kernel, device) pairs have been granted more than 64 KiB of dynamic LDS so far (the opt-in is made once per
 * kernel instantiation and device, on first use, and before any graph capture). */
NB_API int nb_lds_optin_count(int* count);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_HIP_TUNING_H */
