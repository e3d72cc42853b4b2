/*
 * nbody_hip.h -- C-ABI of the MI355X (gfx950) all-pairs N-body hot path.
 *
 * This is the drop-in boundary for the reference's device seam: plain pointers and sizes, no C++,
 * HIP or torch types.  The library (libnbody_hip.so) keeps the reference's ownership rules: the
 * CALLER owns every device array and passes raw device pointers; the callee allocates nothing per
 * call and keeps no state except the two per-precision softening^2 values (process-global, exactly
 * like the reference's __constant__ pair).
 *
 * Every entry point returns 0 on success or a non-zero hipError_t value (NB_ERR_* for argument
 * errors detected on the host); nb_error_string() names it.  Nothing here prints or exits: the
 * reference's print+exit(1) / throw behaviour is reproduced by the C++ wrappers in
 * cuda-nbody_amd/host/integrate_nbody_hip.hpp so a host can keep its own error policy.
 *
 * The reference draws its initial conditions from the process-global libc rand() stream, and the HIP runtime
 * consumes draws from it (its first pageable host-to-device copy does).  The entry points that set up state --
 * device selection and query, allocation, copies, the blocking synchronize calls, graph creation, communicator creation and
 * the RCCL exchange -- therefore park the caller's rand() state for the duration of the call (a process-wide lock around
 * initstate/setstate: another thread calling rand() inside that window draws from a scratch state).  The launch path --
 * nb_integrate_*, nb_integrate_shard_*, nb_graph_launch, nb_event_record, nb_stream_wait_event, and nb_sharded_step_* in
 * a world of one -- takes no lock and touches no such state.  The first call on a device (normally nb_set_device or
 * nb_alloc) warms the runtime up once, under that guard, with a small allocation and copy; a caller that allocates
 * with its own hipMalloc and wants to capture nb_integrate_* into a graph of its own calls nb_set_device first.
 *
 * All reference citations are relative to j-horner/cuda-nbody (/root/reference/).
 *
 * Data layout (identical to the reference's, src/nbody/bodysystemcuda.cu:148):
 *   positions  T[4*N]  = {x, y, z, mass} per body   (float4 / double4, 16-/32-byte aligned)
 *   velocities T[4*N]  = {vx, vy, vz, unused} per body (.w is preserved, never interpreted)
 */
#ifndef NBODY_HIP_H
#define NBODY_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NB_API __attribute__((visibility("default")))

/* The one hipError_t value a host has to tell from the others: nb_alloc / nb_host_alloc_mapped could not get the memory
 * (thrust::device_vector throws a std::bad_alloc there, which the reference's main turns into exit code 3: nbody.cpp:396-408). */
#define NB_ERR_OUT_OF_MEMORY 2 /* = hipErrorOutOfMemory */

/* Host-side argument errors (outside hipError_t's range). */
#define NB_ERR_INVALID_ARGUMENT 10001
#define NB_ERR_UNSUPPORTED      10002
#define NB_ERR_RCCL_BASE        20000 /* 20000 + ncclResult_t for a failed RCCL call of the multi-GPU entry points */

/* Arithmetic mode of the integrate entry points. */
enum {
    /* Bit-reproduces the reference's CPU BodySystem path (src/nbody/bodysystemcpu.cpp:140-303): one lane
     * per body i, j = 0..N-1 in order, the CPU's exact op order, IEEE sqrt and divide, no FMA contraction. */
    NB_MODE_STRICT = 0,
    /* The production kernel: bodyBodyInteraction's op sequence (src/nbody/bodysystemcuda.cu:98-123) with
     * v_rsq / FMA, register-tiled and j-split; summation order differs from the CPU path. */
    NB_MODE_FAST = 1
};

/* Flags of nb_integrate_shard_*. */
enum {
    NB_SHARD_ACC_IN   = 1, /* start from the partial accelerations in acc[] instead of zero             */
    NB_SHARD_FINALIZE = 2  /* integrate and store new_pos / vel; otherwise store partial sums to acc[]  */
};

typedef void* nb_stream_t; /* hipStream_t; NULL = the default stream (what the reference uses)          */
typedef void* nb_event_t;  /* hipEvent_t                                                                */

typedef struct nb_device_info {
    char   name[256];      /* marketing name, e.g. "AMD Instinct MI355X"                                */
    char   arch[64];       /* gcnArchName, e.g. "gfx950:sramecc+:xnack-"                                */
    int    compute_units;  /* replaces multiprocessor_count(), src/nbody/compute_cuda.cpp:113           */
    int    wavefront_size;
    int    clock_khz;
    int    can_map_host_memory; /* replaces can_map_host_memory(), compute_cuda.cpp:78                  */
    int    lds_bytes_per_cu;
    size_t total_memory;
} nb_device_info_t;

/* ---- errors ------------------------------------------------------------------------------------ */
NB_API const char* nb_error_string(int code);            /* replaces cudaGetErrorName/String, bodysystemcuda.cu:50,212 */

/* ---- device query (replaces cuda-api-wrappers use in src/nbody/compute_cuda.cpp:16-48,68-96,113) -- */
NB_API int nb_device_count(int* count);
NB_API int nb_set_device(int device);
NB_API int nb_get_device(int* device);
NB_API int nb_device_info(int device, nb_device_info_t* out);

/* ---- storage (replaces thrust::device_vector + thrust::copy, bodysystemcuda_default.hpp:34-35,
 *      bodysystemcuda_default.cu:26-55; mapped host memory replaces unique_mapped_span.cpp:11-27) ----- */
NB_API int nb_alloc(void** device_ptr, size_t bytes);
NB_API int nb_free(void* device_ptr);
NB_API int nb_memset(void* device_ptr, int value, size_t bytes, nb_stream_t stream);
NB_API int nb_h2d(void* device_dst, const void* host_src, size_t bytes, nb_stream_t stream); /* blocking */
NB_API int nb_d2h(void* host_dst, const void* device_src, size_t bytes, nb_stream_t stream); /* blocking */
NB_API int nb_d2d(void* device_dst, const void* device_src, size_t bytes, nb_stream_t stream); /* async  */
NB_API int nb_host_alloc_mapped(void** host_ptr, void** device_ptr, size_t bytes);
NB_API int nb_host_free(void* host_ptr);

/* ---- streams and events (replaces cuda::event_t, compute_cuda.cpp:66-67,149,188,239,263-271) ------ */
NB_API int nb_stream_create(nb_stream_t* stream);
NB_API int nb_stream_destroy(nb_stream_t stream);
NB_API int nb_stream_synchronize(nb_stream_t stream);
NB_API int nb_stream_wait_event(nb_stream_t stream, nb_event_t event);
NB_API int nb_event_create(nb_event_t* event);
NB_API int nb_event_destroy(nb_event_t event);
NB_API int nb_event_record(nb_event_t event, nb_stream_t stream);
NB_API int nb_event_synchronize(nb_event_t event);
NB_API int nb_event_elapsed_ms(float* ms, nb_event_t start, nb_event_t stop);
NB_API int nb_device_synchronize(void);                  /* replaces cudaDeviceSynchronize, compute_cuda.cpp:155,180 */

/* ---- softening (replaces set_softening_squared(float|double), decl src/nbody/bodysystemcuda.cpp:37-38,
 *      def src/nbody/bodysystemcuda.cu:46-60).  Host-side state handed to each launch as a kernel
 *      argument; one value per precision, as in the reference. ---------------------------------------- */
NB_API int nb_set_softening_sq_f32(float softening_sq);
NB_API int nb_set_softening_sq_f64(double softening_sq);
NB_API int nb_get_softening_sq_f32(float* softening_sq);
NB_API int nb_get_softening_sq_f64(double* softening_sq);

/* ---- the hot path (replaces integrateNbodySystem<T>, decl src/nbody/integrate_nbody_cuda.hpp:5,
 *      def src/nbody/bodysystemcuda.cu:186-215, explicit instantiations :218-220).
 *
 *  One leapfrog-style step for all N bodies: a_i = sum_j m_j r_ij / (|r_ij|^2 + eps^2)^(3/2)
 *  (self-interaction included, exactly 0 for eps > 0), v = (v + a*dt)*damping, p += v*dt, written to
 *  new_positions (must not alias old_positions) and, in place, to velocities.  Asynchronous on
 *  `stream`.  Unlike the reference kernel (wrong unless N % blockSize == 0, :153-155) any N >= 1 works.
 *  `block_size` is the reference's --blockSize (a multiple of 64, <= 1024, validated in STRICT mode): results
 *  never depend on it and both modes pick their own launch geometry, so it is a hint only. ------------ */
NB_API int nb_integrate_f32(float* new_positions, const float* old_positions, float* velocities,
                            float delta_time, float damping, unsigned num_bodies, int block_size,
                            int mode, nb_stream_t stream);
NB_API int nb_integrate_f64(double* new_positions, const double* old_positions, double* velocities,
                            double delta_time, double damping, unsigned num_bodies, int block_size,
                            int mode, nb_stream_t stream);

/* ---- the same step with a caller-owned WORKSPACE (new).  The reference kernel evaluates every directed interaction
 *  (src/nbody/bodysystemcuda.cu:98-146: N^2 calls of bodyBodyInteraction per step).  Given scratch memory the FAST mode
 *  evaluates every PAIR of bodies once and applies it to both (csrc/nbody_pair.hip: the bodies j and their reaction sums
 *  rotate through the wavefront, the reaction sums leave through the workspace and a second kernel adds them in a fixed
 *  order and integrates): same result up to summation order, about half the arithmetic.  The ownership rule stays the
 *  reference's -- the library allocates nothing: the caller asks nb_workspace_bytes_*(N, mode) once (0 = this N / mode has
 *  no use for a workspace), allocates that much device memory, and passes it to every step.  The contents need not be
 *  preserved or cleared between steps, and one workspace serves any number of systems of that size on one stream.
 *  With workspace == NULL, too few bytes, STRICT mode or a system too small to gain, nb_integrate_ws_* IS nb_integrate_*.
 *  Which of the two a call takes is a function of (num_bodies, mode, workspace_bytes) alone: nb_pair_plan_* says whether the
 *  pairwise layout applies to this N and how many bytes it asks for; given fewer bytes the step cuts the tournament into more
 *  slices (nb_workspace_bytes_capped_* says which sizes are useful), given too few for any form it is the one-sided kernel.
 *  The workspace must not overlap any of the three body arrays (NB_ERR_INVALID_ARGUMENT). */
NB_API int nb_workspace_bytes_f32(unsigned num_bodies, int mode, size_t* bytes);
NB_API int nb_workspace_bytes_f64(unsigned num_bodies, int mode, size_t* bytes);
/* The workspace of ONE tournament over the whole system grows with N^2 (0.4 GB at 262 144 bodies fp32, 6.4 GB at 1 Mi, 103 GB at
 * 4 Mi).  When that is more than a third of the device's memory -- or more than the caller wants to spend -- the tournament is cut
 * into K slices of bodies that share one reusable region of reaction planes (12 GB at 4 Mi bodies in four slices, 7 GB in eight; K <= 15):
 * same pairs, same arithmetic, a few more launches.  nb_workspace_bytes_* asks for the fewest slices the device affords;
 * nb_workspace_bytes_capped_* for the fewest whose workspace stays within max_bytes (0 = nothing fits: the one-sided kernel);
 * nb_integrate_ws_* takes the fewest slices that fit the workspace_bytes it is given.  nb_pair_plan_t.slices reports K. */
NB_API int nb_workspace_bytes_capped_f32(unsigned num_bodies, int mode, size_t max_bytes, size_t* bytes);
NB_API int nb_workspace_bytes_capped_f64(unsigned num_bodies, int mode, size_t max_bytes, size_t* bytes);
NB_API int nb_integrate_ws_f32(float* new_positions, const float* old_positions, float* velocities,
                               float delta_time, float damping, unsigned num_bodies, int block_size, int mode,
                               void* workspace, size_t workspace_bytes, nb_stream_t stream);
NB_API int nb_integrate_ws_f64(double* new_positions, const double* old_positions, double* velocities,
                               double delta_time, double damping, unsigned num_bodies, int block_size, int mode,
                               void* workspace, size_t workspace_bytes, nb_stream_t stream);

/* ---- sharded form for multi-GPU body sharding (new; the reference is single-GPU).
 *
 *  Bodies i in [i_begin, i_begin+i_count) accumulate the pull of bodies j in [j_begin, j_begin+j_count)
 *  of old_positions (all arrays are full-size and indexed by global body id).  acc is T[4*N] scratch
 *  holding partial accelerations between calls (flags: NB_SHARD_ACC_IN / NB_SHARD_FINALIZE).
 *  nb_integrate_* == one shard call with i = j = [0,N), flags = NB_SHARD_FINALIZE.
 *  In STRICT mode chunks must be issued in ascending j order to keep the CPU path's summation order.
 *  old_positions is read-only for the whole launch (the kernels read it through the scalar cache): the bodies a call
 *  writes -- [i_begin, i_begin+i_count) of new_positions and velocities, or of acc -- must not overlap the bodies i and j
 *  it reads of old_positions (NB_ERR_INVALID_ARGUMENT otherwise). */
NB_API int nb_integrate_shard_f32(float* new_positions, const float* old_positions, float* velocities, float* acc,
                                  unsigned i_begin, unsigned i_count, unsigned j_begin, unsigned j_count,
                                  unsigned flags, float delta_time, float damping, int block_size, int mode,
                                  nb_stream_t stream);
NB_API int nb_integrate_shard_f64(double* new_positions, const double* old_positions, double* velocities, double* acc,
                                  unsigned i_begin, unsigned i_count, unsigned j_begin, unsigned j_count,
                                  unsigned flags, double delta_time, double damping, int block_size, int mode,
                                  nb_stream_t stream);

/* ---- multi-GPU: bodies sharded over the GPUs of one node, position tiles exchanged over RCCL / xGMI (new: the reference
 *  is single-GPU).  Rank r of G owns bodies [r*N/G, (r+1)*N/G) -- N must be a multiple of G; pad with zero-mass bodies
 *  as tipsy.cpp:111-119 does -- i.e. their velocities and their slice of every new position array; all arrays stay
 *  full-size.  The one exchange step is the all-gather of the new positions, issued as G-1 position TILES: in round s
 *  rank r sends its slice to r-s and receives the slice of r+s (RCCL send/recv pairs on the communicator's own
 *  high-priority stream, a group per round or all rounds of a step in one RCCL group -- nb_comm_set_exchange_grouping --, an
 *  event per tile).  nb_sharded_step_* = the kernels of the own slice,
 *  then of each tile as it arrives (STRICT: ascending rank order, bit-identical to one GPU), integrate, and the start
 *  of the exchange of new_positions -- everything asynchronous; the caller ping-pongs the two position arrays exactly
 *  as with nb_integrate_*.  Process models: one process per GPU (nb_comm_unique_id on rank 0, ship the 128 bytes to
 *  the others by any means, nb_comm_init_rank on each) or one process driving several GPUs (nb_comm_init_all +
 *  the *_all step, which takes one array of each kind per local device; its local ranks are enqueued in parallel by a crew of
 *  persistent threads, one per rank -- each issuing its own rank's RCCL groups, RCCL's thread-per-device model -- so the host needs
 *  what ONE rank needs whatever the number of devices; NBODY_STEP_THREADS=0: the calling thread alone, same bits).
 *  RCCL is loaded on first use. ------------------------------------------------------------------------------------------------- */
#define NB_COMM_ID_BYTES 128
typedef void* nb_comm_t;
NB_API int nb_comm_unique_id(void* id /* NB_COMM_ID_BYTES */);
NB_API int nb_comm_init_rank(nb_comm_t* comm, const void* id, int world_size, int rank);  /* on the current device */
NB_API int nb_comm_init_all(nb_comm_t* comms /* [num_devices] */, int num_devices, const int* devices /* NULL = 0..n-1 */);
NB_API int nb_comm_destroy(nb_comm_t comm);
NB_API int nb_comm_info(nb_comm_t comm, int* rank, int* world_size, int* device);
/* A stream to step this rank on (destroy it with nb_stream_destroy): non-blocking, and well PLACED.  The HIP runtime maps streams onto
 * a few hardware queues; RCCL puts work of its own on the null stream, and a rank that computes on the null stream or on a stream
 * that shares its queue -- about one created stream in three -- steps ~40 % slower (measured with the real RCCL next to the force
 * kernels, profiles/round5_hw_queue_collision.txt).  This one is probed to be clear of that queue.  Any stream works; this one is fast.
 * (The library looks at any OTHER stream a rank steps on once per stream -- two 40 us spin kernels and a synchronisation of that stream and
 * of the null stream inside the first nb_sharded_step_* that sees it, never while the stream is capturing, not at all with
 * NBODY_AUX_PROBE=0 -- to be able to say so: nb_comm_caller_stream_placement, tuning header.  The placement of a stream made here is not looked at again; the rank's second
 * compute stream is settled beside it once, at the first pairwise step with two or more partners.) */
NB_API int nb_comm_stream_create(nb_comm_t comm, nb_stream_t* stream);
/* ... the same without a communicator, on the current device: for a host that runs an RCCL of its own next to these kernels. */
NB_API int nb_stream_create_placed(nb_stream_t* stream);
/* Lend a rank scratch memory (caller-owned, as everywhere; nb_comm_workspace_bytes_* says how much for this communicator,
 * 0 = none needed).  FAST mode then evaluates every PAIR of bodies once, across the ranks too: a communicator of one rank
 * steps through nb_integrate_ws_*; with G ranks, rank r evaluates its own slice against itself and against the slices of
 * ranks r+1 .. r+G/2 pairwise, keeps its own bodies' sums and sends the reaction sums (N/G * 12 B per partner, one more
 * RCCL send/recv round per partner on the exchange stream) to their owners -- half the arithmetic per rank.  STRICT never uses it.
 *
 * The layout of a step is a property of the COMMUNICATOR, never of one rank: a rank that stepped one-sidedly while its peers
 * stepped pairwise would leave them inside unmatched send/recv rounds.  So with several ranks nb_comm_set_workspace is a
 * COLLECTIVE over the communicator: every rank calls it (a rank without memory passes NULL, 0), the ranks exchange what they
 * were lent, and every rank learns the smallest amount -- the step is pairwise if and only if that is enough for the plan
 * (nb_comm_layout_* gives the answer; it is the same on every rank).  One process per GPU: the call blocks until every rank
 * of the communicator has made it (it synchronises the communicator's exchange stream).  One process driving all ranks
 * (nb_comm_init_all): call it once per rank in any order; nothing is exchanged.  Ranks whose process-global plan overrides
 * (nbody_hip_tuning.h) differ get NB_ERR_INVALID_ARGUMENT from the call, all of them.  Before the first call the layout is
 * one-sided.  A workspace larger than a third of the device's memory is never asked for (nb_comm_workspace_bytes_* says 0). */
NB_API int nb_comm_workspace_bytes_f32(nb_comm_t comm, unsigned num_bodies, int mode, size_t* bytes);
NB_API int nb_comm_workspace_bytes_f64(nb_comm_t comm, unsigned num_bodies, int mode, size_t* bytes);
NB_API int nb_comm_set_workspace(nb_comm_t comm, void* workspace, size_t workspace_bytes);
/* *pairwise = 1: nb_sharded_step_* of this communicator, for this system and mode, evaluates every pair once (see above);
 * 0: the one-sided tile schedule.  The same answer on every rank. */
NB_API int nb_comm_layout_f32(nb_comm_t comm, unsigned num_bodies, int mode, int* pairwise);
NB_API int nb_comm_layout_f64(nb_comm_t comm, unsigned num_bodies, int mode, int* pairwise);
/* How the G-1 position rounds of a step are issued, per communicator (every rank must choose the same): one_group = 0 (default
 * since round 5; the environment variable NBODY_EXCHANGE_ONE_GROUP=1 flips the default): a group and an event per round -- the kernel
 * of tile k can start while round k+1 is still moving; one_group = 1: all rounds in ONE RCCL group -- one RCCL kernel per step,
 * every tile's event fires when the whole exchange is done.  Same data, same bits either way.  Measured with the real RCCL next to
 * the force kernels on one GPU (profiles/round5_exchange_contention.jsonl): a group per round is never slower and 5 % faster for
 * 262 144 bodies over 8 ranks.  nb_comm_get_exchange_grouping reports the current setting. */
NB_API int nb_comm_set_exchange_grouping(nb_comm_t comm, int one_group);
NB_API int nb_comm_get_exchange_grouping(nb_comm_t comm, int* one_group);
NB_API int nb_sharded_step_f32(nb_comm_t comm, float* new_positions, const float* old_positions, float* velocities, float* acc,
                               unsigned num_bodies, float delta_time, float damping, int block_size, int mode, nb_stream_t stream);
NB_API int nb_sharded_step_f64(nb_comm_t comm, double* new_positions, const double* old_positions, double* velocities, double* acc,
                               unsigned num_bodies, double delta_time, double damping, int block_size, int mode, nb_stream_t stream);
NB_API int nb_sharded_step_all_f32(const nb_comm_t* comms, int num_local, float* const* new_positions, const float* const* old_positions,
                                   float* const* velocities, float* const* acc, unsigned num_bodies, float delta_time, float damping,
                                   int block_size, int mode, const nb_stream_t* streams);
NB_API int nb_sharded_step_all_f64(const nb_comm_t* comms, int num_local, double* const* new_positions, const double* const* old_positions,
                                   double* const* velocities, double* const* acc, unsigned num_bodies, double delta_time, double damping,
                                   int block_size, int mode, const nb_stream_t* streams);
/* The exchange on its own (what nb_sharded_step_* ends with): tiles of `positions` start moving once the work already
 * enqueued on `after_stream` is done; nb_exchange_wait_tile makes `stream` wait for the tile of rank `peer`.
 * nb_allgather_* is the same exchange as ONE in-place ncclAllGather (every tile then arrives with it). */
NB_API int nb_exchange_tiles_f32(nb_comm_t comm, float* positions, unsigned num_bodies, nb_stream_t after_stream);
NB_API int nb_exchange_tiles_f64(nb_comm_t comm, double* positions, unsigned num_bodies, nb_stream_t after_stream);
NB_API int nb_allgather_f32(nb_comm_t comm, float* positions, unsigned num_bodies, nb_stream_t after_stream);
NB_API int nb_allgather_f64(nb_comm_t comm, double* positions, unsigned num_bodies, nb_stream_t after_stream);
NB_API int nb_exchange_wait_tile(nb_comm_t comm, int peer, nb_stream_t stream);
NB_API int nb_exchange_wait_all(nb_comm_t comm, nb_stream_t stream);

/* ---- hipGraph form of the step loop (new).  Small systems are launch-bound (a 1 024-body step is ~2 us of GPU work):
 *  `steps` consecutive nb_integrate_* launches, ping-ponging between position_a (read first) and position_b, are
 *  captured once and replayed with one host call.  `steps` must be even so every replay starts from position_a again.
 *  The softening^2 current at creation time is baked in.  The arrays must outlive the graph. ------------------------ */
typedef void* nb_graph_t;
NB_API int nb_graph_create_f32(nb_graph_t* graph, float* position_a, float* position_b, float* velocities,
                               float delta_time, float damping, unsigned num_bodies, int block_size, int mode, unsigned steps);
NB_API int nb_graph_create_f64(nb_graph_t* graph, double* position_a, double* position_b, double* velocities,
                               double delta_time, double damping, unsigned num_bodies, int block_size, int mode, unsigned steps);
NB_API int nb_graph_create_ws_f32(nb_graph_t* graph, float* position_a, float* position_b, float* velocities,
                                  float delta_time, float damping, unsigned num_bodies, int block_size, int mode, unsigned steps,
                                  void* workspace, size_t workspace_bytes);   /* the steps are nb_integrate_ws_* */
NB_API int nb_graph_create_ws_f64(nb_graph_t* graph, double* position_a, double* position_b, double* velocities,
                                  double delta_time, double damping, unsigned num_bodies, int block_size, int mode, unsigned steps,
                                  void* workspace, size_t workspace_bytes);
NB_API int nb_graph_launch(nb_graph_t graph, nb_stream_t stream);
NB_API int nb_graph_destroy(nb_graph_t graph);

/* ---- introspection: the launch geometry the FAST path would use for a shard (tests / DESIGN.md) ---- */
typedef struct nb_launch_plan {
    int bodies_per_lane;   /* I  : i-bodies register-tiled per lane                  */
    int lanes_per_body;    /* S  : wave groups of one workgroup that split the j range; 64 = wave-split layout
                              (small shards): the 64 lanes of a wave split j and bodies_per_lane counts per WAVE */
    int tile_bodies;       /* bodies j a workgroup takes per round (S chunks; an LDS tile in the wave-split layout) */
    int block_threads;
    unsigned grid_blocks;
    unsigned lds_bytes;
} nb_launch_plan_t;
NB_API int nb_plan_f32(unsigned i_count, unsigned j_count, nb_launch_plan_t* plan);
NB_API int nb_plan_f64(unsigned i_count, unsigned j_count, nb_launch_plan_t* plan);

/* ... and of the pairwise layout behind nb_integrate_ws_* */
typedef struct nb_pair_plan {
    int      applies;          /* 1: nb_integrate_ws_* takes the pairwise layout for this N (FAST mode)              */
    int      bodies_per_lane;  /* I: bodies i a lane holds for good                                                  */
    int      waves_per_block;  /* S: waves of a workgroup; they share the bodies i and split the tiles of bodies j   */
    unsigned splits;           /* C: workgroups per block of bodies i                                                */
    unsigned blocks;           /* NB = ceil(N / block_bodies); block a meets blocks a .. a + NB/2 (mod NB)           */
    unsigned block_bodies;     /* 64 * I                                                                             */
    unsigned reaction_slots;   /* per body: partial reaction sums in the workspace                                   */
    unsigned grid_blocks;
    unsigned lds_bytes;
    size_t   workspace_bytes;  /* what nb_workspace_bytes_* asks for (of the sliced form when slices > 1)                  */
    unsigned slices;           /* K: 1 = one tournament over the whole system (the fields above describe it); > 1: the     */
                               /* tournament cut into K slices because one tournament's workspace is not affordable         */
} nb_pair_plan_t;
NB_API int nb_pair_plan_f32(unsigned num_bodies, nb_pair_plan_t* plan);
NB_API int nb_pair_plan_f64(unsigned num_bodies, nb_pair_plan_t* plan);
/* (Process-global overrides of both plans, the kernel-time projection of one rank of a multi-GPU step and other lab-bench
 * hooks live in nbody_hip_tuning.h: nothing a host needs to bind.) */

NB_API const char* nb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_HIP_H */
