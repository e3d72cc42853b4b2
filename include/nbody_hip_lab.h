/*
 * nbody_hip_lab.h -- the LAB BENCH: exported by libnbody_hip_lab.so only (the product's object files + csrc/nbody_comm_lab.hip), never
 * by libnbody_hip.so.  What a one-GPU box can prove and measure against the REAL RCCL (a self-loop with every byte checked, a loopback
 * rank that steps as rank r of a nominal G-rank communicator, an in-process world of G ranks on one device), a stream-placement
 * A/B hook and the allocation-failure hook.  Used by tests/, tools/ and the one-GPU
 * projections of bench.py (each in a process that loads the lab library INSTEAD of libnbody_hip.so: it exports everything
 * nbody_hip.h and nbody_hip_tuning.h declare as well).  Nothing a host binds; process-global, not thread-safe.
 */
#ifndef NBODY_HIP_LAB_H
#define NBODY_HIP_LAB_H

#include "nbody_hip_tuning.h"

#ifdef __cplusplus
extern "C" {
#endif

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
 */
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

/* Retire the second compute stream and make another (experiments on how much its placement matters: tools/side_stream_placement.py). */
NB_API int nb_comm_replace_side_stream(nb_comm_t comm);


/* A clock read from the chip itself: `workgroups` single-wave workgroups (consecutive ones land on different XCDs) each stamp the
 * shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around a spin of `microseconds` and write four 64-bit
 * words -- {shader cycles, 100 MHz ticks, XCC_ID, HW_ID} -- to device_words[4 * workgroup] (caller-owned device memory).  MHz = 100 *
 * cycles / ticks.  An experiment (tools/inkernel_clock.py): launched right behind a dense kernel it does NOT read that kernel's clock. */
NB_API int nb_clock_probe_launch(void* device_words, int workgroups, unsigned microseconds, nb_stream_t stream);

/* Tests of the out-of-memory fall-backs (halve the workspace and ask again; step without one): every nb_alloc request above
 * `bytes` is refused BY THE RUNTIME (the request is replaced by one no device can serve), 0 = no limit.  The CLI's
 * --alloc-limit-mib sets it when the lab library is the one in the process (LD_PRELOAD), and says so when it is not. */
NB_API int nb_set_alloc_limit(size_t bytes);


#ifdef __cplusplus
}
#endif
#endif /* NBODY_HIP_LAB_H */
