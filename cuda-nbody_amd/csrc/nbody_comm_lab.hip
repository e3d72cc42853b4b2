// nbody_comm_lab.hip -- the LAB BENCH of the multi-GPU layer (include/nbody_hip_lab.h).  gfx950 only.
//
// Linked into libnbody_hip_lab.so (= the product's own object files + this one) and never into libnbody_hip.so: what a one-GPU box
// can still prove and measure against the REAL RCCL, and the A/B switches of tuning sweeps --
//   * a self-loop with every byte checked (nb_comm_selftest_*), the measuring form of it (nb_comm_self_transfer_f32);
//   * a LOOPBACK rank: rank r of a nominal G-rank communicator whose ncclComm has one rank (nb_comm_loopback_open);
//   * an IN-PROCESS world: all G ranks in one process on one device sharing one real ncclComm (nb_comm_inprocess_open_all);
//   * a new second compute stream on demand (nb_comm_replace_side_stream: how much does its placement matter?);
//   * the allocation-failure hook of the out-of-memory tests (nb_set_alloc_limit), a clock probe kernel (nb_clock_probe_launch).
// It reaches into the product through nbody_comm_internal.h (hidden symbols of the same shared object) and adds no code path to
// it: a loopback rank is a Comm with self_peers set, an in-process world is G such Comms that share their ncclComm and exchange
// stream -- the product's exchange issues its calls in one canonical order whoever the peers are.
#include "../../include/nbody_hip_lab.h"

#include "nbody_comm_internal.h"
#include "rand_stream_guard.h"

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

using namespace nbc;

// nb_clock_probe_launch (lab header): a clock read from the chip itself -- a wave stamps the shader-cycle
// counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around a short spin; cycles / time is the clock, whatever
// the power management's tables (sysfs pp_dpm_sclk / hwmon freq1_input) say.  Built to read the clock a dense kernel ran at from right
// behind it -- and measured NOT to: launched microseconds after pair_forces it reads 2.43 GHz where the kernel's own stamps say 2.20
// (profiles/round6_delivered_clock.txt): the chip raises its clock as soon as the load is gone.  Kept as the experiment it was; the
// product reads the clock INSIDE the kernel (pair_forces_clocked).  Per workgroup four words: {shader cycles, 100 MHz ticks, XCC_ID, HW_ID}.
__global__ void clock_probe(unsigned long long* words, unsigned long long ticks) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long       r1 = r0;
    float                    x  = static_cast<float>(threadIdx.x);
    while (r1 - r0 < ticks) {
#pragma unroll
        for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1                          = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        unsigned long long* out = words + static_cast<size_t>(blockIdx.x) * 4;
        out[0] = c1 - c0, out[1] = r1 - r0;
        out[2] = static_cast<unsigned>(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20));   // XCC_ID
        out[3] = static_cast<unsigned>(__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4));   // HW_ID[15:0]
    }
    if (x == 12345.678f) words[0] = 0;  // (keeps the chain alive)
}


extern "C" {

// ---- the REAL transport on one GPU (lab header): a communicator of one rank that does own an RCCL communicator, and a
// self-loop through it with the event choreography of exchange_tiles -----------------------------------------------------------
int nb_comm_selftest_open(nb_comm_t* comm, const void* id) { return init_rank(comm, id, 1, 0, true); }
int nb_comm_loopback_open(nb_comm_t* comm, const void* id, int nominal_world, int nominal_rank) {
    if (nominal_world < 2) return NB_ERR_INVALID_ARGUMENT;
    return init_rank(comm, id, nominal_world, nominal_rank, true, true);
}

// An IN-PROCESS world: all G ranks in this process, on the current device, sharing ONE real one-rank ncclComm and one exchange stream.
// Every transfer is a self-transfer of that communicator; RCCL matches the sends and receives of one peer first in, first out, so
// rank a's send reaches rank b's receive by the ORDER in which the product issues them (the canonical order of exchange_tiles / reaction_exchange: sends by rank, receives by source rank -- nothing there knows about this world).  The
// full G-rank step -- even G, split rectangles and all -- then runs through the product's own calls into the REAL library, its
// results comparable with the CPU path: what RCCL's refusal of two ranks per device otherwise leaves to the transport double.
int nb_comm_inprocess_open_all(nb_comm_t* comms, int world, const void* id) {
    NB_KEEP_RAND_STREAM;
    if (comms == nullptr || id == nullptr || world < 2) return NB_ERR_INVALID_ARGUMENT;
    Rccl* lib = rccl();
    if (lib == nullptr) return NB_ERR_UNSUPPORTED;
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, sizeof(uid.internal));
    ncclComm_t real = nullptr;
    if (const int rc = lib->CommInitRank(&real, 1, uid, 0); rc != 0) return nccl_status(rc);
    std::shared_ptr<void> owner(real, [lib](void* p) { (void)lib->CommDestroy(static_cast<ncclComm_t>(p)); });
    int device = 0;
    if (const auto err = hipGetDevice(&device); err != hipSuccess) return static_cast<int>(err);
    std::vector<Comm*> made;
    int                rc = 0;
    for (int k = 0; k < world && rc == 0; ++k) {
        auto* c      = new Comm;
        c->nccl      = real, c->shared_nccl = owner;
        c->rank      = k, c->world = world, c->device = device;
        c->self_peers = true;
        made.push_back(c);
        rc = make_resources(c);
        if (rc == 0 && k > 0) {  // one exchange stream for all: the ranks' self-transfers of a round are ONE RCCL group on one stream
            (void)hipStreamDestroy(c->stream);
            c->stream = made.front()->stream, c->owns_stream = false;
        }
    }
    if (rc != 0) {
        for (Comm* c : made) {
            free_resources(c);
            delete c;
        }
        return rc;
    }
    std::shared_ptr<void> stream_owner(made.front()->stream, [device](void* s) {
        DeviceScope scope(device);
        (void)hipStreamDestroy(static_cast<hipStream_t>(s));
    });
    for (Comm* c : made) c->group = made, c->shared_stream = stream_owner, c->owns_stream = false;
    for (int k = 0; k < world; ++k) comms[k] = made[static_cast<size_t>(k)];
    return 0;
}

// `rounds` send/recv pairs from this rank to ITSELF on the communicator's exchange stream -- round k moves `count` floats from
// src + k * count to dst + k * count -- all in one RCCL group or a group per round, after what `after` holds now; `begin` / `end`
// (optional) are recorded on the exchange stream around them, and tile event 0 after them (nb_exchange_wait_tile(comm, .., 0)
// is not usable for it: a rank never waits for its own tile -- the caller waits for `end`).
int nb_comm_self_transfer_f32(nb_comm_t comm, const float* src, float* dst, size_t count, int rounds, int one_group, nb_stream_t after, nb_event_t begin, nb_event_t end) {
    NB_KEEP_RAND_STREAM;
    Comm* c = as_comm(comm);
    if (c == nullptr || c->nccl == nullptr || src == nullptr || dst == nullptr || count == 0 || rounds < 1) return NB_ERR_INVALID_ARGUMENT;
    Rccl* lib = rccl();
    if (lib == nullptr) return NB_ERR_UNSUPPORTED;
    DeviceScope scope(c->device);
    auto        err = hipEventRecord(c->ready, reinterpret_cast<hipStream_t>(after));
    if (err == hipSuccess) err = hipStreamWaitEvent(c->stream, c->ready, 0);
    if (err == hipSuccess && begin != nullptr) err = hipEventRecord(reinterpret_cast<hipEvent_t>(begin), c->stream);
    if (err != hipSuccess) return static_cast<int>(err);
    int rc = one_group ? static_cast<int>(lib->GroupStart()) : 0;
    for (int k = 0; k < rounds && rc == 0; ++k) {
        if (!one_group) rc = lib->GroupStart();
        if (rc == 0) rc = lib->Send(src + static_cast<size_t>(k) * count, count, ncclFloat32, peer_of(c, c->rank), c->nccl, c->stream);
        if (rc == 0) rc = lib->Recv(dst + static_cast<size_t>(k) * count, count, ncclFloat32, peer_of(c, c->rank), c->nccl, c->stream);
        if (!one_group) {
            const int ended = lib->GroupEnd();
            if (rc == 0) rc = ended;
        }
    }
    if (one_group) {
        const int ended = lib->GroupEnd();
        if (rc == 0) rc = ended;
    }
    if (rc != 0) return nccl_status(rc);
    if (end != nullptr) err = hipEventRecord(reinterpret_cast<hipEvent_t>(end), c->stream);
    if (err == hipSuccess) err = hipEventRecord(c->arrived[static_cast<size_t>(c->rank)], c->stream);
    return static_cast<int>(err);
}

int nb_comm_selftest_f32(nb_comm_t comm, size_t bytes, nb_stream_t stream, nb_comm_selftest_t* report) {
    NB_KEEP_RAND_STREAM;
    Comm* c = as_comm(comm);
    if (c == nullptr || report == nullptr || bytes < 4 || bytes % 4 != 0) return NB_ERR_INVALID_ARGUMENT;
    std::memset(report, 0, sizeof(*report));
    if (c->nccl == nullptr || c->world != 1) return NB_ERR_INVALID_ARGUMENT;  // (made by nb_comm_selftest_open)
    Rccl* lib = rccl();
    if (lib == nullptr) return NB_ERR_UNSUPPORTED;
    (void)nb_comm_transport_info(comm, &report->rccl_version, report->library_path, sizeof(report->library_path));
    DeviceScope  scope(c->device);
    hipStream_t  on    = reinterpret_cast<hipStream_t>(stream);
    const size_t count = bytes / 4;
    std::vector<unsigned> sent(count), got(count);
    for (size_t k = 0; k < count; ++k) sent[k] = static_cast<unsigned>(k) * 2654435761u + 0x9e3779b9u;
    float *     src = nullptr, *dst = nullptr, *gathered = nullptr;
    hipEvent_t  t0 = nullptr, t1 = nullptr;
    int         result = 0;
    auto refuse = [&](const char* call, int rc, int* slot) {
        *slot = nccl_status(rc);
        if (report->refused_call[0] == 0) std::strncpy(report->refused_call, call, sizeof(report->refused_call) - 1);
        if (result == 0) result = *slot;
    };
    auto wrong_bytes = [&](const float* device) -> size_t {  // what `device` holds against what was sent (after the exchange stream's work)
        if (hipMemcpyAsync(got.data(), device, bytes, hipMemcpyDeviceToHost, on) != hipSuccess || hipStreamSynchronize(on) != hipSuccess) return bytes;
        size_t wrong = 0;
        for (size_t k = 0; k < count; ++k)
            for (int b = 0; b < 4; ++b) wrong += ((sent[k] >> (8 * b)) & 0xffu) != ((got[k] >> (8 * b)) & 0xffu);
        return wrong;
    };
    hipError_t err = hipMalloc(reinterpret_cast<void**>(&src), bytes);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&dst), bytes);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&gathered), bytes);
    if (err == hipSuccess) err = hipEventCreate(&t0);
    if (err == hipSuccess) err = hipEventCreate(&t1);
    // what is sent is produced on the CALLER's stream (as a step's positions are); the exchange stream waits for `ready`
    if (err == hipSuccess) err = hipMemcpyAsync(src, sent.data(), bytes, hipMemcpyHostToDevice, on);
    if (err == hipSuccess) err = hipMemsetAsync(dst, 0xa5, bytes, on);
    if (err == hipSuccess) err = hipMemsetAsync(gathered, 0x5a, bytes, on);
    if (err == hipSuccess) {
        // (1) GroupStart; Send(to self); Recv(from self); GroupEnd -- exactly one round of exchange_tiles
        const int rc = nb_comm_self_transfer_f32(comm, src, dst, count, 1, 1, stream, t0, t1);
        if (rc >= NB_ERR_RCCL_BASE) refuse("ncclSend/ncclRecv (grouped, to self)", rc - NB_ERR_RCCL_BASE, &report->send_recv_status);
        else if (rc != 0) err = static_cast<hipError_t>(rc);
        else {
            err = hipStreamWaitEvent(on, c->arrived[0], 0);  // the consumer of a tile waits for its event, on the compute stream
            if (err == hipSuccess) report->send_recv_wrong_bytes = wrong_bytes(dst);
            if (err == hipSuccess) err = hipEventElapsedTime(&report->send_recv_ms, t0, t1);
        }
    }
    if (err == hipSuccess) {
        // (2) ncclAllGather as nb_allgather_* issues it; out of place first (bytes to check), then in place (the product's form)
        err = hipEventRecord(c->ready, on);
        if (err == hipSuccess) err = hipStreamWaitEvent(c->stream, c->ready, 0);
        if (err == hipSuccess) err = hipEventRecord(t0, c->stream);
        if (err == hipSuccess) {
            int rc = lib->AllGather(src, gathered, count, ncclFloat32, c->nccl, c->stream);
            if (rc == 0) rc = lib->AllGather(gathered, gathered, count, ncclFloat32, c->nccl, c->stream);
            if (rc != 0) refuse("ncclAllGather", rc, &report->all_gather_status);
            else {
                err = hipEventRecord(t1, c->stream);
                if (err == hipSuccess) err = hipEventRecord(c->arrived[0], c->stream);
                if (err == hipSuccess) err = hipStreamWaitEvent(on, c->arrived[0], 0);
                if (err == hipSuccess) report->all_gather_wrong_bytes = wrong_bytes(gathered);
                if (err == hipSuccess) err = hipEventElapsedTime(&report->all_gather_ms, t0, t1);
            }
        }
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(on);
    if (t0) (void)hipEventDestroy(t0);
    if (t1) (void)hipEventDestroy(t1);
    for (float* p : {src, dst, gathered})
        if (p) (void)hipFree(p);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return static_cast<int>(err);
    }
    if (result == 0 && (report->send_recv_wrong_bytes != 0 || report->all_gather_wrong_bytes != 0)) result = NB_ERR_UNSUPPORTED;
    return result;
}

int nb_comm_replace_side_stream(nb_comm_t comm) {  // (experiments: how much does the placement of the second stream matter?)
    NB_KEEP_RAND_STREAM;
    Comm* c = as_comm(comm);
    if (c == nullptr) return NB_ERR_INVALID_ARGUMENT;
    DeviceScope scope(c->device);
    if (c->aux != nullptr) {
        (void)hipStreamSynchronize(c->aux);
        c->aux_retired.push_back(c->aux);
        c->aux = nullptr;
    }
    c->aux_probed = false;
    for (StreamNote& n : c->seen) n.aux_beside = false;
    return static_cast<int>(create_side_stream(&c->aux));
}


int nb_clock_probe_launch(void* device_words, int workgroups, unsigned microseconds, nb_stream_t stream) {
    if (device_words == nullptr || workgroups < 1 || workgroups > 1024 || microseconds == 0 || microseconds > 100000u) return NB_ERR_INVALID_ARGUMENT;
    if ((reinterpret_cast<std::uintptr_t>(device_words) % sizeof(unsigned long long)) != 0) return NB_ERR_INVALID_ARGUMENT;
    (void)hipGetLastError();
    hipLaunchKernelGGL(clock_probe, dim3(static_cast<unsigned>(workgroups)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), static_cast<unsigned long long*>(device_words),
                       static_cast<unsigned long long>(microseconds) * 100ull);
    return static_cast<int>(hipGetLastError());
}

int nb_set_alloc_limit(size_t bytes) {
    nb::alloc_limit().store(bytes);
    return 0;
}

}  // extern "C"
