// fake_rccl.cpp -- TEST DOUBLE for RCCL (a third-party library), never part of the product.
//
// libnbody_hip.so resolves ten nccl* entry points with dlopen/dlsym and honours NBODY_RCCL_LIB
// (cuda-nbody_amd/csrc/nbody_comm.hip).  This file implements those ten with nothing but the HIP runtime, so that the
// multi-GPU code of the product -- grouped send/recv rounds, per-round events, the tile waits of the sharded step --
// runs on a box with ONE GPU: its "ranks" may share a device (real RCCL refuses that), and a send/recv pair becomes a
// device-to-device copy with RCCL's stream semantics:
//
//     sender's stream:   record A ............................ wait B      (the send returns the buffer when the data has left)
//     receiver's stream:          wait A, copy src -> dst, record B        (the receive completes when the data has landed)
//
// Matching is what RCCL does: a send of rank a to rank b pairs with the oldest unmatched receive of rank b from rank a.
// ncclGroupEnd (or an ungrouped call) posts the caller's operations and then blocks the HOST until each of them has met
// its counterpart -- one thread driving every rank in one group (ncclCommInitAll model) never waits; a thread per rank
// (ncclCommInitRank model) waits for the peer threads exactly as with the real library.  A counterpart that does not
// show up within FAKE_RCCL_TIMEOUT_S (default 60 s) is reported as ncclInternalError rather than hanging the test.
//
// ACROSS PROCESSES (round 4; FAKE_RCCL_IPC=1 in the environment of every rank): one process per rank, as bench.py and
// `torch.distributed.run` start them, all on the one GPU.  The ranks then meet through files under /dev/shm named after the
// unique id: a send is "synchronise the stream, copy the buffer to the host, publish it as <id>_<src>_<dst>_<sequence>"; a
// receive waits for that file (FAKE_RCCL_TIMEOUT_S), copies it to the device and removes it.  Everything happens inside
// ncclGroupEnd / the ungrouped call, sends before receives, so no order of arrival can deadlock.  Host-synchronous on purpose:
// it rehearses the CODE PATH of an N-process job (ids, groups, rounds, who waits for what), never its timing.
//
// Build: tests/fake_rccl/Makefile -> tests/fake_rccl/libfake_rccl.so.
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define FAKE_API extern "C" __attribute__((visibility("default")))

struct ncclComm;

namespace {

enum : int { kSuccess = 0, kUnhandledCudaError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

struct Op {
    bool        is_send = false;
    int         rank    = 0;  // who posted it
    int         peer    = 0;
    int         device  = 0;
    void*       buffer  = nullptr;
    size_t      bytes   = 0;
    hipStream_t stream  = nullptr;
    bool        done    = false;
    int         status  = kSuccess;
    struct ::ncclComm* ipc_comm = nullptr;  // FAKE_RCCL_IPC: the communicator of a cross-process operation (no World)
};

struct Gather {
    int         rank   = 0;
    int         device = 0;
    const void* send   = nullptr;
    void*       recv   = nullptr;
    size_t      bytes  = 0;
    hipStream_t stream = nullptr;
};

struct World {
    int                              size = 1;
    std::mutex                       mutex;
    std::condition_variable          changed;
    std::deque<std::shared_ptr<Op>>  unmatched;
    std::vector<Gather>              gather;             // posts of the collective in progress
    uint64_t                         gathers_done = 0;   // generation counter: a waiter leaves when it moves on
    int                              gather_status = kSuccess;
    int                              members = 0;        // live communicators
    std::vector<hipEvent_t>          events;             // ring of reusable events (a wait captures the record it follows)
    size_t                           next_event = 0;
};

}  // namespace

struct ncclComm {
    World* world  = nullptr;
    int    rank   = 0;
    int    device = 0;
    // across processes (FAKE_RCCL_IPC=1): no shared World -- the key of the unique id names the files the ranks meet through
    bool                  ipc  = false;
    int                   size = 1;
    uint64_t              key  = 0;
    std::vector<uint64_t> sent, received;  // per peer: messages so far (the sequence number a send/receive pair agrees on)
};
using ncclComm_t = ncclComm*;
struct ncclUniqueId {
    char internal[128];
};

namespace {

std::mutex                g_registry_mutex;
std::map<uint64_t, World*> g_registry;  // unique id -> world (ncclCommInitRank)
std::atomic<uint64_t>     g_next_id{1};
std::atomic<long>         g_sends{0}, g_recvs{0}, g_gathers{0}, g_groups{0}, g_copies{0};

thread_local int                              t_group_depth = 0;
thread_local std::vector<std::pair<World*, std::shared_ptr<Op>>> t_group_ops;

double timeout_seconds() {
    const char* s = std::getenv("FAKE_RCCL_TIMEOUT_S");
    const double v = s ? std::atof(s) : 0.0;
    return v > 0.0 ? v : 60.0;
}

size_t type_bytes(int type) {
    switch (type) {
        case 0: case 1: return 1;           // int8 / char, uint8
        case 2: case 3: case 7: return 4;   // int32, uint32, float32
        case 4: case 5: case 8: return 8;   // int64, uint64, float64
        case 6: case 9: return 2;           // float16, bfloat16
        default: return 0;
    }
}

class OnDevice {
 public:
    explicit OnDevice(int device) {
        (void)hipGetDevice(&saved_);
        if (saved_ != device) (void)hipSetDevice(device);
    }
    ~OnDevice() {
        int now = 0;
        (void)hipGetDevice(&now);
        if (now != saved_) (void)hipSetDevice(saved_);
    }

 private:
    int saved_ = 0;
};

// caller holds world.mutex
hipEvent_t take_event(World& world, int device) {
    constexpr size_t ring = 512;
    if (world.events.size() < ring) {
        OnDevice   scope(device);
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        world.events.push_back(e);
        return e;
    }
    hipEvent_t e     = world.events[world.next_event];
    world.next_event = (world.next_event + 1) % ring;
    return e;
}

// One matched pair as a copy with RCCL's stream semantics (see the header of this file).  Caller holds world.mutex.
int transfer(World& world, const Op& send, const Op& recv) {
    if (send.bytes != recv.bytes) return kInvalidArgument;
    hipEvent_t left = take_event(world, send.device), landed = take_event(world, recv.device);
    if (left == nullptr || landed == nullptr) return kUnhandledCudaError;
    hipError_t err;
    {
        OnDevice scope(send.device);
        err = hipEventRecord(left, send.stream);
    }
    if (err == hipSuccess) {
        OnDevice scope(recv.device);
        err = hipStreamWaitEvent(recv.stream, left, 0);
        if (err == hipSuccess && send.bytes != 0 && send.buffer != recv.buffer) err = hipMemcpyAsync(recv.buffer, send.buffer, send.bytes, hipMemcpyDeviceToDevice, recv.stream);
        if (err == hipSuccess) err = hipEventRecord(landed, recv.stream);
    }
    if (err == hipSuccess) {
        OnDevice scope(send.device);
        err = hipStreamWaitEvent(send.stream, landed, 0);
    }
    g_copies.fetch_add(1);
    return err == hipSuccess ? kSuccess : kUnhandledCudaError;
}

// caller holds world.mutex
void match_all(World& world) {
    auto& q = world.unmatched;
    for (bool progress = true; progress;) {
        progress = false;
        for (size_t s = 0; s < q.size() && !progress; ++s) {
            if (!q[s]->is_send) continue;
            for (size_t r = 0; r < q.size(); ++r) {
                if (q[r]->is_send || q[r]->rank != q[s]->peer || q[r]->peer != q[s]->rank) continue;  // the OLDEST such receive
                const int status = transfer(world, *q[s], *q[r]);
                q[s]->status = q[r]->status = status;
                q[s]->done = q[r]->done = true;
                q.erase(q.begin() + static_cast<long>(std::max(s, r)));
                q.erase(q.begin() + static_cast<long>(std::min(s, r)));
                progress = true;
                break;
            }
        }
    }
    world.changed.notify_all();
}

// ---- across processes: messages as files under /dev/shm ---------------------------------------------------------------------
bool ipc_mode() {
    const char* v = std::getenv("FAKE_RCCL_IPC");
    return v != nullptr && v[0] == '1';
}

std::string ipc_name(uint64_t key, int src, int dst, uint64_t sequence) {
    return "/dev/shm/fake_rccl_" + std::to_string(key) + "_" + std::to_string(src) + "_" + std::to_string(dst) + "_" + std::to_string(sequence);
}

int ipc_send(const Op& op) {
    ncclComm& c = *op.ipc_comm;
    OnDevice  scope(op.device);
    if (hipStreamSynchronize(op.stream) != hipSuccess) return kUnhandledCudaError;  // what the send reads has been produced
    std::vector<char> host(op.bytes);
    if (op.bytes != 0 && hipMemcpy(host.data(), op.buffer, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return kUnhandledCudaError;
    const std::string name = ipc_name(c.key, c.rank, op.peer, c.sent[static_cast<size_t>(op.peer)]++), tmp = name + ".tmp";
    const int         fd   = ::open(tmp.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0600);
    if (fd < 0) return kSystemError;
    size_t done = 0;
    while (done < host.size()) {
        const ssize_t w = ::write(fd, host.data() + done, host.size() - done);
        if (w <= 0) {
            ::close(fd);
            return kSystemError;
        }
        done += static_cast<size_t>(w);
    }
    ::close(fd);
    return ::rename(tmp.c_str(), name.c_str()) == 0 ? kSuccess : kSystemError;  // published atomically: a reader never sees half a message
}

int ipc_recv(const Op& op) {
    ncclComm&         c    = *op.ipc_comm;
    const std::string name = ipc_name(c.key, op.peer, c.rank, c.received[static_cast<size_t>(op.peer)]++);
    const auto        deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_seconds());
    int               fd   = -1;
    while ((fd = ::open(name.c_str(), O_RDONLY)) < 0) {
        if (std::chrono::steady_clock::now() > deadline) return kInternalError;  // the counterpart never came
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    struct stat st {};
    if (::fstat(fd, &st) != 0 || static_cast<size_t>(st.st_size) != op.bytes) {
        ::close(fd);
        ::unlink(name.c_str());
        return kInvalidArgument;  // the two sides disagree about the size of the message
    }
    std::vector<char> host(op.bytes);
    size_t            done = 0;
    while (done < host.size()) {
        const ssize_t r = ::read(fd, host.data() + done, host.size() - done);
        if (r <= 0) break;
        done += static_cast<size_t>(r);
    }
    ::close(fd);
    ::unlink(name.c_str());
    if (done != host.size()) return kSystemError;
    OnDevice scope(op.device);
    if (hipStreamSynchronize(op.stream) != hipSuccess) return kUnhandledCudaError;  // earlier readers of the buffer on this stream are done
    if (op.bytes != 0 && hipMemcpy(op.buffer, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return kUnhandledCudaError;
    g_copies.fetch_add(1);
    return kSuccess;
}

int ipc_post_and_wait(std::vector<std::pair<World*, std::shared_ptr<Op>>>& ops) {
    int status = kSuccess;
    for (const bool sends : {true, false}) {  // every send of the group first: whatever order the ranks arrive in, nobody waits for a message not yet written
        for (auto& [world, op] : ops) {
            if (op->is_send != sends) continue;
            const int rc = sends ? ipc_send(*op) : ipc_recv(*op);
            if (rc != kSuccess && status == kSuccess) status = rc;
        }
    }
    ops.clear();
    return status;
}

int post_and_wait(std::vector<std::pair<World*, std::shared_ptr<Op>>>& ops) {
    if (!ops.empty() && ops.front().second->ipc_comm != nullptr) return ipc_post_and_wait(ops);
    // post everything first (a single thread driving all ranks has every counterpart in this very list), then wait
    std::vector<World*> worlds;
    for (auto& [world, op] : ops) {
        std::lock_guard<std::mutex> lock(world->mutex);
        world->unmatched.push_back(op);
        if (worlds.empty() || worlds.back() != world) worlds.push_back(world);
    }
    for (World* world : worlds) {
        std::lock_guard<std::mutex> lock(world->mutex);
        match_all(*world);
    }
    int        status   = kSuccess;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_seconds());
    for (auto& [world, op] : ops) {
        std::unique_lock<std::mutex> lock(world->mutex);
        if (!world->changed.wait_until(lock, deadline, [&op = op] { return op->done; })) {
            for (auto it = world->unmatched.begin(); it != world->unmatched.end(); ++it)
                if (*it == op) {
                    world->unmatched.erase(it);
                    break;
                }
            status = kInternalError;  // the counterpart never came
        } else if (op->status != kSuccess && status == kSuccess) {
            status = op->status;
        }
    }
    ops.clear();
    return status;
}

int enqueue(bool is_send, void* buffer, size_t count, int type, int peer, ncclComm_t comm, hipStream_t stream) {
    if (comm == nullptr || (comm->world == nullptr && !comm->ipc) || peer < 0 || peer >= (comm->ipc ? comm->size : comm->world->size) || type_bytes(type) == 0) return kInvalidArgument;
    if (buffer == nullptr && count != 0) return kInvalidArgument;
    auto op     = std::make_shared<Op>();
    op->is_send = is_send, op->rank = comm->rank, op->peer = peer, op->device = comm->device;
    op->buffer = buffer, op->bytes = count * type_bytes(type), op->stream = stream;
    op->ipc_comm = comm->ipc ? comm : nullptr;
    (is_send ? g_sends : g_recvs).fetch_add(1);
    t_group_ops.emplace_back(comm->world, std::move(op));
    if (t_group_depth > 0) return kSuccess;
    return post_and_wait(t_group_ops);
}

World* world_of_id(const ncclUniqueId& id, int size) {
    uint64_t key = 0;
    std::memcpy(&key, id.internal + 8, sizeof(key));
    if (std::memcmp(id.internal, "FAKERCCL", 8) != 0 || key == 0) return nullptr;
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    auto                        it = g_registry.find(key);
    if (it != g_registry.end()) return it->second->size == size ? it->second : nullptr;
    auto* world = new World;
    world->size = size;
    g_registry.emplace(key, world);
    return world;
}

void leave(World* world) {
    bool last = false;
    {
        std::lock_guard<std::mutex> lock(world->mutex);
        last = --world->members == 0;
    }
    if (!last) return;
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        for (auto it = g_registry.begin(); it != g_registry.end(); ++it)
            if (it->second == world) {
                g_registry.erase(it);
                break;
            }
    }
    for (hipEvent_t e : world->events) (void)hipEventDestroy(e);
    delete world;
}

}  // namespace

FAKE_API int ncclGetVersion(int* version) {  // 0: no RCCL release -- what a record shows when the double was bound
    if (version == nullptr) return kInvalidArgument;
    *version = 0;
    return kSuccess;
}

FAKE_API int ncclGetUniqueId(ncclUniqueId* id) {
    if (id == nullptr) return kInvalidArgument;
    std::memset(id->internal, 0, sizeof(id->internal));
    std::memcpy(id->internal, "FAKERCCL", 8);
    uint64_t key = g_next_id.fetch_add(1);
    if (ipc_mode()) {  // across processes the key must be unique on the machine, not in the process
        key = (static_cast<uint64_t>(::getpid()) << 32) ^ static_cast<uint64_t>(std::chrono::steady_clock::now().time_since_epoch().count()) ^ (key << 56);
        if (key == 0) key = 1;
        id->internal[16] = 'I';
    }
    std::memcpy(id->internal + 8, &key, sizeof(key));
    return kSuccess;
}

FAKE_API int ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (comm == nullptr || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
    // FAKE_RCCL_HANG_INIT=<rank> | all: this call never returns on that rank (a bootstrap that waits for a peer for ever) -- what
    // bench.py's bounded bring-up is tested against
    if (const char* hang = std::getenv("FAKE_RCCL_HANG_INIT"); hang != nullptr && hang[0] != 0 && (std::strcmp(hang, "all") == 0 || std::atoi(hang) == rank)) {
        for (;;) sleep(3600);
    }
    if (id.internal[16] == 'I' && std::memcmp(id.internal, "FAKERCCL", 8) == 0) {  // one process per rank: no shared World
        auto* c = new ncclComm;
        c->ipc = true, c->size = nranks, c->rank = rank;
        std::memcpy(&c->key, id.internal + 8, sizeof(c->key));
        c->sent.assign(static_cast<size_t>(nranks), 0), c->received.assign(static_cast<size_t>(nranks), 0);
        if (hipGetDevice(&c->device) != hipSuccess) {
            delete c;
            return kUnhandledCudaError;
        }
        *comm = c;
        return kSuccess;
    }
    World* world = world_of_id(id, nranks);
    if (world == nullptr) return kInvalidArgument;
    auto* c  = new ncclComm;
    c->world = world, c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        return kUnhandledCudaError;
    }
    {
        std::lock_guard<std::mutex> lock(world->mutex);
        ++world->members;
    }
    *comm = c;
    return kSuccess;
}

FAKE_API int ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (comms == nullptr || ndev < 1) return kInvalidArgument;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess) return kUnhandledCudaError;
    for (int k = 0; k < ndev; ++k)
        if ((devlist ? devlist[k] : k) < 0 || (devlist ? devlist[k] : k) >= visible) return kInvalidArgument;  // duplicates allowed: that is the point
    auto* world    = new World;
    world->size    = ndev;
    world->members = ndev;
    for (int k = 0; k < ndev; ++k) {
        comms[k]         = new ncclComm;
        comms[k]->world  = world;
        comms[k]->rank   = k;
        comms[k]->device = devlist ? devlist[k] : k;
    }
    return kSuccess;
}

FAKE_API int ncclCommDestroy(ncclComm_t comm) {
    if (comm == nullptr) return kInvalidArgument;
    if (!comm->ipc) leave(comm->world);
    delete comm;
    return kSuccess;
}

FAKE_API int ncclGroupStart() {
    ++t_group_depth;
    return kSuccess;
}

FAKE_API int ncclGroupEnd() {
    if (t_group_depth <= 0) return kInvalidUsage;
    if (--t_group_depth > 0) return kSuccess;
    g_groups.fetch_add(1);
    // FAKE_RCCL_HANG_GROUP_END=<rank>: the process whose RANK (torch.distributed.run's) that is never comes back from its first group
    // -- an exchange that waits for a peer for ever: what bench.py's headline watchdog is tested against
    if (const char* hang = std::getenv("FAKE_RCCL_HANG_GROUP_END"), *me = std::getenv("RANK"); hang != nullptr && me != nullptr && hang[0] != 0 && std::atoi(hang) == std::atoi(me)) {
        for (;;) sleep(3600);
    }
    // FAKE_RCCL_HANG_WHEN_EXISTS=<path>: ... from the moment that file exists (a test makes it when the run has reached the stage to hang in)
    if (const char* path = std::getenv("FAKE_RCCL_HANG_WHEN_EXISTS"); path != nullptr && path[0] != 0 && access(path, F_OK) == 0) {
        for (;;) sleep(3600);
    }
    return post_and_wait(t_group_ops);
}

FAKE_API int ncclSend(const void* sendbuff, size_t count, int datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(true, const_cast<void*>(sendbuff), count, datatype, peer, comm, stream);
}

FAKE_API int ncclRecv(void* recvbuff, size_t count, int datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(false, recvbuff, count, datatype, peer, comm, stream);
}

// recvbuff of every rank = the ranks' sendbuffs in rank order (in place when sendbuff == recvbuff + rank * bytes).
// The last rank to arrive issues every copy; each stream then waits until all copies that read or write its buffers are done.
FAKE_API int ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, ncclComm_t comm, hipStream_t stream) {
    if (comm != nullptr && comm->ipc) {  // across processes: the collective as its send/recv pairs
        if (type_bytes(datatype) == 0 || sendbuff == nullptr || recvbuff == nullptr) return kInvalidArgument;
        g_gathers.fetch_add(1);
        const size_t bytes = sendcount * type_bytes(datatype);
        char*        own   = static_cast<char*>(recvbuff) + static_cast<size_t>(comm->rank) * bytes;
        if (own != sendbuff && bytes != 0 && hipMemcpyAsync(own, sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return kUnhandledCudaError;
        ++t_group_depth;
        int rc = kSuccess;
        for (int p = 0; p < comm->size && rc == kSuccess; ++p) {
            if (p == comm->rank) continue;
            rc = enqueue(true, const_cast<void*>(sendbuff), sendcount, datatype, p, comm, stream);
            if (rc == kSuccess) rc = enqueue(false, static_cast<char*>(recvbuff) + static_cast<size_t>(p) * bytes, sendcount, datatype, p, comm, stream);
        }
        --t_group_depth;
        if (t_group_depth > 0) return rc;
        const int end = post_and_wait(t_group_ops);
        return rc != kSuccess ? rc : end;
    }
    if (comm == nullptr || comm->world == nullptr || type_bytes(datatype) == 0 || sendbuff == nullptr || recvbuff == nullptr) return kInvalidArgument;
    World&                       world = *comm->world;
    std::unique_lock<std::mutex> lock(world.mutex);
    const uint64_t               generation = world.gathers_done;
    world.gather.push_back(Gather{comm->rank, comm->device, sendbuff, recvbuff, sendcount * type_bytes(datatype), stream});
    g_gathers.fetch_add(1);
    if (static_cast<int>(world.gather.size()) < world.size) {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_seconds());
        if (!world.changed.wait_until(lock, deadline, [&] { return world.gathers_done != generation; })) {
            for (auto it = world.gather.begin(); it != world.gather.end(); ++it)
                if (it->rank == comm->rank) {
                    world.gather.erase(it);
                    break;
                }
            return kInternalError;
        }
        return world.gather_status;
    }
    int        status = kSuccess;
    hipError_t err    = hipSuccess;
    std::vector<hipEvent_t> ready(world.gather.size()), filled(world.gather.size());
    for (size_t p = 0; p < world.gather.size() && err == hipSuccess; ++p) {  // "my send buffer may be read from here on"
        OnDevice scope(world.gather[p].device);
        ready[p] = take_event(world, world.gather[p].device);
        err      = ready[p] ? hipEventRecord(ready[p], world.gather[p].stream) : hipErrorOutOfMemory;
    }
    for (size_t r = 0; r < world.gather.size() && err == hipSuccess; ++r) {
        const Gather& dst = world.gather[r];
        OnDevice      scope(dst.device);
        for (size_t p = 0; p < world.gather.size() && err == hipSuccess; ++p) {
            const Gather& src = world.gather[p];
            if (src.bytes != dst.bytes) status = kInvalidArgument;
            char* where = static_cast<char*>(dst.recv) + static_cast<size_t>(src.rank) * dst.bytes;
            err         = hipStreamWaitEvent(dst.stream, ready[p], 0);
            if (err == hipSuccess && where != src.send && src.bytes != 0) {
                err = hipMemcpyAsync(where, src.send, src.bytes, hipMemcpyDeviceToDevice, dst.stream);
                g_copies.fetch_add(1);
            }
        }
        filled[r] = take_event(world, dst.device);
        if (err == hipSuccess) err = filled[r] ? hipEventRecord(filled[r], dst.stream) : hipErrorOutOfMemory;
    }
    for (size_t p = 0; p < world.gather.size() && err == hipSuccess; ++p) {  // a collective returns nobody's buffer before all are done
        OnDevice scope(world.gather[p].device);
        for (size_t r = 0; r < world.gather.size() && err == hipSuccess; ++r)
            if (r != p) err = hipStreamWaitEvent(world.gather[p].stream, filled[r], 0);
    }
    if (err != hipSuccess) status = kUnhandledCudaError;
    world.gather.clear();
    world.gather_status = status;
    ++world.gathers_done;
    world.changed.notify_all();
    return status;
}

FAKE_API const char* ncclGetErrorString(int result) {
    switch (result) {
        case kSuccess: return "no error";
        case kUnhandledCudaError: return "unhandled cuda error (fake RCCL)";
        case kSystemError: return "unhandled system error (fake RCCL)";
        case kInternalError: return "internal error (fake RCCL: a send/recv counterpart never arrived)";
        case kInvalidArgument: return "invalid argument (fake RCCL)";
        case kInvalidUsage: return "invalid usage (fake RCCL)";
        default: return "unknown result code (fake RCCL)";
    }
}

// ---- for the tests: proof that the product's exchange code really came through here ------------------------------------
FAKE_API void fake_rccl_counters(long* sends, long* recvs, long* allgathers, long* groups, long* copies) {
    if (sends) *sends = g_sends.load();
    if (recvs) *recvs = g_recvs.load();
    if (allgathers) *allgathers = g_gathers.load();
    if (groups) *groups = g_groups.load();
    if (copies) *copies = g_copies.load();
}
