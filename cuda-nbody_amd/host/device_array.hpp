// device_array.hpp -- caller-owned storage for the three body arrays.
//   DeviceArray<T>  replaces thrust::device_vector<T>  (/root/reference/src/nbody/bodysystemcuda_default.hpp:34-35)
//   MappedArray<T>  replaces UniqueMappedSpan<T>        (/root/reference/src/nbody/unique_mapped_span.{hpp,cpp})
//   HipEvent        replaces cuda::event_t              (/root/reference/src/nbody/compute_cuda.cpp:66-67,263-272)
#pragma once

#include "hip_check.hpp"

#include <algorithm>
#include <cstddef>
#include <new>
#include <span>
#include <utility>

// what thrust::device_vector throws when the device has no room (thrust::system::detail::bad_alloc IS a std::bad_alloc: the
// reference's main answers it with "Unable to allocate memory!" and exit code 3, nbody.cpp:396-408)
struct DeviceBadAlloc : std::bad_alloc {
    auto what() const noexcept -> const char* override { return "nb_alloc: out of device memory"; }
};

template <typename T> class DeviceArray {
 public:
    DeviceArray() = default;
    explicit DeviceArray(std::size_t n) : size_(n) {
        void* p = nullptr;
        const int status = nb_alloc(&p, n * sizeof(T));
        if (status == NB_ERR_OUT_OF_MEMORY) throw DeviceBadAlloc{};
        hip_check(status, "nb_alloc");
        ptr_ = static_cast<T*>(p);
        // cleared AND the clear complete before the constructor returns: nb_memset is asynchronous on the null stream, and the consumers
        // of this memory run on non-blocking streams that the null stream orders nothing against (a sharded body system's shards;
        // round-5 review: a late clear could wipe sums a kernel had already written -- GB-sized workspaces take milliseconds to clear)
        hip_check(nb_memset(ptr_, 0, n * sizeof(T), nullptr), "nb_memset");
        hip_check(nb_stream_synchronize(nullptr), "nb_stream_synchronize");
    }
    DeviceArray(const DeviceArray&)                    = delete;
    auto operator=(const DeviceArray&) -> DeviceArray& = delete;
    DeviceArray(DeviceArray&& o) noexcept : ptr_(std::exchange(o.ptr_, nullptr)), size_(std::exchange(o.size_, 0)) {}
    auto operator=(DeviceArray&& o) noexcept -> DeviceArray& {
        if (this != &o) {
            release();
            ptr_  = std::exchange(o.ptr_, nullptr);
            size_ = std::exchange(o.size_, 0);
        }
        return *this;
    }
    ~DeviceArray() { release(); }

    auto data() const noexcept -> T* { return ptr_; }
    auto size() const noexcept { return size_; }

    auto upload(std::span<const T> host) -> void { hip_check(nb_h2d(ptr_, host.data(), std::min(host.size(), size_) * sizeof(T), nullptr), "nb_h2d"); }
    auto download(std::span<T> host) const -> void { hip_check(nb_d2h(host.data(), ptr_, std::min(host.size(), size_) * sizeof(T), nullptr), "nb_d2h"); }

 private:
    auto release() noexcept -> void {
        if (ptr_ != nullptr) (void)nb_free(ptr_);
        ptr_ = nullptr;
    }
    T*          ptr_  = nullptr;
    std::size_t size_ = 0;
};

// zero-copy host memory the GPU reads and writes over PCIe (--hostmem)
template <typename T> class MappedArray {
 public:
    MappedArray() = default;
    MappedArray(std::size_t n, const T& value) : size_(n) {
        void *h = nullptr, *d = nullptr;
        hip_check(nb_host_alloc_mapped(&h, &d, n * sizeof(T)), "nb_host_alloc_mapped");
        host_   = static_cast<T*>(h);
        device_ = static_cast<T*>(d);
        std::fill(host_, host_ + n, value);
    }
    MappedArray(const MappedArray&)                    = delete;
    auto operator=(const MappedArray&) -> MappedArray& = delete;
    MappedArray(MappedArray&& o) noexcept : host_(std::exchange(o.host_, nullptr)), device_(std::exchange(o.device_, nullptr)), size_(std::exchange(o.size_, 0)) {}
    ~MappedArray() {
        if (host_ != nullptr) (void)nb_host_free(host_);
    }
    auto host_ptr() const noexcept -> T* { return host_; }
    auto device_ptr() const noexcept -> T* { return device_; }
    auto size() const noexcept { return size_; }

 private:
    T*          host_   = nullptr;
    T*          device_ = nullptr;
    std::size_t size_   = 0;
};

class HipEvent {
 public:
    HipEvent() { hip_check(nb_event_create(&event_), "nb_event_create"); }
    HipEvent(const HipEvent&)                    = delete;
    auto operator=(const HipEvent&) -> HipEvent& = delete;
    ~HipEvent() {
        if (event_ != nullptr) (void)nb_event_destroy(event_);
    }
    auto record(nb_stream_t stream = nullptr) -> void { hip_check(nb_event_record(event_, stream), "nb_event_record"); }  // (on the stream the timed work runs on)
    auto synchronize() -> void { hip_check(nb_event_synchronize(event_), "nb_event_synchronize"); }
    auto handle() const noexcept { return event_; }
    static auto elapsed_ms(const HipEvent& start, const HipEvent& stop) -> float {
        float ms = 0.f;
        hip_check(nb_event_elapsed_ms(&ms, start.event_, stop.event_), "nb_event_elapsed_ms");
        return ms;
    }

 private:
    nb_event_t event_ = nullptr;
};
