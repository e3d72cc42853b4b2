// rccl_api.h -- the RCCL entry points nbody_comm.hip resolves with dlsym, spelled ONCE.
//
// The library never links librccl: it binds the copy that belongs to the HIP runtime of the process at run time (a torch
// process carries its own), so the function-pointer types below are what the calls go through.  They are written in terms of
// four aliases -- Result, DataType, Comm, UniqueId -- with two definitions:
//   * the product build (nbody_comm.hip): plain restatements (int, int, an opaque struct pointer, 128 chars) -- no RCCL header
//     is needed to build or to load the library;
//   * the ABI check (rccl_abi_check.cpp, `make check-rccl-abi`, run by tests/test_capi_symbols.py): NB_RCCL_API_REAL_HEADER is
//     defined, the aliases ARE the types of /opt/rocm/include/rccl/rccl.h, and the check static_asserts that every pointer type
//     below is exactly decltype(&nccl...) and that each restated alias has the size, kind and values of the real one.
// A change of an RCCL signature therefore fails the build of the check instead of corrupting a call at run time.
#ifndef NBODY_RCCL_API_H
#define NBODY_RCCL_API_H

#include <hip/hip_runtime_api.h>

#include <cstddef>

namespace nb_rccl {

#ifdef NB_RCCL_API_REAL_HEADER
using Result   = ncclResult_t;
using DataType = ncclDataType_t;
using Comm     = ncclComm_t;
using UniqueId = ncclUniqueId;
#else
using Result   = int;  // ncclResult_t: an enum, ncclSuccess = 0
using DataType = int;  // ncclDataType_t: an enum
struct ncclComm;
using Comm = ncclComm*;
struct UniqueId {
    char internal[128];
};
#endif

// (values of ncclDataType_t / ncclResult_t used by the product; the check compares them with the header's)
inline constexpr int kSuccess = 0;
inline constexpr int kFloat32 = 7;
inline constexpr int kFloat64 = 8;
inline constexpr int kUniqueIdBytes = 128;

using GetVersionFn     = Result (*)(int*);
using GetUniqueIdFn    = Result (*)(UniqueId*);
using CommInitRankFn   = Result (*)(Comm*, int, UniqueId, int);
using CommInitAllFn    = Result (*)(Comm*, int, const int*);
using CommDestroyFn    = Result (*)(Comm);
using SendFn           = Result (*)(const void*, size_t, DataType, int, Comm, hipStream_t);
using RecvFn           = Result (*)(void*, size_t, DataType, int, Comm, hipStream_t);
using AllGatherFn      = Result (*)(const void*, void*, size_t, DataType, Comm, hipStream_t);
using GroupStartFn     = Result (*)();
using GroupEndFn       = Result (*)();
using GetErrorStringFn = const char* (*)(Result);

}  // namespace nb_rccl
#endif  // NBODY_RCCL_API_H
