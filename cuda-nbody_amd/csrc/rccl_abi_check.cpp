// rccl_abi_check.cpp -- compile-time check of rccl_api.h against the RCCL header of this image (`make check-rccl-abi`).
// A host-only translation unit: nothing here runs, nothing is linked into the product; if it compiles, every function-pointer
// type nbody_comm.hip calls RCCL through is the type of the function the header declares, and every restated alias of the
// product build (int for the two enums, a 128-char struct for the id) has the representation of the real type.
#include <rccl/rccl.h>

#define NB_RCCL_API_REAL_HEADER 1
#include "rccl_api.h"

#include <type_traits>

namespace {
using namespace nb_rccl;

// the signatures, argument for argument
static_assert(std::is_same_v<decltype(&ncclGetVersion), GetVersionFn>);
static_assert(std::is_same_v<decltype(&ncclGetUniqueId), GetUniqueIdFn>);
static_assert(std::is_same_v<decltype(&ncclCommInitRank), CommInitRankFn>);
static_assert(std::is_same_v<decltype(&ncclCommInitAll), CommInitAllFn>);
static_assert(std::is_same_v<decltype(&ncclCommDestroy), CommDestroyFn>);
static_assert(std::is_same_v<decltype(&ncclSend), SendFn>);
static_assert(std::is_same_v<decltype(&ncclRecv), RecvFn>);
static_assert(std::is_same_v<decltype(&ncclAllGather), AllGatherFn>);
static_assert(std::is_same_v<decltype(&ncclGroupStart), GroupStartFn>);
static_assert(std::is_same_v<decltype(&ncclGroupEnd), GroupEndFn>);
static_assert(std::is_same_v<decltype(&ncclGetErrorString), GetErrorStringFn>);

// what the product build puts in place of the aliases: same size, same calling-convention class, same values
static_assert(std::is_enum_v<ncclResult_t> && sizeof(ncclResult_t) == sizeof(int), "ncclResult_t is passed and returned as a 4-byte integer");
static_assert(std::is_enum_v<ncclDataType_t> && sizeof(ncclDataType_t) == sizeof(int), "ncclDataType_t is passed as a 4-byte integer");
static_assert(std::is_pointer_v<ncclComm_t> && std::is_class_v<std::remove_pointer_t<ncclComm_t>>, "ncclComm_t is a pointer to an opaque struct");
static_assert(sizeof(ncclUniqueId) == kUniqueIdBytes && NCCL_UNIQUE_ID_BYTES == kUniqueIdBytes);
static_assert(std::is_standard_layout_v<ncclUniqueId> && std::is_trivially_copyable_v<ncclUniqueId> && alignof(ncclUniqueId) == 1,
              "ncclUniqueId is passed BY VALUE to ncclCommInitRank: a 128-byte trivially copyable struct of chars, like the restated one");
static_assert(sizeof(ncclUniqueId::internal) == kUniqueIdBytes);
static_assert(static_cast<int>(ncclSuccess) == kSuccess);
static_assert(static_cast<int>(ncclFloat32) == kFloat32 && static_cast<int>(ncclFloat64) == kFloat64);
}  // namespace

int main() { return 0; }
