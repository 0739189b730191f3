// Compile-only check (tests/test_host_cpu.py): gpexp_amd/csrc/dist.hip binds RCCL through dlopen with hand-declared function
// types (no rccl.h in the product build, so that the library loads on machines without RCCL).  Most of those entry points have
// never been CALLED with more than one rank, where a wrong argument order would first show -- so the declared types are
// pinned against the installed header here.  Integer enums are passed as int on this ABI (ncclDataType_t / ncclRedOp_t).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <type_traits>

template <class Declared, class Real>
constexpr bool same_shape() { return std::is_same<Declared, Real>::value; }

// what dist.hip declares, with the header's own enum types substituted for `int` where it passes enum values
using unique_id_t = ncclResult_t (*)(ncclUniqueId*);
using init_rank_t = ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int);
using destroy_t = ncclResult_t (*)(ncclComm_t);
using bcast_t = ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
using allreduce_t = ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
using allgather_t = ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
using reduce_t = ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t);
using split_t = ncclResult_t (*)(ncclComm_t, int, int, ncclComm_t*, ncclConfig_t*);
using send_t = ncclResult_t (*)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
using recv_t = ncclResult_t (*)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
using group_t = ncclResult_t (*)();
using errstr_t = const char* (*)(ncclResult_t);

static_assert(same_shape<unique_id_t, decltype(&ncclGetUniqueId)>(), "ncclGetUniqueId");
static_assert(same_shape<init_rank_t, decltype(&ncclCommInitRank)>(), "ncclCommInitRank");
static_assert(same_shape<destroy_t, decltype(&ncclCommDestroy)>(), "ncclCommDestroy");
static_assert(same_shape<bcast_t, decltype(&ncclBroadcast)>(), "ncclBroadcast");
static_assert(same_shape<allreduce_t, decltype(&ncclAllReduce)>(), "ncclAllReduce");
static_assert(same_shape<allgather_t, decltype(&ncclAllGather)>(), "ncclAllGather");
static_assert(same_shape<reduce_t, decltype(&ncclReduce)>(), "ncclReduce");
static_assert(same_shape<split_t, decltype(&ncclCommSplit)>(), "ncclCommSplit");
static_assert(same_shape<send_t, decltype(&ncclSend)>(), "ncclSend");
static_assert(same_shape<recv_t, decltype(&ncclRecv)>(), "ncclRecv");
static_assert(same_shape<group_t, decltype(&ncclGroupStart)>(), "ncclGroupStart");
static_assert(same_shape<group_t, decltype(&ncclGroupEnd)>(), "ncclGroupEnd");
static_assert(same_shape<errstr_t, decltype(&ncclGetErrorString)>(), "ncclGetErrorString");
// the constants dist.hip spells out
static_assert(ncclFloat64 == 8 && ncclInt64 == 4 && ncclInt8 == 0 && ncclUint8 == 1, "data type codes");
static_assert(ncclSum == 0 && ncclMax == 2, "reduction codes");
static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int) && sizeof(ncclResult_t) == sizeof(int),
              "enums travel as int");
int main() { return 0; }
