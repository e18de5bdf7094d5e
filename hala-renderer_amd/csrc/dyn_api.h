// dyn_api.h — the two optional libraries of libhalart.so, resolved on first use instead of at load time:
//   RCCL  (multi-GPU tile all-gather, hala_rt_comm_* / hala_rt_tile_allgather*): a host that renders on one GPU never touches it and
//         needs no librccl installed; a host that attaches its own ncclComm_t (hala_rt_comm_attach) gets the RCCL instance that is
//         ALREADY in the process (RTLD_NOLOAD first), i.e. the one its communicator came from, not whatever the linker bound at build time;
//   roctx (ranges around commit / update / refit / tile_allgather for rocprofv3 --marker-trace): no-ops when the profiler SDK is absent.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: nothing is linked

#include <string>

namespace rt {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*);
  ncclResult_t (*CommCount)(const ncclComm_t, int*);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  const char* (*GetErrorString)(ncclResult_t);
};
// nullptr + *err when librccl cannot be loaded or lacks a symbol
const RcclApi* rccl_api(std::string* err);

void roctx_push(const char* name);
void roctx_pop();

}  // namespace rt
