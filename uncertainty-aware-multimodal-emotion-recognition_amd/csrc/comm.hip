// Gradient exchange of the data-parallel step over RCCL: the only collective of the path is ONE
// exchange of the flat gradient buffer per step -- as one all-reduce, or as reduce-scatter + all-gather (every rank reduces 1/N
// of the buffer from all peers at once, then every rank fetches the other shards: on point-to-point xGMI that uses all seven
// links of a GPU simultaneously, where a ring all-reduce is bound by one link).  mmdeer_comm_* wrap an RCCL communicator behind the C ABI so that a
// host without torch.distributed can run the exchange (the Python host, mmdeer/parallel.py, uses torch.distributed's
// RCCL communicator by default and this one with MMDEER_COMM=rccl).  RCCL is bound at run time (dlopen of the
// librccl that is already loaded into the process when there is one): libmmdeer_hip.so itself has no link-time
// dependency on it and loads on hosts without RCCL.
#include <dlfcn.h>

#include <cstring>

#include "common.h"

#include "../../include/mmdeer.h"

namespace mmdeer {
namespace {

// the slice of rccl.h this file needs (ABI-stable NCCL 2.x definitions; /opt/rocm/include/rccl/rccl.h:40-52, 448-468)
struct UniqueId { char internal[128]; };
typedef void* Comm;
enum { kSuccess = 0, kSum = 0, kAvg = 4, kFloat32 = 7, kBfloat16 = 9 };
static_assert(MMDEER_COMM_ID_BYTES == sizeof(UniqueId), "NCCL_UNIQUE_ID_BYTES");

struct Api {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

const Api* api() {
  static Api a;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (int pass = 0; pass < 2 && !a.handle; ++pass)        // pass 0: only a copy that is already loaded (torch's)
      for (const char* n : names)
        if (!a.handle) a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
    if (a.handle) {
      a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
      a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
      a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
      a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.handle, "ncclAllReduce"));
      a.ReduceScatter = reinterpret_cast<decltype(a.ReduceScatter)>(dlsym(a.handle, "ncclReduceScatter"));
      a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.handle, "ncclAllGather"));
      a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
    }
  }
  return (a.handle && a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.ReduceScatter && a.AllGather) ? &a : nullptr;
}

#define RCCL_TRY(call)                                                                                   \
  do {                                                                                                   \
    const int rc_ = (call);                                                                              \
    if (rc_ != kSuccess) {                                                                               \
      set_error("%s failed: %s", #call, r->GetErrorString ? r->GetErrorString(rc_) : "RCCL error");      \
      return -1;                                                                                         \
    }                                                                                                    \
  } while (0)

}  // namespace
}  // namespace mmdeer

using namespace mmdeer;

struct mmdeer_comm {
  Comm comm;
  int rank, world;
};

extern "C" {

int mmdeer_comm_unique_id(void* id_out) {
  MMDEER_CHECK(id_out != nullptr, "comm_unique_id: NULL output");
  const Api* r = api();
  MMDEER_CHECK(r != nullptr, "comm: librccl.so is not available in this process (dlopen failed)");
  RCCL_TRY(r->GetUniqueId(reinterpret_cast<UniqueId*>(id_out)));
  return 0;
}

int mmdeer_comm_init(mmdeer_comm** comm, int rank, int world_size, const void* id) {
  MMDEER_CHECK(comm != nullptr && id != nullptr, "comm_init: NULL argument");
  MMDEER_CHECK(world_size >= 1 && rank >= 0 && rank < world_size, "comm_init: bad rank %d of %d", rank, world_size);
  const Api* r = api();
  MMDEER_CHECK(r != nullptr, "comm: librccl.so is not available in this process (dlopen failed)");
  UniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  Comm c = nullptr;
  RCCL_TRY(r->CommInitRank(&c, world_size, uid, rank));     // binds the calling thread's current HIP device
  *comm = new mmdeer_comm{c, rank, world_size};
  return 0;
}

int mmdeer_comm_destroy(mmdeer_comm* comm) {
  if (!comm) return 0;
  const Api* r = api();
  MMDEER_CHECK(r != nullptr, "comm: librccl.so is not available in this process");
  const int rc = r->CommDestroy(comm->comm);
  delete comm;
  MMDEER_CHECK(rc == kSuccess, "ncclCommDestroy failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "RCCL error");
  return 0;
}

int mmdeer_allreduce(void* buf, long long count, int dtype_f32, int average, mmdeer_comm* comm, void* stream) {
  MMDEER_CHECK(comm != nullptr, "allreduce: NULL communicator");
  MMDEER_CHECK(count >= 0, "allreduce: count must be >= 0 (got %lld)", count);
  if (count == 0) return 0;
  MMDEER_CHECK(buf != nullptr, "allreduce: NULL buffer");
  const Api* r = api();
  MMDEER_CHECK(r != nullptr, "comm: librccl.so is not available in this process");
  RCCL_TRY(r->AllReduce(buf, buf, (size_t)count, dtype_f32 ? kFloat32 : kBfloat16, average ? kAvg : kSum, comm->comm,
                        (hipStream_t)stream));      // in place, enqueued on the caller's stream, capturable into a HIP graph
  return 0;
}

int mmdeer_comm_rank(const mmdeer_comm* comm) { return comm ? comm->rank : -1; }
int mmdeer_comm_world(const mmdeer_comm* comm) { return comm ? comm->world : -1; }

// Reduce-scatter: `send` holds world * recv_count elements; rank r receives the reduction over ranks of elements
// [r * recv_count, (r + 1) * recv_count) in `recv` (which may be send + r * recv_count: in place).
int mmdeer_reduce_scatter(const void* send, void* recv, long long recv_count, int dtype_f32, int average, mmdeer_comm* comm, void* stream) {
  MMDEER_CHECK(comm != nullptr, "reduce_scatter: NULL communicator");
  MMDEER_CHECK(recv_count >= 0, "reduce_scatter: recv_count must be >= 0 (got %lld)", recv_count);
  if (recv_count == 0) return 0;
  MMDEER_CHECK(send != nullptr && recv != nullptr, "reduce_scatter: NULL buffer");
  const Api* r = api();
  MMDEER_CHECK(r != nullptr, "comm: librccl.so is not available in this process");
  RCCL_TRY(r->ReduceScatter(send, recv, (size_t)recv_count, dtype_f32 ? kFloat32 : kBfloat16, average ? kAvg : kSum, comm->comm,
                            (hipStream_t)stream));
  return 0;
}

// All-gather: every rank contributes send_count elements; `recv` receives world * send_count (rank r's at offset r * send_count;
// `send` may be recv + rank * send_count: in place).
int mmdeer_allgather(const void* send, void* recv, long long send_count, int dtype_f32, mmdeer_comm* comm, void* stream) {
  MMDEER_CHECK(comm != nullptr, "allgather: NULL communicator");
  MMDEER_CHECK(send_count >= 0, "allgather: send_count must be >= 0 (got %lld)", send_count);
  if (send_count == 0) return 0;
  MMDEER_CHECK(send != nullptr && recv != nullptr, "allgather: NULL buffer");
  const Api* r = api();
  MMDEER_CHECK(r != nullptr, "comm: librccl.so is not available in this process");
  RCCL_TRY(r->AllGather(send, recv, (size_t)send_count, dtype_f32 ? kFloat32 : kBfloat16, comm->comm, (hipStream_t)stream));
  return 0;
}

}  // extern "C"
