// Gradient exchange over RCCL behind the C-ABI (mm_comm_init / allreduce_bucket / finalize).
//
// Replaces what the reference leaves to DeepSpeed (config/deepspeed.json:5-19: ZeRO gradient reduce + parameter
// gather) and torch.distributed's NCCL backend (cli/train.py:200-201).  One communicator per process (= per GPU);
// the caller owns the stream, so a bucket's collective is ordered after the kernels that wrote the bucket and
// overlaps the rest of backward.
//
// RCCL is bound at RUN time (dlopen of the librccl that the process already holds: torch ships one; /opt/rocm/lib has
// another), so libmmhip.so keeps loading on a box without RCCL and mm_comm_init then fails with MM_ERR_UNSUPPORTED.
// No device code in this file.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/mm_hip.h"

namespace {

// The part of rccl.h this file uses (ROCm 7.2 /opt/rocm/include/rccl/rccl.h:40-43,187,220,260,448-468,611,655,678).
struct UniqueId { char internal[128]; };
typedef void* Comm;
enum { kSum = 0, kFloat32 = 7, kBfloat16 = 9 };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*ReduceScatterFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, Comm, hipStream_t);
typedef int (*GroupFn)();

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  ReduceScatterFn reduce_scatter = nullptr;
  AllGatherFn all_gather = nullptr;
  GroupFn group_start = nullptr, group_end = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    // RTLD_NOLOAD first: reuse the copy torch.distributed already runs on (two RCCL instances in one process would each
    // build their own topology and IPC state)
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      x.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      if (x.handle) break;
    }
    if (!x.handle)
      for (const char* n : names) {
        x.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (x.handle) break;
      }
    if (!x.handle) return x;
    x.get_unique_id = (GetUniqueIdFn)dlsym(x.handle, "ncclGetUniqueId");
    x.comm_init_rank = (CommInitRankFn)dlsym(x.handle, "ncclCommInitRank");
    x.comm_destroy = (CommDestroyFn)dlsym(x.handle, "ncclCommDestroy");
    x.all_reduce = (AllReduceFn)dlsym(x.handle, "ncclAllReduce");
    x.reduce_scatter = (ReduceScatterFn)dlsym(x.handle, "ncclReduceScatter");
    x.all_gather = (AllGatherFn)dlsym(x.handle, "ncclAllGather");
    x.group_start = (GroupFn)dlsym(x.handle, "ncclGroupStart");
    x.group_end = (GroupFn)dlsym(x.handle, "ncclGroupEnd");
    x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_reduce && x.reduce_scatter && x.all_gather &&
           x.group_start && x.group_end;
    return x;
  }();
  return r;
}

struct MmComm {
  Comm comm;
  int rank, world;
};

inline int nccl_dtype(int dtype) { return dtype == MM_BF16 ? kBfloat16 : kFloat32; }
inline size_t elem_size(int dtype) { return dtype == MM_BF16 ? 2 : 4; }

}  // namespace

extern "C" int mm_comm_unique_id(void* id128) {
  if (!id128) return MM_ERR_ARG;
  if (!rccl().ok) return MM_ERR_UNSUPPORTED;
  UniqueId id;
  if (rccl().get_unique_id(&id) != 0) return MM_ERR_LAUNCH;
  memcpy(id128, id.internal, 128);
  return MM_OK;
}

extern "C" int mm_comm_init(const void* id128, int rank, int world, void** comm_out) {
  if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return MM_ERR_ARG;
  if (!rccl().ok) return MM_ERR_UNSUPPORTED;
  UniqueId id;
  memcpy(id.internal, id128, 128);
  Comm c = nullptr;
  if (rccl().comm_init_rank(&c, world, id, rank) != 0 || !c) return MM_ERR_LAUNCH;
  *comm_out = new MmComm{c, rank, world};
  return MM_OK;
}

extern "C" int mm_comm_finalize(void* comm) {
  if (!comm) return MM_ERR_ARG;
  MmComm* m = (MmComm*)comm;
  const int r = rccl().comm_destroy(m->comm);
  delete m;
  return r == 0 ? MM_OK : MM_ERR_LAUNCH;
}

extern "C" int mm_comm_rank(void* comm, int* rank, int* world) {
  if (!comm) return MM_ERR_ARG;
  if (rank) *rank = ((MmComm*)comm)->rank;
  if (world) *world = ((MmComm*)comm)->world;
  return MM_OK;
}

// Sum `count` elements in place across the ranks.
//   algo 0: one ncclAllReduce (RCCL picks ring / tree / direct by size and topology).
//   algo 1: reduce-scatter + all-gather as two collectives of one group, in place (shard r of the bucket is rank r's):
//           the form a sharded optimiser step slots into (update shard r between the two); the tail count % world goes
//           through a small all-reduce.  Same sums in the same order on every rank either way.
extern "C" int mm_comm_allreduce_bucket(void* comm, int dtype, void* buf, int64_t count, int algo, void* stream) {
  if (!comm || !buf || count < 0 || (dtype != MM_BF16 && dtype != MM_F32) || (algo != 0 && algo != 1)) return MM_ERR_ARG;
  if (count == 0) return MM_OK;
  MmComm* m = (MmComm*)comm;
  Rccl& r = rccl();
  hipStream_t s = (hipStream_t)stream;
  const int dt = nccl_dtype(dtype);
  if (algo == 0 || count < m->world) return r.all_reduce(buf, buf, (size_t)count, dt, kSum, m->comm, s) == 0 ? MM_OK : MM_ERR_LAUNCH;
  const int64_t shard = count / m->world, body = shard * m->world;
  char* b = (char*)buf;
  char* mine = b + (size_t)m->rank * shard * elem_size(dtype);
  int rc = r.group_start();
  rc |= r.reduce_scatter(b, mine, (size_t)shard, dt, kSum, m->comm, s);
  rc |= r.all_gather(mine, b, (size_t)shard, dt, m->comm, s);
  if (body < count) rc |= r.all_reduce(b + body * elem_size(dtype), b + body * elem_size(dtype), (size_t)(count - body), dt, kSum, m->comm, s);
  rc |= r.group_end();
  return rc == 0 ? MM_OK : MM_ERR_LAUNCH;
}

// The two halves on their own (sharded optimiser: reduce-scatter the gradients, update shard `rank`, all-gather the
// parameters).  shard = count / world elements; count must be a multiple of world.
extern "C" int mm_comm_reduce_scatter(void* comm, int dtype, void* buf, int64_t count, void* stream) {
  if (!comm || !buf || count < 0 || (dtype != MM_BF16 && dtype != MM_F32)) return MM_ERR_ARG;
  MmComm* m = (MmComm*)comm;
  if (count % m->world) return MM_ERR_ARG;
  if (count == 0) return MM_OK;
  const int64_t shard = count / m->world;
  char* b = (char*)buf;
  return rccl().reduce_scatter(b, b + (size_t)m->rank * shard * elem_size(dtype), (size_t)shard, nccl_dtype(dtype), kSum, m->comm,
                               (hipStream_t)stream) == 0 ? MM_OK : MM_ERR_LAUNCH;
}

extern "C" int mm_comm_all_gather(void* comm, int dtype, void* buf, int64_t count, void* stream) {
  if (!comm || !buf || count < 0 || (dtype != MM_BF16 && dtype != MM_F32)) return MM_ERR_ARG;
  MmComm* m = (MmComm*)comm;
  if (count % m->world) return MM_ERR_ARG;
  if (count == 0) return MM_OK;
  const int64_t shard = count / m->world;
  char* b = (char*)buf;
  return rccl().all_gather(b + (size_t)m->rank * shard * elem_size(dtype), b, (size_t)shard, nccl_dtype(dtype), m->comm,
                           (hipStream_t)stream) == 0 ? MM_OK : MM_ERR_LAUNCH;
}
