// K12 of SURVEY 8(b): the data-parallel gradient exchange as a C-ABI entry point - ONE ncclAllReduce(sum) over a range of the flat fp32
// gradient buffer on RCCL / xGMI (reference practice: DDP under `accelerate`, train_mnist.py:114-126).  One process per GPU.
// RCCL is bound at run time (dlopen / dlsym): a process that already carries a copy (PyTorch ships its own librccl.so) keeps using that one,
// and the library has no link-time dependency on it.  The Python path uses torch.distributed (the same RCCL) by default; these entry points
// serve hosts without PyTorch and `optim.GradReducer(backend='tfx')`.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstring>
#include "../../include/tfx.h"

namespace {
typedef struct { char internal[128]; } nccl_uid;                          // ncclUniqueId: 128 opaque bytes (rccl.h)
typedef void* nccl_comm;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(nccl_comm*, int, nccl_uid, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_destroy)(nccl_comm);
struct Rccl {
  void* h = nullptr; fn_get_uid get_uid = nullptr; fn_init_rank init_rank = nullptr; fn_all_reduce all_reduce = nullptr; fn_destroy destroy = nullptr;
  nccl_comm comm = nullptr; int world = 0, rank = -1;
  int load() {
    if (h) return 0;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return -130;
    get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId"); init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce"); destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    return (get_uid && init_rank && all_reduce && destroy) ? 0 : -131;
  }
} g;
}  // namespace

extern "C" {
int tfx_allreduce_unique_id(void* out128) {            // rank 0 calls this and ships the 128 bytes to the other ranks (any host channel)
  if (!out128) return -1;
  int rc = g.load(); if (rc) return rc;
  nccl_uid id; rc = g.get_uid(&id); if (rc) return 1000 + rc;
  memcpy(out128, &id, 128);
  return 0;
}
int tfx_allreduce_init(int32_t rank, int32_t world, const void* unique_id128) {   // the calling thread's current HIP device joins the communicator
  if (!unique_id128 || world < 1 || rank < 0 || rank >= world) return -1;
  int rc = g.load(); if (rc) return rc;
  if (g.comm) { g.destroy(g.comm); g.comm = nullptr; }
  nccl_uid id; memcpy(&id, unique_id128, 128);
  rc = g.init_rank(&g.comm, world, id, rank); if (rc) return 1000 + rc;
  g.world = world; g.rank = rank;
  return 0;
}
int tfx_allreduce_run(float* buf, int64_t count, void* stream) {                   // in place, fp32 sum over ranks, enqueued on `stream`
  if (!g.comm) return -132;
  if (count <= 0) return 0;
  if (!buf) return -1;
  const int rc = g.all_reduce(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, g.comm, (hipStream_t)stream);
  return rc ? 1000 + rc : 0;
}
int tfx_allreduce_destroy(void) {
  if (g.comm) { const int rc = g.destroy(g.comm); g.comm = nullptr; g.world = 0; g.rank = -1; return rc ? 1000 + rc : 0; }
  return 0;
}
}
