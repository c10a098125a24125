// RCCL wrappers of the C-ABI: the path is data parallel -- every rank denoises its own images -- and the
// only exchange is the one-off broadcast of the packed weight arena from the rank that read the checkpoint.  The reference
// has no communication at all (device_id = 0 hard-coded, storage/device.py:23); a host that is not Python (no
// torch.distributed) gets the broadcast through these four entries.  librccl is opened on first use, so single-GPU users of
// libtinyfusers_hip.so never load it.
#include "common.h"
#include "../../include/tinyfusers_hip.h"
#include <dlfcn.h>
#include <string.h>

namespace {
typedef struct { char internal[128]; } rcclUniqueId;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128, rccl.h:40-43)
typedef void* rcclComm;
typedef int (*fn_get_unique_id)(rcclUniqueId*);
typedef int (*fn_comm_init_rank)(rcclComm*, int, rcclUniqueId, int);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, rcclComm, hipStream_t);
typedef int (*fn_comm_destroy)(rcclComm);
typedef const char* (*fn_error_string)(int);
struct Rccl {
  void* lib = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_broadcast broadcast = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_error_string error_string = nullptr;
} g_rccl;

int rccl_load() {
  if (g_rccl.lib) return TF_OK;
  // a librccl the process already holds (e.g. the one bundled with PyTorch) wins, so that one HIP runtime serves both
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }
  for (const char* n : names) { if (h) break; h = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
  if (!h) { tf_set_error("tf_comm: cannot open librccl (%s)", dlerror()); return TF_E_STATE; }
  g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  g_rccl.broadcast = (fn_broadcast)dlsym(h, "ncclBroadcast");
  g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  g_rccl.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.broadcast || !g_rccl.comm_destroy) {
    tf_set_error("tf_comm: librccl lacks an expected symbol");
    dlclose(h);
    return TF_E_STATE;
  }
  g_rccl.lib = h;
  return TF_OK;
}
}  // namespace

struct tfComm_st { rcclComm comm; int nranks, rank; };

#define TF_RCCL(expr)                                                                                                  \
  do {                                                                                                                 \
    int _r = (expr);                                                                                                   \
    if (_r != 0) {                                                                                                     \
      tf_set_error("%s failed: %s", #expr, g_rccl.error_string ? g_rccl.error_string(_r) : "rccl error");               \
      return 20000 + _r;                                                                                               \
    }                                                                                                                  \
  } while (0)

extern "C" {

int tf_comm_unique_id(void* id_out) {
  TF_REQUIRE(id_out, "tf_comm_unique_id: null output");
  int rc = rccl_load();
  if (rc) return rc;
  rcclUniqueId id;
  TF_RCCL(g_rccl.get_unique_id(&id));
  memcpy(id_out, &id, sizeof(id));
  return TF_OK;
}

int tf_comm_init_rank(tfComm_t* out, const void* unique_id, int nranks, int rank) {
  TF_REQUIRE(out && unique_id && nranks >= 1 && rank >= 0 && rank < nranks, "tf_comm_init_rank: bad arguments (nranks=%d rank=%d)", nranks, rank);
  int rc = rccl_load();
  if (rc) return rc;
  rcclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  tfComm_st* c = new tfComm_st{nullptr, nranks, rank};
  int r = g_rccl.comm_init_rank(&c->comm, nranks, id, rank);     // uses the calling thread's current device (tf_init)
  if (r != 0) {
    tf_set_error("ncclCommInitRank failed: %s", g_rccl.error_string ? g_rccl.error_string(r) : "rccl error");
    delete c;
    return 20000 + r;
  }
  *out = c;
  return TF_OK;
}

int tf_bcast(tfComm_t comm, void* ptr, size_t nbytes, int root, tfStream_t s) {
  TF_REQUIRE(comm && (ptr || nbytes == 0) && root >= 0 && root < comm->nranks, "tf_bcast: bad arguments (root=%d)", root);
  if (nbytes == 0) return TF_OK;
  TF_RCCL(g_rccl.broadcast(ptr, ptr, nbytes, /*ncclChar*/ 0, root, comm->comm, tf_hs(s)));
  return TF_OK;
}

int tf_comm_destroy(tfComm_t comm) {
  if (!comm) return TF_OK;
  int r = g_rccl.comm_destroy ? g_rccl.comm_destroy(comm->comm) : 0;
  delete comm;
  if (r != 0) { tf_set_error("ncclCommDestroy failed"); return 20000 + r; }
  return TF_OK;
}

}  // extern "C"
