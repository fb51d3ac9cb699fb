// RCCL bound at run time (dlopen): the gradient all-reduce of the data-parallel train step issued from native code, on a stream the
// caller names, between the backward stages nv_vit_train_step enqueues (engine.hip) - no Python, no torch.distributed call inside a step.
// The library is NOT linked against librccl: one process per GPU already has RCCL loaded (torch's "nccl" backend IS RCCL on ROCm), and
// nv_comm_load takes the path of that copy so both users share one; a single-GPU process never needs it.
// The communicator is built from a 128-byte unique id that rank 0 creates (nv_comm_unique_id) and the ranks exchange over whatever
// channel they already have (neurovit_amd/parallel.py: one broadcast on the torch.distributed group).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string.h>

#include "../../include/neurovit_hip.h"

extern "C" void nv_set_error(const char* fmt, ...);

namespace {

typedef struct { char internal[128]; } rccl_unique_id;      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm;                                     // ncclComm_t
enum { RCCL_SUM = 0, RCCL_F16 = 6, RCCL_F32 = 7, RCCL_BF16 = 9 };      // ncclSum, ncclFloat16, ncclFloat32, ncclBfloat16 (rccl.h)

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(rccl_unique_id*) = nullptr;
  int (*CommInitRank)(rccl_comm*, int, rccl_unique_id, int) = nullptr;
  int (*CommDestroy)(rccl_comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int bind(const char* path) {
  if (g_rccl.handle) return NV_OK;
  const char* tried[4] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (int i = 0; i < 4 && !h; ++i)
    if (tried[i] && tried[i][0]) h = dlopen(tried[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) { nv_set_error("nv_comm_load: librccl not found (%s)", dlerror()); return NV_ERR_HIP; }
  Rccl r;
  r.handle = h;
  r.GetUniqueId = (int (*)(rccl_unique_id*))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (int (*)(rccl_comm*, int, rccl_unique_id, int))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (int (*)(rccl_comm))dlsym(h, "ncclCommDestroy");
  r.AllReduce = (int (*)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t))dlsym(h, "ncclAllReduce");
  r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) { nv_set_error("nv_comm_load: the library lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce"); return NV_ERR_HIP; }
  g_rccl = r;
  return NV_OK;
}

int fail(const char* what, int rc) {
  nv_set_error("%s: RCCL error %d (%s)", what, rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  return NV_ERR_HIP;
}

}  // namespace

extern "C" int nv_comm_load(const char* path) { return bind(path); }

extern "C" int nv_comm_unique_id(void* id128) {
  if (!id128) { nv_set_error("nv_comm_unique_id: null buffer"); return NV_ERR_ARG; }
  if (int rc = bind(nullptr)) return rc;
  rccl_unique_id id;
  if (int rc = g_rccl.GetUniqueId(&id)) return fail("nv_comm_unique_id", rc);
  memcpy(id128, &id, sizeof(id));
  return NV_OK;
}

extern "C" int nv_comm_init(const void* id128, int world, int rank, void** comm) {
  if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) { nv_set_error("nv_comm_init: bad arguments (world %d, rank %d)", world, rank); return NV_ERR_ARG; }
  if (int rc = bind(nullptr)) return rc;
  rccl_unique_id id;
  memcpy(&id, id128, sizeof(id));
  rccl_comm c = nullptr;
  if (int rc = g_rccl.CommInitRank(&c, world, id, rank)) return fail("nv_comm_init", rc);
  *comm = c;
  return NV_OK;
}

extern "C" int nv_comm_destroy(void* comm) {
  if (!comm || !g_rccl.handle) return NV_OK;
  if (int rc = g_rccl.CommDestroy((rccl_comm)comm)) return fail("nv_comm_destroy", rc);
  return NV_OK;
}

// In-place SUM of buf[0 .. count) over the ranks of `comm`, enqueued on `stream`.  dtype: 0 = float32, 1 = the 16-bit operand format
// (nv_set_operand_format: bf16 or fp16 messages).
extern "C" int nv_comm_all_reduce(void* comm, void* buf, long count, int dtype, void* stream) {
  if (!comm || !buf || count <= 0 || (dtype != 0 && dtype != 1)) { nv_set_error("nv_comm_all_reduce: bad arguments"); return NV_ERR_ARG; }
  if (!g_rccl.handle) { nv_set_error("nv_comm_all_reduce: RCCL is not loaded (nv_comm_init first)"); return NV_ERR_HIP; }
  const int dt = dtype == 0 ? RCCL_F32 : (nv_operand_format() == NV_OPERAND_FP16 ? RCCL_F16 : RCCL_BF16);
  if (int rc = g_rccl.AllReduce(buf, buf, (size_t)count, dt, RCCL_SUM, (rccl_comm)comm, (hipStream_t)stream)) return fail("nv_comm_all_reduce", rc);
  return NV_OK;
}
