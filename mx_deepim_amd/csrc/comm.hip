// Multi-GPU exchange step of the refinement loop: one RCCL all-gather of the refined poses
// ((B,3,4) fp32 = 48 B per pair) per refinement iteration, enqueued on the library's own stream — no host sync
// inside the loop and no PyTorch.  The reference's counterpart is the per-GPU executor group handing every
// host the outputs of all devices (deepim/core/DataParallelExecutorGroup.py:364-388, deepim/test.py:135).
//
// One process per GPU.  librccl.so is opened lazily (dlopen) the first time a communicator is asked for, so a
// single-GPU process never loads it.  Bootstrap: rank 0 calls deepim_comm_unique_id and ships the 128 bytes to the
// other ranks by any out-of-band channel (mx_deepim_amd/parallel.py: a small TCP rendezvous on MASTER_ADDR);
// every rank then calls deepim_comm_init with the same bytes.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;             // optional (reporting only)
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
};
RcclApi g_rccl;

int load_rccl() {
  if (g_rccl.handle) return 0;
  // Resolved ONCE, to an absolute file, in a fixed order — so that every rank of a job binds the same library whatever else its
  // process has loaded: (1) $DEEPIM_RCCL_PATH; (2) a copy already mapped into the process (a host program that linked one: dlopen of
  // the soname then returns THAT handle — RTLD_NOLOAD asks without loading); (3) librccl.so.1 next to the libamdhip64 this library
  // itself runs on (the ROCm installation's own); (4) the bare sonames through the loader's search path.
  void* h = nullptr;
  if (const char* env = getenv("DEEPIM_RCCL_PATH"))
    if (env[0] == '/') h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
  if (!h) {
    Dl_info dh;
    if (dladdr((void*)&hipGetDeviceCount, &dh) && dh.dli_fname) {
      char path[1024];
      snprintf(path, sizeof(path), "%s", dh.dli_fname);
      if (char* slash = strrchr(path, '/')) {
        snprintf(slash + 1, sizeof(path) - (size_t)(slash + 1 - path), "librccl.so.1");
        h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
      }
    }
  }
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    deepim_set_error_msg("comm: librccl.so not found (dlopen)");
    return -1;
  }
#define DI_SYM(field, name)                                         \
  *(void**)(&g_rccl.field) = dlsym(h, name);                        \
  if (!g_rccl.field) {                                              \
    deepim_set_error_msg("comm: symbol " name " missing in librccl"); \
    dlclose(h);                                                     \
    return -1;                                                      \
  }
  DI_SYM(GetUniqueId, "ncclGetUniqueId");
  DI_SYM(CommInitRank, "ncclCommInitRank");
  DI_SYM(CommDestroy, "ncclCommDestroy");
  DI_SYM(AllGather, "ncclAllGather");
  DI_SYM(AllReduce, "ncclAllReduce");
  DI_SYM(GetErrorString, "ncclGetErrorString");
#undef DI_SYM
  *(void**)(&g_rccl.GetVersion) = dlsym(h, "ncclGetVersion");
  *(void**)(&g_rccl.CommCount) = dlsym(h, "ncclCommCount");
  g_rccl.handle = h;
  return 0;
}

int rccl_fail(const char* where, ncclResult_t r) {
  char msg[256];
  snprintf(msg, sizeof(msg), "%s: %s", where, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
  deepim_set_error_msg(msg);
  return 1000 + (int)r;
}
#define DI_RCCL(expr)                                    \
  do {                                                   \
    ncclResult_t _r = (expr);                            \
    if (_r != ncclSuccess) return rccl_fail(#expr, _r);  \
  } while (0)

}  // namespace

extern "C" int deepim_comm_unique_id(void* id_bytes) {
  DI_REQUIRE(id_bytes != nullptr, "comm_unique_id: NULL buffer");
  if (int rc = load_rccl()) return rc;
  static_assert(sizeof(ncclUniqueId) == DEEPIM_COMM_ID_BYTES, "ncclUniqueId size");
  DI_RCCL(g_rccl.GetUniqueId((ncclUniqueId*)id_bytes));
  return 0;
}

extern "C" int deepim_comm_init(deepim_ctx* ctx, int rank, int world, const void* id_bytes) {
  DI_DEVICE(ctx);
  DI_REQUIRE(world >= 1 && rank >= 0 && rank < world && id_bytes != nullptr, "comm_init: bad rank/world/id");
  DI_REQUIRE(ctx->comm == nullptr, "comm_init: communicator already initialised");
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t comm;
  DI_RCCL(g_rccl.CommInitRank(&comm, world, id, rank));
  ctx->comm = (void*)comm;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return 0;
}

extern "C" int deepim_comm_destroy(deepim_ctx* ctx) {
  DI_DEVICE(ctx);
  if (!ctx->comm) return 0;
  DI_CHECK(hipStreamSynchronize(ctx->stream));
  DI_RCCL(g_rccl.CommDestroy((ncclComm_t)ctx->comm));
  ctx->comm = nullptr;
  ctx->comm_world = 1;
  ctx->comm_rank = 0;
  return 0;
}

// all_poses (world·B,3,4) ← every rank's poses (B,3,4), rank-major; one enqueue on ctx->stream, asynchronous
extern "C" int deepim_allgather_poses(deepim_ctx* ctx, float* all_poses, const float* poses, int B) {
  DI_DEVICE(ctx);
  DI_REQUIRE(B >= 0 && all_poses != nullptr && poses != nullptr, "allgather_poses: bad arguments");
  if (B == 0) return 0;
  if (!ctx->comm) {   // single process: the gather is the identity
    if (all_poses != poses) DI_CHECK(hipMemcpyAsync(all_poses, poses, (size_t)B * 48, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
  }
  DI_RCCL(g_rccl.AllGather(poses, all_poses, (size_t)B * 12, ncclFloat, (ncclComm_t)ctx->comm, ctx->stream));
  return 0;
}

// in-place MAX (op 0) / SUM (op 1) all-reduce of n doubles that live on the device; asynchronous on ctx->stream
extern "C" int deepim_comm_allreduce_f64(deepim_ctx* ctx, double* buf, int n, int op) {
  DI_DEVICE(ctx);
  DI_REQUIRE(n >= 0 && (op == 0 || op == 1), "comm_allreduce: bad arguments");
  if (n == 0 || !ctx->comm) return 0;
  DI_RCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclDouble, op == 0 ? ncclMax : ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
  return 0;
}

// What this process actually bound ("nothing records which librccl / libamdhip64 a rank bound"): a text record
//   backend=rccl|none;rccl_ranks=<ncclCommCount of this context's communicator, 0 without one>;rccl_version=<ncclGetVersion>;
//   librccl_path=<file the ncclAllGather symbol in use lives in>;libamdhip64_path=<file the HIP runtime symbols in use live in>
// written into buf (NUL-terminated, truncated to n). bench.py puts it into the `comm` block of its JSON line.
extern "C" int deepim_comm_info(deepim_ctx* ctx, char* buf, int n) {
  DI_REQUIRE(buf != nullptr && n > 0, "comm_info: no buffer");
  int version = 0, ranks = 0;
  const char* rccl_path = "";
  Dl_info di;
  if (g_rccl.handle) {
    if (g_rccl.GetVersion) g_rccl.GetVersion(&version);
    if (dladdr((void*)g_rccl.AllGather, &di) && di.dli_fname) rccl_path = di.dli_fname;
    if (ctx && ctx->comm && g_rccl.CommCount) g_rccl.CommCount((ncclComm_t)ctx->comm, &ranks);
  }
  const char* hip_path = "";
  Dl_info dh;
  if (dladdr((void*)&hipGetDeviceCount, &dh) && dh.dli_fname) hip_path = dh.dli_fname;
  snprintf(buf, (size_t)n, "backend=%s;rccl_ranks=%d;rccl_version=%d;librccl_path=%s;libamdhip64_path=%s",
           (ctx && ctx->comm) ? "rccl" : "none", ranks, version, rccl_path, hip_path);
  return 0;
}
