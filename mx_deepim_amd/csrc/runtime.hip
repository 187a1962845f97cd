// Runtime half of the C ABI: context, device memory, copies, HIP-event stopwatch,
// hipGraph capture. Replaces the MXNet context/NDArray plumbing the reference's
// Predictor relies on (deepim/core/tester.py:27-47).
#include "common.h"
#include <limits.h>

static thread_local char g_err[512] = "";

void deepim_set_error(const char* where, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}
void deepim_set_error_msg(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

extern "C" const char* deepim_last_error(void) { return g_err; }

extern "C" int deepim_device_count(int* n) {
  DI_CHECK(hipGetDeviceCount(n));
  return 0;
}

// The option defaults of a fresh context (also what the host-arithmetic queries — deepim_conv_wino_plan — assume without a context)
void deepim_ctx_default_options(deepim_ctx* c) {
  c->capturing = false;
  c->conv_max_split = 0;
  c->conv_xcd_swizzle = 1;
  c->f16_dev_flags = 0;
  c->wgrad_lds = 1;
  c->dgrad_group = 1;
  c->wino_two_wave = 0;
  c->wino_shared = 1;
  c->wino_persistent = 1;
  c->wino_streamk = 1;
  c->wino_fin = 0;   // measured slower than the second pass at every batch size (profiles/r06_b4_share.md)
  c->conv_fewout_quad = 1;
  c->conv_fewout_blocks = 0;     // 0 = by the grid (launch site)
  c->conv_fewout_minc = 32;
  c->wino_counters = nullptr;
  c->wino_split = 0;
  c->wino_wide = 1;
  c->wino_s2d_skip = 1;
  c->conv_autotune = 0;
  c->conv_direct = 1;
  c->conv_tail_slots = 1024;
  c->conv_force_plan = 0;
  c->fc_slices = 0;
  c->conv_tail_split = 0;
  c->conv_tile256 = 0;   // measured: 113.1 vs 113.7 TF for 128x128 — kept as an option, off by default
}

extern "C" int deepim_create(int device_id, deepim_ctx** out) {
  DI_REQUIRE(out != nullptr, "deepim_create: out is NULL");
  int n = 0;
  DI_CHECK(hipGetDeviceCount(&n));
  DI_REQUIRE(device_id >= 0 && device_id < n, "deepim_create: no such device");
  DI_CHECK(hipSetDevice(device_id));
  deepim_ctx* c = new deepim_ctx();
  c->device = device_id;
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  c->capturing = false;
  c->sync_event = nullptr;
  c->comm = nullptr;
  c->comm_rank = 0;
  c->comm_world = 1;
  deepim_ctx_default_options(c);
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    deepim_set_error("hipStreamCreate", e);
    delete c;
    return (int)e;
  }
  // status word + the bbox accumulators of deepim_mask_box_forward
  const size_t box_bytes = (size_t)DI_MAX_BOX_SAMPLES * 4 * sizeof(int), zbox_bytes = (size_t)DI_MAX_BOX_SAMPLES * 8 * sizeof(int);
  c->zbuf = nullptr;
  c->zbuf_bytes = 0;
  c->zbuf_dirty = c->zoom_box_dirty = 0;
  e = hipMalloc((void**)&c->status, 64 + box_bytes + zbox_bytes);
  if (e == hipSuccess) e = hipMemsetAsync(c->status, 0, 64, c->stream);
  if (e == hipSuccess) {
    c->box_words = c->status + 16;
    c->zoom_box = c->box_words + (size_t)DI_MAX_BOX_SAMPLES * 4;
    std::vector<int> init((size_t)DI_MAX_BOX_SAMPLES * 8);
    for (size_t i = 0; i < init.size(); ++i) init[i] = (i & 1) ? -1 : INT_MAX;      // {minx,maxx,miny,maxy}: empty boxes
    e = hipMemcpy(c->zoom_box, init.data(), zbox_bytes, hipMemcpyHostToDevice);
  }
  // arrival counters of the Winograd kernels' in-kernel finish (stream-K pieces, K slices): one word per tile block, zero between
  // launches (the kernels leave them so). Allocated HERE so that a layer's plan is a function of geometry and options only
  if (e == hipSuccess) e = hipMalloc(&c->wino_counters, (size_t)DI_WINO_COUNTERS * sizeof(int));
  if (e == hipSuccess) e = hipMemsetAsync(c->wino_counters, 0, (size_t)DI_WINO_COUNTERS * sizeof(int), c->stream);
  if (e != hipSuccess) {
    if (c->status) hipFree(c->status);
    if (c->wino_counters) hipFree(c->wino_counters);
    deepim_set_error("hipMalloc(status)", e);
    hipStreamDestroy(c->stream);
    delete c;
    return (int)e;
  }
  *out = c;
  return 0;
}

extern "C" int deepim_destroy(deepim_ctx* ctx) {
  if (!ctx) return 0;
  DI_DEVICE(ctx);
  hipStreamSynchronize(ctx->stream);
  if (ctx->comm) deepim_comm_destroy(ctx);
  for (auto g : ctx->graphs) hipGraphExecDestroy(g);
  if (ctx->zbuf) hipFree(ctx->zbuf);
  if (ctx->wino_counters) hipFree(ctx->wino_counters);
  for (auto e : ctx->timer_start) hipEventDestroy(e);
  for (auto e : ctx->timer_stop) hipEventDestroy(e);
  if (ctx->sync_event) hipEventDestroy(ctx->sync_event);
  for (auto& t : ctx->conv_tabs) hipFree(t.tab);
  for (auto p : ctx->retired_scratch) hipFree(p);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->status) hipFree(ctx->status);
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return 0;
}

int deepim_scratch(deepim_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->scratch_bytes) {
    DI_REQUIRE(!ctx->capturing, "scratch growth during graph capture; run the sequence once eagerly first");
    DI_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) {
      // captured graphs hold the old pointer: retire it (freed at destroy) instead of freeing it under them
      if (!ctx->graphs.empty()) ctx->retired_scratch.push_back(ctx->scratch);
      else DI_CHECK(hipFree(ctx->scratch));
    }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    DI_CHECK(hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return 0;
}

extern "C" int deepim_malloc(deepim_ctx* ctx, size_t bytes, void** dptr) {
  DI_DEVICE(ctx);
  DI_CHECK(hipMalloc(dptr, bytes ? bytes : 4));
  std::lock_guard<std::mutex> lk(ctx->allocs_mu);
  ctx->allocs[(uintptr_t)*dptr] = bytes ? bytes : 4;
  return 0;
}
extern "C" int deepim_free(deepim_ctx* ctx, void* dptr) {
  DI_DEVICE(ctx);
  if (!dptr) return 0;
  DI_CHECK(hipStreamSynchronize(ctx->stream));
  DI_CHECK(hipFree(dptr));          // the table forgets the block only once the driver has (a failed free leaves it listed)
  std::lock_guard<std::mutex> lk(ctx->allocs_mu);
  ctx->allocs.erase((uintptr_t)dptr);
  return 0;
}
extern "C" int deepim_memset(deepim_ctx* ctx, void* dptr, int value, size_t bytes) {
  DI_DEVICE(ctx);
  DI_CHECK(hipMemsetAsync(dptr, value, bytes, ctx->stream));
  return 0;
}
extern "C" int deepim_h2d(deepim_ctx* ctx, void* dst, const void* src, size_t bytes) {
  DI_DEVICE(ctx);
  DI_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  DI_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}
extern "C" int deepim_d2h(deepim_ctx* ctx, void* dst, const void* src, size_t bytes) {
  DI_DEVICE(ctx);
  DI_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  DI_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}
namespace {
__global__ __launch_bounds__(256) void copy_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
}  // namespace

extern "C" int deepim_d2d(deepim_ctx* ctx, void* dst, const void* src, size_t bytes) {
  DI_DEVICE(ctx);
  if (bytes == 0) return 0;
  // small word-aligned copies (poses, head weights, 7-row gradient blocks) as a kernel in stream order: the blit path of
  // hipMemcpyAsync left 8-18 µs gaps around each of them in the training trace
  // (the kernel dereferences both pointers on ctx's device: only when both allocations live there — a source on another
  // device, e.g. DeviceArray.copyfrom across contexts, keeps the runtime's peer copy). Residency: the context's own allocation table
  // first (deepim_malloc / deepim_free keep it: no driver call on the path this kernel exists to shorten); a pointer it does not
  // know is asked of the driver — never inside a stream capture, where an unknown pointer takes the memcpy node instead.
  auto mine = [&](const void* q, size_t n) {
    std::lock_guard<std::mutex> lk(ctx->allocs_mu);
    auto it = ctx->allocs.upper_bound((uintptr_t)q);
    if (it == ctx->allocs.begin()) return false;
    --it;
    return (uintptr_t)q + n <= it->first + it->second;
  };
  auto on_device = [&](const void* q, size_t n) {
    if (mine(q, n)) return true;
    if (ctx->capturing) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice && a.device == ctx->device;
  };
  if (bytes <= (1u << 20) && ((bytes | (size_t)dst | (size_t)src) & 3) == 0 && on_device(dst, bytes) && on_device(src, bytes)) {
    const size_t n = bytes / 4;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)dst,
                       (const uint32_t*)src, n);
    DI_LAUNCH_CHECK();
    return 0;
  }
  DI_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}
extern "C" int deepim_copy_channels(deepim_ctx* ctx, float* dst, int dst_ctotal, int dst_coff, const float* src, int C,
                                    int B, size_t hw) {
  DI_DEVICE(ctx);
  if (B == 0 || C == 0) return 0;
  DI_CHECK(hipMemcpy2DAsync(dst + (size_t)dst_coff * hw, (size_t)dst_ctotal * hw * sizeof(float), src,
                            (size_t)C * hw * sizeof(float), (size_t)C * hw * sizeof(float), (size_t)B,
                            hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}
extern "C" int deepim_set_option(deepim_ctx* ctx, const char* name, int value) {
  DI_DEVICE(ctx);
  if (strcmp(name, "conv_max_split") == 0) {
    DI_REQUIRE(value >= 0, "conv_max_split must be >= 0");
    ctx->conv_max_split = value;
    return 0;
  }
  if (strcmp(name, "conv_direct") == 0) { ctx->conv_direct = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }
  if (strcmp(name, "fc_slices") == 0) { ctx->fc_slices = value > 0 ? value : 0; return 0; }
  if (strcmp(name, "conv_tail_split") == 0) { ctx->conv_tail_split = value ? 1 : 0; return 0; }
  if (strcmp(name, "conv_force_plan") == 0) { ctx->conv_force_plan = value; return 0; }
  if (strcmp(name, "conv_tail_slots") == 0) { DI_REQUIRE(value >= 8 && value % 8 == 0, "conv_tail_slots must be a positive multiple of 8"); ctx->conv_tail_slots = value; return 0; }
  if (strcmp(name, "conv_tile256") == 0) { ctx->conv_tile256 = value ? 1 : 0; return 0; }
  if (strcmp(name, "conv_autotune") == 0) { ctx->conv_autotune = value ? 1 : 0; return 0; }
  if (strcmp(name, "dgrad_group") == 0) { ctx->dgrad_group = value ? 1 : 0; return 0; }
  if (strcmp(name, "wgrad_lds") == 0) { ctx->wgrad_lds = value ? 1 : 0; return 0; }
  if (strcmp(name, "wino_two_wave") == 0) { ctx->wino_two_wave = value ? 1 : 0; return 0; }
  if (strcmp(name, "wino_persistent") == 0) { ctx->wino_persistent = value ? 1 : 0; return 0; }
  if (strcmp(name, "conv_fewout_blocks") == 0) { ctx->conv_fewout_blocks = value < 0 ? 0 : value; return 0; }
  if (strcmp(name, "conv_fewout_minc") == 0) { ctx->conv_fewout_minc = value < 1 ? 1 : value; return 0; }
  if (strcmp(name, "conv_fewout_quad") == 0) { ctx->conv_fewout_quad = value ? 1 : 0; return 0; }
  if (strcmp(name, "wino_fin") == 0) { ctx->wino_fin = value != 0; return 0; }
  if (strcmp(name, "wino_streamk") == 0) { ctx->wino_streamk = value < 0 ? 0 : value > 2 ? 2 : value; return 0; }
  if (strcmp(name, "wino_split") == 0) { ctx->wino_split = value < 0 ? 0 : value; return 0; }
  if (strcmp(name, "wino_wide") == 0) { ctx->wino_wide = (value >= 0 && value <= 3) ? value : 1; return 0; }
  if (strcmp(name, "wino_shared") == 0) { ctx->wino_shared = value ? 1 : 0; return 0; }
  if (strcmp(name, "wino_s2d_skip") == 0) { ctx->wino_s2d_skip = value ? 1 : 0; return 0; }
  if (strcmp(name, "f16_dev_flags") == 0) { DI_REQUIRE(value >= 0 && value < 32, "f16_dev_flags: bits 0..4"); ctx->f16_dev_flags = value; return 0; }
  if (strcmp(name, "conv_xcd_swizzle") == 0) {
    ctx->conv_xcd_swizzle = value ? 1 : 0;
    return 0;
  }
  deepim_set_error_msg("deepim_set_option: unknown option");
  return -1;
}
// the current value of an option deepim_set_option knows (host code that must follow the context's kernel selection reads it
// instead of keeping a shadow copy)
extern "C" int deepim_get_option(deepim_ctx* ctx, const char* name, int* value) {
  DI_REQUIRE(ctx != nullptr && name != nullptr && value != nullptr, "deepim_get_option: NULL argument");
  const struct { const char* n; int v; } opts[] = {
      {"conv_max_split", ctx->conv_max_split}, {"conv_direct", ctx->conv_direct}, {"fc_slices", ctx->fc_slices},
      {"conv_tail_split", ctx->conv_tail_split}, {"conv_force_plan", ctx->conv_force_plan}, {"conv_tail_slots", ctx->conv_tail_slots},
      {"conv_tile256", ctx->conv_tile256}, {"conv_autotune", ctx->conv_autotune}, {"dgrad_group", ctx->dgrad_group},
      {"wgrad_lds", ctx->wgrad_lds}, {"wino_two_wave", ctx->wino_two_wave}, {"wino_shared", ctx->wino_shared}, {"wino_wide", ctx->wino_wide}, {"wino_split", ctx->wino_split}, {"wino_persistent", ctx->wino_persistent}, {"wino_streamk", ctx->wino_streamk}, {"wino_fin", ctx->wino_fin}, {"conv_fewout_quad", ctx->conv_fewout_quad}, {"conv_fewout_blocks", ctx->conv_fewout_blocks}, {"conv_fewout_minc", ctx->conv_fewout_minc}, {"wino_s2d_skip", ctx->wino_s2d_skip}, {"f16_dev_flags", ctx->f16_dev_flags}, {"conv_xcd_swizzle", ctx->conv_xcd_swizzle}};
  for (const auto& o : opts)
    if (strcmp(name, o.n) == 0) { *value = o.v; return 0; }
  deepim_set_error_msg("deepim_get_option: unknown option");
  return -1;
}
extern "C" int deepim_sync(deepim_ctx* ctx) {
  DI_DEVICE(ctx);
  DI_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}
extern "C" void* deepim_stream(deepim_ctx* ctx) { return (void*)ctx->stream; }

// Device-side ordering between two contexts (= two streams) of one GPU: work queued on `waiter` after this call runs only after
// everything queued on `ctx` before it. No host synchronisation.
extern "C" int deepim_stream_wait(deepim_ctx* waiter, deepim_ctx* ctx) {
  DI_REQUIRE(waiter != nullptr && ctx != nullptr && waiter->device == ctx->device, "stream_wait: two contexts of the same device");
  DI_REQUIRE(!waiter->capturing && !ctx->capturing, "stream_wait during graph capture");
  if (waiter == ctx) return 0;
  DI_DEVICE(ctx);
  if (!ctx->sync_event) DI_CHECK(hipEventCreateWithFlags(&ctx->sync_event, hipEventDisableTiming));
  DI_CHECK(hipEventRecord(ctx->sync_event, ctx->stream));
  DI_CHECK(hipStreamWaitEvent(waiter->stream, ctx->sync_event, 0));
  return 0;
}

extern "C" int deepim_timer_create(deepim_ctx* ctx, int* timer_id) {
  DI_DEVICE(ctx);
  hipEvent_t a, b;
  DI_CHECK(hipEventCreate(&a));
  DI_CHECK(hipEventCreate(&b));
  ctx->timer_start.push_back(a);
  ctx->timer_stop.push_back(b);
  *timer_id = (int)ctx->timer_start.size() - 1;
  return 0;
}
extern "C" int deepim_timer_start(deepim_ctx* ctx, int id) {
  DI_DEVICE(ctx);
  DI_REQUIRE(id >= 0 && id < (int)ctx->timer_start.size(), "bad timer id");
  DI_CHECK(hipEventRecord(ctx->timer_start[id], ctx->stream));
  return 0;
}
extern "C" int deepim_timer_stop(deepim_ctx* ctx, int id) {
  DI_DEVICE(ctx);
  DI_REQUIRE(id >= 0 && id < (int)ctx->timer_stop.size(), "bad timer id");
  DI_CHECK(hipEventRecord(ctx->timer_stop[id], ctx->stream));
  return 0;
}
extern "C" int deepim_timer_elapsed_ms(deepim_ctx* ctx, int id, float* ms) {
  DI_DEVICE(ctx);
  DI_REQUIRE(id >= 0 && id < (int)ctx->timer_stop.size(), "bad timer id");
  DI_CHECK(hipEventSynchronize(ctx->timer_stop[id]));
  DI_CHECK(hipEventElapsedTime(ms, ctx->timer_start[id], ctx->timer_stop[id]));
  return 0;
}

extern "C" int deepim_graph_begin(deepim_ctx* ctx) {
  DI_DEVICE(ctx);
  DI_REQUIRE(!ctx->capturing, "graph capture already open");
  DI_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  ctx->capturing = true;
  return 0;
}
extern "C" int deepim_graph_end(deepim_ctx* ctx, int* graph_id) {
  DI_DEVICE(ctx);
  DI_REQUIRE(ctx->capturing, "no graph capture open");
  hipGraph_t g;
  ctx->capturing = false;
  DI_CHECK(hipStreamEndCapture(ctx->stream, &g));
  hipGraphExec_t ge;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e != hipSuccess) {
    deepim_set_error("hipGraphInstantiate", e);
    return (int)e;
  }
  ctx->graphs.push_back(ge);
  *graph_id = (int)ctx->graphs.size() - 1;
  return 0;
}
extern "C" int deepim_graph_launch(deepim_ctx* ctx, int graph_id) {
  DI_DEVICE(ctx);
  DI_REQUIRE(graph_id >= 0 && graph_id < (int)ctx->graphs.size(), "bad graph id");
  DI_CHECK(hipGraphLaunch(ctx->graphs[graph_id], ctx->stream));
  return 0;
}
