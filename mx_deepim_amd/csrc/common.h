// Shared internals of libdeepim_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <map>
#include <mutex>
#include <type_traits>
#include <algorithm>
#include "../../include/deepim_hip.h"

constexpr int DI_MAX_BOX_SAMPLES = 4096;
// bits of the sticky status word (deepim_zoom_status)
constexpr int DI_STATUS_ZOOM_EMPTY = 1;       // observed mask/image empty in a zoom-factor computation (zoom_mask.py:55 raises)
constexpr int DI_STATUS_GROUP_RANGE = 2;      // GroupPicker index out of range
constexpr int DI_STATUS_X3_SATURATED = 8;     // split-fp16 conv: a value left fp16's range after scaling and was clamped (results invalid)
constexpr int DI_STATUS_MASK_BOX_EMPTY = 4;   // mask_box / fused re-render: empty mask (data_pair.py:98 np.min raises)

// deepim_set_option(ctx, "f16_dev_flags", bits): measurement switches of csrc/conv_f16.hip
constexpr int DI_F16_TN4 = 1;      // 128x128 wave tiles, one block per CU (the round-2 starting point)
constexpr int DI_F16_W8 = 2;       // one 8-wave block per CU on a 256x256 tile
constexpr int DI_F16_NO_TAIL = 4;  // no tail split of the under-filled last round
constexpr int DI_F16_NO_DMA = 8;   // register-staged kernel instead of the LDS-DMA ring
constexpr int DI_F16_NO_PP = 16;   // no ping-pong kernel (round 4): every LDS-DMA layer on the 4-wave kernel, as in round 3
#ifndef DI_PP_MIN_TILES
#define DI_PP_MIN_TILES 64
#endif
constexpr int DI_F16_PP_MIN_TILES = DI_PP_MIN_TILES;   // ping-pong kernel from 64 tiles of 256x256 (128x512) on (measured, profiles/r04_fp16_pingpong.md): with fewer the 4-wave kernel on 256x128 tiles, two blocks per CU

struct ConvTab { int mode, Cin, kh, kw, H, W; void* tab; };  // im2col tap table of one conv geometry

struct ConvPlanKey { int mode, B, Cin, H, W, Cout, Ho, Wo, stride, pad, nchunk, below, target; };
struct ConvPlan { ConvPlanKey key; int ksplit; };  // autotuned split-K factor of one conv geometry

struct deepim_ctx {
  int device;
  hipStream_t stream;
  // small device scratch shared by the ops (bbox words, zoom factors, split-K partials)
  void* scratch;
  size_t scratch_bytes;
  std::vector<void*> retired_scratch;  // outgrown scratch buffers still referenced by captured graphs
  int* status;  // persistent device status word, bits DI_STATUS_*
  void* comm;        // ncclComm_t of this rank (csrc/comm.hip), NULL in a single-GPU process
  int comm_rank, comm_world;
  int* box_words;   // DI_MAX_BOX_SAMPLES x {xmin,xmax,ymin,ymax}: bbox accumulators of mask_box, armed inside every call
  int* zoom_box;    // DI_MAX_BOX_SAMPLES x 2 maps x {xmin,xmax,ymin,ymax}: bbox accumulators of the zoom-factor computation; armed at
                    // creation and RE-ARMED BY THEIR CONSUMER (zoom_factor_kernel resets what it read), so a call needs no init launch
  unsigned long long* zbuf;   // rasteriser z-buffer (key = depth bits << 32 | triangle): all-ones between calls — the resolve pass
  size_t zbuf_bytes;          // puts every entry back after reading it, so a draw needs no clearing pass (grow-only, cleared on growth)
  // host-side guards of the two self-re-arming buffers: set before the PRODUCER launches (raster / bbox), cleared once the CONSUMER
  // (resolve / zoom_factor) has been launched without error. Found set on entry — an aborted sequence, a launch error in between —
  // the buffer is re-initialised before it is used again instead of handing stale depth keys / boxes to the next call.
  int zbuf_dirty, zoom_box_dirty;
  // pinned host staging for small per-call attribute uploads (K, means, ...)
  std::vector<hipEvent_t> timer_start, timer_stop;
  hipEvent_t sync_event;   // deepim_stream_wait: "everything queued on this stream so far" (created on first use)
  std::vector<hipGraphExec_t> graphs;
  bool capturing;
  std::vector<ConvTab> conv_tabs;
  std::vector<ConvPlan> conv_plans;
  int conv_direct;    // LDS-free register-fed kernel for 128x128-tiled convs: 0 off, 1 (default) unless conv_max_split == 1, 2 always
  int fc_slices;        // dev: K slices of the FC GEMV (0 = enough for ~1024 blocks)
  int conv_tail_split;  // 1: let the autotuner consider tail splits (default 0, see launch_conv)
  int conv_force_plan;  // dev: 0 = off, n > 0 = uniform split-K n, n < 0 = tail split with -n slices
  int conv_tail_slots;  // resident 128x128 blocks of the LDS-free kernel on the whole chip (256 CUs x 4): round size for the tail split
  int conv_tile256;   // 1: 256x128 tiles (512-thread blocks) when Cout % 256 == 0 (default 0)
  int conv_autotune;  // 1: time split-K candidates on the first call of a geometry (default 0: deterministic cost-model plan)
  int conv_max_split;  // 0 auto, 1 off, n cap
  int conv_xcd_swizzle;  // 1: XCD-aware tile order (default), 0: plain
  int dgrad_group;       // 1 (default): the four parity classes of a stride-2 data gradient share one launch; 0: class by class
  int wgrad_lds;         // 1 (default): LDS-staged weight-gradient kernel; 0: the round-2 register-fed kernel (A/B measurements)
  int wino_s2d_skip;     // 1 (default): stride-2 Winograd layers skip the positions whose weights are identically zero; 0: all 16 (A/B measurements)
  int wino_shared;       // 1 (default): Winograd layers with Cout % 64 == 0 on the 8-wave shared-transform kernel (conv_wino8_kernel); 0: the round-4 one-wave kernel
  int wino_wide;         // block shape of the shared-transform kernel: 1 (default) = per layer by the work per CU, 0 = 64 ch x 64 tiles, 3 = 128 x 32, 2 = 64 x 32 on four waves (two blocks per CU)
  int wino_split;        // K-split of the shared-transform kernel: 0 (default) = the plan of wino8_split_plan, 1 = never, n = at most n slices
  int conv_fewout_blocks, conv_fewout_minc;   // few-filter heads: channel slices so that the grid has about this many blocks (0 = default: 1024 where the pixels alone give >= 32 blocks, else 512), of at least this many channels each (32)
  int conv_fewout_quad;  // 1 (default): the 3x3 stride-1 heads with W % 4 == 0 on the four-pixels-per-lane kernel; 0: one pixel per lane
  int wino_streamk;      // 1 (default): where a grid leaves a partly filled last round, the persistent blocks share the work granule by granule (stream-K; needs wino_persistent, off with wino_split = 1); 2: wherever it applies, whatever the cost model says; 0: never
  std::map<uintptr_t, size_t> allocs;   // deepim_malloc's live allocations (base -> bytes): deepim_d2d's residency test without a driver query
  int wino_fin;          // 1: a K-split Winograd layer is finished inside the kernel by the block whose slice arrives last (no second pass); 0 (default): wino_reduce_kernel — the serial finish of the last slices costs more than the parallel second pass (profiles/r06_b4_share.md)
  std::mutex allocs_mu;  // guards `allocs`: a DeviceArray finalizer on another Python thread may free while the main thread copies (ctypes drops the GIL)
  void* wino_counters;   // arrival counters of the Winograd kernels' in-kernel finish, one per tile block (deepim_create; zero between launches)
  int wino_persistent;   // 1 (default): the shared-transform kernel's grid is one block per resident slot, each walking its share of the tiles; 0: one block per tile block
  int wino_two_wave;     // 0 (default): Winograd layers on the one-wave 16-position kernel; 1: the two-waves-per-SIMD kernel (measured slower on the big layers)
  int f16_dev_flags;     // dev: DI_F16_* bits — alternative tilings of the fp16 / x3 conv kernels (default 0)
  std::vector<const void*> attr_done;  // hipFuncSetAttribute groups already applied on THIS context's device (di_attr_needed)
};

// true the first time `tag` is seen on this context: guards one-off hipFuncSetAttribute calls (per device, hence per context)
static inline bool di_attr_needed(deepim_ctx* ctx, const void* tag) {
  for (const void* t : ctx->attr_done) if (t == tag) return false;
  ctx->attr_done.push_back(tag);
  return true;
}

#define DI_WINO_COUNTERS 16384   /* arrival counters per context (64 KB): tile blocks of one Winograd launch that can finish in-kernel */
void deepim_set_error(const char* where, hipError_t e);
void deepim_set_error_msg(const char* msg);

#define DI_CHECK(expr)                                   \
  do {                                                   \
    hipError_t _e = (expr);                              \
    if (_e != hipSuccess) {                              \
      deepim_set_error(#expr, _e);                       \
      return (int)_e;                                    \
    }                                                    \
  } while (0)

// Every extern "C" entry that takes a context starts with this: allocations, events and launches follow the CURRENT
// device, not the stream's, and the reference drives several GPUs from one process (gpu_flow_wrapper(device_id),
// one executor per context) — so each call makes its context's device current first.
#define DI_DEVICE(ctx)                                   \
  do {                                                   \
    hipError_t _e = hipSetDevice((ctx)->device);         \
    if (_e != hipSuccess) {                              \
      deepim_set_error("hipSetDevice", _e);              \
      return (int)_e;                                    \
    }                                                    \
  } while (0)

#define DI_LAUNCH_CHECK()                                \
  do {                                                   \
    hipError_t _e = hipGetLastError();                   \
    if (_e != hipSuccess) {                              \
      deepim_set_error("kernel launch", _e);             \
      return (int)_e;                                    \
    }                                                    \
  } while (0)

#define DI_REQUIRE(cond, msg)                            \
  do {                                                   \
    if (!(cond)) {                                       \
      deepim_set_error_msg(msg);                         \
      return -1;                                         \
    }                                                    \
  } while (0)

// fill pass of the box_rendered rectangle from accumulated bbox words (csrc/flow.hip)
int deepim_mask_box_fill(deepim_ctx* ctx, float* box, const int* words, int B, int H, int W);

void deepim_ctx_default_options(deepim_ctx* c);
// Grow-only scratch; never reallocated while a graph capture is open.
int deepim_scratch(deepim_ctx* ctx, size_t bytes, void** out);

static inline int di_div_up(long a, long b) { return (int)((a + b - 1) / b); }

// small by-value attribute blocks passed as kernel arguments (live in SGPRs)
struct Mat3 { float v[9]; };
struct Vec3 { float v[3]; };
