// N11/N12: FullyConnected layers and the pose head.
//   fc6/fc7 (+LeakyReLU)            deepIM_flownet.py:112-116
//   rot / trans FCs + ZoomTrans(inv) + Concat → se3   deepIM_flownet.py:715-726, zoom_trans.py:22-46
//
// fc6 is a 256×81920 weight stream (84 MB fp32) against a handful of activation rows:
// HBM-bound on the weights in principle, L1-bound on the activation re-reads in practice
// (16 activation dwordx4 per lane per step against the weight loads). Split-K GEMV-style
// kernel: a block owns 32 output rows × one K slice; each lane streams dwordx4 of 8 weight
// rows and re-uses every activation dwordx4 it loads for those 8 rows (4 rows: 60 µs,
// 8 rows: 43 µs for fc6 at B = 16). Partials are reduced in a fixed order by a second tiny
// kernel (deterministic, no float atomics) that also applies bias + LeakyReLU.
#include "common.h"
#include "pose_head.h"

namespace {

constexpr int FC_BT = 16;   // batch rows per pass
constexpr int FC_RW = 8;    // output rows per wave: every activation dwordx4 (the L1-side bottleneck) feeds 8 rows
constexpr int FC_ROWS = 32; // output rows per block (4 waves)

// partial[s][b][o] = Σ_{k in slice s} x[b][k]·w[o][k]
__global__ __launch_bounds__(256) void fc_partial_kernel(float* __restrict__ partial, const float* __restrict__ x,
                                                         const float* __restrict__ w, int B, int I, int O, int slice) {
  const int b0 = blockIdx.z * FC_BT;   // batch rows in groups of 16: concurrent blocks, the second group's weight reads hit L2/MALL
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int o0 = blockIdx.x * FC_ROWS + wave * FC_RW;
  const int s = blockIdx.y;
  const int kbeg = s * slice, kend = min(I, kbeg + slice);
  float acc[FC_RW][FC_BT];
#pragma unroll
  for (int r = 0; r < FC_RW; ++r)
#pragma unroll
    for (int b = 0; b < FC_BT; ++b) acc[r][b] = 0.f;
  const int nb = min(FC_BT, B - b0);
  for (int k = kbeg + lane * 4; k < kend; k += 256) {
    float4 wv[FC_RW];
#pragma unroll
    for (int r = 0; r < FC_RW; ++r) {
      const int o = o0 + r;
      wv[r] = o < O ? *reinterpret_cast<const float4*>(w + (long)o * I + k) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < FC_BT; ++b) {
      if (b < nb) {
        const float4 xv = *reinterpret_cast<const float4*>(x + (long)(b0 + b) * I + k);
#pragma unroll
        for (int r = 0; r < FC_RW; ++r) {
          float a = acc[r][b];
          a = fmaf(xv.x, wv[r].x, a); a = fmaf(xv.y, wv[r].y, a);
          a = fmaf(xv.z, wv[r].z, a); a = fmaf(xv.w, wv[r].w, a);
          acc[r][b] = a;
        }
      }
    }
  }
  // Butterfly reduce-scatter over the wavefront, 4 rows (64 per-lane partial sums = 4 rows x 16 batch rows) at a time:
  // after 6 exchange steps lane l holds the wave total of value l. 63 shuffles instead of 64 x 6.
#pragma unroll
  for (int h = 0; h < FC_RW / 4; ++h) {
    float v[64];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int b = 0; b < FC_BT; ++b) v[r * FC_BT + b] = acc[h * 4 + r][b];
#pragma unroll
    for (int off = 32, n = 32; off > 0; off >>= 1, n >>= 1) {
      const bool hi = (lane & off) != 0;
#pragma unroll
      for (int i = 0; i < n; ++i) {
        const float keep = hi ? v[i + n] : v[i];
        const float send = hi ? v[i] : v[i + n];
        v[i] = keep + __shfl_xor(send, off, 64);
      }
    }
    const int r = h * 4 + (lane >> 4), b = lane & 15;
    if (o0 + r < O && b < nb) partial[((long)s * B + (b0 + b)) * O + o0 + r] = v[0];
  }
}

// out[i] = lrelu(Σ_s partial[s][i] + bias): 4 outputs per wave, 16 lanes each; lane j sums slices j, j+16, ... in order,
// then a fixed-shape butterfly adds the 16 lane sums — deterministic, and the S dependent loads no longer serialise
__global__ __launch_bounds__(256) void fc_finalize_kernel(float* __restrict__ out, const float* __restrict__ partial,
                                                          const float* __restrict__ bias, int B, int O, int S,
                                                          float slope) {
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int j = threadIdx.x & 15;
  float v = 0.f;
  if (i < B * O)
    for (int s = j; s < S; s += 16) v += partial[(long)s * B * O + i];
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
  if (i < B * O && j == 0) {
    v += bias ? bias[i % O] : 0.f;
    out[i] = v > 0.f ? v : v * slope;
  }
}

// ---------------------------------------------------------------------------------------------- fc6 on the matrix cores ----
// out[b][o] = Σ_k x[b][k]·w[o][k] as D[o][b] on v_mfma_f32_32x32x2_f32 (exact fp32), ONE pass over the weights for up to
// 32 batch rows.  The weights are re-packed once (deepim_fc_pack_weights) into the A-operand order of the instruction:
//   wp[step][o-tile][lane][j] = w[32·tile + lane%32][8·step + 4·(lane/32) + j]
// so one `dwordx4` per lane feeds four MFMA k-steps (k pairs (8s+j, 8s+4+j)) and a wave reads 1 KB contiguous per o-tile,
// 8 KB contiguous per step for fc6.  The activation operand uses the same k pairing: lanes 0-31 read x[b][8s..8s+3], lanes
// 32-63 x[b][8s+4..8s+7] — every x element is read by exactly one wave.  A wave owns a contiguous run of steps and all
// O/32 o-tiles (8 x 16 accumulator registers); the four waves of a block are added through LDS in a fixed tree, the block
// writes one partial tile, and fc_finalize_mfma_kernel adds the partials in a fixed order (+ bias + LeakyReLU):
// deterministic, no float atomics.  fc6 at B = 32: 84 MB of weights read once + 10.5 MB of activations + 8 MB of partials.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int FCM_BLOCKS = 256;   // one block per CU, one wave per SIMD: the 512-register budget buys a 4-step-deep prefetch
constexpr int FCM_DEPTH = 5;      // register sets; a wave keeps 4 steps (36 KB) of loads in flight behind the one it multiplies

// HBM latency under load is a few microseconds, a step of 32 MFMAs is 0.85 µs: with the next step only in flight the kernel
// ran at 3.3 TB/s (latency-bound), so the loop keeps FCM_DEPTH-1 steps in flight. The body is straight-line — every load
// is unconditional (step index clamped), steps past the end multiply x = 0 — and fenced with sched_barrier: a load or an
// MFMA group behind a branch makes the compiler sink the prefetch next to its use or drain vmcnt at the join.
template <int OT>
__global__ __launch_bounds__(256, 1) void fc_mfma_kernel(float* __restrict__ partial, const float* __restrict__ x,
                                                         const float* __restrict__ wp, int B, int b0, int I,
                                                         int steps_total, int steps_per_wave) {
  extern __shared__ float red[];   // 2 x OT x 16 x 64 floats
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave;
  int s0 = gw * steps_per_wave, s1 = min(steps_total, s0 + steps_per_wave);
  if (s0 >= steps_total) { s0 = steps_total - 1; s1 = s0; }   // no work: the loop below does not run
  const int brow = min(b0 + (lane & 31), B - 1);   // columns past the batch replay the last row; they are never stored
  const float* xp = x + (long)brow * I + (lane >> 5) * 4;
  const float* wl = wp + (long)lane * 4;
  f32x16 acc[OT];
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // explicitly named register sets, so that nothing is indexed dynamically
  float4 x0, x1, x2, x3, x4, w0[OT], w1[OT], w2[OT], w3[OT], w4[OT];
  const int last = max(s0, s1 - 1);
#define FCM_LOAD(X, W, ST)                                                                     \
  {                                                                                            \
    const int st_ = min((ST), last);                                                           \
    X = *reinterpret_cast<const float4*>(xp + (long)st_ * 8);                                  \
    _Pragma("unroll") for (int t = 0; t < OT; ++t)                                             \
        W[t] = *reinterpret_cast<const float4*>(wl + ((long)st_ * OT + t) * 256);              \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  }
#define FCM_MUL(X, W, ST)                                                                      \
  {                                                                                            \
    if ((ST) >= s1) X = make_float4(0.f, 0.f, 0.f, 0.f);                                       \
    _Pragma("unroll") for (int t = 0; t < OT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[t].x, X.x, acc[t], 0, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < OT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[t].y, X.y, acc[t], 0, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < OT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[t].z, X.z, acc[t], 0, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < OT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[t].w, X.w, acc[t], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  }
  FCM_LOAD(x0, w0, s0) FCM_LOAD(x1, w1, s0 + 1) FCM_LOAD(x2, w2, s0 + 2) FCM_LOAD(x3, w3, s0 + 3)
  for (int g = s0; g < s1; g += FCM_DEPTH) {
    FCM_LOAD(x4, w4, g + 4) FCM_MUL(x0, w0, g)
    FCM_LOAD(x0, w0, g + 5) FCM_MUL(x1, w1, g + 1)
    FCM_LOAD(x1, w1, g + 6) FCM_MUL(x2, w2, g + 2)
    FCM_LOAD(x2, w2, g + 7) FCM_MUL(x3, w3, g + 3)
    FCM_LOAD(x3, w3, g + 8) FCM_MUL(x4, w4, g + 4)
  }
#undef FCM_LOAD
#undef FCM_MUL
  // fixed tree over the four waves: (w0 + w2) + (w1 + w3)
  float* slot = red + (long)(wave & 1) * OT * 1024 + lane;
  if (wave >= 2) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) slot[(t * 16 + r) * 64] = acc[t][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] += slot[(t * 16 + r) * 64];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) slot[(t * 16 + r) * 64] = acc[t][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* o = partial + (long)blockIdx.x * OT * 1024 + lane;
    const float* other = red + (long)OT * 1024 + lane;
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(t * 16 + r) * 64] = acc[t][r] + other[(t * 16 + r) * 64];
  }
}

// out[b][o] = lrelu(Σ_s partial[s][tile][r][lane] + bias[o]); element i = (tile·16 + r)·64 + lane is row
// o = 32·tile + (r&3) + 8·(r>>2) + 4·(lane>>5) and batch column b0 + (lane&31) of the 32x32 MFMA tile.
// 32 lanes per element: lane j adds slices j, j+32, ... in order (8 independent loads for fc6), then a fixed butterfly.
__global__ __launch_bounds__(256) void fc_finalize_mfma_kernel(float* __restrict__ out, const float* __restrict__ partial,
                                                               const float* __restrict__ bias, int B, int b0, int O, int S,
                                                               float slope) {
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int j = threadIdx.x & 31;
  const int n = O * 32;
  float v = 0.f;
  if (i < n) {
#pragma unroll 8
    for (int s = j; s < S; s += 32) v += partial[(long)s * n + i];
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 32);
  if (i < n && j == 0) {
    const int lane = i & 63, r = (i >> 6) & 15, tile = i >> 10;
    const int o = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), b = b0 + (lane & 31);
    if (b < B) {
      v += bias ? bias[o] : 0.f;
      out[(long)b * O + o] = v > 0.f ? v : v * slope;
    }
  }
}

__global__ __launch_bounds__(256) void fc_pack_kernel(float* __restrict__ wp, const float* __restrict__ w, int O, int I,
                                                      long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int OT = O >> 5;
  const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
  const long q = i >> 8;
  const int tile = (int)(q % OT);
  const long step = q / OT;
  wp[i] = w[(long)(tile * 32 + (lane & 31)) * I + step * 8 + (lane >> 5) * 4 + j];
}

// one block (one wave) per sample: 7 dot products of length F, then inverse ZoomTrans
__global__ __launch_bounds__(64) void pose_head_kernel(float* __restrict__ se3, const float* __restrict__ feat,
                                                       const float* __restrict__ w_rot, const float* __restrict__ b_rot,
                                                       const float* __restrict__ w_trans,
                                                       const float* __restrict__ b_trans,
                                                       const float* __restrict__ zoom_factor, int F) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float o[7];
  di_pose_head_wave(o, feat + (long)b * F, w_rot, b_rot, w_trans, b_trans, zoom_factor[b * 4 + 0], F, lane);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < 7; ++r) se3[b * 7 + r] = o[r];
  }
}

}  // namespace

extern "C" int deepim_fc_forward(deepim_ctx* ctx, float* out, const float* in, const float* w, const float* bias,
                                 int B, int I, int O, float slope) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE((I & 3) == 0, "fc: input width must be a multiple of 4");
  const int rowblocks = di_div_up(O, FC_ROWS);
  // enough K slices to fill the chip (~512 blocks of 2 waves/SIMD), each a multiple of 256 elements
  int S = ctx->fc_slices > 0 ? ctx->fc_slices : di_div_up(512, rowblocks);   // measured best at fc6: 8 row blocks x 64 slices
  int slice = di_div_up(di_div_up(I, S), 256) * 256;
  S = di_div_up(I, slice);
  void* scratch;
  int rc = deepim_scratch(ctx, (size_t)S * B * O * sizeof(float), &scratch);
  if (rc) return rc;
  float* partial = (float*)scratch;
  hipLaunchKernelGGL(fc_partial_kernel, dim3(rowblocks, S, di_div_up(B, FC_BT)), dim3(256), 0, ctx->stream, partial, in, w,
                     B, I, O, slice);
  hipLaunchKernelGGL(fc_finalize_kernel, dim3(di_div_up((long)B * O, 16)), dim3(256), 0, ctx->stream, out, partial,
                     bias, B, O, S, slope);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t deepim_fc_packed_size(int O, int I) { return (size_t)O * I * sizeof(float); }

extern "C" int deepim_fc_pack_weights(deepim_ctx* ctx, float* packed_w, const float* w, int O, int I) {
  DI_DEVICE(ctx);
  DI_REQUIRE(O > 0 && (O & 31) == 0 && (I & 7) == 0, "fc_pack: needs O % 32 == 0 and I % 8 == 0");
  const long total = (long)O * I;
  hipLaunchKernelGGL(fc_pack_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, packed_w, w, O, I, total);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_fc_forward_packed(deepim_ctx* ctx, float* out, const float* in, const float* packed_w,
                                        const float* bias, int B, int I, int O, float slope) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(O == 256 && (I & 7) == 0, "fc_forward_packed: built for O == 256 (fc6/fc7), I % 8 == 0");
  constexpr int OT = 8;
  const int steps_total = I / 8;
  const int blocks = min(FCM_BLOCKS, di_div_up(steps_total, 4 * FCM_DEPTH));
  const int spw = FCM_DEPTH * di_div_up(steps_total, blocks * 4 * FCM_DEPTH);   // whole groups of FCM_DEPTH steps per wave
  const int S = di_div_up(steps_total, spw * 4);                                // blocks that own at least one step
  void* scratch;
  int rc = deepim_scratch(ctx, (size_t)S * OT * 1024 * sizeof(float), &scratch);
  if (rc) return rc;
  float* partial = (float*)scratch;
  const size_t lds = (size_t)2 * OT * 1024 * sizeof(float);
  static const char attr_set_tag = 0;   // function attributes are per DEVICE: remember them per context
  if (di_attr_needed(ctx, &attr_set_tag)) {
    DI_CHECK(hipFuncSetAttribute((const void*)fc_mfma_kernel<OT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  for (int b0 = 0; b0 < B; b0 += 32) {   // one weight pass per 32 batch rows
    hipLaunchKernelGGL(fc_mfma_kernel<OT>, dim3(S), dim3(256), lds, ctx->stream, partial, in, packed_w, B, b0, I, steps_total,
                       spw);
    hipLaunchKernelGGL(fc_finalize_mfma_kernel, dim3(di_div_up((long)O * 32, 8)), dim3(256), 0, ctx->stream, out, partial,
                       bias, B, b0, O, S, slope);
  }
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_pose_head_forward(deepim_ctx* ctx, float* se3, const float* feat, const float* w_rot,
                                        const float* b_rot, const float* w_trans, const float* b_trans,
                                        const float* zoom_factor, int B, int F) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  hipLaunchKernelGGL(pose_head_kernel, dim3(B), dim3(64), 0, ctx->stream, se3, feat, w_rot, b_rot, w_trans, b_trans,
                     zoom_factor, F);
  DI_LAUNCH_CHECK();
  return 0;
}
