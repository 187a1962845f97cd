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

// one block (one wave) per sample: 7 dot products of length F, then inverse ZoomTrans
__global__ __launch_bounds__(64) void pose_head_kernel(float* __restrict__ se3, const float* __restrict__ feat,
                                                       const float* __restrict__ w_rot, const float* __restrict__ b_rot,
                                                       const float* __restrict__ w_trans,
                                                       const float* __restrict__ b_trans,
                                                       const float* __restrict__ zoom_factor, int F) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int k = lane; k < F; k += 64) {
    const float x = feat[(long)b * F + k];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = fmaf(x, w_rot[r * F + k], acc[r]);
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[4 + r] = fmaf(x, w_trans[r * F + k], acc[4 + r]);
  }
#pragma unroll
  for (int r = 0; r < 7; ++r)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off, 64);
  if (lane == 0) {
    const float wx = zoom_factor[b * 4 + 0];
    float* o = se3 + b * 7;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = acc[r] + b_rot[r];
    o[4] = (acc[4] + b_trans[0]) * wx;  // ZoomTrans b_inv_zoom=True (zoom_trans.py:34-37)
    o[5] = (acc[5] + b_trans[1]) * wx;
    o[6] = acc[6] + b_trans[2];
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
