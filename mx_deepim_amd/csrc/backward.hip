// Backward of the matching network's conv stack and FC head + the SGD step: what
// module.backward / the "sgd" optimizer do for these layers in the reference's training loop
// (deepim/core/module.py:1131-1137, deepim/train.py:295-338; layers wired at deepim/symbols/deepIM_flownet.py:63-116,211-215).
//
//   dgrad  dX = Σ dY·W          runs on the FORWARD MFMA conv kernel: weights transposed + flipped by
//                               conv_flip_weights_kernel (W'[ci][co][ky'][kx'] = W[co][ci][kh-1-ky'][kw-1-kx']), stride-2 layers
//                               on a zero-dilated dY (dilate2d_kernel), pad' = k-1-p — composed in
//                               mx_deepim_amd/symbols/deepIM_flownet.py:_conv_backward
//   wgrad  dW = Σ_pix dY·Xcol   wgrad_mfma_kernel: D[co][k] over v_mfma_f32_32x32x2_f32 with the PIXELS as the reduction
//                               dimension, split into slices across the grid, partial sums added in slice order
//                               (deterministic, no float atomics)
//   bias   db = Σ dY            one block per channel, fixed-shape tree
//   LeakyReLU'                  from the saved output: dZ = dY·(y > 0 ? 1 : slope)
//   FC     dX = dY·W, dW = dYᵀ·X, db = Σ dY
//   SGD    mom = m·mom − lr·(rescale·g [clipped] + wd·w); w += mom     (MXNet sgd_mom_update)
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// called in place (dz == dy) by the training graph: no __restrict__ on that pair
__global__ __launch_bounds__(256) void lrelu_backward_kernel(float* dz, const float* dy, const float* __restrict__ y,
                                                             float slope, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dz[i] = y[i] > 0.f ? dy[i] : dy[i] * slope;
}

// db[c] = Σ_n Σ_p dz[n][c][p] in two deterministic passes: block (c, s) sums slice s of every sample's plane (thread t takes
// elements t, t+256, … of the slice in order, float64, fixed LDS tree) into partial[c][s]; the second pass adds the S slice
// sums of a channel in order. One block per channel (round 2) left 64-128 blocks walking 300 k elements each on the first
// layers: 0.32 ms for conv1's gradient alone.
__global__ __launch_bounds__(256) void bias_grad_kernel(double* __restrict__ partial, const float* __restrict__ dz, int B, int C,
                                                        long HW, int S, long per_slice, float* __restrict__ db) {
  const int c = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
  const long lo = (long)sl * per_slice, hi = min(HW, lo + per_slice);
  double acc = 0.0;
  for (int n = 0; n < B; ++n) {
    const float* p = dz + ((long)n * C + c) * HW;
    for (long i = lo + tid; i < hi; i += 256) acc += (double)p[i];
  }
  __shared__ double red[256];
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) {
    if (S == 1) db[c] = (float)red[0];     // a single slice is the sum itself: no second pass
    else partial[(long)c * S + sl] = red[0];
  }
}

// The same walk with the LeakyReLU gradient folded in: dz = lrelu'(y)·(dy [+ add]) is written AND summed by block (c, slice) — one
// read of dy / y instead of three passes (skip-gradient add, activation gradient, bias-gradient first pass). V4: float4 steps.
template <bool V4>
__global__ __launch_bounds__(256) void lrelu_bias_backward_kernel(double* __restrict__ partial, float* dz, const float* dy,
                                                                  const float* add, const float* __restrict__ y, float slope,
                                                                  int B, int C, long HW, int S, long per_slice, float* __restrict__ db) {
  const int c = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
  const long lo = (long)sl * per_slice, hi = min(HW, lo + per_slice);
  double acc = 0.0;
  for (int n = 0; n < B; ++n) {
    const long base = ((long)n * C + c) * HW;
    if (V4) {
      for (long i = lo + 4 * tid; i < hi; i += 1024) {
        float4 g = *reinterpret_cast<const float4*>(dy + base + i);
        const float4 yy = *reinterpret_cast<const float4*>(y + base + i);
        if (add) {
          const float4 a = *reinterpret_cast<const float4*>(add + base + i);
          g.x += a.x; g.y += a.y; g.z += a.z; g.w += a.w;
        }
        g.x = yy.x > 0.f ? g.x : g.x * slope; g.y = yy.y > 0.f ? g.y : g.y * slope;
        g.z = yy.z > 0.f ? g.z : g.z * slope; g.w = yy.w > 0.f ? g.w : g.w * slope;
        *reinterpret_cast<float4*>(dz + base + i) = g;
        acc += (double)g.x; acc += (double)g.y; acc += (double)g.z; acc += (double)g.w;
      }
    } else {
      for (long i = lo + tid; i < hi; i += 256) {
        float g = dy[base + i];
        if (add) g += add[base + i];
        g = y[base + i] > 0.f ? g : g * slope;
        dz[base + i] = g;
        acc += (double)g;
      }
    }
  }
  __shared__ double red[256];
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) {
    if (S == 1) db[c] = (float)red[0];     // a single slice is the sum itself: no second pass
    else partial[(long)c * S + sl] = red[0];
  }
}

// Backward of [Concat slice → LeakyReLU → Crop] in front of a transposed convolution, in one walk: channels [coff, coff + C) of
// the concat gradient (B,ctotal,ho,wo), times lrelu'(y) of the same slice of the saved concat output (y NULL: no activation),
// placed at (off_y, off_x) of the un-cropped frame (B,C,hf,wf) with zeros around, and summed per channel (the bias gradient).
// Block = one channel. Replaces two slice copies, the activation gradient, both bias-gradient passes and the scatter.
__global__ __launch_bounds__(256) void slice_lrelu_bias_scatter_kernel(float* __restrict__ out, float* __restrict__ db,
                                                                       const float* __restrict__ dcat, const float* __restrict__ ycat,
                                                                       int B, int ctotal, int coff, int C, int ho, int wo, int hf,
                                                                       int wf, int off_y, int off_x, float slope) {
  const int c = blockIdx.x, tid = threadIdx.x;
  double acc = 0.0;
  for (int n = 0; n < B; ++n) {
    const long src = ((long)n * ctotal + coff + c) * ho * wo;
    float* o = out + ((long)n * C + c) * hf * wf;
    for (int i = tid; i < hf * wf; i += 256) {
      const int y = i / wf - off_y, x = i % wf - off_x;
      float g = 0.f;
      if ((unsigned)y < (unsigned)ho && (unsigned)x < (unsigned)wo) {
        g = dcat[src + y * wo + x];
        if (ycat) g = ycat[src + y * wo + x] > 0.f ? g : g * slope;
        acc += (double)g;
      }
      o[i] = g;
    }
  }
  __shared__ double red[256];
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0 && db) db[c] = (float)red[0];
}

__global__ __launch_bounds__(256) void bias_grad_final_kernel(float* __restrict__ db, const double* __restrict__ partial, int C,
                                                              int S) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double acc = 0.0;
  for (int s = 0; s < S; ++s) acc += partial[(long)c * S + s];
  db[c] = (float)acc;
}

__global__ __launch_bounds__(256) void conv_flip_weights_kernel(float* __restrict__ wt, const float* __restrict__ w, int Cout,
                                                                int Cin, int kh, int kw, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int kx = (int)(i % kw);
  const int ky = (int)((i / kw) % kh);
  const int co = (int)((i / ((long)kw * kh)) % Cout);
  const int ci = (int)(i / ((long)kw * kh * Cout));
  wt[i] = w[(((long)co * Cin + ci) * kh + (kh - 1 - ky)) * kw + (kw - 1 - kx)];
}

// Data gradient of a STRIDE-2 convolution without the zero-dilated gradient: the output pixels of one parity class
// (y % 2, x % 2) = (py, px) only ever meet the taps ky = ky0 + 2a, kx = kx0 + 2b (ky0 = (py + pad) % 2, …), so the class is a
// stride-1 convolution of the un-dilated dz with this sub-kernel, transposed and flipped:
//   wt (Cin,Cout,nky,nkx)[ci][co][a'][b'] = w (Cout,Cin,kh,kw)[co][ci][ky0 + 2(nky-1-a')][kx0 + 2(nkx-1-b')]
__global__ __launch_bounds__(256) void conv_subkernel_flip_kernel(float* __restrict__ wt, const float* __restrict__ w, int Cout,
                                                                  int Cin, int kh, int kw, int ky0, int kx0, int nky, int nkx,
                                                                  long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i % nkx);
  const int a = (int)((i / nkx) % nky);
  const int co = (int)((i / ((long)nkx * nky)) % Cout);
  const int ci = (int)(i / ((long)nkx * nky * Cout));
  wt[i] = w[(((long)co * Cin + ci) * kh + ky0 + 2 * (nky - 1 - a)) * kw + kx0 + 2 * (nkx - 1 - b)];
}

// dx (BC,H,W)[bc][2i + py][2j + px] = src (BC,Hs,Ws)[bc][i + cy][j + cx] for i < hq, j < wq: the window of a class result put
// on its parity positions (the four classes together write every element of dx exactly once)
__global__ __launch_bounds__(256) void interleave2d_kernel(float* __restrict__ dx, const float* __restrict__ src, int Hs, int Ws,
                                                           int cy, int cx, int H, int W, int py, int px, int hq, int wq,
                                                           long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i % wq);
  const int r = (int)((i / wq) % hq);
  const long bc = i / ((long)wq * hq);
  dx[(bc * H + 2 * r + py) * W + 2 * j + px] = src[(bc * Hs + r + cy) * Ws + j + cx];
}

// out (BC,Hd,Wd) = zeros except out[bc][off_y + stride*y][off_x + stride*x] = in[bc][y][x]  (dilation and/or un-crop)
__global__ __launch_bounds__(256) void dilate2d_kernel(float* __restrict__ out, const float* __restrict__ in, int Ho, int Wo,
                                                       int Hd, int Wd, int stride, int off_y, int off_x, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int xd = (int)(i % Wd) - off_x;
  const int yd = (int)((i / Wd) % Hd) - off_y;
  const long bc = i / ((long)Wd * Hd);
  float v = 0.f;
  if (yd >= 0 && xd >= 0 && yd % stride == 0 && xd % stride == 0) {
    const int y = yd / stride, x = xd / stride;
    if (y < Ho && x < Wo) v = in[(bc * Ho + y) * Wo + x];
  }
  out[i] = v;
}

// data gradient of the depthwise k32 s16 transposed convolution + crop (fixed bilinear weights, deepIM_flownet.py:185-200,
// 326-340): d_in[bc][iy][ix] = scale * sum_{ky,kx} dy[bc][16iy+ky-cy][16ix+kx-cx] * w[c][ky][kx]; one 64-lane group per output
// element, lane = (ky parity rows...) sums 16 taps, fixed-shape butterfly
__global__ __launch_bounds__(256) void upsample16_backward_kernel(float* __restrict__ d_in, const float* __restrict__ dy,
                                                                  const float* __restrict__ w, int C, int H, int W, int Ho,
                                                                  int Wo, int crop_y, int crop_x, float scale, long total) {
  const long e = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wavefront per low-resolution element
  const int lane = threadIdx.x & 63;
  if (e >= total) return;
  const int ix = (int)(e % W), iy = (int)((e / W) % H);
  const long bc = e / ((long)W * H);
  const float* g = dy + bc * (long)Ho * Wo;
  const float* wp = w + (bc % C) * 1024;
  float acc = 0.f;
  for (int t = lane; t < 1024; t += 64) {                      // lane takes taps t, t+64, ... in order
    const int ky = t >> 5, kx = t & 31;
    const int y = 16 * iy + ky - crop_y, x = 16 * ix + kx - crop_x;
    if (y >= 0 && y < Ho && x >= 0 && x < Wo) acc = fmaf(g[(long)y * Wo + x], wp[t], acc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) d_in[e] = acc * scale;
}

// ------------------------------------------------------------------------------------------------------- wgrad ----
// partial[slice][co][k] = Σ_{pixels of the slice} dZ[co][pix]·X[ci(k)][pix shifted by the tap of k],  k = (ci,ky,kx).
// 128 (co) x 128 (k) block tile, four waves of 64x64 (2x2 MFMA 32x32x2 tiles). The reduction runs over the pixels of one
// sample at a time in groups of 8: a lane loads 4 consecutive pixels of its dZ row as one dwordx4 (lanes 0-31 pixels 0-3,
// lanes 32-63 pixels 4-7 → MFMA k-pairs (j, 4+j)) and gathers the matching 4 input pixels of its k (tap offset folded into
// a per-lane base, zero padding by predication). No LDS: the four waves share lines through L1.
struct WgradParams {
  const float* x;    // (B,Cin,H,W)
  const float* dz;   // (B,Cout,Ho,Wo)
  float* partial;    // [S][Cout][K]
  int B, Cin, H, W, Cout, Ho, Wo, kh, kw, stride, pad, K;
  int ktiles, mtiles, S, groups_per_slice;   // groups of 8 pixels per sample: ceil(HW/8); slices cover (n, group) pairs
};

__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradParams p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kt = blockIdx.x % p.ktiles, mt = (blockIdx.x / p.ktiles) % p.mtiles, sl = blockIdx.x / (p.ktiles * p.mtiles);
  const int lcol = lane & 31, lrow = lane >> 5;
  const int co_base = mt * 128 + (wave >> 1) * 64, k_base = kt * 128 + (wave & 1) * 64;
  const int HW = p.Ho * p.Wo, gps = (HW + 7) >> 3;
  // this lane's two dZ rows and two k columns
  int co[2], kk[2], kci[2], kdy[2], kdx[2];
  bool co_ok[2], k_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    co[i] = co_base + i * 32 + lcol;
    co_ok[i] = co[i] < p.Cout;
    kk[i] = k_base + i * 32 + lcol;
    k_ok[i] = kk[i] < p.K;
    const int k = k_ok[i] ? kk[i] : 0;
    kci[i] = k / (p.kh * p.kw);
    const int t = k - kci[i] * (p.kh * p.kw);
    kdy[i] = t / p.kw - p.pad;
    kdx[i] = t % p.kw - p.pad;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const long g_begin = (long)sl * p.groups_per_slice, g_end = min((long)p.B * gps, g_begin + p.groups_per_slice);
  for (long g = g_begin; g < g_end; ++g) {
    const int n = (int)(g / gps);
    const int p0 = (int)(g - (long)n * gps) * 8 + lrow * 4;   // this lane's 4 pixels: p0 .. p0+3
    float a[2][4], b[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* row = p.dz + ((long)n * p.Cout + (co_ok[i] ? co[i] : 0)) * HW;
      if (co_ok[i] && p0 + 3 < HW) {
        const float4 v = *reinterpret_cast<const float4*>(row + p0);   // HW % 4 == 0 is required (checked on the host)
        a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i][j] = (co_ok[i] && p0 + j < HW) ? row[p0 + j] : 0.f;
      }
    }
    int ho = p0 / p.Wo, wo = p0 - ho * p.Wo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool pin = p0 + j < HW;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int hi = ho * p.stride + kdy[i], wi = wo * p.stride + kdx[i];
        const bool ok = pin && k_ok[i] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
        b[i][j] = ok ? p.x[(((long)n * p.Cin + kci[i]) * p.H + hi) * p.W + wi] : 0.f;
      }
      if (++wo == p.Wo) { wo = 0; ++ho; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[q][j], acc[i][q], 0, 0, 0);
  }
  float* out = p.partial + (long)sl * p.Cout * p.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = k_base + q * 32 + lcol;
      if (k >= p.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = co_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
        if (c < p.Cout) out[(long)c * p.K + k] = acc[i][q][r];
      }
    }
}

// The same GEMM with both operands staged through LDS (round 3). In the kernel above a lane owns a dZ row / an im2col column
// and walks along the pixels, so every lane of a load touches a different cache line (64 lines per instruction): it runs at the
// L1 / TA line rate, 45-50 TFLOP/s. Here a 16-pixel chunk of the two operand tiles is loaded with the pixels fastest across the
// lanes — dZ: one dwordx4 per lane, 16 rows x 64 B per wave-instruction; im2col: one dword per lane, 4 columns x 16 consecutive
// output pixels — and kept in LDS ROW-major ([row][16 pixels], pitch 20 floats: 16-byte aligned, and 16 lanes of a
// ds_read_b128 cover the 64 banks once). A lane's MFMA k index is the pixel: lanes 0-31 take pixels 0-7 of the chunk, lanes
// 32-63 pixels 8-15, k-step j multiplies pixels (j, 8 + j) — so one fragment is 8 consecutive floats of one row = two
// ds_read_b128, all 8 reads of a chunk are issued up front and the 32 MFMAs run back to back; dZ goes global → LDS as whole
// dwordx4. Zero padding and ragged edges through the raw-buffer out-of-range offset; the global loads of chunk c+1 fly while
// chunk c is multiplied. BM = 64 (conv1: Cout 64): 32x64 wave tiles instead of a half-empty 128-row tile.
constexpr int WG_PIX = 16, WG_P = 20;
#ifndef WG_INTERLEAVE
#define WG_INTERLEAVE 1   // 1: one iteration is ONE scheduling region and the memory operations are spread over the MFMA slots
#endif
// TAPM ("tap-major", Cin % 8 == 0): the K rows run (tap, channel) instead of (channel, tap), and a thread's eight im2col rows
// are ONE tap of eight consecutive channels — one range check and one address per chunk instead of eight (the plane stride goes
// into the loads' scalar offset), and sixteen fewer registers. An ablation with the per-row range arithmetic removed ran 9 %
// faster (19 % on conv1): that arithmetic, not memory, was what the waves were short of. The gradient then comes out as
// dw_tm[co][tap][ci]; whoever reads it (the SGD kernel, deepim_weight_grad_to_natural) applies the permutation.
template <int BM, bool TAPM = false>
__global__ __launch_bounds__(256, 3) void wgrad_lds_kernel(WgradParams p) {   // 41 KB of LDS per block: three blocks per CU
  constexpr int TM = BM / 64, NA = BM / 64;          // MFMA row tiles per wave; dZ dwordx4 loads per thread and chunk
  __shared__ __attribute__((aligned(16))) float As[2][BM * WG_P];
  __shared__ __attribute__((aligned(16))) float Bs[2][128 * WG_P];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kt = blockIdx.x % p.ktiles, mt = (blockIdx.x / p.ktiles) % p.mtiles, sl = blockIdx.x / (p.ktiles * p.mtiles);
  const int lcol = lane & 31, lrow = lane >> 5;
  const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * 64;
  const int HW = p.Ho * p.Wo, cps = (HW + WG_PIX - 1) / WG_PIX;
  const unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, (int)((long)p.B * p.Cout * HW * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((long)p.B * p.Cin * p.H * p.W * 4), 0x00020000);
  // dZ loader: rows rowA (+ 64) of the tile, pixels 4 qA … 4 qA + 3 of the chunk
  const int rowA = tid >> 2, qA = tid & 3;
  unsigned a_off[NA];       // byte offset of the row inside a sample; bit 31 = row past Cout (→ out of range → 0.0)
#pragma unroll
  for (int r = 0; r < NA; ++r) {
    const int co = mt * BM + rowA + 64 * r;
    a_off[r] = co < p.Cout ? (unsigned)(co * HW * 4) : OOB;
  }
  // im2col loader: pixel pixB of the chunk, columns krow0 + 16 e of the tile
  const int pixB = tid & 15, krow0 = tid >> 4;
  int b_off[8];             // byte offset of tap (ci, ky - pad, kx - pad) from the pixel's (hi0, wi0) inside a sample (may be < 0)
  int b_tap[8];             // (ky - pad) << 16 | (kx - pad) & 0xffff; a column past K gets a row far outside the image
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = kt * 128 + krow0 + 16 * e;
    const int kc = k < p.K ? k : 0;
    const int ci = kc / (p.kh * p.kw), t = kc - ci * (p.kh * p.kw);
    const int ty = k < p.K ? t / p.kw - p.pad : 0x4000, tx = t % p.kw - p.pad;
    b_off[e] = k < p.K ? (ci * p.H * p.W + ty * p.W + tx) * 4 : 0;
    b_tap[e] = (ty << 16) | (tx & 0xffff);
  }
  const int Hm1 = p.H - 1, Wm1 = p.W - 1;
  // TAPM: rows 8 krow0 … 8 krow0 + 7 of the tile = tap tm_tap, channels tm_ci0 … tm_ci0 + 7
  int tm_ty = 0x4000, tm_tx = 0, tm_off = 0;
  if (TAPM) {
    const int k0 = kt * 128 + 8 * krow0;
    if (k0 < p.K) {
      const int t = k0 / p.Cin, ci0 = k0 - t * p.Cin;
      tm_ty = t / p.kw - p.pad; tm_tx = t % p.kw - p.pad;
      tm_off = (ci0 * p.H * p.W + tm_ty * p.W + tm_tx) * 4;
    }
  }
  const int plane_bytes = p.H * p.W * 4;
  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  typedef float f32x4 __attribute__((ext_vector_type(4)));
  // two register sets: the loads of a chunk are issued TWO chunks before its LDS store (one chunk of loads in flight, consumed
  // at the end of the same chunk, left the waves waiting for memory behind their MFMAs)
  f32x4 raA[NA], raB[NA];
  float rbA[8], rbB[8];
  const int chunks = p.B * cps;                      // < 2^31 (checked on the host)
  const int c_begin = sl * p.groups_per_slice, c_end = min(chunks, c_begin + p.groups_per_slice);
  // every byte offset first, then the loads back to back: left to itself the compiler interleaves address arithmetic with
  // the loads, reuses a load's destination register for the next address and waits (vmcnt(0)) in between, and turns
  // `valid ? offset : OOB` into branches — the selects below are plain bit operations (bit 31 set = out of range → 0.0)
  // (chunks at or past c_end load zeros: the loop below is branch-free — with loads under `if` the compiler cannot count what is in
  // flight and drains vmcnt(0) at every LDS store, i.e. the prefetch depth collapses to one chunk)
  // Validity is integer arithmetic on the sign bit, never a compare: `x >= y ? … : …` becomes v_cmp → s_or_b64 → v_cndmask, and a
  // VALU operation that reads an SGPR written by the scalar unit inside the MFMA shadow stalls the SIMD's issue (the effect
  // measured on the forward kernel, tools/mfma_issue_probe.hip). v is outside [0, n) iff (v | (n - 1 - v)) is negative.
  auto load_regs = [&](int c, f32x4 (&ra)[NA], float (&rb)[8]) {
    const int deadm = -(int)(c >= c_end);             // scalar: all ones for a chunk past the slice
    c = min(c, chunks - 1);
    const int n = c / cps;                            // 32-bit: a 64-bit division is a software loop inside the pipeline
    const int pc = (c - n * cps) * WG_PIX;
    const int pa = pc + qA * 4;                       // HW % 4 == 0: the quad is entirely inside or entirely outside the sample
    unsigned offa[NA], offb[8];
    const int basea = (n * p.Cout * HW + pa) * 4;
    const unsigned inva = (unsigned)(deadm | (HW - 1 - pa));
#pragma unroll
    for (int r = 0; r < NA; ++r)
      offa[r] = ((unsigned)(basea + (int)(a_off[r] & 0x7fffffffu)) & 0x7fffffffu) | ((inva | a_off[r]) & OOB);
    const int pb = pc + pixB;
    const int ho = pb / p.Wo, wo = pb - ho * p.Wo;
    const int hi0 = ho * p.stride, wi0 = wo * p.stride;
    const int baseb = (n * p.Cin * p.H * p.W + hi0 * p.W + wi0) * 4;
    const int pinv = deadm | (HW - 1 - pb);
    if (TAPM) {
      const int hi = hi0 + tm_ty, wi = wi0 + tm_tx;
      const int bad = pinv | hi | (Hm1 - hi) | wi | (Wm1 - wi);
      offb[0] = ((unsigned)(baseb + tm_off) & 0x7fffffffu) | ((unsigned)bad & OOB);
    }
#pragma unroll
    for (int e = 0; e < (TAPM ? 0 : 8); ++e) {
      const int hi = hi0 + (b_tap[e] >> 16), wi = wi0 + (int)(short)(b_tap[e] & 0xffff);
#ifdef WG_ABL   // dev ablation (wrong results at the borders): what the per-tap range arithmetic costs
      const int bad = pinv;
      (void)hi; (void)wi;
#else
      const int bad = pinv | hi | (Hm1 - hi) | wi | (Wm1 - wi);
#endif
      offb[e] = ((unsigned)(baseb + b_off[e]) & 0x7fffffffu) | ((unsigned)bad & OOB);
    }
#if !WG_INTERLEAVE
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int r = 0; r < NA; ++r) ra[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dz, (int)offa[r], 0, 0));
#pragma unroll
    for (int e = 0; e < 8; ++e)   // TAPM: the channel plane rides in the scalar offset (not range-checked: bit 31 of the vector offset decides)
      rb[e] = __builtin_bit_cast(float, TAPM ? __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)offb[0], e * plane_bytes, 0)
                                             : __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)offb[e], 0, 0));
#if !WG_INTERLEAVE
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  auto store_regs = [&](int buf, const f32x4 (&ra)[NA], const float (&rb)[8]) {
#pragma unroll
    for (int r = 0; r < NA; ++r) *reinterpret_cast<f32x4*>(&As[buf][(rowA + 64 * r) * WG_P + qA * 4]) = ra[r];
#pragma unroll
    for (int e = 0; e < 8; ++e) Bs[buf][(TAPM ? 8 * krow0 + e : krow0 + 16 * e) * WG_P + pixB] = rb[e];
  };

  auto compute = [&](int buf) {
    f32x4 af[TM][2], bf[2][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int v = 0; v < 2; ++v)
        af[i][v] = *reinterpret_cast<const f32x4*>(&As[buf][(wm0 + i * 32 + lcol) * WG_P + lrow * 8 + v * 4]);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int v = 0; v < 2; ++v)
        bf[q][v] = *reinterpret_cast<const f32x4*>(&Bs[buf][(wn0 + q * 32 + lcol) * WG_P + lrow * 8 + v * 4]);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][j >> 2][j & 3], bf[q][j >> 2][j & 3], acc[i][q], 0, 0, 0);
  };
  // iteration c: the set holds chunk c+1 (loaded during iterations c-2 … c-1) → LDS[buf^1]; its registers then take chunk c+3;
  // chunk c is multiplied out of LDS[buf]
  auto iter = [&](int c, int buf, f32x4 (&ra)[NA], float (&rb)[8]) {
    store_regs(buf ^ 1, ra, rb);
    load_regs(c + 3, ra, rb);
    compute(buf);
#if WG_INTERLEAVE
    // issue order of the region (hipcc alone: all stores, all address arithmetic, all loads, then reads → wait → 16 MFMAs twice,
    // so a wave's own memory operations never overlap its own MFMAs): the first fragments, then one MFMA per slot with the
    // address arithmetic and ONE memory operation behind it — the LDS stores of chunk c+1, the second-half fragments, the ten
    // global loads of chunk c+3
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + 2), 0);          // DS reads: first-half fragments
#pragma unroll
    for (int k = 0; k < 8 * TM * 2; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                   // a few VALU (byte offsets)
      if (k < NA + 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write (NA b128 + 4 write2)
      else if (k < NA + 4 + (TM + 2)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read: second-half fragments
      else if (k < NA + 4 + (TM + 2) + NA + 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
    }
#endif
    __syncthreads();
  };
  load_regs(c_begin, raA, rbA);
  store_regs(0, raA, rbA);
  load_regs(c_begin + 1, raA, rbA);
  load_regs(c_begin + 2, raB, rbB);
  __syncthreads();
  for (int c = c_begin; c < c_end; c += 2) {   // an odd count runs one phantom chunk of zeros
    iter(c, 0, raA, rbA);
    iter(c + 1, 1, raB, rbB);
  }
  float* out = p.partial + (long)sl * p.Cout * p.K;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = kt * 128 + wn0 + q * 32 + lcol;
      if (k >= p.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = mt * BM + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
        if (c < p.Cout) out[(long)c * p.K + k] = acc[i][q][r];
      }
    }
}

// Weight gradient of a layer with very few filters (the 3x3 prediction heads, Cout 1-2; the 2 → 2 k4 flow upsamplers): per input
// channel one dot product over the pixels for every (filter, tap) — a stream over x, not an MFMA problem (on the 64-row MFMA
// tile 97 % of the rows were padding: 59 µs + a reduce pass for Convolution3). Block = one input channel; thread t takes pixels
// t, t + 256, … of every sample in order, then a fixed shuffle + LDS tree: deterministic.
template <int CO, int KS>
__global__ __launch_bounds__(256) void wgrad_fewout_kernel(float* __restrict__ dw, const float* __restrict__ x,
                                                           const float* __restrict__ dz, int B, int Cin, int H, int W, int Cout,
                                                           int Ho, int Wo, int stride, int pad, float* __restrict__ db) {
  constexpr int T = KS * KS;
  const int ci = blockIdx.x, tid = threadIdx.x;
  const bool with_bias = db != nullptr && ci == 0;   // block 0 also sums dz per filter (the bias gradient): it reads dz anyway
  float accb[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) accb[co] = 0.f;
  float acc[CO][T];
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int t = 0; t < T; ++t) acc[co][t] = 0.f;
  const int HW = Ho * Wo;
  for (int n = 0; n < B; ++n) {
    const float* xp = x + ((long)n * Cin + ci) * H * W;
    const float* zp = dz + (long)n * Cout * HW;
    for (int r = tid; r < HW; r += 256) {
      const int ho = r / Wo, wo = r - ho * Wo;
      float g[CO];
#pragma unroll
      for (int co = 0; co < CO; ++co) g[co] = co < Cout ? zp[(long)co * HW + r] : 0.f;
#pragma unroll
      for (int co = 0; co < CO; ++co) accb[co] += g[co];
      const int hi0 = ho * stride - pad, wi0 = wo * stride - pad;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int hi = hi0 + ky, wi = wi0 + kx;
          const float v = ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) ? xp[hi * W + wi] : 0.f;
#pragma unroll
          for (int co = 0; co < CO; ++co) acc[co][ky * KS + kx] = fmaf(g[co], v, acc[co][ky * KS + kx]);
        }
    }
  }
  __shared__ float sh[4][CO * T];
  __shared__ float shb[4][CO];
  const int lane = tid & 63, wave = tid >> 6;
  if (with_bias) {
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      float v = accb[co];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) shb[wave][co] = v;
    }
  }
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float v = acc[co][t];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) sh[wave][co * T + t] = v;
    }
  __syncthreads();
  if (tid < CO * T) {
    const int co = tid / T, t = tid - co * T;
    if (co < Cout) dw[((long)co * Cin + ci) * T + t] = ((sh[0][tid] + sh[1][tid]) + sh[2][tid]) + sh[3][tid];
  }
  if (with_bias && tid < Cout) db[tid] = ((shb[0][tid] + shb[1][tid]) + shb[2][tid]) + shb[3][tid];
}

// dw[i] = Σ_s partial[s][i], slices in order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(float* __restrict__ dw, const float* __restrict__ partial, long n,
                                                           int S) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = partial[i];
  for (int s = 1; s < S; ++s) v += partial[(long)s * n + i];
  dw[i] = v;
}

// Many slices of a small dw (conv1: 25 k weights x 256 slices): one thread per element walks S dependent, 100 KB-strided loads
// (60 µs). Here a block = 64 float4 columns x 4 slice quarters; each quarter keeps four independent partial sums (slices s, s+1,
// s+2, s+3 of its range), the quarters are added in order through LDS. Fixed order, so still deterministic.
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(float* __restrict__ dw, const float* __restrict__ partial, long n4,
                                                            int S) {
  const int tid = threadIdx.x, col = tid & 63, q = tid >> 6;
  const long i = (long)blockIdx.x * 64 + col;
  const int Sq = (S + 3) >> 2, s0 = q * Sq, s1 = min(S, s0 + Sq);
  float4 a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* p = reinterpret_cast<const float4*>(partial) + i;
    int s = s0;
    for (; s + 4 <= s1; s += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = p[(long)(s + u) * n4];
        a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
      }
    }
    for (; s < s1; ++s) {
      const float4 v = p[(long)s * n4];
      a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
    }
  }
  float4 r = make_float4((a[0].x + a[1].x) + (a[2].x + a[3].x), (a[0].y + a[1].y) + (a[2].y + a[3].y),
                         (a[0].z + a[1].z) + (a[2].z + a[3].z), (a[0].w + a[1].w) + (a[2].w + a[3].w));
  __shared__ float4 sh[4][64];
  sh[q][col] = r;
  __syncthreads();
  if (q == 0 && i < n4) {
#pragma unroll
    for (int u = 1; u < 4; ++u) { const float4 v = sh[u][col]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
    reinterpret_cast<float4*>(dw)[i] = r;
  }
}

// ---------------------------------------------------------------------------------------------------------- FC ----
// dW[o][i] = Σ_b dy[b][o]·x[b][i]: thread = (o, 4 consecutive i)
__global__ __launch_bounds__(256) void fc_wgrad_kernel(float* __restrict__ dw, const float* __restrict__ dy,
                                                       const float* __restrict__ x, int B, int I, int O) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  const int i4 = (int)(q % (I / 4));
  const int o = (int)(q / (I / 4));
  if (o >= O) return;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int b = 0; b < B; ++b) {
    const float g = dy[(long)b * O + o];
    const float4 v = *reinterpret_cast<const float4*>(x + (long)b * I + i4 * 4);
    acc.x = fmaf(g, v.x, acc.x); acc.y = fmaf(g, v.y, acc.y); acc.z = fmaf(g, v.z, acc.z); acc.w = fmaf(g, v.w, acc.w);
  }
  *reinterpret_cast<float4*>(dw + (long)o * I + i4 * 4) = acc;
}
// dX[b][i] = Σ_o dy[b][o]·w[o][i] for a tile of 8 batch rows (w is read once per 8 rows). A block = 64 column quads x 4
// output-row quarters: wave q walks rows [q·O/4, (q+1)·O/4) with its 64 lanes on 1 KB of consecutive columns, four rows in
// flight; the four partial sums are added in wave order through LDS (deterministic). (Round 2: one thread per column quad
// over ALL rows and 256 quads per block — 80 blocks for fc6's 84 MB of weights, one for fc7: 122 and 111 µs.)
__global__ __launch_bounds__(256) void fc_dgrad_kernel(float* __restrict__ dx, const float* __restrict__ dy,
                                                       const float* __restrict__ w, int B, int I, int O) {
  __shared__ float4 part[3][8][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i4 = blockIdx.x * 64 + lane;
  const int b0 = blockIdx.y * 8;
  const bool live = i4 < I / 4;
  const int oq = (O + 3) / 4, o_lo = min(O, q * oq), o_hi = min(O, o_lo + oq);
  float4 acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) acc[b] = make_float4(0, 0, 0, 0);
  if (live) {
    const float* wp = w + (long)i4 * 4;
    int o = o_lo;
    for (; o + 4 <= o_hi; o += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(wp + (long)(o + u) * I);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const float g = b0 + b < B ? dy[(long)(b0 + b) * O + o + u] : 0.f;
          acc[b].x = fmaf(g, v[u].x, acc[b].x); acc[b].y = fmaf(g, v[u].y, acc[b].y);
          acc[b].z = fmaf(g, v[u].z, acc[b].z); acc[b].w = fmaf(g, v[u].w, acc[b].w);
        }
    }
    for (; o < o_hi; ++o) {
      const float4 v = *reinterpret_cast<const float4*>(wp + (long)o * I);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const float g = b0 + b < B ? dy[(long)(b0 + b) * O + o] : 0.f;
        acc[b].x = fmaf(g, v.x, acc[b].x); acc[b].y = fmaf(g, v.y, acc[b].y);
        acc[b].z = fmaf(g, v.z, acc[b].z); acc[b].w = fmaf(g, v.w, acc[b].w);
      }
    }
  }
  if (q > 0) {
#pragma unroll
    for (int b = 0; b < 8; ++b) part[q - 1][b][lane] = acc[b];
  }
  __syncthreads();
  if (q == 0 && live) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if (b0 + b >= B) continue;
      float4 r = acc[b];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 t = part[k][b][lane];
        r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
      }
      *reinterpret_cast<float4*>(dx + (long)(b0 + b) * I + i4 * 4) = r;
    }
  }
}
__global__ void fc_bgrad_kernel(float* __restrict__ db, const float* __restrict__ dy, int B, int O) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= O) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += dy[(long)b * O + o];
  db[o] = s;
}

__global__ __launch_bounds__(256) void sgd_mom_kernel(float* __restrict__ w, float* __restrict__ mom, const float* __restrict__ g,
                                                      float lr, float wd, float momentum, float rescale, float clip, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float gg = g[i] * rescale;
  if (clip > 0.f) gg = fminf(fmaxf(gg, -clip), clip);
  const float m = momentum * mom[i] - lr * (gg + wd * w[i]);
  mom[i] = m;
  w[i] = w[i] + m;
}

// every parameter in one launch: row r of the table = {w, mom, g, n, wd bits | first block << 32, Cin | kh*kw << 32 (0: natural g)}; a block finds its row with a
// ballot over the rows' first blocks
__global__ __launch_bounds__(256) void sgd_mom_multi_kernel(const unsigned long long* __restrict__ table, int rows, float lr,
                                                            float momentum, float rescale, float clip) {
  const unsigned b = blockIdx.x;
  // row = number of rows that start at or before this block, minus one: every wave counts them 64 at a time with one load + ballot
  // (a serial walk of the table cost the late rows ~40 dependent loads)
  const int lane = threadIdx.x & 63;
  int r = -1;
  for (int base = 0; base < rows; base += 64) {
    const int idx = base + lane;
    const bool le = idx < rows && (unsigned)(table[idx * 6 + 4] >> 32) <= b;
    r += __popcll(__ballot(le));
  }
  r = __builtin_amdgcn_readfirstlane(r);
  const unsigned long long* e = table + r * 6;
  float* w = reinterpret_cast<float*>(e[0]);
  float* mom = reinterpret_cast<float*>(e[1]);
  const float* g = reinterpret_cast<const float*>(e[2]);
  const size_t n = e[3];
  const float wd = __uint_as_float((unsigned)e[4]);
  // four consecutive parameters per thread (dwordx4 on w / mom and on a natural g; tensors and their row views are 16-byte aligned)
  const size_t i0 = ((size_t)(b - (unsigned)(e[4] >> 32)) * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  const int cin = (int)(unsigned)e[5], khw = (int)(e[5] >> 32);
  const int cnt = (int)min((size_t)4, n - i0);
  float wv[4], mv[4], gv[4];
  if (cnt == 4) {
    const float4 a = *reinterpret_cast<const float4*>(w + i0), c = *reinterpret_cast<const float4*>(mom + i0);
    wv[0] = a.x; wv[1] = a.y; wv[2] = a.z; wv[3] = a.w;
    mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w;
  } else {
    for (int j = 0; j < cnt; ++j) { wv[j] = w[i0 + j]; mv[j] = mom[i0 + j]; }
  }
  if (cin) {   // the gradient lies tap-major (co, tap, ci): deepim_conv2d_wgrad_tm
    const int K = cin * khw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= cnt) break;
      const size_t i = i0 + j, co = i / K;
      const int rem = (int)(i - co * K), ci = rem / khw, t = rem - ci * khw;
      gv[j] = g[co * K + (size_t)t * cin + ci];
    }
  } else if (cnt == 4) {
    const float4 a = *reinterpret_cast<const float4*>(g + i0);
    gv[0] = a.x; gv[1] = a.y; gv[2] = a.z; gv[3] = a.w;
  } else {
    for (int j = 0; j < cnt; ++j) gv[j] = g[i0 + j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float gg = gv[j] * rescale;
    if (clip > 0.f) gg = fminf(fmaxf(gg, -clip), clip);
    const float m = momentum * mv[j] - lr * (gg + wd * wv[j]);
    mv[j] = m;
    wv[j] = wv[j] + m;
  }
  if (cnt == 4) {
    *reinterpret_cast<float4*>(mom + i0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
    *reinterpret_cast<float4*>(w + i0) = make_float4(wv[0], wv[1], wv[2], wv[3]);
  } else {
    for (int j = 0; j < cnt; ++j) { mom[i0 + j] = mv[j]; w[i0 + j] = wv[j]; }
  }
}

void launch_wgrad_reduce(deepim_ctx* ctx, float* dw, const float* partial, long n, int S) {
  if (S >= 8 && n % 4 == 0)
    hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(di_div_up(n / 4, 64)), dim3(256), 0, ctx->stream, dw, partial, n / 4, S);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(di_div_up(n, 256)), dim3(256), 0, ctx->stream, dw, partial, n, S);
}

}  // namespace

extern "C" int deepim_sgd_mom_update_multi(deepim_ctx* ctx, const unsigned long long* table, int rows, int total_blocks,
                                           float lr, float momentum, float rescale, float clip) {
  DI_DEVICE(ctx);
  DI_REQUIRE(rows >= 0 && table != nullptr, "sgd_mom_update_multi: no table");   // (w, mom, g 16-byte aligned: unchecked, device-side data)
  if (rows == 0 || total_blocks <= 0) return 0;
  hipLaunchKernelGGL(sgd_mom_multi_kernel, dim3(total_blocks), dim3(256), 0, ctx->stream, table, rows, lr, momentum, rescale, clip);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_lrelu_bias_backward(deepim_ctx* ctx, float* dz, float* db, const float* dy, const float* add, const float* y,
                                          float slope, int B, int C, size_t hw) {
  DI_DEVICE(ctx);
  if (C == 0) return 0;
  if (B == 0 || hw == 0) {   // nothing to walk: the sum over no elements is zero (what deepim_bias_grad gives)
    DI_CHECK(hipMemsetAsync(db, 0, (size_t)C * sizeof(float), ctx->stream));
    return 0;
  }
  int S = (int)max(1L, min((long)di_div_up(1024, C), (long)di_div_up((long)hw, 4096)));
  const long per_slice = (long)di_div_up(di_div_up((long)hw, S), 1024) * 1024;
  S = di_div_up((long)hw, per_slice);
  void* scratch;
  int rc = deepim_scratch(ctx, (size_t)C * S * sizeof(double), &scratch);
  if (rc) return rc;
  if (hw % 4 == 0)
    hipLaunchKernelGGL(lrelu_bias_backward_kernel<true>, dim3(C, S), dim3(256), 0, ctx->stream, (double*)scratch, dz, dy, add, y,
                       slope, B, C, (long)hw, S, per_slice, db);
  else
    hipLaunchKernelGGL(lrelu_bias_backward_kernel<false>, dim3(C, S), dim3(256), 0, ctx->stream, (double*)scratch, dz, dy, add, y,
                       slope, B, C, (long)hw, S, per_slice, db);
  if (S > 1)
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3(di_div_up(C, 256)), dim3(256), 0, ctx->stream, db, (const double*)scratch, C, S);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_slice_lrelu_bias_scatter(deepim_ctx* ctx, float* out, float* db, const float* dcat, const float* ycat, int B,
                                               int ctotal, int coff, int C, int ho, int wo, int hf, int wf, int off_y, int off_x,
                                               float slope) {
  DI_DEVICE(ctx);
  DI_REQUIRE(coff >= 0 && C >= 0 && coff + C <= ctotal && off_y >= 0 && off_x >= 0 && off_y + ho <= hf && off_x + wo <= wf,
             "slice_lrelu_bias_scatter: slice outside the concat / frame too small");
  if (B == 0 || C == 0) return 0;
  hipLaunchKernelGGL(slice_lrelu_bias_scatter_kernel, dim3(C), dim3(256), 0, ctx->stream, out, db, dcat, ycat, B, ctotal, coff, C, ho,
                     wo, hf, wf, off_y, off_x, slope);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_lrelu_backward(deepim_ctx* ctx, float* dz, const float* dy, const float* y, float slope, size_t n) {
  DI_DEVICE(ctx);
  if (n == 0) return 0;
  hipLaunchKernelGGL(lrelu_backward_kernel, dim3(di_div_up((long)n, 256)), dim3(256), 0, ctx->stream, dz, dy, y, slope, n);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_bias_grad(deepim_ctx* ctx, float* db, const float* dz, int B, int C, size_t hw) {
  DI_DEVICE(ctx);
  if (C == 0) return 0;
  if (B == 0 || hw == 0) {   // empty batch / empty maps: the sum over nothing (as deepim_lrelu_bias_backward does)
    DI_CHECK(hipMemsetAsync(db, 0, (size_t)C * sizeof(float), ctx->stream));
    return 0;
  }
  // slices of whole 1024-element runs, enough blocks for ~4 per CU; fixed by the geometry (deterministic)
  int S = (int)max(1L, min((long)di_div_up(1024, C), (long)di_div_up((long)hw, 4096)));
  const long per_slice = (long)di_div_up(di_div_up((long)hw, S), 1024) * 1024;
  S = di_div_up((long)hw, per_slice);
  void* scratch;
  int rc = deepim_scratch(ctx, (size_t)C * S * sizeof(double), &scratch);
  if (rc) return rc;
  hipLaunchKernelGGL(bias_grad_kernel, dim3(C, S), dim3(256), 0, ctx->stream, (double*)scratch, dz, B, C, (long)hw, S, per_slice, db);
  if (S > 1)
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3(di_div_up(C, 256)), dim3(256), 0, ctx->stream, db, (const double*)scratch, C, S);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_conv_flip_weights(deepim_ctx* ctx, float* wt, const float* w, int Cout, int Cin, int kh, int kw) {
  DI_DEVICE(ctx);
  const long total = (long)Cout * Cin * kh * kw;
  if (total == 0) return 0;
  hipLaunchKernelGGL(conv_flip_weights_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, wt, w, Cout, Cin, kh,
                     kw, total);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_scatter2d(deepim_ctx* ctx, float* out, const float* in, int BC, int Ho, int Wo, int Hd, int Wd,
                                int stride, int off_y, int off_x) {
  DI_DEVICE(ctx);
  DI_REQUIRE(stride >= 1 && off_y >= 0 && off_x >= 0 && Hd >= off_y + (Ho - 1) * stride + 1 && Wd >= off_x + (Wo - 1) * stride + 1,
             "scatter2d: output too small");
  const long total = (long)BC * Hd * Wd;
  if (total == 0) return 0;
  hipLaunchKernelGGL(dilate2d_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, out, in, Ho, Wo, Hd, Wd, stride,
                     off_y, off_x, total);
  DI_LAUNCH_CHECK();
  return 0;
}
extern "C" int deepim_dilate2d(deepim_ctx* ctx, float* out, const float* in, int BC, int Ho, int Wo, int Hd, int Wd,
                               int stride) {
  return deepim_scatter2d(ctx, out, in, BC, Ho, Wo, Hd, Wd, stride, 0, 0);
}

extern "C" int deepim_upsample16_crop_backward(deepim_ctx* ctx, float* d_in, const float* dy, const float* w, int B, int C,
                                               int H, int W, int Ho, int Wo, int crop_y, int crop_x, float scale) {
  DI_DEVICE(ctx);
  const long total = (long)B * C * H * W;
  if (total == 0) return 0;
  hipLaunchKernelGGL(upsample16_backward_kernel, dim3(di_div_up(total, 4)), dim3(256), 0, ctx->stream, d_in, dy, w, C, H, W, Ho,
                     Wo, crop_y, crop_x, scale, total);
  DI_LAUNCH_CHECK();
  return 0;
}

// dst (B,C,hw) = channels [coff, coff+C) of src (B,Ctotal,hw): the backward of Concat (and the inverse of deepim_copy_channels)
extern "C" int deepim_extract_channels(deepim_ctx* ctx, float* dst, const float* src, int src_ctotal, int src_coff, int C, int B,
                                       size_t hw) {
  DI_DEVICE(ctx);
  if (B == 0 || C == 0) return 0;
  DI_CHECK(hipMemcpy2DAsync(dst, (size_t)C * hw * sizeof(float), src + (size_t)src_coff * hw, (size_t)src_ctotal * hw * sizeof(float),
                            (size_t)C * hw * sizeof(float), (size_t)B, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

static int conv2d_wgrad_impl(deepim_ctx* ctx, float* dw, float* db, const float* x, const float* dz, int B, int Cin, int H, int W,
                             int Cout, int kh, int kw, int stride, int pad, bool tap_major = false);

extern "C" int deepim_conv2d_wgrad(deepim_ctx* ctx, float* dw, const float* x, const float* dz, int B, int Cin, int H, int W,
                                   int Cout, int kh, int kw, int stride, int pad) {
  return conv2d_wgrad_impl(ctx, dw, nullptr, x, dz, B, Cin, H, W, Cout, kh, kw, stride, pad);
}

// weight and bias gradient of one layer: one launch for the few-filter layers (the stream kernel's first block sums dz on the way),
// deepim_bias_grad + the MFMA kernel otherwise
extern "C" int deepim_conv2d_wgrad_bias(deepim_ctx* ctx, float* dw, float* db, const float* x, const float* dz, int B, int Cin,
                                        int H, int W, int Cout, int kh, int kw, int stride, int pad) {
  DI_REQUIRE(db != nullptr, "conv2d_wgrad_bias: db is NULL (use deepim_conv2d_wgrad)");
  return conv2d_wgrad_impl(ctx, dw, db, x, dz, B, Cin, H, W, Cout, kh, kw, stride, pad);
}

// The weight gradient in TAP-MAJOR layout dw_tm[co][ky*kw + kx][ci] (Cin % 8 == 0, Cout > 4, the LDS-staged kernel): same sums in the
// same order as deepim_conv2d_wgrad, K rows permuted — see wgrad_lds_kernel<BM, TAPM>.
extern "C" int deepim_conv2d_wgrad_tm(deepim_ctx* ctx, float* dw_tm, const float* x, const float* dz, int B, int Cin, int H, int W,
                                      int Cout, int kh, int kw, int stride, int pad) {
  DI_REQUIRE((Cin & 7) == 0 && Cout > 4 && ctx->wgrad_lds, "conv2d_wgrad_tm: needs Cin % 8 == 0, Cout > 4 and the LDS-staged kernel");
  return conv2d_wgrad_impl(ctx, dw_tm, nullptr, x, dz, B, Cin, H, W, Cout, kh, kw, stride, pad, true);
}

namespace {
// natural (co, ci, tap) ← tap-major (co, tap, ci): thread per natural element (coalesced writes, 4-byte gathers whose neighbours
// in ci are adjacent)
__global__ __launch_bounds__(256) void grad_to_natural_kernel(float* __restrict__ nat, const float* __restrict__ tm, int Cin, int khw,
                                                              long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int K = Cin * khw;
  const long co = i / K;
  const int rem = (int)(i - co * K), ci = rem / khw, t = rem - ci * khw;
  nat[i] = tm[co * K + (long)t * Cin + ci];
}
}  // namespace

extern "C" int deepim_weight_grad_to_natural(deepim_ctx* ctx, float* dw, const float* dw_tm, int Cout, int Cin, int khw) {
  DI_DEVICE(ctx);
  const long total = (long)Cout * Cin * khw;
  if (total == 0) return 0;
  hipLaunchKernelGGL(grad_to_natural_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, dw, dw_tm, Cin, khw, total);
  DI_LAUNCH_CHECK();
  return 0;
}

static int conv2d_wgrad_impl(deepim_ctx* ctx, float* dw, float* db, const float* x, const float* dz, int B, int Cin, int H, int W,
                             int Cout, int kh, int kw, int stride, int pad, bool tap_major) {
  DI_DEVICE(ctx);
  if (B == 0) {   // a rank with an empty shard: the gradients are zero, not whatever the buffers held (update() applies them)
    if (dw) DI_CHECK(hipMemsetAsync(dw, 0, (size_t)Cout * Cin * kh * kw * sizeof(float), ctx->stream));
    if (db) DI_CHECK(hipMemsetAsync(db, 0, (size_t)Cout * sizeof(float), ctx->stream));
    return 0;
  }
  WgradParams p;
  p.x = x; p.dz = dz;
  p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
  p.Ho = (H + 2 * pad - kh) / stride + 1;
  p.Wo = (W + 2 * pad - kw) / stride + 1;
  p.K = Cin * kh * kw;
  const int HW = p.Ho * p.Wo;
  DI_REQUIRE((HW & 3) == 0, "conv2d_wgrad: Ho*Wo must be a multiple of 4");
  p.ktiles = di_div_up(p.K, 128);
  const int bm = (ctx->wgrad_lds && Cout <= 64) ? 64 : 128;
  p.mtiles = di_div_up(Cout, bm);
  const long n_dw = (long)Cout * p.K;
  if (ctx->wgrad_lds && Cout <= 4 && kh == kw && (kh == 3 || kh == 4)) {   // prediction heads, flow upsamplers
#define DI_WG_FEW(CO, KS) hipLaunchKernelGGL((wgrad_fewout_kernel<CO, KS>), dim3(Cin), dim3(256), 0, ctx->stream, dw, x, dz, B, Cin, H, W, Cout, p.Ho, p.Wo, stride, pad, db)
    if (kh == 3) { if (Cout == 1) DI_WG_FEW(1, 3); else if (Cout == 2) DI_WG_FEW(2, 3); else DI_WG_FEW(4, 3); }
    else { if (Cout == 1) DI_WG_FEW(1, 4); else if (Cout == 2) DI_WG_FEW(2, 4); else DI_WG_FEW(4, 4); }
#undef DI_WG_FEW
    DI_LAUNCH_CHECK();
    return 0;
  }
  if (db) {
    const int rc = deepim_bias_grad(ctx, db, dz, B, Cout, (size_t)HW);
    if (rc) return rc;
  }
  if (ctx->wgrad_lds && (size_t)B * Cin * H * W * 4 < 0x7fffffffUL && (size_t)B * Cout * HW * 4 < 0x7fffffffUL) {
    // LDS-staged kernel: chunks of 16 pixels of one sample; slices of whole chunks, fixed by the geometry (deterministic)
    const long chunks = (long)B * di_div_up(HW, WG_PIX);
    DI_REQUIRE(chunks < (1L << 30), "conv2d_wgrad: too many pixel chunks");
    // slice count: the blocks of a launch run in rounds of `slots` (256 CUs x 3 resident blocks, 4 for the 64-row tile); a
    // count just above a multiple of that leaves a nearly empty last round (1024 blocks on 768 slots: a third of the chip
    // idle for half the kernel). Pick S minimising rounds x chunks-per-slice (+ the reduce pass), slices >= 16 chunks.
    const long tiles = (long)p.ktiles * p.mtiles, slots = bm == 64 ? 1024 : 768;
    long S = 1;
    float best = 1e30f;
    for (long cand = 1; cand <= max(1L, min(chunks / 16, 4 * slots / tiles)); ++cand) {
      const long cps_ = di_div_up(chunks, cand), s_eff = di_div_up(chunks, cps_);
      // + the second pass: every slice writes and re-reads Cout·K floats (≈4.5e-7 chunk times per float at ≈4 TB/s)
      const float cost = (float)di_div_up(tiles * s_eff, slots) * (float)cps_ +
                         (s_eff > 1 ? 2.f + (float)s_eff * (0.02f + 4.5e-7f * (float)n_dw) : 0.f);
      if (cost < best * 0.99f) { best = cost; S = cand; }
    }
    p.groups_per_slice = (int)di_div_up(chunks, S);
    p.S = (int)di_div_up(chunks, p.groups_per_slice);
    if (p.S == 1) {
      p.partial = dw;                                  // a single slice writes the gradient itself
    } else {
      void* scratch;
      int rc = deepim_scratch(ctx, (size_t)p.S * n_dw * sizeof(float), &scratch);
      if (rc) return rc;
      p.partial = (float*)scratch;
    }
    const dim3 grid(p.ktiles * p.mtiles * p.S);
    if (tap_major) {
      if (bm == 64) hipLaunchKernelGGL((wgrad_lds_kernel<64, true>), grid, dim3(256), 0, ctx->stream, p);
      else hipLaunchKernelGGL((wgrad_lds_kernel<128, true>), grid, dim3(256), 0, ctx->stream, p);
    } else if (bm == 64) hipLaunchKernelGGL(wgrad_lds_kernel<64>, grid, dim3(256), 0, ctx->stream, p);
    else hipLaunchKernelGGL(wgrad_lds_kernel<128>, grid, dim3(256), 0, ctx->stream, p);
    if (p.S > 1) launch_wgrad_reduce(ctx, dw, p.partial, n_dw, p.S);
    DI_LAUNCH_CHECK();
    return 0;
  }
  DI_REQUIRE(!tap_major, "conv2d_wgrad_tm: tensors of 2 GiB and more take the register-fed kernel, which has no tap-major order");
  const long groups = (long)B * ((HW + 7) >> 3);
  // enough pixel slices to fill the chip (~1024 blocks), each at least 64 groups (512 pixels) long; fixed by the geometry
  long S = di_div_up(1024, p.ktiles * p.mtiles);
  S = min(S, max(1L, groups / 64));
  p.groups_per_slice = (int)di_div_up(groups, S);
  p.S = (int)di_div_up(groups, p.groups_per_slice);
  const long n = (long)Cout * p.K;
  void* scratch;
  int rc = deepim_scratch(ctx, (size_t)p.S * n * sizeof(float), &scratch);
  if (rc) return rc;
  p.partial = (float*)scratch;
  hipLaunchKernelGGL(wgrad_mfma_kernel, dim3(p.ktiles * p.mtiles * p.S), dim3(256), 0, ctx->stream, p);
  launch_wgrad_reduce(ctx, dw, p.partial, n, p.S);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_conv_subkernel_flip(deepim_ctx* ctx, float* wt, const float* w, int Cout, int Cin, int kh, int kw, int ky0,
                                          int kx0, int nky, int nkx) {
  DI_DEVICE(ctx);
  DI_REQUIRE(ky0 >= 0 && kx0 >= 0 && nky >= 1 && nkx >= 1 && ky0 + 2 * (nky - 1) < kh && kx0 + 2 * (nkx - 1) < kw,
             "conv_subkernel_flip: taps outside the kernel");
  const long total = (long)Cin * Cout * nky * nkx;
  hipLaunchKernelGGL(conv_subkernel_flip_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, wt, w, Cout, Cin, kh, kw,
                     ky0, kx0, nky, nkx, total);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_interleave2d(deepim_ctx* ctx, float* dx, const float* src, int BC, int Hs, int Ws, int cy, int cx, int H,
                                   int W, int py, int px) {
  DI_DEVICE(ctx);
  const int hq = (H - py + 1) / 2, wq = (W - px + 1) / 2;
  DI_REQUIRE(py >= 0 && py < 2 && px >= 0 && px < 2 && cy >= 0 && cx >= 0 && hq + cy <= Hs && wq + cx <= Ws,
             "interleave2d: class window outside the source");
  const long total = (long)BC * hq * wq;
  if (total == 0) return 0;
  hipLaunchKernelGGL(interleave2d_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, dx, src, Hs, Ws, cy, cx, H, W,
                     py, px, hq, wq, total);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_fc_backward(deepim_ctx* ctx, float* dx, float* dw, float* db, const float* dy, const float* x,
                                  const float* w, int B, int I, int O) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE((I & 3) == 0, "fc_backward: input width must be a multiple of 4");
  if (dw) hipLaunchKernelGGL(fc_wgrad_kernel, dim3(di_div_up((long)O * (I / 4), 256)), dim3(256), 0, ctx->stream, dw, dy, x, B, I, O);
  if (dx) hipLaunchKernelGGL(fc_dgrad_kernel, dim3(di_div_up(I / 4, 64), di_div_up(B, 8)), dim3(256), 0, ctx->stream, dx, dy, w, B, I, O);
  if (db) hipLaunchKernelGGL(fc_bgrad_kernel, dim3(di_div_up(O, 64)), dim3(64), 0, ctx->stream, db, dy, B, O);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_sgd_mom_update(deepim_ctx* ctx, float* w, float* mom, const float* g, float lr, float wd,
                                     float momentum, float rescale, float clip, size_t n) {
  DI_DEVICE(ctx);
  if (n == 0) return 0;
  hipLaunchKernelGGL(sgd_mom_kernel, dim3(di_div_up((long)n, 256)), dim3(256), 0, ctx->stream, w, mom, g, lr, wd, momentum,
                     rescale, clip, n);
  DI_LAUNCH_CHECK();
  return 0;
}
