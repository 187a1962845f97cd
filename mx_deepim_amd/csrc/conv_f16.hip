// fp16 convolution path of the matching network (BASELINE config 5): activations and
// weights in fp16, products on the fp16 matrix cores (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate),
// accumulation + bias + LeakyReLU in fp32. Same layers as csrc/conv.hip (deepIM_flownet.py:63-107) with a
// documented, looser tolerance (fp16 cannot meet 1e-4).
//
// Layout is chosen for the instruction, not inherited from the fp32 path:
//   * activations are NHWC fp16 and K runs (ky,kx,ci) with ci fastest, so the 8 consecutive k-values one
//     lane feeds to a 32x32x16 MFMA are 8 consecutive channels of one tap = ONE 16-byte buffer load
//     (the fp32 path needs 8 dword gathers for the same amount of K);
//   * zero padding is per tap and done by the hardware (invalid tap → voffset bit 31 → load returns 0);
//   * LDS holds [k-octet][m or pixel][8 halves]: the gather is stored with one ds_write_b128, fragments are
//     read with one conflict-free ds_read_b128 per 32x32x16 operand (lanes 0-31 octet 2t, lanes 32-63 2t+1);
//   * K chunk = 64 (8 octets): 4 dwordx4 of activations + 4 dwordx4 of pre-packed weights per thread per
//     chunk, 16 MFMAs per wave per chunk; LDS double-buffered, one s_barrier per chunk;
//   * output is NHWC fp16 = "pixel-major", which is exactly the flattened (n,ho,wo) pixel index the GEMM
//     uses: each lane stores 4 consecutive output channels (8 bytes) per accumulator quad.
// Roofline: at 128x128 tiles the arithmetic intensity towards L2 is 64 FLOP/B, i.e. this kernel is L1/L2-
// bandwidth-bound, not MFMA-bound (2.5 PFLOP/s dense fp16 peak).
#include "common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int HOCT = 8;   // octets per chunk (K chunk = 8 octets x 8 halves = 64)
constexpr int HBM = 128, HBN = 128;

struct ConvF16Params {
  const void* in;       // NHWC fp16 (B,H,W,Cin)
  const h8* wp;         // packed [mtile][chunk][octet][128][8]
  const int2* tab;      // per k-octet: {byte offset ((ky*W+kx)*Cin + ci0)*2, tap bit ky*8+kx}; padding → bit 63
  const float* bias;
  _Float16* out;        // NHWC fp16 (B,Ho,Wo,Cout)
  int B, Cin, H, W, Cout, Ho, Wo, stride, pad, nchunk;
  float slope;
  long npix;
  int gx, gy;
  int pad_bytes;
  unsigned in_bytes;
  int ksplit, chunks_per_split;  // split-K across grid slices for under-filled grids
  float* partial;                // [ksplit][npix][Cout] fp32 partial sums when ksplit > 1
};

__global__ __launch_bounds__(256) void conv_f16_kernel(ConvF16Params p) {
  __shared__ __attribute__((aligned(16))) h8 As[2][HOCT * HBM];
  __shared__ __attribute__((aligned(16))) h8 Bs[2][HOCT * HBN];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  // XCD-aware tile order (see csrc/conv.hip)
  int vid;
  {
    const int total = p.gx * p.gy * p.ksplit, bid = blockIdx.x;
    const int xcd = bid & 7, qn = total >> 3, rn = total & 7;
    vid = xcd * qn + min(xcd, rn) + (bid >> 3);
  }
  const int bx = vid % p.gx, mb = (vid / p.gx) % p.gy, split = vid / (p.gx * p.gy);
  const long n0 = (long)bx * HBN;
  const int kc_begin = split * p.chunks_per_split, kc_end = min(p.nchunk, kc_begin + p.chunks_per_split);

  // per-thread gather state: one pixel, 4 octet rows per chunk
  const int gp = tid & (HBN - 1);
  const int orow0 = __builtin_amdgcn_readfirstlane((tid >> 7) * 4);
  const long pix = n0 + gp;
  unsigned voff = 0x80000000u;  // bit 31 set = out of range (threads beyond the last pixel)
  unsigned long long m64 = 0;
  if (pix < p.npix) {
    const int hw = p.Ho * p.Wo;
    const int n = (int)(pix / hw);
    const int r = (int)(pix - (long)n * hw);
    const int ho = r / p.Wo, wo = r - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    voff = (unsigned)(((n * p.H + hi0) * p.W + wi0) * p.Cin * 2 + p.pad_bytes);
    unsigned mky = 0, mkx = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      if (hi0 + k >= 0 && hi0 + k < p.H) mky |= 1u << k;
      if (wi0 + k >= 0 && wi0 + k < p.W) mkx |= 1u << k;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((mky >> k) & 1u) m64 |= (unsigned long long)mkx << (8 * k);
  }
  const unsigned long long ninv64 = ~m64;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.in - p.pad_bytes), 0, (int)(p.in_bytes + (unsigned)p.pad_bytes), 0x00020000);

  const h8* wblk = p.wp + (long)mb * p.nchunk * (HOCT * HBM) + tid;
  i32x4 areg[4], breg[4];

#define LOAD_CHUNK(kc)                                                                                   \
  {                                                                                                      \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                        \
      areg[e] = *reinterpret_cast<const i32x4*>(wblk + (long)(kc) * (HOCT * HBM) + e * 256);             \
    const int2* tp = p.tab + (kc) * HOCT + orow0;                                                        \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                      \
      const int2 t = tp[e];                                                                              \
      const unsigned inv = (unsigned)(ninv64 >> t.y);   /* no SALU-produced VALU operand, see conv.hip */  \
      breg[e] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((inv << 31) | voff), t.x, 0)); \
    }                                                                                                    \
  }
#define STORE_CHUNK(buf)                                                                                 \
  {                                                                                                      \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                        \
      *reinterpret_cast<i32x4*>(&As[buf][tid + e * 256]) = areg[e];                                      \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                        \
      *reinterpret_cast<i32x4*>(&Bs[buf][(orow0 + e) * HBN + gp]) = breg[e];                             \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  LOAD_CHUNK(kc_begin);
  STORE_CHUNK(0);
  __syncthreads();

  const int lrow = lane >> 5, lcol = lane & 31;
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const int buf = (kc - kc_begin) & 1;
    const bool more = kc + 1 < kc_end;
    if (more) LOAD_CHUNK(kc + 1);
    const h8* as = &As[buf][lrow * HBM + wm0 + lcol];
    const h8* bs = &Bs[buf][lrow * HBN + wn0 + lcol];
#pragma unroll
    for (int t = 0; t < HOCT / 2; ++t) {
      h8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = as[t * 2 * HBM + i * 32];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = bs[t * 2 * HBN + j * 32];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) STORE_CHUNK(buf ^ 1);
    __syncthreads();
  }
#undef LOAD_CHUNK
#undef STORE_CHUNK

  // epilogue: bias + LeakyReLU in fp32, NHWC fp16 store (offset = pixel*Cout + co; 4 channels = 8 bytes per store)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long op = n0 + wn0 + j * 32 + lcol;
    if (op >= p.npix) continue;
    _Float16* orow = p.out + op * p.Cout;
    float* prow = p.partial ? p.partial + ((long)split * p.npix + op) * p.Cout : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co0 = mb * HBM + wm0 + i * 32 + 8 * g + 4 * lrow;
        if (co0 < p.Cout && prow) {  // split-K: raw fp32 partial sums, reduced by splitk_f16_reduce_kernel
          *reinterpret_cast<float4*>(prow + co0) =
              make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        } else if (co0 < p.Cout) {
          h4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[i][j][4 * g + r] + (p.bias ? p.bias[co0 + r] : 0.f);
            x = x > 0.f ? x : x * p.slope;
            v[r] = (_Float16)x;
          }
          *reinterpret_cast<h4*>(orow + co0) = v;
        }
      }
  }
}

// split-K second pass: out[pix][c] = f16(lrelu(Σ_s partial[s][pix][c] + bias[c])), fixed order
__global__ __launch_bounds__(256) void splitk_f16_reduce_kernel(_Float16* __restrict__ out, const float* __restrict__ partial,
                                                                const float* __restrict__ bias, long total, int S,
                                                                int Cout, float slope) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float v = partial[i];
  for (int s = 1; s < S; ++s) v += partial[(long)s * total + i];
  v = v + (bias ? bias[i % Cout] : 0.f);
  v = v > 0.f ? v : v * slope;
  out[i] = (_Float16)v;
}

// packed[mt][kc][o][m][h] = f16(w[mt*128+m][ci0+h][ky][kx]); k-octet q = kc*8+o → tap = q / (Cin_pad/8)
__global__ void pack_f16_kernel(_Float16* __restrict__ packed, const float* __restrict__ w, int Cout, int Cin,
                                int Cin_pad, int kh, int kw, int nchunk, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int h = (int)(i & 7);
  const int m = (int)((i >> 3) & 127);
  const int o = (int)((i >> 10) & 7);
  const int kc = (int)((i >> 13) % nchunk);
  const int mt = (int)((i >> 13) / nchunk);
  const int opt = Cin_pad / 8;
  const int q = kc * HOCT + o, tap = q / opt, ci = (q % opt) * 8 + h, co = mt * HBM + m;
  float v = 0.f;
  if (co < Cout && ci < Cin && tap < kh * kw) v = w[(((long)co * Cin + ci) * kh + tap / kw) * kw + tap % kw];
  packed[i] = (_Float16)v;
}

__global__ void build_f16_tab_kernel(int2* __restrict__ tab, int noct, int noct_pad, int Cin_pad, int kw, int W) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= noct_pad) return;
  if (q >= noct) { tab[q] = make_int2(0, 63); return; }
  const int opt = Cin_pad / 8, tap = q / opt, ci0 = (q % opt) * 8, ky = tap / kw, kx = tap % kw;
  tab[q] = make_int2(((ky * W + kx) * Cin_pad + ci0) * 2, ky * 8 + kx);
}

// NCHW fp32 → NHWC fp16 (channels zero-padded to Cpad); one thread per (pixel, 8-channel octet)
__global__ __launch_bounds__(256) void nchw_to_nhwc_f16_kernel(_Float16* __restrict__ out, const float* __restrict__ in,
                                                               int C, int Cpad, long hw, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int opp = Cpad / 8;
  const int oc = (int)(i % opp);
  const long pix = i / opp;             // n*hw + r
  const long n = pix / hw, r = pix - n * hw;
  h8 v;
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const int c = oc * 8 + h;
    v[h] = c < C ? (_Float16)in[(n * C + c) * hw + r] : (_Float16)0.f;
  }
  *reinterpret_cast<h8*>(out + pix * Cpad + oc * 8) = v;
}

// NHWC fp16 → NCHW fp32; lanes run along pixels of one channel (coalesced stores)
__global__ __launch_bounds__(256) void nhwc_f16_to_nchw_kernel(float* __restrict__ out, const _Float16* __restrict__ in,
                                                               int C, long hw, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long r = i % hw;
  const long c = (i / hw) % C;
  const long n = i / (hw * C);
  out[i] = (float)in[(n * hw + r) * C + c];
}

inline int f16_chunks(int Cin_pad, int kh, int kw) { return di_div_up(kh * kw * (Cin_pad / 8), HOCT); }

}  // namespace

extern "C" int deepim_nchw_f32_to_nhwc_f16(deepim_ctx* ctx, void* out_f16, const float* in, int B, int C, int H, int W,
                                           int Cpad) {
  DI_DEVICE(ctx);
  DI_REQUIRE(Cpad >= C && (Cpad & 7) == 0, "nchw_to_nhwc_f16: Cpad must be a multiple of 8 and >= C");
  const long total = (long)B * H * W * (Cpad / 8);
  if (total == 0) return 0;
  hipLaunchKernelGGL(nchw_to_nhwc_f16_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream,
                     (_Float16*)out_f16, in, C, Cpad, (long)H * W, total);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_nhwc_f16_to_nchw_f32(deepim_ctx* ctx, float* out, const void* in_f16, int B, int C, int H, int W) {
  DI_DEVICE(ctx);
  const long total = (long)B * C * H * W;
  if (total == 0) return 0;
  hipLaunchKernelGGL(nhwc_f16_to_nchw_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, out,
                     (const _Float16*)in_f16, C, (long)H * W, total);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t deepim_conv_f16_packed_size(int Cout, int Cin_pad, int kh, int kw) {
  return (size_t)di_div_up(Cout, HBM) * f16_chunks(Cin_pad, kh, kw) * HOCT * HBM * 8 * sizeof(_Float16);
}

extern "C" int deepim_conv_f16_pack_weights(deepim_ctx* ctx, void* packed, const float* w, int Cout, int Cin, int Cin_pad,
                                            int kh, int kw) {
  DI_DEVICE(ctx);
  DI_REQUIRE(Cin_pad >= Cin && (Cin_pad & 7) == 0, "conv_f16_pack: Cin_pad must be a multiple of 8 and >= Cin");
  const int nchunk = f16_chunks(Cin_pad, kh, kw);
  const long total = (long)di_div_up(Cout, HBM) * nchunk * HOCT * HBM * 8;
  hipLaunchKernelGGL(pack_f16_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, (_Float16*)packed, w, Cout,
                     Cin, Cin_pad, kh, kw, nchunk, total);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_conv2d_f16_forward(deepim_ctx* ctx, void* out_nhwc_f16, const void* in_nhwc_f16,
                                         const void* packed_w, const float* bias, int B, int Cin_pad, int H, int W,
                                         int Cout, int kh, int kw, int stride, int pad, float slope) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE((Cin_pad & 7) == 0 && (Cout & 3) == 0, "conv2d_f16: Cin_pad % 8 and Cout % 4 must be 0");
  DI_REQUIRE(kh <= 7 && kw <= 7, "conv2d_f16: kernel larger than 7 not supported");
  ConvF16Params p;
  p.in = in_nhwc_f16; p.wp = (const h8*)packed_w; p.bias = bias; p.out = (_Float16*)out_nhwc_f16;
  p.B = B; p.Cin = Cin_pad; p.H = H; p.W = W; p.Cout = Cout;
  p.Ho = (H + 2 * pad - kh) / stride + 1;
  p.Wo = (W + 2 * pad - kw) / stride + 1;
  p.stride = stride; p.pad = pad; p.slope = slope;
  p.nchunk = f16_chunks(Cin_pad, kh, kw);
  p.npix = (long)B * p.Ho * p.Wo;
  p.pad_bytes = (pad * W + pad) * Cin_pad * 2;
  const size_t in_bytes = (size_t)B * H * W * Cin_pad * 2;
  DI_REQUIRE(in_bytes + p.pad_bytes < 0x7fffffffUL, "conv2d_f16: input tensor must be < 2 GiB per launch");
  p.in_bytes = (unsigned)in_bytes;
  // tap table, cached in the context (mode 2 = fp16 octet table)
  int2* tab = nullptr;
  for (auto& k : ctx->conv_tabs)
    if (k.mode == 2 && k.Cin == Cin_pad && k.kh == kh && k.kw == kw && k.H == H && k.W == W) tab = (int2*)k.tab;
  if (!tab) {
    DI_REQUIRE(!ctx->capturing, "conv2d_f16: tap table built during graph capture; run once eagerly first");
    const int noct = kh * kw * (Cin_pad / 8), noct_pad = p.nchunk * HOCT;
    DI_CHECK(hipMalloc((void**)&tab, (size_t)noct_pad * sizeof(int2)));
    hipLaunchKernelGGL(build_f16_tab_kernel, dim3(di_div_up(noct_pad, 256)), dim3(256), 0, ctx->stream, tab, noct,
                       noct_pad, Cin_pad, kw, W);
    ctx->conv_tabs.push_back({2, Cin_pad, kh, kw, H, W, (void*)tab});
  }
  p.tab = tab;
  p.gx = di_div_up(p.npix, HBN); p.gy = di_div_up(Cout, HBM);
  const int blocks = p.gx * p.gy;
  int ks = 1;
  if (blocks < 384 && ctx->conv_max_split != 1) ks = min(min(di_div_up(512, blocks), max(1, p.nchunk / 4)), 16);
  if (ctx->conv_max_split > 1) ks = min(ks, ctx->conv_max_split);
  p.chunks_per_split = di_div_up(p.nchunk, ks);
  p.ksplit = di_div_up(p.nchunk, p.chunks_per_split);
  p.partial = nullptr;
  if (p.ksplit > 1) {
    void* scratch;
    int rc = deepim_scratch(ctx, (size_t)p.ksplit * p.npix * Cout * sizeof(float), &scratch);
    if (rc) return rc;
    p.partial = (float*)scratch;
  }
  hipLaunchKernelGGL(conv_f16_kernel, dim3(blocks * p.ksplit), dim3(256), 0, ctx->stream, p);
  if (p.ksplit > 1) {
    const long total = p.npix * Cout;
    hipLaunchKernelGGL(splitk_f16_reduce_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, p.out,
                       p.partial, bias, total, p.ksplit, Cout, slope);
  }
  DI_LAUNCH_CHECK();
  return 0;
}
