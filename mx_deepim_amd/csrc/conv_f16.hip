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

// Tile shapes. A 32x32x16 MFMA is 32 cycles; a wave tile of TM x TN MFMA tiles reads (TM + TN) KB of fragments from LDS
// per TM·TN MFMAs. With the 2x2 tile of the first version that is 1 KB per MFMA and wave = 128 B/clk per CU at full MFMA
// rate: the whole LDS bandwidth, before the stores (measured 440 TFLOP/s = 18 %), and a K chunk lasted only 512 cycles, less
// than an L2 hit, with the next chunk's loads just one chunk ahead. The shapes below halve the LDS bytes per MFMA
// (2x4: 768 B, 4x4: 512 B) and make a chunk 1024 / 2048 cycles long:
//   Cout <= 64 (conv1)        waves 1x4, wave tile 2x2:  64 x 256 block, 2-3 blocks per CU (7 K chunks only: prologue and
//                             epilogue of one block overlap the main loop of another)
//   Cout % 256 != 0 (conv2)   waves 2x2, wave tile 2x4: 128 x 256 block
//   Cout % 256 == 0           waves 2x2, wave tile 4x4: 256 x 256 block (256 accumulator registers, one wave per SIMD)
inline int f16_bm(int Cout) { return Cout <= 64 ? 64 : ((Cout & 255) == 0 ? 256 : 128); }
// DMA kernel: 128x64 wave tiles, two blocks per CU (DEEPIM_F16_TN4=1 restores the 128x128 wave tiles, one block per CU)
// Dev switches live in the context (deepim_set_option "f16_dev_flags", DI_F16_*), never in the process environment:
// tiling decides the summation order, and every rank must take the same plan.
inline int f16_bn(int Cout, bool dma, int dev) {
  if (!dma) return 256;
  const bool tn4 = (dev & DI_F16_TN4) != 0;
  if (f16_bm(Cout) == 256) return (tn4 || (dev & DI_F16_W8)) ? 256 : 128;
  return f16_bm(Cout) == 128 ? (tn4 ? 512 : 256) : 256;
}

struct ConvF16Params {
  const void* in;       // NHWC fp16 (B,H,W,Cin)
  const h8* wp;         // packed [mtile][chunk][octet][BM][8]
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
  int stride_kw;                 // (kh << 16) | kw, for the DMA kernel's scalar tap counters
  // tail split (DMA kernel, ksplit == 1): tiles [0, n_full) run whole; the R = gx·gy − n_full tiles of the under-filled last
  // round are cut into tail_s K slices of tail_cps chunks → tile-local fp32 partials [slice][R][BN pixels][BM channels]
  int n_full, tail_s, tail_cps;
  float* tail_partial;
  int* status;                   // context status word (X3: saturation flag)
  float acc_scale, out_scale;    // X3 mode: accumulator → real units (2^-(s_act+s_w)), real units → stored activations (2^s_act)
};

// X3 ("split fp16") operands: a real value v is carried as the fp16 pair hi = f16(v·2^s), lo = f16(v·2^s − hi), i.e. 22
// significand bits, and a product as hi·hi + hi·lo + lo·hi on the fp16 matrix cores with fp32 accumulation (the dropped
// lo·lo term is 2^-22 relative): fp32-grade results at 16/3 of the fp32 MFMA rate. Tensors keep the NHWC fp16 machinery:
// 16 real channels are one 32-half record [hi 0..15 | lo 0..15], so a tensor with C real channels looks like an fp16
// tensor with 2C channels and the loaders below run unchanged; one 32-wide K chunk is then 16 real channels of a tap.
struct X3Pair { _Float16 hi, lo; };
__device__ __forceinline__ X3Pair x3_split(float x, float scale) {
  float v = x * scale;
  v = fminf(fmaxf(v, -60000.f), 60000.f);     // saturate instead of inf (fp16 max 65504)
  const _Float16 h = (_Float16)v;
  return {h, (_Float16)(v - (float)h)};
}
// the same, tracking the largest scaled magnitude in `amax` (one v_max per value): the caller reports a clamp once per thread
// through bit DI_STATUS_X3_SATURATED of the context's status word — saturation is never silent
__device__ __forceinline__ X3Pair x3_split(float x, float scale, float& amax) {
  amax = fmaxf(amax, fabsf(x * scale));
  return x3_split(x, scale);
}
__device__ __forceinline__ void x3_report(float amax, int* status) {
  if (amax > 60000.f) atomicOr(status, DI_STATUS_X3_SATURATED);
}

// validity of the (ky,kx) taps of a pixel as a 64-bit word (bit ky*8+kx), kh,kw <= 7: rows/columns hi0+k, wi0+k inside the frame
__device__ __forceinline__ unsigned long long tap_mask64(int hi0, int wi0, int H, int W) {
  auto range7 = [](int lo0, int n) -> unsigned {          // bits k in [0,7) with 0 <= lo0 + k < n
    const int lo = max(0, -lo0), hi = min(7, n - lo0);
    return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
  };
  const unsigned mky = range7(hi0, H), mkx = range7(wi0, W);
  // spread the 7 row bits to byte positions, then every set byte gets the column mask
  const unsigned long long rows = ((unsigned long long)mky * 0x0002040810204081ull) & 0x0101010101010101ull;
  return rows * (unsigned long long)mkx;
}

template <int WGM, int WGN, int TM, int TN, bool UT>   // UT: uniform tap per chunk (Cin_pad % 64 == 0)
__global__ __launch_bounds__(256, 1) void conv_f16_kernel(ConvF16Params p) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int NA = BM * HOCT / 256;      // weight dwordx4 per thread per chunk
  static_assert(WGM * WGN == 4 && BN % 256 == 0 && NA >= 1, "tile shape");
  extern __shared__ __attribute__((aligned(16))) h8 smem[];   // [2][HOCT*BM] weights, then [2][HOCT*BN] activations
  h8* As = smem;
  h8* Bs = smem + 2 * HOCT * BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WGN) * (TM * 32), wn0 = (wave % WGN) * (TN * 32);
  // XCD-aware tile order (see csrc/conv.hip)
  int vid;
  {
    const int total = p.gx * p.gy * p.ksplit, bid = blockIdx.x;
    const int xcd = bid & 7, qn = total >> 3, rn = total & 7;
    vid = xcd * qn + min(xcd, rn) + (bid >> 3);
  }
  const int bx = vid % p.gx, mb = (vid / p.gx) % p.gy, split = vid / (p.gx * p.gy);
  const long n0 = (long)bx * BN;
  const int kc_begin = split * p.chunks_per_split, kc_end = min(p.nchunk, kc_begin + p.chunks_per_split);

  // Activation gather. One load instruction of a wave covers 8 pixels x the 8 octets of the chunk: lane = (pixel l>>3,
  // octet l&7). With Cin_pad % 64 == 0 the 8 octets of a pixel are 128 contiguous bytes = ONE cache line per 8 lanes,
  // 8 lines per wave-instruction. (The first version gave every lane its own pixel: 64 lines per instruction, and the L1's
  // one-tag-lookup-per-line rate, not bytes, bounded the kernel — measured: with the MFMAs removed it ran 20 % faster,
  // with the loads removed 60 %.) Per thread: a fixed octet, NR = BN/32 pixels.
  constexpr int NR = BN / 32;
  const int oct = tid & 7;
  unsigned voff[NR];
  unsigned long long ninv64[NR];
  {
    // the thread's pixels are 32 apart: one division for the first, then (n, ho, wo) advance incrementally
    const int hw = p.Ho * p.Wo;
    long pix = n0 + (tid >> 3);
    int n = (int)(pix / hw);
    int rr = (int)(pix - (long)n * hw);
    int ho = rr / p.Wo, wo = rr - ho * p.Wo;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      voff[r] = 0x80000000u;  // bit 31 set = out of range (threads beyond the last pixel)
      unsigned long long m64 = 0;
      if (pix < p.npix) {
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        voff[r] = (unsigned)(((n * p.H + hi0) * p.W + wi0) * p.Cin * 2 + p.pad_bytes);
        m64 = tap_mask64(hi0, wi0, p.H, p.W);
      }
      ninv64[r] = ~m64;
      pix += 32; wo += 32;
      while (wo >= p.Wo) { wo -= p.Wo; if (++ho == p.Ho) { ho = 0; ++n; } }
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.in - p.pad_bytes), 0, (int)(p.in_bytes + (unsigned)p.pad_bytes), 0x00020000);

  const h8* wblk = p.wp + (long)mb * p.nchunk * (HOCT * BM) + tid;
  i32x4 areg[NA], breg[NR];
  // Tap-table entry {byte offset, tap bit} of this thread's octet. With Cin_pad % 64 == 0 the 8 octets of a chunk are one
  // tap and 64 consecutive channels: the entry of octet 0 is wave-uniform (scalar load) and the lane adds oct*16 bytes.
  // Otherwise (conv1) every octet is its own tap: a per-lane entry, loaded one chunk ahead and issued BEFORE that chunk's
  // gathers — vmcnt counts in order, so a table load issued after the gathers would make its first use wait for all of them.
  constexpr bool uniform_tap = UT;
  int2 tnext = p.tab[kc_begin * HOCT + (uniform_tap ? 0 : oct)];
  const unsigned lane_off = uniform_tap ? (unsigned)oct * 16u : 0u;

#define LOAD_CHUNK(kc)                                                                                   \
  {                                                                                                      \
    _Pragma("unroll") for (int e = 0; e < NA; ++e)                                                       \
      areg[e] = *reinterpret_cast<const i32x4*>(wblk + (long)(kc) * (HOCT * BM) + e * 256);              \
    int2 t;                                                                                              \
    if (uniform_tap) {                                                                                   \
      t = p.tab[(kc) * HOCT];                              /* wave-uniform: s_load */                    \
    } else {                                                                                             \
      t = tnext;                                                                                         \
      tnext = p.tab[min((kc) + 1, p.nchunk - 1) * HOCT + oct];                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
    }                                                                                                    \
    const unsigned toff = (unsigned)t.x + lane_off;                                                      \
    _Pragma("unroll") for (int r = 0; r < NR; ++r) {                                                     \
      const unsigned inv = (unsigned)(ninv64[r] >> t.y);                                                 \
      /* the tap offset is per lane, so it goes into the VGPR offset (a divergent soffset would be waterfalled) */ \
      breg[r] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((inv << 31) | voff[r]) + toff), 0, 0)); \
    }                                                                                                    \
  }
#define STORE_CHUNK(buf)                                                                                 \
  {                                                                                                      \
    _Pragma("unroll") for (int e = 0; e < NA; ++e)                                                       \
      *reinterpret_cast<i32x4*>(&As[(buf) * HOCT * BM + tid + e * 256]) = areg[e];                       \
    _Pragma("unroll") for (int r = 0; r < NR; ++r) {                                                     \
      const int px = r * 32 + (tid >> 3);                                                                \
      *reinterpret_cast<i32x4*>(&Bs[(buf) * HOCT * BN + px * 8 + (oct ^ ((px >> 1) & 7))]) = breg[r];    \
    }                                                                                                    \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
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
    const h8* as = &As[buf * HOCT * BM + lrow * BM + wm0 + lcol];
    const h8* bs = &Bs[buf * HOCT * BN + (wn0 + lcol) * 8];
    const int sw = (lcol >> 1) & 7;   // == ((pixel >> 1) & 7): wn0 and j*32 are multiples of 16
    // fragments of k-step t+1 are read while the MFMAs of k-step t run (two fragment sets, statically indexed)
    h8 af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = as[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = bs[j * 256 + (lrow ^ sw)];
#pragma unroll
    for (int t = 0; t < HOCT / 2; ++t) {
      if (t + 1 < HOCT / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(t + 1) & 1][i] = as[(t + 1) * 2 * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[(t + 1) & 1][j] = bs[j * 256 + (((t + 1) * 2 + lrow) ^ sw)];
      }
      // the next chunk goes to the other LDS buffer before the last k-step, so its ds_writes overlap these MFMAs
      if (t == HOCT / 2 - 1 && more) STORE_CHUNK(buf ^ 1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][i], bf[t & 1][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#undef LOAD_CHUNK
#undef STORE_CHUNK

  // epilogue: bias + LeakyReLU in fp32, NHWC fp16 store (offset = pixel*Cout + co; 4 channels = 8 bytes per store)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long op = n0 + wn0 + j * 32 + lcol;
    if (op >= p.npix) continue;
    _Float16* orow = p.out + op * p.Cout;
    float* prow = p.partial ? p.partial + ((long)split * p.npix + op) * p.Cout : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co0 = mb * BM + wm0 + i * 32 + 8 * g + 4 * lrow;
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

// ---------------------------------------------------------------------------------------------------------------------
// 256x256 tiles with LDS-DMA and a 4-stage ring (layers with Cout % 256 == 0 and Cin_pad % 64 == 0: conv3 … conv6_1).
// Register staging keeps exactly one K chunk of loads in flight per wave and every chunk then ends in "wait for the loads,
// ds_write, barrier": measured, the kernel above spends about half its time in that wait (MFMAs removed: 20 % faster; loads
// issued but never waited for: as fast as with no loads at all). Here the loads go global → LDS directly
// (`buffer_load_dwordx4 … lds`, 1 KB per wave-instruction, hardware zero fill for padding taps), no staging registers and
// no ds_write pass; a K chunk is 32 (4 octets: 16 KB of weights + 16 KB of activations), the ring holds 4 chunks, so three
// chunks (3072 MFMA cycles) of loads are in flight behind the one being multiplied. The DMA is issued from inline asm:
// through the builtin the compiler cannot tell the ring stages apart and drains vmcnt(0) before every ds_read.
//   weights     stage image [octet 0..3][row 0..255][8 halves]  = 16 KB contiguous in the packed buffer (chunk32 c at c·16 KB)
//   activations stage image [pixel 0..255][slot 0..3][8 halves], slot = octet ^ ((pixel >> 2) & 3): the DMA writes lanes
//               linearly, so the swizzle is applied to WHICH 16 bytes of the pixel's 64-byte run a lane fetches; a
//               ds_read_b128 lane group (16 pixels, one octet) then covers all 64 banks once.
// WGM x WGN waves of 128x128 (4x4 MFMA tiles) each: <2,2> = 256x256 block, 4 ring stages of 32 KB; <1,4> = 128x512 block for
// Cout == 128 (conv2), 3 stages of 40 KB.
template <int WGM, int WGN, int NSTAGE, bool X3 = false, int TN = 4>
__global__ __launch_bounds__(WGM * WGN * 64, TN == 2 ? 2 : 1) void conv_f16_dma_kernel(ConvF16Params p) {
  // TN = 2: 128x64 wave tiles (128 accumulator registers) so that two blocks share a CU, i.e. two waves per SIMD — one wave
  // alone issues MFMAs at 71 % of the pipe's rate (tools/mfma_f16_probe.hip) and nothing covers its waits
  // WGM * WGN = 8 (512 threads, TN = 2): eight 128x64 wave tiles share ONE 256x256 LDS image — two waves per SIMD as with two
  // 4-wave blocks, but a third fewer DMA bytes per MFMA and a 4-stage ring
  constexpr int BM = WGM * 128, BN = WGN * TN * 32, TM = 4, NW = WGM * WGN;
  constexpr int NPA = BM / (16 * NW), NPB = BN / (16 * NW), NP = NPA + NPB;   // 1 KB DMA pieces per wave per chunk: weights, activations
  constexpr int STAGE = (BM + BN) * 4;     // h8 per stage: BM*4 weights + BN*4 activations
  static_assert((NW == 4 || NW == 8) && NPA >= 1 && NPB >= 1 && NSTAGE * STAGE * 16 <= 160 * 1024, "tile shape");
  extern __shared__ __attribute__((aligned(16))) h8 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WGN) * 128, wn0 = (wave % WGN) * (TN * 32);
  int vid;
  {
    // XCD-aware order over the whole tiles; the tail slices keep the dispatch order (they must start last)
    const int total = p.n_full > 0 ? p.n_full : p.gx * p.gy * p.ksplit, bid = blockIdx.x;
    const int xcd = bid & 7, qn = total >> 3, rn = total & 7;
    vid = bid < total ? xcd * qn + min(xcd, rn) + (bid >> 3) : bid;
  }
  int bx, mb, split, c_begin, c_end, tail_slot = -1;
  if (p.n_full > 0 && vid >= p.n_full) {
    const int R = p.gx * p.gy - p.n_full, t = vid - p.n_full;
    const int tile = p.n_full + t % R, ts = t / R;
    bx = tile % p.gx; mb = tile / p.gx; split = 0;
    tail_slot = ts * R + (tile - p.n_full);
    c_begin = ts * p.tail_cps * 2; c_end = min(p.nchunk, (ts + 1) * p.tail_cps) * 2;
  } else {
    bx = vid % p.gx; mb = (vid / p.gx) % p.gy; split = vid / (p.gx * p.gy);
    // chunk32 range of this K slice (chunks_per_split counts 64-wide chunks)
    c_begin = split * p.chunks_per_split * 2; c_end = min(p.nchunk, (split + 1) * p.chunks_per_split) * 2;
  }
  const long n0 = (long)bx * BN;

  // activation gather: piece i of wave w fills pixels (w*NPB+i)*16 .. +15 of the stage, lane = (pixel l>>2, slot l&3)
  unsigned voff[NPB];
  unsigned long long ninv64[NPB];
  unsigned lane_off[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int P = (wave * NPB + i) * 16 + (lane >> 2);
    const long pix = n0 + P;
    voff[i] = 0x80000000u;
    unsigned long long m64 = 0;
    if (pix < p.npix) {
      const int hw = p.Ho * p.Wo;
      const int n = (int)(pix / hw);
      const int rr = (int)(pix - (long)n * hw);
      const int ho = rr / p.Wo, wo = rr - ho * p.Wo;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      voff[i] = (unsigned)(((n * p.H + hi0) * p.W + wi0) * p.Cin * 2 + p.pad_bytes);
      m64 = tap_mask64(hi0, wi0, p.H, p.W);
    }
    ninv64[i] = ~m64;
    lane_off[i] = (unsigned)(((lane & 3) ^ ((P >> 2) & 3)) * 16);   // the octet this lane fetches into its slot
  }
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.in - p.pad_bytes), 0, (int)(p.in_bytes + (unsigned)p.pad_bytes), 0x00020000);
  // weights of this M tile: nchunk 64-wide chunks x 8 octets x BM rows x 16 B, chunk32 c at c*BM*64 bytes
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.wp + (long)mb * p.nchunk * (HOCT * BM)), 0, (int)((long)p.nchunk * HOCT * BM * 16), 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)smem;
  const unsigned w_voff = (unsigned)((wave * NPA) * 1024 + lane * 16);          // + i*1024 + c*BM*64

#define DMA(ldsaddr, voffset, rsrc)                                                                      \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"                 \
               :: "s"(ldsaddr), "v"(voffset), "s"(rsrc) : "memory")
  // Issue stream: chunk32 numbers c_begin, c_begin+1, … in order. K order is (64-channel group, ky, kx, 32-channel half)
  // and the position is kept in scalar counters — a tap-table load inside the loop would be a VECTOR load here (the asm
  // statements clobber memory, so the compiler cannot prove the table invariant) whose wait drains the whole DMA ring.
  const int kw_ = p.stride_kw & 0xffff, kh_ = p.stride_kw >> 16;
  int ic = c_begin;                                   // next chunk32 to issue
  int i_half, i_kx, i_ky, i_cg;
  {
    const int kc = c_begin >> 1, ntaps = kh_ * kw_;
    const int tap = kc % ntaps;
    i_half = c_begin & 1; i_cg = kc / ntaps; i_ky = tap / kw_; i_kx = tap - i_ky * kw_;
  }
  // one chunk's NP DMA pieces are issued one at a time (pieces 0..NPA-1 weights, then activations) so that the main loop
  // can spread them over the chunk's MFMAs: a DMA piece costs ~100 issue cycles next to ds_reads and almost nothing in the
  // shadow of an MFMA, and all of them back to back after the barrier left the matrix pipe idle for most of a chunk
  unsigned i_sbase = 0, i_toff = 0, i_wbase = 0;
  int i_tbit = 0;
  auto issue_begin = [&]() {
    i_sbase = lds0 + (unsigned)((ic - c_begin) % NSTAGE) * (STAGE * 16);
    i_toff = (unsigned)(((i_ky * p.W + i_kx) * p.Cin + i_cg * 64 + i_half * 32) * 2);
    i_tbit = i_ky * 8 + i_kx;
    i_wbase = (unsigned)min(ic, c_end - 1) * (unsigned)(BM * 64);
  };
  auto issue_piece = [&](int q) {
    if (q < NPA) {
      const unsigned la = __builtin_amdgcn_readfirstlane(i_sbase + (unsigned)((wave * NPA + q) * 1024));
      DMA(la, w_voff + (unsigned)q * 1024u + i_wbase, rsrc_w);
    } else {
      const int i = q - NPA;
      const unsigned la = __builtin_amdgcn_readfirstlane(i_sbase + (unsigned)(BM * 64) + (unsigned)((wave * NPB + i) * 1024));
      const unsigned inv = (unsigned)(ninv64[i] >> i_tbit);
      DMA(la, ((inv << 31) | voff[i]) + i_toff + lane_off[i], rsrc_in);
    }
  };
  auto issue_end = [&]() {
    if (ic < c_end - 1) {                             // past the end: keep re-issuing the last chunk (ring slot is free)
      i_half ^= 1;
      if (i_half == 0) {
        if (++i_kx == kw_) {
          i_kx = 0;
          if (++i_ky == kh_) { i_ky = 0; ++i_cg; }
        }
      }
    }
    ++ic;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int pre = 0; pre < NSTAGE - 1; ++pre) {
    issue_begin();
#pragma unroll
    for (int q = 0; q < NP; ++q) issue_piece(q);
    issue_end();
  }
  const int lrow = lane >> 5, lcol = lane & 31;
  const int sw = (lcol >> 2) & 3;
  for (int c = c_begin; c < c_end; ++c) {
    // chunk c has landed once at most the loads of the NSTAGE-2 later chunks are outstanding; then every wave's part is visible
    if (NSTAGE == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP) : "memory");
    __builtin_amdgcn_s_barrier();
    issue_begin();                                  // chunk c+NSTAGE-1 goes into the slot chunk c-1 was read from (all waves are past it)
    const h8* as = smem + ((c - c_begin) % NSTAGE) * STAGE + lrow * BM + wm0 + lcol;
    const h8* bs = smem + ((c - c_begin) % NSTAGE) * STAGE + BM * 4 + (wn0 + lcol) * 4;
    h8 af[2][TM], bf[2][TN];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[t][i] = as[t * 2 * BM + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[t][j] = bs[j * 128 + ((t * 2 + lrow) ^ sw)];
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (X3) {
      // octets 0,1 of the chunk = hi, octets 2,3 = lo of the same 16 real channels: af/bf[0] = hi, [1] = lo fragments.
      // Three passes over the 4x4 tile grid (hi·hi, hi·lo, lo·hi): an accumulator is touched once per 16 MFMAs.
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[term == 2 ? 1 : 0][i], bf[term == 1 ? 1 : 0][j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          constexpr int S12 = 12;                   // 12 issue slots per chunk (one after every 4 MFMAs)
          const int slot = term * 4 + i;
#pragma unroll
          for (int q = slot * NP / S12; q < (slot + 1) * NP / S12; ++q) issue_piece(q);
          __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t][i], bf[t][j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          constexpr int S8 = 8;                       // 8 issue slots per chunk (one after every 4 MFMAs)
          const int slot = t * 4 + i;
#pragma unroll
          for (int q = slot * NP / S8; q < (slot + 1) * NP / S8; ++q) issue_piece(q);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    issue_end();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef DMA

  float amax = 0.f;
  if constexpr (TN == 2) if (tail_slot >= 0 || p.partial) {
    // Raw fp32 partial sums (a tail slice: tile-local [pixel][BM channels]; a split-K slice: [slice][pixel][Cout]) through the
    // same LDS staging as the final tiles below: a pixel's 128 channels of this wave are 512 contiguous bytes
    constexpr int PITCH = 512 + 16;
    __syncthreads();
    char* stage = reinterpret_cast<char*>(smem) + wave * (32 * PITCH);
    float* dst;
    size_t row_floats;
    if (tail_slot >= 0) { dst = p.tail_partial + (long)tail_slot * (BM * BN) + (long)wn0 * BM + wm0; row_floats = BM; }
    else { dst = p.partial + ((long)split * p.npix + n0 + wn0) * p.Cout + mb * BM + wm0; row_floats = p.Cout; }
    const bool co_ok = tail_slot >= 0 || mb * BM + wm0 + 128 <= p.Cout;   // Cout % 128 == 0 on this path except ragged fp16 layers
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(stage + lcol * PITCH + (i * 32 + 8 * g + 4 * lrow) * 4) =
              make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int px = k * 2 + (lane >> 5), chunk = lane & 31;
        const i32x4 d = *reinterpret_cast<const i32x4*>(stage + px * PITCH + chunk * 16);
        const long op = n0 + wn0 + j * 32 + px;
        const bool ok = tail_slot >= 0 || (op < p.npix && (co_ok || mb * BM + wm0 + chunk * 4 < p.Cout));
        if (ok) *reinterpret_cast<i32x4*>(dst + (size_t)(j * 32 + px) * row_floats + chunk * 4) = d;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }
  if (tail_slot >= 0) {   // raw fp32 partial sums of a tail slice, tile-local [pixel][channel]
    float* tp = p.tail_partial + (long)tail_slot * (BM * BN);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(tp + (wn0 + j * 32 + lcol) * BM + wm0 + i * 32 + 8 * g + 4 * lrow) =
              make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
    return;
  }
  if constexpr (TN == 2) if (!p.partial && (p.Cout & 127) == 0) {
    // Final output, whole 128-channel wave tiles: stage the tile through LDS (the ring is free once every wave has left the
    // K loop) so that a wave writes each pixel's run of its 128 channels — 256 B of fp16, 512 B of split16 records — as
    // contiguous 16-byte pieces, 1 KB per store instruction. The direct form below (8-byte quads, 64 different cache lines
    // per instruction) cost 10-12 % of the kernel (measured with the stores removed).
    constexpr int RUN = X3 ? 512 : 256, PITCH = RUN + 16, CPP = RUN / 16, PPI = 64 / CPP;
    static_assert(4 * 32 * PITCH <= NSTAGE * STAGE * 16 || NW != 4, "staging area");
    __syncthreads();
    char* stage = reinterpret_cast<char*>(smem) + wave * (32 * PITCH);
    const int cw0 = mb * BM + wm0;                                     // first channel of this wave's tile
    const size_t rec_bytes = (size_t)p.Cout * (X3 ? 4 : 2);            // bytes per output pixel
    const size_t run_off = X3 ? (size_t)(cw0 >> 4) * 64 : (size_t)cw0 * 2;
    // the wave tile's 16 bias quads in ONE batch of loads, before the tile is staged: loaded quad by quad inside the loop below
    // (where every quad is fenced off to keep the register count down) each load's L2 latency was paid in sequence — 64 dependent
    // round trips, ~10 of a tile's ~15 thousand epilogue cycles (profiles/r04_fp16_pingpong.md)
    float4 bias_r[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bias_r[i][g] = p.bias ? *reinterpret_cast<const float4*>(p.bias + cw0 + i * 32 + 8 * g + 4 * lrow) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cl = i * 32 + 8 * g + 4 * lrow;                    // channel within the wave tile
          const float bias_q[4] = {bias_r[i][g].x, bias_r[i][g].y, bias_r[i][g].z, bias_r[i][g].w};
          h4 vh, vl;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = (X3 ? acc[i][j][4 * g + r] * p.acc_scale : acc[i][j][4 * g + r]) + bias_q[r];
            x = x > 0.f ? x : x * p.slope;
            if constexpr (X3) { const X3Pair s2 = x3_split(x, p.out_scale, amax); vh[r] = s2.hi; vl[r] = s2.lo; }
            else vh[r] = (_Float16)x;
          }
          if constexpr (X3) {
            char* rec = stage + lcol * PITCH + (cl >> 4) * 64 + (cl & 15) * 2;
            *reinterpret_cast<h4*>(rec) = vh;
            *reinterpret_cast<h4*>(rec + 32) = vl;
          } else {
            *reinterpret_cast<h4*>(stage + lcol * PITCH + cl * 2) = vh;
          }
          __builtin_amdgcn_sched_barrier(0);   // one quad at a time: hoisting all 32 quads' arithmetic ahead of the writes spills
        }
      // the wave reads back its own writes (LDS operations of a wave complete in order)
#pragma unroll
      for (int k = 0; k < 32 / PPI; ++k) {
        const int px = k * PPI + lane / CPP, chunk = lane % CPP;
        const i32x4 d = *reinterpret_cast<const i32x4*>(stage + px * PITCH + chunk * 16);
        const long op = n0 + wn0 + j * 32 + px;
        if (op < p.npix)
          *reinterpret_cast<i32x4*>(reinterpret_cast<char*>(p.out) + (size_t)op * rec_bytes + run_off + chunk * 16) = d;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (X3) x3_report(amax, p.status);
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long op = n0 + wn0 + j * 32 + lcol;
    if (op >= p.npix) continue;
    _Float16* orow = p.out + op * p.Cout;
    float* prow = p.partial ? p.partial + ((long)split * p.npix + op) * p.Cout : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co0 = mb * BM + wm0 + i * 32 + 8 * g + 4 * lrow;
        if (co0 < p.Cout && prow) {
          *reinterpret_cast<float4*>(prow + co0) =
              make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        } else if (co0 < p.Cout && X3) {   // split16 output: record of 16 channels = [hi 16 | lo 16]
          h4 vh, vl;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[i][j][4 * g + r] * p.acc_scale + (p.bias ? p.bias[co0 + r] : 0.f);
            x = x > 0.f ? x : x * p.slope;
            { const X3Pair s2 = x3_split(x, p.out_scale, amax); vh[r] = s2.hi; vl[r] = s2.lo; }
          }
          _Float16* rec = p.out + op * (2 * p.Cout) + (co0 >> 4) * 32 + (co0 & 15);
          *reinterpret_cast<h4*>(rec) = vh;
          *reinterpret_cast<h4*>(rec + 16) = vl;
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
  if (X3) x3_report(amax, p.status);
}

// ---------------------------------------------------------------------------------------------------------------------
// Ping-pong kernel (round 4): ONE 8-wave block per CU on a 256x256 (Cout % 256 == 0) or 128x512 (conv2) tile, the two waves of
// every SIMD alternating between a MEMORY segment (the 12 fragment reads of one 32-wide K stage + all of a stage's LDS-DMA
// pieces + the counted wait) and a MATRIX segment (that stage's 16 MFMAs under s_setprio 1), separated by raw s_barriers — the
// phase structure of the CDNA4 guide's 8-wave kernels (cdna_hip_programming.md §5 "256² 8-phase template", MI355X_MICROARCH.md
// "Two waves per SIMD") laid over this convolution's tap-ordered K loop. Waves 4-7 run one barrier behind waves 0-3, so that on
// every SIMD one wave multiplies while its partner reads and issues (the 4-wave kernel above reads, waits and multiplies in one
// wave: matrix pipe busy 40-54 %, profiles/r02_pmc_fp16_dma.md).
//   stage  = one 32-wide K chunk: weights [octet 0..3][row][8 halves] + activations [pixel][4 slots][8] — the images of the
//            kernel above, so packed weights and the source-side XOR swizzle are shared.
//   ring   = NSTAGE stages, LA = NSTAGE − 2 issued ahead: phase P reads stage P, issues stage P + LA into the slot of stage
//            P − 2 — last read TWO phases ago: the partner half reads one barrier later, and only the barrier after that orders
//            its reads before a DMA (the guide's "restage >= 2 phases after the last ds_read") — and waits, before its first
//            barrier, until at most (LA − 1)·NP pieces are outstanding: stage P + 1 has landed and is read one phase later, as
//            the guide requires for data another wave's DMA wrote. Never vmcnt(0) inside the loop.
//   memory segment stripped to ds_reads + DMA (a first form with a k16-step per phase and the tap counters / address arithmetic
//            inside the segment measured 220-390 cycles per segment against the partner's 296 — one wave issues an instruction
//            only every ~5 cycles — profiles/r04_fp16_pingpong.md): everything the NEXT phase's pieces need (ring slot, tap
//            offset and validity shift from the per-chunk tap table by ONE scalar load, M0 values, soffsets, the activation
//            pieces' voffsets with the padding bit, the fragment read bases) is computed inside the matrix segment, between the
//            MFMAs, where issue slots are free; the pieces of one operand share one M0 write and differ by the instruction's
//            immediate offset, which advances the LDS and the memory address together (the activation voffsets are pre-biased by
//            −i·1024).
//   fp16 only (the x3 epilogue staging does not fit beside the ring); final / partial tiles leave through the LDS-staged
//   epilogues of the kernel above. Same products in the same order as the 4-wave kernel: bit-identical results (tests).
// Dev builds only (tools/build_variants_f16.sh → variants/lib_<name>.so, never the shipped library): ablation bits and a timeline
// of the ping-pong kernel. DI_PP_ABL: 1 = no DMA issue inside the loop, 8 = no fragment reads, 16 = no MFMAs (results are wrong by
// construction: timing only), 64 = pieces issued before the fragment reads. DI_PP_TRACE: the eight waves of tile 0 stamp s_memtime
// at kernel entry, loop begin / end, kernel end and at the start of every phase (one stamp per phase, written to LDS one phase
// later: no extra waits) and dump them to the buffer registered with deepim_dev_pp_trace() (tools/pp_trace.py).
#ifndef DI_PP_ABL
#define DI_PP_ABL 0
#endif
#ifndef DI_PP_TRACE
#define DI_PP_TRACE 0
#endif
#if DI_PP_TRACE
__device__ unsigned long long* g_pp_trace = nullptr;
#endif

template <int WGM, int WGN, int NSTAGE>
__global__ __launch_bounds__(512, 2) void conv_f16_pp_kernel(ConvF16Params p) {
#if DI_PP_TRACE
  const unsigned long long tr_t0 = __builtin_amdgcn_s_memtime();
#endif
  constexpr int TM = 4, TN = 2, NW = 8, LA = NSTAGE - 2;
  constexpr int BM = WGM * 128, BN = WGN * 64;
  constexpr int NPA = BM / (16 * NW), NPB = BN / (16 * NW), NP = NPA + NPB;
  constexpr int STAGE = (BM + BN) * 4;     // h8 per stage
  constexpr unsigned STAGE_B = STAGE * 16u;
  constexpr unsigned BIAS = 4096u;         // descriptor base shifted down so that the pre-biased activation voffsets stay >= 0
  static_assert(WGM * WGN == NW && NPA >= 1 && NPB >= 1 && NPA <= 4 && NPB <= 4 && NSTAGE >= 4 && NSTAGE * STAGE * 16 <= 160 * 1024, "tile shape");
  extern __shared__ __attribute__((aligned(16))) h8 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WGN) * 128, wn0 = (wave % WGN) * 64;
  int vid;
  {
    const int total = p.n_full > 0 ? p.n_full : p.gx * p.gy * p.ksplit, bid = blockIdx.x;
    const int xcd = bid & 7, qn = total >> 3, rn = total & 7;
    vid = bid < total ? xcd * qn + min(xcd, rn) + (bid >> 3) : bid;
  }
  int bx, mb, split, c_begin, c_end, tail_slot = -1;
  if (p.n_full > 0 && vid >= p.n_full) {
    const int R = p.gx * p.gy - p.n_full, t = vid - p.n_full;
    const int tile = p.n_full + t % R, ts = t / R;
    bx = tile % p.gx; mb = tile / p.gx; split = 0;
    tail_slot = ts * R + (tile - p.n_full);
    c_begin = ts * p.tail_cps * 2; c_end = min(p.nchunk, (ts + 1) * p.tail_cps) * 2;
  } else {
    bx = vid % p.gx; mb = (vid / p.gx) % p.gy; split = vid / (p.gx * p.gy);
    c_begin = split * p.chunks_per_split * 2; c_end = min(p.nchunk, (split + 1) * p.chunks_per_split) * 2;
  }
  const long n0 = (long)bx * BN;

  // activation pieces: piece i of wave w fills pixels (w·NPB + i)·16 … +15 of a stage, lane = (pixel l >> 2, slot l & 3)
  unsigned voffl[NPB];                 // pixel origin + this lane's (swizzled) 16-byte slot − i·1024 + BIAS
  unsigned long long ninv64[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int P = (wave * NPB + i) * 16 + (lane >> 2);
    const long pix = n0 + P;
    voffl[i] = 0x80000000u;
    unsigned long long m64 = 0;
    if (pix < p.npix) {
      const int hw = p.Ho * p.Wo;
      const int n = (int)(pix / hw);
      const int rr = (int)(pix - (long)n * hw);
      const int ho = rr / p.Wo, wo = rr - ho * p.Wo;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      voffl[i] = (unsigned)(((n * p.H + hi0) * p.W + wi0) * p.Cin * 2 + p.pad_bytes) + (unsigned)(((lane & 3) ^ ((P >> 2) & 3)) * 16) +
                 BIAS - (unsigned)i * 1024u;
      m64 = tap_mask64(hi0, wi0, p.H, p.W);
    }
    ninv64[i] = ~m64;
  }
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.in - p.pad_bytes - BIAS), 0, (int)(p.in_bytes + (unsigned)p.pad_bytes + BIAS), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.wp + (long)mb * p.nchunk * (HOCT * BM)), 0, (int)((long)p.nchunk * HOCT * BM * 16), 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)smem;
  const unsigned w_voff = (unsigned)((wave * NPA) * 1024 + lane * 16);
  const unsigned ldsA = lds0 + (unsigned)(wave * NPA) * 1024u, ldsB = lds0 + (unsigned)(BM * 64) + (unsigned)(wave * NPB) * 1024u;

  // all pieces of one stage: one M0 write per operand, the instruction's immediate offset steps LDS and memory address together
  auto issue_stage = [&](unsigned m0a, unsigned soffa, unsigned m0b, unsigned soffb, const unsigned (&nv)[NPB]) {
    if constexpr (NPA == 1)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                   :: "s"(m0a), "v"(w_voff), "s"(rsrc_w), "s"(soffa) : "memory");
    else
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                   "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds"
                   :: "s"(m0a), "v"(w_voff), "s"(rsrc_w), "s"(soffa) : "memory");
    if constexpr (NPB == 2)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                   "buffer_load_dwordx4 %2, %3, %4 offen offset:1024 lds"
                   :: "s"(m0b), "v"(nv[0]), "v"(nv[1]), "s"(rsrc_in), "s"(soffb) : "memory");
    else
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %6 offen lds\n\t"
                   "buffer_load_dwordx4 %2, %5, %6 offen offset:1024 lds\n\t"
                   "buffer_load_dwordx4 %3, %5, %6 offen offset:2048 lds\n\t"
                   "buffer_load_dwordx4 %4, %5, %6 offen offset:3072 lds"
                   :: "s"(m0b), "v"(nv[0]), "v"(nv[1]), "v"(nv[NPB == 4 ? 2 : 0]), "v"(nv[NPB == 4 ? 3 : 0]), "s"(rsrc_in), "s"(soffb) : "memory");
  };
  // the scalars of stage `ic` (chunk32 index; past the end the last chunk is re-issued into a free slot): ring slot, weight
  // soffset, tap table entry {byte offset of (tap, 32-channel half), validity bit}
  // The table entry comes by a SCALAR load issued from inline asm: a compiler-visible load here would be a vector load (the DMA
  // statements clobber memory, so the table is not provably invariant) whose compiler-placed wait is vmcnt(0) — draining the
  // whole DMA ring every phase (cdna_hip_programming.md §5 ".s-level traps" (b)). tab_issue → (≥ one s_waitcnt lgkmcnt(0)
  // later, stated on the register pair) → stage_prep.
  const int2* tab_ptr = p.tab;
  auto tab_issue = [&](int ic, unsigned long long& te) {
    const unsigned off = (unsigned)min(ic, c_end - 1) * 32u;                          // tab[cc * 4], 8-byte entries
    asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(te) : "s"(tab_ptr), "s"(off) : "memory");
  };
  auto tab_wait = [&](unsigned long long& te) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(te) :: "memory"); };
  auto stage_prep = [&](int ic, unsigned long long te, unsigned& m0a, unsigned& soffa, unsigned& m0b, unsigned& soffb, unsigned (&nv)[NPB]) {
    const int cc = min(ic, c_end - 1);
    const unsigned slot = (unsigned)(ic - c_begin) % (unsigned)NSTAGE;
    m0a = ldsA + slot * STAGE_B; m0b = ldsB + slot * STAGE_B;
    soffa = (unsigned)cc * (unsigned)(BM * 64);
    soffb = (unsigned)(te & 0xffffffffull);
    const int tbit = (int)(te >> 32);
#pragma unroll
    for (int i = 0; i < NPB; ++i) nv[i] = ((unsigned)(ninv64[i] >> tbit) << 31) | voffl[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  unsigned m0a, soffa, m0b, soffb, nv[NPB];
  unsigned long long te;
  // prologue: stages 0 … LA−1 (what phases −LA … −1 of the steady state would have issued); stage 0 must have landed
#pragma unroll
  for (int pre = 0; pre < LA; ++pre) {
    tab_issue(c_begin + pre, te);
    tab_wait(te);
    stage_prep(c_begin + pre, te, m0a, soffa, m0b, soffb, nv);
    issue_stage(m0a, soffa, m0b, soffb, nv);
  }
  tab_issue(c_begin + LA, te);
  tab_wait(te);
  stage_prep(c_begin + LA, te, m0a, soffa, m0b, soffb, nv);            // what phase 0 issues
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"((LA - 1) * NP) : "memory");
  asm volatile("s_barrier" ::: "memory");
  if (wave >= 4) asm volatile("s_barrier" ::: "memory");               // the younger half runs one barrier behind

  const int lrow = lane >> 5, lcol = lane & 31;
  const int sw = (lcol >> 2) & 3;
  const int nst = c_end - c_begin;
  // fragment read bases of the current phase's slot (advanced inside the matrix segment for the next phase)
  const unsigned a_lane = (unsigned)(lrow * BM + wm0 + lcol) * 16u;
  const unsigned b_lane0 = (unsigned)(BM * 4 + (wn0 + lcol) * 4 + (lrow ^ sw)) * 16u;
  const unsigned b_lane1 = (unsigned)(BM * 4 + (wn0 + lcol) * 4 + ((2 + lrow) ^ sw)) * 16u;
  const char* lds_c = reinterpret_cast<const char*>(smem);
  unsigned slot_off = 0;
  unsigned rd_a = a_lane, rd_b0 = b_lane0, rd_b1 = b_lane1;
#if DI_PP_TRACE
  // dev timeline, dumped straight to global memory after the loop (global stores inside would disturb the vmcnt counting):
  // [wave][0 entry, 1 loop begin, 2 loop end, 3 kernel end, 8 + s: start of phase s] — phase stamps kept in LDS behind the ring
  unsigned long long* trl = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(smem) + 135168);   // NSTAGE = 4 builds only
  const bool tracing = vid == 0 && g_pp_trace != nullptr && NSTAGE == 4;
  const unsigned long long tr_t1 = __builtin_amdgcn_s_memtime();
  unsigned long long tr_a = 0;
#endif
  // values computed inside the matrix segment for the next memory segment are pinned there: left alone, the compiler sinks the
  // arithmetic to its use — back into the memory segment, where every instruction lengthens the hand-over
#define PP2_PIN(x) asm volatile("" : "+v"(x))
  for (int s = 0; s < nst; ++s) {
    // ---- memory segment: 12 fragment reads of stage s, the NP pieces of stage s + LA, the counted wait
    const char* as = lds_c + rd_a;
    const char* bs0 = lds_c + rd_b0;
    const char* bs1 = lds_c + rd_b1;
    h8 af[2][TM], bf[2][TN];
#if DI_PP_TRACE
    if (tracing) tr_a = __builtin_amdgcn_s_memtime();        // waited for by the lgkmcnt(0) behind the barrier
#endif
#if DI_PP_ABL & 64
    issue_stage(m0a, soffa, m0b, soffb, nv);      // dev: pieces first, reads second
    __builtin_amdgcn_sched_barrier(0);
#endif
#if !(DI_PP_ABL & 8)
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const h8*>(as + i * 512);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const h8*>(bs0 + j * 2048);
#pragma unroll
    for (int i = 0; i < TM; ++i) af[1][i] = *reinterpret_cast<const h8*>(as + 2 * BM * 16 + i * 512);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[1][j] = *reinterpret_cast<const h8*>(bs1 + j * 2048);
#else
#pragma unroll
    for (int i = 0; i < TM; ++i) { af[0][i] = af[1][i] = *reinterpret_cast<const h8*>(lds_c + a_lane + i * 512); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { bf[0][j] = bf[1][j] = *reinterpret_cast<const h8*>(lds_c + b_lane0 + j * 2048); }
#endif
    __builtin_amdgcn_sched_barrier(0);
#if !(DI_PP_ABL & 65)
    issue_stage(m0a, soffa, m0b, soffb, nv);
#endif
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((LA - 1) * NP) : "memory");
    asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- matrix segment: 16 MFMAs; between them the scalars and voffsets of the next phase's pieces and read bases
    __builtin_amdgcn_s_setprio(1);
#if DI_PP_TRACE
    if (tracing && lane == 0 && s < 120) trl[wave * 128 + 8 + s] = tr_a;
#endif
    tab_issue(c_begin + s + 1 + LA, te);          // lands under the first MFMAs (no LDS operation is outstanding here)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (DI_PP_ABL & 16) asm volatile("" :: "v"(af[0][i]), "v"(bf[0][j]), "v"(af[1][i]), "v"(bf[1][j]));
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);            // the table entry is waited for behind the first eight MFMAs, not before them
    tab_wait(te);
    stage_prep(c_begin + s + 1 + LA, te, m0a, soffa, m0b, soffb, nv);
    slot_off = slot_off + STAGE_B == (unsigned)NSTAGE * STAGE_B ? 0u : slot_off + STAGE_B;
    rd_a = slot_off + a_lane; rd_b0 = slot_off + b_lane0; rd_b1 = slot_off + b_lane1;
    PP2_PIN(rd_a); PP2_PIN(rd_b0); PP2_PIN(rd_b1);
#pragma unroll
    for (int i = 0; i < NPB; ++i) PP2_PIN(nv[i]);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        if (!(DI_PP_ABL & 16)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
    // the ~15 scalar / vector instructions above go between the second eight MFMAs, two or three per issue gap
#pragma unroll
    for (int g_ = 0; g_ < 8; ++g_) {
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x4, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
  }
#undef PP2_PIN
#if DI_PP_TRACE
  const unsigned long long tr_t2 = __builtin_amdgcn_s_memtime();
#endif
  if (wave < 4) asm volatile("s_barrier" ::: "memory");                // the older half catches the barrier count up
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if DI_PP_TRACE
  if (tracing) {
    __syncthreads();
    if (lane == 0) { trl[wave * 128 + 0] = tr_t0; trl[wave * 128 + 1] = tr_t1; trl[wave * 128 + 2] = tr_t2; }
    __syncthreads();
    for (int i = tid; i < 8 * 128; i += 512) if ((i & 127) != 3) g_pp_trace[i] = trl[i];
    __syncthreads();
  }
#endif

  if (tail_slot >= 0 || p.partial) {
    // raw fp32 partial sums through LDS staging (see conv_f16_dma_kernel): a pixel's 128 channels of this wave = 512 contiguous bytes
    constexpr int PITCH = 512 + 16;
    __syncthreads();
    char* stage = reinterpret_cast<char*>(smem) + wave * (32 * PITCH);
    float* dst;
    size_t row_floats;
    if (tail_slot >= 0) { dst = p.tail_partial + (long)tail_slot * (BM * BN) + (long)wn0 * BM + wm0; row_floats = BM; }
    else { dst = p.partial + ((long)split * p.npix + n0 + wn0) * p.Cout + mb * BM + wm0; row_floats = p.Cout; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(stage + lcol * PITCH + (i * 32 + 8 * g + 4 * lrow) * 4) =
              make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int px = k * 2 + (lane >> 5), chunk = lane & 31;
        const i32x4 d = *reinterpret_cast<const i32x4*>(stage + px * PITCH + chunk * 16);
        const long op = n0 + wn0 + j * 32 + px;
        const bool ok = tail_slot >= 0 || op < p.npix;          // Cout % 128 == 0 on this kernel
        if (ok) *reinterpret_cast<i32x4*>(dst + (size_t)(j * 32 + px) * row_floats + chunk * 4) = d;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }
  {
    constexpr int RUN = 256, PITCH = RUN + 16, CPP = RUN / 16, PPI = 64 / CPP;
    __syncthreads();
    char* stage = reinterpret_cast<char*>(smem) + wave * (32 * PITCH);
    const int cw0 = mb * BM + wm0;
    const size_t rec_bytes = (size_t)p.Cout * 2;
    const size_t run_off = (size_t)cw0 * 2;
    // the wave tile's 16 bias quads in ONE batch of loads, before the tile is staged: loaded quad by quad inside the loop below
    // (where every quad is fenced off to keep the register count down) each load's L2 latency was paid in sequence — 64 dependent
    // round trips, ~10 of a tile's ~15 thousand epilogue cycles (profiles/r04_fp16_pingpong.md)
    float4 bias_r[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bias_r[i][g] = p.bias ? *reinterpret_cast<const float4*>(p.bias + cw0 + i * 32 + 8 * g + 4 * lrow) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cl = i * 32 + 8 * g + 4 * lrow;
          const float bias_q[4] = {bias_r[i][g].x, bias_r[i][g].y, bias_r[i][g].z, bias_r[i][g].w};
          h4 vh;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[i][j][4 * g + r] + bias_q[r];
            x = x > 0.f ? x : x * p.slope;
            vh[r] = (_Float16)x;
          }
          *reinterpret_cast<h4*>(stage + lcol * PITCH + cl * 2) = vh;
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int k = 0; k < 32 / PPI; ++k) {
        const int px = k * PPI + lane / CPP, chunk = lane % CPP;
        const i32x4 d = *reinterpret_cast<const i32x4*>(stage + px * PITCH + chunk * 16);
        const long op = n0 + wn0 + j * 32 + px;
        if (op < p.npix)
          *reinterpret_cast<i32x4*>(reinterpret_cast<char*>(p.out) + (size_t)op * rec_bytes + run_off + chunk * 16) = d;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#if DI_PP_TRACE
  if (tracing) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) g_pp_trace[wave * 128 + 3] = __builtin_amdgcn_s_memtime();
  }
#endif
}

// X3 split-K second pass: Σ_s partial (fixed order) → real units → bias → LeakyReLU → split16 record
__global__ __launch_bounds__(256) void splitk_x3_reduce_kernel(_Float16* __restrict__ out, const float* __restrict__ partial,
                                                               const float* __restrict__ bias, long total4, int S, int Cout,
                                                               float slope, float acc_scale, float out_scale, int* status) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const float4* p4 = reinterpret_cast<const float4*>(partial);
  float4 v = p4[i];
#pragma unroll 4
  for (int s = 1; s < S; ++s) {
    const float4 u = p4[(long)s * total4 + i];
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  const long e = i * 4;
  const long pix = e / Cout;
  const int c0 = (int)(e - pix * Cout);
  float r[4] = {v.x, v.y, v.z, v.w};
  h4 vh, vl;
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float x = r[k] * acc_scale + (bias ? bias[c0 + k] : 0.f);
    x = x > 0.f ? x : x * slope;
    { const X3Pair s2 = x3_split(x, out_scale, amax); vh[k] = s2.hi; vl[k] = s2.lo; }
  }
  x3_report(amax, status);
  _Float16* rec = out + pix * (2 * Cout) + (c0 >> 4) * 32 + (c0 & 15);
  *reinterpret_cast<h4*>(rec) = vh;
  *reinterpret_cast<h4*>(rec + 16) = vl;
}

// tail-split second pass: one thread per (remainder tile, pixel, 4 channels); slices added in order, then the layer epilogue
template <bool X3>
__global__ __launch_bounds__(256) void tail_f16_reduce_kernel(ConvF16Params p, int BM, int BN) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int R = p.gx * p.gy - p.n_full, q4 = BM >> 2;
  const long per_tile = (long)BN * q4;
  if (i >= R * per_tile) return;
  const int rt = (int)(i / per_tile);
  const long e = i - rt * per_tile;
  const int pl = (int)(e / q4), c4 = (int)(e - (long)pl * q4);
  const int tile = p.n_full + rt, bx = tile % p.gx, mb = tile / p.gx;
  const long op = (long)bx * BN + pl;
  const int co0 = mb * BM + c4 * 4;
  if (op >= p.npix || co0 >= p.Cout) return;
  const float* tp = p.tail_partial + ((long)rt * BN + pl) * BM + c4 * 4;
  float4 v = *reinterpret_cast<const float4*>(tp);
  for (int s = 1; s < p.tail_s; ++s) {
    const float4 u = *reinterpret_cast<const float4*>(tp + (long)s * R * BM * BN);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  float r[4] = {v.x, v.y, v.z, v.w};
  h4 vh, vl;
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float x = (X3 ? r[k] * p.acc_scale : r[k]) + (p.bias ? p.bias[co0 + k] : 0.f);
    x = x > 0.f ? x : x * p.slope;
    if (X3) { const X3Pair s2 = x3_split(x, p.out_scale, amax); vh[k] = s2.hi; vl[k] = s2.lo; }
    else vh[k] = (_Float16)x;
  }
  if (X3) {
    _Float16* rec = p.out + op * (2 * p.Cout) + (co0 >> 4) * 32 + (co0 & 15);
    *reinterpret_cast<h4*>(rec) = vh;
    *reinterpret_cast<h4*>(rec + 16) = vl;
    x3_report(amax, p.status);
  } else {
    *reinterpret_cast<h4*>(p.out + op * p.Cout + co0) = vh;
  }
}

// split-K second pass: out[pix][c] = f16(lrelu(Σ_s partial[s][pix][c] + bias[c])), slices added in order; 4 channels per thread
// (dwordx4 loads of every slice, one 8-byte store); Cout % 4 == 0
__global__ __launch_bounds__(256) void splitk_f16_reduce_kernel(_Float16* __restrict__ out, const float* __restrict__ partial,
                                                                const float* __restrict__ bias, long total4, int S,
                                                                int Cout, float slope) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const float4* p4 = reinterpret_cast<const float4*>(partial);
  float4 v = p4[i];
#pragma unroll 4
  for (int s = 1; s < S; ++s) {
    const float4 u = p4[(long)s * total4 + i];
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  const int c0 = (int)((i * 4) % Cout);
  float r[4] = {v.x, v.y, v.z, v.w};
  h4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float x = r[k] + (bias ? bias[c0 + k] : 0.f);
    x = x > 0.f ? x : x * slope;
    o[k] = (_Float16)x;
  }
  *reinterpret_cast<h4*>(out + i * 4) = o;
}

// K order. A k-octet q is 8 consecutive input channels of one tap. With Cin_pad a multiple of 64 the octets run
// (channel group of 64, tap, octet in group): a K chunk is one tap of one 64-channel group — 128 contiguous bytes per
// pixel — and the kh·kw chunks of a group re-read the same haloed input region back to back, so the taps' re-reads hit in
// L2 (tap-major order swept the whole Cin·2-byte pixel records once per tap: 9-25x the input from HBM; measured
// 6.6 TB/s of loads, the bound of the first version). Otherwise (conv1: Cin_pad 8/16) octets run (tap, octet).
__host__ __device__ inline void f16_octet(int q, int Cin_pad, int ntaps, int* tap, int* ci0) {
  const int opt = Cin_pad / 8;
  if ((opt & 7) == 0) {
    const int o = q & 7, g = q >> 3;           // g = cg * ntaps + tap
    *tap = g % ntaps;
    *ci0 = ((g / ntaps) * 8 + o) * 8;
  } else {
    *tap = q / opt;
    *ci0 = (q % opt) * 8;
  }
}

// packed[mt][kc][o][m][h] = f16(w[mt*BM+m][ci0+h][ky][kx]) for k-octet q = kc*8+o (order: f16_octet)
__global__ void pack_f16_kernel(_Float16* __restrict__ packed, const float* __restrict__ w, int Cout, int Cin,
                                int Cin_pad, int kh, int kw, int nchunk, int BM, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int h = (int)(i & 7);
  const long r1 = i >> 3;
  const int m = (int)(r1 % BM);
  const long r2 = r1 / BM;
  const int o = (int)(r2 & 7);
  const long r3 = r2 >> 3;
  const int kc = (int)(r3 % nchunk);
  const int mt = (int)(r3 / nchunk);
  int tap, ci0;
  f16_octet(kc * HOCT + o, Cin_pad, kh * kw, &tap, &ci0);
  const int ci = ci0 + h, co = mt * BM + m;
  float v = 0.f;
  if (co < Cout && ci < Cin && tap < kh * kw) v = w[(((long)co * Cin + ci) * kh + tap / kw) * kw + tap % kw];
  packed[i] = (_Float16)v;
}

__global__ void build_f16_tab_kernel(int2* __restrict__ tab, int noct, int noct_pad, int Cin_pad, int kh, int kw, int W) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= noct_pad) return;
  if (q >= noct) { tab[q] = make_int2(0, 63); return; }
  int tap, ci0;
  f16_octet(q, Cin_pad, kh * kw, &tap, &ci0);
  const int ky = tap / kw, kx = tap % kw;
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

// X3 weights: the packed layout of pack_f16_kernel over the 2·Cin virtual channels [hi 16 | lo 16] of w·w_scale
__global__ void pack_x3_kernel(_Float16* __restrict__ packed, const float* __restrict__ w, int Cout, int Cin, int kh, int kw,
                               int nchunk, int BM, float w_scale, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int h = (int)(i & 7);
  const long r1 = i >> 3;
  const int m = (int)(r1 % BM);
  const long r2 = r1 / BM;
  const int o = (int)(r2 & 7);
  const long r3 = r2 >> 3;
  const int kc = (int)(r3 % nchunk);
  const int mt = (int)(r3 / nchunk);
  int tap, cv0;
  f16_octet(kc * HOCT + o, 2 * Cin, kh * kw, &tap, &cv0);
  const int cv = cv0 + h, co = mt * BM + m;
  const int ci = (cv >> 5) * 16 + (cv & 15);
  X3Pair s2 = {(_Float16)0.f, (_Float16)0.f};
  if (co < Cout && ci < Cin && tap < kh * kw) s2 = x3_split(w[(((long)co * Cin + ci) * kh + tap / kw) * kw + tap % kw], w_scale);
  packed[i] = (cv & 16) ? s2.lo : s2.hi;
}

// NCHW fp32 → split16 NHWC; one thread per (pixel, 4 channels): two 8-byte stores
__global__ __launch_bounds__(256) void nchw_to_split16_kernel(_Float16* __restrict__ out, const float* __restrict__ in, int C,
                                                              long hw, float scale, long total, int* status) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int q4 = C >> 2;
  const int cq = (int)(i % q4);
  const long pix = i / q4;
  const long n = pix / hw, r = pix - n * hw;
  h4 vh, vl;
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const X3Pair s2 = x3_split(in[(n * C + cq * 4 + k) * hw + r], scale, amax); vh[k] = s2.hi; vl[k] = s2.lo; }
  const int c0 = cq * 4;
  _Float16* rec = out + pix * (2 * C) + (c0 >> 4) * 32 + (c0 & 15);
  *reinterpret_cast<h4*>(rec) = vh;
  *reinterpret_cast<h4*>(rec + 16) = vl;
  x3_report(amax, status);
}

// split16 NHWC → NCHW fp32 (hi + lo, back to real units); lanes run along pixels of one channel
__global__ __launch_bounds__(256) void split16_to_nchw_kernel(float* __restrict__ out, const _Float16* __restrict__ in, int C,
                                                              long hw, float inv_scale, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long r = i % hw;
  const int c = (int)((i / hw) % C);
  const long n = i / (hw * C);
  const _Float16* rec = in + (n * hw + r) * (2 * C) + (c >> 4) * 32 + (c & 15);
  out[i] = ((float)rec[0] + (float)rec[16]) * inv_scale;
}

// ---------------------------------------------------------------------------------------------------------------------
// conv1 of the split-fp16 encoder: Cin = 8, 7x7, stride 2, pad 3, Cout = 64 (deepIM_flownet.py:63-67) straight from the NCHW
// fp32 net input. With 8 input channels a 16-wide MFMA k-step holds TWO taps (lanes 0-31 tap t, lanes 32-63 tap t+1), so the
// 49 taps are 25 k-steps and the three split products 75 MFMAs per 32x32 output tile.
//   * persistent blocks (one per CU), the packed weights of all taps stay in LDS (100 KB: w_hi / w_lo [tap][co][8 halves]);
//   * a block tile is 8 output rows x 32 columns of one image; its 21 x 69 input patch is loaded as fp32 (coalesced along
//     rows, zero outside the frame), split into (hi, lo) and stored in LDS de-interleaved by column parity
//     [row][parity][35][8 halves] — stride 2 then makes the 32 pixels of a fragment 512 contiguous bytes (conflict-free
//     ds_read_b128), for every tap; the next tile's patch is prefetched into registers while the current one is multiplied;
//   * wave w owns output rows 2w, 2w+1 (two 32-pixel fragments) x all 64 channels: 8 ds_read_b128 per 12 MFMAs;
//   * epilogue: accumulator → real units, bias, LeakyReLU, split16 records (NHWC) for conv2.
constexpr int C1_TAPS = 49, C1_PAIRS = 25, C1_ROWS = 21, C1_RW = 35;
constexpr int C1_WH = 50 * 64;                 // h8 entries of w_hi (and of w_lo)
constexpr int C1_PH = C1_ROWS * 2 * C1_RW;     // h8 entries of a_hi (and of a_lo)
constexpr int C1_LDS = (2 * C1_WH + 2 * C1_PH) * 16;
constexpr int C1_LDS_F16 = (C1_WH + C1_PH) * 16;   // plain fp16 variant: hi parts only (73.7 KB → two blocks per CU)

constexpr int C1_DSTAGE = C1_ROWS * 72 * 4;        // DEP: the two extra channels of the patch as [row][72 quad columns][2 halves]
constexpr int C1_LDS_DEP = C1_LDS_F16 + C1_DSTAGE + 256;  // 81 024 B (+ the 64 biases): still two blocks per CU
constexpr int C1_DEP_TAPS = 14;                     // pseudo taps (ky, kx0 in {0, 4}): 4 consecutive columns x 2 channels each

struct Conv1Params {
  const void* in16;     // IN16: (B,H,W,8) fp16 pixel records (the first eight net-input channels)
  const void* in16x;    // IN16 + DEP: (B,H,W,2) fp16, channels 8-9
  const float* in;      // (B,8,H,W) fp32 — (B,10,H,W) for the DEP kernel
  const h8* wdep;       // DEP: [16 pseudo taps][64 co] h8, read from global memory (L1-resident, 14 KB)
  const h8* wp;         // packed [hi|lo][50 taps][64 co] h8 (tap 49 = zeros)
  const float* bias;
  _Float16* out;        // split16 NHWC (B,Ho,Wo,128 halves)
  int B, H, W, Ho, Wo, tiles_x, tiles_y, ntiles;
  float slope, in_scale, acc_scale, out_scale;
  int* status;
};

// X3 = false: the same kernel with plain fp16 operands (BASELINE config 5): one MFMA per tile pair and k-step, NHWC fp16
// output (128 B per pixel), half the LDS — two blocks share a CU and cover each other's load / epilogue phases.
//
// DEP (plain fp16 only): BASELINE config 5's RGB-D input, Cin = 10 (deepIM_flownet.py:33-62 with INPUT_DEPTH). Channels 0-7 run
// exactly as above (25 k-steps); channels 8, 9 ride as 14 PSEUDO TAPS: for (ky, kx0 in {0, 4}) the 8 halves of a k-lane group
// are {ch8, ch9} of the 4 consecutive input columns kx0 … kx0+3 (kx = 7 has zero weights). After the 25 main steps the two
// extra channels — kept aside in LDS as [row][column][2 halves] — are regrouped into the parity-0 slots of the same patch
// array as 16-byte entries E[row][even column c] = columns c … c+3, so the pseudo tap (ky, kx0) is addressed exactly like the
// real tap (ky, kx0): 7 more k-steps instead of the 24 a 16-channel padding would cost. Their weight fragments come from
// global memory (14 KB, L1-resident; the 80 KB per block that keep two blocks on a CU are spent on the patch).
// IN16 (plain fp16 only): the net input arrives as fp16 pixel records from the zoom front end (deepim_zoom_concat_forward_h16)
// instead of NCHW fp32 planes: the patch is 1449 sixteen-byte records, six loads per thread, stored to LDS as they are — no
// conversion pass, a quarter of the input bytes.
template <bool X3, bool DEP = false, bool IN16 = false>
__global__ __launch_bounds__(256, X3 ? 1 : 2) void conv1_x3_kernel(Conv1Params p) {
  static_assert(!(X3 && DEP) && !(X3 && IN16), "the two-channel extension and the fp16-record input exist for the plain fp16 path only");
  constexpr int CIN = IN16 ? 1 : (DEP ? 10 : 8);   // fp32 planes loaded per quad (IN16: none, one dummy)
  extern __shared__ __attribute__((aligned(16))) h8 smem[];
  h8* w_hi = smem;
  h8* w_lo = smem + C1_WH;                         // X3 only
  h8* a_hi = smem + (X3 ? 2 : 1) * C1_WH;
  h8* a_lo = a_hi + C1_PH;                         // X3 only
  unsigned* dstage = reinterpret_cast<unsigned*>(a_hi + C1_PH);   // DEP only
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 5, lcol = lane & 31;
  for (int i = tid; i < (X3 ? 2 : 1) * C1_WH; i += 256) smem[i] = p.wp[i];

  // patch quads of this thread: the 21 x 69 patch inside 21 rows x 18 aligned quads of 4 consecutive pixels (72 columns
  // starting one pixel left of the patch); quad k*256 + tid, k = 0, 1. One dwordx4 per channel per quad: 16 loads per thread instead of 48 dword
  // loads — with the 16 stores of the previous tile still in flight 48 loads ran into the 63-instruction limit of vmcnt and
  // the issue of the prefetch stalled until the stores had drained (measured: 5-10 k cycles per tile).
  constexpr int C1_QUADS = C1_ROWS * 18;
  int qr[2], qc[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int q = min(k * 256 + tid, C1_QUADS - 1);
    qr[k] = q / 18;
    qc[k] = (q - qr[k] * 18) * 4;
  }
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 v[2][CIN];
  float amax = 0.f;
  const long plane = (long)p.H * p.W;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, IN16 ? 0 : (int)((long)p.B * CIN * plane * 4), 0x00020000);
  constexpr int C1_NREC = (C1_ROWS * 69 + 255) / 256;   // IN16: pixel records of the patch per thread
  i32x4 vr[IN16 ? C1_NREC : 1];
  i32x4 vx[IN16 && DEP ? 2 : 1];
  int rrow[IN16 ? C1_NREC : 1], rpc[IN16 ? C1_NREC : 1];
  if constexpr (IN16) {
#pragma unroll
    for (int k = 0; k < C1_NREC; ++k) {
      const int rec = min(k * 256 + tid, C1_ROWS * 69 - 1);
      rrow[k] = rec / 69;
      rpc[k] = rec - rrow[k] * 69;
    }
  }
  const __amdgpu_buffer_rsrc_t rs_in16 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in16, 0, IN16 ? (int)((long)p.B * plane * 16) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_in16x = __builtin_amdgcn_make_buffer_rsrc((void*)p.in16x, 0, IN16 && DEP ? (int)((long)p.B * plane * 4) : 0, 0x00020000);
  auto load_patch = [&](int tile) {
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / (p.tiles_x * p.tiles_y);
    const int gy0 = 2 * (ty * 8) - 3, gx0 = 2 * (tx * 32) - 4;   // quads start one pixel left of the patch: 16-byte aligned
    if constexpr (IN16) {
#pragma unroll
      for (int k = 0; k < C1_NREC; ++k) {
        const int gy = gy0 + rrow[k], gx = gx0 + 1 + rpc[k];
        const bool ok = (k * 256 + tid) < C1_ROWS * 69 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const unsigned off = (((unsigned)((n * p.H + gy) * p.W + gx) * 16u) & 0x7fffffffu) | ((unsigned)(!ok) << 31);
        vr[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_in16, (int)off, 0, 0);
      }
      if constexpr (DEP) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int gy = gy0 + qr[k], gx = gx0 + qc[k];
          const bool ok = (k * 256 + tid) < C1_QUADS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
          const unsigned off = (((unsigned)((n * p.H + gy) * p.W + gx) * 4u) & 0x7fffffffu) | ((unsigned)(!ok) << 31);
          vx[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_in16x, (int)off, 0, 0);
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int gy = gy0 + qr[k], gx = gx0 + qc[k];
      // W % 4 == 0, so an aligned quad is entirely inside or entirely outside the frame; outside (and the unused second quad
      // of the upper threads): offset bit 31 → the load returns zeros
      const bool row_ok = (k * 256 + tid) < C1_QUADS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      // one select per quad, not per load (the compiler turned per-load selects into branches with a vmcnt(0) between the
      // loads); an invalid quad stays invalid for every channel: 0x80000000 + c·cstep < 2^32
      const unsigned b0 = row_ok ? (unsigned)((((long)n * CIN * p.H + gy) * p.W + gx) * 4) : 0x80000000u;
      const unsigned cstep = (unsigned)(plane * 4);
#pragma unroll
      for (int c = 0; c < CIN; ++c)
        v[k][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(b0 + (unsigned)c * cstep), 0, 0));
    }
  };
  auto store_patch = [&]() {
    if constexpr (IN16) {
#pragma unroll
      for (int k = 0; k < C1_NREC; ++k)
        if (k * 256 + tid < C1_ROWS * 69) a_hi[(rrow[k] * 2 + (rpc[k] & 1)) * C1_RW + (rpc[k] >> 1)] = __builtin_bit_cast(h8, vr[k]);
      if constexpr (DEP) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (k * 256 + tid < C1_QUADS) *reinterpret_cast<i32x4*>(dstage + qr[k] * 72 + qc[k]) = vx[k];
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k * 256 + tid >= C1_QUADS) continue;
      if constexpr (DEP && !IN16) {                // channels 8, 9 of the quad's 4 columns: one (ch8, ch9) pair per column
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        i32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h2 pr;
          pr[0] = (_Float16)(e == 0 ? v[k][8 % CIN].x : e == 1 ? v[k][8 % CIN].y : e == 2 ? v[k][8 % CIN].z : v[k][8 % CIN].w);
          pr[1] = (_Float16)(e == 0 ? v[k][9 % CIN].x : e == 1 ? v[k][9 % CIN].y : e == 2 ? v[k][9 % CIN].z : v[k][9 % CIN].w);
          d[e] = __builtin_bit_cast(int, pr);
        }
        *reinterpret_cast<i32x4*>(dstage + qr[k] * 72 + qc[k]) = d;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pc = qc[k] + e - 1;             // patch column of this element
        if (pc < 0 || pc >= 69) continue;
        h8 hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float x = e == 0 ? v[k][c % CIN].x : e == 1 ? v[k][c % CIN].y : e == 2 ? v[k][c % CIN].z : v[k][c % CIN].w;
          if constexpr (X3) {
            const X3Pair s2 = x3_split(x, p.in_scale, amax);
            hi[c] = s2.hi;
            lo[c] = s2.lo;
          } else {
            hi[c] = (_Float16)x;
          }
        }
        const int idx = (qr[k] * 2 + (pc & 1)) * C1_RW + (pc >> 1);
        a_hi[idx] = hi;
        if constexpr (X3) a_lo[idx] = lo;
      }
    }
  };

  // bias of this lane's 8 channel runs (i, g), kept in registers: a load inside the epilogue would wait behind the prefetch
  // loads and the previous tile's stores (vmcnt counts in order)
  // (the DEP kernel has no registers to spare for them and keeps the 64 biases in LDS instead)
  float bias_r[DEP ? 1 : 2][DEP ? 1 : 4][DEP ? 1 : 4];
  float* bias_s = reinterpret_cast<float*>(dstage + C1_DSTAGE / 4);
  if constexpr (DEP) {
    if (tid < 64) bias_s[tid] = p.bias ? p.bias[tid] : 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[i][g][r] = p.bias ? p.bias[i * 32 + 8 * g + 4 * lrow + r] : 0.f;
  }
  int tile = blockIdx.x;
  if (tile < p.ntiles) load_patch(tile);
  // fragment bases: A rows = output channels (lane%32, + 32 for the second tile), the half-wave picks tap t or t+1;
  // B columns = 32 output pixels of row 2*wave + j
  const int a_base = lrow * 64 + lcol;
  const int b_base0 = (2 * (2 * wave) * 2) * C1_RW + lcol, b_base1 = (2 * (2 * wave + 1) * 2) * C1_RW + lcol;
  // epilogue staging (the patch region, once every wave is done with it): this wave's 32 pixels of one output row as
  // records of 256 B at a 272-byte pitch, so the lane-per-pixel writes spread over the banks
  constexpr int C1_REC = X3 ? 256 : 128, C1_PITCH = C1_REC + 16;
  char* stage = reinterpret_cast<char*>(a_hi) + wave * (32 * C1_PITCH);
  h8 fr[2][8];   // [set][ah0 ah1 al0 al1 bh0 bh1 bl0 bl1]
  auto read_frags = [&](int set, int pp) {
    const int t0 = 2 * pp, t1 = min(2 * pp + 1, C1_TAPS - 1);   // tap 49 has zero weights: any valid pixel address will do
    const int ky0 = t0 / 7, kx0 = t0 % 7, ky1 = t1 / 7, kx1 = t1 % 7;
    const int o0 = (ky0 * 2 + (kx0 & 1)) * C1_RW + (kx0 >> 1), o1 = (ky1 * 2 + (kx1 & 1)) * C1_RW + (kx1 >> 1);
    const int bo = lrow ? o1 : o0;
    fr[set][0] = w_hi[t0 * 64 + a_base];
    fr[set][1] = w_hi[t0 * 64 + a_base + 32];
    fr[set][4] = a_hi[b_base0 + bo];
    fr[set][5] = a_hi[b_base1 + bo];
    if constexpr (X3) {
      fr[set][6] = a_lo[b_base0 + bo];
      fr[set][7] = a_lo[b_base1 + bo];
      fr[set][2] = w_lo[t0 * 64 + a_base];
      fr[set][3] = w_lo[t0 * 64 + a_base + 32];
    }
  };
  for (; tile < p.ntiles; tile += gridDim.x) {
    __syncthreads();                       // every wave is done with the staging area (and the weights are in place)
    store_patch();
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) load_patch(tile + gridDim.x);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    {
      read_frags(0, 0);
#pragma unroll
      for (int pp = 0; pp < C1_PAIRS; ++pp) {
        const int cur = pp & 1;
        __builtin_amdgcn_sched_barrier(0);
        if (pp + 1 < C1_PAIRS) read_frags(cur ^ 1, pp + 1);     // next pair's fragments fly while this pair multiplies
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][i], fr[cur][4 + j], acc[i][j], 0, 0, 0);
        if constexpr (X3) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][i], fr[cur][6 + j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][2 + i], fr[cur][4 + j], acc[i][j], 0, 0, 0);
        }
      }
    }
    if constexpr (DEP) {
      __syncthreads();                     // every wave is past the 8-channel patch: its parity-0 slots take the E entries
      for (int i = tid; i < C1_ROWS * C1_RW; i += 256) {
        const int row = i / C1_RW, j = i - row * C1_RW;
        const unsigned* src = dstage + row * 72 + 2 * j + 1;     // patch column 2j sits in quad column 2j + 1
        i32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = (int)src[min(e, 70 - 2 * j)];   // column 72 does not exist: only ever multiplied by the zero weights of kx = 7
        a_hi[(row * 2) * C1_RW + j] = __builtin_bit_cast(h8, d);
      }
      __syncthreads();
      auto read_dep = [&](int set, int pp) {    // into the main loop's fragment registers (dead by now)
        const int s0 = 2 * pp, s1 = 2 * pp + 1;                // pseudo tap s: ky = s / 2, kx0 = 4 (s % 2)
        const int o0 = ((s0 >> 1) * 2) * C1_RW + (s0 & 1) * 2, o1 = ((s1 >> 1) * 2) * C1_RW + (s1 & 1) * 2;
        const int bo = lrow ? o1 : o0;
        fr[set][0] = p.wdep[s0 * 64 + a_base];
        fr[set][1] = p.wdep[s0 * 64 + a_base + 32];
        fr[set][4] = a_hi[b_base0 + bo];
        fr[set][5] = a_hi[b_base1 + bo];
      };
      read_dep(0, 0);
#pragma unroll
      for (int pp = 0; pp < C1_DEP_TAPS / 2; ++pp) {
        const int cur = pp & 1;
        if (pp + 1 < C1_DEP_TAPS / 2) read_dep(cur ^ 1, pp + 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][i], fr[cur][4 + j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();                       // all waves are past their last patch read: the region becomes the staging area
    // epilogue: real units, bias, LeakyReLU, split → LDS record → 16 B per lane, 1 KB (4 pixel records) per wave store
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / (p.tiles_x * p.tiles_y);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.out + (long)n * p.Ho * p.Wo * (C1_REC / 2)), 0, (int)((long)p.Ho * p.Wo * C1_REC), 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int oy = ty * 8 + 2 * wave + j;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co0 = i * 32 + 8 * g + 4 * lrow;
          h4 vh, vl;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = (X3 ? acc[i][j][4 * g + r] * p.acc_scale : acc[i][j][4 * g + r]) + (DEP ? bias_s[co0 + r] : bias_r[DEP ? 0 : i][DEP ? 0 : g][DEP ? 0 : r]);
            x = x > 0.f ? x : x * p.slope;
            if constexpr (X3) {
              const X3Pair s2 = x3_split(x, p.out_scale, amax);
              vh[r] = s2.hi;
              vl[r] = s2.lo;
            } else {
              vh[r] = (_Float16)x;
            }
          }
          if constexpr (X3) {
            char* rec = stage + lcol * C1_PITCH + (co0 >> 4) * 64 + (co0 & 15) * 2;
            *reinterpret_cast<h4*>(rec) = vh;
            *reinterpret_cast<h4*>(rec + 32) = vl;
          } else {
            *reinterpret_cast<h4*>(stage + lcol * C1_PITCH + co0 * 2) = vh;
          }
        }
      // the wave reads back its own writes (LDS operations of a wave complete in order)
      constexpr int CPP = C1_REC / 16, PPI = 64 / CPP;    // 16-byte chunks per pixel record, pixels per wave store
#pragma unroll
      for (int k = 0; k < 32 / PPI; ++k) {
        const int px = k * PPI + lane / CPP, chunk = lane % CPP;
        const i32x4 d = *reinterpret_cast<const i32x4*>(stage + px * C1_PITCH + chunk * 16);
        const int ox = tx * 32 + px;
        const bool ok = oy < p.Ho && ox < p.Wo;
        const unsigned off = ok ? (unsigned)((oy * p.Wo + ox) * C1_REC + chunk * 16) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(d, rs_out, (int)off, 0, 0);
      }
    }
  }
  if constexpr (X3) x3_report(amax, p.status);
}

// conv1 weights (64,8,7,7) fp32 → [hi|lo][50 taps][64 co][8 halves] of w·w_scale (tap 49 zero)
__global__ void pack_conv1_x3_kernel(_Float16* __restrict__ packed, const float* __restrict__ w, float w_scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * C1_WH * 8) return;
  const int c = i & 7, co = (i >> 3) & 63, t = (i >> 9) % 50, part = i / (C1_WH * 8);
  X3Pair s2 = {(_Float16)0.f, (_Float16)0.f};
  if (t < C1_TAPS) s2 = x3_split(w[((co * 8 + c) * 7 + t / 7) * 7 + t % 7], w_scale);
  packed[i] = part ? s2.lo : s2.hi;
}

// conv1 weights of the RGB-D input (64,10,7,7) fp32 → [50 taps][64 co][8 halves] for channels 0-7 (tap 49 zero), then
// [16 pseudo taps][64 co][8 halves]: pseudo tap s = 2 ky + (kx0 / 4) holds {w(8,ky,kx0+e), w(9,ky,kx0+e)} for e = 0 … 3, zero
// where kx0 + e = 7 and for s >= 14
__global__ void pack_conv1_c10_kernel(_Float16* __restrict__ packed, const float* __restrict__ w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (C1_WH + 16 * 64) * 8) return;
  float v = 0.f;
  if (i < C1_WH * 8) {
    const int c = i & 7, co = (i >> 3) & 63, t = i >> 9;
    if (t < C1_TAPS) v = w[((co * 10 + c) * 7 + t / 7) * 7 + t % 7];
  } else {
    const int k = i - C1_WH * 8, h = k & 7, co = (k >> 3) & 63, s_ = k >> 9;
    const int ky = s_ >> 1, kx = (s_ & 1) * 4 + (h >> 1), c = 8 + (h & 1);
    if (s_ < C1_DEP_TAPS && kx < 7) v = w[((co * 10 + c) * 7 + ky) * 7 + kx];
  }
  packed[i] = (_Float16)v;
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
  const int BM = f16_bm(Cout);
  return (size_t)di_div_up(Cout, BM) * f16_chunks(Cin_pad, kh, kw) * HOCT * BM * 8 * sizeof(_Float16);
}

extern "C" int deepim_conv_f16_pack_weights(deepim_ctx* ctx, void* packed, const float* w, int Cout, int Cin, int Cin_pad,
                                            int kh, int kw) {
  DI_DEVICE(ctx);
  DI_REQUIRE(Cin_pad >= Cin && (Cin_pad & 7) == 0, "conv_f16_pack: Cin_pad must be a multiple of 8 and >= Cin");
  const int nchunk = f16_chunks(Cin_pad, kh, kw);
  const int BM = f16_bm(Cout);
  const long total = (long)di_div_up(Cout, BM) * nchunk * HOCT * BM * 8;
  hipLaunchKernelGGL(pack_f16_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, (_Float16*)packed, w, Cout,
                     Cin, Cin_pad, kh, kw, nchunk, BM, total);
  DI_LAUNCH_CHECK();
  return 0;
}

namespace {
// Plan + launch of the LDS-DMA kernel (fp16 or X3). One 256-thread block per CU. Plans, in units of one 64-wide K chunk:
//   uniform split-K s:   ceil(blocks·s / 256) · ceil(nchunk / s)                      + reduce(blocks·s)
//   tail split ts:       floor(blocks / 256) · nchunk + ceil(R·ts / 256) · ceil(nchunk / ts) + reduce(R·ts),  R = blocks mod 256
// reduce(n) = c0 + c1·n·(BM·BN/16384): fp32 partial tiles written and read once; deterministic, no timing involved.
template <bool X3>
int launch_f16_dma(deepim_ctx* ctx, ConvF16Params p, int BM, int BN, float c0, float c1, bool pp = false) {
  const int blocks = p.gx * p.gy;
  const bool w8 = !pp && BM == 256 && BN == 256 && (ctx->f16_dev_flags & DI_F16_W8) != 0;   // dev: one 8-wave block per CU on a 256x256 tile
  const bool tn2 = !pp && ((BM == 256 && BN == 128) || (BM == 128 && BN == 256));       // 128x64 wave tiles, two blocks per CU
  const int slots = tn2 ? 512 : 256;            // resident blocks on the chip (the ping-pong kernel: one 8-wave block per CU)
  int ks = 1, ts = 0;
  if (ctx->conv_max_split != 1) {
    float best = 1e30f;
    const float tile_w = (float)(BM * BN) / 16384.f;
    for (int s_ : {1, 2, 3, 4, 6, 8, 12, 16}) {
      if (s_ > 1 && ((long)blocks * s_ > 2048 || s_ > max(1, p.nchunk / 4))) continue;
      const float cost = (float)di_div_up((long)blocks * s_, slots) * (float)di_div_up(p.nchunk, s_) +
                         (s_ > 1 ? c0 + c1 * (float)((long)blocks * s_) * tile_w : 0.f);
      if (cost < best * 0.985f) { best = cost; ks = s_; }
    }
    const int R = blocks % slots;
    if (blocks > slots && R > 0 && ctx->conv_max_split == 0 && !(ctx->f16_dev_flags & DI_F16_NO_TAIL)) {
      for (int t_ : {2, 3, 4, 5, 6, 8}) {
        if (t_ > max(1, p.nchunk / 4)) continue;
        const float cost = (float)(blocks / slots) * (float)p.nchunk + (float)di_div_up(R * t_, slots) * (float)di_div_up(p.nchunk, t_) +
                           c0 + c1 * (float)(R * t_) * tile_w;
        if (cost < best * 0.97f) { best = cost; ts = t_; ks = 1; }
      }
    }
  }
  if (ctx->conv_max_split > 1) ks = min(ks, ctx->conv_max_split);
  p.chunks_per_split = di_div_up(p.nchunk, ks);
  p.ksplit = di_div_up(p.nchunk, p.chunks_per_split);
  p.partial = nullptr;
  p.n_full = 0; p.tail_s = 0; p.tail_cps = 0; p.tail_partial = nullptr;
  int grid = blocks * p.ksplit;
  if (p.ksplit > 1) {
    void* scratch;
    int rc = deepim_scratch(ctx, (size_t)p.ksplit * p.npix * p.Cout * sizeof(float), &scratch);
    if (rc) return rc;
    p.partial = (float*)scratch;
  } else if (ts > 1) {
    const int R = blocks % slots;
    p.n_full = blocks - R;
    p.tail_cps = di_div_up(p.nchunk, ts);
    p.tail_s = di_div_up(p.nchunk, p.tail_cps);
    void* scratch;
    int rc = deepim_scratch(ctx, (size_t)p.tail_s * R * BM * BN * sizeof(float), &scratch);
    if (rc) return rc;
    p.tail_partial = (float*)scratch;
    grid = p.n_full + R * p.tail_s;
  }
  static const char attr_tag = 0;   // function attributes are per DEVICE: remember them per context
  if (di_attr_needed(ctx, &attr_tag)) {
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_dma_kernel<2, 2, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_dma_kernel<1, 4, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 122880));
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_dma_kernel<2, 2, 3, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_dma_kernel<2, 2, 3, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_dma_kernel<1, 4, 3, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_dma_kernel<1, 4, 3, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_dma_kernel<2, 4, 4, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_dma_kernel<2, 4, 4, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_pp_kernel<2, 4, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
#if DI_PP_TRACE
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_pp_kernel<2, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 135168 + 8192));
#endif
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_pp_kernel<1, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  }
  if (getenv("DEEPIM_CONV_VERBOSE"))
    fprintf(stderr, "[deepim] %s plan B=%d Cin=%d %dx%d Cout=%d: %d tiles of %dx%d, split-K %d, tail split %d (R=%d)\n",
            X3 ? "x3" : "f16", p.B, p.Cin, p.H, p.W, p.Cout, blocks, BM, BN, p.ksplit, p.tail_s, blocks - p.n_full);
  if (pp) {
    if constexpr (!X3) {
#if DI_PP_TRACE
      if (BM == 256) hipLaunchKernelGGL((conv_f16_pp_kernel<2, 4, 4>), dim3(grid), dim3(512), 135168 + 8192, ctx->stream, p);
#else
      if (BM == 256) hipLaunchKernelGGL((conv_f16_pp_kernel<2, 4, 5>), dim3(grid), dim3(512), 163840, ctx->stream, p);
#endif
      else hipLaunchKernelGGL((conv_f16_pp_kernel<1, 8, 4>), dim3(grid), dim3(512), 163840, ctx->stream, p);
    } else {
      DI_REQUIRE(false, "conv2d_x3: no ping-pong kernel");
    }
  } else if (BM == 128 && tn2) hipLaunchKernelGGL((conv_f16_dma_kernel<1, 4, 3, X3, 2>), dim3(grid), dim3(256), 73728, ctx->stream, p);
  else if (w8) hipLaunchKernelGGL((conv_f16_dma_kernel<2, 4, 4, X3, 2>), dim3(grid), dim3(512), 131072, ctx->stream, p);
  else if (tn2) hipLaunchKernelGGL((conv_f16_dma_kernel<2, 2, 3, X3, 2>), dim3(grid), dim3(256), 73728, ctx->stream, p);
  else if constexpr (!X3) {   // 128x128 wave tiles, one block per CU (DEEPIM_F16_TN4=1; plain fp16 only)
    if (BM == 128) hipLaunchKernelGGL((conv_f16_dma_kernel<1, 4, 3, false>), dim3(grid), dim3(256), 122880, ctx->stream, p);
    else hipLaunchKernelGGL((conv_f16_dma_kernel<2, 2, 4, false>), dim3(grid), dim3(256), 131072, ctx->stream, p);
  } else {
    DI_REQUIRE(false, "conv2d_x3: tile shape not built");
  }
  if (p.ksplit > 1) {
    const long total4 = p.npix * p.Cout / 4;
    if (X3)
      hipLaunchKernelGGL(splitk_x3_reduce_kernel, dim3(di_div_up(total4, 256)), dim3(256), 0, ctx->stream, p.out, p.partial, p.bias,
                         total4, p.ksplit, p.Cout, p.slope, p.acc_scale, p.out_scale, p.status);
    else
      hipLaunchKernelGGL(splitk_f16_reduce_kernel, dim3(di_div_up(total4, 256)), dim3(256), 0, ctx->stream, p.out, p.partial, p.bias,
                         total4, p.ksplit, p.Cout, p.slope);
  } else if (p.tail_s > 1) {
    const long total = (long)(blocks - p.n_full) * BN * (BM / 4);
    hipLaunchKernelGGL(tail_f16_reduce_kernel<X3>, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, p, BM, BN);
  }
  DI_LAUNCH_CHECK();
  return 0;
}
}  // namespace

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
  p.stride_kw = (kh << 16) | kw;
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
                       noct_pad, Cin_pad, kh, kw, W);
    ctx->conv_tabs.push_back({2, Cin_pad, kh, kw, H, W, (void*)tab});
  }
  p.tab = tab;
  const bool ut = ((Cin_pad >> 3) & 7) == 0;
  const int BM = f16_bm(Cout);
  int BN = f16_bn(Cout, ut && !(ctx->f16_dev_flags & DI_F16_NO_DMA), ctx->f16_dev_flags);
  p.n_full = 0; p.tail_s = 0; p.tail_cps = 0; p.tail_partial = nullptr; p.acc_scale = p.out_scale = 1.f; p.status = ctx->status;
  // Ping-pong kernel (one 8-wave block per CU, 256x256 / 128x512 tiles) from DI_F16_PP_MIN_TILES tiles on: conv2 … conv5_1 at
  // B = 32, conv2 … conv4_1 at B = 8 (150-tile layers run 1.2x faster on it than on the 4-wave kernel even half filled). Smaller
  // grids keep the 4-wave kernel on 256x128 tiles (two blocks per CU: twice the blocks for the same split factor). Fixed by the
  // geometry: the same plan on every rank.
  if (ut && BM >= 128 && (Cout % 128) == 0 && !(ctx->f16_dev_flags & (DI_F16_NO_DMA | DI_F16_NO_PP | DI_F16_TN4 | DI_F16_W8))) {
    const int bn_pp = BM == 256 ? 256 : 512;
    const long tiles_pp = (long)di_div_up(p.npix, bn_pp) * di_div_up(Cout, BM);
    if (tiles_pp >= DI_F16_PP_MIN_TILES) {
      BN = bn_pp;
      p.gx = di_div_up(p.npix, BN); p.gy = di_div_up(Cout, BM);
      return launch_f16_dma<false>(ctx, p, BM, BN, 1.5f, 0.009f, true);
    }
  }
  p.gx = di_div_up(p.npix, BN); p.gy = di_div_up(Cout, BM);
  if (ut && BM >= 128 && !(ctx->f16_dev_flags & DI_F16_NO_DMA)) return launch_f16_dma<false>(ctx, p, BM, BN, 1.5f, 0.009f);
  const int blocks = p.gx * p.gy;
  // one 256-thread block per CU (the LDS double buffer and the 256 accumulator registers leave room for one): split K when
  // the grid cannot fill the 256 CUs; deterministic (cost model of csrc/conv.hip's plan_ksplit, one slot per CU)
  int ks = 1;
  if (ctx->conv_max_split != 1) {
    float best = 1e30f;
    for (int s_ : {1, 2, 3, 4, 6, 8, 12, 16}) {
      if (s_ > 1 && ((long)blocks * s_ > 2048 || s_ > max(1, p.nchunk / 4))) continue;
      const float cost = (float)di_div_up((long)blocks * s_, 256) * (float)di_div_up(p.nchunk, s_) +
                         (s_ > 1 ? 1.5f + 0.009f * (float)((long)blocks * s_) * (float)(BM * BN) / 16384.f : 0.f);   // reduce: 8 B of fp32 partials written + read per output per slice
      if (cost < best * 0.985f) { best = cost; ks = s_; }
    }
  }
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
  const size_t lds = (size_t)2 * HOCT * (BM + BN) * 16;
  static const char attr_set_tag = 0;   // function attributes are per DEVICE: remember them per context
  if (di_attr_needed(ctx, &attr_set_tag)) {
#define DI_F16_ATTR(A, B2, C, D, BMv, BNv)                                                                                  \
  DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_kernel<A, B2, C, D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               2 * HOCT * (BMv + BNv) * 16));                                                              \
  DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_kernel<A, B2, C, D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               2 * HOCT * (BMv + BNv) * 16));
    DI_F16_ATTR(1, 4, 2, 2, 64, 256)
    DI_F16_ATTR(2, 2, 2, 4, 128, 256)
    DI_F16_ATTR(2, 2, 4, 4, 256, 256)
#undef DI_F16_ATTR
  }
  const dim3 grid(blocks * p.ksplit);
#define DI_F16_LAUNCH(A, B2, C, D)                                                                                 \
  {                                                                                                                \
    if (ut) hipLaunchKernelGGL((conv_f16_kernel<A, B2, C, D, true>), grid, dim3(256), lds, ctx->stream, p);         \
    else hipLaunchKernelGGL((conv_f16_kernel<A, B2, C, D, false>), grid, dim3(256), lds, ctx->stream, p);           \
  }
  static const char dma_attr_tag = 0;   // function attributes are per DEVICE: remember them per context
  if (di_attr_needed(ctx, &dma_attr_tag)) {
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_dma_kernel<2, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    DI_CHECK(hipFuncSetAttribute((const void*)conv_f16_dma_kernel<1, 4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 122880));
  }
  const bool dma = ut && !(ctx->f16_dev_flags & DI_F16_NO_DMA);
  if (BM == 64) DI_F16_LAUNCH(1, 4, 2, 2)
  else if (BM == 128 && dma) hipLaunchKernelGGL((conv_f16_dma_kernel<1, 4, 3>), grid, dim3(256), 122880, ctx->stream, p);
  else if (BM == 128) DI_F16_LAUNCH(2, 2, 2, 4)
  else if (dma) hipLaunchKernelGGL((conv_f16_dma_kernel<2, 2, 4>), grid, dim3(256), 131072, ctx->stream, p);
  else DI_F16_LAUNCH(2, 2, 4, 4)
#undef DI_F16_LAUNCH
  if (p.ksplit > 1) {
    const long total4 = p.npix * Cout / 4;
    hipLaunchKernelGGL(splitk_f16_reduce_kernel, dim3(di_div_up(total4, 256)), dim3(256), 0, ctx->stream, p.out,
                       p.partial, bias, total4, p.ksplit, Cout, slope);
  }
  DI_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------- X3: split-fp16 convolution ----
extern "C" int deepim_nchw_f32_to_split16(deepim_ctx* ctx, void* out_split16, const float* in, int B, int C, int H, int W,
                                          float scale) {
  DI_DEVICE(ctx);
  DI_REQUIRE((C & 15) == 0, "nchw_to_split16: C must be a multiple of 16");
  const long total = (long)B * H * W * (C / 4);
  if (total == 0) return 0;
  hipLaunchKernelGGL(nchw_to_split16_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, (_Float16*)out_split16, in,
                     C, (long)H * W, scale, total, ctx->status);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_split16_to_nchw_f32(deepim_ctx* ctx, float* out, const void* in_split16, int B, int C, int H, int W,
                                          float inv_scale) {
  DI_DEVICE(ctx);
  DI_REQUIRE((C & 15) == 0, "split16_to_nchw: C must be a multiple of 16");
  const long total = (long)B * C * H * W;
  if (total == 0) return 0;
  hipLaunchKernelGGL(split16_to_nchw_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, out,
                     (const _Float16*)in_split16, C, (long)H * W, inv_scale, total);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t deepim_conv_x3_packed_size(int Cout, int Cin, int kh, int kw) {
  return deepim_conv_f16_packed_size(Cout, 2 * Cin, kh, kw);
}

extern "C" int deepim_conv_x3_pack_weights(deepim_ctx* ctx, void* packed, const float* w, int Cout, int Cin, int kh, int kw,
                                           float w_scale) {
  DI_DEVICE(ctx);
  DI_REQUIRE((Cin & 31) == 0 && (Cout & 127) == 0, "conv_x3_pack: needs Cin % 32 == 0 and Cout % 128 == 0");
  const int nchunk = f16_chunks(2 * Cin, kh, kw);
  const int BM = f16_bm(Cout);
  const long total = (long)di_div_up(Cout, BM) * nchunk * HOCT * BM * 8;
  hipLaunchKernelGGL(pack_x3_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, (_Float16*)packed, w, Cout, Cin, kh,
                     kw, nchunk, BM, w_scale, total);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_conv2d_x3_forward(deepim_ctx* ctx, void* out_split16, const void* in_split16, const void* packed_w,
                                        const float* bias, int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride,
                                        int pad, float slope, float acc_scale, float out_scale) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE((Cin & 31) == 0 && (Cout & 127) == 0, "conv2d_x3: needs Cin % 32 == 0 and Cout % 128 == 0");
  DI_REQUIRE(kh <= 7 && kw <= 7, "conv2d_x3: kernel larger than 7 not supported");
  const int Cv = 2 * Cin;   // virtual fp16 channels
  {
    // one raw-buffer descriptor with bit 31 of the offset as the padding marker addresses < 2 GiB of input per launch: larger
    // batches run as consecutive sub-batches (samples are independent)
    const size_t per_sample = (size_t)H * W * Cv * 2, limit = 0x7fffffffUL - (size_t)(pad * W + pad) * Cv * 2;
    if ((size_t)B * per_sample >= limit) {
      const int Bc = (int)((limit - 1) / per_sample);
      DI_REQUIRE(Bc >= 1, "conv2d_x3: one sample exceeds 2 GiB");
      const size_t out_sample = (size_t)((H + 2 * pad - kh) / stride + 1) * ((W + 2 * pad - kw) / stride + 1) * 2 * Cout;
      for (int b0 = 0; b0 < B; b0 += Bc) {
        const int rc = deepim_conv2d_x3_forward(ctx, (_Float16*)out_split16 + (size_t)b0 * out_sample,
                                                (const _Float16*)in_split16 + (size_t)b0 * H * W * Cv, packed_w, bias,
                                                min(Bc, B - b0), Cin, H, W, Cout, kh, kw, stride, pad, slope, acc_scale, out_scale);
        if (rc) return rc;
      }
      return 0;
    }
  }
  ConvF16Params p;
  p.in = in_split16; p.wp = (const h8*)packed_w; p.bias = bias; p.out = (_Float16*)out_split16; p.tab = nullptr;
  p.B = B; p.Cin = Cv; p.H = H; p.W = W; p.Cout = Cout;
  p.Ho = (H + 2 * pad - kh) / stride + 1;
  p.Wo = (W + 2 * pad - kw) / stride + 1;
  p.stride = stride; p.pad = pad; p.slope = slope;
  p.stride_kw = (kh << 16) | kw;
  p.acc_scale = acc_scale; p.out_scale = out_scale; p.status = ctx->status;
  p.nchunk = f16_chunks(Cv, kh, kw);
  p.npix = (long)B * p.Ho * p.Wo;
  p.pad_bytes = (pad * W + pad) * Cv * 2;
  const size_t in_bytes = (size_t)B * H * W * Cv * 2;
  DI_REQUIRE(in_bytes + p.pad_bytes < 0x7fffffffUL, "conv2d_x3: input tensor must be < 2 GiB per launch");
  p.in_bytes = (unsigned)in_bytes;
  const int BM = f16_bm(Cout), BN = (ctx->f16_dev_flags & DI_F16_W8) && BM == 256 ? 256 : (BM == 256 ? 128 : 256);   // 128x64 wave tiles
  p.gx = di_div_up(p.npix, BN); p.gy = di_div_up(Cout, BM);
  // the plan model of the fp16 path with a chunk 1.5x as long (96 instead of 64 MFMAs per wave)
  return launch_f16_dma<true>(ctx, p, BM, BN, 1.0f, 0.006f);
}

extern "C" size_t deepim_conv1_x3_packed_size(void) { return (size_t)2 * C1_WH * 16; }

extern "C" int deepim_conv1_x3_pack_weights(deepim_ctx* ctx, void* packed, const float* w, float w_scale) {
  DI_DEVICE(ctx);
  hipLaunchKernelGGL(pack_conv1_x3_kernel, dim3(di_div_up(2 * C1_WH * 8, 256)), dim3(256), 0, ctx->stream, (_Float16*)packed, w,
                     w_scale);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_conv1_x3_forward(deepim_ctx* ctx, void* out_split16, const float* in, const void* packed_w,
                                       const float* bias, int B, int H, int W, float slope, float in_scale, float acc_scale,
                                       float out_scale) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  Conv1Params p;
  p.in16 = p.in16x = nullptr; p.wdep = nullptr;
  p.in = in; p.wp = (const h8*)packed_w; p.bias = bias; p.out = (_Float16*)out_split16;
  p.B = B; p.H = H; p.W = W;
  p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1;
  p.tiles_x = di_div_up(p.Wo, 32); p.tiles_y = di_div_up(p.Ho, 8);
  const long nt = (long)p.tiles_x * p.tiles_y * B;
  DI_REQUIRE((W & 3) == 0, "conv1_x3: W must be a multiple of 4 (aligned quad loads); use deepim_conv2d_forward_split16 otherwise");
  if ((long)B * 8 * H * W * 4 >= 0x7fffffffL) {   // < 2 GiB of input per launch: consecutive sub-batches
    const int Bc = (int)(0x7ffffffeL / ((long)8 * H * W * 4));
    DI_REQUIRE(Bc >= 1, "conv1_x3: one sample exceeds 2 GiB");
    for (int b0 = 0; b0 < B; b0 += Bc) {
      const int rc = deepim_conv1_x3_forward(ctx, (_Float16*)out_split16 + (size_t)b0 * p.Ho * p.Wo * 128, in + (size_t)b0 * 8 * H * W,
                                             packed_w, bias, min(Bc, B - b0), H, W, slope, in_scale, acc_scale, out_scale);
      if (rc) return rc;
    }
    return 0;
  }
  DI_REQUIRE(nt < (1L << 30), "conv1_x3: too many tiles");
  p.ntiles = (int)nt;
  p.slope = slope; p.in_scale = in_scale; p.acc_scale = acc_scale; p.out_scale = out_scale; p.status = ctx->status;
  static const char attr_tag = 0;   // function attributes are per DEVICE: remember them per context
  if (di_attr_needed(ctx, &attr_tag)) {
    DI_CHECK(hipFuncSetAttribute((const void*)conv1_x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDS));
  }
  const int grid = (int)min(256L, nt);   // persistent: one block per CU
  hipLaunchKernelGGL(conv1_x3_kernel<true>, dim3(grid), dim3(256), C1_LDS, ctx->stream, p);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t deepim_conv1_f16_c10_packed_size(void) { return (size_t)(C1_WH + 16 * 64) * 16; }

extern "C" int deepim_conv1_f16_c10_pack_weights(deepim_ctx* ctx, void* packed, const float* w) {
  DI_DEVICE(ctx);
  hipLaunchKernelGGL(pack_conv1_c10_kernel, dim3(di_div_up((C1_WH + 16 * 64) * 8, 256)), dim3(256), 0, ctx->stream,
                     (_Float16*)packed, w);
  DI_LAUNCH_CHECK();
  return 0;
}

static int conv1_f16_launch(deepim_ctx* ctx, void* out_nhwc_f16, const float* in, const void* packed_w, const float* bias, int B,
                            int H, int W, float slope, int Cin, const void* in16 = nullptr, const void* in16x = nullptr);

// conv1 of the fp16 path from the fp16 pixel records of deepim_zoom_concat_forward_h16: main8 (B,H,W,8) and, for the 10-channel
// RGB-D input, extra2 (B,H,W,2); packed_w as for deepim_conv1_f16_forward (extra2 == NULL) / deepim_conv1_f16_c10_forward
extern "C" int deepim_conv1_f16_h16_forward(deepim_ctx* ctx, void* out_nhwc_f16, const void* main8, const void* extra2,
                                            const void* packed_w, const float* bias, int B, int H, int W, float slope) {
  DI_REQUIRE(main8 != nullptr, "conv1_f16_h16: no input");
  return conv1_f16_launch(ctx, out_nhwc_f16, nullptr, packed_w, bias, B, H, W, slope, extra2 ? 10 : 8, main8, extra2);
}

// conv1 of the plain fp16 path on the same patch kernel: NCHW fp32 net input → NHWC fp16 (B,Ho,Wo,64); the packed weights are
// the hi half of deepim_conv1_x3_pack_weights(..., w_scale = 1)
extern "C" int deepim_conv1_f16_forward(deepim_ctx* ctx, void* out_nhwc_f16, const float* in, const void* packed_w,
                                        const float* bias, int B, int H, int W, float slope) {
  return conv1_f16_launch(ctx, out_nhwc_f16, in, packed_w, bias, B, H, W, slope, 8);
}

// the same for the 10-channel RGB-D net input (BASELINE config 5); packed_w from deepim_conv1_f16_c10_pack_weights
extern "C" int deepim_conv1_f16_c10_forward(deepim_ctx* ctx, void* out_nhwc_f16, const float* in, const void* packed_w,
                                            const float* bias, int B, int H, int W, float slope) {
  return conv1_f16_launch(ctx, out_nhwc_f16, in, packed_w, bias, B, H, W, slope, 10);
}

static int conv1_f16_launch(deepim_ctx* ctx, void* out_nhwc_f16, const float* in, const void* packed_w, const float* bias, int B,
                            int H, int W, float slope, int Cin, const void* in16, const void* in16x) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  Conv1Params p;
  p.in16 = in16; p.in16x = in16x;
  p.in = in; p.wp = (const h8*)packed_w; p.bias = bias; p.out = (_Float16*)out_nhwc_f16;
  p.wdep = (const h8*)packed_w + C1_WH;
  p.B = B; p.H = H; p.W = W;
  p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1;
  p.tiles_x = di_div_up(p.Wo, 32); p.tiles_y = di_div_up(p.Ho, 8);
  const long nt = (long)p.tiles_x * p.tiles_y * B;
  DI_REQUIRE((W & 3) == 0, "conv1_f16: W must be a multiple of 4 (aligned quad loads)");
  const long per_sample = in16 ? (long)H * W * 16 : (long)Cin * H * W * 4;      // bytes behind the (largest) buffer descriptor
  if ((long)B * per_sample >= 0x7fffffffL) {   // < 2 GiB of input per launch: consecutive sub-batches
    const int Bc = (int)(0x7ffffffeL / per_sample);
    DI_REQUIRE(Bc >= 1, "conv1_f16: one sample exceeds 2 GiB");
    for (int b0 = 0; b0 < B; b0 += Bc) {
      const int rc = conv1_f16_launch(ctx, (_Float16*)out_nhwc_f16 + (size_t)b0 * p.Ho * p.Wo * 64,
                                      in ? in + (size_t)b0 * Cin * H * W : nullptr, packed_w, bias, min(Bc, B - b0), H, W, slope, Cin,
                                      in16 ? (const char*)in16 + (size_t)b0 * H * W * 16 : nullptr,
                                      in16x ? (const char*)in16x + (size_t)b0 * H * W * 4 : nullptr);
      if (rc) return rc;
    }
    return 0;
  }
  DI_REQUIRE(nt < (1L << 30), "conv1_f16: too many tiles");
  p.ntiles = (int)nt;
  p.slope = slope; p.in_scale = p.acc_scale = p.out_scale = 1.f; p.status = ctx->status;
  static const char attr_tag = 0;   // function attributes are per DEVICE: remember them per context
  if (di_attr_needed(ctx, &attr_tag)) {
    DI_CHECK(hipFuncSetAttribute((const void*)conv1_x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDS_F16));
    DI_CHECK(hipFuncSetAttribute((const void*)conv1_x3_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDS_DEP));
    DI_CHECK(hipFuncSetAttribute((const void*)conv1_x3_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDS_F16));
    DI_CHECK(hipFuncSetAttribute((const void*)conv1_x3_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDS_DEP));
  }
  const int grid = (int)min(512L, nt);   // persistent: two blocks per CU
  DI_REQUIRE(Cin == 8 || Cin == 10, "conv1_f16: 8 (RGB pair + masks) or 10 (RGB-D pair + masks) input channels");
  if (in16 && Cin == 10) hipLaunchKernelGGL((conv1_x3_kernel<false, true, true>), dim3(grid), dim3(256), C1_LDS_DEP, ctx->stream, p);
  else if (in16) hipLaunchKernelGGL((conv1_x3_kernel<false, false, true>), dim3(grid), dim3(256), C1_LDS_F16, ctx->stream, p);
  else if (Cin == 10) hipLaunchKernelGGL((conv1_x3_kernel<false, true>), dim3(grid), dim3(256), C1_LDS_DEP, ctx->stream, p);
  else hipLaunchKernelGGL(conv1_x3_kernel<false>, dim3(grid), dim3(256), C1_LDS_F16, ctx->stream, p);
  DI_LAUNCH_CHECK();
  return 0;
}

#if DI_PP_TRACE
// dev builds only: register the device buffer (8 waves x 128 64-bit stamps) the traced tile dumps its timeline to
extern "C" int deepim_dev_pp_trace(void* buf) {
  unsigned long long* b = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pp_trace), &b, sizeof(b));
}
#endif
