// N-group: FlowNetS convolution / deconvolution stack on the fp32 matrix cores.
//   MXNet Convolution + bias + LeakyReLU   deepIM_flownet.py:63-107,123,145,176,317
//   MXNet Deconvolution k4 s2 + Crop       deepIM_flownet.py:127-165
//   MXNet Deconvolution k32 s16 grouped    deepIM_flownet.py:185-200,326-340
//
// Implicit GEMM  D[co][pixel] = Σ_k Wp[co][k] · X[k][pixel]  on v_mfma_f32_32x32x2_f32 — exact fp32, each output a
// single fmaf chain in the kernel's K order (bit-identical to the oracle run in that order), at the fp32 peak (157 TF).
// Pixels are flattened over (n,ho,wo); 4 waves per block, each a 64x64 accumulator tile; zero padding by out-of-range
// buffer offsets; XCD-aware 1-D tile order; deterministic split-K, autotuned per geometry; fused bias + LeakyReLU.
// Three kernels:
//   conv_nc8_kernel     LDS-free, channel-blocked (NC8) activations, K = (c/8,ky,kx,s,h): encoder conv2 … conv6_1
//   conv_direct_kernel  LDS-free, NCHW input, K = (ci/2,ky,kx,ci%2): conv1 (writes NC8); the encoder with nc8 off
//   conv_mfma_kernel    LDS-staged, canonical K = (ci,ky,kx): decoder deconvolutions, heads, odd Cin, bit-exact mode:
//     * K is consumed in chunks of 16: the weight chunk is a contiguous 16×BM slab of the pre-packed weights (one
//       dwordx4 per thread, straight into LDS [k][m]); the activation chunk is gathered global→registers→LDS
//       [k][pixel] with the per-thread pixel fixed for the whole K loop, the (ci,ky,kx)→offset table read through the
//       scalar cache;
//     * LDS is double-buffered (one s_barrier per chunk); operand reads are conflict-free ds_read_b32 (lanes 0-31 → 32
//       consecutive dwords of row k, lanes 32-63 → row k+1, matching the 32x32x2 A/B fragment layout);
//     * the output may be a channel slice of a wider tensor (free Concat).
// Roofline: MFMA-bound (arithmetic intensity ≈ 300 FLOP/B).
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// (helpers below)
constexpr int KT = 16;        // K chunk
constexpr int GRAN = 64;      // weight packing granule (rows)
constexpr int MODE_CONV = 0;
constexpr int MODE_DECONV = 1;  // k4 s2 p0 transposed conv, one launch z-slice per output parity
constexpr int MODE_NC8_TAB = 4;     // tap table of the NC8 kernel (one entry per (c8, tap) group)
constexpr int MODE_DIRECT_TAB = 3;  // tap-table flavour of the LDS-free kernel (conv_tabs key only; 2 is taken by conv_f16.hip)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// ds_read2st64_b32: two dwords per lane from addr + OFF*256 bytes (offsets in units of 64 dwords)
template <int O0, int O1>
__device__ __forceinline__ f32x2 lds_read2st64(unsigned addr) {
  static_assert(O0 >= 0 && O0 < 256 && O1 >= 0 && O1 < 256, "ds_read2st64 offset out of range");
  f32x2 v;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1) : "memory");
  return v;
}
// wait until at most N LDS reads are outstanding; the fragments are threaded through the asm so the
// MFMAs that consume them cannot be hoisted above the wait
template <int N, int TM, int TN>
__device__ __forceinline__ void lds_wait(f32x2 (&a)[TM], f32x2 (&b)[TN]) {
  if constexpr (TM == 2 && TN == 2)
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
  else if constexpr (TM == 1 && TN == 2)
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
  else if constexpr (TM == 1 && TN == 4)
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
  else
    static_assert(TM == 0, "unsupported wave tile");
}

struct ConvParams {
  const float* in;
  const float* wp;     // packed [Mgran][nchunk][KT][GRAN]
  const int2* tab;     // per k: {byte offset of tap (ci,ky,kx) from the pixel's tap origin, bit index ky*8+kx}; padded k → bit 63
  const float* bias;
  float* out;
  int B, Cin, H, W;         // input
  int Cout, Ho, Wo;         // output (after crop for deconv)
  int stride, pad;
  int nchunk;               // ceil(K/16)
  int ngran;                // 64-row weight granules per (parity class of the) packed tensor
  int out_ctotal, out_coff;
  float slope;
  int crop_y, crop_x;       // deconv crop offsets
  long npix;                // B*Ho*Wo (conv) or pixels per parity class (deconv)
  int ksplit;               // split-K slices (grid.z = classes * ksplit)
  int chunks_per_split;
  float* partial;           // [ksplit][B][Cout][Ho][Wo] raw partial sums when ksplit > 1
  long partial_stride;
  int swizzle;              // XCD-aware tile order on/off
  int gx, gy, gz;           // logical grid (pixel tiles, M tiles, classes*ksplit); launched 1-D and XCD-swizzled
  int pad_bytes;            // buffer-descriptor base shift: (pad*W + pad)*4 so tap origins are >= 0
  unsigned in_bytes;        // size of the input tensor (must stay < 2 GiB: 0x80000000 is the OOB marker)
  const float* wd;          // direct-kernel weights [Cout/32][k-pair][2][32] (second half of the packed buffer)
  const int2* tab2;         // direct-kernel tap table, one entry per k-pair (ci2,ky,kx): {byte offset of channel 2*ci2, bit}
  unsigned wd_bytes;
  // tail split (LDS-free kernel): tiles [0, n_full) run their whole K range; each of the R = tiles - n_full tiles that
  // would form the under-filled last round is cut into tail_s K slices (tile-local partials → tail_reduce_kernel)
  int n_full, tail_s, tail_cps, n_tail_pad;
  float* tail_partial;      // [tail_s][R][128 co][128 px]
  // channel-blocked activations ("NC8": [n][C/8][h][w][8]) between the encoder layers
  int in_nc8, out_nc8;      // layout of the input / output tensor (0 = NCHW, 1 = NC8; output only: 2 = split16 fp16 pairs)
  int out_s2d;              // NC8 output in space-to-depth order: pixel (y, x) of channel c -> channel ((y&1)*2 + (x&1))*Cout + c at
                            // (y/2, x/2) of a (4*Cout, Ho/2, Wo/2) NC8 tensor — what the stride-2 Winograd layers read (csrc/wino.hip)
  float out_scale;          // split16 output: stored value = result · out_scale
  int* status;              // context status word (split16 output: saturation flag)
  // Output remap of the final NCHW stores (not of split-K partials): the window [rm_cy, rm_cy + rm_hq) x [rm_cx, rm_cx + rm_wq) of
  // the (Ho, Wo) result goes to out[.., 2i + rm_py, 2j + rm_px] of a (rm_H, rm_W) plane; pixels outside the window are dropped.
  // This is how a parity class of a stride-2 data gradient lands in dx without a class buffer (deepim_conv2d_forward_remap).
  int rm_on, rm_cy, rm_cx, rm_hq, rm_wq, rm_py, rm_px, rm_H, rm_W;
  // Activation-gradient epilogue of a DATA GRADIENT (register-fed kernels and the split-K second passes only): the final NCHW store
  // becomes out = lrelu'(ep_y)·(v [+ ep_add]) with ep_y / ep_add laid out like out — the previous layer's saved output and the
  // gradient arriving over a skip connection. NULL ep_y = off.
  const float* ep_y;
  const float* ep_add;
  float ep_slope;
  const float* wd8;         // weights for NC8 inputs [Cout/32][group = (c8,ky,kx)][lane][4]
  const int2* tab8;         // per group: {byte offset (c8*H*W + ky*W + kx)*32, bit ky*8+kx}
};

struct Remap { int on, cy, cx, hq, wq, py, px, H, W, Wo; };
struct ActGrad { const float* y; const float* add; float slope; };   // ConvParams::ep_* of a launch
__host__ __device__ inline Remap remap_of(const ConvParams& p) {
  return {p.rm_on, p.rm_cy, p.rm_cx, p.rm_hq, p.rm_wq, p.rm_py, p.rm_px, p.rm_H, p.rm_W, p.Wo};
}
// pixel r of the (Ho, Wo) result → offset inside a channel plane of the destination and that plane's size; false = dropped
__device__ __forceinline__ bool remap_pixel(const Remap& m, int r, int hw, long& poff, long& plane) {
  if (!m.on) { poff = r; plane = hw; return true; }
  const int ho = r / m.Wo, wo = r - ho * m.Wo;
  const int i = ho - m.cy, j = wo - m.cx;
  if ((unsigned)i >= (unsigned)m.hq || (unsigned)j >= (unsigned)m.wq) return false;
  poff = (long)(2 * i + m.py) * m.W + 2 * j + m.px;
  plane = (long)m.H * m.W;
  return true;
}

template <int BM, int BN, int MODE, int NT = 256>
__global__ __launch_bounds__(NT, 4) void conv_mfma_kernel(ConvParams p) {
  constexpr int WM = BM / (NT / 128), WN = BN / 2;  // wave tile: waves form a (NT/128) x 2 grid over the block tile
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NG = BM / GRAN;            // weight granules per block
  constexpr int NGT = NG * 256 / NT;       // ... of which each thread copies this many (one dwordx4 each)
  constexpr int EB = KT * BN / NT;         // gathered elements per thread per chunk (consecutive k rows)
  constexpr int KS = KT / 2;               // MFMA k-steps per chunk
  constexpr int MPS = TM * TN;             // MFMAs per k-step
  constexpr int Q = KS * MPS;              // issue slots per chunk (one per MFMA)
  __shared__ __attribute__((aligned(16))) float As[2][KT * BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][KT * BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  // XCD-aware tile order: hardware sends block b to XCD b % 8; remap so that each XCD works on a
  // contiguous run of (pixel-tile fastest) virtual ids — neighbours share weights (same M tile / K slice)
  // and halo rows in that XCD's L2 instead of every XCD streaming every weight slab.
  int vid;
  {
    const int total = p.gx * p.gy * p.gz, bid = blockIdx.x;
    const int xcd = bid & 7, qn = total >> 3, rn = total & 7;
    vid = p.swizzle ? xcd * qn + min(xcd, rn) + (bid >> 3) : bid;
  }
  const int bx = vid % p.gx;
  const int mb = (vid / p.gx) % p.gy;      // M tile
  int zz = vid / (p.gx * p.gy);
  const long n0 = (long)bx * BN;           // first pixel of the tile
  const int split = zz % p.ksplit;         // split-K slice
  zz /= p.ksplit;                          // deconv parity class

  // ---- per-thread gather state (fixed for the whole K loop) ----
  // voff: byte offset of the pixel's tap origin inside the (shifted) input buffer descriptor, or the
  // out-of-range marker; m64: bit (ky*8+kx) set when tap (ky,kx) lies inside the image (zero padding).
  const int gp = tid & (BN - 1);
  const int krow0 = __builtin_amdgcn_readfirstlane((tid / BN) * EB);
  const long pix = n0 + gp;
  int voff = 0;  // < 2^31 always; bit 31 set by GATHER marks an invalid tap (or a thread beyond the last pixel)
  unsigned long long m64 = 0;
  int par_y = 0, par_x = 0;
  long npix = p.npix;
  if (MODE == MODE_DECONV) {
    par_y = (zz >> 1); par_x = (zz & 1);
    const int oy0 = (par_y - p.crop_y) & 1, ox0 = (par_x - p.crop_x) & 1;
    npix = (long)p.B * ((p.Ho - oy0 + 1) >> 1) * ((p.Wo - ox0 + 1) >> 1);
  }
  if (pix < npix) {
    unsigned mky = 0, mkx = 0;
    if (MODE == MODE_CONV) {
      const int hw = p.Ho * p.Wo;
      const int n = (int)(pix / hw);
      const int r = (int)(pix - (long)n * hw);
      const int ho = r / p.Wo, wo = r - ho * p.Wo;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      voff = (n * p.Cin * p.H * p.W + hi0 * p.W + wi0) * 4 + p.pad_bytes;
#pragma unroll
      for (int k = 0; k < 7; ++k) {  // kernels are <= 7x7: row/col 7 stay clear, so bit 63 (padded k) is never valid
        if (hi0 + k >= 0 && hi0 + k < p.H) mky |= 1u << k;
        if (wi0 + k >= 0 && wi0 + k < p.W) mkx |= 1u << k;
      }
    } else {
      // parity class (par_y,par_x) of the UNCROPPED deconv output y = yo + crop_y.
      // class pixels: yo = 2*qy + oy0 where oy0 makes (yo+crop_y)&1 == par_y.
      const int oy0 = (par_y - p.crop_y) & 1, ox0 = (par_x - p.crop_x) & 1;
      const int nqy = (p.Ho - oy0 + 1) >> 1, nqx = (p.Wo - ox0 + 1) >> 1;
      const int hw = nqy * nqx;
      const int n = (int)(pix / hw);
      const int r = (int)(pix - (long)n * hw);
      const int qy = r / nqx, qx = r - qy * nqx;
      const int y = 2 * qy + oy0 + p.crop_y, x = 2 * qx + ox0 + p.crop_x;  // uncropped coords
      // taps: ky = par_y + 2*jy (jy∈{0,1}) ↔ iy = (y - ky)/2 = (y>>1) - jy ; tap origin = (iy0-1, ix0-1)
      const int iy0 = (y >> 1), ix0 = (x >> 1);
      voff = (n * p.Cin * p.H * p.W + (iy0 - 1) * p.W + (ix0 - 1)) * 4 + p.pad_bytes;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (iy0 - j >= 0 && iy0 - j < p.H) mky |= 1u << j;
        if (ix0 - j >= 0 && ix0 - j < p.W) mkx |= 1u << j;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((mky >> k) & 1u) m64 |= (unsigned long long)mkx << (8 * k);
  }
  const unsigned long long ninv64 = ~m64;   // inverted validity bits; shifted by the s_load-ed tap index (no SALU in between)
  // raw buffer over the input, base shifted down by pad_bytes so border tap origins stay non-negative
  // (only offsets >= pad_bytes, i.e. addresses inside the tensor, are ever dereferenced)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.in - p.pad_bytes), 0, (int)(p.in_bytes + (unsigned)p.pad_bytes), 0x00020000);

  const int gt0 = (tid >> 8) * NGT, ta = tid & 255;   // this thread's first granule and its dwordx4 slot in the slab
  const float* wblk = p.wp + ((long)mb * NG + gt0) * p.nchunk * (KT * GRAN) + ta * 4;
  if (MODE == MODE_DECONV) wblk += (long)zz * p.ngran * p.nchunk * (KT * GRAN);
  const int kc_begin = split * p.chunks_per_split;
  const int kc_end = min(p.nchunk, kc_begin + p.chunks_per_split);

  // LDS slots this thread fills
  const int a_st = ((ta * 4) / GRAN) * BM + gt0 * GRAN + (ta * 4) % GRAN;
  const int b_st = krow0 * BN + gp;

  float4 areg0 = make_float4(0, 0, 0, 0), areg1 = areg0;  // named scalars: an indexed array lands in scratch
  float breg[EB];

// Zero padding: an out-of-range voffset makes the buffer load return 0.0 without touching memory.
// Tap validity lives in a 64-bit word of INVERTED bits; the wave-uniform tap index comes straight from the scalar
// cache into a 64-bit VALU shift (no SALU-produced operand: see tools/mfma_issue_probe.hip), then shift-or puts
// the "invalid" bit into bit 31 of the voffset (>= num_records → hardware returns 0).
#define GATHER(e, tq)                                                                                  \
  {                                                                                                    \
    const unsigned inv = (unsigned)(ninv64 >> (tq)[e].y);                                              \
    breg[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)((inv << 31) | (unsigned)voff), (tq)[e].x, 0)); \
  }
#define BSEL(e) breg[e]
#define ALOAD0(kc) areg0 = *reinterpret_cast<const float4*>(wblk + (long)(kc) * (KT * GRAN));
#define ALOAD1(kc) areg1 = *reinterpret_cast<const float4*>(wblk + ((long)p.nchunk + (kc)) * (KT * GRAN));

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lrow = lane >> 5, lcol = lane & 31;
  int2 tq[EB];  // tap-table entries (wave-uniform → SGPRs) of the chunk gathered next, loaded one chunk ahead
  const int kc_last = kc_end - 1;
  if (kc_begin < kc_end) {
    // chunk kc_begin → LDS[0]; chunk kc_begin+1 → registers (stored during the first loop iteration)
    const int2* tp0 = p.tab + kc_begin * KT + krow0;
#pragma unroll
    for (int e = 0; e < EB; ++e) tq[e] = tp0[e];
#pragma unroll
    for (int e = 0; e < EB; ++e) GATHER(e, tq)
    ALOAD0(kc_begin)
    if (NGT == 2) ALOAD1(kc_begin)
    *reinterpret_cast<float4*>(&As[0][a_st]) = areg0;
    if (NGT == 2) *reinterpret_cast<float4*>(&As[0][a_st + GRAN]) = areg1;
#pragma unroll
    for (int e = 0; e < EB; ++e) Bs[0][b_st + e * BN] = BSEL(e);
    const int k1 = min(kc_begin + 1, kc_last);
    const int2* tp1 = p.tab + k1 * KT + krow0;
#pragma unroll
    for (int e = 0; e < EB; ++e) tq[e] = tp1[e];
#pragma unroll
    for (int e = 0; e < EB; ++e) GATHER(e, tq)
    ALOAD0(k1)
    if (NGT == 2) ALOAD1(k1)
    const int2* tp2 = p.tab + min(kc_begin + 2, kc_last) * KT + krow0;
#pragma unroll
    for (int e = 0; e < EB; ++e) tq[e] = tp2[e];
  }
  __syncthreads();

  // Main loop. One issue slot after every MFMA, fenced with sched_barrier(0) so hipcc keeps the order:
  // the wave's own gather (tap decode + buffer_load), LDS fragment reads and LDS stores are issued in the
  // shadow of its MFMAs (64 cycles each) instead of before/after the MFMA block.
  // Software pipeline, two chunks deep with one register set: while chunk kc is multiplied out of
  // LDS[buf], element e (slot e*Q/EB) first stores the value gathered during chunk kc-1 (it belongs to
  // chunk kc+1) into LDS[buf^1], then re-issues its gather for chunk kc+2 — a full chunk (≈2000 cycles)
  // of latency tolerance per load. Fragment reads of k-step s+1 are issued at slot s*MPS.
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const int buf = (kc - kc_begin) & 1;
    const int kg = min(kc + 2, kc_last);                               // chunk gathered in this iteration
    const int2* tpn = p.tab + min(kc + 3, kc_last) * KT + krow0;       // its successor's tap table
    const float* as = &As[buf][lrow * BM + wm0 + lcol];
    const float* bs = &Bs[buf][lrow * BN + wn0 + lcol];
    float* asn = &As[buf ^ 1][a_st];
    float* bsn = &Bs[buf ^ 1][b_st];
    float av[2][TM], bv[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[0][i] = as[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[0][j] = bs[j * 32];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int m = 0; m < MPS; ++m) {
        const int i = m / TN, j = m % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][i], bv[s & 1][j], acc[i][j], 0, 0, 0);
        const int q = s * MPS + m;
        if (m == 0 && s + 1 < KS) {
#pragma unroll
          for (int ii = 0; ii < TM; ++ii) av[(s + 1) & 1][ii] = as[(s + 1) * 2 * BM + ii * 32];
#pragma unroll
          for (int jj = 0; jj < TN; ++jj) bv[(s + 1) & 1][jj] = bs[(s + 1) * 2 * BN + jj * 32];
        }
        // one event per slot where the chunk has room (Q >= 4*EB): reads 4s, store(e) 4e+1, gather(e) 4e+2,
        // weight slabs: store 3 / 11, load 7 / 15 — every slot stays well inside the 64-cycle MFMA shadow
        constexpr bool ROOMY = Q >= 4 * EB;
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          const int qs = ROOMY ? 4 * e + 1 : e * Q / EB + (MPS > 1 ? 1 : 0);
          const int qg = ROOMY ? 4 * e + 2 : qs;
          if (qs == q) bsn[e * BN] = BSEL(e);
          if (qg == q) GATHER(e, tq)
        }
        if (q == (ROOMY ? 3 : 2)) *reinterpret_cast<float4*>(asn) = areg0;
        if (q == (ROOMY ? 7 : 2)) ALOAD0(kg)
        if (NGT == 2 && q == (ROOMY ? 11 : 6)) *reinterpret_cast<float4*>(asn + GRAN) = areg1;
        if (NGT == 2 && q == (ROOMY ? 15 : 6)) ALOAD1(kg)
        if (q == Q - 1) {
#pragma unroll
          for (int e = 0; e < EB; ++e) tq[e] = tpn[e];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
#undef GATHER
#undef BSEL
#undef ALOAD0
#undef ALOAD1

  // ---- epilogue: bias + LeakyReLU, NCHW store (or raw partial sums for split-K) ----
  const bool partial = p.ksplit > 1;
  float* outp = partial ? p.partial + (long)split * p.partial_stride : p.out;
  const int ctotal = partial ? p.Cout : p.out_ctotal;
  const int coff = partial ? 0 : p.out_coff;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long op = n0 + wn0 + j * 32 + lcol;
    if (op >= npix) continue;
    long obase;
    long cstride = (long)p.Ho * p.Wo;
    if (MODE == MODE_CONV) {
      const int hw = p.Ho * p.Wo;
      const int n = (int)(op / hw);
      const int r = (int)(op - (long)n * hw);
      long poff = r;
      if (!partial && !remap_pixel(remap_of(p), r, hw, poff, cstride)) continue;
      obase = ((long)n * ctotal + coff) * cstride + poff;
    } else {
      const int oy0 = (par_y - p.crop_y) & 1, ox0 = (par_x - p.crop_x) & 1;
      const int nqy = (p.Ho - oy0 + 1) >> 1, nqx = (p.Wo - ox0 + 1) >> 1;
      const int hwq = nqy * nqx;
      const int n = (int)(op / hwq);
      const int r = (int)(op - (long)n * hwq);
      const int qy = r / nqx, qx = r - qy * nqx;
      obase = ((long)n * ctotal + coff) * p.Ho * p.Wo + (long)(2 * qy + oy0) * p.Wo + (2 * qx + ox0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = mb * BM + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
        if (co < p.Cout) {
          float v = acc[i][j][r];
          if (!partial) {
            v = v + (p.bias ? p.bias[co] : 0.f);
            v = v > 0.f ? v : v * p.slope;
          }
          outp[obase + co * cstride] = v;
        }
      }
    }
  }
}



// Epilogue for channel-blocked output [n][Cout/8][ho][wo][8]: the 16 accumulator registers of a lane are 4 groups of 4
// consecutive channels (rows 8g + 4*lrow + 0..3 of the 32-row MFMA tile) of ONE pixel → four 16-byte stores per tile
// instead of sixteen scattered dwords. Cout must be a multiple of 8.
template <int TM, int TN>
__device__ __forceinline__ void store_tile_nc8(f32x16 (&acc)[TM][TN], float* __restrict__ outp, const ConvParams& p,
                                               bool partial, int co0, long pix0, int lrow, int lcol) {
  const int hw = p.Ho * p.Wo, c8n = p.Cout >> 3;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long op = pix0 + j * 32 + lcol;
    if (op >= p.npix) continue;
    const int n = (int)(op / hw);
    int r0 = (int)(op - (long)n * hw);
    long nbase = (long)n * c8n;      // first channel block of sample n, in blocks of hw pixels
    if (p.out_s2d) {                 // (block, pixel) -> (phase*c8n + block, (y/2, x/2)) of a tensor with hw/4 pixels per block
      const int ho = r0 / p.Wo, wo = r0 - ho * p.Wo;
      nbase = (nbase * 4 + ((ho & 1) * 2 + (wo & 1)) * c8n) * (hw >> 2);
      r0 = (ho >> 1) * (p.Wo >> 1) + (wo >> 1);
    } else {
      nbase *= hw;
    }
    const int hwb = p.out_s2d ? hw >> 2 : hw;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = ((co0 + i * 32) >> 3) + g;
        if (cb >= c8n) continue;
        float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        if (!partial) {
          const int c = cb * 8 + 4 * lrow;
          const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0, 0, 0, 0);
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope;
          v.z = v.z > 0.f ? v.z : v.z * p.slope; v.w = v.w > 0.f ? v.w : v.w * p.slope;
        }
        *reinterpret_cast<float4*>(outp + (nbase + (long)cb * hwb + r0) * 8 + 4 * lrow) = v;
      }
    }
  }
}

// Epilogue for "split16" output (the activation format of the split-fp16 conv path, csrc/conv_f16.hip): NHWC, per 16 channels
// a 32-half record [hi 0..15 | lo 0..15] with hi = f16(v·scale), lo = f16(v·scale − hi). Same ownership as the NC8 epilogue:
// a lane holds 4 runs of 4 consecutive channels of one pixel → one 8-byte hi and one 8-byte lo store per run.
template <int TM, int TN>
__device__ __forceinline__ void store_tile_split16(f32x16 (&acc)[TM][TN], const ConvParams& p, int co0, long pix0, int lrow,
                                                   int lcol) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  _Float16* outh = reinterpret_cast<_Float16*>(p.out);
  float amax = 0.f;   // largest scaled magnitude: a clamp is reported once per thread (bit DI_STATUS_X3_SATURATED)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long op = pix0 + j * 32 + lcol;
    if (op >= p.npix) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = co0 + i * 32 + g * 8 + 4 * lrow;
        if (c >= p.Cout) continue;
        h4 vh, vl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][j][4 * g + r] + (p.bias ? p.bias[c + r] : 0.f);
          v = v > 0.f ? v : v * p.slope;
          v *= p.out_scale;
          amax = fmaxf(amax, fabsf(v));
          v = fminf(fmaxf(v, -60000.f), 60000.f);
          const _Float16 h = (_Float16)v;
          vh[r] = h;
          vl[r] = (_Float16)(v - (float)h);
        }
        _Float16* rec = outh + op * (2 * p.Cout) + (c >> 4) * 32 + (c & 15);
        *reinterpret_cast<h4*>(rec) = vh;
        *reinterpret_cast<h4*>(rec + 16) = vl;
      }
    }
  }
  if (amax > 60000.f) atomicOr(p.status, DI_STATUS_X3_SATURATED);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-free variant for the encoder convolutions (MODE_CONV, even Cin): every wave feeds its MFMAs straight from
// registers loaded with coalesced buffer loads — no LDS staging, no block barrier, four fully independent waves per
// block that only share L1/L2 lines.
//   * K order is (ci/2, ky, kx, ci%2): the two k of one v_mfma_f32_32x32x2_f32 are the SAME tap of two adjacent input
//     channels, so lanes 32-63 (k+1) differ from lanes 0-31 (k) by a constant plane stride folded into their voffset
//     and share the padding-validity bit — the per-load address math is the same 3 VALU as in the LDS kernel.
//   * A operand: weights pre-packed [Cout/32][k-pair/4][lane][4] → one b128 load per 4 k-steps, 1 KB contiguous per wave.
//   * B operand: lane j of a half-wave reads pixel j of 32 consecutive output pixels at one tap: 128 B (stride 1) or
//     256 B (stride 2) runs; neighbouring taps and the sibling wave re-hit the same lines in L1.
//   * software pipeline: a ring of 4 k-steps of B operands in registers, loads issued 3 k-steps (12 MFMAs) ahead;
//     A operands double-buffered per half chunk; at most one load per MFMA issue slot (20 loads per 32 MFMAs).
// The accumulation order per output is the fmaf chain over that K order (results differ from the (ci,ky,kx) kernel in
// the last bits; the oracle has the matching order switch).
constexpr int DK = 8;  // k-pairs per chunk: the same 16-wide K chunks the split-K plumbing counts in
#ifndef DIRECT_RING
#define DIRECT_RING 4   // operand ring (k-steps); loads run DIRECT_RING-1 k-steps ahead of the MFMAs
#endif
#ifndef DIRECT_OCC
#define DIRECT_OCC 3
#endif
constexpr int DR = DIRECT_RING, DPD = DR - 1;
static_assert(DK % DR == 0 && DPD < DK, "ring must divide the chunk");

// (Tried: single-wave workgroups, one 64x64 quadrant each, for 4x finer tail granularity — 60 TF, half the rate.)
// WMW = wave rows: 2 → waves 2x2 over a 128x128 tile; 1 → waves 1x4 over a 64x256 tile (Cout <= 64: conv1)
template <int WMW>
__device__ __forceinline__ void conv_direct_body(const ConvParams& p, const int block_id) {
  constexpr int BM = 64 * WMW, BN = 64 * (4 / WMW), TM = 2, TN = 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = WMW == 2 ? (wave >> 1) * 64 : 0, wn0 = WMW == 2 ? (wave & 1) * 64 : wave * 64;
  int vid, split, tail_item = -1;
  if (p.tail_s > 0) {
    // every XCD (block b → XCD b % 8) first walks its eighth of the full tiles, then its eighth of the tail items,
    // so the short items are dispatched last on all XCDs and fill the round the full tiles leave under-used
    const int bid = block_id, xcd = bid & 7, idx = bid >> 3;
    const int fq = p.n_full >> 3, tq8 = p.n_tail_pad >> 3;
    if (idx < fq) {
      vid = xcd * fq + idx;
      split = 0;
    } else {
      tail_item = xcd * tq8 + (idx - fq);
      if (tail_item >= (p.gx * p.gy - p.n_full) * p.tail_s) return;
      vid = p.n_full + tail_item / p.tail_s;
      split = tail_item % p.tail_s;
    }
  } else {
    const int total = p.gx * p.gy * p.gz, bid = block_id;
    const int xcd = bid & 7, qn = total >> 3, rn = total & 7;
    vid = p.swizzle ? xcd * qn + min(xcd, rn) + (bid >> 3) : bid;
    split = vid / (p.gx * p.gy);
  }
  const int tvid = vid % (p.gx * p.gy);
  const int bx = tvid % p.gx;      // pixel tiles fastest (M-fastest order measured the same)
  const int mb = tvid / p.gx;
  const long n0 = (long)bx * BN;
  const int lrow = lane >> 5, lcol = lane & 31;
  const long npix = p.npix;

  int voff[TN];
  unsigned long long ninv[TN];   // inverted tap-validity bits (bit ky*8+kx)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long pix = n0 + wn0 + j * 32 + lcol;
    unsigned long long m64 = 0;
    voff[j] = 0;
    if (pix < npix) {
      const int hw = p.Ho * p.Wo;
      const int n = (int)(pix / hw);
      const int r = (int)(pix - (long)n * hw);
      const int ho = r / p.Wo, wo = r - ho * p.Wo;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      voff[j] = ((n * p.Cin + lrow) * p.H * p.W + hi0 * p.W + wi0) * 4 + p.pad_bytes;
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
    ninv[j] = ~m64;
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.in - p.pad_bytes), 0, (int)(p.in_bytes + (unsigned)p.pad_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wd, 0, (int)p.wd_bytes, 0x00020000);
  const int npair = p.nchunk * DK;
  int wvo[TM];   // A operand: [32-row tile][4 k-steps][lane][4] → one b128 load per tile per 4 k-steps, 1 KB contiguous per wave
#pragma unroll
  for (int i = 0; i < TM; ++i) wvo[i] = (((mb * (BM / 32) + (wm0 >> 5) + i) * (npair / 4)) * 64 + lane) * 16;

  int kc_begin = split * p.chunks_per_split;
  int kc_end = min(p.nchunk, kc_begin + p.chunks_per_split);
  if (p.tail_s > 0) {
    kc_begin = tail_item < 0 ? 0 : split * p.tail_cps;
    kc_end = tail_item < 0 ? p.nchunk : min(p.nchunk, kc_begin + p.tail_cps);
  }
  kc_begin = __builtin_amdgcn_readfirstlane(kc_begin);   // block-uniform by construction; keeps the table pointers in SGPRs
  kc_end = __builtin_amdgcn_readfirstlane(kc_end);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float rb[DR][TN];
  float4 aq[2][TM];   // A operands of the two 4-k-step halves of a chunk
// The tap bit index goes from the scalar cache straight into the VALU shift: a VALU op that reads an SGPR produced by
// an SALU op (s_and / s_bitcmp+s_cselect) inside the MFMA shadow stalls the SIMD's issue (tools/mfma_issue_probe.hip:
// 155 → 115-120 TF), SGPRs written by s_load do not.
#ifndef DIRECT_ABL
#define DIRECT_ABL 0   // dev ablations (wrong results): 1 = B loads from one L1-resident line, 2 = A loads likewise, 4 = no tap decode
#endif
#define DLOADB(slot, j, ent)                                                                            \
  {                                                                                                     \
    const unsigned inv = (DIRECT_ABL & 4) ? 0u : (unsigned)(ninv[j] >> ((DIRECT_ABL & 8) ? p.pad * 9 : (ent).y)); \
    const int vo_ = (DIRECT_ABL & 1) ? lane * 4 + p.pad_bytes : (int)((inv << 31) | (unsigned)voff[j]); \
    /* 8: real per-lane pattern (line straddling, stride) but one fixed tap, so the lines stay L1-resident */ \
    rb[slot][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo_, (DIRECT_ABL & 9) ? ((DIRECT_ABL & 8) ? p.pad_bytes : 0) : (ent).x, 0)); \
  }
#define DLOADA(buf, i, hc) \
  aq[buf][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrw, (DIRECT_ABL & 2) ? lane * 16 : wvo[i], (DIRECT_ABL & 2) ? 0 : (hc) * 1024, 0));
#define ASEL(u, i) (((u) & 3) == 0 ? aq[((u) >> 2) & 1][i].x : ((u) & 3) == 1 ? aq[((u) >> 2) & 1][i].y : \
                    ((u) & 3) == 2 ? aq[((u) >> 2) & 1][i].z : aq[((u) >> 2) & 1][i].w)

  if (kc_begin < kc_end) {
    // tap-table entries travel through the scalar cache: 8 entries (16 dwords) per chunk, fetched one chunk ahead.
    // (inline asm: after an `asm volatile` hipcc no longer proves the table unclobbered and would fall back to
    // vector loads + waterfall loops)
    const int g0 = kc_begin * DK;
    const int hc_last = p.nchunk * 2 - 1;
    i32x16 tq, tn;
    {
      const int2* t0 = p.tab2 + g0;
      asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(tn) : "s"(t0));
#pragma unroll
      for (int u = 0; u < DPD; ++u) {
        const int2 ent = make_int2(tn[2 * u], tn[2 * u + 1]);
        DLOADB(u, 0, ent) DLOADB(u, 1, ent)
      }
      DLOADA(0, 0, g0 / 4) DLOADA(0, 1, g0 / 4)
      const int2* t1 = p.tab2 + g0 + DPD;
      asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(tq) : "s"(t1));
    }
    for (int kc = kc_begin; kc < kc_end; ++kc) {
      const int g = kc * DK;
      const int2* t2 = p.tab2 + g + DK + DPD;   // entries of the loads the NEXT chunk issues (table is padded past the end)
      asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(tn) : "s"(t2));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < DK; ++u) {
        const int sl = u % DR, ld = (u + DPD) % DR;
        const int2 ent = make_int2(tq[2 * u], tq[2 * u + 1]);
        // the empty asm pins each MFMA inside its slot (a pure intrinsic would otherwise sink past the fences)
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ASEL(u, 0), rb[sl][0], acc[0][0], 0, 0, 0);
        asm volatile("" : "+v"(acc[0][0]));
        DLOADB(ld, 0, ent)
        __builtin_amdgcn_sched_barrier(0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ASEL(u, 0), rb[sl][1], acc[0][1], 0, 0, 0);
        asm volatile("" : "+v"(acc[0][1]));
        DLOADB(ld, 1, ent)
        __builtin_amdgcn_sched_barrier(0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ASEL(u, 1), rb[sl][0], acc[1][0], 0, 0, 0);
        asm volatile("" : "+v"(acc[1][0]));
        // next half's weights, 4 k-steps ahead; clamped: the soffset of a raw buffer load is not range-checked, and the
        // half after the last one of the last M tile would lie past the packed buffer
        if ((u & 3) == 0) DLOADA(((u >> 2) + 1) & 1, 0, min((g + u) / 4 + 1, hc_last))
        __builtin_amdgcn_sched_barrier(0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ASEL(u, 1), rb[sl][1], acc[1][1], 0, 0, 0);
        asm volatile("" : "+v"(acc[1][1]));
        if ((u & 3) == 0) DLOADA(((u >> 2) + 1) & 1, 1, min((g + u) / 4 + 1, hc_last))
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tn));
      tq = tn;
    }
  }
#undef DLOADA
#undef ASEL
#undef DLOADB

  if (tail_item >= 0) {   // raw partial sums, tile-local layout [slice][remainder tile][co 128][px 128]
    const int R = p.gx * p.gy - p.n_full;
    float* tp = p.tail_partial + ((long)split * R + (vid - p.n_full)) * (BM * BN);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tp[(wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow) * BN + wn0 + j * 32 + lcol] = acc[i][j][r];
    return;
  }
  const bool partial = p.ksplit > 1;
  float* outp = partial ? p.partial + (long)split * p.partial_stride : p.out;
  const int ctotal = partial ? p.Cout : p.out_ctotal;
  const int coff = partial ? 0 : p.out_coff;
  if (p.out_nc8 == 2) {   // split16 (never with split-K: the launcher keeps ksplit = 1 for this layout)
    store_tile_split16<TM, TN>(acc, p, mb * BM + wm0, n0 + wn0, lrow, lcol);
    return;
  }
  if (p.out_nc8) {
    store_tile_nc8<TM, TN>(acc, outp, p, partial, mb * BM + wm0, n0 + wn0, lrow, lcol);
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long op = n0 + wn0 + j * 32 + lcol;
    if (op >= npix) continue;
    const int hw = p.Ho * p.Wo;
    const int n = (int)(op / hw);
    const int r0 = (int)(op - (long)n * hw);
    long poff = r0, plane = hw;
    if (!partial && !remap_pixel(remap_of(p), r0, hw, poff, plane)) continue;
    const long obase = ((long)n * ctotal + coff) * plane + poff;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = mb * BM + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
        if (co < p.Cout) {
          float v = acc[i][j][r];
          if (!partial) {
            v = v + (p.bias ? p.bias[co] : 0.f);
            v = v > 0.f ? v : v * p.slope;
            if (p.ep_y) {
              if (p.ep_add) v += p.ep_add[obase + (long)co * plane];
              v = p.ep_y[obase + (long)co * plane] > 0.f ? v : v * p.ep_slope;
            }
          }
          outp[obase + (long)co * plane] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------

#ifndef DIRECT_OCC1
#define DIRECT_OCC1 DIRECT_OCC   // conv1's 64x256-tile instance on its own: dev builds try 5 blocks per CU (<= 96 VGPRs)
#endif
template <int WMW>
__global__ __launch_bounds__(256, WMW == 1 ? DIRECT_OCC1 : DIRECT_OCC) void conv_direct_kernel(ConvParams p) {
  conv_direct_body<WMW>(p, blockIdx.x);
}

// Several independent convolutions in ONE launch (the four output parity classes of a stride-2 data gradient: same input, own
// sub-kernel, own K, own remap): block b belongs to the last convolution whose first block is <= b. Small layers fill the chip
// together instead of one under-filled launch (+ split-K pass) per class.
constexpr int CONV_GROUP_MAX = 4;
struct ConvGroup {
  ConvParams p[CONV_GROUP_MAX];
  int start[CONV_GROUP_MAX + 1];   // first block of each member (multiples of 8: the XCD round-robin stays aligned); start[n] = grid
  int n;
};
template <int WMW>
__global__ __launch_bounds__(256, DIRECT_OCC) void conv_direct_group_kernel(ConvGroup g) {
  const int b = blockIdx.x;
  int c = 0;
#pragma unroll
  for (int i = 1; i < CONV_GROUP_MAX; ++i)
    if (i < g.n && b >= g.start[i]) c = i;
  c = __builtin_amdgcn_readfirstlane(c);
  const int local = b - g.start[c];
  if (local >= g.p[c].gx * g.p[c].gy * g.p[c].gz) return;   // the padding up to the next multiple of 8
  conv_direct_body<WMW>(g.p[c], local);
}

// The same LDS-free design on channel-blocked ("NC8") activations [n][C/8][h][w][8] — the layout the encoder layers
// hand to each other (conv1 writes it, conv6_1 returns to NCHW for fc6). K runs (c8, ky, kx, s, h) with channel
// c8*8 + s + 4h: the four k-steps s of one (c8, tap) group pair channels (s, s+4), so lanes 0-31 read channels 0-3 and
// lanes 32-63 channels 4-7 of the 32-byte pixel record — ONE dwordx4 per lane feeds four k-steps, and a wave reads
// 1 KB (stride 1) of fully used, contiguous bytes per load. 4 loads (2 activation, 2 weight, all b128) per 16 MFMAs
// instead of 10; operands of 4 groups in flight (ring of 4, loads issued 3 groups = 48 MFMAs ahead).
#ifndef NC8_RING
#define NC8_RING 4
#endif
#ifndef NC8_OCC
#define NC8_OCC 3
#endif
// (Tried: 64x128 tiles on 128-thread blocks for layers whose 128x128 tile count divides badly over 256 CUs, e.g. 2400 =
// 9.375 per CU — identical throughput, so per-CU tile quantisation is not what separates conv3 (139 TF) from conv2 (144).
// Note the 2nd __launch_bounds__ argument is waves per SIMD in HIP, not blocks per CU.)
// WIDE: 64 x 256 block tile (the four waves side by side along the pixels, all on the same 64 rows) for Cout = 64 — conv1 on
// the channel-blocked net input the zoom front end writes (round 3).
template <int OUT_NC8, int WIDE = 0>
__global__ __launch_bounds__(256, NC8_OCC) void conv_nc8_kernel(ConvParams p) {
  constexpr int BM = WIDE ? 64 : 128, BN = WIDE ? 256 : 128, TM = 2, TN = 2, NG = NC8_RING, NPD = NG - 1;   // NG: groups per loop body = ring size
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = WIDE ? 0 : (wave >> 1) * 64, wn0 = WIDE ? wave * 64 : (wave & 1) * 64;
  int vid;
  {
    const int total = p.gx * p.gy * p.gz, bid = blockIdx.x;
    const int xcd = bid & 7, qn = total >> 3, rn = total & 7;
    vid = p.swizzle ? xcd * qn + min(xcd, rn) + (bid >> 3) : bid;
  }
  const int bx = vid % p.gx;
  const int mb = (vid / p.gx) % p.gy;
  const int split = vid / (p.gx * p.gy);
  const long n0 = (long)bx * BN;
  const int lrow = lane >> 5, lcol = lane & 31;
  const int pad8 = p.pad_bytes * 8;   // descriptor base shift in bytes of the blocked layout

  int voff[TN];
  unsigned long long ninv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long pix = n0 + wn0 + j * 32 + lcol;
    unsigned long long m64 = 0;
    voff[j] = 0;
    if (pix < p.npix) {
      const int hw = p.Ho * p.Wo;
      const int n = (int)(pix / hw);
      const int r = (int)(pix - (long)n * hw);
      const int ho = r / p.Wo, wo = r - ho * p.Wo;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      voff[j] = (n * (p.Cin >> 3) * p.H * p.W + hi0 * p.W + wi0) * 32 + lrow * 16 + pad8;
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
    ninv[j] = ~m64;
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.in - pad8), 0, (int)(p.in_bytes + (unsigned)pad8), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wd8, 0, (int)p.wd_bytes, 0x00020000);
  const int ngroup = p.nchunk * 2;
  int wvo[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) wvo[i] = (((mb * (BM / 32) + (wm0 >> 5) + i) * ngroup) * 64 + lane) * 16;

  // chunks_per_split is even on this path, so a block always owns whole bodies of NG = 4 groups
  const int g_begin = __builtin_amdgcn_readfirstlane(split * p.chunks_per_split * 2);
  const int g_end = __builtin_amdgcn_readfirstlane(min(p.nchunk, (split + 1) * p.chunks_per_split) * 2);
  const int g_last = ngroup - 1;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 bq[NG][TN], aq[NG][TM];
#define NLOADB(slot, j, off, bit)                                                                        \
  {                                                                                                      \
    const unsigned inv = (unsigned)(ninv[j] >> (bit));                                                   \
    bq[slot][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((inv << 31) | (unsigned)voff[j]), (off), 0)); \
  }
#define NLOADA(slot, i, g) \
  aq[slot][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrw, wvo[i], min((g), g_last) * 1024, 0));
#define QSEL(v, s) ((s) == 0 ? (v).x : (s) == 1 ? (v).y : (s) == 2 ? (v).z : (v).w)

  if (g_begin < g_end) {
    // tap-table entries of 8 groups (16 dwords) per scalar load: [0..5] feed this body's loads (groups +3..+5 are the next
    // body's first three), fetched one body ahead
    i32x16 tq, tn;
    {
      const int2* t0 = p.tab8 + g_begin;
      asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(tn) : "s"(t0));
#pragma unroll
      for (int u = 0; u < NPD; ++u) {
        NLOADB(u, 0, tn[2 * u], tn[2 * u + 1]) NLOADB(u, 1, tn[2 * u], tn[2 * u + 1])
        NLOADA(u, 0, g_begin + u) NLOADA(u, 1, g_begin + u)
      }
      const int2* t1 = p.tab8 + g_begin + NPD;
      asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(tq) : "s"(t1));
    }
    for (int g = g_begin; g < g_end; g += NG) {
      const int2* t2 = p.tab8 + g + NG + NPD;   // table is padded past the end
      asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(tn) : "s"(t2));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        const int ld = (q + NPD) % NG;   // slot of the group loaded while group q is multiplied
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(QSEL(aq[q][0], s_), QSEL(bq[q][0], s_), acc[0][0], 0, 0, 0);
          asm volatile("" : "+v"(acc[0][0]));
          if (s_ == 1) NLOADB(ld, 0, tq[2 * q], tq[2 * q + 1])
          __builtin_amdgcn_sched_barrier(0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(QSEL(aq[q][0], s_), QSEL(bq[q][1], s_), acc[0][1], 0, 0, 0);
          asm volatile("" : "+v"(acc[0][1]));
          if (s_ == 1) NLOADB(ld, 1, tq[2 * q], tq[2 * q + 1])
          __builtin_amdgcn_sched_barrier(0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(QSEL(aq[q][1], s_), QSEL(bq[q][0], s_), acc[1][0], 0, 0, 0);
          asm volatile("" : "+v"(acc[1][0]));
          if (s_ == 1) NLOADA(ld, 0, g + q + NPD)
          __builtin_amdgcn_sched_barrier(0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(QSEL(aq[q][1], s_), QSEL(bq[q][1], s_), acc[1][1], 0, 0, 0);
          asm volatile("" : "+v"(acc[1][1]));
          if (s_ == 1) NLOADA(ld, 1, g + q + NPD)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tn));
      tq = tn;
    }
  }
#undef NLOADB
#undef NLOADA
#undef QSEL

  const bool partial = p.ksplit > 1;
  float* outp = partial ? p.partial + (long)split * p.partial_stride : p.out;
  if (OUT_NC8) {
    store_tile_nc8<TM, TN>(acc, outp, p, partial, mb * BM + wm0, n0 + wn0, lrow, lcol);
    return;
  }
  const int ctotal = partial ? p.Cout : p.out_ctotal;
  const int coff = partial ? 0 : p.out_coff;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long op = n0 + wn0 + j * 32 + lcol;
    if (op >= p.npix) continue;
    const int hw = p.Ho * p.Wo;
    const int n = (int)(op / hw);
    const int r0 = (int)(op - (long)n * hw);
    const long obase = ((long)n * ctotal + coff) * hw + r0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = mb * BM + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
        if (co < p.Cout) {
          float v = acc[i][j][r];
          if (!partial) {
            v = v + (p.bias ? p.bias[co] : 0.f);
            v = v > 0.f ? v : v * p.slope;
          }
          outp[obase + (long)co * hw] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Convs with a handful of output channels (the flow / mask heads: Cout = 2 or 1 over 770-1026 input channels,
// deepIM_flownet.py:123,145,176,317): on the MFMA kernels 62 of 64 tile rows would be padding, and the layer is a pure
// stream over the input (118 MB at B = 32 for 0.5 GFLOP). VALU kernel: a block owns 64 consecutive output pixels
// (lane = pixel, loads coalesced along the row), its 16 waves split the input channels into sixteenths, each wave runs
// the (ci,ky,kx)-ordered fmaf chain of its share with wave-uniform weights from the scalar cache; the sixteen partial
// sums are added in wave order through LDS (deterministic), then bias + LeakyReLU.
// KS > 0: square kernel size known at compile time — the taps of a channel are unrolled so its KS*KS loads are in
// flight together (the loop is latency-bound otherwise: 1.9 ms instead of 0.1 ms for the 770-channel heads)
template <int COUT, int KS>
__global__ __launch_bounds__(1024) void conv_fewout_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                          const float* __restrict__ wp, const float* __restrict__ bias,
                                                          int Cin, int H, int W, int Ho, int Wo, int kh, int kw,
                                                          int stride, int pad, long npix, int out_ctotal, int out_coff,
                                                          float slope, float* __restrict__ partial, int cslice) {
  // few pixels (B = 4 heads: 75 … 5 blocks of 64 pixels for 256 CUs, each wave walking 48-64 channels one after the other —
  // 60 µs for 0.13 GFLOP): gridDim.y slices of `cslice` input channels each, raw sums to partial[slice][n][co][hw], bias +
  // activation in splitk_reduce_kernel
  constexpr int NW = 16;   // waves per block = channel shares
  __shared__ float part[NW - 1][COUT][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long pix = (long)blockIdx.x * 64 + lane;
  const bool live = pix < npix;
  const int hw = Ho * Wo;
  const long n = live ? pix / hw : 0;
  const int r = live ? (int)(pix - n * hw) : 0;
  const int ho = r / Wo, wo = r - ho * Wo;
  const int hi0 = ho * stride - pad, wi0 = wo * stride - pad;
  const int s_lo = blockIdx.y * cslice, s_hi = min(Cin, s_lo + cslice);
  const int cq = (s_hi - s_lo + NW - 1) / NW;
  const int c_lo = min(s_hi, s_lo + wave * cq), c_hi = min(s_hi, c_lo + cq);
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  const float* ip = in + (n * Cin) * (long)H * W;
  const int khw = kh * kw;
  if constexpr (KS > 0) {
    // tap addresses and validity are the same for every channel: hoisted
    int off[KS * KS];
    bool ok[KS * KS];
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int hi = hi0 + ky, wi = wi0 + kx;
        ok[ky * KS + kx] = live && hi >= 0 && hi < H && wi >= 0 && wi < W;
        off[ky * KS + kx] = min(max(hi, 0), H - 1) * W + min(max(wi, 0), W - 1);
      }
    for (int ci = c_lo; ci < c_hi; ++ci) {
      const float* plane = ip + (long)ci * H * W;
      float v[KS * KS];
#pragma unroll
      for (int t = 0; t < KS * KS; ++t) v[t] = plane[off[t]];
#pragma unroll
      for (int t = 0; t < KS * KS; ++t) {
        const float x = ok[t] ? v[t] : 0.f;
#pragma unroll
        for (int co = 0; co < COUT; ++co)   // LDS-kernel packing, granule 0: wp[k*64 + co] = w[co][k], k = (ci,ky,kx)
          acc[co] = fmaf(wp[((long)ci * (KS * KS) + t) * GRAN + co], x, acc[co]);
      }
    }
  } else {
    for (int ci = c_lo; ci < c_hi; ++ci) {
      const float* plane = ip + (long)ci * H * W;
      for (int ky = 0; ky < kh; ++ky) {
        const int hi = hi0 + ky;
        const bool rowok = live && hi >= 0 && hi < H;
        const float* row = plane + (long)min(max(hi, 0), H - 1) * W;
        for (int kx = 0; kx < kw; ++kx) {
          const int wi = wi0 + kx;
          const float v = (rowok && wi >= 0 && wi < W) ? row[min(max(wi, 0), W - 1)] : 0.f;
#pragma unroll
          for (int co = 0; co < COUT; ++co)
            acc[co] = fmaf(wp[((long)ci * khw + ky * kw + kx) * GRAN + co], v, acc[co]);
        }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) part[wave - 1][co][lane] = acc[co];
  }
  __syncthreads();
  if (wave == 0 && live) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float v = acc[co];
#pragma unroll
      for (int q = 0; q < NW - 1; ++q) v += part[q][co][lane];
      if (partial) {
        partial[((long)blockIdx.y * (npix / hw) * COUT + n * COUT + co) * hw + r] = v;
        continue;
      }
      v = v + (bias ? bias[co] : 0.f);
      v = v > 0.f ? v : v * slope;
      out[(n * out_ctotal + out_coff + co) * hw + r] = v;
    }
  }
}

// The 3x3 stride-1 pad-1 heads with W % 4 == 0 (deepIM_flownet.py:123,145,317: the flow / mask predictors at 15x20 and 30x40): a lane
// owns FOUR consecutive output pixels of a row. Per input channel it loads three rows of six columns — one aligned dwordx4 + the two
// neighbours, out-of-image positions through out-of-range buffer offsets (the hardware returns the zero padding: no selects) — and
// runs the 4 x 9 x COUT fmaf: 2.25 loads per output pixel and channel instead of 9, and no masking VALU. The conv_fewout_kernel above
// ran at 0.10-0.20 of the HBM rate (profiles/per_kernel.json); this one at 0.19-0.24 (the 770-channel heads 75-79 -> 61-64 us). Eight waves split the block's
// channel slice into eighths ((ci,ky,kx)-ordered chains, partial sums added in wave order through LDS), grid.y slices the channels
// further where the pixels alone leave the chip empty (raw sums to partial[slice][n][co][hw], fixed-order second pass).
#ifndef FEWOUT_UNROLL
#define FEWOUT_UNROLL 2     // channels of the stream whose loads are in flight together (1: one at a time; same chains, same results)
#endif
#ifndef FEWOUT_HALO_DPP
#define FEWOUT_HALO_DPP 1   // 0: the round-5 form (two dword loads per row for the halo columns), for A/B builds
#endif
template <int COUT>
__global__ __launch_bounds__(512) void conv_fewout_quad_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                              const float* __restrict__ wp, const float* __restrict__ bias, int Cin,
                                                              int H, int W, long nquad, unsigned in_bytes, int out_ctotal, int out_coff,
                                                              float slope, float* __restrict__ partial, int cslice) {
  constexpr int NW = 8;
  __shared__ float4 part[NW - 1][COUT][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long q = (long)blockIdx.x * 64 + lane;
  const bool live = q < nquad;
  const int Wq = W >> 2, hwq = H * Wq, hw = H * W;
  const int n = live ? (int)(q / hwq) : 0;
  const int r = live ? (int)(q - (long)n * hwq) : 0;
  const int ho = r / Wq, x0 = (r - ho * Wq) * 4;
  // byte offsets of (row ho - 1 + ky, columns x0 - 1 | x0 .. x0 + 3 | x0 + 4) in channel 0 of sample n; bit 31 = outside the image
  int voff[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int hi = ho - 1 + ky;
    const bool rowok = live && hi >= 0 && hi < H;
    const int base = ((n * Cin) * H + hi) * W + x0;
    voff[ky][0] = rowok && x0 > 0 ? (base - 1) * 4 : (int)0x80000000;
    voff[ky][1] = rowok ? base * 4 : (int)0x80000000;
    voff[ky][2] = rowok && x0 + 4 < W ? (base + 4) * 4 : (int)0x80000000;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)in_bytes, 0x00020000);
  const int s_lo = blockIdx.y * cslice, s_hi = min(Cin, s_lo + cslice);
  const int cq = (s_hi - s_lo + NW - 1) / NW;
  const int c_lo = min(s_hi, s_lo + wave * cq), c_hi = min(s_hi, c_lo + cq);
  float acc[COUT][4];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[co][o] = 0.f;
  // (Tried, same box: four channels' loads in flight together — 62.5 / 61.1 us against 63.6 / 61.4 on the 770-channel heads; the weights
  // staged in per-wave LDS strips instead of scalar loads from the packed layout's 256-byte lines — 75 us. Neither the load latency of a
  // wave nor the scalar cache is what holds this stream at ~2 TB/s; kept simple.)
  // The two halo columns of a lane's quad are its neighbours' own pixels: lane - 1 holds x0 - 1 as its m.w, lane + 1 holds x0 + 4 as its
  // m.x (quads are dealt in row order; at a row's ends the halo is the zero padding). They arrive by a whole-wave DPP shift instead of
  // two more loads per row — the six dword loads per channel were 2/3 of the kernel's address-coalescer time (64 lanes x 4 bytes at a
  // 16-byte stride cost a full-rate instruction each) —; only lanes 0 and 63, whose neighbour sits in another wave, still load theirs.
  const bool edge = lane == 0 || lane == 63;
  const bool dl = lane > 0 && x0 > 0, dr = lane < 63 && x0 + 4 < W;
  // one channel's loads: three aligned dwordx4 (+ the halo pixels of lanes 0 / 63, whose neighbours sit in another wave)
  auto load_ch = [&](const int ci, float4 (&m)[3], float (&e0)[3], float (&e5)[3]) {
    const int so = ci * hw * 4;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      m[ky] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[ky][1], so, 0));
      e0[ky] = 0.f; e5[ky] = 0.f;
    }
    if (!FEWOUT_HALO_DPP || edge) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        if (!FEWOUT_HALO_DPP || lane == 0) e0[ky] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[ky][0], so, 0));
        if (!FEWOUT_HALO_DPP || lane == 63) e5[ky] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[ky][2], so, 0));
      }
    }
  };
  // ... and its 36 x COUT fmaf, in (ky, kx) order behind the channels before it: the chain of every output is the one-channel-at-a-time
  // chain whatever FEWOUT_UNROLL is
  auto fma_ch = [&](const int ci, const float4 (&m)[3], const float (&e0)[3], const float (&e5)[3]) {
    float x[3][6];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      x[ky][0] = e0[ky]; x[ky][5] = e5[ky];
      if (FEWOUT_HALO_DPP) {
        // wave_shr:1 (0x138): lane i reads lane i - 1; wave_shl:1 (0x130): lane i reads lane i + 1 (GFX9 DPP controls)
        const float fl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m[ky].w), 0x138, 0xf, 0xf, false));
        const float fr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m[ky].x), 0x130, 0xf, 0xf, false));
        x[ky][0] = dl ? fl : x[ky][0];
        x[ky][5] = dr ? fr : x[ky][5];
      }
      x[ky][1] = m[ky].x; x[ky][2] = m[ky].y; x[ky][3] = m[ky].z; x[ky][4] = m[ky].w;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int co = 0; co < COUT; ++co) {   // LDS-kernel packing, granule 0: wp[k*64 + co] = w[co][k], k = (ci,ky,kx)
          const float wv = wp[((long)ci * 9 + ky * 3 + kx) * GRAN + co];
#pragma unroll
          for (int o = 0; o < 4; ++o) acc[co][o] = fmaf(wv, x[ky][o + kx], acc[co][o]);
        }
  };
  {
    int ci = c_lo;
    if (FEWOUT_UNROLL >= 2) {
      for (; ci + 1 < c_hi; ci += 2) {   // two channels' loads in flight together
        float4 ma[3], mb[3];
        float a0[3], a5[3], b0[3], b5[3];
        load_ch(ci, ma, a0, a5);
        load_ch(ci + 1, mb, b0, b5);
        fma_ch(ci, ma, a0, a5);
        fma_ch(ci + 1, mb, b0, b5);
      }
    }
    for (; ci < c_hi; ++ci) {
      float4 ma[3];
      float a0[3], a5[3];
      load_ch(ci, ma, a0, a5);
      fma_ch(ci, ma, a0, a5);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) part[wave - 1][co][lane] = make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
  }
  __syncthreads();
  if (wave == 0 && live) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float4 v = make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
#pragma unroll
      for (int w_ = 0; w_ < NW - 1; ++w_) {
        const float4 t = part[w_][co][lane];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      const long pix0 = (long)ho * W + x0;
      if (partial) {
        *reinterpret_cast<float4*>(partial + (((long)blockIdx.y * (nquad / hwq) + n) * COUT + co) * hw + pix0) = v;
        continue;
      }
      const float bv = bias ? bias[co] : 0.f;
      v.x += bv; v.y += bv; v.z += bv; v.w += bv;
      v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
      v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
      *reinterpret_cast<float4*>(out + ((long)n * out_ctotal + out_coff + co) * hw + pix0) = v;
    }
  }
}

// split-K second pass: out[n][coff+c][hw] = lrelu(Σ_s partial[s][n][c][hw] + bias[c]) in fixed order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(float* __restrict__ out, const float* __restrict__ partial,
                                                            const float* __restrict__ bias, long total, long stride,
                                                            int S, int Cout, int hw, int ctotal, int coff,
                                                            float slope, Remap rm, const float* __restrict__ ep_y,
                                                            const float* __restrict__ ep_add, float ep_slope) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int r = (int)(i % hw);
  long poff, plane;
  if (!remap_pixel(rm, r, hw, poff, plane)) return;
  float v = partial[i];
  for (int s = 1; s < S; ++s) v += partial[(long)s * stride + i];
  const int c = (int)((i / hw) % Cout);
  const long n = i / ((long)hw * Cout);
  v = v + (bias ? bias[c] : 0.f);
  v = v > 0.f ? v : v * slope;
  const long o = (n * ctotal + coff + c) * plane + poff;
  if (ep_y) {
    if (ep_add) v += ep_add[o];
    v = ep_y[o] > 0.f ? v : v * ep_slope;
  }
  out[o] = v;
}

// the second passes of a ConvGroup launch in one: member m sums its own slices onto its own remap window of `out`
struct ReduceGroup {
  float* out;
  const float* partial[CONV_GROUP_MAX];
  long total[CONV_GROUP_MAX], stride[CONV_GROUP_MAX];
  int S[CONV_GROUP_MAX], hw[CONV_GROUP_MAX];
  Remap rm[CONV_GROUP_MAX];
  int start[CONV_GROUP_MAX + 1];
  int n, Cout;
  const float* ep_y;      // activation-gradient epilogue (ConvParams::ep_*)
  const float* ep_add;
  float ep_slope;
  const float* bias;      // forward use (transposed convolution of the decoder): + bias, LeakyReLU, channel slice of a Concat
  float slope;
  int out_ctotal, out_coff;
};
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(ReduceGroup g) {
  const int b = blockIdx.x;
  int m = 0;
#pragma unroll
  for (int i = 1; i < CONV_GROUP_MAX; ++i)
    if (i < g.n && b >= g.start[i]) m = i;
  m = __builtin_amdgcn_readfirstlane(m);
  const long i = (long)(b - g.start[m]) * 256 + threadIdx.x;
  if (i >= g.total[m]) return;
  const int hw = g.hw[m];
  const int r = (int)(i % hw);
  long poff, plane;
  if (!remap_pixel(g.rm[m], r, hw, poff, plane)) return;
  const float* partial = g.partial[m];
  const long stride = g.stride[m];
  float v = partial[i];
  for (int s = 1; s < g.S[m]; ++s) v += partial[(long)s * stride + i];
  const int c = (int)((i / hw) % g.Cout);
  const long n = i / ((long)hw * g.Cout);
  const long o = (n * g.out_ctotal + g.out_coff + c) * plane + poff;
  v += g.bias ? g.bias[c] : 0.f;
  v = v > 0.f ? v : v * g.slope;
  if (g.ep_y) {
    if (g.ep_add) v += g.ep_add[o];
    v = g.ep_y[o] > 0.f ? v : v * g.ep_slope;
  }
  g.out[o] = v;
}

// tail-split second pass: one block per remainder tile; out = lrelu(Σ_slice partial + bias), slices in fixed order
__global__ __launch_bounds__(256) void tail_reduce_kernel(float* __restrict__ out, const float* __restrict__ partial,
                                                          const float* __restrict__ bias, int n_full, int R, int S,
                                                          int gx, int Cout, long npix, int hw, int ctotal, int coff,
                                                          float slope, Remap rm) {
  const int rt = blockIdx.x, vid = n_full + rt;
  const int bx = vid % gx, mb = vid / gx;
  const float* pp = partial + (long)rt * 16384;
  for (int e = threadIdx.x; e < 16384; e += 256) {
    const int co = mb * 128 + (e >> 7);
    const long pix = (long)bx * 128 + (e & 127);
    if (co >= Cout || pix >= npix) continue;
    float v = pp[e];
    for (int s = 1; s < S; ++s) v += pp[(long)s * R * 16384 + e];
    v = v + (bias ? bias[co] : 0.f);
    v = v > 0.f ? v : v * slope;
    const long n = pix / hw;
    const int r = (int)(pix - n * hw);
    long poff, plane;
    if (!remap_pixel(rm, r, hw, poff, plane)) continue;
    out[(n * ctotal + coff + co) * plane + poff] = v;
  }
}

// split-K second pass for channel-blocked outputs: partials and output share the [n][C/8][hw][8] layout, so the pass is
// elementwise on float4 (4 consecutive channels)
__global__ __launch_bounds__(256) void splitk_reduce_nc8_kernel(float* __restrict__ out, const float* __restrict__ partial,
                                                                const float* __restrict__ bias, long total4,
                                                                long stride, int S, int C8, int hw, float slope, int C8mod) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  float4 v = reinterpret_cast<const float4*>(partial)[i];
  for (int s = 1; s < S; ++s) {
    const float4 w = reinterpret_cast<const float4*>(partial + (long)s * stride)[i];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
  }
  // 2 float4 per pixel record; space-to-depth output: C8 = 4 phases x C8mod blocks of hw = Ho*Wo/4 pixels
  const int c = (int)(((i / (2L * hw)) % C8) % C8mod) * 8 + (int)(i & 1) * 4;
  const float4 b = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0, 0, 0, 0);
  v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
  v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
  reinterpret_cast<float4*>(out)[i] = v;
}

// NC8 <-> NCHW re-layout of an activation tensor (decoder skip connections, tests)
// (ctotal / coff: the NCHW side is channels [coff, coff + C) of a ctotal-channel tensor — a decoder concat buffer)
__global__ __launch_bounds__(256) void relayout_nc8_kernel(float* __restrict__ dst, const float* __restrict__ src, int C,
                                                           int hw, long total, int to_nc8, int ctotal = 0, int coff = 0) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // index in the NC8 tensor
  if (i >= total) return;
  const int q = (int)(i & 7);
  const long pix = (i >> 3) % hw;
  const long nc8 = (i >> 3) / hw;
  const int C8 = C >> 3;
  const long n = nc8 / C8;
  const int c = (int)(nc8 % C8) * 8 + q;
  const long j = ctotal ? (n * ctotal + coff + c) * hw + pix : (n * C + c) * hw + pix;                 // index in the NCHW tensor
  if (to_nc8) dst[i] = src[j]; else dst[j] = src[i];
}

// ------------------------------------------------------------ packing ----
// What the pack kernels read. on = 0: w is the convolution's own (Cout, Cin, kh·kw) tensor. on = 1: the weights of a DATA
// GRADIENT read straight out of the layer's raw tensor wl (Co_l, Ci_l, kh_l, kw_l) — the convolution being packed has
// Cout = Ci_l, Cin = Co_l and an (nky x nkx) kernel whose tap (a, b) is the layer's tap (ky0 + st·(nky-1-a), kx0 + st·(nkx-1-b)):
// st = 1, ky0 = kx0 = 0 is the transposed + flipped kernel of a stride-1 layer, st = 2 the sub-kernel of one output parity
// class of a stride-2 layer (csrc/backward.hip). Fusing the view into the pack saves the flip pass and its buffer.
struct WView { int on, Ci_l, kh_l, kw_l, ky0, kx0, st, nky, nkx; };
__device__ __forceinline__ float wview_at(const WView& v, const float* __restrict__ w, int co, int ci, int t, int Cin, int khw) {
  if (!v.on) return w[((long)co * Cin + ci) * khw + t];
  const int a = t / v.nkx, b = t - a * v.nkx;
  return w[(((long)ci * v.Ci_l + co) * v.kh_l + v.ky0 + v.st * (v.nky - 1 - a)) * v.kw_l + v.kx0 + v.st * (v.nkx - 1 - b)];
}
// conv:   packed[g][kc][kk][mm] = w[(g*64+mm)][kc*16+kk]  (w as (Cout, K) row-major), zero padded
__global__ void pack_conv_kernel(float* __restrict__ packed, const float* __restrict__ w, int Cout, int K, int nchunk,
                                 long total, WView v, int khw) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int mm = (int)(i % GRAN);
  const int kk = (int)((i / GRAN) % KT);
  const int kc = (int)((i / (GRAN * KT)) % nchunk);
  const int g = (int)(i / ((long)GRAN * KT * nchunk));
  const int co = g * GRAN + mm, k = kc * KT + kk;
  packed[i] = (co < Cout && k < K) ? wview_at(v, w, co, k / khw, k % khw, K / khw, khw) : 0.f;
}
// deconv (w: Cin,Cout,4,4): 4 parity classes z = py*2+px; per class K = Cin*4, k = ci*4 + jy*2 + jx,
// tap ky = py + 2*jy, kx = px + 2*jx:  packed[z][g][kc][kk][mm] = w[ci][g*64+mm][ky][kx]
__global__ void pack_deconv_kernel(float* __restrict__ packed, const float* __restrict__ w, int Cin, int Cout,
                                   int ngran, int nchunk, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int mm = (int)(i % GRAN);
  const int kk = (int)((i / GRAN) % KT);
  const int kc = (int)((i / (GRAN * KT)) % nchunk);
  const int g = (int)((i / ((long)GRAN * KT * nchunk)) % ngran);
  const int z = (int)(i / ((long)GRAN * KT * nchunk * ngran));
  const int py = z >> 1, px = z & 1;
  const int co = g * GRAN + mm, k = kc * KT + kk;
  float v = 0.f;
  if (co < Cout && k < Cin * 4) {
    const int ci = k >> 2, jy = (k >> 1) & 1, jx = k & 1;
    v = w[(((long)ci * Cout + co) * 4 + (py + 2 * jy)) * 4 + (px + 2 * jx)];
  }
  packed[i] = v;
}

// direct layout: packed[mt][g/4][lane = h*32 + r][g%4] = w[mt*32 + r][2*ci2 + h][ky][kx], g = (ci2*kh + ky)*kw + kx, zero padded
__global__ void pack_direct_kernel(float* __restrict__ packed, const float* __restrict__ w, int Cout, int Cin, int khw,
                                   int npair, long total, WView v) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i & 3), r = (int)((i >> 2) & 31), h = (int)((i >> 7) & 1);
  const int g = (int)((i >> 8) % (npair / 4)) * 4 + q;
  const int mt = (int)(i / (64L * npair));
  const int co = mt * 32 + r, ci = 2 * (g / khw) + h, t = g % khw;
  packed[i] = (co < Cout && ci < Cin) ? wview_at(v, w, co, ci, t, Cin, khw) : 0.f;
}
// the register-fed order of the four parity-class sub-kernels of a stride-2 data gradient in one launch
struct PackGroup {
  float* dst[CONV_GROUP_MAX];
  long total[CONV_GROUP_MAX];
  int khw[CONV_GROUP_MAX], npair[CONV_GROUP_MAX];
  WView v[CONV_GROUP_MAX];
  int start[CONV_GROUP_MAX + 1];
  int n, Cout, Cin;
};
__global__ __launch_bounds__(256) void pack_direct_group_kernel(const float* __restrict__ w, PackGroup g) {
  const int b = blockIdx.x;
  int m = 0;
#pragma unroll
  for (int i = 1; i < CONV_GROUP_MAX; ++i)
    if (i < g.n && b >= g.start[i]) m = i;
  m = __builtin_amdgcn_readfirstlane(m);
  const long i = (long)(b - g.start[m]) * 256 + threadIdx.x;
  if (i >= g.total[m]) return;
  const int khw = g.khw[m], npair = g.npair[m];
  const int q = (int)(i & 3), r = (int)((i >> 2) & 31), h = (int)((i >> 7) & 1);
  const int gq = (int)((i >> 8) % (npair / 4)) * 4 + q;
  const int mt = (int)(i / (64L * npair));
  const int co = mt * 32 + r, ci = 2 * (gq / khw) + h, t = gq % khw;
  g.dst[m][i] = (co < g.Cout && ci < g.Cin) ? wview_at(g.v[m], w, co, ci, t, g.Cin, khw) : 0.f;
}
__global__ void build_direct_tab_kernel(int2* __restrict__ tab, int npair_real, int n, int kh, int kw, int H, int W) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  if (g >= npair_real) { tab[g] = make_int2(0, 63); return; }
  const int kx = g % kw, ky = (g / kw) % kh, ci2 = g / (kw * kh);
  tab[g] = make_int2((2 * ci2 * H * W + ky * W + kx) * 4, ky * 8 + kx);
}

// NC8 layout: packed[mt][g][lane = h*32 + r][s] = w[mt*32 + r][c8*8 + s + 4h][ky][kx], g = (c8*kh + ky)*kw + kx
__global__ void pack_nc8_kernel(float* __restrict__ packed, const float* __restrict__ w, int Cout, int Cin, int khw,
                                int ngroup, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int s = (int)(i & 3), r = (int)((i >> 2) & 31), h = (int)((i >> 7) & 1);
  const int g = (int)((i >> 8) % ngroup);
  const int mt = (int)(i / (256L * ngroup));
  const int co = mt * 32 + r, ci = (g / khw) * 8 + s + 4 * h, t = g % khw;
  packed[i] = (co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * khw + t] : 0.f;
}
__global__ void build_nc8_tab_kernel(int2* __restrict__ tab, int ngroup_real, int n, int kh, int kw, int H, int W) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  if (g >= ngroup_real) { tab[g] = make_int2(0, 63); return; }
  const int kx = g % kw, ky = (g / kw) % kh, c8 = g / (kw * kh);
  tab[g] = make_int2((c8 * H * W + ky * W + kx) * 32, ky * 8 + kx);
}

__global__ void build_conv_tab_kernel(int2* __restrict__ tab, int K, int Kpad, int kh, int kw, int H, int W) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Kpad) return;
  if (k >= K) { tab[k] = make_int2(0, 63); return; }
  const int kx = k % kw, ky = (k / kw) % kh, ci = k / (kw * kh);
  tab[k] = make_int2((ci * H * W + ky * W + kx) * 4, ky * 8 + kx);
}
__global__ void build_deconv_tab_kernel(int2* __restrict__ tab, int K, int Kpad, int H, int W) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Kpad) return;
  if (k >= K) { tab[k] = make_int2(0, 63); return; }
  const int ci = k >> 2, jy = (k >> 1) & 1, jx = k & 1;
  tab[k] = make_int2((ci * H * W + (1 - jy) * W + (1 - jx)) * 4, jy * 8 + jx);  // origin = (iy0-1, ix0-1)
}

// grouped k32 s16 transposed conv (depthwise), cropped, scaled: ≤ 2x2 contributing inputs per output
__global__ __launch_bounds__(256) void upsample16_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                         const float* __restrict__ w, int C, int H, int W, int Ho,
                                                         int Wo, int crop_y, int crop_x, float scale) {
  const int xo = blockIdx.x * 256 + threadIdx.x;
  const int yo = blockIdx.y;
  const int bc = blockIdx.z;
  if (xo >= Wo) return;
  const int c = bc % C;
  const int y = yo + crop_y, x = xo + crop_x;  // uncropped output coords: y = iy*16 + ky
  const float* ip = in + (long)bc * H * W;
  const float* wp = w + (long)c * 1024;
  float acc = 0.f;
  // iy ascending, ix ascending (matches a gather restatement of the scatter definition)
  const int iy_hi = y >> 4, ix_hi = x >> 4;
#pragma unroll
  for (int dy = 1; dy >= 0; --dy) {
    const int iy = iy_hi - dy;
    const int ky = y - iy * 16;
    if (iy < 0 || iy >= H || ky >= 32) continue;
#pragma unroll
    for (int dx = 1; dx >= 0; --dx) {
      const int ix = ix_hi - dx;
      const int kx = x - ix * 16;
      if (ix < 0 || ix >= W || kx >= 32) continue;
      acc = fmaf(ip[iy * W + ix], wp[ky * 32 + kx], acc);
    }
  }
  out[((long)bc * Ho + yo) * Wo + xo] = acc * scale;
}

// the same, four consecutive outputs of a row per thread (Wo % 4 == 0, crop_x % 4 == 0: they share the <= 2x2 contributing inputs, their
// weights are one aligned float4 per input, the store is 16 bytes; every output sums in the scalar kernel's order: bit-identical).
// The one-output kernel moved 0.15-0.16 of the HBM rate at B = 32 (profiles/per_kernel.json): instruction-bound, not byte-bound.
__global__ __launch_bounds__(256) void upsample16x4_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                           const float* __restrict__ w, int C, int H, int W, int Ho, int Wo,
                                                           int crop_y, int crop_x, float scale, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;      // (bc, yo, xo / 4)
  if (i >= total) return;
  const int wq = Wo >> 2;
  const int xq = (int)(i % wq);
  const long r = i / wq;
  const int yo = (int)(r % Ho);
  const long bc = r / Ho;
  const int c = (int)(bc % C);
  const int y = yo + crop_y, x0 = 4 * xq + crop_x;
  const float* ip = in + bc * H * W;
  const float* wp = w + (long)c * 1024;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int iy_hi = y >> 4, ix_hi = x0 >> 4;
#pragma unroll
  for (int dy = 1; dy >= 0; --dy) {
    const int iy = iy_hi - dy;
    const int ky = y - iy * 16;
    if (iy < 0 || iy >= H || ky >= 32) continue;
#pragma unroll
    for (int dx = 1; dx >= 0; --dx) {
      const int ix = ix_hi - dx;
      const int kx = x0 - ix * 16;
      if (ix < 0 || ix >= W || kx + 3 >= 32) continue;
      const float v = ip[iy * W + ix];
      const float4 wv = *reinterpret_cast<const float4*>(wp + ky * 32 + kx);
      acc.x = fmaf(v, wv.x, acc.x); acc.y = fmaf(v, wv.y, acc.y); acc.z = fmaf(v, wv.z, acc.z); acc.w = fmaf(v, wv.w, acc.w);
    }
  }
  *reinterpret_cast<float4*>(out + (bc * Ho + yo) * Wo + 4 * xq) = make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale);
}

struct TileChoice { int bm, bn, ksplit, tail_s; };
// Tile/split plan, deterministic (the same geometry gets the same summation order in every run, process and rank):
// 128x128 tiles (best MFMA density per gathered activation); 64x256 / 64x128 when Cout <= 64; when the grid leaves the
// chip under-filled, split K across grid.z (fixed-order two-pass reduction). The split factor minimises a cost model
// fitted to what the timing autotuner picked on MI355X at B = 1 … 32 (profiles/r02_conv_plans.md):
//   cost(s) = ceil(tiles·s / 256) · ceil(nchunk / s)        MFMA time: blocks sharing a CU share its matrix pipes, so what
//                                                            counts is the most loaded CU, in units of one K chunk (≈1 µs)
//           + [s > 1] · (3 + 0.016 · tiles · s)             split-K reduce: launch + 64 KB of partial sums per block
// over s ∈ {1,…,10,12,14,16,20,24}, s ≤ nchunk/4, tiles·s ≤ 4096; ties go to the smaller s.
int plan_ksplit(long tiles, int nchunk) {
  const int cands[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24};
  float best = 1e30f;
  int best_s = 1;
  for (int s : cands) {
    if (s > 1 && (tiles * s > 4096 || s > max(1, nchunk / 4))) continue;
    const float cost = (float)di_div_up(tiles * s, 256) * (float)di_div_up(nchunk, s) +
                       (s > 1 ? 3.f + 0.016f * (float)(tiles * s) : 0.f);
    if (cost < best * 0.985f) { best = cost; best_s = s; }
  }
  return best_s;
}
TileChoice choose_tile(const deepim_ctx* ctx, int Cout, long npix, int nchunk, int classes, bool wide64 = false) {
  int bm = 128, bn = 128;
  if (Cout <= 64) { bm = 64; bn = (wide64 || npix >= 256L * 1024) ? 256 : 128; }   // wide64: the only 64-row LDS-free shape
  else if (ctx->conv_tile256 && Cout % 256 == 0) bm = 256;   // 8-wave block: every gathered activation feeds 256 channels
  const long tiles = (long)di_div_up(Cout, bm) * di_div_up(npix, bn) * classes;
  return {bm, bn, plan_ksplit(tiles, nchunk), 0};
}

template <int MODE>
int launch_one(deepim_ctx* ctx, ConvParams p, int classes, TileChoice t) {
  if (p.out_nc8 == 2) t.ksplit = 1, t.tail_s = 0;   // the split16 epilogue has no split-K second pass
  if (p.ep_y) {   // only the register-fed kernels' epilogue and the plain split-K second pass apply it
    t.tail_s = 0;
    DI_REQUIRE(MODE == MODE_CONV && p.tab2 != nullptr && !p.in_nc8 && !p.out_nc8 && ((t.bm == 128 && t.bn == 128) || (t.bm == 64 && t.bn == 256)),
               "conv: activation-gradient epilogue asked of a kernel family that has none");
  }
  p.ksplit = t.ksplit;
  p.chunks_per_split = di_div_up(p.nchunk, t.ksplit);
  if (p.in_nc8) p.chunks_per_split = (p.chunks_per_split + 1) & ~1;   // the NC8 kernel consumes whole pairs of chunks
  p.ksplit = di_div_up(p.nchunk, p.chunks_per_split);
  p.partial = nullptr;
  p.partial_stride = (long)p.B * p.Cout * p.Ho * p.Wo;
  if (p.ksplit > 1) {
    void* scratch;
    int rc = deepim_scratch(ctx, (size_t)p.ksplit * p.partial_stride * sizeof(float), &scratch);
    if (rc) return rc;
    p.partial = (float*)scratch;
  }
  p.swizzle = ctx->conv_xcd_swizzle;
  p.gx = di_div_up(p.npix, t.bn); p.gy = di_div_up(p.Cout, t.bm); p.gz = classes * p.ksplit;
  p.n_full = 0; p.tail_s = 0; p.tail_cps = 0; p.n_tail_pad = 0; p.tail_partial = nullptr;
  const bool direct_ok = MODE == MODE_CONV && t.bm == 128 && t.bn == 128 && p.tab2 != nullptr;
  if (direct_ok && t.tail_s > 1 && t.ksplit == 1) {
    const int tiles = p.gx * p.gy, slots = ctx->conv_tail_slots;
    const int n_full = tiles / slots * slots, R = tiles - n_full;
    if (n_full > 0 && R > 0) {
      p.n_full = n_full;
      p.tail_cps = di_div_up(p.nchunk, t.tail_s);
      p.tail_s = di_div_up(p.nchunk, p.tail_cps);
      p.n_tail_pad = di_div_up(R * p.tail_s, 8) * 8;
      void* scratch;
      int rc = deepim_scratch(ctx, (size_t)R * p.tail_s * 16384 * sizeof(float), &scratch);
      if (rc) return rc;
      p.tail_partial = (float*)scratch;
      hipLaunchKernelGGL(conv_direct_kernel<2>, dim3(p.n_full + p.n_tail_pad), dim3(256), 0, ctx->stream, p);
      hipLaunchKernelGGL(tail_reduce_kernel, dim3(R), dim3(256), 0, ctx->stream, p.out, p.tail_partial, p.bias, n_full, R,
                         p.tail_s, p.gx, p.Cout, p.npix, p.Ho * p.Wo, p.out_ctotal, p.out_coff, p.slope, remap_of(p));
      DI_LAUNCH_CHECK();
      return 0;
    }
  }
  DI_REQUIRE((long)p.gx * p.gy * p.gz < (1L << 31) && p.gx > 0, "conv: grid too large");
  dim3 grid(p.gx * p.gy * p.gz);
  if (MODE == MODE_CONV && p.in_nc8) {
    if (t.bm == 64) {
      DI_REQUIRE(t.bn == 256 && p.out_nc8 == 1, "conv: the 64-row NC8 kernel writes NC8 output on 64x256 tiles");
      hipLaunchKernelGGL((conv_nc8_kernel<1, 1>), grid, dim3(256), 0, ctx->stream, p);
    } else if (p.out_nc8) hipLaunchKernelGGL(conv_nc8_kernel<1>, grid, dim3(256), 0, ctx->stream, p);
    else hipLaunchKernelGGL(conv_nc8_kernel<0>, grid, dim3(256), 0, ctx->stream, p);
  } else if (direct_ok)
    hipLaunchKernelGGL(conv_direct_kernel<2>, grid, dim3(256), 0, ctx->stream, p);
  else if (MODE == MODE_CONV && t.bm == 64 && t.bn == 256 && p.tab2 != nullptr)
    hipLaunchKernelGGL(conv_direct_kernel<1>, grid, dim3(256), 0, ctx->stream, p);
  else if (t.bm == 256)
    hipLaunchKernelGGL((conv_mfma_kernel<256, 128, MODE, 512>), grid, dim3(512), 0, ctx->stream, p);
  else if (t.bm == 128)
    hipLaunchKernelGGL((conv_mfma_kernel<128, 128, MODE>), grid, dim3(256), 0, ctx->stream, p);
  else if (t.bn == 256)
    hipLaunchKernelGGL((conv_mfma_kernel<64, 256, MODE>), grid, dim3(256), 0, ctx->stream, p);
  else
    hipLaunchKernelGGL((conv_mfma_kernel<64, 128, MODE>), grid, dim3(256), 0, ctx->stream, p);
  if (p.ksplit > 1) {
    const long total = p.partial_stride;
    if (p.out_nc8)
      hipLaunchKernelGGL(splitk_reduce_nc8_kernel, dim3(di_div_up(total / 4, 256)), dim3(256), 0, ctx->stream, p.out,
                         p.partial, p.bias, total / 4, p.partial_stride, p.ksplit, (p.Cout >> 3) * (p.out_s2d ? 4 : 1),
                         p.Ho * p.Wo / (p.out_s2d ? 4 : 1), p.slope, p.Cout >> 3);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, p.out, p.partial,
                         p.bias, total, p.partial_stride, p.ksplit, p.Cout, p.Ho * p.Wo, p.out_ctotal, p.out_coff,
                         p.slope, remap_of(p), p.ep_y, p.ep_add, p.ep_slope);
  }
  DI_LAUNCH_CHECK();
  return 0;
}

// Opt-in ("conv_autotune" = 1; default 0 so that plans never depend on wall-clock noise): on the first call of a
// geometry (outside graph capture) time a few split-K factors with HIP events and remember the fastest. The kernel is
// idempotent, so the trial launches only rewrite `out`.
template <int MODE>
int launch_conv(deepim_ctx* ctx, const ConvParams& p, int classes) {
  TileChoice t = choose_tile(ctx, p.Cout, p.npix, p.nchunk, classes, p.out_nc8 != 0);
  const ConvPlanKey key = {MODE, p.B, p.Cin, p.H, p.W, p.Cout, p.Ho, p.Wo, p.stride, p.pad, p.nchunk,
                           ctx->conv_tail_split * 8 + ctx->conv_tile256 * 4 + (p.tab2 != nullptr ? ctx->conv_direct : 0) + 1024 * (p.in_nc8 * 2 + p.out_nc8),
                           0};
  if (ctx->conv_autotune && ctx->conv_max_split != 1) {
    bool found = false;
    for (auto& e : ctx->conv_plans)
      if (memcmp(&e.key, &key, sizeof(key)) == 0) {   // plan value: > 0 uniform split-K factor, < 0 tail split with -value slices
        t.ksplit = e.ksplit > 0 ? e.ksplit : 1;
        t.tail_s = e.ksplit < 0 ? -e.ksplit : 0;
        found = true;
        break;
      }
    if (!found && !ctx->capturing) {
      const int tiles = di_div_up(p.Cout, t.bm) * di_div_up(p.npix, t.bn) * classes;
      int cands[24], nc = 0;
      const int base[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24};
      for (int c : base)
        if ((c == 1 || (long)tiles * c <= 4096) && c <= max(1, p.nchunk / 4)) cands[nc++] = c;
      // tail split (LDS-free kernel only, opt-in): cut just the tiles of the under-filled last round, so that their
      // slices about fill one round of the chip; candidates are encoded as negative values. Off by default: measured
      // +2.8 % on conv2, ±1 % elsewhere, and a pair's summation order then depends on its position in the batch
      // (results stop being bit-identical under a permutation of the pairs).
      if (ctx->conv_tail_split && MODE == MODE_CONV && t.bm == 128 && t.bn == 128 && p.tab2 != nullptr && classes == 1) {
        const int slots = ctx->conv_tail_slots, R = tiles % slots;
        if (tiles >= slots && R > 0) {
          const int s0 = min(slots / R, max(1, p.nchunk / 4));
          for (int sN : {s0, s0 - 1, s0 * 2})
            if (sN >= 2 && sN <= max(1, p.nchunk / 4) && nc < 24) cands[nc++] = -sN;
        }
      }
      struct EventPair {   // destroyed on every exit path, including the DI_CHECK early returns below
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
      } ev;
      DI_CHECK(hipEventCreate(&ev.a));
      DI_CHECK(hipEventCreate(&ev.b));
      const hipEvent_t e0 = ev.a, e1 = ev.b;
      float best = 1e30f;
      int best_ks = t.ksplit;
      for (int ci = 0; ci < nc; ++ci) {
        TileChoice tc = t;
        tc.ksplit = cands[ci] > 0 ? cands[ci] : 1;
        tc.tail_s = cands[ci] < 0 ? -cands[ci] : 0;
        int rc = launch_one<MODE>(ctx, p, classes, tc);  // warm (also grows the scratch once)
        if (rc) return rc;
        DI_CHECK(hipEventRecord(e0, ctx->stream));
        for (int r = 0; r < 3; ++r) launch_one<MODE>(ctx, p, classes, tc);
        DI_CHECK(hipEventRecord(e1, ctx->stream));
        DI_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        DI_CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best * 0.985f) { best = ms; best_ks = cands[ci]; }  // prefer fewer splits on near-ties
      }
      if (getenv("DEEPIM_CONV_VERBOSE"))
        fprintf(stderr, "[deepim] conv plan B=%d Cin=%d %dx%d Cout=%d s%d: %d tiles, %s %d (%.3f ms)\n", p.B, p.Cin, p.H, p.W,
                p.Cout, p.stride, tiles, best_ks < 0 ? "tail split" : "split-K", best_ks < 0 ? -best_ks : best_ks, best / 3);
      ctx->conv_plans.push_back({key, best_ks});
      t.ksplit = best_ks > 0 ? best_ks : 1;
      t.tail_s = best_ks < 0 ? -best_ks : 0;
    }
  }
  if (ctx->conv_force_plan != 0) {   // dev knob: bypass heuristic and autotuner (> 0 uniform split-K, < 0 tail split)
    t.ksplit = ctx->conv_force_plan > 0 ? ctx->conv_force_plan : 1;
    t.tail_s = ctx->conv_force_plan < 0 ? -ctx->conv_force_plan : 0;
  }
  if (ctx->conv_max_split > 0 && t.ksplit > ctx->conv_max_split) t.ksplit = ctx->conv_max_split;
  if (ctx->conv_max_split > 0 && t.tail_s > ctx->conv_max_split) t.tail_s = ctx->conv_max_split > 1 ? ctx->conv_max_split : 0;
  return launch_one<MODE>(ctx, p, classes, t);
}

inline int gran_count(int Cout) { return di_div_up(Cout, 128) * 2; }  // whole 128-row tiles → BM=128 never overruns
inline int chunk_count(int K) { return di_div_up(K, KT); }

// tap tables are tiny; they live in the context (freed by deepim_destroy), keyed by geometry
int get_tab(deepim_ctx* ctx, int mode, int Cin, int kh, int kw, int H, int W, int2** out) {
  for (auto& k : ctx->conv_tabs) {
    if (k.mode == mode && k.Cin == Cin && k.kh == kh && k.kw == kw && k.H == H && k.W == W) {
      *out = (int2*)k.tab;
      return 0;
    }
  }
  DI_REQUIRE(!ctx->capturing, "conv tap table built during graph capture; run the sequence once eagerly first");
  const int K = mode == MODE_DECONV ? Cin * 4 : Cin * kh * kw;
  const int Kpad = chunk_count(K) * KT;
  int2* tab;
  DI_CHECK(hipMalloc((void**)&tab, (size_t)(Kpad + 32) * sizeof(int2)));
  if (mode == MODE_NC8_TAB)      // one entry per group of 8 K elements, padded past the end for the read-ahead
    hipLaunchKernelGGL(build_nc8_tab_kernel, dim3(di_div_up(Kpad / 8 + 24, 256)), dim3(256), 0, ctx->stream, tab, K / 8,
                       Kpad / 8 + 24, kh, kw, H, W);
  else if (mode == MODE_DIRECT_TAB)   // one entry per k-pair, padded past the end for the kernel's read-ahead
    hipLaunchKernelGGL(build_direct_tab_kernel, dim3(di_div_up(Kpad / 2 + 16, 256)), dim3(256), 0, ctx->stream, tab, K / 2,
                       Kpad / 2 + 16, kh, kw, H, W);
  else if (mode == MODE_CONV)
    hipLaunchKernelGGL(build_conv_tab_kernel, dim3(di_div_up(Kpad, 256)), dim3(256), 0, ctx->stream, tab, K, Kpad, kh,
                       kw, H, W);
  else
    hipLaunchKernelGGL(build_deconv_tab_kernel, dim3(di_div_up(Kpad, 256)), dim3(256), 0, ctx->stream, tab, K, Kpad, H,
                       W);
  DI_LAUNCH_CHECK();
  ctx->conv_tabs.push_back({mode, Cin, kh, kw, H, W, (void*)tab});
  *out = tab;
  return 0;
}

}  // namespace

// the packed buffer holds three layouts back to back, each gran·nchunk·1024 floats: [granule][chunk][16][64] for the
// LDS kernel, [32-row tile][k-pair/4][lane][4] for the LDS-free kernel on NCHW input, and the same shape in the
// channel order of the NC8 kernel
inline size_t packed_half(int Cout, int K) { return (size_t)gran_count(Cout) * chunk_count(K) * KT * GRAN; }

extern "C" size_t deepim_conv_packed_size(int Cout, int Cin, int kh, int kw) {
  return 3 * packed_half(Cout, Cin * kh * kw) * sizeof(float);
}

extern "C" int deepim_conv_pack_weights(deepim_ctx* ctx, float* packed_w, const float* w, int Cout, int Cin, int kh,
                                        int kw) {
  return deepim_conv_pack_weights_ex(ctx, packed_w, w, Cout, Cin, kh, kw, 7);
}

static int conv_pack_impl(deepim_ctx* ctx, float* packed_w, const float* w, int Cout, int Cin, int kh, int kw, int orders,
                          const WView& v) {
  DI_DEVICE(ctx);
  DI_REQUIRE(orders >= 1 && orders <= 7, "conv_pack_weights: orders = bits 1 (LDS kernel) | 2 (NCHW register-fed) | 4 (NC8)");
  const int K = Cin * kh * kw, nchunk = chunk_count(K);
  const long total = (long)gran_count(Cout) * nchunk * KT * GRAN;
  // three operand orders back to back; a caller that never runs one of the kernel families (the training graph: NCHW only)
  // leaves its order out — the slot stays allocated and unread
  if (orders & 1)
    hipLaunchKernelGGL(pack_conv_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, packed_w, w, Cout, K,
                       nchunk, total, v, kh * kw);
  if (orders & 2)
    hipLaunchKernelGGL(pack_direct_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, packed_w + total, w, Cout,
                       Cin, kh * kw, nchunk * (KT / 2), total, v);
  if (orders & 4) {
    DI_REQUIRE(!v.on, "conv_pack: the NC8 order is not built for data-gradient views");
    hipLaunchKernelGGL(pack_nc8_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, packed_w + 2 * total, w, Cout,
                       Cin, kh * kw, nchunk * 2, total);
  }
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_conv_pack_weights_ex(deepim_ctx* ctx, float* packed_w, const float* w, int Cout, int Cin, int kh, int kw,
                                           int orders) {
  const WView v = {0, 0, 0, 0, 0, 0, 1, kh, kw};
  return conv_pack_impl(ctx, packed_w, w, Cout, Cin, kh, kw, orders, v);
}

// Packed weights of a DATA GRADIENT straight from the layer's raw tensor w_layer (Co_l, Ci_l, kh_l, kw_l): the convolution that is
// packed has Cout = Ci_l, Cin = Co_l and an nky x nkx kernel, tap (a, b) = layer tap (ky0 + st (nky-1-a), kx0 + st (nkx-1-b)).
// st = 1, ky0 = kx0 = 0, nky = kh_l, nkx = kw_l: the transposed + flipped kernel (what deepim_conv_flip_weights + pack gave);
// st = 2: one output parity class of a stride-2 layer (what deepim_conv_subkernel_flip + pack gave). orders: 1 | 2.
extern "C" int deepim_conv_pack_dgrad(deepim_ctx* ctx, float* packed_w, const float* w_layer, int Co_l, int Ci_l, int kh_l,
                                      int kw_l, int ky0, int kx0, int st, int nky, int nkx, int orders) {
  DI_REQUIRE((st == 1 || st == 2) && ky0 >= 0 && kx0 >= 0 && nky >= 1 && nkx >= 1 && ky0 + st * (nky - 1) < kh_l &&
                 kx0 + st * (nkx - 1) < kw_l && (orders & 4) == 0,
             "conv_pack_dgrad: taps outside the layer's kernel (or an NC8 order asked for)");
  const WView v = {1, Ci_l, kh_l, kw_l, ky0, kx0, st, nky, nkx};
  return conv_pack_impl(ctx, packed_w, w_layer, Ci_l, Co_l, nky, nkx, orders, v);
}

// Which ONE of the packed operand orders deepim_conv2d_forward (NCHW in, NCHW out) reads for this geometry under the context's
// current options — the selection of conv2d_forward_impl / choose_tile / launch_one restated, so that a caller that re-packs
// weights it uses once (the data gradients of the training graph) packs only that order. Returns 1 (LDS kernel / few-output
// kernel) or 2 (NCHW register-fed kernel).
extern "C" int deepim_conv_weight_order(deepim_ctx* ctx, int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride,
                                        int pad) {
  const long npix = (long)B * ((H + 2 * pad - kh) / stride + 1) * ((W + 2 * pad - kw) / stride + 1);
  if (Cout <= 4 && ctx->conv_max_split != 1) return 1;
  const bool direct = ctx->conv_direct == 2 || (ctx->conv_direct == 1 && ctx->conv_max_split != 1);
  if (!(direct && (Cin & 1) == 0 && (Cout > 64 || npix >= 256L * 1024))) return 1;
  if (Cout <= 64) return 2;                                   // 64x256 tiles of the register-fed kernel
  return (ctx->conv_tile256 && Cout % 256 == 0) ? 1 : 2;      // 256-row tiles exist in the LDS kernel only
}

extern "C" int deepim_conv2d_forward(deepim_ctx* ctx, float* out, const float* in, const float* packed_w,
                                     const float* bias, int B, int Cin, int H, int W, int Cout, int kh, int kw,
                                     int stride, int pad, float slope, int out_ctotal, int out_coff) {
  DI_DEVICE(ctx);
  return deepim_conv2d_forward_ex(ctx, out, in, packed_w, bias, B, Cin, H, W, Cout, kh, kw, stride, pad, slope, out_ctotal,
                                  out_coff, 0, 0);
}

static int conv2d_forward_impl(deepim_ctx* ctx, float* out, const float* in, const float* packed_w, const float* bias, int B,
                               int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, float slope,
                               int out_ctotal, int out_coff, int in_nc8, int out_nc8, float out_scale,
                               const Remap* rm = nullptr, const ActGrad* ag = nullptr);

extern "C" int deepim_conv2d_forward_ex(deepim_ctx* ctx, float* out, const float* in, const float* packed_w,
                                        const float* bias, int B, int Cin, int H, int W, int Cout, int kh, int kw,
                                        int stride, int pad, float slope, int out_ctotal, int out_coff, int in_nc8,
                                        int out_nc8) {
  return conv2d_forward_impl(ctx, out, in, packed_w, bias, B, Cin, H, W, Cout, kh, kw, stride, pad, slope, out_ctotal, out_coff,
                             in_nc8, out_nc8 == 3 ? 3 : (out_nc8 ? 1 : 0), 1.f);   // 3: NC8 in space-to-depth order
}

// One output parity class of a stride-2 data gradient, straight into dx: a stride-1 convolution of the un-dilated gradient `in`
// (B,Cin,H,W) with the class's flipped sub-kernel (kh x kw, symmetric pad), whose result window [cy, cy + hq) x [cx, cx + wq)
// — hq = ⌈(Hd - py) / 2⌉, wq = ⌈(Wd - px) / 2⌉ — is written to out (B,Cout,Hd,Wd)[.., 2i + py, 2j + px]. No bias, no activation.
extern "C" int deepim_conv2d_forward_remap(deepim_ctx* ctx, float* out, const float* in, const float* packed_w, int B, int Cin,
                                           int H, int W, int Cout, int kh, int kw, int pad, int cy, int cx, int Hd, int Wd,
                                           int py, int px) {
  Remap rm;
  rm.on = 1; rm.cy = cy; rm.cx = cx; rm.py = py; rm.px = px; rm.H = Hd; rm.W = Wd;
  rm.hq = (Hd - py + 1) / 2; rm.wq = (Wd - px + 1) / 2;
  rm.Wo = W + 2 * pad - kw + 1;
  const int Ho = H + 2 * pad - kh + 1;
  DI_REQUIRE(py >= 0 && py < 2 && px >= 0 && px < 2 && cy >= 0 && cx >= 0 && rm.hq + cy <= Ho && rm.wq + cx <= rm.Wo,
             "conv2d_forward_remap: class window outside the convolution result");
  return conv2d_forward_impl(ctx, out, in, packed_w, nullptr, B, Cin, H, W, Cout, kh, kw, 1, pad, 1.f, 0, 0, 0, 0, 1.f, &rm);
}

// ---- the whole data gradient of a stride-2 convolution ----------------------------------------------------------------------
// dx (B,Ci_l,Hd,Wd) from dz (B,Co_l,Ho,Wo) and the layer's raw weights (Co_l,Ci_l,k,k): the four output parity classes
// (deepim_conv_pack_dgrad / deepim_conv2d_forward_remap above), planned and launched TOGETHER when the register-fed 128x128
// kernel takes them — one pack launch, one convolution launch whose blocks are shared out over the four classes (split-K per
// class so that every block runs about the same number of K chunks), one second pass. On conv4 … conv6 of the training graph a
// class alone fills a fraction of the chip (12-48 tiles); four launches + four second passes were launch-bound (19-49 TF).
namespace {
struct S2Class { int py, px, ky0, kx0, nky, nkx, cy, cx, P; };
inline S2Class s2_class(int z, int k, int pad) {
  S2Class c;
  c.py = z >> 1; c.px = z & 1;
  c.ky0 = (c.py + pad) % 2; c.kx0 = (c.px + pad) % 2;
  c.nky = (k - c.ky0 + 1) / 2; c.nkx = (k - c.kx0 + 1) / 2;
  c.P = max(c.nky, c.nkx) - 1;
  c.cy = (c.py + pad - c.ky0) / 2 + c.P - (c.nky - 1);
  c.cx = (c.px + pad - c.kx0) / 2 + c.P - (c.nkx - 1);
  return c;
}
}  // namespace

extern "C" size_t deepim_conv_dgrad_s2_packed_size(int Co_l, int Ci_l, int k, int pad) {
  size_t total = 0;
  for (int z = 0; z < 4; ++z) {
    const S2Class c = s2_class(z, k, pad);
    total += 3 * packed_half(Ci_l, Co_l * c.nky * c.nkx);
  }
  return total * sizeof(float);
}

namespace {
// the epilogue as a pass of its own, for the kernel families that do not carry it
__global__ __launch_bounds__(256) void actgrad_kernel(float* __restrict__ dx, const float* __restrict__ y, const float* __restrict__ add,
                                                      float slope, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = dx[i];
  if (add) v += add[i];
  dx[i] = y[i] > 0.f ? v : v * slope;
}
int actgrad_pass(deepim_ctx* ctx, float* dx, const ActGrad& ag, size_t n) {
  hipLaunchKernelGGL(actgrad_kernel, dim3(di_div_up((long)n, 256)), dim3(256), 0, ctx->stream, dx, ag.y, ag.add, ag.slope, n);
  DI_LAUNCH_CHECK();
  return 0;
}
}  // namespace

// forward use of the same machinery (a transposed convolution IS the data gradient of a stride-2 convolution): explicit size of the
// input map (the Crop may keep fewer rows than the adjoint's full frame), bias + LeakyReLU + channel slice in the final stores,
// weights packed once (s2_pack_direct_group; prepacked skips the pack launch)
struct S2Forward { int Ho, Wo; const float* bias; float slope; int out_ctotal, out_coff; bool prepacked; };
static int conv2d_dgrad_s2_impl(deepim_ctx* ctx, float* dx, const float* dz, const float* w_layer, float* packed_ws, int B,
                                int Ci_l, int Hd, int Wd, int Co_l, int k, int pad, const ActGrad* ag, const S2Forward* fw = nullptr);

extern "C" int deepim_conv2d_dgrad_s2(deepim_ctx* ctx, float* dx, const float* dz, const float* w_layer, float* packed_ws, int B,
                                      int Ci_l, int Hd, int Wd, int Co_l, int k, int pad) {
  return conv2d_dgrad_s2_impl(ctx, dx, dz, w_layer, packed_ws, B, Ci_l, Hd, Wd, Co_l, k, pad, nullptr);
}

extern "C" size_t deepim_conv_dgrad_packed_size(int Co_l, int Ci_l, int k, int stride, int pad) {
  return stride == 2 ? deepim_conv_dgrad_s2_packed_size(Co_l, Ci_l, k, pad) : deepim_conv_packed_size(Ci_l, Co_l, k, k);
}

// Data gradient of a Convolution layer (Co_l,Ci_l,k,k; stride 1 or 2) from its raw weights, optionally already multiplied by the
// activation gradient of the layer below: dx = lrelu'(act_y)·(dgrad [+ add]) — act_y = that layer's saved output, add = the
// gradient reaching it over a skip connection (both laid out like dx; NULL act_y = plain dgrad). The epilogue rides in the final
// stores of the register-fed kernels / their split-K second passes; other kernel families get it as one extra pass.
extern "C" int deepim_conv2d_dgrad(deepim_ctx* ctx, float* dx, const float* dz, const float* w_layer, float* packed_ws, int B,
                                   int Ci_l, int Hd, int Wd, int Co_l, int k, int stride, int pad, const float* act_y,
                                   const float* add, float slope) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(stride == 1 || stride == 2, "conv2d_dgrad: stride 1 or 2");
  DI_REQUIRE(act_y != nullptr || add == nullptr, "conv2d_dgrad: add without act_y");
  const ActGrad ag = {act_y, add, slope};
  if (stride == 2) return conv2d_dgrad_s2_impl(ctx, dx, dz, w_layer, packed_ws, B, Ci_l, Hd, Wd, Co_l, k, pad, act_y ? &ag : nullptr);
  const int Ho = Hd + 2 * pad - k + 1, Wo = Wd + 2 * pad - k + 1, P = k - 1 - pad;
  DI_REQUIRE(P >= 0, "conv2d_dgrad: pad > k - 1");
  const int order = deepim_conv_weight_order(ctx, B, Co_l, Ho, Wo, Ci_l, k, k, 1, P);
  int rc = deepim_conv_pack_dgrad(ctx, packed_ws, w_layer, Co_l, Ci_l, k, k, 0, 0, 1, k, k, order);
  if (rc) return rc;
  const bool fused = act_y && order == 2 && (size_t)B * Co_l * Ho * Wo * 4 + (size_t)(P * Wo + P) * 4 < 0x7fffffffUL;
  rc = conv2d_forward_impl(ctx, dx, dz, packed_ws, nullptr, B, Co_l, Ho, Wo, Ci_l, k, k, 1, P, 1.f, 0, 0, 0, 0, 1.f, nullptr,
                           fused ? &ag : nullptr);
  if (rc) return rc;
  if (act_y && !fused) return actgrad_pass(ctx, dx, ag, (size_t)B * Ci_l * Hd * Wd);
  return 0;
}

// The four parity classes of a stride-2 layer's operand buffer, a function of (Ci_l, Co_l, k, pad) only: class z at slot[z] (three
// halves each: LDS-kernel packing, then the register-fed kernel's operand order), members of a grouped launch in descending K
// (ord: the long blocks are dispatched first). Shared by the pack entry and the launches so the two cannot drift apart.
struct S2Layout {
  S2Class cls[4];
  size_t slot[4];
  int ord[4];
};
static S2Layout s2_layout(int Ci_l, int Co_l, int k, int pad) {
  S2Layout L;
  size_t off = 0;
  for (int z = 0; z < 4; ++z) {
    L.cls[z] = s2_class(z, k, pad);
    L.slot[z] = off;
    off += 3 * packed_half(Ci_l, Co_l * L.cls[z].nky * L.cls[z].nkx);
    L.ord[z] = z;
  }
  for (int a = 0; a < 4; ++a)
    for (int b = a + 1; b < 4; ++b)
      if (L.cls[L.ord[b]].nky * L.cls[L.ord[b]].nkx > L.cls[L.ord[a]].nky * L.cls[L.ord[a]].nkx) { const int t = L.ord[a]; L.ord[a] = L.ord[b]; L.ord[b] = t; }
  return L;
}
// The register-fed kernel's operands of all four classes from the layer's own (Co_l, Ci_l, k, k) weights, one launch. No geometry: the
// operand order does not depend on the image (deepim_deconv_pack_weights packs once per update, the data gradient per call).
static int s2_fill_pack_group(PackGroup& pg, const S2Layout& L, float* packed_ws, int Ci_l, int Co_l, int k) {
  pg.n = 4;
  pg.Cout = Ci_l; pg.Cin = Co_l;
  int pstart = 0;
  for (int m = 0; m < 4; ++m) {
    const S2Class& c = L.cls[L.ord[m]];
    const size_t half = packed_half(Ci_l, Co_l * c.nky * c.nkx);
    pg.dst[m] = packed_ws + L.slot[L.ord[m]] + half;
    pg.total[m] = (long)half;
    pg.khw[m] = c.nky * c.nkx;
    pg.npair[m] = chunk_count(Co_l * c.nky * c.nkx) * (KT / 2);
    pg.v[m] = WView{1, Ci_l, k, k, c.ky0, c.kx0, 2, c.nky, c.nkx};
    pg.start[m] = pstart;
    pstart += (int)di_div_up((long)half, 256);
  }
  pg.start[4] = pstart;
  return pstart;
}
static int s2_pack_direct_group(deepim_ctx* ctx, const float* w_layer, float* packed_ws, int Ci_l, int Co_l, int k, int pad) {
  DI_REQUIRE(k >= 2 && k <= 7 && pad >= 0 && pad < k, "conv2d_dgrad_s2: kernel 2 … 7, pad < k");
  const S2Layout L = s2_layout(Ci_l, Co_l, k, pad);
  PackGroup pg;
  const int pstart = s2_fill_pack_group(pg, L, packed_ws, Ci_l, Co_l, k);
  hipLaunchKernelGGL(pack_direct_group_kernel, dim3(pstart), dim3(256), 0, ctx->stream, w_layer, pg);
  DI_LAUNCH_CHECK();
  return 0;
}

static int conv2d_dgrad_s2_impl(deepim_ctx* ctx, float* dx, const float* dz, const float* w_layer, float* packed_ws, int B,
                                int Ci_l, int Hd, int Wd, int Co_l, int k, int pad, const ActGrad* ag, const S2Forward* fw) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(k >= 2 && k <= 7 && pad >= 0 && pad < k, "conv2d_dgrad_s2: kernel 2 … 7, pad < k");
  const int Ho = fw ? fw->Ho : (Hd + 2 * pad - k) / 2 + 1, Wo = fw ? fw->Wo : (Wd + 2 * pad - k) / 2 + 1;
  const S2Layout L = s2_layout(Ci_l, Co_l, k, pad);
  const S2Class* cls = L.cls;
  const size_t* slot = L.slot;
  const int* ord = L.ord;
  const bool direct = ctx->conv_direct == 2 || (ctx->conv_direct == 1 && ctx->conv_max_split != 1);
  const size_t in_bytes = (size_t)B * Co_l * Ho * Wo * 4;
  const bool grouped = fw != nullptr || (ctx->dgrad_group && direct && (Co_l & 1) == 0 && !(ctx->conv_tile256 && Ci_l % 256 == 0) &&
                                         in_bytes + (size_t)(4 * Wo + 4) * 4 < 0x7fffffffUL);
  // tile of the register-fed kernel: 128 x 128, or 64 rows x 256 pixels when dx has at most 64 channels (conv2's data gradient)
  const int bm = Ci_l <= 64 ? 64 : 128, bn = Ci_l <= 64 ? 256 : 128;
  if (!grouped) {   // class by class: whatever kernel family deepim_conv2d_forward picks for the geometry
    for (int z = 0; z < 4; ++z) {
      const S2Class& c = cls[z];
      const int order = deepim_conv_weight_order(ctx, B, Co_l, Ho, Wo, Ci_l, c.nky, c.nkx, 1, c.P);
      int rc = deepim_conv_pack_dgrad(ctx, packed_ws + slot[z], w_layer, Co_l, Ci_l, k, k, c.ky0, c.kx0, 2, c.nky, c.nkx, order);
      if (rc) return rc;
      rc = deepim_conv2d_forward_remap(ctx, dx, dz, packed_ws + slot[z], B, Co_l, Ho, Wo, Ci_l, c.nky, c.nkx, c.P, c.cy, c.cx, Hd,
                                       Wd, c.py, c.px);
      if (rc) return rc;
    }
    return ag ? actgrad_pass(ctx, dx, *ag, (size_t)B * Ci_l * Hd * Wd) : 0;
  }
  // members in descending K (S2Layout::ord)
  ConvGroup g;
  PackGroup pg;
  g.n = 4;
  const int pstart = s2_fill_pack_group(pg, L, packed_ws, Ci_l, Co_l, k);
  long tiles[4];
  for (int m = 0; m < 4; ++m) {
    const S2Class& c = cls[ord[m]];
    ConvParams& p = g.p[m];
    const size_t half = packed_half(Ci_l, Co_l * c.nky * c.nkx);
    p.in = dz; p.wp = packed_ws + slot[ord[m]]; p.bias = fw ? fw->bias : nullptr; p.out = dx;
    p.B = B; p.Cin = Co_l; p.H = Ho; p.W = Wo; p.Cout = Ci_l;
    p.Ho = Ho + 2 * c.P - c.nky + 1; p.Wo = Wo + 2 * c.P - c.nkx + 1;
    p.stride = 1; p.pad = c.P;
    p.nchunk = chunk_count(Co_l * c.nky * c.nkx);
    p.ngran = gran_count(Ci_l);
    p.out_ctotal = fw && fw->out_ctotal > 0 ? fw->out_ctotal : Ci_l; p.out_coff = fw ? fw->out_coff : 0;
    p.slope = fw ? fw->slope : 1.f; p.crop_y = p.crop_x = 0;
    p.rm_on = 1; p.rm_cy = c.cy; p.rm_cx = c.cx; p.rm_hq = (Hd - c.py + 1) / 2; p.rm_wq = (Wd - c.px + 1) / 2;
    p.rm_py = c.py; p.rm_px = c.px; p.rm_H = Hd; p.rm_W = Wd;
    DI_REQUIRE(p.rm_hq + c.cy <= p.Ho && p.rm_wq + c.cx <= p.Wo, "conv2d_dgrad_s2: class window outside the convolution result");
    p.npix = (long)B * p.Ho * p.Wo;
    p.pad_bytes = (c.P * Wo + c.P) * 4;
    p.in_bytes = (unsigned)in_bytes;
    int2 *tab, *tab2;
    int rc = get_tab(ctx, MODE_CONV, Co_l, c.nky, c.nkx, Ho, Wo, &tab);
    if (rc) return rc;
    rc = get_tab(ctx, MODE_DIRECT_TAB, Co_l, c.nky, c.nkx, Ho, Wo, &tab2);
    if (rc) return rc;
    p.tab = tab; p.tab2 = tab2;
    p.wd = packed_ws + slot[ord[m]] + half; p.wd_bytes = (unsigned)(half * sizeof(float));
    p.in_nc8 = 0; p.out_nc8 = 0; p.out_s2d = 0; p.out_scale = 1.f; p.status = ctx->status; p.wd8 = nullptr; p.tab8 = nullptr;
    p.ep_y = ag ? ag->y : nullptr; p.ep_add = ag ? ag->add : nullptr; p.ep_slope = ag ? ag->slope : 1.f;
    p.swizzle = ctx->conv_xcd_swizzle;
    p.gx = di_div_up(p.npix, bn); p.gy = di_div_up(Ci_l, bm);
    p.n_full = 0; p.tail_s = 0; p.tail_cps = 0; p.n_tail_pad = 0; p.tail_partial = nullptr;
    tiles[m] = (long)p.gx * p.gy;
  }
  // joint plan: T = K chunks per block; member m runs ceil(nchunk_m / T) slices. cost as plan_ksplit: rounds of 256 blocks x the
  // longest block, + the second pass
  const int nch_max = g.p[0].nchunk;
  const int cands[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24, 32};
  float best = 1e30f;
  int best_T = nch_max;
  for (int sN : cands) {
    const int T = di_div_up(nch_max, sN);
    if (sN > 1 && T < 4) continue;
    long blocks = 0, split_blocks = 0;
    for (int m = 0; m < 4; ++m) {
      const int ks = di_div_up(g.p[m].nchunk, T);
      blocks += tiles[m] * ks;
      if (ks > 1) split_blocks += tiles[m] * ks;
    }
    if (blocks > 8192) continue;
    const float cost = (float)di_div_up(blocks, 256) * (float)T + (split_blocks ? 3.f + 0.016f * (float)split_blocks : 0.f);
    if (cost < best * 0.985f) { best = cost; best_T = T; }
  }
  size_t partial_floats = 0;
  for (int m = 0; m < 4; ++m) {
    ConvParams& p = g.p[m];
    p.ksplit = di_div_up(p.nchunk, best_T);
    p.chunks_per_split = di_div_up(p.nchunk, p.ksplit);
    p.ksplit = di_div_up(p.nchunk, p.chunks_per_split);
    p.gz = p.ksplit;
    p.partial_stride = (long)B * Ci_l * p.Ho * p.Wo;
    if (p.ksplit > 1) partial_floats += (size_t)p.ksplit * p.partial_stride;
  }
  float* scratch = nullptr;
  if (partial_floats) {
    void* sp;
    int rc = deepim_scratch(ctx, partial_floats * sizeof(float), &sp);
    if (rc) return rc;
    scratch = (float*)sp;
  }
  ReduceGroup rg;
  rg.out = dx; rg.Cout = Ci_l; rg.n = 0;
  rg.ep_y = ag ? ag->y : nullptr; rg.ep_add = ag ? ag->add : nullptr; rg.ep_slope = ag ? ag->slope : 1.f;
  rg.bias = fw ? fw->bias : nullptr; rg.slope = fw ? fw->slope : 1.f;
  rg.out_ctotal = fw && fw->out_ctotal > 0 ? fw->out_ctotal : Ci_l; rg.out_coff = fw ? fw->out_coff : 0;
  int cstart = 0, rstart = 0;
  size_t poff = 0;
  for (int m = 0; m < 4; ++m) {
    ConvParams& p = g.p[m];
    p.partial = nullptr;
    if (p.ksplit > 1) {
      p.partial = scratch + poff;
      poff += (size_t)p.ksplit * p.partial_stride;
      const int j = rg.n++;
      rg.partial[j] = p.partial; rg.total[j] = p.partial_stride; rg.stride[j] = p.partial_stride; rg.S[j] = p.ksplit;
      rg.hw[j] = p.Ho * p.Wo; rg.rm[j] = remap_of(p); rg.start[j] = rstart;
      rstart += (int)di_div_up(p.partial_stride, 256);
    }
    g.start[m] = cstart;
    cstart += di_div_up(p.gx * p.gy * p.gz, 8) * 8;
  }
  g.start[4] = cstart;
  for (int j = rg.n; j <= CONV_GROUP_MAX; ++j) rg.start[j] = rstart;
  if (!(fw && fw->prepacked)) hipLaunchKernelGGL(pack_direct_group_kernel, dim3(pstart), dim3(256), 0, ctx->stream, w_layer, pg);
  if (bm == 64) hipLaunchKernelGGL(conv_direct_group_kernel<1>, dim3(cstart), dim3(256), 0, ctx->stream, g);
  else hipLaunchKernelGGL(conv_direct_group_kernel<2>, dim3(cstart), dim3(256), 0, ctx->stream, g);
  if (rg.n) hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(rstart), dim3(256), 0, ctx->stream, rg);
  DI_LAUNCH_CHECK();
  return 0;
}

// NCHW fp32 in → split16 out (conv1 of the split-fp16 encoder): the fp32 MFMA convolution with the split folded into its
// epilogue. Cout % 16 == 0; LDS-free kernels only (even Cin).
extern "C" int deepim_conv2d_forward_split16(deepim_ctx* ctx, void* out_split16, const float* in, const float* packed_w,
                                             const float* bias, int B, int Cin, int H, int W, int Cout, int kh, int kw,
                                             int stride, int pad, float slope, float out_scale) {
  DI_REQUIRE((Cout & 15) == 0 && (Cin & 1) == 0, "conv2d_split16: needs Cout % 16 == 0 and an even Cin");
  return conv2d_forward_impl(ctx, (float*)out_split16, in, packed_w, bias, B, Cin, H, W, Cout, kh, kw, stride, pad, slope, 0, 0,
                             0, 2, out_scale);
}

static int conv2d_forward_impl(deepim_ctx* ctx, float* out, const float* in, const float* packed_w, const float* bias, int B,
                               int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, float slope,
                               int out_ctotal, int out_coff, int in_nc8, int out_nc8, float out_scale, const Remap* rm,
                               const ActGrad* ag) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE((long)Cin * H * W < (1L << 31), "conv2d: per-sample input too large for 32-bit offsets");
  {
    // the kernels address the input through one raw-buffer descriptor with bit 31 of the offset as the padding marker,
    // so a launch sees < 2 GiB of input; larger batches run as consecutive sub-batches (samples are independent)
    const size_t per_sample = (size_t)Cin * H * W * 4, limit = 0x7fffffffUL - (size_t)(pad * W + pad) * 4;
    if ((size_t)B * per_sample >= limit) {
      DI_REQUIRE(ag == nullptr, "conv2d: activation-gradient epilogue on a sub-batched launch (input >= 2 GiB)");
      const int Bc = (int)((limit - 1) / per_sample);
      DI_REQUIRE(Bc >= 1, "conv2d: one sample exceeds 2 GiB");
      const int Ho_ = (H + 2 * pad - kh) / stride + 1, Wo_ = (W + 2 * pad - kw) / stride + 1;
      const size_t out_sample = (size_t)(out_ctotal > 0 ? out_ctotal : Cout) * (rm ? (size_t)rm->H * rm->W : (size_t)Ho_ * Wo_);   // split16: 2·Cout halves = Cout floats
      for (int b0 = 0; b0 < B; b0 += Bc) {
        const int rc = conv2d_forward_impl(ctx, out + (size_t)b0 * out_sample, in + (size_t)b0 * Cin * H * W, packed_w, bias,
                                           min(Bc, B - b0), Cin, H, W, Cout, kh, kw, stride, pad, slope, out_ctotal,
                                           out_coff, in_nc8, out_nc8, out_scale, rm);
        if (rc) return rc;
      }
      return 0;
    }
  }
  ConvParams p;
  p.in = in; p.wp = packed_w; p.bias = bias; p.out = out;
  p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
  p.Ho = (H + 2 * pad - kh) / stride + 1;
  p.Wo = (W + 2 * pad - kw) / stride + 1;
  p.stride = stride; p.pad = pad;
  p.nchunk = chunk_count(Cin * kh * kw);
  p.ngran = gran_count(Cout);
  p.out_ctotal = out_ctotal > 0 ? out_ctotal : Cout;
  p.out_coff = out_coff;
  p.slope = slope; p.crop_y = p.crop_x = 0;
  p.rm_on = 0; p.rm_cy = p.rm_cx = p.rm_hq = p.rm_wq = p.rm_py = p.rm_px = p.rm_H = p.rm_W = 0;
  if (rm) {
    DI_REQUIRE(!in_nc8 && !out_nc8 && Cout > 4, "conv2d remap: NCHW in / out on the MFMA kernels only");
    p.rm_on = 1; p.rm_cy = rm->cy; p.rm_cx = rm->cx; p.rm_hq = rm->hq; p.rm_wq = rm->wq; p.rm_py = rm->py; p.rm_px = rm->px;
    p.rm_H = rm->H; p.rm_W = rm->W;
  }
  p.npix = (long)B * p.Ho * p.Wo;
  p.pad_bytes = (pad * W + pad) * 4;
  DI_REQUIRE((size_t)B * Cin * H * W * 4 + p.pad_bytes < 0x7fffffffUL, "conv2d: input tensor must be < 2 GiB per launch");
  DI_REQUIRE(kh <= 7 && kw <= 7, "conv2d: kernel larger than 7 not supported");
  p.in_bytes = (unsigned)((size_t)B * Cin * H * W * 4);
  int2* tab;
  int rc = get_tab(ctx, MODE_CONV, Cin, kh, kw, H, W, &tab);
  if (rc) return rc;
  p.tab = tab;
  p.wd = nullptr; p.tab2 = nullptr; p.wd_bytes = 0;
  p.out_s2d = out_nc8 == 3 ? 1 : 0;
  if (out_nc8 == 3) out_nc8 = 1;
  p.in_nc8 = in_nc8 ? 1 : 0; p.out_nc8 = out_nc8; p.out_scale = out_scale; p.status = ctx->status; p.wd8 = nullptr; p.tab8 = nullptr;
  p.ep_y = ag ? ag->y : nullptr; p.ep_add = ag ? ag->add : nullptr; p.ep_slope = ag ? ag->slope : 1.f;
  if (out_nc8 == 2) DI_REQUIRE(!in_nc8, "conv2d: split16 output is built for the NCHW-input LDS-free kernel (conv1)");
  if (out_nc8) DI_REQUIRE((Cout & 7) == 0 && p.out_ctotal == Cout && out_coff == 0, "conv2d: NC8 output needs Cout % 8 == 0 and no channel slice");
  if (p.out_s2d) DI_REQUIRE(((p.Ho | p.Wo) & 1) == 0, "conv2d: space-to-depth output needs even output height and width");
  if (in_nc8) {
    DI_REQUIRE((Cin & 7) == 0 && (Cout > 64 || (Cout == 64 && out_nc8 == 1)),
               "conv2d: NC8 input needs Cin % 8 == 0 and Cout > 64 (or Cout == 64 with NC8 output: the 64x256-tile kernel)");
    const size_t half = packed_half(Cout, Cin * kh * kw);
    int2* tab8;
    rc = get_tab(ctx, MODE_NC8_TAB, Cin, kh, kw, H, W, &tab8);
    if (rc) return rc;
    p.tab8 = tab8;
    p.wd8 = packed_w + 2 * half;
    p.wd_bytes = (unsigned)(half * sizeof(float));
    return launch_conv<MODE_CONV>(ctx, p, 1);
  }
  if (Cout <= 4 && ctx->conv_max_split != 1 && p.out_nc8 == 0) {   // heads: a stream over the input, not an MFMA problem
    // channel slices when the pixels alone leave the chip empty: ~512 blocks, slices of at least 32 channels (fixed by the geometry)
    // the 3x3 stride-1 pad-1 heads on rows of whole quads: four pixels per lane (conv_fewout_quad_kernel); dev option conv_fewout_quad = 0: the one-pixel form
    const bool quad = kh == 3 && kw == 3 && stride == 1 && pad == 1 && (W & 3) == 0 && ctx->conv_fewout_quad &&
                      (((uintptr_t)out | (uintptr_t)in) & 15) == 0;
    const int nblk = quad ? di_div_up(p.npix / 4, 64) : di_div_up(p.npix, 64);
    // measured with the DPP halo (profiles/r06_heads.md): ~1024 blocks where the pixels give >= 32 blocks of their own (the 30x40 and 15x20
    // heads at B = 32: 53 -> 39 us), ~512 below (the 8x10 head and every head at B = 4: more slices only add second-pass traffic)
    const int target = ctx->conv_fewout_blocks > 0 ? ctx->conv_fewout_blocks : (nblk >= 32 ? 1024 : 512);
    int S = max(1, min(min(target / nblk, Cin / ctx->conv_fewout_minc), 32));
    const int cslice = di_div_up(Cin, S);
    S = di_div_up(Cin, cslice);
    float* partial = nullptr;
    const long total = (long)B * Cout * p.Ho * p.Wo;
    if (S > 1) {
      void* scratch;
      rc = deepim_scratch(ctx, (size_t)S * total * sizeof(float), &scratch);
      if (rc) return rc;
      partial = (float*)scratch;
    }
    const dim3 grid(nblk, S);
    if (quad) {
#define DI_FEWOUT_Q(C)                                                                                                  \
  hipLaunchKernelGGL((conv_fewout_quad_kernel<C>), grid, dim3(512), 0, ctx->stream, out, in, packed_w, bias, Cin, H, W, p.npix / 4, \
                     p.in_bytes, p.out_ctotal, out_coff, slope, partial, cslice)
      if (Cout == 1) DI_FEWOUT_Q(1); else if (Cout == 2) DI_FEWOUT_Q(2); else if (Cout == 3) DI_FEWOUT_Q(3); else DI_FEWOUT_Q(4);
#undef DI_FEWOUT_Q
    } else {
#define DI_FEWOUT(C, KS)                                                                                               \
  hipLaunchKernelGGL((conv_fewout_kernel<C, KS>), grid, dim3(1024), 0, ctx->stream, out, in, packed_w, bias, Cin, H, W, p.Ho, \
                     p.Wo, kh, kw, stride, pad, p.npix, p.out_ctotal, out_coff, slope, partial, cslice)
    const bool k3 = kh == 3 && kw == 3;   // the heads are all 3x3; other sizes take the generic loop
    if (Cout == 1) { if (k3) DI_FEWOUT(1, 3); else DI_FEWOUT(1, 0); }
    else if (Cout == 2) { if (k3) DI_FEWOUT(2, 3); else DI_FEWOUT(2, 0); }
    else if (Cout == 3) { if (k3) DI_FEWOUT(3, 3); else DI_FEWOUT(3, 0); }
    else { if (k3) DI_FEWOUT(4, 3); else DI_FEWOUT(4, 0); }
#undef DI_FEWOUT
    }
    if (S > 1) {
      Remap none = {};
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, out, partial, bias, total, total,
                         S, Cout, p.Ho * p.Wo, p.out_ctotal, out_coff, slope, none, (const float*)nullptr, (const float*)nullptr, 1.f);
    }
    DI_LAUNCH_CHECK();
    return 0;
  }
  // LDS-free kernel: 128x128-tiled layers with even Cin. conv_max_split = 1 asks for the canonical single
  // (ci,ky,kx)-ordered chain per output, which only the LDS kernel provides; conv_direct = 2 forces the LDS-free kernel
  // regardless (its chain runs over (ci/2,ky,kx,ci%2)).
  const bool direct = out_nc8 || ctx->conv_direct == 2 || (ctx->conv_direct == 1 && ctx->conv_max_split != 1);
  if (out_nc8) DI_REQUIRE((Cin & 1) == 0, "conv2d: NC8 output needs an even Cin (LDS-free kernels only)");
  if (direct && (Cin & 1) == 0 && (Cout > 64 || out_nc8 || p.npix >= 256L * 1024)) {   // the tile shapes the LDS-free kernel has
    const size_t half = packed_half(Cout, Cin * kh * kw);
    int2* tab2;
    rc = get_tab(ctx, MODE_DIRECT_TAB, Cin, kh, kw, H, W, &tab2);
    if (rc) return rc;
    p.tab2 = tab2;
    p.wd = packed_w + half;
    p.wd_bytes = (unsigned)(half * sizeof(float));
  }
  return launch_conv<MODE_CONV>(ctx, p, 1);
}

// [LDS-kernel operand order of the four output-parity classes | the same four 2x2-tap sub-kernels in the register-fed kernel's
// operand order (deepim_conv2d_dgrad_s2's packing: Deconvolution k4 s2 + Crop(1,1) is the data gradient of a k4 s2 p1 convolution
// whose (filters, channels) are the MXNet Deconvolution weight's (Cin, Cout))]. The second part exists for even Cin and Cout >= 64.
static size_t deconv_lds_pack_floats(int Cin, int Cout) { return (size_t)4 * gran_count(Cout) * chunk_count(Cin * 4) * KT * GRAN; }
static bool deconv_direct_ok(int Cin, int Cout) { return (Cin & 1) == 0 && Cout >= 64; }
extern "C" size_t deepim_deconv_packed_size(int Cin, int Cout) {
  return deconv_lds_pack_floats(Cin, Cout) * sizeof(float) + (deconv_direct_ok(Cin, Cout) ? deepim_conv_dgrad_s2_packed_size(Cin, Cout, 4, 1) : 0);
}

extern "C" int deepim_deconv_pack_weights(deepim_ctx* ctx, float* packed_w, const float* w, int Cin, int Cout) {
  DI_DEVICE(ctx);
  const int nchunk = chunk_count(Cin * 4), ngran = gran_count(Cout);
  const long total = (long)4 * ngran * nchunk * KT * GRAN;
  hipLaunchKernelGGL(pack_deconv_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, packed_w, w, Cin, Cout,
                     ngran, nchunk, total);
  DI_LAUNCH_CHECK();
  if (deconv_direct_ok(Cin, Cout))   // the register-fed kernel's operands of the four parity classes (k = 4, pad = 1): no geometry involved
    return s2_pack_direct_group(ctx, w, packed_w + deconv_lds_pack_floats(Cin, Cout), Cout, Cin, 4, 1);
  return 0;
}

extern "C" int deepim_deconv4x4s2_crop_forward(deepim_ctx* ctx, float* out, const float* in, const float* packed_w,
                                               const float* bias, int B, int Cin, int H, int W, int Cout, int Ho,
                                               int Wo, int crop_y, int crop_x, float slope, int out_ctotal,
                                               int out_coff) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(Ho + crop_y <= (H - 1) * 2 + 4 && Wo + crop_x <= (W - 1) * 2 + 4, "deconv: crop exceeds output");
  // The decoder's two big transposed convolutions (deconv5 1024 → 512, deconv4 1026 → 256; crop (1,1)) on the register-fed MFMA
  // kernel: four stride-1 2x2-tap convolutions of the input, one per output parity class, in ONE grouped launch whose final stores
  // (or one grouped second pass) put bias + LeakyReLU results onto the class's pixels of the Concat slice — the machinery of the
  // stride-2 data gradient (deepim_conv2d_dgrad_s2), here with weights packed once. Round 3 ran them on the LDS-staged kernel at
  // 64-77 TFLOP/s (profiles/r03_heads_b4_iteration_trace.txt). Not in the bit-exact configuration (conv_max_split = 1 keeps the
  // canonical (ci,ky,kx) order of the LDS kernel, as for the encoder).
  const bool direct = ctx->conv_direct == 2 || (ctx->conv_direct == 1 && ctx->conv_max_split != 1);
  if (direct && crop_y == 1 && crop_x == 1 && deconv_direct_ok(Cin, Cout) && !ctx->conv_tile256 &&
      (size_t)B * Cin * H * W * 4 + (size_t)(4 * W + 4) * 4 < 0x7fffffffUL) {
    const S2Forward fw = {H, W, bias, slope, out_ctotal > 0 ? out_ctotal : Cout, out_coff, true};
    return conv2d_dgrad_s2_impl(ctx, out, in, nullptr, const_cast<float*>(packed_w) + deconv_lds_pack_floats(Cin, Cout), B, Cout, Ho, Wo,
                                Cin, 4, 1, nullptr, &fw);
  }
  ConvParams p;
  p.in = in; p.wp = packed_w; p.bias = bias; p.out = out;
  p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
  p.Ho = Ho; p.Wo = Wo; p.stride = 2; p.pad = 0;
  p.nchunk = chunk_count(Cin * 4);
  p.ngran = gran_count(Cout);
  p.out_ctotal = out_ctotal > 0 ? out_ctotal : Cout;
  p.out_coff = out_coff;
  p.slope = slope; p.crop_y = crop_y; p.crop_x = crop_x;
  p.rm_on = 0; p.rm_cy = p.rm_cx = p.rm_hq = p.rm_wq = p.rm_py = p.rm_px = p.rm_H = p.rm_W = 0;
  // every parity class has at most ceil(Ho/2)*ceil(Wo/2) pixels; size the grid for the largest, the
  // kernel masks with its own per-class count
  p.npix = (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2);
  p.pad_bytes = (W + 1) * 4;
  DI_REQUIRE((size_t)B * Cin * H * W * 4 + p.pad_bytes < 0x7fffffffUL, "deconv: input tensor must be < 2 GiB per launch");
  p.in_bytes = (unsigned)((size_t)B * Cin * H * W * 4);
  int2* tab;
  int rc = get_tab(ctx, MODE_DECONV, Cin, 4, 4, H, W, &tab);
  if (rc) return rc;
  p.tab = tab;
  p.wd = nullptr; p.tab2 = nullptr; p.wd_bytes = 0;
  p.in_nc8 = p.out_nc8 = p.out_s2d = 0; p.wd8 = nullptr; p.tab8 = nullptr;
  p.ep_y = p.ep_add = nullptr; p.ep_slope = 1.f;
  return launch_conv<MODE_DECONV>(ctx, p, 4);
}

extern "C" int deepim_upsample16_crop_forward(deepim_ctx* ctx, float* out, const float* in, const float* w, int B,
                                              int C, int H, int W, int Ho, int Wo, int crop_y, int crop_x,
                                              float scale) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  if ((Wo & 3) == 0 && (crop_x & 3) == 0 && crop_x >= 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)w & 15) == 0) {
    const long total = (long)B * C * Ho * (Wo >> 2);      // four outputs of a row per thread
    hipLaunchKernelGGL(upsample16x4_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, out, in, w, C, H, W, Ho, Wo, crop_y,
                       crop_x, scale, total);
  } else {
    dim3 grid(di_div_up(Wo, 256), Ho, B * C);
    hipLaunchKernelGGL(upsample16_kernel, grid, dim3(256), 0, ctx->stream, out, in, w, C, H, W, Ho, Wo, crop_y, crop_x,
                       scale);
  }
  DI_LAUNCH_CHECK();
  return 0;
}

// NC8 in space-to-depth order <-> NCHW: (B, C, H, W) <-> [n][(phase*C + c)/8][H/2][W/2][8], phase = (y&1)*2 + (x&1)
__global__ __launch_bounds__(256) void relayout_s2d_kernel(float* __restrict__ dst, const float* __restrict__ src, int C, int H, int W,
                                                           long total, int to_s2d) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // index in the space-to-depth tensor
  if (i >= total) return;
  const int q = (int)(i & 7), hw4 = (H >> 1) * (W >> 1), C8 = C >> 3;
  const int pix = (int)((i >> 3) % hw4);
  const long plane = (i >> 3) / hw4;
  const long n = plane / (4 * C8);
  const int r = (int)(plane % (4 * C8)), ph = r / C8, c = (r % C8) * 8 + q;
  const int y = 2 * (pix / (W >> 1)) + (ph >> 1), x = 2 * (pix % (W >> 1)) + (ph & 1);
  const long j = ((n * C + c) * H + y) * W + x;
  if (to_s2d) dst[i] = src[j]; else dst[j] = src[i];
}
extern "C" int deepim_relayout_nc8_s2d(deepim_ctx* ctx, float* dst, const float* src, int B, int C, int H, int W, int to_s2d) {
  DI_DEVICE(ctx);
  if (B == 0 || C == 0) return 0;
  DI_REQUIRE((C & 7) == 0 && ((H | W) & 1) == 0, "relayout_nc8_s2d: C % 8 == 0, even H and W");
  const long total = (long)B * C * H * W;
  hipLaunchKernelGGL(relayout_s2d_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, dst, src, C, H, W, total, to_s2d);
  DI_LAUNCH_CHECK();
  return 0;
}

// NC8 -> NCHW straight into channels [dst_coff, dst_coff + C) of a dst_ctotal-channel tensor: the skip connections of the refinement
// decoder (Concat2 / Concat3, deepIM_flownet.py:128-131, :143-146) without the intermediate NCHW copy and its 2-D blit
extern "C" int deepim_relayout_nc8_slice(deepim_ctx* ctx, float* dst, int dst_ctotal, int dst_coff, const float* src_nc8, int B, int C,
                                         size_t hw) {
  DI_DEVICE(ctx);
  if (B == 0 || C == 0) return 0;
  DI_REQUIRE((C & 7) == 0 && dst_ctotal >= dst_coff + C && dst_coff >= 0, "relayout_nc8_slice: C % 8 == 0 and the slice inside the tensor");
  const long total = (long)B * C * (long)hw;
  hipLaunchKernelGGL(relayout_nc8_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, dst, src_nc8, C, (int)hw, total, 0,
                     dst_ctotal, dst_coff);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_relayout_nc8(deepim_ctx* ctx, float* dst, const float* src, int B, int C, size_t hw, int to_nc8) {
  DI_DEVICE(ctx);
  if (B == 0 || C == 0) return 0;
  DI_REQUIRE((C & 7) == 0, "relayout_nc8: C must be a multiple of 8");
  const long total = (long)B * C * (long)hw;
  hipLaunchKernelGGL(relayout_nc8_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, dst, src, C, (int)hw, total,
                     to_nc8);
  DI_LAUNCH_CHECK();
  return 0;
}
