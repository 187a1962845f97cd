// fp32 Winograd F(2x2, 3x3) for the encoder's 3x3 stride-1 layers (conv3_1 / conv4_1 / conv5_1 / conv6_1 of
// deepIM_flownet.py:69-101 — 36.6 % of the encoder's multiply-adds) on channel-blocked activations [n][C/8][h][w][8].
// Same fp32 arithmetic as the direct kernels, 2.25x fewer multiplies: per 2x2 output tile and channel the 4x4 input patch d
// becomes V = B^T d B, the 3x3 kernel g becomes U = G g G^T (packed once), the sixteen positions (xi, nu) are sixteen
// independent GEMMs M = U·V over the input channels, and the tile is Y = A^T M A. Results differ from the direct sum in the
// last bits (bounded by the tests at 1e-5 of the layer's range). The host class asks deepim_conv_wino_preferred[_s2d] per layer at
// bind time (network.WINOGRAD_CONV, default on); a context in the canonical-order configuration (conv_max_split = 1: every sum one
// fmaf chain, bit-exact against the oracle) gets 0 from it. Which layers qualify, and the K-split plan of the shared-transform kernel,
// depend on the batch size — so does the rounding of a sample's output (never its 1e-5 bound).
//
// conv_wino_kernel (round 4; now the fallback for Cout % 64 != 0):
// One kernel, no LDS, no intermediate tensor in HBM:
//   * a wave owns 32 output channels x 32 tiles x ALL 16 positions: sixteen 32x32 accumulators = 256 AGPRs, one wave per SIMD
//     (the 512-register budget of gfx950 at occupancy 1); the MFMAs of one position are independent of the next, so a single
//     wave keeps the matrix pipe busy as long as its operands are loaded a body ahead.
//   * lane (h = lane / 32, t = lane % 32): tile t of the wave's 32 consecutive tiles, k index h of v_mfma_f32_32x32x2_f32.
//     Per block of 8 input channels lane h reads channels 4h..4h+3 of each patch pixel with ONE 16-byte load (a wave touches
//     32 bytes of every pixel record it visits): body j = 0, 1 of the block multiplies channels 4h + 2j + s in its two k-steps
//     s. 16 pixel loads + 16 weight loads (all 16 bytes) + 128 adds of transform per 64 MFMAs. (Measured: 8-byte pixel loads, one
//     body at a time, spend 26 % of the kernel in the texture path — 902 vs 667 µs on conv3_1 without them.)
//   * patch addresses: 16 per-lane voffsets computed once (out-of-image pixels carry bit 31 → the buffer load returns 0, which
//     IS the zero padding); the channel walk lives in the scalar offset. No per-load VALU.
//   * input transform in registers while the previous body multiplies; output transform + bias + LeakyReLU in the lane that
//     holds all 16 positions of its (channel, tile) pairs; NC8 stores of 16 bytes (or NCHW for the layer fc6 reads).
//   * block = 4 waves on 32 channels x 128 tiles; blocks are dealt to the XCDs so that each XCD keeps a fixed slice of the
//     transformed weights (<= 2 MB for conv4_1 / conv5_1) resident in its L2.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoParams {
  const float* in;
  float* out;
  const float* wd;     // [Cout/32][Cin/8][16 positions][64 lanes][4]
  const float* bias;
  float slope;
  int Cin, Cout, H, W, TY, TX, ntiles, gx, gy;
  unsigned in_bytes, wd_bytes, out_bytes;   // out_bytes: one channel-blocked copy of the output (conv_wino8_kernel's buffer stores; < 2 GB)
  int out_ctotal, out_coff;   // NCHW output: channel slice of a wider tensor
  int out_s2d;                // NC8 output in space-to-depth order (the input format of the next stride-2 layer on this kernel)
  // conv_wino8_kernel, split over the input channels: slice = blockIdx.x / grid0 walks the 8-channel blocks [slice*kslice, +kslice) and
  // writes its raw sums (no bias, no activation) to out + slice*part_stride — `out` is then the partial buffer, wino_reduce_kernel finishes
  int grid0, kslice, ksplit, nvb;   // nvb = grid0 * ksplit virtual blocks (conv_wino8_kernel's blocks are persistent)
  long part_stride;
  float* part;                // raw sums of the pieces that do not cover all of Cin: copy c at part + c * part_stride, laid out like the output
  // stream-K of the last, partly filled round (sk_G > 0). XCD x (the blocks b with b % 8 == x) owns the tile blocks bid % 8 == x, as in the
  // plain walk; its sk_nlb blocks first walk sk_F whole tile blocks each, side by side (local tile block r * sk_nlb + lb: the blocks of an
  // XCD read the same weights at the same time), then deal the remaining tile blocks' granules — sk_G per tile block, sk_gran K steps
  // each — in consecutive runs of sk_q (+ 1 for the first sk_rem blocks). A run cuts tile blocks into pieces: piece c (in K order)
  // writes raw sums to copy c of `part`, bumps the tile block's counter in sk_count, and whichever piece arrives last adds the copies in
  // order, the bias and the activation (a fixed order of fp32 adds: deterministic whatever the arrival order). Counters return to zero.
  int sk_G, sk_gran, sk_F, sk_q, sk_rem, sk_nlb;
  int* sk_count;
  int fin_nchw, fin_ctotal, fin_coff;   // in-kernel finish of K slices whose output is NCHW (dense NCHW partials): the real channel slice of the output
};

#ifndef WINO_ABL
#define WINO_ABL 0   // dev ablations (wrong results): 1 = no pixel loads in the loop, 2 = no weight loads, 4 = no transform
#endif

// WINO_PK 0 (with -fno-slp-vectorize): component-wise scalar adds instead of the compiler's mix of v_pk_add_f32 and scalar ops —
// measured the same within noise (0.779 vs 0.758 ms on conv3_1), so the plain vector form stays
#ifndef WINO_PK
#define WINO_PK 1
#endif
__device__ __forceinline__ f32x4 wsub(f32x4 a, f32x4 b) {
  if (WINO_PK) return a - b;
  f32x4 r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; r.w = a.w - b.w; return r;
}
__device__ __forceinline__ f32x4 wadd(f32x4 a, f32x4 b) {
  if (WINO_PK) return a + b;
  f32x4 r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; r.w = a.w + b.w; return r;
}

// dev builds (-DWINO_TRACE=1, tools/wino_trace.py): the four waves of one block in the middle of the grid sum the s_memtime ticks of
// their body 0 (pixel + weight loads beside the MFMAs) and body 1 (transform + weight loads) segments; stamps are read a segment
// after they were issued, so the scalar wait is free
#ifndef WINO_TRACE
#define WINO_TRACE 0
#endif
#if WINO_TRACE
__device__ unsigned long long* g_wino_trace = nullptr;
#define WTR_STAMP(dst) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0" : "=s"(dst) : "s"(tr_prev) : "memory"); 
#endif

// S2D: the input is the space-to-depth form of a stride-2 layer's input and the weights come from deepim_conv_wino_pack_weights_s2d:
// the channels of input phase (py, px) — a quarter of Cin each, in the order (0,0) (0,1) (1,0) (1,1) — carry a 3x3 kernel whose
// third row (py = 1) / column (px = 1) is zero, so U[xi = 3][.] / U[.][nu = 3] vanish identically for them: the K loop runs phase by
// phase and skips those positions' MFMAs, weight loads, transform adds and the patch pixels only they read — 49 of 64
// (position, phase) pairs are left, the skipped terms are exact zeros.
template <int OUT_NC8, int S2D = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_wino_kernel(WinoParams p) {
#if WINO_TRACE
  const unsigned long long tr_t0 = __builtin_amdgcn_s_memtime();
  unsigned long long tr_a = 0, tr_b = 0, tr_prev = 0, tr_sum0 = 0, tr_sum1 = 0, tr_loop0 = 0;
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 5, lcol = lane & 31;
  int mb, bx;
  {
    const int bid = blockIdx.x;
    if ((p.gy & 7) == 0) {   // XCD x (blocks x, x+8, ...) owns channel blocks [x·gy/8, (x+1)·gy/8): channel block fastest inside
      const int per = p.gy >> 3, xcd = bid & 7, idx = bid >> 3;
      mb = xcd * per + idx % per;
      bx = idx / per;
    } else {
      mb = bid % p.gy;
      bx = bid / p.gy;
    }
  }
  const int t = bx * 128 + wave * 32 + lcol;
  const int tpi = p.TY * p.TX;
  const bool tvalid = t < p.ntiles;
  const int n = tvalid ? t / tpi : 0;
  const int tr = tvalid ? t - n * tpi : 0;
  const int ty = tr / p.TX, tx = tr - ty * p.TX;

  // patch pixel (i, j) = input (2ty - 1 + i, 2tx - 1 + j); column-major slot k = j*4 + i (the order the loads are issued in)
  int voff[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = k & 3, j = k >> 2;
    const int y = 2 * ty - 1 + i, x = 2 * tx - 1 + j;
    const bool ok = tvalid && y >= 0 && y < p.H && x >= 0 && x < p.W;
    voff[k] = ok ? (((n * (p.Cin >> 3)) * p.H + y) * p.W + x) * 32 + lrow * 16 : (int)0x80000000;
    if (WINO_ABL & 8) {   // dev: same bytes per load, but the wave's 64 lanes read two contiguous 512-byte runs (wrong results)
      const int t0 = min(bx * 128 + wave * 32, p.ntiles - 32);
      const int n0 = t0 / tpi, r0 = t0 - n0 * tpi, ty0 = r0 / p.TX, tx0 = r0 - ty0 * p.TX;
      const int y0 = min(max(2 * ty0 - 1 + i, 0), p.H - 2), x0 = min(max(2 * tx0 - 1 + j, 0), p.W - 1);
      voff[k] = (((n0 * (p.Cin >> 3)) * p.H + y0) * p.W + x0) * 32 + lcol * 16 + lrow * 512;
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wd, 0, (int)p.wd_bytes, 0x00020000);
  int wvo[4];   // weights: 1 KB per position, 16 KB per block of 8 channels; immediate offsets reach 4 KB
#pragma unroll
  for (int i = 0; i < 4; ++i) wvo[i] = lane * 16 + i * 4096;
  const int hw32 = p.H * p.W * 32;
  const int c8n = p.Cin >> 3;
  const int abase = mb * c8n;   // first block of 8 input channels of this channel block in the packed weights (16 KB each)

  f32x16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  f32x4 raw[16], T[16];        // pixels of the NEXT channel block (4 channels per lane = both bodies) and their row pass
  f32x2 Va[16], Vb[16], Vc[16];
  f32x4 A[16];   // weights of both bodies of a position (.xy body 0, .zw body 1): ONE 16-byte load, reloaded right behind its last use
#define WLOADB(k, soff) \
  raw[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[k], (soff), 0));
#define WLOADA(q, soff) \
  A[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrw, wvo[(q) >> 2] + ((q) & 3) * 1024, (soff), 0));
// row pass of B^T d B, one output per call: T[xi*4 + j] from patch column j (raw slots j*4 + i). (One call per MFMA slot: a slot
// that carries a whole column's 16 scalar adds overruns the 64-cycle MFMA gap of the wave's only SIMD.)
#define WROW(k)                                                                                \
  {                                                                                            \
    const int j_ = (k) >> 2, w_ = (k) & 3;                                                     \
    const f32x4 d0 = raw[j_ * 4 + 0], d1 = raw[j_ * 4 + 1], d2 = raw[j_ * 4 + 2], d3 = raw[j_ * 4 + 3]; \
    T[w_ * 4 + j_] = w_ == 0 ? wsub(d0, d2) : w_ == 1 ? wadd(d1, d2) : w_ == 2 ? wsub(d2, d1) : wsub(d1, d3); \
  }
// column pass: position k = xi*4 + nu from T[xi*4 + 0..3]; channels .xy go to body 0's V, .zw to body 1's
#define WCOL(V0_, V1_, k)                                                                      \
  {                                                                                            \
    const int x_ = (k) >> 2, w_ = (k) & 3;                                                     \
    const f32x4 t0 = T[x_ * 4 + 0], t1 = T[x_ * 4 + 1], t2 = T[x_ * 4 + 2], t3 = T[x_ * 4 + 3]; \
    const f32x4 r_ = w_ == 0 ? wsub(t0, t2) : w_ == 1 ? wadd(t1, t2) : w_ == 2 ? wsub(t2, t1) : wsub(t3, t1);   /* nu = 3 NEGATED on both sides (pack_wino_kernel) */ \
    V0_[k] = r_.xy; V1_[k] = r_.zw;                                                            \
  }

  // prologue: weights of bodies 0 and 1, pixels of channel block 0 -> V of both its bodies, then the pixels of block 1
  // (issue order pinned: the loop header's one s_waitcnt serves both the entry and the back edge — with the weights loaded
  // last here, as the scheduler would have it, it becomes vmcnt(0) and drains everything in flight once per trip)
#pragma unroll
  for (int q = 0; q < 16; ++q) WLOADA(q, abase * 16384)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 16; ++k) WLOADB(k, 0)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 16; ++k) WROW(k)
#pragma unroll
  for (int k = 0; k < 16; ++k) WCOL(Va, Vb, k)
  __builtin_amdgcn_sched_barrier(0);
  {
    const int sb1 = __builtin_amdgcn_readfirstlane(min(1, c8n - 1) * hw32);
#pragma unroll
    for (int k = 0; k < 16; ++k) WLOADB(k, sb1)
  }
  __builtin_amdgcn_sched_barrier(0);

  // channel block c = two bodies of 4 channels (32 MFMAs each). Body 0 multiplies Va; body 1 multiplies X while the transform of
  // block c + 1 runs in its slots — row passes in the first 16, column passes in the last 16 (into Va, free by now, and Y), and
  // behind each row pass slot's column the 16 pixel loads of block c + 2 refill `raw` (48 slots ahead of their first use).
  // Every A[q] (16 bytes = both bodies of position q) is reloaded for the next block right behind its last use in body 1, 32 slots
  // ahead of its next. Wide loads, few of them: a single wave pays ~23 cycles of MFMA issue per load instruction
  // (tools/wino_issue_probe.hip: a b64 load every other slot costs 18 %, the same bytes as a b128 every fourth slot 12 %; two
  // v_pk_add_f32 per slot cost MORE than the four scalar adds they replace; a second wave per SIMD hides the loads, not the adds).
  // X / Y swap roles from block to block.
#define WMFMA(J, VV, q, s_) \
  acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32((J) ? ((s_) ? A[q].w : A[q].z) : ((s_) ? A[q].y : A[q].x), (s_) ? VV[q].y : VV[q].x, acc[q], 0, 0, 0); \
  asm volatile("" : "+a"(acc[q]));
// MFMA order inside a body: positions in pairs, (q0,s0) (q1,s0) (q0,s1) (q1,s1) — WINO_PAIR 0 puts the two k-steps of a
// position back to back (an accumulator dependent on the MFMA right before it)
#ifndef WINO_PAIR
#define WINO_PAIR 1
#endif
#define WQ(sl) (WINO_PAIR ? (((sl) >> 2) * 2 + ((sl) & 1)) : ((sl) >> 1))
#define WS(sl) (WINO_PAIR ? (((sl) >> 1) & 1) : ((sl) & 1))
#if WINO_TRACE
#define WTR0 { tr_prev = tr_b; WTR_STAMP(tr_a) if (tr_prev) tr_sum1 += tr_a - tr_prev; else tr_loop0 = tr_a; }
#define WTR1 { tr_prev = tr_a; WTR_STAMP(tr_b) tr_sum0 += tr_b - tr_prev; }
#else
#define WTR0
#define WTR1
#endif
// phase masks: P < 0 = every position; else py = P >> 1, px = P & 1
#define WACT(P, q) ((P) < 0 || !((((P) >> 1) && ((q) >> 2) == 3) || (((P) & 1) && ((q) & 3) == 3)))   /* position q = xi*4 + nu */
#define WPIX(P, k) ((P) < 0 || !((((P) >> 1) && ((k) & 3) == 3) || (((P) & 1) && ((k) >> 2) == 3)))   /* patch slot k = j*4 + i */
// PC: phase of the block being multiplied; PN: of the next block (its transform and weight loads run here); PB: of the block
// after that (its pixel loads are issued here)
#define WSUPER(X, Y, sb, sa, PC, PN, PB)                                                       \
  WTR0                                                                                         \
  _Pragma("unroll") for (int sl = 0; sl < 32; ++sl) {                                          \
    if (WACT(PC, WQ(sl))) { WMFMA(0, Va, WQ(sl), WS(sl)) }                                     \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  }                                                                                            \
  WTR1                                                                                         \
  _Pragma("unroll") for (int sl = 0; sl < 32; ++sl) {                                          \
    if (WACT(PC, WQ(sl))) { WMFMA(1, X, WQ(sl), WS(sl)) }                                      \
    if (!(WINO_ABL & 2) && WS(sl) && WACT(PN, WQ(sl))) { WLOADA(WQ(sl), sa) }                  \
    if (!(WINO_ABL & 4)) {                                                                     \
      if (sl < 16) { if (WPIX(PN, ((sl) >> 2) * 4 + 0) && WACT(PN, ((sl) & 3) * 4)) WROW(sl) } /* row op sl: column sl/4, xi = sl%4 */ \
      else if (WACT(PN, sl - 16)) WCOL(Va, Y, sl - 16)                                         \
    }                                                                                          \
    if ((WINO_ABL & 16) && sl < 16) asm volatile("" :: "v"(raw[sl]));   /* dev: pixel loads kept alive without the transform */ \
    if (!(WINO_ABL & 1) && sl >= 16 && WPIX(PB, sl - 16)) { WLOADB(sl - 16, sb) }              \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  }
#define WOFFS(c8)                                                                              \
  const int sb = __builtin_amdgcn_readfirstlane(min((c8) + 2, c8n - 1) * hw32); /* clamped: loads past the end re-read the last block */ \
  const int sa = __builtin_amdgcn_readfirstlane((abase + min((c8) + 1, c8n - 1)) * 16384);

  int c8 = 0;
  if (!S2D) {
    for (; c8 + 2 <= c8n; c8 += 2) {
      { WOFFS(c8) WSUPER(Vb, Vc, sb, sa, -1, -1, -1) }
      { WOFFS(c8 + 1) WSUPER(Vc, Vb, sb, sa, -1, -1, -1) }
    }
    if (c8 < c8n) { WOFFS(c8) WSUPER(Vb, Vc, sb, sa, -1, -1, -1) }
  } else {
    // four phases of c8n / 4 blocks each (an even number: the launcher checks); the last two blocks of a phase prepare the next one's
    const int npb = c8n >> 2;
#define WPHASE(PH, NX)                                                                         \
    for (const int e_ = ((PH) + 1) * npb - 2; c8 < e_; c8 += 2) {                              \
      { WOFFS(c8) WSUPER(Vb, Vc, sb, sa, PH, PH, PH) }                                         \
      { WOFFS(c8 + 1) WSUPER(Vc, Vb, sb, sa, PH, PH, PH) }                                     \
    }                                                                                          \
    { WOFFS(c8) WSUPER(Vb, Vc, sb, sa, PH, PH, NX) }                                           \
    { WOFFS(c8 + 1) WSUPER(Vc, Vb, sb, sa, PH, NX, NX) }                                       \
    c8 += 2;
    WPHASE(0, 1) WPHASE(1, 2) WPHASE(2, 3) WPHASE(3, -1)
#undef WPHASE
  }
#undef WOFFS
#undef WSUPER
#undef WACT
#undef WPIX
#undef WMFMA
#undef WCOL
#undef WROW
#undef WLOADA
#undef WLOADB

#if WINO_TRACE
  const unsigned long long tr_t2 = __builtin_amdgcn_s_memtime();
#endif
  // output transform Y = A^T M A per (channel, tile), bias, LeakyReLU, store. acc[q][r]: channel (r&3) + 8(r>>2) + 4·lrow, tile lcol
  if (!tvalid) return;
  const int y0 = 2 * ty, x0 = 2 * tx;
  const bool y1ok = y0 + 1 < p.H, x1ok = x0 + 1 < p.W;
  // the lane's 16 bias values in one batch (element-wise loads behind a null check each serialise four round trips per g); the
  // opaque copy of mb keeps the loads BELOW the K loop, where their 16 registers would spill
  float4 bq[4];
  int mbe = mb;
  asm volatile("" : "+s"(mbe));
#pragma unroll
  for (int g = 0; g < 4; ++g)
    bq[g] = p.bias ? *reinterpret_cast<const float4*>(p.bias + mbe * 32 + 8 * g + 4 * lrow) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float o[4][4];   // [a*2+b][channel within the run of 4]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * g + e;
      float s[4][2];
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        const float m0 = acc[xi * 4 + 0][r], m1 = acc[xi * 4 + 1][r], m2 = acc[xi * 4 + 2][r], m3 = acc[xi * 4 + 3][r];
        s[xi][0] = (m0 + m1) + m2;
        s[xi][1] = (m1 - m2) - m3;
      }
      const float bv = e == 0 ? bq[g].x : e == 1 ? bq[g].y : e == 2 ? bq[g].z : bq[g].w;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float v0 = ((s[0][b] + s[1][b]) + s[2][b]) + bv;
        float v1 = ((s[1][b] - s[2][b]) - s[3][b]) + bv;
        o[0 * 2 + b][e] = v0 > 0.f ? v0 : v0 * p.slope;
        o[1 * 2 + b][e] = v1 > 0.f ? v1 : v1 * p.slope;
      }
    }
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      const int a = ab >> 1, b = ab & 1;
      if ((a && !y1ok) || (b && !x1ok)) continue;
      const long pix = (long)(y0 + a) * p.W + x0 + b;
      if (OUT_NC8) {
        long rec;
        if (p.out_s2d)   // output (2ty + a, 2tx + b) is pixel (ty, tx) of phase plane a*2 + b: consecutive tiles, consecutive records
          rec = (((long)n * 4 + ab) * (p.Cout >> 3) + mb * 4 + g) * (p.TY * p.TX) + ty * p.TX + tx;
        else
          rec = ((long)n * (p.Cout >> 3) + mb * 4 + g) * p.H * p.W + pix;
        *reinterpret_cast<float4*>(p.out + rec * 8 + 4 * lrow) = make_float4(o[ab][0], o[ab][1], o[ab][2], o[ab][3]);
      } else {
        const long c0 = (long)n * p.out_ctotal + p.out_coff + mb * 32 + 8 * g + 4 * lrow;
#pragma unroll
        for (int e = 0; e < 4; ++e) p.out[(c0 + e) * p.H * p.W + pix] = o[ab][e];
      }
    }
  }
#if WINO_TRACE
  if (g_wino_trace && blockIdx.x == gridDim.x / 2 && lane == 0) {   // [wave][8]: prologue, sum body 0, sum body 1, epilogue, total, blocks of 8 channels
    const unsigned long long tr_t3 = __builtin_amdgcn_s_memtime();
    unsigned long long* o = g_wino_trace + wave * 8;
    o[0] = tr_loop0 - tr_t0; o[1] = tr_sum0; o[2] = tr_sum1 + (tr_t2 - tr_b); o[3] = tr_t3 - tr_t2; o[4] = tr_t3 - tr_t0; o[5] = (unsigned long long)c8n;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Two waves per SIMD (deepim_set_option "wino_two_wave" = 1; NOT the default). The same algorithm with the sixteen positions of a
// (32 channels x 32 tiles) tile split over TWO waves — "top": xi = 0, 1 (patch rows 0..2), "bottom": xi = 2, 3 (rows 1..3) —
// 8 accumulators = 128 AGPRs + 128 VGPRs each, so that two waves share a SIMD and one wave's loads, waits, prologue and epilogue
// run under the other's MFMAs (tools/wino_issue_probe.hip: a second wave hides the load issue a single wave pays for). Measured:
// each half loads 3 of the 4 patch rows and its own weights, 28 loads per 32 MFMAs instead of 32 per 64, and that costs more than
// the overlap gains — conv3_1 0.92 vs 0.75 ms, conv4_1 0.83 vs 0.71 at B = 32 (0.84 with the same bytes on contiguous addresses:
// the texture path); it wins only where the one-wave grid is between one and two rounds (conv5_1 at B = 32: 0.27 vs 0.31 ms).
// Block = 4 waves = 2 tile groups x (top, bottom) on 32 channels x 64 tiles, two blocks per CU. Both halves run the same stream:
// with (f0, f1, f2) = patch rows (0, 1, 2) / (2, 3, 1) and sgn = +1 / -1 the row pass is Ta = f0 - f2, Tb = fma(f1, sgn, f2)
// (= d0 - d2, d1 + d2 / d2 - d1, d1 - d3); only the row each load slot addresses differs. The output transform needs the other
// half's two partial rows: bottom waves hand them over through 16 KB of LDS per tile group, top waves finish and store.
template <int OUT_NC8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino2_kernel(WinoParams p) {
  __shared__ float xch[2][64][64];   // [tile group][value][lane]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 1, grp = wave & 1;
  const int lrow = lane >> 5, lcol = lane & 31;
  int mb, bx;
  {
    const int bid = blockIdx.x;
    if ((p.gy & 7) == 0) {
      const int per = p.gy >> 3, xcd = bid & 7, idx = bid >> 3;
      mb = xcd * per + idx % per;
      bx = idx / per;
    } else {
      mb = bid % p.gy;
      bx = bid / p.gy;
    }
  }
  const int t = bx * 64 + grp * 32 + lcol;
  const int tpi = p.TY * p.TX;
  const bool tvalid = t < p.ntiles;
  const int n = tvalid ? t / tpi : 0;
  const int tr = tvalid ? t - n * tpi : 0;
  const int ty = tr / p.TX, tx = tr - ty * p.TX;

  // load slot k = j*3 + e: patch column j, row operand f_e of this half
  int voff[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int j = k / 3, e = k - 3 * j;
    const int i = half ? (e == 0 ? 2 : e == 1 ? 3 : 1) : e;
    const int y = 2 * ty - 1 + i, x = 2 * tx - 1 + j;
    const bool ok = tvalid && y >= 0 && y < p.H && x >= 0 && x < p.W;
    voff[k] = ok ? (((n * (p.Cin >> 3)) * p.H + y) * p.W + x) * 32 + lrow * 16 : (int)0x80000000;
    if (WINO_ABL & 8) {   // dev: same bytes per load, the wave's 64 lanes on two contiguous 512-byte runs (wrong results)
      const int t0 = min(bx * 64 + grp * 32, p.ntiles - 32);
      const int n0 = t0 / tpi, r0 = t0 - n0 * tpi, ty0 = r0 / p.TX, tx0 = r0 - ty0 * p.TX;
      const int y0 = min(max(2 * ty0 - 1 + i, 0), p.H - 2), x0 = min(max(2 * tx0 - 1 + j, 0), p.W - 1);
      voff[k] = (((n0 * (p.Cin >> 3)) * p.H + y0) * p.W + x0) * 32 + lcol * 16 + lrow * 512;
    }
  }
  const float sgn = half ? -1.f : 1.f;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wd, 0, (int)p.wd_bytes, 0x00020000);
  int wvo[2];   // this half's 8 positions: 1 KB each (16 bytes per lane, body 0 in the first 8)
#pragma unroll
  for (int i = 0; i < 2; ++i) wvo[i] = lane * 16 + half * 8192 + i * 4096;
  const int hw32 = p.H * p.W * 32;
  const int c8n = p.Cin >> 3;
  const int abase = mb * c8n;

  f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  f32x4 raw[12], T[8];
  f32x2 Va[8], Vb[8], Vc[8], A[8];   // A: ONE body's weights, reloaded for the next body right behind its last use
#define W2LOADB(k, soff) \
  raw[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[k], (soff), 0));
#define W2LOADA(q, soff, body) \
  A[q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrw, wvo[(q) >> 2] + ((q) & 3) * 1024 + (body) * 8, (soff), 0));
// row pass, one output per call: o = j*2 + which -> T[which*4 + j]
#define W2ROW(o)                                                                               \
  {                                                                                            \
    const int j_ = (o) >> 1;                                                                   \
    const f32x4 f0 = raw[j_ * 3 + 0], f1 = raw[j_ * 3 + 1], f2 = raw[j_ * 3 + 2];              \
    if ((o) & 1) { f32x4 r_; r_.x = __builtin_fmaf(f1.x, sgn, f2.x); r_.y = __builtin_fmaf(f1.y, sgn, f2.y);                  \
                   r_.z = __builtin_fmaf(f1.z, sgn, f2.z); r_.w = __builtin_fmaf(f1.w, sgn, f2.w); T[4 + j_] = r_; }         \
    else T[j_] = wsub(f0, f2);                                                                 \
  }
#define W2COL(V0_, V1_, k)                                                                     \
  {                                                                                            \
    const int x_ = (k) >> 2, w_ = (k) & 3;                                                     \
    const f32x4 t0 = T[x_ * 4 + 0], t1 = T[x_ * 4 + 1], t2 = T[x_ * 4 + 2], t3 = T[x_ * 4 + 3]; \
    const f32x4 r_ = w_ == 0 ? wsub(t0, t2) : w_ == 1 ? wadd(t1, t2) : w_ == 2 ? wsub(t2, t1) : wsub(t3, t1);   /* nu = 3 NEGATED on both sides (pack_wino_kernel) */ \
    V0_[k] = r_.xy; V1_[k] = r_.zw;                                                            \
  }
#define W2MFMA(VV, q, s_) \
  acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32((s_) ? A[q].y : A[q].x, (s_) ? VV[q].y : VV[q].x, acc[q], 0, 0, 0); \
  asm volatile("" : "+a"(acc[q]));

#pragma unroll
  for (int q = 0; q < 8; ++q) W2LOADA(q, abase * 16384, 0)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 12; ++k) W2LOADB(k, 0)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 8; ++o) W2ROW(o)
#pragma unroll
  for (int k = 0; k < 8; ++k) W2COL(Va, Vb, k)
  __builtin_amdgcn_sched_barrier(0);

  // block c: body 0 multiplies Va (weights .xy) and receives the 12 pixel loads of block c + 1; body 1 multiplies X (weights .zw)
  // while the transform of block c + 1 runs in its slots (8 row passes, 8 column passes into Va and Y)
#define W2SUPER(X, Y, sb, sa, san)                                                             \
  _Pragma("unroll") for (int sl = 0; sl < 16; ++sl) {                                          \
    W2MFMA(Va, WQ(sl), WS(sl))                                                                 \
    if (sl < 12) { W2LOADB(sl, sb) }                                                           \
    if (WS(sl)) { W2LOADA(WQ(sl), sa, 1) }                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  }                                                                                            \
  _Pragma("unroll") for (int sl = 0; sl < 16; ++sl) {                                          \
    W2MFMA(X, WQ(sl), WS(sl))                                                                  \
    if (WS(sl)) { W2LOADA(WQ(sl), san, 0) }                                                    \
    if (sl < 8) W2ROW(sl) else W2COL(Va, Y, sl - 8)                                            \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  }
#define W2OFFS(c8)                                                                             \
  const int sb = __builtin_amdgcn_readfirstlane(min((c8) + 1, c8n - 1) * hw32);                \
  const int sa = __builtin_amdgcn_readfirstlane((abase + (c8)) * 16384);                       \
  const int san = __builtin_amdgcn_readfirstlane((abase + min((c8) + 1, c8n - 1)) * 16384);
  int c8 = 0;
  for (; c8 + 2 <= c8n; c8 += 2) {
    { W2OFFS(c8) W2SUPER(Vb, Vc, sb, sa, san) }
    { W2OFFS(c8 + 1) W2SUPER(Vc, Vb, sb, sa, san) }
  }
  if (c8 < c8n) { W2OFFS(c8) W2SUPER(Vb, Vc, sb, sa, san) }
#undef W2OFFS
#undef W2SUPER
#undef W2MFMA
#undef W2COL
#undef W2ROW
#undef W2LOADA
#undef W2LOADB

  // output transform. This half's partial rows per accumulator row r: sA[b], sB[b] from its xi = 0 / 1 (top: s0, s1; bottom: s2, s3)
  // Y[0][b] = (s0 + s1) + s2, Y[1][b] = s1 + (-s2 - s3): bottom hands {s2, -s2 - s3} over, top adds and stores.
  float* xl = &xch[grp][0][lane];
  if (half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float a0 = acc[0][r], a1 = acc[1][r], a2 = acc[2][r], a3 = acc[3][r];
        const float c0 = acc[4][r], c1 = acc[5][r], c2 = acc[6][r], c3 = acc[7][r];
        const float sA = b == 0 ? (a0 + a1) + a2 : (a1 - a2) - a3;
        const float sB = b == 0 ? (c0 + c1) + c2 : (c1 - c2) - c3;
        xl[(r * 4 + b) * 64] = sA;
        xl[(r * 4 + 2 + b) * 64] = -sA - sB;
      }
    }
  }
  __syncthreads();
  if (half || !tvalid) return;
  const int y0 = 2 * ty, x0 = 2 * tx;
  const bool y1ok = y0 + 1 < p.H, x1ok = x0 + 1 < p.W;
  float4 bq[4];
  int mbe = mb;
  asm volatile("" : "+s"(mbe));
#pragma unroll
  for (int g = 0; g < 4; ++g)
    bq[g] = p.bias ? *reinterpret_cast<const float4*>(p.bias + mbe * 32 + 8 * g + 4 * lrow) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float o[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * g + e;
      const float bv = e == 0 ? bq[g].x : e == 1 ? bq[g].y : e == 2 ? bq[g].z : bq[g].w;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float a0 = acc[0][r], a1 = acc[1][r], a2 = acc[2][r], a3 = acc[3][r];
        const float c0 = acc[4][r], c1 = acc[5][r], c2 = acc[6][r], c3 = acc[7][r];
        const float sA = b == 0 ? (a0 + a1) + a2 : (a1 - a2) - a3;
        const float sB = b == 0 ? (c0 + c1) + c2 : (c1 - c2) - c3;
        float v0 = ((sA + sB) + xl[(r * 4 + b) * 64]) + bv;
        float v1 = (sB + xl[(r * 4 + 2 + b) * 64]) + bv;
        o[0 * 2 + b][e] = v0 > 0.f ? v0 : v0 * p.slope;
        o[1 * 2 + b][e] = v1 > 0.f ? v1 : v1 * p.slope;
      }
    }
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      const int a = ab >> 1, b = ab & 1;
      if ((a && !y1ok) || (b && !x1ok)) continue;
      const long pix = (long)(y0 + a) * p.W + x0 + b;
      if (OUT_NC8) {
        long rec;
        if (p.out_s2d)
          rec = (((long)n * 4 + ab) * (p.Cout >> 3) + mb * 4 + g) * (p.TY * p.TX) + ty * p.TX + tx;
        else
          rec = ((long)n * (p.Cout >> 3) + mb * 4 + g) * p.H * p.W + pix;
        *reinterpret_cast<float4*>(p.out + rec * 8 + 4 * lrow) = make_float4(o[ab][0], o[ab][1], o[ab][2], o[ab][3]);
      } else {
        const long c0 = (long)n * p.out_ctotal + p.out_coff + mb * 32 + 8 * g + 4 * lrow;
#pragma unroll
        for (int e = 0; e < 4; ++e) p.out[(c0 + e) * p.H * p.W + pix] = o[ab][e];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Shared-transform kernel (round 5, the default wherever Cout % 64 == 0): the same algorithm as conv_wino_kernel (3x3 layers: the same
// bits), re-cut so that the work a lone wave pays for with matrix-pipe time is shared.
// What the measurements say (profiles/r05_winograd.md): the fp32 matrix pipe runs on the fp32 vector lanes, and EVERY other
// instruction of a SIMD's waves — fp32 or integer VALU, moves, LDS and memory instructions — comes out of its time, with one wave
// per SIMD or with two. So the lever is the instruction count per MFMA: the one-wave kernel spends 128 transform adds and 32 loads per
// 64 MFMAs (2.5 per MFMA; 0.68 MFMA-busy), this kernel 32 + 24 per 32 (1.75).
// A 512-thread block (8 waves, two per SIMD, 128 + 128 registers) owns 64 output channels x 64 tiles x 16 positions:
//   * the input transform of a step (8 input channels x 64 tiles) is computed ONCE per block — a lane owns one patch COLUMN of
//     one tile and four channels: 4 pixel loads (16 B), row pass in registers, the column pass across the four lanes of a quad
//     as ONE v_fmac_f32 with a DPP quad_perm source per value (positions nu = 3 are stored negated on both sides for that),
//     4 ds_write_b128 of V — and every V is multiplied by 64 output channels instead of 32;
//   * wave (ph, mh, tg) multiplies positions xi in {2ph, 2ph+1} of channel half mh and tile half tg: V comes from LDS (one
//     ds_read_b128 per position and step = 4 k-steps), the transformed weights straight from global memory (one 16-byte load per
//     position and step; the two tile halves read the same bytes, L1 / the XCD's L2 serve them), both re-read right behind their
//     last use;
//   * two V slots in LDS (36.25 KB each), ONE barrier per step; stage s is loaded in step s-3, transformed in step s-2, read into
//     registers in step s-1 and multiplied in step s. The two waves of a SIMD run the same stream 16 MFMAs apart (in lockstep both
//     would leave the matrix pipe idle at the same time);
//   * output transform: s[xi][b] per wave, the halves swap 32 values per lane through LDS and each finishes 16 of the 32 channels
//     in the one-wave kernel's order of operations: the 3x3 layers are bit-identical to conv_wino_kernel;
//   * S2D (the 5x5 stride-2 layers): the stages of the four input phases are walked INTERLEAVED — two of phase 0, two of phase 1, … —
//     so that one loop body of eight steps has a compile-time phase per step and the identically-zero positions are dropped with no
//     branch at all (wave-uniform branches around MFMAs make the register allocator move accumulator tuples between the arms and
//     spill them). The channel sum runs in that order, so these layers agree with the one-wave kernel to rounding, not bit for bit.
// dev builds (-DW8_TRACE=1, tools/wino8_trace.py): every wave of ONE block (blockIdx.x == gridDim.x / 2) writes s_memtime stamps of its
// first tile — entry, first MFMA, loop end, exchange done, exit — to a device buffer: where a block's fixed cost goes
#ifndef W8_TRACE
#define W8_TRACE 0
#endif
#if W8_TRACE
__device__ unsigned long long* g_w8_trace = nullptr;
#define W8_STAMP(k_) if (g_w8_trace && blockIdx.x == gridDim.x / 2 && vb == W8_TRACE - 1 && (threadIdx.x & 63) == 0) g_w8_trace[wave * 8 + (k_)] = __builtin_amdgcn_s_memtime();   /* W8_TRACE - 1 = which tile of the persistent block */
#else
#define W8_STAMP(k_)
#endif
#ifndef W8_PRO_AFIRST
#define W8_PRO_AFIRST 0
#endif
#ifndef W8_ABL
#define W8_ABL 0   // dev ablations (wrong results): 1 no step barriers, 4 no pixel loads + transform + V stores, 8 no operand reads in the loop, 16 no output transform / stores, 32 no MFMAs
#endif
#define W8_QS 2320                       /* V: bytes per position: 2 tile halves x 2 k halves x (512 + 64 pad) + 16 */
#define W8_HS 576
#define W8_TGS 1152
#define W8_QSW 1168                      /* WIDE: one tile half */
#define W8_LDS_HALF_BYTES (2 * 16 * W8_QSW)   /* 37 376 B: two slots of one tile half (SHAPE 2; its output exchange takes 32 KB) */
#define W8_LDS_BYTES (2 * 16 * W8_QS)     /* 74 240 B: two V slots of 37 120 B — every offset of either slot fits the ds 16-bit immediate; the output exchange reuses the first 64 KB */

// PH: the wave's half (0 top: xi = 0, 1; 1 bottom: xi = 2, 3) as a COMPILE-TIME constant — the two halves run different streams, and
// with the choice behind run-time branches inside one body the register allocator cannot keep the accumulator tuples in place
// across the joins (hundreds of scratch spills); the kernel branches once, at its top, into two complete bodies.
// WIDE: the block is 128 output channels x 32 tiles instead of 64 x 64 (every V feeds 128 channels: half the transform work per MFMA,
// a lane of the transform owns TWO channels of a patch column; the weights of a step are read once per 32 tiles instead of per 64)
// SHAPE 2: the block is 64 channels x 32 tiles on FOUR waves (one per SIMD), two blocks per CU: each SIMD then runs a wave of either
// block, the blocks run independently — one block's prologue / output transform / stores under the other's MFMAs, and a last partial
// round of lone blocks runs faster instead of idling half the chip's issue slots.
// tile block bid -> (channel block mb2, tile block bx). bid % 8 names the XCD that works on it (persistent blocks: block b takes tile
// blocks bid = b mod 8 only)
__device__ __forceinline__ void w8_block_coords(const WinoParams& p, const int bid, int& mb2, int& bx) {
  const int xcd = bid & 7, idx = bid >> 3;
  if ((p.gy & 7) == 0) {          // XCD x owns the channel blocks [x·gy/8, (x+1)·gy/8): its slice of U stays in its L2
    const int per = p.gy >> 3;
    mb2 = xcd * per + idx % per;
    bx = idx / per;
  } else if ((8 % p.gy) == 0) {   // 8 / gy XCDs per channel block
    const int r = 8 / p.gy;
    mb2 = xcd % p.gy;
    bx = idx * r + xcd / p.gy;
  } else {
    mb2 = bid % p.gy;
    bx = bid / p.gy;
  }
}

// One piece of work: tile block `bid`, stages [kb, ke) of its K walk, results to the output (copy < 0: bias + activation applied) or as
// raw sums to copy `copy` of p.part. vb: the piece's number within its block (dev trace only).
template <int OUT_NC8, int S2D, int PH, int SHAPE>
__device__ __forceinline__ void wino8_body(const WinoParams& p, char* smem, const int wave, const int vb, const int bid, const int kb,
                                           const int ke, const int copy) {
  constexpr int WIDE = SHAPE == 1, HALF = SHAPE == 2;
  W8_STAMP(0)
  // opaque per tile: what a lane derives from its index is recomputed for every tile of a persistent block instead of being carried
  // (hoisted out of the tile loop it would live through the K loop and the output transform, i.e. in scratch)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  constexpr int ph = PH;
  constexpr int NSUB = WIDE ? 4 : 2;          // 32-channel sub-blocks of the block
  constexpr int TB = (WIDE || HALF) ? 32 : 64;          // tiles of the block
  constexpr int QS = (WIDE || HALF) ? W8_QSW : W8_QS;   // V bytes per position
  constexpr int VSLOT = 16 * QS;
  using tvec = typename std::conditional<WIDE != 0, f32x2, f32x4>::type;   // a transform lane's channels
  const int msub = WIDE ? (wave & 3) : (wave & 1), tg = (WIDE || HALF) ? 0 : (wave >> 1) & 1;
  const int lrow = lane >> 5, lcol = lane & 31;
  int mb2, bx;
  w8_block_coords(p, bid, mb2, bx);
  if (bx >= p.gx) return;
  int tpi = p.TY * p.TX, TXp = p.TX;
  asm volatile("" : "+s"(tpi), "+s"(TXp));   // per tile: the divisions' reciprocals are not carried across the tile loop (in scratch)
  const int c8n = p.Cin >> 3;                               // 8-channel blocks of the input: strides
  const int hw32 = p.H * p.W * 32;

  // ---- transform role: lane = (tile Tl of the block's TB, channel group cg of the 8 — 2 groups of 4 / WIDE: 4 groups of 2 —, patch column j)
  const int j = lane & 3, cg = WIDE ? (lane >> 2) & 3 : (lane >> 2) & 1, Tl = WIDE ? wave * 4 + (lane >> 4) : wave * 8 + (lane >> 3);
  int voffT[4];
  {
    const int tT = bx * TB + Tl;
    const bool tv = tT < p.ntiles;
    const int n = tv ? tT / tpi : 0;
    const int tr = tv ? tT - n * tpi : 0;
    const int ty = tr / TXp, tx = tr - ty * TXp;
    const int x = 2 * tx - 1 + j;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = 2 * ty - 1 + i;
      const bool ok = tv && y >= 0 && y < p.H && x >= 0 && x < p.W;
      voffT[i] = ok ? (((n * c8n) * p.H + y) * p.W + x) * 32 + cg * (WIDE ? 8 : 16) : (int)0x80000000;
    }
  }
  const float sgn = j == 1 ? 1.f : -1.f;
  // V of a position: [tile half][k half h][tile][16 B = channels 4h .. 4h+3] (+ pads: the 8 / 16 lanes of a store group hit 32 banks)
  const unsigned vw = WIDE ? (unsigned)((cg >> 1) * W8_HS + (cg & 1) * 8 + Tl * 16 + j * QS) :
                      HALF ? (unsigned)(cg * W8_HS + Tl * 16 + j * QS)
                           : (unsigned)((Tl >> 5) * W8_TGS + cg * W8_HS + (Tl & 31) * 16 + j * QS);
  // ---- multiply role
  const unsigned rb = (unsigned)(ph * 8 * QS + tg * W8_TGS + lrow * W8_HS + lcol * 16);
  const int ra_g = lane * 16 + ph * 8192;                  // this lane's 16 bytes of position 8 ph + i at + i * 1024 ...
  const int ra_s0 = ((mb2 * NSUB + msub) * c8n) * 16384;    // ... of the 32-channel sub-block's 16 KB per 8-channel block
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wd, 0, (int)p.wd_bytes, 0x00020000);

  // zeroed by the matrix pipe itself (0 * 0 + the inline constant 0): a zero tuple built in vector registers is carried from tile to
  // tile of a persistent block through scratch, and its reload sits in front of the first loads' wait
  f32x16 acc[8];
  {
    const float zf = 0.f;
#pragma unroll
    // (s_nop: the v_mov that makes zf may be scheduled right in front of the first of these, and the hazard pass cannot see inside the asm —
    // a VALU write followed by an MFMA read of the register needs wait states; found in csrc/wino42.hip, where acc[0] started from garbage)
    for (int q = 0; q < 8; ++q) asm volatile("s_nop 4\n\tv_mfma_f32_32x32x2_f32 %0, %1, %1, 0" : "=a"(acc[q]) : "v"(zf));
  }
  tvec raw[4], T[4];
  f32x4 A[8], Bv[8];

  // S2D: the s2d tensor holds the input phases (py, px) as four runs of Cin/4 channels = npb 8-channel blocks each; positions xi = 3
  // (py) / nu = 3 (px) of a phase's blocks vanish. Stage s of the K walk is block W8_CB(s): phases interleaved two blocks at a time,
  // so that stage s has phase (s >> 1) & 3 — a compile-time constant at every place of the eight-step loop body.
  const int npb = S2D ? (c8n >> 2) : 0;
#define W8_CB(s) (S2D ? (((s) >> 1) & 3) * npb + (((s) >> 3) << 1) + ((s) & 1) : (s))
// position i of this wave in a stage of phase P (compile-time)
#define W8_ACT(i, P) (!S2D || !((((P) & 1) && ((i) & 3) == 3) || (((P) >> 1) && ph && (i) >= 4)))
#define W8_LDS4(off) (*reinterpret_cast<f32x4*>(smem + (off)))
// the 4 pixel loads of this lane's patch column for stage `stage` of phase P. S2D: patch row 3 feeds only xi = 3 — where the phase drops
// those positions neither the load, nor that row of the row pass, nor its column pass and store are issued (compile-time).
// (Column 3 feeds only nu = 3 as well; masking those lanes would cost 4 vector ORs per step, more than their L1 hits.)
#define W8_PIX(stage, P)                                                                              \
  if (!(W8_ABL & 4)) {                                                                                \
    const int st_ = min((stage), ke - 1);                                                             \
    const int so_ = __builtin_amdgcn_readfirstlane(W8_CB(st_) * hw32);                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                  \
      if (i_ < 3 || !(S2D && ((P) >> 1))) {                                                           \
        if constexpr (WIDE) raw[i_] = __builtin_bit_cast(tvec, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voffT[i_], so_, 0)); \
        else raw[i_] = __builtin_bit_cast(tvec, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffT[i_], so_, 0)); \
      }                                                                                               \
  }
#define W8_ROW(P)                                                                                     \
  {                                                                                                   \
    T[0] = raw[0] - raw[2]; T[1] = raw[1] + raw[2]; T[2] = raw[2] - raw[1];                           \
    if (!(S2D && ((P) >> 1))) T[3] = raw[1] - raw[3];                                                 \
  }
// column pass of row xi across the quad, IN PLACE: lane j holds t_j and needs (t0 - t2, t1 + t2, t2 - t1, t3 - t1)[j] (nu = 3 negated, as
// packed) = self + sgn * T[lane (2, 2, 1, 1)[j]], sgn = (-1, +1, -1, -1): one v_fmac_f32 with a DPP quad_perm source per value.
// s_nop 1: a DPP source written by the VALU instruction right before needs two wait states, and the hazard pass cannot see inside the
// asm — so the wait states and the DPP reads they protect are ONE asm statement (nothing can be scheduled between them)
#define W8_DPPQ "quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf"
#define W8_COL(xi, slot_)                                                                             \
  {                                                                                                   \
    if constexpr (WIDE) {                                                                             \
      float c0_ = T[xi].x, c1_ = T[xi].y;                                                             \
      asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 " W8_DPPQ "\n\tv_fmac_f32_dpp %1, %1, %2 " W8_DPPQ \
                   : "+v"(c0_), "+v"(c1_) : "v"(sgn));                                                \
      f32x2 v_; v_.x = c0_; v_.y = c1_;                                                               \
      *reinterpret_cast<f32x2*>(smem + (vw + (unsigned)((slot_) * VSLOT + (xi) * 4 * QS))) = v_;      \
    } else {                                                                                          \
      float c0_ = T[xi][0], c1_ = T[xi][1], c2_ = T[xi][2 % (WIDE ? 2 : 4)], c3_ = T[xi][3 % (WIDE ? 2 : 4)]; \
      asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %4 " W8_DPPQ "\n\tv_fmac_f32_dpp %1, %1, %4 " W8_DPPQ \
                   "\n\tv_fmac_f32_dpp %2, %2, %4 " W8_DPPQ "\n\tv_fmac_f32_dpp %3, %3, %4 " W8_DPPQ    \
                   : "+v"(c0_), "+v"(c1_), "+v"(c2_), "+v"(c3_) : "v"(sgn));                          \
      f32x4 v_; v_.x = c0_; v_.y = c1_; v_.z = c2_; v_.w = c3_;                                       \
      W8_LDS4(vw + (unsigned)((slot_) * VSLOT + (xi) * 4 * QS)) = v_;                                 \
    }                                                                                                 \
  }
// operands of position i for the stage whose weights sit at scalar offset rsa_: the weights straight from global memory (L1 / L2:
// the two tile halves of a channel half read the same bytes), V from LDS slot slot_
#define W8_RDA(i) A[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrw, ra_g + (i) * 1024, rsa_, 0));
#define W8_RDB(i, slot_) Bv[i] = W8_LDS4(rb + (unsigned)((slot_) * VSLOT + (i) * QS));
#define W8_RD(i, slot_) { W8_RDA(i) W8_RDB(i, slot_) }
#define W8_MFMA(i, s_)                                                                                \
  acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32((s_) == 0 ? A[i].x : (s_) == 1 ? A[i].y : (s_) == 2 ? A[i].z : A[i].w, \
                                                (s_) == 0 ? Bv[i].x : (s_) == 1 ? Bv[i].y : (s_) == 2 ? Bv[i].z : Bv[i].w, acc[i], 0, 0, 0); \
  asm volatile("" : "+a"(acc[i]));
// a wave's own LDS traffic is all a barrier has to wait for: the global loads are tracked by the compiler where they are used
#define W8_SYNC()                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
  __builtin_amdgcn_s_barrier();                                                                       \
  asm volatile("" ::: "memory");
// MFMA blocks: two positions interleaved (i0,s0) (i1,s0) (i0,s1) ... or one position's four k-steps; X(sl) = the slot's share of the
// other roles. P: the stage's phase — a block whose positions the phase drops shrinks or vanishes at compile time.
#define W8_M2X(i0, i1, P, X)                                                                          \
  _Pragma("unroll") for (int sl = 0; sl < 8; ++sl) {                                                  \
    if (!(W8_ABL & 32)) {                                                                             \
      if (sl & 1) { if (W8_ACT(i1, P)) { W8_MFMA(i1, sl >> 1) } } else { if (W8_ACT(i0, P)) { W8_MFMA(i0, sl >> 1) } } \
    }                                                                                                 \
    X(sl) __builtin_amdgcn_sched_barrier(0);                                                          \
  }
#define W8_M1X(i0, P, X)                                                                              \
  _Pragma("unroll") for (int sl = 0; sl < 8; sl += 2) {                                               \
    if (!(W8_ABL & 32) && W8_ACT(i0, P)) { W8_MFMA(i0, sl >> 1) }                                     \
    X(sl) __builtin_amdgcn_sched_barrier(0);                                                          \
  }
// the operand reads of positions for the next stage (phase PN: what it drops is not read)
#define W8_R2(i0, i1, PN, SLR)                                                                        \
    if (!(W8_ABL & 8)) { if (W8_ACT(i0, PN)) { W8_RD(i0, SLR) } if (W8_ACT(i1, PN)) { W8_RD(i1, SLR) } } \
    __builtin_amdgcn_sched_barrier(0);
#define W8_R1(i0, PN, SLR)                                                                            \
    if (!(W8_ABL & 8) && W8_ACT(i0, PN)) { W8_RD(i0, SLR) }                                           \
    __builtin_amdgcn_sched_barrier(0);
#define W8_XNONE(sl)
// slots with the transform of stage xst_ into slot xsl_ and, behind the column pass (the loads' 16 registers are the row pass's), the
// pixel loads of stage xst_ + 1 (phase xpp_); xtp_ = the phase of stage xst_
#define W8_XWORK(sl)                                                                                  \
      if (sl == 0 && !(W8_ABL & 4)) { W8_ROW(xtp_) }                                                  \
      if (sl >= 1 && sl <= 4 && !(W8_ABL & 4) && !(S2D && sl == 4 && (xtp_ >> 1))) { W8_COL(sl - 1, xsl_) } \
      if (sl == 5) { W8_PIX(xst_ + 1, xpp_) }

  // ---- prologue: stage 0 into slot 0 and into the operand registers; the top wave also transforms stage 1 into slot 1 and loads the
  // pixels of stage 2 (the bottom wave does both in its first step). S2D: stages 0, 1 have phase 0, stages 2, 3 phase 1
#if W8_PRO_AFIRST
  {   // dev: the first stage's weight operands requested before anything else
    const int rsa_ = __builtin_amdgcn_readfirstlane(ra_s0 + W8_CB(kb) * 16384);
#pragma unroll
    for (int i = 0; i < 8; ++i) W8_RDA(i)
  }
  __builtin_amdgcn_sched_barrier(0);
#endif
  W8_PIX(kb, 0)
  W8_ROW(0)
  W8_PIX(kb + 1, 0)
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) W8_COL(xi, 0)
  W8_SYNC()
  W8_STAMP(5)
  {
    const int rsa_ = __builtin_amdgcn_readfirstlane(ra_s0 + W8_CB(kb) * 16384);
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (!W8_PRO_AFIRST) { W8_RDA(i) } W8_RDB(i, 0) }
  }
  if (!ph) {
    W8_ROW(0)
    W8_PIX(kb + 2, 1)
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) W8_COL(xi, 1)
  }
  __builtin_amdgcn_sched_barrier(0);

  W8_STAMP(1)
  // ---- one step per stage k (SL = k & 1), ONE barrier each: behind barrier k every wave reads stage k + 1 into its operand
  // registers (V from slot SL ^ 1); between barriers k and k + 1 it transforms its share of stage k + 2 (pixels loaded a step earlier)
  // into slot SL and loads its pixels of stage k + 3. The two waves of a SIMD run 16 MFMAs apart:
  //   top    (ph 0): M(0,1) | barrier k | R(0,1) | M(2,4) + transform k+2 | R(2,4) | M(5,6) | R(5,6) | M(3,7) | R(3,7)
  //   bottom (ph 1): M(0,1) + transform k+1 | M(3) | barrier k | R(0,1) R(3) | M(2) | R(2) | M(4,5) | R(4,5) | M(6,7) | R(6,7)
  // (M = the MFMAs of the listed positions i = x*4 + nu of stage k, R = their operand reads for stage k + 1). The blocks a phase can
  // drop — nu = 3 in the top wave; 3, row xi = 3 = {4..7} in the bottom wave — carry none of the other roles.
  // P0..P3: the phases of stages k .. k + 3 (S2D; compile-time)
#define W8_STEP(k, SL, P0, P1, P2, P3)                                                                \
  {                                                                                                   \
    const int rsa_ = __builtin_amdgcn_readfirstlane(ra_s0 + W8_CB(min((k) + 1, ke - 1)) * 16384);   \
    if (!ph) {                                                                                        \
      const int xst_ = (k) + 2; constexpr int xsl_ = (SL), xtp_ = (P2), xpp_ = (P3);                               \
      W8_M2X(0, 1, P0, W8_XNONE)                                                                      \
      if (!(W8_ABL & 1)) { W8_SYNC() }                                                                \
      W8_R2(0, 1, P1, (SL) ^ 1)                                                                       \
      W8_M2X(2, 4, P0, W8_XWORK)                                                                      \
      W8_R2(2, 4, P1, (SL) ^ 1)                                                                       \
      W8_M2X(5, 6, P0, W8_XNONE)                                                                      \
      W8_R2(5, 6, P1, (SL) ^ 1)                                                                       \
      W8_M2X(3, 7, P0, W8_XNONE)                                                                      \
      W8_R2(3, 7, P1, (SL) ^ 1)                                                                       \
    } else {                                                                                          \
      const int xst_ = (k) + 1; constexpr int xsl_ = (SL) ^ 1, xtp_ = (P1), xpp_ = (P2);                           \
      W8_M2X(0, 1, P0, W8_XWORK)                                                                      \
      W8_M1X(3, P0, W8_XNONE)                                                                         \
      if (!(W8_ABL & 1)) { W8_SYNC() }                                                                \
      W8_R2(0, 1, P1, (SL) ^ 1)                                                                       \
      W8_R1(3, P1, (SL) ^ 1)                                                                          \
      W8_M1X(2, P0, W8_XNONE)                                                                         \
      W8_R1(2, P1, (SL) ^ 1)                                                                          \
      W8_M2X(4, 5, P0, W8_XNONE)                                                                      \
      W8_R2(4, 5, P1, (SL) ^ 1)                                                                       \
      W8_M2X(6, 7, P0, W8_XNONE)                                                                      \
      W8_R2(6, 7, P1, (SL) ^ 1)                                                                       \
    }                                                                                                 \
  }
  if (!S2D) {
    int k = kb;   // kb is even (the launcher's slices are): SL = k & 1
    for (; k + 2 <= ke; k += 2) {
      W8_STEP(k, 0, 0, 0, 0, 0)
      W8_STEP(k + 1, 1, 0, 0, 0, 0)
    }
    if (k < ke) W8_STEP(k, 0, 0, 0, 0, 0)
  } else {   // kb % 8 == 0 and (ke - kb) % 8 == 0 (the launcher checks): phases 0 0 1 1 2 2 3 3 | 0 0 ...
    for (int k = kb; k < ke; k += 8) {
      W8_STEP(k, 0, 0, 0, 1, 1)
      W8_STEP(k + 1, 1, 0, 1, 1, 2)
      W8_STEP(k + 2, 0, 1, 1, 2, 2)
      W8_STEP(k + 3, 1, 1, 2, 2, 3)
      W8_STEP(k + 4, 0, 2, 2, 3, 3)
      W8_STEP(k + 5, 1, 2, 3, 3, 0)
      W8_STEP(k + 6, 0, 3, 3, 0, 0)
      W8_STEP(k + 7, 1, 3, 0, 0, 1)
    }
  }
#undef W8_STEP
#undef W8_XWORK
#undef W8_XNONE
#undef W8_R1
#undef W8_R2
#undef W8_M1X
#undef W8_M2X
#undef W8_MFMA
#undef W8_RD
#undef W8_RDA
#undef W8_RDB
#undef W8_COL
#undef W8_DPPQ
#undef W8_ROW
#undef W8_PIX
#undef W8_ACT
#undef W8_CB

  W8_STAMP(2)
  // ---- output transform. acc[x*4 + nu][r] with xi = 2ph + x: channel (r&3) + 8(r>>2) + 4·lrow of the wave's 32, tile lcol.
  // s[xi][0] = (m0 + m1) + m2, s[xi][1] = (m1 - m2) - m3; Y[0][b] = ((s0 + s1) + s2) + bias, Y[1][b] = ((s1 - s2) - s3) + bias.
  // The top wave (s0, s1) finishes accumulator rows 0-7, the bottom wave (s2, s3) rows 8-15; each hands the other its s of the
  // rows it does not finish: 32 floats per lane through LDS, [wave][value][lane].
  W8_SYNC()   // every wave is done with both slots, every DMA piece has landed
  if (W8_ABL & 16) { if (acc[0][0] == 123.f && acc[7][15] == 4.f) p.out[0] = acc[3][2]; return; }
  // opaque copies: everything the epilogue derives from the block index is computed HERE (hoisted above the K loop it would sit in
  // registers the loop needs, i.e. in scratch)
  int bxe = __builtin_amdgcn_readfirstlane(bx), mb2e = __builtin_amdgcn_readfirstlane(mb2), tpie = __builtin_amdgcn_readfirstlane(tpi),
      TXe = __builtin_amdgcn_readfirstlane(p.TX);   // (bx, mb2 come out of a VALU division: uniform, but in vector registers)
  asm volatile("" : "+s"(bxe), "+s"(mb2e), "+s"(tpie), "+s"(TXe));
  const int lanee = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int lrow_e = lanee >> 5, lcol_e = lanee & 31;
  const int t = bxe * TB + tg * 32 + lcol_e;
  const bool tvalid = t < p.ntiles;
  const int n = tvalid ? t / tpie : 0;
  const int trm = tvalid ? t - n * tpie : 0;
  const int ty = trm / TXe, tx = trm - ty * TXe;
  const int y0 = 2 * ty, x0 = 2 * tx;
  const bool y1ok = y0 + 1 < p.H, x1ok = x0 + 1 < p.W;
  const int mb = mb2e * NSUB + msub;
  // a piece of a split K walk writes raw sums to its own copy of the output; bias and activation wait for the second pass
  float* const outp = copy >= 0 ? p.part + (long)copy * p.part_stride : p.out;
  const float* const biasp = copy >= 0 ? nullptr : p.bias;
  float slope_e = copy >= 0 ? 1.f : p.slope;
  const bool through = copy >= 0 && p.sk_count != nullptr;   // a piece some block reads back inside this launch: straight to memory
  // channel-blocked output: 32-byte records addressed as the lane's base (its tile's first pixel in channel block mb*4) + a wave-uniform
  // scalar offset per (channel block, pixel) — no 64-bit address arithmetic per store (it was a third of this epilogue's vector work)
  const __amdgpu_buffer_rsrc_t rsrp = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, OUT_NC8 ? (int)p.out_bytes : 0, 0x00020000);
  const int C8o = p.Cout >> 3, tpo = p.TY * TXe;
  const int gstep = p.out_s2d ? tpo : p.H * p.W;                    // records from one channel block to the next
  const int ab1 = p.out_s2d ? C8o * tpo : 1, ab2 = p.out_s2d ? 2 * C8o * tpo : p.W;   // ... to the pixel to the right / below
  const int vo0 = ((p.out_s2d ? (n * 4 * C8o + mb * 4) * tpo + ty * TXe + tx : (n * C8o + mb * 4) * (p.H * p.W) + y0 * p.W + x0) * 32) + 16 * lrow_e;
  asm volatile("" : "+v"(slope_e));
  float* xw = reinterpret_cast<float*>(smem) + wave * 2048 + lanee;
  const float* xr = reinterpret_cast<const float*>(smem) + (wave ^ (HALF ? 2 : 4)) * 2048 + lanee;
  // whole-vector forms: reading single elements of an AGPR-resident f32x16 makes the compiler copy all 16 registers each time
  f32x16 sx[2][2];   // [x][b], all 16 rows
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const f32x16 t_ = acc[x * 4 + 0] + acc[x * 4 + 1];
    sx[x][0] = t_ + acc[x * 4 + 2];
    const f32x16 u_ = acc[x * 4 + 1] - acc[x * 4 + 2];
    sx[x][1] = u_ - acc[x * 4 + 3];
    __builtin_amdgcn_sched_barrier(0);
  }
#define W8_SEND(PHC)                                                                                  \
  {                                                                                                   \
    _Pragma("unroll") for (int rr = 0; rr < 8; ++rr) {                                                \
      const int r = (PHC) ? rr : rr + 8;   /* the rows the OTHER half finishes */                     \
      xw[(rr * 4 + 0) * 64] = sx[0][0][r]; xw[(rr * 4 + 1) * 64] = sx[0][1][r];                       \
      xw[(rr * 4 + 2) * 64] = sx[1][0][r]; xw[(rr * 4 + 3) * 64] = sx[1][1][r];                       \
    }                                                                                                 \
  }
#define W8_FINISH(PHC)                                                                                \
  {                                                                                                   \
      float4 bq[2];                                                                                   \
      _Pragma("unroll") for (int gg = 0; gg < 2; ++gg)                                                \
        bq[gg] = biasp ? *reinterpret_cast<const float4*>(biasp + mb * 32 + 8 * ((PHC) * 2 + gg) + 4 * lrow_e) : make_float4(0.f, 0.f, 0.f, 0.f); \
      _Pragma("unroll") for (int gg = 0; gg < 2; ++gg) {                                              \
        const int g = (PHC) * 2 + gg;                                                                 \
        float o[4][4];                                                                                \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                               \
          const int r = 4 * g + e, rr = r - (PHC) * 8;                                                \
          const float bv = e == 0 ? bq[gg].x : e == 1 ? bq[gg].y : e == 2 ? bq[gg].z : bq[gg].w;      \
          _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                             \
            const float mine0 = sx[0][b][r], mine1 = sx[1][b][r];                                     \
            const float oth0 = xr[(rr * 4 + b) * 64], oth1 = xr[(rr * 4 + 2 + b) * 64];               \
            const float s0 = (PHC) ? oth0 : mine0, s1 = (PHC) ? oth1 : mine1;                         \
            const float s2 = (PHC) ? mine0 : oth0, s3 = (PHC) ? mine1 : oth1;                         \
            float v0 = ((s0 + s1) + s2) + bv;                                                         \
            float v1 = ((s1 - s2) - s3) + bv;                                                         \
            o[0 * 2 + b][e] = v0 > 0.f ? v0 : v0 * slope_e;                                           \
            o[1 * 2 + b][e] = v1 > 0.f ? v1 : v1 * slope_e;                                           \
          }                                                                                           \
        }                                                                                             \
        _Pragma("unroll") for (int ab = 0; ab < 4; ++ab) {                                            \
          const int a = ab >> 1, b = ab & 1;                                                          \
          if ((a && !y1ok) || (b && !x1ok)) continue;                                                 \
          if (OUT_NC8) {   /* record (channel block g, pixel ab) of this lane's tile: the lane's base + a wave-uniform offset */ \
            const int so_ = (g * gstep + (ab == 0 ? 0 : ab == 1 ? ab1 : ab == 2 ? ab2 : ab1 + ab2)) * 32; \
            f32x4 ov; ov.x = o[ab][0]; ov.y = o[ab][1]; ov.z = o[ab][2]; ov.w = o[ab][3];              \
            const auto od = __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, ov); \
            if (through) __builtin_amdgcn_raw_buffer_store_b128(od, rsrp, vo0, so_, 17);   /* a stream-K piece: straight to memory (sc0 sc1), whichever XCD's block reads it back */ \
            else __builtin_amdgcn_raw_buffer_store_b128(od, rsrp, vo0, so_, 0);                       \
          } else {                                                                                    \
            const long pix = (long)(y0 + a) * p.W + x0 + b;                                           \
            const long c0 = (long)n * p.out_ctotal + p.out_coff + mb * 32 + 8 * g + 4 * lrow_e;         \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) outp[(c0 + e) * p.H * p.W + pix] = o[ab][e]; \
          }                                                                                           \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
      }                                                                                               \
  }
  if (ph == 0) W8_SEND(0) else W8_SEND(1)
  W8_STAMP(6)
  W8_SYNC()
  W8_STAMP(3)
  if (!tvalid) return;
  if (ph == 0) W8_FINISH(0) else W8_FINISH(1)
  W8_STAMP(4)
#undef W8_SEND
#undef W8_FINISH
#undef W8_SYNC
#undef W8_LDS4
}

// stream-K: which block's run holds granule u of the XCD's last-round granules (runs of sk_q + 1 for the first sk_rem blocks, sk_q after)
__device__ __forceinline__ int w8_run_owner(const WinoParams& p, const int u) {
  const int big = p.sk_rem * (p.sk_q + 1);
  return u < big ? u / (p.sk_q + 1) : p.sk_rem + (u - big) / p.sk_q;
}

// stream-K: this block has just written one of the `npieces` raw copies of tile block `bid` (TB tiles x CB channels, channel-blocked
// output). Count it; the block whose piece arrives last reads the copies back (same XCD, same L2: the runs of a cut tile block belong
// to neighbouring blocks of one XCD) and writes out = act((copy 0 + copy 1 + ...) + bias) — the same sum whoever comes last.
__device__ __forceinline__ void w8_finish_cut_tile_block(const WinoParams& p, char* smem, const int bid, const int npieces, const int TB,
                                                         const int CB) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's raw sums have reached memory (written through: sc0 sc1) ...
  __syncthreads();                                       // ... every thread's have
  if (threadIdx.x == 0) {
    const int old = atomicAdd(p.sk_count + bid, 1);
    if (old == npieces - 1) atomicExch(p.sk_count + bid, 0);   // the last piece leaves the counter as it found it before the launch
    *reinterpret_cast<volatile int*>(smem) = old;
  }
  __syncthreads();
  const int old = *reinterpret_cast<volatile int*>(smem);
  __syncthreads();                                       // (the word is free again for the next piece's V)
  if (old != npieces - 1) return;
  int mb2, bx;
  w8_block_coords(p, bid, mb2, bx);
  const int tpi = p.TY * p.TX, c4n = CB >> 2;
  if (p.fin_nchw) {            // dense NCHW partials (Cout channels each) -> channels [fin_coff, +Cout) of the fin_ctotal-channel output
    const long hw = (long)p.H * p.W;
    for (int i = threadIdx.x; i < TB * CB * 4; i += blockDim.x) {   // (channel, tile, pixel of its 2x2): pixels fastest
      const int ab = i & 3, tl = (i >> 2) % TB, cl = (i >> 2) / TB;
      const int t = bx * TB + tl;
      if (t >= p.ntiles) continue;
      const int n = t / tpi, tr = t - n * tpi;
      const int ty = tr / p.TX, tx = tr - ty * p.TX;
      const int y = 2 * ty + (ab >> 1), x = 2 * tx + (ab & 1);
      if (y >= p.H || x >= p.W) continue;
      const int c = mb2 * CB + cl;
      const long src = ((long)n * p.Cout + c) * hw + (long)y * p.W + x;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.part, 0, (int)(p.part_stride * 4), 0x00020000);
      float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(src * 4), 0, 17));
      for (int s_ = 1; s_ < npieces; ++s_) {
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.part + (long)s_ * p.part_stride), 0, (int)(p.part_stride * 4), 0x00020000);
        v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_, (int)(src * 4), 0, 17));
      }
      v += p.bias ? p.bias[c] : 0.f;
      p.out[((long)n * p.fin_ctotal + p.fin_coff + c) * hw + (long)y * p.W + x] = v > 0.f ? v : v * p.slope;
    }
    return;
  }
  for (int i = threadIdx.x; i < TB * CB; i += blockDim.x) {     // (tile, pixel of its 2x2, 4 channels)
    const int c4 = i % c4n, ab = (i / c4n) & 3, tl = i / CB;
    const int t = bx * TB + tl;
    if (t >= p.ntiles) continue;
    const int n = t / tpi, tr = t - n * tpi;
    const int ty = tr / p.TX, tx = tr - ty * p.TX;
    const int y = 2 * ty + (ab >> 1), x = 2 * tx + (ab & 1);
    if (y >= p.H || x >= p.W) continue;
    const int c = mb2 * CB + c4 * 4;
    long rec;
    if (p.out_s2d) rec = (((long)n * 4 + ab) * (p.Cout >> 3) + (c >> 3)) * tpi + ty * p.TX + tx;
    else rec = ((long)n * (p.Cout >> 3) + (c >> 3)) * p.H * p.W + (long)y * p.W + x;
    const long o4 = rec * 2 + ((c >> 2) & 1);
    // read from memory, past both cache levels (sc0 sc1): the pieces may have been written from another XCD
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.part, 0, (int)(p.part_stride * 4), 0x00020000);
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(o4 * 16), 0, 17));
    for (int s_ = 1; s_ < npieces; ++s_) {
      const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.part + (long)s_ * p.part_stride), 0, (int)(p.part_stride * 4), 0x00020000);
      const f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)(o4 * 16), 0, 17));
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope;
    v.z = v.z > 0.f ? v.z : v.z * p.slope; v.w = v.w > 0.f ? v.w : v.w * p.slope;
    reinterpret_cast<float4*>(p.out)[o4] = make_float4(v.x, v.y, v.z, v.w);
  }
}

// Persistent blocks: the grid is one block per resident slot (256 of 8 waves, 512 of 4) and block b works on the virtual blocks b,
// b + gridDim.x, … (gridDim.x % 8 == 0 keeps a virtual block on the XCD its index names), or — stream-K — on its column of whole
// tile blocks and then its run of the last round's granules. What it buys is measured in profiles/r05_winograd.md: a block's fixed
// cost is ~10 us, and part of it is the hardware's own turn-around between two workgroups; stream-K shortens the last, partly filled
// round to its share of the work (9.375 rounds of tile blocks cost 10 without it on conv3 / conv3_1 at B = 32).
// Every value that names a piece is wave-uniform and lives in scalar registers (readfirstlane: the divisions run on the VALU).
#define W8_PERSIST(BODY, TB_, CB_)                                                                    \
  {                                                                                                   \
    int u = 0, uend = 0, xcd = 0, lb = 0;                                                             \
    if (p.sk_G > 0) {                                                                                 \
      lb = blockIdx.x >> 3;                                                                           \
      xcd = blockIdx.x & 7;                                                                           \
      u = lb * p.sk_q + min(lb, p.sk_rem);                                                            \
      uend = u + p.sk_q + (lb < p.sk_rem ? 1 : 0);                                                    \
    }                                                                                                 \
    for (int vb = 0;; ++vb) {                                                                         \
      int bid, kb, ke, copy, npieces = 0;                                                             \
      if (p.sk_G > 0) {                                                                               \
        if (vb < p.sk_F) {            /* the whole rounds: tile block vb of this block's column */      \
          bid = (vb * p.sk_nlb + lb) * 8 + xcd; kb = 0; ke = p.Cin >> 3; copy = -1;                   \
        } else {                                                                                      \
          if (u >= uend) break;                                                                       \
          const int lt = __builtin_amdgcn_readfirstlane(u / p.sk_G), g0 = u - lt * p.sk_G, g1 = min(p.sk_G, g0 + uend - u); \
          const int o0 = __builtin_amdgcn_readfirstlane(w8_run_owner(p, lt * p.sk_G)), o1 = __builtin_amdgcn_readfirstlane(w8_run_owner(p, lt * p.sk_G + p.sk_G - 1)); \
          bid = (p.sk_F * p.sk_nlb + lt) * 8 + xcd; kb = g0 * p.sk_gran; ke = g1 * p.sk_gran;         \
          npieces = o1 - o0 + 1;                                                                      \
          copy = npieces == 1 ? -1 : lb - o0;                                                         \
          if (npieces == 1) npieces = 0;                                                              \
          u += g1 - g0;                                                                               \
        }                                                                                             \
      } else {                                                                                        \
        const int v = blockIdx.x + vb * gridDim.x;                                                    \
        if (v >= p.nvb) break;                                                                        \
        const int slice = __builtin_amdgcn_readfirstlane(v / p.grid0);                                \
        bid = v - slice * p.grid0; kb = slice * p.kslice; ke = min(p.Cin >> 3, kb + p.kslice);        \
        copy = p.ksplit > 1 ? slice : -1;                                                             \
        if (p.ksplit > 1 && p.sk_count) npieces = p.ksplit;   /* K slices: the slice that arrives last sums them (no second pass) */ \
      }                                                                                               \
      BODY;                                                                                           \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* the output exchange has been read: the slots are free for the next piece */ \
      __builtin_amdgcn_s_barrier();                                                                   \
      asm volatile("" ::: "memory");                                                                  \
      if (npieces > 0) w8_finish_cut_tile_block(p, smem, bid, npieces, TB_, CB_);                     \
    }                                                                                                 \
  }
template <int OUT_NC8, int S2D, int SHAPE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino8_kernel(WinoParams p) {
  __shared__ __attribute__((aligned(16))) char smem[W8_LDS_BYTES];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // waves w and w + 4 share a SIMD: a top and a bottom half each
  if (((wave >> 2) & 1) == 0) { W8_PERSIST((wino8_body<OUT_NC8, S2D, 0, SHAPE>(p, smem, wave, vb, bid, kb, ke, copy)), (SHAPE == 0 ? 64 : 32), (SHAPE == 1 ? 128 : 64)) }
  else { W8_PERSIST((wino8_body<OUT_NC8, S2D, 1, SHAPE>(p, smem, wave, vb, bid, kb, ke, copy)), (SHAPE == 0 ? 64 : 32), (SHAPE == 1 ? 128 : 64)) }
}

// SHAPE 2: four waves, two blocks per CU (the register file holds two waves of 256 registers per SIMD: one of each block)
template <int OUT_NC8, int S2D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino4_kernel(WinoParams p) {
  __shared__ __attribute__((aligned(16))) char smem[W8_LDS_HALF_BYTES];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (((wave >> 1) & 1) == 0) { W8_PERSIST((wino8_body<OUT_NC8, S2D, 0, 2>(p, smem, wave, vb, bid, kb, ke, copy)), 32, 64) }
  else { W8_PERSIST((wino8_body<OUT_NC8, S2D, 1, 2>(p, smem, wave, vb, bid, kb, ke, copy)), 32, 64) }
}
#undef W8_PERSIST

// Second pass of a split K loop of conv_wino8_kernel: sums the S raw copies (same layout as the output), adds the bias, applies the
// LeakyReLU. nc8: channel-blocked output (float4 = 4 consecutive channels; space-to-depth order keeps the channel of a record:
// C8 = 4 phases x C8mod blocks); else NCHW into channels [coff, coff + Cout) of a ctotal-channel tensor (partials dense).
__global__ __launch_bounds__(256) void wino_reduce_kernel(float* __restrict__ out, const float* __restrict__ partial, const float* __restrict__ bias,
                                                          long total, long stride, int S, float slope, int nc8, int C8, int hw, int C8mod,
                                                          int Cout, int ctotal, int coff) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  if (nc8) {
    float4 v = reinterpret_cast<const float4*>(partial)[i];
    for (int s = 1; s < S; ++s) {
      const float4 w = reinterpret_cast<const float4*>(partial + (long)s * stride)[i];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    const int c = (int)(((i / (2L * hw)) % C8) % C8mod) * 8 + (int)(i & 1) * 4;
    const float4 b = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0, 0, 0, 0);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
    reinterpret_cast<float4*>(out)[i] = v;
  } else {
    float v = partial[i];
    for (int s = 1; s < S; ++s) v += partial[(long)s * stride + i];
    const int r = (int)(i % hw), c = (int)((i / hw) % Cout);
    const long n = i / ((long)hw * Cout);
    v += bias ? bias[c] : 0.f;
    out[(n * ctotal + coff + c) * hw + r] = v > 0.f ? v : v * slope;
  }
}

// K-split plan of conv_wino8_kernel for a grid of `blocks` full-K blocks of nK steps on the chip's 256 CUs (one block per CU at a
// time): S slices of ks steps each so that blocks x S fills whole rounds; cost model = rounds x (steps + per-block prologue/epilogue,
// ~6 steps' worth) + the second pass (S + 1 passes over the output at ~4 TB/s, in steps of ~1.7 us). Deterministic: a function of
// the geometry only. step_granule: 2 (slot parity) or 8 (the stride-2 form's loop body).
static int wino8_split_plan(long blocks, int nK, int step_granule, double out_mb, int max_split, int* kslice, int slots = 256,
                            double* cost_out = nullptr) {
  int best = 1;
  double best_cost = 1e30;
  for (int S = 1; S <= 16; ++S) {
    if (max_split > 0 && S > max_split) break;
    int ks = di_div_up(di_div_up(nK, S), step_granule) * step_granule;
    if (S > 1 && ks < 8) break;
    const int Seff = di_div_up(nK, ks);
    if (Seff != S) continue;
    const double rounds = (double)di_div_up(blocks * S, slots);
    double cost = rounds * (ks + 6.0);
    if (S > 1) cost += (S + 1) * out_mb / 4000.0 / 1.7e-3 + 1.5;   // MB / (MB per ms) -> ms -> steps; + a launch boundary
    if (cost < best_cost - 1e-9) { best_cost = cost; best = S; *kslice = ks; }
  }
  if (best == 1) *kslice = nK;
  if (cost_out) *cost_out = best_cost;
  return best;
}

// Stream-K of the last round of the same grid (WinoParams::sk_*): F whole tile blocks per persistent block, then every block an equal
// run (+- 1) of the remaining tile blocks' granules — at the price of up to two more pieces per block and the read-back of the cut tile
// blocks. Same cost unit as wino8_split_plan (steps). Returns the granules per tile block, 0 where it does not apply: the XCD deal
// needs grid % 8 == 0, and no tile block is cut into more than W8_SK_MAX_COPIES pieces.
#define W8_SK_MAX_COPIES 8
#define W8_SK_MAX_TILE_BLOCKS DI_WINO_COUNTERS   /* counters per context (64 KB) */
static int wino8_streamk_plan(long grid, int nK, int step_granule, int slots, double* cost, int* F, int* q, int* rem) {
  if (grid % 8 != 0 || nK % step_granule != 0 || grid > W8_SK_MAX_TILE_BLOCKS) return 0;
  const int G = nK / step_granule, nlb = slots / 8;
  const long ltiles = grid / 8;
  *F = (int)(ltiles / nlb);
  const long units = (ltiles - (long)*F * nlb) * G;
  if (units == 0) return 0;
  *q = (int)(units / nlb); *rem = (int)(units % nlb);
  if (*q < 1 || di_div_up(G, *q) + 1 > W8_SK_MAX_COPIES) return 0;
  *cost = (double)*F * (nK + 6.0) + ((double)*q + (*rem ? 1 : 0)) * step_granule + 2 * 6.0 + 2.0;
  return G;
}

// U = G g G^T in double, rounded once; packed [Cout/32][Cin/8][position][lane = h*32 + row][4] with channel 8(c/8) + 4h + s (s = 0, 1: body 0; 2, 3: body 1)
// s2d: `w` is a (Cout, Cin/4, 5, 5) stride-2 pad-2 kernel read as the 3x3 stride-1 pad-1 kernel over the 4 input phases it is
// equivalent to — channel phase*(Cin/4) + c, phase = py*2 + px, tap (a, b) = w[2a + py][2b + px] (zero where 2a + py or 2b + px = 5)
__global__ void pack_wino_kernel(float* __restrict__ packed, const float* __restrict__ w, int Cout, int Cin, long total, int s2d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int s = (int)(i & 3), r = (int)((i >> 2) & 31), h = (int)((i >> 7) & 1), q = (int)((i >> 8) & 15);
  const long bi = i >> 12;
  const int c8n = Cin >> 3;
  const int c8 = (int)(bi % c8n), mb = (int)(bi / c8n);
  const int co = mb * 32 + r, ci = c8 * 8 + 4 * h + s;
  const int xi = q >> 2, nu = q & 3;
  const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  double u = 0;
  if (s2d) {
    const int C0 = Cin >> 2, ph = ci / C0, py = ph >> 1, px = ph & 1;
    const float* g = w + ((long)co * C0 + (ci - ph * C0)) * 25;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b)
        if (2 * a + py < 5 && 2 * b + px < 5) u += G[xi][a] * (double)g[(2 * a + py) * 5 + 2 * b + px] * G[nu][b];
  } else {
    const float* g = w + ((long)co * Cin + ci) * 9;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) u += G[xi][a] * (double)g[a * 3 + b] * G[nu][b];
  }
  // positions nu = 3 are stored NEGATED, and every kernel multiplies them with the negated V (t3 - t1 instead of t1 - t3): the
  // products are the same bits, and conv_wino8_kernel's column pass becomes ONE v_fmac_f32_dpp per value (self + sgn * neighbour)
  packed[i] = nu == 3 ? -(float)u : (float)u;
}

extern "C" size_t deepim_conv_wino_packed_size(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || (Cout & 31) || (Cin & 7)) return 0;
  return (size_t)Cout * Cin * 16 * sizeof(float);
}

// Whether the layer should take the Winograd path. Cout % 64 == 0 (every encoder layer): the shared-transform kernel splits the input
// channels where the grid would not fill the chip, so it pays from two tile blocks on (measured at B = 4 / 8 / 32, tools/bench_wino.py:
// 1.45-2.1x over the direct kernels on every layer, conv5_1 / conv6_1 at B = 4 included); other channel counts fall back to the
// one-wave kernel, which walks all of Cin per block and needs >= 128 (stride-2 form: 256) blocks of 32 channels x 128 tiles.
// conv_max_split == 1 is the canonical-summation-order configuration (bit-exact against the oracle's default order): no Winograd there.
#ifndef WINO_MIN_BLOCKS
#define WINO_MIN_BLOCKS 128
#endif
#ifndef WINO_MIN_BLOCKS_S2D
#define WINO_MIN_BLOCKS_S2D 256
#endif
#ifndef WINO_MIN_TILES
#define WINO_MIN_TILES 64
#endif
static long wino_blocks(int B, int H, int W, int Cout) {
  const long tiles = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
  return (long)di_div_up(tiles, 128) * (Cout / 32);
}
static int wino_pays(deepim_ctx* ctx, int B, int H, int W, int Cout, long min_blocks) {
  if (ctx && ctx->conv_max_split == 1) return 0;
  if ((!ctx || (ctx->wino_shared && !ctx->wino_two_wave)) && (Cout & 63) == 0)   // (no context: the defaults)
    return (long)B * ((H + 1) / 2) * ((W + 1) / 2) >= WINO_MIN_TILES ? 1 : 0;
  return wino_blocks(B, H, W, Cout) >= min_blocks ? 1 : 0;
}
extern "C" int deepim_conv_wino_preferred(deepim_ctx* ctx, int B, int Cin, int H, int W, int Cout) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (Cout & 31) || (Cin & 7)) return 0;
  if ((size_t)B * Cin * H * W * 4 >= (1ull << 31)) return 0;   // one buffer descriptor per launch: larger inputs stay on the direct kernels (sub-batched there)
  if ((size_t)B * Cout * H * W * 4 >= (1ull << 31)) return 0;  // ... and one for the output
  return wino_pays(ctx, B, H, W, Cout, WINO_MIN_BLOCKS);
}
// the same question for a 5x5 stride-2 pad-2 layer with input (B, Cin, H, W) run over its space-to-depth form
extern "C" int deepim_conv_wino_preferred_s2d(deepim_ctx* ctx, int B, int Cin, int H, int W, int Cout) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (Cout & 31) || (Cin & 7) || ((H | W) & 1)) return 0;
  if ((size_t)B * Cin * H * W * 4 >= (1ull << 31) || (size_t)B * Cout * (H / 2) * (W / 2) * 4 >= (1ull << 31)) return 0;
  return wino_pays(ctx, B, H / 2, W / 2, Cout, WINO_MIN_BLOCKS_S2D);
}

extern "C" int deepim_conv_wino_pack_weights(deepim_ctx* ctx, float* packed_w, const float* w, int Cout, int Cin) {
  DI_DEVICE(ctx);
  DI_REQUIRE(Cout > 0 && Cin > 0 && (Cout & 31) == 0 && (Cin & 7) == 0, "conv_wino_pack_weights: Cout % 32 == 0 and Cin % 8 == 0 required");
  const long total = (long)Cout * Cin * 16;
  pack_wino_kernel<<<di_div_up(total, 256), 256, 0, ctx->stream>>>(packed_w, w, Cout, Cin, total, 0);
  DI_LAUNCH_CHECK();
  return 0;
}

// A 5x5 stride-2 pad-2 layer (conv2 / conv3) as this kernel's 3x3 stride-1 pad-1 problem over the space-to-depth input:
// (B, Cin, H, W) -> (B, 4 Cin, H/2, W/2), 16 positions x 4 Cin instead of 25 taps x 4 outputs x Cin per 2x2 tile = 1.56x fewer
// multiplies. w is the layer's own (Cout, Cin, 5, 5) tensor; the packed size is deepim_conv_wino_packed_size(Cout, 4 * Cin) and
// the forward call is deepim_conv2d_wino_forward(..., Cin = 4 * Cin, H / 2, W / 2, ...) on the space-to-depth NC8 tensor.
extern "C" int deepim_conv_wino_pack_weights_s2d(deepim_ctx* ctx, float* packed_w, const float* w, int Cout, int Cin) {
  DI_DEVICE(ctx);
  DI_REQUIRE(Cout > 0 && Cin > 0 && (Cout & 31) == 0 && (Cin & 7) == 0, "conv_wino_pack_weights_s2d: Cout % 32 == 0 and Cin % 8 == 0 required");
  const long total = (long)Cout * Cin * 4 * 16;
  pack_wino_kernel<<<di_div_up(total, 256), 256, 0, ctx->stream>>>(packed_w, w, Cout, Cin * 4, total, 1);
  DI_LAUNCH_CHECK();
  return 0;
}

// 3x3, stride 1, pad 1 convolution + bias + LeakyReLU(slope) from channel-blocked `in` (B, Cin/8, H, W, 8) into channel-blocked
// `out` (out_nc8 = 1; 3 = channel-blocked in space-to-depth order, what a stride-2 layer on this kernel reads) or into channels
// [out_coff, out_coff + Cout) of an NCHW tensor of out_ctotal channels (out_nc8 = 0).
static int wino_forward_impl(deepim_ctx* ctx, float* out, const float* in, const float* packed_w, const float* bias, int B, int Cin,
                             int H, int W, int Cout, float slope, int out_nc8, int out_ctotal, int out_coff, bool s2d,
                             int* plan_only = nullptr) {
  if (!plan_only) DI_DEVICE(ctx);      // (the plan is host arithmetic)
  DI_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv2d_wino_forward: bad shape");
  DI_REQUIRE((Cout & 31) == 0 && (Cin & 7) == 0, "conv2d_wino_forward: Cout % 32 == 0 and Cin % 8 == 0 required");
  if (B == 0) return 0;
  const size_t in_bytes = (size_t)B * Cin * H * W * 4, wd_bytes = (size_t)Cout * Cin * 64;
  DI_REQUIRE(in_bytes < (1ull << 31) && wd_bytes < (1ull << 31), "conv2d_wino_forward: tensor beyond the 2 GB buffer range");
  WinoParams p;
  p.in = in; p.out = out; p.wd = packed_w; p.bias = bias; p.slope = slope;
  p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
  p.TY = (H + 1) / 2; p.TX = (W + 1) / 2;
  p.ntiles = B * p.TY * p.TX;
  const bool two_wave = ctx->wino_two_wave != 0;   // dev option: two 8-position waves per SIMD on 64-tile blocks instead of one 16-position wave
  const bool shared = !two_wave && ctx->wino_shared && (Cout & 63) == 0;   // default: the 8-wave shared-transform kernel
  // block shape (ctx->wino_wide): 0 = 64 x 64; 3 = 128 channels x 32 tiles where Cout % 128 == 0; 2 = 64 x 32 on four waves, two blocks
  // per CU; 1 (default) = by the work per CU: the wide blocks share each V among 128 channels and win the long grids (2-4 %), the
  // four-wave blocks overlap one block's prologue / epilogue with the other's loop and win where a CU sees few blocks — measured
  // (tools/bench_wino.py at B = 4 / 8 / 32, both forms): the crossover sits near 100 steps of 8 input channels per CU
  bool half = shared && ctx->wino_wide == 2;
  if (shared && ctx->wino_wide == 1) {
    const long wide_blocks = (long)di_div_up(p.ntiles, 32) * di_div_up(Cout, 128);
    half = wide_blocks * (Cin / 8) <= 100L * 256 || (Cout & 127) != 0;
  }
  const bool wide = shared && !half && (Cout & 127) == 0 && ctx->wino_wide != 0;
  p.gx = di_div_up(p.ntiles, (wide || half) ? 32 : (two_wave || shared) ? 64 : 128);
  p.gy = wide ? Cout / 128 : shared ? Cout / 64 : Cout / 32;
  p.in_bytes = (unsigned)in_bytes; p.wd_bytes = (unsigned)wd_bytes;
  p.out_bytes = (unsigned)std::min<size_t>((size_t)B * Cout * H * W * 4, 0x7fffffffu);
  p.out_ctotal = out_ctotal > 0 ? out_ctotal : Cout;
  p.out_coff = out_coff;
  p.out_s2d = out_nc8 == 3 ? 1 : 0;
  p.grid0 = 1; p.kslice = Cin / 8; p.ksplit = 1; p.part_stride = 0; p.nvb = 1; p.part = nullptr;
  p.sk_G = 0; p.sk_gran = 0; p.sk_F = 0; p.sk_q = 0; p.sk_rem = 0; p.sk_nlb = 0; p.sk_count = nullptr;
  if (p.out_s2d) DI_REQUIRE(((H | W) & 1) == 0, "conv2d_wino_forward: space-to-depth output needs even H and W");
  int grid = p.gx * p.gy;
  if (shared) {
    DI_REQUIRE(!out_nc8 || (size_t)B * Cout * H * W * 4 < (1ull << 31), "conv2d_wino_forward: channel-blocked output beyond the 2 GB buffer range");
    // block -> (channel block, tile block) as conv_wino8_kernel maps it: gy < 8 dividing 8 deals 8 / gy XCDs to each channel block
    if ((p.gy & 7) != 0 && (8 % p.gy) == 0) grid = 8 * di_div_up(p.gx, 8 / p.gy);
    // the zero positions are dropped along an interleaved walk of the four input phases: two 8-channel blocks of each per loop body
    const bool ph8 = s2d && (Cin % 64) == 0 && ctx->wino_s2d_skip;
    // under-filled grids split the input channels (conv5_1 / conv6_1 at B = 32, every layer at the per-GPU shares of an 8-GPU node)
    const int nK = Cin / 8;
    const size_t out_elems = (size_t)B * Cout * H * W;
    int ks = nK;
    const int slots = half ? 512 : 256, gran = ph8 ? 8 : 2;
    double cost_split = 0, cost_sk = 0;
    int S = ctx->wino_split == 1 ? 1 : wino8_split_plan(grid, nK, gran, out_elems * 4 / 1e6, ctx->wino_split, &ks, slots, &cost_split);
    int skF = 0, skq = 0, skrem = 0;
    const int skG = (ctx->wino_split != 1 && ctx->wino_persistent && ctx->wino_streamk && out_nc8 && out_elems * 4 < (1ull << 31)) ? wino8_streamk_plan(grid, nK, gran, slots, &cost_sk, &skF, &skq, &skrem) : 0;
    // measured (bench.py A/B in one box): +1 % at B = 32 (4-18 whole rounds before the cut one), -1 % at B = 4 (one): from two whole rounds on
    // (the arrival counters are allocated on the first such launch: not inside a graph capture — run the sequence once eagerly first)
    const bool streamk = skG > 0 &&
                         (ctx->wino_streamk == 2 || (skF >= 2 && cost_sk < cost_split * 0.98));   // 2: wherever it applies (tests)
    if (streamk) { S = 1; ks = nK; }
    p.grid0 = grid; p.kslice = ks; p.ksplit = S; p.part_stride = 0; p.part = nullptr;
    if ((S > 1 || streamk) && !plan_only) {
      void* scr = nullptr;
      const int copies = streamk ? di_div_up(skG, skq) + 1 : S;   // a tile block of G granules cut by runs of >= q: at most that many pieces
      if (deepim_scratch(ctx, (size_t)copies * out_elems * 4, &scr) != 0) return -1;
      p.part = (float*)scr;
      p.part_stride = (long)out_elems;
    }
    if (S > 1) {
      if (!out_nc8) { p.out_ctotal = Cout; p.out_coff = 0; }   // dense NCHW partials
      grid *= S;
    }
    p.nvb = grid;
    if (ctx->wino_persistent) grid = (int)std::min<long>(grid, slots);   // one block per resident slot, each walks its share
    if (streamk) {
      p.sk_G = skG; p.sk_gran = gran; p.sk_F = skF; p.sk_q = skq; p.sk_rem = skrem; p.sk_nlb = slots / 8; grid = slots;
      if (!plan_only) p.sk_count = (int*)ctx->wino_counters;   // one word per tile block, zero between launches (the kernel leaves them so)
    }
    // K slices finished inside the kernel: the block whose slice of a tile block arrives last adds the S raw copies in slice order, the bias
    // and the activation — the sums wino_reduce_kernel would form, without its launch and its pass over the whole output
    const bool fin = S > 1 && ctx->wino_fin && p.grid0 <= DI_WINO_COUNTERS && (size_t)out_elems * 4 < (1ull << 31);
    p.fin_nchw = 0; p.fin_ctotal = 0; p.fin_coff = 0;
    if (fin) {
      if (!plan_only) p.sk_count = (int*)ctx->wino_counters;
      if (!out_nc8) { p.fin_nchw = 1; p.fin_ctotal = out_ctotal > 0 ? out_ctotal : Cout; p.fin_coff = out_coff; }
    }
    if (plan_only) {   // {block shape 0 / 1 wide / 2 four-wave, grid, K slices, K steps per slice, stream-K granules per tile block (0: off), granules per run, whole tile blocks per block before the run, tile blocks of the layer (incl. the padding of the XCD deal), runs that are one granule longer}
      plan_only[0] = half ? 2 : wide ? 1 : 0; plan_only[1] = grid; plan_only[2] = S; plan_only[3] = ks; plan_only[4] = p.sk_G; plan_only[5] = p.sk_q; plan_only[6] = p.sk_F; plan_only[7] = p.grid0; plan_only[8] = p.sk_rem;
      return 0;
    }
#define W8_LAUNCH(O, S)                                                                               \
    if (half) conv_wino4_kernel<O, S><<<grid, 256, 0, ctx->stream>>>(p);                              \
    else if (wide) conv_wino8_kernel<O, S, 1><<<grid, 512, 0, ctx->stream>>>(p);                      \
    else conv_wino8_kernel<O, S, 0><<<grid, 512, 0, ctx->stream>>>(p);
    if (ph8) {
      if (out_nc8) { W8_LAUNCH(1, 1) } else { W8_LAUNCH(0, 1) }
    } else {
      if (out_nc8) { W8_LAUNCH(1, 0) } else { W8_LAUNCH(0, 0) }
    }
#undef W8_LAUNCH
    if (S > 1 && !fin) {
      const long total = out_nc8 ? (long)(out_elems / 4) : (long)out_elems;
      const int hw = p.out_s2d ? H * W / 4 : H * W;
      wino_reduce_kernel<<<di_div_up(total, 256), 256, 0, ctx->stream>>>(out, p.part, bias, total, p.part_stride, S, slope, out_nc8 ? 1 : 0,
                                                                           (Cout >> 3) * (p.out_s2d ? 4 : 1), hw, Cout >> 3, Cout,
                                                                           out_ctotal > 0 ? out_ctotal : Cout, out_coff);
    }
    DI_LAUNCH_CHECK();
    return 0;
  }
  if (plan_only) return 0;
  // phase-by-phase K loop with the zero positions skipped: needs an even number of 8-channel blocks per input phase
  const bool phases = s2d && (Cin % 64) == 0 && ctx->wino_s2d_skip;
  if (two_wave) {
    if (out_nc8) conv_wino2_kernel<1><<<grid, 256, 0, ctx->stream>>>(p);
    else conv_wino2_kernel<0><<<grid, 256, 0, ctx->stream>>>(p);
  } else if (phases) {
    if (out_nc8) conv_wino_kernel<1, 1><<<grid, 256, 0, ctx->stream>>>(p);
    else conv_wino_kernel<0, 1><<<grid, 256, 0, ctx->stream>>>(p);
  } else {
    if (out_nc8) conv_wino_kernel<1><<<grid, 256, 0, ctx->stream>>>(p);
    else conv_wino_kernel<0><<<grid, 256, 0, ctx->stream>>>(p);
  }
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_conv2d_wino_forward(deepim_ctx* ctx, float* out, const float* in, const float* packed_w, const float* bias,
                                          int B, int Cin, int H, int W, int Cout, float slope, int out_nc8, int out_ctotal,
                                          int out_coff) {
  return wino_forward_impl(ctx, out, in, packed_w, bias, B, Cin, H, W, Cout, slope, out_nc8, out_ctotal, out_coff, false);
}

// The 5x5 stride-2 pad-2 layer itself: `in` = the space-to-depth NC8 form (B, 4 Cin, H/2, W/2) of its (B, Cin, H, W) input, packed_w
// from deepim_conv_wino_pack_weights_s2d, output (B, Cout, H/2, W/2). Same kernel as deepim_conv2d_wino_forward on (4 Cin, H/2, W/2),
// with the positions whose transformed weights are identically zero skipped (bit-identical results, 49 of 64 MFMAs).
extern "C" int deepim_conv2d_wino_forward_s2d(deepim_ctx* ctx, float* out, const float* in_s2d, const float* packed_w, const float* bias,
                                              int B, int Cin, int H, int W, int Cout, float slope, int out_nc8, int out_ctotal,
                                              int out_coff) {
  DI_REQUIRE(H > 0 && W > 0 && ((H | W) & 1) == 0, "conv2d_wino_forward_s2d: even H and W required");
  return wino_forward_impl(ctx, out, in_s2d, packed_w, bias, B, 4 * Cin, H / 2, W / 2, Cout, slope, out_nc8, out_ctotal, out_coff, true);
}

// The launch plan of the shared-transform kernel for a layer geometry (s2d: the arguments are the space-to-depth problem's, as
// wino_forward_impl sees them) under the context's options; plan[6] as documented at the fill site. -1 where that kernel is not used.
extern "C" int deepim_conv_wino_plan(deepim_ctx* ctx, int B, int Cin, int H, int W, int Cout, int out_nc8, int s2d, int* plan) {
  DI_REQUIRE(plan != nullptr, "conv_wino_plan: null plan");
  for (int i = 0; i < 9; ++i) plan[i] = -1;
  deepim_ctx defaults;                 // ctx == NULL: the plan under the default options (no device involved)
  if (!ctx) { deepim_ctx_default_options(&defaults); ctx = &defaults; }
  return wino_forward_impl(ctx, nullptr, nullptr, nullptr, nullptr, B, Cin, H, W, Cout, 0.f, out_nc8, 0, 0, s2d != 0, plan);
}

#if W8_TRACE
extern "C" int deepim_dev_w8_trace(void* buf) {
  unsigned long long* b = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_w8_trace), &b, sizeof(b));
}
#endif
#if WINO_TRACE
extern "C" int deepim_dev_wino_trace(void* buf) {
  unsigned long long* b = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wino_trace), &b, sizeof(b));
}
#endif
