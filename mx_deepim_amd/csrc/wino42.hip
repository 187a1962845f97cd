// fp32 Winograd F(4,3) x F(2,3) — 4x2 output tiles — for the encoder's 3x3 stride-1 layers on channel-blocked activations
// [n][C/8][h][w][8] (round 6; an entry point of its own, NOT bound by the network: profiles/r06_wino44.md sections 4-5 say why this
// tile, and what the kernel measured — 0.96-0.98x of the tuned F(2x2,3x3) kernel in its first, untuned form). Per 4-row x 2-column output tile and channel the 6x4 input patch d becomes V = B4^T d B2 (24 positions), the 3x3
// kernel g becomes U = G4 g G2^T (packed once, in double), the 24 positions are 24 independent GEMMs M = U.V over the input channels,
// and the tile is Y = A4^T M A2: 3 multiply-adds per output, input and output channel where F(2x2,3x3) needs 4 and the direct sum 9.
// The 6-point transform of F(4,3) (interpolation points 0, +-1, +-2, inf) costs accuracy: ~3e-6 of a layer's range against ~7e-7 for
// F(2x2,3x3) (tests/test_gpu_wino42.py: bar 1e-5 of range against the oracle's direct convolution, as for the other Winograd kernels).
//
// Built on conv_wino8_kernel's skeleton (csrc/wino.hip), cut for the register file: 24 positions x 64 channels x 32 tiles per block =
// 48 accumulator tuples on 8 waves = SIX per wave (96 of the 128 accumulation registers a wave has at two waves per SIMD; F(4x4,3x3)
// would need nine: 144).
//   * wave (pg, msub): column nu = pg of the 6x4 position grid (rows xi = 0..5), 32-channel sub-block msub, all 32 tiles;
//   * input transform ONCE per block: lane = (tile, 2-channel group, patch column j): six 8-byte pixel loads, the 6-point row pass in
//     registers, the 4-point column pass across the quad as one v_fmac_f32_dpp per value (positions nu = 3 negated on both sides, as in
//     wino.hip), six ds_write_b64;
//   * two V slots (28 KB each), one barrier per 8 input channels: stage s is loaded in step s-3, transformed in step s-2, read into the
//     operand registers in step s-1, multiplied in step s; weights straight from global memory (16 bytes per lane, position and step);
//   * output transform: the row direction (6 -> 4) in the wave's own registers, the column direction (4 -> 2) across the four pg waves
//     through LDS — each wave finishes a quarter of the channel rows: bias + LeakyReLU, 16-byte NC8 stores.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct W42Params {
  const float* in;
  float* out;
  const float* wd;     // [Cout/32][Cin/8][24 positions][64 lanes][4]
  const float* bias;
  float slope;
  int Cin, Cout, H, W, TY, TX, ntiles, gx, gy;
  unsigned in_bytes, wd_bytes, out_bytes;
  int out_ctotal, out_coff;   // NCHW output: channel slice of a wider tensor
};

#define W42_QS 1168                          /* V bytes per position: 2 k halves x (512 + 64 pad) + 16 */
#define W42_HS 576
#define W42_VSLOT (24 * W42_QS)              /* 28 032 B */
#define W42_XCH (8 * 48 * 64 * 4)            /* output exchange: 8 waves x 48 floats per lane */
#define W42_LDS_BYTES (W42_XCH > 2 * W42_VSLOT ? W42_XCH : 2 * W42_VSLOT)

__device__ __forceinline__ void w42_block_coords(const W42Params& p, const int bid, int& mb2, int& bx) {
  const int xcd = bid & 7, idx = bid >> 3;
  if ((p.gy & 7) == 0) {          // XCD x owns the channel blocks [x gy/8, (x+1) gy/8): its slice of U stays in its L2
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

template <int OUT_NC8>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino42_kernel(W42Params p) {
  __shared__ __attribute__((aligned(16))) char smem[W42_LDS_BYTES];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int msub = wave & 1, pg = wave >> 1;
  const int lrow = lane >> 5, lcol = lane & 31;
  int mb2, bx;
  w42_block_coords(p, blockIdx.x, mb2, bx);
  if (bx >= p.gx) return;
  const int tpi = p.TY * p.TX;
  const int c8n = p.Cin >> 3;
  const int hw32 = p.H * p.W * 32;
  constexpr int QS = W42_QS, VSLOT = W42_VSLOT;

  // ---- transform role: lane = (tile Tl of the block's 32, channel pair cg of the 8, patch column j)
  const int j = lane & 3, cg = (lane >> 2) & 3, Tl = wave * 4 + (lane >> 4);
  int voffT[6];
  {
    const int tT = bx * 32 + Tl;
    const bool tv = tT < p.ntiles;
    const int n = tv ? tT / tpi : 0;
    const int tr = tv ? tT - n * tpi : 0;
    const int ty = tr / p.TX, tx = tr - ty * p.TX;
    const int x = 2 * tx - 1 + j;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int y = 4 * ty - 1 + i;
      const bool ok = tv && y >= 0 && y < p.H && x >= 0 && x < p.W;
      voffT[i] = ok ? (((n * c8n) * p.H + y) * p.W + x) * 32 + cg * 8 : (int)0x80000000;
    }
  }
  const float sgn = j == 1 ? 1.f : -1.f;
  const unsigned vw = (unsigned)((cg >> 1) * W42_HS + (cg & 1) * 8 + Tl * 16 + j * QS);
  // ---- multiply role: positions (xi, nu = pg), xi = 0 .. 5
  const unsigned rb = (unsigned)(pg * QS + lrow * W42_HS + lcol * 16);
  const int ra_g = lane * 16 + pg * 1024;
  const int ra_s0 = ((mb2 * 2 + msub) * c8n) * 24576;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wd, 0, (int)p.wd_bytes, 0x00020000);

  f32x16 acc[6];
  {
    const float zf = 0.f;
#pragma unroll
    // (s_nop: the hazard pass cannot see inside the asm, and the v_mov that makes zf may sit right in front of the first MFMA —
    // a VALU write followed by an MFMA read of the same register needs wait states; without them acc[0] started from garbage)
    for (int q = 0; q < 6; ++q) asm volatile("s_nop 4\n\tv_mfma_f32_32x32x2_f32 %0, %1, %1, 0" : "=a"(acc[q]) : "v"(zf));
  }
  f32x2 raw[6], T[6];
  f32x4 A[6], Bv[6];
  const int ke = c8n;

#ifndef W42_ABL
#define W42_ABL 0   // dev ablations (wrong results): 1 no step barriers, 4 no pixel loads + transform + V stores in the loop, 8 no operand reads in the loop, 16 no output transform / stores, 32 no MFMAs
#endif
#define W42_LDS4(off) (*reinterpret_cast<f32x4*>(smem + (off)))
#define W42_PIX(stage)                                                                                \
  {                                                                                                   \
    const int st_ = min((stage), ke - 1);                                                             \
    const int so_ = __builtin_amdgcn_readfirstlane(st_ * hw32);                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_)                                                  \
      raw[i_] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voffT[i_], so_, 0)); \
  }
// the 6-point row pass of F(4,3) (B4^T), per channel: t0 = 4 d0 - 5 d2 + d4, t1 = (d4 - 4 d2) + (d3 - 4 d1), t2 = (d4 - 4 d2) - (d3 - 4 d1),
// t3 = (d4 - d2) + 2 (d3 - d1), t4 = (d4 - d2) - 2 (d3 - d1), t5 = 4 d1 - 5 d3 + d5 — 12 operations
#define W42_ROW1(c)                                                                                   \
  {                                                                                                   \
    const float d0 = raw[0].c, d1 = raw[1].c, d2 = raw[2].c, d3 = raw[3].c, d4 = raw[4].c, d5 = raw[5].c; \
    const float a_ = __builtin_fmaf(-4.f, d2, d4), b_ = __builtin_fmaf(-4.f, d1, d3);                 \
    const float c_ = d4 - d2, e_ = d3 - d1;                                                           \
    T[0].c = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));                                   \
    T[1].c = a_ + b_; T[2].c = a_ - b_;                                                               \
    T[3].c = __builtin_fmaf(2.f, e_, c_); T[4].c = __builtin_fmaf(-2.f, e_, c_);                      \
    T[5].c = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));                                   \
  }
#define W42_ROW() { W42_ROW1(x) W42_ROW1(y) }
// column pass of row xi across the quad, in place (see wino.hip W8_COL): self + sgn * T[lane (2, 2, 1, 1)[j]]; the wait states and the DPP
// reads they protect are one asm statement
#define W42_DPPQ "quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf"
#define W42_COL(xi, slot_)                                                                            \
  {                                                                                                   \
    float c0_ = T[xi].x, c1_ = T[xi].y;                                                               \
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 " W42_DPPQ "\n\tv_fmac_f32_dpp %1, %1, %2 " W42_DPPQ \
                 : "+v"(c0_), "+v"(c1_) : "v"(sgn));                                                  \
    f32x2 v_; v_.x = c0_; v_.y = c1_;                                                                 \
    *reinterpret_cast<f32x2*>(smem + (vw + (unsigned)((slot_) * VSLOT + (xi) * 4 * QS))) = v_;        \
  }
#define W42_RDA(i) A[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrw, ra_g + (i) * 4096, rsa_, 0));
#define W42_RDB(i, slot_) Bv[i] = W42_LDS4(rb + (unsigned)((slot_) * VSLOT + (i) * 4 * QS));
#define W42_RD(i, slot_) { W42_RDA(i) W42_RDB(i, slot_) }
#define W42_MFMA(i, s_)                                                                               \
  acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32((s_) == 0 ? A[i].x : (s_) == 1 ? A[i].y : (s_) == 2 ? A[i].z : A[i].w, \
                                                (s_) == 0 ? Bv[i].x : (s_) == 1 ? Bv[i].y : (s_) == 2 ? Bv[i].z : Bv[i].w, acc[i], 0, 0, 0); \
  asm volatile("" : "+a"(acc[i]));
#define W42_SYNC()                                                                                    \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
  __builtin_amdgcn_s_barrier();                                                                       \
  asm volatile("" ::: "memory");
// two positions interleaved over eight MFMA slots; X(sl) = the slot's share of the other roles
#define W42_M2X(i0, i1, X)                                                                            \
  _Pragma("unroll") for (int sl = 0; sl < 8; ++sl) {                                                  \
    if (!(W42_ABL & 32)) { if (sl & 1) { W42_MFMA(i1, sl >> 1) } else { W42_MFMA(i0, sl >> 1) } }     \
    X(sl) __builtin_amdgcn_sched_barrier(0);                                                          \
  }
#define W42_R2(i0, i1, SLR) if (!(W42_ABL & 8)) { W42_RD(i0, SLR) W42_RD(i1, SLR) } __builtin_amdgcn_sched_barrier(0);
#define W42_XNONE(sl)
// transform of stage xst_ into slot xsl_ over the eight slots of an MFMA block pair: row pass, six column passes, then the pixel loads
// of stage xst_ + 1 (behind the row pass: the loads' registers are its inputs)
#define W42_XA(sl)                                                                                    \
    if (!(W42_ABL & 4)) {                                                                             \
      if (sl == 0) { W42_ROW() }                                                                      \
      if (sl >= 1 && sl <= 3) { W42_COL(sl - 1, xsl_) }                                               \
      if (sl == 4) { W42_PIX(xst_ + 1) }                                                              \
      if (sl >= 5 && sl <= 7) { W42_COL(sl - 2, xsl_) }                                               \
    }

  // ---- prologue: stage 0 into slot 0 and into the operand registers, stage 1 into slot 1, pixels of stage 2 in flight
  W42_PIX(0)
  W42_ROW()
  W42_PIX(1)
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) W42_COL(xi, 0)
  W42_SYNC()
  {
    const int rsa_ = __builtin_amdgcn_readfirstlane(ra_s0);
#pragma unroll
    for (int i = 0; i < 6; ++i) { W42_RDA(i) W42_RDB(i, 0) }
  }
  W42_ROW()
  W42_PIX(2)
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) W42_COL(xi, 1)
  __builtin_amdgcn_sched_barrier(0);

  // ---- one step per stage k (SL = k & 1), one barrier each
#define W42_STEP(k, SL)                                                                               \
  {                                                                                                   \
    const int rsa_ = __builtin_amdgcn_readfirstlane(ra_s0 + min((k) + 1, ke - 1) * 24576);            \
    const int xst_ = (k) + 2; constexpr int xsl_ = (SL);                                              \
    W42_M2X(0, 1, W42_XNONE)                                                                          \
    if (!(W42_ABL & 1)) { W42_SYNC() }                                                                \
    W42_R2(0, 1, (SL) ^ 1)                                                                            \
    W42_M2X(2, 3, W42_XA)                                                                             \
    W42_R2(2, 3, (SL) ^ 1)                                                                            \
    W42_M2X(4, 5, W42_XNONE)                                                                          \
    W42_R2(4, 5, (SL) ^ 1)                                                                            \
  }
  {
    int k = 0;
    for (; k + 2 <= ke; k += 2) {
      W42_STEP(k, 0)
      W42_STEP(k + 1, 1)
    }
    if (k < ke) W42_STEP(k, 0)
  }
#undef W42_STEP

  // ---- output transform. acc[xi][r]: channel (r&3) + 8(r>>2) + 4 lrow of the wave's 32, tile lcol, position (xi, nu = pg).
  W42_SYNC()   // every wave is done with both slots
  if (W42_ABL & 16) { if (acc[0][0] == 123.f && acc[5][15] == 4.f) p.out[0] = acc[3][2]; return; }
  // row direction (A4^T, 6 -> 4) on whole vectors: p = m1 + m2, q = m1 - m2, u = m3 + m4, v = m3 - m4;
  // r0 = (m0 + p) + u, r1 = q + 2 v, r2 = p + 4 u, r3 = (q + 8 v) + m5
  f32x16 rr[4];
  {
    const f32x16 pp = acc[1] + acc[2], qq = acc[1] - acc[2], uu = acc[3] + acc[4], vv = acc[3] - acc[4];
    rr[0] = (acc[0] + pp) + uu;
    rr[1] = qq + 2.f * vv;
    rr[2] = pp + 4.f * uu;
    rr[3] = (qq + 8.f * vv) + acc[5];
  }
  float* xw = reinterpret_cast<float*>(smem) + wave * (48 * 64) + lane;
  // what the other three column waves of this channel sub-block finish: rows 4 pg' .. 4 pg' + 3, slot d = index of pg' among them
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int pd = d < pg ? d : d + 1;      // destination wave's pg (wave-uniform)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      // rows 4 pd + e of rr[a]: pd is not a compile-time constant — select among the four row groups
      float v0, v1, v2, v3;
      if (pd == 0) { v0 = rr[a][0]; v1 = rr[a][1]; v2 = rr[a][2]; v3 = rr[a][3]; }
      else if (pd == 1) { v0 = rr[a][4]; v1 = rr[a][5]; v2 = rr[a][6]; v3 = rr[a][7]; }
      else if (pd == 2) { v0 = rr[a][8]; v1 = rr[a][9]; v2 = rr[a][10]; v3 = rr[a][11]; }
      else { v0 = rr[a][12]; v1 = rr[a][13]; v2 = rr[a][14]; v3 = rr[a][15]; }
      xw[(d * 16 + a * 4 + 0) * 64] = v0; xw[(d * 16 + a * 4 + 1) * 64] = v1;
      xw[(d * 16 + a * 4 + 2) * 64] = v2; xw[(d * 16 + a * 4 + 3) * 64] = v3;
    }
  }
  W42_SYNC()
  const int t = bx * 32 + lcol;
  if (t >= p.ntiles) return;
  const int n = t / tpi, trm = t - n * tpi;
  const int ty = trm / p.TX, tx = trm - ty * p.TX;
  const int y0 = 4 * ty, x0 = 2 * tx;
  const int mb = mb2 * 2 + msub;
  // column direction (A2^T, 4 -> 2) for this wave's rows 4 pg + e: values of nu = 0 .. 3 — its own and the three others'
  float val[4][4][4];   // [nu][a][e]
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) {
    if (nu == pg) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (pg == 0) { val[nu][a][0] = rr[a][0]; val[nu][a][1] = rr[a][1]; val[nu][a][2] = rr[a][2]; val[nu][a][3] = rr[a][3]; }
        else if (pg == 1) { val[nu][a][0] = rr[a][4]; val[nu][a][1] = rr[a][5]; val[nu][a][2] = rr[a][6]; val[nu][a][3] = rr[a][7]; }
        else if (pg == 2) { val[nu][a][0] = rr[a][8]; val[nu][a][1] = rr[a][9]; val[nu][a][2] = rr[a][10]; val[nu][a][3] = rr[a][11]; }
        else { val[nu][a][0] = rr[a][12]; val[nu][a][1] = rr[a][13]; val[nu][a][2] = rr[a][14]; val[nu][a][3] = rr[a][15]; }
      }
    } else {
      const int d = pg < nu ? pg : pg - 1;      // this wave's slot among wave nu's three destinations
      const float* xr = reinterpret_cast<const float*>(smem) + (nu * 2 + msub) * (48 * 64) + lane;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) val[nu][a][e] = xr[(d * 16 + a * 4 + e) * 64];
    }
  }
  const float4 bq = p.bias ? *reinterpret_cast<const float4*>(p.bias + mb * 32 + 8 * pg + 4 * lrow) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int C8o = p.Cout >> 3;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int y = y0 + a;
    if (y >= p.H) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int x = x0 + b;
      if (x >= p.W) continue;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float bv = e == 0 ? bq.x : e == 1 ? bq.y : e == 2 ? bq.z : bq.w;
        float v = b == 0 ? (val[0][a][e] + val[1][a][e]) + val[2][a][e] : (val[1][a][e] - val[2][a][e]) - val[3][a][e];
        v += bv;
        o[e] = v > 0.f ? v : v * p.slope;
      }
      if (OUT_NC8) {
        const long rec = (((long)n * C8o + mb * 4 + pg) * p.H + y) * p.W + x;
        *reinterpret_cast<float4*>(p.out + rec * 8 + lrow * 4) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        const long c0 = (long)n * p.out_ctotal + p.out_coff + mb * 32 + 8 * pg + 4 * lrow;
#pragma unroll
        for (int e = 0; e < 4; ++e) p.out[((c0 + e) * p.H + y) * p.W + x] = o[e];
      }
    }
  }
}

// U = G4 g G2^T in double, rounded once; packed [Cout/32][Cin/8][position = xi*4 + nu][lane = h*32 + row][4] with input channel
// 8 (c/8) + 4 h + s; positions nu = 3 negated (the column pass computes t3 - t1)
__global__ void pack_wino42_kernel(float* __restrict__ packed, const float* __restrict__ w, int Cout, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int within = (int)(i % 6144);
  const long bi = i / 6144;
  const int s = within & 3, r = (within >> 2) & 31, h = (within >> 7) & 1, q = within >> 8;
  const int c8n = Cin >> 3;
  const int c8 = (int)(bi % c8n), mb = (int)(bi / c8n);
  const int co = mb * 32 + r, ci = c8 * 8 + 4 * h + s;
  const int xi = q >> 2, nu = q & 3;
  const double G4[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                           {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const float* g = w + ((long)co * Cin + ci) * 9;
  double u = 0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) u += G4[xi][a] * (double)g[a * 3 + b] * G2[nu][b];
  packed[i] = nu == 3 ? -(float)u : (float)u;
}

extern "C" size_t deepim_conv_wino42_packed_size(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || (Cout & 63) || (Cin & 7)) return 0;
  return (size_t)Cout * Cin * 24 * sizeof(float);
}

extern "C" int deepim_conv_wino42_pack_weights(deepim_ctx* ctx, float* packed_w, const float* w, int Cout, int Cin) {
  DI_DEVICE(ctx);
  DI_REQUIRE(Cout > 0 && Cin > 0 && (Cout & 63) == 0 && (Cin & 7) == 0, "conv_wino42_pack_weights: Cout % 64 == 0 and Cin % 8 == 0 required");
  const long total = (long)Cout * Cin * 24;
  pack_wino42_kernel<<<di_div_up(total, 256), 256, 0, ctx->stream>>>(packed_w, w, Cout, Cin, total);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_conv2d_wino42_forward(deepim_ctx* ctx, float* out, const float* in, const float* packed_w, const float* bias, int B,
                                            int Cin, int H, int W, int Cout, float slope, int out_nc8, int out_ctotal, int out_coff) {
  DI_DEVICE(ctx);
  DI_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv2d_wino42_forward: bad shape");
  DI_REQUIRE((Cout & 63) == 0 && (Cin & 7) == 0, "conv2d_wino42_forward: Cout % 64 == 0 and Cin % 8 == 0 required");
  DI_REQUIRE(out_nc8 == 0 || out_nc8 == 1, "conv2d_wino42_forward: NC8 or NCHW output");
  if (B == 0) return 0;
  const size_t in_bytes = (size_t)B * Cin * H * W * 4, wd_bytes = (size_t)Cout * Cin * 96;
  DI_REQUIRE(in_bytes < (1ull << 31) && wd_bytes < (1ull << 31), "conv2d_wino42_forward: tensor beyond the 2 GB buffer range");
  W42Params p;
  p.in = in; p.out = out; p.wd = packed_w; p.bias = bias; p.slope = slope;
  p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
  p.TY = (H + 3) / 4; p.TX = (W + 1) / 2;
  p.ntiles = B * p.TY * p.TX;
  p.gx = di_div_up(p.ntiles, 32);
  p.gy = Cout / 64;
  p.in_bytes = (unsigned)in_bytes; p.wd_bytes = (unsigned)wd_bytes;
  p.out_bytes = (unsigned)std::min<size_t>((size_t)B * Cout * H * W * 4, 0x7fffffffu);
  p.out_ctotal = out_ctotal > 0 ? out_ctotal : Cout;
  p.out_coff = out_coff;
  int grid = p.gx * p.gy;
  if ((p.gy & 7) != 0 && (8 % p.gy) == 0) grid = 8 * di_div_up(p.gx, 8 / p.gy);
  if (out_nc8) conv_wino42_kernel<1><<<grid, 512, 0, ctx->stream>>>(p);
  else conv_wino42_kernel<0><<<grid, 512, 0, ctx->stream>>>(p);
  DI_LAUNCH_CHECK();
  return 0;
}
