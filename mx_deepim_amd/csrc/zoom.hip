// Z-group: zoom-in crop/warp.
//   Z0 MXNet GridGenerator(affine)+BilinearSampler (third-party, call sites zoom_mask.py:96-107)
//   Z1 zoom_mask.py:29-112            Z2 zoom_image_with_factor.py:31-65
//   Z3 zoom_image.py:26-107           Z4 zoom_depth.py:24-44
//   Z5 zoom_flow.py:28-71             Z6 zoom_mask_with_factor.py:29-64
//   Z7 zoom_trans.py:22-74
//
// Three kernels: (1) per-sample bounding boxes of the two validity maps
// (wavefront min/max reductions via DPP row/bank shuffles + one atomic per block),
// (2) one thread per sample turns boxes + projected object centre into the zoom
// factor with the reference's float64/float32 promotion pattern, (3) one generic
// affine bilinear resampler that serves every Z op through a small by-value
// channel plan (pre-op, post-op per channel) — lanes run along W so writes are
// full 256 B wave stores and the 4-tap reads hit the same few source rows.
// HBM-bound: algorithmic bytes = crop-region read + full-frame write per channel.
#include "common.h"
#include <limits.h>

namespace {

// ---------------------------------------------------------------- bbox ----
enum BBoxMode { BB_MASK_GT = 0 /* v > 0.3 */, BB_MASK_RENDERED = 1 /* v > 0.2 */, BB_IMAGE = 2 /* sum_c(v+mean) > 0.01 */ };

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

__global__ void bbox_init_kernel(int* __restrict__ box, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) box[i] = (i & 1) ? -1 : INT_MAX;  // {minx,maxx,miny,maxy}
}

// grid (rows/ROWS_PER_BLOCK, B, 2 maps); block 256 threads sweep ROWS_PER_BLOCK rows
constexpr int BB_ROWS = 8;
__global__ __launch_bounds__(256) void bbox_kernel(int* __restrict__ box, const float* __restrict__ map0,
                                                   const float* __restrict__ map1, int mode0, int mode1, Vec3 means,
                                                   int H, int W) {
  const int b = blockIdx.y, m = blockIdx.z;
  const float* src = m == 0 ? map0 : map1;
  const int mode = m == 0 ? mode0 : mode1;
  const size_t plane = (size_t)H * W;
  const int nch = mode == BB_IMAGE ? 3 : 1;
  src += (size_t)b * nch * plane;
  const int r0 = blockIdx.x * BB_ROWS;
  int minx = INT_MAX, maxx = -1, miny = INT_MAX, maxy = -1;
  for (int r = r0; r < min(r0 + BB_ROWS, H); ++r) {
    for (int x = threadIdx.x; x < W; x += 256) {
      const size_t p = (size_t)r * W + x;
      bool valid;
      if (mode == BB_MASK_GT) valid = src[p] > 0.3f;
      else if (mode == BB_MASK_RENDERED) valid = src[p] > 0.2f;
      else {
        float s = (src[p] + means.v[0]) + (src[plane + p] + means.v[1]);
        s = s + (src[2 * plane + p] + means.v[2]);
        valid = s > 0.01f;
      }
      if (valid) { minx = min(minx, x); maxx = max(maxx, x); miny = min(miny, r); maxy = max(maxy, r); }
    }
  }
  minx = wave_min(minx); maxx = wave_max(maxx); miny = wave_min(miny); maxy = wave_max(maxy);
  __shared__ int red[4][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[wave][0] = minx; red[wave][1] = maxx; red[wave][2] = miny; red[wave][3] = maxy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) {
      minx = min(minx, red[i][0]); maxx = max(maxx, red[i][1]);
      miny = min(miny, red[i][2]); maxy = max(maxy, red[i][3]);
    }
    int* o = box + (b * 2 + m) * 4;
    if (maxx >= 0) { atomicMin(o + 0, minx); atomicMax(o + 1, maxx); atomicMin(o + 2, miny); atomicMax(o + 3, maxy); }
  }
}

// ---------------------------------------------------------- zoom factor ----
// zoom_mask.py:47-103 / zoom_image.py:41-98 under NumPy-1.x ("legacy") scalar promotion, the
// reference's era and the parity target (pinned by tests/golden/zoom_golden.npz, which runs the
// reference's own lines): K·t and the centre cx = c0/c2 are float32; everything after that mixes the
// float32 scalar with Python ints / int64 box edges and is therefore float64 — distances, crop, and
// tx = cx / W * 2 - 1 — rounded once when stored to the float32 zoom_factor.
// rearm: the accumulators are the context's persistent ones — put them back to "empty" once read, so that the NEXT zoom-factor
// computation needs no init launch (they are armed at context creation; every consumer re-arms exactly what it read)
__global__ void zoom_factor_kernel(float* __restrict__ zoom_factor, int* __restrict__ status,
                                   int* __restrict__ box, const float* __restrict__ src_pose, Mat3 K, int B,
                                   int H, int W, int rearm) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int real[4], rend[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { real[i] = box[(b * 2 + 0) * 4 + i]; rend[i] = box[(b * 2 + 1) * 4 + i]; }
  if (rearm) {
#pragma unroll
    for (int i = 0; i < 8; ++i) box[b * 8 + i] = (i & 1) ? -1 : INT_MAX;
  }
  float* zf = zoom_factor + b * 4;
  if (real[1] < 0) {  // reference raises ValueError (np.min of empty) — flag it
    const float nanv = __int_as_float(0x7fc00000);
    zf[0] = zf[1] = zf[2] = zf[3] = nanv;
    atomicOr(status, DI_STATUS_ZOOM_EMPTY);
    return;
  }
  const double rsx = real[0], rex = real[1], rsy = real[2], rey = real[3];
  const float* t = src_pose + b * 12;
  const float t0 = t[3], t1 = t[7], t2 = t[11];
  const float c0 = K.v[0] * t0 + K.v[1] * t1 + K.v[2] * t2;
  const float c1 = K.v[3] * t0 + K.v[4] * t1 + K.v[5] * t2;
  const float c2 = K.v[6] * t0 + K.v[7] * t1 + K.v[8] * t2;
  const float cxf = c0 / c2, cyf = c1 / c2;
  double osx, oex, osy, oey, zcx, zcy;
  float txf, tyf;
  if (rend[1] < 0) {
    osx = rsx; oex = rex; osy = rsy; oey = rey;
    zcx = (rsx + rex) * 0.5;
    zcy = (rsy + rey) * 0.5;
    txf = (float)(zcx / W * 2 - 1);
    tyf = (float)(zcy / H * 2 - 1);
  } else {
    osx = rend[0]; oex = rend[1]; osy = rend[2]; oey = rend[3];
    zcx = (double)cxf;
    zcy = (double)cyf;
    txf = (float)(zcx / W * 2 - 1);
    tyf = (float)(zcy / H * 2 - 1);
  }
  const double left = fmax(zcx - osx, zcx - rsx);
  const double right = fmax(oex - zcx, rex - zcx);
  const double up = fmax(zcy - osy, zcy - rsy);
  const double down = fmax(rey - zcy, oey - zcy);
  const double crop = fmax(fmax(0.75 * right, 0.75 * left), fmax(up, down)) * 1.4 * 2;
  const float wx = (float)(crop / H);
  zf[0] = wx; zf[1] = wx; zf[2] = txf; zf[3] = tyf;
}

// ------------------------------------------------------------ resampler ----
enum ChanFlags {
  CF_PRE_BIN02 = 1,      // v = v > 0.2 ? 1 : 0 before sampling (zoom_mask.py:40-41)
  CF_POST_ROUND = 2,     // roundf (mx.nd.round, half away from zero)
  CF_POST_DIV255 = 4,    // /255 (deepIM_flownet.py:35-36)
  CF_POST_MUL_WX = 8,    // *wx (zoom_flow.py:62)
  CF_POST_DIV_WX = 16,   // /wx (zoom_flow.py:64)
  CF_POST_ROUND_M045 = 32,  // round(v - 0.45) (zoom_flow.py:70-72)
  CF_HIGHLIGHT = 64,     // maximum(v, centre map) before un-mean (zoom_image_with_factor.py:59-60)
  CF_HIGHLIGHT_RED = 128,
  CF_PRE_SIGMOID = 256,  // v = sigmoid(v) before the pre-binarise (mask head, deepIM_flownet.py:650)
};
struct Chan {
  const float* src;  // plane of sample 0
  float* dst;
  long src_bstride, dst_bstride;  // elements between samples
  float mean;                     // added before sampling, subtracted after
  int flags;
};
constexpr int MAX_CHAN = 10;
struct Plan {
  Chan ch[MAX_CHAN];
  int n;
  int inverse;  // build the inverse zoom from the stored factor (zoom_flow.py:36-44)
};

struct Affine { float wx, wy, tx, ty, wx_in; };

__device__ __forceinline__ Affine load_affine(const float* __restrict__ zoom_factor, int b, int inverse, int H, int W) {
  const float wx_in = zoom_factor[b * 4 + 0], wy_in = zoom_factor[b * 4 + 1];
  const float tx_in = zoom_factor[b * 4 + 2], ty_in = zoom_factor[b * 4 + 3];
  Affine a;
  a.wx_in = wx_in;
  if (!inverse) { a.wx = wx_in; a.wy = wy_in; a.tx = tx_in; a.ty = ty_in; return a; }
  // zoom_flow.py:36-44 / zoom_mask_with_factor.py:43-52, legacy promotion: the float32 scalars meet Python
  // ints/floats on every line, so the whole chain is float64, rounded once into the float32 affine matrix
  const double wxi = wx_in, wyi = wy_in, txi = tx_in, tyi = ty_in;
  a.wx = (float)(1 / wxi);
  a.wy = (float)(1 / wyi);
  const double crop_w = wxi * W, crop_h = wyi * H;
  const double cx = txi * 0.5 * W + 0.5 * W;
  const double cy = tyi * 0.5 * H + 0.5 * H;
  a.tx = (float)((W * 0.5 - cx) / crop_w * 2);
  a.ty = (float)((H * 0.5 - cy) / crop_h * 2);
  return a;
}

struct Taps {
  int x0, y0;
  bool in00, in01, in10, in11;  // [y][x]
  float wy0, wx0;               // weight of the top row / left column
};

// GridGenerator(affine) + BilinearSampler coordinate math (Z0), f32, unfused
__device__ __forceinline__ Taps make_taps(const Affine& a, int h, int w, int H, int W, float gx_step, float gy_step) {
  const float xd = -1.0f + (float)w * gx_step;
  const float yd = -1.0f + (float)h * gy_step;
  const float xs = a.wx * xd + a.tx;
  const float ys = a.wy * yd + a.ty;
  const float xr = (xs + 1.f) * (float)(W - 1) / 2.f;
  const float yr = (ys + 1.f) * (float)(H - 1) / 2.f;
  Taps t;
  // clamp before the int cast so huge/NaN coordinates are simply "all taps outside"
  const float xf = floorf(fminf(fmaxf(xr, -4.f), (float)W + 4.f));
  const float yf = floorf(fminf(fmaxf(yr, -4.f), (float)H + 4.f));
  t.x0 = (int)xf;
  t.y0 = (int)yf;
  t.wx0 = 1.0f - (xr - xf);
  t.wy0 = 1.0f - (yr - yf);
  const bool x0in = t.x0 >= 0 && t.x0 <= W - 1, x1in = t.x0 + 1 >= 0 && t.x0 + 1 <= W - 1;
  const bool y0in = t.y0 >= 0 && t.y0 <= H - 1, y1in = t.y0 + 1 >= 0 && t.y0 + 1 <= H - 1;
  const bool finite = (xr == xr) && (yr == yr);
  t.in00 = finite && y0in && x0in; t.in01 = finite && y0in && x1in;
  t.in10 = finite && y1in && x0in; t.in11 = finite && y1in && x1in;
  return t;
}

// v / 255.0f, bit for bit, in 5 VALU ops instead of the ~12 of an IEEE division sequence: Markstein's correction of
// q0 = v·RN(1/255) with the exact FMA residual. Checked against true division for ALL 2^32 float bit patterns
// (deepim_selfcheck_div255 on the device, tests/test_oracle_thirdparty.py on the host): the only inputs where the
// corrected quotient differs are ±0 and ±inf, for which q0 itself is the exact answer.
__device__ __forceinline__ float div255(float v) {
  const float c = 1.0f / 255.0f;
  const float q0 = v * c;
  const float r = fmaf(-255.0f, q0, v);
  const float q1 = fmaf(r, c, q0);
  return __builtin_amdgcn_classf(v, 0x264) ? q0 : q1;   // class mask: -inf | -0 | +0 | +inf
}

__device__ __forceinline__ float pre_op(float v, float mean, int flags) {
  if (flags & CF_PRE_SIGMOID) v = 1.0f / (1.0f + expf(-v));
  if (flags & CF_PRE_BIN02) v = v > 0.2f ? 1.f : (v <= 0.2f ? 0.f : v);
  return v + mean;
}

// BilinearSampler blend with the reference's float/double mix (bilinear_sampler.cc forward)
__device__ __forceinline__ float blend(float tl, float tr, float bl, float br, float wy0, float wx0) {
  const float t1 = tl * wy0 * wx0;
  const double t2 = (double)(tr * wy0) * (1.0 - (double)wx0);
  const double t3 = (double)bl * (1.0 - (double)wy0) * (double)wx0;
  const double t4 = (double)br * (1.0 - (double)wy0) * (1.0 - (double)wx0);
  return (float)((((double)t1 + t2) + t3) + t4);
}

// grid (ceil(W/256), H, B)
__global__ __launch_bounds__(256) void resample_kernel(Plan plan, const float* __restrict__ zoom_factor, int H, int W,
                                                       float gx_step, float gy_step) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  if (w >= W) return;
  const Affine a = load_affine(zoom_factor, b, plan.inverse, H, W);
  const Taps t = make_taps(a, h, w, H, W, gx_step, gy_step);
  const long o00 = (long)t.y0 * W + t.x0;
  const long opix = (long)h * W + w;
  const bool spot = h >= (int)floorf((float)H / 2 - 5) && h < (int)ceilf((float)H / 2 + 5) &&
                    w >= (int)floorf((float)W / 2 - 5) && w < (int)ceilf((float)W / 2 + 5);
  for (int c = 0; c < plan.n; ++c) {
    const Chan& ch = plan.ch[c];
    const float* s = ch.src + (long)b * ch.src_bstride;
    const int fl = ch.flags;
    const float tl = t.in00 ? pre_op(s[o00], ch.mean, fl) : 0.f;
    const float tr = t.in01 ? pre_op(s[o00 + 1], ch.mean, fl) : 0.f;
    const float bl = t.in10 ? pre_op(s[o00 + W], ch.mean, fl) : 0.f;
    const float br = t.in11 ? pre_op(s[o00 + W + 1], ch.mean, fl) : 0.f;
    float v = blend(tl, tr, bl, br, t.wy0, t.wx0);
    if (fl & CF_HIGHLIGHT) v = fmaxf(v, ((fl & CF_HIGHLIGHT_RED) && spot) ? 255.f : 0.f);
    v = v - ch.mean;
    if (fl & CF_POST_ROUND) v = roundf(v);
    if (fl & CF_POST_DIV255) v = div255(v);
    if (fl & CF_POST_MUL_WX) v = v * a.wx_in;
    if (fl & CF_POST_DIV_WX) v = v / a.wx_in;
    if (fl & CF_POST_ROUND_M045) v = roundf(v - 0.45f);
    ch.dst[(long)b * ch.dst_bstride + opix] = v;
  }
}

// 4 consecutive output pixels per thread (W % 4 == 0): the row taps are shared, the 16 tap loads of a channel are issued
// together, results leave as one dwordx4 store; 1-D grid over pixel quads so no lane is idle on the 640-wide rows. Same
// arithmetic, operation by operation, as resample_kernel.
struct Quad {             // everything of a pixel quad that does not depend on the channel
  Taps t[4];
  long row0, row1, opix;
  int xc0[4], xc1[4];
  int h, w0;
};
__device__ __forceinline__ Quad make_quad(const Affine& a, int h, int w0, int H, int W, float gx_step, float gy_step) {
  Quad q;
  q.h = h; q.w0 = w0;
#pragma unroll
  for (int i = 0; i < 4; ++i) q.t[i] = make_taps(a, h, w0 + i, H, W, gx_step, gy_step);
  // taps outside the frame contribute 0 (BilinearSampler zero padding): load from a clamped in-frame address and select,
  // so the 16 tap loads of a channel are unconditional (no exec-mask branches around them)
  const int yc0 = min(max(q.t[0].y0, 0), H - 1), yc1 = min(max(q.t[0].y0 + 1, 0), H - 1);
  q.row0 = (long)yc0 * W; q.row1 = (long)yc1 * W;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    q.xc0[i] = min(max(q.t[i].x0, 0), W - 1);
    q.xc1[i] = min(max(q.t[i].x0 + 1, 0), W - 1);
  }
  q.opix = (long)h * W + w0;
  return q;
}

// One channel of a quad. FLS >= 0: the channel's flags are a compile-time constant (the fused front end: every flag test
// folds away); FLS < 0: flags come from the plan at run time (the generic Z ops).
template <int FLS>
__device__ __forceinline__ void resample_channel4_vals(const float* __restrict__ s, float (&o)[4], float mean, int fl_dyn,
                                                       const Quad& q, const Affine& a, int H, int W) {
  const int fl = FLS >= 0 ? FLS : fl_dyn;
  float tl[4], tr[4], bl[4], br[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    tl[i] = s[q.row0 + q.xc0[i]];
    tr[i] = s[q.row0 + q.xc1[i]];
    bl[i] = s[q.row1 + q.xc0[i]];
    br[i] = s[q.row1 + q.xc1[i]];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const Taps& t = q.t[i];
    const float ptl = t.in00 ? pre_op(tl[i], mean, fl) : 0.f;
    const float ptr_ = t.in01 ? pre_op(tr[i], mean, fl) : 0.f;
    const float pbl = t.in10 ? pre_op(bl[i], mean, fl) : 0.f;
    const float pbr = t.in11 ? pre_op(br[i], mean, fl) : 0.f;
    float v = blend(ptl, ptr_, pbl, pbr, t.wy0, t.wx0);
    if (fl & CF_HIGHLIGHT) {
      const int sy0 = (int)floorf((float)H / 2 - 5), sy1 = (int)ceilf((float)H / 2 + 5);
      const int sx0 = (int)floorf((float)W / 2 - 5), sx1 = (int)ceilf((float)W / 2 + 5);
      const bool spot = q.h >= sy0 && q.h < sy1 && q.w0 + i >= sx0 && q.w0 + i < sx1;
      v = fmaxf(v, ((fl & CF_HIGHLIGHT_RED) && spot) ? 255.f : 0.f);
    }
    v = v - mean;
    if (fl & CF_POST_ROUND) v = roundf(v);
    if (fl & CF_POST_DIV255) v = div255(v);
    if (fl & CF_POST_MUL_WX) v = v * a.wx_in;
    if (fl & CF_POST_DIV_WX) v = v / a.wx_in;
    if (fl & CF_POST_ROUND_M045) v = roundf(v - 0.45f);
    o[i] = v;
  }
}

template <int FLS>
__device__ __forceinline__ void resample_channel4(const float* __restrict__ s, float* __restrict__ d, float mean, int fl_dyn,
                                                  const Quad& q, const Affine& a, int H, int W) {
  float o[4];
  resample_channel4_vals<FLS>(s, o, mean, fl_dyn, q, a, H, W);
  *reinterpret_cast<float4*>(d + q.opix) = make_float4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void resample4_kernel(Plan plan, const float* __restrict__ zoom_factor, int H, int W,
                                                        float gx_step, float gy_step) {
  const int qpr = W >> 2;
  const int qid = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (qid >= qpr * H) return;
  const int h = qid / qpr, w0 = (qid - h * qpr) << 2;
  const Affine a = load_affine(zoom_factor, b, plan.inverse, H, W);
  const Quad q = make_quad(a, h, w0, H, W, gx_step, gy_step);
#pragma unroll 2
  for (int c = 0; c < plan.n; ++c) {
    const Chan& ch = plan.ch[c];
    resample_channel4<-1>(ch.src + (long)b * ch.src_bstride, ch.dst + (long)b * ch.dst_bstride, ch.mean, ch.flags, q, a, H, W);
  }
}

// The fused front end of the test graph (deepIM_flownet.py:563-622 + :33-62: ZoomMask / ZoomImageWithFactor / ZoomDepth +
// /255 + Concat) with every channel's pre/post op known at compile time: the generic plan kernel spends more instructions
// on run-time flag tests and IEEE divisions than on the blend itself (85 VALU ops per output there).
struct ConcatArgs {
  const float* image_observed; const float* image_rendered;   // (B,3,H,W)
  const float* depth_observed; const float* depth_rendered;   // (B,1,H,W) or unused
  const float* mask_observed;  const float* mask_rendered;    // (B,1,H,W) or unused
  float* net_input;                                           // (B,C,H,W)
  Vec3 means;
  int C;
};
template <bool DEPTH, bool MASK>
__global__ __launch_bounds__(256) void zoom_concat4_kernel(ConcatArgs g, const float* __restrict__ zoom_factor, int H, int W,
                                                           float gx_step, float gy_step) {
  const int qpr = W >> 2;
  const int qid = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (qid >= qpr * H) return;
  const int h = qid / qpr, w0 = (qid - h * qpr) << 2;
  const Affine a = load_affine(zoom_factor, b, 0, H, W);
  const Quad q = make_quad(a, h, w0, H, W, gx_step, gy_step);
  const long p = (long)H * W;
  float* out = g.net_input + (long)b * g.C * p;
#pragma unroll
  for (int c = 0; c < 3; ++c)
    resample_channel4<CF_POST_DIV255>(g.image_observed + ((long)b * 3 + c) * p, out + c * p, g.means.v[c], 0, q, a, H, W);
#pragma unroll
  for (int c = 0; c < 3; ++c)
    resample_channel4<CF_POST_DIV255>(g.image_rendered + ((long)b * 3 + c) * p, out + (3 + c) * p, g.means.v[c], 0, q, a, H, W);
  int n = 6;
  if (DEPTH) {
    resample_channel4<CF_POST_DIV255>(g.depth_observed + (long)b * p, out + (long)n * p, 0.f, 0, q, a, H, W);
    resample_channel4<CF_POST_DIV255>(g.depth_rendered + (long)b * p, out + (long)(n + 1) * p, 0.f, 0, q, a, H, W);
    n += 2;
  }
  if (MASK) {
    resample_channel4<CF_POST_ROUND>(g.mask_observed + (long)b * p, out + (long)n * p, 0.f, 0, q, a, H, W);
    resample_channel4<CF_PRE_BIN02 | CF_POST_ROUND>(g.mask_rendered + (long)b * p, out + (long)(n + 1) * p, 0.f, 0, q, a, H, W);
  }
}

// The same front end for the shipped 8-channel input (RGB pair + masks) writing CHANNEL-BLOCKED records — [n][h][w][8], the
// "NC8" layout with C = 8 — so that conv1 runs on the 16-byte-load kernel of the other encoder layers instead of gathering
// dwords from eight NCHW planes. A thread owns 4 consecutive pixels of all 8 channels: 128 contiguous bytes, eight dwordx4
// stores; values identical to the NCHW kernel's (same resample_channel4_vals calls).
__global__ __launch_bounds__(256) void zoom_concat4_nc8_kernel(ConcatArgs g, const float* __restrict__ zoom_factor, int H, int W,
                                                               float gx_step, float gy_step) {
  const int qpr = W >> 2;
  const int qid = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (qid >= qpr * H) return;
  const int h = qid / qpr, w0 = (qid - h * qpr) << 2;
  const Affine a = load_affine(zoom_factor, b, 0, H, W);
  const Quad q = make_quad(a, h, w0, H, W, gx_step, gy_step);
  const long p = (long)H * W;
  float o[8][4];
#pragma unroll
  for (int c = 0; c < 3; ++c)
    resample_channel4_vals<CF_POST_DIV255>(g.image_observed + ((long)b * 3 + c) * p, o[c], g.means.v[c], 0, q, a, H, W);
#pragma unroll
  for (int c = 0; c < 3; ++c)
    resample_channel4_vals<CF_POST_DIV255>(g.image_rendered + ((long)b * 3 + c) * p, o[3 + c], g.means.v[c], 0, q, a, H, W);
  resample_channel4_vals<CF_POST_ROUND>(g.mask_observed + (long)b * p, o[6], 0.f, 0, q, a, H, W);
  resample_channel4_vals<CF_PRE_BIN02 | CF_POST_ROUND>(g.mask_rendered + (long)b * p, o[7], 0.f, 0, q, a, H, W);
  float4* rec = reinterpret_cast<float4*>(g.net_input + ((long)b * p + q.opix) * 8);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    rec[2 * i] = make_float4(o[0][i], o[1][i], o[2][i], o[3][i]);
    rec[2 * i + 1] = make_float4(o[4][i], o[5][i], o[6][i], o[7][i]);
  }
}

// The front end of the fp16 conv path (BASELINE config 5): the same values rounded once to fp16 (RNE, what the separate
// NCHW fp32 → NHWC fp16 pass did) and written as the pixel records conv1's patch kernel reads — main (B,H,W,8 halves) = the
// first eight net-input channels, and with DEPTH the two remaining ones (the masks) as (B,H,W,2 halves). Half the bytes of
// the fp32 net input on the way out, a quarter on conv1's way in.
template <bool DEPTH>
__global__ __launch_bounds__(256) void zoom_concat4_h16_kernel(ConcatArgs g, _Float16* __restrict__ main8,
                                                               _Float16* __restrict__ extra2,
                                                               const float* __restrict__ zoom_factor, int H, int W,
                                                               float gx_step, float gy_step) {
  constexpr int C = DEPTH ? 10 : 8;
  const int qpr = W >> 2;
  const int qid = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (qid >= qpr * H) return;
  const int h = qid / qpr, w0 = (qid - h * qpr) << 2;
  const Affine a = load_affine(zoom_factor, b, 0, H, W);
  const Quad q = make_quad(a, h, w0, H, W, gx_step, gy_step);
  const long p = (long)H * W;
  float o[C][4];
#pragma unroll
  for (int c = 0; c < 3; ++c)
    resample_channel4_vals<CF_POST_DIV255>(g.image_observed + ((long)b * 3 + c) * p, o[c], g.means.v[c], 0, q, a, H, W);
#pragma unroll
  for (int c = 0; c < 3; ++c)
    resample_channel4_vals<CF_POST_DIV255>(g.image_rendered + ((long)b * 3 + c) * p, o[3 + c], g.means.v[c], 0, q, a, H, W);
  if (DEPTH) {
    resample_channel4_vals<CF_POST_DIV255>(g.depth_observed + (long)b * p, o[6], 0.f, 0, q, a, H, W);
    resample_channel4_vals<CF_POST_DIV255>(g.depth_rendered + (long)b * p, o[7], 0.f, 0, q, a, H, W);
  }
  resample_channel4_vals<CF_POST_ROUND>(g.mask_observed + (long)b * p, o[C - 2], 0.f, 0, q, a, H, W);
  resample_channel4_vals<CF_PRE_BIN02 | CF_POST_ROUND>(g.mask_rendered + (long)b * p, o[C - 1], 0.f, 0, q, a, H, W);
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  h8v* rec = reinterpret_cast<h8v*>(main8 + ((long)b * p + q.opix) * 8);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h8v r;
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = (_Float16)o[c][i];
    rec[i] = r;
  }
  if (DEPTH) {
    h8v r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[2 * i] = (_Float16)o[8][i]; r[2 * i + 1] = (_Float16)o[9][i]; }
    *reinterpret_cast<h8v*>(extra2 + ((long)b * p + q.opix) * 2) = r;
  }
}

// every float bit pattern through div255 against the IEEE division (parity hook)
__global__ __launch_bounds__(256) void selfcheck_div255_kernel(unsigned long long* __restrict__ bad) {
  const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  unsigned long long n = 0;
  for (unsigned long long u = i; u < (1ull << 32); u += (unsigned long long)gridDim.x * 256) {
    const float v = __uint_as_float((unsigned)u);
    const float x = v / 255.0f, y = div255(v);
    if (__float_as_uint(x) != __float_as_uint(y) && !(x != x && y != y)) ++n;
  }
  if (n) atomicAdd(bad, n);
}

__global__ __launch_bounds__(256) void indices_kernel(int32_t* __restrict__ idx, const float* __restrict__ zoom_factor,
                                                      int H, int W, float gx_step, float gy_step) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  if (w >= W) return;
  const Affine a = load_affine(zoom_factor, b, 0, H, W);
  const Taps t = make_taps(a, h, w, H, W, gx_step, gy_step);
  const long plane = (long)H * W;
  idx[((long)b * 2 + 0) * plane + (long)h * W + w] = t.x0;
  idx[((long)b * 2 + 1) * plane + (long)h * W + w] = t.y0;
}

__global__ void inverse_factor_kernel(float* __restrict__ out, const float* __restrict__ zoom_factor, int B, int H, int W) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const Affine a = load_affine(zoom_factor, b, 1, H, W);
  out[b * 4 + 0] = a.wx; out[b * 4 + 1] = a.wy; out[b * 4 + 2] = a.tx; out[b * 4 + 3] = a.ty;
}

__global__ void zoom_trans_kernel(float* __restrict__ out, const float* __restrict__ zoom_factor,
                                  const float* __restrict__ in, int mul, int scale_xy, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float wx = zoom_factor[b * 4 + 0];  // the reference reads wy from column 0 too (zoom_trans.py:31)
  float x = in[b * 3 + 0], y = in[b * 3 + 1];
  if (scale_xy) {
    if (mul) { x = x * wx; y = y * wx; } else { x = x / wx; y = y / wx; }
  }
  out[b * 3 + 0] = x; out[b * 3 + 1] = y; out[b * 3 + 2] = in[b * 3 + 2];
}

int launch_resample(deepim_ctx* ctx, const Plan& plan, const float* zoom_factor, int B, int H, int W) {
  if (B == 0) return 0;
  const float gx = (float)(2.0 / (W - 1)), gy = (float)(2.0 / (H - 1));
  bool vec4 = (W & 3) == 0;
  for (int c = 0; c < plan.n && vec4; ++c)
    vec4 = (((size_t)plan.ch[c].dst & 15) == 0) && ((plan.ch[c].dst_bstride & 3) == 0);
  if (vec4) {
    dim3 grid(di_div_up((long)(W / 4) * H, 256), B);
    hipLaunchKernelGGL(resample4_kernel, grid, dim3(256), 0, ctx->stream, plan, zoom_factor, H, W, gx, gy);
  } else {
    dim3 grid(di_div_up(W, 256), H, B);
    hipLaunchKernelGGL(resample_kernel, grid, dim3(256), 0, ctx->stream, plan, zoom_factor, H, W, gx, gy);
  }
  DI_LAUNCH_CHECK();
  return 0;
}

Chan make_chan(const float* src, float* dst, long sb, long db, float mean, int flags) {
  Chan c; c.src = src; c.dst = dst; c.src_bstride = sb; c.dst_bstride = db; c.mean = mean; c.flags = flags;
  return c;
}

Mat3 mat3_from(const float* h) { Mat3 m; for (int i = 0; i < 9; ++i) m.v[i] = h[i]; return m; }

// boxes + factor; scratch layout: [B*2*4 int boxes]; status word lives in ctx->status
int compute_zoom_factor(deepim_ctx* ctx, float* zoom_factor, const float* map0, const float* map1, int mode0, int mode1,
                        const float* means3, const float* src_pose, const float* K_host, int B, int H, int W) {
  // up to DI_MAX_BOX_SAMPLES pairs the accumulators are the context's persistent, self-re-arming ones (no init launch: one of the
  // ~4 µs launches per refinement iteration that carry no work); larger batches arm a scratch copy as before
  const bool persistent = B <= DI_MAX_BOX_SAMPLES;
  int* box = ctx->zoom_box;
  if (!persistent) {
    void* scratch;
    int rc = deepim_scratch(ctx, (size_t)(B * 8) * sizeof(int), &scratch);
    if (rc) return rc;
    box = (int*)scratch;
    hipLaunchKernelGGL(bbox_init_kernel, dim3(di_div_up(B * 8, 256)), dim3(256), 0, ctx->stream, box, B * 8);
  }
  if (persistent && ctx->zoom_box_dirty) {   // an earlier call accumulated boxes that no zoom_factor_kernel consumed and re-armed
    hipLaunchKernelGGL(bbox_init_kernel, dim3(di_div_up(DI_MAX_BOX_SAMPLES * 8, 256)), dim3(256), 0, ctx->stream, box, DI_MAX_BOX_SAMPLES * 8);
    DI_LAUNCH_CHECK();
    ctx->zoom_box_dirty = 0;
  }
  int* status = ctx->status;
  Vec3 means = {{0, 0, 0}};
  if (means3) for (int i = 0; i < 3; ++i) means.v[i] = means3[i];
  dim3 grid(di_div_up(H, BB_ROWS), B, 2);
  if (persistent) ctx->zoom_box_dirty = 1;
  hipLaunchKernelGGL(bbox_kernel, grid, dim3(256), 0, ctx->stream, box, map0, map1, mode0, mode1, means, H, W);
  DI_LAUNCH_CHECK();
  hipLaunchKernelGGL(zoom_factor_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, zoom_factor, status, box,
                     src_pose, mat3_from(K_host), B, H, W, persistent ? 1 : 0);
  DI_LAUNCH_CHECK();
  if (persistent) ctx->zoom_box_dirty = 0;   // the consumer is queued: it resets what it reads
  return 0;
}

}  // namespace

extern "C" int deepim_zoom_mask_forward(deepim_ctx* ctx, const float* mask_observed, const float* mask_gt_observed,
                                        const float* mask_rendered, const float* src_pose, const float* K_host,
                                        float* zoom_mask_observed, float* zoom_mask_gt_observed,
                                        float* zoom_mask_rendered, float* zoom_factor, int B, int H, int W) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  int rc = compute_zoom_factor(ctx, zoom_factor, mask_gt_observed, mask_rendered, BB_MASK_GT, BB_MASK_RENDERED, nullptr,
                               src_pose, K_host, B, H, W);
  if (rc) return rc;
  const long p = (long)H * W;
  Plan plan; plan.n = 3; plan.inverse = 0;
  plan.ch[0] = make_chan(mask_observed, zoom_mask_observed, p, p, 0.f, CF_POST_ROUND);
  plan.ch[1] = make_chan(mask_gt_observed, zoom_mask_gt_observed, p, p, 0.f, CF_POST_ROUND);
  plan.ch[2] = make_chan(mask_rendered, zoom_mask_rendered, p, p, 0.f, CF_PRE_BIN02 | CF_POST_ROUND);
  return launch_resample(ctx, plan, zoom_factor, B, H, W);
}

static void image_plan(Plan& plan, int base, const float* src, float* dst, long sb, long db, long plane,
                       const float* means, int flags, int highlight) {
  for (int c = 0; c < 3; ++c) {
    int fl = flags;
    if (highlight) fl |= CF_HIGHLIGHT | (c == 0 ? CF_HIGHLIGHT_RED : 0);
    plan.ch[base + c] = make_chan(src + c * plane, dst + c * plane, sb, db, means[c], fl);
  }
}

extern "C" int deepim_zoom_image_forward(deepim_ctx* ctx, const float* image_observed, const float* image_rendered,
                                         const float* src_pose, const float* K_host, const float* pixel_means_host,
                                         float* zoom_image_observed, float* zoom_image_rendered, float* zoom_factor,
                                         int B, int H, int W) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  int rc = compute_zoom_factor(ctx, zoom_factor, image_observed, image_rendered, BB_IMAGE, BB_IMAGE, pixel_means_host,
                               src_pose, K_host, B, H, W);
  if (rc) return rc;
  const long p = (long)H * W;
  Plan plan; plan.n = 6; plan.inverse = 0;
  image_plan(plan, 0, image_observed, zoom_image_observed, 3 * p, 3 * p, p, pixel_means_host, 0, 0);
  image_plan(plan, 3, image_rendered, zoom_image_rendered, 3 * p, 3 * p, p, pixel_means_host, 0, 0);
  return launch_resample(ctx, plan, zoom_factor, B, H, W);
}

extern "C" int deepim_zoom_image_with_factor_forward(deepim_ctx* ctx, const float* zoom_factor,
                                                     const float* image_observed, const float* image_rendered,
                                                     const float* pixel_means_host, int high_light_center,
                                                     float* zoom_image_observed, float* zoom_image_rendered, int B,
                                                     int H, int W) {
  DI_DEVICE(ctx);
  const long p = (long)H * W;
  Plan plan; plan.n = 6; plan.inverse = 0;
  image_plan(plan, 0, image_observed, zoom_image_observed, 3 * p, 3 * p, p, pixel_means_host, 0, 0);
  image_plan(plan, 3, image_rendered, zoom_image_rendered, 3 * p, 3 * p, p, pixel_means_host, 0, high_light_center);
  return launch_resample(ctx, plan, zoom_factor, B, H, W);
}

extern "C" int deepim_zoom_depth_forward(deepim_ctx* ctx, const float* zoom_factor, const float* depth_observed,
                                         const float* depth_rendered, float* zoom_depth_observed,
                                         float* zoom_depth_rendered, int B, int H, int W) {
  DI_DEVICE(ctx);
  const long p = (long)H * W;
  Plan plan; plan.n = 2; plan.inverse = 0;
  plan.ch[0] = make_chan(depth_observed, zoom_depth_observed, p, p, 0.f, 0);
  plan.ch[1] = make_chan(depth_rendered, zoom_depth_rendered, p, p, 0.f, 0);
  return launch_resample(ctx, plan, zoom_factor, B, H, W);
}

extern "C" int deepim_zoom_flow_forward(deepim_ctx* ctx, const float* zoom_factor, const float* flow,
                                        const float* flow_weights, float* zoom_flow, float* zoom_flow_weights,
                                        int b_inv_zoom, int B, int H, int W) {
  DI_DEVICE(ctx);
  const long p = (long)H * W;
  Plan plan; plan.n = 2; plan.inverse = b_inv_zoom ? 1 : 0;
  const int fl = b_inv_zoom ? CF_POST_MUL_WX : CF_POST_DIV_WX;
  plan.ch[0] = make_chan(flow, zoom_flow, 2 * p, 2 * p, 0.f, fl);
  plan.ch[1] = make_chan(flow + p, zoom_flow + p, 2 * p, 2 * p, 0.f, fl);
  if (!b_inv_zoom) {
    DI_REQUIRE(flow_weights && zoom_flow_weights, "ZoomFlow: flow_weights required when b_inv_zoom is false");
    plan.ch[2] = make_chan(flow_weights, zoom_flow_weights, 2 * p, 2 * p, 0.f, CF_POST_ROUND_M045);
    plan.ch[3] = make_chan(flow_weights + p, zoom_flow_weights + p, 2 * p, 2 * p, 0.f, CF_POST_ROUND_M045);
    plan.n = 4;
  }
  return launch_resample(ctx, plan, zoom_factor, B, H, W);
}

extern "C" int deepim_zoom_mask_with_factor_forward(deepim_ctx* ctx, const float* zoom_factor, const float* mask,
                                                    float* zoom_mask, int b_inv_zoom, int B, int H, int W) {
  DI_DEVICE(ctx);
  const long p = (long)H * W;
  Plan plan; plan.n = 1; plan.inverse = b_inv_zoom ? 1 : 0;
  plan.ch[0] = make_chan(mask, zoom_mask, p, p, 0.f, CF_PRE_BIN02 | CF_POST_ROUND);
  return launch_resample(ctx, plan, zoom_factor, B, H, W);
}

// mask head test path, fused: sigmoid → (>0.2) → inverse zoom → round → round (deepIM_flownet.py:647-666)
extern "C" int deepim_mask_head_forward(deepim_ctx* ctx, float* mask_pred, float* prob, const float* logits,
                                        const float* zoom_factor, int B, int H, int W) {
  DI_DEVICE(ctx);
  const long p = (long)H * W;
  Plan plan; plan.n = 1; plan.inverse = 1;
  plan.ch[0] = make_chan(logits, mask_pred, p, p, 0.f, CF_PRE_SIGMOID | CF_PRE_BIN02 | CF_POST_ROUND);
  int rc = launch_resample(ctx, plan, zoom_factor, B, H, W);
  if (rc || !prob) return rc;
  return deepim_mask_logistic(ctx, prob, nullptr, logits, nullptr, 0.f, (size_t)B * p);
}

extern "C" int deepim_zoom_trans_forward(deepim_ctx* ctx, const float* zoom_factor, const float* trans_delta,
                                         float* zoom_trans_delta, int b_inv_zoom, int B) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  hipLaunchKernelGGL(zoom_trans_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, zoom_trans_delta,
                     zoom_factor, trans_delta, b_inv_zoom ? 1 : 0, 1, B);
  DI_LAUNCH_CHECK();
  return 0;
}
extern "C" int deepim_zoom_trans_backward(deepim_ctx* ctx, const float* zoom_factor, const float* out_grad,
                                          float* in_grad, int b_inv_zoom, int b_zoom_grad, int B) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  hipLaunchKernelGGL(zoom_trans_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, in_grad, zoom_factor,
                     out_grad, b_inv_zoom ? 1 : 0, b_zoom_grad ? 1 : 0, B);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_zoom_concat_forward(deepim_ctx* ctx, const float* image_observed, const float* image_rendered,
                                          const float* mask_observed, const float* mask_rendered,
                                          const float* depth_observed, const float* depth_rendered,
                                          const float* src_pose, const float* K_host, const float* pixel_means_host,
                                          float* net_input, float* zoom_factor, int B, int H, int W) {
  return deepim_zoom_concat_train_forward(ctx, image_observed, image_rendered, mask_observed, nullptr, mask_rendered,
                                          depth_observed, depth_rendered, src_pose, K_host, pixel_means_host, net_input,
                                          zoom_factor, B, H, W);
}

// the shipped 8-channel front end (masks, no depth) writing channel-blocked records (B,H,W,8) for conv1's NC8 kernel
extern "C" int deepim_zoom_concat_forward_nc8(deepim_ctx* ctx, const float* image_observed, const float* image_rendered,
                                              const float* mask_observed, const float* mask_rendered, const float* src_pose,
                                              const float* K_host, const float* pixel_means_host, float* net_input_nc8,
                                              float* zoom_factor, int B, int H, int W) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(mask_observed && mask_rendered, "zoom_concat_nc8: the 8-channel input needs both masks");
  DI_REQUIRE((W & 3) == 0 && ((size_t)net_input_nc8 & 15) == 0, "zoom_concat_nc8: W % 4 == 0 and a 16-byte aligned output");
  int rc = compute_zoom_factor(ctx, zoom_factor, mask_observed, mask_rendered, BB_MASK_GT, BB_MASK_RENDERED, nullptr, src_pose,
                               K_host, B, H, W);
  if (rc) return rc;
  ConcatArgs g;
  g.image_observed = image_observed; g.image_rendered = image_rendered;
  g.depth_observed = g.depth_rendered = nullptr;
  g.mask_observed = mask_observed; g.mask_rendered = mask_rendered;
  g.net_input = net_input_nc8; g.C = 8;
  for (int i = 0; i < 3; ++i) g.means.v[i] = pixel_means_host ? pixel_means_host[i] : 0.f;
  const float gx = (float)(2.0 / (W - 1)), gy = (float)(2.0 / (H - 1));
  dim3 grid(di_div_up((long)(W / 4) * H, 256), B);
  hipLaunchKernelGGL(zoom_concat4_nc8_kernel, grid, dim3(256), 0, ctx->stream, g, zoom_factor, H, W, gx, gy);
  DI_LAUNCH_CHECK();
  return 0;
}

// front end of the fp16 conv path: fp16 pixel records for deepim_conv1_f16_h16_forward (depth_* and extra2 both NULL or both set)
extern "C" int deepim_zoom_concat_forward_h16(deepim_ctx* ctx, const float* image_observed, const float* image_rendered,
                                              const float* mask_observed, const float* mask_rendered,
                                              const float* depth_observed, const float* depth_rendered, const float* src_pose,
                                              const float* K_host, const float* pixel_means_host, void* main8, void* extra2,
                                              float* zoom_factor, int B, int H, int W) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(mask_observed && mask_rendered, "zoom_concat_h16: needs both masks (8- or 10-channel input)");
  const bool dep = depth_observed != nullptr;
  DI_REQUIRE(dep == (depth_rendered != nullptr) && dep == (extra2 != nullptr), "zoom_concat_h16: depth pair and extra2 go together");
  DI_REQUIRE((W & 3) == 0 && ((size_t)main8 & 15) == 0 && ((size_t)extra2 & 15) == 0, "zoom_concat_h16: W % 4 == 0, 16-byte aligned outputs");
  int rc = compute_zoom_factor(ctx, zoom_factor, mask_observed, mask_rendered, BB_MASK_GT, BB_MASK_RENDERED, nullptr, src_pose,
                               K_host, B, H, W);
  if (rc) return rc;
  ConcatArgs g;
  g.image_observed = image_observed; g.image_rendered = image_rendered;
  g.depth_observed = depth_observed; g.depth_rendered = depth_rendered;
  g.mask_observed = mask_observed; g.mask_rendered = mask_rendered;
  g.net_input = nullptr; g.C = dep ? 10 : 8;
  for (int i = 0; i < 3; ++i) g.means.v[i] = pixel_means_host ? pixel_means_host[i] : 0.f;
  const float gx = (float)(2.0 / (W - 1)), gy = (float)(2.0 / (H - 1));
  dim3 grid(di_div_up((long)(W / 4) * H, 256), B);
  if (dep) hipLaunchKernelGGL(zoom_concat4_h16_kernel<true>, grid, dim3(256), 0, ctx->stream, g, (_Float16*)main8, (_Float16*)extra2, zoom_factor, H, W, gx, gy);
  else hipLaunchKernelGGL(zoom_concat4_h16_kernel<false>, grid, dim3(256), 0, ctx->stream, g, (_Float16*)main8, (_Float16*)nullptr, zoom_factor, H, W, gx, gy);
  DI_LAUNCH_CHECK();
  return 0;
}

// training graph (deepIM_flownet.py:392-412): ZoomMask takes the zoom region from mask_GT_observed (NULL = the test graph,
// where mask_gt_observed is mask_observed, :564)
extern "C" int deepim_zoom_concat_train_forward(deepim_ctx* ctx, const float* image_observed, const float* image_rendered,
                                                const float* mask_observed, const float* mask_gt_observed,
                                                const float* mask_rendered, const float* depth_observed,
                                                const float* depth_rendered, const float* src_pose, const float* K_host,
                                                const float* pixel_means_host, float* net_input, float* zoom_factor, int B,
                                                int H, int W) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE((mask_observed == nullptr) == (mask_rendered == nullptr), "zoom_concat: pass both masks or neither");
  const bool with_mask = mask_observed != nullptr;
  const long p = (long)H * W;
  const int C = 6 + (with_mask ? 2 : 0) + (depth_observed ? 2 : 0);
  int rc;
  if (with_mask)  // ZoomMask; test graph: mask_gt_observed ≡ mask_observed (deepIM_flownet.py:564)
    rc = compute_zoom_factor(ctx, zoom_factor, mask_gt_observed ? mask_gt_observed : mask_observed, mask_rendered, BB_MASK_GT,
                             BB_MASK_RENDERED, nullptr, src_pose, K_host, B, H, W);
  else            // ZoomImage: boxes of the non-black pixels (deepIM_flownet.py:594-605, zoom_image.py:31-37)
    rc = compute_zoom_factor(ctx, zoom_factor, image_observed, image_rendered, BB_IMAGE, BB_IMAGE, pixel_means_host,
                             src_pose, K_host, B, H, W);
  if (rc) return rc;
  const long db = (long)C * p;
  if ((W & 3) == 0 && ((size_t)net_input & 15) == 0) {   // the fused, compile-time-specialised front end
    ConcatArgs g;
    g.image_observed = image_observed; g.image_rendered = image_rendered;
    g.depth_observed = depth_observed; g.depth_rendered = depth_rendered;
    g.mask_observed = mask_observed; g.mask_rendered = mask_rendered;
    g.net_input = net_input; g.C = C;
    for (int i = 0; i < 3; ++i) g.means.v[i] = pixel_means_host ? pixel_means_host[i] : 0.f;
    const float gx = (float)(2.0 / (W - 1)), gy = (float)(2.0 / (H - 1));
    dim3 grid(di_div_up((long)(W / 4) * H, 256), B);
    const bool dep = depth_observed != nullptr;
#define DI_CONCAT(D, M) hipLaunchKernelGGL((zoom_concat4_kernel<D, M>), grid, dim3(256), 0, ctx->stream, g, zoom_factor, H, W, gx, gy)
    if (dep && with_mask) DI_CONCAT(true, true);
    else if (dep) DI_CONCAT(true, false);
    else if (with_mask) DI_CONCAT(false, true);
    else DI_CONCAT(false, false);
#undef DI_CONCAT
    DI_LAUNCH_CHECK();
    return 0;
  }
  Plan plan; plan.inverse = 0;
  int n = 0;
  image_plan(plan, 0, image_observed, net_input, 3 * p, db, p, pixel_means_host, CF_POST_DIV255, 0);
  image_plan(plan, 3, image_rendered, net_input + 3 * p, 3 * p, db, p, pixel_means_host, CF_POST_DIV255, 0);
  n = 6;
  if (depth_observed) {
    plan.ch[n] = make_chan(depth_observed, net_input + n * p, p, db, 0.f, CF_POST_DIV255); ++n;
    plan.ch[n] = make_chan(depth_rendered, net_input + n * p, p, db, 0.f, CF_POST_DIV255); ++n;
  }
  if (with_mask) {
    plan.ch[n] = make_chan(mask_observed, net_input + n * p, p, db, 0.f, CF_POST_ROUND); ++n;
    plan.ch[n] = make_chan(mask_rendered, net_input + n * p, p, db, 0.f, CF_PRE_BIN02 | CF_POST_ROUND); ++n;
  }
  plan.n = n;
  return launch_resample(ctx, plan, zoom_factor, B, H, W);
}

extern "C" int deepim_selfcheck_div255(deepim_ctx* ctx, unsigned long long* mismatches_host) {
  DI_DEVICE(ctx);
  void* scratch;
  int rc = deepim_scratch(ctx, 64, &scratch);
  if (rc) return rc;
  DI_CHECK(hipMemsetAsync(scratch, 0, 8, ctx->stream));
  hipLaunchKernelGGL(selfcheck_div255_kernel, dim3(4096), dim3(256), 0, ctx->stream, (unsigned long long*)scratch);
  DI_LAUNCH_CHECK();
  DI_CHECK(hipMemcpyAsync(mismatches_host, scratch, 8, hipMemcpyDeviceToHost, ctx->stream));
  DI_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int deepim_zoom_indices(deepim_ctx* ctx, const float* zoom_factor, int32_t* idx, int B, int H, int W) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  const float gx = (float)(2.0 / (W - 1)), gy = (float)(2.0 / (H - 1));
  dim3 grid(di_div_up(W, 256), H, B);
  hipLaunchKernelGGL(indices_kernel, grid, dim3(256), 0, ctx->stream, idx, zoom_factor, H, W, gx, gy);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_zoom_inverse_factor(deepim_ctx* ctx, const float* zoom_factor, float* inv_factor, int B, int H, int W) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  hipLaunchKernelGGL(inverse_factor_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, inv_factor, zoom_factor, B, H, W);
  DI_LAUNCH_CHECK();
  return 0;
}

// status word of the last zoom-factor computation: bit0 = an observed mask/image was empty
// (sticky until read; reading clears it)
extern "C" int deepim_zoom_status(deepim_ctx* ctx, int* status) {
  DI_DEVICE(ctx);
  DI_CHECK(hipMemcpyAsync(status, ctx->status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  DI_CHECK(hipMemsetAsync(ctx->status, 0, sizeof(int), ctx->stream));
  DI_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}
