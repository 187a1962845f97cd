// F-group: depth warp / ground-truth flow kernels.
//   F1  lib/flow_c/gpu_flow_kernel.cu:32-69   (flow_kernel; authoritative for lib/flow_c)
//   F2  lib/pair_matching/flow.py:12-63        (calc_flow, float64 numpy variant)
//   F3  deepim/operator_py/flow_updater.py:42-102 (FlowUpdater, integer flow)
//   F4  lib/pair_matching/batch_updater_py_multi.py:255-265 (KT, mask from depth)
//
// All three are one-thread-per-pixel HBM-bound kernels: 4 B read (depth_src) + one
// 4 B gather (depth_tgt) + 12 B written per pixel = 20 B/px algorithmic.
// Layout: each thread owns 4 consecutive pixels of a row → dwordx4 loads/stores; the
// per-sample 3x4 transform is read through the scalar cache (wave-uniform).
// The library is built with -ffp-contract=off so the f32 operation order below is
// exactly the order of the reference expressions (no fused multiply-adds).
#include "common.h"
#include <limits.h>

namespace {

struct FlowOut { float dh, dw, valid; };

// gpu_flow_kernel.cu:40-66, one pixel
__device__ __forceinline__ FlowOut flow_pixel(int w, int h, float d_src, const float* __restrict__ KT,
                                              const Mat3& Kinv, const float* __restrict__ depth_tgt_b,
                                              int height, int width) {
  FlowOut o = {0.f, 0.f, 0.f};
  float x = ((float)w * Kinv.v[0] + (float)h * Kinv.v[1] + Kinv.v[2]) * d_src;
  float y = ((float)w * Kinv.v[3] + (float)h * Kinv.v[4] + Kinv.v[5]) * d_src;
  float z = d_src;
  if (d_src > 1E-3) {  // float vs double literal, as in the reference
    float x_proj = x * KT[0] + y * KT[1] + z * KT[2] + KT[3];
    float y_proj = x * KT[4] + y * KT[5] + z * KT[6] + KT[7];
    float z_proj = (float)((double)(x * KT[8] + y * KT[9] + z * KT[10] + KT[11]) + 1E-15);
    float w_proj = x_proj / z_proj;
    float h_proj = y_proj / z_proj;
    int w_proj_i = (int)roundf(w_proj);
    int h_proj_i = (int)roundf(h_proj);
    if (w_proj >= 0 && w_proj <= width - 1 && h_proj >= 0 && h_proj <= height - 1) {
      float d_tgt = depth_tgt_b[h_proj_i * width + w_proj_i];
      if (fabsf(z_proj - d_tgt) < 3E-3) {
        o.dh = h_proj - h;
        o.dw = w_proj - w;
        o.valid = 1.f;
      }
    }
  }
  return o;
}

// grid: (ceil(W/4/256)·H rows folded, B); thread → 4 consecutive pixels
__global__ __launch_bounds__(256) void flow_kernel(float* __restrict__ flow, float* __restrict__ valid,
                                                   const float* __restrict__ depth_src,
                                                   const float* __restrict__ depth_tgt,
                                                   const float* __restrict__ KT_all, Mat3 Kinv,
                                                   int height, int width) {
  const int b = blockIdx.y;
  const int quads_per_row = width >> 2;  // width % 4 == 0 path
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= quads_per_row * height) return;
  const int h = q / quads_per_row;
  const int w0 = (q - h * quads_per_row) << 2;
  const size_t plane = (size_t)height * width;
  const float* KT = KT_all + b * 12;
  const float* tgt = depth_tgt + (size_t)b * plane;
  const size_t pix = (size_t)h * width + w0;
  const float4 d = *reinterpret_cast<const float4*>(depth_src + (size_t)b * plane + pix);
  const float ds[4] = {d.x, d.y, d.z, d.w};
  float dh[4], dw[4], vv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    FlowOut o = flow_pixel(w0 + i, h, ds[i], KT, Kinv, tgt, height, width);
    dh[i] = o.dh; dw[i] = o.dw; vv[i] = o.valid;
  }
  float* f0 = flow + ((size_t)b * 2 + 0) * plane + pix;
  float* f1 = flow + ((size_t)b * 2 + 1) * plane + pix;
  *reinterpret_cast<float4*>(f0) = make_float4(dh[0], dh[1], dh[2], dh[3]);
  *reinterpret_cast<float4*>(f1) = make_float4(dw[0], dw[1], dw[2], dw[3]);
  *reinterpret_cast<float4*>(valid + (size_t)b * plane + pix) = make_float4(vv[0], vv[1], vv[2], vv[3]);
}

// generic-width fallback (one pixel per thread)
__global__ __launch_bounds__(256) void flow_kernel_scalar(float* __restrict__ flow, float* __restrict__ valid,
                                                          const float* __restrict__ depth_src,
                                                          const float* __restrict__ depth_tgt,
                                                          const float* __restrict__ KT_all, Mat3 Kinv,
                                                          int height, int width) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int plane = height * width;
  if (p >= plane) return;
  const int h = p / width, w = p - h * width;
  FlowOut o = flow_pixel(w, h, depth_src[(size_t)b * plane + p], KT_all + b * 12, Kinv,
                         depth_tgt + (size_t)b * plane, height, width);
  flow[((size_t)b * 2 + 0) * plane + p] = o.dh;
  flow[((size_t)b * 2 + 1) * plane + p] = o.dw;
  valid[(size_t)b * plane + p] = o.valid;
}

// F2: flow.py:12-63. float64 arithmetic like numpy: X = d·(Kinv·[w,h,1]) with Kinv f32→f64,
// Xp = KT(f32→f64)·[X;1]; np.round = half-to-even (rint); flow from UN-rounded projections.
struct CalcFlowOut { float f0, f1, vis; };
__device__ __forceinline__ CalcFlowOut calc_flow_pixel(int b, int p, const float* __restrict__ depth_src,
                                                       const float* __restrict__ depth_tgt,
                                                       const float* __restrict__ KT_all, const Mat3& Kinv, float thresh,
                                                       int standard_rep, int height, int width, float* dsf_out) {
  const int plane = height * width;
  const int h = p / width, w = p - h * width;
  const float* KT = KT_all + b * 12;
  const float dsf = depth_src[(size_t)b * plane + p];
  *dsf_out = dsf;
  const double d = (double)dsf;
  const double rx = (double)Kinv.v[0] * w + (double)Kinv.v[1] * h + (double)Kinv.v[2];
  const double ry = (double)Kinv.v[3] * w + (double)Kinv.v[4] * h + (double)Kinv.v[5];
  const double rz = (double)Kinv.v[6] * w + (double)Kinv.v[7] * h + (double)Kinv.v[8];
  const double X = d * rx, Y = d * ry, Z = d * rz;
  const double xp = (double)KT[0] * X + (double)KT[1] * Y + (double)KT[2] * Z + (double)KT[3];
  const double yp = (double)KT[4] * X + (double)KT[5] * Y + (double)KT[6] * Z + (double)KT[7];
  const double zp = (double)KT[8] * X + (double)KT[9] * Y + (double)KT[10] * Z + (double)KT[11];
  const double pz = zp + 1e-15;
  const double pw = xp / pz, ph = yp / pz;
  float vis = 0.f;
  if (dsf != 0.f) {
    const long pwr = (long)rint(pw), phr = (long)rint(ph);
    const bool within = pwr >= 0 && pwr < width && phr >= 0 && phr < height;
    const long pwc = pwr < 0 ? 0 : (pwr > width - 1 ? width - 1 : pwr);
    const long phc = phr < 0 ? 0 : (phr > height - 1 ? height - 1 : phr);
    const double dt = (double)depth_tgt[(size_t)b * plane + phc * width + pwc];
    if (within && fabs(dt - pz) < (double)thresh && fabs(dt) > 1e-10) vis = 1.f;
  }
  CalcFlowOut o = {0.f, 0.f, vis};
  if (vis == 1.f) {
    const float fw = (float)(pw - (double)w), fh = (float)(ph - (double)h);
    if (standard_rep) { o.f0 = fw; o.f1 = fh; } else { o.f0 = fh; o.f1 = fw; }
  }
  return o;
}

__global__ __launch_bounds__(256) void calc_flow_kernel(float* __restrict__ flow, float* __restrict__ visible,
                                                        const float* __restrict__ depth_src,
                                                        const float* __restrict__ depth_tgt,
                                                        const float* __restrict__ KT_all, Mat3 Kinv,
                                                        float thresh, int standard_rep, int height, int width) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int plane = height * width;
  if (p >= plane) return;
  float dsf;
  const CalcFlowOut o = calc_flow_pixel(b, p, depth_src, depth_tgt, KT_all, Kinv, thresh, standard_rep, height, width, &dsf);
  float2* fo = reinterpret_cast<float2*>(flow + ((size_t)b * plane + p) * 2);
  *fo = make_float2(o.f0, o.f1);
  visible[(size_t)b * plane + p] = o.vis;
}

// Loader-side flow labels, lib/utils/image.py:402-450 (get_pair_flow): calc_flow per pair, flow.transpose((2,0,1)) →
// (B,2,H,W), weights by TRAIN.FLOW_WEIGHT_TYPE — 0 'all' (ones), 1 'viz' (visible), 2 'valid' (depth_rendered == 0 or
// visible) — tiled over the two channels. One pass: 8 B read + 16 B written per pixel.
__global__ __launch_bounds__(256) void pair_flow_labels_kernel(float* __restrict__ flow, float* __restrict__ weights,
                                                               const float* __restrict__ depth_src,
                                                               const float* __restrict__ depth_tgt,
                                                               const float* __restrict__ KT_all, Mat3 Kinv, float thresh,
                                                               int standard_rep, int weight_type, int height, int width) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int plane = height * width;
  if (p >= plane) return;
  float dsf;
  const CalcFlowOut o = calc_flow_pixel(b, p, depth_src, depth_tgt, KT_all, Kinv, thresh, standard_rep, height, width, &dsf);
  const float wgt = weight_type == 0 ? 1.f : (weight_type == 1 ? o.vis : ((dsf == 0.f || o.vis == 1.f) ? 1.f : 0.f));
  const size_t o0 = ((size_t)b * 2 + 0) * plane + p, o1 = o0 + plane;
  flow[o0] = o.f0; flow[o1] = o.f1;
  weights[o0] = wgt; weights[o1] = wgt;
}

// image.py:381-387: mask_rendered = depth_rendered with every value > thresh set to 1 (smaller values are KEPT, not zeroed;
// ZoomMask binarises later)
__global__ __launch_bounds__(256) void depth_clip_mask_kernel(float* __restrict__ mask, const float* __restrict__ depth,
                                                              float thresh, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float d = depth[i]; mask[i] = d > thresh ? 1.f : d; }
}

// F3: flow_updater.py:42-102. f32 MXNet ops; R = f32(Kinv64·[w,h,1]); round = half away from zero.
struct Mat3d { double v[9]; };
__global__ __launch_bounds__(256) void flow_updater_kernel(float* __restrict__ flow, float* __restrict__ weights,
                                                           const float* __restrict__ depth_src,
                                                           const float* __restrict__ depth_tgt,
                                                           const float* __restrict__ KT_all, Mat3d Kinv,
                                                           float thresh, int wh_rep, int height, int width) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int plane = height * width;
  if (p >= plane) return;
  const int h = p / width, w = p - h * width;
  const float* T = KT_all + b * 12;
  const float d = depth_src[(size_t)b * plane + p];
  const float r0 = (float)(Kinv.v[0] * w + Kinv.v[1] * h + Kinv.v[2]);
  const float r1 = (float)(Kinv.v[3] * w + Kinv.v[4] * h + Kinv.v[5]);
  const float r2 = (float)(Kinv.v[6] * w + Kinv.v[7] * h + Kinv.v[8]);
  const float X = d * r0, Y = d * r1, Z = d * r2;
  const float wp = T[0] * X + T[1] * Y + T[2] * Z + T[3];
  const float hp = T[4] * X + T[5] * Y + T[6] * Z + T[7];
  const float zp = (T[8] * X + T[9] * Y + T[10] * Z + T[11]) + 1e-15f;
  float pwf = fminf(fmaxf(roundf(wp / zp), 0.f), (float)(width - 1));
  float phf = fminf(fmaxf(roundf(hp / zp), 0.f), (float)(height - 1));
  const int pw = (int)pwf, ph = (int)phf;
  const bool valid_src = d > 1e-10f;
  const float dm = depth_tgt[(size_t)b * plane + (size_t)ph * width + pw];
  const bool vis = fabsf(dm - zp) < thresh;
  const bool ok = valid_src && vis;
  const float wd = ok ? (float)(pw - w) : 0.f;
  const float hd = ok ? (float)(ph - h) : 0.f;
  const size_t o0 = ((size_t)b * 2 + 0) * plane + p, o1 = ((size_t)b * 2 + 1) * plane + p;
  flow[o0] = wh_rep ? wd : hd;
  flow[o1] = wh_rep ? hd : wd;
  weights[o0] = ok ? 1.f : 0.f;
  weights[o1] = ok ? 1.f : 0.f;
}

// calc_se3 (RT_transform.py:176-187 → projection.py:12-43, f32 results) then K·se3 (f32).
__global__ void calc_KT_kernel(float* __restrict__ KT, const float* __restrict__ pose_src,
                               const float* __restrict__ pose_tgt, Mat3 K, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* S = pose_src + b * 12;
  const float* T = pose_tgt + b * 12;
  // se3_inverse(src): Rinv = R^T ; tinv = -1 * (R^T · t)
  float Ri[9], ti[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Ri[i * 3 + j] = S[j * 4 + i];
  for (int i = 0; i < 3; ++i) ti[i] = -1.f * (Ri[i * 3 + 0] * S[3] + Ri[i * 3 + 1] * S[7] + Ri[i * 3 + 2] * S[11]);
  // se3_mul(tgt, inv)
  float M[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      M[i * 4 + j] = T[i * 4 + 0] * Ri[0 * 3 + j] + T[i * 4 + 1] * Ri[1 * 3 + j] + T[i * 4 + 2] * Ri[2 * 3 + j];
    M[i * 4 + 3] = (T[i * 4 + 0] * ti[0] + T[i * 4 + 1] * ti[1] + T[i * 4 + 2] * ti[2]) + T[i * 4 + 3];
  }
  float* O = KT + b * 12;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j)
      O[i * 4 + j] = K.v[i * 3 + 0] * M[0 * 4 + j] + K.v[i * 3 + 1] * M[1 * 4 + j] + K.v[i * 3 + 2] * M[2 * 4 + j];
}

__global__ __launch_bounds__(256) void depth_to_mask_kernel(float* __restrict__ mask, const float* __restrict__ depth,
                                                            float thresh, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mask[i] = depth[i] > thresh ? 1.f : 0.f;
}

// rectangle of a mask the way the reference draws it (lib/pair_matching/data_pair.py:94-105, lib/utils/image.py:363-372):
// x/y_start = first, x/y_end = last column/row holding a non-zero, filled [y_start:y_end, x_start:x_end] — numpy slices,
// so the last row and column stay 0. words: {xmin, xmax, ymin, ymax} per sample, armed to {INT_MAX,-1,INT_MAX,-1} by
// mask_box_arm_kernel (or by the project pass of the fused re-render) earlier in the same call, in stream order.
__global__ void mask_box_arm_kernel(int* __restrict__ words, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) words[i] = (i & 1) ? -1 : INT_MAX;
}
__global__ __launch_bounds__(256) void mask_bbox_kernel(int* __restrict__ words, const float* __restrict__ mask, int H,
                                                        int W) {
  const int b = blockIdx.y;
  const float* m = mask + (long)b * H * W;
  int xmin = INT_MAX, xmax = -1, ymin = INT_MAX, ymax = -1;
  const int y0 = blockIdx.x * 8;
  for (int i = threadIdx.x; i < 8 * W; i += 256) {
    const int y = y0 + i / W, x = i - (i / W) * W;
    if (y < H && m[(long)y * W + x] != 0.f) {
      xmin = min(xmin, x); xmax = max(xmax, x); ymin = min(ymin, y); ymax = max(ymax, y);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    xmin = min(xmin, __shfl_xor(xmin, off, 64)); xmax = max(xmax, __shfl_xor(xmax, off, 64));
    ymin = min(ymin, __shfl_xor(ymin, off, 64)); ymax = max(ymax, __shfl_xor(ymax, off, 64));
  }
  if ((threadIdx.x & 63) == 0 && xmax >= 0) {
    atomicMin(&words[b * 4 + 0], xmin); atomicMax(&words[b * 4 + 1], xmax);
    atomicMin(&words[b * 4 + 2], ymin); atomicMax(&words[b * 4 + 3], ymax);
  }
}
__global__ __launch_bounds__(256) void mask_box_fill_kernel(float* __restrict__ box, const int* __restrict__ words,
                                                            int* __restrict__ status, int H, int W) {
  const int b = blockIdx.y;
  const int xs = words[b * 4 + 0], xe = words[b * 4 + 1], ys = words[b * 4 + 2], ye = words[b * 4 + 3];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i == 0 && xe < 0) atomicOr(status, DI_STATUS_MASK_BOX_EMPTY);   // the reference's np.min of an empty array raises
  if (i >= (long)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (long)y * W);
  box[(long)b * H * W + i] = (xe >= 0 && y >= ys && y < ye && x >= xs && x < xe) ? 1.f : 0.f;
}

Mat3 load_mat3(const float* h) {
  Mat3 m;
  for (int i = 0; i < 9; ++i) m.v[i] = h[i];
  return m;
}

// 3x3 inverse in double (adjugate) — for intrinsics this matches numpy's LU result to f64 ulps
void inv3d(const float* K, double* o) {
  double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
  double A = e * i - f * h, Bc = -(d * i - f * g), C = d * h - e * g;
  double det = a * A + b * Bc + c * C;
  o[0] = A / det; o[1] = -(b * i - c * h) / det; o[2] = (b * f - c * e) / det;
  o[3] = Bc / det; o[4] = (a * i - c * g) / det; o[5] = -(a * f - c * d) / det;
  o[6] = C / det; o[7] = -(a * h - b * g) / det; o[8] = (a * e - b * d) / det;
}

}  // namespace

extern "C" int deepim_flow_forward(deepim_ctx* ctx, float* flow, float* valid, const float* depth_src,
                                   const float* depth_tgt, const float* KT, const float* Kinv_host,
                                   int B, int H, int W) {
  DI_DEVICE(ctx);
  DI_REQUIRE(B >= 0 && H > 0 && W > 0, "deepim_flow_forward: bad shape");
  if (B == 0) return 0;
  Mat3 Kinv = load_mat3(Kinv_host);
  if ((W & 3) == 0) {
    dim3 grid(di_div_up((long)(W / 4) * H, 256), B);
    hipLaunchKernelGGL(flow_kernel, grid, dim3(256), 0, ctx->stream, flow, valid, depth_src, depth_tgt, KT, Kinv, H, W);
  } else {
    dim3 grid(di_div_up((long)W * H, 256), B);
    hipLaunchKernelGGL(flow_kernel_scalar, grid, dim3(256), 0, ctx->stream, flow, valid, depth_src, depth_tgt, KT,
                       Kinv, H, W);
  }
  DI_LAUNCH_CHECK();
  return 0;
}

static int g_flow_status = 0;
extern "C" int deepim_flow_status(void) { return g_flow_status; }

// B2 drop-in body: host pointers, synchronous (gpu_flow_kernel.cu:82-148). A per-device
// context + device buffers are cached between calls instead of malloc/free per call.  Two symbols forward here:
// the C-linkage `_flow` below (ctypes / cgo-style binders) and the C++-linkage `_flow` of flow_cxx.cpp, which is
// the symbol the reference's own Cython extension links (gpu_flow.hpp:1-3 has no extern "C"; gpu_flow.pyx:13-16 is
// built language="c++", setup_linux.py:116-125).
extern "C" void deepim_flow_host(float* flow, float* valid, float* depth_src, float* depth_tgt, float* KT, float* Kinv,
                                 int batch_size, int height, int width, int device_id) {
  static deepim_ctx* ctxs[64] = {nullptr};
  static float* bufs[64] = {nullptr};
  static size_t caps[64] = {0};
  g_flow_status = -1;
  if (device_id < 0 || device_id >= 64) {
    deepim_set_error_msg("_flow: bad device id");
    fprintf(stderr, "_flow: %s\n", deepim_last_error());
    return;
  }
  if (!ctxs[device_id] && deepim_create(device_id, &ctxs[device_id]) != 0) {
    fprintf(stderr, "_flow: %s\n", deepim_last_error());
    return;
  }
  deepim_ctx* c = ctxs[device_id];
  if (hipSetDevice(device_id) != hipSuccess) {   // gpu_flow_kernel.cu:71-80 selects the device itself, so does the drop-in
    deepim_set_error_msg("_flow: hipSetDevice failed");
    fprintf(stderr, "_flow: %s\n", deepim_last_error());
    return;
  }
  const size_t plane = (size_t)height * width, n = (size_t)batch_size * plane;
  const size_t need = n * 5 + (size_t)batch_size * 12 + 16;  // src,tgt,flow(2),valid,KT
  if (need > caps[device_id]) {
    if (bufs[device_id]) hipFree(bufs[device_id]);
    bufs[device_id] = nullptr;
    caps[device_id] = 0;
    if (hipMalloc((void**)&bufs[device_id], need * sizeof(float)) != hipSuccess) {
      deepim_set_error_msg("_flow: hipMalloc failed");
      fprintf(stderr, "_flow: %s\n", deepim_last_error());
      return;
    }
    caps[device_id] = need;
  }
  float* d_src = bufs[device_id];
  float* d_tgt = d_src + n;
  float* d_flow = d_tgt + n;
  float* d_valid = d_flow + 2 * n;
  float* d_KT = d_valid + n;
  int rc = 0;
  rc |= hipMemcpyAsync(d_src, depth_src, n * 4, hipMemcpyHostToDevice, c->stream);
  rc |= hipMemcpyAsync(d_tgt, depth_tgt, n * 4, hipMemcpyHostToDevice, c->stream);
  rc |= hipMemcpyAsync(d_KT, KT, (size_t)batch_size * 48, hipMemcpyHostToDevice, c->stream);
  if (rc == 0) rc = deepim_flow_forward(c, d_flow, d_valid, d_src, d_tgt, d_KT, Kinv, batch_size, height, width);
  if (rc == 0) rc |= hipMemcpyAsync(flow, d_flow, 2 * n * 4, hipMemcpyDeviceToHost, c->stream);
  if (rc == 0) rc |= hipMemcpyAsync(valid, d_valid, n * 4, hipMemcpyDeviceToHost, c->stream);
  if (rc == 0) rc |= hipStreamSynchronize(c->stream);
  if (rc != 0) {
    if (deepim_last_error()[0] == 0) deepim_set_error_msg("_flow: HIP failure");
    fprintf(stderr, "_flow: HIP failure: %s\n", deepim_last_error());
    return;
  }
  g_flow_status = 0;
}

extern "C" void _flow(float* flow, float* valid, float* depth_src, float* depth_tgt, float* KT, float* Kinv,
                      int batch_size, int height, int width, int device_id) {
  deepim_flow_host(flow, valid, depth_src, depth_tgt, KT, Kinv, batch_size, height, width, device_id);
}

extern "C" int deepim_calc_flow_forward(deepim_ctx* ctx, float* flow, float* visible, const float* depth_src,
                                        const float* depth_tgt, const float* KT, const float* Kinv_host, float thresh,
                                        int standard_rep, int B, int H, int W) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  dim3 grid(di_div_up((long)W * H, 256), B);
  hipLaunchKernelGGL(calc_flow_kernel, grid, dim3(256), 0, ctx->stream, flow, visible, depth_src, depth_tgt, KT,
                     load_mat3(Kinv_host), thresh, standard_rep, H, W);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_pair_flow_labels(deepim_ctx* ctx, float* flow, float* flow_weights, const float* depth_rendered,
                                       const float* depth_observed, const float* pose_rendered,
                                       const float* pose_observed, const float* K_host, float thresh, int standard_rep,
                                       int weight_type, int B, int H, int W) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(weight_type >= 0 && weight_type <= 2, "pair_flow_labels: unknown FLOW_WEIGHT_TYPE");
  void* scratch;
  int rc = deepim_scratch(ctx, (size_t)B * 48, &scratch);
  if (rc) return rc;
  float* KT = (float*)scratch;
  rc = deepim_calc_KT(ctx, KT, pose_rendered, pose_observed, K_host, B);   // np.matmul(K, se3_mul(tgt, se3_inverse(src))), f32
  if (rc) return rc;
  double kinv[9];
  inv3d(K_host, kinv);                    // np.linalg.inv(np.matrix(K)): float32 result for a float32 K
  Mat3 Kinv;
  for (int i = 0; i < 9; ++i) Kinv.v[i] = (float)kinv[i];
  dim3 grid(di_div_up((long)W * H, 256), B);
  hipLaunchKernelGGL(pair_flow_labels_kernel, grid, dim3(256), 0, ctx->stream, flow, flow_weights, depth_rendered,
                     depth_observed, KT, Kinv, thresh, standard_rep, weight_type, H, W);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_depth_clip_mask(deepim_ctx* ctx, float* mask, const float* depth, float thresh, size_t n) {
  DI_DEVICE(ctx);
  if (n == 0) return 0;
  hipLaunchKernelGGL(depth_clip_mask_kernel, dim3(di_div_up((long)n, 256)), dim3(256), 0, ctx->stream, mask, depth, thresh,
                     n);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_calc_KT(deepim_ctx* ctx, float* KT, const float* pose_src, const float* pose_tgt,
                              const float* K_host, int B) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  hipLaunchKernelGGL(calc_KT_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, KT, pose_src, pose_tgt,
                     load_mat3(K_host), B);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_flow_updater_forward(deepim_ctx* ctx, float* flow, float* flow_weights, const float* depth_src,
                                           const float* depth_tgt, const float* pose_src, const float* pose_tgt,
                                           const float* K_host, float thresh, int wh_rep, int B, int H, int W) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  void* scratch;
  int rc = deepim_scratch(ctx, (size_t)B * 48, &scratch);
  if (rc) return rc;
  float* KT = (float*)scratch;
  rc = deepim_calc_KT(ctx, KT, pose_src, pose_tgt, K_host, B);
  if (rc) return rc;
  Mat3d Kinv;
  inv3d(K_host, Kinv.v);
  dim3 grid(di_div_up((long)W * H, 256), B);
  hipLaunchKernelGGL(flow_updater_kernel, grid, dim3(256), 0, ctx->stream, flow, flow_weights, depth_src, depth_tgt,
                     KT, Kinv, thresh, wh_rep, H, W);
  DI_LAUNCH_CHECK();
  return 0;
}

// second half of the rectangle op, shared with the fused re-render (csrc/render.hip accumulates the bbox itself)
int deepim_mask_box_fill(deepim_ctx* ctx, float* box, const int* words, int B, int H, int W) {
  hipLaunchKernelGGL(mask_box_fill_kernel, dim3(di_div_up((long)H * W, 256), B), dim3(256), 0, ctx->stream, box, words,
                     ctx->status, H, W);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_mask_box_forward(deepim_ctx* ctx, float* box, const float* mask, int B, int H, int W) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(B <= DI_MAX_BOX_SAMPLES, "mask_box: batch too large");
  int* words = ctx->box_words;
  hipLaunchKernelGGL(mask_box_arm_kernel, dim3(di_div_up(B * 4, 256)), dim3(256), 0, ctx->stream, words, B * 4);
  hipLaunchKernelGGL(mask_bbox_kernel, dim3(di_div_up(H, 8), B), dim3(256), 0, ctx->stream, words, mask, H, W);
  return deepim_mask_box_fill(ctx, box, words, B, H, W);
}

extern "C" int deepim_depth_to_mask(deepim_ctx* ctx, float* mask, const float* depth, float thresh, size_t n) {
  DI_DEVICE(ctx);
  if (n == 0) return 0;
  hipLaunchKernelGGL(depth_to_mask_kernel, dim3(di_div_up((long)n, 256)), dim3(256), 0, ctx->stream, mask, depth,
                     thresh, n);
  DI_LAUNCH_CHECK();
  return 0;
}
