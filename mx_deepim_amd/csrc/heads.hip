// H-group: losses and small heads (HBM-bound elementwise + reductions).
//   H3 flow loss            deepIM_flownet.py:200-207
//   H4 mask logistic        deepIM_flownet.py:342-349 (MXNet LogisticRegressionOutput)
//   H6 point-matching loss  deepIM_flownet.py:265-312 (MXNet abs / square / smooth_l1 + MakeLoss)
//   GroupPicker             deepim/operator_py/group_picker.py:22-56
// Loss sums are deterministic: per-block partials (wave shuffles + LDS) then a
// single-block pass in fixed order — no float atomics.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float block_sum(float v) {
  __shared__ float red[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

constexpr int LOSS_BLOCKS = 1024;

// loss_type 0: |x|, 1: x², 2: smooth_l1(x, σ) (0.5(σx)² if |x| < 1/σ², else |x| − 0.5/σ²)
__device__ __forceinline__ void loss_fn(float x, int type, float sigma, float& f, float& df) {
  if (type == 0) { f = fabsf(x); df = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
  else if (type == 1) { f = x * x; df = 2.f * x; }
  else {
    const float s2 = sigma * sigma;
    if (fabsf(x) < 1.f / s2) { f = 0.5f * (sigma * x) * (sigma * x); df = s2 * x; }
    else { f = fabsf(x) - 0.5f / s2; df = x > 0.f ? 1.f : -1.f; }
  }
}

__global__ __launch_bounds__(256) void pm_loss_kernel(float* __restrict__ loss, float* __restrict__ partial,
                                                      float* __restrict__ d_est, const float* __restrict__ est,
                                                      const float* __restrict__ gt, const float* __restrict__ weights,
                                                      float normalize, int type, float sigma, float grad_scale,
                                                      size_t n) {
  float local = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float x = (est[i] - gt[i]) / normalize;
    float f, df;
    loss_fn(x, type, sigma, f, df);
    const float w = weights ? weights[i] : 1.f;
    const float l = w * f;
    loss[i] = l;
    local += l;
    if (d_est) d_est[i] = grad_scale * w * df / normalize;
  }
  const float s = block_sum(local);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void flow_loss_kernel(float* __restrict__ loss, float* __restrict__ partial,
                                                        float* __restrict__ d_est, const float* __restrict__ est,
                                                        const float* __restrict__ gt, const float* __restrict__ weights,
                                                        float normalize_flow, float grad_scale, size_t n) {
  float local = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = est[i] - gt[i] / normalize_flow;
    const float w = weights ? weights[i] : 1.f;
    const float l = w * (d * d);
    loss[i] = l;
    local += l;
    if (d_est) d_est[i] = grad_scale * w * 2.f * d;
  }
  const float s = block_sum(local);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void final_sum_kernel(float* __restrict__ out, const float* __restrict__ partial,
                                                        int n) {
  float local = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) local += partial[i];
  const float s = block_sum(local);
  if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(256) void logistic_kernel(float* __restrict__ prob, float* __restrict__ d_logits,
                                                       const float* __restrict__ logits, const float* __restrict__ label,
                                                       float grad_scale, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float p = 1.0f / (1.0f + expf(-logits[i]));
  prob[i] = p;
  if (d_logits) d_logits[i] = (p - label[i]) * grad_scale;
}

__global__ __launch_bounds__(256) void group_pick_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                         const float* __restrict__ group_idx, int* __restrict__ status,
                                                         int group_num, int C, int backward, int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cg = C / group_num;
  if (!backward) {  // out (B, cg) ← in (B, C)
    const int b = i / cg, c = i - b * cg;
    const int g = (int)group_idx[b];
    if (g < 0 || g >= group_num) { atomicOr(status, DI_STATUS_GROUP_RANGE); out[i] = 0.f; return; }
    out[i] = in[(long)b * C + g * cg + c];
  } else {          // out = in_grad (B, C) ← in = out_grad (B, cg)
    const int b = i / C, c = i - b * C;
    const int g = (int)group_idx[b];
    if (g < 0 || g >= group_num) { atomicOr(status, DI_STATUS_GROUP_RANGE); out[i] = 0.f; return; }
    out[i] = (c >= g * cg && c < (g + 1) * cg) ? in[(long)b * cg + (c - g * cg)] : 0.f;
  }
}

// one thread per row (D is 3 or 4 here)
__global__ void l2norm_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ d_out,
                              int B, int D, float eps, int backward) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* x = in + (long)b * D;
  float ss = 0.f;
  for (int i = 0; i < D; ++i) ss += x[i] * x[i];
  const float nrm = sqrtf(ss + eps);
  if (!backward) {
    for (int i = 0; i < D; ++i) out[(long)b * D + i] = x[i] / nrm;
  } else {  // d_x = (d_y − y·Σ(d_y·y)) / norm
    const float* g = d_out + (long)b * D;
    float dot = 0.f;
    for (int i = 0; i < D; ++i) dot += g[i] * (x[i] / nrm);
    for (int i = 0; i < D; ++i) out[(long)b * D + i] = (g[i] - (x[i] / nrm) * dot) / nrm;
  }
}

__global__ void rot_dist_kernel(float* __restrict__ loss, float* __restrict__ d_q, const float* __restrict__ q_gt,
                                const float* __restrict__ q_est, float grad_scale, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* g = q_gt + b * 4;
  const float* e = q_est + b * 4;
  const float dot = ((g[0] * e[0] + g[1] * e[1]) + g[2] * e[2]) + g[3] * e[3];
  loss[b] = 1.f - dot * dot;
  if (d_q)
    for (int i = 0; i < 4; ++i) d_q[b * 4 + i] = -2.f * dot * g[i] * grad_scale;
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha,
                                                   size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = y[i] + alpha * x[i];
}

}  // namespace

extern "C" int deepim_l2_normalize_forward(deepim_ctx* ctx, float* out, const float* in, int B, int D, float eps) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  hipLaunchKernelGGL(l2norm_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, out, in, nullptr, B, D, eps, 0);
  DI_LAUNCH_CHECK();
  return 0;
}
extern "C" int deepim_l2_normalize_backward(deepim_ctx* ctx, float* d_in, const float* d_out, const float* in, int B,
                                            int D, float eps) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  hipLaunchKernelGGL(l2norm_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, d_in, in, d_out, B, D, eps, 1);
  DI_LAUNCH_CHECK();
  return 0;
}
extern "C" int deepim_rot_dist_loss(deepim_ctx* ctx, float* loss, float* d_q_est, const float* q_gt, const float* q_est,
                                    float grad_scale, int B) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  hipLaunchKernelGGL(rot_dist_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, loss, d_q_est, q_gt, q_est,
                     grad_scale, B);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_axpy(deepim_ctx* ctx, float* y, const float* x, float alpha, size_t n) {
  DI_DEVICE(ctx);
  if (n == 0) return 0;
  hipLaunchKernelGGL(axpy_kernel, dim3(di_div_up((long)n, 256)), dim3(256), 0, ctx->stream, y, x, alpha, n);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_point_matching_loss(deepim_ctx* ctx, float* loss, float* loss_sum, float* d_est,
                                          const float* est, const float* gt, const float* weights, float normalize,
                                          int loss_type, float sigma, float grad_scale, int B, int N) {
  DI_DEVICE(ctx);
  const size_t n = (size_t)B * 3 * N;
  if (n == 0) return 0;
  DI_REQUIRE(loss_type >= 0 && loss_type <= 2, "point_matching_loss: unknown loss type");
  void* scratch;
  int rc = deepim_scratch(ctx, LOSS_BLOCKS * sizeof(float), &scratch);
  if (rc) return rc;
  const int blocks = (int)((n + 255) / 256 < LOSS_BLOCKS ? (n + 255) / 256 : LOSS_BLOCKS);
  hipLaunchKernelGGL(pm_loss_kernel, dim3(blocks), dim3(256), 0, ctx->stream, loss, (float*)scratch, d_est, est, gt,
                     weights, normalize, loss_type, sigma, grad_scale, n);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, loss_sum, (const float*)scratch, blocks);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_flow_loss(deepim_ctx* ctx, float* loss, float* loss_sum, float* d_est, const float* est,
                                const float* gt, const float* weights, float normalize_flow, float grad_scale,
                                size_t n) {
  DI_DEVICE(ctx);
  if (n == 0) return 0;
  void* scratch;
  int rc = deepim_scratch(ctx, LOSS_BLOCKS * sizeof(float), &scratch);
  if (rc) return rc;
  const int blocks = (int)((n + 255) / 256 < LOSS_BLOCKS ? (n + 255) / 256 : LOSS_BLOCKS);
  hipLaunchKernelGGL(flow_loss_kernel, dim3(blocks), dim3(256), 0, ctx->stream, loss, (float*)scratch, d_est, est, gt,
                     weights, normalize_flow, grad_scale, n);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, loss_sum, (const float*)scratch, blocks);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_mask_logistic(deepim_ctx* ctx, float* prob, float* d_logits, const float* logits,
                                    const float* label, float grad_scale, size_t n) {
  DI_DEVICE(ctx);
  if (n == 0) return 0;
  DI_REQUIRE(!d_logits || label, "mask_logistic: label required for the gradient");
  hipLaunchKernelGGL(logistic_kernel, dim3(di_div_up((long)n, 256)), dim3(256), 0, ctx->stream, prob, d_logits, logits,
                     label, grad_scale, n);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_group_picker_forward(deepim_ctx* ctx, float* out, const float* in, const float* group_idx,
                                           int group_num, int B, int C) {
  DI_DEVICE(ctx);
  DI_REQUIRE(group_num > 0 && C % group_num == 0, "GroupPicker: channels not divisible by group_num");
  const int total = B * (C / group_num);
  if (total == 0) return 0;
  hipLaunchKernelGGL(group_pick_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, out, in, group_idx,
                     ctx->status, group_num, C, 0, total);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_group_picker_backward(deepim_ctx* ctx, float* in_grad, const float* out_grad,
                                            const float* group_idx, int group_num, int B, int C) {
  DI_DEVICE(ctx);
  DI_REQUIRE(group_num > 0 && C % group_num == 0, "GroupPicker: channels not divisible by group_num");
  const int total = B * C;
  if (total == 0) return 0;
  hipLaunchKernelGGL(group_pick_kernel, dim3(di_div_up(total, 256)), dim3(256), 0, ctx->stream, in_grad, out_grad,
                     group_idx, ctx->status, group_num, C, 1, total);
  DI_LAUNCH_CHECK();
  return 0;
}
