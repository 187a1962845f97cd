// rot / trans FullyConnected + inverse ZoomTrans of ONE sample by one wavefront (deepIM_flownet.py:715-726, zoom_trans.py:22-46):
// shared by pose_head_kernel (csrc/fc.hip) and the fused pose tail (csrc/se3.hip) so that both add in the same order.
#pragma once
#include <hip/hip_runtime.h>

// feat: the sample's F fc7 outputs (global or LDS); returns the 7 se3 values in every lane
__device__ __forceinline__ void di_pose_head_wave(float (&o)[7], const float* feat, const float* __restrict__ w_rot,
                                                  const float* __restrict__ b_rot, const float* __restrict__ w_trans,
                                                  const float* __restrict__ b_trans, float wx, int F, int lane) {
  float acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int k = lane; k < F; k += 64) {
    const float x = feat[k];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = fmaf(x, w_rot[r * F + k], acc[r]);
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[4 + r] = fmaf(x, w_trans[r * F + k], acc[4 + r]);
  }
#pragma unroll
  for (int r = 0; r < 7; ++r)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off, 64);
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = acc[r] + b_rot[r];
  o[4] = (acc[4] + b_trans[0]) * wx;  // ZoomTrans b_inv_zoom=True (zoom_trans.py:34-37)
  o[5] = (acc[5] + b_trans[1]) * wx;
  o[6] = acc[6] + b_trans[2];
}
