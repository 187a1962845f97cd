// Dev probe: sustained fp32 MFMA rate (v_mfma_f32_32x32x2_f32, 4 independent accumulators per wave, no memory
// traffic) for ~ms-long kernels — the practical ceiling of the conv kernel at the clock the chip sustains.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int RANDOM>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  unsigned rs = threadIdx.x * 2654435761u + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (RANDOM) {  // fresh pseudo-random operands every 32 MFMAs (2 VALU per 2048 MFMA cycles)
      rs = rs * 1664525u + 1013904223u;
      a = __uint_as_float(0x3f800000u | (rs >> 9)) - 1.5f;
      b = __uint_as_float(0x3f800000u | ((rs * 747796405u) >> 9)) - 1.5f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rnd = 0; rnd < 2; ++rnd) for (int blocks : {256, 768}) for (int iters : {2000, 20000}) {
    auto run = [&]() { if (rnd) k<1><<<blocks, 256>>>(out, iters, 1.0f, 0.5f); else k<0><<<blocks, 256>>>(out, iters, 1.0f, 0.5f); };
    run(); hipDeviceSynchronize();
    hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("%s operands, blocks %d iters %d: %.3f ms  %.1f TFLOP/s\n", rnd ? "random  " : "constant", blocks, iters, ms, fl / ms / 1e9);
  }
  return 0;
}
