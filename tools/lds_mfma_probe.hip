// Dev probe: fp32 MFMA loop fed from LDS, no global traffic — isolates the cost of the fragment-read pattern.
//   variant 0: ds_read2_b32 per k-step (current conv layout [k][m])
//   variant 1: ds_read_b128 per 4 k-steps (layout [k/8][k&1][m][4])
//   variant 2: no LDS at all (register operands)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int VAR>
__global__ __launch_bounds__(256) void k(float* out, int chunks) {
  __shared__ __attribute__((aligned(16))) float As[2][16 * 128];
  __shared__ __attribute__((aligned(16))) float Bs[2][16 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 16 * 128; i += 256) { (&As[0][0])[i] = (float)((i * 37) % 101) * 0.01f - 0.5f; (&Bs[0][0])[i] = (float)((i * 53) % 97) * 0.01f - 0.5f; }
  __syncthreads();
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64, lrow = lane >> 5, lcol = lane & 31;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int c = 0; c < chunks; ++c) {
    const int buf = c & 1;
    if (VAR == 0) {
      const float* as = &As[buf][lrow * 128 + wm0 + lcol];
      const float* bs = &Bs[buf][lrow * 128 + wn0 + lcol];
      float av[2][2], bv[2][2];
      for (int i = 0; i < 2; ++i) { av[0][i] = as[i * 32]; bv[0][i] = bs[i * 32]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][m >> 1], bv[s & 1][m & 1], acc[m >> 1][m & 1], 0, 0, 0);
          if (m == 0 && s + 1 < 8) for (int i = 0; i < 2; ++i) { av[(s + 1) & 1][i] = as[(s + 1) * 256 + i * 32]; bv[(s + 1) & 1][i] = bs[(s + 1) * 256 + i * 32]; }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if (VAR == 1) {
      // [u][h][m][4]: lane (m, h) reads one 16-byte quad per u: k = 8u + 2*slot + h
      const float4* a4 = reinterpret_cast<const float4*>(&As[buf][0]);
      const float4* b4 = reinterpret_cast<const float4*>(&Bs[buf][0]);
      float4 av[2][2], bv[2][2];
      for (int i = 0; i < 2; ++i) { av[0][i] = a4[(0 * 2 + lrow) * 128 + wm0 + i * 32 + lcol]; bv[0][i] = b4[(0 * 2 + lrow) * 128 + wn0 + i * 32 + lcol]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const float a = sl == 0 ? av[u][m >> 1].x : sl == 1 ? av[u][m >> 1].y : sl == 2 ? av[u][m >> 1].z : av[u][m >> 1].w;
            const float b = sl == 0 ? bv[u][m & 1].x : sl == 1 ? bv[u][m & 1].y : sl == 2 ? bv[u][m & 1].z : bv[u][m & 1].w;
            acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m >> 1][m & 1], 0, 0, 0);
            if (u == 0 && sl == 0 && m == 0) for (int i = 0; i < 2; ++i) { av[1][i] = a4[(1 * 2 + lrow) * 128 + wm0 + i * 32 + lcol]; bv[1][i] = b4[(1 * 2 + lrow) * 128 + wn0 + i * 32 + lcol]; }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    } else {
      float a = As[0][tid], b = Bs[0][tid];
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m >> 1][m & 1], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}
template <int VAR> void run(float* out, int blocks, int chunks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<VAR><<<blocks, 256>>>(out, chunks); hipDeviceSynchronize();
  hipEventRecord(e0); k<VAR><<<blocks, 256>>>(out, chunks); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("variant %d blocks %d: %.3f ms %.1f TFLOP/s\n", VAR, blocks, ms, (double)blocks * 4 * chunks * 32 * 4096.0 / ms / 1e9);
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  for (int blocks : {256, 768, 3072}) { run<0>(out, blocks, 2000); run<1>(out, blocks, 2000); run<2>(out, blocks, 2000); }
  return 0;
}
