// Dev probe: what the fp16 matrix pipe sustains with this kernel family's loop shapes (no global memory):
//   mode 0: MFMAs only (16 accumulators of a 4x4 wave tile, register operands)
//   mode 1: + the fragment reads of conv_f16_kernel (8 ds_read_b128 per 16 MFMAs, double-buffered)
//   mode 2: 2x4 wave tile (8 accumulators, 6 ds_read_b128 per 8 MFMAs)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_probe.hip -o /tmp/mfma_f16_probe && /tmp/mfma_f16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int TM, int TN, int LDS>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) h8 sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) {
    h8 v;
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)(((i * 8 + j + seed) * 2654435761u >> 20) * 1e-4f - 0.2f);
    sm[i] = v;
  }
  __syncthreads();
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const h8* as = sm + lane + (tid >> 6) * 256;
  const h8* bs = sm + 4096 + lane + (tid >> 6) * 256;
  h8 af[2][TM], bf[2][TN];
  for (int i = 0; i < TM; ++i) af[0][i] = as[i * 32];
  for (int j = 0; j < TN; ++j) bf[0][j] = bs[j * 32];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (LDS) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(t + 1) & 1][i] = as[((t + 1) & 3) * 512 + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[(t + 1) & 1][j] = bs[((t + 1) & 3) * 512 + j * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[LDS ? (t & 1) : 0][i], bf[LDS ? (t & 1) : 0][j], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}
static int g_lds = 131072;
template <int TM, int TN, int LDS>
void run(const char* name, float* out, int blocks) {
  hipFuncSetAttribute((const void*)k<TM, TN, LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, g_lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int iters : {2000, 20000}) {
    k<TM, TN, LDS><<<blocks, 256, g_lds>>>(out, iters, 1u); hipDeviceSynchronize();
    hipEventRecord(e0); k<TM, TN, LDS><<<blocks, 256, g_lds>>>(out, iters, 7u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 4 * TM * TN * 32768.0;
    printf("%-34s blocks %d iters %5d: %8.3f ms  %7.1f TFLOP/s\n", name, blocks, iters, ms, fl / ms / 1e9);
  }
}
int main(int argc, char** argv) {
  if (argc > 1) g_lds = atoi(argv[1]);
  printf("dynamic LDS %d bytes\n", g_lds);
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  run<4, 4, 0>("4x4 tile, MFMA only", out, 256);
  run<4, 4, 1>("4x4 tile, + 8 ds_read_b128/k-step", out, 256);
  run<2, 4, 0>("2x4 tile, MFMA only", out, 256);
  run<2, 4, 1>("2x4 tile, + 6 ds_read_b128/k-step", out, 256);
  run<2, 2, 0>("2x2 tile, MFMA only", out, 256);
  run<1, 2, 0>("1x2 tile, MFMA only", out, 256);
  run<1, 1, 0>("1x1 tile, MFMA only", out, 256);
  run<2, 2, 1>("2x2 tile, + 4 ds_read_b128/k-step", out, 256);
  run<2, 2, 1>("2x2 tile, 2 blocks/CU", out, 512);
  return 0;
}
