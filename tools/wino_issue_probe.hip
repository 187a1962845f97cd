// Dev probe for csrc/wino.hip: does a second wave per SIMD hide the VALU / load issue that a single wave adds to its fp32 MFMAs?
// One v_mfma_f32_32x32x2_f32 per slot on NACC rotating accumulators; per slot NV independent fp32 adds and (LD) a b64 buffer load
// every other slot from an L2-resident buffer. NACC = 16 -> 256 AGPRs, one wave per SIMD; NACC = 8 -> 128, two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/wino_issue_probe.hip -o tools/wino_issue_probe.bin && tools/wino_issue_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NV, int LD, int PK = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NACC == 16 ? 1 : 2, NACC == 16 ? 1 : 2)))
void k(float* out, const float* src, int iters, unsigned nbytes) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63;
  f32x2 a[8], v[16];
  float t[8];
  for (int i = 0; i < 8; ++i) { a[i] = (f32x2){1.0f + i * 0.01f, 0.5f + lane * 1e-3f}; t[i] = lane * 0.25f + i; }
  for (int i = 0; i < 16; ++i) v[i] = (f32x2){0.3f + i * 0.02f, 0.7f - lane * 1e-3f};
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)nbytes, 0x00020000);
  const int voff = ((threadIdx.x * 8 + blockIdx.x * 4096) & (nbytes - 1));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int q = u % NACC;
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32((u & 1) ? a[q & 7].y : a[q & 7].x, (u & 1) ? v[u & 15].y : v[u & 15].x, acc[q], 0, 0, 0);
      asm volatile("" : "+a"(acc[q]));
      if (PK) {   // the same adds as packed pairs: NV / 2 v_pk_add_f32
#pragma unroll
        for (int n = 0; n < NV / 2; ++n) v[(u + 8 + n) & 15] = v[(u + 8 + n) & 15] + v[(u + 11 + n) & 15];
      } else {
#pragma unroll
        for (int n = 0; n < NV; ++n) t[(u + n) & 7] = t[(u + n) & 7] + t[(u + n + 3) & 7];
      }
      if (LD == 2 && (u & 3) == 3) {   // the same bytes as one b128 load every fourth slot
        const f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff * 2, ((it * 8 + (u >> 2)) * 1024) & 0xffff, 0));
        a[(u >> 2) & 7] = w.xy; a[((u >> 2) + 1) & 7] = w.zw;
      }
      if (LD == 1 && (u & 1)) a[(u >> 1) & 7] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, ((it * 16 + (u >> 1)) * 512) & 0xffff, 0));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += t[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int NV, int LD, int PK = 0>
static void run(const char* name, float* out, float* src, unsigned nbytes) {
  const int iters = 2000, grid = 256 * (NACC == 16 ? 1 : 2) * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, NV, LD, PK>), dim3(grid), dim3(256), 0, 0, out, src, iters, nbytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * 4 * iters * 32 * 4096.0;
  printf("%-44s %7.3f ms  %6.1f TF (%.3f of 157.3)\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
}

int main() {
  float *out, *src; const unsigned nbytes = 1u << 22;
  hipMalloc(&out, 256 * 8 * 256 * 4 * 4); hipMalloc(&src, nbytes); hipMemset(src, 0, nbytes);
  run<16, 0, 0>("1 wave/SIMD, MFMA only", out, src, nbytes);
  run<16, 2, 0>("1 wave/SIMD, + 2 VALU per slot", out, src, nbytes);
  run<16, 4, 0>("1 wave/SIMD, + 4 VALU per slot", out, src, nbytes);
  run<16, 0, 1>("1 wave/SIMD, + b64 load every other slot", out, src, nbytes);
  run<16, 2, 1>("1 wave/SIMD, + 2 VALU + load", out, src, nbytes);
  run<16, 4, 0, 1>("1 wave/SIMD, + 4 adds as 2 v_pk_add_f32 per slot", out, src, nbytes);
  run<16, 0, 2>("1 wave/SIMD, + b128 load every fourth slot", out, src, nbytes);
  run<16, 4, 2, 1>("1 wave/SIMD, + 2 pk adds + b128 load / 4 slots", out, src, nbytes);
  run<8, 0, 0>("2 waves/SIMD, MFMA only", out, src, nbytes);
  run<8, 2, 0>("2 waves/SIMD, + 2 VALU per slot", out, src, nbytes);
  run<8, 4, 0>("2 waves/SIMD, + 4 VALU per slot", out, src, nbytes);
  run<8, 0, 1>("2 waves/SIMD, + b64 load every other slot", out, src, nbytes);
  run<8, 2, 1>("2 waves/SIMD, + 2 VALU + load", out, src, nbytes);
  return 0;
}
