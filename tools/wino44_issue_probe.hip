// Dev probe for profiles/r06_wino44.md: what K-loop rate would an fp32 Winograd F(4x4,3x3) kernel reach on gfx950, given the per-step
// instruction mix its design needs? Same skeleton as conv_wino8_kernel (8 waves per block, two per SIMD, one barrier per step, V handed
// over through LDS, weights straight from an L2-resident buffer), with the mix as template parameters:
//   NM MFMAs (v_mfma_f32_32x32x2_f32 on NACC accumulators) + NV fp32 VALU + NPL b64 pixel loads + NW ds_write_b64 + NRA global b128 +
//   NRB ds_read_b128 per wave and step.
//   F(2x2,3x3) as shipped (calibration):  NACC 8, NM 32, NV 16, NPL 4, NW 4, NRA 8, NRB 8   -> the real kernel's loop runs at 0.90
//   F(4x4,3x3), 64 ch x 32 tiles blocks:  NACC 9, NM 36, NV 72, NPL 9, NW 9, NRA 9, NRB 9
// hipcc --offload-arch=gfx950 -O3 tools/wino44_issue_probe.hip -o tools/wino44_issue_probe.bin && tools/wino44_issue_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NM, int NV, int NPL, int NW, int NRA, int NRB, int WPS = 2>
__global__ __launch_bounds__(WPS == 2 ? 512 : 256) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
void k(float* out, const float* wsrc, const float* psrc, int steps, unsigned wbytes, unsigned pbytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NA_ = NRA > 0 ? NRA : 1, NB_ = NRB > 0 ? 3 : 1, NP_ = NPL > 0 ? 4 : 1;   // B operands and pixels: short rings (read right before use)
  f32x4 A[NA_], Bv[NB_];
  f32x2 px[NP_];
  float t[12];
  for (int i = 0; i < 12; ++i) t[i] = lane * 0.25f + i;
  for (int i = 0; i < NA_; ++i) A[i] = (f32x4){1.f, 0.5f, 0.25f, 0.125f};
  for (int i = 0; i < NB_; ++i) Bv[i] = (f32x4){0.3f, 0.2f, 0.1f, 0.05f};
  for (int i = 0; i < NP_; ++i) px[i] = (f32x2){0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, (int)wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)psrc, 0, (int)pbytes, 0x00020000);
  const int vw = lane * 16 + wave * 1024, vp = (threadIdx.x * 8 + blockIdx.x * 4096) & (pbytes - 1);
  const unsigned lw = (threadIdx.x & 511) * 8, lr = (wave & 3) * 9216 + lane * 16;
  constexpr int SLOTS = NM;
  for (int s = 0; s < steps; ++s) {
    const int so = (s * 16384) & (wbytes - 1) & ~16383;
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
      const int q = u % NACC, e = (u / NACC) & 3;
      const float a = e == 0 ? A[q % NA_].x : e == 1 ? A[q % NA_].y : e == 2 ? A[q % NA_].z : A[q % NA_].w;
      const float b = e == 0 ? Bv[q % NB_].x : e == 1 ? Bv[q % NB_].y : e == 2 ? Bv[q % NB_].z : Bv[q % NB_].w;
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
      asm volatile("" : "+a"(acc[q]));
      // the other roles, spread evenly over the MFMA slots
#pragma unroll
      for (int n = 0; n < (NV * (u + 1)) / SLOTS - (NV * u) / SLOTS; ++n) {
        const int i = (NV * u) / SLOTS + n;
        t[i % 12] = fmaf(t[(i + 5) % 12], -4.f, t[(i + 7) % 12]);
      }
      if (NPL && (NPL * (u + 1)) / SLOTS != (NPL * u) / SLOTS) {
        const int i = (NPL * u) / SLOTS;
        t[i % 12] += px[i % NP_].x + px[i % NP_].y;                                         // consume the previous step's pixel
        px[i % NP_] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rp, vp, (s * 64 + i * 4096) & (pbytes - 1) & ~63, 0));
      }
      if (NW && (NW * (u + 1)) / SLOTS != (NW * u) / SLOTS) {
        const int i = (NW * u) / SLOTS;
        f32x2 v; v.x = t[i % 12]; v.y = t[(i + 1) % 12];
        *reinterpret_cast<f32x2*>(smem + ((s & 1) * 36864 + lw + (i % 9) * 4096)) = v;
      }
      if (NRA && (NRA * (u + 1)) / SLOTS != (NRA * u) / SLOTS) {
        const int i = (NRA * u) / SLOTS;
        A[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, vw, so + i * 1024 * 8, 0));
      }
      if (NRB && (NRB * (u + 1)) / SLOTS != (NRB * u) / SLOTS) {
        const int i = (NRB * u) / SLOTS;
        Bv[i % NB_] = *reinterpret_cast<f32x4*>(smem + (((s + 1) & 1) * 36864 + lr + (i % 9) * 1024));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  float sum = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
  for (int i = 0; i < 12; ++i) sum += t[i];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}

template <int NACC, int NM, int NV, int NPL, int NW, int NRA, int NRB, int WPS = 2>
static double run(const char* name, float* out, float* w, float* p, unsigned wb, unsigned pb) {
  const int steps = 2048, grid = 256 * 4;
  hipFuncSetAttribute((const void*)k<NACC, NM, NV, NPL, NW, NRA, NRB, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, NM, NV, NPL, NW, NRA, NRB, WPS>), dim3(grid), dim3(WPS == 2 ? 512 : 256), 73728, 0, out, w, p, steps, wb, pb);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * (WPS == 2 ? 8 : 4) * steps * NM * 4096.0, tf = fl / ms / 1e9;
  printf("%-72s %8.3f ms  %6.1f TF  %.3f of 157.3\n", name, ms, tf, tf / 157.3);
  return tf / 157.3;
}

int main() {
  float *out, *w, *p; const unsigned wb = 1u << 22, pb = 1u << 24;
  hipMalloc(&out, 256 * 4 * 512 * 4); hipMalloc(&w, wb); hipMalloc(&p, pb); hipMemset(w, 0, wb); hipMemset(p, 0, pb);
  run<8, 32, 0, 0, 0, 0, 0>("F(2,3) shape: 32 MFMAs only (8 acc)", out, w, p, wb, pb);
  const double c23 = run<8, 32, 16, 4, 4, 8, 8>("F(2,3) as shipped: +16 VALU +4 px +4 dsw +8 A +8 B per step", out, w, p, wb, pb);
  run<9, 36, 0, 0, 0, 0, 0>("F(4,3) shape: 36 MFMAs only (9 acc = 144 AGPRs)", out, w, p, wb, pb);
  run<9, 36, 0, 0, 0, 9, 9>("F(4,3): + operand reads only (9 A + 9 B)", out, w, p, wb, pb);
  const double c43 = run<9, 36, 72, 9, 9, 9, 9>("F(4,3) 64ch x 32 tiles: +72 VALU +9 px +9 dsw +9 A +9 B", out, w, p, wb, pb);
  const double c43h = run<9, 36, 36, 5, 5, 9, 9>("F(4,3) if every V fed 128 channels: +36 VALU +5 px +5 dsw +9 A +9 B", out, w, p, wb, pb);
  run<9, 36, 108, 9, 9, 9, 9>("F(4,3) with 1.5x the transform ops (address math, masks): +108 VALU ...", out, w, p, wb, pb);
  // one wave per SIMD (512 registers): 18 accumulator tuples = 288 registers per wave, the whole transform on the four waves
  run<16, 64, 0, 0, 0, 0, 0, 1>("1 wave/SIMD: 64 MFMAs on 16 acc, nothing else (round 4's shape)", out, w, p, wb, pb);
  run<18, 72, 0, 0, 0, 0, 0, 1>("1 wave/SIMD, F(4,3) shape: 72 MFMAs on 18 acc (288 regs), nothing else", out, w, p, wb, pb);
  const double c43o = run<18, 72, 144, 18, 18, 18, 18, 1>("1 wave/SIMD, F(4,3) 64ch x 32 tiles: +144 VALU +18 px +18 dsw +18 A +18 B", out, w, p, wb, pb);
  printf("one-wave form: 1.778 x (%.3f / %.3f) = %.2f\n", c43o, c23, 1.7778 * c43o / c23);
  // F(4,3) x F(2,3): 4x2 output tiles, 6x4 patch, 24 positions (3 multiplies per output instead of 4): 6 accumulator tuples per wave on a
  // 64 ch x 32 tiles block — the same outputs x channels per wave and step as the shipped 128 x 32 block of 2x2 tiles
  run<6, 24, 0, 0, 0, 0, 0>("F(4,3)xF(2,3) shape: 24 MFMAs on 6 acc, nothing else", out, w, p, wb, pb);
  const double c42 = run<6, 24, 36, 6, 6, 6, 6>("F(4,3)xF(2,3) 64ch x 32 tiles(4x2): +36 VALU +6 px +6 dsw +6 A +6 B", out, w, p, wb, pb);
  const double c42b = run<6, 24, 48, 6, 6, 6, 6>("  same with 48 VALU", out, w, p, wb, pb);
  printf("F(4,3)xF(2,3): time per wave-step %.0f vs shipped mix %.0f ticks-equivalent -> K-loop speed-up %.2f (48 VALU: %.2f)\n",
         24 * 64 / c42, 32 * 64 / c23, (32.0 / c23) / (24.0 / c42), (32.0 / c23) / (24.0 / c42b));
  printf("K-loop speed-up of F(4,3) over F(2,3) at equal loop quality: MFMA count 16/4 : 36/16 = 1.778 x (%.3f / %.3f) = %.2f (wide: %.2f)\n",
         c43, c23, 1.7778 * c43 / c23, 1.7778 * c43h / c23);
  return 0;
}
