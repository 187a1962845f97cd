// Dev probe: what does it cost to issue other work between fp32 MFMAs? Same issue pattern as the conv kernels
// (one slot per v_mfma_f32_32x32x2_f32), with the extras switched on one at a time:
//   bit0: 3 VALU per slot   bit1: 1 buffer_load per slot feeding the operands (L1/L2-resident source)
//   bit2: 2 SALU per slot   bit3: operands change every MFMA (ring of 16 registers, no loads)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_issue_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SArgs { int v[8]; };
template <int MODE>
__global__ __launch_bounds__(256, 3) void k(float* out, const float* src, int iters, unsigned nbytes, int sstep, SArgs sa) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float ring[16];
  for (int i = 0; i < 16; ++i) ring[i] = 1.0f + i * 0.01f + lane * 1e-3f;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)nbytes, 0x00020000);
  int voff = (threadIdx.x * 4 + blockIdx.x * 1024) & (nbytes - 1);
  unsigned m0 = 0x12345678u ^ threadIdx.x, m1 = ~m0;
  int soff = 0;
  unsigned sbits = blockIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int cur = (MODE & (2 | 8)) ? (u & 15) : 0;
      const int nxt = (u + 12) & 15;
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[cur], ring[(cur + 1) & 15], acc[u & 3], 0, 0, 0);
      asm volatile("" : "+v"(acc[u & 3]));
      unsigned vo = (unsigned)voff;
      if (MODE & 4) {
        sbits = sbits * 5u + 1u;
        soff = (int)((sbits >> 7) & 0xffcu);
      }
      if (MODE & 32) {          // VALU consumes a long-settled SGPR (kernel argument), no SALU in the slot
        const unsigned inv = __builtin_amdgcn_ubfe(m0, (unsigned)sa.v[u & 7], 1u);
        vo = (inv << 20) | vo;
      } else if (MODE & 64) {   // SALU result produced one slot earlier than the VALU that consumes it
        const unsigned prev = sbits;
        sbits = sbits * 5u + 1u;
        const unsigned inv = __builtin_amdgcn_ubfe(m0, prev & 31u, 1u);
        vo = (inv << 20) | vo;
      } else if (MODE & 1) {
        const unsigned word = (sbits & 32) ? m1 : m0;
        const unsigned inv = __builtin_amdgcn_ubfe(word, sbits & 31u, 1u);
        vo = (inv << 20) | vo;   // keeps the address in range (bit 20 < nbytes when nbytes >= 2 MB)
      }
      if ((MODE & 2) && (MODE & 16)) {   // load issued, but its address does not depend on this slot's VALU/SALU results
        ring[nxt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, u * sstep, 0));
        asm volatile("" :: "v"(vo), "s"(soff));
      } else if ((MODE & 2) && sstep < 0) {   // walk a region of -sstep bytes: a fresh 256 B line per load
        const int region = -sstep;
        const int so = (int)(((long)it * 8192 + u * 256 + (MODE & 128 ? blockIdx.x * 65536 : 0)) & (region - 1));
        ring[nxt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, so, 0));
      } else if (MODE & 2) {
        ring[nxt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)vo, soff + u * sstep, 0));
      } else if (MODE & (1 | 32 | 64)) {
        asm volatile("" :: "v"(vo));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += ring[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)soff;
}

template <int MODE>
void run(const char* name, float* out, float* src, unsigned nbytes, int sstep) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  SArgs sa; for (int i = 0; i < 8; ++i) sa.v[i] = (i * 7 + 3) & 31;
  for (int blocks : {768, 1280}) {
    const int iters = 1000;
    k<MODE><<<blocks, 256>>>(out, src, iters, nbytes, sstep, sa); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, src, iters, nbytes, sstep, sa); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("%-44s blocks %4d: %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
  }
}
int main() {
  float *out, *src; hipMalloc(&out, 4096 * 256 * 4);
  const unsigned nbytes = 64u << 20;
  hipMalloc(&src, nbytes); hipMemset(src, 0, nbytes);
  run<0>("pure MFMA", out, src, nbytes, 0);
  run<8>("MFMA, operands rotate", out, src, nbytes, 0);
  run<1>("+3 VALU/slot", out, src, nbytes, 0);
  run<4>("+2 SALU/slot", out, src, nbytes, 0);
  run<2>("+1 buffer_load/slot (same 1 KB/block: L1)", out, src, nbytes, 0);
  run<2>("+1 buffer_load/slot (256 B stride: streams)", out, src, nbytes, 256);
  run<2>("+load, all blocks walk the same 2 MB (L2-resident)", out, src, nbytes, -(2 << 20));
  run<2>("+load, all blocks walk the same 32 MB (MALL/HBM)", out, src, nbytes, -(32 << 20));
  run<130>("+load, per-block walk of 64 MB (HBM stream)", out, src, nbytes, -(64 << 20));
  run<32>("+2 VALU reading settled SGPRs (kernel args)", out, src, nbytes, 0);
  run<34>("+2 VALU reading settled SGPRs + dependent load", out, src, nbytes, 0);
  run<64>("+SALU, VALU reads the PREVIOUS slot's result", out, src, nbytes, 0);
  run<66>("  same + dependent load", out, src, nbytes, 0);
  run<7>("+VALU+SALU+load (L1)", out, src, nbytes, 0);
  run<3>("+VALU+load (voffset from VALU)", out, src, nbytes, 0);
  run<6>("+SALU+load (soffset from SALU)", out, src, nbytes, 0);
  run<5>("+VALU+SALU (VALU reads SALU result), no load", out, src, nbytes, 0);
  run<23>("+VALU+SALU+load, load independent of both", out, src, nbytes, 0);
  run<19>("+VALU+load, load independent", out, src, nbytes, 0);
  return 0;
}
