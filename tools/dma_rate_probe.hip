// Dev probe: what ONE CU sustains for global → LDS (buffer_load_dwordx4 … lds, 1 KB per wave-instruction) and global → VGPR
// loads, by source pattern — the quantity that bounds the fp16 conv kernels' K loop (every 256x256x32 stage needs 32 pieces per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/dma_rate_probe.hip -o tools/dma_rate_probe.bin && tools/dma_rate_probe.bin
// modes: 0 contiguous 1 KB pieces, L2-resident window (96 KB per CU: misses L1)     = weight pieces
//        1 16 x 64 B segments at a 256 B pitch, same window                          = activation pieces (conv3: Cin·2 = 256 B pixel pitch)
//        2 contiguous pieces, 8 KB window (L1-resident)
//        3 contiguous pieces to VGPRs (global_load_dwordx4), L2-resident window
//        4 8 x 128 B segments at a 256 B pitch (whole cache lines)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k(const char* src, long per_cu, int iters, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long)blockIdx.x * per_cu), 0, (int)per_cu, 0x00020000);
  const unsigned win = MODE == 2 ? 8192u : (unsigned)per_cu;
  unsigned lane_off;
  if (MODE == 1) lane_off = (unsigned)((lane >> 2) * 256 + (lane & 3) * 16);
  else if (MODE == 4) lane_off = (unsigned)((lane >> 3) * 256 + (lane & 7) * 16);
  else lane_off = (unsigned)lane * 16;
  const unsigned span = (MODE == 1) ? 4096u : (MODE == 4 ? 2048u : 1024u);
  unsigned pos = (unsigned)wave * span;
  const unsigned ldsbase = (unsigned)(size_t)lds + (unsigned)wave * 8192u;
  i32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned off = (pos % win) + lane_off;
      pos += span * NW;
      if (MODE == 3) {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        acc += v;
      } else {
        const unsigned la = __builtin_amdgcn_readfirstlane(ldsbase + (unsigned)q * 1024u);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(la), "v"(off), "s"(rs) : "memory");
      }
    }
    if (MODE != 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // one batch of 8 in flight behind the one being issued
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (MODE == 3 && acc.x == 12345) sink[0] = (float)acc.y;
}
template <int MODE, int NW>
void run(const char* name, const char* src, long per_cu, unsigned long long* cyc, float* sink) {
  const int iters = 400;
  hipFuncSetAttribute((const void*)k<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) k<MODE, NW><<<256, NW * 64, 65536>>>(src, per_cu, iters, cyc, sink);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 256; ++i) s += (double)h[i];
  const double cycles = s / 256, bytes = (double)iters * 8 * NW * 1024;
  printf("%-58s %d waves/CU: %6.1f B/clk/CU  (%5.1f cycles per 1 KB piece)\n", name, NW, bytes / cycles, cycles / (iters * 8.0 * NW));
}
int main() {
  const long per_cu = 96 * 1024;
  char* src; unsigned long long* cyc; float* sink;
  hipMalloc(&src, per_cu * 256 + 65536); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 64);
  hipMemset(src, 1, per_cu * 256 + 65536);
#define RUN(M, NAME) run<M, 4>(NAME, src, per_cu, cyc, sink); run<M, 8>(NAME, src, per_cu, cyc, sink);
  RUN(0, "LDS-DMA, contiguous 1 KB pieces, L2-resident")
  RUN(1, "LDS-DMA, 16 x 64 B segments at 256 B pitch, L2-resident")
  RUN(4, "LDS-DMA, 8 x 128 B segments at 256 B pitch, L2-resident")
  RUN(2, "LDS-DMA, contiguous pieces, L1-resident (8 KB window)")
  RUN(3, "global_load_dwordx4 to VGPRs, contiguous, L2-resident")
  return 0;
}
