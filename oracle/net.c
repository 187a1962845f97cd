/* N-group oracle: MXNet Convolution / Deconvolution / FullyConnected semantics in plain C.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — never linked into the product.
 *
 * Follows the layer definitions wired at deepim/symbols/deepIM_flownet.py:63-167 (conv stack),
 * :176-200/:317-340 (heads + k32 s16 grouped upsampling) and MXNet 1.2's documented operator
 * definitions (third-party, not vendored: PARITY UNPINNED by reference tests; cross-checked
 * against torch-CPU in tests/).  Accumulation is a float32 fmaf chain per output element in the
 * order the MI355X fp32-MFMA kernels use — (ci,ky,kx), or channel-pair-interleaved on request — so
 * conv/deconv parity is checked bit-for-bit.
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

static inline float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

/* out (B,Cout,Ho,Wo).  pair_order = 0: the chain runs over k = (ci,ky,kx), ci slowest (the natural MXNet weight order);
 * pair_order = 1: over (ci/2, ky, kx, ci%2) — adjacent input channels interleaved per tap, the order of the LDS-free
 * MI355X kernel on NCHW input (needs even Cin); pair_order = 2: over (ci/8, ky, kx, s, h) with channel 8(ci/8) + s + 4h
 * — the order of the same kernel on channel-blocked (NC8) input (needs Cin % 8 == 0).  fp32 addition is not associative and the reference's own order (cuDNN / MKL-DNN
 * under MXNet) is unspecified, so both are equally faithful restatements; each kernel is checked bit-for-bit against
 * the order it implements and to 1e-5 against the other. */
void oracle_conv2d_order(float* out, const float* in, const float* w, const float* bias, int B, int Cin, int H, int W,
                         int Cout, int kh, int kw, int stride, int pad, float slope, int pair_order) {
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  const int khw = kh * kw, K = Cin * khw;
  /* tasks = (sample, output channel, block of output rows): enough of them for every host core even at B = 1 with 64
   * channels; splitting rows does not change any output's own fmaf chain */
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  int RB = (4 * nthreads + B * Cout - 1) / (B * Cout);
  if (RB < 1) RB = 1;
  if (RB > Ho) RB = Ho;
  const int rows_per = (Ho + RB - 1) / RB;
#pragma omp parallel for collapse(3) schedule(dynamic)
  for (int n = 0; n < B; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int rb = 0; rb < RB; ++rb) {
        const int r0 = rb * rows_per, r1 = r0 + rows_per < Ho ? r0 + rows_per : Ho;
        if (r0 >= r1) continue;
        float* acc = out + ((size_t)n * Cout + co) * Ho * Wo;
        memset(acc + (size_t)r0 * Wo, 0, sizeof(float) * (size_t)(r1 - r0) * Wo);
        for (int s = 0; s < K; ++s) {
          int ci, t;
          if (pair_order == 2) { const int g = s >> 3, e = s & 7; ci = 8 * (g / khw) + (e >> 1) + 4 * (e & 1); t = g % khw; }
          else if (pair_order) { const int g = s >> 1; ci = 2 * (g / khw) + (s & 1); t = g % khw; }
          else { ci = s / khw; t = s % khw; }
          const int ky = t / kw, kx = t % kw;
          const float* ip = in + ((size_t)n * Cin + ci) * H * W;
          const float wv = w[(((size_t)co * Cin + ci) * kh + ky) * kw + kx];
          int wo_lo = 0, wo_hi = Wo;
          while (wo_lo < Wo && wo_lo * stride - pad + kx < 0) ++wo_lo;
          while (wo_hi > wo_lo && (wo_hi - 1) * stride - pad + kx >= W) --wo_hi;
          for (int ho = r0; ho < r1; ++ho) {
            const int hi = ho * stride - pad + ky;
            if (hi < 0 || hi >= H) continue;
            const float* row = ip + (size_t)hi * W - pad + kx;
            float* arow = acc + (size_t)ho * Wo;
            if (stride == 1) {
              for (int wo = wo_lo; wo < wo_hi; ++wo) arow[wo] = fmaf(wv, row[wo], arow[wo]);
            } else {
              for (int wo = wo_lo; wo < wo_hi; ++wo) arow[wo] = fmaf(wv, row[wo * stride], arow[wo]);
            }
          }
        }
        const float bv = bias ? bias[co] : 0.f;
        for (size_t i = (size_t)r0 * Wo; i < (size_t)r1 * Wo; ++i) acc[i] = lrelu(acc[i] + bv, slope);
      }
}

void oracle_conv2d(float* out, const float* in, const float* w, const float* bias, int B, int Cin, int H, int W,
                   int Cout, int kh, int kw, int stride, int pad, float slope) {
  oracle_conv2d_order(out, in, w, bias, B, Cin, H, W, Cout, kh, kw, stride, pad, slope, 0);
}

/* The same convolution — every output the same float32 fmaf chain over (ci,ky,kx) as oracle_conv2d, hence bit-identical
 * (tests/test_oracle_thirdparty.py) — laid out for the host cores instead of for readability: a task is one output row of a
 * block of 8 output channels (B x Cout/8 x Ho tasks: thousands, whatever the layer), the accumulators live as [wo][8] so that
 * the eight channels of a pixel are one AVX2 register (broadcast input value x 8 weights: vectorises for stride 1 and 2
 * alike), and an input row is read once per 8 channels instead of once per channel. bench.py's `cpu_baseline` times THIS
 * build (the stated baseline should use the cores it names); the checker stays oracle_conv2d_order. */
void oracle_conv2d_blocked(float* out, const float* in, const float* w, const float* bias, int B, int Cin, int H, int W,
                           int Cout, int kh, int kw, int stride, int pad, float slope) {
  enum { CB = 8, WMAX = 1024 };
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  if (Wo > WMAX) { oracle_conv2d_order(out, in, w, bias, B, Cin, H, W, Cout, kh, kw, stride, pad, slope, 0); return; }
  const int khw = kh * kw, NCB = (Cout + CB - 1) / CB;
#pragma omp parallel for collapse(3) schedule(dynamic, 4)
  for (int n = 0; n < B; ++n)
    for (int cb = 0; cb < NCB; ++cb)
      for (int ho = 0; ho < Ho; ++ho) {
        float acc[WMAX][CB] __attribute__((aligned(32)));
        const int co0 = cb * CB, nc = Cout - co0 < CB ? Cout - co0 : CB;
        memset(acc, 0, sizeof(float) * CB * (size_t)Wo);
        for (int ci = 0; ci < Cin; ++ci)
          for (int ky = 0; ky < kh; ++ky) {
            const int hi = ho * stride - pad + ky;
            if (hi < 0 || hi >= H) continue;
            for (int kx = 0; kx < kw; ++kx) {
              float wv[CB] __attribute__((aligned(32)));
              for (int c = 0; c < CB; ++c) wv[c] = c < nc ? w[((size_t)(co0 + c) * Cin + ci) * khw + ky * kw + kx] : 0.f;
              int wo_lo = 0, wo_hi = Wo;
              while (wo_lo < Wo && wo_lo * stride - pad + kx < 0) ++wo_lo;
              while (wo_hi > wo_lo && (wo_hi - 1) * stride - pad + kx >= W) --wo_hi;
              const float* row = in + (((size_t)n * Cin + ci) * H + hi) * W - pad + kx;
              for (int wo = wo_lo; wo < wo_hi; ++wo) {
                const float x = row[(size_t)wo * stride];
                for (int c = 0; c < CB; ++c) acc[wo][c] = fmaf(wv[c], x, acc[wo][c]);
              }
            }
          }
        for (int c = 0; c < nc; ++c) {
          const float bv = bias ? bias[co0 + c] : 0.f;
          float* orow = out + (((size_t)n * Cout + co0 + c) * Ho + ho) * Wo;
          for (int wo = 0; wo < Wo; ++wo) orow[wo] = lrelu(acc[wo][c] + bv, slope);
        }
      }
}

/* bench.py's cpu_baseline pins the OpenMP team to the physical cores (the default, one thread per hardware thread, oversubscribes
 * the FMA units of an SMT host) */
void oracle_set_omp_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int oracle_omp_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* MXNet Deconvolution k4 s2 p0, w (Cin,Cout,4,4), cropped at (crop_y,crop_x) to (Ho,Wo). */
void oracle_deconv4x4s2_crop(float* out, const float* in, const float* w, const float* bias, int B, int Cin, int H,
                             int W, int Cout, int Ho, int Wo, int crop_y, int crop_x, float slope) {
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int n = 0; n < B; ++n)
    for (int co = 0; co < Cout; ++co) {
      float* o = out + ((size_t)n * Cout + co) * Ho * Wo;
      for (int yo = 0; yo < Ho; ++yo)
        for (int xo = 0; xo < Wo; ++xo) {
          const int y = yo + crop_y, x = xo + crop_x;
          float acc = 0.f;
          for (int ci = 0; ci < Cin; ++ci) {
            const float* ip = in + ((size_t)n * Cin + ci) * H * W;
            const float* wp = w + ((size_t)ci * Cout + co) * 16;
            for (int ky = y & 1; ky < 4; ky += 2) {
              const int iy = (y - ky) / 2;
              const int yok = (y - ky) >= 0 && iy < H;
              for (int kx = x & 1; kx < 4; kx += 2) {
                const int ix = (x - kx) / 2;
                const int ok = yok && (x - kx) >= 0 && ix < W;
                acc = fmaf(wp[ky * 4 + kx], ok ? ip[iy * W + ix] : 0.f, acc);
              }
            }
          }
          o[yo * Wo + xo] = lrelu(acc + (bias ? bias[co] : 0.f), slope);
        }
    }
}

/* depthwise Deconvolution k32 s16 p0 no-bias, w (C,1,32,32), crop, times scale */
void oracle_upsample16_crop(float* out, const float* in, const float* w, int B, int C, int H, int W, int Ho, int Wo,
                            int crop_y, int crop_x, float scale) {
#pragma omp parallel for schedule(dynamic)
  for (int bc = 0; bc < B * C; ++bc) {
    const float* ip = in + (size_t)bc * H * W;
    const float* wp = w + (size_t)(bc % C) * 1024;
    for (int yo = 0; yo < Ho; ++yo)
      for (int xo = 0; xo < Wo; ++xo) {
        const int y = yo + crop_y, x = xo + crop_x;
        float acc = 0.f;
        for (int iy = 0; iy < H; ++iy) {
          const int ky = y - iy * 16;
          if (ky < 0 || ky >= 32) continue;
          for (int ix = 0; ix < W; ++ix) {
            const int kx = x - ix * 16;
            if (kx < 0 || kx >= 32) continue;
            acc = fmaf(ip[iy * W + ix], wp[ky * 32 + kx], acc);
          }
        }
        out[((size_t)bc * Ho + yo) * Wo + xo] = acc * scale;
      }
  }
}

/* FullyConnected y = x·Wᵀ + b, float64 accumulation (most accurate order-free reference) */
void oracle_fc(float* out, const float* in, const float* w, const float* bias, int B, int I, int O, float slope) {
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < O; ++o) {
      double acc = 0.0;
      const float* x = in + (size_t)b * I;
      const float* wr = w + (size_t)o * I;
      for (int k = 0; k < I; ++k) acc += (double)x[k] * (double)wr[k];
      out[(size_t)b * O + o] = lrelu((float)acc + (bias ? bias[o] : 0.f), slope);
    }
}

/* ---------------------------------------------------------------------------------------------- backward ----
 * Gradients of MXNet Convolution / FullyConnected (the training graph, deepim/symbols/deepIM_flownet.py:367-546 →
 * module.backward, deepim/core/module.py:1131-1137).  Plain definitions, float64 accumulation (order-free reference;
 * cross-checked against torch autograd in tests/test_oracle_thirdparty.py). */

/* dX[n,ci,h,w] = Σ_{co,ky,kx} dY[n,co,ho,wo]·W[co,ci,ky,kx] with h = ho·s − p + ky, w = wo·s − p + kx */
void oracle_conv2d_dgrad(float* dx, const float* dy, const float* w, int B, int Cin, int H, int W, int Cout, int kh,
                         int kw, int stride, int pad) {
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int n = 0; n < B; ++n)
    for (int ci = 0; ci < Cin; ++ci) {
      double* acc = (double*)calloc((size_t)H * W, sizeof(double));
      for (int co = 0; co < Cout; ++co) {
        const float* g = dy + ((size_t)n * Cout + co) * Ho * Wo;
        const float* wp = w + (((size_t)co * Cin + ci) * kh) * kw;
        for (int ky = 0; ky < kh; ++ky)
          for (int kx = 0; kx < kw; ++kx) {
            const double wv = wp[ky * kw + kx];
            for (int ho = 0; ho < Ho; ++ho) {
              const int h = ho * stride - pad + ky;
              if (h < 0 || h >= H) continue;
              for (int wo = 0; wo < Wo; ++wo) {
                const int x = wo * stride - pad + kx;
                if (x < 0 || x >= W) continue;
                acc[(size_t)h * W + x] += wv * (double)g[(size_t)ho * Wo + wo];
              }
            }
          }
      }
      float* o = dx + ((size_t)n * Cin + ci) * H * W;
      for (size_t i = 0; i < (size_t)H * W; ++i) o[i] = (float)acc[i];
      free(acc);
    }
}

/* dW[co,ci,ky,kx] = Σ_{n,ho,wo} dY[n,co,ho,wo]·X[n,ci,ho·s−p+ky,wo·s−p+kx];  db[co] = Σ dY[n,co,:,:] */
void oracle_conv2d_wgrad(float* dw, float* db, const float* x, const float* dy, int B, int Cin, int H, int W, int Cout,
                         int kh, int kw, int stride, int pad) {
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx) {
          double acc = 0.0;
          for (int n = 0; n < B; ++n) {
            const float* g = dy + ((size_t)n * Cout + co) * Ho * Wo;
            const float* ip = x + ((size_t)n * Cin + ci) * H * W;
            for (int ho = 0; ho < Ho; ++ho) {
              const int h = ho * stride - pad + ky;
              if (h < 0 || h >= H) continue;
              for (int wo = 0; wo < Wo; ++wo) {
                const int xx = wo * stride - pad + kx;
                if (xx < 0 || xx >= W) continue;
                acc += (double)g[(size_t)ho * Wo + wo] * (double)ip[(size_t)h * W + xx];
              }
            }
          }
          dw[(((size_t)co * Cin + ci) * kh + ky) * kw + kx] = (float)acc;
        }
  if (db) {
#pragma omp parallel for
    for (int co = 0; co < Cout; ++co) {
      double acc = 0.0;
      for (int n = 0; n < B; ++n) {
        const float* g = dy + ((size_t)n * Cout + co) * Ho * Wo;
        for (int i = 0; i < Ho * Wo; ++i) acc += g[i];
      }
      db[co] = (float)acc;
    }
  }
}

/* FullyConnected backward: dX = dY·W, dW = dYᵀ·X, db = Σ_b dY */
void oracle_fc_backward(float* dx, float* dw, float* db, const float* dy, const float* x, const float* w, int B, int I, int O) {
#pragma omp parallel for
  for (int b = 0; b < B; ++b) {
    double* acc = (double*)calloc((size_t)I, sizeof(double));
    for (int o = 0; o < O; ++o) {
      const double g = dy[(size_t)b * O + o];
      const float* wr = w + (size_t)o * I;
      for (int k = 0; k < I; ++k) acc[k] += g * (double)wr[k];
    }
    for (int k = 0; k < I; ++k) dx[(size_t)b * I + k] = (float)acc[k];
    free(acc);
  }
#pragma omp parallel for
  for (int o = 0; o < O; ++o) {
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += dy[(size_t)b * O + o];
    db[o] = (float)s;
    for (int k = 0; k < I; ++k) {
      double a = 0.0;
      for (int b = 0; b < B; ++b) a += (double)dy[(size_t)b * O + o] * (double)x[(size_t)b * I + k];
      dw[(size_t)o * I + k] = (float)a;
    }
  }
}

/* Deconvolution k4 s2 p0 + Crop(crop_y,crop_x → Ho,Wo) backward: dy is the gradient of the CROPPED output (B,Cout,Ho,Wo);
 * w (Cin,Cout,4,4).  d_in[n,ci,iy,ix] = Σ_{co,ky,kx} dy[n,co,2iy+ky−crop_y,2ix+kx−crop_x]·w[ci,co,ky,kx];
 * dw[ci,co,ky,kx] = Σ_{n,iy,ix} in[n,ci,iy,ix]·dy[n,co,2iy+ky−crop_y,2ix+kx−crop_x];  db[co] = Σ dy */
void oracle_deconv4x4s2_crop_backward(float* d_in, float* dw, float* db, const float* in, const float* w, const float* dy, int B,
                                      int Cin, int H, int W, int Cout, int Ho, int Wo, int crop_y, int crop_x) {
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int n = 0; n < B; ++n)
    for (int ci = 0; ci < Cin; ++ci) {
      float* o = d_in + ((size_t)n * Cin + ci) * H * W;
      for (int iy = 0; iy < H; ++iy)
        for (int ix = 0; ix < W; ++ix) {
          double acc = 0.0;
          for (int co = 0; co < Cout; ++co) {
            const float* g = dy + ((size_t)n * Cout + co) * Ho * Wo;
            const float* wp = w + ((size_t)ci * Cout + co) * 16;
            for (int ky = 0; ky < 4; ++ky) {
              const int y = 2 * iy + ky - crop_y;
              if (y < 0 || y >= Ho) continue;
              for (int kx = 0; kx < 4; ++kx) {
                const int x = 2 * ix + kx - crop_x;
                if (x < 0 || x >= Wo) continue;
                acc += (double)g[(size_t)y * Wo + x] * (double)wp[ky * 4 + kx];
              }
            }
          }
          o[(size_t)iy * W + ix] = (float)acc;
        }
    }
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int ci = 0; ci < Cin; ++ci)
    for (int co = 0; co < Cout; ++co)
      for (int ky = 0; ky < 4; ++ky)
        for (int kx = 0; kx < 4; ++kx) {
          double acc = 0.0;
          for (int n = 0; n < B; ++n) {
            const float* ip = in + ((size_t)n * Cin + ci) * H * W;
            const float* g = dy + ((size_t)n * Cout + co) * Ho * Wo;
            for (int iy = 0; iy < H; ++iy) {
              const int y = 2 * iy + ky - crop_y;
              if (y < 0 || y >= Ho) continue;
              for (int ix = 0; ix < W; ++ix) {
                const int x = 2 * ix + kx - crop_x;
                if (x < 0 || x >= Wo) continue;
                acc += (double)ip[(size_t)iy * W + ix] * (double)g[(size_t)y * Wo + x];
              }
            }
          }
          dw[((size_t)ci * Cout + co) * 16 + ky * 4 + kx] = (float)acc;
        }
  if (db) {
#pragma omp parallel for
    for (int co = 0; co < Cout; ++co) {
      double acc = 0.0;
      for (int n = 0; n < B; ++n) {
        const float* g = dy + ((size_t)n * Cout + co) * Ho * Wo;
        for (int i = 0; i < Ho * Wo; ++i) acc += g[i];
      }
      db[co] = (float)acc;
    }
  }
}

/* data gradient of the depthwise k32 s16 upsampling (fixed weights, lr_mult 0): d_in[bc,iy,ix] = scale·Σ dy[bc,16iy+ky−cy,16ix+kx−cx]·w[c,ky,kx] */
void oracle_upsample16_crop_backward(float* d_in, const float* dy, const float* w, int B, int C, int H, int W, int Ho, int Wo,
                                     int crop_y, int crop_x, float scale) {
#pragma omp parallel for schedule(dynamic)
  for (int bc = 0; bc < B * C; ++bc) {
    const float* g = dy + (size_t)bc * Ho * Wo;
    const float* wp = w + (size_t)(bc % C) * 1024;
    for (int iy = 0; iy < H; ++iy)
      for (int ix = 0; ix < W; ++ix) {
        double acc = 0.0;
        for (int ky = 0; ky < 32; ++ky) {
          const int y = 16 * iy + ky - crop_y;
          if (y < 0 || y >= Ho) continue;
          for (int kx = 0; kx < 32; ++kx) {
            const int x = 16 * ix + kx - crop_x;
            if (x < 0 || x >= Wo) continue;
            acc += (double)g[(size_t)y * Wo + x] * (double)wp[ky * 32 + kx];
          }
        }
        d_in[((size_t)bc * H + iy) * W + ix] = (float)(acc * (double)scale);
      }
  }
}
