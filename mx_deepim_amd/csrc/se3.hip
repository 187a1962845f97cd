// S-group: SE(3) pose update and 3-D point transform.
//   S1 RT_transform            lib/pair_matching/RT_transform.py:127-151 (+ :47-61, :74-95, :383-429)
//   S5 Transform3D forward     deepim/operator_py/transform3d.py:34-97, :185-212
//   S6 Transform3D backward    deepim/operator_py/transform3d.py:99-151, :153-183, :214-281
//
// The reference does this on the host with NumPy-1.x scalar promotion: float32 inputs,
// `2.0 / Nq` and everything downstream in float64, a few float32 islands
// (T_src[0] / T_src[2]). The kernels keep that pattern operation by operation (fp64 is
// full rate on CDNA4; the work is a few hundred flops per pair). Reductions over the
// N model points use wavefront shuffles + one LDS hop.
#include "common.h"
#include "pose_head.h"

namespace {

enum RotCoord { RC_MODEL = 0, RC_CAMERA = 1, RC_CAMERA_NEW = 2, RC_NAIVE = 3 };

struct Vec3d { double v[3]; };

// quaternion (already float32 scalars) → float64 matrix, RT_transform.py:409-429
__device__ void quat2mat_f64(float w, float x, float y, float z, double* M) {
  const double s = 2.0 / (double)(w * w + x * x + y * y + z * z);
  const double X = x * s, Y = y * s, Z = z * s;
  const double wX = w * X, wY = w * Y, wZ = w * Z;
  const double xX = x * X, xY = x * Y, xZ = x * Z;
  const double yY = y * Y, yZ = y * Z, zZ = z * Z;
  M[0] = 1.0 - (yY + zZ); M[1] = xY - wZ; M[2] = xZ + wY;
  M[3] = xY + wZ; M[4] = 1.0 - (xX + zZ); M[5] = yZ - wX;
  M[6] = xZ - wY; M[7] = yZ + wX; M[8] = 1.0 - (xX + yY);
}

// T_transform, RT_transform.py:74-95 (T_src, T_delta float32; means/stds float64)
__device__ void t_transform_f64(const float* Tsrc, const float* Td, const Vec3d& mu, const Vec3d& sd, int rc,
                                double* T) {
  double d1[3];
  for (int i = 0; i < 3; ++i) d1[i] = (double)Td[i] * sd.v[i] + mu.v[i];
  const double z2 = (double)Tsrc[2] / exp(d1[2]);
  T[2] = z2;
  if (rc == RC_CAMERA || rc == RC_MODEL) {
    T[0] = z2 * (d1[0] + (double)(Tsrc[0] / Tsrc[2]));
    T[1] = z2 * (d1[1] + (double)(Tsrc[1] / Tsrc[2]));
  } else {  // CAMERA_NEW
    T[0] = (double)Tsrc[2] * d1[0] + (double)Tsrc[0];
    T[1] = (double)Tsrc[2] * d1[1] + (double)Tsrc[1];
  }
}

// euler2mat(ai, aj, ak, 'sxyz') (RT_transform.py:240-307, the default axes RT_transform calls it with at :131):
// math.sin/cos take Python floats, so everything is float64
__device__ void euler2mat_sxyz_f64(double ai, double aj, double ak, double* M) {
  const double si = sin(ai), sj = sin(aj), sk = sin(ak);
  const double ci = cos(ai), cj = cos(aj), ck = cos(ak);
  const double cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  M[0] = cj * ck; M[1] = sj * sc - cs; M[2] = sj * cc + ss;
  M[3] = cj * sk; M[4] = sj * ss + cc; M[5] = sj * cs - sc;
  M[6] = -sj;     M[7] = cj * si;      M[8] = cj * ci;
}

// mat2euler(M, 'sxyz') (RT_transform.py:310-373), float64
__device__ void mat2euler_sxyz_f64(const double* M, double* e) {
  const double eps4 = 2.220446049250313e-16 * 4.0;
  const double cy = sqrt(M[0] * M[0] + M[3] * M[3]);
  if (cy > eps4) {
    e[0] = atan2(M[7], M[8]);
    e[1] = atan2(-M[6], cy);
    e[2] = atan2(M[3], M[0]);
  } else {
    e[0] = atan2(-M[5], M[4]);
    e[1] = atan2(-M[6], cy);
    e[2] = 0.0;
  }
}

// quat2mat (RT_transform.py:383-429) on float32 scalars: eye(3) when Nq < _FLOAT_EPS (:236,412)
__device__ void quat2mat_checked_f64(float w, float x, float y, float z, double* Rd) {
  const float Nq = ((w * w + x * x) + y * y) + z * z;
  if (!((double)Nq < 2.220446049250313e-16)) quat2mat_f64(w, x, y, z, Rd);
  else for (int i = 0; i < 9; ++i) Rd[i] = (i % 4 == 0) ? 1.0 : 0.0;
}

// r_len = 4: se3 rows are [quat(4) | trans(3)] (ROT_TYPE QUAT); r_len = 3: [euler(3) | trans(3)] (ROT_TYPE EULER, :130-131)
// one sample: P = its source pose (12 floats), q = its se3 row [rotation(r_len) | translation(3)]
__device__ void rt_transform_one(float* __restrict__ pose_est, double* __restrict__ pose_est64, const float* P, const float* q,
                                 const Vec3d& mu, const Vec3d& sd, int rc, int b, int r_len) {
  const float* t = q + r_len;
  double Rd[9];
  if (r_len == 3) {
    euler2mat_sxyz_f64((double)q[0], (double)q[1], (double)q[2], Rd);
  } else {
    // quat = r / LA.norm(r)  (float32)
    const float nrm = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
    quat2mat_checked_f64(q[0] / nrm, q[1] / nrm, q[2] / nrm, q[3] / nrm, Rd);
  }
  double out[12];
  if (rc == RC_NAIVE) {
    // se3_mul(se3_mx, pose_src): float64·float32 products, result cast to float32 (projection.py:26-43)
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j)
        out[i * 4 + j] = (double)(float)(Rd[i * 3 + 0] * P[0 * 4 + j] + Rd[i * 3 + 1] * P[1 * 4 + j] + Rd[i * 3 + 2] * P[2 * 4 + j]);
      out[i * 4 + 3] = (double)(float)((Rd[i * 3 + 0] * P[3] + Rd[i * 3 + 1] * P[7] + Rd[i * 3 + 2] * P[11]) + (double)t[i]);
    }
  } else {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        if (rc == RC_MODEL)
          out[i * 4 + j] = (double)P[i * 4 + 0] * Rd[0 * 3 + j] + (double)P[i * 4 + 1] * Rd[1 * 3 + j] + (double)P[i * 4 + 2] * Rd[2 * 3 + j];
        else
          out[i * 4 + j] = Rd[i * 3 + 0] * P[0 * 4 + j] + Rd[i * 3 + 1] * P[1 * 4 + j] + Rd[i * 3 + 2] * P[2 * 4 + j];
      }
    const float Tsrc[3] = {P[3], P[7], P[11]};
    double T[3];
    t_transform_f64(Tsrc, t, mu, sd, rc, T);
    out[3] = T[0]; out[7] = T[1]; out[11] = T[2];
  }
  for (int i = 0; i < 12; ++i) {
    pose_est[b * 12 + i] = (float)out[i];
    if (pose_est64) pose_est64[b * 12 + i] = out[i];
  }
}

__global__ void rt_transform_kernel(float* __restrict__ pose_est, double* __restrict__ pose_est64,
                                    const float* __restrict__ pose_src, const float* __restrict__ se3, Vec3d mu,
                                    Vec3d sd, int rc, int B, int r_len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float P[12], q[7];
  for (int i = 0; i < 12; ++i) P[i] = pose_src[b * 12 + i];      // copied first: pose_est may be pose_src (in-place update)
  for (int i = 0; i < r_len + 3; ++i) q[i] = se3[b * (r_len + 3) + i];
  rt_transform_one(pose_est, pose_est64, P, q, mu, sd, rc, b, r_len);
}

// The tail of a test-graph refinement iteration in ONE launch (one 1024-thread block per sample): fc7 (256 → 256, LeakyReLU)
// → rot / trans FullyConnected + inverse ZoomTrans → se3 → RT_transform — what deepim_fc_forward (two kernels) +
// deepim_pose_head_forward + deepim_rt_transform run as four dependent 4-5 µs launches (at the per-GPU share of config 3, B = 4,
// that tail was 1.5 % of the iteration). Same sums in the same order, so bit-identical to the four launches:
//   fc7: the order of fc_partial_kernel / fc_finalize_kernel for I = 256 (one K slice): lane l multiplies k = 4l … 4l+3 in a
//   fmaf chain from 0, the 64 lane sums are added by the xor butterfly 32 → 1 (fp addition is commutative, so the reduce-scatter
//   form there and the all-reduce form here give the same value), then + bias, LeakyReLU;
//   pose head: di_pose_head_wave (csrc/pose_head.h); RT_transform: rt_transform_one above.
__global__ __launch_bounds__(1024) void pose_tail_kernel(float* __restrict__ fc7_out, float* __restrict__ se3, float* __restrict__ pose_est,
                                                         const float* __restrict__ fc6, const float* __restrict__ w7,
                                                         const float* __restrict__ b7, const float* __restrict__ w_rot,
                                                         const float* __restrict__ b_rot, const float* __restrict__ w_trans,
                                                         const float* __restrict__ b_trans, const float* __restrict__ zoom_factor,
                                                         const float* __restrict__ pose_src, Vec3d mu, Vec3d sd, int rc, float slope) {
  // 16 waves per sample, 16 fc7 outputs per wave: every wave issues its 16 weight rows (1 KB each, one dwordx4 per lane) in ONE
  // batch — with few blocks on the chip the kernel is pure load latency, so the loads must not queue behind one another
  __shared__ float f7[256];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 xv = *reinterpret_cast<const float4*>(fc6 + (long)b * 256 + lane * 4);
  float4 wv[16];
#pragma unroll
  for (int oo = 0; oo < 16; ++oo) wv[oo] = *reinterpret_cast<const float4*>(w7 + (long)(wave * 16 + oo) * 256 + lane * 4);
  const float bias = (lane < 16 && b7) ? b7[wave * 16 + lane] : 0.f;
  float keep = 0.f;
#pragma unroll
  for (int oo = 0; oo < 16; ++oo) {
    float a = 0.f;
    a = fmaf(xv.x, wv[oo].x, a); a = fmaf(xv.y, wv[oo].y, a); a = fmaf(xv.z, wv[oo].z, a); a = fmaf(xv.w, wv[oo].w, a);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == oo) keep = a;        // lane oo keeps output oo of this wave
  }
  if (lane < 16) {
    float v = keep + bias;
    v = v > 0.f ? v : v * slope;
    f7[wave * 16 + lane] = v;
    fc7_out[(long)b * 256 + wave * 16 + lane] = v;
  }
  __syncthreads();
  if (wave == 0) {
    float q[7], P[12];
    di_pose_head_wave(q, f7, w_rot, b_rot, w_trans, b_trans, zoom_factor[b * 4 + 0], 256, lane);
    if (lane == 0) {
      for (int i = 0; i < 12; ++i) P[i] = pose_src[b * 12 + i];
#pragma unroll
      for (int r = 0; r < 7; ++r) se3[b * 7 + r] = q[r];
      rt_transform_one(pose_est, nullptr, P, q, mu, sd, rc, b, 4);
    }
  }
}

// --- calc_RT_delta (ground-truth labels) --------------------------------------------
// largest eigenvector of a symmetric 4x4 (cyclic Jacobi, float64) — mat2quat, RT_transform.py:485-509
__device__ void sym4_max_eigvec(double A[4][4], double* vec) {
  double V[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
    if (off < 1e-30) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 4; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - sn * akq;
          A[k][q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 4; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - sn * aqk;
          A[q][k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - sn * vkq;
          V[k][q] = sn * vkp + c * vkq;
        }
      }
  }
  int best = 0;
  for (int i = 1; i < 4; ++i)
    if (A[i][i] > A[best][best]) best = i;
  for (int k = 0; k < 4; ++k) vec[k] = V[k][best];
}

// mat2quat (RT_transform.py:432-509) of a float32 3x3: float64 quaternion, w >= 0
__device__ void mat2quat_f64(const float* Rd, double* q) {
  // K from the 3x3 (Q_ab = contribution of input a to output b = M[b][a])
  const double Qxx = Rd[0], Qyx = Rd[1], Qzx = Rd[2], Qxy = Rd[3], Qyy = Rd[4], Qzy = Rd[5], Qxz = Rd[6], Qyz = Rd[7],
               Qzz = Rd[8];
  double K[4][4] = {{Qxx - Qyy - Qzz, Qyx + Qxy, Qzx + Qxz, Qyz - Qzy},
                    {Qyx + Qxy, Qyy - Qxx - Qzz, Qzy + Qyz, Qzx - Qxz},
                    {Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, Qxy - Qyx},
                    {Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz}};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) K[i][j] /= 3.0;
  double v[4];
  sym4_max_eigvec(K, v);
  q[0] = v[3]; q[1] = v[0]; q[2] = v[1]; q[3] = v[2];
  if (q[0] < 0)
    for (int i = 0; i < 4; ++i) q[i] = -q[i];
}

// se3_mul(tgt, se3_inverse(src)) in float32 (lib/utils/projection.py:12-43): Rd (3x3), td (3)
__device__ void se3_src2tgt_f32(const float* S, const float* T, float* Rd, float* td) {
  float Ri[9], ti[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Ri[i * 3 + j] = S[j * 4 + i];
  for (int i = 0; i < 3; ++i) ti[i] = -1.f * ((Ri[i * 3 + 0] * S[3] + Ri[i * 3 + 1] * S[7]) + Ri[i * 3 + 2] * S[11]);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      Rd[i * 3 + j] = (T[i * 4 + 0] * Ri[0 * 3 + j] + T[i * 4 + 1] * Ri[1 * 3 + j]) + T[i * 4 + 2] * Ri[2 * 3 + j];
    td[i] = ((T[i * 4 + 0] * ti[0] + T[i * 4 + 1] * ti[1]) + T[i * 4 + 2] * ti[2]) + T[i * 4 + 3];
  }
}

// rot_type (RT_transform.py:34-41): 0 QUAT → rot (B,4), 1 EULER → rot (B,3), 2 MATRIX → rot (B,9)
__global__ void calc_rt_delta_kernel(float* __restrict__ rot, float* __restrict__ trans,
                                     const float* __restrict__ pose_src, const float* __restrict__ pose_tgt, Vec3d mu,
                                     Vec3d sd, int rc, int B, int rot_type) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* S = pose_src + b * 12;
  const float* T = pose_tgt + b * 12;
  float Rd[9];
  double Td[3];
  if (rc == RC_NAIVE) {  // se3_mul(tgt, se3_inverse(src)), float32 (projection.py)
    float td[3];
    se3_src2tgt_f32(S, T, Rd, td);
    for (int i = 0; i < 3; ++i) Td[i] = (double)td[i];
  } else {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        if (rc == RC_MODEL)  // R_src^T · R_tgt
          Rd[i * 3 + j] = (S[0 * 4 + i] * T[0 * 4 + j] + S[1 * 4 + i] * T[1 * 4 + j]) + S[2 * 4 + i] * T[2 * 4 + j];
        else                 // R_tgt · R_src^T
          Rd[i * 3 + j] = (T[i * 4 + 0] * S[j * 4 + 0] + T[i * 4 + 1] * S[j * 4 + 1]) + T[i * 4 + 2] * S[j * 4 + 2];
      }
    const float sx = S[3], sy = S[7], sz = S[11], tx = T[3], ty = T[7], tz = T[11];
    double d[3];
    if (rc == RC_CAMERA_NEW) {
      d[0] = (double)((tx - sx) / sz);
      d[1] = (double)((ty - sy) / sz);
    } else {
      d[0] = (double)(tx / tz - sx / sz);
      d[1] = (double)(ty / tz - sy / sz);
    }
    d[2] = (double)logf(sz / tz);
    for (int i = 0; i < 3; ++i) Td[i] = (d[i] - mu.v[i]) / sd.v[i];
  }
  if (rot_type == 0) {
    double q[4];
    mat2quat_f64(Rd, q);
    for (int i = 0; i < 4; ++i) rot[b * 4 + i] = (float)q[i];
  } else if (rot_type == 1) {
    double Md[9], e[3];
    for (int i = 0; i < 9; ++i) Md[i] = (double)Rd[i];
    mat2euler_sxyz_f64(Md, e);
    for (int i = 0; i < 3; ++i) rot[b * 3 + i] = (float)e[i];
  } else {
    for (int i = 0; i < 9; ++i) rot[b * 9 + i] = Rd[i];
  }
  for (int i = 0; i < 3; ++i) trans[b * 3 + i] = (float)Td[i];
}

// quat2mat / mat2quat / euler2mat / mat2euler / calc_se3 as stand-alone batched ops (the reference-named Python entries of
// lib/pair_matching/RT_transform.py bind to these). op: 0 quat2mat (B,4)->(B,9) f64, 1 mat2quat (B,9)->(B,4) f64,
// 2 euler2mat (B,3)->(B,9) f64, 3 mat2euler (B,9)->(B,3) f64
__global__ void rot_convert_kernel(double* __restrict__ out, const float* __restrict__ in, int op, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (op == 0) {
    const float* q = in + b * 4;
    quat2mat_checked_f64(q[0], q[1], q[2], q[3], out + b * 9);
  } else if (op == 1) {
    mat2quat_f64(in + b * 9, out + b * 4);
  } else if (op == 2) {
    euler2mat_sxyz_f64((double)in[b * 3], (double)in[b * 3 + 1], (double)in[b * 3 + 2], out + b * 9);
  } else {
    double Md[9];
    for (int i = 0; i < 9; ++i) Md[i] = (double)in[b * 9 + i];
    mat2euler_sxyz_f64(Md, out + b * 3);
  }
}
__global__ void calc_se3_kernel(float* __restrict__ rotm, float* __restrict__ t, const float* __restrict__ pose_src,
                                const float* __restrict__ pose_tgt, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  se3_src2tgt_f32(pose_src + b * 12, pose_tgt + b * 12, rotm + b * 9, t + b * 3);
}

// --- Transform3D --------------------------------------------------------------
// quat2mat_forward (transform3d.py:185-212): identity unless |Nq-1| < 1e-2; float64 math, float32 result
__device__ void t3d_quat2mat(const float* q, float* M) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  const float Nq = ((w * w + x * x) + y * y) + z * z;
  const double dn = (double)(Nq - 1.f);
  if (!(-1e-2 < dn && dn < 1e-2)) {
    for (int i = 0; i < 9; ++i) M[i] = (i % 4 == 0) ? 1.f : 0.f;
    return;
  }
  double Md[9];
  quat2mat_f64(w, x, y, z, Md);
  for (int i = 0; i < 9; ++i) M[i] = (float)Md[i];
}

// per-sample R_tgt (float32 3x3) and t_tgt (float32 3) → rt (B,12) as [R | t] rows
__global__ void t3d_prepare_kernel(float* __restrict__ rt, const float* __restrict__ rotation,
                                   const float* __restrict__ translation, const float* __restrict__ pose_src, Vec3d mu,
                                   Vec3d sd, int rc, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* P = pose_src + b * 12;
  float Rd[9];
  t3d_quat2mat(rotation + b * 4, Rd);
  float* o = rt + b * 12;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float v;
      if (rc == RC_MODEL) v = P[i * 4 + 0] * Rd[0 * 3 + j] + P[i * 4 + 1] * Rd[1 * 3 + j] + P[i * 4 + 2] * Rd[2 * 3 + j];
      else v = Rd[i * 3 + 0] * P[0 * 4 + j] + Rd[i * 3 + 1] * P[1 * 4 + j] + Rd[i * 3 + 2] * P[2 * 4 + j];
      o[i * 4 + j] = v;
    }
  const float Tsrc[3] = {P[3], P[7], P[11]};
  const float* Td = translation + b * 3;
  if (rc == RC_NAIVE) {  // T_transform_naive (RT_transform.py:98-102), float32
    for (int i = 0; i < 3; ++i)
      o[i * 4 + 3] = (Rd[i * 3 + 0] * Tsrc[0] + Rd[i * 3 + 1] * Tsrc[1] + Rd[i * 3 + 2] * Tsrc[2]) + Td[i];
  } else {
    double T[3];
    t_transform_f64(Tsrc, Td, mu, sd, rc, T);
    o[3] = (float)T[0]; o[7] = (float)T[1]; o[11] = (float)T[2];
  }
}

__global__ __launch_bounds__(256) void t3d_apply_kernel(float* __restrict__ out, const float* __restrict__ pts,
                                                        const float* __restrict__ rt, int N) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float* R = rt + b * 12;
  const float* p = pts + (long)b * 3 * N;
  const float x = p[n], y = p[N + n], z = p[2 * N + n];
  float* o = out + (long)b * 3 * N;
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i * N + n] = ((R[i * 4 + 0] * x + R[i * 4 + 1] * y) + R[i * 4 + 2] * z) + R[i * 4 + 3];
}

// get_point_cloud_observed (lib/pair_matching/data_pair.py): np.dot(R, points_model) + T — float64 (the model points
// are float64 there), stored float32
__global__ __launch_bounds__(256) void points_transform_kernel(float* __restrict__ out, const float* __restrict__ pts,
                                                               const float* __restrict__ pose, int N) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float* R = pose + b * 12;
  const float* p = pts + (long)b * 3 * N;
  const double x = p[n], y = p[N + n], z = p[2 * N + n];
  float* o = out + (long)b * 3 * N;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    o[i * N + n] = (float)((((double)R[i * 4 + 0] * x + (double)R[i * 4 + 1] * y) + (double)R[i * 4 + 2] * z) + (double)R[i * 4 + 3]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one block per sample: 3 + 9 reductions over N, then the quaternion / translation chain rule
__global__ __launch_bounds__(256) void t3d_backward_kernel(float* __restrict__ d_rot, float* __restrict__ d_trans,
                                                           const float* __restrict__ og, const float* __restrict__ pts,
                                                           const float* __restrict__ rotation,
                                                           const float* __restrict__ translation,
                                                           const float* __restrict__ pose_src, Vec3 mu, Vec3 sd, int rc,
                                                           int N) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* g = og + (long)b * 3 * N;
  const float* p = pts + (long)b * 3 * N;
  const float* P = pose_src + b * 12;
  float s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = 0.f;
  for (int n = tid; n < N; n += 256) {
    float gv[3] = {g[n], g[N + n], g[2 * N + n]};
    float pv[3] = {p[n], p[N + n], p[2 * N + n]};
    if (rc == RC_NAIVE) {  // src_3d_points = Rm_src·P + T_src (transform3d.py:131-133)
      float q[3];
      for (int i = 0; i < 3; ++i) q[i] = ((P[i * 4 + 0] * pv[0] + P[i * 4 + 1] * pv[1]) + P[i * 4 + 2] * pv[2]) + P[i * 4 + 3];
      for (int i = 0; i < 3; ++i) pv[i] = q[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      s[i] += gv[i];
#pragma unroll
      for (int j = 0; j < 3; ++j) s[3 + i * 3 + j] = fmaf(gv[i], pv[j], s[3 + i * 3 + j]);
    }
  }
  __shared__ float red[4][12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float v = wave_sum(s[i]);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (tid != 0) return;
  float tot[12];
  for (int i = 0; i < 12; ++i) tot[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  const float* Dt = tot;      // T_tgt_diff (3)
  const float* Dr = tot + 3;  // Rm_tgt_diff (3x3) — or Rm_delta_diff directly for NAIVE
  // ---- translation ----
  float* dt = d_trans + b * 3;
  const float* Td = translation + b * 3;
  if (rc == RC_NAIVE) {
    dt[0] = Dt[0]; dt[1] = Dt[1]; dt[2] = Dt[2];
  } else {
    const float Tsrc[3] = {P[3], P[7], P[11]};
    float d1[3];
    for (int i = 0; i < 3; ++i) d1[i] = Td[i] * sd.v[i] + mu.v[i];
    const float z2 = Tsrc[2] / expf(d1[2]);
    if (rc == RC_CAMERA || rc == RC_MODEL) {
      dt[0] = (Dt[0] * (sd.v[0] * z2) + Dt[1] * 0.f) + Dt[2] * 0.f;
      dt[1] = (Dt[0] * 0.f + Dt[1] * (sd.v[1] * z2)) + Dt[2] * 0.f;
      const float share = -sd.v[2] * z2;
      dt[2] = (Dt[0] * (share * (d1[0] + Tsrc[0] / Tsrc[2])) + Dt[1] * (share * (d1[1] + Tsrc[1] / Tsrc[2]))) +
              Dt[2] * (-sd.v[2] * z2);
    } else {
      dt[0] = (Dt[0] * (sd.v[0] * Tsrc[2]) + Dt[1] * 0.f) + Dt[2] * 0.f;
      dt[1] = (Dt[0] * 0.f + Dt[1] * (sd.v[1] * Tsrc[2])) + Dt[2] * 0.f;
      dt[2] = (Dt[0] * 0.f + Dt[1] * 0.f) + Dt[2] * (-sd.v[2] * z2);
    }
  }
  // ---- rotation: Rm_delta_diff then quat2mat_backward ----
  float D[9];
  if (rc == RC_MODEL) {  // Rm_src^T · Rm_tgt_diff
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) D[i * 3 + j] = (P[0 * 4 + i] * Dr[0 * 3 + j] + P[1 * 4 + i] * Dr[1 * 3 + j]) + P[2 * 4 + i] * Dr[2 * 3 + j];
  } else if (rc == RC_NAIVE) {
    for (int i = 0; i < 9; ++i) D[i] = Dr[i];
  } else {  // Rm_tgt_diff · Rm_src^T
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) D[i * 3 + j] = (Dr[i * 3 + 0] * P[j * 4 + 0] + Dr[i * 3 + 1] * P[j * 4 + 1]) + Dr[i * 3 + 2] * P[j * 4 + 2];
  }
  const float* q = rotation + b * 4;
  float* dq = d_rot + b * 4;
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  const float Nq = ((w * w + x * x) + y * y) + z * z;
  const double dn = (double)(Nq - 1.f);
  if (!(-1e-4 < dn && dn < 1e-4)) { dq[0] = dq[1] = dq[2] = dq[3] = 0.f; return; }
  const float Ns = sqrtf(Nq);
  const float w_ = w / Ns, x_ = x / Ns, y_ = y / Ns, z_ = z / Ns;
  // transform3d.py:223-270 — float32 nine-term sums, left to right
  float wd = 0.f * D[0];
  wd = wd - z_ * D[1]; wd = wd + y_ * D[2]; wd = wd + z_ * D[3]; wd = wd + 0.f * D[4];
  wd = wd - x_ * D[5]; wd = wd - y_ * D[6]; wd = wd + x_ * D[7]; wd = wd + 0.f * D[8];
  float xd = 0.f * D[0];
  xd = xd + y_ * D[1]; xd = xd + z_ * D[2]; xd = xd + y_ * D[3]; xd = xd - (2.f * x_) * D[4];
  xd = xd - w_ * D[5]; xd = xd + z_ * D[6]; xd = xd + w_ * D[7]; xd = xd - (2.f * x_) * D[8];
  float yd = (-2.f * y_) * D[0];
  yd = yd + x_ * D[1]; yd = yd + w_ * D[2]; yd = yd + x_ * D[3]; yd = yd + 0.f * D[4];
  yd = yd + z_ * D[5]; yd = yd - w_ * D[6]; yd = yd + z_ * D[7]; yd = yd - (2.f * y_) * D[8];
  float zd = (-2.f * z_) * D[0];
  zd = zd - w_ * D[1]; zd = zd + x_ * D[2]; zd = zd + w_ * D[3]; zd = zd - (2.f * z_) * D[4];
  zd = zd + y_ * D[5]; zd = zd + x_ * D[6]; zd = zd + y_ * D[7]; zd = zd + 0.f * D[8];
  const double wD = (double)wd * 2.0, xD = (double)xd * 2.0, yD = (double)yd * 2.0, zD = (double)zd * 2.0;
  const float Ns3 = Ns * Ns * Ns;  // Nq_sqrt ** 3, float32
  const double share = (double)Ns3 * ((((double)w * wD + (double)x * xD) + (double)y * yD) + (double)z * zD);
  dq[0] = (float)((double)Ns * wD - (double)w * share);
  dq[1] = (float)((double)Ns * xD - (double)x * share);
  dq[2] = (float)((double)Ns * yD - (double)y * share);
  dq[3] = (float)((double)Ns * zD - (double)z * share);
}

// --- pose-error metrics (lib/utils/pose_error.py) ------------------------------------
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one block per pair; est/gt transformed points staged in LDS tiles for the brute-force nearest neighbour
constexpr int PE_TILE = 1024;
__global__ __launch_bounds__(256) void pose_error_kernel(float* __restrict__ out, const float* __restrict__ pose_est,
                                                         const float* __restrict__ pose_gt, const float* __restrict__ pts,
                                                         long pts_bstride, Mat3 K, int N) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* E = pose_est + b * 12;
  const float* G = pose_gt + b * 12;
  const float* p = pts + (long)b * pts_bstride;
  __shared__ float est_tile[3][PE_TILE];
  __shared__ double red[4][3];
  double Ed[12], Gd[12];
  for (int i = 0; i < 12; ++i) { Ed[i] = E[i]; Gd[i] = G[i]; }
  double s_add = 0.0, s_adi = 0.0, s_arp = 0.0;
  for (int n0 = 0; n0 < N; n0 += 256) {   // every thread owns one gt point per pass
    const int n = n0 + tid;
    double ge[3] = {0, 0, 0};
    double best = 1e300;
    if (n < N) {
      const double x = p[n], y = p[N + n], z = p[2 * N + n];
      double ee[3];
      for (int i = 0; i < 3; ++i) {
        ee[i] = Ed[i * 4] * x + Ed[i * 4 + 1] * y + Ed[i * 4 + 2] * z + Ed[i * 4 + 3];
        ge[i] = Gd[i * 4] * x + Gd[i * 4 + 1] * y + Gd[i * 4 + 2] * z + Gd[i * 4 + 3];
      }
      s_add += sqrt((ee[0] - ge[0]) * (ee[0] - ge[0]) + (ee[1] - ge[1]) * (ee[1] - ge[1]) + (ee[2] - ge[2]) * (ee[2] - ge[2]));
      double pe[3], pg[3];
      for (int i = 0; i < 3; ++i) {
        pe[i] = (double)K.v[i * 3] * ee[0] + (double)K.v[i * 3 + 1] * ee[1] + (double)K.v[i * 3 + 2] * ee[2];
        pg[i] = (double)K.v[i * 3] * ge[0] + (double)K.v[i * 3 + 1] * ge[1] + (double)K.v[i * 3 + 2] * ge[2];
      }
      const double du = pe[0] / pe[2] - pg[0] / pg[2], dv = pe[1] / pe[2] - pg[1] / pg[2];
      s_arp += sqrt(du * du + dv * dv);
    }
    // nearest est point for this gt point: sweep all est points through LDS (float32 staging, float64 distance)
    for (int m0 = 0; m0 < N; m0 += PE_TILE) {
      __syncthreads();
      for (int m = tid; m < PE_TILE; m += 256) {
        const int mm = m0 + m;
        if (mm < N) {
          const double x = p[mm], y = p[N + mm], z = p[2 * N + mm];
          for (int i = 0; i < 3; ++i)
            est_tile[i][m] = (float)(Ed[i * 4] * x + Ed[i * 4 + 1] * y + Ed[i * 4 + 2] * z + Ed[i * 4 + 3]);
        }
      }
      __syncthreads();
      if (n < N) {
        const int cnt = min(PE_TILE, N - m0);
        for (int m = 0; m < cnt; ++m) {
          const double dx = (double)est_tile[0][m] - ge[0], dy = (double)est_tile[1][m] - ge[1], dz = (double)est_tile[2][m] - ge[2];
          best = fmin(best, dx * dx + dy * dy + dz * dz);
        }
      }
    }
    if (n < N) s_adi += sqrt(best);
  }
  double v[3] = {wave_sum_d(s_add), wave_sum_d(s_adi), wave_sum_d(s_arp)};
  __syncthreads();
  if (lane == 0) for (int i = 0; i < 3; ++i) red[wave][i] = v[i];
  __syncthreads();
  if (tid == 0) {
    double tot[3];
    for (int i = 0; i < 3; ++i) tot[i] = ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i])) / N;
    // re: angle of R_est^T R_gt; te: ||t_gt - t_est||
    double tr = 0.0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) tr += Ed[j * 4 + i] * Gd[j * 4 + i];
    const double c = fmin(1.0, fmax(-1.0, (tr - 1.0) * 0.5));
    const double dtx = Gd[3] - Ed[3], dty = Gd[7] - Ed[7], dtz = Gd[11] - Ed[11];
    float* o = out + b * 5;
    o[0] = (float)(acos(c) * 180.0 / 3.141592653589793);
    o[1] = (float)sqrt(dtx * dtx + dty * dty + dtz * dtz);
    o[2] = (float)tot[0];
    o[3] = (float)tot[1];
    o[4] = (float)tot[2];
  }
}

Vec3d vec3d_from(const float* h, double dflt) {
  Vec3d v;
  for (int i = 0; i < 3; ++i) v.v[i] = h ? (double)h[i] : dflt;
  return v;
}
Vec3 vec3_from(const float* h, float dflt) {
  Vec3 v;
  for (int i = 0; i < 3; ++i) v.v[i] = h ? h[i] : dflt;
  return v;
}

}  // namespace

extern "C" int deepim_rt_transform(deepim_ctx* ctx, float* pose_est, double* pose_est64, const float* pose_src,
                                   const float* se3, const float* T_means_host, const float* T_stds_host,
                                   int rot_coord, int B) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(rot_coord >= 0 && rot_coord <= 3, "rt_transform: unknown rot_coord");
  hipLaunchKernelGGL(rt_transform_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, pose_est, pose_est64,
                     pose_src, se3, vec3d_from(T_means_host, 0.0), vec3d_from(T_stds_host, 1.0), rot_coord, B, 4);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_pose_tail_forward(deepim_ctx* ctx, float* fc7_out, float* se3, float* pose_est, const float* fc6,
                                        const float* w_fc7, const float* b_fc7, const float* w_rot, const float* b_rot,
                                        const float* w_trans, const float* b_trans, const float* zoom_factor,
                                        const float* pose_src, const float* T_means_host, const float* T_stds_host, int rot_coord,
                                        int B, int feat, float slope) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(feat == 256, "pose_tail: built for the network's 256-wide fc6 / fc7 (deepIM_flownet.py:112-116)");
  DI_REQUIRE(rot_coord >= 0 && rot_coord <= 3, "pose_tail: unknown rot_coord");
  DI_REQUIRE(fc7_out && se3 && pose_est && fc6 && w_fc7 && w_rot && w_trans && b_rot && b_trans && zoom_factor && pose_src,
             "pose_tail: NULL argument");
  hipLaunchKernelGGL(pose_tail_kernel, dim3(B), dim3(1024), 0, ctx->stream, fc7_out, se3, pose_est, fc6, w_fc7, b_fc7, w_rot, b_rot,
                     w_trans, b_trans, zoom_factor, pose_src, vec3d_from(T_means_host, 0.0), vec3d_from(T_stds_host, 1.0), rot_coord,
                     slope);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_rt_transform_euler(deepim_ctx* ctx, float* pose_est, double* pose_est64, const float* pose_src,
                                         const float* euler_trans, const float* T_means_host, const float* T_stds_host,
                                         int rot_coord, int B) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(rot_coord >= 0 && rot_coord <= 3, "rt_transform: unknown rot_coord");
  hipLaunchKernelGGL(rt_transform_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, pose_est, pose_est64,
                     pose_src, euler_trans, vec3d_from(T_means_host, 0.0), vec3d_from(T_stds_host, 1.0), rot_coord, B, 3);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_rot_convert(deepim_ctx* ctx, double* out, const float* in, int op, int B) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(op >= 0 && op <= 3, "rot_convert: unknown op");
  hipLaunchKernelGGL(rot_convert_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, out, in, op, B);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_points_transform(deepim_ctx* ctx, float* out, const float* points, const float* pose, int B, int N) {
  DI_DEVICE(ctx);
  if (B == 0 || N == 0) return 0;
  hipLaunchKernelGGL(points_transform_kernel, dim3(di_div_up(N, 256), B), dim3(256), 0, ctx->stream, out, points, pose, N);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_calc_se3(deepim_ctx* ctx, float* rotm, float* t, const float* pose_src, const float* pose_tgt,
                               int B) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  hipLaunchKernelGGL(calc_se3_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, rotm, t, pose_src, pose_tgt, B);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_calc_rt_delta(deepim_ctx* ctx, float* rot, float* trans, const float* pose_src,
                                    const float* pose_tgt, const float* T_means_host, const float* T_stds_host,
                                    int rot_coord, int B) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(rot_coord >= 0 && rot_coord <= 3, "calc_rt_delta: unknown rot_coord");
  hipLaunchKernelGGL(calc_rt_delta_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, rot, trans, pose_src,
                     pose_tgt, vec3d_from(T_means_host, 0.0), vec3d_from(T_stds_host, 1.0), rot_coord, B, 0);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_calc_rt_delta_ex(deepim_ctx* ctx, float* rot, float* trans, const float* pose_src,
                                       const float* pose_tgt, const float* T_means_host, const float* T_stds_host,
                                       int rot_coord, int rot_type, int B) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(rot_coord >= 0 && rot_coord <= 3, "calc_rt_delta: unknown rot_coord");
  DI_REQUIRE(rot_type >= 0 && rot_type <= 2, "calc_rt_delta: unknown rot_type");
  hipLaunchKernelGGL(calc_rt_delta_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, rot, trans, pose_src,
                     pose_tgt, vec3d_from(T_means_host, 0.0), vec3d_from(T_stds_host, 1.0), rot_coord, B, rot_type);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_pose_error(deepim_ctx* ctx, float* out, const float* pose_est, const float* pose_gt,
                                 const float* points, int points_shared, const float* K_host, int B, int N) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(N > 0, "pose_error: need at least one model point");
  Mat3 K;
  for (int i = 0; i < 9; ++i) K.v[i] = K_host[i];
  hipLaunchKernelGGL(pose_error_kernel, dim3(B), dim3(256), 0, ctx->stream, out, pose_est, pose_gt, points,
                     points_shared ? 0L : (long)3 * N, K, N);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_transform3d_forward(deepim_ctx* ctx, float* out, const float* points, const float* rotation,
                                          const float* translation, const float* pose_src,
                                          const float* T_means_host, const float* T_stds_host, int rot_coord, int B,
                                          int N) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(rot_coord >= 0 && rot_coord <= 3, "transform3d: unknown rot_coord");
  void* scratch;
  int rc = deepim_scratch(ctx, (size_t)B * 12 * sizeof(float), &scratch);
  if (rc) return rc;
  float* rt = (float*)scratch;
  hipLaunchKernelGGL(t3d_prepare_kernel, dim3(di_div_up(B, 64)), dim3(64), 0, ctx->stream, rt, rotation, translation,
                     pose_src, vec3d_from(T_means_host, 0.0), vec3d_from(T_stds_host, 1.0), rot_coord, B);
  hipLaunchKernelGGL(t3d_apply_kernel, dim3(di_div_up(N, 256), B), dim3(256), 0, ctx->stream, out, points, rt, N);
  DI_LAUNCH_CHECK();
  return 0;
}

extern "C" int deepim_transform3d_backward(deepim_ctx* ctx, float* d_rotation, float* d_translation,
                                           const float* out_grad, const float* points, const float* rotation,
                                           const float* translation, const float* pose_src,
                                           const float* T_means_host, const float* T_stds_host, int rot_coord, int B,
                                           int N) {
  DI_DEVICE(ctx);
  if (B == 0) return 0;
  DI_REQUIRE(rot_coord >= 0 && rot_coord <= 3, "transform3d: unknown rot_coord");
  hipLaunchKernelGGL(t3d_backward_kernel, dim3(B), dim3(256), 0, ctx->stream, d_rotation, d_translation, out_grad,
                     points, rotation, translation, pose_src, vec3_from(T_means_host, 0.f), vec3_from(T_stds_host, 1.f),
                     rot_coord, N);
  DI_LAUNCH_CHECK();
  return 0;
}
