// Per-sample Gaussian and trigonometric helpers of the integrated positional encoding, shared by the stand-alone
// featurisation kernel (features.hip) and the fused chain's in-kernel layer-0 producer (fused_mlp.hip).
//
// Include it with floating-point contraction switched OFF (#pragma clang fp contract(off) in front of the include): both
// users must evaluate the same separately rounded operations, or the features the fused chain builds in LDS stop being
// bit-identical to the rows cast_rays_ipe_kernel writes (the explicit fmaf calls are fused in both).
#pragma once

#include "common.h"

#define FE_PI_2 1.57079632679489661923f
#define FE_100PI 314.159265358979323846f

__device__ __forceinline__ float fe_safe_sin(float x) {
  // math.py:26-28: sin(x if |x| < 100*pi else x mod 100*pi); `%` takes the divisor's sign.
  const float t = FE_100PI;
  if (!(fabsf(x) < t)) {
    float m = fmodf(x, t);
    if (m != 0.0f && (m < 0.0f)) m += t;
    x = m;
  }
  return sinf(x);
}

// math.safe_sin's argument wrap (math.py:26-28): y if |y| < 100 pi, else y mod 100 pi with the divisor's sign -- the value
// cast_rays_ipe_kernel gets from fmodf, without the library call's ~150-instruction loop (the fused chain's in-kernel
// encoding is bound by its instruction stream, not by HBM).  fmod is exact, so it can be had from one FMA: with q the
// truncated quotient, y - q t is representable and fmaf(-q, t, y) returns it unrounded.  q comes from a multiplication by
// 1/t (|error| < 0.2 for |y| < 1e8, i.e. off by at most one, and only next to a multiple of t): a quotient one too large
// shows as a remainder of the wrong sign, one too small as |r| >= t (rounding is monotonic and t is representable), and
// the remainder is then taken again from y with the corrected quotient.  Bit-identical to the fmodf path.
__device__ __forceinline__ float fe_wrap_100pi(float y) {
  const float t = FE_100PI;
  if (!(fabsf(y) < t)) {
    float m;
    if (fabsf(y) < 1e8f) {
      const float qf = truncf(y * (1.0f / FE_100PI));
      m = fmaf(-qf, t, y);
      const float toward0 = qf - copysignf(1.0f, y), away = qf + copysignf(1.0f, y);
      if ((y > 0.0f) ? (m < 0.0f) : (m > 0.0f)) m = fmaf(-toward0, t, y);
      else if (fabsf(m) >= t) m = fmaf(-away, t, y);
    } else {
      m = fmodf(y, t);
    }
    if (m != 0.0f && (m < 0.0f)) m += t;
    y = m;
  }
  return y;
}

// sin and cos of y for |y| <= 100 pi (the argument after math.safe_sin's wrap): Cody-Waite reduction by pi/2 in three parts
// (k <= 200: k * FE_PIO2_A is exact), Cephes single-precision kernels on [-pi/4, pi/4] (|error| ~1e-7, the features then go to
// bf16).  ~20 VALU operations without a branch; the library sincosf carries its large-argument reduction along.
#define FE_PIO2_A 1.5703125f
#define FE_PIO2_B 4.837512969970703125e-4f
#define FE_PIO2_C 7.54978995489188216e-8f
__device__ __forceinline__ void fe_sincos_wrapped(float y, float* sn, float* cs) {
  const float kf = rintf(y * 0.63661977236758134308f);
  const int q = (int)kf;
  float r = fmaf(-kf, FE_PIO2_A, y);
  r = fmaf(-kf, FE_PIO2_B, r);
  r = fmaf(-kf, FE_PIO2_C, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  const float s = fmaf(r * z, ps, r);
  const float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  const float c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  const float a = (q & 1) ? c : s, b = (q & 1) ? s : c;
  *sn = (q & 2) ? -a : a;
  *cs = ((q + 1) & 2) ? -b : b;
}

struct FeSample {
  float mean[3];
  float cov[6];   // xx, xy, xz, yy, yz, zz (symmetric)
};

__device__ __forceinline__ void fe_gaussian(const mnr_ipe_cfg& c, float t0, float t1, const float* o,
                                            const float* d, float radius, FeSample& g) {
  float t_mean, t_var, r_var;
  if (c.ray_shape == 0) {
    // render.py:62-70 (stable form of mip-NeRF eq. 7).
    const float mu = (t0 + t1) / 2.0f;
    const float hw = (t1 - t0) / 2.0f;
    const float denom = fmaxf(MNR_F32_EPS, 3.0f * mu * mu + hw * hw);
    const float hw2 = hw * hw, hw4 = hw2 * hw2;
    t_mean = mu + (2.0f * mu * hw2) / denom;
    t_var = hw2 / 3.0f - (4.0f / 15.0f) * hw4 * (12.0f * mu * mu - hw2) / (denom * denom);
    r_var = (mu * mu) / 4.0f + (5.0f / 12.0f) * hw2 - (4.0f / 15.0f) * hw4 / denom;
    r_var *= radius * radius;
  } else {
    // render.py:97-99.
    t_mean = (t0 + t1) / 2.0f;
    r_var = radius * radius / 4.0f;
    t_var = (t1 - t0) * (t1 - t0) / 12.0f;
  }
  // render.py:23-41 lift_gaussian(diag=False) and :126 (+ origins).
  const float dmag = fmaxf(1e-10f, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  float mean[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) mean[i] = o[i] + d[i] * t_mean;
  float cov[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float dd = d[i] * d[j];
      const float null_outer = (i == j ? 1.0f : 0.0f) - d[i] * (d[j] / dmag);
      cov[i][j] = t_var * dd + r_var * null_outer;
    }
  if (c.disable_integration) {
    t_var = 0.0f;
    r_var = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) cov[i][j] = 0.0f;
  }
  if (c.warp_contract) {
    // coord.py:21-27 and its Jacobian (what jax.linearize yields, coord.py:58-59): identity inside
    // the unit ball; outside z = s x, J = s I + cc x x^T, s = (2 sqrt(m) - 1)/m, cc = 2 (1 - sqrt(m))/m^2.
    //
    // J cov J^T is evaluated from the STRUCTURE of cov = t_var d d^T + r_var (I - d d^T/|d|^2):
    //   J cov J^T = t_var u u^T + r_var (J^2 - u u^T/|d|^2),   u = J d,
    //   J^2 = s^2 I + (2 cc / sqrt(m)) x x^T            (2 s cc + cc^2 m = 2 cc / sqrt(m)),
    //   u   = [m - 2 (1 - sqrt(m)) (|o|^2 + t (o.d))] / m^2 * d + cc (x.d) o     (x = o + t d).
    // Same function as the reference's two matrix products, but without forming s*t_var*d d^T
    // (~1e5 for the far samples of 360.gin) only to cancel it against cc*(x.d)*t_var*x d^T: the
    // plain products lose ~10% of the tangential variance there in fp32.
    const float m = fmaxf(MNR_F32_EPS, mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2]);
    if (!(m <= 1.0f)) {
      const float sq = sqrtf(m);
      const float s = (2.0f * sq - 1.0f) / m;
      const float cc = 2.0f * (1.0f - sq) / (m * m);
      const float oo = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
      const float od = o[0] * d[0] + o[1] * d[1] + o[2] * d[2];
      const float xd = mean[0] * d[0] + mean[1] * d[1] + mean[2] * d[2];
      const float brk = (m - 2.0f * (1.0f - sq) * (oo + t_mean * od)) / (m * m);
      float u[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) u[i] = brk * d[i] + cc * xd * o[i];
      const float j2x = 2.0f * cc / sq;
      const float ku = t_var - r_var / dmag;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          cov[i][j] = ku * u[i] * u[j] + r_var * ((i == j ? s * s : 0.0f) + j2x * mean[i] * mean[j]);
#pragma unroll
      for (int i = 0; i < 3; ++i) mean[i] = s * mean[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) g.mean[i] = mean[i];
  // Symmetrise exactly the way diag(P^T C P) sees it: keep both triangles' average.
  g.cov[0] = cov[0][0];
  g.cov[1] = 0.5f * (cov[0][1] + cov[1][0]);
  g.cov[2] = 0.5f * (cov[0][2] + cov[2][0]);
  g.cov[3] = cov[1][1];
  g.cov[4] = 0.5f * (cov[1][2] + cov[2][1]);
  g.cov[5] = cov[2][2];
}
