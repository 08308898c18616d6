// Per-ray pieces of the proposal losses (stepfun.py:30-86), shared by the stand-alone loss kernels (losses.hip) and the
// fused per-level backward kernel (render.hip: level_bwd_kernel).
#pragma once

#include "common.h"

#define LS_THREADS 64

__device__ __forceinline__ float ls_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// lossfun_outer for one ray: the histogram (t[n+1], w[n]) against the envelope (te[ne+1], we[ne]; element i at
// te[i * es] / we[i * es], so that the envelope may sit in LDS as [elem][ray]).  Writes, per fence-post k of t,
// lo_f / hi_f (searchsorted indices as floats, stride S) and per interval either the scaled gradient factor
// d loss / d w_outer (want_grad) or the loss term itself into g_or_loss (stride S); adds the loss terms to loss_sum.
__device__ __forceinline__ void ls_outer_sweep(int n, const float* t, const float* w, int ne, const float* te,
                                               const float* we, int es, float* lo_f, float* hi_f, float* g_or_loss,
                                               int S, float scale, bool want_grad, float& loss_sum) {
  // searchsorted(t_env, t) (stepfun.py:49-53) with cursors:
  //   lo = last idx with te[idx] <= v (0 if none), hi = first idx with te[idx] > v (ne if none).
  // cy[idx] = sum_{j<idx} we[j] accumulated left to right as the cursors advance.
  int lo = 0, hi = 0;
  float cy_lo = 0.0f, cy_hi = 0.0f, cy_lo_prev = 0.0f;
  const float eps = MNR_F32_EPS;
  const float te0 = te[0];
  for (int k = 0; k <= n; ++k) {
    const float v = t[k];
    while (lo + 1 <= ne && te[(lo + 1) * es] <= v) { cy_lo += we[lo * es]; ++lo; }
    const int lo_k = (te0 <= v) ? lo : 0;                     // none true -> i[0]
    const float cylo_k = (te0 <= v) ? cy_lo : 0.0f;
    while (hi <= ne && !(te[hi * es] > v)) { if (hi < ne) cy_hi += we[hi * es]; ++hi; }
    const int hi_k = hi <= ne ? hi : ne;                      // none false... -> last index
    // cy at hi_k: if hi ran past ne, cy_hi holds the full sum = cy[ne].
    lo_f[k * S] = (float)lo_k;
    hi_f[k * S] = (float)hi_k;
    if (k >= 1) {
      const float w_outer = cy_hi - cy_lo_prev;               // stepfun.py:74
      const float wi = w[k - 1];
      const float d = fmaxf(0.0f, wi - w_outer);
      const float l = d * d / (wi + eps);                     // stepfun.py:86
      loss_sum += l;
      g_or_loss[(k - 1) * S] = want_grad ? (-2.0f * d / (wi + eps)) * scale : l;
    }
    cy_lo_prev = cylo_k;
  }
}
