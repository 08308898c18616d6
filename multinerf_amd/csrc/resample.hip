// Hierarchical interval resampling, one level (gfx950).
//
// Replaces, for a batch of rays, the chain of reference internal/models.py:153-204:
//   stepfun.max_dilate_weights (stepfun.py:99-128)   [optional]
//   anneal + logits            (models.py:173-185)
//   stepfun.sample_intervals   (stepfun.py:164-263: softmax, integrate_weights,
//                               math.sorted_interp math.py:108-127, midpoints)
//   s_to_t                     (coord.py:63-99)
//
// The reference materialises O(n^2)-per-ray boolean masks ([B,3n+1,n] for the
// dilation, [B,n_cdf,n_samples] x4 for the interpolation).  Every list involved
// is sorted, so here each ray is ONE lane walking monotone cursors: the sort of
// concat[t, t0-d, t1+d] is a 3-way merge, the dilation max is a sliding window,
// the inverse CDF is a merge of the sorted u against the sorted cw.  A ray's
// working arrays live in LDS in [element][ray] order (bank-conflict-free when
// lanes index the same element, which the common-trip-count loops do).
//
// Summation-order contract (bit-exact sample indices): the softmax denominator
// and the CDF are accumulated strictly left to right in fp32, and this file is
// compiled with floating-point contraction OFF (no FMA fusing), matching
// oracle/stepfun.py::softmax_seq / integrate_weights.
#include "common.h"

#pragma clang fp contract(off)

#define RS_THREADS 64

// LDS layout per ray (floats), element i of an array at [i * rpb + ray]:
//   region A: T[np+1] | spare | W[np]                       (a_len >= 2np+2)
//   region B: dilation:    TD[3np+1] | WD[3np]              (b_len = 6np+1)
//             no dilation: centers[n] .. | sdist[n+1]       (b_len = 2n+2)
// With dilation, region A is dead after the dilation and hosts centers/sdist.
// The CDF (nb+1 fence-posts for nb bins) is built in place one slot BEFORE the
// array of bin weights it integrates: WD[0] (the trimmed-away first weight) with
// dilation, the spare slot without.
struct RsLayout {
  int rpb;     // rays per block (<= 64)
  int a_len;
  int b_len;
};

__device__ __forceinline__ float rs_s_to_t(int fn, float s, float near, float far) {
  // coord.py:96-98: fn_inv(s * fn(far) + (1 - s) * fn(near)).
  float fn_near, fn_far;
  switch (fn) {
    case MNR_RAYDIST_RECIPROCAL: fn_near = 1.0f / near; fn_far = 1.0f / far; break;
    case MNR_RAYDIST_PIECEWISE:
      fn_near = near < 1.0f ? 0.5f * near : 1.0f - 0.5f / near;
      fn_far = far < 1.0f ? 0.5f * far : 1.0f - 0.5f / far;
      break;
    case MNR_RAYDIST_LOG: fn_near = logf(near); fn_far = logf(far); break;
    case MNR_RAYDIST_EXP: fn_near = expf(near); fn_far = expf(far); break;
    case MNR_RAYDIST_SQRT: fn_near = sqrtf(near); fn_far = sqrtf(far); break;
    case MNR_RAYDIST_SQUARE: fn_near = near * near; fn_far = far * far; break;
    default: fn_near = near; fn_far = far; break;
  }
  const float x = s * fn_far + (1.0f - s) * fn_near;
  switch (fn) {
    case MNR_RAYDIST_RECIPROCAL: return 1.0f / x;
    case MNR_RAYDIST_PIECEWISE: return x < 0.5f ? 2.0f * x : 0.5f / (1.0f - x);
    case MNR_RAYDIST_LOG: return expf(x);
    case MNR_RAYDIST_EXP: return logf(x);
    case MNR_RAYDIST_SQRT: return x * x;
    case MNR_RAYDIST_SQUARE: return sqrtf(x);
    default: return x;
  }
}

// stepfun.max_dilate_weights(renormalize=True): (t[0..n], w[0..n-1]) -> (td[0..3n], wd[0..3n-1]).
// `p` holds w on entry and is overwritten by the pdf.  All arrays indexed [i * stride].
__device__ __forceinline__ void rs_max_dilate(int n, const float* t, float* p, float* td, float* wd,
                                              int stride, float dilation, float lo, float hi) {
  const float eps2 = MNR_F32_EPS * MNR_F32_EPS;
  // stepfun.py:89-91: pdf = w / max(eps^2, dt).
  for (int j = 0; j < n; ++j) {
    const float dt = t[(j + 1) * stride] - t[j * stride];
    p[j * stride] = p[j * stride] / fmaxf(eps2, dt);
  }
  // stepfun.py:101-104: sort(concat[t, t[:-1]-d, t[1:]+d]) = 3-way merge of sorted lists; clip.
  int ia = 0, ib = 0, ic = 0;
  const int m = 3 * n + 1;
  for (int k = 0; k < m; ++k) {
    const float va = ia <= n ? t[ia * stride] : INFINITY;
    const float vb = ib < n ? t[ib * stride] - dilation : INFINITY;
    const float vc = ic < n ? t[(ic + 1) * stride] + dilation : INFINITY;
    float v;
    if (vb <= va && vb <= vc) { v = vb; ++ib; }
    else if (va <= vc) { v = va; ++ia; }
    else { v = vc; ++ic; }
    td[k * stride] = fminf(fmaxf(v, lo), hi);
  }
  // stepfun.py:105-112: wd[k] = max_j { p[j] : t0[j] <= td[k] < t1[j] } for k < 3n.
  // t0 and t1 ascend, so the admissible j form a window [jlo, jhi] that only moves right.
  int jlo = 0, jhi = -1;
  for (int k = 0; k < m - 1; ++k) {
    const float x = td[k * stride];
    while (jhi + 1 < n && t[(jhi + 1) * stride] - dilation <= x) ++jhi;
    while (jlo < n && !(t[(jlo + 1) * stride] + dilation > x)) ++jlo;
    float best = 0.0f;
    for (int j = jlo; j <= jhi; ++j) best = fmaxf(best, p[j * stride]);
    wd[k * stride] = best;
  }
  // stepfun.py:125-127: back to weights, renormalise by max(eps^2, sum).
  float sum = 0.0f;
  for (int k = 0; k < m - 1; ++k) {
    const float w = wd[k * stride] * (td[(k + 1) * stride] - td[k * stride]);
    wd[k * stride] = w;
    sum += w;
  }
  const float denom = fmaxf(eps2, sum);
  for (int k = 0; k < m - 1; ++k) wd[k * stride] = wd[k * stride] / denom;
}

// One query of math.sorted_interp (math.py:108-127).  `i` is a cursor holding the last
// index with xp[i] <= x (or -1); since the queries ascend it moves O(1) amortised.
__device__ __forceinline__ float rs_interp_one(float x, const float* xp, const float* fp, int stride, int nc,
                                               int& i) {
  while (i >= 0 && xp[i * stride] > x) --i;
  while (i + 1 < nc && xp[(i + 1) * stride] <= x) ++i;
  const int i0 = i >= 0 ? i : 0;                 // no True in the mask -> v[0]
  const int i1 = i + 1 < nc ? i + 1 : nc - 1;    // no False in the mask -> v[-1]
  const float x0 = xp[i0 * stride], x1 = xp[i1 * stride];
  const float f0 = fp[i0 * stride], f1 = fp[i1 * stride];
  const float off = mnr_nan0_clip01((x - x0) / (x1 - x0));
  return f0 + off * (f1 - f0);
}

__global__ __launch_bounds__(RS_THREADS) void resample_level_kernel(
    mnr_resample_cfg c, int64_t B, RsLayout lay, const float* __restrict__ sdist_prev,
    const float* __restrict__ w_prev, const float* __restrict__ u_base, const float* __restrict__ jitter,
    const float* __restrict__ near, const float* __restrict__ far, float* __restrict__ sdist_out,
    float* __restrict__ tdist_out, int32_t* __restrict__ idx_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int rpb = lay.rpb;
  float* regA = lds;
  float* regB = lds + lay.a_len * rpb;
  const int64_t ray0 = (int64_t)blockIdx.x * rpb;
  const int nrays = (int)min((int64_t)rpb, B - ray0);
  const int np = c.n_prev, n = c.n_samples;
  const int w_off = np + 2;                      // T[np+1] | spare | W

  // Coalesced load of the block's rows into [elem][ray] order.
  for (int e = threadIdx.x; e < nrays * (np + 1); e += RS_THREADS) {
    const int r = e / (np + 1), i = e % (np + 1);
    regA[i * rpb + r] = sdist_prev[ray0 * (np + 1) + e];
  }
  for (int e = threadIdx.x; e < nrays * np; e += RS_THREADS) {
    const int r = e / np, i = e % np;
    regA[(w_off + i) * rpb + r] = w_prev[ray0 * np + e];
  }
  __syncthreads();

  float* out_reg = c.use_dilation ? regA : regB;  // centers[n] | sdist[n+1]
  const int r = threadIdx.x;
  if (r < nrays) {
    float* t = regA + r;
    float* w = regA + w_off * rpb + r;
    float* td;
    float* wd;
    int nb;                                      // bins of the histogram being sampled
    if (c.use_dilation) {
      float* TD = regB + r;
      float* WD = regB + (3 * np + 1) * rpb + r;
      rs_max_dilate(np, t, w, TD, WD, rpb, c.dilation, c.domain_lo, c.domain_hi);
      td = TD + rpb;                             // models.py:170-171: drop first/last fence-post
      wd = WD + rpb;                             //                    and first/last weight
      nb = 3 * np - 2;
    } else {
      td = t;
      wd = w;
      nb = np;
    }
    // models.py:183-185 logits; jax.nn.softmax (stepfun.py:156) with a sequential denominator.
    float mx = -INFINITY;
    for (int k = 0; k < nb; ++k) {
      const bool open = td[(k + 1) * rpb] > td[k * rpb];
      const float lg = open ? c.anneal * logf(wd[k * rpb] + c.resample_padding) : -INFINITY;
      wd[k * rpb] = lg;
      mx = (lg != lg || mx != mx) ? NAN : fmaxf(mx, lg);      // jnp.max propagates NaN (0 * log 0 at train_frac 0)
    }
    float denom = 0.0f;
    for (int k = 0; k < nb; ++k) {
      const float e = expf(wd[k * rpb] - mx);
      wd[k * rpb] = e;
      denom += e;
    }
    // stepfun.py:146-149: cw = [0, min(1, cumsum(w[:-1])), 1] built one slot before wd:
    // iteration k reads wd[k] (slot k+1 of cw's storage) and writes cw[k] (slot k).
    float* cw = wd - rpb;
    float run = 0.0f, prev = 0.0f;
    for (int k = 0; k < nb; ++k) {
      const float wk = wd[k * rpb] / denom;
      cw[k * rpb] = prev;
      run += wk;
      prev = (run != run) ? run : fminf(1.0f, run);           // jnp.minimum propagates NaN
    }
    cw[0] = 0.0f;
    cw[nb * rpb] = 1.0f;

    // Inverse CDF (stepfun.py:153-161 -> math.py:108-127).
    float* centers = out_reg + r;
    int cur = 0;
    const float jit1 = (jitter && c.single_jitter) ? jitter[ray0 + r] * c.max_jitter : 0.0f;
    for (int j = 0; j < n; ++j) {
      float u = u_base[j];
      if (jitter) u = u + (c.single_jitter ? jit1 : jitter[(ray0 + r) * n + j] * c.max_jitter);
      centers[j * rpb] = rs_interp_one(u, cw, td, rpb, nb + 1, cur);
      if (idx_out) idx_out[(ray0 + r) * n + j] = cur;
    }
    // stepfun.py:252-262: fence-posts at midpoints; reflected + clamped ends.
    float* so = out_reg + n * rpb + r;
    const float c0 = centers[0], c1 = centers[rpb];
    const float cl = centers[(n - 1) * rpb], cl1 = centers[(n - 2) * rpb];
    so[0] = fmaxf(c.domain_lo, 2.0f * c0 - (c1 + c0) / 2.0f);
    for (int j = 1; j < n; ++j) so[j * rpb] = (centers[j * rpb] + centers[(j - 1) * rpb]) / 2.0f;
    so[n * rpb] = fminf(c.domain_hi, 2.0f * cl - (cl + cl1) / 2.0f);
  }
  __syncthreads();
  // Coalesced write-out of sdist and tdist = s_to_t(sdist).
  const float* so_base = out_reg + n * rpb;
  for (int e = threadIdx.x; e < nrays * (n + 1); e += RS_THREADS) {
    const int rr = e / (n + 1), i = e % (n + 1);
    const float s = so_base[i * rpb + rr];
    sdist_out[ray0 * (n + 1) + e] = s;
    tdist_out[ray0 * (n + 1) + e] = rs_s_to_t(c.raydist_fn, s, near[ray0 + rr], far[ray0 + rr]);
  }
}

static RsLayout rs_layout(const mnr_resample_cfg* c) {
  RsLayout l;
  const int np = c->n_prev, n = c->n_samples;
  const int out_need = 2 * n + 1;                // centers[n] + sdist[n+1]
  if (c->use_dilation) {
    l.a_len = max(2 * np + 2, out_need);
    l.b_len = 6 * np + 1;
  } else {
    l.a_len = 2 * np + 2;
    l.b_len = out_need;
  }
  int rpb = 64;
  while (rpb > 1 && (size_t)(l.a_len + l.b_len) * rpb * 4 > 150 * 1024) rpb >>= 1;
  l.rpb = rpb;
  return l;
}

extern "C" int mnr_resample_level(const mnr_resample_cfg* cfg, int64_t B, const float* sdist_prev,
                                  const float* w_prev, const float* u_base, const float* jitter,
                                  const float* near, const float* far, float* sdist_out, float* tdist_out,
                                  int32_t* idx_out, void* stream) {
  MNR_CHECK_ARG(cfg && B > 0 && sdist_prev && w_prev && u_base && near && far && sdist_out && tdist_out,
                "mnr_resample_level: null argument");
  MNR_CHECK_ARG(cfg->n_samples > 1, "num_samples must be > 1, is %d.", cfg->n_samples);   // stepfun.py:239-240
  MNR_CHECK_ARG(cfg->n_prev >= 1 && cfg->n_prev <= 1024 && cfg->n_samples <= 1024,
                "mnr_resample_level: n_prev=%d / n_samples=%d out of range", cfg->n_prev, cfg->n_samples);
  MNR_CHECK_ARG(cfg->raydist_fn >= 0 && cfg->raydist_fn <= MNR_RAYDIST_SQUARE, "mnr_resample_level: bad raydist_fn");
  RsLayout lay = rs_layout(cfg);
  // One lane per ray is latency-bound; with 64 rays per wave a 16384-ray batch is only 256 waves for 1024
  // SIMDs.  Fewer rays per (64-thread) workgroup spreads the same serial work over more SIMDs.
  while (lay.rpb > 8 && B / lay.rpb < mnr_ray_wave_target()) lay.rpb >>= 1;
  const size_t lds_bytes = (size_t)(lay.a_len + lay.b_len) * lay.rpb * 4;
  MNR_CHECK_ARG(lds_bytes <= 160 * 1024, "mnr_resample_level: step function too long for LDS");
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if (mnr_attr_needed(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)resample_level_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  const int grid = mnr_cdiv(B, lay.rpb);
  hipLaunchKernelGGL(resample_level_kernel, dim3(grid), dim3(RS_THREADS), lds_bytes, (hipStream_t)stream, *cfg,
                     B, lay, sdist_prev, w_prev, u_base, jitter, near, far, sdist_out, tdist_out, idx_out);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// Parity-test leaves (one lane per ray, arrays in global memory, stride 1).

__global__ void sorted_interp_kernel(int64_t B, int nc, int nu, const float* u, const float* cw, const float* t,
                                     float* out, int32_t* idx) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int cur = -1;
  for (int j = 0; j < nu; ++j) {
    out[b * nu + j] = rs_interp_one(u[b * nu + j], cw + b * nc, t + b * nc, 1, nc, cur);
    if (idx) idx[b * nu + j] = cur;
  }
}

extern "C" int mnr_sorted_interp(int64_t B, int nc, int nu, const float* u, const float* cw, const float* t,
                                 float* out, int32_t* idx, void* stream) {
  MNR_CHECK_ARG(B > 0 && nc > 0 && nu > 0 && u && cw && t && out, "mnr_sorted_interp: bad arguments");
  hipLaunchKernelGGL(sorted_interp_kernel, dim3(mnr_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, B, nc, nu, u,
                     cw, t, out, idx);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

__global__ void max_dilate_kernel(int64_t B, int n, const float* t, const float* w, float dilation, float lo,
                                  float hi, float* t_out, float* w_out, float* scratch) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float* p = scratch + b * n;
  for (int j = 0; j < n; ++j) p[j] = w[b * n + j];
  rs_max_dilate(n, t + b * (n + 1), p, t_out + b * (3 * n + 1), w_out + b * 3 * n, 1, dilation, lo, hi);
}

extern "C" int mnr_max_dilate_weights(int64_t B, int n, const float* t, const float* w, float dilation,
                                      float domain_lo, float domain_hi, float* t_out, float* w_out,
                                      float* scratch, void* stream) {
  MNR_CHECK_ARG(B > 0 && n > 0 && t && w && t_out && w_out && scratch, "mnr_max_dilate_weights: bad arguments");
  hipLaunchKernelGGL(max_dilate_kernel, dim3(mnr_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, B, n, t, w,
                     dilation, domain_lo, domain_hi, t_out, w_out, scratch);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}
