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

// exp / log of the sampling path, written out (Cephes' single-precision expf / logf: Cody-Waite reduction + polynomial,
// every operation a separately rounded IEEE fp32 +, -, * under `fp contract(off)`) instead of the device library's, whose
// last bit differs from the host library's.  oracle/math.py restates them operation for operation in NumPy float32
// (kexp / klog), so the logits, the softmax and with them the CDF are the SAME BITS on the host and on the device, and
// the sample indices are bit-exact by construction, not by luck of the libm.  Accuracy ~1 ulp like the libraries'.
__device__ __forceinline__ float rs_pow2i(int n) { return __int_as_float((n + 127) << 23); }   // 2^n, -126 <= n <= 127

__device__ __forceinline__ float rs_exp(float x) {
  if (x != x) return x;
  if (x < -103.9720840454f) return 0.0f;               // < log(2^-150): rounds to zero
  if (x > 88.7228317261f) return INFINITY;
  const float fn = floorf(x * 1.44269504088896341f + 0.5f);
  float r = x - fn * 0.693359375f;
  r = r - fn * -2.12194440e-4f;
  const float z = r * r;
  float p = 1.9875691500E-4f;
  p = p * r + 1.3981999507E-3f;
  p = p * r + 8.3334519073E-3f;
  p = p * r + 4.1665795894E-2f;
  p = p * r + 1.6666665459E-1f;
  p = p * r + 5.0000001201E-1f;
  p = p * z + r;
  p = p + 1.0f;
  const int n = (int)fn;
  if (n >= -126) return p * rs_pow2i(n);               // exact scaling
  return (p * rs_pow2i(n + 64)) * rs_pow2i(-64);        // subnormal result: one rounding, in the second product
}

__device__ __forceinline__ float rs_log(float x) {
  if (x != x) return x;
  if (x < 0.0f) return NAN;
  if (x == 0.0f) return -INFINITY;
  if (x == INFINITY) return x;
  int e = 0;
  if (x < 1.17549435e-38f) {                            // subnormal: scale by 2^23 (exact)
    x = x * 8388608.0f;
    e = -23;
  }
  const unsigned bits = __float_as_uint(x);
  e += (int)((bits >> 23) & 0xffu) - 126;
  float m = __uint_as_float((bits & 0x007fffffu) | 0x3f000000u);     // mantissa in [0.5, 1)
  if (m < 0.707106781186547524f) {
    e -= 1;
    m = m + m - 1.0f;
  } else {
    m = m - 1.0f;
  }
  const float z = m * m;
  float y = 7.0376836292E-2f;
  y = y * m - 1.1514610310E-1f;
  y = y * m + 1.1676998740E-1f;
  y = y * m - 1.2420140846E-1f;
  y = y * m + 1.4249322787E-1f;
  y = y * m - 1.6668057665E-1f;
  y = y * m + 2.0000714765E-1f;
  y = y * m - 2.4999993993E-1f;
  y = y * m + 3.3333331174E-1f;
  y = y * m;
  y = y * z;
  const float fe = (float)e;
  y = y + -2.12194440e-4f * fe;
  y = y + -0.5f * z;
  float r = m + y;
  r = r + 0.693359375f * fe;
  return r;
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

// ---------------------------------------------------------------------------
// The level kernel: RSP_LPR = 16 lanes per ray, 4 rays per 64-lane workgroup.
//
// One lane per ray (round 1) is a ~35 k-instruction dependent stream per wave whatever the number of rays: 0.23 ms per
// level.  Here the 16 lanes of a ray split every pass:
//   * the 3-way merge of max_dilate is a RANKING: the position of an element in the sorted union is its own index plus
//     the number of elements of the other two lists in front of it (two binary searches; ties ordered b < a < c as the
//     sequential merge takes them, so the ranks are a permutation);
//   * the sliding-window max and the inverse CDF keep their monotone cursors, but every lane owns a contiguous chunk of
//     the queries and finds its first cursor position with a binary search;
//   * the three sums (renormalisation of the dilated weights, softmax denominator, CDF) are BLOCKED: a lane sums its
//     contiguous chunk of ceil(len / 16) elements left to right, the 16 chunk sums are then added left to right, and the
//     running value at element k is (sum of the chunks before k's) + (k's prefix inside its chunk).  That association
//     order is part of the contract for bit-exact sample indices and is restated in oracle/stepfun.py
//     (blocked_cumsum / blocked_sum): given the same inputs the oracle and the kernel produce the same CDF bit for bit
//     (up to the device's expf / logf), hence the same indices.
#define RSP_LPR 16
#define RSP_RPW 4

struct RspLay {
  int T, P, TD, WD, CW, CEN, SO, RED, per_ray;      // offsets (floats) inside one ray's LDS block
};

// number of i in [0, n) with a[i] + shift <= x (le) or < x; a ascending.  (t[j] - d is computed as t[j] + (-d): the same
// IEEE operation, so the searched values are exactly the merged ones.)
__device__ __forceinline__ int rsp_count(const float* a, int n, float shift, float x, bool le) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const float v = a[mid] + shift;
    if (le ? (v <= x) : (v < x)) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// Exclusive prefix of the 16 lanes' chunk sums, added left to right (identical on every lane); `total` = all 16.
__device__ __forceinline__ float rsp_chunk_offset(float chunk_sum, float* red, int l, float& total) {
  __syncthreads();
  red[l] = chunk_sum;
  __syncthreads();
  float off = 0.0f, tot = 0.0f;
#pragma unroll
  for (int i = 0; i < RSP_LPR; ++i) {
    if (i == l) off = tot;
    tot += red[i];
  }
  total = tot;
  return off;
}

// stepfun.max_dilate_weights(renormalize=True) by the 16 lanes of one ray: t[0..n], p[0..n-1] (weights on entry, pdf
// after) -> td[0..3n], wd[0..3n-1].  Every thread of the workgroup must call it (barriers inside).
__device__ __forceinline__ void rsp_max_dilate(int n, const float* t, float* p, float* td, float* wd, float* red, int l,
                                               float dilation, float lo, float hi) {
  const float eps2 = MNR_F32_EPS * MNR_F32_EPS;
  const int m = 3 * n + 1;
  // stepfun.py:89-91: pdf = w / max(eps^2, dt).
  for (int j = l; j < n; j += RSP_LPR) p[j] = p[j] / fmaxf(eps2, t[j + 1] - t[j]);
  // stepfun.py:101-104: sort(concat[a = t, b = t[:-1] - d, c = t[1:] + d]) by ranking; clip.
  for (int e = l; e < m; e += RSP_LPR) {
    float v;
    int rank;
    if (e <= n) {                                        // a[i]: b's equal to it go first, c's equal to it after
      v = t[e];
      rank = e + rsp_count(t, n, -dilation, v, true) + rsp_count(t + 1, n, dilation, v, false);
    } else if (e <= 2 * n) {                             // b[j]
      const int j = e - (n + 1);
      v = t[j] - dilation;
      rank = j + rsp_count(t, n + 1, 0.0f, v, false) + rsp_count(t + 1, n, dilation, v, false);
    } else {                                             // c[k]
      const int k = e - (2 * n + 1);
      v = t[k + 1] + dilation;
      rank = k + rsp_count(t, n + 1, 0.0f, v, true) + rsp_count(t, n, -dilation, v, true);
    }
    td[rank] = fminf(fmaxf(v, lo), hi);
  }
  __syncthreads();
  // stepfun.py:105-112: wd[k] = max_j { p[j] : t[j] - d <= td[k] < t[j+1] + d } for k < 3n; the admissible j are a
  // window [jlo, jhi] that only moves right with k: chunked queries, first window by binary search.
  const int nw = m - 1;
  const int ch = (nw + RSP_LPR - 1) / RSP_LPR;
  const int k0 = min(nw, l * ch), k1 = min(nw, k0 + ch);
  float csum = 0.0f;
  if (k0 < k1) {
    int jhi = rsp_count(t, n, -dilation, td[k0], true) - 1;
    int jlo = rsp_count(t + 1, n, dilation, td[k0], true);
    for (int k = k0; k < k1; ++k) {
      const float x = td[k];
      while (jhi + 1 < n && t[jhi + 1] - dilation <= x) ++jhi;
      while (jlo < n && !(t[jlo + 1] + dilation > x)) ++jlo;
      float best = 0.0f;
      for (int j = jlo; j <= jhi; ++j) best = fmaxf(best, p[j]);
      // stepfun.py:125-127: back to weights; the sum runs over the lane's chunk left to right
      const float w = best * (td[k + 1] - x);
      wd[k] = w;
      csum += w;
    }
  }
  float total;
  (void)rsp_chunk_offset(csum, red, l, total);
  const float denom = fmaxf(eps2, total);
  for (int k = k0; k < k1; ++k) wd[k] = wd[k] / denom;
  __syncthreads();
}

__global__ __launch_bounds__(RS_THREADS) void resample_level_kernel(
    mnr_resample_cfg c, int64_t B, RspLay lay, const float* __restrict__ sdist_prev,
    const float* __restrict__ w_prev, const float* __restrict__ u_base, const float* __restrict__ jitter,
    const float* __restrict__ near, const float* __restrict__ far, float* __restrict__ sdist_out,
    float* __restrict__ tdist_out, int32_t* __restrict__ idx_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int np = c.n_prev, n = c.n_samples;
  const int64_t ray0 = (int64_t)blockIdx.x * RSP_RPW;
  const int nrays = (int)min((int64_t)RSP_RPW, B - ray0);
  // coalesced load of the workgroup's rows (contiguous in HBM) into the per-ray blocks; a short last workgroup loads its
  // last ray again into the unused slots (every lane runs every phase: barriers inside; nothing of those slots is stored)
  for (int e = threadIdx.x; e < RSP_RPW * (np + 1); e += RS_THREADS) {
    const int r = e / (np + 1), i = e % (np + 1);
    lds[r * lay.per_ray + lay.T + i] = sdist_prev[(ray0 + min(r, nrays - 1)) * (np + 1) + i];
  }
  for (int e = threadIdx.x; e < RSP_RPW * np; e += RS_THREADS) {
    const int r = e / np, i = e % np;
    lds[r * lay.per_ray + lay.P + i] = w_prev[(ray0 + min(r, nrays - 1)) * np + i];
  }
  const int g = threadIdx.x / RSP_LPR, l = threadIdx.x % RSP_LPR;
  const int gr = min(g, nrays - 1);
  __syncthreads();
  float* base = lds + g * lay.per_ray;
  float* t = base + lay.T;
  float* w = base + lay.P;
  float* red = base + lay.RED;
  const float* td;
  float* wd;
  int nb;                                              // bins of the histogram being sampled
  if (c.use_dilation) {
    rsp_max_dilate(np, t, w, base + lay.TD, base + lay.WD, red, l, c.dilation, c.domain_lo, c.domain_hi);
    td = base + lay.TD + 1;                            // models.py:170-171: drop first/last fence-post
    wd = base + lay.WD + 1;                            //                    and first/last weight
    nb = 3 * np - 2;
  } else {
    td = t;
    wd = w;
    nb = np;
  }
  // models.py:183-185 logits; jax.nn.softmax (stepfun.py:156).  Chunked ownership: lane l holds bins [k0, k1).
  const int ch = (nb + RSP_LPR - 1) / RSP_LPR;
  const int k0 = min(nb, l * ch), k1 = min(nb, k0 + ch);
  float mx = -INFINITY;
  for (int k = k0; k < k1; ++k) {
    const bool open = td[k + 1] > td[k];
    const float lg = open ? c.anneal * rs_log(wd[k] + c.resample_padding) : -INFINITY;
    wd[k] = lg;
    mx = (lg != lg || mx != mx) ? NAN : fmaxf(mx, lg);          // jnp.max propagates NaN (0 * log 0 at train_frac 0)
  }
  __syncthreads();
  red[l] = mx;
  __syncthreads();
  mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < RSP_LPR; ++i) {
    const float v = red[i];
    mx = (v != v || mx != mx) ? NAN : fmaxf(mx, v);
  }
  float csum = 0.0f;
  for (int k = k0; k < k1; ++k) {
    const float e = rs_exp(wd[k] - mx);
    wd[k] = e;
    csum += e;
  }
  float denom;
  (void)rsp_chunk_offset(csum, red, l, denom);
  // stepfun.py:146-149: cw = [0, min(1, cumsum(w[:-1])), 1], blocked cumulative sum.
  float* cw = base + lay.CW;
  csum = 0.0f;
  for (int k = k0; k < k1; ++k) {
    const float wk = wd[k] / denom;
    wd[k] = wk;
    if (k < nb - 1) csum += wk;
  }
  float unused;
  const float off = rsp_chunk_offset(csum, red, l, unused);
  float run = 0.0f;
  for (int k = k0; k < k1 && k < nb - 1; ++k) {
    run += wd[k];
    const float cs = off + run;
    cw[k + 1] = (cs != cs) ? cs : fminf(1.0f, cs);              // jnp.minimum propagates NaN
  }
  if (l == 0) {
    cw[0] = 0.0f;
    cw[nb] = 1.0f;
  }
  __syncthreads();

  // Inverse CDF (stepfun.py:153-161 -> math.py:108-127): lane l owns samples [j0, j1); the queries ascend, so after a
  // binary search for the first one the cursor (last index with cw[i] <= u, or -1) only moves up.
  float* centers = base + lay.CEN;
  const int64_t ray = ray0 + gr;
  const int chs = (n + RSP_LPR - 1) / RSP_LPR;
  const int j0 = min(n, l * chs), j1 = min(n, j0 + chs);
  const float jit1 = (jitter && c.single_jitter) ? jitter[ray] * c.max_jitter : 0.0f;
  int cur = -2;
  for (int j = j0; j < j1; ++j) {
    float u = u_base[j];
    if (jitter) u = u + (c.single_jitter ? jit1 : jitter[ray * n + j] * c.max_jitter);
    if (cur == -2) {
      // largest i with cw[i] <= u over the sorted CDF (a NaN CDF compares false everywhere: i stays at the start)
      int lo_i = -1, hi_i = nb + 1;
      while (hi_i - lo_i > 1) {
        const int mid = (lo_i + hi_i) >> 1;
        if (cw[mid] <= u) lo_i = mid;
        else hi_i = mid;
      }
      cur = lo_i;
    }
    centers[j] = rs_interp_one(u, cw, td, 1, nb + 1, cur);
    if (idx_out && g < nrays) idx_out[ray * n + j] = cur;
  }
  __syncthreads();
  // stepfun.py:252-262: fence-posts at midpoints; reflected + clamped ends.
  float* so = base + lay.SO;
  for (int j = l; j <= n; j += RSP_LPR) {
    float v;
    if (j == 0) v = fmaxf(c.domain_lo, 2.0f * centers[0] - (centers[1] + centers[0]) / 2.0f);
    else if (j == n) v = fminf(c.domain_hi, 2.0f * centers[n - 1] - (centers[n - 1] + centers[n - 2]) / 2.0f);
    else v = (centers[j] + centers[j - 1]) / 2.0f;
    so[j] = v;
  }
  __syncthreads();
  // Coalesced write-out of sdist and tdist = s_to_t(sdist).
  for (int e = threadIdx.x; e < nrays * (n + 1); e += RS_THREADS) {
    const int rr = e / (n + 1), i = e % (n + 1);
    const float sv = lds[rr * lay.per_ray + lay.SO + i];
    sdist_out[ray0 * (n + 1) + e] = sv;
    tdist_out[ray0 * (n + 1) + e] = rs_s_to_t(c.raydist_fn, sv, near[ray0 + rr], far[ray0 + rr]);
  }
}

static RspLay rsp_layout(int np, int n) {
  RspLay l;
  const int m = 3 * np + 1;
  int o = 0;
  l.T = o; o += np + 1;
  l.P = o; o += np;
  l.TD = o; o += m;
  l.WD = o; o += m;
  l.CW = o; o += m;
  l.CEN = o; o += n;
  l.SO = o; o += n + 1;
  l.RED = o; o += RSP_LPR;
  l.per_ray = (o + 3) & ~3;
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
  const RspLay lay = rsp_layout(cfg->n_prev, cfg->n_samples);
  const size_t lds_bytes = (size_t)lay.per_ray * RSP_RPW * 4;
  MNR_CHECK_ARG(lds_bytes <= 160 * 1024, "mnr_resample_level: step function too long for LDS");
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if (mnr_attr_needed(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)resample_level_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  const int grid = mnr_cdiv(B, RSP_RPW);
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

// Leaf: the dilation phase of the level kernel alone (same device function, 16 lanes per ray).
__global__ __launch_bounds__(RS_THREADS) void max_dilate_kernel(int64_t B, int n, const float* t, const float* w, float dilation,
                                                                float lo, float hi, float* t_out, float* w_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int m = 3 * n + 1;
  const int per_ray = (n + 1) + n + 2 * m + RSP_LPR;
  const int g = threadIdx.x / RSP_LPR, l = threadIdx.x % RSP_LPR;
  const int64_t ray = min(B - 1, (int64_t)blockIdx.x * RSP_RPW + g);
  float* tt = lds + g * per_ray;
  float* p = tt + n + 1;
  float* td = p + n;
  float* wd = td + m;
  float* red = wd + m;
  for (int i = l; i <= n; i += RSP_LPR) tt[i] = t[ray * (n + 1) + i];
  for (int i = l; i < n; i += RSP_LPR) p[i] = w[ray * n + i];
  __syncthreads();
  rsp_max_dilate(n, tt, p, td, wd, red, l, dilation, lo, hi);
  if ((int64_t)blockIdx.x * RSP_RPW + g < B) {
    for (int i = l; i < m; i += RSP_LPR) t_out[ray * m + i] = td[i];
    for (int i = l; i < m - 1; i += RSP_LPR) w_out[ray * (m - 1) + i] = wd[i];
  }
}

extern "C" int mnr_max_dilate_weights(int64_t B, int n, const float* t, const float* w, float dilation,
                                      float domain_lo, float domain_hi, float* t_out, float* w_out,
                                      float* scratch, void* stream) {
  (void)scratch;                                           // (kept in the signature: callers of the round-1 ABI pass one)
  MNR_CHECK_ARG(B > 0 && n > 0 && n <= 1024 && t && w && t_out && w_out, "mnr_max_dilate_weights: bad arguments");
  const size_t lds_bytes = (size_t)((n + 1) + n + 2 * (3 * n + 1) + RSP_LPR) * RSP_RPW * 4;
  MNR_CHECK_ARG(lds_bytes <= 160 * 1024, "mnr_max_dilate_weights: step function too long for LDS");
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if (mnr_attr_needed(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)max_dilate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL(max_dilate_kernel, dim3(mnr_cdiv(B, RSP_RPW)), dim3(RS_THREADS), lds_bytes, (hipStream_t)stream, B, n, t, w,
                     dilation, domain_lo, domain_hi, t_out, w_out);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// VJP of the level kernel with respect to the incoming step function (Model.stop_level_grad = False, reference
// models.py:198-201: the sample positions of a level then carry gradient into the previous level's sdist and weights).
//
// Given g_sdist [B, n+1] = d loss / d sdist_out the kernel re-runs the level's forward pass for its rays (the SAME device
// code in the same order, so the bins every sample fell into are the forward pass's own) and walks it backwards:
//   fence-posts -> centers (stepfun.py:252-262; the clamped ends pass gradient while inside the domain)
//   math.sorted_interp (math.py:108-127): c = f0 + off (f1 - f0), off = clip((u - x0) / (x1 - x0), 0, 1): gradient to the two
//     bracketing fence-posts f0, f1 of the (dilated) histogram and to the two CDF values x0, x1
//   integrate_weights (stepfun.py:131-150): transposed cumulative sum (suffix sums) behind the min(1, .) clamp
//   jax.nn.softmax (stepfun.py:156), anneal * log(w + padding) (models.py:183-185; closed bins are constants)
//   max_dilate_weights (stepfun.py:99-128): renormalisation, pdf <-> weight, the window maximum (gradient to the bin that
//     attains it), sort + clip (each dilated fence-post is one of t[i], t[i] - d, t[i+1] + d, or a clipped constant)
// to g_sdist_prev [B, n_prev+1] and g_w_prev [B, n_prev].  The indices (which bin, which maximum, which source of a sorted
// fence-post) are piecewise constant and carry no gradient, as in the reference's autodiff.  One deliberate deviation: where
// a bin's weight + padding is exactly 0 its logit is -inf, its softmax weight 0, and autodiff yields 0 * inf = NaN for that
// bin's weight gradient; the NaN then propagates to every parameter-gradient element that weight depends on, and the reference's
// train_step applies nan_to_num ELEMENT-WISE (train_utils.py:326-328): those elements become 0, the others stay.  Here that
// product is 0, i.e. the gradient of the function that is evaluated; with resample_padding = 0 the two training trajectories
// differ (the reference is effectively undefined there: models.Model.build warns once).
//
// 16 lanes per ray as in the forward kernel: element-wise passes are split over the lanes, the three scatters / scans
// (sample -> bracketing bins, suffix sum, window maximum -> bin) run on the ray's first lane (deterministic order).

struct RspBLay {
  int T, P, W0, TD, WR, BEST, JB, RANK, WN, LG, CW, PASS, CEN, I0, UQ, GC, GTD, GCW, GWK, GWN, GTF, GP, GT, RED, per_ray;
};

static RspBLay rspb_layout(int np, int n) {
  RspBLay l;
  const int m = 3 * np + 1;
  int o = 0;
  l.T = o; o += np + 1;
  l.P = o; o += np;
  l.W0 = o; o += np;
  l.TD = o; o += m;
  l.WR = o; o += m;
  l.BEST = o; o += m;
  l.JB = o; o += m;
  l.RANK = o; o += m;
  l.WN = o; o += m;
  l.LG = o; o += m;
  l.CW = o; o += m;
  l.PASS = o; o += m;
  l.CEN = o; o += n;
  l.I0 = o; o += n;
  l.UQ = o; o += n;
  l.GC = o; o += n;
  l.GTD = o; o += m;
  l.GCW = o; o += m;
  l.GWK = o; o += m;
  l.GWN = o; o += m;
  l.GTF = o; o += m;
  l.GP = o; o += np;
  l.GT = o; o += np + 1;
  l.RED = o; o += RSP_LPR;
  l.per_ray = (o + 3) & ~3;
  return l;
}

__global__ __launch_bounds__(RS_THREADS) void resample_level_bwd_kernel(
    mnr_resample_cfg c, int64_t B, RspBLay lay, const float* __restrict__ sdist_prev, const float* __restrict__ w_prev,
    const float* __restrict__ u_base, const float* __restrict__ jitter, const float* __restrict__ g_sdist,
    float* __restrict__ g_sdist_prev, float* __restrict__ g_w_prev) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int np = c.n_prev, n = c.n_samples;
  const float eps2 = MNR_F32_EPS * MNR_F32_EPS;
  const int64_t ray0 = (int64_t)blockIdx.x * RSP_RPW;
  const int nrays = (int)min((int64_t)RSP_RPW, B - ray0);
  for (int e = threadIdx.x; e < RSP_RPW * (np + 1); e += RS_THREADS) {
    const int r = e / (np + 1), i = e % (np + 1);
    lds[r * lay.per_ray + lay.T + i] = sdist_prev[(ray0 + min(r, nrays - 1)) * (np + 1) + i];
  }
  for (int e = threadIdx.x; e < RSP_RPW * np; e += RS_THREADS) {
    const int r = e / np, i = e % np;
    const float wv = w_prev[(ray0 + min(r, nrays - 1)) * np + i];
    lds[r * lay.per_ray + lay.P + i] = wv;
    lds[r * lay.per_ray + lay.W0 + i] = wv;
  }
  const int g = threadIdx.x / RSP_LPR, l = threadIdx.x % RSP_LPR;
  const int gr = min(g, nrays - 1);
  const int64_t ray = ray0 + gr;
  __syncthreads();
  float* base = lds + g * lay.per_ray;
  float* t = base + lay.T;
  float* p = base + lay.P;
  const float* w0 = base + lay.W0;
  float* red = base + lay.RED;
  float* tdf = base + lay.TD;                          // full dilated fence-posts [3 np + 1]
  float* wr = base + lay.WR;                           // un-normalised dilated weights [3 np]
  float* best = base + lay.BEST;
  int* jb = (int*)(base + lay.JB);
  int* rank = (int*)(base + lay.RANK);
  float* wn = base + lay.WN;                           // normalised dilated weights (full) / the incoming weights
  const int m = 3 * np + 1;
  float S_total = 1.0f;
  const float* td;                                     // the histogram being sampled: fence-posts, weights, bins
  const float* wdn;
  int nb;
  if (c.use_dilation) {
    // ---- forward: rsp_max_dilate with its indices recorded
    for (int j = l; j < np; j += RSP_LPR) p[j] = p[j] / fmaxf(eps2, t[j + 1] - t[j]);
    __syncthreads();
    for (int e = l; e < m; e += RSP_LPR) {
      float v;
      int rk;
      if (e <= np) {
        v = t[e];
        rk = e + rsp_count(t, np, -c.dilation, v, true) + rsp_count(t + 1, np, c.dilation, v, false);
      } else if (e <= 2 * np) {
        const int j = e - (np + 1);
        v = t[j] - c.dilation;
        rk = j + rsp_count(t, np + 1, 0.0f, v, false) + rsp_count(t + 1, np, c.dilation, v, false);
      } else {
        const int k = e - (2 * np + 1);
        v = t[k + 1] + c.dilation;
        rk = k + rsp_count(t, np + 1, 0.0f, v, true) + rsp_count(t, np, -c.dilation, v, true);
      }
      tdf[rk] = fminf(fmaxf(v, c.domain_lo), c.domain_hi);
      rank[e] = rk;
    }
    __syncthreads();
    const int nw = m - 1;
    const int ch = (nw + RSP_LPR - 1) / RSP_LPR;
    const int k0 = min(nw, l * ch), k1 = min(nw, k0 + ch);
    float csum = 0.0f;
    if (k0 < k1) {
      int jhi = rsp_count(t, np, -c.dilation, tdf[k0], true) - 1;
      int jlo = rsp_count(t + 1, np, c.dilation, tdf[k0], true);
      for (int k = k0; k < k1; ++k) {
        const float x = tdf[k];
        while (jhi + 1 < np && t[jhi + 1] - c.dilation <= x) ++jhi;
        while (jlo < np && !(t[jlo + 1] + c.dilation > x)) ++jlo;
        float bst = 0.0f;
        int jbst = -1;
        for (int j = jlo; j <= jhi; ++j) {
          if (p[j] > bst) {
            bst = p[j];
            jbst = j;
          }
        }
        best[k] = bst;
        jb[k] = jbst;
        const float w = bst * (tdf[k + 1] - x);
        wr[k] = w;
        csum += w;
      }
    }
    (void)rsp_chunk_offset(csum, red, l, S_total);
    const float denom = fmaxf(eps2, S_total);
    for (int k = k0; k < k1; ++k) wn[k] = wr[k] / denom;
    __syncthreads();
    td = tdf + 1;
    wdn = wn + 1;
    nb = 3 * np - 2;
  } else {
    for (int j = l; j < np; j += RSP_LPR) wn[j] = w0[j];
    __syncthreads();
    td = t;
    wdn = wn;
    nb = np;
  }
  // ---- forward: logits, softmax, CDF (the level kernel's order)
  float* lg = base + lay.LG;                           // logits, then softmax weights
  const int ch = (nb + RSP_LPR - 1) / RSP_LPR;
  const int k0 = min(nb, l * ch), k1 = min(nb, k0 + ch);
  float mx = -INFINITY;
  for (int k = k0; k < k1; ++k) {
    const bool open = td[k + 1] > td[k];
    const float v = open ? c.anneal * rs_log(wdn[k] + c.resample_padding) : -INFINITY;
    lg[k] = v;
    mx = (v != v || mx != mx) ? NAN : fmaxf(mx, v);
  }
  __syncthreads();
  red[l] = mx;
  __syncthreads();
  mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < RSP_LPR; ++i) {
    const float v = red[i];
    mx = (v != v || mx != mx) ? NAN : fmaxf(mx, v);
  }
  float csum = 0.0f;
  for (int k = k0; k < k1; ++k) {
    const float e = rs_exp(lg[k] - mx);
    lg[k] = e;
    csum += e;
  }
  float denom2;
  (void)rsp_chunk_offset(csum, red, l, denom2);
  float* cw = base + lay.CW;
  csum = 0.0f;
  for (int k = k0; k < k1; ++k) {
    const float wk = lg[k] / denom2;
    lg[k] = wk;
    if (k < nb - 1) csum += wk;
  }
  float unused;
  const float off0 = rsp_chunk_offset(csum, red, l, unused);
  float* pass_cs = base + lay.PASS;                    // [k + 1] = 1 where min(1, cs[k]) passes gradient
  float run = 0.0f;
  for (int k = k0; k < k1 && k < nb - 1; ++k) {
    run += lg[k];
    const float cs = off0 + run;
    cw[k + 1] = (cs != cs) ? cs : fminf(1.0f, cs);
    pass_cs[k + 1] = (cs <= 1.0f) ? 1.0f : 0.0f;
  }
  if (l == 0) {
    cw[0] = 0.0f;
    cw[nb] = 1.0f;
  }
  __syncthreads();
  // ---- forward: inverse CDF with the bracketing indices recorded
  float* centers = base + lay.CEN;
  int* i0s = (int*)(base + lay.I0);
  float* uq = base + lay.UQ;                           // the queries u
  const int chs = (n + RSP_LPR - 1) / RSP_LPR;
  const int j0 = min(n, l * chs), j1 = min(n, j0 + chs);
  const float jit1 = (jitter && c.single_jitter) ? jitter[ray] * c.max_jitter : 0.0f;
  int cur = -2;
  for (int j = j0; j < j1; ++j) {
    float u = u_base[j];
    if (jitter) u = u + (c.single_jitter ? jit1 : jitter[ray * n + j] * c.max_jitter);
    if (cur == -2) {
      int lo_i = -1, hi_i = nb + 1;
      while (hi_i - lo_i > 1) {
        const int mid = (lo_i + hi_i) >> 1;
        if (cw[mid] <= u) lo_i = mid;
        else hi_i = mid;
      }
      cur = lo_i;
    }
    centers[j] = rs_interp_one(u, cw, td, 1, nb + 1, cur);
    i0s[j] = cur;
    uq[j] = u;
  }
  __syncthreads();

  // ================= backward
  float* gc = base + lay.GC;
  float* gtd = base + lay.GTD;                         // d loss / d td (trimmed indexing: the sampled histogram's fence-posts)
  float* gwn = base + lay.GWN;                         // d loss / d (normalised dilated weights), full indexing when dilated
  float* gp = base + lay.GP;
  float* gt = base + lay.GT;
  const float* gso = g_sdist + ray * (n + 1);
  // fence-posts -> centers
  {
    const float x_first = 2.0f * centers[0] - (centers[1] + centers[0]) / 2.0f;
    const float x_last = 2.0f * centers[n - 1] - (centers[n - 1] + centers[n - 2]) / 2.0f;
    const float G0 = (x_first >= c.domain_lo) ? gso[0] : 0.0f;
    const float Gn = (x_last <= c.domain_hi) ? gso[n] : 0.0f;
    for (int j = l; j < n; j += RSP_LPR) {
      float v = 0.0f;
      if (j >= 1) v += 0.5f * gso[j];
      if (j + 1 <= n - 1) v += 0.5f * gso[j + 1];
      if (j == 0) v += 1.5f * G0;
      if (j == 1) v -= 0.5f * G0;
      if (j == n - 1) v += 1.5f * Gn;
      if (j == n - 2) v -= 0.5f * Gn;
      gc[j] = v;
    }
  }
  for (int k = l; k <= nb; k += RSP_LPR) gtd[k] = 0.0f;
  __syncthreads();
  // samples -> bracketing fence-posts and CDF values (first lane of the ray: the samples ascend, so do the bins)
  float* gcw = base + lay.GCW;                         // d loss / d cw
  float* gwk = base + lay.GWK;                         // d loss / d softmax weights
  for (int k = l; k <= nb; k += RSP_LPR) gcw[k] = 0.0f;
  for (int k = l; k < m; k += RSP_LPR) gwn[k] = 0.0f;
  __syncthreads();
  if (l == 0) {
    for (int j = 0; j < n; ++j) {
      const int i = i0s[j];
      const int a = i >= 0 ? i : 0;
      const int b = i + 1 < nb + 1 ? i + 1 : nb;
      const float u = uq[j];
      const float x0 = cw[a], x1 = cw[b], f0 = td[a], f1 = td[b];
      const float den = x1 - x0;
      const float raw = (u - x0) / den;
      const bool fin = (raw == raw) && fabsf(raw) <= 3.4028234664e38f;
      const float off = mnr_nan0_clip01(raw);
      const float gcj = gc[j];
      gtd[a] += gcj * (1.0f - off);
      gtd[b] += gcj * off;
      if (fin && raw >= 0.0f && raw <= 1.0f) {
        const float goff = gcj * (f1 - f0);
        gcw[a] += goff * (raw - 1.0f) / den;
        gcw[b] -= goff * raw / den;
      }
    }
    // cw[k+1] = min(1, cs[k]), cs = cumsum(w[:-1]): suffix sums behind the clamp; the last bin's weight gets nothing
    float suffix = 0.0f;
    gwk[nb - 1] = 0.0f;
    for (int k = nb - 2; k >= 0; --k) {
      suffix += gcw[k + 1] * pass_cs[k + 1];
      gwk[k] = suffix;
    }
  }
  __syncthreads();
  // softmax VJP, then the logits' log; lg[] holds the softmax weights
  float dotp = 0.0f;
  for (int k = k0; k < k1; ++k) dotp += lg[k] * gwk[k];
  float dot_total;
  (void)rsp_chunk_offset(dotp, red, l, dot_total);
  {
    float* gdst = c.use_dilation ? gwn + 1 : gwn;
    for (int k = k0; k < k1; ++k) {
      const float glog = lg[k] * (gwk[k] - dot_total);
      const bool open = td[k + 1] > td[k];
      const float den = wdn[k] + c.resample_padding;
      gdst[k] = (open && glog != 0.0f) ? glog * c.anneal / den : 0.0f;
    }
  }
  __syncthreads();
  if (!c.use_dilation) {
    if (g < nrays) {
      for (int i = l; i <= np; i += RSP_LPR) g_sdist_prev[ray * (np + 1) + i] = gtd[i];
      for (int i = l; i < np; i += RSP_LPR) g_w_prev[ray * np + i] = gwn[i];
    }
    return;
  }
  // ---- max_dilate_weights backwards.  Full indexing from here: gtd_full[r] = gtd[r - 1] for 1 <= r <= m - 2.
  const int nw = m - 1;
  {
    const int chw = (nw + RSP_LPR - 1) / RSP_LPR;
    const int q0 = min(nw, l * chw), q1 = min(nw, q0 + chw);
    float d2 = 0.0f;
    for (int k = q0; k < q1; ++k) d2 += gwn[k] * wn[k];
    float dot2;
    (void)rsp_chunk_offset(d2, red, l, dot2);
    const float denom = fmaxf(eps2, S_total);
    const bool pass = S_total >= eps2;
    // g_wr[k] in place of gwn[k]
    for (int k = q0; k < q1; ++k) gwn[k] = pass ? (gwn[k] - dot2) / denom : gwn[k] / denom;
  }
  for (int j = l; j < np; j += RSP_LPR) gp[j] = 0.0f;
  __syncthreads();
  // fence-post gradients of the dilated histogram: from the sampling (shifted) and from wr[k] = best[k] (td[k+1] - td[k])
  float* gtf = base + lay.GTF;                         // [m] full-index d loss / d tdf
  for (int r = l; r < m; r += RSP_LPR) {
    float v = (r >= 1 && r <= m - 2) ? gtd[r - 1] : 0.0f;
    if (r >= 1) v += gwn[r - 1] * best[r - 1];
    if (r < nw) v -= gwn[r] * best[r];
    gtf[r] = v;
  }
  if (l == 0) {
    for (int k = 0; k < nw; ++k) {
      const int j = jb[k];
      if (j >= 0) gp[j] += gwn[k] * (tdf[k + 1] - tdf[k]);
    }
  }
  __syncthreads();
  // sort + clip: every source element receives the gradient of the fence-post it became (unless clipped away)
  for (int i = l; i <= np; i += RSP_LPR) {
    float v = 0.0f;
    {
      const float a = t[i];
      if (a >= c.domain_lo && a <= c.domain_hi) v += gtf[rank[i]];
    }
    if (i < np) {
      const float bb = t[i] - c.dilation;
      if (bb >= c.domain_lo && bb <= c.domain_hi) v += gtf[rank[np + 1 + i]];
    }
    if (i >= 1) {
      const float cc = t[i] + c.dilation;
      if (cc >= c.domain_lo && cc <= c.domain_hi) v += gtf[rank[2 * np + 1 + (i - 1)]];
    }
    // pdf = w / max(eps^2, dt): the two bins next to this fence-post
    if (i < np) {
      const float dt = t[i + 1] - t[i];
      if (dt >= eps2) v += gp[i] * p[i] / dt;
    }
    if (i >= 1) {
      const float dt = t[i] - t[i - 1];
      if (dt >= eps2) v -= gp[i - 1] * p[i - 1] / dt;
    }
    gt[i] = v;
  }
  __syncthreads();
  if (g < nrays) {
    for (int i = l; i <= np; i += RSP_LPR) g_sdist_prev[ray * (np + 1) + i] = gt[i];
    for (int i = l; i < np; i += RSP_LPR) g_w_prev[ray * np + i] = gp[i] / fmaxf(eps2, t[i + 1] - t[i]);
  }
}

extern "C" int mnr_resample_level_bwd(const mnr_resample_cfg* cfg, int64_t B, const float* sdist_prev, const float* w_prev,
                                      const float* u_base, const float* jitter, const float* g_sdist, float* g_sdist_prev,
                                      float* g_w_prev, void* stream) {
  MNR_CHECK_ARG(cfg && B > 0 && sdist_prev && w_prev && u_base && g_sdist && g_sdist_prev && g_w_prev,
                "mnr_resample_level_bwd: null argument");
  MNR_CHECK_ARG(cfg->n_samples > 1, "num_samples must be > 1, is %d.", cfg->n_samples);
  MNR_CHECK_ARG(cfg->n_prev >= 1 && cfg->n_prev <= 1024 && cfg->n_samples <= 1024,
                "mnr_resample_level_bwd: n_prev=%d / n_samples=%d out of range", cfg->n_prev, cfg->n_samples);
  const RspBLay lay = rspb_layout(cfg->n_prev, cfg->n_samples);
  const size_t lds_bytes = (size_t)lay.per_ray * RSP_RPW * 4;
  MNR_CHECK_ARG(lds_bytes <= 160 * 1024, "mnr_resample_level_bwd: step function too long for LDS");
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if (mnr_attr_needed(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)resample_level_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL(resample_level_bwd_kernel, dim3(mnr_cdiv(B, RSP_RPW)), dim3(RS_THREADS), lds_bytes, (hipStream_t)stream,
                     *cfg, B, lay, sdist_prev, w_prev, u_base, jitter, g_sdist, g_sdist_prev, g_w_prev);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}
