// Ref-NeRF branch of the MLP (gfx950): predicted / density-gradient normals, reflection
// directions, integrated directional encoding (IDE), diffuse + tinted specular colour, and the
// orientation / predicted-normal losses, forward and VJP.
//
// Replaces reference internal/models.py:478-503,512-523,540-563,588-602, internal/ref_utils.py:22-42
// and :99-159, internal/image.py:48-56 and internal/train_utils.py:162-197.
//
// Density-gradient normals: the reference takes vmap(value_and_grad(predict_density)) w.r.t. the
// sample means (models.py:478-481).  Here the three tangents d/d mean_c are pushed FORWARD through the
// trunk as 3*M extra GEMM rows (tangent features from features.hip, each layer masked by the primal
// ReLU bits), so the "double backward" of the predicted-normal loss is an ordinary backward pass
// through that linear tangent network with the same GEMM kernels.  This file holds the per-sample
// glue around those GEMMs; one lane owns one sample.
//
// Merged head layout used throughout (columns of the head GEMM output / gradient matrix):
//   [0, bw) bottleneck | bw density | bw+1..3 grad_pred | bw+4..6 raw diffuse | bw+7..9 raw tint | bw+10 raw roughness
// `small` = the fp32 side output [M, 11] holding columns bw .. bw+10.  The layout is the same for every feature set
// (MNR_REF_* bits, models.py:468-563 takes each flag on its own): a head that is switched off is a zero column nobody reads.
#include "common.h"

#define RF_THREADS 256
#define RF_MAX_T 36
#define RF_MAX_L 16
#define RF_MAX_ENC (2 * RF_MAX_T)       // IDE: 2 T columns; coord.pos_enc of a direction: 3 + 6 deg_view (deg_view <= 11)
#define RF_PI_2 1.57079632679489661923f
#define RF_LOG3 1.09861228866810969140f

struct IdeTab {
  int T, lmax;
  int m[RF_MAX_T], l[RF_MAX_T];
  float sigma[RF_MAX_T];
  float mat[(RF_MAX_L + 1) * RF_MAX_T];     // [k][t]
};

__device__ __forceinline__ void rf_load_tab(IdeTab& tab, const mnr_ide_tables& g) {
  if (threadIdx.x == 0) { tab.T = g.T; tab.lmax = g.lmax; }
  for (int i = threadIdx.x; i < g.T; i += blockDim.x) {
    tab.m[i] = g.m[i];
    tab.l[i] = g.l[i];
    tab.sigma[i] = g.sigma[i];
  }
  for (int i = threadIdx.x; i < (g.lmax + 1) * g.T; i += blockDim.x) tab.mat[i] = g.mat[i];
}

// ref_utils.l2_normalize (ref_utils.py:40-42) of -x, and its VJP.
__device__ __forceinline__ void rf_neg_normalize(const float* x, float* out, float& r, bool& clamped) {
  const float s = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  clamped = !(s > MNR_F32_EPS);
  r = sqrtf(fmaxf(s, MNR_F32_EPS));
#pragma unroll
  for (int i = 0; i < 3; ++i) out[i] = -x[i] / r;
}

__device__ __forceinline__ void rf_neg_normalize_bwd(const float* x, float r, bool clamped, const float* g_out,
                                                     float* g_x) {
  // out = -x / r ;  r = sqrt(max(|x|^2, eps))
  float dot = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; ++i) dot += x[i] * g_out[i];
  const float r3 = r * r * r;
#pragma unroll
  for (int i = 0; i < 3; ++i) g_x[i] = -(g_out[i] / r - (clamped ? 0.0f : x[i] * dot / r3));
}

// IDE (ref_utils.py:127-157) at direction (x,y,z) with kappa_inv = kinv.  out: [2T] (real block, imag block).
// With BWD: g [2T] -> accumulates d/dx, d/dy, d/dz, d/dkinv.
template <bool BWD>
__device__ __forceinline__ void rf_ide(const IdeTab& tab, float x, float y, float z, float kinv, float* out,
                                       const float* g, float* gxyz, float* gk) {
  float pr[RF_MAX_L + 1], pi[RF_MAX_L + 1];           // (x + iy)^m
  pr[0] = 1.0f;
  pi[0] = 0.0f;
  for (int m = 1; m <= tab.lmax; ++m) {
    pr[m] = pr[m - 1] * x - pi[m - 1] * y;
    pi[m] = pr[m - 1] * y + pi[m - 1] * x;
  }
  const int T = tab.T;
  for (int t = 0; t < T; ++t) {
    const int m = tab.m[t], l = tab.l[t];
    const int K = l - m;
    // P(z) = sum_k mat[k][t] z^k (Horner), and P'(z).
    float p = 0.0f, dp = 0.0f;
    for (int k = K; k >= 0; --k) {
      dp = dp * z + p;
      p = p * z + tab.mat[k * T + t];
    }
    const float att = expf(-tab.sigma[t] * kinv);
    const float re = pr[m] * p * att;
    const float im = pi[m] * p * att;
    if (!BWD) {
      out[t] = re;
      out[T + t] = im;
    } else {
      const float gr = g[t], gi = g[T + t];
      const float gA_re = gr * p * att, gA_im = gi * p * att;
      const float gP = (gr * pr[m] + gi * pi[m]) * att;
      *gk += (gr * re + gi * im) * (-tab.sigma[t]);
      if (m > 0) {
        const float br = pr[m - 1] * (float)m, bi = pi[m - 1] * (float)m;     // m (x+iy)^(m-1)
        gxyz[0] += gA_re * br + gA_im * bi;
        gxyz[1] += gA_re * (-bi) + gA_im * br;
      }
      gxyz[2] += gP * dp;
    }
  }
}

// coord.pos_enc(u, 0, deg, append_identity=True) (coord.py:136-147; models.py:438-441 without the IDE):
// [u | sin(2^l u) (l-major) | sin(2^l u + pi/2)], and its VJP w.r.t. u.
__device__ __forceinline__ void rf_posenc(const float* u, int deg, float* out) {
  int c = 0;
  for (int i = 0; i < 3; ++i) out[c++] = u[i];
  for (int l = 0; l < deg; ++l)
    for (int i = 0; i < 3; ++i) out[c++] = sinf(u[i] * ldexpf(1.0f, l));
  for (int l = 0; l < deg; ++l)
    for (int i = 0; i < 3; ++i) out[c++] = sinf(u[i] * ldexpf(1.0f, l) + RF_PI_2);
}

__device__ __forceinline__ void rf_posenc_bwd(const float* u, int deg, const float* g, float* gu) {
  for (int i = 0; i < 3; ++i) {
    float acc = g[i];
    for (int l = 0; l < deg; ++l) {
      const float sc = ldexpf(1.0f, l), a = u[i] * sc;
      acc += sc * (g[3 + 3 * l + i] * cosf(a) + g[3 + 3 * deg + 3 * l + i] * cosf(a + RF_PI_2));
    }
    gu[i] = acc;
  }
}

__host__ __device__ __forceinline__ int rf_enc_dim(int features, int T, int deg_view) {
  return (features & MNR_REF_IDE) ? 2 * T : 3 + 6 * deg_view;
}

// ---------------------------------------------------------------------------
// Forward: normals, predicted normals, roughness, reflection dirs, IDE / pos_enc, n.v  ->  view-MLP input columns.

__global__ __launch_bounds__(RF_THREADS) void ref_head_fwd_kernel(
    int64_t M, int n, const float* __restrict__ small, const float* __restrict__ raw_grad,
    const float* __restrict__ viewdirs, mnr_ide_tables tabs, int features, int deg_view, float roughness_bias,
    bf16* __restrict__ vi, int ldvi, int col0, int col_end, float* __restrict__ normals_out, float* __restrict__ npred_out,
    float* __restrict__ rough_out, int vec8) {
  __shared__ IdeTab tab;
  if (features & MNR_REF_IDE) rf_load_tab(tab, tabs);
  __syncthreads();
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= M) return;
  const int64_t ray = s / n;
  const float v[3] = {viewdirs[ray * 3], viewdirs[ray * 3 + 1], viewdirs[ray * 3 + 2]};
  const float* sm = small + s * 11;
  float npred[3] = {0.0f, 0.0f, 0.0f}, nrm[3] = {0.0f, 0.0f, 0.0f}, r0, r1;
  bool c0, c1;
  if (features & MNR_REF_PRED_NORMALS) {
    const float gp[3] = {sm[1], sm[2], sm[3]};
    rf_neg_normalize(gp, npred, r0, c0);                        // models.py:498
  }
  if (features & MNR_REF_DENSITY_NORMALS) {
    const float rg[3] = {raw_grad[s], raw_grad[M + s], raw_grad[2 * M + s]};
    rf_neg_normalize(rg, nrm, r1, c1);                          // models.py:492
  }
  const float* nu = (features & MNR_REF_PRED_NORMALS) ? npred : nrm;                       // normals_to_use, models.py:499-503
  const float rough = (features & MNR_REF_ROUGHNESS) ? mnr_softplus(sm[10] + roughness_bias) : 0.0f;   // models.py:521-523
  const float ndv = nu[0] * v[0] + nu[1] * v[1] + nu[2] * v[2];
  // reflect(-v, n) = 2 (n . -v) n - (-v) = v - 2 (n.v) n      (ref_utils.py:22-37, models.py:545); else the view direction (:550)
  float u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = (features & MNR_REF_REFLECT) ? v[i] - 2.0f * ndv * nu[i] : v[i];
  float enc[RF_MAX_ENC];
  if (features & MNR_REF_IDE) rf_ide<false>(tab, u[0], u[1], u[2], rough, enc, nullptr, nullptr, nullptr);
  else rf_posenc(u, deg_view, enc);
  const int E = rf_enc_dim(features, (features & MNR_REF_IDE) ? tab.T : 0, deg_view);
  const bool has_ndv = (features & MNR_REF_N_DOT_V) != 0;
  bf16* o = vi + s * ldvi + col0;
  if (vec8) {
    // [encoding (E) | n.v | zeros up to col_end] in 16-byte pieces (col0 and col_end are multiples of 8 here)
    for (int c8 = 0; col0 + c8 * 8 < col_end; ++c8) {
      bf16x8 w;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = c8 * 8 + e;
        w[e] = (bf16)(i < E ? enc[i] : ((i == E && has_ndv) ? ndv : 0.0f));      // models.py:560-563 (un-negated viewdirs)
      }
      *(bf16x8*)(o + c8 * 8) = w;
    }
  } else {
    for (int i = 0; i < E; ++i) o[i] = (bf16)enc[i];
    if (has_ndv) o[E] = (bf16)ndv;                              // models.py:560-563 (un-negated viewdirs)
    for (int c = col0 + E + (has_ndv ? 1 : 0); c < col_end; ++c) vi[s * ldvi + c] = (bf16)0.0f;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (features & MNR_REF_DENSITY_NORMALS) normals_out[s * 3 + i] = nrm[i];
    if (features & MNR_REF_PRED_NORMALS) npred_out[s * 3 + i] = npred[i];
  }
  if (features & MNR_REF_ROUGHNESS) rough_out[s] = rough;
}

// Which feature sets the reference itself can run (models.py:538-563, ref_utils.py:147-154): the IDE multiplies by the
// roughness and is evaluated per sample, i.e. on reflection directions; reflections and n.v need a normal field.
static int rf_check_features(const char* who, int features, const mnr_ide_tables* tabs, int deg_view) {
  MNR_CHECK_ARG((features & ~(MNR_REF_PRED_NORMALS | MNR_REF_DENSITY_NORMALS | MNR_REF_REFLECT | MNR_REF_IDE | MNR_REF_N_DOT_V |
                              MNR_REF_ROUGHNESS)) == 0, "%s: unknown feature bits", who);
  const bool has_normals = (features & (MNR_REF_PRED_NORMALS | MNR_REF_DENSITY_NORMALS)) != 0;
  MNR_CHECK_ARG(has_normals || !(features & (MNR_REF_REFLECT | MNR_REF_N_DOT_V)), "Normals must be computed for reflection directions.");
  if (features & MNR_REF_IDE) {
    MNR_CHECK_ARG((features & MNR_REF_ROUGHNESS) && (features & MNR_REF_REFLECT),
                  "%s: the IDE needs the predicted roughness and reflection directions (ref_utils.py:147-154)", who);
    MNR_CHECK_ARG(tabs && tabs->T >= 1 && tabs->T <= RF_MAX_T && tabs->lmax <= RF_MAX_L, "Only deg_view of at most 5 is numerically stable.");
  } else {
    MNR_CHECK_ARG(deg_view >= 0 && 3 + 6 * deg_view <= RF_MAX_ENC, "%s: deg_view of the positional encoding out of range", who);
  }
  return MNR_OK;
}

extern "C" int mnr_ref_head_fwd(int64_t M, int n, const float* small, const float* raw_grad, const float* viewdirs,
                                const mnr_ide_tables* tabs, int features, int deg_view, float roughness_bias, uint16_t* vi,
                                int ldvi, int col0, int col_end, float* normals_out, float* normals_pred_out,
                                float* roughness_out, void* stream) {
  MNR_CHECK_ARG(M > 0 && n > 0 && small && viewdirs && vi, "mnr_ref_head_fwd: null argument");
  if (int rc = rf_check_features("mnr_ref_head_fwd", features, tabs, deg_view)) return rc;
  MNR_CHECK_ARG(!(features & MNR_REF_DENSITY_NORMALS) || (raw_grad && normals_out), "mnr_ref_head_fwd: density normals without raw_grad / normals_out");
  MNR_CHECK_ARG(!(features & MNR_REF_PRED_NORMALS) || normals_pred_out, "mnr_ref_head_fwd: predicted normals without normals_pred_out");
  MNR_CHECK_ARG(!(features & MNR_REF_ROUGHNESS) || roughness_out, "mnr_ref_head_fwd: roughness without roughness_out");
  const int ncols = rf_enc_dim(features, tabs ? tabs->T : 0, deg_view) + ((features & MNR_REF_N_DOT_V) ? 1 : 0);
  MNR_CHECK_ARG(col0 + ncols <= col_end && col_end <= ldvi, "mnr_ref_head_fwd: columns out of range");
  const mnr_ide_tables none = {0, 0, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(ref_head_fwd_kernel, dim3(mnr_cdiv(M, RF_THREADS)), dim3(RF_THREADS), 0, (hipStream_t)stream, M, n,
                     small, raw_grad, viewdirs, (features & MNR_REF_IDE) ? *tabs : none, features, deg_view, roughness_bias,
                     (bf16*)vi, ldvi, col0, col_end, normals_out, normals_pred_out, roughness_out,
                     (col0 % 8 == 0 && col_end % 8 == 0 && ldvi % 8 == 0 && ((uintptr_t)vi % 16) == 0) ? 1 : 0);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// VJP of the above.  Upstream: d/d(view-MLP input columns) (bf16 dVI_a [+ dVI_b], the two places the view
// MLP reads its input: layer 0 and the skip concat), g_npred / g_n [M,3] from the normal losses (may be
// NULL).  Outputs: gradient columns grad_pred (3) and raw roughness (1) of the merged head matrix (bf16),
// and g_raw_grad [3, M] (fp32) for the tangent network.

__global__ __launch_bounds__(RF_THREADS) void ref_head_bwd_kernel(
    int64_t M, int n, const float* __restrict__ small, const float* __restrict__ raw_grad,
    const float* __restrict__ viewdirs, mnr_ide_tables tabs, int features, int deg_view, float roughness_bias,
    const bf16* __restrict__ dvi_a, const bf16* __restrict__ dvi_b, int lddvi, int col0, const float* __restrict__ g_npred_in,
    const float* __restrict__ g_n_in, bf16* __restrict__ dhb, int lddhb, int col_gp, int col_rough,
    float* __restrict__ g_raw_grad, int vec8) {
  __shared__ IdeTab tab;
  if (features & MNR_REF_IDE) rf_load_tab(tab, tabs);
  __syncthreads();
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= M) return;
  const int64_t ray = s / n;
  const float v[3] = {viewdirs[ray * 3], viewdirs[ray * 3 + 1], viewdirs[ray * 3 + 2]};
  const float* sm = small + s * 11;
  float gp[3] = {0.0f, 0.0f, 0.0f}, rg[3] = {0.0f, 0.0f, 0.0f};
  float npred[3] = {0.0f, 0.0f, 0.0f}, nrm[3] = {0.0f, 0.0f, 0.0f}, r0 = 1.0f, r1 = 1.0f;
  bool c0 = true, c1 = true;
  if (features & MNR_REF_PRED_NORMALS) {
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] = sm[1 + i];
    rf_neg_normalize(gp, npred, r0, c0);
  }
  if (features & MNR_REF_DENSITY_NORMALS) {
#pragma unroll
    for (int i = 0; i < 3; ++i) rg[i] = raw_grad[(int64_t)i * M + s];
    rf_neg_normalize(rg, nrm, r1, c1);
  }
  const float* nu = (features & MNR_REF_PRED_NORMALS) ? npred : nrm;
  const float raw_r = sm[10] + roughness_bias;
  const float rough = (features & MNR_REF_ROUGHNESS) ? mnr_softplus(raw_r) : 0.0f;
  const float ndv = nu[0] * v[0] + nu[1] * v[1] + nu[2] * v[2];
  float u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = (features & MNR_REF_REFLECT) ? v[i] - 2.0f * ndv * nu[i] : v[i];
  const int E = rf_enc_dim(features, (features & MNR_REF_IDE) ? tab.T : 0, deg_view);
  const bool has_ndv = (features & MNR_REF_N_DOT_V) != 0;
  const int ncols = E + (has_ndv ? 1 : 0);
  float g[RF_MAX_ENC];
  float g_ndv = 0.0f;
  if (vec8) {
    // the gradient columns of this sample's row in 16-byte pieces (73 two-byte loads per matrix and thread before:
    // the kernel was bound by their issue, 5.6 ms per level at 2^21 samples)
    const bf16* pa = dvi_a + s * lddvi + col0;
    const bf16* pb = dvi_b ? dvi_b + s * lddvi + col0 : nullptr;
    for (int c8 = 0; c8 * 8 < ncols; ++c8) {
      const bf16x8 a = *(const bf16x8*)(pa + c8 * 8);
      bf16x8 b;
      if (pb) b = *(const bf16x8*)(pb + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = c8 * 8 + e;
        float x = (float)a[e];
        if (pb) x += (float)b[e];
        if (i < E) g[i] = x;
        else if (i == E && has_ndv) g_ndv = x;
      }
    }
  } else {
    for (int i = 0; i < E; ++i) {
      float x = (float)dvi_a[s * lddvi + col0 + i];
      if (dvi_b) x += (float)dvi_b[s * lddvi + col0 + i];
      g[i] = x;
    }
    if (has_ndv) {
      g_ndv = (float)dvi_a[s * lddvi + col0 + E];
      if (dvi_b) g_ndv += (float)dvi_b[s * lddvi + col0 + E];
    }
  }
  float gu[3] = {0.0f, 0.0f, 0.0f}, gk = 0.0f;
  if (features & MNR_REF_IDE) rf_ide<true>(tab, u[0], u[1], u[2], rough, nullptr, g, gu, &gk);
  else if (features & MNR_REF_REFLECT) rf_posenc_bwd(u, deg_view, g, gu);        // (of the view direction itself: an input, no gradient)
  // u = v - 2 (n.v) n  ->  d/dn ; n.v -> d/dn: the gradient of normals_to_use
  float g_nu[3];
  const float gun = gu[0] * nu[0] + gu[1] * nu[1] + gu[2] * nu[2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    g_nu[i] = ((features & MNR_REF_REFLECT) ? -2.0f * (gun * v[i] + ndv * gu[i]) : 0.0f) + g_ndv * v[i];
  if (features & MNR_REF_PRED_NORMALS) {
    float g_np[3], g_gp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) g_np[i] = g_nu[i] + (g_npred_in ? g_npred_in[s * 3 + i] : 0.0f);
    rf_neg_normalize_bwd(gp, r0, c0, g_np, g_gp);
#pragma unroll
    for (int i = 0; i < 3; ++i) dhb[s * lddhb + col_gp + i] = (bf16)g_gp[i];
  }
  if (features & MNR_REF_ROUGHNESS)       // (without the IDE the roughness is an output only: models.py:521-523, 550)
    dhb[s * lddhb + col_rough] = (bf16)((features & MNR_REF_IDE) ? gk * mnr_sigmoid(raw_r) : 0.0f);
  if (features & MNR_REF_DENSITY_NORMALS) {
    float g_n[3], g_rg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      g_n[i] = (g_n_in ? g_n_in[s * 3 + i] : 0.0f) + ((features & MNR_REF_PRED_NORMALS) ? 0.0f : g_nu[i]);
    rf_neg_normalize_bwd(rg, r1, c1, g_n, g_rg);
#pragma unroll
    for (int i = 0; i < 3; ++i) g_raw_grad[(int64_t)i * M + s] = g_rg[i];
  }
}

// dhb[:, :cols] = dvi_a[:, :cols] (+ dvi_b[:, :cols]): the bottleneck part of the view-input gradient.
__global__ void ref_bottleneck_grad_kernel(int64_t M, int cols8, const bf16* __restrict__ dvi_a,
                                           const bf16* __restrict__ dvi_b, int lddvi, int lddvi_b,
                                           bf16* __restrict__ dhb, int lddhb) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * cols8) return;
  const int64_t r = t / cols8;
  const int c = (int)(t % cols8) * 8;
  bf16x8 a = *reinterpret_cast<const bf16x8*>(dvi_a + r * lddvi + c);
  if (dvi_b) {
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(dvi_b + r * lddvi_b + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (bf16)((float)a[e] + (float)b[e]);
  }
  *reinterpret_cast<bf16x8*>(dhb + r * lddhb + c) = a;
}

extern "C" int mnr_add_cols_bf16(int64_t M, int cols, const uint16_t* a, int lda, const uint16_t* b, int ldb,
                                 uint16_t* dst, int lddst, void* stream) {
  MNR_CHECK_ARG(M > 0 && cols > 0 && a && dst, "mnr_add_cols_bf16: null argument");
  MNR_CHECK_ARG(cols % 8 == 0 && lda % 8 == 0 && lddst % 8 == 0 && (!b || ldb % 8 == 0),
                "mnr_add_cols_bf16: widths / strides must be multiples of 8");
  const int cols8 = cols / 8;
  hipLaunchKernelGGL(ref_bottleneck_grad_kernel, dim3(mnr_cdiv(M * cols8, 256)), dim3(256), 0, (hipStream_t)stream, M,
                     cols8, (const bf16*)a, (const bf16*)b, lda, ldb, (bf16*)dst, lddst);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_ref_head_bwd(int64_t M, int n, const float* small, const float* raw_grad, const float* viewdirs,
                                const mnr_ide_tables* tabs, int features, int deg_view, float roughness_bias,
                                const uint16_t* dvi_a, const uint16_t* dvi_b, int lddvi, int col0, const float* g_npred,
                                const float* g_n, uint16_t* dhb, int lddhb, int col_gp, int col_rough, float* g_raw_grad,
                                void* stream) {
  MNR_CHECK_ARG(M > 0 && n > 0 && small && viewdirs && dvi_a && dhb, "mnr_ref_head_bwd: null argument");
  if (int rc = rf_check_features("mnr_ref_head_bwd", features, tabs, deg_view)) return rc;
  MNR_CHECK_ARG(!(features & MNR_REF_DENSITY_NORMALS) || (raw_grad && g_raw_grad), "mnr_ref_head_bwd: density normals without raw_grad / g_raw_grad");
  MNR_CHECK_ARG((features & MNR_REF_PRED_NORMALS) || !g_npred, "mnr_ref_head_bwd: g_npred without predicted normals");
  MNR_CHECK_ARG((features & MNR_REF_DENSITY_NORMALS) || !g_n, "mnr_ref_head_bwd: g_n without density-gradient normals");
  // 16-byte row pieces when the columns [col0, col0 + 8 ceil(ncols / 8)) are aligned and inside the row
  const int ncols = rf_enc_dim(features, tabs ? tabs->T : 0, deg_view) + ((features & MNR_REF_N_DOT_V) ? 1 : 0);
  MNR_CHECK_ARG(col0 + ncols <= lddvi, "mnr_ref_head_bwd: columns out of range");
  const int ncol8 = (ncols + 7) / 8 * 8;
  const int vec8 = (col0 % 8 == 0 && lddvi % 8 == 0 && col0 + ncol8 <= lddvi && ((uintptr_t)dvi_a % 16) == 0 &&
                    (!dvi_b || ((uintptr_t)dvi_b % 16) == 0)) ? 1 : 0;
  const mnr_ide_tables none = {0, 0, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(ref_head_bwd_kernel, dim3(mnr_cdiv(M, RF_THREADS)), dim3(RF_THREADS), 0, (hipStream_t)stream, M, n,
                     small, raw_grad, viewdirs, (features & MNR_REF_IDE) ? *tabs : none, features, deg_view, roughness_bias,
                     (const bf16*)dvi_a, (const bf16*)dvi_b, lddvi, col0, g_npred, g_n, (bf16*)dhb, lddhb, col_gp, col_rough,
                     g_raw_grad, vec8);
  MNR_CHECK_LAUNCH();
  if (col0 > 0) {
    MNR_CHECK_ARG(col0 % 8 == 0 && lddvi % 8 == 0 && lddhb % 8 == 0, "mnr_ref_head_bwd: bottleneck width / strides must be multiples of 8");
    const int cols8 = col0 / 8;
    hipLaunchKernelGGL(ref_bottleneck_grad_kernel, dim3(mnr_cdiv(M * cols8, 256)), dim3(256), 0, (hipStream_t)stream,
                       M, cols8, (const bf16*)dvi_a, (const bf16*)dvi_b, lddvi, lddvi, (bf16*)dhb, lddhb);
    MNR_CHECK_LAUNCH();
  }
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// Predicted normals WITHOUT the rest of the Ref-NeRF head (models.py:494-503 with enable_pred_normals only): normals_pred =
// -l2_normalize(grad_pred) from three columns of the head's fp32 side output, and the VJP into the head's gradient matrix.

__global__ void pred_normals_kernel(int64_t M, const float* __restrict__ small, int ld, int col, float* __restrict__ npred_out,
                                    const float* __restrict__ g_npred, bf16* __restrict__ dhb, int lddhb, int col_g) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= M) return;
  const float gp[3] = {small[s * ld + col], small[s * ld + col + 1], small[s * ld + col + 2]};
  float np[3], r;
  bool c;
  rf_neg_normalize(gp, np, r, c);                              // models.py:498
  if (npred_out) {
#pragma unroll
    for (int i = 0; i < 3; ++i) npred_out[s * 3 + i] = np[i];
  }
  if (g_npred) {
    const float g[3] = {g_npred[s * 3], g_npred[s * 3 + 1], g_npred[s * 3 + 2]};
    float gx[3];
    rf_neg_normalize_bwd(gp, r, c, g, gx);
#pragma unroll
    for (int i = 0; i < 3; ++i) dhb[s * lddhb + col_g + i] = (bf16)gx[i];
  }
}

// Density-gradient normals WITHOUT the rest of the Ref-NeRF head (models.py:478-492 with disable_density_normals = False only, what
// configs/llff_raw.gin's comment asks for together with the orientation loss): normals = -l2_normalize(raw_grad) from the tangent
// network's output raw_grad [3, M] (component-major), and the VJP back into g_raw_grad [3, M].
__global__ void density_normals_kernel(int64_t M, const float* __restrict__ raw_grad, float* __restrict__ normals_out,
                                       const float* __restrict__ g_normals, float* __restrict__ g_raw_grad) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= M) return;
  const float rg[3] = {raw_grad[s], raw_grad[M + s], raw_grad[2 * M + s]};
  float nrm[3], r;
  bool c;
  rf_neg_normalize(rg, nrm, r, c);                             // models.py:492
  if (normals_out) {
#pragma unroll
    for (int i = 0; i < 3; ++i) normals_out[s * 3 + i] = nrm[i];
  }
  if (g_normals) {
    const float g[3] = {g_normals[s * 3], g_normals[s * 3 + 1], g_normals[s * 3 + 2]};
    float gx[3];
    rf_neg_normalize_bwd(rg, r, c, g, gx);
#pragma unroll
    for (int i = 0; i < 3; ++i) g_raw_grad[i * M + s] = gx[i];
  }
}

extern "C" int mnr_density_normals_fwd(int64_t M, const float* raw_grad, float* normals_out, void* stream) {
  MNR_CHECK_ARG(M > 0 && raw_grad && normals_out, "mnr_density_normals_fwd: bad argument");
  hipLaunchKernelGGL(density_normals_kernel, dim3(mnr_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, M, raw_grad, normals_out,
                     (const float*)nullptr, (float*)nullptr);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_density_normals_bwd(int64_t M, const float* raw_grad, const float* g_normals, float* g_raw_grad, void* stream) {
  MNR_CHECK_ARG(M > 0 && raw_grad && g_normals && g_raw_grad, "mnr_density_normals_bwd: bad argument");
  hipLaunchKernelGGL(density_normals_kernel, dim3(mnr_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, M, raw_grad, (float*)nullptr,
                     g_normals, g_raw_grad);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_pred_normals_fwd(int64_t M, const float* small, int ld, int col, float* normals_pred_out, void* stream) {
  MNR_CHECK_ARG(M > 0 && small && normals_pred_out && col >= 0 && col + 3 <= ld, "mnr_pred_normals_fwd: bad argument");
  hipLaunchKernelGGL(pred_normals_kernel, dim3(mnr_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, M, small, ld, col,
                     normals_pred_out, (const float*)nullptr, (bf16*)nullptr, 0, 0);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_pred_normals_bwd(int64_t M, const float* small, int ld, int col, const float* g_normals_pred, uint16_t* dhb,
                                    int lddhb, int col_g, void* stream) {
  MNR_CHECK_ARG(M > 0 && small && g_normals_pred && dhb && col >= 0 && col + 3 <= ld && col_g >= 0 && col_g + 3 <= lddhb,
                "mnr_pred_normals_bwd: bad argument");
  hipLaunchKernelGGL(pred_normals_kernel, dim3(mnr_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, M, small, ld, col,
                     (float*)nullptr, g_normals_pred, (bf16*)dhb, lddhb, col_g);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// Colour combine (models.py:584-602 with use_diffuse_color / use_specular_tint; image.py:48-56).

__device__ __forceinline__ float rf_linear_to_srgb(float lin) {
  const float srgb0 = (323.0f / 25.0f) * lin;
  const float srgb1 = (211.0f * powf(fmaxf(MNR_F32_EPS, lin), 5.0f / 12.0f) - 11.0f) / 200.0f;
  return lin <= 0.0031308f ? srgb0 : srgb1;
}

__global__ void ref_color_kernel(int64_t M, const float* __restrict__ raw_rgb, const float* __restrict__ small,
                                 float premult, float rgb_bias, float pad, int use_tint,
                                 const float* __restrict__ g_rgb, float* __restrict__ rgb_out,
                                 float* __restrict__ g_raw_rgb, bf16* __restrict__ dhb, int lddhb, int col_dif,
                                 int col_tint) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= M) return;
  const float* sm = small + s * 11;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float spec = mnr_sigmoid(premult * raw_rgb[s * 3 + ch] + rgb_bias);       // models.py:584-586
    const float tint = use_tint ? mnr_sigmoid(sm[7 + ch]) : 0.5f;                    // :518, :592-595
    const float dif = mnr_sigmoid(sm[4 + ch] - RF_LOG3);                             // :591
    const float lin = tint * spec + dif;
    const float srgb = rf_linear_to_srgb(lin);
    const float clipped = fminf(fmaxf(srgb, 0.0f), 1.0f);                            // :598-599
    if (rgb_out) rgb_out[s * 3 + ch] = clipped * (1.0f + 2.0f * pad) - pad;          // :602
    if (g_rgb) {
      const float g_srgb = (srgb > 0.0f && srgb < 1.0f) ? g_rgb[s * 3 + ch] * (1.0f + 2.0f * pad) : 0.0f;
      float dsrgb;
      if (lin <= 0.0031308f) dsrgb = 323.0f / 25.0f;
      else dsrgb = lin > MNR_F32_EPS ? (211.0f / 200.0f) * (5.0f / 12.0f) * powf(lin, -7.0f / 12.0f) : 0.0f;
      const float g_lin = g_srgb * dsrgb;
      g_raw_rgb[s * 3 + ch] = g_lin * tint * spec * (1.0f - spec) * premult;
      dhb[s * lddhb + col_dif + ch] = (bf16)(g_lin * dif * (1.0f - dif));
      if (use_tint) dhb[s * lddhb + col_tint + ch] = (bf16)(g_lin * spec * tint * (1.0f - tint));
    }
  }
}

extern "C" int mnr_ref_color_fwd(int64_t M, const float* raw_rgb, const float* small, float rgb_premultiplier,
                                 float rgb_bias, float rgb_padding, int use_tint, float* rgb_out, void* stream) {
  MNR_CHECK_ARG(M > 0 && raw_rgb && small && rgb_out, "mnr_ref_color_fwd: null argument");
  hipLaunchKernelGGL(ref_color_kernel, dim3(mnr_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, M, raw_rgb, small,
                     rgb_premultiplier, rgb_bias, rgb_padding, use_tint, (const float*)nullptr, rgb_out,
                     (float*)nullptr, (bf16*)nullptr, 0, 0, 0);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_ref_color_bwd(int64_t M, const float* raw_rgb, const float* small, float rgb_premultiplier,
                                 float rgb_bias, float rgb_padding, int use_tint, const float* g_rgb,
                                 float* g_raw_rgb, uint16_t* dhb, int lddhb, int col_diffuse, int col_tint,
                                 void* stream) {
  MNR_CHECK_ARG(M > 0 && raw_rgb && small && g_rgb && g_raw_rgb && dhb, "mnr_ref_color_bwd: null argument");
  hipLaunchKernelGGL(ref_color_kernel, dim3(mnr_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, M, raw_rgb, small,
                     rgb_premultiplier, rgb_bias, rgb_padding, use_tint, g_rgb, (float*)nullptr, g_raw_rgb, (bf16*)dhb,
                     lddhb, col_diffuse, col_tint);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// orientation_loss + predicted_normal_loss for one level (train_utils.py:162-197) and their VJPs.

__global__ void ref_losses_kernel(int64_t B_valid, int n, float mult_orient, float mult_pred, int target_is_pred,
                                  const float* __restrict__ weights, const float* __restrict__ normals,
                                  const float* __restrict__ npred, const float* __restrict__ viewdirs, float* stats,
                                  float* __restrict__ g_w, float* __restrict__ g_n, float* __restrict__ g_npred) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float lo = 0.0f, lp = 0.0f;
  if (b < B_valid) {
    const float v[3] = {-viewdirs[b * 3], -viewdirs[b * 3 + 1], -viewdirs[b * 3 + 2]};   // point -> camera
    const float invB = 1.0f / (float)B_valid;
    for (int i = 0; i < n; ++i) {
      const int64_t s = b * n + i;
      const float w = weights[s];
      const float* nt = (target_is_pred ? npred : normals) + s * 3;
      const float ndv = nt[0] * v[0] + nt[1] * v[1] + nt[2] * v[2];
      const float neg = fminf(0.0f, ndv);
      // (normals == NULL: an MLP with predicted normals only, models.py:494-503 with disable_density_normals; the host admits it
      // with mult_pred == 0 and the predicted normals as the orientation target)
      // (npred == NULL: density-gradient normals only, models.py:478-492 without enable_pred_normals: mult_pred == 0 and
      // the density normals as the orientation target)
      const float dot = (normals && npred) ? normals[s * 3] * npred[s * 3] + normals[s * 3 + 1] * npred[s * 3 + 1] +
                                                 normals[s * 3 + 2] * npred[s * 3 + 2]
                                           : 1.0f;
      lo += w * neg * neg;                                     // :173
      lp += w * (1.0f - dot);                                  // :192
      if (g_w) g_w[s] += (mult_orient * neg * neg + mult_pred * (1.0f - dot)) * invB;
      if (g_npred || g_n) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float gn = npred ? -mult_pred * w * npred[s * 3 + c] * invB : 0.0f;
          float gp = normals ? -mult_pred * w * normals[s * 3 + c] * invB : 0.0f;
          const float go = mult_orient * w * 2.0f * neg * v[c] * invB;
          if (target_is_pred) gp += go; else gn += go;
          if (g_n) g_n[s * 3 + c] = gn;
          if (g_npred) g_npred[s * 3 + c] = gp;
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo += __shfl_down(lo, off, 64);
    lp += __shfl_down(lp, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    unsafeAtomicAdd(stats + 0, mult_orient * lo / (float)B_valid);
    unsafeAtomicAdd(stats + 1, mult_pred * lp / (float)B_valid);
  }
}

extern "C" int mnr_ref_losses(int64_t B_valid, int n, float mult_orientation, float mult_pred_normal,
                              int target_is_pred, const float* weights, const float* normals,
                              const float* normals_pred, const float* viewdirs, float* stats, float* g_weights,
                              float* g_normals, float* g_normals_pred, void* stream) {
  MNR_CHECK_ARG(B_valid > 0 && n > 0 && weights && (normals || normals_pred) && viewdirs && stats, "mnr_ref_losses: null argument");
  MNR_CHECK_ARG(normals || (mult_pred_normal == 0.0f && target_is_pred && !g_normals),
                "mnr_ref_losses: without density-gradient normals only the orientation loss on the predicted normals is defined");
  MNR_CHECK_ARG(normals_pred || (mult_pred_normal == 0.0f && !target_is_pred && !g_normals_pred),
                "mnr_ref_losses: without predicted normals only the orientation loss on the density-gradient normals is defined");
  MNR_CHECK_ARG(!(normals && normals_pred) || (g_normals == nullptr) == (g_normals_pred == nullptr),
                "mnr_ref_losses: g_normals and g_normals_pred go together");
  hipLaunchKernelGGL(ref_losses_kernel, dim3(mnr_cdiv(B_valid, 64)), dim3(64), 0, (hipStream_t)stream, B_valid, n,
                     mult_orientation, mult_pred_normal, target_is_pred, weights, normals, normals_pred, viewdirs,
                     stats, g_weights, g_normals, g_normals_pred);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// out[b, c] = sum_i weights[b, i] * values[b, i, c]   (render.py:187-190 extras: normals, roughness).

__global__ void weighted_sum_kernel(int64_t B, int n, int C, const float* __restrict__ weights,
                                    const float* __restrict__ values, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int64_t b = e / C;
  const int c = (int)(e % C);
  float acc = 0.0f;
  for (int i = 0; i < n; ++i) acc += weights[b * n + i] * values[(b * n + i) * C + c];
  out[e] = acc;
}

extern "C" int mnr_weighted_sum(int64_t B, int n, int C, const float* weights, const float* values, float* out,
                                void* stream) {
  MNR_CHECK_ARG(B > 0 && n > 0 && C > 0 && weights && values && out, "mnr_weighted_sum: bad arguments");
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(mnr_cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, B, n, C,
                     weights, values, out);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}
