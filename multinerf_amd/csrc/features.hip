// Ray casting + integrated positional encoding -> bf16 feature rows (gfx950).
//
// Per sample (interval of a ray) this fuses, in fp32:
//   render.cast_rays            (render.py:103-127; conical frustum eq. 7 :62-70 or cylinder :97-99,
//                                lift_gaussian full covariance :35-41)
//   coord.track_linearize(coord.contract)   (coord.py:21-27, 39-60; closed-form Jacobian)
//   coord.lift_and_diagonalize  (coord.py:129-133)
//   coord.integrated_pos_enc    (coord.py:102-126) with math.safe_sin (math.py:26-38)
// and writes the [2*K*L] features of the sample as one bf16 row of the first
// Dense layer's A operand.  The reference materialises means [B,n,3], covs
// [B,n,3,3], lifted [B,n,K] x2 and the fp32 features [B,n,2KL] in HBM; here only
// the bf16 row leaves the CU, staged through LDS so the global stores are full
// 16-byte-per-lane row segments.
//
// Work split inside a 256-thread block handling SPB consecutive samples:
//   phase 1: one thread per sample: Gaussian (mean, cov), warp -> LDS
//   phase 2: one thread per (sample, basis direction): projection, then the L degrees
//   phase 3: all threads: coalesced copy of the [SPB, ld] bf16 rows to HBM
#include "common.h"

// No floating-point contraction in this file: the instantiations of cast_rays_ipe_kernel (bf16 rows, f32 rows for the parity
// tests, tangent rows) must evaluate the same separately rounded operations, whatever hipcc would fuse in each of them (the
// explicit fmaf calls are fused in all).
#pragma clang fp contract(off)

#ifndef FE_THREADS                                     // (probe builds: -DFE_THREADS=... -DFE_STAGE_BYTES=..., tools/ipe_probe.py)
#define FE_THREADS 256
#endif
#ifndef FE_STAGE_BYTES
#define FE_STAGE_BYTES (32 * 1024)                      // feature rows a block stages in LDS
#endif
#include "ipe_math.h"

// Tangents of the contraction for the density-gradient normals (models.py:445-446 applies warp_fn INSIDE predict_density, so
// jax.value_and_grad, :478-481, differentiates through it): with x the pre-warp mean and the covariance an INPUT of
// predict_density (held fixed),  z = s x,  cov' = J cov J^T = ku u u^T + r_var (s^2 I + j2x x x^T)  (fe_gaussian's structured
// form, u = J d, J = s I + cc x x^T), this returns for c = x, y, z:
//   dz[c]  = d z / d x_c  = cc x_c x + s e_c                                              (column c of J)
//   dC[c]  = d cov' / d x_c = ku (du_c u^T + u du_c^T) + r_var (2 s ds_c I + dj2x_c x x^T + j2x (e_c x^T + x e_c^T))
// with du_c = cc (x_c d + d_c x + (x.d) e_c) + 2 cc' x_c (x.d) x,  ds_c = cc x_c,  cc' = (3 sqrt(m) - 4) / m^3 and
// dj2x_c = d(2 cc / sqrt(m))/dm * 2 x_c.  u itself comes from fe_gaussian's cancellation-free form.  Inside the unit ball the
// contraction is the identity: dz[c] = e_c, dC = 0 (the no-warp case).
struct FeTangent {
  float dz[3][3];
  float dC[3][6];          // xx, xy, xz, yy, yz, zz
};

__device__ __forceinline__ void fe_contract_tangent(const mnr_ipe_cfg& c, float t0, float t1, const float* o, const float* d,
                                                    float radius, FeTangent& T) {
#pragma unroll
  for (int cc_ = 0; cc_ < 3; ++cc_) {
#pragma unroll
    for (int i = 0; i < 3; ++i) T.dz[cc_][i] = (i == cc_) ? 1.0f : 0.0f;
#pragma unroll
    for (int e = 0; e < 6; ++e) T.dC[cc_][e] = 0.0f;
  }
  if (!c.warp_contract) return;
  float t_mean, t_var, r_var;
  if (c.ray_shape == 0) {
    const float mu = (t0 + t1) / 2.0f;
    const float hw = (t1 - t0) / 2.0f;
    const float denom = fmaxf(MNR_F32_EPS, 3.0f * mu * mu + hw * hw);
    const float hw2 = hw * hw, hw4 = hw2 * hw2;
    t_mean = mu + (2.0f * mu * hw2) / denom;
    t_var = hw2 / 3.0f - (4.0f / 15.0f) * hw4 * (12.0f * mu * mu - hw2) / (denom * denom);
    r_var = (mu * mu) / 4.0f + (5.0f / 12.0f) * hw2 - (4.0f / 15.0f) * hw4 / denom;
    r_var *= radius * radius;
  } else {
    t_mean = (t0 + t1) / 2.0f;
    r_var = radius * radius / 4.0f;
    t_var = (t1 - t0) * (t1 - t0) / 12.0f;
  }
  if (c.disable_integration) {
    t_var = 0.0f;
    r_var = 0.0f;
  }
  const float dmag = fmaxf(1e-10f, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  float x[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) x[i] = o[i] + d[i] * t_mean;
  const float m = fmaxf(MNR_F32_EPS, x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  if (m <= 1.0f) return;
  const float sq = sqrtf(m);
  const float s = (2.0f * sq - 1.0f) / m;
  const float cc = 2.0f * (1.0f - sq) / (m * m);
  const float ccp = (3.0f * sq - 4.0f) / (m * m * m);              // d cc / d m
  const float j2x = 2.0f * cc / sq;
  const float j2xp = (2.0f * ccp - cc / m) / sq;                   // d (2 cc / sqrt(m)) / d m  = 2 cc' / sq - cc / (m sq)
  const float oo = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
  const float od = o[0] * d[0] + o[1] * d[1] + o[2] * d[2];
  const float xd = x[0] * d[0] + x[1] * d[1] + x[2] * d[2];
  const float brk = (m - 2.0f * (1.0f - sq) * (oo + t_mean * od)) / (m * m);
  float u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = brk * d[i] + cc * xd * o[i];
  const float ku = t_var - r_var / dmag;
  const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float xk = x[k];
    float du[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      du[i] = cc * (xk * d[i] + d[k] * x[i] + (i == k ? xd : 0.0f)) + 2.0f * ccp * xk * xd * x[i];
      T.dz[k][i] = cc * xk * x[i] + (i == k ? s : 0.0f);
    }
    const float ds = cc * xk;
    const float dj = j2xp * 2.0f * xk;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int i = ii[e], j = jj[e];
      float v = ku * (du[i] * u[j] + u[i] * du[j]);
      float w = dj * x[i] * x[j] + j2x * ((i == k ? x[j] : 0.0f) + (j == k ? x[i] : 0.0f));
      if (i == j) w += 2.0f * s * ds;
      T.dC[k][e] = v + r_var * w;
    }
  }
}

template <bool OUT_F32, bool TANGENT>
__global__ __launch_bounds__(FE_THREADS) void cast_rays_ipe_kernel(
    mnr_ipe_cfg c, int64_t total, int n, int spb, int pitch, const float* __restrict__ tdist,
    const float* __restrict__ origins, const float* __restrict__ directions, const float* __restrict__ radii,
    const float* __restrict__ basis, void* __restrict__ feat_out, int ld_feat, float* __restrict__ means_out,
    float* __restrict__ covs_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int K = c.basis_k;
  const int L = c.max_deg - c.min_deg;
  const int nfeat = 2 * K * L;
  // LDS: samples [spb] FeSample | basis [K*3] | rows [spb][ld] (bf16 or f32)
  FeSample* gs = (FeSample*)smem;
  FeTangent* gt = (FeTangent*)(gs + spb);                           // (TANGENT only: [spb] behind the samples)
  float* bs = TANGENT ? (float*)(gt + spb) : (float*)(gs + spb);
  char* rows = (char*)(bs + ((K * 3 + 3) & ~3));
  const int64_t s0 = (int64_t)blockIdx.x * spb;
  const int ns = (int)min((int64_t)spb, total - s0);

  for (int i = threadIdx.x; i < K * 3; i += FE_THREADS) bs[i] = basis[i];
  if (threadIdx.x < ns) {
    const int64_t s = s0 + threadIdx.x;
    const int64_t ray = s / n;
    const int j = (int)(s % n);
    const float t0 = tdist[ray * (n + 1) + j], t1 = tdist[ray * (n + 1) + j + 1];
    float o[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      o[i] = origins[ray * 3 + i];
      d[i] = directions[ray * 3 + i];
    }
    FeSample g;
#if defined(FE_DBG) && FE_DBG == 3                      // timing probe: no Gaussian arithmetic
    g.mean[0] = t0 + o[0], g.mean[1] = t1 + o[1], g.mean[2] = d[2] + radii[ray];
    for (int i = 0; i < 6; ++i) g.cov[i] = 1e-3f * d[i % 3];
#else
    fe_gaussian(c, t0, t1, o, d, radii[ray], g);
#endif
    gs[threadIdx.x] = g;
    if (TANGENT) {
      FeTangent T;
      fe_contract_tangent(c, t0, t1, o, d, radii[ray], T);
      gt[threadIdx.x] = T;
    }
    if (means_out) {
#pragma unroll
      for (int i = 0; i < 3; ++i) means_out[s * 3 + i] = g.mean[i];
    }
    if (covs_out) {
      const float* cv = g.cov;
      const float full[9] = {cv[0], cv[1], cv[2], cv[1], cv[3], cv[4], cv[2], cv[4], cv[5]};
#pragma unroll
      for (int i = 0; i < 9; ++i) covs_out[s * 9 + i] = full[i];
    }
  }
  __syncthreads();

  const int row_elems = OUT_F32 ? nfeat : ld_feat;
  // `pitch`: bytes between staged rows in LDS (row bytes + padding: with 1-KiB rows every sample of a wave hits the same banks).
  // The kernel streams its feature rows out at 3.0-3.9 TB/s (1 GB per 64-sample proposal level in 0.27-0.36 ms).  That is NOT the
  // write rate HBM sustains (a plain fill writes 6.9 TB/s, profiles/r5k_write_rate.txt): without its write-out the kernel takes 246 of
  // its 357 us, without the encoding loop 191 (profiles/r5m_ipe_probe.txt): the two phases of a block barely overlap with the
  // other blocks of its CU.  Round 2 cut the loop from ≈470 to ≈340 instructions per (sample, direction) without changing the
  // time; round 5's two-directions-per-thread loop on v_pk_mul_f32 (1.7x fewer instructions, same bits) was 5-9 % SLOWER
  // (profiles/r5n_probe.txt) and is not here.  The inner loop is short rather than clever:
  // an anchor every 4th degree = one sin / cos of the wrapped argument (math.safe_sin's wrap at float32(100 pi),
  // math.py:26-28; fe_sincos_wrapped) and one hardware exp2 for the attenuation; the 3 degrees behind it by the double-angle
  // recurrence (sin 2x = 2 sin x cos x, cos 2x = 1 - 2 sin^2 x; cf. stable_pos_enc in the reference's tests/coord_test.py:34-43)
  // and by att(l+1) = att(l)^4 (exp(-v 4^l / 2): two squarings): at most 3 steps of a ~1e-7 error, each at most x4.
  // cos is the reference's sin(x + pi/2).  sin and cos feature of a (degree, direction) leave as one packed bf16 pair.
  // TANGENT: three rows per sample (d/d mean_x, d/d mean_y, d/d mean_z), staged as [c][sample][ld].
  const float inv_k = 1.0f / (float)K;
#if defined(FE_DBG) && FE_DBG == 1                      // timing probe (tools/ipe_probe.py): no encoding (rows are whatever LDS holds)
  for (int pair = threadIdx.x; pair < 0; pair += FE_THREADS) {
#else
  for (int pair = threadIdx.x; pair < ns * K; pair += FE_THREADS) {
#endif
    const int si = (int)(((float)pair + 0.5f) * inv_k);       // pair / K, exact for pair < 2^20
    const int k = pair - si * K;
    const FeSample g = gs[si];
    const float px = bs[k * 3 + 0], py = bs[k * 3 + 1], pz = bs[k * 3 + 2];
    // coord.py:131-132: mean . p_k ; p_k^T cov p_k.
    const float lm = g.mean[0] * px + g.mean[1] * py + g.mean[2] * pz;
    const float cx = g.cov[0] * px + g.cov[1] * py + g.cov[2] * pz;
    const float cy = g.cov[1] * px + g.cov[3] * py + g.cov[4] * pz;
    const float cz = g.cov[2] * px + g.cov[4] * py + g.cov[5] * pz;
    const float lv = px * cx + py * cy + pz * cz;
    const float vscale = -0.5f * 1.44269504088896340736f * lv;       // exp(-v/2) = exp2(vscale * 4^deg)
    char* rowp = rows + (size_t)si * pitch + (size_t)k * (OUT_F32 ? 4 : (int)sizeof(bf16));       // column k of the sample's row (row 0 of 3 if TANGENT)
    const int half = K * L * (OUT_F32 ? 4 : (int)sizeof(bf16));                      // byte offset of the cos half of the row
    const int lstep = K * (OUT_F32 ? 4 : (int)sizeof(bf16));
    float sc = ldexpf(1.0f, c.min_deg);                              // 2^deg, exact
    float sn = 0.0f, cs = 1.0f, att = 1.0f;
    float dlm[3] = {px, py, pz}, dlv[3] = {0.0f, 0.0f, 0.0f};
    if (TANGENT) {
      const FeTangent& T = gt[si];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        dlm[cc] = px * T.dz[cc][0] + py * T.dz[cc][1] + pz * T.dz[cc][2];
        const float* C6 = T.dC[cc];
        dlv[cc] = px * (C6[0] * px + C6[1] * py + C6[2] * pz) + py * (C6[1] * px + C6[3] * py + C6[4] * pz) +
                  pz * (C6[2] * px + C6[4] * py + C6[5] * pz);
      }
    }
    for (int l = 0; l < L; ++l) {
      if ((l & 3) == 0) {
        fe_sincos_wrapped(fe_wrap_100pi(lm * sc), &sn, &cs);
        att = exp2f(vscale * sc * sc);
      }
      const float fs = att * sn;
      const float fc = att * cs;
      if (TANGENT) {
        // d/d mean_c of att sin(lm 2^l) = att 2^l cos(.) dlm_c - 1/2 4^l att sin(.) dlv_c;  of att cos(.): -att 2^l sin(.) dlm_c
        // - 1/2 4^l att cos(.) dlv_c, with dlm_c = p_k . dz[c], dlv_c = p_k^T dC[c] p_k (no warp: dlm_c = p_k[c], dlv_c = 0:
        // the variance does not depend on the mean).
        const float hv = -0.5f * sc * sc;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          char* rp = rowp + (size_t)cc * spb * pitch;
          *(bf16*)rp = (bf16)(fc * sc * dlm[cc] + hv * fs * dlv[cc]);
          *(bf16*)(rp + half) = (bf16)(-fs * sc * dlm[cc] + hv * fc * dlv[cc]);
        }
      } else if (OUT_F32) {
        *(float*)rowp = fs;
        *(float*)(rowp + half) = fc;
      } else {
        const f32x2 pr = {fs, fc};
        const bf16x2 pb = __builtin_convertvector(pr, bf16x2);       // one v_cvt_pk_bf16_f32
#if defined(FE_DBG) && FE_DBG == 4                      // timing probe: one LDS store per (sample, direction) instead of 2 L
        if (l == L - 1) *(bf16x2*)(rows + (size_t)si * pitch + (size_t)k * 4) = pb;
#else
        *(bf16*)rowp = pb[0];
        *(bf16*)(rowp + half) = pb[1];
#endif
      }
      rowp += lstep;
      const float s2 = 2.0f * sn * cs;
      cs = 1.0f - 2.0f * sn * sn;
      sn = s2;
      const float a2 = att * att;
      att = a2 * a2;
      sc *= 2.0f;
    }
  }
  if (!OUT_F32) {
    // zero the padding columns [nfeat, ld)
    const int pad = ld_feat - nfeat;
    const int nrows = TANGENT ? 3 * spb : ns;
    for (int e = threadIdx.x; e < nrows * pad; e += FE_THREADS) {
      const int si = e / pad, cidx = nfeat + e % pad;
      ((bf16*)(rows + (size_t)si * pitch))[cidx] = (bf16)0.0f;
    }
  }
  __syncthreads();
  // Coalesced write-out: the block's rows are contiguous in HBM (16 B per lane); in LDS they are `pitch` apart.
  const int row_bytes = row_elems * (OUT_F32 ? 4 : (int)sizeof(bf16));
  const int cpr = row_bytes >> 4;                         // 16-B chunks per row (row_bytes is a multiple of 16)
  for (int cc = 0; cc < (TANGENT ? 3 : 1); ++cc) {
    char* dst = (char*)feat_out + ((size_t)cc * total + s0) * row_bytes;
    const char* src = rows + (size_t)cc * spb * pitch;
#if defined(FE_DBG) && FE_DBG == 2                      // timing probe: no write-out (one chunk per block keeps the encoding alive)
    for (int ch = threadIdx.x; ch < 1; ch += FE_THREADS) {
#else
    for (int ch = threadIdx.x; ch < ns * cpr; ch += FE_THREADS) {
#endif
      const int r = ch / cpr, o = (ch - r * cpr) << 4;
      *(uint4*)(dst + (size_t)r * row_bytes + o) = *(const uint4*)(src + (size_t)r * pitch + o);
    }
  }
}

static int fe_launch(int mode /*0 bf16, 1 f32, 2 tangent*/, const mnr_ipe_cfg* cfg, int64_t B, int n, const float* tdist, const float* origins,
                     const float* directions, const float* radii, const float* basis, void* feat_out, int ld_feat,
                     float* means_out, float* covs_out, void* stream) {
  MNR_CHECK_ARG(cfg && B > 0 && n > 0 && tdist && origins && directions && radii && basis && feat_out,
                "mnr_cast_rays_ipe: null argument");
  MNR_CHECK_ARG(cfg->ray_shape == 0 || cfg->ray_shape == 1, "ray_shape must be 'cone' or 'cylinder'");  // render.py:124
  const int K = cfg->basis_k, L = cfg->max_deg - cfg->min_deg;
  MNR_CHECK_ARG(K >= 1 && K <= 128 && L >= 1 && L <= 32, "mnr_cast_rays_ipe: basis_k=%d / degrees=%d out of range", K, L);
  const bool f32 = mode == 1;
  const bool tangent = mode == 2;
  const int nfeat = 2 * K * L;
  const int row_elems = f32 ? nfeat : ld_feat;
  MNR_CHECK_ARG(f32 || (ld_feat >= nfeat && ld_feat % 8 == 0), "mnr_cast_rays_ipe: ld_feat=%d must be >= %d and a multiple of 8", ld_feat, nfeat);
  MNR_CHECK_ARG(!f32 || nfeat % 4 == 0, "mnr_cast_rays_ipe_f32: feature count must be a multiple of 4");
  const size_t row_bytes = (size_t)row_elems * (f32 ? 4 : sizeof(bf16));
  int spb = (int)(FE_STAGE_BYTES / (row_bytes * (tangent ? 3 : 1)));
  if (spb > FE_THREADS) spb = FE_THREADS;
  spb &= ~3;                       // keeps the row buffer 16-byte aligned behind the FeSample array
  MNR_CHECK_ARG(spb >= 4, "mnr_cast_rays_ipe: feature row too long");
  // 48 B of padding per staged row: consecutive samples then sit 12 banks apart (a wave covers ~3 samples x 21 directions)
  const int pitch = (int)row_bytes + 48;
  const size_t lds = (size_t)spb * (sizeof(FeSample) + (tangent ? sizeof(FeTangent) : 0)) + (size_t)((K * 3 + 3) & ~3) * 4 +
                     (size_t)spb * pitch * (tangent ? 3 : 1);
  const int64_t total = B * n;
  const int grid = mnr_cdiv(total, spb);
  if (tangent) {
    hipLaunchKernelGGL((cast_rays_ipe_kernel<false, true>), dim3(grid), dim3(FE_THREADS), lds, (hipStream_t)stream, *cfg,
                       total, n, spb, pitch, tdist, origins, directions, radii, basis, feat_out, ld_feat, means_out, covs_out);
  } else if (f32) {
    hipLaunchKernelGGL((cast_rays_ipe_kernel<true, false>), dim3(grid), dim3(FE_THREADS), lds, (hipStream_t)stream, *cfg,
                       total, n, spb, pitch, tdist, origins, directions, radii, basis, feat_out, ld_feat, means_out, covs_out);
  } else {
    hipLaunchKernelGGL((cast_rays_ipe_kernel<false, false>), dim3(grid), dim3(FE_THREADS), lds, (hipStream_t)stream, *cfg,
                       total, n, spb, pitch, tdist, origins, directions, radii, basis, feat_out, ld_feat, means_out, covs_out);
  }
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_cast_rays_ipe(const mnr_ipe_cfg* cfg, int64_t B, int n, const float* tdist,
                                 const float* origins, const float* directions, const float* radii,
                                 const float* basis, uint16_t* feat_out, int ld_feat, float* means_out,
                                 float* covs_out, void* stream) {
  return fe_launch(0, cfg, B, n, tdist, origins, directions, radii, basis, feat_out, ld_feat, means_out,
                   covs_out, stream);
}

extern "C" int mnr_cast_rays_ipe_f32(const mnr_ipe_cfg* cfg, int64_t B, int n, const float* tdist,
                                     const float* origins, const float* directions, const float* radii,
                                     const float* basis, float* feat_out, void* stream) {
  return fe_launch(1, cfg, B, n, tdist, origins, directions, radii, basis, feat_out, 0, nullptr, nullptr, stream);
}

extern "C" int mnr_cast_rays_ipe_tangent(const mnr_ipe_cfg* cfg, int64_t B, int n, const float* tdist,
                                         const float* origins, const float* directions, const float* radii,
                                         const float* basis, uint16_t* feat_out, int ld_feat, void* stream) {
  return fe_launch(2, cfg, B, n, tdist, origins, directions, radii, basis, feat_out, ld_feat, nullptr, nullptr, stream);
}

// ---------------------------------------------------------------------------
// View-direction positional encoding broadcast into the view-MLP input matrix.

__global__ void viewdir_enc_fill_kernel(int64_t total_rows, int n, const float* __restrict__ viewdirs,
                                        int deg_view, bf16* __restrict__ dst, int ld, int col0, int col_end) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= total_rows) return;
  const int64_t ray = row / n;
  const float x[3] = {viewdirs[ray * 3 + 0], viewdirs[ray * 3 + 1], viewdirs[ray * 3 + 2]};
  bf16* o = dst + row * ld + col0;
  const int nenc = 3 + 6 * deg_view;
  // coord.py:136-147: [x, sin(2^l x) (l-major), sin(2^l x + pi/2)], plain sin.
  int c = 0;
  for (int i = 0; i < 3; ++i) o[c++] = (bf16)x[i];
  for (int l = 0; l < deg_view; ++l)
    for (int i = 0; i < 3; ++i) o[c++] = (bf16)sinf(x[i] * ldexpf(1.0f, l));
  for (int l = 0; l < deg_view; ++l)
    for (int i = 0; i < 3; ++i) o[c++] = (bf16)sinf(x[i] * ldexpf(1.0f, l) + FE_PI_2);
  for (int k = col0 + nenc; k < col_end; ++k) dst[row * ld + k] = (bf16)0.0f;
}

// The same with 16-byte stores: a lane owns 8 consecutive columns of a row (the encoding is evaluated per column; the
// element-wise kernel above writes 2 bytes at a time, 0.16 ms per step for 28 MB at 360.gin).
// Every sample of a ray gets the same encoding: a thread evaluates its 8 columns ONCE and stores them into VD_SPT consecutive
// samples' rows (evaluated per row the kernel was bound by its 8 library sines per lane, 1.6 TB/s of stores: 80 us per step at
// 360.gin, 2 x 286 us at llff_raw's 2 M rows per level).
#define VD_SPT 8
__global__ __launch_bounds__(256) void viewdir_enc_fill_vec_kernel(int64_t total_rows, int n, const float* __restrict__ viewdirs,
                                                                   int deg_view, bf16* __restrict__ dst, int ld, int col0, int chunks) {
  const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t grp = item / chunks;
  const int ch = (int)(item - grp * chunks);
  const int gpr = (n + VD_SPT - 1) / VD_SPT;             // sample groups per ray
  const int64_t ray = grp / gpr;
  const int s0 = (int)(grp - ray * gpr) * VD_SPT;
  if (ray * n >= total_rows) return;
  const float x[3] = {viewdirs[ray * 3 + 0], viewdirs[ray * 3 + 1], viewdirs[ray * 3 + 2]};
  const int nenc = 3 + 6 * deg_view, half = 3 + 3 * deg_view;
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = ch * 8 + i;              // coord.py:136-147: [x, sin(2^l x) (l-major), sin(2^l x + pi/2)], plain sin
    float val = 0.0f;
    if (e < 3) {
      val = x[e];
    } else if (e < nenc) {
      const int k = e < half ? e - 3 : e - half;
      const float arg = x[k % 3] * ldexpf(1.0f, k / 3);
      val = sinf(e < half ? arg : arg + FE_PI_2);
    }
    v[i] = (bf16)val;
  }
  bf16* out = dst + (ray * n + s0) * ld + col0 + ch * 8;
  const int ns = min(VD_SPT, n - s0);
  for (int sidx = 0; sidx < ns; ++sidx) *(bf16x8*)(out + (int64_t)sidx * ld) = v;
}

extern "C" int mnr_viewdir_enc_fill(int64_t B, int n, const float* viewdirs, int deg_view, uint16_t* dst, int ld,
                                    int col0, int col_end, void* stream) {
  MNR_CHECK_ARG(B > 0 && n > 0 && viewdirs && dst && deg_view >= 0 && deg_view <= 16, "mnr_viewdir_enc_fill: bad arguments");
  MNR_CHECK_ARG(col0 + 3 + 6 * deg_view <= col_end && col_end <= ld, "mnr_viewdir_enc_fill: columns out of range");
  const int64_t rows = B * n;
  if (col0 % 8 == 0 && (col_end - col0) % 8 == 0 && ld % 8 == 0 && ((uintptr_t)dst % 16) == 0) {
    const int chunks = (col_end - col0) / 8;
    const int64_t items = B * (int64_t)((n + VD_SPT - 1) / VD_SPT) * chunks;
    hipLaunchKernelGGL(viewdir_enc_fill_vec_kernel, dim3(mnr_cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, rows,
                       n, viewdirs, deg_view, (bf16*)dst, ld, col0, chunks);
    MNR_CHECK_LAUNCH();
    return MNR_OK;
  }
  hipLaunchKernelGGL(viewdir_enc_fill_kernel, dim3(mnr_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, rows, n,
                     viewdirs, deg_view, (bf16*)dst, ld, col0, col_end);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// GLO vectors (models.py:101-110 nn.Embed lookup by cam_idx; :565-568 broadcast + concat into the view input).

__global__ void glo_fill_kernel(int64_t total, int n, int G, const float* __restrict__ table,
                                const int32_t* __restrict__ cam_idx, int num_embeddings, bf16* __restrict__ dst,
                                int ld, int col0) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t row = e / G;
  const int g = (int)(e % G);
  float v = 0.0f;                                             // zero_glo (models.py:109-110)
  if (cam_idx) {
    int idx = cam_idx[row / n];
    idx = idx < 0 ? 0 : (idx >= num_embeddings ? num_embeddings - 1 : idx);   // jnp gather clamps
    v = table[(int64_t)idx * G + g];
  }
  dst[row * ld + col0 + g] = (bf16)v;
}

extern "C" int mnr_glo_fill(int64_t B, int n, int G, const float* table, const int32_t* cam_idx, int num_embeddings,
                            uint16_t* dst, int ld, int col0, void* stream) {
  MNR_CHECK_ARG(B > 0 && n > 0 && G > 0 && dst && (cam_idx == nullptr || table), "mnr_glo_fill: bad arguments");
  MNR_CHECK_ARG(col0 >= 0 && col0 + G <= ld && num_embeddings > 0, "mnr_glo_fill: columns out of range");
  const int64_t total = B * n * G;
  hipLaunchKernelGGL(glo_fill_kernel, dim3(mnr_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, total, n, G, table,
                     cam_idx, num_embeddings, (bf16*)dst, ld, col0);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// VJP: grad_table[cam_idx[b], g] += sum_i (g_a[b*n+i, g] + g_b[b*n+i, g]).
__global__ void glo_bwd_kernel(int64_t B, int n, int G, const float* __restrict__ g_a, const float* __restrict__ g_b,
                               const int32_t* __restrict__ cam_idx, int num_embeddings, float* grad_table) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * G) return;
  const int64_t b = e / G;
  const int g = (int)(e % G);
  float acc = 0.0f;
  for (int i = 0; i < n; ++i) {
    acc += g_a[(b * n + i) * G + g];
    if (g_b) acc += g_b[(b * n + i) * G + g];
  }
  int idx = cam_idx[b];
  idx = idx < 0 ? 0 : (idx >= num_embeddings ? num_embeddings - 1 : idx);
  unsafeAtomicAdd(grad_table + (int64_t)idx * G + g, acc);
}

extern "C" int mnr_glo_bwd(int64_t B, int n, int G, const float* g_a, const float* g_b, const int32_t* cam_idx,
                           int num_embeddings, float* grad_table, void* stream) {
  MNR_CHECK_ARG(B > 0 && n > 0 && G > 0 && g_a && cam_idx && grad_table && num_embeddings > 0, "mnr_glo_bwd: bad arguments");
  hipLaunchKernelGGL(glo_bwd_kernel, dim3(mnr_cdiv(B * G, 256)), dim3(256), 0, (hipStream_t)stream, B, n, G, g_a, g_b,
                     cam_idx, num_embeddings, grad_table);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// VJP of cast_rays_ipe_kernel with respect to the interval ends (Model.stop_level_grad = False, models.py:198-201: the sample
// distances then carry gradient, and the features are a function of them through render.cast_rays render.py:103-127,
// coord.track_linearize(contract) coord.py:21-60, lift_and_diagonalize :129-133 and integrated_pos_enc :102-126).
//
// Input: g_feat [M, ld] bf16 = d loss / d features (the dX GEMM of trunk layer 0, plus that of the skip layer's feature
// segment in g_feat_b).  Output: g_t0 [M], g_t1 [M] = d loss / d (t0, t1) of every sample.
//
// Reverse mode down to the per-sample Gaussian, forward mode below it:
//   phase 1: one thread per sample: the Gaussian (fe_gaussian: the forward pass's own code)
//   phase 2: one thread per (sample, basis direction k): the features' sin / cos / attenuation once more, and
//            g_lm[k] = sum_l 2^l (g_sin att cos - g_cos att sin),  g_lv[k] = -1/2 sum_l 4^l (g_sin att sin + g_cos att cos)
//   phase 3: one thread per sample: g_mean = sum_k g_lm[k] p_k, G = sum_k g_lv[k] p_k p_k^T (9 numbers), then the Gaussian
//            again on DUAL numbers in (t0, t1) -- the conical-frustum moments, the lift and the contraction with its Jacobian
//            are differentiated by carrying (value, d/dt0, d/dt1) through the same closed forms (the contraction's second
//            derivative never has to be written down) -- and g_t = g_mean . d mean/dt + <G, d cov/dt>.
// The features reach the MLP rounded to bf16; the rounding is treated as the identity (straight through), as autodiff does.

struct FeD2 {
  float v, a, b;           // value, d/dt0, d/dt1
};
__device__ __forceinline__ FeD2 fe_d2(float v) { return FeD2{v, 0.0f, 0.0f}; }
__device__ __forceinline__ FeD2 operator+(FeD2 x, FeD2 y) { return FeD2{x.v + y.v, x.a + y.a, x.b + y.b}; }
__device__ __forceinline__ FeD2 operator-(FeD2 x, FeD2 y) { return FeD2{x.v - y.v, x.a - y.a, x.b - y.b}; }
__device__ __forceinline__ FeD2 operator*(FeD2 x, FeD2 y) { return FeD2{x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b}; }
__device__ __forceinline__ FeD2 operator*(float s, FeD2 x) { return FeD2{s * x.v, s * x.a, s * x.b}; }
__device__ __forceinline__ FeD2 operator*(FeD2 x, float s) { return FeD2{s * x.v, s * x.a, s * x.b}; }
__device__ __forceinline__ FeD2 operator+(FeD2 x, float s) { return FeD2{x.v + s, x.a, x.b}; }
__device__ __forceinline__ FeD2 operator+(float s, FeD2 x) { return FeD2{x.v + s, x.a, x.b}; }
__device__ __forceinline__ FeD2 operator-(FeD2 x, float s) { return FeD2{x.v - s, x.a, x.b}; }
__device__ __forceinline__ FeD2 operator-(float s, FeD2 x) { return FeD2{s - x.v, -x.a, -x.b}; }
__device__ __forceinline__ FeD2 operator/(FeD2 x, FeD2 y) {
  const float q = x.v / y.v;
  return FeD2{q, (x.a - q * y.a) / y.v, (x.b - q * y.b) / y.v};
}
__device__ __forceinline__ FeD2 operator/(FeD2 x, float s) { return FeD2{x.v / s, x.a / s, x.b / s}; }
__device__ __forceinline__ FeD2 fe_sqrt(FeD2 x) {
  const float r = sqrtf(x.v);
  return FeD2{r, 0.5f * x.a / r, 0.5f * x.b / r};
}
__device__ __forceinline__ FeD2 fe_max_const(float lo, FeD2 x) { return x.v >= lo ? x : fe_d2(lo); }   // max(lo, x)

// fe_gaussian on dual numbers: mean[3], cov[6] (xx, xy, xz, yy, yz, zz) as functions of (t0, t1).
__device__ __forceinline__ void fe_gaussian_dual(const mnr_ipe_cfg& c, float t0v, float t1v, const float* o, const float* d,
                                                 float radius, FeD2* mean, FeD2* cov) {
  const FeD2 t0 = {t0v, 1.0f, 0.0f}, t1 = {t1v, 0.0f, 1.0f};
  FeD2 t_mean, t_var, r_var;
  if (c.ray_shape == 0) {
    const FeD2 mu = (t0 + t1) / 2.0f;
    const FeD2 hw = (t1 - t0) / 2.0f;
    const FeD2 denom = fe_max_const(MNR_F32_EPS, 3.0f * (mu * mu) + hw * hw);
    const FeD2 hw2 = hw * hw, hw4 = hw2 * hw2;
    t_mean = mu + (2.0f * (mu * hw2)) / denom;
    t_var = hw2 / 3.0f - ((4.0f / 15.0f) * (hw4 * (12.0f * (mu * mu) - hw2))) / (denom * denom);
    r_var = (mu * mu) / 4.0f + (5.0f / 12.0f) * hw2 - ((4.0f / 15.0f) * hw4) / denom;
    r_var = r_var * (radius * radius);
  } else {
    t_mean = (t0 + t1) / 2.0f;
    r_var = fe_d2(radius * radius / 4.0f);
    t_var = ((t1 - t0) * (t1 - t0)) / 12.0f;
  }
  const float dmag = fmaxf(1e-10f, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) mean[i] = o[i] + d[i] * t_mean;
  if (c.disable_integration) {
    t_var = fe_d2(0.0f);
    r_var = fe_d2(0.0f);
  }
  const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
  for (int e = 0; e < 6; ++e) {
    const int i = ii[e], j = jj[e];
    const float dd = d[i] * d[j];
    const float null_outer = (i == j ? 1.0f : 0.0f) - d[i] * (d[j] / dmag);
    cov[e] = t_var * dd + r_var * null_outer;
  }
  if (c.warp_contract) {
    // (the structured form of fe_gaussian: J cov J^T = ku u u^T + r_var (s^2 I + j2x x x^T), mean' = s x)
    FeD2 m = mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2];
    m = fe_max_const(MNR_F32_EPS, m);
    if (!(m.v <= 1.0f)) {
      const FeD2 sq = fe_sqrt(m);
      const FeD2 s = (2.0f * sq - 1.0f) / m;
      const FeD2 cc = (2.0f * (1.0f - sq)) / (m * m);
      const float oo = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
      const float od = o[0] * d[0] + o[1] * d[1] + o[2] * d[2];
      const FeD2 xd = mean[0] * d[0] + mean[1] * d[1] + mean[2] * d[2];
      const FeD2 brk = (m - (2.0f * (1.0f - sq)) * (oo + t_mean * od)) / (m * m);
      FeD2 u[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) u[i] = brk * d[i] + (cc * xd) * o[i];
      const FeD2 j2x = (2.0f * cc) / sq;
      const FeD2 ku = t_var - r_var / dmag;
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const int i = ii[e], j = jj[e];
        FeD2 inner = j2x * (mean[i] * mean[j]);
        if (i == j) inner = inner + s * s;
        cov[e] = ku * (u[i] * u[j]) + r_var * inner;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) mean[i] = s * mean[i];
    }
  }
}

__global__ __launch_bounds__(FE_THREADS) void cast_rays_ipe_bwd_kernel(
    mnr_ipe_cfg c, int64_t total, int n, int spb, int pitch, const float* __restrict__ tdist, const float* __restrict__ origins,
    const float* __restrict__ directions, const float* __restrict__ radii, const float* __restrict__ basis,
    const bf16* __restrict__ g_feat_a, const bf16* __restrict__ g_feat_b, int ld_feat, float* __restrict__ g_t0,
    float* __restrict__ g_t1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int K = c.basis_k;
  const int L = c.max_deg - c.min_deg;
  const int nfeat = 2 * K * L;
  // LDS: samples [spb] FeSample | basis [K*3] | g rows [spb][pitch] f32 | per-direction partials [spb][K][2] f32
  FeSample* gs = (FeSample*)smem;
  float* bs = (float*)(gs + spb);
  float* rows = bs + ((K * 3 + 3) & ~3);
  float* part = rows + (size_t)spb * pitch;
  const int64_t s0 = (int64_t)blockIdx.x * spb;
  const int ns = (int)min((int64_t)spb, total - s0);
  for (int i = threadIdx.x; i < K * 3; i += FE_THREADS) bs[i] = basis[i];
  if (threadIdx.x < ns) {
    const int64_t s = s0 + threadIdx.x;
    const int64_t ray = s / n;
    const int j = (int)(s % n);
    const float t0 = tdist[ray * (n + 1) + j], t1 = tdist[ray * (n + 1) + j + 1];
    float o[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      o[i] = origins[ray * 3 + i];
      d[i] = directions[ray * 3 + i];
    }
    FeSample g;
    fe_gaussian(c, t0, t1, o, d, radii[ray], g);
    gs[threadIdx.x] = g;
  }
  // the block's gradient rows: 16-byte loads (8 bf16), summed over the two sources, kept as fp32
  const int cpr = ld_feat >> 3;
  for (int ch = threadIdx.x; ch < ns * cpr; ch += FE_THREADS) {
    const int r = ch / cpr, c0 = (ch - r * cpr) << 3;
    const bf16x8 va = *(const bf16x8*)(g_feat_a + (size_t)(s0 + r) * ld_feat + c0);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)va[i];
    if (g_feat_b) {
      const bf16x8 vb = *(const bf16x8*)(g_feat_b + (size_t)(s0 + r) * ld_feat + c0);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += (float)vb[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (c0 + i < nfeat) rows[(size_t)r * pitch + c0 + i] = v[i];
  }
  __syncthreads();
  const float inv_k = 1.0f / (float)K;
  for (int pair = threadIdx.x; pair < ns * K; pair += FE_THREADS) {
    const int si = (int)(((float)pair + 0.5f) * inv_k);
    const int k = pair - si * K;
    const FeSample g = gs[si];
    const float px = bs[k * 3 + 0], py = bs[k * 3 + 1], pz = bs[k * 3 + 2];
    const float lm = g.mean[0] * px + g.mean[1] * py + g.mean[2] * pz;
    const float cx = g.cov[0] * px + g.cov[1] * py + g.cov[2] * pz;
    const float cy = g.cov[1] * px + g.cov[3] * py + g.cov[4] * pz;
    const float cz = g.cov[2] * px + g.cov[4] * py + g.cov[5] * pz;
    const float lv = px * cx + py * cy + pz * cz;
    const float vscale = -0.5f * 1.44269504088896340736f * lv;
    const float* grow = rows + (size_t)si * pitch + k;
    const int half = K * L;
    float sc = ldexpf(1.0f, c.min_deg);
    float sn = 0.0f, cs = 1.0f, att = 1.0f;
    float g_lm = 0.0f, g_lv = 0.0f;
    for (int l = 0; l < L; ++l) {
      if ((l & 3) == 0) {
        fe_sincos_wrapped(fe_wrap_100pi(lm * sc), &sn, &cs);
        att = exp2f(vscale * sc * sc);
      }
      const float fs = att * sn, fc = att * cs;
      const float gsn = grow[l * K], gcs = grow[half + l * K];
      g_lm += sc * (gsn * fc - gcs * fs);
      g_lv += -0.5f * sc * sc * (gsn * fs + gcs * fc);
      const float s2 = 2.0f * sn * cs;
      cs = 1.0f - 2.0f * sn * sn;
      sn = s2;
      const float a2 = att * att;
      att = a2 * a2;
      sc *= 2.0f;
    }
    part[(size_t)pair * 2] = g_lm;
    part[(size_t)pair * 2 + 1] = g_lv;
  }
  __syncthreads();
  if (threadIdx.x < ns) {
    const int si = threadIdx.x;
    float gm[3] = {0.0f, 0.0f, 0.0f}, G[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < K; ++k) {
      const float px = bs[k * 3 + 0], py = bs[k * 3 + 1], pz = bs[k * 3 + 2];
      const float a = part[(size_t)(si * K + k) * 2], v = part[(size_t)(si * K + k) * 2 + 1];
      gm[0] += a * px; gm[1] += a * py; gm[2] += a * pz;
      G[0] += v * px * px; G[1] += v * px * py; G[2] += v * px * pz;
      G[3] += v * py * py; G[4] += v * py * pz; G[5] += v * pz * pz;
    }
    const int64_t s = s0 + si;
    const int64_t ray = s / n;
    const int j = (int)(s % n);
    const float t0 = tdist[ray * (n + 1) + j], t1 = tdist[ray * (n + 1) + j + 1];
    float o[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      o[i] = origins[ray * 3 + i];
      d[i] = directions[ray * 3 + i];
    }
    FeD2 mean[3], cov[6];
    fe_gaussian_dual(c, t0, t1, o, d, radii[ray], mean, cov);
    const float wgt[6] = {1.0f, 2.0f, 2.0f, 1.0f, 2.0f, 1.0f};      // the symmetric matrix's off-diagonal entries count twice
    float ga = 0.0f, gb = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      ga += gm[i] * mean[i].a;
      gb += gm[i] * mean[i].b;
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      ga += wgt[e] * G[e] * cov[e].a;
      gb += wgt[e] * G[e] * cov[e].b;
    }
    g_t0[s] = ga;
    g_t1[s] = gb;
  }
}

extern "C" int mnr_cast_rays_ipe_bwd(const mnr_ipe_cfg* cfg, int64_t B, int n, const float* tdist, const float* origins,
                                     const float* directions, const float* radii, const float* basis, const uint16_t* g_feat_a,
                                     const uint16_t* g_feat_b, int ld_feat, float* g_t0, float* g_t1, void* stream) {
  MNR_CHECK_ARG(cfg && B > 0 && n > 0 && tdist && origins && directions && radii && basis && g_feat_a && g_t0 && g_t1,
                "mnr_cast_rays_ipe_bwd: null argument");
  MNR_CHECK_ARG(cfg->ray_shape == 0 || cfg->ray_shape == 1, "ray_shape must be 'cone' or 'cylinder'");
  const int K = cfg->basis_k, L = cfg->max_deg - cfg->min_deg;
  MNR_CHECK_ARG(K >= 1 && K <= 128 && L >= 1 && L <= 32, "mnr_cast_rays_ipe_bwd: basis_k=%d / degrees=%d out of range", K, L);
  const int nfeat = 2 * K * L;
  MNR_CHECK_ARG(ld_feat >= nfeat && ld_feat % 8 == 0 && ((uintptr_t)g_feat_a % 16) == 0 && ((uintptr_t)g_feat_b % 16) == 0,
                "mnr_cast_rays_ipe_bwd: ld_feat=%d must be >= %d and a multiple of 8, rows 16-byte aligned", ld_feat, nfeat);
  const int pitch = nfeat | 1;                                  // floats; odd: the (sample, direction) threads' column reads spread over the banks
  int spb = (int)((40 * 1024) / ((size_t)pitch * 4 + (size_t)K * 8));
  if (spb > 64) spb = 64;
  spb &= ~3;
  MNR_CHECK_ARG(spb >= 4, "mnr_cast_rays_ipe_bwd: feature row too long");
  const size_t lds = (size_t)spb * sizeof(FeSample) + (size_t)((K * 3 + 3) & ~3) * 4 + (size_t)spb * pitch * 4 + (size_t)spb * K * 8;
  const int64_t total = B * n;
  hipLaunchKernelGGL(cast_rays_ipe_bwd_kernel, dim3(mnr_cdiv(total, spb)), dim3(FE_THREADS), lds, (hipStream_t)stream, *cfg, total,
                     n, spb, pitch, tdist, origins, directions, radii, basis, (const bf16*)g_feat_a, (const bf16*)g_feat_b, ld_feat,
                     g_t0, g_t1);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// VJP of cast_rays_ipe_kernel<.., TANGENT> with respect to the interval ends: Model.stop_level_grad = False NEXT TO
// density-gradient normals (models.py:198-201 with :478-492).  The normals are -l2_normalize(d raw_density / d mean), computed in
// forward mode through the tangent network, whose input rows
//   T_sin[c] = att 2^l cos(lm 2^l) dlm_c - 1/2 4^l att sin(lm 2^l) dlv_c,   T_cos[c] = -att 2^l sin(.) dlm_c - 1/2 4^l att cos(.) dlv_c
// (c = x, y, z; per basis direction k and degree l; lm = p_k . mean', lv = p_k^T cov' p_k, att = exp(-1/2 4^l lv), dlm_c = p_k . dz[c],
// dlv_c = p_k^T dC[c] p_k with dz / dC the contraction's tangents, fe_contract_tangent) depend on the sample's (t0, t1) as well.
//
// Input: g_T [3 M, ld] = d loss / d (tangent rows) (the tangent network's dX GEMM of trunk layer 0, plus that of the skip
// layer's feature segment in g_T_b).  Output: g_t0, g_t1 [M] += d loss / d (t0, t1)  (ACCUMULATED onto mnr_cast_rays_ipe_bwd's).
//
// Reverse mode down to the eight scalars (lm, lv, dlm_c, dlv_c) of a (sample, direction), forward mode below them:
//   phase 1: one thread per sample: the Gaussian and the contraction's tangents (the forward pass's own code)
//   phase 2: one thread per (sample, k): sin / cos / attenuation once more, and with S = sin, C = cos, a = att, sc = 2^l, hv = -1/2 4^l
//              d T_sin[c] / d lm = -a sc^2 S dlm_c + hv a sc C dlv_c      d T_cos[c] / d lm = -a sc^2 C dlm_c - hv a sc S dlv_c
//              d T_x[c] / d lv = hv T_x[c]      d T_sin[c] / d dlm_c = a sc C   d T_cos[c] / d dlm_c = -a sc S
//              d T_sin[c] / d dlv_c = hv a S    d T_cos[c] / d dlv_c = hv a C
//   phase 3: one thread per sample: g_mean' = sum_k g_lm p_k, G_cov' = sum_k g_lv p_k p_k^T, g_dz[c] = sum_k g_dlm[c] p_k,
//            G_dC[c] = sum_k g_dlv[c] p_k p_k^T, then the Gaussian AND the contraction's tangents on dual numbers in (t0, t1)
//            (fe_gaussian_dual, fe_contract_tangent_dual: the same closed forms carrying (value, d/dt0, d/dt1), so the
//            contraction's THIRD derivative never has to be written down) and the inner products with those duals.

// fe_contract_tangent on dual numbers.
__device__ __forceinline__ void fe_contract_tangent_dual(const mnr_ipe_cfg& c, float t0v, float t1v, const float* o, const float* d,
                                                         float radius, FeD2 (&dz)[3][3], FeD2 (&dC)[3][6]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int i = 0; i < 3; ++i) dz[k][i] = fe_d2((i == k) ? 1.0f : 0.0f);
#pragma unroll
    for (int e = 0; e < 6; ++e) dC[k][e] = fe_d2(0.0f);
  }
  if (!c.warp_contract) return;
  const FeD2 t0 = {t0v, 1.0f, 0.0f}, t1 = {t1v, 0.0f, 1.0f};
  FeD2 t_mean, t_var, r_var;
  if (c.ray_shape == 0) {
    const FeD2 mu = (t0 + t1) / 2.0f;
    const FeD2 hw = (t1 - t0) / 2.0f;
    const FeD2 denom = fe_max_const(MNR_F32_EPS, 3.0f * (mu * mu) + hw * hw);
    const FeD2 hw2 = hw * hw, hw4 = hw2 * hw2;
    t_mean = mu + (2.0f * (mu * hw2)) / denom;
    t_var = hw2 / 3.0f - ((4.0f / 15.0f) * (hw4 * (12.0f * (mu * mu) - hw2))) / (denom * denom);
    r_var = (mu * mu) / 4.0f + (5.0f / 12.0f) * hw2 - ((4.0f / 15.0f) * hw4) / denom;
    r_var = r_var * (radius * radius);
  } else {
    t_mean = (t0 + t1) / 2.0f;
    r_var = fe_d2(radius * radius / 4.0f);
    t_var = ((t1 - t0) * (t1 - t0)) / 12.0f;
  }
  if (c.disable_integration) {
    t_var = fe_d2(0.0f);
    r_var = fe_d2(0.0f);
  }
  const float dmag = fmaxf(1e-10f, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  FeD2 x[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) x[i] = o[i] + d[i] * t_mean;
  const FeD2 m = fe_max_const(MNR_F32_EPS, x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  if (m.v <= 1.0f) return;
  const FeD2 sq = fe_sqrt(m);
  const FeD2 s = (2.0f * sq - 1.0f) / m;
  const FeD2 m2 = m * m;
  const FeD2 cc = (2.0f * (1.0f - sq)) / m2;
  const FeD2 ccp = (3.0f * sq - 4.0f) / (m2 * m);                  // d cc / d m
  const FeD2 j2x = (2.0f * cc) / sq;
  const FeD2 j2xp = (2.0f * ccp - cc / m) / sq;                    // d (2 cc / sqrt(m)) / d m
  const float oo = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
  const float od = o[0] * d[0] + o[1] * d[1] + o[2] * d[2];
  const FeD2 xd = x[0] * d[0] + x[1] * d[1] + x[2] * d[2];
  const FeD2 brk = (m - (2.0f * (1.0f - sq)) * (oo + t_mean * od)) / m2;
  FeD2 u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = brk * d[i] + (cc * xd) * o[i];
  const FeD2 ku = t_var - r_var / dmag;
  const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const FeD2 xk = x[k];
    FeD2 du[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      FeD2 lin = xk * d[i] + d[k] * x[i];
      if (i == k) lin = lin + xd;
      du[i] = cc * lin + (2.0f * ccp) * (xk * xd) * x[i];
      dz[k][i] = cc * (xk * x[i]);
      if (i == k) dz[k][i] = dz[k][i] + s;
    }
    const FeD2 ds = cc * xk;
    const FeD2 dj = (2.0f * j2xp) * xk;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int i = ii[e], j = jj[e];
      const FeD2 v = ku * (du[i] * u[j] + u[i] * du[j]);
      FeD2 w = dj * (x[i] * x[j]);
      if (i == k) w = w + j2x * x[j];
      if (j == k) w = w + j2x * x[i];
      if (i == j) w = w + (2.0f * s) * ds;
      dC[k][e] = v + r_var * w;
    }
  }
}

#define FE_TB_SPB 32                                    // samples per block of the tangent VJP (8 partial sums per (sample, direction))

__global__ __launch_bounds__(FE_THREADS) void cast_rays_ipe_tangent_bwd_kernel(
    mnr_ipe_cfg c, int64_t total, int n, const float* __restrict__ tdist, const float* __restrict__ origins,
    const float* __restrict__ directions, const float* __restrict__ radii, const float* __restrict__ basis,
    const bf16* __restrict__ g_T_a, const bf16* __restrict__ g_T_b, int ld_feat, float* __restrict__ g_t0, float* __restrict__ g_t1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int K = c.basis_k;
  const int L = c.max_deg - c.min_deg;
  // LDS: samples [spb] FeSample | tangents [spb] FeTangent | basis [K*3] | partials [spb][K][8]
  FeSample* gs = (FeSample*)smem;
  FeTangent* gt = (FeTangent*)(gs + FE_TB_SPB);
  float* bs = (float*)(gt + FE_TB_SPB);
  float* part = bs + ((K * 3 + 3) & ~3);
  const int64_t s0 = (int64_t)blockIdx.x * FE_TB_SPB;
  const int ns = (int)min((int64_t)FE_TB_SPB, total - s0);
  for (int i = threadIdx.x; i < K * 3; i += FE_THREADS) bs[i] = basis[i];
  if (threadIdx.x < ns) {
    const int64_t s = s0 + threadIdx.x;
    const int64_t ray = s / n;
    const int j = (int)(s % n);
    const float t0 = tdist[ray * (n + 1) + j], t1 = tdist[ray * (n + 1) + j + 1];
    float o[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      o[i] = origins[ray * 3 + i];
      d[i] = directions[ray * 3 + i];
    }
    FeSample g;
    fe_gaussian(c, t0, t1, o, d, radii[ray], g);
    gs[threadIdx.x] = g;
    FeTangent T;
    fe_contract_tangent(c, t0, t1, o, d, radii[ray], T);
    gt[threadIdx.x] = T;
  }
  __syncthreads();
  const int half = K * L;
  for (int pair = threadIdx.x; pair < ns * K; pair += FE_THREADS) {
    const int si = pair / K;
    const int k = pair - si * K;
    const FeSample g = gs[si];
    const FeTangent& T = gt[si];
    const float px = bs[k * 3 + 0], py = bs[k * 3 + 1], pz = bs[k * 3 + 2];
    const float lm = g.mean[0] * px + g.mean[1] * py + g.mean[2] * pz;
    const float cx = g.cov[0] * px + g.cov[1] * py + g.cov[2] * pz;
    const float cy = g.cov[1] * px + g.cov[3] * py + g.cov[4] * pz;
    const float cz = g.cov[2] * px + g.cov[4] * py + g.cov[5] * pz;
    const float lv = px * cx + py * cy + pz * cz;
    const float vscale = -0.5f * 1.44269504088896340736f * lv;
    float dlm[3], dlv[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      dlm[cc] = px * T.dz[cc][0] + py * T.dz[cc][1] + pz * T.dz[cc][2];
      const float* C6 = T.dC[cc];
      dlv[cc] = px * (C6[0] * px + C6[1] * py + C6[2] * pz) + py * (C6[1] * px + C6[3] * py + C6[4] * pz) +
                pz * (C6[2] * px + C6[4] * py + C6[5] * pz);
    }
    float sc = ldexpf(1.0f, c.min_deg);
    float sn = 0.0f, cs = 1.0f, att = 1.0f;
    float g_lm = 0.0f, g_lv = 0.0f, g_dlm[3] = {0.0f, 0.0f, 0.0f}, g_dlv[3] = {0.0f, 0.0f, 0.0f};
    for (int l = 0; l < L; ++l) {
      if ((l & 3) == 0) {
        fe_sincos_wrapped(fe_wrap_100pi(lm * sc), &sn, &cs);
        att = exp2f(vscale * sc * sc);
      }
      const float hv = -0.5f * sc * sc;
      const float aS = att * sn, aC = att * cs;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const size_t row = ((size_t)cc * total + s0 + si) * ld_feat + (size_t)l * K + k;
        float gsn = (float)g_T_a[row], gcs = (float)g_T_a[row + half];
        if (g_T_b) {
          gsn += (float)g_T_b[row];
          gcs += (float)g_T_b[row + half];
        }
        const float Tsn = aC * sc * dlm[cc] + hv * aS * dlv[cc];
        const float Tcs = -aS * sc * dlm[cc] + hv * aC * dlv[cc];
        g_lm += gsn * (-aS * sc * sc * dlm[cc] + hv * aC * sc * dlv[cc]) + gcs * (-aC * sc * sc * dlm[cc] - hv * aS * sc * dlv[cc]);
        g_lv += hv * (gsn * Tsn + gcs * Tcs);
        g_dlm[cc] += sc * (gsn * aC - gcs * aS);
        g_dlv[cc] += hv * (gsn * aS + gcs * aC);
      }
      const float s2 = 2.0f * sn * cs;
      cs = 1.0f - 2.0f * sn * sn;
      sn = s2;
      const float a2 = att * att;
      att = a2 * a2;
      sc *= 2.0f;
    }
    float* pp = part + (size_t)pair * 8;
    pp[0] = g_lm;
    pp[1] = g_lv;
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      pp[2 + cc] = g_dlm[cc];
      pp[5 + cc] = g_dlv[cc];
    }
  }
  __syncthreads();
  if (threadIdx.x < ns) {
    const int si = threadIdx.x;
    float gm[3] = {0.0f, 0.0f, 0.0f}, G[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    float gz[3][3], GC[3][6];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
#pragma unroll
      for (int i = 0; i < 3; ++i) gz[cc][i] = 0.0f;
#pragma unroll
      for (int e = 0; e < 6; ++e) GC[cc][e] = 0.0f;
    }
    for (int k = 0; k < K; ++k) {
      const float p[3] = {bs[k * 3 + 0], bs[k * 3 + 1], bs[k * 3 + 2]};
      const float pp6[6] = {p[0] * p[0], p[0] * p[1], p[0] * p[2], p[1] * p[1], p[1] * p[2], p[2] * p[2]};
      const float* pr = part + (size_t)(si * K + k) * 8;
#pragma unroll
      for (int i = 0; i < 3; ++i) gm[i] += pr[0] * p[i];
#pragma unroll
      for (int e = 0; e < 6; ++e) G[e] += pr[1] * pp6[e];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
#pragma unroll
        for (int i = 0; i < 3; ++i) gz[cc][i] += pr[2 + cc] * p[i];
#pragma unroll
        for (int e = 0; e < 6; ++e) GC[cc][e] += pr[5 + cc] * pp6[e];
      }
    }
    const int64_t s = s0 + si;
    const int64_t ray = s / n;
    const int j = (int)(s % n);
    const float t0 = tdist[ray * (n + 1) + j], t1 = tdist[ray * (n + 1) + j + 1];
    float o[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      o[i] = origins[ray * 3 + i];
      d[i] = directions[ray * 3 + i];
    }
    FeD2 mean[3], cov[6];
    fe_gaussian_dual(c, t0, t1, o, d, radii[ray], mean, cov);
    FeD2 dz[3][3], dC[3][6];
    fe_contract_tangent_dual(c, t0, t1, o, d, radii[ray], dz, dC);
    const float wgt[6] = {1.0f, 2.0f, 2.0f, 1.0f, 2.0f, 1.0f};      // the symmetric matrices' off-diagonal entries count twice
    float ga = 0.0f, gb = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      ga += gm[i] * mean[i].a;
      gb += gm[i] * mean[i].b;
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      ga += wgt[e] * G[e] * cov[e].a;
      gb += wgt[e] * G[e] * cov[e].b;
    }
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        ga += gz[cc][i] * dz[cc][i].a;
        gb += gz[cc][i] * dz[cc][i].b;
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        ga += wgt[e] * GC[cc][e] * dC[cc][e].a;
        gb += wgt[e] * GC[cc][e] * dC[cc][e].b;
      }
    }
    g_t0[s] += ga;
    g_t1[s] += gb;
  }
}

extern "C" int mnr_cast_rays_ipe_tangent_bwd(const mnr_ipe_cfg* cfg, int64_t B, int n, const float* tdist, const float* origins,
                                             const float* directions, const float* radii, const float* basis, const uint16_t* g_T_a,
                                             const uint16_t* g_T_b, int ld_feat, float* g_t0, float* g_t1, void* stream) {
  MNR_CHECK_ARG(cfg && B > 0 && n > 0 && tdist && origins && directions && radii && basis && g_T_a && g_t0 && g_t1,
                "mnr_cast_rays_ipe_tangent_bwd: null argument");
  MNR_CHECK_ARG(cfg->ray_shape == 0 || cfg->ray_shape == 1, "ray_shape must be 'cone' or 'cylinder'");
  const int K = cfg->basis_k, L = cfg->max_deg - cfg->min_deg;
  MNR_CHECK_ARG(K >= 1 && K <= 128 && L >= 1 && L <= 32, "mnr_cast_rays_ipe_tangent_bwd: basis_k=%d / degrees=%d out of range", K, L);
  MNR_CHECK_ARG(ld_feat >= 2 * K * L, "mnr_cast_rays_ipe_tangent_bwd: ld_feat=%d must be >= %d", ld_feat, 2 * K * L);
  const size_t lds = (size_t)FE_TB_SPB * (sizeof(FeSample) + sizeof(FeTangent)) + (size_t)((K * 3 + 3) & ~3) * 4 +
                     (size_t)FE_TB_SPB * K * 8 * 4;
  const int64_t total = B * n;
  hipLaunchKernelGGL(cast_rays_ipe_tangent_bwd_kernel, dim3(mnr_cdiv(total, FE_TB_SPB)), dim3(FE_THREADS), lds, (hipStream_t)stream,
                     *cfg, total, n, tdist, origins, directions, radii, basis, (const bf16*)g_T_a, (const bf16*)g_T_b, ld_feat, g_t0,
                     g_t1);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}
