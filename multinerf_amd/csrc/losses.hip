// Training losses and their VJPs on device (gfx950).
//
// Replaces train_utils.compute_data_loss (train_utils.py:72-136), interlevel_loss
// (:139-150) with stepfun.lossfun_outer / inner_outer / searchsorted
// (stepfun.py:30-86) and distortion_loss (:153-159) with stepfun.lossfun_distortion
// (stepfun.py:266-276).  The reference builds [B,n_env,n] comparison masks for
// searchsorted and a [B,n,n] |u_i-u_j| tensor; both operands are sorted per ray, so
// here one lane walks one ray with monotone cursors (O(n + n_env)) and the pairwise
// distortion term is an in-LDS double loop.  Scalars are reduced per wave and added
// atomically to a small fp32 stats array.
#include "common.h"

#include "ray_losses.h"

__global__ void lossmult_sum_kernel(int64_t B, const float* __restrict__ lm, int lm_c, float* out) {
  float s = 0.0f;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
    if (lm_c == 1) s += 3.0f * lm[b];
    else s += lm[b * 3] + lm[b * 3 + 1] + lm[b * 3 + 2];
  }
  s = ls_wave_sum(s);
  if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(out, s);
}

extern "C" int mnr_lossmult_sum(int64_t B_valid, const float* lossmult, int lm_c, float* out, void* stream) {
  MNR_CHECK_ARG(B_valid > 0 && lossmult && out && (lm_c == 1 || lm_c == 3), "mnr_lossmult_sum: bad arguments");
  int grid = mnr_cdiv(B_valid, 256);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(lossmult_sum_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, B_valid, lossmult, lm_c, out);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// Training-time metrics of compute_data_loss (train_utils.py:113-128): no gradient, one block.
//   out_disp   = mean_b (1 / (1 + distance_mean_b) - disps_b)^2
//   out_normal = sum_b w_b acos(clip(n_b . ngt_b)) / sum_b w_b * 180/pi,  w = acc * alphas, n / ngt l2-normalised
__global__ __launch_bounds__(1024) void render_metrics_kernel(int64_t B_valid, const float* __restrict__ distance_mean,
                                                              const float* __restrict__ disps,
                                                              const float* __restrict__ acc,
                                                              const float* __restrict__ alphas,
                                                              const float* __restrict__ normals,
                                                              const float* __restrict__ normals_gt,
                                                              float* out_disp, float* out_normal) {
  __shared__ float red[3][16];
  float sd = 0.0f, sn = 0.0f, sw = 0.0f;
  for (int64_t b = threadIdx.x; b < B_valid; b += blockDim.x) {
    if (out_disp) {
      const float d = 1.0f / (1.0f + distance_mean[b]) - disps[b];
      sd += d * d;
    }
    if (out_normal) {
      const float w = acc[b] * alphas[b];
      float n[3], g[3], nn = 0.0f, gg = 0.0f, dot = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        n[c] = normals[b * 3 + c];
        g[c] = normals_gt[b * 3 + c];
        nn += n[c] * n[c];
        gg += g[c] * g[c];
      }
      const float rn = 1.0f / sqrtf(fmaxf(nn, MNR_F32_EPS)), rg = 1.0f / sqrtf(fmaxf(gg, MNR_F32_EPS));   // ref_utils.py:40-42
#pragma unroll
      for (int c = 0; c < 3; ++c) dot += (n[c] * rn) * (g[c] * rg);
      const float one_eps = 1.0f - MNR_F32_EPS;
      sn += w * acosf(fminf(fmaxf(dot, -one_eps), one_eps));                                              // :45-50
      sw += w;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sd += __shfl_down(sd, off, 64);
    sn += __shfl_down(sn, off, 64);
    sw += __shfl_down(sw, off, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wave] = sd;
    red[1][wave] = sn;
    red[2][wave] = sw;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, b = 0.0f, c = 0.0f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
      a += red[0][i];
      b += red[1][i];
      c += red[2][i];
    }
    if (out_disp) *out_disp = a / (float)B_valid;
    if (out_normal) *out_normal = b / c * (180.0f / 3.14159265358979323846f);
  }
}

extern "C" int mnr_render_metrics(int64_t B_valid, const float* distance_mean, const float* disps, const float* acc,
                                  const float* alphas, const float* normals, const float* normals_gt, float* out_disp,
                                  float* out_normal, void* stream) {
  MNR_CHECK_ARG(B_valid > 0 && (out_disp || out_normal), "mnr_render_metrics: nothing to compute");
  MNR_CHECK_ARG(!out_disp || (distance_mean && disps), "mnr_render_metrics: disparity metric needs distance_mean and disps");
  MNR_CHECK_ARG(!out_normal || (acc && alphas && normals && normals_gt),
                "mnr_render_metrics: normal metric needs acc, alphas, normals and normals_gt");
  hipLaunchKernelGGL(render_metrics_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, B_valid, distance_mean, disps,
                     acc, alphas, normals, normals_gt, out_disp, out_normal);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

__global__ void data_loss_kernel(int loss_type, float charb_padding, float loss_mult, int64_t B, int64_t B_valid,
                                 const float* __restrict__ rgb, const float* __restrict__ gt,
                                 const float* __restrict__ lm, int lm_c, const float* __restrict__ denom_p,
                                 float* stats, float* __restrict__ g_rgb) {
  const float denom = *denom_p;
  float s_mse = 0.0f, s_loss = 0.0f;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float g = 0.0f;
      if (b < B_valid) {
        const float y = rgb[b * 3 + ch], t = gt[b * 3 + ch];
        const float w = lm_c == 1 ? lm[b] : lm[b * 3 + ch];
        const float r = y - t;
        s_mse += w * r * r;                                   // train_utils.py:86-88
        float dl, dd;
        if (loss_type == MNR_LOSS_MSE) {                      // :90-92
          dl = r * r;
          dd = 2.0f * r;
        } else if (loss_type == MNR_LOSS_CHARB) {             // :93-95
          dl = sqrtf(r * r + charb_padding * charb_padding);
          dd = r / dl;
        } else {                                              // :96-103 rawnerf
          const float yc = fminf(1.0f, y);
          const float rc = yc - t;
          const float sg = 1.0f / (1e-3f + yc);
          dl = rc * rc * sg * sg;
          dd = y < 1.0f ? 2.0f * rc * sg * sg : 0.0f;
        }
        s_loss += w * dl;
        g = loss_mult * w * dd / denom;
      }
      if (g_rgb) g_rgb[b * 3 + ch] = g;
    }
  }
  s_mse = ls_wave_sum(s_mse);
  s_loss = ls_wave_sum(s_loss);
  if ((threadIdx.x & 63) == 0) {
    unsafeAtomicAdd(stats + 0, s_mse / denom);
    unsafeAtomicAdd(stats + 1, loss_mult * s_loss / denom);
  }
}

extern "C" int mnr_data_loss(int loss_type, float charb_padding, float loss_mult, int64_t B, int64_t B_valid,
                             const float* rgb, const float* gt, const float* lossmult, int lm_c,
                             const float* denom, float* stats, float* g_rgb, void* stream) {
  MNR_CHECK_ARG(B > 0 && B_valid > 0 && B_valid <= B && rgb && gt && lossmult && denom && stats,
                "mnr_data_loss: bad arguments");
  MNR_CHECK_ARG(loss_type >= MNR_LOSS_MSE && loss_type <= MNR_LOSS_RAWNERF && (lm_c == 1 || lm_c == 3),
                "mnr_data_loss: unsupported data_loss_type / lossmult channels");
  int grid = mnr_cdiv(B, 256);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(data_loss_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, loss_type, charb_padding,
                     loss_mult, B, B_valid, rgb, gt, lossmult, lm_c, denom, stats, g_rgb);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// interlevel loss.  Per ray, in LDS [elem][ray]: lo[n+1], hi[n+1] (as floats), g[n].

__global__ __launch_bounds__(LS_THREADS) void interlevel_kernel(float mult, int64_t B, int64_t B_valid, int n,
                                                                const float* __restrict__ t,
                                                                const float* __restrict__ w, int ne,
                                                                const float* __restrict__ te,
                                                                const float* __restrict__ we, float* stats,
                                                                float* __restrict__ g_we, float* __restrict__ per_elem) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int S = LS_THREADS;
  float* lo_f = lds + threadIdx.x;
  float* hi_f = lo_f + (n + 1) * S;
  float* gi = hi_f + (n + 1) * S;
  const int64_t b = (int64_t)blockIdx.x * S + threadIdx.x;
  float loss_sum = 0.0f;
  if (b < B_valid) {
    const float scale = mult / ((float)B_valid * (float)n);   // jnp.mean over [B, n]
    ls_outer_sweep(n, t + b * (n + 1), w + b * n, ne, te + b * (ne + 1), we + b * ne, 1, lo_f, hi_f, gi, S, scale,
                   per_elem == nullptr, loss_sum);
    if (per_elem) {
      for (int i = 0; i < n; ++i) per_elem[b * n + i] = gi[i * S];
    } else if (g_we) {
      // d w_outer[i] / d we[j] = [lo[i] <= j < hi[i+1]]; starts and ends both ascend with i.
      int is = 0, ie = 0;
      float active = 0.0f;
      for (int j = 0; j < ne; ++j) {
        while (is < n && (int)lo_f[is * S] <= j) active += gi[(is++) * S];
        while (ie < n && (int)hi_f[(ie + 1) * S] <= j) active -= gi[(ie++) * S];
        g_we[b * ne + j] += active;
      }
    }
  }
  if (stats) {
    loss_sum = ls_wave_sum(loss_sum);
    if (threadIdx.x == 0) unsafeAtomicAdd(stats, mult * loss_sum / ((float)B_valid * (float)n));
  }
}

static int ls_interlevel_launch(float mult, int64_t B, int64_t B_valid, int n, const float* t, const float* w, int ne,
                                const float* te, const float* we, float* stats, float* g_we, float* per_elem,
                                void* stream) {
  MNR_CHECK_ARG(B > 0 && B_valid > 0 && B_valid <= B && n >= 1 && ne >= 1 && t && w && te && we,
                "mnr_interlevel_loss: bad arguments");
  const size_t lds = (size_t)(3 * n + 2) * LS_THREADS * 4;
  MNR_CHECK_ARG(lds <= 160 * 1024, "mnr_interlevel_loss: n too large");
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if (mnr_attr_needed(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)interlevel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL(interlevel_kernel, dim3(mnr_cdiv(B_valid, LS_THREADS)), dim3(LS_THREADS), lds,
                     (hipStream_t)stream, mult, B, B_valid, n, t, w, ne, te, we, stats, g_we, per_elem);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_interlevel_loss(float mult, int64_t B, int64_t B_valid, int n, const float* t, const float* w,
                                   int ne, const float* t_env, const float* w_env, float* stats, float* g_w_env,
                                   void* stream) {
  return ls_interlevel_launch(mult, B, B_valid, n, t, w, ne, t_env, w_env, stats, g_w_env, nullptr, stream);
}

extern "C" int mnr_lossfun_outer(int64_t B, int n, const float* t, const float* w, int ne, const float* t_env,
                                 const float* w_env, float* out, void* stream) {
  MNR_CHECK_ARG(out, "mnr_lossfun_outer: null output");
  return ls_interlevel_launch(1.0f, B, B, n, t, w, ne, t_env, w_env, nullptr, nullptr, out, stream);
}

// ---------------------------------------------------------------------------
// distortion loss.  LDS [elem][ray]: ut[n], w[n].

__global__ __launch_bounds__(LS_THREADS) void distortion_kernel(float mult, int64_t B, int64_t B_valid, int n,
                                                                const float* __restrict__ t,
                                                                const float* __restrict__ w, float* stats,
                                                                float* __restrict__ g_w, float* __restrict__ per_ray) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int S = LS_THREADS;
  float* ut = lds + threadIdx.x;
  float* ww = ut + n * S;
  const int64_t b = (int64_t)blockIdx.x * S + threadIdx.x;
  float loss = 0.0f;
  if (b < B_valid) {
    const float* tb = t + b * (n + 1);
    const float* wb = w + b * n;
    for (int i = 0; i < n; ++i) {
      ut[i * S] = (tb[i + 1] + tb[i]) / 2.0f;                 // stepfun.py:269
      ww[i * S] = wb[i];
    }
    const float scale = mult / (float)B_valid;                // jnp.mean over rays
    for (int i = 0; i < n; ++i) {
      const float ui = ut[i * S], wi = ww[i * S];
      float inner = 0.0f;
      for (int j = 0; j < n; ++j) inner += ww[j * S] * fabsf(ui - ut[j * S]);   // :270-271
      const float dt = tb[i + 1] - tb[i];
      loss += wi * inner + wi * wi * dt / 3.0f;               // :271, :274
      if (g_w) g_w[b * n + i] += scale * (2.0f * inner + (2.0f / 3.0f) * wi * dt);
    }
    if (per_ray) per_ray[b] = loss;
  }
  if (stats) {
    loss = ls_wave_sum(loss);
    if (threadIdx.x == 0) unsafeAtomicAdd(stats, mult * loss / (float)B_valid);
  }
}

static int ls_distortion_launch(float mult, int64_t B, int64_t B_valid, int n, const float* t, const float* w,
                                float* stats, float* g_w, float* per_ray, void* stream) {
  MNR_CHECK_ARG(B > 0 && B_valid > 0 && B_valid <= B && n >= 1 && t && w, "mnr_distortion_loss: bad arguments");
  const size_t lds = (size_t)(2 * n) * LS_THREADS * 4;
  MNR_CHECK_ARG(lds <= 160 * 1024, "mnr_distortion_loss: n too large");
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if (mnr_attr_needed(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)distortion_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL(distortion_kernel, dim3(mnr_cdiv(B_valid, LS_THREADS)), dim3(LS_THREADS), lds,
                     (hipStream_t)stream, mult, B, B_valid, n, t, w, stats, g_w, per_ray);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_distortion_loss(float mult, int64_t B, int64_t B_valid, int n, const float* t, const float* w,
                                   float* stats, float* g_w, void* stream) {
  return ls_distortion_launch(mult, B, B_valid, n, t, w, stats, g_w, nullptr, stream);
}

extern "C" int mnr_lossfun_distortion(int64_t B, int n, const float* t, const float* w, float* out, void* stream) {
  MNR_CHECK_ARG(out, "mnr_lossfun_distortion: null output");
  return ls_distortion_launch(1.0f, B, B, n, t, w, nullptr, nullptr, out, stream);
}
