// Alpha compositing along rays, forward and VJP (gfx950).
//
// Replaces the head activations (models.py:506, 584-602), RawNeRF exposure
// scaling (models.py:257-267), render.compute_alpha_weights (render.py:130-151)
// and render.volumetric_rendering (render.py:154-213).  The reference runs these
// as whole-batch tensor ops with an exclusive cumsum; here one lane owns one ray
// and the scan stays in registers: a forward running sum of density*delta and, in
// the VJP, one reverse running sum (formulas: SURVEY.md Appendix A "Backward").
// A workgroup's rows are staged through LDS so that global loads/stores are
// coalesced although each lane walks its own row.
#include "common.h"
#include "ray_losses.h"

#define CP_THREADS 64

__device__ __forceinline__ float cp_act(int kind, float x) {
  switch (kind) {
    case MNR_ACT_SIGMOID: return mnr_sigmoid(x);
    case MNR_ACT_SAFE_EXP: return expf(fminf(x, 88.0f));     // math.py:41-44
    case MNR_ACT_SOFTPLUS: return mnr_softplus(x);
    case MNR_ACT_EXP: return expf(x);
    case MNR_ACT_RELU: return fmaxf(x, 0.0f);
    default: return x;
  }
}

// d act(x) / dx given x and y = act(x).
__device__ __forceinline__ float cp_act_grad(int kind, float x, float y) {
  switch (kind) {
    case MNR_ACT_SIGMOID: return y * (1.0f - y);
    case MNR_ACT_SAFE_EXP: return y;                          // math.py:47-54 custom JVP
    case MNR_ACT_SOFTPLUS: return mnr_sigmoid(x);
    case MNR_ACT_EXP: return y;
    case MNR_ACT_RELU: return x > 0.0f ? 1.0f : 0.0f;
    default: return 1.0f;
  }
}

// Stage `rows` rows of `len` floats (contiguous in HBM from `src`) into LDS as [elem][ray].
__device__ __forceinline__ void cp_load_rows(float* lds, const float* src, int rows, int len, int stride) {
  for (int e = threadIdx.x; e < rows * len; e += CP_THREADS) {
    const int r = e / len, i = e % len;
    lds[i * stride + r] = src[e];
  }
}

__device__ __forceinline__ void cp_store_rows(const float* lds, float* dst, int rows, int len, int stride) {
  for (int e = threadIdx.x; e < rows * len; e += CP_THREADS) {
    const int r = e / len, i = e % len;
    dst[e] = lds[i * stride + r];
  }
}

// LDS plan (floats per ray): raw density n | tdist n+1 | raw rgb 3n | weights n (out) | rgb 3n (out)
__global__ __launch_bounds__(CP_THREADS) void composite_fwd_kernel(
    mnr_composite_cfg c, int64_t B, int S, const float* __restrict__ raw_density, const float* __restrict__ noise,
    const float* __restrict__ raw_rgb, const float* __restrict__ tdist, const float* __restrict__ dirs,
    const float* __restrict__ bg, const float* __restrict__ expo, float* __restrict__ density,
    float* __restrict__ rgb, float* __restrict__ weights, float* __restrict__ rgb_out, float* __restrict__ acc_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = c.n;
  float* l_den = lds;                       // n  (raw in, activated out)
  float* l_t = l_den + n * S;               // n+1
  float* l_w = l_t + (n + 1) * S;           // n
  float* l_rgb = l_w + n * S;               // 3n (raw in, activated out)
  const int64_t ray0 = (int64_t)blockIdx.x * S;
  const int rows = (int)min((int64_t)S, B - ray0);
  cp_load_rows(l_den, raw_density + ray0 * n, rows, n, S);
  cp_load_rows(l_t, tdist + ray0 * (n + 1), rows, n + 1, S);
  if (c.has_rgb) cp_load_rows(l_rgb, raw_rgb + ray0 * n * 3, rows, 3 * n, S);
  if (noise && c.density_noise_std > 0.0f) {
    __syncthreads();
    for (int e = threadIdx.x; e < rows * n; e += CP_THREADS) {
      const int r = e / n, i = e % n;
      l_den[i * S + r] += c.density_noise_std * noise[ray0 * n + e];     // models.py:462-464
    }
  }
  __syncthreads();
  const int r = threadIdx.x;
  if (r < rows) {
    const int64_t ray = ray0 + r;
    const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    float ex[3] = {1.0f, 1.0f, 1.0f};
    if (expo) { ex[0] = expo[ray * 3]; ex[1] = expo[ray * 3 + 1]; ex[2] = expo[ray * 3 + 2]; }
    float run = 0.0f, acc = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
    for (int i = 0; i < n; ++i) {
      const float sigma = cp_act(c.density_act, l_den[i * S + r] + c.density_bias);   // models.py:506
      l_den[i * S + r] = sigma;
      const float delta = (l_t[(i + 1) * S + r] - l_t[i * S + r]) * dnorm;           // render.py:132-133
      float x = sigma * delta;
      if (c.opaque_background && i == n - 1) x = INFINITY;                            // render.py:136-142
      const float alpha = 1.0f - expf(-x);
      const float trans = expf(-run);
      const float w = alpha * trans;
      run += x;
      l_w[i * S + r] = w;
      acc += w;
      if (c.has_rgb) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          // models.py:584-586, 602, then exposure scaling :257-267.
          float v = cp_act(c.rgb_act, c.rgb_premultiplier * l_rgb[(3 * i + ch) * S + r] + c.rgb_bias);
          v = v * (1.0f + 2.0f * c.rgb_padding) - c.rgb_padding;
          v *= ex[ch];
          l_rgb[(3 * i + ch) * S + r] = v;
          if (ch == 0) cr += w * v; else if (ch == 1) cg += w * v; else cb += w * v;
        }
      }
    }
    const float bgw = fmaxf(0.0f, 1.0f - acc);                                         // render.py:180
    float b0 = c.bg_value, b1 = c.bg_value, b2 = c.bg_value;
    if (c.bg_mode == 1) { b0 = bg[ray * 3]; b1 = bg[ray * 3 + 1]; b2 = bg[ray * 3 + 2]; }
    rgb_out[ray * 3 + 0] = cr + bgw * b0;                                              // render.py:181
    rgb_out[ray * 3 + 1] = cg + bgw * b1;
    rgb_out[ray * 3 + 2] = cb + bgw * b2;
    if (acc_out) acc_out[ray] = acc;
  }
  __syncthreads();
  cp_store_rows(l_den, density + ray0 * n, rows, n, S);
  cp_store_rows(l_w, weights + ray0 * n, rows, n, S);
  if (c.has_rgb && rgb) cp_store_rows(l_rgb, rgb + ray0 * n * 3, rows, 3 * n, S);
}

// Rays per block: 64 if the staging fits in LDS, else halved until it does.
static int cp_rays_per_block(int n, int64_t B) {
  int s = CP_THREADS;
  while (s > 1 && (size_t)(6 * n + 1) * s * 4 > 150 * 1024) s >>= 1;
  // one wave per SIMD of the chip (common.h: mnr_ray_wave_target)
  while (s > 8 && B / s < mnr_ray_wave_target()) s >>= 1;
  return s;
}
static size_t cp_lds_bytes(int n, int s) { return (size_t)(6 * n + 1) * s * 4; }

// A/B switch of the per-ray kernels: 1 = four lanes per ray (composite_fwd_quad_kernel, level_bwd_quad_kernel) where a
// wave's 16 rays fit LDS, 0 = lane per ray for every shape
static int g_level_bwd_quad = 1;
// Largest LDS footprint (bytes per single-wave workgroup) at which the four-lanes-per-ray kernels are used.  Up to 40 KiB four
// workgroups (one per SIMD) fit a CU; rays of 128 samples need 49 KiB (forward) / 66 KiB (backward), i.e. three / two
// workgroups per CU, and are still faster there than on the lane-per-ray kernels: round-3 same-box A/B with the limit at
// 40 / 52 / 80 KiB: blender_256 1.670 / 1.658 / 1.737 M rays/s, llff_raw 488 / 496 / 495 k (profiles/r3_ab.md).  The levels of
// llff_raw (128 samples WITH rgb, RawNeRF loss, exposure scaling) need 82 / 99 KiB, one workgroup per CU, and are the ones the
// lane-per-ray stream hurts most: with the limit at 80 / 88 / 104 / 160 KiB llff_raw runs 512 / 545 / 545 / 544 k rays/s,
// blender_refnerf 150.8 / - / 152.6 / - k, blender_256 1.771 / - / 1.765 / - M (three pairs, inside the noise): 104 KiB.
static size_t quad_lds_max() { return 104 * 1024; }
extern "C" int mnr_level_bwd_set_quad(int on) {
  g_level_bwd_quad = on;
  return MNR_OK;
}

__global__ void composite_fwd_quad_kernel(mnr_composite_cfg c, int64_t B, int stride, const float* __restrict__ raw_density,
                                          const float* __restrict__ noise, const float* __restrict__ raw_rgb,
                                          const float* __restrict__ tdist, const float* __restrict__ dirs,
                                          const float* __restrict__ bg, const float* __restrict__ expo,
                                          float* __restrict__ density, float* __restrict__ rgb, float* __restrict__ weights,
                                          float* __restrict__ rgb_out, float* __restrict__ acc_out);

extern "C" int mnr_composite_fwd(const mnr_composite_cfg* cfg, int64_t B, const float* raw_density,
                                 const float* density_noise, const float* raw_rgb, const float* tdist,
                                 const float* dirs, const float* bg, const float* exposure_scale, float* density,
                                 float* rgb, float* weights, float* rgb_out, float* acc, void* stream) {
  MNR_CHECK_ARG(cfg && B > 0 && raw_density && tdist && dirs && density && weights && rgb_out,
                "mnr_composite_fwd: null argument");
  MNR_CHECK_ARG(cfg->n >= 1 && cfg->n <= 1024, "mnr_composite_fwd: n out of range");
  MNR_CHECK_ARG(!cfg->has_rgb || raw_rgb, "mnr_composite_fwd: has_rgb needs raw_rgb");
  MNR_CHECK_ARG(cfg->bg_mode == 0 || bg, "mnr_composite_fwd: bg_mode 1 needs bg");
  {
    // four lanes per ray: den n | t n+1 | w n | [rgb 3n], ray stride 4 x odd floats (conflict-free quads)
    int o = 3 * cfg->n + 1 + (cfg->has_rgb ? 3 * cfg->n : 0);
    o = (o + 3) & ~3;
    if (((o >> 2) & 1) == 0) o += 4;
    const size_t quad_lds = (size_t)o * (CP_THREADS / 4) * 4;
    if (g_level_bwd_quad && quad_lds <= quad_lds_max()) {
      static unsigned long long attr_q = 0;
      if (quad_lds > 64 * 1024 && mnr_attr_needed(&attr_q))
        (void)hipFuncSetAttribute((const void*)composite_fwd_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(composite_fwd_quad_kernel, dim3(mnr_cdiv(B, CP_THREADS / 4)), dim3(CP_THREADS), quad_lds,
                         (hipStream_t)stream, *cfg, B, o, raw_density, density_noise, raw_rgb, tdist, dirs, bg, exposure_scale,
                         density, rgb, weights, rgb_out, acc);
      MNR_CHECK_LAUNCH();
      return MNR_OK;
    }
  }
  const int S = cp_rays_per_block(cfg->n, B);
  const size_t lds = cp_lds_bytes(cfg->n, S);
  MNR_CHECK_ARG(lds <= 160 * 1024, "mnr_composite_fwd: n=%d too long for LDS staging", cfg->n);
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if (mnr_attr_needed(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)composite_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL(composite_fwd_kernel, dim3(mnr_cdiv(B, S)), dim3(CP_THREADS), lds, (hipStream_t)stream,
                     *cfg, B, S, raw_density, density_noise, raw_rgb, tdist, dirs, bg, exposure_scale, density, rgb,
                     weights, rgb_out, acc);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// One level's backward pass up to the MLP heads: the training losses that act on this level's rendering and their
// gradients (train_utils.py:72-159: data loss on the composited colour; interlevel loss of a proposal level, whose
// histogram is the envelope of the final level's; distortion loss of the final level), then the compositing VJP.
// One launch per level instead of data_loss + interlevel / distortion + composite_bwd: d loss / d weights lives in LDS
// and d loss / d rgb in registers, never in HBM.
//
// Compositing VJP.  With x_i = sigma_i * delta_i, T_i = exp(-sum_{k<i} x_k), w_i = (1 - exp(-x_i)) T_i:
//   g^_i = g_w[i] + g_rgb . c_i - [acc < 1] (g_rgb . bg)
//   dL/dx_i = g^_i (T_i - w_i) - sum_{k>i} g^_k w_k ;  dL/dsigma_i = delta_i dL/dx_i ; dL/dc_i = w_i g_rgb.
// LDS per ray (floats): raw density n | tdist n+1 | weights n | raw rgb 3n | g_w n | [sdist n+1 | lo n_ref+1 | hi n_ref+1 | gi n_ref]
__global__ __launch_bounds__(CP_THREADS) void level_bwd_kernel(mnr_level_bwd_args a, int S) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const mnr_composite_cfg c = a.cfg;
  const int n = c.n;
  float* l_den = lds;                       // n   raw density in, g_raw_density out
  float* l_t = l_den + n * S;               // n+1
  float* l_w = l_t + (n + 1) * S;           // n   weights
  float* l_rgb = l_w + n * S;               // 3n  raw rgb in, g_raw_rgb out
  float* l_gw = l_rgb + 3 * n * S;          // n   d loss / d weights
  float* l_s = l_gw + n * S;                // n+1 sdist (weight losses only)
  float* l_lo = l_s + (n + 1) * S;          // n_ref+1, n_ref+1, n_ref (interlevel only)
  float* l_hi = l_lo + (a.n_ref + 1) * S;
  float* l_gi = l_hi + (a.n_ref + 1) * S;
  const int64_t B = a.B;
  const int64_t ray0 = (int64_t)blockIdx.x * S;
  const int rows = (int)min((int64_t)S, B - ray0);
  cp_load_rows(l_den, a.raw_density + ray0 * n, rows, n, S);
  cp_load_rows(l_t, a.tdist + ray0 * (n + 1), rows, n + 1, S);
  cp_load_rows(l_w, a.weights + ray0 * n, rows, n, S);
  if (c.has_rgb) cp_load_rows(l_rgb, a.raw_rgb + ray0 * n * 3, rows, 3 * n, S);
  if (a.g_weights) {
    cp_load_rows(l_gw, a.g_weights + ray0 * n, rows, n, S);
  } else {
    for (int e = threadIdx.x; e < S * n; e += CP_THREADS) l_gw[e] = 0.0f;
  }
  if (a.wloss_mode != 0) cp_load_rows(l_s, a.sdist + ray0 * (n + 1), rows, n + 1, S);
  if (a.density_noise && c.density_noise_std > 0.0f) {
    __syncthreads();
    for (int e = threadIdx.x; e < rows * n; e += CP_THREADS) {
      const int r = e / n, i = e % n;
      l_den[i * S + r] += c.density_noise_std * a.density_noise[ray0 * n + e];
    }
  }
  __syncthreads();
  const int r = threadIdx.x;
  float s_mse = 0.0f, s_dloss = 0.0f, s_wloss = 0.0f;
  if (r < rows) {
    const int64_t ray = ray0 + r;
    const bool valid = ray < a.B_valid;      // padding rays take part in no loss
    float go[3] = {0.0f, 0.0f, 0.0f};
    if (a.g_rgb_out) { go[0] = a.g_rgb_out[ray * 3]; go[1] = a.g_rgb_out[ray * 3 + 1]; go[2] = a.g_rgb_out[ray * 3 + 2]; }
    // ---- data loss on the composited colour (train_utils.py:85-111)
    if (a.data_loss_type >= 0 && valid) {
      const float denom = *a.denom;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float y = a.rgb_out[ray * 3 + ch], t = a.gt[ray * 3 + ch];
        const float w = a.lm_c == 1 ? a.lossmult[ray] : a.lossmult[ray * 3 + ch];
        const float rs = y - t;
        s_mse += w * rs * rs;                                 // train_utils.py:86-88
        float dl, dd;
        if (a.data_loss_type == MNR_LOSS_MSE) {               // :90-92
          dl = rs * rs;
          dd = 2.0f * rs;
        } else if (a.data_loss_type == MNR_LOSS_CHARB) {      // :93-95
          dl = sqrtf(rs * rs + a.charb_padding * a.charb_padding);
          dd = rs / dl;
        } else {                                              // :96-103 rawnerf
          const float yc = fminf(1.0f, y);
          const float rc = yc - t;
          const float sg = 1.0f / (1e-3f + yc);
          dl = rc * rc * sg * sg;
          dd = y < 1.0f ? 2.0f * rc * sg * sg : 0.0f;
        }
        s_dloss += w * dl;
        go[ch] += a.data_loss_mult * w * dd / denom;
      }
    }
    // ---- losses on the weights: d loss / d w_i accumulates in l_gw
    if (a.wloss_mode == 1 && valid) {
      // interlevel (train_utils.py:139-150, stepfun.py:64-86): this level (l_s, l_w) is the envelope of the final
      // level's histogram (t_ref, w_ref), whose values carry no gradient
      const int nr = a.n_ref;
      const float scale = a.wloss_mult / ((float)a.B_valid * (float)nr);       // jnp.mean over [B, n_ref]
      ls_outer_sweep(nr, a.t_ref + ray * (nr + 1), a.w_ref + ray * nr, n, l_s + r, l_w + r, S, l_lo + r, l_hi + r, l_gi + r, S,
                     scale, true, s_wloss);
      // d w_outer[i] / d we[j] = [lo[i] <= j < hi[i+1]]; starts and ends both ascend with i.
      int is = 0, ie = 0;
      float active = 0.0f;
      for (int j = 0; j < n; ++j) {
        while (is < nr && (int)l_lo[is * S + r] <= j) active += l_gi[(is++) * S + r];
        while (ie < nr && (int)l_hi[(ie + 1) * S + r] <= j) active -= l_gi[(ie++) * S + r];
        l_gw[j * S + r] += active;
      }
    } else if (a.wloss_mode == 2 && valid) {
      // distortion (train_utils.py:153-159, stepfun.py:266-276) on this level's own (sdist, weights)
      const float scale = a.wloss_mult / (float)a.B_valid;                     // jnp.mean over rays
      for (int i = 0; i < n; ++i) {
        const float ui = (l_s[(i + 1) * S + r] + l_s[i * S + r]) / 2.0f, wi = l_w[i * S + r];
        float inner = 0.0f;
        for (int j = 0; j < n; ++j) inner += l_w[j * S + r] * fabsf(ui - (l_s[(j + 1) * S + r] + l_s[j * S + r]) / 2.0f);
        const float dt = l_s[(i + 1) * S + r] - l_s[i * S + r];
        s_wloss += wi * inner + wi * wi * dt / 3.0f;
        l_gw[i * S + r] += scale * (2.0f * inner + (2.0f / 3.0f) * wi * dt);
      }
    }
    // ---- compositing VJP
    const float dx = a.dirs[ray * 3], dy = a.dirs[ray * 3 + 1], dz = a.dirs[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    float ex[3] = {1.0f, 1.0f, 1.0f};
    if (a.exposure_scale) { ex[0] = a.exposure_scale[ray * 3]; ex[1] = a.exposure_scale[ray * 3 + 1]; ex[2] = a.exposure_scale[ray * 3 + 2]; }
    float b[3] = {c.bg_value, c.bg_value, c.bg_value};
    if (c.bg_mode == 1) { b[0] = a.bg[ray * 3]; b[1] = a.bg[ray * 3 + 1]; b[2] = a.bg[ray * 3 + 2]; }
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) acc += l_w[i * S + r];
    const float g_bgw = (1.0f - acc > 0.0f) ? (go[0] * b[0] + go[1] * b[1] + go[2] * b[2]) : 0.0f;
    // Exclusive prefix of x at the last sample, then walk backwards:
    // run_i = sum_{k<i} x_k, T_i = exp(-run_i), T_{i+1} = exp(-(run_i + x_i)).
    float run = 0.0f;
    for (int i = 0; i + 1 < n; ++i) {
      const float sigma = cp_act(c.density_act, l_den[i * S + r] + c.density_bias);
      run += sigma * (l_t[(i + 1) * S + r] - l_t[i * S + r]) * dnorm;
    }
    float suffix = 0.0f;                    // sum_{k>i} g^_k w_k
    float ges[3] = {0.0f, 0.0f, 0.0f};      // d L / d exposure_scale[ray]
    for (int i = n - 1; i >= 0; --i) {
      const float w = l_w[i * S + r];
      const float raw = l_den[i * S + r] + c.density_bias;
      const float sigma = cp_act(c.density_act, raw);
      const float delta = (l_t[(i + 1) * S + r] - l_t[i * S + r]) * dnorm;
      const bool opaque_last = c.opaque_background && i == n - 1;
      const float x = opaque_last ? INFINITY : sigma * delta;
      const float t_next = expf(-(run + x));              // T_{i+1} = T_i - w_i
      float ghat = l_gw[i * S + r] - g_bgw;
      float gc[3] = {0.0f, 0.0f, 0.0f};
      if (c.has_rgb) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float z = c.rgb_premultiplier * l_rgb[(3 * i + ch) * S + r] + c.rgb_bias;
          const float y = cp_act(c.rgb_act, z);
          const float col_pre = y * (1.0f + 2.0f * c.rgb_padding) - c.rgb_padding;
          const float col = col_pre * ex[ch];
          ghat += go[ch] * col;
          ges[ch] += w * go[ch] * col_pre;
          // d col / d raw = ex * (1+2pad) * act'(z) * premult
          gc[ch] = w * go[ch] * ex[ch] * (1.0f + 2.0f * c.rgb_padding) * cp_act_grad(c.rgb_act, z, y) *
                   c.rgb_premultiplier;
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) l_rgb[(3 * i + ch) * S + r] = gc[ch];
      }
      float gx = ghat * t_next - suffix;
      if (opaque_last) gx = 0.0f;           // x = +inf is a constant
      suffix += ghat * w;
      l_den[i * S + r] = gx * delta * cp_act_grad(c.density_act, raw, sigma);
      l_gw[i * S + r] = gx;                 // d loss / d (sigma_i delta_i), for a.g_x (d loss / d w_i is consumed)
      if (i > 0) {
        const float sp = cp_act(c.density_act, l_den[(i - 1) * S + r] + c.density_bias);
        run -= sp * (l_t[i * S + r] - l_t[(i - 1) * S + r]) * dnorm;
      }
    }
    if (a.g_exposure_scale) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) a.g_exposure_scale[ray * 3 + ch] += ges[ch];
    }
  }
  // loss values: wave sums, one atomic each
  if (a.data_loss_type >= 0 && a.data_stats) {
    s_mse = ls_wave_sum(s_mse);
    s_dloss = ls_wave_sum(s_dloss);
    if (threadIdx.x == 0) {
      const float denom = *a.denom;
      unsafeAtomicAdd(a.data_stats + 0, s_mse / denom);
      unsafeAtomicAdd(a.data_stats + 1, a.data_loss_mult * s_dloss / denom);
    }
  }
  if (a.wloss_mode != 0 && a.wloss_stat) {
    s_wloss = ls_wave_sum(s_wloss);
    const float norm = a.wloss_mode == 1 ? (float)a.B_valid * (float)a.n_ref : (float)a.B_valid;
    if (threadIdx.x == 0) unsafeAtomicAdd(a.wloss_stat, a.wloss_mult * s_wloss / norm);
  }
  __syncthreads();
  if (a.g_raw_density) cp_store_rows(l_den, a.g_raw_density + ray0 * n, rows, n, S);
  if (a.g_raw_density_bf16) {
    bf16* gb = (bf16*)a.g_raw_density_bf16;
    for (int e = threadIdx.x; e < rows * n; e += CP_THREADS) {
      const int rr = e / n, i = e % n;
      gb[(ray0 * n + e) * (int64_t)a.ld_bf16] = (bf16)l_den[i * S + rr];
    }
  }
  if (c.has_rgb && a.g_raw_rgb) cp_store_rows(l_rgb, a.g_raw_rgb + ray0 * n * 3, rows, 3 * n, S);
  if (a.g_x) cp_store_rows(l_gw, a.g_x + ray0 * n, rows, n, S);
}

// ---------------------------------------------------------------------------
// The same backward pass on FOUR lanes per ray (one DPP quad), 16 rays per wave.  The lane-per-ray kernel above is bound by the
// instruction stream of its wave (one wave per SIMD, 16 of 64 lanes active: more lanes per wave were slower, there are
// only 16384 rays for 1024 SIMDs); here lane q of a ray's quad owns the samples i = 4j + q and
//   * prefix / suffix sums (cumulative envelope weights, optical depth, the reverse sum of g^_k w_k) are quad scans (two
//     DPP steps) with a carry over j,
//   * the searchsorted cursors of the interlevel loss become binary searches: with ub(v) = #{idx : te[idx] <= v},
//     lo = max(ub - 1, 0), hi = min(ub, ne), w_outer[i] = CW[hi[i+1]] - CW[lo[i]] (CW = exclusive cumsum of the envelope
//     weights), and d loss / d we[j] = PG[#{i : lo[i] <= j}] - PG[#{i : hi[i+1] <= j}] (PG = exclusive cumsum of the
//     per-interval gradient factors; both index sequences ascend),
//   * the distortion loss splits its outer index over the quad.
// Sums are associated differently from the lane-per-ray kernel (blocked instead of sequential); both are held to the oracle
// by the same tolerances.  LDS per ray, contiguous: den n | t n+1 | w n | [rgb 3n] | gw n | scratch, with a ray stride of
// 4 x odd floats: the 32 lanes of a ds_read group (8 rays x 4 lanes, addresses r * stride + 4j + q) then hit 32 banks.
#define LB_LPR 4
#define LB_RPW (CP_THREADS / LB_LPR)

template <int CTRL>
__device__ __forceinline__ float lb_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lb_quad_sum(float v) {
  v += lb_dpp<0xB1>(v);                      // quad_perm [1,0,3,2]
  v += lb_dpp<0x4E>(v);                      // quad_perm [2,3,0,1]
  return v;
}
// inclusive sum over the quad's lanes 0..q
__device__ __forceinline__ float lb_quad_prefix(float v, int q) {
  const float a = lb_dpp<0x90>(v);           // [0,0,1,2]: lane q-1
  if (q >= 1) v += a;
  const float b = lb_dpp<0x40>(v);           // [0,0,0,1]: lane q-2
  if (q >= 2) v += b;
  return v;
}
// inclusive sum over the quad's lanes q..3
__device__ __forceinline__ float lb_quad_suffix(float v, int q) {
  const float a = lb_dpp<0xF9>(v);           // [1,2,3,3]: lane q+1
  if (q <= 2) v += a;
  const float b = lb_dpp<0xFE>(v);           // [2,3,3,3]: lane q+2
  if (q <= 1) v += b;
  return v;
}

struct LbLay {
  int den, t, w, rgb, gw, scr, stride;       // float offsets inside a ray's LDS block, ray stride
};
__host__ __device__ inline LbLay lb_layout(int n, int has_rgb, int wloss_mode, int n_ref) {
  LbLay L;
  int o = 0;
  L.den = o; o += n;
  L.t = o; o += n + 1;
  L.w = o; o += n;
  L.rgb = o; o += has_rgb ? 3 * n : 0;
  L.gw = o; o += n;
  L.scr = o;
  // scratch: [sdist n+1 | CW n+1 | lo n_ref+1 | hi n_ref+1 | PG n_ref+1] during the losses, then [TN n | DG n | GH n]
  int loss = wloss_mode != 0 ? n + 1 : 0;
  if (wloss_mode == 1) loss += (n + 1) + 3 * (n_ref + 1);
  o += loss > 3 * n ? loss : 3 * n;
  o = (o + 3) & ~3;
  if (((o >> 2) & 1) == 0) o += 4;           // 4 x odd
  L.stride = o;
  return L;
}

__global__ __launch_bounds__(CP_THREADS) void level_bwd_quad_kernel(mnr_level_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const mnr_composite_cfg c = a.cfg;
  const int n = c.n;
  const LbLay L = lb_layout(n, c.has_rgb, a.wloss_mode, a.n_ref);
  const int64_t B = a.B;
  const int64_t ray0 = (int64_t)blockIdx.x * LB_RPW;
  const int rows = (int)min((int64_t)LB_RPW, B - ray0);
  auto load_rows = [&](int off, const float* src, int len) {      // rows of `len` floats, contiguous in HBM
    for (int e = threadIdx.x; e < rows * len; e += CP_THREADS) {
      const int rr = e / len, i = e - rr * len;
      lds[rr * L.stride + off + i] = src[e];
    }
  };
  load_rows(L.den, a.raw_density + ray0 * n, n);
  load_rows(L.t, a.tdist + ray0 * (n + 1), n + 1);
  load_rows(L.w, a.weights + ray0 * n, n);
  if (c.has_rgb) load_rows(L.rgb, a.raw_rgb + ray0 * n * 3, 3 * n);
  if (a.g_weights) {
    load_rows(L.gw, a.g_weights + ray0 * n, n);
  } else {
    for (int e = threadIdx.x; e < LB_RPW * n; e += CP_THREADS) lds[(e / n) * L.stride + L.gw + e % n] = 0.0f;
  }
  if (a.wloss_mode != 0) load_rows(L.scr, a.sdist + ray0 * (n + 1), n + 1);
  if (a.density_noise && c.density_noise_std > 0.0f) {
    __syncthreads();
    for (int e = threadIdx.x; e < rows * n; e += CP_THREADS) {
      const int rr = e / n, i = e - rr * n;
      lds[rr * L.stride + L.den + i] += c.density_noise_std * a.density_noise[ray0 * n + e];
    }
  }
  __syncthreads();
  // Every lane runs every phase (the barriers between phases are workgroup barriers); rays past the batch end compute on
  // whatever their LDS block holds and write nothing.
  const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
  const bool live = r < rows;
  const int64_t ray = live ? ray0 + r : B - 1;
  const bool valid = live && ray < a.B_valid;      // padding rays take part in no loss
  float* X = lds + r * L.stride;
  float* l_den = X + L.den;
  float* l_t = X + L.t;
  float* l_w = X + L.w;
  float* l_rgb = X + L.rgb;
  float* l_gw = X + L.gw;
  float* l_s = X + L.scr;
  const int J = (n + 3) >> 2;                      // samples per lane
  float s_mse = 0.0f, s_dloss = 0.0f, s_wloss = 0.0f;
  float go[3] = {0.0f, 0.0f, 0.0f};
  if (a.g_rgb_out) { go[0] = a.g_rgb_out[ray * 3]; go[1] = a.g_rgb_out[ray * 3 + 1]; go[2] = a.g_rgb_out[ray * 3 + 2]; }
  // ---- data loss on the composited colour (train_utils.py:85-111): every lane of the quad needs go[], lane 0 keeps the sums
  if (a.data_loss_type >= 0 && valid) {
    const float denom = *a.denom;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float y = a.rgb_out[ray * 3 + ch], t = a.gt[ray * 3 + ch];
      const float w = a.lm_c == 1 ? a.lossmult[ray] : a.lossmult[ray * 3 + ch];
      const float rs = y - t;
      float dl, dd;
      if (a.data_loss_type == MNR_LOSS_MSE) {               // :90-92
        dl = rs * rs;
        dd = 2.0f * rs;
      } else if (a.data_loss_type == MNR_LOSS_CHARB) {      // :93-95
        dl = sqrtf(rs * rs + a.charb_padding * a.charb_padding);
        dd = rs / dl;
      } else {                                              // :96-103 rawnerf
        const float yc = fminf(1.0f, y);
        const float rc = yc - t;
        const float sg = 1.0f / (1e-3f + yc);
        dl = rc * rc * sg * sg;
        dd = y < 1.0f ? 2.0f * rc * sg * sg : 0.0f;
      }
      if (q == 0) {
        s_mse += w * rs * rs;                               // train_utils.py:86-88
        s_dloss += w * dl;
      }
      go[ch] += a.data_loss_mult * w * dd / denom;
    }
  }
  // ---- losses on the weights: d loss / d w_i accumulates in l_gw
  if (a.wloss_mode == 1) {
    // interlevel (train_utils.py:139-150, stepfun.py:64-86): this level (l_s, l_w) is the envelope of the final level's
    // histogram (t_ref, w_ref), whose values carry no gradient
    const int nr = a.n_ref;
    const float scale = a.wloss_mult / ((float)a.B_valid * (float)nr);       // jnp.mean over [B, n_ref]
    const float* tr = a.t_ref + ray * (nr + 1);
    const float* wr = a.w_ref + ray * nr;
    float* l_cw = l_s + (n + 1);                   // CW[idx] = sum_{j<idx} we[j], idx = 0..n
    int* l_lo = (int*)(l_cw + (n + 1));            // per fence-post k of t_ref
    int* l_hi = l_lo + (nr + 1);
    float* l_pg = (float*)(l_hi + (nr + 1));       // gradient factors, then their exclusive cumsum (nr + 1)
    {
      float carry = 0.0f;
      if (q == 0) l_cw[0] = 0.0f;
      for (int j = 0; j < J; ++j) {
        const int i = 4 * j + q;
        const float inc = lb_quad_prefix(i < n ? l_w[i] : 0.0f, q) + carry;
        if (i < n) l_cw[i + 1] = inc;
        carry = lb_dpp<0xFF>(inc);
      }
    }
    // searchsorted of the fence-posts (stepfun.py:49-53): ub = #{idx in [0, n] : te[idx] <= v}
    for (int k = q; k <= nr; k += LB_LPR) {
      const float v = tr[k];
      int lo_b = 0, hi_b = n + 1;                  // ub in [lo_b, hi_b]
      while (lo_b < hi_b) {
        const int mid = (lo_b + hi_b) >> 1;
        if (l_s[mid] <= v) lo_b = mid + 1; else hi_b = mid;
      }
      l_lo[k] = max(lo_b - 1, 0);
      l_hi[k] = min(lo_b, n);
    }
    __syncthreads();
    const float eps = MNR_F32_EPS;
    for (int i = q; i < nr; i += LB_LPR) {
      const float w_outer = l_cw[l_hi[i + 1]] - l_cw[l_lo[i]];             // stepfun.py:74
      const float wi = wr[i];
      const float d = fmaxf(0.0f, wi - w_outer);
      if (valid) s_wloss += d * d / (wi + eps);                             // stepfun.py:86
      l_pg[i + 1] = (-2.0f * d / (wi + eps)) * scale;
    }
    __syncthreads();
    {
      // exclusive cumsum in place: PG[0] = 0, PG[i+1] = sum_{i' <= i} g[i']
      float carry = 0.0f;
      if (q == 0) l_pg[0] = 0.0f;
      const int JR = (nr + 3) >> 2;
      for (int j = 0; j < JR; ++j) {
        const int i = 4 * j + q;
        const float inc = lb_quad_prefix(i < nr ? l_pg[i + 1] : 0.0f, q) + carry;
        if (i < nr) l_pg[i + 1] = inc;
        carry = lb_dpp<0xFF>(inc);
      }
    }
    __syncthreads();
    if (valid) {
      // d w_outer[i] / d we[j] = [lo[i] <= j < hi[i+1]]; starts and ends both ascend with i
      for (int j = q; j < n; j += LB_LPR) {
        int lb = 0, hb = nr;                       // #{i < nr : lo[i] <= j}
        while (lb < hb) {
          const int mid = (lb + hb) >> 1;
          if (l_lo[mid] <= j) lb = mid + 1; else hb = mid;
        }
        int le = 0, he = nr;                       // #{i < nr : hi[i+1] <= j}
        while (le < he) {
          const int mid = (le + he) >> 1;
          if (l_hi[mid + 1] <= j) le = mid + 1; else he = mid;
        }
        l_gw[j] += l_pg[lb] - l_pg[le];
      }
    }
  } else if (a.wloss_mode == 2) {
    // distortion (train_utils.py:153-159, stepfun.py:266-276) on this level's own (sdist, weights)
    const float scale = a.wloss_mult / (float)a.B_valid;                     // jnp.mean over rays
    if (valid) {
      for (int i = q; i < n; i += LB_LPR) {
        const float ui = (l_s[i + 1] + l_s[i]) / 2.0f, wi = l_w[i];
        float inner = 0.0f;
        for (int j = 0; j < n; ++j) inner += l_w[j] * fabsf(ui - (l_s[j + 1] + l_s[j]) / 2.0f);
        const float dt = l_s[i + 1] - l_s[i];
        s_wloss += wi * inner + wi * wi * dt / 3.0f;
        l_gw[i] += scale * (2.0f * inner + (2.0f / 3.0f) * wi * dt);
      }
    }
  }
  __syncthreads();                                 // l_gw complete; the loss scratch is free
  // ---- compositing VJP
  float* l_tn = X + L.scr;                         // T_{i+1}
  float* l_dg = l_tn + n;                          // delta_i * act'(raw_i)
  float* l_gh = l_dg + n;                          // g^_i
  const float dx = a.dirs[ray * 3], dy = a.dirs[ray * 3 + 1], dz = a.dirs[ray * 3 + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  float ex[3] = {1.0f, 1.0f, 1.0f};
  if (a.exposure_scale) { ex[0] = a.exposure_scale[ray * 3]; ex[1] = a.exposure_scale[ray * 3 + 1]; ex[2] = a.exposure_scale[ray * 3 + 2]; }
  float b[3] = {c.bg_value, c.bg_value, c.bg_value};
  if (c.bg_mode == 1) { b[0] = a.bg[ray * 3]; b[1] = a.bg[ray * 3 + 1]; b[2] = a.bg[ray * 3 + 2]; }
  float acc = 0.0f;
  for (int j = 0; j < J; ++j) {
    const int i = 4 * j + q;
    acc += i < n ? l_w[i] : 0.0f;
  }
  acc = lb_quad_sum(acc);
  const float g_bgw = (1.0f - acc > 0.0f) ? (go[0] * b[0] + go[1] * b[1] + go[2] * b[2]) : 0.0f;
  float ges[3] = {0.0f, 0.0f, 0.0f};               // d L / d exposure_scale[ray]
  {
    // forward over the samples: run_i = sum_{k<i} x_k (exclusive), T_{i+1} = exp(-(run_i + x_i)), g^_i, colour gradients
    float carry = 0.0f;
    for (int j = 0; j < J; ++j) {
      const int i = 4 * j + q;
      const bool in = i < n;
      const float raw = (in ? l_den[i] : 0.0f) + c.density_bias;
      const float sigma = cp_act(c.density_act, raw);
      const float delta = in ? (l_t[i + 1] - l_t[i]) * dnorm : 0.0f;
      const bool opaque_last = c.opaque_background && i == n - 1;
      const float x = opaque_last ? INFINITY : (in ? sigma * delta : 0.0f);
      const float inc = lb_quad_prefix(x, q);
      const float below = lb_dpp<0x90>(inc);       // (DPP outside any lane-dependent branch: a disabled source lane reads as 0)
      const float before = (q == 0 ? 0.0f : below) + carry;      // exclusive: the lanes below, no subtraction (x may be inf)
      carry += lb_dpp<0xFF>(inc);
      if (in) {
        const float w = l_w[i];
        l_tn[i] = expf(-(before + x));             // T_{i+1} = T_i - w_i
        l_dg[i] = delta * cp_act_grad(c.density_act, raw, sigma);
        float ghat = l_gw[i] - g_bgw;
        if (c.has_rgb) {
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float z = c.rgb_premultiplier * l_rgb[3 * i + ch] + c.rgb_bias;
            const float y = cp_act(c.rgb_act, z);
            const float col_pre = y * (1.0f + 2.0f * c.rgb_padding) - c.rgb_padding;
            const float col = col_pre * ex[ch];
            ghat += go[ch] * col;
            ges[ch] += w * go[ch] * col_pre;
            // d col / d raw = ex * (1+2pad) * act'(z) * premult
            l_rgb[3 * i + ch] = w * go[ch] * ex[ch] * (1.0f + 2.0f * c.rgb_padding) * cp_act_grad(c.rgb_act, z, y) *
                                c.rgb_premultiplier;
          }
        }
        l_gh[i] = ghat;
      }
    }
  }
  {
    // backward: suffix_i = sum_{k>i} g^_k w_k (exclusive), dL/dx_i = g^_i T_{i+1} - suffix_i
    float carry = 0.0f;
    for (int j = J - 1; j >= 0; --j) {
      const int i = 4 * j + q;
      const bool in = i < n;
      const float ghat = in ? l_gh[i] : 0.0f;
      const float inc = lb_quad_suffix(in ? ghat * l_w[i] : 0.0f, q);
      const float above = lb_dpp<0xF9>(inc);
      const float after = (q == 3 ? 0.0f : above) + carry;
      carry += lb_dpp<0x00>(inc);
      if (in) {
        float gx = ghat * l_tn[i] - after;
        if (c.opaque_background && i == n - 1) gx = 0.0f;      // x = +inf is a constant
        l_den[i] = gx * l_dg[i];
        l_gh[i] = gx;                              // d loss / d (sigma_i delta_i), for a.g_x (g^_i is consumed)
      }
    }
  }
  if (a.g_exposure_scale) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float g = lb_quad_sum(ges[ch]);
      if (live && q == 0) a.g_exposure_scale[ray * 3 + ch] += g;
    }
  }
  // loss values: wave sums, one atomic each
  if (a.data_loss_type >= 0 && a.data_stats) {
    s_mse = ls_wave_sum(s_mse);
    s_dloss = ls_wave_sum(s_dloss);
    if (threadIdx.x == 0) {
      const float denom = *a.denom;
      unsafeAtomicAdd(a.data_stats + 0, s_mse / denom);
      unsafeAtomicAdd(a.data_stats + 1, a.data_loss_mult * s_dloss / denom);
    }
  }
  if (a.wloss_mode != 0 && a.wloss_stat) {
    s_wloss = ls_wave_sum(s_wloss);
    const float norm = a.wloss_mode == 1 ? (float)a.B_valid * (float)a.n_ref : (float)a.B_valid;
    if (threadIdx.x == 0) unsafeAtomicAdd(a.wloss_stat, a.wloss_mult * s_wloss / norm);
  }
  __syncthreads();
  auto store_rows = [&](int off, float* dst, int len) {
    for (int e = threadIdx.x; e < rows * len; e += CP_THREADS) {
      const int rr = e / len, i = e - rr * len;
      dst[e] = lds[rr * L.stride + off + i];
    }
  };
  if (a.g_raw_density) store_rows(L.den, a.g_raw_density + ray0 * n, n);
  if (a.g_raw_density_bf16) {
    bf16* gb = (bf16*)a.g_raw_density_bf16;
    for (int e = threadIdx.x; e < rows * n; e += CP_THREADS) {
      const int rr = e / n, i = e - rr * n;
      gb[(ray0 * n + e) * (int64_t)a.ld_bf16] = (bf16)lds[rr * L.stride + L.den + i];
    }
  }
  if (c.has_rgb && a.g_raw_rgb) store_rows(L.rgb, a.g_raw_rgb + ray0 * n * 3, 3 * n);
  if (a.g_x) store_rows(L.scr + 2 * n, a.g_x + ray0 * n, n);
}

// The forward compositing on four lanes per ray (the same quad helpers and LDS plan as level_bwd_quad_kernel): lane q owns the
// samples 4j + q, the optical depth in front of a sample is a quad scan with a carry (summed in blocked instead of sequential
// order: weights differ from the lane-per-ray kernel in the last bits, both are held to the oracle by the same tolerances).
// LDS per ray: den n | t n+1 | w n | [rgb 3n], ray stride 4 x odd floats.
__global__ __launch_bounds__(CP_THREADS) void composite_fwd_quad_kernel(
    mnr_composite_cfg c, int64_t B, int stride, const float* __restrict__ raw_density, const float* __restrict__ noise,
    const float* __restrict__ raw_rgb, const float* __restrict__ tdist, const float* __restrict__ dirs,
    const float* __restrict__ bg, const float* __restrict__ expo, float* __restrict__ density,
    float* __restrict__ rgb, float* __restrict__ weights, float* __restrict__ rgb_out, float* __restrict__ acc_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = c.n;
  const int o_den = 0, o_t = n, o_w = 2 * n + 1, o_rgb = 3 * n + 1;
  const int64_t ray0 = (int64_t)blockIdx.x * LB_RPW;
  const int rows = (int)min((int64_t)LB_RPW, B - ray0);
  auto load_rows = [&](int off, const float* src, int len) {
    for (int e = threadIdx.x; e < rows * len; e += CP_THREADS) {
      const int rr = e / len, i = e - rr * len;
      lds[rr * stride + off + i] = src[e];
    }
  };
  auto store_rows = [&](int off, float* dst, int len) {
    for (int e = threadIdx.x; e < rows * len; e += CP_THREADS) {
      const int rr = e / len, i = e - rr * len;
      dst[e] = lds[rr * stride + off + i];
    }
  };
  load_rows(o_den, raw_density + ray0 * n, n);
  load_rows(o_t, tdist + ray0 * (n + 1), n + 1);
  if (c.has_rgb) load_rows(o_rgb, raw_rgb + ray0 * n * 3, 3 * n);
  if (noise && c.density_noise_std > 0.0f) {
    __syncthreads();
    for (int e = threadIdx.x; e < rows * n; e += CP_THREADS) {
      const int rr = e / n, i = e - rr * n;
      lds[rr * stride + o_den + i] += c.density_noise_std * noise[ray0 * n + e];     // models.py:462-464
    }
  }
  __syncthreads();
  const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
  const bool live = r < rows;
  const int64_t ray = live ? ray0 + r : B - 1;
  float* X = lds + r * stride;
  const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  float ex[3] = {1.0f, 1.0f, 1.0f};
  if (expo) { ex[0] = expo[ray * 3]; ex[1] = expo[ray * 3 + 1]; ex[2] = expo[ray * 3 + 2]; }
  float carry = 0.0f, acc = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
  const int J = (n + 3) >> 2;
  for (int j = 0; j < J; ++j) {
    const int i = 4 * j + q;
    const bool in = i < n;
    const float sigma = cp_act(c.density_act, (in ? X[o_den + i] : 0.0f) + c.density_bias);   // models.py:506
    const float delta = in ? (X[o_t + i + 1] - X[o_t + i]) * dnorm : 0.0f;                    // render.py:132-133
    float x = in ? sigma * delta : 0.0f;
    if (c.opaque_background && i == n - 1) x = INFINITY;                                      // render.py:136-142
    const float inc = lb_quad_prefix(x, q);
    const float below = lb_dpp<0x90>(inc);
    const float before = (q == 0 ? 0.0f : below) + carry;       // sum_{k<i} x_k (never contains the opaque last sample)
    carry += lb_dpp<0xFF>(inc);
    if (in) {
      const float w = (1.0f - expf(-x)) * expf(-before);
      X[o_den + i] = sigma;
      X[o_w + i] = w;
      acc += w;
      if (c.has_rgb) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          // models.py:584-586, 602, then exposure scaling :257-267.
          float v = cp_act(c.rgb_act, c.rgb_premultiplier * X[o_rgb + 3 * i + ch] + c.rgb_bias);
          v = v * (1.0f + 2.0f * c.rgb_padding) - c.rgb_padding;
          v *= ex[ch];
          X[o_rgb + 3 * i + ch] = v;
          if (ch == 0) cr += w * v; else if (ch == 1) cg += w * v; else cb += w * v;
        }
      }
    }
  }
  acc = lb_quad_sum(acc);
  cr = lb_quad_sum(cr);
  cg = lb_quad_sum(cg);
  cb = lb_quad_sum(cb);
  if (live && q == 0) {
    const float bgw = fmaxf(0.0f, 1.0f - acc);                                         // render.py:180
    float b0 = c.bg_value, b1 = c.bg_value, b2 = c.bg_value;
    if (c.bg_mode == 1) { b0 = bg[ray * 3]; b1 = bg[ray * 3 + 1]; b2 = bg[ray * 3 + 2]; }
    rgb_out[ray * 3 + 0] = cr + bgw * b0;                                              // render.py:181
    rgb_out[ray * 3 + 1] = cg + bgw * b1;
    rgb_out[ray * 3 + 2] = cb + bgw * b2;
    if (acc_out) acc_out[ray] = acc;
  }
  __syncthreads();
  store_rows(o_den, density + ray0 * n, n);
  store_rows(o_w, weights + ray0 * n, n);
  if (c.has_rgb && rgb) store_rows(o_rgb, rgb + ray0 * n * 3, 3 * n);
}

static size_t lb_floats_per_ray(const mnr_level_bwd_args* a) {
  const int n = a->cfg.n;
  size_t f = (size_t)(6 * n + 1) + n;                        // compositing VJP + d loss / d weights
  if (a->wloss_mode != 0) f += n + 1;                        // sdist
  if (a->wloss_mode == 1) f += 3 * (size_t)a->n_ref + 2;     // lo, hi, gi
  return f;
}

extern "C" int mnr_level_bwd(const mnr_level_bwd_args* a, void* stream) {
  MNR_CHECK_ARG(a != nullptr, "mnr_level_bwd: null args");
  const mnr_composite_cfg* cfg = &a->cfg;
  MNR_CHECK_ARG(a->B > 0 && a->B_valid > 0 && a->B_valid <= a->B && a->raw_density && a->tdist && a->dirs && a->weights,
                "mnr_level_bwd: null argument");
  MNR_CHECK_ARG(cfg->n >= 1 && cfg->n <= 1024, "mnr_level_bwd: n out of range");
  MNR_CHECK_ARG(a->g_raw_density || a->g_raw_density_bf16, "mnr_level_bwd: no density-gradient output");
  MNR_CHECK_ARG(!cfg->has_rgb || a->raw_rgb, "mnr_level_bwd: has_rgb needs raw_rgb");
  MNR_CHECK_ARG(cfg->bg_mode == 0 || a->bg, "mnr_level_bwd: bg_mode 1 needs bg");
  MNR_CHECK_ARG(a->data_loss_type < 0 || (a->data_loss_type <= MNR_LOSS_RAWNERF && a->rgb_out && a->gt && a->lossmult &&
                                          a->denom && (a->lm_c == 1 || a->lm_c == 3)),
                "mnr_level_bwd: the fused data loss needs rgb_out, gt, lossmult [B,1|3] and denom");
  MNR_CHECK_ARG(a->wloss_mode >= 0 && a->wloss_mode <= 2, "mnr_level_bwd: wloss_mode must be 0, 1 or 2");
  MNR_CHECK_ARG(a->wloss_mode == 0 || a->sdist, "mnr_level_bwd: weight losses need this level's sdist");
  MNR_CHECK_ARG(a->wloss_mode != 1 || (a->n_ref >= 1 && a->n_ref <= 1024 && a->t_ref && a->w_ref),
                "mnr_level_bwd: the interlevel loss needs the final level's (t_ref, w_ref)");
  // four lanes per ray whenever a wave's 16 rays fit LDS four workgroups per CU (n <= ~128); else lane per ray
  const LbLay lay = lb_layout(cfg->n, cfg->has_rgb, a->wloss_mode, a->n_ref);
  const size_t quad_lds = (size_t)lay.stride * LB_RPW * 4;
  if (g_level_bwd_quad && quad_lds <= quad_lds_max()) {
    static unsigned long long attr_q = 0;
    if (quad_lds > 64 * 1024 && mnr_attr_needed(&attr_q))
      (void)hipFuncSetAttribute((const void*)level_bwd_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(level_bwd_quad_kernel, dim3(mnr_cdiv(a->B, LB_RPW)), dim3(CP_THREADS), quad_lds, (hipStream_t)stream, *a);
    MNR_CHECK_LAUNCH();
    return MNR_OK;
  }
  const size_t fpr = lb_floats_per_ray(a);
  int S = CP_THREADS;
  while (S > 1 && fpr * S * 4 > 150 * 1024) S >>= 1;
  // one wave per SIMD of the chip (common.h: mnr_ray_wave_target)
  while (S > 8 && a->B / S < mnr_ray_wave_target()) S >>= 1;
  const size_t lds = fpr * S * 4;
  MNR_CHECK_ARG(lds <= 160 * 1024, "mnr_level_bwd: n=%d too long for LDS staging", cfg->n);
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if (mnr_attr_needed(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)level_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL(level_bwd_kernel, dim3(mnr_cdiv(a->B, S)), dim3(CP_THREADS), lds, (hipStream_t)stream, *a, S);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// The compositing VJP alone (no fused loss): the same kernel.
extern "C" int mnr_composite_bwd(const mnr_composite_cfg* cfg, int64_t B, const float* raw_density,
                                 const float* density_noise, const float* raw_rgb, const float* tdist,
                                 const float* dirs, const float* bg, const float* exposure_scale,
                                 const float* weights, const float* g_rgb_out, const float* g_weights,
                                 float* g_raw_density, uint16_t* g_raw_density_bf16, int ld_bf16,
                                 float* g_raw_rgb, float* g_exposure_scale, void* stream) {
  MNR_CHECK_ARG(cfg != nullptr, "mnr_composite_bwd: null cfg");
  mnr_level_bwd_args a = {};
  a.cfg = *cfg;
  a.B = a.B_valid = B;
  a.raw_density = raw_density;
  a.density_noise = density_noise;
  a.raw_rgb = raw_rgb;
  a.tdist = tdist;
  a.dirs = dirs;
  a.bg = bg;
  a.exposure_scale = exposure_scale;
  a.weights = weights;
  a.g_rgb_out = g_rgb_out;
  a.g_weights = g_weights;
  a.g_raw_density = g_raw_density;
  a.g_raw_density_bf16 = g_raw_density_bf16;
  a.ld_bf16 = ld_bf16;
  a.g_raw_rgb = g_raw_rgb;
  a.g_exposure_scale = g_exposure_scale;
  a.data_loss_type = -1;
  a.wloss_mode = 0;
  return mnr_level_bwd(&a, stream);
}

// ---------------------------------------------------------------------------
// compute_extras outputs (render.py:184-211): distance_mean and the 5/50/95 percentiles.

__global__ void render_extras_kernel(int64_t B, int n, const float* __restrict__ weights,
                                     const float* __restrict__ tdist, const float* __restrict__ t_far,
                                     float* __restrict__ out) {
  const int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= B) return;
  const float* w = weights + ray * n;
  const float* t = tdist + ray * (n + 1);
  float acc = 0.0f, e = 0.0f;
  for (int i = 0; i < n; ++i) {
    acc += w[i];
    e += w[i] * logf(0.5f * (t[i] + t[i + 1]));              // render.py:192-197
  }
  float dm = expf(e / fmaxf(MNR_F32_EPS, acc));
  if (dm != dm) dm = INFINITY;                                // nan_to_num(.., inf)
  dm = fminf(fmaxf(dm, t[0]), t[n]);
  out[ray * 4 + 0] = dm;
  // stepfun.weighted_percentile on (t_aug = [tdist, t_far], w_aug = [w, bg_w]) (render.py:203-207,
  // stepfun.py:298-308): np.interp of p/100 into cw = [0, min(1, cumsum(w_aug[:-1])), 1].
  const float ps[3] = {0.05f, 0.5f, 0.95f};
  const int nb = n + 1;                                       // bins of the augmented histogram
  const float far = t_far[ray];
  for (int q = 0; q < 3; ++q) {
    const float p = ps[q];
    float run = 0.0f, cw_lo = 0.0f;
    float res = far;
    bool found = false;
    for (int k = 0; k < nb && !found; ++k) {
      // fence-post k has cw_lo; fence-post k+1 has cw_hi.
      float cw_hi;
      if (k == nb - 1) cw_hi = 1.0f;
      else { run += w[k]; cw_hi = fminf(1.0f, run); }
      const float t_lo = t[k];
      const float t_hi = (k + 1 <= n) ? t[k + 1] : far;
      if (p < cw_hi || k == nb - 1) {
        // np.interp: last segment [cw_lo, cw_hi] with cw_lo <= p < cw_hi (ties resolve to the
        // right-most duplicate, as searchsorted(side='right') - 1 does).
        const float d = cw_hi - cw_lo;
        res = d > 0.0f ? t_lo + (p - cw_lo) / d * (t_hi - t_lo) : t_lo;
        if (p >= 1.0f) res = far;
        found = true;
      }
      cw_lo = cw_hi;
    }
    out[ray * 4 + 1 + q] = res;
  }
}

extern "C" int mnr_render_extras(int64_t B, int n, const float* weights, const float* tdist, const float* t_far,
                                 float* out, void* stream) {
  MNR_CHECK_ARG(B > 0 && n > 0 && weights && tdist && t_far && out, "mnr_render_extras: bad arguments");
  hipLaunchKernelGGL(render_extras_kernel, dim3(mnr_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, B, n, weights,
                     tdist, t_far, out);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// RawNeRF exposure scaling (models.py:257-267): rgb *= exposure_values; rgb *= 1 + [idx > 0] * offsets[idx].
// Both factors are per ray, so they are folded into one [B,3] scale consumed by the compositing kernels.

__global__ void exposure_scale_kernel(int64_t B, const float* __restrict__ ev, const int32_t* __restrict__ idx,
                                      const float* __restrict__ offsets, float* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int i = idx[b];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float sc = (offsets && i > 0) ? 1.0f + offsets[(int64_t)i * 3 + c] : 1.0f;
    out[b * 3 + c] = ev[b] * sc;
  }
}

extern "C" int mnr_exposure_scale(int64_t B, const float* exposure_values, const int32_t* exposure_idx,
                                  const float* offsets, float* out, void* stream) {
  MNR_CHECK_ARG(B > 0 && exposure_values && exposure_idx && out, "mnr_exposure_scale: bad arguments");
  hipLaunchKernelGGL(exposure_scale_kernel, dim3(mnr_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, B,
                     exposure_values, exposure_idx, offsets, out);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

__global__ void exposure_scale_bwd_kernel(int64_t B, const float* __restrict__ ev, const int32_t* __restrict__ idx,
                                          const float* __restrict__ g_scale, float* g_offsets) {
  // A batch holds a handful of distinct exposure indices, so one atomic per ray and channel is 16384 adds onto a dozen addresses
  // (221 us per step at llff_raw).  The lanes of a wave that share an index are summed first (one pass per distinct index in
  // the wave: its lowest pending lane names the index, a masked butterfly sums its lanes), then ONE lane adds the three sums.
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int i = b < B ? idx[b] : 0;
  if (i < 0) i = 0;                          // (<= 0: no gradient; the reference pins exposure 0 as the brightness reference)
  float v[3] = {0.0f, 0.0f, 0.0f};
  if (i > 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = ev[b] * g_scale[b * 3 + c];
  }
  const int lane = threadIdx.x & 63;
  unsigned long long pending = __ballot(i > 0);
  while (pending) {
    const int leader = __builtin_ctzll(pending);
    const int li = __shfl(i, leader, 64);
    const bool mine = i == li;
    float s[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      s[c] = mine ? v[c] : 0.0f;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) s[c] += __shfl_xor(s[c], d, 64);
    }
    if (lane == leader) {
#pragma unroll
      for (int c = 0; c < 3; ++c) unsafeAtomicAdd(g_offsets + (int64_t)li * 3 + c, s[c]);
    }
    pending &= ~__ballot(mine);
  }
}

extern "C" int mnr_exposure_scale_bwd(int64_t B_valid, const float* exposure_values, const int32_t* exposure_idx,
                                      const float* g_scale, float* g_offsets, void* stream) {
  MNR_CHECK_ARG(B_valid > 0 && exposure_values && exposure_idx && g_scale && g_offsets, "mnr_exposure_scale_bwd: bad arguments");
  hipLaunchKernelGGL(exposure_scale_bwd_kernel, dim3(mnr_cdiv(B_valid, 256)), dim3(256), 0, (hipStream_t)stream,
                     B_valid, exposure_values, exposure_idx, g_scale, g_offsets);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// Model.stop_level_grad = False (models.py:198-201): everything a level's loss says about its own sample distances,
// gathered per fence-post into d loss / d sdist [B, n+1]:
//   * the optical-depth increments x_i = sigma_i * (t_{i+1} - t_i) * |d| of render.py:144-145 (g_x from mnr_level_bwd),
//   * the Gaussians' dependence on the interval ends (g_t0, g_t1 per sample from mnr_cast_rays_ipe_bwd),
//     both through s_to_t (coord.py:96-98),
//   * the distortion loss's own dependence on sdist (stepfun.py:266-276; the final level),
//   * and the gradient that arrives from the NEXT level's resampling (mnr_resample_level_bwd's g_sdist_prev).
// The interlevel loss is piecewise constant in the distances (stepfun.py:64-77: searchsorted indices) and contributes nothing.
// One thread per fence-post.

__device__ __forceinline__ float sd_dtds(int fn, float s, float near, float far) {
  // d/ds of fn_inv(s * fn(far) + (1 - s) * fn(near)), coord.py:63-99
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
  const float dx = fn_far - fn_near;
  switch (fn) {
    case MNR_RAYDIST_RECIPROCAL: return -dx / (x * x);
    case MNR_RAYDIST_PIECEWISE: return x < 0.5f ? 2.0f * dx : 0.5f * dx / ((1.0f - x) * (1.0f - x));
    case MNR_RAYDIST_LOG: return expf(x) * dx;
    case MNR_RAYDIST_EXP: return dx / x;
    case MNR_RAYDIST_SQRT: return 2.0f * x * dx;
    case MNR_RAYDIST_SQUARE: return 0.5f * dx / sqrtf(x);
    default: return dx;
  }
}

__global__ __launch_bounds__(256) void sdist_bwd_kernel(mnr_sdist_bwd_args a) {
  const int n = a.n;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.B * (n + 1)) return;
  const int64_t ray = e / (n + 1);
  const int k = (int)(e - ray * (n + 1));
  float gt = 0.0f;
  if (a.g_x) {
    const float dx = a.dirs[ray * 3], dy = a.dirs[ray * 3 + 1], dz = a.dirs[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    auto sigma = [&](int i) {
      float raw = a.raw_density[ray * n + i] + a.density_bias;
      if (a.density_noise && a.density_noise_std > 0.0f) raw += a.density_noise_std * a.density_noise[ray * n + i];
      return cp_act(a.density_act, raw);
    };
    if (k >= 1) gt += a.g_x[ray * n + k - 1] * sigma(k - 1) * dnorm;
    if (k < n) gt -= a.g_x[ray * n + k] * sigma(k) * dnorm;
  }
  if (a.g_t0) {
    if (k < n) gt += a.g_t0[ray * n + k];
    if (k >= 1) gt += a.g_t1[ray * n + k - 1];
  }
  float gs = gt * sd_dtds(a.raydist_fn, a.sdist[e], a.near[ray], a.far[ray]);
  if (a.distortion_mult != 0.0f && ray < a.B_valid) {
    // loss = mult / B_valid * [ sum_ij w_i w_j |u_i - u_j| + 1/3 sum_i w_i^2 (s_{i+1} - s_i) ],  u_i = (s_i + s_{i+1}) / 2
    const float* s = a.sdist + ray * (n + 1);
    const float* w = a.weights + ray * n;
    float acc = 0.0f;
    for (int side = 0; side < 2; ++side) {
      const int i = k - 1 + side;                  // the two intervals this fence-post bounds
      if (i < 0 || i >= n) continue;
      const float ui = (s[i + 1] + s[i]) / 2.0f;
      float sg = 0.0f;
      for (int j = 0; j < n; ++j) {
        const float d = ui - (s[j + 1] + s[j]) / 2.0f;
        sg += d > 0.0f ? w[j] : (d < 0.0f ? -w[j] : 0.0f);
      }
      acc += 0.5f * (2.0f * w[i] * sg);            // d loss / d u_i, half of it to each end
      acc += (side == 0 ? 1.0f : -1.0f) * w[i] * w[i] / 3.0f;
    }
    gs += a.distortion_mult / (float)a.B_valid * acc;
  }
  if (a.g_sdist_in) gs += a.g_sdist_in[e];
  a.g_sdist[e] = gs;
}

extern "C" int mnr_sdist_bwd(const mnr_sdist_bwd_args* a, void* stream) {
  MNR_CHECK_ARG(a && a->B > 0 && a->n > 0 && a->sdist && a->near && a->far && a->g_sdist, "mnr_sdist_bwd: null argument");
  MNR_CHECK_ARG(!a->g_x || (a->raw_density && a->dirs), "mnr_sdist_bwd: g_x needs raw_density and dirs");
  MNR_CHECK_ARG((a->g_t0 == nullptr) == (a->g_t1 == nullptr), "mnr_sdist_bwd: g_t0 and g_t1 come together");
  MNR_CHECK_ARG(a->distortion_mult == 0.0f || a->weights, "mnr_sdist_bwd: the distortion term needs the weights");
  MNR_CHECK_ARG(a->raydist_fn >= 0 && a->raydist_fn <= MNR_RAYDIST_SQUARE, "mnr_sdist_bwd: bad raydist_fn");
  const int64_t total = a->B * (a->n + 1);
  hipLaunchKernelGGL(sdist_bwd_kernel, dim3(mnr_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, *a);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}
