// Gradient clipping + Adam (gfx950): one streaming pass over {grad, param, mu, nu}.
//
// Replaces train_utils.clip_gradients (train_utils.py:200-218: per top-level
// module, clip by value then by norm with mult = min(1, max_norm/(eps + |g|))),
// jnp.nan_to_num (:328) and optax.adam via state.apply_gradients (:330, :372).
// HBM-bound: 16 B/param read + 12 B/param written.
#include "common.h"

__global__ void grad_sqnorm_kernel(const float* __restrict__ g, int64_t begin, int64_t end, float max_val,
                                   float* out) {
  float s = 0.0f;
  for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end;
       i += (int64_t)gridDim.x * blockDim.x) {
    float v = g[i];
    if (max_val > 0.0f && v == v) v = fminf(fmaxf(v, -max_val), max_val);   // jnp.clip keeps NaN
    s += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

extern "C" int mnr_grad_sqnorm(const float* grad, int64_t begin, int64_t end, float max_val, float* out,
                               void* stream) {
  MNR_CHECK_ARG(grad && out && end > begin, "mnr_grad_sqnorm: bad arguments");
  int grid = mnr_cdiv(end - begin, 256 * 8);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad, begin, end, max_val, out);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// losses['weight'] term of one top-level module (train_utils.py:302-305): loss += mult * sum p^2, grad += 2 mult p.
__global__ void weight_decay_kernel(const float* __restrict__ p, int64_t begin, int64_t end, float mult,
                                    float* __restrict__ grad, float* loss_out, float* sqnorm_out) {
  float s = 0.0f;
  for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float v = p[i];
    s += v * v;
    if (grad && mult != 0.0f) grad[i] += 2.0f * mult * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = part[0] + part[1] + part[2] + part[3];
    if (loss_out) unsafeAtomicAdd(loss_out, mult * t);
    if (sqnorm_out) unsafeAtomicAdd(sqnorm_out, t);
  }
}

extern "C" int mnr_weight_decay(const float* params, int64_t begin, int64_t end, float mult, float* grad,
                                float* loss_out, float* sqnorm_out, void* stream) {
  MNR_CHECK_ARG(params && end > begin, "mnr_weight_decay: bad arguments");
  int grid = mnr_cdiv(end - begin, 256 * 8);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(weight_decay_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, params, begin, end, mult, grad,
                     loss_out, sqnorm_out);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

__device__ __forceinline__ float op_nan_to_num(float x) {
  if (x != x) return 0.0f;
  if (x > MNR_F32_MAX) return MNR_F32_MAX;
  if (x < -MNR_F32_MAX) return -MNR_F32_MAX;
  return x;
}

__global__ void clip_adam_kernel(mnr_adam_cfg c, int64_t begin, int64_t end, const float* __restrict__ sqnorm,
                                 const float* __restrict__ grad, float* __restrict__ params, float* __restrict__ mu,
                                 float* __restrict__ nu) {
  float mult = 1.0f;
  if (c.grad_max_norm > 0.0f) {
    // jnp.minimum propagates NaN (a NaN anywhere in the module zeroes its whole update after
    // nan_to_num, train_utils.py:212-214,328); fminf would drop it.
    const float r = c.grad_max_norm / (MNR_F32_EPS + sqrtf(*sqnorm));
    mult = (r != r) ? r : fminf(1.0f, r);
  }
  for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end;
       i += (int64_t)gridDim.x * blockDim.x) {
    float g = grad[i];
    if (c.grad_max_val > 0.0f && g == g) g = fminf(fmaxf(g, -c.grad_max_val), c.grad_max_val);
    g = op_nan_to_num(mult * g);
    const float m = c.b1 * mu[i] + (1.0f - c.b1) * g;
    const float v = c.b2 * nu[i] + (1.0f - c.b2) * g * g;
    mu[i] = m;
    nu[i] = v;
    const float mh = m / c.bias_corr1;
    const float vh = v / c.bias_corr2;
    params[i] = params[i] - c.lr * mh / (sqrtf(vh) + c.eps);
  }
}

extern "C" int mnr_clip_adam(const mnr_adam_cfg* cfg, int64_t begin, int64_t end, const float* sqnorm_seg,
                             const float* grad, float* params, float* mu, float* nu, void* stream) {
  MNR_CHECK_ARG(cfg && grad && params && mu && nu && end > begin, "mnr_clip_adam: bad arguments");
  MNR_CHECK_ARG(cfg->grad_max_norm <= 0.0f || sqnorm_seg, "mnr_clip_adam: grad_max_norm needs sqnorm_seg");
  int grid = mnr_cdiv(end - begin, 256 * 4);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(clip_adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *cfg, begin, end, sqnorm_seg,
                     grad, params, mu, nu);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}
