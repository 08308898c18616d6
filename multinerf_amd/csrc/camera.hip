// Pixel -> ray generation on device (replaces camera_utils.pixels_to_rays / cast_ray_batch,
// internal/camera_utils.py:520-688, enabled by Config.cast_rays_in_train_step, train_utils.py:267-268).
// One thread per pixel: three pixel directions (centre, +x, +y) through the inverse intrinsics, optional
// radial/tangential undistortion (10 Newton steps, camera_utils.py:477-511), optional fisheye mapping,
// OpenCV -> OpenGL flip, camera rotation, optional NDC projection (camera_utils.py:32-98), cone radii.
// fp32 throughout, FP contraction off so that the arithmetic follows the reference's operation order.

#include "common.h"

#pragma clang fp contract(off)

struct CamDist {
  float k1, k2, k3, k4, p1, p2;
};

__device__ __forceinline__ void cam_undistort(const CamDist& c, float xd, float yd, float& xo, float& yo) {
  float x = xd, y = yd;
  for (int it = 0; it < 10; ++it) {
    const float r = x * x + y * y;
    const float d = 1.0f + r * (c.k1 + r * (c.k2 + r * (c.k3 + r * c.k4)));
    const float fx = d * x + 2.0f * c.p1 * x * y + c.p2 * (r + 2.0f * x * x) - xd;
    const float fy = d * y + 2.0f * c.p2 * x * y + c.p1 * (r + 2.0f * y * y) - yd;
    const float d_r = c.k1 + r * (2.0f * c.k2 + r * (3.0f * c.k3 + r * 4.0f * c.k4));
    const float d_x = 2.0f * x * d_r, d_y = 2.0f * y * d_r;
    const float fx_x = d + d_x * x + 2.0f * c.p1 * y + 6.0f * c.p2 * x;
    const float fx_y = d_y * x + 2.0f * c.p1 * x + 2.0f * c.p2 * y;
    const float fy_x = d_x * y + 2.0f * c.p2 * y + 2.0f * c.p1 * x;
    const float fy_y = d + d_y * y + 2.0f * c.p2 * x + 6.0f * c.p1 * y;
    const float den = fy_x * fx_y - fx_x * fy_y;
    const float xn = fx * fy_y - fy * fx_y;
    const float yn = fy * fx_x - fx * fy_x;
    const bool ok = fabsf(den) > 1e-9f;
    x += ok ? xn / den : 0.0f;
    y += ok ? yn / den : 0.0f;
  }
  xo = x;
  yo = y;
}

// camera_utils.py:76-98 with near = 1.
__device__ __forceinline__ void cam_to_ndc(const float* __restrict__ pndc, const float o[3], const float d[3],
                                           float on[3], float dn[3]) {
  const float t = -(1.0f + o[2]) / d[2];
  const float ox = o[0] + t * d[0], oy = o[1] + t * d[1], oz = o[2] + t * d[2];
  const float xm = 1.0f / pndc[2], ym = 1.0f / pndc[5];
  on[0] = xm * ox / oz;
  on[1] = ym * oy / oz;
  on[2] = -1.0f;
  dn[0] = xm * d[0] / d[2] - on[0];
  dn[1] = ym * d[1] / d[2] - on[1];
  dn[2] = 1.0f - on[2];
}

__global__ void pixels_to_rays_kernel(int64_t B, const int32_t* __restrict__ pix_x, const int32_t* __restrict__ pix_y,
                                      const int32_t* __restrict__ cam_idx, int num_cams,
                                      const float* __restrict__ pixtocams, const float* __restrict__ camtoworlds,
                                      int has_dist, CamDist dist, const float* __restrict__ pixtocam_ndc, int fisheye,
                                      float* __restrict__ origins, float* __restrict__ directions,
                                      float* __restrict__ viewdirs, float* __restrict__ radii,
                                      float* __restrict__ imageplane) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int ci = 0;
  if (cam_idx && num_cams > 1) {
    ci = cam_idx[b];
    ci = ci < 0 ? 0 : (ci >= num_cams ? num_cams - 1 : ci);
  }
  const float* P = pixtocams + (int64_t)ci * 9;
  const float* C = camtoworlds + (int64_t)ci * 12;
  const float px = (float)pix_x[b], py = (float)pix_y[b];
  float wd[3][3];                              // world-space directions of (centre, +x, +y)
  float ip[2];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const float u = px + (s == 1 ? 1.0f : 0.0f) + 0.5f;
    const float v = py + (s == 2 ? 1.0f : 0.0f) + 0.5f;
    float c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) c[i] = P[i * 3 + 0] * u + P[i * 3 + 1] * v + P[i * 3 + 2];
    if (has_dist) {
      float xu, yu;
      cam_undistort(dist, c[0], c[1], xu, yu);
      c[0] = xu;
      c[1] = yu;
      c[2] = 1.0f;
    }
    if (fisheye) {
      float theta = sqrtf(c[0] * c[0] + c[1] * c[1]);
      theta = fminf(3.14159265358979323846f, theta);
      const float sot = sinf(theta) / theta;
      c[0] *= sot;
      c[1] *= sot;
      c[2] = cosf(theta);
    }
    c[1] = -c[1];                              // OpenCV -> OpenGL
    c[2] = -c[2];
    if (s == 0) {
      ip[0] = c[0];
      ip[1] = c[1];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) wd[s][i] = C[i * 4 + 0] * c[0] + C[i * 4 + 1] * c[1] + C[i * 4 + 2] * c[2];
  }
  float o[3] = {C[3], C[7], C[11]};
  float d[3] = {wd[0][0], wd[0][1], wd[0][2]};
  const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float vd[3] = {d[0] * inv, d[1] * inv, d[2] * inv};
  float dxn, dyn;
  if (!pixtocam_ndc) {
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float ex = wd[1][i] - d[i], ey = wd[2][i] - d[i];
      sx += ex * ex;
      sy += ey * ey;
    }
    dxn = sqrtf(sx);
    dyn = sqrtf(sy);
  } else {
    float on[3], dn[3], ox[3], oy[3], tmp[3];
    cam_to_ndc(pixtocam_ndc, o, wd[1], ox, tmp);
    cam_to_ndc(pixtocam_ndc, o, wd[2], oy, tmp);
    cam_to_ndc(pixtocam_ndc, o, d, on, dn);
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float ex = ox[i] - on[i], ey = oy[i] - on[i];
      sx += ex * ex;
      sy += ey * ey;
      o[i] = on[i];
      d[i] = dn[i];
    }
    dxn = sqrtf(sx);
    dyn = sqrtf(sy);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    origins[b * 3 + i] = o[i];
    directions[b * 3 + i] = d[i];
    viewdirs[b * 3 + i] = vd[i];
  }
  radii[b] = (0.5f * (dxn + dyn)) * 2.0f / sqrtf(12.0f);
  imageplane[b * 2 + 0] = ip[0];
  imageplane[b * 2 + 1] = ip[1];
}

extern "C" int mnr_pixels_to_rays(int64_t B, const int32_t* pix_x_int, const int32_t* pix_y_int, const int32_t* cam_idx,
                                  int num_cams, const float* pixtocams, const float* camtoworlds,
                                  const float* distortion6, const float* pixtocam_ndc, int camtype, float* origins,
                                  float* directions, float* viewdirs, float* radii, float* imageplane, void* stream) {
  MNR_CHECK_ARG(B > 0 && pix_x_int && pix_y_int && pixtocams && camtoworlds && origins && directions && viewdirs &&
                    radii && imageplane,
                "mnr_pixels_to_rays: null argument");
  MNR_CHECK_ARG(num_cams >= 1 && (num_cams == 1 || cam_idx), "mnr_pixels_to_rays: stacked cameras need cam_idx");
  MNR_CHECK_ARG(camtype == MNR_CAM_PERSPECTIVE || camtype == MNR_CAM_FISHEYE, "mnr_pixels_to_rays: unknown camtype %d", camtype);
  CamDist dist = {0, 0, 0, 0, 0, 0};
  if (distortion6) {
    // host copy of six floats: the distortion parameters are a Python dict of floats in the reference
    dist = {distortion6[0], distortion6[1], distortion6[2], distortion6[3], distortion6[4], distortion6[5]};
  }
  hipLaunchKernelGGL(pixels_to_rays_kernel, dim3(mnr_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, B, pix_x_int,
                     pix_y_int, cam_idx, num_cams, pixtocams, camtoworlds, distortion6 ? 1 : 0, dist, pixtocam_ndc,
                     camtype == MNR_CAM_FISHEYE, origins, directions, viewdirs, radii, imageplane);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}
