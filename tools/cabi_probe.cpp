// GEMM A/B through the C ABI alone (no Python, no torch: starts in milliseconds on a fresh box).
//
//   hipcc -O2 -o tools/_bin/cabi_probe tools/cabi_probe.cpp -Iinclude -Lmultinerf_amd -lmnerf_hip -Wl,-rpath,'$ORIGIN/../../multinerf_amd'
//   tools/_bin/cabi_probe [cfg ...]            (default: every prepared 256-row NT configuration, then the TN pair)
//
// For every NT configuration: a bitwise screen against NtC2 (forward layer with bias + ReLU + bit masks over a [A1|A2]
// concat, the dX layer reading those masks) at M = 8192, then timings at the shapes of the 360.gin step: the 1024-wide
// trunk layer forward and dX at M = 524288, the 256-wide proposal layer at M = 1048576.  Then the weight-gradient kernel,
// default and split-path: relative difference at M = 8192, timing at the trunk shape.  Everything the Python probe
// (tools/gemm_probe.py) screens, minus the interpreter start-up.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "mnerf.h"

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)
#define MNR(x)                                                                \
  do {                                                                        \
    int s_ = (x);                                                             \
    if (s_ != 0) {                                                            \
      fprintf(stderr, "%s:%d: status %d: %s\n", __FILE__, __LINE__, s_, mnr_last_error()); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// seeded uniform(-scale, scale) bf16 on the device (filled on the host: no kernel of our own needed)
static uint16_t* dev_bf16(size_t n, float scale, uint32_t seed) {
  std::vector<uint16_t> h(n);
  uint32_t s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = f2bf(((s >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale);
  }
  uint16_t* d;
  CHECK(hipMalloc(&d, n * 2));
  CHECK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}

template <class T>
static T* dev_zero(size_t n) {
  T* d;
  CHECK(hipMalloc(&d, n * sizeof(T)));
  CHECK(hipMemset(d, 0, n * sizeof(T)));
  return d;
}

template <class F>
static float time_us(F fn, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < (reps > 1 ? 3 : 0); ++i) fn();
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) fn();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

static bool same(const void* a, const void* b, size_t bytes) {
  std::vector<char> ha(bytes), hb(bytes);
  CHECK(hipMemcpy(ha.data(), a, bytes, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hb.data(), b, bytes, hipMemcpyDeviceToHost));
  return memcmp(ha.data(), hb.data(), bytes) == 0;
}

int main(int argc, char** argv) {
  std::vector<int> cfgs;
  bool small = false;                                    // --small: toy sizes (the simulator build of this probe)
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--small")) small = true;
    else cfgs.push_back(atoi(argv[i]));
  }
  if (cfgs.empty()) cfgs = {2, 43, 44, 45, 41, 40, 42, 35, 36, 37, 38, 39, 18};
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("%s, %d CUs, ABI %d\n", prop.gcnArchName, prop.multiProcessorCount, mnr_abi_version());

  const int K = 1024, N = 1024, K1 = 768, K2 = 256;
  const int64_t Mc = small ? 256 : 8192, Mt = small ? 512 : 524288, Mp = small ? 512 : 1048576;
  const int reps = small ? 1 : 10;
  uint16_t* A = dev_bf16((size_t)Mt * K, 1.0f, 1);            // [Mt, 1024]; its first Mc rows serve the screens
  uint16_t* Bt = dev_bf16((size_t)N * K, 0.05f, 2);
  uint16_t* A2 = dev_bf16((size_t)Mp * 256, 1.0f, 3);         // [Mp, 256]
  uint16_t* B2 = dev_bf16((size_t)256 * 256, 0.05f, 4);
  uint16_t* dY = dev_bf16((size_t)Mt * N, 1.0f, 5);           // TN second operand
  std::vector<float> hb(N);
  for (int i = 0; i < N; ++i) hb[i] = 0.01f * (float)((i * 37) % 101 - 50);
  float* bias = dev_zero<float>(N);
  CHECK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  uint16_t* C = dev_zero<uint16_t>((size_t)Mp * 256 > (size_t)Mt * N ? (size_t)Mp * 256 : (size_t)Mt * N);
  uint8_t* bits = dev_zero<uint8_t>((size_t)Mt * N / 8);
  uint16_t *Cr = dev_zero<uint16_t>(Mc * N), *Cx = dev_zero<uint16_t>(Mc * N), *Dr = dev_zero<uint16_t>(Mc * N), *Dx = dev_zero<uint16_t>(Mc * N);
  uint8_t *br = dev_zero<uint8_t>(Mc * N / 8), *bx = dev_zero<uint8_t>(Mc * N / 8);

  auto fwd = [&](int64_t M, uint16_t* out, uint8_t* bo, bool concat) {
    mnr_gemm_nt_args a;
    memset(&a, 0, sizeof(a));
    a.A1 = A; a.lda1 = K; a.K1 = concat ? K1 : K;
    if (concat) { a.A2 = A + K1; a.lda2 = K; a.K2 = K2; }
    a.Bt = Bt; a.ldb = K; a.M = M; a.N = N; a.bias = bias; a.n_bias = N; a.relu = 1;
    a.Cb = out; a.ldcb = N; a.nb = N; a.mask_bits_out = bo; a.ld_bits_out = N / 8;
    MNR(mnr_gemm_nt_bf16(&a, nullptr));
  };
  auto dx = [&](int64_t M, uint16_t* out, const uint8_t* bi) {
    mnr_gemm_nt_args a;
    memset(&a, 0, sizeof(a));
    a.A1 = A; a.lda1 = K; a.K1 = K; a.Bt = Bt; a.ldb = K; a.M = M; a.N = N;
    a.Cb = out; a.ldcb = N; a.nb = N; a.mask_bits_in = bi; a.ld_bits_in = N / 8;
    MNR(mnr_gemm_nt_bf16(&a, nullptr));
  };
  auto prop_fwd = [&]() {
    mnr_gemm_nt_args a;
    memset(&a, 0, sizeof(a));
    a.A1 = A2; a.lda1 = 256; a.K1 = 256; a.Bt = B2; a.ldb = 256; a.M = Mp; a.N = 256; a.bias = bias; a.n_bias = 256; a.relu = 1;
    a.Cb = C; a.ldcb = 256; a.nb = 256; a.mask_bits_out = bits; a.ld_bits_out = 32;
    MNR(mnr_gemm_nt_bf16(&a, nullptr));
  };

  MNR(mnr_gemm_nt_set_config(2, 0));
  fwd(Mc, Cr, br, true);
  dx(Mc, Dr, br);
  CHECK(hipDeviceSynchronize());
  for (int c : cfgs) {
    if (mnr_gemm_nt_set_config(c, 0) != 0) {
      printf("cfg %d: not compiled in\n", c);
      continue;
    }
    bool ok = true;
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(Cx, 0xff, Mc * N * 2));
      CHECK(hipMemset(Dx, 0xff, Mc * N * 2));
      fwd(Mc, Cx, bx, true);
      dx(Mc, Dx, br);
      CHECK(hipDeviceSynchronize());
      ok = ok && same(Cx, Cr, Mc * N * 2) && same(bx, br, Mc * N / 8) && same(Dx, Dr, Mc * N * 2);
    }
    const float tf = time_us([&] { fwd(Mt, C, bits, false); }, reps);
    const float td = time_us([&] { dx(Mt, C, bits); }, reps);
    const float tp = time_us([&] { prop_fwd(); }, reps);
    printf("cfg %2d: %s | fwd 1024 %7.1f us %6.1f TF/s | dX 1024 %7.1f us %6.1f TF/s | prop 256 %7.1f us %6.1f TF/s\n", c,
           ok ? "bitwise = cfg 2" : "MISMATCH      ", tf, 2.0 * Mt * N * K / tf / 1e6, td, 2.0 * Mt * N * K / td / 1e6, tp,
           2.0 * Mp * 256 * 256 / tp / 1e6);
    fflush(stdout);
  }
  MNR(mnr_gemm_nt_set_config(2, 0));
  {
    // weights-resident persistent kernel on the proposal shape: bitwise screen at M = Mc rows of it, then timing
    uint16_t *Pr = dev_zero<uint16_t>(Mc * 256), *Px = dev_zero<uint16_t>(Mc * 256);
    auto prop_small = [&](uint16_t* out) {
      mnr_gemm_nt_args a;
      memset(&a, 0, sizeof(a));
      a.A1 = A2; a.lda1 = 256; a.K1 = 256; a.Bt = B2; a.ldb = 256; a.M = Mc; a.N = 256; a.bias = bias; a.n_bias = 256; a.relu = 1;
      a.Cb = out; a.ldcb = 256; a.nb = 256; a.mask_bits_out = bits; a.ld_bits_out = 32;
      MNR(mnr_gemm_nt_bf16(&a, nullptr));
    };
    prop_small(Pr);
    for (int wgs : {1, 128}) {
      MNR(mnr_gemm_nt_set_wres(wgs));
      CHECK(hipMemset(Px, 0xff, Mc * 256 * 2));
      prop_small(Px);
      CHECK(hipDeviceSynchronize());
      const bool ok = same(Px, Pr, Mc * 256 * 2);
      const float tp = time_us([&] { prop_fwd(); }, reps);
      printf("wres %3d: %s | prop 256 %7.1f us %6.1f TF/s\n", wgs, ok ? "bitwise = cfg 2" : "MISMATCH      ", tp, 2.0 * Mp * 256 * 256 / tp / 1e6);
    }
    MNR(mnr_gemm_nt_set_wres(0));
  }

  // weight gradient: C[K, N] += A^T dY
  float *W0 = dev_zero<float>((size_t)K * N), *W1 = dev_zero<float>((size_t)K * N), *db = dev_zero<float>(N);
  auto tn = [&](int64_t M, float* out) {
    mnr_gemm_tn_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = K; a.K = K; a.B = dY; a.ldb = N; a.N = N; a.M = M; a.C = out; a.ldc = N; a.k_valid = K; a.n_valid = N;
    a.bias_out = db; a.bias_n_valid = N;
    MNR(mnr_gemm_tn_bf16(&a, nullptr));
  };
  MNR(mnr_gemm_tn_set_split(0));
  tn(Mc, W0);
  MNR(mnr_gemm_tn_set_split(1));
  tn(Mc, W1);
  CHECK(hipDeviceSynchronize());
  {
    std::vector<float> h0((size_t)K * N), h1((size_t)K * N);
    CHECK(hipMemcpy(h0.data(), W0, h0.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h1.data(), W1, h1.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (size_t i = 0; i < h0.size(); ++i) {
      num += (double)(h0[i] - h1[i]) * (h0[i] - h1[i]);
      den += (double)h0[i] * h0[i];
    }
    printf("tn split vs default at M = %lld: relative difference %.2e %s\n", (long long)Mc, den > 0 ? sqrt(num / den) : -1.0,
           (den > 0 && num / den < 1e-10) ? "(ok: fp32 atomics order only)" : "(MISMATCH)");
  }
  for (int split = 0; split < 3; ++split) {
    MNR(mnr_gemm_tn_set_split(split));
    const float t = time_us([&] { tn(Mt, W0); }, reps);
    printf("tn %s: dW 1024x1024 over M = %lld: %7.1f us %6.1f TF/s\n", split == 1 ? "split  " : split == 2 ? "imm    " : "default",
           (long long)Mt, t, 2.0 * Mt * N * K / t / 1e6);
  }
  MNR(mnr_gemm_tn_set_split(0));
  return 0;
}
