/* mnerf.h -- C ABI of libmnerf_hip.so: MI355X (gfx950) kernels for the MultiNeRF
 * per-ray render/train hot path.
 *
 * The reference (google-research/multinerf) is pure Python/JAX and has no
 * native operator interface; its hot path is the Python call surface
 * Model.__call__ / MLP.__call__ / train_step.  This header is the boundary a
 * host (ctypes here; cffi/pybind in a maintainer's tree)
 * binds to replace the jax.numpy bodies of those functions.  Each entry point
 * names the reference function(s) it replaces (file:line under the reference
 * repository root).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (row-major,
 *    contiguous unless a leading dimension is given); the library never
 *    allocates, frees or retains them;
 *  - `stream` is a hipStream_t passed as void*; calls only enqueue work;
 *  - return value: 0 (MNR_OK) or a negative mnr_status; mnr_last_error()
 *    returns a thread-local NUL-terminated description of the last failure;
 *  - fp32 unless a name says bf16 (raw IEEE bfloat16, uint16_t storage);
 *  - "B" = rays, "n" = intervals along a ray, fence-post arrays have n+1
 *    entries (reference internal/stepfun.py:15-23).
 */
#ifndef MNERF_H_
#define MNERF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MNR_OK = 0,
  MNR_ERR_INVALID_ARGUMENT = -1,
  MNR_ERR_HIP = -2,
  MNR_ERR_UNSUPPORTED = -3
} mnr_status;

const char* mnr_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int mnr_abi_version(void);
/* Device properties the host needs to size launches: fills cu_count, lds_bytes, gcn arch name. */
int mnr_device_info(int device, int* cu_count, int* lds_bytes_per_cu, char* arch_name, int arch_name_len);

/* ------------------------------------------------------------------------- *
 * Sampling  (replaces models.py:153-204 = stepfun.max_dilate_weights
 * stepfun.py:99-128, anneal/logits models.py:173-185, stepfun.sample_intervals
 * stepfun.py:164-263 incl. invert_cdf :153-161, integrate_weights :131-150,
 * math.sorted_interp math.py:108-127, and coord.construct_ray_warps s_to_t
 * coord.py:63-99)
 * ------------------------------------------------------------------------- */
typedef enum { MNR_RAYDIST_IDENTITY = 0, MNR_RAYDIST_RECIPROCAL = 1, MNR_RAYDIST_PIECEWISE = 2,
               MNR_RAYDIST_LOG = 3, MNR_RAYDIST_EXP = 4, MNR_RAYDIST_SQRT = 5, MNR_RAYDIST_SQUARE = 6 } mnr_raydist;

typedef struct {
  int n_prev;              /* intervals in the incoming step function (1 at level 0)  */
  int n_samples;           /* intervals to draw (Model.num_prop_samples / num_nerf_samples) */
  int use_dilation;        /* models.py:162-171 (level > 0 and dilation enabled)      */
  float dilation;          /* models.py:153-154                                       */
  float domain_lo, domain_hi; /* (init_s_near, init_s_far)                            */
  float anneal;            /* Schlick bias of train_frac, models.py:174-179 (1 if slope==0) */
  float resample_padding;  /* models.py:70                                            */
  int single_jitter;       /* jitter shape [B] (1) or [B,n_samples] (0)               */
  float max_jitter;        /* stepfun.py:204-205; ignored when jitter == NULL         */
  int raydist_fn;          /* mnr_raydist for s_to_t                                  */
} mnr_resample_cfg;

/* One hierarchical-sampling level for B rays.
 *  sdist_prev [B,n_prev+1], w_prev [B,n_prev]: incoming step function (s-space).
 *  u_base [n_samples]: the linspace of stepfun.py:198 / :208 (host-computed).
 *  jitter: NULL (rng=None) or uniform[0,1) draws, [B] or [B,n_samples].
 *  near,far [B].  Outputs: sdist [B,n+1], tdist [B,n+1]; idx_out (optional,
 *  may be NULL) [B,n] int32 = count(cw <= u) - 1, the bit-exact sample index. */
int mnr_resample_level(const mnr_resample_cfg* cfg, int64_t B,
                       const float* sdist_prev, const float* w_prev,
                       const float* u_base, const float* jitter,
                       const float* near, const float* far,
                       float* sdist_out, float* tdist_out, int32_t* idx_out, void* stream);

/* VJP of mnr_resample_level w.r.t. its incoming step function: Model.stop_level_grad = False (models.py:56,198-201; what
 * jax differentiates there: stepfun.max_dilate_weights stepfun.py:99-128, the logits models.py:183-185, jax.nn.softmax and
 * integrate_weights stepfun.py:131-156, math.sorted_interp math.py:108-127, the interval fence-posts stepfun.py:252-262).
 *  Inputs as the forward call's (same cfg, sdist_prev, w_prev, u_base, jitter: the forward pass is re-run inside);
 *  g_sdist [B,n+1] = d loss / d sdist_out.  Outputs (overwritten): g_sdist_prev [B,n_prev+1], g_w_prev [B,n_prev]. */
int mnr_resample_level_bwd(const mnr_resample_cfg* cfg, int64_t B,
                           const float* sdist_prev, const float* w_prev,
                           const float* u_base, const float* jitter, const float* g_sdist,
                           float* g_sdist_prev, float* g_w_prev, void* stream);

/* Leaf: math.sorted_interp(u, cw, t) (math.py:108-127) with the integer index.
 * cw,t [B,nc]; u [B,nu] -> out [B,nu], idx [B,nu] (may be NULL). */
int mnr_sorted_interp(int64_t B, int nc, int nu, const float* u, const float* cw, const float* t,
                      float* out, int32_t* idx, void* stream);

/* Leaf: stepfun.max_dilate_weights(t, w, dilation, domain, renormalize=True)
 * (stepfun.py:116-128). t [B,n+1], w [B,n] -> t_out [B,3n+1], w_out [B,3n];
 * scratch: unused since round 2 (the kernel works in LDS); kept in the signature, may be NULL. */
int mnr_max_dilate_weights(int64_t B, int n, const float* t, const float* w, float dilation,
                           float domain_lo, float domain_hi, float* t_out, float* w_out,
                           float* scratch, void* stream);

/* ------------------------------------------------------------------------- *
 * Featurisation  (replaces render.cast_rays render.py:103-127 incl.
 * conical_frustum_to_gaussian :44-78 / cylinder_to_gaussian :81-100 /
 * lift_gaussian :21-41, coord.track_linearize(coord.contract) coord.py:21-60,
 * coord.lift_and_diagonalize :129-133, coord.integrated_pos_enc :102-126 with
 * math.safe_sin math.py:26-38; and coord.pos_enc :136-147 for view directions)
 * ------------------------------------------------------------------------- */
typedef struct {
  int ray_shape;       /* 0 cone, 1 cylinder (Model.ray_shape)                */
  int warp_contract;   /* 1: MLP.warp_fn = coord.contract                     */
  int disable_integration; /* Model.disable_integration: zero covariances     */
  int basis_k;         /* K columns of pos_basis_t (21 icosa-2, 3 octa-1)     */
  int min_deg, max_deg;
} mnr_ipe_cfg;

/* tdist [B,n+1]; origins, directions [B,3]; radii [B]; basis [K,3] (rows =
 * geopoly.generate_basis rows, geopoly.py:78-124).  feat_out: bf16
 * [B*n, ld_feat], columns [0, 2*K*(max_deg-min_deg)) = IPE features in the
 * reference order, remaining columns zero.  means_out / covs_out (optional,
 * fp32 [B*n,3] / [B*n,9]) are the POST-warp Gaussians, for parity tests. */
int mnr_cast_rays_ipe(const mnr_ipe_cfg* cfg, int64_t B, int n, const float* tdist,
                      const float* origins, const float* directions, const float* radii,
                      const float* basis, uint16_t* feat_out, int ld_feat,
                      float* means_out, float* covs_out, void* stream);

/* Same, fp32 features [B*n, 2*K*L] (no padding): parity-test leaf for
 * coord.integrated_pos_enc. */
int mnr_cast_rays_ipe_f32(const mnr_ipe_cfg* cfg, int64_t B, int n, const float* tdist,
                          const float* origins, const float* directions, const float* radii,
                          const float* basis, float* feat_out, void* stream);

/* Tangent features for the density-gradient normals (models.py:478-492 via forward mode):
 * row c*B*n + s of feat_out (bf16 [3*B*n, ld_feat]) = d(IPE features of sample s)/d(mean_c), c = x,y,z, mean = the Gaussian's
 * mean as predict_density receives it (models.py:441-446: BEFORE warp_fn; with cfg->warp_contract the rows carry the
 * contraction's Jacobian and, through J cov J^T with the covariance held fixed, its derivative). */
int mnr_cast_rays_ipe_tangent(const mnr_ipe_cfg* cfg, int64_t B, int n, const float* tdist,
                              const float* origins, const float* directions, const float* radii,
                              const float* basis, uint16_t* feat_out, int ld_feat, void* stream);

/* VJP of mnr_cast_rays_ipe w.r.t. the interval ends (Model.stop_level_grad = False, models.py:198-201: render.cast_rays
 * render.py:103-127 incl. the conical-frustum / cylinder moments :44-100 and lift_gaussian :21-41, coord.track_linearize(
 * contract) coord.py:21-60 with the contraction's second derivative, lift_and_diagonalize :129-133, integrated_pos_enc
 * :102-126).  g_feat_a (and optionally g_feat_b, summed) bf16 [B*n, ld_feat] = d loss / d features;
 * outputs g_t0, g_t1 fp32 [B*n] = d loss / d (tdist[ray, j], tdist[ray, j+1]) of sample (ray, j). */
int mnr_cast_rays_ipe_bwd(const mnr_ipe_cfg* cfg, int64_t B, int n, const float* tdist,
                          const float* origins, const float* directions, const float* radii,
                          const float* basis, const uint16_t* g_feat_a, const uint16_t* g_feat_b, int ld_feat,
                          float* g_t0, float* g_t1, void* stream);

/* VJP of mnr_cast_rays_ipe_tangent w.r.t. the interval ends: Model.stop_level_grad = False next to density-gradient normals
 * (models.py:198-201 with :478-492: the normals are a derivative of predict_density AT the sample's Gaussian, so they depend on
 * the sample positions as well).  g_T_a (and optionally g_T_b, summed) bf16 [3*B*n, ld_feat] = d loss / d (tangent rows) (the
 * tangent network's dX GEMMs of the trunk layers that read the features); g_t0, g_t1 fp32 [B*n] are ACCUMULATED into (call
 * after mnr_cast_rays_ipe_bwd).  With cfg->warp_contract the contraction's third derivative enters (forward-mode duals). */
int mnr_cast_rays_ipe_tangent_bwd(const mnr_ipe_cfg* cfg, int64_t B, int n, const float* tdist,
                                  const float* origins, const float* directions, const float* radii,
                                  const float* basis, const uint16_t* g_T_a, const uint16_t* g_T_b, int ld_feat,
                                  float* g_t0, float* g_t1, void* stream);

/* coord.pos_enc(viewdirs, 0, deg_view, append_identity=True) per ray, written
 * (bf16) into columns [col0, col0+3+6*deg_view) of every one of the ray's n
 * rows of `dst` [B*n, ld]; columns up to col_end are zero-filled
 * (models.py:550-557 broadcast + concat). */
int mnr_viewdir_enc_fill(int64_t B, int n, const float* viewdirs, int deg_view,
                         uint16_t* dst, int ld, int col0, int col_end, void* stream);

/* ------------------------------------------------------------------------- *
 * Ray generation  (replaces camera_utils.pixels_to_rays / cast_ray_batch,
 * internal/camera_utils.py:520-688; used when Config.cast_rays_in_train_step,
 * train_utils.py:267-268)
 * pix_x_int / pix_y_int / cam_idx: int32 [B] (device).  pixtocams [num_cams,3,3],
 * camtoworlds [num_cams,3,4] fp32 (device; num_cams == 1: one camera for all pixels,
 * cam_idx may be NULL).  distortion6 = HOST pointer to {k1,k2,k3,k4,p1,p2} or NULL.
 * pixtocam_ndc: device [3,3] or NULL (NDC rays, camera_utils.py:32-98).
 * Outputs fp32 device: origins, directions, viewdirs [B,3], radii [B], imageplane [B,2].
 * ------------------------------------------------------------------------- */
typedef enum { MNR_CAM_PERSPECTIVE = 0, MNR_CAM_FISHEYE = 1 } mnr_camtype;
int mnr_pixels_to_rays(int64_t B, const int32_t* pix_x_int, const int32_t* pix_y_int, const int32_t* cam_idx,
                       int num_cams, const float* pixtocams, const float* camtoworlds, const float* distortion6,
                       const float* pixtocam_ndc, int camtype, float* origins, float* directions, float* viewdirs,
                       float* radii, float* imageplane, void* stream);

/* GLO vectors (models.py:101-110,565-568): dst[b*n+i, col0+g] = table[cam_idx[b], g] (bf16), or 0 when
 * cam_idx is NULL (zero_glo=True).  table [num_embeddings, G] fp32 = params['Embed_0']['embedding']. */
int mnr_glo_fill(int64_t B, int n, int G, const float* table, const int32_t* cam_idx, int num_embeddings,
                 uint16_t* dst, int ld, int col0, void* stream);
/* VJP: grad_table[cam_idx[b], g] += sum_i (g_a[b*n+i, g] + g_b[b*n+i, g]); g_a / g_b fp32 [B*n, G] (g_b may be NULL). */
int mnr_glo_bwd(int64_t B, int n, int G, const float* g_a, const float* g_b, const int32_t* cam_idx,
                int num_embeddings, float* grad_table, void* stream);

/* ------------------------------------------------------------------------- *
 * Dense layers  (replaces flax.linen.Dense at models.py:436-437,456,460,495,
 * 515,518,521,527,577,585: y = x @ kernel[in,out] + bias, and their VJPs)
 * bf16 operands, fp32 accumulation on the MFMA units.
 * ------------------------------------------------------------------------- */
typedef struct {
  /* A = [A1 | A2] : [M, K1+K2] bf16; K1,K2 multiples of 64 (of 32 with c_layout = PANEL); A2 may be NULL (K2=0).
   * The second segment is the skip concat of models.py:458-459. */
  const uint16_t* A1; int lda1; int K1;
  const uint16_t* A2; int lda2; int K2;
  const uint16_t* Bt; int ldb;        /* [N, K1+K2] bf16: row n = output column n */
  int64_t M;                          /* multiple of 128 */
  int N;                              /* multiple of 128 (padded rows of Bt are zero) */
  const float* bias; int n_bias;      /* bias[n] added for n < n_bias (may be NULL) */
  int relu;                           /* nn.relu epilogue                          */
  const uint16_t* mask; int ldmask;   /* optional: out *= (mask[m,n] > 0)  (ReLU VJP) */
  uint16_t* Cb; int ldcb; int nb;     /* bf16 output for columns n < nb (may be NULL) */
  float* Cf; int ldcf; int f0; int nf;/* fp32 output for f0 <= n < f0+nf at column n-f0 */
  /* 1-bit ReLU masks, bit (n & 7) of byte [m*ld + n/8]: the forward layer writes (out > 0), the dX
   * GEMM of the next layer reads it instead of re-reading the bf16 activation (16x less traffic). */
  uint8_t* mask_bits_out; int ld_bits_out;
  const uint8_t* mask_bits_in; int ld_bits_in;
  int64_t bits_row_mod;               /* > 0: row m reads the bits of row m % bits_row_mod (tangent rows
                                         c*M + s share the primal mask of sample s) */
  /* Storage of the activation operand A1 and of the bf16 result (MNR_LAYOUT_*).  MNR_LAYOUT_PANEL is the layout the
   * 1024-wide trunk's activations and gradients live in between this library's own GEMMs (every producer and consumer
   * of such a matrix is a kernel of this file; the reference's per-layer tensors models.py:441-465 never leave it):
   * element (m, n) of an [M, W] matrix sits at ((m / 32) * (W / 16) + n / 16) * 512 + (m % 32) * 16 + n % 16, i.e.
   * 1-KiB blocks of 32 rows x 16 columns, which an MFMA wave writes straight from its accumulators as whole blocks
   * (no transposition through LDS) and which the next GEMM's LDS-DMA reads in 512-byte runs.  M %% 256 == 0, W %% 256 == 0.
   * With c_layout = PANEL the 1-bit ReLU masks are in TILE order: 8 KiB per 256 x 256 output tile (tile index
   * m_tile * (N / 256) + n_tile), thread t of the tile's 512 owns 16 bytes; byte (j * 4 + i) * 2 + h of thread
   * t = (wm * 4 + wn) * 64 + kh * 32 + r covers row wm * 128 + i * 32 + r, columns wn * 64 + j * 32 + h * 16 + kh * 8 .. + 7
   * of the tile (bit e = column + e); written by the forward layer, read by the dX layer of the same tile shape
   * (ld_bits_* are ignored).  Restrictions of c_layout = PANEL: K1 + K2 >= 192, no fp32 side output, no bf16 mask,
   * nb == N, forward layers carry a full bias (n_bias == N), bits_row_mod == 0; A2 and Bt are always row-major. */
  int a1_layout; int c_layout;
  /* One more output column supplied as a VECTOR: vcol_out[m] = sum_k [A1|A2][m, k] * vcol[k] + *vcol_bias (fp32), next to an
   * N = 256 result.  The NeRF MLP's density head (internal/models.py:460) next to its 256-wide bottleneck (:527): both read the
   * last trunk activation, and as column 257 of a merged operand the density would cost a second 256-column tile of MFMAs.
   * Each of the four waves that share a row block spends one extra MFMA per k-step on it (+12.5 %).  vcol: bf16 [K1 + K2]
   * (16-byte aligned); needs a1_layout = PANEL, N == 256, K1 + K2 <= 1536, a row-major bf16 result, no fp32 side output. */
  const uint16_t* vcol; float* vcol_out; const float* vcol_bias;     /* vcol_bias: one device float, or NULL for 0 */
  /* c_layout = PANEL: 1 = walk the M-tiles in descending order (the caller alternates it between consecutive layers: a layer
   * then starts with the rows the previous one wrote last).  Performance only; results do not depend on it. */
  int walk_descending;
  int max_wgs;                 /* c_layout = PANEL: > 0 caps the persistent grid (a paired launch shares the chip, see mnr_gemm_tn_args) */
} mnr_gemm_nt_args;
#define MNR_LAYOUT_ROWMAJOR 0
#define MNR_LAYOUT_PANEL 1

/* C[M,N] = epilogue([A1|A2] * Bt^T). */
int mnr_gemm_nt_bf16(const mnr_gemm_nt_args* args, void* stream);

/* ---- Fused Dense chain (csrc/fused_mlp.hip): the trunk of internal/models.py:441-465 (Dense + ReLU layers, optionally
 * one skip concat of the input features, :458-459) and, for a density-only MLP, its Dense(1) head (:460) as ONE
 * persistent kernel per sampling level; replaces depth (+ 1) mnr_gemm_nt_bf16 launches (forward) / the dX chain
 * (backward).  The activation tile of a 256-row block stays in LDS between layers; weights go global -> registers.
 * Widths 128 / 256: the proposal MLP of every config and the 256-wide NeRF MLPs of blender_256 / llff_raw / blender_refnerf. */
#define MNR_CHAIN_MAX_DEPTH 8
typedef struct {
  int64_t M;                  /* rows (samples of the level): a multiple of 256 */
  int W;                      /* layer width: 128 or 256 */
  int depth;                  /* number of Dense + ReLU layers, 1..MNR_CHAIN_MAX_DEPTH */
  const uint16_t* feat;       /* [M, ld_feat] bf16: input of layer 0 (IPE features), columns >= fan_in zero */
  int ld_feat; int K0;        /* K0 = padded fan_in of layer 0, a multiple of 64 */
  const uint16_t* Bt[MNR_CHAIN_MAX_DEPTH];   /* layer i: [W, ldb[i]] bf16 = kernel^T (row = output column) */
  int ldb[MNR_CHAIN_MAX_DEPTH];
  const float* bias[MNR_CHAIN_MAX_DEPTH];    /* [W] fp32 */
  const uint16_t* w_head;     /* [W] bf16 Dense(1) kernel applied to the last activation; NULL: no head */
  const float* b_head;        /* [1] fp32 on the device (NULL: 0) */
  float* head_out;            /* [M] fp32: x_last . w_head + b_head */
  uint16_t* acts[MNR_CHAIN_MAX_DEPTH];       /* optional out: activation of layer i, [M, W] bf16 (training: dW inputs) */
  uint8_t* bits[MNR_CHAIN_MAX_DEPTH];        /* optional out: its ReLU mask, 1 bit per element, [M, W/8] */
  int skip_layer;             /* > 0: layer `skip_layer` reads [x_{skip_layer-1} | feat] (the skip concat of models.py:458-459):
                                 Bt[skip_layer] has W + K0 columns, the feature part behind the W activation columns; 0: none */
} mnr_mlp_chain_fwd_args;
int mnr_mlp_chain_fwd(const mnr_mlp_chain_fwd_args* args, void* stream);

/* The same forward chain with layer 0's A operand PRODUCED IN THE KERNEL (inference: no per-layer outputs, no skip concat):
 * render.cast_rays (render.py:103-127), coord.track_linearize(coord.contract) (coord.py:21-60), lift_and_diagonalize
 * (:129-133) and integrated_pos_enc (:102-126) are evaluated per 256-sample tile straight into the LDS tile the MFMAs read,
 * four encoding degrees (one "group") at a time; the [M, 2KL] feature matrix of mnr_cast_rays_ipe never exists.
 * chain->feat / ld_feat / K0 are unused; acts[i] / bits[i] must be NULL except acts[depth-1]; chain->Bt[0] is layer 0's
 * kernel^T with its K columns REORDERED group-major (L = max_deg - min_deg, K = basis_k, G = MNR_CHAIN_IPE_GROUP_COLS):
 *   column g*G + dl*2K + s*K + k  =  kernel row  s*K*L + (4g+dl)*K + k     (s = 0 sin / 1 cos, dl = 0..3, k < K)
 * and columns [8K, G) of every group zero; chain->ldb[0] >= (L/4)*G.  Needs L % 4 == 0 and K <= 24.  The features are bit-identical to mnr_cast_rays_ipe's rows; only the order
 * of the fp32 MFMA accumulation over layer 0's K differs from mnr_mlp_chain_fwd on that matrix. */
#define MNR_CHAIN_IPE_GROUP_COLS 192
typedef struct {
  mnr_ipe_cfg cfg;
  int n;                       /* samples per ray: row r of the level is sample r % n of ray r / n */
  const float* tdist;          /* [M/n, n+1] */
  const float* origins; const float* directions; const float* radii; const float* basis;   /* as mnr_cast_rays_ipe */
} mnr_chain_ipe_args;
int mnr_mlp_chain_fwd_ipe(const mnr_mlp_chain_fwd_args* chain, const mnr_chain_ipe_args* ipe, void* stream);

typedef struct {
  int64_t M; int W; int depth;
  const float* g_head;        /* [M] fp32: gradient w.r.t. the head output (from mnr_composite_bwd) */
  const float* w_head;        /* [W] fp32 head kernel */
  const uint8_t* bits[MNR_CHAIN_MAX_DEPTH];  /* ReLU masks written by the forward pass */
  const uint16_t* Bw[MNR_CHAIN_MAX_DEPTH];   /* layer i >= 1: [W, ldb[i]] bf16 = kernel as flax stores it (row = input) */
  int ldb[MNR_CHAIN_MAX_DEPTH];
  uint16_t* dY[MNR_CHAIN_MAX_DEPTH];         /* out: gradient w.r.t. layer i's pre-activation, [M, W] bf16
                                                (dY[depth-1] = mask * (g_head (x) w_head) may be NULL: not stored) */
  const uint16_t* dY_in;      /* optional: [M, W] bf16 gradient w.r.t. the LAST layer's pre-activation, already masked (an MLP
                                 with heads: the merged head's dX GEMM wrote it); then g_head / w_head / bits[depth-1] are unused
                                 and dY[depth-1] must be NULL */
} mnr_mlp_chain_bwd_args;
int mnr_mlp_chain_bwd(const mnr_mlp_chain_bwd_args* args, void* stream);

typedef struct {
  const uint16_t* A; int lda; int K;   /* A [M, lda] bf16, K columns used, K multiple of 128 */
  const uint16_t* B; int ldb; int N;   /* B [M, ldb] bf16, N columns used, N multiple of 128 */
  int64_t M;                           /* multiple of 64 */
  float* C; int ldc;                   /* fp32 [k_valid, ldc]: C[k,n] += sum_m A[m,k] B[m,n] */
  int k_valid, n_valid;                /* only k < k_valid, n < n_valid are written */
  float* bias_out; int bias_n_valid;   /* optional: bias_out[n] += sum_m B[m,n] for n < bias_n_valid
                                          (the Dense bias gradient, fused: B is read once) */
  const uint16_t* gcol; float* gcol_out;  /* optional (K, N multiples of 256): one more column of B given as a contiguous bf16
                                          vector [M] (32-byte aligned): gcol_out[k] += sum_m A[m,k] gcol[m] for k < k_valid.  The Dense(1) density head
                                          of the merged NeRF head (models.py:460 next to :527): its gradient column rides in
                                          the bottleneck's weight-gradient GEMM as one extra MFMA per k-step instead of
                                          widening N from 256 to 384 */
  int a_layout, b_layout;              /* MNR_LAYOUT_* of A and B (PANEL: lda == K resp. ldb == N of the whole matrix, K and N
                                          multiples of 256; the 256 x 256 output tile only) */
  /* Pairing with the dX GEMM that reads the same B (= dY) matrix (Model backward, one layer): m_interleave = 1 hands the
   * M-splits the 256-row M-tiles block-cyclically (split s takes M-tiles s, s + splits, ...) so that every split walks M from
   * top to bottom at the pace of a persistent NT launch running next to it; max_wgs > 0 caps the grid (the two launches then
   * share the chip).  Performance only: where M / 256 is not a multiple of the split count the launcher picks
   * (shape- and CU-count-dependent), m_interleave is dropped and the splits are contiguous. */
  int m_interleave, max_wgs;
  /* B given by its factors instead of as a matrix (then B may be NULL; row-major A, K and N multiples of 256):
   *   B[m, n] = bit n of rank1_bits[m, :] ? bf16(rank1_g[m] * rank1_w[n]) : 0
   * the gradient w.r.t. the LAST hidden layer's pre-activation of an MLP whose only head is Dense(1) (the proposal MLP,
   * models.py:460: dY_last = relu'(z_last) * (g_density (x) w_density)), value for value what mnr_mlp_chain_bwd computes and would
   * otherwise have to store for this launch to read back (M x N x 2 bytes each way).  rank1_g: fp32 [M] (4-byte aligned),
   * rank1_w: fp32 [N] (16-byte aligned), rank1_bits: the forward pass's 1-bit ReLU masks, row-major, bit (n & 7) of byte
   * [m * ld_rank1_bits + n / 8] (4-byte aligned, ld_rank1_bits a multiple of 4). */
  const float* rank1_g; const float* rank1_w; const uint8_t* rank1_bits; int ld_rank1_bits;
} mnr_gemm_tn_args;

/* Weight gradient: C += A^T B (fp32 atomics; C must be initialised by the caller). */
int mnr_gemm_tn_bf16(const mnr_gemm_tn_args* args, void* stream);

/* out[n] += sum_m X[m,n] for n < n_valid (bias gradient). X bf16 [M, ld]. */
int mnr_colsum_bf16(const uint16_t* X, int ld, int64_t M, int n_valid, float* out, void* stream);

/* One entry of the weight-packing table: copy an fp32 [rows_in, cols_out] flax
 * kernel (params + src_off) into a bf16 matrix at dst_off, as itself
 * (transpose=0: dst[r*ld + c]) or transposed (transpose=1: dst[c*ld + r]),
 * starting at (row0, col0) of the destination. */
typedef struct {
  int64_t src_off; int rows_in; int cols_out;
  int64_t dst_off; int ld; int row0; int col0; int transpose;
} mnr_pack_desc;
int mnr_pack_weights_bf16(const float* params, const mnr_pack_desc* descs_device, int n_desc,
                          int max_elems, uint16_t* dst, void* stream);

/* dst[k*ld_dst + c] (fp32, flat-gradient layout) += src[(row0+k)*ld_src + col0 + c] for
 * k < rows, c < cols: scatter a (padded / merged) weight-gradient block into
 * the flat gradient vector. */
int mnr_scatter_add_f32(const float* src, int ld_src, int row0, int col0, int rows, int cols,
                        float* dst, int ld_dst, void* stream);

/* fp32 -> bf16 cast of a strided matrix [M, n] (ld_src) into dst [M, ld_dst] at col0. */
int mnr_cast_f32_to_bf16(const float* src, int ld_src, int64_t M, int n, uint16_t* dst, int ld_dst,
                         int col0, void* stream);

/* Non-ReLU MLP.net_activation (models.py:348,457,578; jax.nn.softplus / jax.nn.silu are the ones the reference registers,
 * configs.py:29-31).  The Dense GEMM stores the bf16 pre-activation z [n elements, contiguous]:
 *   mnr_act_fwd_bf16: a = act(z);   mnr_act_bwd_bf16: d *= act'(z) in place (d = gradient w.r.t. a -> w.r.t. z).
 * kind: 1 = softplus, 2 = silu; arithmetic in fp32, one bf16 rounding. */
int mnr_act_fwd_bf16(int kind, int64_t n, const uint16_t* z, uint16_t* a, void* stream);
int mnr_act_bwd_bf16(int kind, int64_t n, const uint16_t* z, uint16_t* d, void* stream);
/* The same activations inside the forward-mode tangent network of the density-gradient normals (models.py:478-492: with a
 * non-ReLU net_activation jax differentiates through act' as well).  z [n = M*W] the layer's primal pre-activation, U [3n] its
 * tangent pre-activation (rows c*M + s = direction c of sample s):
 *   mnr_act_tangent_fwd_bf16: T[3n] = act'(z) * U;
 *   mnr_act_tangent_bwd_bf16: G[3n] (d loss / d T) becomes G * act'(z) in place, and extra[n] = sum_c G_c * U_c * act''(z) =
 *   d loss / d z through act', to be added to the primal backward pass's gradient of that layer. */
int mnr_act_tangent_fwd_bf16(int kind, int64_t n, const uint16_t* z, const uint16_t* U, uint16_t* T, void* stream);
int mnr_act_tangent_bwd_bf16(int kind, int64_t n, const uint16_t* z, const uint16_t* U, uint16_t* G, uint16_t* extra, void* stream);

/* X[m, c] += scale * noise[m, c] (fp32 add, one bf16 rounding) for the first `cols` columns of the bf16 matrix X [M, ld]:
 * the bottleneck noise of models.py:530-533 (`bottleneck += bottleneck_noise * random.normal(...)`) on the
 * bottleneck columns of the view-MLP input.  noise fp32 [M, cols], cols %% 8 == 0. */
int mnr_add_noise_bf16(int64_t M, int cols, uint16_t* X, int ld, const float* noise, float scale, void* stream);

/* Small-N dense VJP pieces (heads with 1..4 outputs, e.g. the rgb Dense(3)):
 *  dX[m,k] = relu'(H[m,k]) * sum_c g[m,c] W[k,c]          (bf16 out)
 *  dW[k,c] += sum_m H[m,k] g[m,c];  db[c] += sum_m g[m,c]  (fp32 atomics)
 * H bf16 [M, ldh] (K columns), g fp32 [M, C], W fp32 [K, C] (flax kernel). */
int mnr_small_head_bwd(int64_t M, int K, int C, const uint16_t* H, int ldh, const float* g,
                       const float* W, uint16_t* dX, int lddx, int apply_relu_mask,
                       float* dW, float* db,
                       const uint8_t* mask_bits /* optional 1-bit mask [*, ld_bits], row m %% bits_row_mod */,
                       int ld_bits, int64_t bits_row_mod,
                       float* scratch /* optional workspace for per-workgroup dW/db partials (else fp32 atomics) */,
                       int64_t scratch_floats, void* stream);

/* ------------------------------------------------------------------------- *
 * Compositing  (replaces the density/rgb activations models.py:506,584-602,
 * render.compute_alpha_weights render.py:130-151, render.volumetric_rendering
 * render.py:154-213 and stepfun.weighted_percentile stepfun.py:298-308)
 * ------------------------------------------------------------------------- */
typedef enum { MNR_ACT_SIGMOID = 0, MNR_ACT_SAFE_EXP = 1, MNR_ACT_SOFTPLUS = 2, MNR_ACT_EXP = 3,
               MNR_ACT_RELU = 4 } mnr_act;

typedef struct {
  int n;                   /* intervals per ray                                      */
  int opaque_background;   /* Model.opaque_background                                */
  int density_act;         /* mnr_act (softplus default)                             */
  float density_bias;      /* MLP.density_bias                                       */
  float density_noise_std; /* MLP.density_noise (0: none)                            */
  int has_rgb;             /* 0: MLP.disable_rgb (rgb = 0)                           */
  int rgb_act;             /* mnr_act                                                */
  float rgb_premultiplier, rgb_bias, rgb_padding;
  int bg_mode;             /* 0: scalar bg_value; 1: per-ray bg [B,3]                */
  float bg_value;
} mnr_composite_cfg;

/* raw_density [B,n] (Dense(1) output incl. its bias), density_noise [B,n] or
 * NULL, raw_rgb [B,n,3] or NULL, tdist [B,n+1], dirs [B,3], bg [B,3] or NULL,
 * exposure_scale [B,3] or NULL (RawNeRF models.py:257-267, already combined).
 * Outputs: density [B,n], rgb [B,n,3] (NULL if !has_rgb), weights [B,n],
 * rgb_out [B,3], acc [B]. */
int mnr_composite_fwd(const mnr_composite_cfg* cfg, int64_t B, const float* raw_density,
                      const float* density_noise, const float* raw_rgb, const float* tdist,
                      const float* dirs, const float* bg, const float* exposure_scale,
                      float* density, float* rgb, float* weights, float* rgb_out, float* acc,
                      void* stream);

/* VJP of mnr_composite_fwd.  g_rgb_out [B,3] or NULL, g_weights [B,n] or NULL
 * (from the interlevel / distortion losses).  Outputs: g_raw_density [B,n]
 * fp32 (and, if g_raw_density_bf16 != NULL, the same value as bf16 at
 * g_raw_density_bf16[row*ld_bf16], the density column of the head-gradient
 * matrix), g_raw_rgb [B,n,3] or NULL. */
int mnr_composite_bwd(const mnr_composite_cfg* cfg, int64_t B, const float* raw_density,
                      const float* density_noise, const float* raw_rgb, const float* tdist,
                      const float* dirs, const float* bg, const float* exposure_scale,
                      const float* weights, const float* g_rgb_out, const float* g_weights,
                      float* g_raw_density, uint16_t* g_raw_density_bf16, int ld_bf16,
                      float* g_raw_rgb, float* g_exposure_scale /* [B,3] +=, may be NULL */, void* stream);

/* One level's backward pass in one launch: the training losses acting on this level's rendering with their gradients
 * (train_utils.py:72-159: compute_data_loss on the composited colour; interlevel_loss of a proposal level against the
 * final level's histogram, stepfun.py:64-86; distortion_loss of the final level, stepfun.py:266-276) fused with the
 * compositing VJP (render.py:130-213).  d loss / d weights and d loss / d rgb never touch HBM.  Replaces one
 * mnr_data_loss + mnr_interlevel_loss | mnr_distortion_loss + mnr_composite_bwd sequence; mnr_composite_bwd is this
 * entry with both losses off. */
typedef struct {
  mnr_composite_cfg cfg;
  int64_t B, B_valid;           /* rays (padded), rays that take part in the losses */
  /* compositing VJP: as mnr_composite_bwd */
  const float* raw_density; const float* density_noise; const float* raw_rgb; const float* tdist; const float* dirs;
  const float* bg; const float* exposure_scale; const float* weights;
  const float* g_rgb_out;       /* optional upstream [B,3], added to the fused data-loss gradient */
  const float* g_weights;       /* optional upstream [B,n] (Ref-NeRF normal losses), added to the fused weight loss's */
  float* g_raw_density; uint16_t* g_raw_density_bf16; int ld_bf16; float* g_raw_rgb; float* g_exposure_scale;
  /* data loss; data_loss_type < 0: off.  data_stats[0] += weighted mse, [1] += data_loss_mult * loss (both / *denom) */
  int data_loss_type; float charb_padding; float data_loss_mult;
  const float* rgb_out; const float* gt; const float* lossmult; int lm_c; const float* denom; float* data_stats;
  /* loss on the weights: 0 none; 1 interlevel, this level = envelope of (t_ref [B,n_ref+1], w_ref [B,n_ref]);
   * 2 distortion on this level's own histogram.  sdist [B,n+1]: this level's normalised distances.  *wloss_stat += loss */
  int wloss_mode; float wloss_mult; const float* sdist; int n_ref; const float* t_ref; const float* w_ref; float* wloss_stat;
  /* optional output [B,n] fp32: d loss / d (sigma_i * delta_i), the optical-depth increments of render.py:144-145 -- what
   * Model.stop_level_grad = False needs to carry the compositing's gradient on to the sample distances (mnr_sdist_bwd) */
  float* g_x;
} mnr_level_bwd_args;
int mnr_level_bwd(const mnr_level_bwd_args* args, void* stream);

/* Model.stop_level_grad = False (models.py:56,198-201): d loss / d sdist [B,n+1] of one level, gathered per fence-post from
 *   g_x [B,n] (mnr_level_bwd: the optical-depth increments sigma_i * (t_{i+1} - t_i) * |d|, render.py:144-145; needs raw_density
 *     [B,n] with density_noise / density_noise_std / density_bias / density_act as in mnr_composite_cfg, and dirs [B,3]),
 *   g_t0, g_t1 [B*n] (mnr_cast_rays_ipe_bwd: the Gaussians' dependence on the interval ends, render.py:103-127),
 *     both through s_to_t' (coord.py:96-98; raydist_fn, near, far [B], sdist [B,n+1]),
 *   the distortion loss's own dependence on sdist (stepfun.py:266-276; distortion_mult != 0: weights [B,n], mean over B_valid rays),
 *   g_sdist_in [B,n+1]: what the NEXT level's resampling sends back (mnr_resample_level_bwd).
 * Every input group is optional (NULL / 0).  The interlevel loss is piecewise constant in sdist (stepfun.py:64-77). */
typedef struct {
  int64_t B, B_valid; int n;
  const float* sdist; const float* near; const float* far; int raydist_fn;
  const float* g_x; const float* raw_density; const float* density_noise; float density_noise_std; float density_bias;
  int density_act; const float* dirs;
  const float* g_t0; const float* g_t1;
  float distortion_mult; const float* weights;
  const float* g_sdist_in;
  float* g_sdist;
} mnr_sdist_bwd_args;
int mnr_sdist_bwd(const mnr_sdist_bwd_args* args, void* stream);

/* RawNeRF exposure (replaces models.py:257-267).  out[b,c] = exposure_values[b] *
 * (1 + [idx[b] > 0] * offsets[idx[b], c]); offsets = the 'exposure_scaling_offsets' embedding
 * [num_embeddings,3] or NULL when Model.learned_exposure_scaling is off.  The backward scatters
 * g_offsets[idx[b], c] += exposure_values[b] * g_scale[b, c] for idx[b] > 0. */
int mnr_exposure_scale(int64_t B, const float* exposure_values, const int32_t* exposure_idx,
                       const float* offsets, float* out, void* stream);
int mnr_exposure_scale_bwd(int64_t B_valid, const float* exposure_values, const int32_t* exposure_idx,
                           const float* g_scale, float* g_offsets, void* stream);

/* The compute_extras outputs of render.volumetric_rendering (render.py:184-211):
 * distance_mean, distance_percentile_5 / median / percentile_95 -> out [B,4]. */
int mnr_render_extras(int64_t B, int n, const float* weights, const float* tdist, const float* t_far,
                      float* out, void* stream);

/* ------------------------------------------------------------------------- *
 * Ref-NeRF branch  (replaces models.py:478-503,512-523,540-563,588-602, ref_utils.py:22-42,99-159,
 * image.py:48-56, train_utils.py:162-197).  Merged head column layout, bw = bottleneck width:
 *   [0,bw) bottleneck | bw density | bw+1..3 grad_pred | bw+4..6 raw diffuse | bw+7..9 raw tint | bw+10 raw roughness
 * `small` [M,11] fp32 = columns bw..bw+10 of the head GEMM; `raw_grad` [3,M] fp32 = d raw_density/d mean.
 * ------------------------------------------------------------------------- */
typedef struct {
  int T; int lmax;            /* number of (m,l) terms; largest degree (<= 16) */
  const int32_t* m; const int32_t* l;   /* [T] device arrays (ref_utils.get_ml_array) */
  const float* sigma;         /* [T] 0.5 l (l+1) */
  const float* mat;           /* [lmax+1, T] z-polynomial coefficients (ref_utils.py:119-125) */
} mnr_ide_tables;

/* Which parts of the branch an MLP has (models.py:468-563 takes every flag on its own; a head that is switched off is a zero
 * column of `small` nobody reads).  What the reference itself cannot run is refused with its own message where it has one:
 * reflections or n.v without a normal field (models.py:434-435), the IDE without the predicted roughness or on the per-ray
 * view direction (ref_utils.py:147-154: `kappa_inv` None / [B,36] against [B,n,1]). */
#define MNR_REF_PRED_NORMALS    1   /* enable_pred_normals: grad_pred columns, normals_to_use = normals_pred (:494-499) */
#define MNR_REF_DENSITY_NORMALS 2   /* disable_density_normals = False: raw_grad, normals (:478-492) */
#define MNR_REF_REFLECT         4   /* use_reflections (:540-547); else the view direction is encoded (:549-555) */
#define MNR_REF_IDE             8   /* use_directional_enc (:436-437); else coord.pos_enc(dir, 0, deg_view, True) (:438-441) */
#define MNR_REF_N_DOT_V        16   /* use_n_dot_v (:560-563) */
#define MNR_REF_ROUGHNESS      32   /* enable_pred_roughness (:520-523) */

/* normals = -l2norm(raw_grad), normals_pred = -l2norm(grad_pred), roughness = softplus(raw + bias),
 * dir = reflect(-viewdirs, normals_to_use) or viewdirs, IDE(dir, roughness) or pos_enc(dir) and n.v written (bf16) into
 * columns [col0, col0 + E (+ 1)) of every row of `vi` [M, ldvi], E = 2T or 3 + 6 deg_view; columns up to col_end
 * zero-filled.  Outputs (and `raw_grad`, `tabs`) of parts that are off may be NULL. */
int mnr_ref_head_fwd(int64_t M, int n, const float* small, const float* raw_grad, const float* viewdirs,
                     const mnr_ide_tables* tabs, int features, int deg_view, float roughness_bias, uint16_t* vi, int ldvi,
                     int col0, int col_end, float* normals_out, float* normals_pred_out, float* roughness_out, void* stream);
/* VJP: dvi_a (+ dvi_b, may be NULL) bf16 [M, lddvi] = gradient w.r.t. the view-MLP input; g_npred / g_n
 * [M,3] fp32 from mnr_ref_losses (may be NULL).  Writes bf16 columns [0,col0) (bottleneck = dvi_a + dvi_b),
 * col_gp..+2 (predicted normals) and col_rough (roughness) of dhb [M, lddhb] and g_raw_grad [3,M] fp32 (density normals). */
int mnr_ref_head_bwd(int64_t M, int n, const float* small, const float* raw_grad, const float* viewdirs,
                     const mnr_ide_tables* tabs, int features, int deg_view, float roughness_bias, const uint16_t* dvi_a,
                     const uint16_t* dvi_b, int lddvi, int col0, const float* g_npred, const float* g_n,
                     uint16_t* dhb, int lddhb, int col_gp, int col_rough, float* g_raw_grad, void* stream);
/* dst[:, :cols] = a[:, :cols] + b[:, :cols] (bf16, b may be NULL; dst may alias a): gradient joins of skip concats. */
int mnr_add_cols_bf16(int64_t M, int cols, const uint16_t* a, int lda, const uint16_t* b, int ldb, uint16_t* dst,
                      int lddst, void* stream);
/* rgb = clip(linear_to_srgb(tint * sigmoid(premult raw_rgb + bias) + sigmoid(raw_diffuse - log 3)), 0, 1)
 * * (1 + 2 pad) - pad; VJP writes g_raw_rgb [M,3] fp32 and the diffuse / tint columns of dhb (bf16). */
int mnr_ref_color_fwd(int64_t M, const float* raw_rgb, const float* small, float rgb_premultiplier, float rgb_bias,
                      float rgb_padding, int use_tint, float* rgb_out, void* stream);
int mnr_ref_color_bwd(int64_t M, const float* raw_rgb, const float* small, float rgb_premultiplier, float rgb_bias,
                      float rgb_padding, int use_tint, const float* g_rgb, float* g_raw_rgb, uint16_t* dhb,
                      int lddhb, int col_diffuse, int col_tint, void* stream);
/* One level of orientation_loss + predicted_normal_loss: stats[0] += mult_o * mean_rays sum_i w min(0, n.(-v))^2,
 * stats[1] += mult_p * mean_rays sum_i w (1 - n.n_pred); g_weights [B,n] +=, g_normals / g_normals_pred [M,3] =. */
int mnr_ref_losses(int64_t B_valid, int n, float mult_orientation, float mult_pred_normal, int target_is_pred,
                   const float* weights, const float* normals, const float* normals_pred, const float* viewdirs,
                   float* stats, float* g_weights, float* g_normals, float* g_normals_pred, void* stream);
/* out[b,c] = sum_i weights[b,i] values[b,i,c]  (render.py:187-190 extras). */
/* Predicted normals without the rest of the Ref-NeRF head (internal/models.py:494-503 with enable_pred_normals only):
 * normals_pred[M,3] = -l2_normalize(small[:, col .. col+2]) (ref_utils.py:40-42) from the head GEMM's fp32 side output [M, ld];
 * _bwd: the VJP of the same into columns col_g .. col_g+2 of the head's bf16 gradient matrix [M, lddhb].
 * mnr_ref_losses takes normals = NULL for such an MLP (mult_pred_normal == 0, target_is_pred = 1, g_normals = NULL). */
/* Density-gradient normals without the rest of the Ref-NeRF head (internal/models.py:478-492 with disable_density_normals = False
 * only): normals[M,3] = -l2_normalize(raw_grad) from the tangent network's raw_grad [3, M] (component-major); _bwd: g_raw_grad [3, M].
 * mnr_ref_losses takes normals_pred = NULL for such an MLP (mult_pred_normal == 0, target_is_pred = 0, g_normals_pred = NULL). */
int mnr_density_normals_fwd(int64_t M, const float* raw_grad, float* normals_out, void* stream);
int mnr_density_normals_bwd(int64_t M, const float* raw_grad, const float* g_normals, float* g_raw_grad, void* stream);
int mnr_pred_normals_fwd(int64_t M, const float* small, int ld, int col, float* normals_pred_out, void* stream);
int mnr_pred_normals_bwd(int64_t M, const float* small, int ld, int col, const float* g_normals_pred, uint16_t* dhb, int lddhb,
                         int col_g, void* stream);
int mnr_weighted_sum(int64_t B, int n, int C, const float* weights, const float* values, float* out, void* stream);

/* ------------------------------------------------------------------------- *
 * Losses  (replaces train_utils.compute_data_loss train_utils.py:72-136,
 * interlevel_loss :139-150 with stepfun.lossfun_outer stepfun.py:30-86,
 * distortion_loss :153-159 with stepfun.lossfun_distortion stepfun.py:266-276)
 * Scalars accumulate into `stats` (fp32 device array, zeroed by the caller).
 * ------------------------------------------------------------------------- */
typedef enum { MNR_LOSS_MSE = 0, MNR_LOSS_CHARB = 1, MNR_LOSS_RAWNERF = 2 } mnr_data_loss_type;

/* out[0] += sum(lossmult broadcast to [B,3]) over the first B_valid rays. lossmult [B,lm_c], lm_c in {1,3}. */
int mnr_lossmult_sum(int64_t B_valid, const float* lossmult, int lm_c, float* out, void* stream);

/* compute_data_loss's gradient-free metrics (train_utils.py:113-128), either may be NULL:
 * *out_disp = mean (1/(1+distance_mean) - disps)^2;  *out_normal = weighted mean angular error in degrees
 * (ref_utils.compute_weighted_mae ref_utils.py:45-50) with weights acc * alphas. */
int mnr_render_metrics(int64_t B_valid, const float* distance_mean, const float* disps, const float* acc,
                       const float* alphas, const float* normals, const float* normals_gt, float* out_disp,
                       float* out_normal, void* stream);

/* One level of compute_data_loss.  rgb [B,3] rendered, gt [B,3], denom = device
 * scalar from mnr_lossmult_sum.  stats[0] += mse numerator/denom, stats[1] +=
 * loss_mult * data loss.  g_rgb [B,3] (may be NULL) = d(loss_mult*loss)/d rgb. */
int mnr_data_loss(int loss_type, float charb_padding, float loss_mult, int64_t B, int64_t B_valid,
                  const float* rgb, const float* gt, const float* lossmult, int lm_c,
                  const float* denom, float* stats, float* g_rgb, void* stream);

/* interlevel: t [B,n+1], w [B,n] (final level, constants); t_env [B,ne+1], w_env
 * [B,ne] (proposal).  stats[0] += mult * mean(lossfun_outer); g_w_env [B,ne]
 * (+=, may be NULL) = d/d w_env. */
int mnr_interlevel_loss(float mult, int64_t B, int64_t B_valid, int n, const float* t, const float* w,
                        int ne, const float* t_env, const float* w_env, float* stats, float* g_w_env,
                        void* stream);

/* distortion on (t = sdist, w): stats[0] += mult * mean(lossfun_distortion); g_w [B,n] += d/d w. */
int mnr_distortion_loss(float mult, int64_t B, int64_t B_valid, int n, const float* t, const float* w,
                        float* stats, float* g_w, void* stream);

/* Leaf for parity: per-ray lossfun_outer [B,n] and lossfun_distortion [B]. */
int mnr_lossfun_outer(int64_t B, int n, const float* t, const float* w, int ne, const float* t_env,
                      const float* w_env, float* out, void* stream);
int mnr_lossfun_distortion(int64_t B, int n, const float* t, const float* w, float* out, void* stream);

/* ------------------------------------------------------------------------- *
 * Optimiser  (replaces train_utils.clip_gradients train_utils.py:200-218,
 * jnp.nan_to_num :328 and optax.adam state.apply_gradients :330,372)
 * ------------------------------------------------------------------------- */
/* Weight regulariser of one top-level module (train_utils.py:300-305, Config.weight_decay_mults):
 * *loss_out += mult * sum(params[begin:end]^2); grad[begin:end] += 2 mult params; *sqnorm_out += sum p^2
 * (stats['weight_l2s']).  grad / loss_out / sqnorm_out may be NULL. */
int mnr_weight_decay(const float* params, int64_t begin, int64_t end, float mult, float* grad, float* loss_out,
                     float* sqnorm_out, void* stream);

/* out[0] += sum of squares of grad[begin:end] (each element first clipped to
 * +-max_val when max_val > 0): one call per top-level module. */
int mnr_grad_sqnorm(const float* grad, int64_t begin, int64_t end, float max_val, float* out, void* stream);

typedef struct {
  float lr, b1, b2, eps;
  float bias_corr1, bias_corr2;  /* 1 - b1^t, 1 - b2^t                          */
  float grad_max_val;            /* 0: off                                      */
  float grad_max_norm;           /* 0: off                                      */
} mnr_adam_cfg;

/* For one segment [begin,end): g = clip_value(g); g *= min(1, max_norm/(eps32+sqrt(sqnorm[seg])));
 * g = nan_to_num(g); Adam moments and parameter update in place. */
int mnr_clip_adam(const mnr_adam_cfg* cfg, int64_t begin, int64_t end, const float* sqnorm_seg,
                  const float* grad, float* params, float* mu, float* nu, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MNERF_H_ */
