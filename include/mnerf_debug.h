/* libmnerf_hip.so -- development and test hooks.  NOT part of the product ABI (include/mnerf.h): nothing in multinerf_amd/
 * calls these; tests/ and tools/ bind them through multinerf_amd._lib.debug().  They are PROCESS-GLOBAL switches and stamps:
 * they exist to A/B a kernel variant on one box, to force small persistent grids in tests, and to read per-workgroup timelines.
 * A deployment never needs them, and a multi-threaded host must not touch them while launches are in flight. */
#ifndef MNERF_DEBUG_H_
#define MNERF_DEBUG_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Profiling hook: device buffer of 16 uint64 per workgroup (s_memtime at entry, K-loop start, K-loop end, exit;
 * s_memrealtime at entry, exit; XCC_ID<<32|HW_ID; unused; epilogue pass stamps) written by
 * every following mnr_gemm_nt_bf16 / mnr_gemm_tn_bf16 launch (TN: [7] = steps << 32 | does-bias); NULL switches it off. */
int mnr_debug_gemm_timeline(unsigned long long* device_buffer);
/* A/B switch: 1 (default) = the 256x256 tiles use the hand-pipelined K loop (BK = 32 x 4 stages, LDS-DMA issued between the
   MFMAs), 0 = the two-stage BK = 64 loop.  Bitwise equal results. */
int mnr_gemm_nt_set_pipelined(int on);
/* A/B switch: 0 = one workgroup per output tile; n > 0 (default 1) = persistent launches, n workgroups per CU walk the
 * tiles; n < 0 = at most -n workgroups in total. */
int mnr_gemm_nt_set_persistent(int wgs_per_cu);
/* A/B switch: 1 (default) = eligible short-K launches (N = 256, K1 <= 256, K2 = 0, full-width bf16 output, no fp32 side
 * output, no bf16 mask) go to the weights-resident persistent kernel (weights in registers, one workgroup per CU walking
 * the M tiles); n > 1 = the same with at most n workgroups; 0 = off. */
int mnr_gemm_nt_set_wres(int max_wgs);
/* Test hook of the panel-result kernel (c_layout = MNR_LAYOUT_PANEL, csrc/gemm_blk.hip): at most n persistent workgroups
 * (every workgroup then walks several tiles at small sizes); 0 (default) = one per CU. */
int mnr_gemm_nt_panel_set_max_wgs(int n);

/* A/B switch of both chain kernels: 1 (default) = a layer's copy-out (activation / gradient rows, mask bits) is issued from
 * inside the NEXT layer's MFMA pass, behind that pass's last weight request; 0 = in front of the pass.  Bitwise equal. */
int mnr_mlp_chain_set_deferred(int on);

/* Test hook of the chain kernels: at most n persistent workgroups (every workgroup then walks several tiles at small sizes);
 * 0 (default) = one per CU. */
int mnr_mlp_chain_set_max_wgs(int n);

/* Profiling hook: device buffer of 32 uint64 per workgroup, stamped (s_memtime per phase of the workgroup's second tile,
 * see csrc/fused_mlp.hip) by every following mnr_mlp_chain_fwd / _bwd launch; NULL switches it off. */
int mnr_debug_chain_timeline(unsigned long long* device_buffer);

/* A/B switch: 1 (default) = four lanes per ray where a wave's 16 rays fit LDS, 0 = the lane-per-ray kernel everywhere.
 * Sums are associated differently in the two; both are held to the oracle by the same tolerances. */
int mnr_level_bwd_set_quad(int on);

#ifdef __cplusplus
}
#endif
#endif  /* MNERF_DEBUG_H_ */
