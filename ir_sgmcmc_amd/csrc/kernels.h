// Host-side launchers of the gfx950 kernels (internal; the public surface is include/irsgmcmc.h).
#pragma once
#include "common.h"

namespace irs {

struct Taps {
    float k[2 * IRS_MAX_HALF_WIDTH + 1];
    int s;
};

struct SplineTaps {
    float k[32];  // sampled cubic B-spline, 4*cps - 1 taps (cps <= 8)
    int cps;
};

// identity-grid tables live in device memory; built once per (D,H,W)
struct LinTables {
    float* dev = nullptr;  // [W | H | D]
    int D = 0, H = 0, W = 0;
    Lin lin() const { return Lin{dev, dev + W, dev + W + H}; }
};
int ensure_lin_tables(LinTables& t, int D, int H, int W, hipStream_t st);
// process-wide cache used by the stateless operators
int cached_lin(int D, int H, int W, hipStream_t st, Lin* out);

// ---- field_kernels.hip
void launch_perturb(const float* v, const float* sigma, const float* eps, float amp, float* out, int C, Vol vol,
                    uint64_t seed, uint64_t iteration, const uint64_t* dev_iteration, hipStream_t st);
void launch_svf_outputs(const float* d, float* transformation, float* displacement, int C, Vol vol, Lin lin,
                        hipStream_t st);
void launch_warp_fwd(const float* im, int64_t im_stride, const float* d, const float* unif, float alpha, float* out,
                     float* gradm, int gradm_aos, int C, Vol vol, Lin lin, uint64_t seed, uint64_t iteration,
                     const uint64_t* dev_iteration, hipStream_t st);
void launch_warp_bwd(const float* im, int64_t im_stride, const float* d, const float* unif, float alpha,
                     const float* g_warped, float* g_d, int C, Vol vol, Lin lin, uint64_t seed, uint64_t iteration,
                     const uint64_t* dev_iteration, hipStream_t st);
void launch_warp_transformation(const float* im, int64_t im_stride, const float* t, float* out, int C, Vol vol,
                                hipStream_t st);
void launch_warp_nearest_u8(const uint8_t* im, int64_t im_stride, const float* t, uint8_t* out, int C, Vol vol,
                            hipStream_t st);
void launch_warp_nearest_i16(const int16_t* im, int64_t im_stride, const float* t, int16_t* out, int C, Vol vol,
                             hipStream_t st);
// rows [w_lo, w_lo + w_n) of the dense axis produced (up) / summed (adjoint); the windowed array stores rows
// [store_lo, store_lo + store_n) of that axis.  w_n < 0 / store_n < 0: the whole axis.
void launch_ffd_axis(const float* in, float* out, const SplineTaps& taps, bool adjoint, int64_t outer, int n_in,
                     int n_out, int64_t inner, hipStream_t st, int w_lo = 0, int w_n = -1, int store_lo = 0, int store_n = -1);
void launch_scale_channels(const float* in, float* out, float s0, float s1, float s2, int C, Vol vol, hipStream_t st);

// ---- exp_kernels.hip (z-marching squaring step + owner-computes gather adjoint, LDS-scatter fallback)
// dmax_in: published bound of the input field (nullptr = unknown), dmax_out: receives the bound of the output field
void launch_exp_step_fwd_march(const float* din, float* dout, bool prescale, int no_steps, int C, Vol vol, Lin lin,
                               const unsigned* dmax_in, unsigned* dmax_out, bool only_r1, int lay, hipStream_t st);
// the adjoint is launched as a set: gather radius 1, gather radius 2 and the scatter fallback; exactly one of them does
// the work, chosen on the device from the bound max|d_k|
void launch_exp_step_bwd_march(const float* G, const float* dk, float* gout, bool prescale, int no_steps, int C, Vol vol,
                               Lin lin, const unsigned* dmax, int max_radius, bool r2_owns_rest, const float* gscale, int lay,
                               hipEvent_t after_primary, hipStream_t st);
// cmm: scratch of coarse_minmax_bytes(vol, C) for the per-cell displacement extrema (nullptr: sources are bounded by the
// global bound around the tile only -- correct, slow for large displacements)
size_t coarse_minmax_bytes(Vol vol, int C);
void launch_exp_step_bwd_lds(const float* G, const float* dk, float* gout, bool prescale, int no_steps, int C, Vol vol,
                             Lin lin, const unsigned* dmax, int halo, int gather_radius, const float* gscale, int lay,
                             float* cmm, hipStream_t st);
void launch_field_absmax(const float* d, bool prescale, int no_steps, unsigned* dmax, int C, Vol vol, hipStream_t st);

// ---- data_kernels.hip
void launch_sobolev_march(const float* in, float* out, const Taps& taps, int planes, Vol vol, unsigned* dmax0, int no_steps,
                          hipStream_t st);  // z-marching version (stencil_kernels.hip)
// SGLD perturbation generated while the smoothing kernel stages its planes (SVF_3D path, s > 0): v + noise is never materialised
void launch_perturb_sobolev_march(const float* v, const float* sigma, const float* eps, float amp, float* out, const Taps& taps,
                                  int C, Vol vol, unsigned* dmax0, int no_steps, uint64_t seed, uint64_t iteration,
                                  const uint64_t* dev_iteration, hipStream_t st);  // stencil_kernels.hip
void launch_lcc_fwd_march(const float* fhat, int64_t fhat_stride, const float* im, float* z, float* sigma_out, int s, int C,
                    Vol vol, hipStream_t st);  // z-marching version (stencil_kernels.hip)
struct GmmDev;  // device-side mixture parameters (scalar_kernels.hip)
// data term + its gradient w.r.t. the warped image (LCC adjoint fused); mode: IRS_DATA_*.  C_launch > 1 (GMM / LCC only): that many
// chains from `chain` on in one launch -- z, sigma_m, g_warped and the partial sums at their chain offsets, fhat / mask every
// f_stride / mask_stride elements, the mixture of each chain from its snapshot (scalar_kernels.h: DevState::snapA)
void launch_data_bwd(int mode, const float* fhat_or_fixed, int64_t f_stride, const float* z, const float* sigma_m,
                     const uint8_t* mask, int64_t mask_stride, const float* g_z_override, const void* dev_state,
                     int chain, float* g_warped, double* nll_partials, int s, int C_launch, Vol vol, hipStream_t st, int seg_C = 1);
// K: number of mixture components if the caller knows it (selects the K <= 4 build of the kernel), 0 = unknown
void launch_stats(int want_vd, const float* z, const uint8_t* mask, const void* dev_state, double* partials, Vol vol,
                  hipStream_t st, int K = 0);
void launch_residual_ssd(const float* fixed, int64_t f_stride, const float* warped, float* z, int C, Vol vol,
                         hipStream_t st);
// z-marching fused data-term backward (stencil_kernels.hip)
// seg_C: chains the segment length is chosen for (1: a launch per chain; C: all chains in one launch -- longer segments, one resident set)
int lcc_data_bwd_march_blocks(Vol vol, int seg_C = 1);
void launch_lcc_data_bwd_march(const float* fhat, const float* z, const float* sigma_m, const uint8_t* mask,
                               const float* g_z_override, const void* dev_state, int chain, float* g_warped,
                               double* nll_partials, int s, Vol vol, hipStream_t st, int batch = 0, int64_t f_stride = 0,
                               int64_t m_stride = 0, int seg_C = 1);  // batch > 0: chains chain .. chain + batch - 1 in one launch, each against its snapshot
int data_bwd_blocks(int mode, Vol vol, int seg_C = 1);
void launch_masked_moments(const float* z, const uint8_t* mask, double* partials, Vol vol, hipStream_t st);
void launch_reg_energy(const float* v, double* partials, int C, Vol vol, hipStream_t st);
void launch_reduce_partials(const double* partials, int nblocks, int nvals, double* out, hipStream_t st);
// out[j] = sum_b partials[b * ncols + j]
void launch_reduce_cols(const double* partials, int nblocks, int ncols, double* out, hipStream_t st);
// energy_partials (optional, [C][sgld_update_blocks_per_chain]): regulariser energy of v_s as a by-product; coef_from_w: the
// regulariser coefficient is w / 2 from the state (L2 family) instead of the one reg_scalar_kernel left in the state
void launch_sgld_update(float* v, const float* sigma, const float* g_d0, const float* v_s, const void* dev_state,
                        float lr, float s0, float s1, float s2, float* grad_out, int C, Vol vol, hipStream_t st,
                        double* energy_partials = nullptr, bool coef_from_w = false);
int sgld_update_blocks_per_chain(Vol vol, int C);
void launch_gradient_operator(const float* v, float* nabla, int transformation, int C, Vol vol, hipStream_t st);
void launch_log_det_jacobian(const float* t, float* log_det, long long* nan_count, int C, Vol vol, hipStream_t st);
void launch_stats_march(int want_vd, const float* z, const uint8_t* mask, const void* dev_state, double* partials, int blocks,
                        Vol vol, int K, hipStream_t st);  // stencil_kernels.hip
void launch_reg_energy_march(const float* v, double* partials, int blocks, int C, Vol vol, hipStream_t st);
void launch_sgld_update_march(float* v, const float* sigma, const float* g_d0, const float* v_s, const void* dev_state,
                              float lr, float s0, float s1, float s2, float* grad_out, int C, Vol vol, hipStream_t st,
                              double* energy_partials = nullptr, bool coef_from_w = false);
int stats_blocks(Vol vol);
int energy_blocks(Vol vol);

// ---- scalar_kernels.hip
struct DevState;  // full definition in scalar_kernels.h
}  // namespace irs
