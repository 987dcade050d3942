// gfx950 kernels for the similarity / regulariser / update half of the SG-MCMC transition:
//   (the LCC map, its adjoint and the Sobolev smoothing moved to stencil_kernels.hip)
//   mixture statistics for virtual decimation + the GMM step (utils/util.py:330-347,446-485; trainer.py:68-77)
//   data term and d/dz with the updated mixture              (model/loss.py:87-100)
//   regulariser energy, its adjoint stencil and the SGLD/SGD update (utils/diff_op.py:78-96; model/loss.py:152-161;
//                                                             utils/functions.py:83-84; trainer.py:349-351)
#include "kernels.h"
#include "scalar_kernels.h"

namespace irs {

// (The LCC map, its fused adjoint / data term and the Sobolev smoothing live in stencil_kernels.hip.)

// SSD (builder-defined): z = F - M,  g_M = -alpha * mask * z / sigma^2,  nll = 0.5 mask (z / sigma)^2
__global__ __launch_bounds__(kBlock) void ssd_bwd_kernel(const float* __restrict__ z, const uint8_t* __restrict__ mask,
                                                         const DevState* __restrict__ state, int chain,
                                                         float* __restrict__ g_m, double* __restrict__ nll_out, Vol vol) {
    __shared__ double smem[kBlock / kWave];
    const float alpha = (float)state->sc.alpha[chain];
    const float is = state->ssd_inv_sigma;
    double acc[1] = {0.0};
    IRS_ROWS_BEGIN(vol, x, y, zc, v)
        (void)y; (void)zc;
        float g = 0.0f;
        if (mask[v]) {
            const float u = z[v] * is;
            acc[0] += 0.5 * (double)(u * u);
            g = -alpha * u * is;
        }
        g_m[v] = g;
    IRS_ROWS_END
    block_sum<1>(acc, smem);
    if (threadIdx.x == 0) nll_out[blockIdx.x] = acc[0];
}

int data_bwd_blocks(int mode, Vol vol, int seg_C) {
    if (mode == IRS_DATA_SSD) return stats_blocks(vol);
    return lcc_data_bwd_march_blocks(vol, seg_C);
}

void launch_data_bwd(int mode, const float* fhat_or_fixed, int64_t f_stride, const float* z, const float* sigma_m,
                     const uint8_t* mask, int64_t mask_stride, const float* g_z_override, const void* dev_state,
                     int chain, float* g_warped, double* nll_partials, int s, int C_launch, Vol vol, hipStream_t st, int seg_C) {
    const DevState* state = (const DevState*)dev_state;
    if (mode == IRS_DATA_SSD) {
        hipLaunchKernelGGL(ssd_bwd_kernel, dim3(stats_blocks(vol)), dim3(kBlock), 0, st, z, mask, state, chain, g_warped,
                           nll_partials, vol);
        return;
    }
    launch_lcc_data_bwd_march(fhat_or_fixed, z, sigma_m, mask, g_z_override, dev_state, chain, g_warped, nll_partials, s, vol, st,
                              C_launch > 1 ? C_launch : 0, f_stride, mask_stride, seg_C);
}

int stats_blocks(Vol vol) {
    const int64_t b = (win_voxels(vol) + kBlock - 1) / kBlock;
    return (int)(b < kMaxPartialBlocks ? b : kMaxPartialBlocks);
}

void launch_stats(int want_vd, const float* z, const uint8_t* mask, const void* dev_state, double* partials, Vol vol,
                  hipStream_t st, int K) {
    launch_stats_march(want_vd, z, mask, dev_state, partials, stats_blocks(vol), vol, K, st);
}

// SSD residual z = F - M o phi (builder-defined data term)
__global__ __launch_bounds__(kBlock) void residual_ssd_kernel(const float* __restrict__ fixed, int64_t f_stride,
                                                              const float* __restrict__ warped, float* __restrict__ z,
                                                              Vol vol) {
    IRS_VOXEL(vol, chain, x, y, zz, p);
    (void)x; (void)y; (void)zz;
    const int64_t i = (int64_t)chain * vol.V + p;
    z[i] = fixed[(int64_t)chain * f_stride + p] - warped[i];
}

void launch_residual_ssd(const float* fixed, int64_t f_stride, const float* warped, float* z, int C, Vol vol,
                         hipStream_t st) {
    hipLaunchKernelGGL(residual_ssd_kernel, vox_grid(vol, C), dim3(kBlock), 0, st, fixed, f_stride, warped, z, vol);
}

// n, sum z, sum z^2 over the mask (GMM initialisation, trainer.py:537-541); partials: [blocks][3]
__global__ __launch_bounds__(kBlock) void masked_moments_kernel(const float* __restrict__ z,
                                                                const uint8_t* __restrict__ mask,
                                                                double* __restrict__ partials, Vol vol) {
    __shared__ double smem[3 * (kBlock / kWave)];
    double acc[3] = {0.0, 0.0, 0.0};
    IRS_ROWS_BEGIN(vol, x, y, zc, v)
        (void)x; (void)y; (void)zc;
        if (mask[v]) {
            const double zz = (double)z[v];
            acc[0] += 1.0;
            acc[1] += zz;
            acc[2] += zz * zz;
        }
    IRS_ROWS_END
    block_sum<3>(acc, smem);
    if (threadIdx.x == 0)
        for (int j = 0; j < 3; ++j) partials[(int64_t)blockIdx.x * 3 + j] = acc[j];
}

void launch_masked_moments(const float* z, const uint8_t* mask, double* partials, Vol vol, hipStream_t st) {
    hipLaunchKernelGGL(masked_moments_kernel, dim3(stats_blocks(vol)), dim3(kBlock), 0, st, z, mask, partials, vol);
}

// ------------------------------------------------------------------------------------------------
// regulariser energy y_c = sum over 3 components x 3 axes of (forward difference)^2, where the difference array is
// replicate-padded, i.e. the last interior difference counts twice (utils/diff_op.py:83-85).  partials: [C][blocks]
// ------------------------------------------------------------------------------------------------

int energy_blocks(Vol vol) { return stats_blocks(vol); }

void launch_reg_energy(const float* v, double* partials, int C, Vol vol, hipStream_t st) {
    launch_reg_energy_march(v, partials, energy_blocks(vol), C, vol, st);
}

// out[c] = sum_b partials[c][b] (fixed order) -- used by the stand-alone irs_reg_energy
__global__ __launch_bounds__(kBlock) void reduce_rows_kernel(const double* __restrict__ partials, int nblocks,
                                                             double* __restrict__ out) {
    __shared__ double smem[kBlock / kWave];
    double acc[1] = {0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) acc[0] += partials[(int64_t)blockIdx.x * nblocks + b];
    block_sum<1>(acc, smem);
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(kBlock) void reduce_cols_kernel(const double* __restrict__ partials, int nblocks, int ncols,
                                                             double* __restrict__ out) {
    __shared__ double smem[kBlock / kWave];
    double acc[1] = {0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) acc[0] += partials[(int64_t)b * ncols + blockIdx.x];
    block_sum<1>(acc, smem);
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0];
}

void launch_reduce_cols(const double* partials, int nblocks, int ncols, double* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_cols_kernel, dim3(ncols), dim3(kBlock), 0, st, partials, nblocks, ncols, out);
}

void launch_reduce_partials(const double* partials, int nblocks, int nvals, double* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(nvals), dim3(kBlock), 0, st, partials, nblocks, out);
}

// ------------------------------------------------------------------------------------------------
// gradient assembly + SGLD/SGD update (trainer.py:349-351, utils/functions.py:83-84,107-109):
//   grad = sigma^2 * ( g * scale_c  +  coef_c * d(energy)/d(v_s) ),   v <- v - lr * grad
// d(energy)/d(v_s) = 2 D^T D v_s with the last forward difference of each axis weighted twice.
// The Sobolev backward is the identity (straight-through), so nothing else sits between v_s and v.
// ------------------------------------------------------------------------------------------------
void launch_sgld_update(float* v, const float* sigma, const float* g_d0, const float* v_s, const void* dev_state,
                        float lr, float s0, float s1, float s2, float* grad_out, int C, Vol vol, hipStream_t st,
                        double* energy_partials, bool coef_from_w) {
    launch_sgld_update_march(v, sigma, g_d0, v_s, dev_state, lr, s0, s1, s2, grad_out, C, vol, st, energy_partials, coef_from_w);
}

// ------------------------------------------------------------------------------------------------
// GradientOperator.forward (utils/diff_op.py:78-96): nabla[c, a, z, y, x, comp] = d v_comp / d axis_a
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fwd_diff(const float* __restrict__ f, int64_t p, int pos, int n, int64_t stride) {
    return pos + 1 < n ? f[p + stride] - f[p] : f[p] - f[p - stride];  // replicated last difference
}

__global__ __launch_bounds__(kBlock) void gradient_operator_kernel(const float* __restrict__ v, float* __restrict__ nabla,
                                                                   int transformation, Vol vol) {
    IRS_VOXEL(vol, chain, x, y, z, p);
    const int64_t cb = (int64_t)chain * 3 * vol.V;
    const int64_t plane = (int64_t)vol.W * vol.H;
    const float sp[3] = {2.0f / (float)(vol.W - 1), 2.0f / (float)(vol.H - 1), 2.0f / (float)(vol.D - 1)};
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
        const float* f = v + cb + comp * vol.V;
        float d[3] = {fwd_diff(f, p, x, vol.W, 1), fwd_diff(f, p, y, vol.H, vol.W), fwd_diff(f, p, z, vol.D, plane)};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (transformation) d[a] = d[a] / sp[a];
            nabla[(((int64_t)chain * 3 + a) * vol.V + p) * 3 + comp] = d[a];
        }
    }
}

void launch_gradient_operator(const float* v, float* nabla, int transformation, int C, Vol vol, hipStream_t st) {
    const dim3 grid = vox_grid(vol, C);
    hipLaunchKernelGGL(gradient_operator_kernel, grid, dim3(kBlock), 0, st, v, nabla, transformation, vol);
}

// log det J of a transformation in [-1,1] (GradientOperator(transformation=True) + calc_det_J + log,
// utils/util.py:72-91,209-212) and the count of NaNs (= folded voxels) per chain
__global__ __launch_bounds__(kBlock) void log_det_kernel(const float* __restrict__ t, float* __restrict__ log_det,
                                                         long long* __restrict__ nan_count, Vol vol) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int chain = blockIdx.z / vol.nz, z = vol.z0 + blockIdx.z - chain * vol.nz;
    int bad = 0;
    if (x < vol.W && y < vol.H) {
        const int64_t p = ((int64_t)z * vol.H + y) * vol.W + x;
        const int64_t cb = (int64_t)chain * 3 * vol.V;
        const int64_t plane = (int64_t)vol.W * vol.H;
        const float sp[3] = {2.0f / (float)(vol.W - 1), 2.0f / (float)(vol.H - 1), 2.0f / (float)(vol.D - 1)};
        float n[3][3];  // n[a][comp]
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) {
            const float* f = t + cb + comp * vol.V;
            n[0][comp] = fwd_diff(f, p, x, vol.W, 1) / sp[0];
            n[1][comp] = fwd_diff(f, p, y, vol.H, vol.W) / sp[1];
            n[2][comp] = fwd_diff(f, p, z, vol.D, plane) / sp[2];
        }
        // nabla_x = n[.][0], nabla_y = n[.][1], nabla_z = n[.][2]; formula of utils/util.py:84-89
        const float det = n[0][0] * n[1][1] * n[2][2] + n[0][1] * n[1][2] * n[2][0] + n[0][2] * n[1][0] * n[2][1] -
                          n[2][0] * n[1][1] * n[0][2] - n[2][1] * n[1][2] * n[0][0] - n[2][2] * n[1][0] * n[0][1];
        const float ld = logf(det);
        if (log_det) log_det[(int64_t)chain * vol.V + p] = ld;
        bad = ld != ld;
    }
    const unsigned long long b = __ballot(bad);
    if ((threadIdx.x & (kWave - 1)) == 0 && b) atomicAdd((unsigned long long*)(nan_count + chain), (unsigned long long)__popcll(b));
}

void launch_log_det_jacobian(const float* t, float* log_det, long long* nan_count, int C, Vol vol, hipStream_t st) {
    (void)hipMemsetAsync(nan_count, 0, sizeof(long long) * C, st);
    const dim3 grid = vox_grid(vol, C);
    hipLaunchKernelGGL(log_det_kernel, grid, dim3(kBlock), 0, st, t, log_det, nan_count, vol);
}

}  // namespace irs
