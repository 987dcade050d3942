// gfx950 kernels for the similarity / regulariser / update half of the SG-MCMC transition:
//   LCC map forward + adjoint through an LDS tile with halo   (reference model/loss.py:53-59,102-111)
//   mixture statistics for virtual decimation + the GMM step (utils/util.py:330-347,446-485; trainer.py:68-77)
//   data term and d/dz with the updated mixture              (model/loss.py:87-100)
//   regulariser energy, its adjoint stencil and the SGLD/SGD update (utils/diff_op.py:78-96; model/loss.py:152-161;
//                                                             utils/functions.py:83-84; trainer.py:349-351)
#include "kernels.h"
#include "scalar_kernels.h"

namespace irs {

// ------------------------------------------------------------------------------------------------
// LDS tile machinery.  A tile of TX x TY x TZ output voxels per 512-thread workgroup; box filters are run as three
// 1-D passes over LDS-resident boxes that shrink by the half width per pass.
//   forward pass : dst(p) = sum_o src(clamp(p + o))           (all-ones Conv3d with padding_mode='replicate')
//   adjoint pass : dst(p) = sum_q src(q) * #{o : clamp(q + o) = p}  -- a plain box sum plus a ramp of extra weights
//                  on the two boundary planes; src must be zero outside the volume, dst is forced to zero there.
// ------------------------------------------------------------------------------------------------
constexpr int TX = 32, TY = 8, TZ = 8;
constexpr int kLdsBlock = 512;

struct Box {
    int ox, oy, oz;  // global coordinates of element (0,0,0)
    int ex, ey, ez;  // extents
};

template <int AXIS, bool ADJ>
__device__ __forceinline__ void lds_pass(const float* __restrict__ src, const Box sb, float* __restrict__ dst,
                                         const Box db, const int s, const Vol vol) {
    const int n = db.ex * db.ey * db.ez;
    const int nA = AXIS == 0 ? vol.W : (AXIS == 1 ? vol.H : vol.D);
    const int so = AXIS == 0 ? sb.ox : (AXIS == 1 ? sb.oy : sb.oz);
    const int stride = AXIS == 0 ? 1 : (AXIS == 1 ? sb.ex : sb.ex * sb.ey);
    for (int i = threadIdx.x; i < n; i += kLdsBlock) {
        const int lx = i % db.ex, ly = (i / db.ex) % db.ey, lz = i / (db.ex * db.ey);
        const int gx = db.ox + lx, gy = db.oy + ly, gz = db.oz + lz;
        const int g = AXIS == 0 ? gx : (AXIS == 1 ? gy : gz);
        // LDS index of the element whose AXIS coordinate is `so` (local 0) on this line
        const int base = ((gz - sb.oz) * sb.ey + (gy - sb.oy)) * sb.ex + (gx - sb.ox) - (g - so) * stride;
        float acc = 0.0f;
        if (!ADJ) {
            for (int o = -s; o <= s; ++o) {
                const int q = min(max(g + o, 0), nA - 1);
                acc += src[base + (q - so) * stride];
            }
        } else {
            const bool inside = gx >= 0 && gx < vol.W && gy >= 0 && gy < vol.H && gz >= 0 && gz < vol.D;
            if (inside) {
                for (int o = -s; o <= s; ++o) acc += src[base + (g + o - so) * stride];
                if (g == 0)
                    for (int q = 0; q < s; ++q) acc += (float)(s - q) * src[base + (q - so) * stride];
                if (g == nA - 1)
                    for (int q = nA - s; q < nA; ++q) acc += (float)(s - (nA - 1 - q)) * src[base + (q - so) * stride];
            }
        }
        dst[i] = acc;
    }
}

__device__ __forceinline__ Box shrink(Box b, int axis, int s) {
    if (axis == 0) { b.ox += s; b.ex -= 2 * s; }
    else if (axis == 1) { b.oy += s; b.ey -= 2 * s; }
    else { b.oz += s; b.ez -= 2 * s; }
    return b;
}

template <int S>
struct LccSizes {
    static constexpr int R2 = (TX + 4 * S) * (TY + 4 * S) * (TZ + 4 * S);
    static constexpr int P1 = (TX + 2 * S) * (TY + 4 * S) * (TZ + 4 * S);
    static constexpr int P2 = (TX + 2 * S) * (TY + 2 * S) * (TZ + 4 * S);
    static constexpr int R1 = (TX + 2 * S) * (TY + 2 * S) * (TZ + 2 * S);
};

// ------------------------------------------------------------------------------------------------
// LCC forward.  MAP = false: out = (I - u) / sigma;  MAP = true: out = fhat - (I - u) / sigma (GMM.map with the
// fixed side pre-normalised).  u = box(I)/n, sigma = sqrt(box((I-u)^2)/n + 1e-10)   (model/loss.py:103-111)
// ------------------------------------------------------------------------------------------------
template <int S, bool MAP>
__global__ __launch_bounds__(kLdsBlock) void lcc_fwd_kernel(const float* __restrict__ fhat, int64_t fhat_stride,
                                                            const float* __restrict__ im, float* __restrict__ out,
                                                            float* __restrict__ sigma_out, Vol vol, int tiles_z) {
    using Z = LccSizes<S>;
    __shared__ float lds[Z::R2 + Z::P1 + Z::P2];
    float* bufM = lds;
    float* bufP = bufM + Z::R2;
    float* bufQ = bufP + Z::P1;

    const int chain = blockIdx.z / tiles_z;
    const int ox = blockIdx.x * TX, oy = blockIdx.y * TY, oz = vol.z0 + (blockIdx.z % tiles_z) * TZ;
    const int zend = vol.z0 + vol.nz;
    const float* src = im + (int64_t)chain * vol.V;
    const Box R2{ox - 2 * S, oy - 2 * S, oz - 2 * S, TX + 4 * S, TY + 4 * S, TZ + 4 * S};
    const float inv_n_dummy = 0.0f;
    (void)inv_n_dummy;
    const float n = (float)((2 * S + 1) * (2 * S + 1) * (2 * S + 1));

    for (int i = threadIdx.x; i < Z::R2; i += kLdsBlock) {
        const int lx = i % R2.ex, ly = (i / R2.ex) % R2.ey, lz = i / (R2.ex * R2.ey);
        const int gx = min(max(R2.ox + lx, 0), vol.W - 1), gy = min(max(R2.oy + ly, 0), vol.H - 1),
                  gz = min(max(R2.oz + lz, 0), vol.D - 1);
        bufM[i] = src[((int64_t)gz * vol.H + gy) * vol.W + gx];
    }
    __syncthreads();
    // u * n over R1
    const Box Bx = shrink(R2, 0, S), Bxy = shrink(Bx, 1, S), R1 = shrink(Bxy, 2, S);
    lds_pass<0, false>(bufM, R2, bufP, Bx, S, vol);
    __syncthreads();
    lds_pass<1, false>(bufP, Bx, bufQ, Bxy, S, vol);
    __syncthreads();
    lds_pass<2, false>(bufQ, Bxy, bufP, R1, S, vol);
    __syncthreads();
    // a = I - u (kept in bufP), t = a^2 (bufQ), both over R1
    for (int i = threadIdx.x; i < Z::R1; i += kLdsBlock) {
        const int lx = i % R1.ex, ly = (i / R1.ex) % R1.ey, lz = i / (R1.ex * R1.ey);
        const float m = bufM[((lz + S) * R2.ey + (ly + S)) * R2.ex + (lx + S)];
        const float a = m - bufP[i] / n;
        bufP[i] = a;
        bufQ[i] = a * a;
    }
    __syncthreads();
    // var * n over the tile: x and y passes through LDS, z pass fused with the epilogue
    const Box Cx = shrink(R1, 0, S), Cxy = shrink(Cx, 1, S);
    lds_pass<0, false>(bufQ, R1, bufM, Cx, S, vol);
    __syncthreads();
    lds_pass<1, false>(bufM, Cx, bufQ, Cxy, S, vol);
    __syncthreads();
    for (int i = threadIdx.x; i < TX * TY * TZ; i += kLdsBlock) {
        const int lx = i % TX, ly = (i / TX) % TY, lz = i / (TX * TY);
        const int gx = ox + lx, gy = oy + ly, gz = oz + lz;
        if (gx >= vol.W || gy >= vol.H || gz >= zend) continue;
        float acc = 0.0f;
        for (int o = -S; o <= S; ++o) {
            const int q = min(max(gz + o, 0), vol.D - 1);
            acc += bufQ[((q - Cxy.oz) * Cxy.ey + ly) * Cxy.ex + lx];
        }
        const float sigma = sqrtf(acc / n + 1e-10f);
        const float a = bufP[((lz + S) * R1.ey + (ly + S)) * R1.ex + (lx + S)];
        const int64_t g = ((int64_t)gz * vol.H + gy) * vol.W + gx;
        const float w = a / sigma;
        out[(int64_t)chain * vol.V + g] = MAP ? fhat[(int64_t)chain * fhat_stride + g] - w : w;
        if (sigma_out) sigma_out[(int64_t)chain * vol.V + g] = sigma;
    }
}

static dim3 tile_grid(Vol vol, int C, int* tiles_z) {
    *tiles_z = (vol.nz + TZ - 1) / TZ;
    return dim3((vol.W + TX - 1) / TX, (vol.H + TY - 1) / TY, (unsigned)(*tiles_z * C));
}

void launch_lcc_fwd(const float* fhat, int64_t fhat_stride, const float* im, float* z, float* sigma_out, int s, int C,
                    Vol vol, hipStream_t st) {
    int tz;
    const dim3 grid = tile_grid(vol, C, &tz);
    const bool map = fhat != nullptr;
#define IRS_LCC_FWD(SS)                                                                                              \
    if (map) hipLaunchKernelGGL((lcc_fwd_kernel<SS, true>), grid, dim3(kLdsBlock), 0, st, fhat, fhat_stride, im, z, \
                                sigma_out, vol, tz);                                                                \
    else hipLaunchKernelGGL((lcc_fwd_kernel<SS, false>), grid, dim3(kLdsBlock), 0, st, fhat, fhat_stride, im, z,    \
                            sigma_out, vol, tz);
    if (s == 1) { IRS_LCC_FWD(1) } else { IRS_LCC_FWD(2) }
#undef IRS_LCC_FWD
}

// ------------------------------------------------------------------------------------------------
// data term + adjoint of the LCC map, fused (one launch per chain because the mixture changes between chains):
//   g_z   = alpha * mask * d(-log p)/dz           (or an explicit g_z for the stand-alone operator)
//   w = fhat - z = a / sigma;   g_w = -g_z
//   g_var = -g_w w / (2 sigma^2);  g_t = B^T(g_var) / n;  g_a = g_w / sigma + 2 a g_t;  g_M = g_a - B^T(g_a) / n
// The block also accumulates sum(mask * -log p) over its tile (fp64) for the reported data term.
// ------------------------------------------------------------------------------------------------
template <int S, bool EXPLICIT_GZ>
__global__ __launch_bounds__(kLdsBlock) void lcc_data_bwd_kernel(const float* __restrict__ fhat,
                                                                 const float* __restrict__ z,
                                                                 const float* __restrict__ sigma_m,
                                                                 const uint8_t* __restrict__ mask,
                                                                 const float* __restrict__ gz_in,
                                                                 const DevState* __restrict__ state, int chain,
                                                                 float* __restrict__ g_m, double* __restrict__ nll_out,
                                                                 Vol vol) {
    using Z = LccSizes<S>;
    // LDS regions, aliased along the lifetime of the buffers (S = 1: 67.5 KB -> two workgroups per CU):
    //   region G : g_var over R2  -> y-pass temporary of the first adjoint box -> g_a over R1
    //   region P : x-pass / z-pass temporaries (both adjoint boxes)
    //   region W : g_w / sigma over R1 -> y-pass temporary of the second adjoint box
    //   region A : 2 a / n over R1
    static_assert(Z::P2 <= Z::R2 && Z::R1 <= Z::R2 && TX * (TY + 2 * S) * (TZ + 2 * S) <= Z::P1 && TX * TY * (TZ + 2 * S) <= Z::R1,
                  "LDS aliasing assumptions");
    __shared__ float lds[Z::R2 + Z::P1 + 2 * Z::R1];
    __shared__ double red[kLdsBlock / kWave];
    float* bufG = lds;
    float* bufP = bufG + Z::R2;
    float* bufW = bufP + Z::P1;
    float* bufA = bufW + Z::R1;
    float* bufQ = bufG;   // first box: written by the y pass, when g_var (x-pass input) is dead
    float* bufQ2 = bufW;  // second box: written by the y pass, when g_w / sigma has been folded into g_a

    const int ox = blockIdx.x * TX, oy = blockIdx.y * TY, oz = vol.z0 + blockIdx.z * TZ;
    const int zend = vol.z0 + vol.nz;
    const Box R2{ox - 2 * S, oy - 2 * S, oz - 2 * S, TX + 4 * S, TY + 4 * S, TZ + 4 * S};
    const Box Bx = shrink(R2, 0, S), Bxy = shrink(Bx, 1, S), R1 = shrink(Bxy, 2, S);
    const float n = (float)((2 * S + 1) * (2 * S + 1) * (2 * S + 1));
    const float alpha = EXPLICIT_GZ ? 1.0f : (float)state->sc.alpha[chain];

    double nll = 0.0;
    for (int i = threadIdx.x; i < Z::R2; i += kLdsBlock) {
        const int lx = i % R2.ex, ly = (i / R2.ex) % R2.ey, lz = i / (R2.ex * R2.ey);
        const int gx = R2.ox + lx, gy = R2.oy + ly, gz = R2.oz + lz;
        const bool inside = gx >= 0 && gx < vol.W && gy >= 0 && gy < vol.H && gz >= 0 && gz < vol.D;
        float gvar = 0.0f, pw = 0.0f, a2 = 0.0f;
        if (inside) {
            const int64_t g = ((int64_t)gz * vol.H + gy) * vol.W + gx;
            const float zz = z[g], sg = sigma_m[g];
            float gzv;
            if (EXPLICIT_GZ) {
                gzv = gz_in[g];
            } else {
                gzv = 0.0f;
                if (mask[g]) {
                    const MixEval e = mix_eval<false>(zz, state, nullptr, nullptr);
                    gzv = alpha * e.gz;
                    if (lx >= 2 * S && lx < 2 * S + TX && ly >= 2 * S && ly < 2 * S + TY && lz >= 2 * S && lz < 2 * S + TZ && gz < zend)
                        nll += (double)e.nll;
                }
            }
            const float w = fhat[g] - zz;
            const float gw = -gzv;
            gvar = -gw * w / (2.0f * sg * sg);
            pw = gw / sg;
            a2 = 2.0f * w * sg / n;
        }
        bufG[i] = gvar;
        if (lx >= S && lx < R2.ex - S && ly >= S && ly < R2.ey - S && lz >= S && lz < R2.ez - S) {
            const int j = ((lz - S) * R1.ey + (ly - S)) * R1.ex + (lx - S);
            bufW[j] = pw;
            bufA[j] = a2;
        }
    }
    __syncthreads();
    lds_pass<0, true>(bufG, R2, bufP, Bx, S, vol);
    __syncthreads();
    lds_pass<1, true>(bufP, Bx, bufQ, Bxy, S, vol);
    __syncthreads();
    lds_pass<2, true>(bufQ, Bxy, bufP, R1, S, vol);
    __syncthreads();
    for (int i = threadIdx.x; i < Z::R1; i += kLdsBlock) bufG[i] = bufW[i] + bufA[i] * bufP[i];  // g_a (0 outside)
    __syncthreads();
    const Box Cx = shrink(R1, 0, S), Cxy = shrink(Cx, 1, S);
    lds_pass<0, true>(bufG, R1, bufP, Cx, S, vol);
    __syncthreads();
    lds_pass<1, true>(bufP, Cx, bufQ2, Cxy, S, vol);
    __syncthreads();
    for (int i = threadIdx.x; i < TX * TY * TZ; i += kLdsBlock) {
        const int lx = i % TX, ly = (i / TX) % TY, lz = i / (TX * TY);
        const int gx = ox + lx, gy = oy + ly, gz = oz + lz;
        if (gx >= vol.W || gy >= vol.H || gz >= zend) continue;
        float acc = 0.0f;
        const int col = ly * Cxy.ex + lx;
        for (int o = -S; o <= S; ++o) {
            const int q = gz + o;
            if (q >= 0 && q < vol.D) acc += bufQ2[(q - Cxy.oz) * Cxy.ey * Cxy.ex + col];
        }
        if (gz == 0)
            for (int q = 0; q < S; ++q) acc += (float)(S - q) * bufQ2[(q - Cxy.oz) * Cxy.ey * Cxy.ex + col];
        if (gz == vol.D - 1)
            for (int q = vol.D - S; q < vol.D; ++q)
                acc += (float)(S - (vol.D - 1 - q)) * bufQ2[(q - Cxy.oz) * Cxy.ey * Cxy.ex + col];
        const float ga = bufG[((lz + S) * R1.ey + (ly + S)) * R1.ex + (lx + S)];
        g_m[((int64_t)gz * vol.H + gy) * vol.W + gx] = ga - acc / n;
    }
    if (!EXPLICIT_GZ) {
        nll = wave_sum(nll);
        if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = nll;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < kLdsBlock / kWave; ++w) t += red[w];
            nll_out[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
        }
    }
}

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

int data_bwd_blocks(int mode, Vol vol) {
    if (mode == IRS_DATA_SSD) return stats_blocks(vol);
    return ((vol.W + TX - 1) / TX) * ((vol.H + TY - 1) / TY) * ((vol.nz + TZ - 1) / TZ);
}

void launch_data_bwd(int mode, const float* fhat_or_fixed, int64_t f_stride, const float* z, const float* sigma_m,
                     const uint8_t* mask, int64_t mask_stride, const float* g_z_override, const void* dev_state,
                     int chain, float* g_warped, double* nll_partials, int s, int C_launch, Vol vol, hipStream_t st) {
    (void)f_stride;
    (void)mask_stride;
    (void)C_launch;
    const DevState* state = (const DevState*)dev_state;
    if (mode == IRS_DATA_SSD) {
        hipLaunchKernelGGL(ssd_bwd_kernel, dim3(stats_blocks(vol)), dim3(kBlock), 0, st, z, mask, state, chain, g_warped,
                           nll_partials, vol);
        return;
    }
    int tz;
    const dim3 grid = tile_grid(vol, 1, &tz);
#define IRS_LCC_BWD(SS)                                                                                                \
    if (g_z_override) hipLaunchKernelGGL((lcc_data_bwd_kernel<SS, true>), grid, dim3(kLdsBlock), 0, st, fhat_or_fixed, \
                                         z, sigma_m, mask, g_z_override, state, chain, g_warped, nll_partials, vol);  \
    else hipLaunchKernelGGL((lcc_data_bwd_kernel<SS, false>), grid, dim3(kLdsBlock), 0, st, fhat_or_fixed, z, sigma_m, \
                            mask, g_z_override, state, chain, g_warped, nll_partials, vol);
    if (s == 1) { IRS_LCC_BWD(1) } else { IRS_LCC_BWD(2) }
#undef IRS_LCC_BWD
}

// ------------------------------------------------------------------------------------------------
// per-chain statistics with the CURRENT mixture (before its Adam step): n_mask, sum x^2, the three lag-1 products
// sum x(p) x(p + e_a) (utils/util.py:466-475), and the sums that make up d(NLL)/d(log_std_k), d(NLL)/d(log pi_k).
// partials: [gridDim.x][kStatVals]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void stats_kernel(const float* __restrict__ z, const uint8_t* __restrict__ mask,
                                                       const DevState* __restrict__ state, int want_vd,
                                                       double* __restrict__ partials, Vol vol) {
    __shared__ double smem[kStatVals * (kBlock / kWave)];
    double acc[kStatVals];
#pragma unroll
    for (int j = 0; j < kStatVals; ++j) acc[j] = 0.0;
    const int64_t plane = (int64_t)vol.W * vol.H;
    IRS_ROWS_BEGIN(vol, x, y, zc, v)
        if (!mask[v]) continue;
        float resp[IRS_MAX_COMPONENTS], q[IRS_MAX_COMPONENTS];
        const MixEval e = mix_eval<true>(z[v], state, resp, q);
        acc[0] += 1.0;
        acc[1] += (double)(e.x * e.x);
        if (state->mode == IRS_DATA_GMM_LCC) {
#pragma unroll
            for (int k = 0; k < IRS_MAX_COMPONENTS; ++k)
                if (k < state->K) {
                    acc[5 + k] += (double)(resp[k] * (1.0f - q[k]));
                    acc[5 + IRS_MAX_COMPONENTS + k] += (double)resp[k];
                }
        }
        if (want_vd) {
            // lag-1 neighbours along D (reference "cov_x", dim 2), H (dim 3), W (dim 4)
            if (zc + 1 < vol.D && mask[v + plane]) acc[2] += (double)(e.x * mix_eval<false>(z[v + plane], state, nullptr, nullptr).x);
            if (y + 1 < vol.H && mask[v + vol.W]) acc[3] += (double)(e.x * mix_eval<false>(z[v + vol.W], state, nullptr, nullptr).x);
            if (x + 1 < vol.W && mask[v + 1]) acc[4] += (double)(e.x * mix_eval<false>(z[v + 1], state, nullptr, nullptr).x);
        }
    IRS_ROWS_END
    block_sum<kStatVals>(acc, smem);
    if (threadIdx.x == 0)
#pragma unroll
        for (int j = 0; j < kStatVals; ++j) partials[(int64_t)blockIdx.x * kStatVals + j] = acc[j];
}

int stats_blocks(Vol vol) {
    const int64_t b = (vol.V + kBlock - 1) / kBlock;
    return (int)(b < kMaxPartialBlocks ? b : kMaxPartialBlocks);
}

void launch_stats(int want_vd, const float* z, const uint8_t* mask, const void* dev_state, double* partials, Vol vol,
                  hipStream_t st) {
    hipLaunchKernelGGL(stats_kernel, dim3(stats_blocks(vol)), dim3(kBlock), 0, st, z, mask, (const DevState*)dev_state,
                       want_vd, partials, vol);
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
__global__ __launch_bounds__(kBlock) void reg_energy_kernel(const float* __restrict__ v, double* __restrict__ partials,
                                                            Vol vol) {
    __shared__ double smem[kBlock / kWave];
    const int64_t plane = (int64_t)vol.W * vol.H;
    const float* f = v + (int64_t)blockIdx.y * 3 * vol.V;
    double acc[1] = {0.0};
    IRS_ROWS_BEGIN(vol, x, y, z, p)
        float e = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* fc = f + c * vol.V;
            const float v0 = fc[p];
            if (x + 1 < vol.W) { const float d = fc[p + 1] - v0; e += (x + 2 == vol.W ? 2.0f : 1.0f) * d * d; }
            if (y + 1 < vol.H) { const float d = fc[p + vol.W] - v0; e += (y + 2 == vol.H ? 2.0f : 1.0f) * d * d; }
            if (z + 1 < vol.D) { const float d = fc[p + plane] - v0; e += (z + 2 == vol.D ? 2.0f : 1.0f) * d * d; }
        }
        acc[0] += (double)e;
    IRS_ROWS_END
    block_sum<1>(acc, smem);
    if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = acc[0];
}

int energy_blocks(Vol vol) { return stats_blocks(vol); }

void launch_reg_energy(const float* v, double* partials, int C, Vol vol, hipStream_t st) {
    hipLaunchKernelGGL(reg_energy_kernel, dim3(energy_blocks(vol), C), dim3(kBlock), 0, st, v, partials, vol);
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
__device__ __forceinline__ float dtd_axis(const float* __restrict__ f, int64_t p, int pos, int n, int64_t stride) {
    // sum over the axis' difference terms touching position `pos`: w_{pos-1} d_{pos-1} - w_pos d_pos
    const float v0 = f[p];
    float r = 0.0f;
    if (pos >= 1) r += (pos - 1 == n - 2 ? 2.0f : 1.0f) * (v0 - f[p - stride]);
    if (pos <= n - 2) r -= (pos == n - 2 ? 2.0f : 1.0f) * (f[p + stride] - v0);
    return r;
}

__global__ __launch_bounds__(kBlock) void sgld_update_kernel(float* __restrict__ v, const float* __restrict__ sigma,
                                                             const float* __restrict__ g, const float* __restrict__ v_s,
                                                             const DevState* __restrict__ state, float lr, float s0,
                                                             float s1, float s2, float* __restrict__ grad_out, Vol vol) {
    IRS_VOXEL(vol, chain, x, y, z, p);
    const int64_t cb = (int64_t)chain * 3 * vol.V;
    const int64_t plane = (int64_t)vol.W * vol.H;
    const float coef2 = 2.0f * (float)state->coef[chain];
    const float sc[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int64_t i = cb + c * vol.V + p;
        const float* f = v_s + cb + c * vol.V;
        const float lap = dtd_axis(f, p, x, vol.W, 1) + dtd_axis(f, p, y, vol.H, vol.W) + dtd_axis(f, p, z, vol.D, plane);
        const float gr = g[i] * sc[c] + coef2 * lap;
        const float sg = sigma ? sigma[i] : 1.0f;
        const float gs = sg * sg * gr;  // SGLD.backward: sigma^2 * grad (utils/functions.py:83-84) == v.grad in the reference
        if (grad_out) grad_out[i] = gs;
        v[i] = v[i] - lr * gs;
    }
}

void launch_sgld_update(float* v, const float* sigma, const float* g_d0, const float* v_s, const void* dev_state,
                        float lr, float s0, float s1, float s2, float* grad_out, int C, Vol vol, hipStream_t st) {
    const dim3 grid = vox_grid(vol, C);
    hipLaunchKernelGGL(sgld_update_kernel, grid, dim3(kBlock), 0, st, v, sigma, g_d0, v_s, (const DevState*)dev_state, lr,
                       s0, s1, s2, grad_out, vol);
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
