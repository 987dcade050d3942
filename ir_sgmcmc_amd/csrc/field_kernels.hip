// gfx950 kernels for the velocity/displacement-field half of the SG-MCMC transition:
//   SGLD perturbation                       (reference utils/functions.py:76-84, utils/util.py:48-58)
//   outputs of scaling and squaring         (utils/transformation.py:75-76; the steps themselves: exp_kernels.hip)
//   trilinear / nearest warps               (utils/registration.py:17-30, utils/util.py:44-53)
//   cubic B-spline FFD up-sampling / adjoint (utils/transformation.py:105-153)
// All of it is HBM/L2-bound gather/stencil work on planar fp32 fields (C,3,D,H,W); no MFMA by design.
#include "kernels.h"
#include "warp_device.h"

namespace irs {

// ------------------------------------------------------------------------------------------------
// SGLD.forward: out = v + (amp * sigma) * eps          (utils/util.py:56-58; amp = sqrt(2 tau))
// One thread handles the three channels of the voxels (x, y, 2p) and (x, y, 2p + 1): ONE Philox4x32-10 call yields the six
// normals of the pair (3 Box-Muller pairs from 6 x 21 bits) -- with a call per voxel the kernel was VALU-bound, not
// HBM-bound.  Pairs are aligned to absolute even z, so a z-window (slab) draws the same noise as the full volume.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void perturb_kernel(const float* __restrict__ v, const float* __restrict__ sigma,
                                                         const float* __restrict__ eps, float amp,
                                                         float* __restrict__ out, Vol vol, uint64_t seed,
                                                         uint64_t iteration, const uint64_t* dev_iter, int npair) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int chain = blockIdx.z / npair;
    const int za = 2 * ((vol.z0 >> 1) + (int)blockIdx.z - chain * npair);
    if (x >= vol.W || y >= vol.H) return;
    const int64_t V = vol.V, HW = (int64_t)vol.H * vol.W;
    const int64_t voxa = ((int64_t)za * vol.H + y) * vol.W + x;
    float n[2][3];
    if (!eps) {
        const uint64_t it = dev_iter ? *dev_iter : iteration;
        const uint64_t idx = (uint64_t)chain * (uint64_t)vol.Vg + (uint64_t)voxa;  // global voxel index: slab-independent noise
        const U4 r = philox4x32_10(U4{(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)it, 0x5347u ^ (uint32_t)(it >> 32)},
                                   (uint32_t)seed, (uint32_t)(seed >> 32));
        uint32_t w[6];
        split21(r, w);
        box_muller21(w[0], w[1], n[0][0], n[0][1]);
        box_muller21(w[2], w[3], n[0][2], n[1][0]);
        box_muller21(w[4], w[5], n[1][1], n[1][2]);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int z = za + h;
        if (z < vol.z0 || z >= vol.z0 + vol.nz) continue;
        const int64_t base = (int64_t)chain * 3 * V + voxa + h * HW;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t i = base + c * V;
            const float nn = eps ? eps[i] : n[h][c];
            const float sg = sigma ? sigma[i] : 1.0f;
            out[i] = __fadd_rn(v[i], __fmul_rn(__fmul_rn(amp, sg), nn));
        }
    }
}

void launch_perturb(const float* v, const float* sigma, const float* eps, float amp, float* out, int C, Vol vol,
                    uint64_t seed, uint64_t iteration, const uint64_t* dev_iteration, hipStream_t st) {
    const int npair = ((vol.z0 + vol.nz - 1) >> 1) - (vol.z0 >> 1) + 1;
    const dim3 grid((unsigned)((vol.W + 63) / 64), (unsigned)((vol.H + 3) / 4), (unsigned)(npair * C));
    hipLaunchKernelGGL(perturb_kernel, grid, dim3(kBlock), 0, st, v, sigma, eps, amp, out, vol, seed, iteration, dev_iteration,
                       npair);
}

// per-channel constants of the coordinate transforms: x <-> W, y <-> H, z <-> D (the reference pairs channel c with shape[2 + c],
// which is the same thing for the cubic volumes it supports)
struct Scale3 {
    float nm1[3];   // axis length - 1 for channel c
    float rnm1[3];  // correctly rounded 1 / nm1 (common.h: exact_rcp, div_exact)
    float inv_pow;  // 1 / 2^no_steps
};

static Scale3 make_scale(Vol vol, int no_steps) {
    Scale3 s;
    s.nm1[0] = (float)(vol.W - 1);
    s.nm1[1] = (float)(vol.H - 1);
    s.nm1[2] = (float)(vol.D - 1);
    for (int c = 0; c < 3; ++c) s.rnm1[c] = exact_rcp(s.nm1[c]);
    s.inv_pow = 1.0f / (float)(1 << no_steps);
    return s;
}

// ------------------------------------------------------------------------------------------------
// transformation = id + d, displacement = (d * (n_c - 1)) / 2     (utils/transformation.py:75-76)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void svf_outputs_kernel(const float* __restrict__ d, float* __restrict__ transf,
                                                             float* __restrict__ disp, Vol vol, Lin lin, Scale3 sc) {
    IRS_VOXEL(vol, chain, x, y, z, vox);
    const int64_t cb = (int64_t)chain * 3 * vol.V;
    const float idv[3] = {lin.x[x], lin.y[y], lin.z[z]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float dv = d[cb + c * vol.V + vox];
        if (transf) transf[cb + c * vol.V + vox] = __fadd_rn(idv[c], dv);
        if (disp) disp[cb + c * vol.V + vox] = __fmul_rn(__fmul_rn(dv, sc.nm1[c]), 0.5f);
    }
}

// four x-consecutive voxels per thread (one 16-byte access per stream and lane): W % 4 == 0, 16-byte aligned bases
__global__ __launch_bounds__(kBlock) void svf_outputs_x4_kernel(const float* __restrict__ d, float* __restrict__ transf,
                                                                float* __restrict__ disp, Vol vol, Lin lin, Scale3 sc) {
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int chain = blockIdx.z / vol.nz, z = vol.z0 + blockIdx.z - chain * vol.nz;
    if (x >= vol.W || y >= vol.H) return;
    const int64_t vox = ((int64_t)z * vol.H + y) * vol.W + x, cb = (int64_t)chain * 3 * vol.V;
    const float4 lx = *reinterpret_cast<const float4*>(lin.x + x);
    const float ly = lin.y[y], lz = lin.z[z];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float4 dv = *reinterpret_cast<const float4*>(d + cb + c * vol.V + vox);
        const float4 id = c == 0 ? lx : make_float4(c == 1 ? ly : lz, c == 1 ? ly : lz, c == 1 ? ly : lz, c == 1 ? ly : lz);
        if (transf)
            *reinterpret_cast<float4*>(transf + cb + c * vol.V + vox) =
                make_float4(__fadd_rn(id.x, dv.x), __fadd_rn(id.y, dv.y), __fadd_rn(id.z, dv.z), __fadd_rn(id.w, dv.w));
        if (disp)
            *reinterpret_cast<float4*>(disp + cb + c * vol.V + vox) =
                make_float4(__fmul_rn(__fmul_rn(dv.x, sc.nm1[c]), 0.5f), __fmul_rn(__fmul_rn(dv.y, sc.nm1[c]), 0.5f),
                            __fmul_rn(__fmul_rn(dv.z, sc.nm1[c]), 0.5f), __fmul_rn(__fmul_rn(dv.w, sc.nm1[c]), 0.5f));
    }
}

void launch_svf_outputs(const float* d, float* transformation, float* displacement, int C, Vol vol, Lin lin,
                        hipStream_t st) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (vol.W % 4 == 0 && vol.W >= 128 && al16(d) && al16(transformation) && al16(displacement) && al16(lin.x)) {
        const dim3 g4((unsigned)((vol.W / 4 + 63) / 64), (unsigned)((vol.H + 3) / 4), (unsigned)(vol.nz * C));
        hipLaunchKernelGGL(svf_outputs_x4_kernel, g4, dim3(kBlock), 0, st, d, transformation, displacement, vol, lin,
                           make_scale(vol, 0));
        return;
    }
    const dim3 grid = vox_grid(vol, C);
    hipLaunchKernelGGL(svf_outputs_kernel, grid, dim3(kBlock), 0, st, d, transformation, displacement, vol, lin,
                       make_scale(vol, 0));
}

// ------------------------------------------------------------------------------------------------
// warp of the moving image at id + d (+ uniform jitter, utils/util.py:44-53) and its grid-gradient
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void grid_point(const float* __restrict__ d, const float* __restrict__ unif, const Jitter& jt,
                                           int64_t cb3, int chain, int64_t vox, Vol vol, Lin lin, int x, int y, int z,
                                           float (&g)[3]) {
    g[0] = __fadd_rn(lin.x[x], d[cb3 + vox]);
    g[1] = __fadd_rn(lin.y[y], d[cb3 + vol.V + vox]);
    g[2] = __fadd_rn(lin.z[z], d[cb3 + 2 * vol.V + vox]);
    jitter_point(g, unif, jt, cb3, chain, vox, vol.V, vol.Vg);
}

// `gradm` (optional, (C,3,D,H,W)): d(warped)/d(d_last) = the trilinear gradient of the moving image at the sampling
// position, with the clamp mask -- exactly what warp_bwd_kernel multiplies the upstream gradient with.  The fused transition
// lets the forward warp write it (the 8 taps are loaded anyway) and folds the product with g_warped into the staging of the
// first adjoint squaring step, so the backward warp and its 24 B/voxel round trip disappear.
// Each thread handles IRS_WARP_NV voxels (rows y, y + 4, ...) with all their loads issued before the first use: the kernel
// is two dependent memory round trips (d, then the eight taps at the position d names) and nothing else, so the number of
// independent voxels in flight per lane is what sets its rate.
#ifndef IRS_WARP_NV
#define IRS_WARP_NV 2
#endif
__global__ __launch_bounds__(kBlock) void warp_fwd_kernel(const float* __restrict__ im, int64_t im_stride,
                                                          const float* __restrict__ d, const float* __restrict__ unif,
                                                          Jitter jt, float* __restrict__ out, float* __restrict__ gradm,
                                                          int gradm_aos, Vol vol, Lin lin) {
    constexpr int NV = IRS_WARP_NV;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y0 = blockIdx.y * (4 * NV) + (threadIdx.x >> 6);
    const int chain = blockIdx.z / vol.nz, z = vol.z0 + blockIdx.z - chain * vol.nz;
    if (x >= vol.W || y0 >= vol.H) return;
    const int64_t cb3 = (int64_t)chain * 3 * vol.V;
    const float* src = im + (int64_t)chain * im_stride;
    float g[NV][3], gm[NV][3], wv[NV];
    int64_t vox[NV];
#pragma unroll
    for (int h = 0; h < NV; ++h) {  // rows past the volume repeat the last one (loads stay unconditional), their stores are skipped
        const int y = min(y0 + 4 * h, vol.H - 1);
        vox[h] = ((int64_t)z * vol.H + y) * vol.W + x;
        grid_point(d, unif, jt, cb3, chain, vox[h], vol, lin, x, y, z, g[h]);
    }
#pragma unroll
    for (int h = 0; h < NV; ++h) wv[h] = gradm ? warp_sample<true>(src, g[h], vol, gm[h]) : warp_sample<false>(src, g[h], vol, gm[h]);
#pragma unroll
    for (int h = 0; h < NV; ++h) {
        if (y0 + 4 * h >= vol.H) continue;
        out[(int64_t)chain * vol.V + vox[h]] = wv[h];
        if (gradm) {
            if (gradm_aos) {  // interleaved ([V][3]): the layout the fused adjoint steps read (exp_kernels.hip: Lay3)
                typedef float f3v __attribute__((ext_vector_type(3)));
                const f3v v = {gm[h][0], gm[h][1], gm[h][2]};
                __builtin_memcpy(gradm + cb3 + vox[h] * 3, &v, 12);
            } else {
                gradm[cb3 + vox[h]] = gm[h][0];
                gradm[cb3 + vol.V + vox[h]] = gm[h][1];
                gradm[cb3 + 2 * vol.V + vox[h]] = gm[h][2];
            }
        }
    }
}

void launch_warp_fwd(const float* im, int64_t im_stride, const float* d, const float* unif, float alpha, float* out,
                     float* gradm, int gradm_aos, int C, Vol vol, Lin lin, uint64_t seed, uint64_t iteration,
                     const uint64_t* dev_iteration, hipStream_t st) {
    const dim3 grid((unsigned)((vol.W + 63) / 64), (unsigned)((vol.H + 4 * IRS_WARP_NV - 1) / (4 * IRS_WARP_NV)), (unsigned)(vol.nz * C));
    hipLaunchKernelGGL(warp_fwd_kernel, grid, dim3(kBlock), 0, st, im, im_stride, d, unif,
                       make_jitter(alpha, vol, seed, iteration, dev_iteration), out, gradm, gradm_aos, vol, lin);
}

__global__ __launch_bounds__(kBlock) void warp_bwd_kernel(const float* __restrict__ im, int64_t im_stride,
                                                          const float* __restrict__ d, const float* __restrict__ unif,
                                                          Jitter jt, const float* __restrict__ gw,
                                                          float* __restrict__ gd, Vol vol, Lin lin) {
    IRS_VOXEL(vol, chain, x, y, z, vox);
    const int64_t cb3 = (int64_t)chain * 3 * vol.V;
    float g[3];
    grid_point(d, unif, jt, cb3, chain, vox, vol, lin, x, y, z, g);
    const AxisTap tx = axis_tap(g[0], vol.W), ty = axis_tap(g[1], vol.H), tz = axis_tap(g[2], vol.D);
    const float* src = im + (int64_t)chain * im_stride;
    float gix = 0.0f, giy = 0.0f, giz = 0.0f;
#pragma unroll
    for (int cz = 0; cz < 2; ++cz)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy) {
            const int64_t rowoff = ((int64_t)(cz ? tz.i1 : tz.i0) * vol.H + (cy ? ty.i1 : ty.i0)) * vol.W;
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
                const float wx = cx ? tx.w1 : tx.w0, wy = cy ? ty.w1 : ty.w0, wz = cz ? tz.w1 : tz.w0;
                const float val = src[rowoff + (cx ? tx.i1 : tx.i0)];
                gix += (cx ? val : -val) * (wy * wz);
                giy += (cy ? val : -val) * (wx * wz);
                giz += (cz ? val : -val) * (wx * wy);
            }
        }
    const float go = gw[(int64_t)chain * vol.V + vox];
    gd[cb3 + vox] = tx.gmul * gix * go;
    gd[cb3 + vol.V + vox] = ty.gmul * giy * go;
    gd[cb3 + 2 * vol.V + vox] = tz.gmul * giz * go;
}

void launch_warp_bwd(const float* im, int64_t im_stride, const float* d, const float* unif, float alpha,
                     const float* g_warped, float* g_d, int C, Vol vol, Lin lin, uint64_t seed, uint64_t iteration,
                     const uint64_t* dev_iteration, hipStream_t st) {
    const dim3 grid = vox_grid(vol, C);
    hipLaunchKernelGGL(warp_bwd_kernel, grid, dim3(kBlock), 0, st, im, im_stride, d, unif,
                       make_jitter(alpha, vol, seed, iteration, dev_iteration), g_warped, g_d, vol, lin);
}

// RegistrationModule.forward on an explicit transformation tensor (C,3,D,H,W) in [-1,1]
template <typename T, bool NEAREST>
__global__ __launch_bounds__(kBlock) void warp_transformation_kernel(const T* __restrict__ im, int64_t im_stride,
                                                                     const float* __restrict__ t, T* __restrict__ out,
                                                                     Vol vol) {
    IRS_VOXEL(vol, chain, x, y, z, vox);
    (void)x; (void)y; (void)z;
    const int64_t cb3 = (int64_t)chain * 3 * vol.V;
    const T* src = im + (int64_t)chain * im_stride;
    const float gx = t[cb3 + vox], gy = t[cb3 + vol.V + vox], gz = t[cb3 + 2 * vol.V + vox];
    if (NEAREST) {
        // ATen nearest: unnormalise, clip to the border, round half to even (nearbyint), in-bounds by construction
        auto nearest = [](float g, int n) {
            const float nm1 = (float)(n - 1);
            float i = __fmul_rn(__fmul_rn(__fadd_rn(g, 1.0f), 0.5f), nm1);
            i = fminf(fmaxf(i, 0.0f), nm1);
            return (int)nearbyintf(i);
        };
        const int xi = nearest(gx, vol.W), yi = nearest(gy, vol.H), zi = nearest(gz, vol.D);
        out[(int64_t)chain * vol.V + vox] = src[((int64_t)zi * vol.H + yi) * vol.W + xi];
    } else {
        const AxisTap tx = axis_tap(gx, vol.W), ty = axis_tap(gy, vol.H), tz = axis_tap(gz, vol.D);
        float acc = 0.0f;
#pragma unroll
        for (int cz = 0; cz < 2; ++cz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) {
                const int64_t rowoff = ((int64_t)(cz ? tz.i1 : tz.i0) * vol.H + (cy ? ty.i1 : ty.i0)) * vol.W;
#pragma unroll
                for (int cx = 0; cx < 2; ++cx) {
                    const float w = __fmul_rn(__fmul_rn(cx ? tx.w1 : tx.w0, cy ? ty.w1 : ty.w0), cz ? tz.w1 : tz.w0);
                    acc = __fadd_rn(acc, __fmul_rn((float)src[rowoff + (cx ? tx.i1 : tx.i0)], w));
                }
            }
        out[(int64_t)chain * vol.V + vox] = (T)acc;
    }
}

void launch_warp_transformation(const float* im, int64_t im_stride, const float* t, float* out, int C, Vol vol,
                                hipStream_t st) {
    const dim3 grid = vox_grid(vol, C);
    hipLaunchKernelGGL((warp_transformation_kernel<float, false>), grid, dim3(kBlock), 0, st, im, im_stride, t, out, vol);
}
void launch_warp_nearest_u8(const uint8_t* im, int64_t im_stride, const float* t, uint8_t* out, int C, Vol vol,
                            hipStream_t st) {
    const dim3 grid = vox_grid(vol, C);
    hipLaunchKernelGGL((warp_transformation_kernel<uint8_t, true>), grid, dim3(kBlock), 0, st, im, im_stride, t, out, vol);
}
void launch_warp_nearest_i16(const int16_t* im, int64_t im_stride, const float* t, int16_t* out, int C, Vol vol,
                             hipStream_t st) {
    const dim3 grid = vox_grid(vol, C);
    hipLaunchKernelGGL((warp_transformation_kernel<int16_t, true>), grid, dim3(kBlock), 0, st, im, im_stride, t, out, vol);
}

// ------------------------------------------------------------------------------------------------
// cubic B-spline FFD, one axis.  Arrays are viewed as [outer][n][inner].
//   up      : out[o][x][i] = sum_j in[o][j][i] * k[x + 3c - 1 - j c],   j = floor(x/c) .. floor(x/c)+3
//   adjoint : out[o][j][i] = sum_x in[o][x][i] * k[x + 3c - 1 - j c],   x = (j-3)c+1 .. (j+1)c-1
// which is conv_transpose1d(stride c, padding 2c-1) followed by the crop [c : c+N] of
// utils/transformation.py:146-153, and its transpose.
// ------------------------------------------------------------------------------------------------
// Row window (z-slab decomposition, slab.hip): the dense side of the D-axis pass is slab-local.  UP produces the output rows
// [w_lo, w_lo + w_n) only; ADJOINT sums the input rows [w_lo, w_lo + w_n) only (a rank's own planes: the partial sums are
// all-reduced).  The windowed array stores the rows [store_lo, store_lo + store_n) of its axis.  Full arrays: w = store = whole axis.
template <bool ADJ>
__global__ __launch_bounds__(kBlock) void ffd_axis_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          SplineTaps taps, int64_t outer, int n_in, int n_out,
                                                          int64_t inner, int w_lo, int w_n, int store_lo, int store_n) {
    const int rows = ADJ ? n_out : w_n;
    const int64_t total = outer * rows * inner;
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total) return;
    const int64_t i = e % inner;
    const int p = (int)((e / inner) % rows) + (ADJ ? 0 : w_lo);
    const int64_t o = e / (inner * rows);
    const int c = taps.cps;
    float acc = 0.0f;
    if (!ADJ) {
        const float* src = in + o * n_in * inner + i;
        const int j0 = p / c;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = j0 + jj;
            if (j < n_in) acc = fmaf(src[(int64_t)j * inner], taps.k[p + 3 * c - 1 - j * c], acc);
        }
        out[(o * store_n + (p - store_lo)) * inner + i] = acc;
    } else {
        const float* src = in + (o * store_n - store_lo) * inner + i;
        const int lo = max(max((p - 3) * c + 1, 0), w_lo), hi = min(min((p + 1) * c - 1, n_in - 1), w_lo + w_n - 1);
        for (int xq = lo; xq <= hi; ++xq) acc = fmaf(src[(int64_t)xq * inner], taps.k[xq + 3 * c - 1 - p * c], acc);
        out[(o * n_out + p) * inner + i] = acc;
    }
}

void launch_ffd_axis(const float* in, float* out, const SplineTaps& taps, bool adjoint, int64_t outer, int n_in,
                     int n_out, int64_t inner, hipStream_t st, int w_lo, int w_n, int store_lo, int store_n) {
    const int axis = adjoint ? n_in : n_out;  // the dense side
    if (w_n < 0) {
        w_lo = 0;
        w_n = axis;
    }
    if (store_n < 0) {
        store_lo = 0;
        store_n = axis;
    }
    const int64_t total = outer * (adjoint ? n_out : w_n) * inner;
    if (total <= 0) return;
    dim3 grid((unsigned)((total + kBlock - 1) / kBlock));
    if (adjoint) hipLaunchKernelGGL(ffd_axis_kernel<true>, grid, dim3(kBlock), 0, st, in, out, taps, outer, n_in, n_out, inner, w_lo, w_n, store_lo, store_n);
    else hipLaunchKernelGGL(ffd_axis_kernel<false>, grid, dim3(kBlock), 0, st, in, out, taps, outer, n_in, n_out, inner, w_lo, w_n, store_lo, store_n);
}

// out[c] = in[c] * s_c  (chain rule of the prescale: d_0 = v * 2/(n_c-1) / 2^steps)
__global__ __launch_bounds__(kBlock) void scale_channels_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                float s0, float s1, float s2, Vol vol) {
    IRS_VOXEL(vol, chain, x, y, z, vox);
    (void)x; (void)y; (void)z;
    const int64_t V = vol.V;
    const int64_t cb = (int64_t)chain * 3 * V;
    out[cb + vox] = in[cb + vox] * s0;
    out[cb + V + vox] = in[cb + V + vox] * s1;
    out[cb + 2 * V + vox] = in[cb + 2 * V + vox] * s2;
}

void launch_scale_channels(const float* in, float* out, float s0, float s1, float s2, int C, Vol vol, hipStream_t st) {
    hipLaunchKernelGGL(scale_channels_kernel, vox_grid(vol, C), dim3(kBlock), 0, st, in, out, s0, s1, s2, vol);
}

// ------------------------------------------------------------------------------------------------
// identity-grid tables: torch.linspace(-1, 1, n) as the CPU kernel evaluates it
// (start + step*i below the midpoint, end - step*(n-1-i) above; ATen RangeFactories)
// ------------------------------------------------------------------------------------------------
static void fill_linspace(float* dst, int n) {
    const float step = (1.0f - (-1.0f)) / (float)(n - 1);
    const int half = n / 2;
    for (int i = 0; i < n; ++i) dst[i] = i < half ? (-1.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

int ensure_lin_tables(LinTables& t, int D, int H, int W, hipStream_t st) {
    if (t.dev && t.D == D && t.H == H && t.W == W) return 0;
    if (t.dev) (void)hipFree(t.dev);
    t.dev = nullptr;
    const size_t n = (size_t)D + H + W;
    float* host = (float*)malloc(n * sizeof(float));
    if (!host) return 1;
    fill_linspace(host, W);
    fill_linspace(host + W, H);
    fill_linspace(host + W + H, D);
    hipError_t e = hipMalloc((void**)&t.dev, n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(t.dev, host, n * sizeof(float), hipMemcpyHostToDevice);
    free(host);
    if (e != hipSuccess) {
        t.dev = nullptr;
        return 1;
    }
    t.D = D;
    t.H = H;
    t.W = W;
    (void)st;
    return 0;
}

int cached_lin(int D, int H, int W, hipStream_t st, Lin* out) {
    static LinTables cache[8];
    static int next = 0;
    for (auto& c : cache)
        if (c.dev && c.D == D && c.H == H && c.W == W) {
            *out = c.lin();
            return 0;
        }
    LinTables& slot = cache[next];
    next = (next + 1) % 8;
    if (slot.dev) {  // evicting a table that an in-flight kernel might read: drain first
        (void)hipDeviceSynchronize();
    }
    if (ensure_lin_tables(slot, D, H, W, st)) return 1;
    *out = slot.lin();
    return 0;
}

}  // namespace irs
