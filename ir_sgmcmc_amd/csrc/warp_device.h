// Device code shared by the stand-alone warp kernels (field_kernels.hip) and the warp fused into the last squaring step
// (exp_kernels.hip): jitter of the sampling grid (utils/util.py:44-53) and the trilinear sample of the moving image with its
// grid-gradient (utils/registration.py:17-30).  Same expressions in both places -> bit-identical results.
#pragma once
#include <math.h>

#include "common.h"

namespace irs {

struct Jitter {
    float alpha;       // <= 0: disabled
    float nm1[3];      // transform_coordinates scaling of the jitter (x <-> W, y <-> H, z <-> D)
    float rnm1[3];     // correctly rounded 1 / nm1 (host): the division below stays exact at 3 instructions
    uint64_t seed, iteration;
    const uint64_t* dev_iter;
};

inline Jitter make_jitter(float alpha, Vol vol, uint64_t seed, uint64_t iteration, const uint64_t* dev_iter) {
    Jitter j;
    j.alpha = alpha;
    j.nm1[0] = (float)(vol.W - 1);
    j.nm1[1] = (float)(vol.H - 1);
    j.nm1[2] = (float)(vol.D - 1);
    for (int c = 0; c < 3; ++c) j.rnm1[c] = exact_rcp(j.nm1[c]);
    j.seed = seed;
    j.iteration = iteration;
    j.dev_iter = dev_iter;
    return j;
}

// g += transform_coordinates(U(-alpha, alpha)); `unif` (C,3,D,H,W) injected U[0,1) draws or nullptr -> Philox2x32-10
__device__ __forceinline__ void jitter_point(float (&g)[3], const float* __restrict__ unif, const Jitter& jt, int64_t cb3,
                                             int chain, int64_t vox, int64_t V, int64_t Vg) {
    if (jt.alpha <= 0.0f) return;
    float u[3];
    if (unif) {
        u[0] = unif[cb3 + vox];
        u[1] = unif[cb3 + V + vox];
        u[2] = unif[cb3 + 2 * V + vox];
    } else {
        const uint64_t it = jt.dev_iter ? *jt.dev_iter : jt.iteration;
        const uint64_t idx = (uint64_t)chain * (uint64_t)Vg + (uint64_t)vox;  // global voxel index, < 2^36 (dims_ok, chains)
        const U2 r = philox2x32_10(U2{(uint32_t)idx, (uint32_t)(idx >> 32) | ((uint32_t)it << 4)}, key_mix(jt.seed, it, 0x554Eu));
        u[0] = u01_21(r.x >> 11);
        u[1] = u01_21(r.y >> 11);
        u[2] = u01_21((r.x & 0x7FFu) | ((r.y & 0x3FFu) << 11));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float nz = __fadd_rn(__fmul_rn(-2.0f * jt.alpha, u[c]), jt.alpha);  // -2 a u + a
        // (nz * 2) / (n - 1), correctly rounded at 3 instructions (common.h: div_exact)
        g[c] = __fadd_rn(g[c], div_exact(__fmul_rn(nz, 2.0f), jt.nm1[c], jt.rnm1[c]));
    }
}

// trilinear sample of `src` (one chain's volume) at the normalised position g, and -- when WANT_GRAD -- its gradient with
// respect to g (clamp mask included): exactly what the backward warp multiplies the upstream gradient with
template <bool WANT_GRAD>
__device__ __forceinline__ float warp_sample(const float* __restrict__ src, const float (&g)[3], const Vol vol, float (&gm)[3]) {
    const AxisTap tx = axis_tap(g[0], vol.W), ty = axis_tap(g[1], vol.H), tz = axis_tap(g[2], vol.D);
    float acc = 0.0f, gix = 0.0f, giy = 0.0f, giz = 0.0f;
    // 32-bit element offsets from the (uniform) image base: a chain's volume has < 2^31 voxels (dims_ok).  Two integer
    // multiplies (quarter rate) instead of one per corner row: i1 is i0 or i0 + 1
    const unsigned hw = (unsigned)(vol.H * vol.W);
    const unsigned offz[2] = {(unsigned)tz.i0 * hw, (unsigned)tz.i0 * hw + (tz.i1 != tz.i0 ? hw : 0u)};
    const unsigned offy[2] = {(unsigned)ty.i0 * (unsigned)vol.W, (unsigned)ty.i0 * (unsigned)vol.W + (ty.i1 != ty.i0 ? (unsigned)vol.W : 0u)};
#pragma unroll
    for (int cz = 0; cz < 2; ++cz)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy) {
            const unsigned rowoff = offz[cz] + offy[cy];
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
                const float wx = cx ? tx.w1 : tx.w0, wy = cy ? ty.w1 : ty.w0, wz = cz ? tz.w1 : tz.w0;
                const float val = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(src) + (size_t)((rowoff + (unsigned)(cx ? tx.i1 : tx.i0)) * 4u));
                acc = __fadd_rn(acc, __fmul_rn(val, __fmul_rn(__fmul_rn(wx, wy), wz)));
                if (WANT_GRAD) {  // same expressions as warp_bwd_kernel
                    gix += (cx ? val : -val) * (wy * wz);
                    giy += (cy ? val : -val) * (wx * wz);
                    giz += (cz ? val : -val) * (wx * wy);
                }
            }
        }
    if (WANT_GRAD) {
        gm[0] = tx.gmul * gix;
        gm[1] = ty.gmul * giy;
        gm[2] = tz.gmul * giz;
    }
    return acc;
}

// what the fused last squaring step needs to finish the warp (exp_kernels.hip)
struct WarpArgs {
    const float* im;       // moving image, chain c at im + c * im_stride
    int64_t im_stride;
    const float* unif;     // injected jitter draws (C,3,D,H,W) or nullptr
    Jitter jt;
    float* warped;         // (C,1,D,H,W)
    float* gradm;          // d(warped)/d(d_last), (C,3,D,H,W) planar or interleaved; nullptr = not wanted
    int gradm_aos;
};

}  // namespace irs
