// z-marching separable stencils: the Sobolev smoothing, the LCC map and its adjoint.
//
// Schedule shared by the kernels in this file (same idea as the squaring-step kernels in exp_kernels.hip):
//   * a 256-thread workgroup owns a column tile of (x, y) and marches along z over a segment of planes;
//   * the z direction of every stencil lives in REGISTERS (a sliding window per thread), the x / y directions go through
//     one LDS plane per stage -- no 3-D LDS tile, no per-element div / mod / clamp in the inner loops;
//   * replicate padding is obtained by CLAMPED LOADS at staging time (a tile row / column / plane outside the volume holds
//     a copy of the border one), so the taps index the LDS planes directly;
//   * every input element is staged once per tile (+ the in-plane halo), the z run-in of a segment re-reads 2S planes.
// The 3-D-tile kernels these replace spent 300-800 VALU instructions per voxel on index arithmetic (rocprofv3
// SQ_INSTS_VALU); these need 50-250.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "scalar_kernels.h"

namespace irs {

namespace {

constexpr int kStBlock = 256;

__device__ __forceinline__ float ldg_off(const float* __restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

__device__ __forceinline__ void atomic_max_bits(unsigned* slot, float m) {  // m >= 0: uint order == float order
    if (__float_as_uint(m) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, __float_as_uint(m));
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// SobolevGrad.forward (utils/functions.py:98-109, utils/util.py:394-404): replicate-pad by S, then the (2S+1)-tap
// 1-D kernel along z, then y, then x (the reference's order; each pass accumulates taps in index order with fmaf,
// exactly like the per-axis kernel conv_axis_kernel, so the three implementations are bit-identical).
// Tile 32 x 16 columns (two outputs per thread); the z pass runs on the tile + S halo columns (1.63x for S = 3).
// Optionally publishes max|v_s| / 2^steps per channel = the displacement bound of d_0 in voxels (exp_kernels.hip).
// ------------------------------------------------------------------------------------------------
constexpr int SMX = 32, SMY = 16;

template <int S>
__global__ __launch_bounds__(kStBlock) void sobolev_march_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 Taps taps, Vol vol, unsigned* __restrict__ dmax0,
                                                                 float inv_pow, int seg_len, int nseg) {
    constexpr int NT = 2 * S + 1, PX = SMX + 2 * S, PY = SMY + 2 * S, PN = PX * PY;
    constexpr int NIT = (PN + kStBlock - 1) / kStBlock;
    constexpr int NY = PX * SMY, NITY = (NY + kStBlock - 1) / kStBlock;
    __shared__ float P1[PN];  // z-filtered plane over the tile + halo
    __shared__ float P2[NY];  // then y-filtered, tile rows only
    __shared__ float red[kStBlock / kWave];

    const int plane = blockIdx.z / nseg, seg = blockIdx.z % nseg;
    const int ox = blockIdx.x * SMX, oy = blockIdx.y * SMY;
    const int z0 = vol.z0 + seg * seg_len, z1 = min(z0 + seg_len, vol.z0 + vol.nz);
    const float* __restrict__ src = in + (int64_t)plane * vol.V;
    float* __restrict__ dst = out + (int64_t)plane * vol.V;
    const int64_t HW = (int64_t)vol.H * vol.W;

    float k[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) k[t] = taps.k[t];

    // columns of the z pass owned by this thread (clamped = replicate padding in x / y)
    unsigned coff[NIT];
    bool cval[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * kStBlock;
        const int px = i % PX, py = i / PX;
        const int cx = min(max(ox - S + px, 0), vol.W - 1), cy = min(max(oy - S + py, 0), vol.H - 1);
        cval[it] = i < PN;
        coff[it] = (unsigned)(cy * vol.W + cx) * 4u;
    }
    // y-pass elements owned by this thread
    int ybase[NITY];
#pragma unroll
    for (int it = 0; it < NITY; ++it) {
        const int j = threadIdx.x + it * kStBlock;
        ybase[it] = j < NY ? j : -1;  // P1 index of tap 0 is the same j: row qy + t of P1 <-> output row qy of P2
    }
    const int lx = threadIdx.x % SMX, ly = threadIdx.x / SMX;  // ly in [0, 8): outputs (lx, ly) and (lx, ly + 8)

    float win[NIT][NT];
    auto load_plane = [&](int p, float (&dstv)[NIT]) {
        const int pc = min(max(p, 0), vol.D - 1);  // replicate padding in z
        const float* __restrict__ base = src + (int64_t)pc * HW;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (cval[it]) dstv[it] = ldg_off(base, coff[it]);
    };
    // run-in: planes z0 - S .. z0 + S - 1 occupy window entries 1 .. 2S (entry 0 is shifted out first)
#pragma unroll
    for (int t = 1; t < NT; ++t) {
        float v[NIT];
        load_plane(z0 - S + t - 1, v);
#pragma unroll
        for (int it = 0; it < NIT; ++it) win[it][t] = v[it];
    }
    float pre[NIT];
    load_plane(z0 + S, pre);

    float m = 0.0f;
    for (int z = z0; z < z1; ++z) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
#pragma unroll
            for (int t = 0; t < NT - 1; ++t) win[it][t] = win[it][t + 1];
            win[it][NT - 1] = pre[it];
        }
        if (z + 1 < z1) load_plane(z + 1 + S, pre);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (!cval[it]) continue;
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc = fmaf(k[t], win[it][t], acc);
            P1[threadIdx.x + it * kStBlock] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
            if (ybase[it] < 0) continue;
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc = fmaf(k[t], P1[ybase[it] + t * PX], acc);
            P2[ybase[it]] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int yy = ly + o * (SMY / 2);
            const int gx = ox + lx, gy = oy + yy;
            if (gx >= vol.W || gy >= vol.H) continue;
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc = fmaf(k[t], P2[yy * PX + lx + t], acc);
            dst[(int64_t)z * HW + (unsigned)(gy * vol.W + gx)] = acc;
            m = fmaxf(m, fabsf(acc));
        }
        // P1 is rewritten after the next iteration's window shift; every thread has passed the second barrier by then,
        // i.e. all y-pass reads of P1 are done.  P2 is rewritten only after the next first barrier.
    }
    if (dmax0) {
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, kWave));
        if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            float mm = 0.0f;
            for (int w = 0; w < kStBlock / kWave; ++w) mm = fmaxf(mm, red[w]);
            mm *= inv_pow;  // |d_0| in voxels = |v_s| * (2/(n-1)/2^steps) * ((n-1)/2)
            atomic_max_bits(dmax0 + (plane / 3) * 4 + (plane % 3), mm);
        }
    }
}

void launch_sobolev_march(const float* in, float* out, const Taps& taps, int planes, Vol vol, unsigned* dmax0, int no_steps,
                          hipStream_t st) {
    static const int seg_env = getenv("IRS_SOBOLEV_SEG") ? atoi(getenv("IRS_SOBOLEV_SEG")) : 32;
    const int seg_len = seg_env;
    const int nseg = (vol.nz + seg_len - 1) / seg_len;
    const dim3 grid((vol.W + SMX - 1) / SMX, (vol.H + SMY - 1) / SMY, (unsigned)(nseg * planes));
    const float inv_pow = 1.0f / (float)(1 << no_steps);
#define IRS_SOB(SS) hipLaunchKernelGGL((sobolev_march_kernel<SS>), grid, dim3(kStBlock), 0, st, in, out, taps, vol, dmax0, inv_pow, seg_len, nseg)
    switch (taps.s) {
        case 1: IRS_SOB(1); break;
        case 2: IRS_SOB(2); break;
        case 3: IRS_SOB(3); break;
        default: IRS_SOB(4); break;
    }
#undef IRS_SOB
}

}  // namespace irs
