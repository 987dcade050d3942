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
// Tile SMX x SMY columns, SMX * SMY / 256 outputs per thread; the z pass runs on the tile + S halo columns.
// Optionally publishes max|v_s| / 2^steps per channel = the displacement bound of d_0 in voxels (exp_kernels.hip).
// ------------------------------------------------------------------------------------------------

template <int S, int SMX, int SMY>
__global__ __launch_bounds__(kStBlock) void sobolev_march_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 Taps taps, Vol vol, unsigned* __restrict__ dmax0,
                                                                 float inv_pow, int seg_len, int nseg) {
    constexpr int NT = 2 * S + 1, PX = SMX + 2 * S, PY = SMY + 2 * S, PN = PX * PY;
    constexpr int NIT = (PN + kStBlock - 1) / kStBlock;
    constexpr int NY = PX * SMY, NITY = (NY + kStBlock - 1) / kStBlock;
    __shared__ float P1[PN];  // z-filtered plane over the tile + halo
    __shared__ float P2[NY];  // then y-filtered, tile rows only
    __shared__ float red[kStBlock / kWave];

    const Blk3 blk = swizzled_block();
    const int plane = blk.z / nseg, seg = blk.z % nseg;
    const int ox = blk.x * SMX, oy = blk.y * SMY;
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
    constexpr int NOUT = SMX * SMY / kStBlock, ROWS = kStBlock / SMX;  // outputs per thread: rows ly + o * ROWS
    static_assert(SMX * SMY % kStBlock == 0 && kStBlock % SMX == 0, "tile shape");
    const int lx = threadIdx.x % SMX, ly = threadIdx.x / SMX;

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
        for (int o = 0; o < NOUT; ++o) {
            const int yy = ly + o * ROWS;
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
    const int seg_env = global_knobs().sobolev_seg;
    int seg_len = pick_seg_len(vol.nz, (int64_t)((vol.W + 31) / 32) * ((vol.H + 15) / 16) * planes, 4, seg_env);
    int nseg = (vol.nz + seg_len - 1) / seg_len;
    const float inv_pow = 1.0f / (float)(1 << no_steps);
    // big tiles (64 x 32, eight outputs per thread: 1.33x halo work in the z pass, a quarter of the barriers per output)
    // when they still fill the GPU, 32 x 16 otherwise
    const int64_t big_blocks = (int64_t)((vol.W + 63) / 64) * ((vol.H + 31) / 32) * nseg * planes;
    const int force = global_knobs().sobolev_tile;  // 1 small / 2 big: used by the parity test of the two shapes
    const bool big = force ? force == 2 : big_blocks >= 512;
    const int tx = big ? 64 : 32, ty = big ? 32 : 16;
    if (global_knobs().seg_fit && seg_env <= 0) {  // segment length for the chosen shape from the resident-set cost model (run-in 2 s planes)
        static int cache_b = 0, cache_s = 0;
        const int64_t res = big ? resident_blocks((const void*)sobolev_march_kernel<3, 64, 32>, kStBlock, &cache_b)
                                : resident_blocks((const void*)sobolev_march_kernel<3, 32, 16>, kStBlock, &cache_s);
        if (res > 0) {
            seg_len = pick_seg_len_fit(vol.nz, 0, (int64_t)((vol.W + tx - 1) / tx) * ((vol.H + ty - 1) / ty) * planes, 4, 2 * taps.s, res, 0);
            nseg = (vol.nz + seg_len - 1) / seg_len;
        }
    }
    const dim3 grid((vol.W + tx - 1) / tx, (vol.H + ty - 1) / ty, (unsigned)(nseg * planes));
#define IRS_SOB(SS)                                                                                                        \
    if (big) hipLaunchKernelGGL((sobolev_march_kernel<SS, 64, 32>), grid, dim3(kStBlock), 0, st, in, out, taps, vol, dmax0, \
                                inv_pow, seg_len, nseg);                                                                   \
    else hipLaunchKernelGGL((sobolev_march_kernel<SS, 32, 16>), grid, dim3(kStBlock), 0, st, in, out, taps, vol, dmax0,    \
                            inv_pow, seg_len, nseg)
    switch (taps.s) {
        case 1: IRS_SOB(1); break;
        case 2: IRS_SOB(2); break;
        case 3: IRS_SOB(3); break;
        default: IRS_SOB(4); break;
    }
#undef IRS_SOB
}

// ------------------------------------------------------------------------------------------------
// SGLD.forward fused into the smoothing kernel (utils/functions.py:76-84 + 98-109): the perturbed velocity v + sqrt(2 tau) sigma eps
// is formed while the planes are staged and never written to memory (the two-kernel form moved 878 MB for a 403 MB stage at
// 256^3).  One Philox4x32-10 call yields the six normals of a voxel PAIR (z even, z + 1) x three channels -- exactly the mapping
// of perturb_kernel (field_kernels.hip), so a chain draws the same noise with either form -- which is why this kernel takes
// all three channels of a tile at once: 512 threads, 32 x 32 columns, the z windows of 3 columns x 3 channels per thread in
// registers, the normals of the odd plane of the pair generated last kept for the next iteration.  Replicate padding: a plane /
// column outside the volume is a copy of the border one INCLUDING its noise (it is the perturbed field that is padded), which
// the clamped coordinates of the counter give for free.  Same tap order as sobolev_march_kernel: bit-identical results.
// ------------------------------------------------------------------------------------------------
constexpr int kPsBlock = 512, PSX = 32;

struct NoiseSrc {
    const float* sigma;       // (C,3,D,H,W) or nullptr (= 1)
    const float* eps;         // injected standard normals or nullptr (Philox)
    float amp;                // sqrt(2 tau)
    uint64_t seed, iteration;
    const uint64_t* dev_iter; // device-side Philox counter (overrides `iteration`)
};

// PSY rows per tile: 32, or -- a sigma FIELD with generated noise, chains started from the VI posterior (MCMC_init 'VI',
// utils/functions.py:78-84) -- 16: on the 32 x 32 tile that variant needs 149 VGPRs (nine more live values across the Philox
// rounds, and 64-bit column pointers for one more array), one workgroup per CU; two columns per thread instead of three bring it
// to the register count of the sigma = 1 kernel (two workgroups per CU) for 16 % more staged elements per output.
template <int S, bool SIGMA, bool EPS, int PSY>
__global__ __launch_bounds__(kPsBlock) void perturb_sobolev_march_kernel(const float* __restrict__ v, NoiseSrc ns,
                                                                         float* __restrict__ out, Taps taps, Vol vol,
                                                                         unsigned* __restrict__ dmax0, float inv_pow,
                                                                         int seg_len, int nseg) {
    constexpr int NT = 2 * S + 1, PX = PSX + 2 * S, PY = PSY + 2 * S, PN = PX * PY;
    constexpr int NIT = (PN + kPsBlock - 1) / kPsBlock;
    constexpr int NY = PX * PSY, NITY = (NY + kPsBlock - 1) / kPsBlock;
    constexpr int NOUT = PSX * PSY / kPsBlock, ROWS = kPsBlock / PSX;
    static_assert(PSX * PSY % kPsBlock == 0 && kPsBlock % PSX == 0, "tile shape");
    __shared__ float P1[3 * PN];
    __shared__ float P2[3 * NY];
    __shared__ float red[3 * (kPsBlock / kWave)];

    const Blk3 blk = swizzled_block();
    const int chain = blk.z / nseg, seg = blk.z % nseg;
    const int ox = blk.x * PSX, oy = blk.y * PSY;
    const int z0 = vol.z0 + seg * seg_len, z1 = min(z0 + seg_len, vol.z0 + vol.nz);
    const int64_t HW = (int64_t)vol.H * vol.W, cb = (int64_t)chain * 3 * vol.V;
    const float* __restrict__ src = v + cb;
    const float* __restrict__ sgp = SIGMA ? ns.sigma + cb : nullptr;
    const float* __restrict__ epp = EPS ? ns.eps + cb : nullptr;
    float* __restrict__ dst = out + cb;
    const uint64_t iter = EPS ? 0ull : (ns.dev_iter ? *ns.dev_iter : ns.iteration);
    const uint64_t chain_base = (uint64_t)chain * (uint64_t)vol.Vg;

    float k[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) k[t] = taps.k[t];

    unsigned coff[NIT];  // element offset (y * W + x, clamped = replicate padding in x / y) of the columns this thread owns
    bool cval[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * kPsBlock;
        const int px = i % PX, py = i / PX;
        const int cx = min(max(ox - S + px, 0), vol.W - 1), cy = min(max(oy - S + py, 0), vol.H - 1);
        cval[it] = i < PN;
        coff[it] = (unsigned)(cy * vol.W + cx);
    }
    const int lx = threadIdx.x % PSX, ly = threadIdx.x / PSX;

    float win[NIT][3][NT];
    float pend[NIT][3];                         // the perturbed plane in flight (enters the windows at the next shift)
    float stash[NIT][3];                        // normals of the odd plane of the pair generated last
    int stash_pair = -(1 << 30);
    int plane_in_flight = -(1 << 30);           // clamped index of the plane held in `pend`
    // issue the loads of plane p, generate its noise meanwhile (the Philox rounds hide the load latency), combine:
    // SGLD.forward with perturb_kernel's arithmetic, v + (amp * sigma) * n
    auto fetch = [&](int p) {
        const int pc = min(max(p, 0), vol.D - 1);
        if (pc == plane_in_flight) return;      // replicate padding in z: the border plane again, noise included
        plane_in_flight = pc;
        const int64_t zo = (int64_t)pc * HW;
        float tv[NIT][3], tn[NIT][3], tsg[NIT][3];
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (cval[it]) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    tv[it][c] = (src + c * vol.V + zo)[coff[it]];
                    if (SIGMA) tsg[it][c] = (sgp + c * vol.V + zo)[coff[it]];
                    if (EPS) tn[it][c] = (epp + c * vol.V + zo)[coff[it]];
                }
            }
        if (!EPS) {
            const int pair = pc >> 1;
            if (pair != stash_pair) {           // (uniform branch)
                stash_pair = pair;
                const uint64_t za = (uint64_t)(2 * pair) * (uint64_t)HW;
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    if (cval[it]) {
                        const uint64_t idx = chain_base + za + coff[it];  // global voxel index of the pair's even plane
                        const U4 r = philox4x32_10(U4{(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)iter, 0x5347u ^ (uint32_t)(iter >> 32)},
                                                   (uint32_t)ns.seed, (uint32_t)(ns.seed >> 32));
                        uint32_t w[6];
                        split21(r, w);
                        float e0, e1, e2, o0, o1, o2;  // even plane: channels 0..2, odd plane: channels 0..2 (perturb_kernel's order)
                        box_muller21(w[0], w[1], e0, e1);
                        box_muller21(w[2], w[3], e2, o0);
                        box_muller21(w[4], w[5], o1, o2);
                        const bool odd = (pc & 1) != 0;
                        tn[it][0] = odd ? o0 : e0;
                        tn[it][1] = odd ? o1 : e1;
                        tn[it][2] = odd ? o2 : e2;
                        stash[it][0] = o0;
                        stash[it][1] = o1;
                        stash[it][2] = o2;
                    }
            } else {                            // the odd plane of the pair generated for the plane before
#pragma unroll
                for (int it = 0; it < NIT; ++it)
#pragma unroll
                    for (int c = 0; c < 3; ++c) tn[it][c] = stash[it][c];
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                pend[it][c] = cval[it] ? __fadd_rn(tv[it][c], __fmul_rn(__fmul_rn(ns.amp, SIGMA ? tsg[it][c] : 1.0f), tn[it][c])) : 0.0f;
    };
    auto perturbed = [&](int it, int c) { return pend[it][c]; };

    // run-in: planes z0 - S .. z0 + S - 1 occupy window entries 1 .. 2S (entry 0 is shifted out first)
#pragma unroll
    for (int t = 1; t < NT; ++t) {
        fetch(z0 - S + t - 1);
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int c = 0; c < 3; ++c) win[it][c][t] = perturbed(it, c);
    }
    fetch(z0 + S);

    float m[3] = {0.0f, 0.0f, 0.0f};
    for (int z = z0; z < z1; ++z) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int t = 0; t < NT - 1; ++t) win[it][c][t] = win[it][c][t + 1];
                win[it][c][NT - 1] = perturbed(it, c);
            }
        if (z + 1 < z1) fetch(z + 1 + S);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (!cval[it]) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc = fmaf(k[t], win[it][c][t], acc);
                P1[c * PN + threadIdx.x + it * kPsBlock] = acc;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
            const int j = threadIdx.x + it * kPsBlock;
            if (j >= NY) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc = fmaf(k[t], P1[c * PN + j + t * PX], acc);
                P2[c * NY + j] = acc;
            }
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const int yy = ly + o * ROWS;
            const int gx = ox + lx, gy = oy + yy;
            if (gx >= vol.W || gy >= vol.H) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc = fmaf(k[t], P2[c * NY + yy * PX + lx + t], acc);
                dst[c * vol.V + (int64_t)z * HW + (unsigned)(gy * vol.W + gx)] = acc;
                m[c] = fmaxf(m[c], fabsf(acc));
            }
        }
        // (P1 is rewritten after the next iteration's window shift, P2 only after the next first barrier: see sobolev_march_kernel)
    }
    if (dmax0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float mc = m[c];
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) mc = fmaxf(mc, __shfl_down(mc, off, kWave));
            if ((threadIdx.x & (kWave - 1)) == 0) red[c * (kPsBlock / kWave) + threadIdx.x / kWave] = mc;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            float mm = 0.0f;
            for (int w = 0; w < kPsBlock / kWave; ++w) mm = fmaxf(mm, red[threadIdx.x * (kPsBlock / kWave) + w]);
            atomic_max_bits(dmax0 + chain * 4 + threadIdx.x, mm * inv_pow);  // |d_0| in voxels (see sobolev_march_kernel)
        }
    }
}

void launch_perturb_sobolev_march(const float* v, const float* sigma, const float* eps, float amp, float* out, const Taps& taps,
                                  int C, Vol vol, unsigned* dmax0, int no_steps, uint64_t seed, uint64_t iteration,
                                  const uint64_t* dev_iteration, hipStream_t st) {
    // the sigma-field variant runs on 32 x 16 tiles (see the kernel); IRS_PS_ROWS = 16 / 32 forces one shape for both (measurements)
    const int forced_rows = global_knobs().ps_rows;
    const bool half = !eps && (forced_rows == 16 || (forced_rows != 32 && sigma));
    const int psy = half ? 16 : 32;
    const int ntx = (vol.W + PSX - 1) / PSX, nty = (vol.H + psy - 1) / psy;
    int seg_len = pick_seg_len(vol.nz, (int64_t)ntx * nty * C, 4, global_knobs().sobolev_seg, 512);
    if (global_knobs().seg_fit && global_knobs().sobolev_seg <= 0) {
        static int cache = 0, cache_half = 0;
        const int64_t res = half ? resident_blocks(sigma ? (const void*)perturb_sobolev_march_kernel<3, true, false, 16> : (const void*)perturb_sobolev_march_kernel<3, false, false, 16>, kPsBlock, &cache_half)
                                 : resident_blocks((const void*)perturb_sobolev_march_kernel<3, false, false, 32>, kPsBlock, &cache);
        if (res > 0) seg_len = pick_seg_len_fit(vol.nz, 0, (int64_t)ntx * nty * C, 4, 2 * taps.s, res, 0);
    }
    const int nseg = (vol.nz + seg_len - 1) / seg_len;
    const float inv_pow = 1.0f / (float)(1 << no_steps);
    const dim3 grid(ntx, nty, (unsigned)(nseg * C));
    if (global_knobs().launch_log) {
        static int cache_l = 0, cache_lh = 0;
        log_launch(half ? "perturb_sobolev_march_kernel<rows=16>" : "perturb_sobolev_march_kernel<rows=32>", PSX, psy, (int64_t)ntx * nty * nseg * C, kPsBlock,
                   seg_len, 2 * taps.s, vol.nz, C,
                   half ? resident_blocks((const void*)perturb_sobolev_march_kernel<3, false, false, 16>, kPsBlock, &cache_lh)
                        : resident_blocks((const void*)perturb_sobolev_march_kernel<3, false, false, 32>, kPsBlock, &cache_l));
    }
    const NoiseSrc ns{sigma, eps, amp, seed, iteration, dev_iteration};
#define IRS_PS(SS, SG, EP, PY) hipLaunchKernelGGL((perturb_sobolev_march_kernel<SS, SG, EP, PY>), grid, dim3(kPsBlock), 0, st, v, ns, out, taps, vol, dmax0, inv_pow, seg_len, nseg)
#define IRS_PS2(SS)                                   \
    if (sigma && eps) IRS_PS(SS, true, true, 32);             \
    else if (sigma && half) IRS_PS(SS, true, false, 16);      \
    else if (sigma) IRS_PS(SS, true, false, 32);              \
    else if (eps) IRS_PS(SS, false, true, 32);                \
    else if (half) IRS_PS(SS, false, false, 16);              \
    else IRS_PS(SS, false, false, 32)
    switch (taps.s) {
        case 1: IRS_PS2(1); break;
        case 2: IRS_PS2(2); break;
        case 3: IRS_PS2(3); break;
        default: IRS_PS2(4); break;
    }
#undef IRS_PS2
#undef IRS_PS
}

// ------------------------------------------------------------------------------------------------
// LCC map (model/loss.py:53-59,102-111):  u = box(I)/n,  w = I - u,  var = box(w^2)/n,  sigma = sqrt(var + 1e-10),
// out = w / sigma  (MAP: fhat - w / sigma, the fixed side being pre-normalised once).  box = (2S+1)^3 all-ones filter
// with REPLICATE padding -- of I for the first box, of w^2 for the second (so the second box indexes clamped
// COORDINATES: a w computed from replicated I outside the volume would be a different number).
//
// One pipeline iteration per staged input plane `pin` (segment [z0, z1) -> pin = z0 - 2S .. z1 - 1 + 2S):
//   A  I(clamp(pin)) over the tile + 2S halo, clamped loads -> LDS plane LA
//   B  per element of the tile + S region: in-plane (2S+1)^2 sum of LA -> register ring over 2S+1 planes -> u, w at
//      plane pu = pin - S -> LDS ring LW (S+2 planes)
//   C  per output: in-plane (2S+1)^2 sum of w^2 (clamped coordinates) -> register ring (replicated at the volume ends)
//      -> var, sigma, output at plane pv = pin - 2S.
// Two barriers per plane.  Tile 32 x 16, two outputs per thread.
// ------------------------------------------------------------------------------------------------
#ifndef IRS_LMX
#define IRS_LMX 32
#define IRS_LMY 16
#endif
constexpr int LMX = IRS_LMX, LMY = IRS_LMY;

template <int S, bool MAP>
__global__ __launch_bounds__(kStBlock) void lcc_fwd_march_kernel(const float* __restrict__ fhat, int64_t fhat_stride,
                                                                 const float* __restrict__ im, float* __restrict__ out,
                                                                 float* __restrict__ sigma_out, Vol vol, int seg_len, int nseg) {
    constexpr int NT = 2 * S + 1, NS = S + 2;
    constexpr int AX = LMX + 4 * S, AY = LMY + 4 * S, AN = AX * AY, NITA = (AN + kStBlock - 1) / kStBlock;
    constexpr int BX = LMX + 2 * S, BY = LMY + 2 * S, BN = BX * BY, NITB = (BN + kStBlock - 1) / kStBlock;
    __shared__ float LA[AN];
    __shared__ float LW[NS * BN];

    const Blk3 blk = swizzled_block();
    const int chain = blk.z / nseg, seg = blk.z % nseg;
    const int ox = blk.x * LMX, oy = blk.y * LMY;
    const int z0 = vol.z0 + seg * seg_len, z1 = min(z0 + seg_len, vol.z0 + vol.nz);
    const int64_t HW = (int64_t)vol.H * vol.W;
    const float* __restrict__ src = im + (int64_t)chain * vol.V;
    const float n = (float)(NT * NT * NT);

    unsigned aoff[NITA];
    bool aval[NITA];
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
        const int i = threadIdx.x + it * kStBlock;
        const int px = i % AX, py = i / AX;
        const int cx = min(max(ox - 2 * S + px, 0), vol.W - 1), cy = min(max(oy - 2 * S + py, 0), vol.H - 1);
        aval[it] = i < AN;
        aoff[it] = (unsigned)(cy * vol.W + cx) * 4u;
    }
    int bwin[NITB];  // LA index of the top-left tap of the element's window; -1: no element
#pragma unroll
    for (int it = 0; it < NITB; ++it) {
        const int e = threadIdx.x + it * kStBlock;
        bwin[it] = e < BN ? (e / BX) * AX + (e % BX) : -1;
    }
    const int lx = threadIdx.x % LMX, ly = threadIdx.x / LMX;
    const int gx = ox + lx;
    int xo[NT], yo[2][NT];
    bool oval[2];
#pragma unroll
    for (int i = 0; i < NT; ++i) xo[i] = min(max(gx + i - S, 0), vol.W - 1) - (ox - S);
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int gy = oy + ly + o * (LMY / 2);
        oval[o] = gx < vol.W && gy < vol.H;
#pragma unroll
        for (int j = 0; j < NT; ++j) yo[o][j] = (min(max(gy + j - S, 0), vol.H - 1) - (oy - S)) * BX;
    }

    float ring1[NITB][NT], iring[NITB][S + 1], ring2[2][NT];
#pragma unroll
    for (int it = 0; it < NITB; ++it) {
#pragma unroll
        for (int t = 0; t < NT; ++t) ring1[it][t] = 0.0f;
#pragma unroll
        for (int t = 0; t <= S; ++t) iring[it][t] = 0.0f;
    }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int t = 0; t < NT; ++t) ring2[o][t] = 0.0f;

    float pre[NITA];
    auto load_plane = [&](int p) {
        const float* __restrict__ base = src + (int64_t)min(max(p, 0), vol.D - 1) * HW;
#pragma unroll
        for (int it = 0; it < NITA; ++it)
            if (aval[it]) pre[it] = ldg_off(base, aoff[it]);
    };
    const int pfirst = z0 - 2 * S, plast = z1 - 1 + 2 * S;
    // MAP: the fixed-image value each output is subtracted from, requested one plane ahead and BEFORE that step's prefetch (a load
    // issued in phase C could only be waited for together with the prefetch: in-order counter); two buffers, used alternately
    struct OwnIn {
        float f[2];
    };
    auto load_own = [&](int pv, OwnIn& q) {
        if (!MAP || pv < z0 || pv >= z1) return;
#pragma unroll
        for (int o = 0; o < 2; ++o)
            if (oval[o]) q.f[o] = fhat[(int64_t)chain * fhat_stride + (int64_t)pv * HW + (unsigned)((oy + ly + o * (LMY / 2)) * vol.W + gx)];
    };
    auto step = [&](int pin, const OwnIn& cur, OwnIn& nxt) {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): phase A needs the newest loads anyway; unconditional, so that `cur` counts as arrived
        // ---- A
#pragma unroll
        for (int it = 0; it < NITA; ++it)
            if (aval[it]) LA[threadIdx.x + it * kStBlock] = pre[it];
        load_own(pin + 1 - 2 * S, nxt);
        if (pin < plast) load_plane(pin + 1);
        __syncthreads();
        // ---- B
        const int pu = pin - S;
        const bool pu_real = pu >= 0 && pu < vol.D;
        const int slot_u = ((pu % NS) + NS) % NS;
#pragma unroll
        for (int it = 0; it < NITB; ++it) {
            if (bwin[it] < 0) continue;
            float sxy = 0.0f;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i) sxy += LA[bwin[it] + j * AX + i];
#pragma unroll
            for (int t = 0; t < NT - 1; ++t) ring1[it][t] = ring1[it][t + 1];
            ring1[it][NT - 1] = sxy;
#pragma unroll
            for (int t = 0; t < S; ++t) iring[it][t] = iring[it][t + 1];
            iring[it][S] = LA[bwin[it] + S * AX + S];
            if (pu_real && pin >= z0) {  // ring1 holds planes pin - 2S .. pin = pu - S .. pu + S
                float sum = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) sum += ring1[it][t];
                LW[slot_u * BN + threadIdx.x + it * kStBlock] = iring[it][0] - sum / n;
            }
        }
        __syncthreads();
        // ---- C
        const int pv = pin - 2 * S;
        const int slot_v = ((pv % NS) + NS) % NS;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            if (!oval[o]) continue;
            if (pu_real && pin >= z0) {
                float q = 0.0f;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        const float w = LW[slot_u * BN + yo[o][j] + xo[i]];
                        q = fmaf(w, w, q);
                    }
                if (pu == 0) {  // volume start: planes -S .. -1 replicate plane 0
#pragma unroll
                    for (int t = 0; t < NT; ++t) ring2[o][t] = q;
                } else {
#pragma unroll
                    for (int t = 0; t < NT - 1; ++t) ring2[o][t] = ring2[o][t + 1];
                    ring2[o][NT - 1] = q;
                }
            } else if (pu >= vol.D) {  // volume end: replicate plane D - 1
#pragma unroll
                for (int t = 0; t < NT - 1; ++t) ring2[o][t] = ring2[o][t + 1];
            }
            if (pv >= z0 && pv < z1) {
                float var = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) var += ring2[o][t];
                // 1-ulp hardware sqrt / reciprocal: relative errors of 1e-7 on sigma and z (the mean above keeps its exact division:
                // an error there is amplified by 1 / sigma in flat regions)
                const float sigma = __builtin_amdgcn_sqrtf(fmaf(var, 1.0f / n, 1e-10f));
                const int yy = ly + o * (LMY / 2);
                const float wc = LW[slot_v * BN + (yy + S) * BX + lx + S];
                const float r = wc * __builtin_amdgcn_rcpf(sigma);
                const int64_t g = (int64_t)pv * HW + (unsigned)((oy + yy) * vol.W + gx);
                out[(int64_t)chain * vol.V + g] = MAP ? cur.f[o] - r : r;
                if (sigma_out) sigma_out[(int64_t)chain * vol.V + g] = sigma;
            }
        }
    };
    OwnIn qa = {}, qb = {};
    load_own(pfirst - 2 * S, qa);
    load_plane(pfirst);
    for (int pin = pfirst; pin <= plast; pin += 2) {
        step(pin, qa, qb);
        if (pin + 1 <= plast) step(pin + 1, qb, qa);
    }
}

// the LCC kernels re-read 4S planes per segment; measured at 128^3: 8-plane segments (1.5x staging, 4x the workgroups)
// still beat 16 and 32
template <int S, bool G, bool BATCH>
__global__ void lcc_data_bwd_march_kernel(const float*, const float*, const float*, const uint8_t*, const float*, const DevState*, int,
                                          float*, double*, Vol, int, int, int, int64_t, int64_t);
static int lcc_seg_len(Vol vol, int C, bool bwd) {
    const int seg_env = global_knobs().lcc_seg;
    const int64_t per_layer = (int64_t)((vol.W + LMX - 1) / LMX) * ((vol.H + LMY - 1) / LMY) * C;
    // measured at 256^3: 2048 workgroups beat 1024; at 128^3: 4-plane segments beat 8 (data stage 0.144 -> 0.132 ms)
    int len = pick_seg_len(vol.nz, per_layer, 4, seg_env, 2048);
    // ... and, with the resident-set cost model of the squaring steps (common.h: pick_seg_len_fit; run-in 4 S planes), per kernel:
    // 192^3 data stage 0.267 -> 0.248 ms, 256^3 0.511 -> 0.496, 128^3 unchanged (the rule's 4-plane segments)
    if (global_knobs().seg_fit && seg_env <= 0) {
        static int cache_f = 0, cache_b = 0;
        const int64_t res = bwd ? resident_blocks((const void*)lcc_data_bwd_march_kernel<1, false, false>, kStBlock, &cache_b)
                                : resident_blocks((const void*)lcc_fwd_march_kernel<1, true>, kStBlock, &cache_f);
        if (res > 0) len = pick_seg_len_fit(vol.nz, 0, per_layer, 4, 4, res, 0);
    }
    return len;
}

void launch_lcc_fwd_march(const float* fhat, int64_t fhat_stride, const float* im, float* z, float* sigma_out, int s, int C,
                          Vol vol, hipStream_t st) {
    const int seg_len = lcc_seg_len(vol, C, false);
    const int nseg = (vol.nz + seg_len - 1) / seg_len;
    const dim3 grid((vol.W + LMX - 1) / LMX, (vol.H + LMY - 1) / LMY, (unsigned)(nseg * C));
    const bool map = fhat != nullptr;
    if (global_knobs().launch_log) {
        static int cache_l = 0;
        log_launch("lcc_fwd_march_kernel", LMX, LMY, (int64_t)grid.x * grid.y * grid.z, kStBlock, seg_len, 4 * s, vol.nz, C,
                   resident_blocks((const void*)lcc_fwd_march_kernel<1, true>, kStBlock, &cache_l));
    }
#define IRS_LCC_FWD(SS)                                                                                                    \
    if (map) hipLaunchKernelGGL((lcc_fwd_march_kernel<SS, true>), grid, dim3(kStBlock), 0, st, fhat, fhat_stride, im, z,  \
                                sigma_out, vol, seg_len, nseg);                                                           \
    else hipLaunchKernelGGL((lcc_fwd_march_kernel<SS, false>), grid, dim3(kStBlock), 0, st, fhat, fhat_stride, im, z,     \
                            sigma_out, vol, seg_len, nseg);
    if (s == 1) { IRS_LCC_FWD(1) } else { IRS_LCC_FWD(2) }
#undef IRS_LCC_FWD
}

// ------------------------------------------------------------------------------------------------
// Data term backward, fused: GMM d(-log p)/dz (with the UPDATED mixture, trainer.py:316-330), the NLL partial sums, and
// the adjoint of the LCC map down to g_M = dL/d(warped image).
//   w' = fhat - z (= w / sigma),  gw = -gz,  gvar = -gw w' / (2 sigma^2),  pw = gw / sigma,  a2 = 2 w' sigma / n
//   T1 = box^T(gvar),  ga = pw + a2 T1,  T2 = box^T(ga),  g_M = ga - T2 / n
// box^T is the adjoint of the replicate-padded box filter: a plain (2S+1)-sum over the voxels that exist (nothing is
// replicated, out-of-volume terms are zero) plus, on the first / last index of an axis, the extra weights of the padding
// that folded onto it:  index 0 gets (S - x) more of x = 0..S-1,  index n-1 gets (S - (n-1-x)) more of x = n-S..n-1.
//
// Pipeline per staged plane `pin` (segment [z0, z1) -> pin = z0 - 2S .. z1 - 1 + 2S), tile 32 x 16:
//   A  per element of the tile + 2S halo: mixture evaluation -> gvar (LDS plane), pw / a2 (LDS rings, tile + S region)
//   B  per element of the tile + S region: weighted in-plane sum of gvar -> register ring -> T1, ga at plane q = pin - S
//   C  per output: weighted in-plane sum of ga -> register ring -> T2, g_M at plane r = pin - 2S.
// ------------------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void adjoint_weights(int g, int n, float (&w)[2 * S + 1]) {
#pragma unroll
    for (int i = 0; i <= 2 * S; ++i) {
        float v = 1.0f;
        if (g == 0 && i >= S && i < 2 * S) v += (float)(2 * S - i);
        if (g == n - 1 && i >= 1 && i <= S) v += (float)i;
        w[i] = v;
    }
}

// sum of a register ring over planes c-S .. c+S with the z border weights of box^T for centre plane c
template <int S>
__device__ __forceinline__ float adjoint_ring_sum(const float (&ring)[2 * S + 1], int c, int D) {
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t <= 2 * S; ++t) acc += ring[t];
    if (c == 0) {
#pragma unroll
        for (int x = 0; x < S; ++x) acc += (float)(S - x) * ring[x + S];
    }
    if (c == D - 1) {
#pragma unroll
        for (int t = 1; t <= S; ++t) acc += (float)t * ring[t];
    }
    return acc;
}

// BATCH: several chains in one launch.  A template parameter, not a run-time switch: with the mixture constants behind a pointer that
// is one of two (DevState::A or a chain's snapshot) the compiler fetched them with vector loads inside the plane loop -- 132 global
// loads instead of 24 in the kernel, 187 instead of 141 us at 256^3 with ONE chain; each instantiation reads its constants through
// `state` directly and keeps the scalar loads
template <int S, bool EXPLICIT_GZ, bool BATCH>
__global__ __launch_bounds__(kStBlock) void lcc_data_bwd_march_kernel(const float* __restrict__ fhat,
                                                                      const float* __restrict__ z,
                                                                      const float* __restrict__ sigma_m,
                                                                      const uint8_t* __restrict__ mask,
                                                                      const float* __restrict__ gz_in,
                                                                      const DevState* __restrict__ state, int chain,
                                                                      float* __restrict__ g_m, double* __restrict__ nll_out,
                                                                      Vol vol, int seg_len, int nseg, int batch,
                                                                      int64_t f_stride, int64_t m_stride) {
    constexpr int NT = 2 * S + 1, NS1 = S + 1, NS2 = S + 2;
    constexpr int AX = LMX + 4 * S, AY = LMY + 4 * S, AN = AX * AY, NITA = (AN + kStBlock - 1) / kStBlock;
    constexpr int BX = LMX + 2 * S, BY = LMY + 2 * S, BN = BX * BY, NITB = (BN + kStBlock - 1) / kStBlock;
    __shared__ float LGV[AN];
    __shared__ float LPW[NS1 * BN], LA2[NS1 * BN];
    __shared__ float LG[NS2 * BN];
    __shared__ double red[kStBlock / kWave];

    const Blk3 blk = swizzled_block();
    int seg = blk.z;
    // `batch` chains in one launch (grid.z = nseg x batch, chain-major): chain `chain + cl` against the snapshot its GMM step left
    // (scalar_kernels.h: DevState::snapA), planes and partial sums at its chain offset -- the same values, sums and slots as `batch`
    // launches of one chain each, every one right behind its chain's step
    if (BATCH) {
        const int cl = __builtin_amdgcn_readfirstlane(blk.z / nseg);  // (uniform, but the quotient comes out of the vector ALU)
        seg = blk.z - cl * nseg;
        chain += cl;
        fhat += cl * f_stride;
        mask += cl * m_stride;
        z += cl * vol.V;
        sigma_m += cl * vol.V;
        g_m += cl * vol.V;
    }
    const float* __restrict__ mixA = BATCH ? state->snapA[chain] : state->A;
    const float* __restrict__ mix_iv = BATCH ? state->snap_inv_var[chain] : state->inv_var;
    const int ox = blk.x * LMX, oy = blk.y * LMY;
    const int z0 = vol.z0 + seg * seg_len, z1 = min(z0 + seg_len, vol.z0 + vol.nz);
    const int64_t HW = (int64_t)vol.H * vol.W;
    const float n = (float)(NT * NT * NT);
    const float alpha = EXPLICIT_GZ ? 1.0f : (float)state->sc.alpha[chain];
    const bool k_le4 = state->K <= 4;

    // stage A elements
    unsigned aoff[NITA];
    int abidx[NITA];  // index in the tile + S region, -1 if the element is halo-only
    bool aexist[NITA], ainside[NITA], acentre[NITA];
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
        const int i = threadIdx.x + it * kStBlock;
        const int px = i % AX, py = i / AX;
        const int gx = ox - 2 * S + px, gy = oy - 2 * S + py;
        aexist[it] = i < AN;
        ainside[it] = aexist[it] && gx >= 0 && gx < vol.W && gy >= 0 && gy < vol.H;
        aoff[it] = ainside[it] ? (unsigned)(gy * vol.W + gx) : 0u;
        const bool inb = aexist[it] && px >= S && px < AX - S && py >= S && py < AY - S;
        abidx[it] = inb ? (py - S) * BX + (px - S) : -1;
        acentre[it] = ainside[it] && px >= 2 * S && px < AX - 2 * S && py >= 2 * S && py < AY - 2 * S;
    }
    // stage B elements
    int bwin[NITB];
    float bwx[NITB][NT], bwy[NITB][NT];
#pragma unroll
    for (int it = 0; it < NITB; ++it) {
        const int e = threadIdx.x + it * kStBlock;
        const int bx = e % BX, by = e / BX;
        bwin[it] = e < BN ? by * AX + bx : -1;
        adjoint_weights<S>(ox - S + bx, vol.W, bwx[it]);
        adjoint_weights<S>(oy - S + by, vol.H, bwy[it]);
    }
    // stage C outputs
    const int lx = threadIdx.x % LMX, ly = threadIdx.x / LMX;
    const int gxo = ox + lx;
    float cwx[NT], cwy[2][NT];
    bool oval[2];
    adjoint_weights<S>(gxo, vol.W, cwx);
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int gy = oy + ly + o * (LMY / 2);
        oval[o] = gxo < vol.W && gy < vol.H;
        adjoint_weights<S>(gy, vol.H, cwy[o]);
    }

    float ring1[NITB][NT], ring2[2][NT];
#pragma unroll
    for (int it = 0; it < NITB; ++it)
#pragma unroll
        for (int t = 0; t < NT; ++t) ring1[it][t] = 0.0f;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int t = 0; t < NT; ++t) ring2[o][t] = 0.0f;

    double nll = 0.0;
    float pz[NITA], ps[NITA], pf[NITA], pg[NITA];
    uint8_t pm[NITA];
    auto load_plane = [&](int p) {
        if (p < 0 || p >= vol.D) return;
        const int64_t zo = (int64_t)p * HW;
#pragma unroll
        for (int it = 0; it < NITA; ++it) {
            if (!ainside[it]) continue;
            const unsigned g = aoff[it];
            pz[it] = (z + zo)[g];
            ps[it] = (sigma_m + zo)[g];
            pf[it] = (fhat + zo)[g];
            if (EXPLICIT_GZ) pg[it] = (gz_in + zo)[g];
            else pm[it] = (mask + zo)[g];
        }
    };
    const int pfirst = z0 - 2 * S, plast = z1 - 1 + 2 * S;
    load_plane(pfirst);
    for (int pin = pfirst; pin <= plast; ++pin) {
        // ---- A
        const bool pin_real = pin >= 0 && pin < vol.D;
        const int slot_a = ((pin % NS1) + NS1) % NS1;
#pragma unroll
        for (int it = 0; it < NITA; ++it) {
            if (!aexist[it]) continue;
            float gvar = 0.0f, pw = 0.0f, a2 = 0.0f;
            if (pin_real && ainside[it]) {
                const float zz = pz[it], sg = ps[it];
                float gzv;
                if (EXPLICIT_GZ) {
                    gzv = pg[it];
                } else {
                    gzv = 0.0f;
                    if (pm[it]) {
                        // (uniform branch: the component loops of a K <= 4 mixture are half as long)
                        const MixEval e = k_le4 ? mix_eval_with<false, 4>(zz, state, mixA, mix_iv, nullptr, nullptr)
                                                 : mix_eval_with<false>(zz, state, mixA, mix_iv, nullptr, nullptr);
                        gzv = alpha * e.gz;
                        if (acentre[it] && pin >= z0 && pin < z1) nll += (double)e.nll;
                    }
                }
                const float w = pf[it] - zz;
                const float gw = -gzv;
                const float isg = __builtin_amdgcn_rcpf(sg);  // 1 ulp; three IEEE divisions cost ~30 VALU per element
                pw = gw * isg;
                gvar = -0.5f * pw * w * isg;
                a2 = (2.0f / n) * w * sg;
            }
            LGV[threadIdx.x + it * kStBlock] = gvar;
            if (abidx[it] >= 0) {
                LPW[slot_a * BN + abidx[it]] = pw;
                LA2[slot_a * BN + abidx[it]] = a2;
            }
        }
        if (pin < plast) load_plane(pin + 1);
        __syncthreads();
        // ---- B
        const int q = pin - S;
        const bool q_real = q >= 0 && q < vol.D && pin >= z0;
        const int slot_q1 = ((q % NS1) + NS1) % NS1, slot_q2 = ((q % NS2) + NS2) % NS2;
#pragma unroll
        for (int it = 0; it < NITB; ++it) {
            if (bwin[it] < 0) continue;
            float p = 0.0f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float row = 0.0f;
#pragma unroll
                for (int i = 0; i < NT; ++i) row = fmaf(bwx[it][i], LGV[bwin[it] + j * AX + i], row);
                p = fmaf(bwy[it][j], row, p);
            }
#pragma unroll
            for (int t = 0; t < NT - 1; ++t) ring1[it][t] = ring1[it][t + 1];
            ring1[it][NT - 1] = p;
            if (q_real) {
                const int e = threadIdx.x + it * kStBlock;
                const float t1 = adjoint_ring_sum<S>(ring1[it], q, vol.D);
                LG[slot_q2 * BN + e] = LPW[slot_q1 * BN + e] + LA2[slot_q1 * BN + e] * t1;
            }
        }
        __syncthreads();
        // ---- C
        const int r = pin - 2 * S;
        const int slot_r2 = ((r % NS2) + NS2) % NS2;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            if (!oval[o]) continue;
            const int yy = ly + o * (LMY / 2);
            float p = 0.0f;
            if (q_real) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float row = 0.0f;
#pragma unroll
                    for (int i = 0; i < NT; ++i) row = fmaf(cwx[i], LG[slot_q2 * BN + (yy + j) * BX + lx + i], row);
                    p = fmaf(cwy[o][j], row, p);
                }
            }
#pragma unroll
            for (int t = 0; t < NT - 1; ++t) ring2[o][t] = ring2[o][t + 1];
            ring2[o][NT - 1] = p;
            if (r >= z0 && r < z1) {
                const float t2 = adjoint_ring_sum<S>(ring2[o], r, vol.D);
                const float ga = LG[slot_r2 * BN + (yy + S) * BX + lx + S];
                g_m[(int64_t)r * HW + (unsigned)((oy + yy) * vol.W + gxo)] = fmaf(-(1.0f / n), t2, ga);
            }
        }
    }
    if (!EXPLICIT_GZ) {
        nll = wave_sum(nll);
        if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = nll;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < kStBlock / kWave; ++w) t += red[w];
            nll_out[(blk.z * gridDim.y + blk.y) * gridDim.x + blk.x] = t;  // slot of the logical tile: same partial layout as unswizzled
        }
    }
}

int lcc_data_bwd_march_blocks(Vol vol, int seg_C) {
    const int seg_len = lcc_seg_len(vol, seg_C, true);
    return ((vol.W + LMX - 1) / LMX) * ((vol.H + LMY - 1) / LMY) * ((vol.nz + seg_len - 1) / seg_len);
}

void launch_lcc_data_bwd_march(const float* fhat, const float* z, const float* sigma_m, const uint8_t* mask,
                               const float* g_z_override, const void* dev_state, int chain, float* g_warped,
                               double* nll_partials, int s, Vol vol, hipStream_t st, int batch, int64_t f_stride,
                               int64_t m_stride, int seg_C) {
    // the segment length is the one the context sized its partial sums for (lcc_data_bwd_march_blocks), whether the chains come in
    // one launch or one by one: the sums keep their slots and their order
    const int seg_len = lcc_seg_len(vol, seg_C, true);
    const int nseg = (vol.nz + seg_len - 1) / seg_len;
    if (g_z_override) batch = 0;
    const dim3 grid((vol.W + LMX - 1) / LMX, (vol.H + LMY - 1) / LMY, (unsigned)(nseg * (batch > 0 ? batch : 1)));
    const DevState* state = (const DevState*)dev_state;
    if (global_knobs().launch_log) {
        static int cache_l = 0;
        log_launch(batch > 0 ? "lcc_data_bwd_march_kernel (all chains)" : "lcc_data_bwd_march_kernel (per chain)", LMX, LMY,
                   (int64_t)grid.x * grid.y * grid.z, kStBlock, seg_len, 4 * s, vol.nz, batch > 0 ? batch : 1,
                   resident_blocks((const void*)lcc_data_bwd_march_kernel<1, false, false>, kStBlock, &cache_l));
    }
#define IRS_LCC_BWD(SS)                                                                                                      \
    if (g_z_override) hipLaunchKernelGGL((lcc_data_bwd_march_kernel<SS, true, false>), grid, dim3(kStBlock), 0, st, fhat, z,       \
                                         sigma_m, mask, g_z_override, state, chain, g_warped, nll_partials, vol, seg_len,   \
                                         nseg, batch, f_stride, m_stride);                                                  \
    else if (batch > 0) hipLaunchKernelGGL((lcc_data_bwd_march_kernel<SS, false, true>), grid, dim3(kStBlock), 0, st, fhat, z, sigma_m, mask, \
                                           g_z_override, state, chain, g_warped, nll_partials, vol, seg_len, nseg, batch, f_stride, m_stride); \
    else hipLaunchKernelGGL((lcc_data_bwd_march_kernel<SS, false, false>), grid, dim3(kStBlock), 0, st, fhat, z, sigma_m, mask,    \
                            g_z_override, state, chain, g_warped, nll_partials, vol, seg_len, nseg, batch, f_stride, m_stride);
    if (s == 1) { IRS_LCC_BWD(1) } else { IRS_LCC_BWD(2) }
#undef IRS_LCC_BWD
}

// ------------------------------------------------------------------------------------------------
// Mixture statistics for virtual decimation and the GMM step (utils/util.py:330-347,446-485; trainer.py:68-77):
// per masked voxel the VD-rescaled residual x and the responsibilities; sums of 1, x^2, the 2K GMM-gradient partials and the
// lag-1 products x[i] x[i+1] along the three axes.  z-marching over 64 x 4 column tiles: the mixture is evaluated ONCE per
// voxel (+ a one-voxel halo on the high x / y side and one run-out plane for the z pairs: 1.3 evaluations per voxel; the
// pointwise version evaluated it for every neighbour again, up to 4 per voxel), the neighbours come from an LDS plane
// (x, y) and a register (z).  partials: [kStatVals][gridDim.x], reduced by the scalar kernels in fixed order.
// ------------------------------------------------------------------------------------------------
#ifndef IRS_STATS_TY
#define IRS_STATS_TY 8
#endif
// 64 x 8 tile, two rows per thread: the tile + halo (65 x 9) takes 3 passes of the 256 threads for 2 rows of outputs; with a
// 64 x 4 tile it was 2 passes per row, the second one (halo column and row) on 69 lanes only.  Measured at 256^3: 64 x 8 with
// 8-plane segments beats 64 x 4 and 64 x 16 (data stage 0.664 / 0.683 / 0.707 ms)
constexpr int QTX = 64, QTY = 4, QPX = QTX + 1, QPN = QPX * (QTY + 1);  // reg_energy_march_kernel: one output per thread
constexpr int TTY = IRS_STATS_TY, TPN = QPX * (TTY + 1), QROWS = kStBlock / QTX, QNOUT = TTY / QROWS;

// KMAX >= K (the host knows K): accumulators and mixture temporaries sized for it -- 13 fp64 sums instead of 21 for K <= 4
// (the partial layout stays [kStatVals]: slots of components >= KMAX are written as zeros)
template <int KMAX>
__global__ __launch_bounds__(kStBlock) void stats_march_kernel(const float* __restrict__ z, const uint8_t* __restrict__ mask,
                                                               const DevState* __restrict__ state, int want_vd,
                                                               double* __restrict__ partials, Vol vol, int seg_len, int nseg,
                                                               int ntx, int nty) {
    constexpr int NIT = (TPN + kStBlock - 1) / kStBlock;
    constexpr int NACC = 5 + 2 * KMAX;  // n, sum x^2, 3 lag-1 products, KMAX dNLL/dlog_std, KMAX responsibility sums
    __shared__ float X[TPN];
    __shared__ double smem[NACC * (kStBlock / kWave)];
    double acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.0;
    const int64_t HW = (int64_t)vol.H * vol.W;
    const int lx = threadIdx.x % QTX, ly = threadIdx.x / QTX;
    const int K = state->K;
    const bool gmm = state->mode == IRS_DATA_GMM_LCC;
    const int total = ntx * nty * nseg;
    for (int tile_ = blockIdx.x; tile_ < total; tile_ += gridDim.x) {
        const int tile = xcd_swizzle_runs(tile_, total, ntx * IRS_SWZ_STENCIL_ROWS);  // x-neighbouring tiles on one XCD
        const int ox = (tile % ntx) * QTX, oy = ((tile / ntx) % nty) * TTY, seg = tile / (ntx * nty);
        const int z0 = vol.z0 + seg * seg_len, z1 = min(z0 + seg_len, vol.z0 + vol.nz);
        unsigned off[NIT];
        bool valid[NIT], owned[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + it * kStBlock;
            const int ex = i % QPX, ey = i / QPX;
            const int gx = ox + ex, gy = oy + ey;
            valid[it] = i < TPN && gx < vol.W && gy < vol.H;
            owned[it] = valid[it] && ex < QTX && ey < TTY;
            if (!want_vd) valid[it] = owned[it];  // no neighbour products: the halo is not needed
            off[it] = valid[it] ? (unsigned)(gy * vol.W + gx) : 0u;
        }
        const int zend = want_vd && z1 < vol.D ? z1 + 1 : z1;  // one run-out plane closes the z pairs of the segment
        float pz[NIT];
        uint8_t pm[NIT];
        auto load_plane = [&](int p) {
            const int64_t zo = (int64_t)p * HW;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (valid[it]) {
                    pm[it] = (mask + zo)[off[it]];
                    pz[it] = (z + zo)[off[it]];
                }
        };
        load_plane(z0);
        float xprev[QNOUT];
#pragma unroll
        for (int o = 0; o < QNOUT; ++o) xprev[o] = 0.0f;
        for (int zc = z0; zc < zend; ++zc) {
            const bool inseg = zc < z1;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = threadIdx.x + it * kStBlock;
                float xv = 0.0f;
                if (valid[it] && pm[it]) {
                    if (owned[it] && inseg) {
                        float resp[KMAX], q[KMAX];
                        const MixEval e = mix_eval<true, KMAX>(pz[it], state, resp, q);
                        xv = e.x;
                        acc[0] += 1.0;
                        acc[1] += (double)(e.x * e.x);
                        if (gmm) {
#pragma unroll
                            for (int k = 0; k < KMAX; ++k)
                                if (k < K) {
                                    acc[5 + k] += (double)(resp[k] * (1.0f - q[k]));
                                    acc[5 + KMAX + k] += (double)resp[k];
                                }
                        }
                    } else {
                        xv = mix_eval<false, KMAX>(pz[it], state, nullptr, nullptr).x;
                    }
                }
                if (i < TPN) X[i] = xv;
            }
            if (zc + 1 < zend) load_plane(zc + 1);
            __syncthreads();
            if (want_vd) {
                // lag-1 neighbours along D (reference "cov_x", dim 2), H (dim 3), W (dim 4); x = 0 off the mask / volume
#pragma unroll
                for (int o = 0; o < QNOUT; ++o) {
                    const int yy = ly + o * QROWS;
                    const float own = X[yy * QPX + lx];
                    if (inseg) {
                        acc[3] += (double)(own * X[(yy + 1) * QPX + lx]);
                        acc[4] += (double)(own * X[yy * QPX + lx + 1]);
                    }
                    if (zc > z0) acc[2] += (double)(xprev[o] * own);
                    xprev[o] = own;
                }
            }
            __syncthreads();
        }
    }
    block_sum<NACC>(acc, smem);
    if (threadIdx.x == 0) {
        // [kStatVals][gridDim.x]: the scalar kernel that sums them reads a column with consecutive lanes
        double* __restrict__ p = partials + blockIdx.x;
        const int64_t cs = gridDim.x;
#pragma unroll
        for (int j = 0; j < 5; ++j) p[j * cs] = acc[j];
#pragma unroll
        for (int k = 0; k < IRS_MAX_COMPONENTS; ++k) {
            p[(5 + k) * cs] = k < KMAX ? acc[5 + (k < KMAX ? k : 0)] : 0.0;
            p[(5 + IRS_MAX_COMPONENTS + k) * cs] = k < KMAX ? acc[5 + KMAX + (k < KMAX ? k : 0)] : 0.0;
        }
    }
}

void launch_stats_march(int want_vd, const float* z, const uint8_t* mask, const void* dev_state, double* partials, int blocks,
                        Vol vol, int K, hipStream_t st) {
    const int seg_env = global_knobs().stats_seg;
    const int seg_len = pick_seg_len(vol.nz, (int64_t)((vol.W + QTX - 1) / QTX) * ((vol.H + TTY - 1) / TTY), 4, seg_env, 2048);
    const int nseg = (vol.nz + seg_len - 1) / seg_len;
    const int ntx = (vol.W + QTX - 1) / QTX, nty = (vol.H + TTY - 1) / TTY;
    if (global_knobs().launch_log) {
        static int cache_l = 0;
        const int64_t tiles = (int64_t)ntx * nty * nseg;
        log_launch("stats_march_kernel<4> (tiles walked by a capped grid)", QTX, TTY, tiles < blocks ? tiles : blocks, kStBlock, seg_len, 1, vol.nz, 1,
                   resident_blocks((const void*)stats_march_kernel<4>, kStBlock, &cache_l));
    }
    if (K >= 1 && K <= 4)
        hipLaunchKernelGGL(stats_march_kernel<4>, dim3(blocks), dim3(kStBlock), 0, st, z, mask, (const DevState*)dev_state, want_vd,
                           partials, vol, seg_len, nseg, ntx, nty);
    else
        hipLaunchKernelGGL(stats_march_kernel<IRS_MAX_COMPONENTS>, dim3(blocks), dim3(kStBlock), 0, st, z, mask,
                           (const DevState*)dev_state, want_vd, partials, vol, seg_len, nseg, ntx, nty);
}

// ------------------------------------------------------------------------------------------------
// Regulariser energy y_c = sum over the 9 forward differences of v_s, the replicated last difference counted twice
// (model/loss.py:152-161, utils/diff_op.py:62-96).  z-marching over 64 x 4 column tiles: every value is read once
// (+ a one-voxel halo on the high x / y side), the x / y neighbours come from an LDS plane, the z neighbour from a register.
// partials: [C][gridDim.x]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kStBlock) void reg_energy_march_kernel(const float* __restrict__ v, double* __restrict__ partials,
                                                                    Vol vol, int seg_len, int nseg, int ntx, int nty) {
    constexpr int NIT = (QPN + kStBlock - 1) / kStBlock;
    __shared__ float F[3 * QPN];
    __shared__ double smem[kStBlock / kWave];
    const float* __restrict__ f = v + (int64_t)blockIdx.y * 3 * vol.V;
    const int64_t HW = (int64_t)vol.H * vol.W;
    const int lx = threadIdx.x % QTX, ly = threadIdx.x / QTX;
    double acc[1] = {0.0};
    const int total = ntx * nty * nseg;
    for (int tile_ = blockIdx.x; tile_ < total; tile_ += gridDim.x) {
        const int tile = xcd_swizzle_runs(tile_, total, ntx * IRS_SWZ_STENCIL_ROWS);
        const int ox = (tile % ntx) * QTX, oy = ((tile / ntx) % nty) * QTY, seg = tile / (ntx * nty);
        const int z0 = vol.z0 + seg * seg_len, z1 = min(z0 + seg_len, vol.z0 + vol.nz);
        const int x = ox + lx, y = oy + ly;
        const bool col_in = x < vol.W && y < vol.H;
        const float wx = x + 1 < vol.W ? (x + 2 == vol.W ? 2.0f : 1.0f) : 0.0f;
        const float wy = y + 1 < vol.H ? (y + 2 == vol.H ? 2.0f : 1.0f) : 0.0f;
        unsigned off[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + it * kStBlock;
            const int ex = i % QPX, ey = i / QPX;
            off[it] = (unsigned)(min(oy + ey, vol.H - 1) * vol.W + min(ox + ex, vol.W - 1)) * 4u;
        }
        float pre[NIT][3];
        auto load_plane = [&](int p) {
            const float* __restrict__ base = f + (int64_t)p * HW;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (threadIdx.x + it * kStBlock < QPN) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) pre[it][c] = ldg_off(base + c * vol.V, off[it]);
                }
        };
        const int zend = z1 < vol.D ? z1 + 1 : z1;  // one run-out plane closes the z differences of the segment
        load_plane(z0);
        float prev[3] = {0.0f, 0.0f, 0.0f};
        for (int zc = z0; zc < zend; ++zc) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = threadIdx.x + it * kStBlock;
                if (i < QPN) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) F[c * QPN + i] = pre[it][c];
                }
            }
            if (zc + 1 < zend) load_plane(zc + 1);
            __syncthreads();
            if (col_in) {
                const bool inseg = zc < z1;
                const float wz = zc > z0 ? (zc + 1 == vol.D ? 2.0f : 1.0f) : 0.0f;  // weight of the difference (zc - 1, zc)
                float e = 0.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v0 = F[c * QPN + ly * QPX + lx];
                    if (inseg) {
                        const float dx = F[c * QPN + ly * QPX + lx + 1] - v0, dy = F[c * QPN + (ly + 1) * QPX + lx] - v0;
                        e += wx * dx * dx;
                        e += wy * dy * dy;
                    }
                    const float dz = v0 - prev[c];
                    e += wz * dz * dz;
                    prev[c] = v0;
                }
                acc[0] += (double)e;
            }
            __syncthreads();
        }
    }
    block_sum<1>(acc, smem);
    if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = acc[0];
}

void launch_reg_energy_march(const float* v, double* partials, int blocks, int C, Vol vol, hipStream_t st) {
    const int seg_len = pick_seg_len(vol.nz, (int64_t)((vol.W + QTX - 1) / QTX) * ((vol.H + QTY - 1) / QTY), 8, 0);
    const int nseg = (vol.nz + seg_len - 1) / seg_len;
    const int ntx = (vol.W + QTX - 1) / QTX, nty = (vol.H + QTY - 1) / QTY;
    hipLaunchKernelGGL(reg_energy_march_kernel, dim3(blocks, C), dim3(kStBlock), 0, st, v, partials, vol, seg_len, nseg, ntx, nty);
}

// ------------------------------------------------------------------------------------------------
// SGLD / SGD update (trainer.py:341-356; utils/functions.py:83-84):
//   grad_v = sigma^2 (g_data * s_c + 2 coef D^T D v_s),   v <- v - lr grad_v
// D^T D is the adjoint of the forward-difference stencil with the replicated (double-weight) last difference.  The
// Sobolev backward is the identity, so nothing else sits between v_s and v.  z-marching over 64 x 4 column tiles: v_s
// goes through an LDS ring of planes (one-voxel halo), the other streams are read and written once, coalesced.
// ------------------------------------------------------------------------------------------------
constexpr int UPX = QTX + 2, UPY = QTY + 2, UPN = UPX * UPY, UNS = 4;
#ifndef IRS_SWZ_UPDATE_ROWS
#define IRS_SWZ_UPDATE_ROWS 4
#endif

// `energy_partials` (optional, [C][tiles per chain]): the regulariser energy sum (forward difference)^2 of v_s as a by-product -- the
// stencil above is built from exactly those differences and weights.  Used for the regularisers whose coefficient does not depend
// on the energy (RegLoss_L2, RegLoss_LogNormal_L2: coef = w / 2, `coef_from_w`), where the energy is only needed AFTER the update
// (loss term, Adam step on log w): one kernel and one pass over v_s less per transition.
template <bool SIGMA>  // sigma field present (compile-time: a run-time branch around its loads leaves waits in the own-voxel prefetch)
__global__ __launch_bounds__(kStBlock) void sgld_update_march_kernel(float* __restrict__ v, const float* __restrict__ sigma,
                                                                     const float* __restrict__ g, const float* __restrict__ v_s,
                                                                     const DevState* __restrict__ state, float lr, float s0,
                                                                     float s1, float s2, float* __restrict__ grad_out, Vol vol,
                                                                     int seg_len, int nseg, int ntx, int nty,
                                                                     double* __restrict__ energy_partials, int coef_from_w) {
    constexpr int NIT = (UPN + kStBlock - 1) / kStBlock;
    __shared__ float F[UNS * 3 * UPN];
    __shared__ double esm[kStBlock / kWave];
    double eacc[1] = {0.0};
    // x-neighbouring tiles AND four y-neighbouring rows of tiles on one XCD: this kernel's 64 x 4 tile reads two halo rows per four
    // rows of v_s, and it is the one memory-bound kernel of a transition (0.57 of the roofline, 1.36 x its algorithmic bytes in
    // round 4); with the halo rows of three of four tile rows served by the XCD's L2: 172 against 186 us at 256^3, three alternating
    // runs on one box (profiles/r05_swz_ab.txt).  The same grouping for the other stencil kernels measured flat: theirs stays 1.
    const int tile = xcd_swizzle_runs((int)blockIdx.x, (int)gridDim.x, ntx * IRS_SWZ_UPDATE_ROWS);
    const int chain = tile / (ntx * nty * nseg);
    const int t_ = tile % (ntx * nty * nseg);
    const int ox = (t_ % ntx) * QTX, oy = ((t_ / ntx) % nty) * QTY, seg = t_ / (ntx * nty);
    const int z0 = vol.z0 + seg * seg_len, z1 = min(z0 + seg_len, vol.z0 + vol.nz);
    const int64_t HW = (int64_t)vol.H * vol.W;
    const int64_t cb = (int64_t)chain * 3 * vol.V;
    const float* __restrict__ f = v_s + cb;
    const int lx = threadIdx.x % QTX, ly = threadIdx.x / QTX;
    const int x = ox + lx, y = oy + ly;
    const bool col_in = x < vol.W && y < vol.H;
    const float coef2 = coef_from_w ? (float)exp(state->st.reg_param[0]) : 2.0f * (float)state->coef[chain];  // 2 coef; L2 family: coef = w / 2
    const bool frozen = state->bad_now != 0u || comm_bad(state);  // the transition turned out to be a no-op (scalar_kernels.h: Verdict, comm_bad): v stays
    const float sc[3] = {s0, s1, s2};
    // weights of the two difference terms touching a position: w(q) = 2 for the last (replicated) difference q = n - 2
    auto wm = [](int pos, int n) { return pos >= 1 ? (pos - 1 == n - 2 ? 2.0f : 1.0f) : 0.0f; };
    auto wp = [](int pos, int n) { return pos <= n - 2 ? (pos == n - 2 ? 2.0f : 1.0f) : 0.0f; };
    const float wxm = wm(x, vol.W), wxp = wp(x, vol.W), wym = wm(y, vol.H), wyp = wp(y, vol.H);

    unsigned off[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * kStBlock;
        const int ex = i % UPX, ey = i / UPX;
        off[it] = (unsigned)(min(max(oy - 1 + ey, 0), vol.H - 1) * vol.W + min(max(ox - 1 + ex, 0), vol.W - 1)) * 4u;
    }
    float pre[NIT][3];
    auto load_plane = [&](int p) {
        const float* __restrict__ base = f + (int64_t)min(max(p, 0), vol.D - 1) * HW;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (threadIdx.x + it * kStBlock < UPN) {
#pragma unroll
                for (int c = 0; c < 3; ++c) pre[it][c] = ldg_off(base + c * vol.V, off[it]);
            }
    };
    auto commit = [&](int p) {
        const int slot = ((p % UNS) + UNS) % UNS;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + it * kStBlock;
            if (i < UPN) {
#pragma unroll
                for (int c = 0; c < 3; ++c) F[(slot * 3 + c) * UPN + i] = pre[it][c];
            }
        }
    };
    load_plane(z0 - 1);
    commit(z0 - 1);
    load_plane(z0);
    commit(z0);
    const unsigned own = col_in ? (unsigned)(y * vol.W + x) * 4u : 0u;
    // The streams read once per voxel (incoming gradient, v itself, sigma) are requested ONE PLANE AHEAD and BEFORE the stencil
    // prefetch of that step: the wavefront's memory counter completes in order, so a load issued in the compute phase -- after the
    // prefetch -- can only be waited for together with the whole prefetch, three times per plane (once per channel: the store
    // to v orders the next channel's load behind it).  That made this kernel latency-bound at 2.5x its HBM time.
    struct OwnIn {
        float g[3], v[3], s[3];
    };
    auto load_own = [&](int z, OwnIn& q) {
        const int64_t pl = (int64_t)z * HW;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t base = cb + c * vol.V + pl;
            q.g[c] = ldg_off(g + base, own);
            q.v[c] = ldg_off(v + base, own);
            if (SIGMA) q.s[c] = ldg_off(sigma + base, own);
        }
    };
    // one plane: `cur` holds the plane's own-voxel streams (requested a step ago), `nxt` receives the next plane's
    auto step = [&](int zc, const OwnIn& cur, OwnIn& nxt) {
        // commit needs the newest loads anyway; said unconditionally (the commit's own waits sit behind per-lane guards) it also
        // tells the compiler that `cur` has arrived -- otherwise it re-waits for it in the compute phase, with a count that covers
        // the loads issued below
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        commit(zc + 1);
        if (zc + 1 < z1) {
            load_own(zc + 1, nxt);
            load_plane(zc + 2);
        }
        __syncthreads();  // ring of 4: plane zc + 1 landed in the slot of plane zc - 3, which nobody reads any more
        if (col_in) {
            const int sm = (((zc - 1) % UNS) + UNS) % UNS, s0_ = ((zc % UNS) + UNS) % UNS, sp = (((zc + 1) % UNS) + UNS) % UNS;
            const float wzm = wm(zc, vol.D), wzp = wp(zc, vol.D);
            const int ci = (ly + 1) * UPX + lx + 1;
            const int64_t pl = (int64_t)zc * HW;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* __restrict__ P = F + (s0_ * 3 + c) * UPN;
                const float v0 = P[ci];
                const float dxp = P[ci + 1] - v0, dyp = P[ci + UPX] - v0, dzp = F[(sp * 3 + c) * UPN + ci] - v0;  // forward differences
                float rx = wxm * (v0 - P[ci - 1]);
                rx -= wxp * dxp;
                float ry = wym * (v0 - P[ci - UPX]);
                ry -= wyp * dyp;
                float rz = wzm * (v0 - F[(sm * 3 + c) * UPN + ci]);
                rz -= wzp * dzp;
                if (energy_partials) eacc[0] += (double)(wxp * dxp * dxp + wyp * dyp * dyp + wzp * dzp * dzp);
                const float lap = rx + ry + rz;
                const int64_t base = cb + c * vol.V + pl;
                const float gr = cur.g[c] * sc[c] + coef2 * lap;
                const float sg = SIGMA ? cur.s[c] : 1.0f;
                const float gs = sg * sg * gr;  // SGLD.backward: sigma^2 * grad == v.grad in the reference
                float* __restrict__ vp = reinterpret_cast<float*>(reinterpret_cast<char*>(v + base) + own);
                if (grad_out) *reinterpret_cast<float*>(reinterpret_cast<char*>(grad_out + base) + own) = gs;
                if (!frozen) *vp = cur.v[c] - lr * gs;
            }
        }
    };
    OwnIn qa = {}, qb = {};  // two buffers used alternately (no copies: a copy would have to wait for the loads it copies)
    load_own(z0, qa);
    load_plane(z0 + 1);
    for (int zc = z0; zc < z1; zc += 2) {
        step(zc, qa, qb);
        if (zc + 1 < z1) step(zc + 1, qb, qa);
    }
    if (energy_partials) {  // (uniform branch: every thread of the block arrives)
        block_sum<1>(eacc, esm);
        if (threadIdx.x == 0) energy_partials[tile] = eacc[0];  // tile = chain * tiles_per_chain + tile in chain: [C][tiles]
    }
}

static int update_seg_len(Vol vol, int C) {
    const int seg_env = global_knobs().update_seg;
    const int64_t per_layer = (int64_t)((vol.W + QTX - 1) / QTX) * ((vol.H + QTY - 1) / QTY) * C;
    int len = pick_seg_len(vol.nz, per_layer, 4, seg_env);
    if (global_knobs().seg_fit && seg_env <= 0) {  // resident-set cost model (common.h), run-in 2 planes: 224^3 0.133 -> 0.125 ms
        static int cache = 0;
        const int64_t res = resident_blocks((const void*)sgld_update_march_kernel<false>, kStBlock, &cache);
        if (res > 0) len = pick_seg_len_fit(vol.nz, 0, per_layer, 4, 2, res, 0);
    }
    return len;
}

int sgld_update_blocks_per_chain(Vol vol, int C) {
    const int seg_len = update_seg_len(vol, C);
    return ((vol.W + QTX - 1) / QTX) * ((vol.H + QTY - 1) / QTY) * ((vol.nz + seg_len - 1) / seg_len);
}

void launch_sgld_update_march(float* v, const float* sigma, const float* g_d0, const float* v_s, const void* dev_state,
                              float lr, float s0, float s1, float s2, float* grad_out, int C, Vol vol, hipStream_t st,
                              double* energy_partials, bool coef_from_w) {
    const int seg_len = update_seg_len(vol, C);
    const int nseg = (vol.nz + seg_len - 1) / seg_len;
    const int ntx = (vol.W + QTX - 1) / QTX, nty = (vol.H + QTY - 1) / QTY;
    if (global_knobs().launch_log) {
        static int cache_l = 0;
        log_launch("sgld_update_march_kernel", QTX, QTY, (int64_t)ntx * nty * nseg * C, kStBlock, seg_len, 2, vol.nz, C,
                   resident_blocks((const void*)sgld_update_march_kernel<false>, kStBlock, &cache_l));
    }
    if (sigma)
        hipLaunchKernelGGL(sgld_update_march_kernel<true>, dim3((unsigned)(ntx * nty * nseg * C)), dim3(kStBlock), 0, st, v, sigma, g_d0,
                           v_s, (const DevState*)dev_state, lr, s0, s1, s2, grad_out, vol, seg_len, nseg, ntx, nty, energy_partials,
                           coef_from_w ? 1 : 0);
    else
        hipLaunchKernelGGL(sgld_update_march_kernel<false>, dim3((unsigned)(ntx * nty * nseg * C)), dim3(kStBlock), 0, st, v, sigma, g_d0,
                           v_s, (const DevState*)dev_state, lr, s0, s1, s2, grad_out, vol, seg_len, nseg, ntx, nty, energy_partials,
                           coef_from_w ? 1 : 0);
}

}  // namespace irs
