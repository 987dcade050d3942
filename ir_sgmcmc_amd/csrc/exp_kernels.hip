// Scaling-and-squaring step and its adjoint for gfx950 -- the dominant kernels of a transition (720 of its 878
// algorithmic bytes per voxel).  Reference semantics: utils/transformation.py:70-73 and autograd through it.
//
// Both directions use a z-MARCHING schedule: a 256-thread workgroup owns a column tile of output voxels (adjoint 32x8,
// forward 64x8 with two rows per thread) over a segment of up to 32 planes, keeps a ring of the last source planes in LDS,
// has the next plane's global loads in flight while it works on the current one, and is placed so that x-adjacent tiles
// (which share halo cache lines) sit on one XCD.
//
//   forward   d_out = d + sample(d, id + d): taps from the ring (global memory only if a tap leaves it: any displacement
//             is handled, the ring radius R in {1, 2} is a speed knob picked on the device from the published bound
//             max|d_k|); publishes max|d_{k+1}| per chain with one guarded atomicMax per workgroup.
//   adjoint   g = G + grid-gradient + trilinear scatter of G, computed as an owner-computes GATHER: every output voxel
//             sums G(x) * prod_a hat(p_a(x) - y_a) over its (2R+1)^3 candidate sources.  No atomics (LDS float atomics
//             retire one lane per clock on gfx950, global ones cost 7.5 ms per step at 256^3), fixed summation order,
//             plain coalesced stores, no zero-fill.  For max|d_k| >= 2 voxels the LDS-atomic scatter kernel takes over
//             (correct for any displacement).  Which kernel does the work of a step is decided on the device.
#include "kernels.h"

namespace irs {

#ifndef IRS_ETX
#define IRS_ETX 32
#define IRS_ETY 8
#define IRS_ETZ 8
#endif
constexpr int ETX = IRS_ETX, ETY = IRS_ETY, ETZ = IRS_ETZ, ETN = ETX * ETY * ETZ;
constexpr int kExpBlock = 512;

struct Scale3L {
    float nm1[3];
    float rnm1[3];  // correctly rounded 1 / nm1 (common.h: exact_rcp, div_exact)
    float inv_pow;
};

typedef float F3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ F3 ld3g(const float* __restrict__ p) {  // three consecutive floats, one global_load_dwordx3
    F3 v;
    __builtin_memcpy(&v, p, 12);
    return v;
}

template <bool PRESCALE>
__device__ __forceinline__ float ldp(const float* __restrict__ p, int64_t i, float nm1, float rnm1, float inv_pow) {
    const float v = p[i];
    return PRESCALE ? prescale(v, nm1, rnm1, inv_pow) : v;
}

// Layout of the squaring-step working fields: planar ([3][V], the reference's tensors) or interleaved ([V][3]).  The fused
// transition keeps its INTERNAL fields (d_1 .. d_{n-1}, the gradients between adjoint steps) interleaved: a halo row of an
// interleaved field is one contiguous 12 (W + 2) byte segment, and the staging rate of a marching kernel grows with the
// length of the segments it reads (tools/bw_probe.hip: 3.9 -> 4.9 TB/s for the 32x8 skeleton).  Element (c, i) of a chain's
// field lives at base[c * cs + i * em]; `lay` bits: 1 displacement input, 2 gradient input, 4 output.
struct Lay3 {
    int64_t cs;
    int em;
};
__host__ __device__ __forceinline__ Lay3 lay3(int aos, int64_t V) { return aos ? Lay3{1, 3} : Lay3{V, 1}; }
template <int L>
struct LayC {  // a layout bit set as a type: the marching tiles instantiate their body per layout
    static constexpr int value = L;
};

__device__ __forceinline__ void atomic_max_nonneg(unsigned* addr, float v) {
    // non-negative floats order like their bit patterns
    atomicMax(addr, __float_as_uint(v));
}

// The squaring-step kernels walk the tiles of all chains with a grid-stride loop over a capped grid: of the variants
// launched per step only one does work (chosen on the device), and an idle variant should cost a few thousand empty
// workgroups, not one per tile.
struct TileGrid {
    int ntx, nty, ntz, total;  // tiles per axis, total over all chains
};

constexpr int kExpGridCap = 1024;  // also bounds the cost of a variant that is launched but not selected

template <int H>
struct ExpBox {
    static constexpr int SX = ETX + 2 * H, SY = ETY + 2 * H, SZ = ETZ + 2 * H, SN = SX * SY * SZ;
};

// stage d (3 channels, optionally prescaled) over tile +/- H into LDS; coordinates are clamped to the volume
template <bool PRESCALE, int H>
__device__ __forceinline__ void stage_field(const float* __restrict__ c0, const Lay3 LD, float* __restrict__ lds, int ox, int oy,
                                            int oz, const Vol vol, const Scale3L sc) {
    using B = ExpBox<H>;
    for (int i = threadIdx.x; i < B::SN; i += kExpBlock) {
        const int lx = i % B::SX, ly = (i / B::SX) % B::SY, lz = i / (B::SX * B::SY);
        const int gx = min(max(ox - H + lx, 0), vol.W - 1), gy = min(max(oy - H + ly, 0), vol.H - 1),
                  gz = min(max(oz - H + lz, 0), vol.D - 1);
        const int64_t g = ((int64_t)gz * vol.H + gy) * vol.W + gx;
        lds[i] = ldp<PRESCALE>(c0, g * LD.em, sc.nm1[0], sc.rnm1[0], sc.inv_pow);
        lds[B::SN + i] = ldp<PRESCALE>(c0 + LD.cs, g * LD.em, sc.nm1[1], sc.rnm1[1], sc.inv_pow);
        lds[2 * B::SN + i] = ldp<PRESCALE>(c0 + 2 * LD.cs, g * LD.em, sc.nm1[2], sc.rnm1[2], sc.inv_pow);
    }
}

static Scale3L make_scale_l(Vol vol, int no_steps) {
    Scale3L s;
    s.nm1[0] = (float)(vol.W - 1);
    s.nm1[1] = (float)(vol.H - 1);
    s.nm1[2] = (float)(vol.D - 1);
    for (int c = 0; c < 3; ++c) s.rnm1[c] = exact_rcp(s.nm1[c]);
    s.inv_pow = 1.0f / (float)(1 << no_steps);
    return s;
}

static dim3 exp_grid(Vol vol, int C, TileGrid* tg) {
    tg->ntx = (vol.W + ETX - 1) / ETX;
    tg->nty = (vol.H + ETY - 1) / ETY;
    tg->ntz = (vol.nz + ETZ - 1) / ETZ;
    tg->total = tg->ntx * tg->nty * tg->ntz * C;
    return dim3((unsigned)(tg->total < kExpGridCap ? tg->total : kExpGridCap));
}

// ------------------------------------------------------------------------------------------------
// adjoint step, any displacement (owner computes): a workgroup owns a 32 x 8 x 8 tile of outputs, walks every source voxel
// that can reach it and scatters the corner contributions into LDS accumulators.
//
// Accumulation is FIXED POINT: contributions w * G are scaled by a per-tile power of two, rounded to integers and added
// with 64-bit LDS atomics.  On gfx950 ds_add_u64 retires ~10 lanes per clock and CU, ds_add_f32 0.4
// (tools/lds_atomic_probe.hip); with float atomics 60 % of this kernel was the atomic unit.  The scale comes from the
// largest |G| a source of the tile can carry (coarse grid below), so one contribution uses < 2^30 and 2^33 of them fit;
// the rounding step is max|G| * 2^-30 per contribution, and integer sums do not depend on the order of the adds: the kernel
// is deterministic, which the float version was not.
//
// Which sources: a source s reaches the tile iff clip(s + d(s)) lies in [lo - 1, hi + 1) per axis, i.e.
// s in [lo - 1 - max d, hi + 1 - min d] with the extrema over any superset of the sources.  B0 = tile +- (global bound + 1);
// two rounds over the coarse cells (8^3 voxels: min / max of d per axis, max |G|) covering the current box shrink it to
// (tile - local displacement range) +- 1: for a smooth 12-voxel field 3x fewer sources than B0.
// ------------------------------------------------------------------------------------------------
constexpr int kCell = 8, kCmm = 7;  // per cell: min, max of d_x, d_y, d_z (voxels), max |G|

template <bool PRESCALE>
__global__ __launch_bounds__(kWave) void coarse_minmax_kernel(const float* __restrict__ dk, const float* __restrict__ G,
                                                              const float* __restrict__ gscale, float* __restrict__ cmm,
                                                              Vol vol, Scale3L sc, const unsigned* __restrict__ dmax,
                                                              int gather_radius, int ncx, int ncy, int ncz, int lay) {
    const int chain = blockIdx.y;
    {   // leave at once unless this chain needs the any-radius kernel (same test as exp_bwd_lds_kernel)
        const int h = (int)floorf(fmaxf(fmaxf(__uint_as_float(dmax[chain * 4 + 0]), __uint_as_float(dmax[chain * 4 + 1])), __uint_as_float(dmax[chain * 4 + 2]))) + 1;
        if (h <= gather_radius) return;
    }
    const Lay3 LD = lay3(lay & 1, vol.V), LG = lay3(lay & 2, vol.V);
    // one wavefront per run of 8 cells along x: lane = x offset, 64 (y, z) iterations of fully coalesced row loads, then a
    // reduction over each group of 8 lanes
    const int nrx = (ncx + 7) / 8;
    const int run = blockIdx.x % nrx, cy = (blockIdx.x / nrx) % ncy, cz = blockIdx.x / (nrx * ncy);
    const int x = run * 64 + (int)threadIdx.x;
    const float* c0 = dk + (int64_t)chain * 3 * vol.V;
    const float* Gc = G + (int64_t)chain * 3 * vol.V;
    const float* gs_ = gscale ? gscale + (int64_t)chain * vol.V : nullptr;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, gm = 0.0f;
    if (x < vol.W) {
        for (int z = cz * kCell; z < min(cz * kCell + kCell, vol.D); ++z)
#pragma unroll
            for (int yy = 0; yy < kCell; ++yy) {  // fixed trip count: eight rows of loads in flight (a clamped row repeats: harmless)
                const int y = min(cy * kCell + yy, vol.H - 1);
                const int64_t g = ((int64_t)z * vol.H + y) * vol.W + x;
                const float gsc = gs_ ? gs_[g] : 1.0f;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float v = ldp<PRESCALE>(c0 + a * LD.cs, g * LD.em, sc.nm1[a], sc.rnm1[a], sc.inv_pow) * (0.5f * sc.nm1[a]);
                    mn[a] = fminf(mn[a], v);
                    mx[a] = fmaxf(mx[a], v);
                    gm = fmaxf(gm, fabsf(Gc[a * LG.cs + g * LG.em] * gsc));  // the product the scatter forms
                }
            }
    }
#pragma unroll
    for (int off = 1; off < kCell; off <<= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, kWave));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, kWave));
        }
        gm = fmaxf(gm, __shfl_xor(gm, off, kWave));
    }
    const int cx = run * 8 + (int)(threadIdx.x >> 3);
    if ((threadIdx.x & 7) == 0 && cx < ncx) {
        float* o = cmm + ((int64_t)chain * ncx * ncy * ncz + ((int64_t)cz * ncy + cy) * ncx + cx) * kCmm;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            o[2 * a] = mn[a];
            o[2 * a + 1] = mx[a];
        }
        o[6] = gm;
    }
}

// Source box and fixed-point scale of every tile of the any-radius kernel, one wavefront per tile (round 5, late): the same two rounds
// over the coarse cells that a workgroup of exp_bwd_lds_kernel otherwise runs at the start of each of its tiles -- 256 threads, two
// block reductions, four barriers, two dependent global round trips: 5.8 of the 38 us of a tile (profiles/r05_lds_phase_trace.txt).
// Minima / maxima do not depend on the order they are taken in: the boxes and scales are the in-kernel ones, bit for bit.
// boxes[tile]: lo[3], hi[3] (int bits), max |G| of the box, unused.
constexpr int kBoxWords = 8;
// (A small persistent grid of four-wavefront workgroups, every wavefront striding over the tiles: launched for a step whose chains all
// stay with the gather kernels -- the host predicts, the device decides -- it costs the dispatch of 512 workgroups, not of one per tile.)
constexpr int kBoxBlock = 4 * kWave, kBoxGrid = 512;
__global__ __launch_bounds__(kBoxBlock) void tile_box_kernel(const float* __restrict__ cmm, float* __restrict__ boxes, Vol vol,
                                                             const unsigned* __restrict__ dmax, TileGrid tg, int gather_radius) {
  const int lane = (int)threadIdx.x & (kWave - 1);
  for (int tile = (int)blockIdx.x * (kBoxBlock / kWave) + (int)threadIdx.x / kWave; tile < tg.total; tile += (int)gridDim.x * (kBoxBlock / kWave)) {
    int t_ = tile;
    const int ox = (t_ % tg.ntx) * ETX;
    t_ /= tg.ntx;
    const int oy = (t_ % tg.nty) * ETY;
    t_ /= tg.nty;
    const int oz = vol.z0 + (t_ % tg.ntz) * ETZ;
    const int chain = t_ / tg.ntz;
    const int hx = (int)floorf(__uint_as_float(dmax[chain * 4 + 0])) + 1;
    const int hy = (int)floorf(__uint_as_float(dmax[chain * 4 + 1])) + 1;
    const int hz = (int)floorf(__uint_as_float(dmax[chain * 4 + 2])) + 1;
    if (hx <= gather_radius && hy <= gather_radius && hz <= gather_radius) continue;  // a gather kernel owns this chain
    const int zend = min(oz + ETZ, vol.z0 + vol.nz) - 1;
    int lo[3] = {max(ox - hx, 0), max(oy - hy, 0), max(oz - hz, 0)};
    int hi[3] = {min(ox + ETX - 1 + hx, vol.W - 1), min(oy + ETY - 1 + hy, vol.H - 1), min(zend + hz, vol.D - 1)};
    float gmax = 0.0f;
    const int nax[3] = {vol.W, vol.H, vol.D};
    const int tlo[3] = {ox, oy, oz};
    const int thi[3] = {min(ox + ETX, vol.W) - 1, min(oy + ETY, vol.H) - 1, min(zend, vol.D - 1)};
    const int ncx = (vol.W + kCell - 1) / kCell, ncy = (vol.H + kCell - 1) / kCell, ncz = (vol.D + kCell - 1) / kCell;
    const float* __restrict__ cm = cmm + (int64_t)chain * ncx * ncy * ncz * kCmm;
    for (int round = 0; round < 2; ++round) {
        const int c0x = lo[0] / kCell, c0y = lo[1] / kCell, c0z = lo[2] / kCell;
        const int nx_ = hi[0] / kCell - c0x + 1, ny_ = hi[1] / kCell - c0y + 1, nz_ = hi[2] / kCell - c0z + 1;
        float m7[kCmm] = {3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f, 0.0f};
        for (int i = lane; i < nx_ * ny_ * nz_; i += kWave) {
            const int cell = ((c0z + i / (nx_ * ny_)) * ncy + c0y + (i / nx_) % ny_) * ncx + c0x + i % nx_;
#pragma unroll
            for (int j = 0; j < kCmm; ++j) {
                const float o = cm[cell * kCmm + j];
                m7[j] = (j < 6 && !(j & 1)) ? fminf(m7[j], o) : fmaxf(m7[j], o);
            }
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1)
#pragma unroll
            for (int j = 0; j < kCmm; ++j) {
                const float o = __shfl_xor(m7[j], off, kWave);
                m7[j] = (j < 6 && !(j & 1)) ? fminf(m7[j], o) : fmaxf(m7[j], o);
            }
        gmax = m7[6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float mn = m7[2 * a], mx = m7[2 * a + 1];
            if (!(mn <= mx) || fabsf(mn) > 1.0e6f || fabsf(mx) > 1.0e6f) continue;  // (exp_bwd_lds_kernel: no usable extrema -> keep the box)
            if (tlo[a] - 1 > 0) lo[a] = max(lo[a], (int)floorf((float)(tlo[a] - 1) - mx - 1e-3f));
            if (thi[a] + 1 < nax[a] - 1) hi[a] = min(hi[a], (int)ceilf((float)(thi[a] + 1) - mn + 1e-3f));
        }
    }
    if (lane == 0) {
        float* b = boxes + (int64_t)tile * kBoxWords;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            b[a] = __int_as_float(lo[a]);
            b[3 + a] = __int_as_float(hi[a]);
        }
        b[6] = gmax;
    }
  }
}

// corner contributions of one source voxel that land in the owned tile
// (gx, gy, gz): the source's sampling coordinates lin[.] + d, formed by the caller
__device__ __forceinline__ void adjoint_scatter(const float gx, const float gy, const float gz, const int ox, const int oy, const int oz,
                                                const float G0, const float G1, const float G2,
                                                unsigned long long* __restrict__ acc, const float scale, const Vol vol) {
    const AxisTap tx = axis_tap(gx, vol.W);
    const AxisTap ty = axis_tap(gy, vol.H);
    const AxisTap tz = axis_tap(gz, vol.D);
    // tile-local corner coordinates; a corner contributes iff it lies inside the owned tile
    const int ax0 = tx.i0 - ox, ax1 = tx.i1 - ox, ay0 = ty.i0 - oy, ay1 = ty.i1 - oy, az0 = tz.i0 - oz, az1 = tz.i1 - oz;
    if (!(ax1 >= 0 && ax0 < ETX && ay1 >= 0 && ay0 < ETY && az1 >= 0 && az0 < ETZ)) return;
    // per-axis weights with out-of-tile corners zeroed: the scatter becomes 8 unconditional-weight products
    const float wx0 = (unsigned)ax0 < (unsigned)ETX ? tx.w0 : 0.0f, wx1 = (unsigned)ax1 < (unsigned)ETX ? tx.w1 : 0.0f;
    const float wy0 = (unsigned)ay0 < (unsigned)ETY ? ty.w0 : 0.0f, wy1 = (unsigned)ay1 < (unsigned)ETY ? ty.w1 : 0.0f;
    const float wz0 = (unsigned)az0 < (unsigned)ETZ ? tz.w0 : 0.0f, wz1 = (unsigned)az1 < (unsigned)ETZ ? tz.w1 : 0.0f;
    const int cxa = min(max(ax0, 0), ETX - 1), cxb = min(max(ax1, 0), ETX - 1);
    const int cya = min(max(ay0, 0), ETY - 1) * ETX, cyb = min(max(ay1, 0), ETY - 1) * ETX;
    const int cza = min(max(az0, 0), ETZ - 1) * (ETX * ETY), czb = min(max(az1, 0), ETZ - 1) * (ETX * ETY);
    const float S0 = G0 * scale, S1 = G1 * scale, S2 = G2 * scale;  // |S| < 2^30 (scale is a power of two: exact)
#pragma unroll
    for (int cz = 0; cz < 2; ++cz)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy) {
            const float wyz = (cy ? wy1 : wy0) * (cz ? wz1 : wz0);
            if (wyz == 0.0f) continue;
            const int row = (cz ? czb : cza) + (cy ? cyb : cya);
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
                const float w = (cx ? wx1 : wx0) * wyz;
                if (w == 0.0f) continue;
                const int t = row + (cx ? cxb : cxa);
                atomicAdd(&acc[t], (unsigned long long)(long long)__float2int_rn(w * S0));
                atomicAdd(&acc[ETN + t], (unsigned long long)(long long)__float2int_rn(w * S1));
                atomicAdd(&acc[2 * ETN + t], (unsigned long long)(long long)__float2int_rn(w * S2));
            }
        }
}

// identity path + grid-gradient of an own voxel; the tap values come from the staged box when the position stays inside it
template <bool PRESCALE, int H>
__device__ __forceinline__ void adjoint_self(const int x, const int y, const int z, const float d0, const float d1, const float d2,
                                             const int ox, const int oy, const int oz, const float* __restrict__ Gc, const Lay3 LG,
                                             const float* __restrict__ gs_, const float* __restrict__ c0, const Lay3 LD,
                                             const float* __restrict__ lds, const Vol vol, const Lin lin, const Scale3L sc,
                                             float (&out)[3]) {
    using B = ExpBox<H>;
    const AxisTap tx = axis_tap(__fadd_rn(lin.x[x], d0), vol.W);
    const AxisTap ty = axis_tap(__fadd_rn(lin.y[y], d1), vol.H);
    const AxisTap tz = axis_tap(__fadd_rn(lin.z[z], d2), vol.D);
    const int64_t g = ((int64_t)z * vol.H + y) * vol.W + x;
    const float gsc = gs_ ? gs_[g] : 1.0f;
    float G0, G1, G2;
    if (LG.em == 3) {
        const F3 v = ld3g(Gc + g * 3);
        G0 = v.x * gsc;
        G1 = v.y * gsc;
        G2 = v.z * gsc;
    } else {
        G0 = Gc[g] * gsc;
        G1 = Gc[LG.cs + g] * gsc;
        G2 = Gc[2 * LG.cs + g] * gsc;
    }
    const int bx0 = tx.i0 - (ox - H), bx1 = tx.i1 - (ox - H);
    const int by0 = ty.i0 - (oy - H), by1 = ty.i1 - (oy - H);
    const int bz0 = tz.i0 - (oz - H), bz1 = tz.i1 - (oz - H);
    const bool in_lds = H > 0 && bx0 >= 0 && bx1 < B::SX && by0 >= 0 && by1 < B::SY && bz0 >= 0 && bz1 < B::SZ;
    float gix = 0.0f, giy = 0.0f, giz = 0.0f;
#pragma unroll
    for (int cz = 0; cz < 2; ++cz)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
                float v0, v1, v2;
                if (in_lds) {
                    const int idx = ((cz ? bz1 : bz0) * B::SY + (cy ? by1 : by0)) * B::SX + (cx ? bx1 : bx0);
                    v0 = lds[idx];
                    v1 = lds[B::SN + idx];
                    v2 = lds[2 * B::SN + idx];
                } else {
                    const int64_t idx = ((int64_t)(cz ? tz.i1 : tz.i0) * vol.H + (cy ? ty.i1 : ty.i0)) * vol.W + (cx ? tx.i1 : tx.i0);
                    if (LD.em == 3) {  // interleaved field: the three channels of a corner in one 12-byte load
                        const F3 v = ld3g(c0 + idx * 3);
                        v0 = PRESCALE ? prescale(v.x, sc.nm1[0], sc.rnm1[0], sc.inv_pow) : v.x;
                        v1 = PRESCALE ? prescale(v.y, sc.nm1[1], sc.rnm1[1], sc.inv_pow) : v.y;
                        v2 = PRESCALE ? prescale(v.z, sc.nm1[2], sc.rnm1[2], sc.inv_pow) : v.z;
                    } else {
                        v0 = ldp<PRESCALE>(c0, idx, sc.nm1[0], sc.rnm1[0], sc.inv_pow);
                        v1 = ldp<PRESCALE>(c0 + LD.cs, idx, sc.nm1[1], sc.rnm1[1], sc.inv_pow);
                        v2 = ldp<PRESCALE>(c0 + 2 * LD.cs, idx, sc.nm1[2], sc.rnm1[2], sc.inv_pow);
                    }
                }
                const float wx = cx ? tx.w1 : tx.w0, wy = cy ? ty.w1 : ty.w0, wz = cz ? tz.w1 : tz.w0;
                const float dot = v0 * G0 + v1 * G1 + v2 * G2;
                gix += (cx ? dot : -dot) * (wy * wz);
                giy += (cy ? dot : -dot) * (wx * wz);
                giz += (cz ? dot : -dot) * (wx * wy);
            }
    out[0] = G0 + tx.gmul * gix;
    out[1] = G1 + ty.gmul * giy;
    out[2] = G2 + tz.gmul * giz;
}

#ifdef IRS_LDS_TRACE
// timing experiment (tools/lds_phase_trace.py; build with tools/build_variant.sh ldstrace -DIRS_LDS_TRACE): wall-clock stamps
// (100 MHz) of one thread of one workgroup at the phase boundaries of its tiles, plus the source box it walked
__device__ unsigned long long g_lds_trace[8 * 32];
extern "C" __attribute__((visibility("default"))) int irs_debug_lds_trace(unsigned long long* out) {  // (trace builds only: the library is compiled -fvisibility=hidden)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lds_trace), sizeof(g_lds_trace)) == hipSuccess ? 0 : 1;
}
#define IRS_LT(slot, val)                                                                                  \
    do {                                                                                                   \
        if (blockIdx.x == 37 && threadIdx.x == 0 && lt_i < 32) g_lds_trace[lt_i * 8 + (slot)] = (val);     \
    } while (0)
#else
#define IRS_LT(slot, val)
#endif
// waves per SIMD the unstaged variant is COMPILED for.  Its 49 KB of accumulators allow three workgroups per CU whatever the
// register count, so the cap of 128 registers that "4" implies (60 bytes of scratch) buys no occupancy -- and still wins: compiled
// for 3 (153 registers, no scratch) a transition of the 6- / 12-voxel displaced runs took 5.79 / 6.80 instead of 5.41 / 6.09 ms on
// one box (round 4; two sources in flight per wave or three made no difference)
#ifndef IRS_LDS_WAVES
#define IRS_LDS_WAVES 4
#endif
#ifndef IRS_LDS_PIPE
#define IRS_LDS_PIPE 2  // wavefront steps of sources requested ahead of their scatter
#endif
template <bool PRESCALE, int H>
__global__ __launch_bounds__(kExpBlock, H == 0 ? IRS_LDS_WAVES : 2) void exp_bwd_lds_kernel(const float* __restrict__ G, const float* __restrict__ dk,
                                                                float* __restrict__ gout, Vol vol, Lin lin, Scale3L sc,
                                                                const unsigned* __restrict__ dmax, TileGrid tg,
                                                                int gather_radius, const float* __restrict__ gscale, int lay,
                                                                const float* __restrict__ cmm, const float* __restrict__ boxes) {
    using B = ExpBox<H>;
    const Lay3 LD = lay3(lay & 1, vol.V), LG = lay3(lay & 2, vol.V), LO = lay3(lay & 4, vol.V);
    constexpr bool STAGED = H > 0;  // H = 0: no staged copy of d at all (sources and taps are far from the tile anyway)
    __shared__ float lds[STAGED ? 3 * B::SN : 1];
    __shared__ unsigned long long acc[3 * ETN];
    __shared__ float bred[kCmm * (kExpBlock / kWave)];
    {   // nothing to do for any chain (the usual case): leave before walking the tile list
        const int chains = tg.total / (tg.ntx * tg.nty * tg.ntz);
        bool any = false;
        for (int c = 0; c < chains; ++c)
            any |= (int)floorf(fmaxf(fmaxf(__uint_as_float(dmax[c * 4 + 0]), __uint_as_float(dmax[c * 4 + 1])), __uint_as_float(dmax[c * 4 + 2]))) + 1 > gather_radius;
        if (!any) return;
    }
#ifdef IRS_LDS_TRACE
  int lt_i = -1;
#endif
  for (int tile = blockIdx.x; tile < tg.total; tile += gridDim.x) {
#ifdef IRS_LDS_TRACE
    ++lt_i;
#endif
    IRS_LT(0, wall_clock64());
    int t_ = tile;
    const int ox = (t_ % tg.ntx) * ETX;
    t_ /= tg.ntx;
    const int oy = (t_ % tg.nty) * ETY;
    t_ /= tg.nty;
    const int oz = vol.z0 + (t_ % tg.ntz) * ETZ;
    const int chain = t_ / tg.ntz;
    // source halo: a voxel at distance h from the tile can reach it iff h <= floor(max|d|) + 1 (corner = floor(x+d) + {0,1})
    const int hx = (int)floorf(__uint_as_float(dmax[chain * 4 + 0])) + 1;
    const int hy = (int)floorf(__uint_as_float(dmax[chain * 4 + 1])) + 1;
    const int hz = (int)floorf(__uint_as_float(dmax[chain * 4 + 2])) + 1;
    if (hx <= gather_radius && hy <= gather_radius && hz <= gather_radius) continue;  // a gather kernel owns this chain
    const int64_t V = vol.V;
    const int64_t cb = (int64_t)chain * 3 * V;
    const float* c0 = dk + cb;
    const float* Gc = G + cb;
    const float* gsc_ = gscale ? gscale + (int64_t)chain * V : nullptr;

    if (STAGED) stage_field<PRESCALE, H>(c0, LD, lds, ox, oy, oz, vol, sc);
    for (int i = threadIdx.x; i < 3 * ETN; i += kExpBlock) acc[i] = 0ull;

    // ---- source box and the largest |G| inside it
    // (z: the tile may stick out of the launch WINDOW -- a slab's boundary strip of a few planes.  Only sources within hz of the
    // window's last plane can reach an output inside it; the planes beyond are not part of this exchange round and hold whatever
    // an earlier transition left there -- a stale source with a stale, larger displacement would otherwise scatter into the strip)
    const int zend = min(oz + ETZ, vol.z0 + vol.nz) - 1;
    int lo[3] = {max(ox - hx, 0), max(oy - hy, 0), max(oz - hz, 0)};
    int hi[3] = {min(ox + ETX - 1 + hx, vol.W - 1), min(oy + ETY - 1 + hy, vol.H - 1), min(zend + hz, vol.D - 1)};
    float gmax = 0.0f;
    auto block_reduce7 = [&](float (&m7)[kCmm]) {  // min over even slots < 6, max over the others; result in every thread
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1)
#pragma unroll
            for (int j = 0; j < kCmm; ++j) {
                const float o = __shfl_down(m7[j], off, kWave);
                m7[j] = (j < 6 && !(j & 1)) ? fminf(m7[j], o) : fmaxf(m7[j], o);
            }
        if ((threadIdx.x & (kWave - 1)) == 0)
#pragma unroll
            for (int j = 0; j < kCmm; ++j) bred[j * (kExpBlock / kWave) + threadIdx.x / kWave] = m7[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kCmm; ++j) {
            float r = bred[j * (kExpBlock / kWave)];
            for (int w = 1; w < kExpBlock / kWave; ++w) {
                const float o = bred[j * (kExpBlock / kWave) + w];
                r = (j < 6 && !(j & 1)) ? fminf(r, o) : fmaxf(r, o);
            }
            m7[j] = r;
        }
        __syncthreads();  // bred is reused
    };
    if (boxes) {  // source box and max |G| of this tile from tile_box_kernel (the same values as the rounds below produce)
        const float* __restrict__ b = boxes + (int64_t)tile * kBoxWords;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = __float_as_int(b[a]);
            hi[a] = __float_as_int(b[3 + a]);
        }
        gmax = b[6];
    } else if (cmm) {
        const int nax[3] = {vol.W, vol.H, vol.D};
        const int tlo[3] = {ox, oy, oz};
        const int thi[3] = {min(ox + ETX, vol.W) - 1, min(oy + ETY, vol.H) - 1, min(zend, vol.D - 1)};
        const int ncx = (vol.W + kCell - 1) / kCell, ncy = (vol.H + kCell - 1) / kCell, ncz = (vol.D + kCell - 1) / kCell;
        const float* __restrict__ cm = cmm + (int64_t)chain * ncx * ncy * ncz * kCmm;
        // two rounds; max |G| is the one of the second round's cells, a superset of the final box: the scale may come out a power
        // of two smaller than the final box would allow, never too large
        for (int round = 0; round < 2; ++round) {
            const int c0x = lo[0] / kCell, c0y = lo[1] / kCell, c0z = lo[2] / kCell;
            const int nx_ = hi[0] / kCell - c0x + 1, ny_ = hi[1] / kCell - c0y + 1, nz_ = hi[2] / kCell - c0z + 1;
            float m7[kCmm] = {3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f, 0.0f};
            for (int i = threadIdx.x; i < nx_ * ny_ * nz_; i += kExpBlock) {
                const int cell = ((c0z + i / (nx_ * ny_)) * ncy + c0y + (i / nx_) % ny_) * ncx + c0x + i % nx_;
#pragma unroll
                for (int j = 0; j < kCmm; ++j) {
                    const float o = cm[cell * kCmm + j];
                    m7[j] = (j < 6 && !(j & 1)) ? fminf(m7[j], o) : fmaxf(m7[j], o);
                }
            }
            block_reduce7(m7);
            gmax = m7[6];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float mn = m7[2 * a], mx = m7[2 * a + 1];
                if (!(mn <= mx) || fabsf(mn) > 1.0e6f || fabsf(mx) > 1.0e6f) continue;  // no usable extrema: keep the box
                // on a tile that touches the volume border the clip folds arbitrarily distant positions onto it: that side
                // keeps B0's extent.  1e-3: the cell extrema are v (n-1)/2, the positions ((g + 1) / 2) (n-1) -- other rounding
                if (tlo[a] - 1 > 0) lo[a] = max(lo[a], (int)floorf((float)(tlo[a] - 1) - mx - 1e-3f));
                if (thi[a] + 1 < nax[a] - 1) hi[a] = min(hi[a], (int)ceilf((float)(thi[a] + 1) - mn + 1e-3f));
            }
        }
    } else {  // no coarse grid (IRS_COARSE_BOX=0, parity runs): B0 and a pass over its sources for max |G|
        float m7[kCmm] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        const int ex0 = hi[0] - lo[0] + 1, ey0 = hi[1] - lo[1] + 1, ez0 = hi[2] - lo[2] + 1;
        for (int i = threadIdx.x; i < ex0 * ey0 * ez0; i += kExpBlock) {
            const int64_t g = ((int64_t)(lo[2] + i / (ex0 * ey0)) * vol.H + lo[1] + (i / ex0) % ey0) * vol.W + lo[0] + i % ex0;
            const float gsc = gsc_ ? gsc_[g] : 1.0f;
#pragma unroll
            for (int a = 0; a < 3; ++a) m7[6] = fmaxf(m7[6], fabsf(Gc[a * LG.cs + g * LG.em] * gsc));
        }
        block_reduce7(m7);
        gmax = m7[6];
    }
    // per-tile power-of-two scale: gmax = m 2^e with m in [0.5, 1) -> |w G scale| < 2^30 for every contribution
    int e2 = 0;
    if (gmax > 0.0f) (void)frexpf(gmax, &e2);
    const bool usable = gmax > 0.0f && gmax < 3.0e38f;
    const float scale = usable ? ldexpf(1.0f, 30 - e2) : 0.0f, inv_scale = usable ? ldexpf(1.0f, e2 - 30) : 0.0f;
    __syncthreads();  // acc zeroed, lds staged
    IRS_LT(1, wall_clock64());

    auto disp_at = [&](int x, int y, int z, float& d0, float& d1, float& d2) {
        const int bx = x - (ox - H), by = y - (oy - H), bz = z - (oz - H);
        if (STAGED && (unsigned)bx < (unsigned)B::SX && (unsigned)by < (unsigned)B::SY && (unsigned)bz < (unsigned)B::SZ) {
            const int ctr = (bz * B::SY + by) * B::SX + bx;
            d0 = lds[ctr];
            d1 = lds[B::SN + ctr];
            d2 = lds[2 * B::SN + ctr];
        } else {
            const int64_t g = ((int64_t)z * vol.H + y) * vol.W + x;
            d0 = ldp<PRESCALE>(c0, g * LD.em, sc.nm1[0], sc.rnm1[0], sc.inv_pow);
            d1 = ldp<PRESCALE>(c0 + LD.cs, g * LD.em, sc.nm1[1], sc.rnm1[1], sc.inv_pow);
            d2 = ldp<PRESCALE>(c0 + 2 * LD.cs, g * LD.em, sc.nm1[2], sc.rnm1[2], sc.inv_pow);
        }
    };
    // ---- scatter: the sources of the box in one flat order (x fastest), 64 consecutive ones per wavefront step -- a row of the box
    // is ~36 sources long, so a wavefront per row left 44 % of the lanes idle, and the loop is bound by instruction issue (~250
    // per source).  Everything a source needs from memory (d, G, the three identity-grid coordinates) is requested kPipe steps
    // ahead of its scatter.
    const int ex = max(hi[0] - lo[0] + 1, 0), ey = max(hi[1] - lo[1] + 1, 0), ez = max(hi[2] - lo[2] + 1, 0);
    const unsigned exy = (unsigned)(ex * ey), nsrc = exy * (unsigned)ez;
    // idx / exy and rem / ex by multiply-high with m = floor(2^32 / d) + 1: exact while idx * d < 2^32 -- and for d >= 2: with
    // d == 1 the multiplier wraps to 0 (a clipped edge tile whose box is one column wide: plain division then)
    const bool magic_ok = ex > 1 && exy > 1 && (unsigned long long)nsrc * exy < (1ull << 32);
    const unsigned m_xy = magic_ok ? 0xFFFFFFFFu / exy + 1u : 0u, m_x = magic_ok ? 0xFFFFFFFFu / (unsigned)ex + 1u : 0u;
    const int nitems = (int)((nsrc + kWave - 1) / kWave);
    constexpr int kWaves = kExpBlock / kWave, kPipe = IRS_LDS_PIPE;
    struct Src {  // the identity-grid coordinates stay separate from d until the scatter: adding them here would wait for the loads
        float d0, d1, d2, l0, l1, l2, G0, G1, G2, gsc;
        bool ok;
    };
    auto load_item = [&](int it, Src& q) {
        q.ok = false;
        if (it >= nitems) return;  // wave-uniform
        const unsigned idx = (unsigned)it * kWave + (threadIdx.x & (kWave - 1));
        q.ok = idx < nsrc;
        if (!q.ok) return;
        unsigned zq, yq, xq;
        if (magic_ok) {
            zq = __umulhi(idx, m_xy);
            const unsigned rem = idx - zq * exy;
            yq = __umulhi(rem, m_x);
            xq = rem - yq * (unsigned)ex;
        } else {
            zq = idx / exy;
            const unsigned rem = idx % exy;
            yq = rem / (unsigned)ex;
            xq = rem % (unsigned)ex;
        }
        const int x = lo[0] + (int)xq, y = lo[1] + (int)yq, z = lo[2] + (int)zq;
        const int64_t g = ((int64_t)z * vol.H + y) * vol.W + x;
        if (!STAGED && LD.em == 3) {  // interleaved field: one 12-byte load
            const F3 v = ld3g(c0 + g * 3);
            q.d0 = v.x;
            q.d1 = v.y;
            q.d2 = v.z;
        } else if (!STAGED) {
            q.d0 = c0[g];
            q.d1 = c0[LD.cs + g];
            q.d2 = c0[2 * LD.cs + g];
        } else {
            disp_at(x, y, z, q.d0, q.d1, q.d2);  // staged copies are already prescaled
        }
        q.gsc = gsc_ ? gsc_[g] : 1.0f;  // optional scalar factor of the incoming gradient (fused backward warp)
        if (LG.em == 3) {
            const F3 v = ld3g(Gc + g * 3);
            q.G0 = v.x;
            q.G1 = v.y;
            q.G2 = v.z;
        } else {
            q.G0 = Gc[g];
            q.G1 = Gc[LG.cs + g];
            q.G2 = Gc[2 * LG.cs + g];
        }
        q.l0 = lin.x[x];
        q.l1 = lin.y[y];
        q.l2 = lin.z[z];
    };
    auto scatter_item = [&](const Src& q) {
        const bool raw = PRESCALE && !STAGED;  // values straight from global memory still need the prescale of step 0
        const float d0 = raw ? prescale(q.d0, sc.nm1[0], sc.rnm1[0], sc.inv_pow) : q.d0;
        const float d1 = raw ? prescale(q.d1, sc.nm1[1], sc.rnm1[1], sc.inv_pow) : q.d1;
        const float d2 = raw ? prescale(q.d2, sc.nm1[2], sc.rnm1[2], sc.inv_pow) : q.d2;
        adjoint_scatter(__fadd_rn(q.l0, d0), __fadd_rn(q.l1, d1), __fadd_rn(q.l2, d2), ox, oy, oz, q.G0 * q.gsc, q.G1 * q.gsc, q.G2 * q.gsc,
                        acc, scale, vol);
    };
    {
        const int w0 = threadIdx.x / kWave;
        Src cur[kPipe], nxt[kPipe];
#pragma unroll
        for (int u = 0; u < kPipe; ++u) load_item(w0 + u * kWaves, cur[u]);
        for (int it = w0; it < nitems; it += kWaves * kPipe) {
#pragma unroll
            for (int u = 0; u < kPipe; ++u) load_item(it + (kPipe + u) * kWaves, nxt[u]);
#pragma unroll
            for (int u = 0; u < kPipe; ++u)
                if (cur[u].ok) scatter_item(cur[u]);
#pragma unroll
            for (int u = 0; u < kPipe; ++u) cur[u] = nxt[u];
        }
    }
    __syncthreads();
    IRS_LT(2, wall_clock64());
    IRS_LT(4, (unsigned long long)ex | ((unsigned long long)ey << 16) | ((unsigned long long)ez << 32));
    // ---- own voxels: identity path + grid-gradient (no atomics: one thread per output) + the scattered sum.  The displacements
    // of all of a thread's outputs are requested first (their taps depend on them)
    float* o = gout + cb;
    constexpr int kOwn = ETN / kExpBlock;
    float sd[kOwn][3];
#pragma unroll
    for (int j = 0; j < kOwn; ++j) {
        const int i = threadIdx.x + j * kExpBlock;
        const int x = min(ox + i % ETX, vol.W - 1), y = min(oy + (i / ETX) % ETY, vol.H - 1), z = min(oz + i / (ETX * ETY), vol.z0 + vol.nz - 1);
        if (!STAGED && LD.em == 3) {
            const F3 v = ld3g(c0 + (((int64_t)z * vol.H + y) * vol.W + x) * 3);
            sd[j][0] = PRESCALE ? prescale(v.x, sc.nm1[0], sc.rnm1[0], sc.inv_pow) : v.x;
            sd[j][1] = PRESCALE ? prescale(v.y, sc.nm1[1], sc.rnm1[1], sc.inv_pow) : v.y;
            sd[j][2] = PRESCALE ? prescale(v.z, sc.nm1[2], sc.rnm1[2], sc.inv_pow) : v.z;
        } else {
            disp_at(x, y, z, sd[j][0], sd[j][1], sd[j][2]);
        }
    }
#pragma unroll
    for (int j = 0; j < kOwn; ++j) {
        const int i = threadIdx.x + j * kExpBlock;
        const int lx = i % ETX, ly = (i / ETX) % ETY, lz = i / (ETX * ETY);
        const bool valid = ox + lx < vol.W && oy + ly < vol.H && oz + lz < vol.z0 + vol.nz;
        const int x = min(ox + lx, vol.W - 1), y = min(oy + ly, vol.H - 1), z = min(oz + lz, vol.z0 + vol.nz - 1);
        float self[3];
        adjoint_self<PRESCALE, H>(x, y, z, sd[j][0], sd[j][1], sd[j][2], ox, oy, oz, Gc, LG, gsc_, c0, LD, lds, vol, lin, sc, self);
        if (!valid) continue;
        const int64_t g = ((int64_t)z * vol.H + y) * vol.W + x;
        const float o0 = self[0] + (float)(long long)acc[i] * inv_scale, o1 = self[1] + (float)(long long)acc[ETN + i] * inv_scale,
                    o2 = self[2] + (float)(long long)acc[2 * ETN + i] * inv_scale;
        if (LO.em == 3) {
            const F3 v = {o0, o1, o2};
            __builtin_memcpy(o + g * 3, &v, 12);
        } else {
            o[g] = o0;
            o[LO.cs + g] = o1;
            o[2 * LO.cs + g] = o2;
        }
    }
    __syncthreads();
    IRS_LT(3, wall_clock64());
  }
}

static size_t coarse_cells_floats(Vol vol, int C) {
    const size_t n = kCmm * (size_t)C * ((vol.W + kCell - 1) / kCell) * ((vol.H + kCell - 1) / kCell) * ((vol.D + kCell - 1) / kCell);
    return (n + 63) / 64 * 64;
}
// the coarse cells, then the per-tile boxes of tile_box_kernel (a launch window never has more tiles than the whole volume)
size_t coarse_minmax_bytes(Vol vol, int C) {
    const size_t tiles = (size_t)C * ((vol.W + ETX - 1) / ETX) * ((vol.H + ETY - 1) / ETY) * ((vol.D + ETZ - 1) / ETZ + 1);
    return sizeof(float) * (coarse_cells_floats(vol, C) + tiles * kBoxWords);
}

void launch_exp_step_bwd_lds(const float* G, const float* dk, float* gout, bool prescale_in, int no_steps, int C, Vol vol,
                             Lin lin, const unsigned* dmax, int halo, int gather_radius, const float* gscale, int lay,
                             float* cmm, hipStream_t st) {
    if (vol.nzb > 0) {  // two windows (slab boundary strips): this rarely selected kernel takes them one launch each
        launch_exp_step_bwd_lds(G, dk, gout, prescale_in, no_steps, C, window(vol, vol.z0, vol.z0 + vol.nz), lin, dmax, halo,
                                gather_radius, gscale, lay, cmm, st);
        launch_exp_step_bwd_lds(G, dk, gout, prescale_in, no_steps, C, window(vol, vol.z0b, vol.z0b + vol.nzb), lin, dmax, halo,
                                gather_radius, gscale, lay, cmm, st);
        return;
    }
    TileGrid tz;
    (void)exp_grid(vol, C, &tz);

    const Scale3L sc = make_scale_l(vol, no_steps);
    float* boxes = nullptr;
    if (!global_knobs().coarse_box) cmm = nullptr;  // parity test of the two source boxes
    if (cmm) {  // coarse displacement extrema / gradient maxima (coarse_minmax_bytes(vol, C) of scratch)
        const int ncx = (vol.W + kCell - 1) / kCell, ncy = (vol.H + kCell - 1) / kCell, ncz = (vol.D + kCell - 1) / kCell;
        const dim3 cg((unsigned)(((ncx + 7) / 8) * ncy * ncz), (unsigned)C);
        if (prescale_in) hipLaunchKernelGGL((coarse_minmax_kernel<true>), cg, dim3(kWave), 0, st, dk, G, gscale, cmm, vol, sc, dmax, gather_radius, ncx, ncy, ncz, lay);
        else hipLaunchKernelGGL((coarse_minmax_kernel<false>), cg, dim3(kWave), 0, st, dk, G, gscale, cmm, vol, sc, dmax, gather_radius, ncx, ncy, ncz, lay);
        if (global_knobs().tile_box) {  // ... and from them the source box + scale of every tile, one wavefront each
            boxes = cmm + coarse_cells_floats(vol, C);
            hipLaunchKernelGGL(tile_box_kernel, dim3((unsigned)min((tz.total + 3) / 4, kBoxGrid)), dim3(kBoxBlock), 0, st, cmm, boxes, vol, dmax, tz, gather_radius);
        }
    }
    // the persistent grid is ONE resident set of workgroups (asked from the runtime once per variant): a grid that is not a
    // multiple of it leaves a partial last round in which most of the chip idles
#define IRS_BWD(P, HH)                                                                                                           \
    do {                                                                                                                         \
        static int resident = 0;                                                                                                 \
        if (!resident) {                                                                                                         \
            int per_cu = 0, dev = 0, cus = 0;                                                                                    \
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && \
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, exp_bwd_lds_kernel<P, HH>, kExpBlock, 0) == hipSuccess && per_cu > 0 && cus > 0) \
                resident = per_cu * cus;                                                                                         \
            else                                                                                                                 \
                resident = kExpGridCap;                                                                                          \
        }                                                                                                                        \
        const dim3 g_((unsigned)(tz.total < resident ? tz.total : resident));                                                    \
        hipLaunchKernelGGL((exp_bwd_lds_kernel<P, HH>), g_, dim3(kExpBlock), 0, st, G, dk, gout, vol, lin, sc, dmax, tz, gather_radius, gscale, lay, cmm, (const float*)boxes); \
    } while (0)
    // With the gather variants in front (gather_radius >= 2) the staged box of d around the tile is of little use (sources
    // and taps are far away): H = 0 stages nothing (49 KB of accumulators instead of 111 KB of LDS -> three workgroups per CU)
    if (gather_radius >= 2 || global_knobs().lds_from == 2) halo = 0;
    if (prescale_in) {
        if (halo <= 0) IRS_BWD(true, 0); else if (halo <= 1) IRS_BWD(true, 1); else IRS_BWD(true, 2);
    } else {
        if (halo <= 0) IRS_BWD(false, 0); else if (halo <= 1) IRS_BWD(false, 1); else IRS_BWD(false, 2);
    }
#undef IRS_BWD
}

// ------------------------------------------------------------------------------------------------
// adjoint step, z-marching gather.
// A 256-thread workgroup owns a 32x8 column of outputs over a z-segment and marches through the source planes, keeping
// only a ring of NP = 2R+1 planes of source records in LDS (R=1: 37 KB -> 4 workgroups per CU).  Per plane step s:
//   write the prefetched plane s into ring slot s mod NP (d, clipped sampling position relative to the source voxel, G:
//   9 floats per source, 8-byte fields),
//   issue the global loads of plane s+1 (in flight during the compute), barrier,
//   every thread reads the (2R+1)^2 records around its (x,y) once and adds them into the NP output accumulators it holds
//   in registers (planes s-R..s+R), then output plane s-R is complete: add the identity path + grid-gradient
//   (taps from the ring) and store it.
// Halo redundancy is (32+2R)(8+2R)/(32*8) in-plane only (1.33x for R=1).
// ------------------------------------------------------------------------------------------------
#ifndef IRS_MTX
#define IRS_MTX 32
#define IRS_MTY 8
#endif
constexpr int MTX = IRS_MTX, MTY = IRS_MTY, kMarchBlock = MTX * MTY;
#ifndef IRS_GATHER_UNROLL_Y
#define IRS_GATHER_UNROLL_Y 1
#endif
#ifndef IRS_MARCH_WAVES
#define IRS_MARCH_WAVES 4
#endif
#ifndef IRS_MARCH_WAVES_R2
#define IRS_MARCH_WAVES_R2 3  // radius-2 gather: 36 KB of LDS since round 4 -- three workgroups per CU (<= 168 VGPRs)
#endif
// Two round-3 experiments on the radius-1 adjoint, both measured SLOWER and not kept:
//  * its own-term global-memory branch is dead code for max|d_k| < 1; compiled out (125 instead of 128 VGPRs) the kernel took
//    217.5 instead of 202.0 us per launch -- the smaller kernel schedules worse;
//  * a build that records, per committed source plane, which signs the relative positions take per axis (six __any + one LDS
//    word per wave) and skips the gather candidates whose hat weight is then exactly 0 behind scalar branches (27 -> 18 / 12 / 8
//    candidates; bit-identical results): 199 instead of 166 us per launch on a smooth 6-voxel field where most planes qualify,
//    243 instead of 199 us on a smooth 3-voxel one.
// The radius-1 adjoint only runs for max|d_k| < 1 (selected on the device from the exact bound): the eight corners of a voxel's
// own sample then ALWAYS sit in the ring, and the global-memory fallback for taps that leave it is dead code -- which stays in
// (1): compiled out (0) the kernel is 3 VGPRs smaller and 7 % SLOWER (219 against 204 us per launch, two same-box A/B runs).
// Compile-time layouts in the adjoint (as in the forward step, where they removed a wait from the middle of the prefetch and gave
// 5 %): measured 221 against 204 us per radius-1 launch (24 bytes of scratch at the 128-VGPR limit) -- 0
#ifndef IRS_BWD_LAYC
#define IRS_BWD_LAYC 0
#endif
#ifndef IRS_BWD_TAPS
#define IRS_BWD_TAPS 0  // 1: the eight corner reads of the own term as single ds_read_b64 / ds_read_b32 (volatile, as IRS_FWD_TAPS); A/B in round 5
#endif
#ifndef IRS_FWD_REC16
#define IRS_FWD_REC16 0
#endif
#ifndef IRS_FWD_FMA
// The 24 tap products of a sample accumulated with fused multiply-adds (round 5).  The COORDINATE arithmetic -- positions, cell
// indices, weights: what decides which cell a sample falls into and what the reference's gradient is compared against -- keeps
// ATen's order (common.h: axis_tap); the accumulation a += t * w rounds once instead of twice per tap.  Until round 4 it was kept
// FMA-free for bit-identity with ATen's CPU sampler; measured now that the LDS is no longer what the kernel waits for: 87.0-88.1
// against 93.5-94.8 us per step at 256^3, 4.296-4.331 against 4.385-4.404 ms per transition, 2 spilled VGPRs instead of 10 --
// and every parity test with unchanged margins (256^3: 1.037e-4 voxels against the oracle, as before; the 32^3 fixtures go from
// exactly 0 to 1.6e-6 of a tolerance of 1e-4): profiles/r05_fwdfma_ab.txt, profiles/parity_report_r05.json.
#define IRS_FWD_FMA 1
#endif
#ifndef IRS_FWD_BUFLOAD
#define IRS_FWD_BUFLOAD 0  // marching forward step: staging loads as buffer loads; A/B in round 5
#endif
#ifndef IRS_BWD_BUFLOAD
#define IRS_BWD_BUFLOAD 0  // marching adjoint: staging loads as buffer loads (descriptor + 32-bit lane offset); A/B in round 5
#endif
#ifndef IRS_BWD_R1_FALLBACK
#define IRS_BWD_R1_FALLBACK 1
#endif
#ifndef IRS_BWD_PEEL
// radius-1 adjoint: run-in plane steps update only the accumulators whose output plane is inside the segment (6 of the 30
// accumulator updates of an 8-plane segment land outside it).  Built in round 5 as asked (per-mask instantiations of the gather
// behind a wave-uniform dispatch; the full gather, commit and own-term blocks unchanged instruction for instruction,
// tools/debug/isa_block_diff.py; chains bit-identical) and measured SLOWER on one box, three / two alternating repetitions:
// 128^3 36.6-37.2 against 35.5-35.8 us per launch, 256^3 223-229 against 215-218 (profiles/r05_peel_ab.txt) -- the kernel grows from
// 2 687 to 4 363 instructions and the dispatch sits in front of every gather.  Off.
#define IRS_BWD_PEEL 0
#endif
// hat of (r + c) for a relative position r and a compile-time integer offset c.
template <int R>
__device__ __forceinline__ float rel_hat(float r, int c);

// hat function max(0, 1 - |t|).  v_med3_f32 folds into the clamp output modifier of the subtraction (one VALU op);
// HIP's __saturatef compiles to two compares and two selects.
// load p[byte_off / 4] with the address written as (uniform 64-bit base) + (32-bit lane byte offset).  Planes are < 4 GiB, so a
// 32-bit byte offset always suffices.  (What the compiler makes of it in the marching kernels is NOT the global_load saddr form the
// expression invites: the sum is formed in the block in front of the layout branch, so each staging load costs one v_lshl_add_u64
// and the lane offsets live in 64-bit VGPR pairs -- 146 of the adjoint's 148 global loads, as the round-4 review found in the ISA.
// Buffer loads -- descriptor in SGPRs, 32-bit lane offset: IRS_BWD_BUFLOAD / IRS_FWD_BUFLOAD -- remove both (122 instead of 128
// VGPRs in the adjoint, 28 VALU instructions fewer in its loop) and measured no faster: profiles/r05_bufload_ab.txt.)
__device__ __forceinline__ float ld_off(const float* __restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
// one 12-byte access per lane for an interleaved field (global_load / store_dwordx3; 4-byte alignment suffices)
__device__ __forceinline__ F3 ld3_off(const float* __restrict__ base, unsigned byte_off) {
    F3 v;
    __builtin_memcpy(&v, reinterpret_cast<const char*>(base) + byte_off, 12);
    return v;
}
__device__ __forceinline__ void st3_off(float* __restrict__ base, unsigned byte_off, float a, float b, float c) {
    const F3 v = {a, b, c};
    __builtin_memcpy(reinterpret_cast<char*>(base) + byte_off, &v, 12);
}
__device__ __forceinline__ void st_off(float* __restrict__ base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// buffer loads: a raw descriptor over "everything from `base` on" (in SGPRs: the base is wave-uniform), 32-bit lane byte offset
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ float buf_ld1(const float* __restrict__ base, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(buf_rsrc(base), (int)byte_off, 0, 0));
}
__device__ __forceinline__ F3 buf_ld3(const float* __restrict__ base, unsigned byte_off) {
    typedef unsigned U3 __attribute__((ext_vector_type(3)));
    const U3 v = __builtin_amdgcn_raw_buffer_load_b96(buf_rsrc(base), (int)byte_off, 0, 0);
    F3 f;
    f.x = __uint_as_float(v.x);
    f.y = __uint_as_float(v.y);
    f.z = __uint_as_float(v.z);
    return f;
}
__device__ __forceinline__ float clamp01(float t) { return __builtin_amdgcn_fmed3f(t, 0.0f, 1.0f); }
__device__ __forceinline__ float hat01(float t) { return clamp01(1.0f - fabsf(t)); }

template <int R>
__device__ __forceinline__ float rel_hat(float r, int c) {
    if (R == 1 && c == 1) return clamp01(-r);
    if (R == 1 && c == -1) return clamp01(r);
    return c == 0 ? hat01(r) : hat01(r + (float)c);
}

template <bool PRESCALE, int R>
struct March {
    static constexpr int NP = 2 * R + 1, PX = MTX + 2 * R, PY = MTY + 2 * R, PN = PX * PY;
    static constexpr int NIT = (PN + kMarchBlock - 1) / kMarchBlock;
};

// Slow but always-correct adjoint of one squaring step for one (x, y) column over [z0, z1): every source within `hs` voxels is
// read from global memory.  Lives only in the rarely selected radius-2 kernel, and only runs when the host did not launch the
// any-radius LDS-scatter kernel for this step (it predicts the displacement bound from earlier transitions, see api.hip) and
// the bound then turned out larger than 2 voxels -- a transient that costs time, never parity.
template <bool PRESCALE>
__device__ __forceinline__ void exp_bwd_generic_column(const float* __restrict__ Gc, const Lay3 LG, const float* __restrict__ gsc,
                                                       const float* __restrict__ c0, const Lay3 LD, float* __restrict__ oc,
                                                       const Lay3 LO, const Vol vol, const Lin lin, const Scale3L sc, const int x,
                                                       const int y, const int z0, const int z1, const int hs) {
    const float nxm = (float)(vol.W - 1), nym = (float)(vol.H - 1), nzm = (float)(vol.D - 1);
    for (int zo = z0; zo < z1; ++zo) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
        for (int sz = max(zo - hs, 0); sz <= min(zo + hs, vol.D - 1); ++sz)
            for (int sy = max(y - hs, 0); sy <= min(y + hs, vol.H - 1); ++sy)
                for (int sx = max(x - hs, 0); sx <= min(x + hs, vol.W - 1); ++sx) {
                    const int64_t idx = ((int64_t)sz * vol.H + sy) * vol.W + sx;
                    const float d0 = ldp<PRESCALE>(c0, idx * LD.em, sc.nm1[0], sc.rnm1[0], sc.inv_pow),
                                d1 = ldp<PRESCALE>(c0 + LD.cs, idx * LD.em, sc.nm1[1], sc.rnm1[1], sc.inv_pow),
                                d2 = ldp<PRESCALE>(c0 + 2 * LD.cs, idx * LD.em, sc.nm1[2], sc.rnm1[2], sc.inv_pow);
                    const float px = __builtin_amdgcn_fmed3f(__fmul_rn(__fmul_rn(__fadd_rn(__fadd_rn(lin.x[sx], d0), 1.0f), 0.5f), nxm), 0.0f, nxm);
                    const float py = __builtin_amdgcn_fmed3f(__fmul_rn(__fmul_rn(__fadd_rn(__fadd_rn(lin.y[sy], d1), 1.0f), 0.5f), nym), 0.0f, nym);
                    const float pz = __builtin_amdgcn_fmed3f(__fmul_rn(__fmul_rn(__fadd_rn(__fadd_rn(lin.z[sz], d2), 1.0f), 0.5f), nzm), 0.0f, nzm);
                    const float w = hat01(px - (float)x) * hat01(py - (float)y) * hat01(pz - (float)zo);
                    if (w > 0.0f) {
                        const float gm = gsc ? gsc[idx] : 1.0f;
                        a0 = fmaf(w, Gc[idx * LG.em] * gm, a0);
                        a1 = fmaf(w, Gc[LG.cs + idx * LG.em] * gm, a1);
                        a2 = fmaf(w, Gc[2 * LG.cs + idx * LG.em] * gm, a2);
                    }
                }
        // grid gradient of the sample taken at this voxel
        const int64_t own = ((int64_t)zo * vol.H + y) * vol.W + x;
        const float gmo = gsc ? gsc[own] : 1.0f;
        const float G0 = Gc[own * LG.em] * gmo, G1 = Gc[LG.cs + own * LG.em] * gmo, G2 = Gc[2 * LG.cs + own * LG.em] * gmo;
        const AxisTap tx = axis_tap(__fadd_rn(lin.x[x], ldp<PRESCALE>(c0, own * LD.em, sc.nm1[0], sc.rnm1[0], sc.inv_pow)), vol.W);
        const AxisTap ty = axis_tap(__fadd_rn(lin.y[y], ldp<PRESCALE>(c0 + LD.cs, own * LD.em, sc.nm1[1], sc.rnm1[1], sc.inv_pow)), vol.H);
        const AxisTap tz = axis_tap(__fadd_rn(lin.z[zo], ldp<PRESCALE>(c0 + 2 * LD.cs, own * LD.em, sc.nm1[2], sc.rnm1[2], sc.inv_pow)), vol.D);
        float gix = 0.0f, giy = 0.0f, giz = 0.0f;
        for (int cz = 0; cz < 2; ++cz)
            for (int cy = 0; cy < 2; ++cy)
                for (int cx = 0; cx < 2; ++cx) {
                    const int64_t idx = ((int64_t)(cz ? tz.i1 : tz.i0) * vol.H + (cy ? ty.i1 : ty.i0)) * vol.W + (cx ? tx.i1 : tx.i0);
                    const float dot = fmaf(ldp<PRESCALE>(c0 + 2 * LD.cs, idx * LD.em, sc.nm1[2], sc.rnm1[2], sc.inv_pow), G2,
                                           fmaf(ldp<PRESCALE>(c0 + LD.cs, idx * LD.em, sc.nm1[1], sc.rnm1[1], sc.inv_pow), G1,
                                                ldp<PRESCALE>(c0, idx * LD.em, sc.nm1[0], sc.rnm1[0], sc.inv_pow) * G0));
                    const float wx = cx ? tx.w1 : tx.w0, wy = cy ? ty.w1 : ty.w0, wz = cz ? tz.w1 : tz.w0;
                    gix += (cx ? dot : -dot) * (wy * wz);
                    giy += (cy ? dot : -dot) * (wx * wz);
                    giz += (cz ? dot : -dot) * (wx * wy);
                }
        oc[own * LO.em] = (G0 + tx.gmul * gix) + a0;
        oc[LO.cs + own * LO.em] = (G1 + ty.gmul * giy) + a1;
        oc[2 * LO.cs + own * LO.em] = (G2 + tz.gmul * giz) + a2;
    }
}

#ifdef IRS_BWD_TRACE
// timing experiment (tools/bwd_phase_trace.py; build with tools/build_variant.sh bwdtrace -DIRS_BWD_TRACE): clock stamps of one wave
// of one workgroup at the phase boundaries of the marching loop.  For reading proportions, not for timing the kernel.
__device__ unsigned long long g_bwd_trace[8 * 64];
extern "C" __attribute__((visibility("default"))) int irs_debug_bwd_trace(unsigned long long* out) {  // (trace builds only: the library is compiled -fvisibility=hidden)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_trace), sizeof(g_bwd_trace)) == hipSuccess ? 0 : 1;
}
#define IRS_BT(slot)                                                                                  \
    do {                                                                                              \
        if (btrace_on && btrace_it < 64) g_bwd_trace[btrace_it * 8 + (slot)] = wall_clock64();        \
    } while (0)
#else
#define IRS_BT(slot)
#endif
template <bool PRESCALE, int R>
__device__ __forceinline__ void exp_bwd_march_tile(const float* __restrict__ G, const float* __restrict__ dk,
                                                   float* __restrict__ gout, const Vol vol, const Lin lin, const Scale3L sc,
                                                   const unsigned* __restrict__ dmax, const int seg_len, const int nseg,
                                                   const int r_lo, const int own_rest, const int swz_run, const int tile_id,
                                                   const dim3 tiles, const float* __restrict__ gscale, const int lay) {
    using M = March<PRESCALE, R>;
    constexpr int NP = M::NP, PX = M::PX, PN = M::PN, NIT = M::NIT;
    // ring slot layout (9 floats per source, 8-byte fields so that the gather needs three ds_read_b64 per candidate):
    //   q_xy = (rx, ry)   q_zg = (rz, G2)   q_g = (G0, G1)   q_d = (d0, d1)   q_dz = d2
    // where r = clipped sampling position - the source's own coordinate (|r| <= max|d|): every hat weight of the
    // gather is then a function of r plus a compile-time offset.  (Forming the three z weights of a source once, at commit
    // time, instead of in each of its nine gatherers was built and measured slower -- wider LDS records: DESIGN.md section 4.)
    // Radius 2: only d is needed across the NP planes of the ring (the corner taps of the own term); the (position, G) records
    // are read by the gather of the CURRENT plane alone -- and R steps later by the voxel's own term, which keeps its column's
    // record in registers meanwhile.  One slot of them instead of five: 36 instead of 78 KB, three workgroups per CU instead of two.
    constexpr int NPQ = R == 2 ? 1 : NP;
    __shared__ float2 q_xy[NPQ * PN], q_zg[NPQ * PN], q_g[NPQ * PN], q_d[NP * PN];
    __shared__ float q_dz[NP * PN];
    // XCD-aware tile assignment: consecutive tiles (x fastest, then y, then z-segment, then chain) stay on one L2
    const int tile_ = xcd_swizzle_runs(tile_id, (int)(tiles.x * tiles.y * tiles.z), swz_run);
    const int tbx = tile_ % tiles.x, tby = (tile_ / tiles.x) % tiles.y, tbz = tile_ / (tiles.x * tiles.y);
    const int chain = tbz / nseg, seg = tbz % nseg;
    const int hs = max(max((int)floorf(__uint_as_float(dmax[chain * 4 + 0])), (int)floorf(__uint_as_float(dmax[chain * 4 + 1]))),
                       (int)floorf(__uint_as_float(dmax[chain * 4 + 2]))) + 1;
    if (hs <= r_lo || (hs > R && !(R == 2 && own_rest))) return;  // another variant of this step owns the chain
    const int ox = tbx * MTX, oy = tby * MTY;
    int z0, z1;
    seg_range(vol, seg, seg_len, z0, z1);
    const int64_t V = vol.V;
    const int64_t cb = (int64_t)chain * 3 * V;
    auto run = [&](auto LC) {  // layouts as compile-time constants (see the forward tile), or (< 0) the run-time ones
    constexpr int LAYC = decltype(LC)::value;
    const int lay_ = LAYC < 0 ? lay : LAYC;
    const Lay3 LD = lay3(lay_ & 1, V), LG = lay3(lay_ & 2, V), LO = lay3(lay_ & 4, V);
    if (R == 2 && hs > R) {  // the any-radius kernel was not launched for this step and the bound outgrew the ring
        const int gx = ox + (int)(threadIdx.x % MTX), gy = oy + (int)(threadIdx.x / MTX);
        if (gx < vol.W && gy < vol.H)
            exp_bwd_generic_column<PRESCALE>(G + cb, LG, gscale ? gscale + (int64_t)chain * V : nullptr, dk + cb, LD, gout + cb, LO, vol,
                                             lin, sc, gx, gy, z0, z1, hs);
        return;
    }
    const float* __restrict__ dx_ = dk + cb;
    const float* __restrict__ dy_ = dx_ + LD.cs;
    const float* __restrict__ dz_ = dy_ + LD.cs;
    const float* __restrict__ Gx_ = G + cb;
    const float* __restrict__ Gy_ = Gx_ + LG.cs;
    const float* __restrict__ Gz_ = Gy_ + LG.cs;
    const float* __restrict__ gs_ = gscale ? gscale + (int64_t)chain * V : nullptr;  // fused backward warp: G *= gs
    float* __restrict__ o = gout + cb;

    const int lx = threadIdx.x % MTX, ly = threadIdx.x / MTX;
    const int x = ox + lx, y = oy + ly;
    const bool col_in = x < vol.W && y < vol.H;
    const float fx = (float)x, fy = (float)y;
    const float nxm = (float)(vol.W - 1), nym = (float)(vol.H - 1), nzm = (float)(vol.D - 1);

    int sxy[NIT];
    bool sin_[NIT];
    float slx[NIT], sly[NIT], sfx[NIT], sfy[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * kMarchBlock;
        const int px = i % PX, py = i / PX;
        const int ux = ox - R + px, uy = oy - R + py;
        sin_[it] = i < PN && (unsigned)ux < (unsigned)vol.W && (unsigned)uy < (unsigned)vol.H;
        const int cx = min(max(ux, 0), vol.W - 1), cy = min(max(uy, 0), vol.H - 1);
        sxy[it] = i < PN ? cy * vol.W + cx : -1;
        slx[it] = lin.x[cx];
        sly[it] = lin.y[cy];
        sfx[it] = (float)cx;
        sfy[it] = (float)cy;
    }
    float pre[NIT][6], pgs[NIT];
    auto prefetch = [&](int s) {
        if (s < 0 || s >= vol.D) return;
        const int64_t zo = (int64_t)s * vol.H * vol.W;
        const float* __restrict__ p0_ = dx_ + zo * LD.em;  // uniform plane bases + 32-bit lane offsets
        const float* __restrict__ p1_ = dy_ + zo * LD.em;
        const float* __restrict__ p2_ = dz_ + zo * LD.em;
        const float* __restrict__ p3_ = Gx_ + zo * LG.em;
        const float* __restrict__ p4_ = Gy_ + zo * LG.em;
        const float* __restrict__ p5_ = Gz_ + zo * LG.em;
        if (IRS_BWD_BUFLOAD && R == 1) {
        // the staging loads as BUFFER loads: descriptor of the plane in SGPRs, 32-bit lane offset -- no 64-bit VALU address arithmetic
        // and no 64-bit lane-offset pairs (the global_load form cost one v_lshl_add_u64 per load and six VGPR pairs: the address is
        // formed in the block in front of the layout branch, so instruction selection never saw base + offset at the load)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (sxy[it] < 0) continue;
            const unsigned g = (unsigned)sxy[it] * 4u, gd = g * (unsigned)LD.em, gg = g * (unsigned)LG.em;
            if (LD.em == 3) {
                const F3 v = buf_ld3(p0_, gd);
                pre[it][0] = v.x;
                pre[it][1] = v.y;
                pre[it][2] = v.z;
            } else {
                pre[it][0] = buf_ld1(p0_, gd);
                pre[it][1] = buf_ld1(p1_, gd);
                pre[it][2] = buf_ld1(p2_, gd);
            }
            if (LG.em == 3) {
                const F3 v = buf_ld3(p3_, gg);
                pre[it][3] = v.x;
                pre[it][4] = v.y;
                pre[it][5] = v.z;
            } else {
                pre[it][3] = buf_ld1(p3_, gg);
                pre[it][4] = buf_ld1(p4_, gg);
                pre[it][5] = buf_ld1(p5_, gg);
            }
            if (gs_) pgs[it] = buf_ld1(gs_ + zo, g);
        }
        return;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (sxy[it] < 0) continue;
            const unsigned g = (unsigned)sxy[it] * 4u, gd = g * (unsigned)LD.em, gg = g * (unsigned)LG.em;
            if (LD.em == 3) {
                const F3 v = ld3_off(p0_, gd);
                pre[it][0] = v.x;
                pre[it][1] = v.y;
                pre[it][2] = v.z;
            } else {
                pre[it][0] = ld_off(p0_, gd);
                pre[it][1] = ld_off(p1_, gd);
                pre[it][2] = ld_off(p2_, gd);
            }
            if (LG.em == 3) {
                const F3 v = ld3_off(p3_, gg);
                pre[it][3] = v.x;
                pre[it][4] = v.y;
                pre[it][5] = v.z;
            } else {
                pre[it][3] = ld_off(p3_, gg);
                pre[it][4] = ld_off(p4_, gg);
                pre[it][5] = ld_off(p5_, gg);
            }
            if (gs_) pgs[it] = ld_off(gs_ + zo, g);
        }
    };
    auto commit = [&](int s, int slot) {  // registers -> ring slot (with the clipped sampling position)
        const bool zin = s >= 0 && s < vol.D;
        const float lz_ = zin ? lin.z[s] : 0.0f, fs_ = (float)s;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (sxy[it] < 0) continue;
            const int i = slot * PN + threadIdx.x + it * kMarchBlock;
            const int iq = R == 2 ? (int)threadIdx.x + it * kMarchBlock : i;  // (radius 2: the single slot of the (position, G) records)
            if (!zin) {  // plane outside the volume: no source there (G = 0 removes it from every gather)
                q_xy[iq] = make_float2(0.0f, 0.0f);
                q_zg[iq] = make_float2(0.0f, 0.0f);
                q_g[iq] = make_float2(0.0f, 0.0f);
                q_d[i] = make_float2(0.0f, 0.0f);
                q_dz[i] = 0.0f;
                continue;
            }
            const float d0 = PRESCALE ? prescale(pre[it][0], sc.nm1[0], sc.rnm1[0], sc.inv_pow) : pre[it][0];
            const float d1 = PRESCALE ? prescale(pre[it][1], sc.nm1[1], sc.rnm1[1], sc.inv_pow) : pre[it][1];
            const float d2 = PRESCALE ? prescale(pre[it][2], sc.nm1[2], sc.rnm1[2], sc.inv_pow) : pre[it][2];
            const float qx = __fmul_rn(__fmul_rn(__fadd_rn(__fadd_rn(slx[it], d0), 1.0f), 0.5f), nxm);
            const float qy = __fmul_rn(__fmul_rn(__fadd_rn(__fadd_rn(sly[it], d1), 1.0f), 0.5f), nym);
            const float qz = __fmul_rn(__fmul_rn(__fadd_rn(__fadd_rn(lz_, d2), 1.0f), 0.5f), nzm);
            // clip to [0, n-1] (one v_med3).  A halo element outside the volume is a copy of the clamped voxel; it must not
            // act as a source, which G = 0 achieves whatever its position.
            const float p0 = __builtin_amdgcn_fmed3f(qx, 0.0f, nxm), p1 = __builtin_amdgcn_fmed3f(qy, 0.0f, nym),
                        p2 = __builtin_amdgcn_fmed3f(qz, 0.0f, nzm);
            const float gm_ = gs_ ? pgs[it] : 1.0f;
            const float g0 = sin_[it] ? pre[it][3] * gm_ : 0.0f, g1 = sin_[it] ? pre[it][4] * gm_ : 0.0f,
                        g2 = sin_[it] ? pre[it][5] * gm_ : 0.0f;
            q_xy[iq] = make_float2(p0 - sfx[it], p1 - sfy[it]);
            q_zg[iq] = make_float2(p2 - fs_, g2);
            q_g[iq] = make_float2(g0, g1);
            q_d[i] = make_float2(d0, d1);
            q_dz[i] = d2;
        }
    };

    float2 acc01[NP];  // channels 0, 1 of the NP output planes in flight (packed FMA)
    float acc2[NP];
#pragma unroll
    for (int a = 0; a < NP; ++a) {
        acc01[a] = make_float2(0.0f, 0.0f);
        acc2[a] = 0.0f;
    }

    // radius 2: this column's own (position, G) record of the planes s, s - 1, s - 2 (shifted along once per plane step)
    float2 own_xy[R == 2 ? R + 1 : 1], own_zg[R == 2 ? R + 1 : 1], own_g[R == 2 ? R + 1 : 1];
    if (R == 2) {
#pragma unroll
        for (int q = 0; q <= R; ++q) own_xy[q] = own_zg[q] = own_g[q] = make_float2(0.0f, 0.0f);
    }

    const int sbase = z0 - R;            // ring slot of plane s is (s - sbase) % NP
    const int slast = z1 - 1 + R;
#ifdef IRS_BWD_TRACE
    const bool btrace_on = R == 1 && !PRESCALE && tile_id == (int)(tiles.x * tiles.y * tiles.z) / 2 + 3 && threadIdx.x == 64;
    int btrace_it = 0;
#endif
    prefetch(sbase);
    for (int sb = sbase; sb <= slast; sb += NP) {
#pragma unroll
        for (int PH = 0; PH < NP; ++PH) {
            const int s = sb + PH;
            if (s > slast) break;
#ifdef IRS_BWD_TRACE
            IRS_BT(0);
            __builtin_amdgcn_s_waitcnt(0);  // the loads of plane s have arrived
            IRS_BT(1);
#endif
            commit(s, PH);
            IRS_BT(2);
            prefetch(s + 1);
            IRS_BT(3);
            __syncthreads();
            IRS_BT(4);
            if (R == 2) {  // the record of plane s enters the column's register queue (its own term is due R steps from now)
#pragma unroll
                for (int q = R; q > 0; --q) {
                    own_xy[q] = own_xy[q - 1];
                    own_zg[q] = own_zg[q - 1];
                    own_g[q] = own_g[q - 1];
                }
                const int co = (ly + R) * PX + (lx + R);
                own_xy[0] = q_xy[co];
                own_zg[0] = q_zg[co];
                own_g[0] = q_g[co];
            }
            // ---- contributions of source plane s to output planes s-R .. s+R
            // MK: which of those output planes lie inside the segment (bit oo + R).  The run-in steps of a segment reach one or
            // two of them only, and on a small volume or a thin slab the run-in is a quarter of all plane steps (8-plane
            // segments: 10 steps, 6 of their 30 accumulator updates land outside) -- those steps run an instantiation that leaves
            // the other accumulators alone (IRS_BWD_PEEL; same operations in the same order for every plane that is stored)
            auto gather = [&](auto MK) {
                constexpr int MASK = decltype(MK)::value;
#pragma unroll IRS_GATHER_UNROLL_Y
                for (int dy = 0; dy <= 2 * R; ++dy)
#pragma unroll
                    for (int dx = 0; dx <= 2 * R; ++dx) {
                        const int ri = (R == 2 ? 0 : PH * PN) + (ly + dy) * PX + (lx + dx);
                        const float2 rxy = q_xy[ri];
                        // weight of source (x + dx - R, y + dy - R, s) on output (x, y, s + oo): hat(r + offset) per axis.
                        // R == 1 guarantees |r| < 1 (variant selection by the displacement bound), where
                        // hat(r + 1) = max(0, -r) and hat(r - 1) = max(0, r): one clamped op each
                        const float hx = rel_hat<R>(rxy.x, dx - R), hy = rel_hat<R>(rxy.y, dy - R);
                        const float hxy = hx * hy;
                        // radius 2: a source reaches 2 of the 5 offsets per axis, so 21 of the 25 in-plane candidates of an output
                        // carry weight 0 -- and where the field is smooth they are the SAME 21 for every lane of the wavefront:
                        // skip the rest of such a candidate (two LDS reads, 5 weights, 15 multiply-adds) behind one vote
                        if (R == 2 && !__any(hxy != 0.0f)) continue;
                        const float2 rzg = q_zg[ri], g01 = q_g[ri];
#pragma unroll
                        for (int oo = -R; oo <= R; ++oo) {
                            if (!((MASK >> (oo + R)) & 1)) continue;
                            const int a = (PH + oo + NP) % NP;  // accumulator of output plane s + oo (static index)
                            const float w = hxy * rel_hat<R>(rzg.x, -oo);
                            acc01[a].x = fmaf(w, g01.x, acc01[a].x);
                            acc01[a].y = fmaf(w, g01.y, acc01[a].y);
                            acc2[a] = fmaf(w, rzg.y, acc2[a]);
                        }
                    }
            };
            if (s >= 0 && s < vol.D && col_in) {
                constexpr int ALL = (1 << NP) - 1;
                if (R == 1 && IRS_BWD_PEEL) {
                    const int m = (s - 1 >= z0 && s - 1 < z1 ? 1 : 0) | (s >= z0 && s < z1 ? 2 : 0) | (s + 1 >= z0 && s + 1 < z1 ? 4 : 0);  // wave-uniform
                    if (m == ALL) gather(LayC<ALL>{});
                    else if (m == 4) gather(LayC<4>{});
                    else if (m == 6) gather(LayC<6>{});
                    else if (m == 3) gather(LayC<3>{});
                    else if (m == 1) gather(LayC<1>{});
                    else gather(LayC<ALL>{});  // (segments of one or two planes: 2, 5 never occur, a lone 2 is rare)
                } else {
                    gather(LayC<ALL>{});
                }
            }
            IRS_BT(5);
            // ---- output plane zo = s - R is complete
            {
                const int zo = s - R;
                const int a = (PH - R + NP) % NP;
                if (zo >= z0 && zo < z1 && col_in) {
                    const int ci = a * PN + (ly + R) * PX + (lx + R);  // plane zo sits in slot (zo - sbase) % NP == a
                    // the sample this voxel took in the forward step: its clipped position is already in the ring (radius 2: in
                    // the register queue, R steps old)
                    constexpr int OQ = R == 2 ? R : 0;  // (oldest entry of the register queue; the arrays have one entry for R = 1)
                    const float2 pc = R == 2 ? own_xy[OQ] : q_xy[R == 2 ? 0 : ci], zg = R == 2 ? own_zg[OQ] : q_zg[R == 2 ? 0 : ci],
                                 Gc01 = R == 2 ? own_g[OQ] : q_g[R == 2 ? 0 : ci];
                    const float G0 = Gc01.x, G1 = Gc01.y, G2 = zg.y;
                    const float fx0 = floorf(pc.x), fy0 = floorf(pc.y), fz0 = floorf(zg.x);  // of the RELATIVE position
                    const float wx1 = __fsub_rn(pc.x, fx0), wx0 = __fsub_rn(__fadd_rn(fx0, 1.0f), pc.x);
                    const float wy1 = __fsub_rn(pc.y, fy0), wy0 = __fsub_rn(__fadd_rn(fy0, 1.0f), pc.y);
                    const float wz1 = __fsub_rn(zg.x, fz0), wz0 = __fsub_rn(__fadd_rn(fz0, 1.0f), zg.x);
                    const float pax = pc.x + fx, pay = pc.y + fy, paz = zg.x + (float)zo;  // absolute clipped position
                    const float gmx = pax > 0.0f && pax < nxm ? 0.5f * nxm : 0.0f;  // d(i)/d(g): 0 on / outside the border
                    const float gmy = pay > 0.0f && pay < nym ? 0.5f * nym : 0.0f;
                    const float gmz = paz > 0.0f && paz < nzm ? 0.5f * nzm : 0.0f;
                    const int rx0 = (int)fx0, ry0 = (int)fy0, rel = (int)fz0;
                    const int ix0 = x + rx0, iy0 = y + ry0, iz0 = zo + rel;
                    // the "+1" corners are read unconditionally: where ATen clamps them (i0 = n-1) their weight is exactly 0 and
                    // the ring holds a finite halo value there
                    const int bx0 = lx + R + rx0, by0 = ly + R + ry0;
                    const bool in_ring = (R == 1 && !IRS_BWD_R1_FALLBACK) ||
                                         ((unsigned)bx0 < (unsigned)(PX - 1) && (unsigned)by0 < (unsigned)(M::PY - 1) && rel >= -R && rel < R);
                    float dot[2][2][2];
                    auto ring_dots = [&]() {
                        int sl0 = ((a - R + NP) % NP) * PN, sl1 = ((a - R + 1 + NP) % NP) * PN;  // rel == -R
#pragma unroll
                        for (int q = -R + 1; q < R; ++q) {
                            sl0 = rel == q ? ((a + q + NP) % NP) * PN : sl0;
                            sl1 = rel == q ? ((a + q + 1 + NP) % NP) * PN : sl1;
                        }
                        const int off = by0 * PX + bx0;
#pragma unroll
                        for (int cz = 0; cz < 2; ++cz) {
                            const int bs = (cz ? sl1 : sl0) + off;
#pragma unroll
                            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                                for (int cx = 0; cx < 2; ++cx) {
#if IRS_BWD_TAPS == 1
                                    typedef float VF2 __attribute__((ext_vector_type(2)));
                                    const VF2 tv = *(const volatile __attribute__((address_space(3))) VF2*)(&q_d[bs + cy * PX + cx]);
                                    const float2 v01 = make_float2(tv.x, tv.y);
                                    const float vz = *(const volatile __attribute__((address_space(3))) float*)(&q_dz[bs + cy * PX + cx]);
#else
                                    const float2 v01 = q_d[bs + cy * PX + cx];
                                    const float vz = q_dz[bs + cy * PX + cx];
#endif
                                    dot[cz][cy][cx] = fmaf(vz, G2, fmaf(v01.y, G1, v01.x * G0));
                                }
                        }
                    };
                    // (the same path behind a wave-uniform `__all(in_ring)` branch, which took 7 % off the forward step, changes nothing
                    // here: 204-205 us either way -- by the time a wave reaches its own term the prefetch has long arrived)
                    if (in_ring) {
                        ring_dots();
                    } else {  // cannot happen while max|d| < R
#pragma unroll
                        for (int cz = 0; cz < 2; ++cz)
#pragma unroll
                            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                                for (int cx = 0; cx < 2; ++cx) {
                                    const int64_t idx = (((int64_t)min(iz0 + cz, vol.D - 1) * vol.H + min(iy0 + cy, vol.H - 1)) * vol.W + min(ix0 + cx, vol.W - 1)) * LD.em;
                                    dot[cz][cy][cx] = fmaf(ldp<PRESCALE>(dz_, idx, sc.nm1[2], sc.rnm1[2], sc.inv_pow), G2,
                                                           fmaf(ldp<PRESCALE>(dy_, idx, sc.nm1[1], sc.rnm1[1], sc.inv_pow), G1,
                                                                ldp<PRESCALE>(dx_, idx, sc.nm1[0], sc.rnm1[0], sc.inv_pow) * G0));
                                }
                    }
                    // J^T G of the trilinear interpolant: differences of the corner dots along one axis, bilinear in the others
                    const float wyz[2][2] = {{wy0 * wz0, wy1 * wz0}, {wy0 * wz1, wy1 * wz1}};  // [cz][cy]
                    const float wxz[2][2] = {{wx0 * wz0, wx1 * wz0}, {wx0 * wz1, wx1 * wz1}};  // [cz][cx]
                    const float wxy[2][2] = {{wx0 * wy0, wx1 * wy0}, {wx0 * wy1, wx1 * wy1}};  // [cy][cx]
                    float gix = 0.0f, giy = 0.0f, giz = 0.0f;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            gix = fmaf(wyz[u][w], dot[u][w][1] - dot[u][w][0], gix);
                            giy = fmaf(wxz[u][w], dot[u][1][w] - dot[u][0][w], giy);
                            giz = fmaf(wxy[u][w], dot[1][u][w] - dot[0][u][w], giz);
                        }
                    const int64_t pl = (int64_t)zo * vol.H * vol.W * LO.em;
                    const unsigned g = (unsigned)(y * vol.W + x) * 4u * (unsigned)LO.em;
                    const float o0 = (G0 + gmx * gix) + acc01[a].x, o1 = (G1 + gmy * giy) + acc01[a].y, o2 = (G2 + gmz * giz) + acc2[a];
                    if (LO.em == 3) {
                        st3_off(o + pl, g, o0, o1, o2);
                    } else {
                        st_off(o + pl, g, o0);
                        st_off(o + LO.cs + pl, g, o1);
                        st_off(o + 2 * LO.cs + pl, g, o2);
                    }
                }
                acc01[a] = make_float2(0.0f, 0.0f);
                acc2[a] = 0.0f;
            }
            IRS_BT(6);
            __syncthreads();  // the next commit overwrites the oldest ring slot
            IRS_BT(7);
#ifdef IRS_BWD_TRACE
            ++btrace_it;
#endif
        }
    }
    };  // run
#if IRS_BWD_LAYC
    switch (lay & 7) {  // the combinations the library produces (ctx.h: bwd_lay; 0 = the planar operator API)
        case 0: run(LayC<0>{}); break;
        case 2: run(LayC<2>{}); break;
        case 5: run(LayC<5>{}); break;
        case 7: run(LayC<7>{}); break;
        default: run(LayC<-1>{}); break;
    }
#else
    run(LayC<-1>{});
#endif
}

// One tile per workgroup for the common radius-1 variant (XCD-aware order); the rarely selected variants are launched on a
// small persistent grid that strides over the tiles, so that a launch whose variant is not selected costs ~2 us instead of
// the dispatch of thousands of workgroups that exit at once.
template <bool PRESCALE, int R>
__global__ __launch_bounds__(kMarchBlock, R == 1 ? IRS_MARCH_WAVES : IRS_MARCH_WAVES_R2) void exp_bwd_march_kernel(
    const float* __restrict__ G, const float* __restrict__ dk, float* __restrict__ gout, Vol vol, Lin lin, Scale3L sc,
    const unsigned* __restrict__ dmax, int seg_len, int nseg, int r_lo, int own_rest, int swz_run, dim3 tiles,
    const float* __restrict__ gscale, int lay) {
    const int total = (int)(tiles.x * tiles.y * tiles.z);
    for (int id = blockIdx.x; id < total; id += gridDim.x)
        exp_bwd_march_tile<PRESCALE, R>(G, dk, gout, vol, lin, sc, dmax, seg_len, nseg, r_lo, own_rest, swz_run, id, tiles, gscale, lay);
}


#ifndef IRS_BWD_MAX_SEG
#define IRS_BWD_MAX_SEG 64  // longest z-segment the resident-set rule may give the adjoint step (common.h: pick_seg_len_fit)
#endif
constexpr int kRareGrid = 256 * IRS_MARCH_WAVES_R2;  // persistent grid of the rarely selected radius-2 variant (what the chip holds at once)


void launch_exp_step_bwd_march(const float* G, const float* dk, float* gout, bool prescale_in, int no_steps, int C, Vol vol,
                               Lin lin, const unsigned* dmax, int max_radius, bool r2_owns_rest, const float* gscale, int lay,
                               hipEvent_t after_primary, hipStream_t st) {
    const int seg_env = global_knobs().march_seg;
    const int64_t per_layer = (int64_t)((vol.W + MTX - 1) / MTX) * ((vol.H + MTY - 1) / MTY) * C;
    int seg_len = pick_seg_len(vol.nz + vol.nzb, per_layer, 8, seg_env);
    if (global_knobs().seg_fit && seg_env <= 0) {
        static int cache = 0;
        const int64_t res = resident_blocks((const void*)exp_bwd_march_kernel<false, 1>, kMarchBlock, &cache);
        if (res > 0) seg_len = pick_seg_len_fit(vol.nz, vol.nzb, per_layer, 8, 2, res, 0, IRS_BWD_MAX_SEG);
    }
    const int nseg = vol_nseg(vol, seg_len);  // segments of both windows
    const dim3 tiles((vol.W + MTX - 1) / MTX, (vol.H + MTY - 1) / MTY, (unsigned)(nseg * C));
    const int total = (int)(tiles.x * tiles.y * tiles.z);
    const Scale3L sc = make_scale_l(vol, no_steps);
    const int swz_env = global_knobs().swz_run;
    const int swz_run = swz_env >= 0 ? swz_env : (int)tiles.x;  // 0/1: no remap; default: one x-row of tiles per XCD run
    if (global_knobs().launch_log) {
        static int cache_l = 0;
        log_launch("exp_bwd_march_kernel<R=1>", MTX, MTY, total, kMarchBlock, seg_len, 2, vol.nz + vol.nzb, C,
                   resident_blocks((const void*)exp_bwd_march_kernel<false, 1>, kMarchBlock, &cache_l));
    }
#define IRS_BWM(P, RR, LO, GRID, SEG, NSEG, TILES, TOTAL) hipLaunchKernelGGL((exp_bwd_march_kernel<P, RR>), dim3(GRID), dim3(kMarchBlock), 0, st, G, dk, gout, vol, lin, sc, dmax, SEG, NSEG, LO, r2_owns_rest ? 1 : 0, (GRID) == (TOTAL) ? swz_run : 0, TILES, gscale, lay)
    // the radius-1 kernel first (the one the roofline is quoted on: `after_primary` brackets exactly its launch), then the
    // rarely selected radius-2 variant on the small persistent grid
    if (prescale_in) IRS_BWM(true, 1, 0, total, seg_len, nseg, tiles, total); else IRS_BWM(false, 1, 0, total, seg_len, nseg, tiles, total);
    if (after_primary) (void)hipEventRecord(after_primary, st);
    if (max_radius >= 2) {
        // its own segments: the variant walks its tiles on a persistent grid of what the chip holds of IT (three workgroups per CU,
        // four run-in planes), and a whole number of rounds of that grid is not the radius-1 kernel's (256^3: 6 segments of 43 planes
        // = 1536 tiles, two rounds of 47 steps, instead of 2048 tiles of 36 in three)
        int seg2 = seg_len;
        if (global_knobs().seg_fit && seg_env <= 0)
            seg2 = pick_seg_len_fit(vol.nz, vol.nzb, per_layer, 8, 4, kRareGrid, 0, IRS_BWD_MAX_SEG);
        const int nseg2 = vol_nseg(vol, seg2);
        const dim3 tiles2(tiles.x, tiles.y, (unsigned)(nseg2 * C));
        const int total2 = (int)(tiles2.x * tiles2.y * tiles2.z);
        const int rare = total2 < kRareGrid ? total2 : kRareGrid;
        if (prescale_in) IRS_BWM(true, 2, 1, rare, seg2, nseg2, tiles2, total2); else IRS_BWM(false, 2, 1, rare, seg2, nseg2, tiles2, total2);
    }
#undef IRS_BWM
}

// ------------------------------------------------------------------------------------------------
// forward step, z-marching: same schedule as the adjoint above -- a 256-thread workgroup owns a 32x8 column over a
// z-segment, keeps a ring of 2R+1 planes of d (3 floats per voxel: 12 KB for R = 1) in LDS, prefetches plane s+1 while
// plane s-R is being interpolated, and reads its 24 taps from the ring (global memory only if a tap leaves the ring).
// ------------------------------------------------------------------------------------------------
// forward tile: 64 x 8 columns per 256-thread workgroup, FROWS = 2 output rows per thread.  The staging rate of a marching
// kernel depends on the LENGTH of the row segments a tile reads (tools/bw_probe.hip, bare load -> LDS -> store skeleton with
// a one-voxel halo: 32 x 8 -> 4.1 TB/s, 64 x 8 -> 4.8 TB/s, 128 x 4 -> 5.0 TB/s); 64 x 8 also has the smaller halo overhead
// (1.29x vs 1.33x), and two rows per thread halve the barriers per output.
#ifndef IRS_FTX
#define IRS_FTX 64
#define IRS_FTY 8
#define IRS_FROWS 2
#endif
#ifndef IRS_FWD_PITCH_ALIGN
#define IRS_FWD_PITCH_ALIGN 16
#endif
#ifndef IRS_FWD_WAVES
#define IRS_FWD_WAVES 4
#endif
#ifndef IRS_FWD_TAPS
// How the eight corner taps of a sample leave the LDS ring.  0: plain reads -- the compiler pairs the cx = 0 / 1 corners into
// ds_read2_b64 + ds_read2_b32.  1 (round 5): every corner its own ds_read_b64 + ds_read_b32, through VOLATILE LDS pointers (the
// load / store optimiser leaves volatile accesses unpaired; the compiler still counts and schedules them).  A ds_read2_b64 holds the
// LDS for 8 cycles where two ds_read_b64 take 4, and is banked per 16 lanes (MI355X_MICROARCH.md, LDS table): with per-lane cell
// shifts the isolated tap pattern costs 99 against 69 clocks per wave tap-set (tools/probes/lds_tap_probe.hip, white shifts; 71
// against 60 with runs of four lanes), and this kernel 95.8 against 104.5 us per launch at 256^3 on one box (profiles/r05_fwd_taps_ab.txt).
// Same values, same products, same order of the additions: chains bit-identical (tools/debug/chain_bits.py).
#define IRS_FWD_TAPS 1
#endif
// FROWS is a template parameter of the radius-1 kernel: two rows per thread (256 threads) where the launch fills the chip; ONE row
// per thread (512 threads, 8 waves per tile) on small volumes and thin slabs, where a launch has a workgroup or two per CU and
// twice the waves per tile hide more of a plane step's latency (128^3: 22.9 against 24.7 us per step; at 256^3 it loses, 142
// against 118 us)
constexpr int FTX = IRS_FTX, FTY = IRS_FTY, FROWS_BIG = IRS_FROWS;
template <int R, int FROWS = FROWS_BIG>
struct MarchF {
    static constexpr int kFwdBlock = FTX * FTY / FROWS;
    static constexpr int NP = 2 * R + 1, PX = FTX + 2 * R, PY = FTY + 2 * R, PN = PX * PY;
    static constexpr int NIT = (PN + kFwdBlock - 1) / kFwdBlock;
    // LDS row pitch: a multiple of 16 elements, so that rows and ring slots of the 8-byte (d0, d1) records start on the same
    // bank -- lanes of a wave whose taps fall into different rows / planes (any displacement field whose sign changes inside
    // a wave) then collide only through the x shift, not through the row they read
    static constexpr int PITCH = (IRS_FWD_PITCH_ALIGN > 1 && R == 1) ? (PX + IRS_FWD_PITCH_ALIGN - 1) / IRS_FWD_PITCH_ALIGN * IRS_FWD_PITCH_ALIGN : PX;  // the radius-2 ring stays below 64 KB
    static constexpr int PNP = PITCH * PY;
};
#ifdef IRS_FWD_TRACE
// timing experiment (tools/fwd_phase_trace.py; build with tools/build_variant.sh trace -DIRS_FWD_TRACE): clock64 stamps of ONE
// wave of one workgroup at six points of the marching loop.  The stamps serialise the wave's LDS / scalar-memory queue, so the
// build is for reading proportions, not for timing the kernel.
__device__ unsigned long long g_fwd_trace[8 * 64];
extern "C" __attribute__((visibility("default"))) int irs_debug_fwd_trace(unsigned long long* out) {  // (trace builds only: the library is compiled -fvisibility=hidden)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_trace), sizeof(g_fwd_trace)) == hipSuccess ? 0 : 1;
}
#define IRS_TR(slot)                                                                           \
    do {                                                                                       \
        if (trace_on && trace_it < 64) g_fwd_trace[trace_it * 8 + (slot)] = clock64();         \
    } while (0)
#else
#define IRS_TR(slot)
#endif
// PF: how many planes ahead of the one being committed the global loads run (1: the next plane is in flight while this one is
// sampled; 2: two planes -- short segments on small volumes, where a plane step is otherwise one load latency long)
template <bool PRESCALE, int R, int FROWS, int PF>
__device__ __forceinline__ void exp_fwd_march_tile(const float* __restrict__ din, float* __restrict__ dout, const Vol vol,
                                                   const Lin lin, const Scale3L sc, const unsigned* __restrict__ dmax_in,
                                                   unsigned* __restrict__ dmax_out, const int seg_len, const int nseg,
                                                   const int h_lo, const int h_hi, const int swz_run, const int tile_id,
                                                   const dim3 tiles, const int lay) {
    using M = MarchF<R, FROWS>;
    constexpr int PX = M::PX, PN = M::PN, NIT = M::NIT, PITCH = M::PITCH, PNP = M::PNP, kFwdBlock = M::kFwdBlock;
    // ring of 2R+2 slots: one more than a sample can reach, so that the commit of the next source plane never overwrites
    // a plane another wavefront is still sampling -> ONE barrier per plane instead of two
    // (IRS_FWD_REC16, A/B of round 5: 16-byte records (d0, d1, d2, -), one ds_read_b128 per tap -- the layout the tap probe ranks
    // first in isolation.  At 16 bytes per record only 2R+1 slots fit the 40 KB that four workgroups per CU allow, so the one-barrier
    // trick above goes: a second barrier per plane step.  Radius-1 kernel with one plane of prefetch only.)
    // (IRS_FWD_REC16 = 2: also the one-row variant of small launches -- two workgroups of 512 threads per CU either way, so it keeps
    // its spare slot and its single barrier at 51 KB)
    constexpr bool REC16 = R == 1 && ((IRS_FWD_REC16 >= 1 && PF == 1) || IRS_FWD_REC16 >= 2);
    constexpr int NS = REC16 && PF == 1 ? M::NP : M::NP + 1;
    __shared__ float2 r_xy[REC16 ? 1 : NS * PNP];  // (d0, d1): one ds_read_b64 per tap
    __shared__ float r_z[REC16 ? 1 : NS * PNP];    // d2
    __shared__ float4 r_q[REC16 ? NS * PNP : 1];
    __shared__ float red[3 * (kFwdBlock / kWave)];
    const int tile_ = xcd_swizzle_runs(tile_id, (int)(tiles.x * tiles.y * tiles.z), swz_run);
    const int tbx = tile_ % tiles.x, tby = (tile_ / tiles.x) % tiles.y, tbz = tile_ / (tiles.x * tiles.y);
    const int chain = tbz / nseg, seg = tbz % nseg;
    if (dmax_in) {
        const int need = max(max((int)ceilf(__uint_as_float(dmax_in[chain * 4 + 0])), (int)ceilf(__uint_as_float(dmax_in[chain * 4 + 1]))),
                             (int)ceilf(__uint_as_float(dmax_in[chain * 4 + 2])));
        if (need <= h_lo || need > h_hi) return;
    }
    const int ox = tbx * FTX, oy = tby * FTY;
    int z0, z1;
    seg_range(vol, seg, seg_len, z0, z1);
    const int64_t V = vol.V;
    const int64_t cb = (int64_t)chain * 3 * V;
    // The layouts are COMPILE-TIME constants inside `run`: with a run-time layout both forms of a prefetch load (one 12-byte load /
    // three 4-byte loads) target the same registers, the 12-byte one then lands in a scratch triple and is COPIED over -- behind
    // an s_waitcnt vmcnt(0) in the middle of the prefetch, which serialises it (seen in the ISA of rounds 1-3)
    auto run = [&](auto LC) {
    constexpr int LAYC = decltype(LC)::value;
    const Lay3 LD = lay3(LAYC & 1, V), LO = lay3(LAYC & 4, V);
    const float* __restrict__ dx_ = din + cb;
    const float* __restrict__ dy_ = dx_ + LD.cs;
    const float* __restrict__ dz_ = dy_ + LD.cs;
    float* __restrict__ o = dout + cb;
    const int lx = threadIdx.x % FTX, ly0 = threadIdx.x / FTX;  // outputs (lx, ly0 + j * FTY / FROWS), j < FROWS
    const int x = ox + lx;
    const float linx = x < vol.W ? lin.x[x] : 0.0f;

    int sxy[NIT], sld[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * kFwdBlock;
        const int px = i % PX, py = i / PX;
        const int cx = min(max(ox - R + px, 0), vol.W - 1), cy = min(max(oy - R + py, 0), vol.H - 1);
        sxy[it] = i < PN ? cy * vol.W + cx : -1;
        sld[it] = py * PITCH + px;
    }
    float pre[PF][NIT][3];
    auto prefetch = [&](int s, float (&pre)[NIT][3]) {
        const int sc_ = min(max(s, 0), vol.D - 1);  // planes outside the volume replicate the border plane
        const int64_t zo = (int64_t)sc_ * vol.H * vol.W * LD.em;
        const float* __restrict__ px_ = dx_ + zo;  // uniform plane bases + 32-bit lane offsets
        const float* __restrict__ py_ = dy_ + zo;
        const float* __restrict__ pz_ = dz_ + zo;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (sxy[it] < 0) continue;
            const unsigned g = (unsigned)sxy[it] * 4u * (unsigned)LD.em;
            if (IRS_FWD_BUFLOAD) {  // buffer loads: plane descriptor in SGPRs + 32-bit lane offset (see the adjoint's staging)
                if (LD.em == 3) {
                    const F3 v = buf_ld3(px_, g);
                    pre[it][0] = v.x;
                    pre[it][1] = v.y;
                    pre[it][2] = v.z;
                } else {
                    pre[it][0] = buf_ld1(px_, g);
                    pre[it][1] = buf_ld1(py_, g);
                    pre[it][2] = buf_ld1(pz_, g);
                }
            } else if (LD.em == 3) {
                const F3 v = ld3_off(px_, g);
                pre[it][0] = v.x;
                pre[it][1] = v.y;
                pre[it][2] = v.z;
            } else {
                pre[it][0] = ld_off(px_, g);
                pre[it][1] = ld_off(py_, g);
                pre[it][2] = ld_off(pz_, g);
            }
        }
    };
    auto commit = [&](int slot, const float (&pre)[NIT][3]) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (sxy[it] < 0) continue;
            const int i = slot * PNP + sld[it];
            if (REC16) {
                r_q[i] = make_float4(PRESCALE ? prescale(pre[it][0], sc.nm1[0], sc.rnm1[0], sc.inv_pow) : pre[it][0],
                                     PRESCALE ? prescale(pre[it][1], sc.nm1[1], sc.rnm1[1], sc.inv_pow) : pre[it][1],
                                     PRESCALE ? prescale(pre[it][2], sc.nm1[2], sc.rnm1[2], sc.inv_pow) : pre[it][2], 0.0f);
                continue;
            }
            r_xy[i] = make_float2(PRESCALE ? prescale(pre[it][0], sc.nm1[0], sc.rnm1[0], sc.inv_pow) : pre[it][0],
                                  PRESCALE ? prescale(pre[it][1], sc.nm1[1], sc.rnm1[1], sc.inv_pow) : pre[it][1]);
            r_z[i] = PRESCALE ? prescale(pre[it][2], sc.nm1[2], sc.rnm1[2], sc.inv_pow) : pre[it][2];
        }
    };

    float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
    const int sbase = z0 - R, slast = z1 - 1 + R;
    // Identity-grid coordinates.  The counter a wavefront waits on completes IN ORDER: a load issued after the prefetch of the
    // next plane cannot be waited for without waiting for that whole plane first -- which, at the top of the sampling phase, is
    // exactly the overlap the prefetch exists for.  So the y coordinates are read once per tile, and the z coordinate of the plane
    // sampled in step s + 1 is requested in step s BEFORE that step's prefetch.
    float liny_[FROWS];
#pragma unroll
    for (int j = 0; j < FROWS; ++j) liny_[j] = lin.y[min(oy + ly0 + j * (FTY / FROWS), vol.H - 1)];
    float linz_nx = lin.z[min(max(sbase - R, 0), vol.D - 1)];
#ifdef IRS_FWD_TRACE
    const bool trace_on = R == 1 && !PRESCALE && tile_id == (int)(tiles.x * tiles.y * tiles.z) / 2 + 3 && threadIdx.x == 64;
    int trace_it = 0;
    if (trace_on) {  // entry 63: tile set-up done (clock64, 100 MHz wall clock) ... loop left (both again): prologue and clock rate
        g_fwd_trace[63 * 8 + 0] = clock64();
        g_fwd_trace[63 * 8 + 1] = wall_clock64();
        g_fwd_trace[63 * 8 + 4] = (unsigned long long)(slast - sbase + 1);
    }
#endif
    static_assert(NS % PF == 0, "the buffer of plane s is (s - sbase) % PF == PH % PF");
    prefetch(sbase, pre[0]);
    if (PF > 1) prefetch(sbase + 1, pre[PF - 1]);
    for (int sb = sbase; sb <= slast; sb += NS) {
#pragma unroll
        for (int PH = 0; PH < NS; ++PH) {
            const int s = sb + PH;
            if (s > slast) break;
#ifdef IRS_FWD_TRACE
            IRS_TR(0);
            __builtin_amdgcn_s_waitcnt(0);  // the loads of plane s have arrived
            IRS_TR(1);
#endif
            // the commit needs the newest loads anyway; said unconditionally (its own waits sit behind per-lane guards) it also tells the
            // compiler that everything older -- the z coordinate requested a step ago -- has arrived, which it otherwise re-waits for
            // in the sampling phase with a count that covers most of the prefetch issued below
            if (PF == 1) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            commit(PH, pre[PH % PF]);
            IRS_TR(2);
            const float linz = linz_nx;
            linz_nx = lin.z[min(max(s + 1 - R, 0), vol.D - 1)];
            if (s + PF <= slast) prefetch(s + PF, pre[PH % PF]);
            IRS_TR(3);
            __syncthreads();
            IRS_TR(4);
            const int zo = s - R;
            if (zo >= z0 && zo < z1) {
#pragma unroll
                for (int j = 0; j < FROWS; ++j) {
                const int ly = ly0 + j * (FTY / FROWS), y = oy + ly;
                if (x >= vol.W || y >= vol.H) continue;
                const float liny = liny_[j];
                const int a = (PH - R + NS) % NS;  // slot of plane zo (compile-time)
                const int ci = a * PNP + (ly + R) * PITCH + (lx + R);
                float d0, d1, d2;
                if (REC16) {
                    typedef float VF4 __attribute__((ext_vector_type(4)));  // (volatile: a plain read of three components becomes a ds_read_b96, 8 LDS cycles)
                    const VF4 c4 = *(const volatile __attribute__((address_space(3))) VF4*)(&r_q[ci]);
                    d0 = c4.x;
                    d1 = c4.y;
                    d2 = c4.z;
                } else {
                    const float2 dxy = r_xy[ci];
                    d0 = dxy.x;
                    d1 = dxy.y;
                    d2 = r_z[ci];
                }
                const AxisTap tx = axis_tap(__fadd_rn(linx, d0), vol.W);
                const AxisTap ty = axis_tap(__fadd_rn(liny, d1), vol.H);
                const AxisTap tz = axis_tap(__fadd_rn(linz, d2), vol.D);
                // the "+1" corners are read unconditionally: where ATen clamps them (i0 = n-1) their weight is exactly 0 and the
                // ring holds a finite (replicated) value there, so the product vanishes exactly
                const int bx0 = tx.i0 - (ox - R), by0 = ty.i0 - (oy - R), rel = tz.i0 - zo;
                const bool in_ring = (unsigned)bx0 < (unsigned)(PX - 1) && (unsigned)by0 < (unsigned)(M::PY - 1) && rel >= -R && rel < R;
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
                auto ring_taps = [&]() {
                    // (wx * wy) is shared by the two z corners: same products, same rounding as ((wx * wy) * wz)
                    const float wxy[2][2] = {{__fmul_rn(tx.w0, ty.w0), __fmul_rn(tx.w1, ty.w0)},
                                             {__fmul_rn(tx.w0, ty.w1), __fmul_rn(tx.w1, ty.w1)}};
                    // plane zo + q sits in ring slot (a + q) mod NS; with a compile-time `a` the slots are constants picked by `rel`
                    int sl0 = ((a - R + NS) % NS) * PNP, sl1 = ((a - R + 1 + NS) % NS) * PNP;
#pragma unroll
                    for (int q = -R + 1; q < R; ++q) {
                        sl0 = rel == q ? ((a + q + NS) % NS) * PNP : sl0;
                        sl1 = rel == q ? ((a + q + 1 + NS) % NS) * PNP : sl1;
                    }
                    const int off = by0 * PITCH + bx0;
#pragma unroll
                    for (int cz = 0; cz < 2; ++cz) {
                        const int bs = (cz ? sl1 : sl0) + off;
#pragma unroll
                        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                            for (int cx = 0; cx < 2; ++cx) {
                                const float w = __fmul_rn(wxy[cy][cx], cz ? tz.w1 : tz.w0);
                                float2 t2;
                                float t1;
                                if (REC16) {
                                    typedef float VF4 __attribute__((ext_vector_type(4)));
                                    const VF4 tv = *(const volatile __attribute__((address_space(3))) VF4*)(&r_q[bs + cy * PITCH + cx]);
                                    t2 = make_float2(tv.x, tv.y);
                                    t1 = tv.z;
                                } else {
#if IRS_FWD_TAPS == 1
                                    typedef float VF2 __attribute__((ext_vector_type(2)));
                                    const VF2 tv = *(const volatile __attribute__((address_space(3))) VF2*)(&r_xy[bs + cy * PITCH + cx]);
                                    t2 = make_float2(tv.x, tv.y);
                                    t1 = *(const volatile __attribute__((address_space(3))) float*)(&r_z[bs + cy * PITCH + cx]);
#else
                                    t2 = r_xy[bs + cy * PITCH + cx];
                                    t1 = r_z[bs + cy * PITCH + cx];
#endif
                                }
#if IRS_FWD_FMA
                                a0 = fmaf(t2.x, w, a0);
                                a1 = fmaf(t2.y, w, a1);
                                a2 = fmaf(t1, w, a2);
#else
                                a0 = __fadd_rn(a0, __fmul_rn(t2.x, w));
                                a1 = __fadd_rn(a1, __fmul_rn(t2.y, w));
                                a2 = __fadd_rn(a2, __fmul_rn(t1, w));
#endif
                            }
                    }
                };
                // the usual case -- every tap of every lane inside the ring -- behind a WAVE-UNIFORM branch: in a divergent
                // if / else the global-memory path is laid out first, and the registers its loads target make the ring path wait
                // for "their" loads, i.e. (in-order counter) for the prefetch of the next plane
                if (__all(in_ring)) {
                    ring_taps();
                } else if (in_ring) {
                    ring_taps();
                } else {
#pragma unroll
                    for (int cz = 0; cz < 2; ++cz)
#pragma unroll
                        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                            for (int cx = 0; cx < 2; ++cx) {
                                const float w = __fmul_rn(__fmul_rn(cx ? tx.w1 : tx.w0, cy ? ty.w1 : ty.w0), cz ? tz.w1 : tz.w0);
                                const int64_t idx = (((int64_t)(cz ? tz.i1 : tz.i0) * vol.H + (cy ? ty.i1 : ty.i0)) * vol.W + (cx ? tx.i1 : tx.i0)) * LD.em;
#if IRS_FWD_FMA
                                a0 = fmaf(ldp<PRESCALE>(dx_, idx, sc.nm1[0], sc.rnm1[0], sc.inv_pow), w, a0);
                                a1 = fmaf(ldp<PRESCALE>(dy_, idx, sc.nm1[1], sc.rnm1[1], sc.inv_pow), w, a1);
                                a2 = fmaf(ldp<PRESCALE>(dz_, idx, sc.nm1[2], sc.rnm1[2], sc.inv_pow), w, a2);
#else
                                a0 = __fadd_rn(a0, __fmul_rn(ldp<PRESCALE>(dx_, idx, sc.nm1[0], sc.rnm1[0], sc.inv_pow), w));
                                a1 = __fadd_rn(a1, __fmul_rn(ldp<PRESCALE>(dy_, idx, sc.nm1[1], sc.rnm1[1], sc.inv_pow), w));
                                a2 = __fadd_rn(a2, __fmul_rn(ldp<PRESCALE>(dz_, idx, sc.nm1[2], sc.rnm1[2], sc.inv_pow), w));
#endif
                            }
                }
                const int64_t pl = (int64_t)zo * vol.H * vol.W * LO.em;
                const unsigned g = (unsigned)(y * vol.W + x) * 4u * (unsigned)LO.em;
                const float r0 = __fadd_rn(d0, a0), r1 = __fadd_rn(d1, a1), r2 = __fadd_rn(d2, a2);
                if (LO.em == 3) {
                    st3_off(o + pl, g, r0, r1, r2);
                } else {
                    st_off(o + pl, g, r0);
                    st_off(o + LO.cs + pl, g, r1);
                    st_off(o + 2 * LO.cs + pl, g, r2);
                }
                m0 = fmaxf(m0, fabsf(r0));
                m1 = fmaxf(m1, fabsf(r1));
                m2 = fmaxf(m2, fabsf(r2));
                }
            }
#ifdef IRS_FWD_TRACE
            IRS_TR(5);
            ++trace_it;
#endif
            if (NS == M::NP) __syncthreads();  // (a ring without the spare slot: the next commit overwrites the oldest plane of this step)
        }
    }
#ifdef IRS_FWD_TRACE
    if (trace_on) {
        g_fwd_trace[63 * 8 + 2] = clock64();
        g_fwd_trace[63 * 8 + 3] = wall_clock64();
    }
#endif
    if (dmax_out) {
        m0 *= 0.5f * sc.nm1[0];
        m1 *= 0.5f * sc.nm1[1];
        m2 *= 0.5f * sc.nm1[2];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            m0 = fmaxf(m0, __shfl_down(m0, off, kWave));
            m1 = fmaxf(m1, __shfl_down(m1, off, kWave));
            m2 = fmaxf(m2, __shfl_down(m2, off, kWave));
        }
        const int wid = threadIdx.x / kWave;
        if ((threadIdx.x & (kWave - 1)) == 0) {
            red[wid] = m0;
            red[(kFwdBlock / kWave) + wid] = m1;
            red[2 * (kFwdBlock / kWave) + wid] = m2;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            float m = 0.0f;
#pragma unroll
            for (int w = 0; w < kFwdBlock / kWave; ++w) m = fmaxf(m, red[threadIdx.x * (kFwdBlock / kWave) + w]);
            unsigned* slot = dmax_out + chain * 4 + threadIdx.x;
            if (__float_as_uint(m) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_nonneg(slot, m);
        }
    }
    };  // run
    switch (lay & 5) {
        case 0: run(LayC<0>{}); break;
        case 1: run(LayC<1>{}); break;
        case 4: run(LayC<4>{}); break;
        default: run(LayC<5>{}); break;
    }
}

#ifndef IRS_FWD_SMALL_TILES
#define IRS_FWD_SMALL_TILES 1024  // launches of at most this many tiles (power-of-two segment rule) take the one-row-per-thread kernel
#endif
template <bool PRESCALE, int R, int FROWS = FROWS_BIG, int PF = 1>
__global__ __launch_bounds__(FTX * FTY / FROWS, R == 1 ? IRS_FWD_WAVES : 1) void exp_fwd_march_kernel(const float* __restrict__ din, float* __restrict__ dout,
                                                                  Vol vol, Lin lin, Scale3L sc,
                                                                  const unsigned* __restrict__ dmax_in,
                                                                  unsigned* __restrict__ dmax_out, int seg_len, int nseg,
                                                                  int h_lo, int h_hi, int swz_run, dim3 tiles, int lay) {
    const int total = (int)(tiles.x * tiles.y * tiles.z);
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
        exp_fwd_march_tile<PRESCALE, R, FROWS, PF>(din, dout, vol, lin, sc, dmax_in, dmax_out, seg_len, nseg, h_lo, h_hi, swz_run, id, tiles, lay);
        __syncthreads();  // the ring and the reduction scratch are reused by the next tile
    }
}

// ------------------------------------------------------------------------------------------------
// forward step of SMALL launches, TWO planes per marching step (round 5, late).  The one-row-per-thread kernel above pays its
// per-step fixed work -- barrier, loop, the wait for the newest loads, the z coordinate -- once per plane (0.5 of the 1.9 us of a
// plane step at 128^3, profiles/r05_phase_trace_128.txt); here a step commits planes s and s + 1, passes ONE barrier and samples
// the output planes s - 1 and s.  Radius 1, one row per thread (512 threads), ring of SIX slots (the planes s - 2 .. s + 1 being
// sampled and the two the next step commits before its barrier: 58 KB, two workgroups per CU as before), the loads of the next
// step's two planes in flight during the sampling.  The arithmetic of an output -- taps, weights, order of the FMAs -- is the
// one-plane kernel's: chains bit-identical (tools/debug/chain_bits.py).
// MEASURED (profiles/r05_fwd_z2_ab.txt): 0.835-0.839 against 0.842-0.845 ms per transition at 128^3 (-0.7 %), 0.678-0.680 against
// 0.680-0.681 with two chains, flat at 96^3 / 64^3 and on slab ranks of 4 / 8 -- the per-step fixed work is not what the small
// forward step loses to; what remains per row is the staging (660 halo elements on 512 threads: two passes, the second 29 % full)
// and the sampling itself.  Below the round's 1 % line: built, kept behind `fwd_z2`, OFF.
// (A second form staged the halo planes of both source planes as ONE list -- three passes of the 512 threads, 86 % of the lanes,
// instead of four -- : digests equal again, 0.836-0.838 against 0.835-0.838 ms, no gain at all; not kept.)
// ------------------------------------------------------------------------------------------------
template <bool PRESCALE>
__device__ __forceinline__ void exp_fwd_march_tile_z2(const float* __restrict__ din, float* __restrict__ dout, const Vol vol,
                                                      const Lin lin, const Scale3L sc, const unsigned* __restrict__ dmax_in,
                                                      unsigned* __restrict__ dmax_out, const int seg_len, const int nseg,
                                                      const int h_lo, const int h_hi, const int swz_run, const int tile_id,
                                                      const dim3 tiles, const int lay) {
    constexpr int R = 1;
    using M = MarchF<R, 1>;
    constexpr int PX = M::PX, PN = M::PN, NIT = M::NIT, PITCH = M::PITCH, PNP = M::PNP, kFwdBlock = M::kFwdBlock;
    constexpr int NS = 6;
    __shared__ float2 r_xy[NS * PNP];  // (d0, d1)
    __shared__ float r_z[NS * PNP];    // d2
    __shared__ float red[3 * (kFwdBlock / kWave)];
    const int tile_ = xcd_swizzle_runs(tile_id, (int)(tiles.x * tiles.y * tiles.z), swz_run);
    const int tbx = tile_ % tiles.x, tby = (tile_ / tiles.x) % tiles.y, tbz = tile_ / (tiles.x * tiles.y);
    const int chain = tbz / nseg, seg = tbz % nseg;
    if (dmax_in) {
        const int need = max(max((int)ceilf(__uint_as_float(dmax_in[chain * 4 + 0])), (int)ceilf(__uint_as_float(dmax_in[chain * 4 + 1]))),
                             (int)ceilf(__uint_as_float(dmax_in[chain * 4 + 2])));
        if (need <= h_lo || need > h_hi) return;
    }
    const int ox = tbx * FTX, oy = tby * FTY;
    int z0, z1;
    seg_range(vol, seg, seg_len, z0, z1);
    const int64_t V = vol.V;
    const int64_t cb = (int64_t)chain * 3 * V;
    auto run = [&](auto LC) {  // layouts as compile-time constants (see exp_fwd_march_tile)
    constexpr int LAYC = decltype(LC)::value;
    const Lay3 LD = lay3(LAYC & 1, V), LO = lay3(LAYC & 4, V);
    const float* __restrict__ dx_ = din + cb;
    const float* __restrict__ dy_ = dx_ + LD.cs;
    const float* __restrict__ dz_ = dy_ + LD.cs;
    float* __restrict__ o = dout + cb;
    const int lx = threadIdx.x % FTX, ly = threadIdx.x / FTX;
    const int x = ox + lx, y = oy + ly;
    const bool live = x < vol.W && y < vol.H;
    const float linx = x < vol.W ? lin.x[x] : 0.0f;
    const float liny = lin.y[min(y, vol.H - 1)];

    int sxy[NIT], sld[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * kFwdBlock;
        const int px = i % PX, py = i / PX;
        const int cx = min(max(ox - R + px, 0), vol.W - 1), cy = min(max(oy - R + py, 0), vol.H - 1);
        sxy[it] = i < PN ? cy * vol.W + cx : -1;
        sld[it] = py * PITCH + px;
    }
    float pre[2][NIT][3];
    auto prefetch = [&](int s, float (&pre)[NIT][3]) {
        const int sc_ = min(max(s, 0), vol.D - 1);  // planes outside the volume replicate the border plane
        const int64_t zo = (int64_t)sc_ * vol.H * vol.W * LD.em;
        const float* __restrict__ px_ = dx_ + zo;
        const float* __restrict__ py_ = dy_ + zo;
        const float* __restrict__ pz_ = dz_ + zo;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (sxy[it] < 0) continue;
            const unsigned g = (unsigned)sxy[it] * 4u * (unsigned)LD.em;
            if (LD.em == 3) {
                const F3 v = ld3_off(px_, g);
                pre[it][0] = v.x;
                pre[it][1] = v.y;
                pre[it][2] = v.z;
            } else {
                pre[it][0] = ld_off(px_, g);
                pre[it][1] = ld_off(py_, g);
                pre[it][2] = ld_off(pz_, g);
            }
        }
    };
    auto commit = [&](int slot, const float (&pre)[NIT][3]) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (sxy[it] < 0) continue;
            const int i = slot * PNP + sld[it];
            r_xy[i] = make_float2(PRESCALE ? prescale(pre[it][0], sc.nm1[0], sc.rnm1[0], sc.inv_pow) : pre[it][0],
                                  PRESCALE ? prescale(pre[it][1], sc.nm1[1], sc.rnm1[1], sc.inv_pow) : pre[it][1]);
            r_z[i] = PRESCALE ? prescale(pre[it][2], sc.nm1[2], sc.rnm1[2], sc.inv_pow) : pre[it][2];
        }
    };

    float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
    // one output of plane zo, whose own value sits in ring slot `a` (a compile-time constant at every call site)
    auto sample = [&](const int zo, const int a, const float linz) {
        const int ci = a * PNP + (ly + R) * PITCH + (lx + R);
        const float2 dxy = r_xy[ci];
        const float d0 = dxy.x, d1 = dxy.y, d2 = r_z[ci];
        const AxisTap tx = axis_tap(__fadd_rn(linx, d0), vol.W);
        const AxisTap ty = axis_tap(__fadd_rn(liny, d1), vol.H);
        const AxisTap tz = axis_tap(__fadd_rn(linz, d2), vol.D);
        const int bx0 = tx.i0 - (ox - R), by0 = ty.i0 - (oy - R), rel = tz.i0 - zo;
        const bool in_ring = (unsigned)bx0 < (unsigned)(PX - 1) && (unsigned)by0 < (unsigned)(M::PY - 1) && rel >= -R && rel < R;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
        auto ring_taps = [&]() {
            const float wxy[2][2] = {{__fmul_rn(tx.w0, ty.w0), __fmul_rn(tx.w1, ty.w0)}, {__fmul_rn(tx.w0, ty.w1), __fmul_rn(tx.w1, ty.w1)}};
            // rel is -1 or 0: planes zo - 1, zo or zo, zo + 1
            const int sl0 = (rel == 0 ? a : (a - 1 + NS) % NS) * PNP, sl1 = (rel == 0 ? (a + 1) % NS : a) * PNP;
            const int off = by0 * PITCH + bx0;
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                const int bs = (cz ? sl1 : sl0) + off;
#pragma unroll
                for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                    for (int cx = 0; cx < 2; ++cx) {
                        const float w = __fmul_rn(wxy[cy][cx], cz ? tz.w1 : tz.w0);
                        typedef float VF2 __attribute__((ext_vector_type(2)));  // single reads: see IRS_FWD_TAPS
                        const VF2 tv = *(const volatile __attribute__((address_space(3))) VF2*)(&r_xy[bs + cy * PITCH + cx]);
                        const float t1 = *(const volatile __attribute__((address_space(3))) float*)(&r_z[bs + cy * PITCH + cx]);
                        a0 = fmaf(tv.x, w, a0);
                        a1 = fmaf(tv.y, w, a1);
                        a2 = fmaf(t1, w, a2);
                    }
            }
        };
        if (__all(in_ring)) {
            ring_taps();
        } else if (in_ring) {
            ring_taps();
        } else {
#pragma unroll
            for (int cz = 0; cz < 2; ++cz)
#pragma unroll
                for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                    for (int cx = 0; cx < 2; ++cx) {
                        const float w = __fmul_rn(__fmul_rn(cx ? tx.w1 : tx.w0, cy ? ty.w1 : ty.w0), cz ? tz.w1 : tz.w0);
                        const int64_t idx = (((int64_t)(cz ? tz.i1 : tz.i0) * vol.H + (cy ? ty.i1 : ty.i0)) * vol.W + (cx ? tx.i1 : tx.i0)) * LD.em;
                        a0 = fmaf(ldp<PRESCALE>(dx_, idx, sc.nm1[0], sc.rnm1[0], sc.inv_pow), w, a0);
                        a1 = fmaf(ldp<PRESCALE>(dy_, idx, sc.nm1[1], sc.rnm1[1], sc.inv_pow), w, a1);
                        a2 = fmaf(ldp<PRESCALE>(dz_, idx, sc.nm1[2], sc.rnm1[2], sc.inv_pow), w, a2);
                    }
        }
        const int64_t pl = (int64_t)zo * vol.H * vol.W * LO.em;
        const unsigned g = (unsigned)(y * vol.W + x) * 4u * (unsigned)LO.em;
        const float r0 = __fadd_rn(d0, a0), r1 = __fadd_rn(d1, a1), r2 = __fadd_rn(d2, a2);
        if (LO.em == 3) {
            st3_off(o + pl, g, r0, r1, r2);
        } else {
            st_off(o + pl, g, r0);
            st_off(o + LO.cs + pl, g, r1);
            st_off(o + 2 * LO.cs + pl, g, r2);
        }
        m0 = fmaxf(m0, fabsf(r0));
        m1 = fmaxf(m1, fabsf(r1));
        m2 = fmaxf(m2, fabsf(r2));
    };

    // planes sbase .. slast are staged, two per step: step i commits sbase + 2 i and sbase + 2 i + 1 (a plane beyond slast is a clamped,
    // harmless re-read that nothing samples) and samples the outputs one plane behind each
    const int sbase = z0 - R, slast = z1 - 1 + R;
    float linz_a = lin.z[min(max(sbase - R, 0), vol.D - 1)], linz_b = lin.z[min(max(sbase + 1 - R, 0), vol.D - 1)];
    prefetch(sbase, pre[0]);
    prefetch(sbase + 1, pre[1]);
    for (int sb = sbase; sb <= slast; sb += NS) {
#pragma unroll
        for (int PH = 0; PH < NS; PH += 2) {
            const int s = sb + PH;
            if (s > slast) break;
            commit(PH, pre[0]);
            commit(PH + 1, pre[1]);
            const float lz0 = linz_a, lz1 = linz_b;
            linz_a = lin.z[min(max(s + 2 - R, 0), vol.D - 1)];  // (before the prefetch: the load counter completes in order)
            linz_b = lin.z[min(max(s + 3 - R, 0), vol.D - 1)];
            if (s + 2 <= slast) {
                prefetch(s + 2, pre[0]);
                prefetch(s + 3, pre[1]);
            }
            __syncthreads();
            const int zo = s - R;
            if (live && zo >= z0 && zo < z1) sample(zo, (PH - R + NS) % NS, lz0);
            if (live && zo + 1 >= z0 && zo + 1 < z1) sample(zo + 1, PH, lz1);
        }
    }
    if (dmax_out) {
        m0 *= 0.5f * sc.nm1[0];
        m1 *= 0.5f * sc.nm1[1];
        m2 *= 0.5f * sc.nm1[2];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            m0 = fmaxf(m0, __shfl_down(m0, off, kWave));
            m1 = fmaxf(m1, __shfl_down(m1, off, kWave));
            m2 = fmaxf(m2, __shfl_down(m2, off, kWave));
        }
        const int wid = threadIdx.x / kWave;
        if ((threadIdx.x & (kWave - 1)) == 0) {
            red[wid] = m0;
            red[(kFwdBlock / kWave) + wid] = m1;
            red[2 * (kFwdBlock / kWave) + wid] = m2;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            float m = 0.0f;
#pragma unroll
            for (int w = 0; w < kFwdBlock / kWave; ++w) m = fmaxf(m, red[threadIdx.x * (kFwdBlock / kWave) + w]);
            unsigned* slot = dmax_out + chain * 4 + threadIdx.x;
            if (__float_as_uint(m) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_nonneg(slot, m);
        }
    }
    };  // run
    switch (lay & 5) {
        case 0: run(LayC<0>{}); break;
        case 1: run(LayC<1>{}); break;
        case 4: run(LayC<4>{}); break;
        default: run(LayC<5>{}); break;
    }
}

template <bool PRESCALE>
__global__ __launch_bounds__(FTX * FTY, IRS_FWD_WAVES) void exp_fwd_march_z2_kernel(const float* __restrict__ din, float* __restrict__ dout, Vol vol, Lin lin,
                                                                        Scale3L sc, const unsigned* __restrict__ dmax_in,
                                                                        unsigned* __restrict__ dmax_out, int seg_len, int nseg, int h_lo,
                                                                        int h_hi, int swz_run, dim3 tiles, int lay) {
    const int total = (int)(tiles.x * tiles.y * tiles.z);
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
        exp_fwd_march_tile_z2<PRESCALE>(din, dout, vol, lin, sc, dmax_in, dmax_out, seg_len, nseg, h_lo, h_hi, swz_run, id, tiles, lay);
        __syncthreads();  // the ring and the reduction scratch are reused by the next tile
    }
}

void launch_exp_step_fwd_march(const float* din, float* dout, bool prescale_in, int no_steps, int C, Vol vol, Lin lin,
                               const unsigned* dmax_in, unsigned* dmax_out, bool only_r1, int lay, hipStream_t st) {
    const int seg_env = global_knobs().march_seg_fwd;
    const int64_t per_layer = (int64_t)((vol.W + FTX - 1) / FTX) * ((vol.H + FTY - 1) / FTY) * C;
    int seg_len = pick_seg_len(vol.nz + vol.nzb, per_layer, 8, seg_env);
    // small launches (the power-of-two rule is down at its shortest segments and still has at most four workgroups of tiles per CU):
    // the radius-1 kernel with one output row per thread -- 512 threads per tile, so a resident set is 512 tiles and the fitted
    // segments come out twice as long (two chains at 128^3: 512 workgroups of 18 plane steps instead of 1024 of 10, the run-in a
    // ninth instead of a fifth of the march: 0.699 -> 0.686 ms per chain-transition; up to 640 tiles until round 5).  Not beyond
    // 8-plane segments: four chains at 128^3 are 1024 tiles of 16 planes, and 512 workgroups of 34 steps are SLOWER there than 1024
    // of 18 (0.632 against 0.621) -- profiles/r05_fwd_small_tiles_ab.txt
    const int small_env = global_knobs().fwd_rows1;
    const bool small = FROWS_BIG != 1 && (small_env >= 0 ? small_env != 0 : seg_len <= 8 && per_layer * vol_nseg(vol, seg_len) <= IRS_FWD_SMALL_TILES);
    if (global_knobs().seg_fit && seg_env <= 0) {
        static int cache_big = 0, cache_small = 0;
        const int64_t res = small ? resident_blocks((const void*)exp_fwd_march_kernel<false, 1, 1, 2>, FTX * FTY, &cache_small)
                                  : resident_blocks((const void*)exp_fwd_march_kernel<false, 1, FROWS_BIG, 1>, FTX * FTY / FROWS_BIG, &cache_big);
        if (res > 0) seg_len = pick_seg_len_fit(vol.nz, vol.nzb, per_layer, 8, 2, res, 0);
    }
    const int nseg = vol_nseg(vol, seg_len);  // segments of both windows
    const dim3 tiles((vol.W + FTX - 1) / FTX, (vol.H + FTY - 1) / FTY, (unsigned)(nseg * C));
    const int total = (int)(tiles.x * tiles.y * tiles.z);
    const Scale3L sc = make_scale_l(vol, no_steps);
    const int swz_env = global_knobs().swz_run;
    const int swz_run = swz_env >= 0 ? swz_env : (int)tiles.x;
    if (global_knobs().launch_log) {
        static int cache_ls = 0, cache_lb = 0;
        log_launch(small ? "exp_fwd_march_kernel<R=1,rows=1>" : "exp_fwd_march_kernel<R=1,rows=2>", FTX, FTY, total, small ? FTX * FTY : FTX * FTY / FROWS_BIG,
                   seg_len, 2, vol.nz + vol.nzb, C,
                   small ? resident_blocks((const void*)exp_fwd_march_kernel<false, 1, 1, 2>, FTX * FTY, &cache_ls)
                         : resident_blocks((const void*)exp_fwd_march_kernel<false, 1, FROWS_BIG, 1>, FTX * FTY / FROWS_BIG, &cache_lb));
    }
#define IRS_FWM(P, RR, LO, HI, GRID) hipLaunchKernelGGL((exp_fwd_march_kernel<P, RR>), dim3(GRID), dim3(FTX * FTY / FROWS_BIG), 0, st, din, dout, vol, lin, sc, dmax_in, dmax_out, seg_len, nseg, LO, HI, (GRID) == total ? swz_run : 0, tiles, lay)
#define IRS_FWM2(P, LO, HI, GRID) hipLaunchKernelGGL((exp_fwd_march_kernel<P, 2>), dim3(GRID), dim3(FTX * FTY / FROWS_BIG), 0, st, din, dout, vol, lin, sc, dmax_in, dmax_out, seg2, nseg2, LO, HI, (GRID) == total2 ? swz_run : 0, tiles2, lay)
#define IRS_FW2(P, LO, HI, GRID) hipLaunchKernelGGL((exp_fwd_march_kernel<P, 2, 1>), dim3(GRID), dim3(FTX * FTY), 0, st, din, dout, vol, lin, sc, dmax_in, dmax_out, seg2, nseg2, LO, HI, (GRID) == total2 ? swz_run : 0, tiles2, lay)
#define IRS_FWS(P, LO, HI)                                                                                                      \
    if (global_knobs().fwd_z2) hipLaunchKernelGGL((exp_fwd_march_z2_kernel<P>), dim3(total), dim3(FTX * FTY), 0, st, din, dout, vol, lin, sc, dmax_in, dmax_out, seg_len, nseg, LO, HI, swz_run, tiles, lay); \
    else if (global_knobs().fwd_pf >= 2) IRS_FWS_(P, 2, LO, HI);                                                                \
    else IRS_FWS_(P, 1, LO, HI)
#define IRS_FWS_(P, PFF, LO, HI) hipLaunchKernelGGL((exp_fwd_march_kernel<P, 1, 1, PFF>), dim3(total), dim3(FTX * FTY), 0, st, din, dout, vol, lin, sc, dmax_in, dmax_out, seg_len, nseg, LO, HI, swz_run, tiles, lay)
    // the radius-2 variant's own segments (as for the adjoint): a persistent grid of what the chip holds of it, four run-in planes
    const bool r2_rows1 = global_knobs().fwd_r2_rows1 != 0;
    int seg2 = seg_len;
    int64_t res2 = kRareGrid;
    if (global_knobs().seg_fit && seg_env <= 0) {
        static int cache_r2a = 0, cache_r2b = 0;
        const int64_t r = r2_rows1 ? resident_blocks((const void*)exp_fwd_march_kernel<false, 2, 1>, FTX * FTY, &cache_r2a)
                                   : resident_blocks((const void*)exp_fwd_march_kernel<false, 2>, FTX * FTY / FROWS_BIG, &cache_r2b);
        if (r > 0) {
            res2 = r;
            seg2 = pick_seg_len_fit(vol.nz, vol.nzb, per_layer, 8, 4, res2, 0, 64);
        }
    }
    const int nseg2 = vol_nseg(vol, seg2);
    const dim3 tiles2(tiles.x, tiles.y, (unsigned)(nseg2 * C));
    const int total2 = (int)(tiles2.x * tiles2.y * tiles2.z);
    const int rare = total2 < res2 ? total2 : (int)res2;
    if (!dmax_in || only_r1) {  // no bound / predicted small: the radius-1 ring is correct for any displacement (far taps go to global memory)
        if (small) { if (prescale_in) IRS_FWS(true, -1, 1 << 30); else IRS_FWS(false, -1, 1 << 30); }
        else if (prescale_in) IRS_FWM(true, 1, -1, 1 << 30, total); else IRS_FWM(false, 1, -1, 1 << 30, total);
    } else {
        if (small) { if (prescale_in) IRS_FWS(true, -1, 1); else IRS_FWS(false, -1, 1); }
        else if (prescale_in) IRS_FWM(true, 1, -1, 1, total); else IRS_FWM(false, 1, -1, 1, total);
        // radius-2 ring: 59 KB, two workgroups per CU -- with one output row per thread they are 16 waves instead of 8
        if (global_knobs().fwd_r2_rows1) { if (prescale_in) IRS_FW2(true, 1, 1 << 30, rare); else IRS_FW2(false, 1, 1 << 30, rare); }
        else if (prescale_in) IRS_FWM2(true, 1, 1 << 30, rare); else IRS_FWM2(false, 1, 1 << 30, rare);
    }
#undef IRS_FWM2
#undef IRS_FWS
#undef IRS_FWS_
#undef IRS_FW2
#undef IRS_FWM
}

// per-chain max |d| (voxels, per axis) of a field -- used by the stateless adjoint, which has no forward by-product
template <bool PRESCALE>
__global__ __launch_bounds__(kBlock) void field_absmax_kernel(const float* __restrict__ d, unsigned* __restrict__ dmax,
                                                              Vol vol, Scale3L sc) {
    __shared__ float red[3 * (kBlock / kWave)];
    const int chain = blockIdx.y;
    const float* c0 = d + (int64_t)chain * 3 * vol.V;
    float m[3] = {0.0f, 0.0f, 0.0f};
    IRS_ROWS_BEGIN(vol, x, y, z, v)
        (void)x; (void)y; (void)z;
#pragma unroll
        for (int c = 0; c < 3; ++c) m[c] = fmaxf(m[c], fabsf(ldp<PRESCALE>(c0 + c * vol.V, v, sc.nm1[c], sc.rnm1[c], sc.inv_pow)));
    IRS_ROWS_END
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        m[c] *= 0.5f * sc.nm1[c];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) m[c] = fmaxf(m[c], __shfl_down(m[c], off, kWave));
        if ((threadIdx.x & (kWave - 1)) == 0) red[c * (kBlock / kWave) + threadIdx.x / kWave] = m[c];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float mm = 0.0f;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) mm = fmaxf(mm, red[threadIdx.x * (kBlock / kWave) + w]);
        unsigned* slot = dmax + chain * 4 + threadIdx.x;
        if (__float_as_uint(mm) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_nonneg(slot, mm);
    }
}

void launch_field_absmax(const float* d, bool prescale_in, int no_steps, unsigned* dmax, int C, Vol vol, hipStream_t st) {
    const int rows = vol.H * vol.nz;
    dim3 grid((unsigned)(rows / 4 < 2048 ? (rows + 3) / 4 : 2048), C);
    const Scale3L sc = make_scale_l(vol, no_steps);
    if (prescale_in) hipLaunchKernelGGL(field_absmax_kernel<true>, grid, dim3(kBlock), 0, st, d, dmax, vol, sc);
    else hipLaunchKernelGGL(field_absmax_kernel<false>, grid, dim3(kBlock), 0, st, d, dmax, vol, sc);
}

}  // namespace irs
