// Shared device helpers for the gfx950 kernels.  CDNA4 only: 64-wide wavefronts, no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/irsgmcmc.h"

namespace irs {

constexpr int kBlock = 256;        // 4 wavefronts
constexpr int kWave = 64;
constexpr int kMaxPartialBlocks = 2048;  // cap for per-block partial sums (reduced in fixed order)

// A volume plus the z-window(s) of OUTPUT planes a launch covers.  Indexing is always global in z (clamping uses D); the
// window only restricts which planes a kernel writes.  A single GPU uses the full window.  The z-slab decomposition
// (slab.hip) launches every stage on its own slab (+ the ghost planes it recomputes) and allocates SLAB-LOCAL arrays: `V` is
// the element stride between the channels / chains of an array (planes held * H * W), while the base pointers handed to the
// kernels are shifted down by (first held plane) * H * W elements, so that global z indexing lands inside the allocation.
// `Vg` = D * H * W is the voxel count of the whole volume (counter of the in-kernel Philox noise: a slab draws the noise
// the full volume would).  The squaring-step kernels accept a second window [z0b, z0b + nzb) -- the two boundary strips
// of a slab in one launch (slab.hip: interior / boundary split around a ghost-plane exchange).
struct Vol {
    int D, H, W;
    int64_t V;   // channel / chain stride in elements
    int z0, nz;
    int z0b, nzb;
    int64_t Vg;  // D*H*W
};

__host__ __device__ inline Vol make_vol(int D, int H, int W) {
    return Vol{D, H, W, (int64_t)D * H * W, 0, D, 0, 0, (int64_t)D * H * W};
}

__host__ __device__ inline Vol window(Vol v, int zlo, int zhi) {
    zlo = zlo < 0 ? 0 : zlo;
    zhi = zhi > v.D ? v.D : zhi;
    v.z0 = zlo;
    v.nz = zhi > zlo ? zhi - zlo : 0;
    v.z0b = 0;
    v.nzb = 0;
    return v;
}

// two disjoint windows [lo0, hi0) and [lo1, hi1), lo1 >= hi0 (an empty first window is replaced by the second)
__host__ __device__ inline Vol window2(Vol v, int lo0, int hi0, int lo1, int hi1) {
    v = window(v, lo0, hi0);
    lo1 = lo1 < 0 ? 0 : lo1;
    hi1 = hi1 > v.D ? v.D : hi1;
    const int n1 = hi1 > lo1 ? hi1 - lo1 : 0;
    if (v.nz == 0) {
        v.z0 = lo1;
        v.nz = n1;
    } else {
        v.z0b = lo1;
        v.nzb = n1;
    }
    return v;
}

// z-segments of a (two-window) marching launch: segments never straddle the two windows
__host__ __device__ inline int vol_nseg(const Vol& v, int seg_len) {
    return (v.nz + seg_len - 1) / seg_len + (v.nzb + seg_len - 1) / seg_len;
}
__device__ __forceinline__ void seg_range(const Vol& v, int seg, int seg_len, int& z0, int& z1) {
    const int na = (v.nz + seg_len - 1) / seg_len;
    if (seg < na) {
        z0 = v.z0 + seg * seg_len;
        z1 = min(z0 + seg_len, v.z0 + v.nz);
    } else {
        z0 = v.z0b + (seg - na) * seg_len;
        z1 = min(z0 + seg_len, v.z0b + v.nzb);
    }
}
// voxels a launch writes (sizes the partial-sum grids)
__host__ __device__ inline int64_t win_voxels(const Vol& v) { return (int64_t)(v.nz + v.nzb) * v.H * v.W; }

// Tuning / test switches.  The environment is read ONCE per process (first use); a context takes a copy when it is created
// and nothing on the per-transition path calls getenv.  None of them selects other arithmetic: they pick launch shapes,
// which (always-correct) kernel variants are launched, and test hooks.  irs_option_set changes one by name afterwards
// (tests, tools/): on a context, or process-wide (the stateless operators and every context created later).
struct Knobs {
    int predict_variants = 1;  // IRS_PREDICT_VARIANTS  0: launch every variant; 1: production heuristic; 2: always predict small (tests)
    int run_ahead = 2;         // IRS_RUN_AHEAD         transitions the host may run ahead of the device (1..3)
    int fuse_warp_bwd = 1;     // IRS_FUSE_WARP_BWD     backward warp folded into the first adjoint step
    int energy_in_update = 1;  // IRS_ENERGY_IN_UPDATE  regulariser energy as a by-product of the update kernel (L2 family)
    int fuse_noise = 1;        // IRS_FUSE_NOISE        Langevin noise generated while the smoothing kernel stages its planes
    int recover = 1;           // IRS_RECOVER           keep the previous velocity and re-run a transition whose variant prediction failed
    int fwd_rows1 = -1;        // IRS_FWD_ROWS1         forward squaring step with one output row per thread (512 threads): -1 by launch size, 0 / 1
    int lds_from = 3;          // IRS_LDS_FROM          adjoint: smallest source halo (floor(max|d_k|) + 1) the any-radius kernel takes; 3 = radius-2 gather in front of it, 2 = not
    int fwd_r2_rows1 = 1;      // IRS_FWD_R2_ROWS1      radius-2 forward squaring step with one output row per thread (512 threads)
    int fwd_z2 = 0;            // IRS_FWD_Z2            one-row forward variant (small launches): two planes per marching step (exp_fwd_march_z2_kernel);
                               //                       bit-identical, 0.837 against 0.843 ms at 128^3, flat with two chains and on slab ranks: off
                               //                       (profiles/r05_fwd_z2_ab.txt)
    int tile_box = 1;          // IRS_TILE_BOX          any-radius adjoint: source boxes / scales of its tiles from tile_box_kernel (0: every workgroup its own)
    int fwd_pf = 2;            // IRS_FWD_PF            one-row forward variant (small launches): planes of global loads in flight ahead of the commit (1 / 2)
    int seg_fit = 1;           // IRS_SEG_FIT           squaring-step kernels: segment length from the resident-set cost model (0: power-of-two rule)
    int coarse_box = 1;        // IRS_COARSE_BOX        any-radius adjoint: source boxes from the coarse displacement extrema
    int sobolev_tile = 0;      // IRS_SOBOLEV_TILE      0: by size; 1: 32 x 16; 2: 64 x 32 ("small" / "big")
    int ps_rows = 0;           // IRS_PS_ROWS           fused perturbation + smoothing: rows per tile, 0: 16 with a sigma field, 32 without; 16 / 32
    int march_seg = 0, march_seg_fwd = 0, swz_run = -1, seg_min_blocks = 0, seg_min_len = 0;  // IRS_MARCH_SEG, _FWD, IRS_SWZ_RUN, IRS_SEG_MIN_*
    int sobolev_seg = 0, lcc_seg = 0, stats_seg = 0, update_seg = 0;                          // IRS_*_SEG
    int slab_buffers = 3;      // IRS_SLAB_BUFFERS      gradient fields the adjoint of a multi-rank slab rotates through (2: rounds of at most two steps)
    int slab_split = 1;        // IRS_SLAB_SPLIT        interior / boundary split around an exchange
    int slab_exact = 0;        // IRS_SLAB_EXACT        every transition in measuring mode
    int slab_force_h = 0;      // IRS_SLAB_FORCE_H      test hook: a deliberately wrong ghost-width plan
    int data_batch = 1;        // IRS_DATA_BATCH        C > 1, GMM / LCC: the data terms of all chains in ONE launch behind the serial statistics -> step loop
    int chain_overlap = 0;     // IRS_CHAIN_OVERLAP     C > 1: data term of chain c on a side stream, overlapping the statistics of chain c + 1
                               //                       (round 5, asked for; measured SLOWER: 0.728 against 0.711 ms per chain-transition at 128^3
                               //                       C = 2, 1.931 against 1.917 at 192^3, three alternating runs -- off; read when a context is created)
    int launch_log = 0;        // IRS_LAUNCH_LOG        print the shape of every distinct marching launch once (stderr; tools/launch_shapes.py)
};
Knobs& global_knobs();                                   // api.hip; initialised from the environment on first use
int knob_set(Knobs& k, const char* name, int value, bool on_context);  // 0 on success (api.hip: scopes)

// Segment length of the z-marching kernels.  32 planes amortise the run-in of a segment (2R .. 4S extra planes) when the
// launch still has enough workgroups to fill 256 CUs; smaller volumes trade run-in overhead for parallelism, down to
// `min_len`.  `forced` (> 0) is the environment override of the kernel family.
inline int pick_seg_len(int nz, int64_t tiles_per_layer, int min_len, int forced, int64_t want_blocks = 1024) {
    if (forced > 0) return forced;
    const Knobs& kn = global_knobs();
    if (kn.seg_min_len > 0) min_len = kn.seg_min_len;
    const int64_t want = kn.seg_min_blocks > 0 ? kn.seg_min_blocks : want_blocks;
    int len = 32;
    while (len > min_len && tiles_per_layer * ((nz + len - 1) / len) < want) len >>= 1;
    return len < 1 ? 1 : len;
}

// Segment length from a cost model of the launch: workgroups run in rounds of `resident` (what the chip holds at once), a
// workgroup marches seg_len + run_in planes, so a launch costs about ceil(blocks / resident) * (seg_len + run_in) plane steps.
// The power-of-two rule above can land badly on sizes that are not powers of two: two chains at 192^3 are 1728 adjoint workgroups
// of 34 plane steps on 1024 slots (two rounds, the second 70 % full) where 7 segments of 28 planes are 2016 workgroups of 30 steps
// -- config 5 runs 454 -> 474 samples/s.  At 256^3, 128^3 and on a slab rank of eight the two rules agree or measure the same.
// `max_len` > 32 (the adjoint squaring step, round 5): 256 x 256 tiles in-plane are two rounds of 34 plane steps with 32-plane segments
// and ONE round of 66 with 64-plane ones -- 4.43-4.51 -> 4.38-4.41 ms per transition at 256^3 (profiles/r05_bwd_long_seg_sweep.txt;
// the forward step, whose 32-plane launch is one round already, is slower with longer segments)
inline int pick_seg_len_fit(int nz, int nzb, int64_t tiles_per_layer, int min_len, int run_in, int64_t resident, int forced,
                            int max_len = 32) {
    if (forced > 0) return forced;
    const int longest = nz > nzb ? nz : nzb;
    int best = 32;
    int64_t best_cost = -1;
    for (int len = max_len; len >= min_len; --len) {
        if (len > longest && len != 32) continue;
        const int64_t nseg = (nz + len - 1) / len + (nzb + len - 1) / len;
        const int64_t rounds = (tiles_per_layer * nseg + resident - 1) / resident;
        const int64_t cost = rounds * ((len < longest ? len : longest) + run_in);
        if (best_cost < 0 || cost < best_cost) {  // ties: the longer segment (less run-in traffic)
            best_cost = cost;
            best = len;
        }
    }
    return best;
}

// workgroups of `kernel` the chip holds at once (asked from the runtime once per kernel; 0 if it cannot be had)
inline int64_t resident_blocks(const void* kernel, int block, int* cache) {
    if (*cache == 0) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) == hipSuccess && per_cu > 0 && cus > 0)
            *cache = per_cu * cus;
        else
            *cache = -1;
    }
    return *cache > 0 ? *cache : 0;
}

// IRS_LAUNCH_LOG=1: one stderr line per distinct (kernel, shape) -- workgroups, threads, segment length, run-in planes, plane steps a
// workgroup marches, workgroups the chip holds at once -> rounds.  What a launch on a small volume or a thin slab looks like
// ("where a small launch loses"); api.hip.
void log_launch(const char* kernel, int tile_x, int tile_y, int64_t blocks, int threads, int seg_len, int run_in, int planes_out,
                int chains, int64_t resident);

inline dim3 vox_grid(const Vol& vol, int planes) {
    return dim3((unsigned)((vol.W + 63) / 64), (unsigned)((vol.H + 3) / 4), (unsigned)(vol.nz * planes));
}

// Division-free voxel mapping for the pointwise kernels: 256-thread blocks laid out 64 (x) x 4 (y),
// grid = (ceil(W/64), ceil(H/4), D * planes) where a "plane" is a chain or a (chain, channel) pair.
// (A flat 1-D index needs three 64-bit divisions per thread -- hundreds of instructions in kernels that move 20-50 bytes.)
#define IRS_VOXEL(vol, PLANE, X, Y, Z, VOX)                                   \
    const int X = blockIdx.x * 64 + (threadIdx.x & 63);                        \
    const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);                         \
    const int PLANE = blockIdx.z / (vol).nz;                                   \
    const int Z = (vol).z0 + blockIdx.z - PLANE * (vol).nz;                    \
    if (X >= (vol).W || Y >= (vol).H) return;                                  \
    const int64_t VOX = ((int64_t)Z * (vol).H + Y) * (vol).W + X

// Grid-stride loop over the rows of a volume for the reduction kernels: each wavefront of a 256-thread block walks
// rows (y, z) and its lanes stride along x, so the only division is one 32-bit one per row.
#define IRS_ROWS_BEGIN(vol, X, Y, Z, VOX)                                                                             \
    for (int row_ = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6); row_ < (vol).H * (vol).nz;                    \
         row_ += gridDim.x * (kBlock / kWave)) {                                                                       \
        const int zr_ = row_ / (vol).H, Y = row_ - zr_ * (vol).H, Z = (vol).z0 + zr_;                                  \
        for (int X = threadIdx.x & 63; X < (vol).W; X += kWave) {                                                      \
            const int64_t VOX = ((int64_t)Z * (vol).H + Y) * (vol).W + X;
#define IRS_ROWS_END \
        }            \
    }

// XCD-aware block -> tile remap (cdna guide T1, bijective form).  Workgroups are dealt round-robin over the 8 XCDs, each
// with a private L2, so linearly adjacent tiles -- which share halo cache lines -- land on 8 different L2s and every one
// re-fetches the shared lines (measured: 3.3x the algorithmic read bytes at the fabric for the marching adjoint).
// Returns the linear tile index this workgroup should process so that each XCD owns a contiguous run of tiles.
__device__ __forceinline__ int xcd_swizzle(int id, int total) {
    const int q = total >> 3, r = total & 7, xcd = id & 7, pos = id >> 3;
    return xcd * q + min(xcd, r) + pos;
}

// Finer variant: only runs of `run` consecutive tiles (one x-row of tiles, the ones that share the cache lines a halo row
// straddles) are kept on one XCD; successive runs still rotate over the XCDs.  Bijective when total % (8 * run) == 0,
// identity otherwise.
__device__ __forceinline__ int xcd_swizzle_runs(int id, int total, int run) {
    if (run <= 1) return id;
    const int group = 8 * run;
    if (total % group) return id;
    const int g = id / group, w = id - g * group;
    return g * group + (w & 7) * run + (w >> 3);
}

// The stencil kernels' 3-D grids (tile x, tile y, segment): the dispatcher walks them x fastest and deals the workgroups round-
// robin over the XCDs, so x-adjacent tiles -- which share the cache lines their halo columns straddle -- meet in eight different
// L2s.  Logical block after the remap: IRS_SWZ_STENCIL_ROWS x-rows of tiles form one run on one XCD (0: no remap).
#ifndef IRS_SWZ_STENCIL_ROWS
#define IRS_SWZ_STENCIL_ROWS 1
#endif
struct Blk3 {
    int x, y, z;
};
__device__ __forceinline__ Blk3 swizzled_block() {
    const int gx = (int)gridDim.x, gy = (int)gridDim.y;
    const int id = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);
    const int t = xcd_swizzle_runs(id, gx * gy * (int)gridDim.z, gx * IRS_SWZ_STENCIL_ROWS);
    const int q = t / gx;
    return {t - q * gx, q % gy, q / gy};
}

// identity-grid tables: linspace(-1, 1, n) per axis as torch's CPU kernel computes it
// (utils/util.py:263-278).  x <-> W, y <-> H, z <-> D.
struct Lin {
    const float* x;
    const float* y;
    const float* z;
};

// ------------------------------------------------------------------------------------------------
// trilinear sampling, ATen semantics (bilinear / border / align_corners=True).
// Arithmetic order follows ATen's CPU grid_sampler_3d (no FMA contraction in the coordinate arithmetic) so that positions, cell
// indices and weights are the reference's CPU path's bit for bit (the squaring step accumulates its tap PRODUCTS with FMAs since
// round 5, exp_kernels.hip: IRS_FWD_FMA): i = ((g + 1) / 2) * (n - 1); clip to [0, n-1] with a zero
// gradient on and outside the border; weights (i0 + 1 - i) and (i - i0); corners accumulated x fastest.
// ------------------------------------------------------------------------------------------------
struct AxisTap {
    int i0, i1;    // corner indices (i1 clamped; its weight is exactly 0 whenever it would be out of range)
    float w0, w1;  // weights
    float gmul;    // d(i)/d(g): (n-1)/2 strictly inside, 0 on / outside the border
};

__device__ __forceinline__ AxisTap axis_tap(float g, int n) {
    AxisTap t;
    const float nm1 = (float)(n - 1);
    const float raw = __fmul_rn(__fmul_rn(__fadd_rn(g, 1.0f), 0.5f), nm1);
    // branch-free clip (selects, not divergent branches: this runs three times per voxel in every sampling kernel)
    const bool interior = raw > 0.0f && raw < nm1;
    const float i = raw <= 0.0f ? 0.0f : (raw >= nm1 ? nm1 : raw);
    t.gmul = interior ? nm1 * 0.5f : 0.0f;
    const float f = floorf(i);
    t.i0 = (int)f;
    t.i1 = min(t.i0 + 1, n - 1);
    t.w1 = __fsub_rn(i, f);
    t.w0 = __fsub_rn(__fadd_rn(f, 1.0f), i);
    return t;
}

// transform_coordinates (utils/util.py:418-429): (v * 2) / (n - 1), then / 2^steps (utils/transformation.py:68)
// a / b correctly rounded at 3 instructions, given rb = the correctly rounded reciprocal of b (host: exact_rcp, a kernel
// argument and therefore an SGPR): quotient estimate, exact remainder, one correction (Markstein).  The IEEE division
// sequence is ~10 instructions per element; tests/csrc/div_exact_check.c checks bit-equality with a / b.
inline float exact_rcp(float nf) {  // the float closest to 1 / n: minimise |r n - 1| (exact in double) over the neighbours
    const double n = (double)nf;
    float best = (float)(1.0 / (n > 0.0 ? n : 1.0));
    const float cand[2] = {nextafterf(best, 0.0f), nextafterf(best, 2.0f)};
    for (float r : cand)
        if (fabs((double)r * n - 1.0) < fabs((double)best * n - 1.0)) best = r;
    return best;
}
__device__ __forceinline__ float div_exact(float a, float b, float rb) {
    const float q = __fmul_rn(a, rb);
    return __fmaf_rn(__fmaf_rn(-q, b, a), rb, q);
}
__device__ __forceinline__ float prescale(float v, float nm1, float rnm1, float inv_pow) {
    return __fmul_rn(div_exact(__fmul_rn(v, 2.0f), nm1, rnm1), inv_pow);
}

// ------------------------------------------------------------------------------------------------
// wavefront / block reductions (fp64 accumulators: 1e-5 relative on sums over up to 2^24 voxels needs
// better than fp32 running sums)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// reduce NV values per thread across the block; result valid in thread 0.  smem: NV * (kBlock/kWave) doubles.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* smem) {
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    // the butterfly of wave_sum, one LEVEL for all NV values at a time: value by value it is a chain of 6 NV dependent
    // ds_bpermute round trips (21 values: 126 x ~150 cycles = 8 us of the 22 us chain_scalar_kernel took).  Same additions, same order.
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        double t[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) t[i] = __shfl_down(v[i], off, kWave);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += t[i];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane == 0) smem[i * (kBlock / kWave) + wid] = v[i];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kBlock / kWave; ++w) s += smem[i * (kBlock / kWave) + w];
            v[i] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG (in-kernel noise when the caller injects none).
// counter = (index lo, index hi, iteration lo, stream id), key = seed.
// ------------------------------------------------------------------------------------------------
struct U4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per lane pair (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: the integer multiplies
        // are quarter-rate on CDNA and dominate the cost of the generator
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// Philox2x32-10: half the multiplies of the 4x32 generator for 64 output bits -- three 21-bit uniforms, what the jitter of
// one voxel needs.  96 input bits: counter = (index lo, index hi[3:0] | iteration[27:0] << 4), key = key_mix(...).
struct U2 {
    uint32_t x, y;
};
__device__ __forceinline__ U2 philox2x32_10(U2 c, uint32_t k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p = (uint64_t)0xD256D193u * c.x;
        c = U2{(uint32_t)(p >> 32) ^ k ^ c.y, (uint32_t)p};
        k += 0x9E3779B9u;
    }
    return c;
}
// 32-bit key from the 64-bit seed, the stream id and the iteration bits that do not fit the counter (wave-uniform: SALU)
__device__ __forceinline__ uint32_t key_mix(uint64_t seed, uint64_t iteration, uint32_t stream) {
    uint64_t zz = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(stream + 1u) + (iteration >> 28) * 0xD1B54A32D192ED03ull;
    zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
    zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
    zz ^= zz >> 31;
    return (uint32_t)zz ^ (uint32_t)(zz >> 32);
}
__device__ __forceinline__ float u01_21(uint32_t r21) { return (float)r21 * (1.0f / 2097152.0f); }  // 21 bits -> [0, 1)

// six 21-bit words from the 128 bits of one Philox4x32 call
__device__ __forceinline__ void split21(const U4 r, uint32_t (&w)[6]) {
    w[0] = r.x >> 11;
    w[1] = r.y >> 11;
    w[2] = r.z >> 11;
    w[3] = (r.x & 0x7FFu) | ((r.w & 0x3FFu) << 11);
    w[4] = (r.y & 0x7FFu) | (((r.w >> 10) & 0x3FFu) << 11);
    w[5] = (r.z & 0x7FFu) | (((r.w >> 20) & 0x3FFu) << 11);
}

// two standard normals from two 21-bit words: radius from u in (0, 1] (tail cut at 5.4 sigma, 7e-8 of the mass), angle in
// revolutions straight into v_sin / v_cos; hardware log2 / sqrt (1 ulp)
__device__ __forceinline__ void box_muller21(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u = ((float)a + 1.0f) * (1.0f / 2097152.0f);
    const float v = u01_21(b);
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));  // -2 ln u = -2 ln2 log2 u
    n0 = r * __builtin_amdgcn_cosf(v);
    n1 = r * __builtin_amdgcn_sinf(v);
}

}  // namespace irs
