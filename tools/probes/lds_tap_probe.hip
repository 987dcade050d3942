// The TAP pattern of the forward squaring step (exp_kernels.hip: exp_fwd_march_tile) in isolation: no HBM in the loop, the ring
// filled once, then `steps` plane steps of what the sampling phase does to the LDS -- per output voxel the centre record and the
// eight corners of the cell its sampling position falls into, the cell chosen PER LANE by a shift in {-1, 0} per axis (lane = x).
// Which ring layout / which instruction moves a tap fastest, and what do the per-lane shifts cost in bank conflicts?
//
//   layout 0  float2 (d0, d1) + float d2, as the shipped kernel writes it: the compiler pairs the cx = 0 / 1 corners into
//             ds_read2_b64 + ds_read2_b32
//   layout 1  the same arrays, every corner its own ds_read_b64 + ds_read_b32 (inline asm: nothing is paired)
//   layout 2  three planar float arrays (compiler: ds_read2_b32 pairs)
//   layout 3  16-byte records (d0, d1, d2, -), one ds_read_b128 per corner
//   layout 4  float2 (d0, d1) + float2 (d2[x], d2[x + 1]): three ds_read_b64 per corner PAIR
//   layout 6  layout 0's arrays through volatile LDS pointers: single ds_read_b64 + ds_read_b32 issued by the COMPILER (no pairing, its
//             own waits) -- the form the kernel ships (IRS_FWD_TAPS=3)
//   layout 5  layout 1's arrays, FIXED columns x - 1, x, x + 1 in the lane's two rows and planes (12 corner reads, 4 of them
//             with weight 0): no per-lane x shift, so no bank conflict whatever the field does
//   pitch     row pitch in records (80 = the shipped 16-record alignment of 66; 66 = unpadded; 81 / 82 = a row start rotated by one
//             / two records per row, and by 10 per plane)
//   pattern   0: every lane the same shifts; L > 0: the shifts change every L lanes along x, independently per axis (1 = white)
//
// Build:  hipcc -O3 --offload-arch=gfx950 tools/probes/lds_tap_probe.hip -o gpurun_out/lds_tap_probe
// Run:    gpurun_out/lds_tap_probe            (prints one line per layout x pitch x pattern: ns per wave tap-set, taps per clock and CU,
//                                              and a checksum that must agree between the layouts of one pattern)
//         rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS ... -- gpurun_out/lds_tap_probe   (counters per variant:
//         every variant is its own kernel instantiation)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int FTX = 64, FTY = 8, R = 1, NS = 4, PX = FTX + 2 * R, PY = FTY + 2 * R, kBlock = 256;

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// The asm layouts issue ALL reads of a tap-set and the wait for them in ONE asm statement: the compiler does not know that a
// ds_read's result arrives later, and between separate statements it is free to copy an "output" before it has landed (the first
// version of this probe -- and an asm variant of the kernel -- read garbage that way).  Early-clobber outputs, the row / plane
// bases as separate address registers so that every immediate offset is 0, 8 or 16 whatever the pitch.
#define RD64(o, a, off) "ds_read_b64 %" #o ", %" #a " offset:" #off "\n"
#define RD32(o, a, off) "ds_read_b32 %" #o ", %" #a " offset:" #off "\n"
#define RD128(o, a, off) "ds_read_b128 %" #o ", %" #a " offset:" #off "\n"
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)p; }

__device__ __forceinline__ float ring_value(int slot, int row, int col, int ch) {  // what the ring holds (any layout)
    return (float)((slot * 16 + row) * 128 + col) * (1.0f / 1024.0f) + 0.25f * (float)ch;
}

template <int LAYOUT, int PITCH>
__global__ __launch_bounds__(kBlock, 4) void tap_probe(float* __restrict__ out, int steps, int run_len) {
    constexpr int PNP = PITCH * PY + (PITCH % 16 ? 10 : 0);  // (a rotated pitch also rotates the plane start)
    constexpr int NREC = NS * PNP + 4;
    // one allocation per layout (bytes): 0, 1, 5: 12 per record; 2: 12; 3, 4: 16
    __shared__ __attribute__((aligned(16))) float lds[NREC * ((LAYOUT == 3 || LAYOUT == 4) ? 4 : 3)];
    float* const xy = lds;                      // layouts 0, 1, 4, 5: float2 per record
    float* const zz = lds + 2 * NREC;           // layouts 0, 1, 5: float per record; layout 4: float2 per record
    float* const p0 = lds;                      // layout 2: planes
    float* const p1 = lds + NREC;
    float* const p2 = lds + 2 * NREC;
    for (int i = threadIdx.x; i < NS * PY * PX; i += kBlock) {
        const int col = i % PX, row = (i / PX) % PY, slot = i / (PX * PY);
        const int e = slot * PNP + row * PITCH + col;
        const float v0 = ring_value(slot, row, col, 0), v1 = ring_value(slot, row, col, 1), v2 = ring_value(slot, row, col, 2);
        if (LAYOUT == 2) {
            p0[e] = v0;
            p1[e] = v1;
            p2[e] = v2;
        } else if (LAYOUT == 3) {
            lds[4 * e] = v0;
            lds[4 * e + 1] = v1;
            lds[4 * e + 2] = v2;
            lds[4 * e + 3] = 0.0f;
        } else if (LAYOUT == 4) {
            xy[2 * e] = v0;
            xy[2 * e + 1] = v1;
            zz[2 * e] = v2;
            zz[2 * e + 1] = ring_value(slot, row, col + 1, 2);
        } else {
            xy[2 * e] = v0;
            xy[2 * e + 1] = v1;
            zz[e] = v2;
        }
    }
    __syncthreads();
    const int lx = threadIdx.x % FTX, ly0 = threadIdx.x / FTX;
    // the shifts: one generator per axis, seeded per run of `run_len` lanes (0: one seed for everybody)
    auto seed = [&](int axis) {
        unsigned g = run_len > 0 ? (unsigned)((lx + 5 * axis) / run_len) : 0u;
        g = (g + 1u) * 2654435761u + (unsigned)axis * 40503u + (run_len > 0 ? (unsigned)ly0 * 97u : 0u);
        return g ^ (g >> 15);
    };
    unsigned gx = seed(0), gy = seed(1), gz = seed(2), hx = 0, hy = 0, hz = 0;
    const float w0 = 1.0f + 0.001f * (float)lx, w1 = 0.5f, w2 = 0.25f;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (int it = 0; it < steps; ++it) {
        if ((it & 15) == 0) {  // new shift words every 16 steps (2 bits per step: one per output row)
            gx = gx * 1664525u + 1013904223u;
            gy = gy * 1664525u + 1013904223u;
            gz = gz * 1664525u + 1013904223u;
            hx = gx ^ (gx >> 16);   // (the low bits of a power-of-two LCG are short-period)
            hy = gy ^ (gy >> 16);
            hz = gz ^ (gz >> 16);
        }
        const int a = it & (NS - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int bit = (it & 15) + 16 * j;
            const int sx = (hx >> bit) & 1, sy = (hy >> bit) & 1, sz = (hz >> bit) & 1;   // 1: the cell one step back
            const int ly = ly0 + j * (FTY / 2);
            const int bx0 = lx + R - sx, by0 = ly + R - sy;
            const int sl0 = ((a - sz) & (NS - 1)) * PNP, sl1 = ((a - sz + 1) & (NS - 1)) * PNP;
            const int ctr = a * PNP + (ly + R) * PITCH + lx + R;
            const int off = by0 * PITCH + bx0;
            if (LAYOUT == 6) {  // layout 0's arrays read through volatile LDS pointers: the compiler issues, counts and schedules the reads, but
                                // its load / store optimiser leaves volatile accesses unpaired -- what IRS_FWD_TAPS=3 does in the kernel
                typedef const volatile __attribute__((address_space(3))) f2* VQ;
                typedef const volatile __attribute__((address_space(3))) float* VZ;
                const VQ q = (VQ)(reinterpret_cast<const f2*>(xy));
                const VZ zq = (VZ)zz;
                const f2 c = q[ctr];
                a0 += c.x + zq[ctr];
                a1 += c.y;
#pragma unroll
                for (int cz = 0; cz < 2; ++cz)
#pragma unroll
                    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                        for (int cx = 0; cx < 2; ++cx) {
                            const int e = (cz ? sl1 : sl0) + off + cy * PITCH + cx;
                            const f2 t = q[e];
                            a0 = fmaf(t.x, w0, a0);
                            a1 = fmaf(t.y, w1, a1);
                            a2 = fmaf(zq[e], w2, a2);
                        }
            } else if (LAYOUT == 0) {
                const f2* q = reinterpret_cast<const f2*>(xy);
                const f2 c = q[ctr];
                a0 += c.x + zz[ctr];
                a1 += c.y;
#pragma unroll
                for (int cz = 0; cz < 2; ++cz)
#pragma unroll
                    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                        for (int cx = 0; cx < 2; ++cx) {
                            const int e = (cz ? sl1 : sl0) + off + cy * PITCH + cx;
                            const f2 t = q[e];
                            a0 = fmaf(t.x, w0, a0);
                            a1 = fmaf(t.y, w1, a1);
                            a2 = fmaf(zz[e], w2, a2);
                        }
            } else if (LAYOUT == 2) {
                a0 += p0[ctr] + p2[ctr];
                a1 += p1[ctr];
#pragma unroll
                for (int cz = 0; cz < 2; ++cz)
#pragma unroll
                    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                        for (int cx = 0; cx < 2; ++cx) {
                            const int e = (cz ? sl1 : sl0) + off + cy * PITCH + cx;
                            a0 = fmaf(p0[e], w0, a0);
                            a1 = fmaf(p1[e], w1, a1);
                            a2 = fmaf(p2[e], w2, a2);
                        }
            } else if (LAYOUT == 1) {
                const unsigned bxy = lds_addr(xy), bz = lds_addr(zz);
                const unsigned cA = bxy + 8u * (unsigned)ctr, cB = bz + 4u * (unsigned)ctr;
                const unsigned e0 = (unsigned)(sl0 + off), e1 = (unsigned)(sl1 + off);
                const unsigned x00 = bxy + 8u * e0, x01 = x00 + 8u * PITCH, x10 = bxy + 8u * e1, x11 = x10 + 8u * PITCH;
                const unsigned z00 = bz + 4u * e0, z01 = z00 + 4u * PITCH, z10 = bz + 4u * e1, z11 = z10 + 4u * PITCH;
                f2 c, t[8];
                float cz_, u[8];
                asm volatile(RD64(0, 18, 0) RD32(1, 19, 0)
                             RD64(2, 20, 0) RD64(3, 20, 8) RD64(4, 21, 0) RD64(5, 21, 8) RD64(6, 22, 0) RD64(7, 22, 8) RD64(8, 23, 0) RD64(9, 23, 8)
                             RD32(10, 24, 0) RD32(11, 24, 4) RD32(12, 25, 0) RD32(13, 25, 4) RD32(14, 26, 0) RD32(15, 26, 4) RD32(16, 27, 0) RD32(17, 27, 4)
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(c), "=&v"(cz_), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
                               "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
                             : "v"(cA), "v"(cB), "v"(x00), "v"(x01), "v"(x10), "v"(x11), "v"(z00), "v"(z01), "v"(z10), "v"(z11)
                             : "memory");
                a0 += c.x + cz_;
                a1 += c.y;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    a0 = fmaf(t[k].x, w0, a0);
                    a1 = fmaf(t[k].y, w1, a1);
                    a2 = fmaf(u[k], w2, a2);
                }
            } else if (LAYOUT == 3) {
                const unsigned bb = lds_addr(lds);
                const unsigned cA = bb + 16u * (unsigned)ctr, x00 = bb + 16u * (unsigned)(sl0 + off), x01 = x00 + 16u * PITCH,
                               x10 = bb + 16u * (unsigned)(sl1 + off), x11 = x10 + 16u * PITCH;
                f4 c, t[8];
                asm volatile(RD128(0, 9, 0) RD128(1, 10, 0) RD128(2, 10, 16) RD128(3, 11, 0) RD128(4, 11, 16) RD128(5, 12, 0) RD128(6, 12, 16)
                             RD128(7, 13, 0) RD128(8, 13, 16) "s_waitcnt lgkmcnt(0)"
                             : "=&v"(c), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                             : "v"(cA), "v"(x00), "v"(x01), "v"(x10), "v"(x11)
                             : "memory");
                a0 += c.x + c.z;
                a1 += c.y;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    a0 = fmaf(t[k].x, w0, a0);
                    a1 = fmaf(t[k].y, w1, a1);
                    a2 = fmaf(t[k].z, w2, a2);
                }
            } else if (LAYOUT == 4) {
                const unsigned bxy = lds_addr(xy), bz = lds_addr(zz);
                const unsigned cA = bxy + 8u * (unsigned)ctr, cB = bz + 8u * (unsigned)ctr;
                const unsigned e0 = (unsigned)(sl0 + off), e1 = (unsigned)(sl1 + off);
                const unsigned x00 = bxy + 8u * e0, x01 = x00 + 8u * PITCH, x10 = bxy + 8u * e1, x11 = x10 + 8u * PITCH;
                const unsigned z00 = bz + 8u * e0, z01 = z00 + 8u * PITCH, z10 = bz + 8u * e1, z11 = z10 + 8u * PITCH;
                f2 c, t[8], u[4];  // u: (d2[x0], d2[x0 + 1]) per (cz, cy)
                float cz_;
                asm volatile(RD64(0, 14, 0) RD32(1, 15, 0)
                             RD64(2, 16, 0) RD64(3, 16, 8) RD64(4, 17, 0) RD64(5, 17, 8) RD64(6, 18, 0) RD64(7, 18, 8) RD64(8, 19, 0) RD64(9, 19, 8)
                             RD64(10, 20, 0) RD64(11, 21, 0) RD64(12, 22, 0) RD64(13, 23, 0) "s_waitcnt lgkmcnt(0)"
                             : "=&v"(c), "=&v"(cz_), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
                               "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])
                             : "v"(cA), "v"(cB), "v"(x00), "v"(x01), "v"(x10), "v"(x11), "v"(z00), "v"(z01), "v"(z10), "v"(z11)
                             : "memory");
                a0 += c.x + cz_;
                a1 += c.y;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    a0 = fmaf(t[k].x, w0, a0);
                    a1 = fmaf(t[k].y, w1, a1);
                    a2 = fmaf((k & 1) ? u[k >> 1].y : u[k >> 1].x, w2, a2);
                }
            } else {  // 5: fixed columns lx + R - 1 .. lx + R + 1 in the lane's own rows / planes; the x shift picks the weights
                const unsigned bxy = lds_addr(xy), bz = lds_addr(zz);
                const unsigned cA = bxy + 8u * (unsigned)ctr, cB = bz + 4u * (unsigned)ctr;
                const unsigned e0 = (unsigned)(sl0 + by0 * PITCH + lx + R - 1), e1 = (unsigned)(sl1 + by0 * PITCH + lx + R - 1);
                const unsigned x00 = bxy + 8u * e0, x01 = x00 + 8u * PITCH, x10 = bxy + 8u * e1, x11 = x10 + 8u * PITCH;
                const unsigned z00 = bz + 4u * e0, z01 = z00 + 4u * PITCH, z10 = bz + 4u * e1, z11 = z10 + 4u * PITCH;
                f2 c, t[12];
                float cz_, u[12];
                asm volatile(RD64(0, 14, 0) RD32(1, 15, 0)
                             RD64(2, 16, 0) RD64(3, 16, 8) RD64(4, 16, 16) RD64(5, 17, 0) RD64(6, 17, 8) RD64(7, 17, 16)
                             RD32(8, 18, 0) RD32(9, 18, 4) RD32(10, 18, 8) RD32(11, 19, 0) RD32(12, 19, 4) RD32(13, 19, 8) "s_waitcnt lgkmcnt(0)"
                             : "=&v"(c), "=&v"(cz_), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]),
                               "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5])
                             : "v"(cA), "v"(cB), "v"(x00), "v"(x01), "v"(z00), "v"(z01)
                             : "memory");  // (26 results + 10 addresses exceed the 30 operands of one statement: two blocks, each with its wait)
                asm volatile(RD64(0, 12, 0) RD64(1, 12, 8) RD64(2, 12, 16) RD64(3, 13, 0) RD64(4, 13, 8) RD64(5, 13, 16)
                             RD32(6, 14, 0) RD32(7, 14, 4) RD32(8, 14, 8) RD32(9, 15, 0) RD32(10, 15, 4) RD32(11, 15, 8) "s_waitcnt lgkmcnt(0)"
                             : "=&v"(t[6]), "=&v"(t[7]), "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]),
                               "=&v"(u[6]), "=&v"(u[7]), "=&v"(u[8]), "=&v"(u[9]), "=&v"(u[10]), "=&v"(u[11])
                             : "v"(x10), "v"(x11), "v"(z10), "v"(z11)
                             : "memory");
                a0 += c.x + cz_;
                a1 += c.y;
                const float m0 = sx ? 1.0f : 0.0f, m2 = sx ? 0.0f : 1.0f;  // column -1 counts for a shifted lane, column +1 for an unshifted one
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const float m = (k % 3 == 0) ? m0 : (k % 3 == 2) ? m2 : 1.0f;
                    a0 = fmaf(t[k].x, w0 * m, a0);
                    a1 = fmaf(t[k].y, w1 * m, a1);
                    a2 = fmaf(u[k], w2 * m, a2);
                }
            }
        }
    }
    out[(size_t)blockIdx.x * kBlock + threadIdx.x] = a0 + a1 + a2;
}

static int g_cus = 256;

template <int LAYOUT, int PITCH>
static void run(float* out, float* host, int run_len, int steps) {
    int per_cu = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tap_probe<LAYOUT, PITCH>, kBlock, 0);
    if (per_cu < 1) per_cu = 1;
    const int blocks = per_cu * g_cus;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((tap_probe<LAYOUT, PITCH>), dim3(blocks), dim3(kBlock), 0, 0, out, 64, run_len);
    hipEventRecord(a);
    hipLaunchKernelGGL((tap_probe<LAYOUT, PITCH>), dim3(blocks), dim3(kBlock), 0, 0, out, steps, run_len);
    hipEventRecord(b);
    if (hipEventSynchronize(b) != hipSuccess) {
        printf("layout %d pitch %d: launch failed\n", LAYOUT, PITCH);
        exit(1);
    }
    float ms = 0.0f;
    hipEventElapsedTime(&ms, a, b);
    hipMemcpy(host, out, sizeof(float) * kBlock, hipMemcpyDeviceToHost);  // workgroup 0 (every workgroup computes the same)
    double sum = 0.0;
    for (int i = 0; i < kBlock; ++i) sum += host[i];
    const double wave_sets = (double)blocks * (kBlock / 64) * steps * 2;  // a tap-set = centre + 8 corners of one output row of a wave
    const double ns_cu = ms * 1e6 / (wave_sets / g_cus);                  // ns of one CU per wave tap-set
    printf("layout %d pitch %2d run_len %2d: %d wg/CU  %.3f ms  %.2f ns per wave tap-set and CU (%.1f clk at 2.1 GHz)  checksum %.6e\n", LAYOUT, PITCH,
           run_len, per_cu, ms, ns_cu, ns_cu * 2.1, sum);
    hipEventDestroy(a);
    hipEventDestroy(b);
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 4000;
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, dev);
    float* out;
    hipMalloc(&out, sizeof(float) * kBlock * 256 * 8);
    float* host = (float*)malloc(sizeof(float) * kBlock);
    const int runs[4] = {0, 16, 4, 1};
    for (int r = 0; r < 4; ++r) {
        const int L = runs[r];
        run<0, 80>(out, host, L, steps);
        run<0, 66>(out, host, L, steps);
        run<0, 81>(out, host, L, steps);
        run<6, 80>(out, host, L, steps);
        run<1, 80>(out, host, L, steps);
        run<1, 66>(out, host, L, steps);
        run<1, 81>(out, host, L, steps);
        run<1, 82>(out, host, L, steps);
        run<2, 80>(out, host, L, steps);
        run<3, 80>(out, host, L, steps);
        run<3, 66>(out, host, L, steps);
        run<4, 80>(out, host, L, steps);
        run<5, 80>(out, host, L, steps);
        printf("\n");
    }
    return 0;
}
