// Streaming-bandwidth probe: what a pure copy reaches on this GPU as a function of access width and shape.
// Build: hipcc -O3 --offload-arch=gfx950 tools/bw_probe.hip -o tools/bw_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <typename T>
__global__ __launch_bounds__(256) void copy_flat(const T* __restrict__ a, T* __restrict__ b, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
// 3 planar read streams + 3 planar write streams, dword per lane, 64 x 4 block mapping over a 256^3 volume
__global__ __launch_bounds__(256) void copy3_dword(const float* __restrict__ a, float* __restrict__ b, int64_t V) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < V) {
        b[i] = a[i];
        b[i + V] = a[i + V];
        b[i + 2 * V] = a[i + 2 * V];
    }
}
// marching skeleton: block 32x8 column, loops over 32 planes, rows of 34 with halo (1.33x redundant), dword loads
__global__ __launch_bounds__(256) void march_skeleton(const float* __restrict__ a, float* __restrict__ b, int N, int halo) {
    const int PX = 32 + 2 * halo, PY = 8 + 2 * halo, PN = PX * PY;
    __shared__ float lds[3 * 12 * 36];
    const int ox = blockIdx.x * 32, oy = blockIdx.y * 8, z0 = blockIdx.z * 32;
    const int64_t V = (int64_t)N * N * N;
    const int lx = threadIdx.x % 32, ly = threadIdx.x / 32;
    for (int z = z0; z < z0 + 32; ++z) {
        for (int i = threadIdx.x; i < PN; i += 256) {
            const int px = i % PX, py = i / PX;
            const int cx = min(max(ox - halo + px, 0), N - 1), cy = min(max(oy - halo + py, 0), N - 1);
            const int64_t g = ((int64_t)z * N + cy) * N + cx;
            lds[i] = a[g];
            lds[PN + i] = a[g + V];
            lds[2 * PN + i] = a[g + 2 * V];
        }
        __syncthreads();
        const int ci = (ly + halo) * PX + lx + halo;
        const int64_t g = ((int64_t)z * N + oy + ly) * N + ox + lx;
        b[g] = lds[ci];
        b[g + V] = lds[PN + ci];
        b[g + 2 * V] = lds[2 * PN + ci];
        __syncthreads();
    }
}


// same, with the next plane prefetched into registers before the barrier (DEPTH planes in flight)
template <int DEPTH>
__global__ __launch_bounds__(256) void march_prefetch(const float* __restrict__ a, float* __restrict__ b, int N) {
    constexpr int halo = 1, PX = 34, PY = 10, PN = PX * PY, NIT = 2;
    __shared__ float lds[3 * PN];
    const int ox = blockIdx.x * 32, oy = blockIdx.y * 8, z0 = blockIdx.z * 32;
    const int64_t V = (int64_t)N * N * N, HW = (int64_t)N * N;
    const int lx = threadIdx.x % 32, ly = threadIdx.x / 32;
    unsigned off[NIT];
    bool val[NIT];
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * 256;
        const int px = i % PX, py = i / PX;
        const int cx = min(max(ox - halo + px, 0), N - 1), cy = min(max(oy - halo + py, 0), N - 1);
        val[it] = i < PN;
        off[it] = (unsigned)(cy * N + cx);
    }
    float pre[DEPTH][NIT][3];
    auto load = [&](int z, int d) {
        const float* p = a + (int64_t)min(z, N - 1) * HW;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (val[it]) {
                pre[d][it][0] = p[off[it]];
                pre[d][it][1] = (p + V)[off[it]];
                pre[d][it][2] = (p + 2 * V)[off[it]];
            }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(z0 + d, d);
    for (int zb = z0; zb < z0 + 32; zb += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int z = zb + d;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (val[it]) {
                    const int i = threadIdx.x + it * 256;
                    lds[i] = pre[d][it][0];
                    lds[PN + i] = pre[d][it][1];
                    lds[2 * PN + i] = pre[d][it][2];
                }
            load(z + DEPTH, d);
            __syncthreads();
            const int ci = (ly + halo) * PX + lx + halo;
            const int64_t g = ((int64_t)z * N + oy + ly) * N + ox + lx;
            b[g] = lds[ci];
            b[g + V] = lds[PN + ci];
            b[g + 2 * V] = lds[2 * PN + ci];
            __syncthreads();
        }
    }
}

// same, staged by LDS-DMA (global_load_lds, 4 B per lane) with PF planes in flight across a raw barrier
template <int PF>
__global__ __launch_bounds__(256) void march_glds(const float* __restrict__ a, float* __restrict__ b, int N) {
    constexpr int halo = 1, PX = 34, PY = 10, PN = PX * PY;
    constexpr int SLOT = 3 * 512;  // 3 components x 2 wave-instructions of 256 lanes (padded)
    constexpr int NS = PF + 2;
    __shared__ float lds[NS * SLOT];
    const int ox = blockIdx.x * 32, oy = blockIdx.y * 8, z0 = blockIdx.z * 32;
    const int64_t V = (int64_t)N * N * N, HW = (int64_t)N * N;
    const int lx = threadIdx.x % 32, ly = threadIdx.x / 32;
    unsigned off[2];
    for (int it = 0; it < 2; ++it) {
        const int i = min((int)threadIdx.x + it * 256, PN - 1);
        const int px = i % PX, py = i / PX;
        const int cx = min(max(ox - halo + px, 0), N - 1), cy = min(max(oy - halo + py, 0), N - 1);
        off[it] = (unsigned)(cy * N + cx);
    }
    const int wave = threadIdx.x / 64;
    auto issue = [&](int z, int slot) {
        const float* p = a + (int64_t)min(z, N - 1) * HW;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                float* dst = lds + slot * SLOT + c * 512 + it * 256 + wave * 64;  // wave-uniform base; lane * 4 is implicit
                __builtin_amdgcn_global_load_lds((const void*)(p + c * V + off[it]), (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
            }
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) issue(z0 + d, d % NS);
    int slot = 0;
    for (int z = z0; z < z0 + 32; ++z) {
        issue(z + PF, (slot + PF) % NS);
        // 6 LDS-DMA per plane per wave; leave the PF newest planes in flight
        if (PF == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if (PF == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        if (PF == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float* s = lds + slot * SLOT;
        const int ci = (ly + halo) * PX + lx + halo;
        const int64_t g = ((int64_t)z * N + oy + ly) * N + ox + lx;
        b[g] = s[ci];
        b[g + V] = s[512 + ci];
        b[g + 2 * V] = s[1024 + ci];
        slot = (slot + 1) % NS;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// tile-shape study: TX x TY threads, halo 1, register prefetch of one plane, SEG planes per block, optional XCD remap
template <int TX, int TY, int SEG, bool SWZ>
__global__ __launch_bounds__(TX * TY) void march_shape(const float* __restrict__ a, float* __restrict__ b, int N) {
    constexpr int halo = 1, PX = TX + 2, PY = TY + 2, PN = PX * PY, NT = TX * TY, NIT = (PN + NT - 1) / NT;
    __shared__ float lds[3 * PN];
    int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (SWZ) {
        const int total = gridDim.x * gridDim.y * gridDim.z, run = gridDim.x, group = 8 * run;
        if (total % group == 0) {
            const int g = bid / group, w = bid - g * group;
            bid = g * group + (w & 7) * run + (w >> 3);
        }
    }
    const int bx = bid % gridDim.x, by = (bid / gridDim.x) % gridDim.y, bz = bid / (gridDim.x * gridDim.y);
    const int ox = bx * TX, oy = by * TY, z0 = bz * SEG;
    const int64_t V = (int64_t)N * N * N, HW = (int64_t)N * N;
    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    unsigned off[NIT];
    bool val[NIT];
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * NT;
        const int px = i % PX, py = i / PX;
        const int cx = min(max(ox - halo + px, 0), N - 1), cy = min(max(oy - halo + py, 0), N - 1);
        val[it] = i < PN;
        off[it] = (unsigned)(cy * N + cx);
    }
    float pre[NIT][3];
    auto load = [&](int z) {
        const float* p = a + (int64_t)min(z, N - 1) * HW;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (val[it]) {
                pre[it][0] = p[off[it]];
                pre[it][1] = (p + V)[off[it]];
                pre[it][2] = (p + 2 * V)[off[it]];
            }
    };
    load(z0);
    for (int z = z0; z < z0 + SEG; ++z) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (val[it]) {
                const int i = threadIdx.x + it * NT;
                lds[i] = pre[it][0];
                lds[PN + i] = pre[it][1];
                lds[2 * PN + i] = pre[it][2];
            }
        load(z + 1);
        __syncthreads();
        const int ci = (ly + halo) * PX + lx + halo;
        const int64_t g = ((int64_t)z * N + oy + ly) * N + ox + lx;
        b[g] = lds[ci];
        b[g + V] = lds[PN + ci];
        b[g + 2 * V] = lds[2 * PN + ci];
        __syncthreads();
    }
}


// array-of-structures variant of the marching skeleton: one 16-byte (float4: x, y, z, pad) or 12-byte (3 floats) record per
// voxel instead of three planar dwords.  Rates are quoted on the 12 useful bytes per voxel and direction.
template <int TX, int TY, int NF>
__global__ __launch_bounds__(TX * TY) void march_aos(const float* __restrict__ a, float* __restrict__ b, int N) {
    constexpr int halo = 1, PX = TX + 2, PY = TY + 2, PN = PX * PY, NT = TX * TY, NIT = (PN + NT - 1) / NT;
    __shared__ float lds[NF * PN];
    const int ox = blockIdx.x * TX, oy = blockIdx.y * TY, z0 = blockIdx.z * 32;
    const int64_t HW = (int64_t)N * N;
    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    unsigned off[NIT];
    bool val[NIT];
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * NT;
        const int px = i % PX, py = i / PX;
        const int cx = min(max(ox - halo + px, 0), N - 1), cy = min(max(oy - halo + py, 0), N - 1);
        val[it] = i < PN;
        off[it] = (unsigned)(cy * N + cx);
    }
    float pre[NIT][NF];
    auto load = [&](int z) {
        const float* p = a + (int64_t)min(z, N - 1) * HW * NF;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (val[it]) {
                if (NF == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(p + (size_t)off[it] * 4);
                    pre[it][0] = v.x; pre[it][1] = v.y; pre[it][2] = v.z; pre[it][3] = v.w;
                } else {
#pragma unroll
                    for (int c = 0; c < NF; ++c) pre[it][c] = p[(size_t)off[it] * NF + c];
                }
            }
    };
    load(z0);
    for (int z = z0; z < z0 + 32; ++z) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (val[it]) {
                const int i = threadIdx.x + it * NT;
#pragma unroll
                for (int c = 0; c < NF; ++c) lds[c * PN + i] = pre[it][c];
            }
        load(z + 1);
        __syncthreads();
        const int ci = (ly + halo) * PX + lx + halo;
        float* q = b + ((int64_t)z * HW + (oy + ly) * N + ox + lx) * NF;
        if (NF == 4) *reinterpret_cast<float4*>(q) = make_float4(lds[ci], lds[PN + ci], lds[2 * PN + ci], lds[3 * PN + ci]);
        else {
#pragma unroll
            for (int c = 0; c < NF; ++c) q[c] = lds[c * PN + ci];
        }
        __syncthreads();
    }
}

int main() {
    const int N = 256;
    const int64_t V = (int64_t)N * N * N;
    float *a, *b;
    hipMalloc(&a, 3 * V * 4);
    hipMalloc(&b, 3 * V * 4);
    hipMemset(a, 0, 3 * V * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch, double bytes) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.1f us  %7.1f GB/s (algorithmic)\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e9);
    };
    const double bytes = 2.0 * 3 * V * 4;
    time("copy_flat<float>", [&] { hipLaunchKernelGGL(copy_flat<float>, dim3((3 * V + 255) / 256), dim3(256), 0, 0, a, b, 3 * V); }, bytes);
    time("copy_flat<float2>", [&] { hipLaunchKernelGGL(copy_flat<float2>, dim3((3 * V / 2 + 255) / 256), dim3(256), 0, 0, (const float2*)a, (float2*)b, 3 * V / 2); }, bytes);
    time("copy_flat<float4>", [&] { hipLaunchKernelGGL(copy_flat<float4>, dim3((3 * V / 4 + 255) / 256), dim3(256), 0, 0, (const float4*)a, (float4*)b, 3 * V / 4); }, bytes);
    time("copy3_dword (3 planar streams)", [&] { hipLaunchKernelGGL(copy3_dword, dim3((V + 255) / 256), dim3(256), 0, 0, a, b, V); }, bytes);
    time("march_skeleton halo 0", [&] { hipLaunchKernelGGL(march_skeleton, dim3(N / 32, N / 8, N / 32), dim3(256), 0, 0, a, b, N, 0); }, bytes);
    time("march_skeleton halo 1", [&] { hipLaunchKernelGGL(march_skeleton, dim3(N / 32, N / 8, N / 32), dim3(256), 0, 0, a, b, N, 1); }, bytes);
    time("march_prefetch<1> halo 1", [&] { hipLaunchKernelGGL(march_prefetch<1>, dim3(N / 32, N / 8, N / 32), dim3(256), 0, 0, a, b, N); }, bytes);
    time("march_prefetch<2> halo 1", [&] { hipLaunchKernelGGL(march_prefetch<2>, dim3(N / 32, N / 8, N / 32), dim3(256), 0, 0, a, b, N); }, bytes);
    time("march_glds<1> halo 1", [&] { hipLaunchKernelGGL(march_glds<1>, dim3(N / 32, N / 8, N / 32), dim3(256), 0, 0, a, b, N); }, bytes);
    time("march_glds<2> halo 1", [&] { hipLaunchKernelGGL(march_glds<2>, dim3(N / 32, N / 8, N / 32), dim3(256), 0, 0, a, b, N); }, bytes);
    time("march_glds<3> halo 1", [&] { hipLaunchKernelGGL(march_glds<3>, dim3(N / 32, N / 8, N / 32), dim3(256), 0, 0, a, b, N); }, bytes);
    time("march_shape<32,8,seg 32,swz 0>", [&] { hipLaunchKernelGGL((march_shape<32, 8, 32, false>), dim3(N / 32, N / 8, N / 32), dim3(256), 0, 0, a, b, N); }, bytes);
    time("march_shape<32,8,seg 32,swz 1>", [&] { hipLaunchKernelGGL((march_shape<32, 8, 32, true>), dim3(N / 32, N / 8, N / 32), dim3(256), 0, 0, a, b, N); }, bytes);
    time("march_shape<64,4,seg 32,swz 0>", [&] { hipLaunchKernelGGL((march_shape<64, 4, 32, false>), dim3(N / 64, N / 4, N / 32), dim3(256), 0, 0, a, b, N); }, bytes);
    time("march_shape<64,4,seg 32,swz 1>", [&] { hipLaunchKernelGGL((march_shape<64, 4, 32, true>), dim3(N / 64, N / 4, N / 32), dim3(256), 0, 0, a, b, N); }, bytes);
    time("march_shape<64,8,seg 32,swz 1>", [&] { hipLaunchKernelGGL((march_shape<64, 8, 32, true>), dim3(N / 64, N / 8, N / 32), dim3(512), 0, 0, a, b, N); }, bytes);
    time("march_shape<64,8,seg 64,swz 1>", [&] { hipLaunchKernelGGL((march_shape<64, 8, 64, true>), dim3(N / 64, N / 8, N / 64), dim3(512), 0, 0, a, b, N); }, bytes);
    time("march_shape<128,4,seg 32,swz 1>", [&] { hipLaunchKernelGGL((march_shape<128, 4, 32, true>), dim3(N / 128, N / 4, N / 32), dim3(512), 0, 0, a, b, N); }, bytes);
    time("march_shape<128,4,seg 64,swz 1>", [&] { hipLaunchKernelGGL((march_shape<128, 4, 64, true>), dim3(N / 128, N / 4, N / 64), dim3(512), 0, 0, a, b, N); }, bytes);
    time("march_shape<128,8,seg 64,swz 1>", [&] { hipLaunchKernelGGL((march_shape<128, 8, 64, true>), dim3(N / 128, N / 8, N / 64), dim3(1024), 0, 0, a, b, N); }, bytes);
    time("march_shape<256,4,seg 64,swz 1>", [&] { hipLaunchKernelGGL((march_shape<256, 4, 64, true>), dim3(N / 256, N / 4, N / 64), dim3(1024), 0, 0, a, b, N); }, bytes);
    time("march_shape<256,2,seg 32,swz 1>", [&] { hipLaunchKernelGGL((march_shape<256, 2, 32, true>), dim3(N / 256, N / 2, N / 32), dim3(512), 0, 0, a, b, N); }, bytes);
    time("march_shape<64,16,seg 64,swz 1>", [&] { hipLaunchKernelGGL((march_shape<64, 16, 64, true>), dim3(N / 64, N / 16, N / 64), dim3(1024), 0, 0, a, b, N); }, bytes);
    float *a4, *b4;
    hipMalloc(&a4, 4 * V * 4);
    hipMalloc(&b4, 4 * V * 4);
    hipMemset(a4, 0, 4 * V * 4);
    time("march_aos<32,8,float4>", [&] { hipLaunchKernelGGL((march_aos<32, 8, 4>), dim3(N / 32, N / 8, N / 32), dim3(256), 0, 0, a4, b4, N); }, bytes);
    time("march_aos<32,8,float3>", [&] { hipLaunchKernelGGL((march_aos<32, 8, 3>), dim3(N / 32, N / 8, N / 32), dim3(256), 0, 0, a4, b4, N); }, bytes);
    time("march_aos<64,8,float4>", [&] { hipLaunchKernelGGL((march_aos<64, 8, 4>), dim3(N / 64, N / 8, N / 32), dim3(512), 0, 0, a4, b4, N); }, bytes);
    time("march_aos<64,4,float3>", [&] { hipLaunchKernelGGL((march_aos<64, 4, 3>), dim3(N / 64, N / 4, N / 32), dim3(256), 0, 0, a4, b4, N); }, bytes);
    time("march_aos<64,8,float3>", [&] { hipLaunchKernelGGL((march_aos<64, 8, 3>), dim3(N / 64, N / 8, N / 32), dim3(512), 0, 0, a4, b4, N); }, bytes);
    time("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, 3 * V * 4, hipMemcpyDeviceToDevice, 0); }, bytes);
    return 0;
}
