// What does a 12-byte LDS-DMA (global_load_lds_dwordx3) lay down in LDS?  One wave copies 64 records of (3i, 3i+1, 3i+2) and
// the LDS image is dumped; a second pass masks the odd lanes off.
// Build: hipcc -O2 --offload-arch=gfx950 tools/probes/glds12_probe.hip -o gpurun_out/glds12_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void probe(const float* __restrict__ a, float* __restrict__ out, int mask_odd, int size) {
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.0f;
    __syncthreads();
    if (!mask_odd || (threadIdx.x & 1) == 0) {
        if (size == 12)
            __builtin_amdgcn_global_load_lds((const void*)(a + 3 * threadIdx.x), (__attribute__((address_space(3))) void*)(lds + 16), 12, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const void*)(a + 4 * threadIdx.x), (__attribute__((address_space(3))) void*)(lds + 16), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    float *a, *o, h[1024];
    hipMalloc(&a, 4096);
    hipMalloc(&o, 4096);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    hipMemcpy(a, h, 4096, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 4; ++pass) {
        const int size = pass < 2 ? 12 : 16;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, o, pass & 1, size);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(h, o, 4096, hipMemcpyDeviceToHost);
        printf("size %d mask_odd %d:", size, pass & 1);
        for (int i = 0; i < 300; ++i) printf(" %g", h[i]);
        printf("\n");
        for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    }
    return 0;
}
