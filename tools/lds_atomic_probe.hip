// LDS atomic throughput on gfx950: ds_add_f32 vs ds_add_u32 vs ds_add_u64 (distinct addresses per lane, no conflicts).
// hipcc --offload-arch=gfx950 -O3 -o tools/lds_atomic_probe tools/lds_atomic_probe.hip && ./tools/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <typename T>
__global__ __launch_bounds__(256) void probe(T* out, int iters) {
    __shared__ T acc[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) acc[i] = (T)0;
    __syncthreads();
    const T v = (T)(threadIdx.x + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) atomicAdd(&acc[(threadIdx.x + 256 * j + it) & 4095], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[blockIdx.x & 4095];
}
template <typename T>
void run(const char* name) {
    T* out;
    hipMalloc(&out, sizeof(T) * 4096);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int blocks = 1024, iters = 200;
    probe<T><<<blocks, 256>>>(out, 10);
    hipEventRecord(a);
    probe<T><<<blocks, 256>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double lane_ops = (double)blocks * 256 * iters * 16;
    printf("%-10s %.3f ms  %.1f G lane-atomics/s  = %.2f lanes/clk/CU (256 CUs, 2.1 GHz)\n", name, ms, lane_ops / ms / 1e6, lane_ops / (ms * 1e-3) / 256 / 2.1e9);
    hipFree(out);
}
int main() {
    run<float>("f32");
    run<unsigned>("u32");
    run<unsigned long long>("u64");
    run<int>("i32");
    return 0;
}
