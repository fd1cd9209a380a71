// lds_atomic_rates.hip -- cost of LDS atomic adds on gfx950 (nominal 2.4 GHz cycles per wave-instruction per CU),
// by type, by address pattern (random histogram bins, lane-linear, one address) and by replication factor R
// (R interleaved copies of the histogram: address = bin * R + lane % R, so lanes of different residue never share a
// bank when R = 32).   hipcc --offload-arch=gfx950 -O3 -w tools/lds_atomic_rates.hip -o tools/lds_atomic_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kBins = 1024, kIters = 2048;

template <typename T, int PATTERN, int R>
__global__ __launch_bounds__(1024) void k(T *out, const uint32_t *idx)
{
    extern __shared__ char raw[];
    T *hist = reinterpret_cast<T *>(raw);
    for (int i = threadIdx.x; i < kBins * R + 64; i += blockDim.x) hist[i] = T(0);
    __syncthreads();
    uint32_t h = idx[threadIdx.x + blockIdx.x * blockDim.x] + threadIdx.x * 2654435761u;
    const uint32_t rep = threadIdx.x & (R - 1);
    for (int it = 0; it < kIters; ++it) {
        uint32_t a;
        if (PATTERN == 0) {  // random bin per lane per iteration
            h = h * 1664525u + 1013904223u;
            a = ((h >> 10) & (kBins - 1)) * R + rep;
        } else if (PATTERN == 1) {
            a = (threadIdx.x & 63) + (it & 511);
        } else {
            a = it & 511;
        }
        atomicAdd(&hist[a], T(1));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += blockDim.x) out[blockIdx.x * kBins + i] = hist[i * R];
}

template <typename T, int PATTERN, int R>
void run(const char *name, int threads)
{
    const int blocks = 256;
    T *out;
    uint32_t *idx;
    hipMalloc(&out, sizeof(T) * kBins * blocks);
    hipMalloc(&idx, 4 * 1024 * blocks);
    hipMemset(idx, 0x5a, 4 * 1024 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = sizeof(T) * (kBins * R + 64);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<T, PATTERN, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k<T, PATTERN, R>), dim3(blocks), dim3(threads), lds, 0, out, idx);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<T, PATTERN, R>), dim3(blocks), dim3(threads), lds, 0, out, idx);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = threads / 64.0;
    printf("%-22s R=%2d threads %4d  %.3f ms  %6.1f cycles per wave-atomic per CU\n", name, R, threads, ms,
           ms * 1e-3 * 2.4e9 / (kIters * waves));
    hipFree(out);
    hipFree(idx);
}

int main()
{
    const int threads = 1024;
    run<float, 1, 1>("f32 lane-linear", threads);
    run<float, 2, 1>("f32 same address", threads);
    run<unsigned int, 1, 1>("u32 lane-linear", threads);
    run<unsigned int, 2, 1>("u32 same address", threads);
    run<unsigned long long, 1, 1>("u64 lane-linear", threads);
    run<double, 1, 1>("f64 lane-linear", threads);
    run<double, 2, 1>("f64 same address", threads);
    run<float, 0, 1>("f32 random", threads);
    run<float, 0, 4>("f32 random", threads);
    run<float, 0, 16>("f32 random", threads);
    run<float, 0, 32>("f32 random", threads);
    run<unsigned int, 0, 1>("u32 random", threads);
    run<unsigned int, 0, 2>("u32 random", threads);
    run<unsigned int, 0, 4>("u32 random", threads);
    run<unsigned int, 0, 8>("u32 random", threads);
    run<unsigned int, 0, 16>("u32 random", threads);
    run<unsigned int, 0, 32>("u32 random", threads);
    run<unsigned long long, 0, 1>("u64 random", threads);
    run<unsigned long long, 0, 4>("u64 random", threads);
    run<unsigned long long, 0, 16>("u64 random", threads);
    run<double, 0, 1>("f64 random", threads);
    run<double, 0, 2>("f64 random", threads);
    run<double, 0, 4>("f64 random", threads);
    run<double, 0, 8>("f64 random", threads);
    run<double, 0, 16>("f64 random", threads);
    return 0;
}
