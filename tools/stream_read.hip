// stream_read.hip -- HBM read ceilings on MI355X for the access patterns of the merge kernel:
//   linear   : one sweep over a 3.2 GB buffer, every workgroup reads consecutive 4 KB chunks (grid-stride)
//   streams32: the merge pattern -- 32 exposures 100 MB apart, a workgroup reads the same 2/4/8 KB slice of each
//   persistent variants of both (grid = resident workgroups)
//   hipcc --offload-arch=gfx950 -O3 -w tools/stream_read.hip -o tools/stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int BYTES>
struct Pk;
template <> struct Pk<8> { uint2 v; };
template <> struct Pk<16> { uint4 v; };

template <int BYTES>
__device__ __forceinline__ uint32_t fold(const Pk<BYTES> &p)
{
    if constexpr (BYTES == 8) return p.v.x ^ p.v.y;
    else return p.v.x ^ p.v.y ^ p.v.z ^ p.v.w;
}

// every thread: one packet per exposure, N exposures `stride` bytes apart; PER = packets per thread (adjacent slices)
template <int BYTES, int PER>
__global__ __launch_bounds__(256) void streams(const char *base, size_t stride, int n, size_t packets, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t p0 = ((size_t)blockIdx.x * PER) * 256 + threadIdx.x; p0 < packets; p0 += (size_t)gridDim.x * PER * 256) {
#pragma unroll 4
        for (int e = 0; e < n; ++e) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const size_t p = p0 + (size_t)k * 256;
                if (p < packets) acc ^= fold<BYTES>(*reinterpret_cast<const Pk<BYTES> *>(base + (size_t)e * stride + p * BYTES));
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int BYTES>
__global__ __launch_bounds__(256) void linear(const char *base, size_t packets, uint32_t *out)
{
    uint32_t acc = 0;
#pragma unroll 8
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < packets; p += (size_t)gridDim.x * 256)
        acc ^= fold<BYTES>(*reinterpret_cast<const Pk<BYTES> *>(base + p * BYTES));
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
void timeit(const char *name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-44s %.3f ms  %.2f TB/s\n", name, best, bytes / best / 1e9);
}

int main()
{
    const int n = 32;
    const size_t per = (size_t)3 * 4096 * 4096 * 2, total = per * n;  // the C2 stack
    char *buf;
    uint32_t *out;
    hipMalloc(&buf, total);
    hipMalloc(&out, 64);
    hipMemset(buf, 1, total);
    const double B = (double)total;
    for (int grid : {2048, 4096, 16384, 1 << 20}) {
        char nm[128];
        snprintf(nm, sizeof nm, "linear 16B/lane grid %d", grid);
        const size_t pk = total / 16;
        const int g = (int)((size_t)grid < (pk + 255) / 256 ? grid : (pk + 255) / 256);
        timeit(nm, B, [&] { hipLaunchKernelGGL(linear<16>, dim3(g), dim3(256), 0, 0, buf, pk, out); });
        snprintf(nm, sizeof nm, "linear 8B/lane grid %d", grid);
        const size_t pk8 = total / 8;
        const int g8 = (int)((size_t)grid < (pk8 + 255) / 256 ? grid : (pk8 + 255) / 256);
        timeit(nm, B, [&] { hipLaunchKernelGGL(linear<8>, dim3(g8), dim3(256), 0, 0, buf, pk8, out); });
    }
    {
        const size_t pk16 = per / 16, pk8 = per / 8;
        timeit("32 streams 16B/lane, 4 KB per WG, full grid", B, [&] { hipLaunchKernelGGL((streams<16, 1>), dim3((pk16 + 255) / 256), dim3(256), 0, 0, buf, per, n, pk16, out); });
        timeit("32 streams 8B/lane, 2 KB per WG, full grid", B, [&] { hipLaunchKernelGGL((streams<8, 1>), dim3((pk8 + 255) / 256), dim3(256), 0, 0, buf, per, n, pk8, out); });
        timeit("32 streams 8B/lane, 2x2 KB per WG, full grid", B, [&] { hipLaunchKernelGGL((streams<8, 2>), dim3((pk8 + 511) / 512), dim3(256), 0, 0, buf, per, n, pk8, out); });
        timeit("32 streams 8B/lane, 4x2 KB per WG, full grid", B, [&] { hipLaunchKernelGGL((streams<8, 4>), dim3((pk8 + 1023) / 1024), dim3(256), 0, 0, buf, per, n, pk8, out); });
        timeit("32 streams 16B/lane, 2x4 KB per WG, full grid", B, [&] { hipLaunchKernelGGL((streams<16, 2>), dim3((pk16 + 511) / 512), dim3(256), 0, 0, buf, per, n, pk16, out); });
        timeit("32 streams 8B/lane, persistent 2048", B, [&] { hipLaunchKernelGGL((streams<8, 1>), dim3(2048), dim3(256), 0, 0, buf, per, n, pk8, out); });
        timeit("32 streams 16B/lane, persistent 2048", B, [&] { hipLaunchKernelGGL((streams<16, 1>), dim3(2048), dim3(256), 0, 0, buf, per, n, pk16, out); });
        timeit("32 streams 8B/lane, persistent 4096", B, [&] { hipLaunchKernelGGL((streams<8, 1>), dim3(4096), dim3(256), 0, 0, buf, per, n, pk8, out); });
    }
    return 0;
}
