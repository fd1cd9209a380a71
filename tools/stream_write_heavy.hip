// stream_write_heavy.hip -- yardstick for the linearize kernel's traffic shape (read 2 B, write 8 B per sample) and an
// elimination experiment: add the kernel's other ingredients one at a time and see which one costs bandwidth.
//   STAGE : every workgroup stages a 768-entry float2 table into LDS and barriers before its packet
//   GATHER: 8 random-index ds_read_b64 per thread feed the outputs
//   VALU  : ~18 dependent-free float ops per sample (the linearize arithmetic's volume)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT, bool STAGE, bool GATHER, int VALU>
__global__ __launch_bounds__(256) void k(const u4 *in, f4 *o1, f4 *o2, size_t nvec, const float *lut)
{
    __shared__ float2 tab[768];
    if (STAGE) {
        for (int i = threadIdx.x; i < 768; i += 256) tab[i] = make_float2(lut[i], lut[i < 767 ? i + 1 : i]);
        __syncthreads();
    }
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= nvec) return;
    const u4 v = in[i];
    float x[8] = {(float)(v.x & 0xffff), (float)(v.x >> 16), (float)(v.y & 0xffff), (float)(v.y >> 16),
                  (float)(v.z & 0xffff), (float)(v.z >> 16), (float)(v.w & 0xffff), (float)(v.w >> 16)};
    float y[8], z[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float s = x[e] * (255.0f / 65535.0f);
        float g0 = s, g1 = s + 1.0f;
        if (GATHER) { const float2 g = tab[((int)s) + 256 * (e % 3)]; g0 = g.x; g1 = g.y; }
        float fr = s - floorf(s);
        float a = g0 * (1.0f - fr) + g1 * fr;
        float b = (g1 - g0) * 255.0f * x[e] * 7.6e-7f;
#pragma unroll
        for (int r = 0; r < VALU; ++r) { a = __builtin_fmaf(a, 1.0000001f, 1e-9f); b = __builtin_fmaf(b, 0.9999999f, 1e-9f); }
        y[e] = a; z[e] = sqrtf(b * b);
    }
    f4 a0 = {y[0], y[1], y[2], y[3]}, a1 = {y[4], y[5], y[6], y[7]}, b0 = {z[0], z[1], z[2], z[3]}, b1 = {z[4], z[5], z[6], z[7]};
    if (NT) { __builtin_nontemporal_store(a0, &o1[2 * i]); __builtin_nontemporal_store(a1, &o1[2 * i + 1]); __builtin_nontemporal_store(b0, &o2[2 * i]); __builtin_nontemporal_store(b1, &o2[2 * i + 1]); }
    else { o1[2 * i] = a0; o1[2 * i + 1] = a1; o2[2 * i] = b0; o2[2 * i + 1] = b1; }
}
template <bool NT, bool STAGE, bool GATHER, int VALU>
void run(const char *name, const u4 *in, f4 *o1, f4 *o2, size_t nvec, const float *lut, size_t samples)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NT, STAGE, GATHER, VALU>), dim3((nvec + 255) / 256), dim3(256), 0, 0, in, o1, o2, nvec, lut);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<NT, STAGE, GATHER, VALU>), dim3((nvec + 255) / 256), dim3(256), 0, 0, in, o1, o2, nvec, lut);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-44s %.3f ms  %.2f TB/s (%.1f%% of 8 TB/s)\n", name, ms, samples * 10.0 / ms / 1e9, samples * 10.0 / ms / 1e9 / 8 * 100);
}
__global__ void fill(uint16_t *p, size_t n) { size_t i = blockIdx.x * (size_t)256 + threadIdx.x; if (i < n) { uint32_t g = (uint32_t)i * 2654435761u; g ^= g >> 16; g *= 2246822519u; p[i] = (uint16_t)(g >> 11); } }
int main()
{
    const size_t samples = (size_t)64 * 3 * 1080 * 1920, nvec = samples / 8;
    u4 *in; f4 *o1, *o2; float *lut; CK(hipMalloc(&in, samples * 2)); CK(hipMalloc(&o1, samples * 4)); CK(hipMalloc(&o2, samples * 4)); CK(hipMalloc(&lut, 768 * 4));
    hipLaunchKernelGGL(fill, dim3((samples + 255) / 256), dim3(256), 0, 0, (uint16_t *)in, samples);
    CK(hipMemset(lut, 0, 768 * 4));
    run<false, false, false, 0>("plain stores", in, o1, o2, nvec, lut, samples);
    run<true, false, false, 0>("non-temporal stores", in, o1, o2, nvec, lut, samples);
    run<false, true, false, 0>("plain + LUT staging/barrier", in, o1, o2, nvec, lut, samples);
    run<false, true, true, 0>("plain + staging + 8 LDS gathers", in, o1, o2, nvec, lut, samples);
    run<false, true, true, 4>("plain + staging + gathers + VALU x4", in, o1, o2, nvec, lut, samples);
    run<false, true, true, 8>("plain + staging + gathers + VALU x8", in, o1, o2, nvec, lut, samples);
    run<true, true, true, 8>("nt    + staging + gathers + VALU x8", in, o1, o2, nvec, lut, samples);
    run<false, false, false, 8>("plain + VALU x8 only", in, o1, o2, nvec, lut, samples);
    return 0;
}
