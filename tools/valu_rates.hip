// valu_rates.hip -- development microbenchmark: sustained per-CU issue rate of the VALU instructions the merge
// kernel uses (wave64, many waves per SIMD, independent chains), to price design choices (packed f32? f64 moments?).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;

template <int OP> __global__ __launch_bounds__(256) void k(float *out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0000001f, c = 1e-9f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float2v pb = {b, b}, pc = {c, c};
    for (int i = 0; i < ITERS; ++i) {
        if constexpr (OP == 0) {  // v_fma_f32 x8
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
            a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
        } else if constexpr (OP == 1) {  // v_pk_fma_f32 x4 (8 lanes-values)
            p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc);
            p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc);
        } else if constexpr (OP == 2) {  // v_fma_f64 x4
            d0 = __builtin_fma(d0, 1.0000001, 1e-9); d1 = __builtin_fma(d1, 1.0000001, 1e-9);
            d2 = __builtin_fma(d2, 1.0000001, 1e-9); d3 = __builtin_fma(d3, 1.0000001, 1e-9);
        } else if constexpr (OP == 3) {  // v_exp_f32 x8
            a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
            a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
        } else if constexpr (OP == 4) {  // v_cvt_f64_f32 + v_cvt_f32_f64 x4 (2 cvt each)
            d0 = (double)a0; a0 = (float)d0 * b; d1 = (double)a1; a1 = (float)d1 * b; d2 = (double)a2; a2 = (float)d2 * b; d3 = (double)a3; a3 = (float)d3 * b;
        } else if constexpr (OP == 5) {  // v_mul_f32 x8
            a0 *= b; a1 *= b; a2 *= b; a3 *= b; a4 *= b; a5 *= b; a6 *= b; a7 *= b;
        } else if constexpr (OP == 6) {  // v_fract_f32 x8
            a0 = __builtin_amdgcn_fractf(a0) + 1.5f; a1 = __builtin_amdgcn_fractf(a1) + 1.5f; a2 = __builtin_amdgcn_fractf(a2) + 1.5f; a3 = __builtin_amdgcn_fractf(a3) + 1.5f;
            a4 = __builtin_amdgcn_fractf(a4) + 1.5f; a5 = __builtin_amdgcn_fractf(a5) + 1.5f; a6 = __builtin_amdgcn_fractf(a6) + 1.5f; a7 = __builtin_amdgcn_fractf(a7) + 1.5f;
        } else if constexpr (OP == 7) {  // v_pk_mul_f32 x4
            p0 *= pb; p1 *= pb; p2 *= pb; p3 *= pb;
        } else if constexpr (OP == 8) {  // v_pk_add_f32 x4
            p0 += pc; p1 += pc; p2 += pc; p3 += pc;
        } else if constexpr (OP == 9) {  // v_cvt_i32_f32 + v_cvt_f32_i32 x8
            a0 = (float)(int)a0 + 0.5f; a1 = (float)(int)a1 + 0.5f; a2 = (float)(int)a2 + 0.5f; a3 = (float)(int)a3 + 0.5f;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int OP> void run(const char *name, double insts_per_iter, float *out)
{
    const int blocks = 256 * 8;  // 8 blocks of 256 per CU -> 8 waves per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double wave_insts = (double)blocks * 4 * ITERS * insts_per_iter;  // per launch
    const double per_simd_per_us = wave_insts / 1024.0 / (ms * 1e3);
    printf("%-34s %.3f ms  %.1f wave-instr/us/SIMD  => %.2f cycles per wave-instr at 2.4 GHz\n", name, ms, per_simd_per_us, 2400.0 / per_simd_per_us);
}

int main()
{
    float *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0>("v_fma_f32 (8/iter)", 8, out);
    run<1>("v_pk_fma_f32 (4/iter, 8 values)", 4, out);
    run<2>("v_fma_f64 (4/iter)", 4, out);
    run<3>("v_exp_f32 (8/iter)", 8, out);
    run<4>("cvt f64<->f32 + mul (12/iter)", 12, out);
    run<5>("v_mul_f32 (8/iter)", 8, out);
    run<6>("v_fract_f32 + add (16/iter)", 16, out);
    run<7>("v_pk_mul_f32 (4/iter, 8 values)", 4, out);
    run<8>("v_pk_add_f32 (4/iter, 8 values)", 4, out);
    run<9>("cvt i32<->f32 + add x4 (12/iter)", 12, out);
    return 0;
}
