// valu_rates2.hip -- issue cost of the exact instructions in the merge kernel's inner loop (inline asm, wave64,
// 8 waves/SIMD resident, independent destination registers), plus LDS gather cost with random / coherent indices.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITERS = 2048;

#define REP8(S) S S S S S S S S
template <int OP> __global__ __launch_bounds__(256) void k(float *out, float seed, const unsigned *idx)
{
    float a = seed + threadIdx.x * 0.001f, b = 1.0000001f, c = 1e-9f, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    double d0 = a, d1 = a + 1, d2 = 0, d3 = 0;
    unsigned u = __float_as_uint(a) & 0xffff, w0 = 0, w1 = 0;
    __shared__ float2 tab[768];
    for (int i = threadIdx.x; i < 768; i += 256) tab[i] = make_float2(i, i + 1);
    __syncthreads();
    unsigned ia = idx[threadIdx.x] * 8, ib = idx[threadIdx.x + 256] * 8;
    float2 g0 = {0, 0}, g1 = {0, 0};
    for (int i = 0; i < ITERS; ++i) {
        if constexpr (OP == 0) { REP8(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a), "v"(b), "v"(c));) }
        if constexpr (OP == 1) { REP8(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d2) : "v"(a));) }
        if constexpr (OP == 2) { REP8(asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d3) : "v"(d0), "v"(d1), "v"(d0));) }
        if constexpr (OP == 3) { REP8(asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r1) : "v"(u));) }
        if constexpr (OP == 4) { REP8(asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(w0) : "v"(a));) }
        if constexpr (OP == 5) { REP8(asm volatile("v_fract_f32 %0, %1" : "=v"(r2) : "v"(a));) }
        if constexpr (OP == 6) { REP8(asm volatile("v_exp_f32 %0, %1" : "=v"(r3) : "v"(a));) }
        if constexpr (OP == 7) { REP8(asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(w1) : "v"(u), "v"(u));) }
        if constexpr (OP == 8) { REP8(asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(a), "v"(b));) }
        if constexpr (OP == 9) { REP8(asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(r1) : "v"(u));) }
        if constexpr (OP == 10) { REP8(asm volatile("v_rcp_f32 %0, %1" : "=v"(r3) : "v"(a));) }
        if constexpr (OP == 11) {  // LDS gather, per-lane random index (worst case)
            REP8(asm volatile("ds_read_b64 %0, %1" : "=v"(g0) : "v"(ia)); asm volatile("ds_read_b64 %0, %1" : "=v"(g1) : "v"(ib));)
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        if constexpr (OP == 12) {  // LDS gather, all lanes same index (broadcast)
            unsigned z = 64;
            REP8(asm volatile("ds_read_b64 %0, %1" : "=v"(g0) : "v"(z)); asm volatile("ds_read_b64 %0, %1" : "=v"(g1) : "v"(z));)
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        if constexpr (OP == 13) { REP8(asm volatile("v_add_f64 %0, %1, %2" : "=v"(d3) : "v"(d0), "v"(d1));) }
        if constexpr (OP == 14) { REP8(asm volatile("v_sqrt_f32 %0, %1" : "=v"(r3) : "v"(a));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + (float)(d2 + d3) + w0 + w1 + g0.x + g1.y;
}

template <int OP> void run(const char *name, double per_iter, float *out, const unsigned *idx)
{
    const int blocks = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, idx);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, idx);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double wave_insts = (double)blocks * 4 * ITERS * per_iter;
    const double per_simd_per_us = wave_insts / 1024.0 / (ms * 1e3);
    printf("%-36s %.3f ms  => %.2f cycles per wave-instr per SIMD at 2.4 GHz (%.2f at 2.1)\n", name, ms, 2400.0 / per_simd_per_us, 2100.0 / per_simd_per_us);
}

int main()
{
    float *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    unsigned h[512]; for (int i = 0; i < 512; ++i) h[i] = (unsigned)(rand() % 768);
    unsigned *idx; CK(hipMalloc(&idx, sizeof(h))); CK(hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice));
    run<0>("v_fma_f32", 8, out, idx);
    run<8>("v_mul_f32", 8, out, idx);
    run<1>("v_cvt_f64_f32", 8, out, idx);
    run<2>("v_fma_f64", 8, out, idx);
    run<13>("v_add_f64", 8, out, idx);
    run<3>("v_cvt_f32_u32_sdwa WORD_1", 8, out, idx);
    run<9>("v_cvt_f32_u32", 8, out, idx);
    run<4>("v_cvt_i32_f32", 8, out, idx);
    run<5>("v_fract_f32", 8, out, idx);
    run<6>("v_exp_f32", 8, out, idx);
    run<10>("v_rcp_f32", 8, out, idx);
    run<14>("v_sqrt_f32", 8, out, idx);
    run<7>("v_lshl_add_u32", 8, out, idx);
    run<11>("ds_read_b64 random idx (per CU: /4)", 16, out, idx);
    run<12>("ds_read_b64 broadcast", 16, out, idx);
    return 0;
}
