// valu_occupancy.hip -- VALU issue cost (cycles per wave-instruction per SIMD) as a function of the number of
// wavefronts per SIMD, for v_fma_f32, v_pk_fma_f32, v_fma_f64 and a mul/cvt/exp mix: is the ~4 cycles per instruction
// seen in the real kernels (3-4 waves per SIMD) a property of low occupancy, and do packed ops escape it?
//   hipcc --offload-arch=gfx950 -O3 -w tools/valu_occupancy.hip -o tools/valu_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 4096;

template <int OP>
__global__ void k(float *out, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b0 = seed, b1 = seed + 1, b2 = seed + 2, b3 = seed + 3, b4 = seed + 4, b5 = seed + 5, b6 = seed + 6, b7 = seed + 7;
    const float m = 1.0000001f, c = 1e-9f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
    const f2 pm = {m, m}, pc = {c, c};
    double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7;
    const double dm = 1.0000001, dc = 1e-9;
    for (int it = 0; it < kIters; ++it) {
        if (OP == 0) {  // 8 independent scalar FMAs
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (OP == 1) {  // 8 independent packed FMAs (16 FMAs)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
        } else if (OP == 2) {  // 8 independent double FMAs
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dm), "v"(dc));
        } else {  // a dependent-ish mix like the kernels': mul, fma, exp, cvt, med3, add (8 instructions, 2 chains)
            asm volatile("v_mul_f32 %0, %0, %4\n v_fma_f32 %1, %0, %4, %5\n v_exp_f32 %2, %1\n v_cvt_i32_f32 %3, %2\n"
                         "v_mul_f32 %0, %2, %4\n v_med3_f32 %1, %0, %5, %4\n v_add_f32 %2, %1, %5\n v_fract_f32 %3, %2\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x +
                                                 p7.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

template <int OP>
void run(const char *name, int waves_per_simd)
{
    const int threads = 256, blocks = 256 * waves_per_simd;  // 4 waves per block -> one per SIMD; blocks per CU = waves per SIMD
    float *out;
    hipMalloc(&out, (size_t)threads * blocks * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // every SIMD executes waves_per_simd waves x kIters x 8 instructions
    const double instr = (double)waves_per_simd * kIters * 8;
    printf("%-14s %d waves/SIMD  %.3f ms  %.2f nominal (2.4 GHz) cycles per wave-instruction per SIMD\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / instr);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 3, 4, 8}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<2>("v_fma_f64", w);
        run<3>("mixed chain", w);
    }
    return 0;
}
