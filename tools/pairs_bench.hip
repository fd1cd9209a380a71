// pairs_bench.hip -- standalone timing harness for ct_pair_residual_fwd / ct_pair_residual_bwd on the C3 shape
// (64 x 2048 x 2048 x 3 uint16, 888 pairs).  Compiles the product source directly so -D switches can select
// experimental variants:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize
//                         tools/pairs_bench.hip clair_torch_amd/csrc/ct_api.cpp -o tools/pairs_bench
#include "../clair_torch_amd/csrc/ct_pairs.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__global__ void make_stack(uint16_t *out, int N, size_t per_image, double t0, double stops, double escale)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= per_image) return;
    uint32_t k = (uint32_t)q + 1237u * 1000003u;
    k = (k ^ (k >> 16)) * 0x45D9F3Bu;
    k = (k ^ (k >> 16)) * 0x45D9F3Bu;
    k ^= k >> 16;
    const double e = (double)k * escale;
    for (int n = 0; n < N; ++n) {
        const double t = t0 * exp2(n * stops);
        double lin = fmin(fmax(e * t, 0.0), 1.0);
        out[(size_t)n * per_image + q] = (uint16_t)lrint(pow(lin, 1.0 / 2.2) * 65535.0);
    }
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 64, S = argc > 2 ? atoi(argv[2]) : 2048, C = 3, L = 256, reps = 5;
    const size_t per_image = (size_t)C * S * S;
    uint16_t *stack;
    CHECK(hipMalloc(&stack, per_image * N * 2));
    const double t0 = 1e-3, stops = 0.125;
    const double tmid = sqrt(t0 * t0 * exp2((N - 1) * stops));
    hipLaunchKernelGGL(make_stack, dim3((per_image + 255) / 256), dim3(256), 0, 0, stack, N, per_image, t0, stops,
                       2.0 / tmid / 4294967296.0);
    CHECK(hipDeviceSynchronize());
    // pairs (triu order, ratio >= 0.25) and the partner CSR
    std::vector<int> pi, pj;
    std::vector<double> pr;
    for (int i = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j) {
            const double r = exp2((i - j) * stops);
            if (r >= 0.25) {
                pi.push_back(i);
                pj.push_back(j);
                pr.push_back(r);
            }
        }
    const int P = (int)pi.size();
    std::vector<int> off(N + 1, 0), ps, pp;
    for (int n = 0; n < N; ++n) {
        for (int p = 0; p < P; ++p) {
            if (pi[p] == n) {
                ps.push_back(pj[p]);
                pp.push_back(p);
            }
            if (pj[p] == n) {
                ps.push_back(pi[p]);
                pp.push_back(~p);
            }
        }
        off[n + 1] = (int)ps.size();
    }
    std::vector<float> lut((size_t)C * L);
    for (int c = 0; c < C; ++c)
        for (int k = 0; k < L; ++k) lut[(size_t)c * L + k] = powf((float)k / (L - 1), 2.2f + 0.2f * c);
    std::vector<double> coef((size_t)P * C);
    const int only_channel = argc > 3 ? atoi(argv[3]) : -1;  // >= 0: upstream gradient for that channel only
    for (size_t k = 0; k < coef.size(); ++k)
        coef[k] = (only_channel < 0 || (int)(k % C) == only_channel) ? 1e-7 * (1.0 + (k % 7)) : 0.0;
    int *d_i, *d_j, *d_off, *d_ps, *d_pp;
    double *d_r, *d_sums, *d_coef, *d_grad;
    float *d_lut;
    CHECK(hipMalloc(&d_i, P * 4));
    CHECK(hipMalloc(&d_j, P * 4));
    CHECK(hipMalloc(&d_r, P * 8));
    CHECK(hipMalloc(&d_off, (N + 1) * 4));
    CHECK(hipMalloc(&d_ps, ps.size() * 4));
    CHECK(hipMalloc(&d_pp, pp.size() * 4));
    CHECK(hipMalloc(&d_sums, (size_t)P * C * 5 * 8));
    CHECK(hipMalloc(&d_coef, (size_t)P * C * 8));
    CHECK(hipMalloc(&d_grad, (size_t)C * L * 8));
    CHECK(hipMalloc(&d_lut, (size_t)C * L * 4));
    CHECK(hipMemcpy(d_i, pi.data(), P * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_j, pj.data(), P * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_r, pr.data(), P * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_off, off.data(), (N + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ps, ps.data(), ps.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_pp, pp.data(), pp.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_coef, coef.data(), coef.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_lut, lut.data(), lut.size() * 4, hipMemcpyHostToDevice));
    const int64_t ws_bytes = ct_pair_residual_bwd_workspace(N, P, C);
    void *d_ws;
    CHECK(hipMalloc(&d_ws, ws_bytes));
    ct_geometry g{};
    g.channels = C;
    g.h_tile = g.h_global = S;
    g.width = S;
    g.image_stride = (int64_t)per_image;
    ct_icrf icrf{d_lut, L, CT_INTERP_LINEAR};
    ct_pair_params prm{};
    prm.lower = 1.0f / 255;
    prm.upper = 254.0f / 255;
    prm.weight_scale = 10.0f;
    prm.use_relative = 1;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // the backward twice: without the band hint (generic kernel) and with it (lane <-> sample kernel when eligible)
    int band = 0;
    for (int p = 0; p < P; ++p) band = std::max(band, pj[p] - pi[p]);
    std::vector<double> grad_ref;
    const char *pick = getenv("PAIRS_BENCH_VARIANT");  // "generic" / "lane": only that one (for rocprofv3 --pmc)
    for (int variant = 0; variant < 2; ++variant) {
        if (pick && ((variant == 0) != (pick[0] == 'g'))) continue;
        prm.pair_band = variant == 0 ? 0 : band;
        float ms_f = 0, ms_b = 0;
        for (int rep = 0; rep < reps + 1; ++rep) {
            CHECK(hipMemset(d_sums, 0, (size_t)P * C * 5 * 8));
            CHECK(hipMemset(d_grad, 0, (size_t)C * L * 8));
            CHECK(hipEventRecord(e0));
            int rc = ct_pair_residual_fwd(stack, CT_DTYPE_U16, 65535.0f, N, &g, nullptr, &icrf, d_i, d_j, d_r, P, &prm, 0,
                                          nullptr, d_sums, nullptr);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float t;
            CHECK(hipEventElapsedTime(&t, e0, e1));
            if (rc) fprintf(stderr, "fwd rc=%d\n", rc);
            if (rep) ms_f += t;
            CHECK(hipEventRecord(e0));
            rc = ct_pair_residual_bwd(stack, CT_DTYPE_U16, 65535.0f, N, &g, nullptr, &icrf, d_r, P, d_off, d_ps, d_pp, &prm,
                                      d_coef, nullptr, d_grad, d_ws, ws_bytes, nullptr);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&t, e0, e1));
            if (rc) fprintf(stderr, "bwd rc=%d\n", rc);
            if (rep) ms_b += t;
        }
        CHECK(hipGetLastError());
        std::vector<double> sums((size_t)P * C * 5), grad((size_t)C * L);
        CHECK(hipMemcpy(sums.data(), d_sums, sums.size() * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(grad.data(), d_grad, grad.size() * 8, hipMemcpyDeviceToHost));
        double cs = 0, cg = 0;
        for (double v : sums) cs += v;
        for (double v : grad) cg += fabs(v);
        printf("N=%d S=%d P=%d band hint %2d:  fwd %.3f ms  bwd %.3f ms   (checksums: sums %.9e  |grad| %.9e)\n", N, S, P,
               prm.pair_band, ms_f / reps, ms_b / reps, cs, cg);
        if (variant == 0) {
            grad_ref = grad;
        } else if (!grad_ref.empty()) {
            double num = 0, den = 0, worst = 0;
            for (size_t k = 0; k < grad.size(); ++k) {
                num += (grad[k] - grad_ref[k]) * (grad[k] - grad_ref[k]);
                den += grad_ref[k] * grad_ref[k];
                if (grad_ref[k] != 0.0) worst = fmax(worst, fabs(grad[k] - grad_ref[k]) / fabs(grad_ref[k]));
            }
            printf("lane vs generic LUT gradient: norm-wise %.3g, worst element %.3g\n", sqrt(num / fmax(den, 1e-300)), worst);
        }
    }
    return 0;
}
