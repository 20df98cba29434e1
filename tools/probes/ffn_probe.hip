// ffn_probe.hip - the fused feed-forward kernel (csrc/ffn_kernels.hip; modes 3 = balanced tail tiles, 4 = 64-token tiles only; modes 0-2: the
// shared-stage first form, ffn_shared_stage_variant.hip) against linear1 -> ReLU -> linear2 as two launches of
// conv2d_nhwc_glds: bit-for-bit comparison and time per shape.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/ffn_probe tools/probes/ffn_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#include "../../objectpermanence_amd/csrc/conv_kernels.hip"
#include "../../objectpermanence_amd/csrc/ffn_kernels.hip"
#include "ffn_shared_stage_variant.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void fill(float *d, size_t n, float scale)
{
    std::vector<float> h(n);
    for (auto &v : h) v = scale * (float)((rand() & 2047) - 1024) / 1024.f;
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
}

static void gemm(const float *A, const float *W, const float *b, float *C, int M, int N, int K, int relu, bool narrow)
{
    ConvArgs c = {};
    c.X = A; c.Wt = W; c.bias = b; c.Y = C;
    c.N = 1; c.H = 1; c.W = M; c.Cin = K; c.Cout = N; c.KH = 1; c.KW = 1; c.stride = 1; c.pad = 0; c.OH = 1; c.OW = M; c.KP = K; c.relu = relu;
    const unsigned gx = (unsigned)((M + 127) / 128);
    if (narrow) conv2d_nhwc_glds<64, 3><<<dim3(gx, (N + 63) / 64, 1), 256>>>(c);
    else conv2d_nhwc_glds<128, 3><<<dim3(gx, (N + 127) / 128, 1), 256>>>(c);
}

static void fused(const FfnArgs &a0, int mode, int slots)
{
    // mode 0: all BM = 64; 1: all BM = 32; 2: full rounds of BM = 64, the rest as BM = 32
    FfnArgs a = a0;
    const int M = a.M;
    if (mode == 0) { a.m_begin = 0; ffn_fused_glds<64><<<(M + 63) / 64, 256>>>(a); return; }
    if (mode == 1) { a.m_begin = 0; ffn_fused_glds<32><<<(M + 31) / 32, 256>>>(a); return; }
    if (mode == 3) { unsigned g; a.m_begin = 0; ffn_w8_plan(M, 256, &a, &g); ffn_fused_w8<<<g, 512>>>(a); return; }
    if (mode == 4) { a.m_begin = 0; a.n_full = (M + 63) / 64; a.tail_frags = 4; ffn_fused_w8<<<a.n_full, 512>>>(a); return; }
    const int n64 = M / 64, full = n64 / slots * slots;
    if (full) { a.m_begin = 0; ffn_fused_glds<64><<<full, 256>>>(a); }
    const int rest = M - full * 64;
    if (rest) { a.m_begin = full * 64; ffn_fused_glds<32><<<(rest + 31) / 32, 256>>>(a); }
}

int main(int argc, char **argv)
{
    const int F = 2048, E = 256;
    int occ64 = 0, occ32 = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ64, ffn_fused_glds<64>, 256, 0));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ32, ffn_fused_glds<32>, 256, 0));
    int occ8 = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ8, ffn_fused_w8, 512, 0));
    printf("resident workgroups per CU: BM=64 %d, BM=32 %d, eight-wave %d\n", occ64, occ32, occ8);
    const int slots = occ64 * 256;
    const int Ms[] = {300, 4800, 19200, 38400, 76800, 76800 + 17, 153600};
    for (int M : Ms) {
        float *X, *W1, *b1, *W2, *b2, *H, *Y0, *Y1;
        CK(hipMalloc(&X, (size_t)M * E * 4)); CK(hipMalloc(&W1, (size_t)F * E * 4)); CK(hipMalloc(&b1, F * 4));
        CK(hipMalloc(&W2, (size_t)F * E * 4)); CK(hipMalloc(&b2, E * 4)); CK(hipMalloc(&H, (size_t)M * F * 4));
        CK(hipMalloc(&Y0, (size_t)M * E * 4)); CK(hipMalloc(&Y1, (size_t)M * E * 4));
        srand(M);
        fill(X, (size_t)M * E, 1.f); fill(W1, (size_t)F * E, 0.06f); fill(b1, F, 0.1f); fill(W2, (size_t)F * E, 0.03f); fill(b2, E, 0.1f);
        FfnArgs a = {X, W1, b1, W2, b2, Y1, M, F, 0};
#ifdef FFN_TRACE
        unsigned long long *clk;
        CK(hipMalloc(&clk, 64)); CK(hipMemset(clk, 0, 64));
        a.clk = clk;
#endif
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](auto fn) {
            std::vector<float> ts;
            for (int r = 0; r < 7; ++r) {
                CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            return ts[ts.size() / 2];
        };
        const float t_ref = timeit([&] { gemm(X, W1, b1, H, M, F, E, 1, true); gemm(H, W2, b2, Y0, M, E, F, 0, false); });
        CK(hipGetLastError());
        std::vector<float> y0((size_t)M * E), y1((size_t)M * E);
        CK(hipMemcpy(y0.data(), Y0, y0.size() * 4, hipMemcpyDeviceToHost));
        const double gf = 4.0 * M * E * F * 1e-9;
        printf("M = %6d: two launches %.3f ms (%.1f TF)", M, t_ref, gf / t_ref);
        for (int mode = (argc > 1 ? 3 : 0); mode < 5; ++mode) {
            CK(hipMemset(Y1, 0xff, (size_t)M * E * 4));
            const float t = timeit([&] { fused(a, mode, slots); });
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(y1.data(), Y1, y1.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0; double maxd = 0;
            for (size_t k = 0; k < y0.size(); ++k) {
                if (memcmp(&y0[k], &y1[k], 4)) { ++bad; maxd = std::max(maxd, (double)fabsf(y0[k] - y1[k])); }
            }
            printf(" | fused mode %d %.3f ms (%.1f TF) %s", mode, t, gf / t, bad ? "DIFFERENT" : "bit-identical");
            if (bad) printf(" (%zu words, max %.3g)", bad, maxd);
        }
#ifdef FFN_TRACE
        {
            unsigned long long h[3];
            CK(hipMemcpy(h, clk, 24, hipMemcpyDeviceToHost));
            if (h[2]) printf(" | eight-wave tiles: %.0f shader ticks, %.2f us each -> %.3f GHz if a tick is a shader cycle", (double)h[0] / h[2], (double)h[1] / h[2] * 0.01,
                             (double)h[0] / ((double)h[1] * 10.0));
        }
#endif
        printf("\n");
        CK(hipFree(X)); CK(hipFree(W1)); CK(hipFree(b1)); CK(hipFree(W2)); CK(hipFree(b2)); CK(hipFree(H)); CK(hipFree(Y0)); CK(hipFree(Y1));
    }
    // ---- the single K = 256 product (gemm_k256_w8) against conv2d_nhwc_glds<64, 3> / <128, 3> ----
    const int Ns[] = {768, 2048, 256};
    for (int N : Ns)
        for (int M : {300, 19200, 76800, 76817, 153600}) {
            float *X, *W, *b, *Y0, *Y1;
            CK(hipMalloc(&X, (size_t)M * E * 4)); CK(hipMalloc(&W, (size_t)N * E * 4)); CK(hipMalloc(&b, N * 4));
            CK(hipMalloc(&Y0, (size_t)M * N * 4)); CK(hipMalloc(&Y1, (size_t)M * N * 4));
            srand(M + N);
            fill(X, (size_t)M * E, 1.f); fill(W, (size_t)N * E, 0.06f); fill(b, N, 0.1f);
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto timeit = [&](auto fn) {
                std::vector<float> ts;
                for (int r = 0; r < 7; ++r) {
                    CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
                }
                std::sort(ts.begin(), ts.end());
                return ts[ts.size() / 2];
            };
            const float t_n = timeit([&] { gemm(X, W, b, Y0, M, N, E, 1, true); });
            const float t_w = timeit([&] { gemm(X, W, b, Y0, M, N, E, 1, false); });
            Gemm256Args g = {X, W, b, Y1, M, N, 1, 0, 0};
            FfnArgs plan = {};
            unsigned grid;
            ffn_w8_plan(M, 256, &plan, &grid);
            g.n_full = plan.n_full; g.tail_frags = plan.tail_frags;
            CK(hipMemset(Y1, 0xff, (size_t)M * N * 4));
            const float t_g = timeit([&] { gemm_k256_w8<<<grid, 512>>>(g); });
            CK(hipGetLastError()); CK(hipDeviceSynchronize());
            std::vector<float> y0((size_t)M * N), y1((size_t)M * N);
            CK(hipMemcpy(y0.data(), Y0, y0.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(y1.data(), Y1, y1.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t k = 0; k < y0.size(); ++k) bad += memcmp(&y0[k], &y1[k], 4) != 0;
            const double gf = 2.0 * M * E * N * 1e-9;
            printf("K = 256 product M = %6d N = %4d: 128 x 64 tiles %.3f ms (%.1f TF), 128 x 128 tiles %.3f ms (%.1f TF) | gemm_k256_w8 %.3f ms (%.1f TF) %s\n", M, N,
                   t_n, gf / t_n, t_w, gf / t_w, t_g, gf / t_g, bad ? "DIFFERENT" : "bit-identical");
            CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(b)); CK(hipFree(Y0)); CK(hipFree(Y1));
        }
    return 0;
}
