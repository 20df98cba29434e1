// winograd_probe.hip - ONE bounded probe: Winograd F(2 x 2, 3 x 3) on the detector's P2-level 256 -> 256
// 3 x 3 conv (16 frames x 200 x 272, 29 % of a detector pass) against the direct LDS-DMA conv (csrc/conv_kernels.hip), fp32.
//   V = B^T d B per 4 x 4 input tile (input transform), M_p = V_p U_p^T for the 16 tile positions p (16 GEMMs [tiles x 256] x [256 x 256]
//   on the SAME conv2d_nhwc_glds kernel, as 1 x 1 convs), Y = A^T M A + bias, ReLU (output transform).  2.25 x fewer MACs; the transforms
//   and the 16 position planes are extra HBM traffic (4 x the input written, read, 4 x the output written, read).
// Adopt only if >= 1.25 x on this layer with max error <= 1e-5 relative; otherwise the table goes to profiles/ and the idea is dropped.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/winograd_probe tools/probes/winograd_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#include "../../objectpermanence_amd/csrc/conv_kernels.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// X [N][H][W][C] -> V [16][NT][C], NT = N (H / 2) (W / 2); one thread per (tile, channel quad)
__global__ void __launch_bounds__(256) wino_input(const float *__restrict__ X, float *__restrict__ V, int N, int H, int W, int C)
{
    const int C4 = C >> 2, TH = H >> 1, TW = W >> 1;
    const long NT = (long)N * TH * TW, n = NT * C4;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const long tile = idx / C4;
        const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), img = (int)(tile / ((long)TW * TH));
        float4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = 2 * ty - 1 + i, x = 2 * tx - 1 + j;
                d[i][j] = (y >= 0 && y < H && x >= 0 && x < W) ? ((const float4 *)X)[(((long)img * H + y) * W + x) * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        float4 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = f4sub(d[0][j], d[2][j]); t[1][j] = f4add(d[1][j], d[2][j]);
            t[2][j] = f4sub(d[2][j], d[1][j]); t[3][j] = f4sub(d[1][j], d[3][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 v0 = f4sub(t[i][0], t[i][2]), v1 = f4add(t[i][1], t[i][2]), v2 = f4sub(t[i][2], t[i][1]), v3 = f4sub(t[i][1], t[i][3]);
            float4 *o = (float4 *)V + ((long)(4 * i) * NT + tile) * C4 + c4;
            o[0] = v0; o[NT * C4] = v1; o[2 * NT * C4] = v2; o[3 * NT * C4] = v3;
        }
    }
}

// M [16][NT][Co] -> Y [N][H][W][Co] = relu(A^T M A + bias)
__global__ void __launch_bounds__(256) wino_output(const float *__restrict__ M, const float *__restrict__ bias, float *__restrict__ Y, int N, int H,
                                                   int W, int Co)
{
    const int C4 = Co >> 2, TH = H >> 1, TW = W >> 1;
    const long NT = (long)N * TH * TW, n = NT * C4;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const long tile = idx / C4;
        const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), img = (int)(tile / ((long)TW * TH));
        float4 m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = ((const float4 *)M)[((long)(4 * i + j) * NT + tile) * C4 + c4];
        float4 r[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[0][j] = f4add(f4add(m[0][j], m[1][j]), m[2][j]);
            r[1][j] = f4sub(f4sub(m[1][j], m[2][j]), m[3][j]);
        }
        const float4 b = ((const float4 *)bias)[c4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float4 y0 = f4add(f4add(f4add(r[a][0], r[a][1]), r[a][2]), b), y1 = f4add(f4sub(f4sub(r[a][1], r[a][2]), r[a][3]), b);
            y0 = make_float4(fmaxf(y0.x, 0.f), fmaxf(y0.y, 0.f), fmaxf(y0.z, 0.f), fmaxf(y0.w, 0.f));
            y1 = make_float4(fmaxf(y1.x, 0.f), fmaxf(y1.y, 0.f), fmaxf(y1.z, 0.f), fmaxf(y1.w, 0.f));
            float4 *o = (float4 *)Y + (((long)img * H + 2 * ty + a) * W + 2 * tx) * C4 + c4;
            o[0] = y0; o[C4] = y1;
        }
    }
}

int main()
{
    const int N = 16, H = 200, W = 272, C = 256, Co = 256;
    const long Mpix = (long)N * H * W, NT = Mpix / 4;
    const int K = 9 * C;
    std::vector<float> hx((size_t)Mpix * C), hw((size_t)Co * K), hb(Co), hu((size_t)16 * Co * C);
    srand(7);
    for (auto &v : hx) v = (float)((rand() & 2047) - 1024) / 1024.f;
    for (auto &v : hw) v = (float)((rand() & 2047) - 1024) / 1024.f / 48.f;        // |y| ~ 1 like a BN-folded layer
    for (auto &v : hb) v = (float)((rand() & 255) - 128) / 256.f;
    // U = G g G^T per (co, ci), float64 on the host (an offline weight transform)
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < C; ++ci) {
            double g[3][3], t[4][3];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) g[ky][kx] = hw[(size_t)co * K + (ky * 3 + kx) * C + ci];
            for (int i = 0; i < 4; ++i)
                for (int kx = 0; kx < 3; ++kx) t[i][kx] = G[i][0] * g[0][kx] + G[i][1] * g[1][kx] + G[i][2] * g[2][kx];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    hu[((size_t)(4 * i + j) * Co + co) * C + ci] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
    float *X, *Wt, *B, *Yd, *Yw, *U, *V, *Mm;
    CK(hipMalloc(&X, hx.size() * 4)); CK(hipMalloc(&Wt, hw.size() * 4)); CK(hipMalloc(&B, Co * 4));
    CK(hipMalloc(&Yd, (size_t)Mpix * Co * 4)); CK(hipMalloc(&Yw, (size_t)Mpix * Co * 4)); CK(hipMalloc(&U, hu.size() * 4));
    CK(hipMalloc(&V, (size_t)16 * NT * C * 4)); CK(hipMalloc(&Mm, (size_t)16 * NT * Co * 4));
    CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Wt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), Co * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(U, hu.data(), hu.size() * 4, hipMemcpyHostToDevice));
    ConvArgs d = {};
    d.X = X; d.Wt = Wt; d.bias = B; d.Y = Yd; d.N = N; d.H = H; d.W = W; d.Cin = C; d.Cout = Co; d.KH = 3; d.KW = 3; d.stride = 1; d.pad = 1;
    d.OH = H; d.OW = W; d.KP = K; d.relu = 1;
    auto direct = [&]() { conv2d_nhwc_glds<128, 3><<<dim3((unsigned)((Mpix + 127) / 128), (Co + 127) / 128, 1), 256>>>(d); };
    auto gemms = [&](bool bn64) {
        for (int p = 0; p < 16; ++p) {
            ConvArgs g = {};
            g.X = V + (size_t)p * NT * C; g.Wt = U + (size_t)p * Co * C; g.Y = Mm + (size_t)p * NT * Co;
            g.N = 1; g.H = 1; g.W = (int)NT; g.Cin = C; g.Cout = Co; g.KH = 1; g.KW = 1; g.stride = 1; g.pad = 0; g.OH = 1; g.OW = (int)NT; g.KP = C;
            if (bn64) conv2d_nhwc_glds<64, 3><<<dim3((unsigned)((NT + 127) / 128), (Co + 63) / 64, 1), 256>>>(g);
            else conv2d_nhwc_glds<128, 3><<<dim3((unsigned)((NT + 127) / 128), (Co + 127) / 128, 1), 256>>>(g);
        }
    };
    auto tin = [&]() { wino_input<<<8192, 256>>>(X, V, N, H, W, C); };
    auto tout = [&]() { wino_output<<<8192, 256>>>(Mm, B, Yw, N, H, W, Co); };
    auto time = [&](auto fn, const char *what, double flop) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        fn(); CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-58s %8.3f ms", what, best);
        if (flop > 0) printf("   %6.1f TF", flop / best / 1e9);
        printf("\n");
        return best;
    };
    const double fd = 2.0 * Mpix * Co * K;
    printf("P2-level conv 16 x 200 x 272, 256 -> 256, 3 x 3, fp32 (MI355X)\n");
    const float t_d = time(direct, "direct: conv2d_nhwc_glds<128, 3>", fd);
    const float t_i = time(tin, "Winograd input transform (56 MB in, 223 MB out / frame)", 0);
    const float t_g = time([&]() { gemms(false); }, "Winograd 16 GEMMs [217600 x 256] x [256 x 256], 128-wide tiles", fd / 2.25);
    const float t_g64 = time([&]() { gemms(true); }, "Winograd 16 GEMMs, 64-wide tiles", fd / 2.25);
    const float t_o = time(tout, "Winograd output transform (223 MB in, 56 MB out / frame)", 0);
    const float t_w = time([&]() { tin(); gemms(t_g64 < t_g); tout(); }, "Winograd whole layer", fd);
    printf("speed-up over the direct conv: %.3f x (adopt from 1.25 x)\n", t_d / t_w);
    direct(); tin(); gemms(false); tout();
    CK(hipDeviceSynchronize());
    std::vector<float> ya((size_t)Mpix * Co), yb((size_t)Mpix * Co);
    CK(hipMemcpy(ya.data(), Yd, ya.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(yb.data(), Yw, yb.size() * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxy = 0;
    for (size_t k = 0; k < ya.size(); ++k) { maxd = fmax(maxd, fabs((double)ya[k] - yb[k])); maxy = fmax(maxy, fabs((double)ya[k])); }
    printf("max |direct - Winograd| = %.3e, max |y| = %.3f, relative %.3e (bar: 1e-5)\n", maxd, maxy, maxd / maxy);
    return 0;
}
