// wino_kernels.hip - Winograd F(2 x 2, 3 x 3) for the detector's stride-1 3 x 3 convolutions with 256+ input channels (the FPN output
// convs, the RPN head conv, the deep bottlenecks' conv2: reference object_detection/models.py:6-20 -> torchvision's
// fasterrcnn_resnet50_fpn).  Parity of the detector is UNPINNED; cuDNN 7.6 - what the
// reference ran on - picks Winograd for these layers itself, so the transform is inside the reference's own numerics.
//
//   V_p = (B^T d B)_p   per 4 x 4 input tile d (stride 2), p = 16 tile positions     wino_input:  X [N][H][W][C] -> V [16][NT][C]
//   M_p = V_p U_p^T     16 products [NT x Cin] x [Cin x Cout] in ONE launch of the LDS-DMA GEMM (conv2d_nhwc_glds, batch = blockIdx.z)
//   Y   = A^T M A + bias (+ ReLU)                                                    wino_output: M [16][NT][Cout] -> Y [N][H][W][Cout]
//   U_p = (G g G^T)_p   once per weight set                                          wino_weights
// 2.25 x fewer MACs than the direct conv; the 16 position planes are extra traffic (4 x the input written and read, 4 x the output
// written and read), most of it absorbed by the 256 MB Infinity Cache.  Measured on the P2-level 256 -> 256 conv of a 16-frame pass
// (tools/probes/winograd_probe.hip, profiles/r6_winograd_probe.txt): 6.03 ms against 7.61 ms direct = 1.26 x, max error 2.2e-6 of max|y|.
#pragma once
#include "conv_kernels.hip"

__device__ __forceinline__ float4 wino_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 wino_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// U [16][Cout][Cin] from the packed conv weight w [Cout][KP], k = (ky * 3 + kx) * Cin + ci;  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void __launch_bounds__(256) wino_weights(const float *__restrict__ w, float *__restrict__ U, int Cin, int Cout, int KP)
{
    const long n = (long)Cout * Cin;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int ci = (int)(idx % Cin), co = (int)(idx / Cin);
        float g[3][3], t[4][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[(long)co * KP + (ky * 3 + kx) * Cin + ci];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            t[0][kx] = g[0][kx];
            t[1][kx] = 0.5f * ((g[0][kx] + g[2][kx]) + g[1][kx]);
            t[2][kx] = 0.5f * ((g[0][kx] + g[2][kx]) - g[1][kx]);
            t[3][kx] = g[2][kx];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float u0 = t[i][0], u1 = 0.5f * ((t[i][0] + t[i][2]) + t[i][1]), u2 = 0.5f * ((t[i][0] + t[i][2]) - t[i][1]), u3 = t[i][2];
            U[((long)(4 * i + 0) * Cout + co) * Cin + ci] = u0;
            U[((long)(4 * i + 1) * Cout + co) * Cin + ci] = u1;
            U[((long)(4 * i + 2) * Cout + co) * Cin + ci] = u2;
            U[((long)(4 * i + 3) * Cout + co) * Cin + ci] = u3;
        }
    }
}

// tiles [t0, t0 + NT) (tile = (image, ty, tx), TH = ceil(H / 2) x TW per image) of X [N][H][W][C] -> V [16][NT][C]; one thread per
// (tile, channel quad); pixels outside the image (the conv's zero padding, the odd row / column of an odd-sized map) read as zero
__global__ void __launch_bounds__(256) wino_input(const float *__restrict__ X, float *__restrict__ V, long t0, long NT, int H, int W, int C)
{
    const int C4 = C >> 2, TH = (H + 1) >> 1, TW = (W + 1) >> 1;
    const long n = NT * C4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const long tile = idx / C4, gt = t0 + tile;
        const int tx = (int)(gt % TW), ty = (int)((gt / TW) % TH), img = (int)(gt / ((long)TW * TH));
        float4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = 2 * ty - 1 + i, x = 2 * tx - 1 + j;
                d[i][j] = (y >= 0 && y < H && x >= 0 && x < W) ? ((const float4 *)X)[(((long)img * H + y) * W + x) * C4 + c4] : z;
            }
        float4 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = wino_sub(d[0][j], d[2][j]); t[1][j] = wino_add(d[1][j], d[2][j]);
            t[2][j] = wino_sub(d[2][j], d[1][j]); t[3][j] = wino_sub(d[1][j], d[3][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 *o = (float4 *)V + ((long)(4 * i) * NT + tile) * C4 + c4;
            o[0] = wino_sub(t[i][0], t[i][2]);
            o[NT * C4] = wino_add(t[i][1], t[i][2]);
            o[2 * NT * C4] = wino_sub(t[i][2], t[i][1]);
            o[3 * NT * C4] = wino_sub(t[i][1], t[i][3]);
        }
    }
}

// M [16][NT][Co] -> tiles [t0, t0 + NT) of Y [N][H][W][Co] = act(A^T M A + bias)
__global__ void __launch_bounds__(256) wino_output(const float *__restrict__ M, const float *__restrict__ bias, float *__restrict__ Y, long t0,
                                                   long NT, int H, int W, int Co, int relu)
{
    const int C4 = Co >> 2, TH = (H + 1) >> 1, TW = (W + 1) >> 1;
    const long n = NT * C4;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const long tile = idx / C4, gt = t0 + tile;
        const int tx = (int)(gt % TW), ty = (int)((gt / TW) % TH), img = (int)(gt / ((long)TW * TH));
        float4 m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = ((const float4 *)M)[((long)(4 * i + j) * NT + tile) * C4 + c4];
        float4 r[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[0][j] = wino_add(wino_add(m[0][j], m[1][j]), m[2][j]);
            r[1][j] = wino_sub(wino_sub(m[1][j], m[2][j]), m[3][j]);
        }
        const float4 b = bias ? ((const float4 *)bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int y = 2 * ty + a;
            if (y >= H) continue;
            float4 y0 = wino_add(wino_add(wino_add(r[a][0], r[a][1]), r[a][2]), b), y1 = wino_add(wino_sub(wino_sub(r[a][1], r[a][2]), r[a][3]), b);
            if (relu) {
                y0 = make_float4(fmaxf(y0.x, 0.f), fmaxf(y0.y, 0.f), fmaxf(y0.z, 0.f), fmaxf(y0.w, 0.f));
                y1 = make_float4(fmaxf(y1.x, 0.f), fmaxf(y1.y, 0.f), fmaxf(y1.z, 0.f), fmaxf(y1.w, 0.f));
            }
            float4 *o = (float4 *)Y + (((long)img * H + y) * W + 2 * tx) * C4 + c4;
            o[0] = y0;
            if (2 * tx + 1 < W) o[C4] = y1;
        }
    }
}
