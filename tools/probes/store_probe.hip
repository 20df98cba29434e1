// store_probe.hip - how fast 256-thread workgroups write 128 x 128 fp32 tiles of a row-major [M][N] matrix, by the shape of one
// store instruction (64 lanes x 16 B): 16 rows x 64 B (the MFMA fragment layout of conv2d_nhwc_glds's epilogue), 4 rows x 256 B,
// 2 rows x 512 B (a wave's 64 columns ... a tile's 128), with the kernel's tile order; GB/s over the whole matrix.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float *Y, long M, int N)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned nt = gridDim.x * gridDim.y;
    unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    { const unsigned per = nt >> 3, rem = nt & 7u, xcd = lin & 7u, slot = lin >> 3; lin = xcd * per + min(xcd, rem) + slot; }
    const long m0 = (long)(lin / gridDim.y) * 128;
    const int n0 = (lin % gridDim.y) * 128;
    const float4 v = make_float4((float)tid, 1.f, 2.f, 3.f);
    if (MODE == 0) {            // fragment layout: wave (wm, wn) 64 x 64; lane (i = pixel, kk): 16 rows x 64 B per instruction
        const int wm = w >> 1, wn = w & 1, i = lane & 15, kk = lane >> 4;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y)
                *(float4 *)(Y + (m0 + wm * 64 + x * 16 + i) * N + n0 + wn * 64 + y * 16 + 4 * kk) = v;
    } else if (MODE == 1) {     // wave (wm, wn) 64 x 64; 4 rows x 256 B per instruction
        const int wm = w >> 1, wn = w & 1, r = lane >> 4, c = lane & 15;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            *(float4 *)(Y + (m0 + wm * 64 + j * 4 + r) * N + n0 + wn * 64 + 4 * c) = v;
    } else if (MODE == 2) {     // wave w: rows 32 w .. 32 w + 31, all 128 columns; 2 rows x 512 B per instruction
        const int r = lane >> 5, c = lane & 31;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            *(float4 *)(Y + (m0 + w * 32 + j * 2 + r) * N + n0 + 4 * c) = v;
    } else {                    // nt variant of mode 1
        const int wm = w >> 1, wn = w & 1, r = lane >> 4, c = lane & 15;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            __builtin_nontemporal_store((f4){v.x, v.y, v.z, v.w}, (f4 *)(Y + (m0 + wm * 64 + j * 4 + r) * N + n0 + wn * 64 + 4 * c));
    }
}

template <int MODE>
int run(float *Y, long M, int N, const char *what)
{
    const dim3 g((unsigned)(M / 128), N / 128, 1);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<g, 256>>>(Y, M, N);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k<MODE><<<g, 256>>>(Y, M, N);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("M=%ld N=%4d %-28s %8.1f us  %7.1f GB/s\n", M, N, what, ms * 1e3, (double)M * N * 4 / ms / 1e6);
    return 0;
}

int main()
{
    float *Y;
    CK(hipMalloc(&Y, (size_t)870400 * 256 * 4 + (size_t)76800 * 2048 * 4));
    for (int N : {2048, 768, 256}) {
        run<0>(Y, 76800, N, "16 rows x 64 B (fragment)");
        run<1>(Y, 76800, N, "4 rows x 256 B");
        run<2>(Y, 76800, N, "2 rows x 512 B");
        run<3>(Y, 76800, N, "4 rows x 256 B nontemporal");
    }
    run<0>(Y, 870400, 256, "16 rows x 64 B (fragment)");
    run<1>(Y, 870400, 256, "4 rows x 256 B");
    run<2>(Y, 870400, 256, "2 rows x 512 B");
    return 0;
}
