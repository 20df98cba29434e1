// mfma4x4_probe.hip - operand layout and issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950 (the 4-clip persistent step's MFMA).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma4x4_probe tools/probes/mfma4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// D = A x B with A[lane] = lane + 1, B[lane] = 1000 * (lane + 1): which (A lane, B lane) pair ends up in D[lane][reg]?
__global__ void layout(float *out)
{
    const int lane = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), 1000.f * (lane + 1), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

template <int CHAINS>
__global__ void rate(float *out, long long *cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    f32x4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = lane * 0.001f, b = 0.5f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            c[k % CHAINS] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[k % CHAINS], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS>
__global__ void rate16(float *out, long long *cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    f32x4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = lane * 0.001f, b = 0.5f;
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            c[k % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[k % CHAINS], 0, 0, 0);
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

int main()
{
    float *d; long long *dc;
    hipMalloc(&d, 1 << 20); hipMalloc(&dc, 16);
    float h[256];
    layout<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            // expected: D[block b][row i = r][col j] at lane 4 b + j: A lane 4 b + r, B lane 4 b + j
            const int b = lane >> 2, j = lane & 3;
            const float want = (float)(4 * b + r + 1) * 1000.f * (4 * b + j + 1);
            if (h[lane * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", lane, r, h[lane * 4 + r], want); ++bad; }
        }
    printf("layout D[lane 4b+j][reg i] = A[lane 4b+i] * B[lane 4b+j]: %s\n", bad ? "NO" : "yes");
    const int iters = 2000;
    long long c;
#define RUN(CH, BLK, THR) do { rate<CH><<<BLK, THR>>>(d, dc, iters); hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); \
        printf("chains %d, %d waves/CU: %.2f clock64 cycles per MFMA per wave\n", CH, THR / 64, (double)c / (16.0 * iters)); } while (0)
    RUN(1, 1, 64); RUN(2, 1, 64); RUN(4, 1, 64); RUN(4, 1, 256); RUN(4, 1, 512);
    long long c2[2];
#define RUN16(CH, THR) do { rate16<CH><<<1, THR>>>(d, dc, iters); hipDeviceSynchronize(); hipMemcpy(c2, dc, 16, hipMemcpyDeviceToHost); \
        printf("16x16x4: chains %d, %d waves/CU: %.2f clock64 per MFMA per wave; wall_clock64 ticks %lld for clock64 %lld\n", CH, THR / 64, (double)c2[0] / (16.0 * iters), c2[1], c2[0]); } while (0)
    RUN16(4, 64); RUN16(4, 256); RUN16(4, 512);
    return 0;
}
