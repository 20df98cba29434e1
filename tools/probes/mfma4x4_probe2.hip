// mfma4x4_probe2.hip - what sets the issue rate of v_mfma_f32_4x4x1_16b_f32 in a product loop: distinct A registers, B operands
// fresh from ds_read_b128, the number of accumulator chains.  (seqx_forward measured 13.4 cycles per MFMA against 9.5 in
// mfma4x4_probe's constant-operand loop.)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mp2 tools/probes/mfma4x4_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(acc, a, b) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFA(acc, a, b) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b))

// MODE bit 0: distinct A registers (128); bit 1: B from LDS (ring of 8, 6 ahead); bit 2: the A operands are AccVGPRs; CH chains
template <int MODE, int CH>
__global__ void __launch_bounds__(256) rate(const float *w, float *out, long long *cyc, int iters)
{
    __shared__ float4 sH[4][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float a[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) a[i] = w[i * 64 + lane];
    float ag[128];
    if (MODE & 4) {
#pragma unroll
        for (int i = 0; i < 128; ++i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ag[i]) : "v"(a[i]));
    }
    sH[wv][lane] = make_float4(lane, 1.f, 2.f, 3.f);
    sH[wv][64 + lane] = make_float4(lane, 1.f, 2.f, 3.f);
    __syncthreads();
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float4 *F = &sH[wv][0] + (lane & 3);
    const float4 cb = make_float4(0.5f, 0.25f, 0.125f, 1.f);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        float4 bf[8];
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 6; ++i) bf[i] = F[i * 4];
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            if ((MODE & 2) && q + 6 < 32) bf[(q + 6) & 7] = F[(q + 6) * 4];
            __builtin_amdgcn_sched_barrier(0);
            const float4 bq = (MODE & 2) ? bf[q & 7] : cb;
            const int o = (q * 4) % CH;
            const int ai = (MODE & 1) ? 4 * q : 0;
            if (MODE & 4) {
                MFA(c[(o + 0) % CH], ag[ai + 0], bq.x);
                MFA(c[(o + 1) % CH], ag[ai + 1], bq.y);
                MFA(c[(o + 2) % CH], ag[ai + 2], bq.z);
                MFA(c[(o + 3) % CH], ag[ai + 3], bq.w);
            } else {
                MF(c[(o + 0) % CH], a[ai + 0], bq.x);
                MF(c[(o + 1) % CH], a[ai + 1], bq.y);
                MF(c[(o + 2) % CH], a[ai + 2], bq.z);
                MF(c[(o + 3) % CH], a[ai + 3], bq.w);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    float *d, *w; long long *dc;
    hipMalloc(&d, 1 << 20); hipMalloc(&w, 1 << 20); hipMalloc(&dc, 16);
    hipMemset(w, 0, 1 << 20);
    const int iters = 500;
    long long c;
#define RUN(M, CH) do { rate<M, CH><<<1, 256>>>(w, d, dc, iters); hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); \
        printf("distinct A %d, B from LDS %d, A in AGPRs %d, chains %d: %.2f cycles per MFMA\n", M & 1, (M >> 1) & 1, (M >> 2) & 1, CH, (double)c / (128.0 * iters)); } while (0)
    RUN(0, 4); RUN(1, 4); RUN(2, 4); RUN(3, 4); RUN(3, 8); RUN(5, 4); RUN(7, 4); RUN(7, 8);
    return 0;
}
