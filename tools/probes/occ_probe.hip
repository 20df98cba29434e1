// occ_probe.hip - what bounds the residency of a 256-thread workgroup on gfx950: LDS bytes x registers, by the occupancy API and a census
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int KB, int REGS>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *census)
{
    __shared__ float s[KB * 256];
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = out[threadIdx.x + i * 256];
    s[threadIdx.x] = r[0];
    if (threadIdx.x == 0) { const unsigned long long now = atomicAdd(census, 1ull) + 1; atomicMax(census + 1, now); }
    __syncthreads();
    for (int it = 0; it < 2000; ++it) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) r[i] = r[i] * 1.0001f + s[(threadIdx.x + it) & 255];
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) acc += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) atomicAdd(census, ~0ull);
}
template <int KB, int REGS>
int run(float *out, unsigned long long *census)
{
    int api = 0;
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void *)k<KB, REGS>));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k<KB, REGS>, 256, 0));
    CK(hipMemset(census, 0, 16));
    k<KB, REGS><<<4096, 256>>>(out, census);
    CK(hipDeviceSynchronize());
    unsigned long long c[2];
    CK(hipMemcpy(c, census, 16, hipMemcpyDeviceToHost));
    printf("LDS %3d KB  regs %3d (numRegs %d, static LDS %zu): API %d per CU, census %.2f per CU\n", KB, REGS, fa.numRegs, fa.sharedSizeBytes, api, c[1] / 256.0);
    return 0;
}
int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s: CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerBlock %d, regsPerMultiprocessor %d\n", p.gcnArchName,
           p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.regsPerMultiprocessor);
    float *out; unsigned long long *census;
    CK(hipMalloc(&out, 4096 * 256 * 4 * 2)); CK(hipMalloc(&census, 16));
    run<16, 16>(out, census); run<32, 16>(out, census); run<40, 16>(out, census); run<48, 16>(out, census); run<52, 16>(out, census);
    run<64, 16>(out, census); run<80, 16>(out, census);
    run<16, 64>(out, census); run<16, 100>(out, census); run<16, 120>(out, census); run<16, 140>(out, census); run<32, 140>(out, census);
    return 0;
}
