// probe (INCONCLUSIVE as a stand-alone program: a bare-HIP graph chain runs at 12-23 us per node on this stack, bimodal,
// so the effect was measured on the product kernel instead - opnet_step_pl): does kernarg preloading (user SGPRs filled by the CP at dispatch) shorten a dependent launch whose first memory
// access needs a pointer argument?  Build twice: plain, and with -mllvm -amdgpu-kernarg-preload-count=12.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) step(const float4 *src, float4 *dst, int n, int s)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 v = src[i ^ (s & 255)];
    v.x += 1.f;
    dst[i] = v;
}
int main()
{
    const int nwg = 194, n = nwg * 256, T = 300;
    float4 *a, *b;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipGraphCreate(&g, 0));
    hipGraphNode_t prev = nullptr, node;
    for (int s = 0; s < T; ++s) {
        const float4 *src = (s & 1) ? b : a;
        float4 *dst = (s & 1) ? a : b;
        int nn = n, ss = s;
        void *args[] = {(void *)&src, (void *)&dst, (void *)&nn, (void *)&ss};
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        kp.func = (void *)step;
        kp.gridDim = dim3(nwg, 1, 1);
        kp.blockDim = dim3(256, 1, 1);
        kp.kernelParams = args;
        CK(hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
        prev = node;
    }
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 600; ++rep) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%.3f us per dependent launch\n", best * 1e3f / T);
    return 0;
}
