// probe: what does an in-kernel, flag-based exchange of the recurrent state cost per step on gfx950, compared with the
// dependent kernel boundary the product uses?  A persistent grid of NWG workgroups runs T steps;
// in step t every workgroup (1) waits until all workgroups have published step t-1 (one agent-scope counter per step),
// (2) loads the whole 64-KiB state of step t-1 from addresses nobody has touched before (so no cache can hold a stale
// copy and no invalidate is needed), (3) optionally burns MFMA time, (4) publishes its 512-byte slice of step t with
// agent-scope (write-through) stores and bumps the counter.  Every loaded value is checked (state of step t == t), so
// a visibility bug shows up as a mismatch count instead of as a timing artefact.  Spins are bounded: no hangs.
// Build: hipcc --offload-arch=gfx950 -O2 -o persist_probe persist_probe.hip ; run: ./persist_probe [T] [nkernels] [mfma]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define NWG 192
#define STATE_FLOATS (NWG * 128)      // 96 KiB per step: workgroup w owns floats [128 w, 128 w + 128)
#define LOAD_F4 4096                  // each workgroup reads the first 64 KiB of the previous step
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) chain(float *state, unsigned *cnt, unsigned *err, int T, int mfma_iters, int fence_mode, int nload, int nsleep)
{
    const int tid = threadIdx.x;
    __shared__ unsigned stop;
    if (tid == 0) stop = 0;
    __syncthreads();
    unsigned bad = 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 1; t <= T; ++t) {
        if (t > 1) {
            if (fence_mode == 3) {
                // one flag word per producer workgroup (plain agent-scope stores, nothing serialises on one address)
                if (tid < 64) {
                    unsigned spins = 0;
                    const unsigned *f = cnt + (size_t)(t - 1) * NWG;
                    for (;;) {
                        const unsigned a = __hip_atomic_load(&f[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned b = __hip_atomic_load(&f[tid + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned c = __hip_atomic_load(&f[tid + 128], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__all((a & b & c) != 0)) break;
                        if (++spins > (1u << 20)) {
                            if (tid == 0) { __hip_atomic_store(&err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); stop = 1; }
                            break;
                        }
                        for (int z = 0; z < nsleep; ++z) __builtin_amdgcn_s_sleep(1);
                    }
                }
            } else if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(&cnt[t - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NWG) {
                    if (++spins > (1u << 21) || __hip_atomic_load(&err[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(&err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        stop = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
            if (stop) return;
            if (fence_mode == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);   // hip: agent-scope acquire (buffer_inv sc1)
        }
        const float4 *src = (const float4 *)(state + (size_t)(t - 1) * STATE_FLOATS);
        float4 v[LOAD_F4 / 256];
        const float want = (float)(t - 1);
#pragma unroll
        for (int j = 0; j < LOAD_F4 / 256; ++j) v[j] = j < nload ? src[j * 256 + tid] : make_float4(want, want, want, want);
#pragma unroll
        for (int j = 0; j < LOAD_F4 / 256; ++j)
            bad += (v[j].x != want) + (v[j].y != want) + (v[j].z != want) + (v[j].w != want);
        for (int i = 0; i < mfma_iters; ++i)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0].x, v[1].y, acc, 0, 0, 0);
        float *dst = state + (size_t)t * STATE_FLOATS + blockIdx.x * 128;
        const float outv = (float)t + (acc[0] != acc[0] ? 1.f : 0.f);    // keeps the MFMA chain alive; acc is finite
        if (fence_mode == 3) {
            if (tid < 128) __hip_atomic_store(&dst[tid], outv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&cnt[(size_t)t * NWG + blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (fence_mode == 0) {
            if (tid < 128) __hip_atomic_store(&dst[tid], outv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&cnt[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (tid < 128) dst[tid] = outv;
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&cnt[t], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (bad) atomicAdd(&err[1], bad);
}

// the product's way: one dependent launch per step doing the same loads / stores
__global__ void __launch_bounds__(256) step(float *state, unsigned *err, int t, int mfma_iters)
{
    const int tid = threadIdx.x;
    const float4 *src = (const float4 *)(state + (size_t)(t - 1) * STATE_FLOATS);
    float4 v[LOAD_F4 / 256];
    unsigned bad = 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < LOAD_F4 / 256; ++j) v[j] = src[j * 256 + tid];
    const float want = (float)(t - 1);
#pragma unroll
    for (int j = 0; j < LOAD_F4 / 256; ++j)
        bad += (v[j].x != want) + (v[j].y != want) + (v[j].z != want) + (v[j].w != want);
    for (int i = 0; i < mfma_iters; ++i)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i & 15].x, v[(i + 1) & 15].y, acc, 0, 0, 0);
    float *dst = state + (size_t)t * STATE_FLOATS + blockIdx.x * 128;
    if (tid < 128) dst[tid] = (float)t + (acc[0] != acc[0] ? 1.f : 0.f);
    if (bad) atomicAdd(&err[1], bad);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 300;
    const int NK = argc > 2 ? atoi(argv[2]) : 4;
    const int mf = argc > 3 ? atoi(argv[3]) : 64;
    const int nload = argc > 4 ? atoi(argv[4]) : 16;
    const int nsleep = argc > 5 ? atoi(argv[5]) : 1;
    const int only = argc > 6 ? atoi(argv[6]) : -1;
    std::vector<float *> st(NK);
    std::vector<unsigned *> cnt(NK), err(NK);
    std::vector<hipStream_t> s(NK);
    const size_t sbytes = (size_t)(T + 1) * STATE_FLOATS * 4;
    for (int k = 0; k < NK; ++k) {
        CK(hipMalloc(&st[k], sbytes));
        CK(hipMalloc(&cnt[k], (size_t)(T + 2) * NWG * 4));
        CK(hipMalloc(&err[k], 8));
        CK(hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 4; ++mode) {
        if (only >= 0 && mode != only) continue;          // 0: write-through stores + relaxed flag; 1: release/acquire fences; 2: launches (host-bound here: no graph)
        for (int nk = 1; nk <= NK; nk += (NK > 1 ? NK - 1 : 1)) {
            float best = 1e30f;
            unsigned herr[2] = {0, 0};
            for (int rep = 0; rep < 4; ++rep) {
                for (int k = 0; k < nk; ++k) {
                    CK(hipMemsetAsync(st[k], 0, sbytes, 0));
                    CK(hipMemsetAsync(cnt[k], 0, (size_t)(T + 2) * NWG * 4, 0));
                    CK(hipMemsetAsync(err[k], 0, 8, 0));
                }
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                CK(hipDeviceSynchronize());
                for (int k = 0; k < nk; ++k) {
                    if (mode != 2) {
                        hipLaunchKernelGGL(chain, dim3(NWG), dim3(256), 0, s[k], st[k], cnt[k], err[k], T, mf, mode, nload, nsleep);
                    } else {
                        for (int t = 1; t <= T; ++t) hipLaunchKernelGGL(step, dim3(NWG), dim3(256), 0, s[k], st[k], err[k], t, mf);
                    }
                }
                for (int k = 0; k < nk; ++k) CK(hipStreamSynchronize(s[k]));
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                for (int k = 0; k < nk; ++k) {
                    unsigned h2[2];
                    CK(hipMemcpy(h2, err[k], 8, hipMemcpyDeviceToHost));
                    herr[0] |= h2[0];
                    herr[1] += h2[1];
                }
            }
            printf("load %d KiB sleep %d | mode %d (%s)  kernels in flight %d  T %d  mfma/step %d : %.3f ms  = %.2f us per step, %.2f us amortised  timeout %u  stale values %u\n",
                   nload * 4, nsleep, mode, mode == 0 ? "write-through + one counter" : mode == 1 ? "release/acquire fences" : mode == 2 ? "one launch per step" : "write-through + flag per workgroup",
                   nk, T, mf, best, best * 1e3f / T, best * 1e3f / T / nk, herr[0], herr[1]);
            if (NK == 1) break;
        }
    }
    return 0;
}
