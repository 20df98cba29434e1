// probe: buffer_load_dwordx4 ... offen lds  (LDS-DMA) semantics on gfx950: lane-linear destination at M0, the
// instruction offset, out-of-range lanes.  Build: hipcc --offload-arch=gfx950 -O2 -o glds_probe glds_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(u32x4 rsrc, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

__global__ void k(const float *src, unsigned bytes, float *out, const unsigned *offs)
{
    __shared__ __attribute__((aligned(16))) float smem[4096];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4096; i += 256) smem[i] = -1.f;
    __syncthreads();
    const unsigned long long base = (unsigned long long)src;
    u32x4 rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)base);
    rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
    rsrc.z = __builtin_amdgcn_readfirstlane(bytes);
    rsrc.w = 0x00020000u;
    glds16(rsrc, offs[tid], w * 1024 * 4);       // wave w -> smem[w*1024 .. +256 floats)? (64 lanes x 16 B = 1 KiB = 256 floats)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __syncthreads();
    for (int i = tid; i < 4096; i += 256) out[i] = smem[i];
}

int main()
{
    const int n = 8192;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    std::vector<unsigned> offs(256);
    for (int t = 0; t < 256; ++t) offs[t] = (unsigned)(((t * 7) % 500) * 16);       // arbitrary 16-B aligned sources
    offs[3] = 0xffffffffu; offs[70] = 0x80000000u; offs[130] = n * 4 - 8;            // out of range in three ways
    float *d, *o; unsigned *doff;
    hipMalloc(&d, n * 4); hipMalloc(&o, 4096 * 4); hipMalloc(&doff, 256 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(doff, offs.data(), 256 * 4, hipMemcpyHostToDevice);
    k<<<1, 256>>>(d, n * 4, o, doff);
    std::vector<float> r(4096);
    hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const int w = t / 64, l = t % 64;
        const float *got = &r[w * 1024 + l * 4];
        const bool oob = t == 3 || t == 70 || t == 130;
        for (int e = 0; e < 4; ++e) {
            const float want = oob ? 0.f : (float)(offs[t] / 4 + e);
            if (got[e] != want) { if (bad < 10) printf("t=%d e=%d got %g want %g\n", t, e, got[e], want); ++bad; }
        }
    }
    // untouched part of each wave's KiB region stays -1
    int touched = 0;
    for (int w = 0; w < 4; ++w) for (int i = 256; i < 1024; ++i) touched += r[w * 1024 + i] != -1.f;
    printf("mismatches %d, stray writes %d (t=130 partial: %g %g %g %g)\n", bad, touched, r[2*1024+2*4], r[2*1024+2*4+1], r[2*1024+2*4+2], r[2*1024+2*4+3]);
    return bad != 0;
}
