// gemm_probe.hip - the LDS-DMA conv / GEMM kernel (csrc/conv_kernels.hip conv2d_nhwc_glds) alone: time per shape, and - built
// with -DCONV_TRACE - where a K step's cycles go (wave 0 of every 32nd workgroup), at the kernel's own occupancy and with the
// occupancy forced down by padding the launch's LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCONV_TRACE -o tools/probes/gemm_probe tools/probes/gemm_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#include "../../objectpermanence_amd/csrc/conv_kernels.hip"
#include "conv_persistent_variant.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { const char *name; int N, H, W, Cin, Cout, k, pad; };

int main(int argc, char **argv)
{
    const Shape shapes[] = {
        {"ffn1  M=76800 N=2048 K=256 ", 1, 1, 76800, 256, 2048, 1, 0},
        {"ffn2  M=76800 N=256  K=2048", 1, 1, 76800, 2048, 256, 1, 0},
        {"qkv   M=76800 N=768  K=256 ", 1, 1, 76800, 256, 768, 1, 0},
        {"ffn1  M=19200 N=2048 K=256 ", 1, 1, 19200, 256, 2048, 1, 0},
        {"outer0 16x200x272 256->256 3x3", 16, 200, 272, 256, 256, 3, 1},
        {"l2c2   16x100x136 128->128 3x3", 16, 100, 136, 128, 128, 3, 1},
        {"l4c2   16x25x34   512->512 3x3", 16, 25, 34, 512, 512, 3, 1},
        {"l3c1   16x50x68  1024->256 1x1", 16, 50, 68, 1024, 256, 1, 0},
        {"l3c2   16x50x68   256->256 3x3", 16, 50, 68, 256, 256, 3, 1},
        {"l3c3   16x50x68   256->1024 1x1", 16, 50, 68, 256, 1024, 1, 0},
        {"fc6    M=16000 N=1024 K=12544 ", 1, 1, 16000, 12544, 1024, 1, 0},
        {"p3     16x100x136 256->256 3x3", 16, 100, 136, 256, 256, 3, 1},
    };
    unsigned long long *trace;
    CK(hipMalloc(&trace, 64));
    for (const Shape &s : shapes) {
        ConvArgs c = {};
        const long M = (long)s.N * s.H * s.W;
        const int K = s.k * s.k * s.Cin;
        float *X, *Wt, *Y, *B;
        CK(hipMalloc(&X, (size_t)M * s.Cin * 4)); CK(hipMalloc(&Wt, (size_t)s.Cout * K * 4)); CK(hipMalloc(&Y, (size_t)M * s.Cout * 4));
        CK(hipMalloc(&B, s.Cout * 4));
        std::vector<float> h((size_t)1 << 20);
        for (auto &v : h) v = (float)((rand() & 1023) - 512) / 512.f;
        for (size_t o = 0; o < (size_t)M * s.Cin; o += h.size())
            CK(hipMemcpy(X + o, h.data(), std::min(h.size(), (size_t)M * s.Cin - o) * 4, hipMemcpyHostToDevice));
        for (size_t o = 0; o < (size_t)s.Cout * K; o += h.size())
            CK(hipMemcpy(Wt + o, h.data(), std::min(h.size(), (size_t)s.Cout * K - o) * 4, hipMemcpyHostToDevice));
        CK(hipMemset(B, 0, s.Cout * 4));
        c.X = X; c.Wt = Wt; c.bias = B; c.R = nullptr; c.Y = Y;
        c.N = s.N; c.H = s.H; c.W = s.W; c.Cin = s.Cin; c.Cout = s.Cout; c.KH = s.k; c.KW = s.k; c.stride = 1; c.pad = s.pad;
        c.OH = s.H; c.OW = s.W; c.KP = K; c.relu = 1;
#ifdef CONV_TRACE
        c.trace = trace;
#endif
        const dim3 g((unsigned)((M + 127) / 128), (s.Cout + 127) / 128, 1);
        {   // the persistent tile loop (conv2d_nhwc_pglds): 3 workgroups per CU resident, one stage ring across tiles; checked against the launch-per-tile form
            int per_cu = 0;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv2d_nhwc_pglds<128, 3>, 256, 0));
            const unsigned long long nt = (unsigned long long)g.x * g.y;
            const unsigned slots = (unsigned)per_cu * 256, gp = nt >= slots ? slots : (unsigned)((nt + 7) / 8 * 8);
            c.relu = 1;
            float *Y2;
            CK(hipMalloc(&Y2, (size_t)M * s.Cout * 4));
            conv2d_nhwc_glds<128, 3><<<g, 256>>>(c);
            ConvArgs c2 = c; c2.Y = Y2;
            conv2d_nhwc_pglds<128, 3><<<gp, 256>>>(c2);
            CK(hipDeviceSynchronize());
            std::vector<float> ya((size_t)M * s.Cout), yb((size_t)M * s.Cout);
            CK(hipMemcpy(ya.data(), Y, ya.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(yb.data(), Y2, yb.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t k = 0; k < ya.size(); ++k) bad += ya[k] != yb[k];
            for (int mode : {0, 1, 2, 3}) {       // 0 as is, 1 no stores, 2 stores spread over the K loop (wrong values), 3 as is with start skew
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                const int skew = mode == 3 ? 100000 : 0;
                c2.ksteps = skew;
                c2.relu = mode == 1 ? 257 : mode == 2 ? 513 : 1;
                CK(hipEventRecord(e0));
                for (int r = 0; r < 5; ++r) conv2d_nhwc_pglds<128, 3><<<gp, 256>>>(c2);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= 5;
                const double fl = 2.0 * M * s.Cout * K;
                printf("%-32s persistent mode %d (%u workgroups, %d per CU, start skew %6d cycles): %8.1f us %6.1f TF (%.3f)   differing from the launch-per-tile form: %zu\n",
                       s.name, mode, gp, per_cu, skew, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 157.3, bad);
            }
            CK(hipFree(Y2));
        }
        {   // the 128 x 64-tile form of the product kernel (4 resident per CU): twice the tiles - better rounds on mid-size layers?
            const dim3 g64((unsigned)((M + 127) / 128), (s.Cout + 63) / 64, 1);
            c.relu = 1;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            conv2d_nhwc_glds<64, 3><<<g64, 256>>>(c);
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) conv2d_nhwc_glds<64, 3><<<g64, 256>>>(c);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 5;
            const double fl = 2.0 * M * s.Cout * K;
            printf("%-32s BN = 64 tiles (%u workgroups = %.2f rounds of 1024; BN = 128: %u = %.2f of 768): %8.1f us %6.1f TF (%.3f)\n", s.name,
                   g64.x * g64.y, g64.x * g64.y / 1024.0, g.x * g.y, g.x * g.y / 768.0, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
        }
        for (int variant = 0; variant < 2; ++variant) {    // 0: as is (3 workgroups per CU); 1: no epilogue stores; 2, 3: the same at 1 per CU
            const int pad_kb = variant >= 2 ? 40 : 0;
            const int occ = 160 / (48 + pad_kb);
            c.relu = (variant & 1) ? 257 : 1;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipMemset(trace, 0, 64));
            conv2d_nhwc_glds<128, 3><<<g, 256, pad_kb * 1024>>>(c);
            CK(hipDeviceSynchronize());
            CK(hipMemset(trace, 0, 64));
            CK(hipEventRecord(e0));
            const int reps = 5;
            for (int r = 0; r < reps; ++r) conv2d_nhwc_glds<128, 3><<<g, 256, pad_kb * 1024>>>(c);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            unsigned long long t[6];
            CK(hipMemcpy(t, trace, 48, hipMemcpyDeviceToHost));
            int api = 0;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, conv2d_nhwc_glds<128, 3>, 256, pad_kb * 1024));
            const double fl = 2.0 * M * s.Cout * K;
            printf("%-32s %s wg/CU %d: %8.1f us %6.1f TF (%.3f)", s.name, (variant & 1) ? "no stores" : "         ", occ, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
            if (t[0]) printf("   per K step: top->MFMAs issued %6.0f  waits %5.0f  barrier+issue %5.0f  = %6.0f cycles (64 MFMAs = 2048 x %d waves/SIMD = %d)",
                             (double)t[1] / t[0], (double)t[2] / t[0], (double)t[3] / t[0], (double)(t[1] + t[2] + t[3]) / t[0], occ, 2048 * occ);
            printf("   resident workgroups: most at once %llu (API says %d per CU)\n", t[5], api);
        }
        CK(hipFree(X)); CK(hipFree(Wt)); CK(hipFree(Y)); CK(hipFree(B));
    }
    return 0;
}
