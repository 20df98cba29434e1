// conv_kernels.hip - convolution path of the detector backbone (torchvision
// fasterrcnn_resnet50_fpn as built at reference object_detection/models.py:6-20 and called per frame at
// baselines/detector.py:71-86).  fp32, NHWC activations, implicit GEMM on v_mfma_f32_16x16x4_f32.
//
// The arithmetic of the network itself lives in torchvision 0.5.0, which is neither under
// /root/reference nor installed: these kernels are checked against a build-authored torch restatement
// (oracle/detector_oracle.py) - parity with the reference's detector is UNPINNED.
//
//   conv2d_nhwc      Y[p][co] = act( sum_{tap,ci} X[pix(p,tap)][ci] * W[co][tap][ci] + bias[co] (+ R[p][co]) )
//                    GEMM view: M = N*OH*OW pixels, N = Cout, K = KH*KW*Cin with k = tap*Cin + ci, so
//                    both operands are K-contiguous and one float4 per lane at k = 16q + 4(l>>4) feeds four
//                    consecutive MFMAs (the step kernels' "hexadecet" trick).  Cin % 4 == 0: for the stem
//                    (Cin = 3 -> 4) a float4 is one tap's channels, for Cin >= 16 a hexadecet lies in one
//                    tap.  Zero padding = masked loads.  BatchNorm (frozen) is folded into W / bias on load.
//   maxpool3x3s2     ResNet stem pool (kernel 3, stride 2, pad 1)
//   subsample2       FPN LastLevelMaxPool (kernel 1, stride 2)
//   upsample_add     FPN top-down: Y = lateral + nearest_upsample(top) to lateral's size
//   preprocess_frame detector.py:74-80 (BGR->RGB, /256) + GeneralizedRCNNTransform (normalize, bilinear
//                    resize align_corners=False, zero pad) -> NHWC with C = 4 (4th channel 0)
#pragma once
#include <hip/hip_runtime.h>

struct ConvArgs {
    const float *X;     // [N][H][W][Cin]
    const float *Wt;    // [Cout][KP]  KP = K padded to a multiple of 16, k = (dy*KW + dx)*Cin + ci
    const float *bias;  // [Cout]
    const float *R;     // residual [N][OH][OW][Cout] or null
    float *Y;           // [N][OH][OW][Cout]
    int N, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, KP, relu;
    // row strides in floats, 0 = dense (XS = Cin, WS = KP, YS = Cout).  Honoured by conv2d_nhwc_glds only: they let
    // the GEMMs of the encoder backward read / write column slices of wider matrices (one head of qkv, d qkv ...)
    int XS, WS, YS;
    // split K (conv2d_nhwc_glds only, ksplit >= 2): workgroup z walks K steps [z * ksteps, (z + 1) * ksteps) and stores its raw
    // 128 x BN sums to P[z][M][Cout]; conv_splitk_reduce adds the slices in order and applies bias / residual / ReLU
    float *P;
    int ksplit, ksteps;
    // conv2d_nhwc_glds only, RH > 0: R is a COARSER map [N][RH][RW][Cout] added nearest-upsampled (F.interpolate to the output's size):
    // the FPN's top-down step  lateral(x) + upsample(top)  in the lateral conv's own epilogue (FeaturePyramidNetwork.forward)
    int RH, RW;
    // conv2d_nhwc_glds only, X2 != null (1 x 1, stride 1, pad 0 on X): a SECOND 1 x 1 source [N][H2][W2][Cin2] sampled at stride2 joins
    // the same K loop - k < Cin from X, Cin <= k < Cin + Cin2 from X2, weights [Cout][Cin + Cin2]: conv3 and the downsample branch of a
    // stage's first bottleneck as ONE product (relu(bn3(conv3(out)) + bn_d(downsample(x))), torchvision Bottleneck.forward)
    const float *X2;
    int Cin2, H2, W2, stride2;
    // conv2d_nhwc_glds only, ksplit <= 1 and gridDim.z > 1: a BATCH of independent products of one shape - workgroup z works on
    // X + z * bsx, Wt + z * bsw, Y + z * bsy (floats): the 16 tile positions of a Winograd layer in one launch (csrc/wino_kernels.hip)
    long bsx, bsw, bsy;
#ifdef CONV_TRACE                 // tools/probes/gemm_probe.hip only: per-K-step cycle sums of wave 0 of every workgroup
    unsigned long long *trace;    // [0] steps, [1] top -> MFMAs issued, [2] -> waits done, [3] -> next top (barrier + DMA issue)
#endif
};

__global__ void __launch_bounds__(256) conv2d_nhwc(const ConvArgs a)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const long M = (long)a.N * a.OH * a.OW;
    const long m0 = (long)blockIdx.x * 64 + (w >> 1) * 32;
    const int n0 = blockIdx.y * 64 + (w & 1) * 32;
    const int K = a.KH * a.KW * a.Cin;
    // this lane's two A rows (output pixels) and two B rows (output channels)
    int iy0[2], ix0[2];
    const float *xn[2];
    bool pv[2];
    const float4 *w_row[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        long p = m0 + f * 16 + i;
        pv[f] = p < M;
        if (!pv[f]) p = M - 1;
        const int ox = p % a.OW;
        const long t = p / a.OW;
        const int oy = t % a.OH;
        const int n = t / a.OH;
        iy0[f] = oy * a.stride - a.pad;
        ix0[f] = ox * a.stride - a.pad;
        xn[f] = a.X + (long)n * a.H * a.W * a.Cin;
        const int co = min(n0 + f * 16 + i, a.Cout - 1);
        w_row[f] = (const float4 *)(a.Wt + (long)co * a.KP) + kk;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nhex = a.KP >> 4;
    for (int q = 0; q < nhex; ++q) {
        const int k4 = 16 * q + 4 * kk;           // this lane's 4 consecutive k
        const int tap = k4 / a.Cin, c = k4 - tap * a.Cin;
        const int dy = tap / a.KW, dx = tap - dy * a.KW;
        float4 av[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int y = iy0[f] + dy, x = ix0[f] + dx;
            const bool ok = k4 < K && y >= 0 && y < a.H && x >= 0 && x < a.W;
            av[f] = ok ? *(const float4 *)(xn[f] + ((long)y * a.W + x) * a.Cin + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 wv0 = w_row[0][q * 4], wv1 = w_row[1][q * 4];
        const float ae[2][4] = {{av[0].x, av[0].y, av[0].z, av[0].w}, {av[1].x, av[1].y, av[1].z, av[1].w}};
        const float we[2][4] = {{wv0.x, wv0.y, wv0.z, wv0.w}, {wv1.x, wv1.y, wv1.z, wv1.w}};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[x][e], we[y][e], acc[x][y], 0, 0, 0);
    }
    // D layout: lane holds column j = l&15 (output channel), rows 4*(l>>4) + r (pixels)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int co = n0 + y * 16 + i;
            if (co >= a.Cout) continue;
            const float b = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long p = m0 + x * 16 + kk * 4 + r;
                if (p < M) {
                    float v = acc[x][y][r] + b;
                    if (a.R) v += a.R[p * a.Cout + co];
                    if (a.relu) v = fmaxf(v, 0.f);
                    a.Y[p * a.Cout + co] = v;
                }
            }
        }
}

// C % 4 == 0 (the host checks): one float4 of channels per thread
__global__ void __launch_bounds__(256) maxpool3x3s2(const float *__restrict__ X, float *__restrict__ Y, int N,
                                                    int H, int W, int C, int OH, int OW)
{
    const int C4 = C >> 2;
    const long n_out = (long)N * OH * OW * C4;
    const float4 *X4 = (const float4 *)X;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n_out; idx += (long)gridDim.x * 256) {
        const int c = idx % C4;
        long t = idx / C4;
        const int ox = t % OW; t /= OW;
        const int oy = t % OH;
        const int n = t / OH;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int y = oy * 2 - 1 + dy, x = ox * 2 - 1 + dx;
                if (y >= 0 && y < H && x >= 0 && x < W) {
                    const float4 v = X4[(((long)n * H + y) * W + x) * C4 + c];
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
        ((float4 *)Y)[idx] = m;
    }
}

__global__ void __launch_bounds__(256) subsample2(const float *__restrict__ X, float *__restrict__ Y, int N, int H,
                                                  int W, int C, int OH, int OW)
{
    const long n_out = (long)N * OH * OW * C;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n_out; idx += (long)gridDim.x * 256) {
        const int c = idx % C;
        long t = idx / C;
        const int ox = t % OW; t /= OW;
        const int oy = t % OH;
        const int n = t / OH;
        Y[idx] = X[(((long)n * H + oy * 2) * W + ox * 2) * C + c];
    }
}

// Y[n][y][x][c] = L[n][y][x][c] + T[n][floor(y*TH/H)][floor(x*TW/W)][c]   (F.interpolate nearest to L's size)
__global__ void __launch_bounds__(256) upsample_add(const float *__restrict__ L, const float *__restrict__ T,
                                                    float *__restrict__ Y, int N, int H, int W, int C, int TH, int TW)
{
    const long n_out = (long)N * H * W * C;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n_out; idx += (long)gridDim.x * 256) {
        const int c = idx % C;
        long t = idx / C;
        const int x = t % W; t /= W;
        const int y = t % H;
        const int n = t / H;
        const int ty = min((int)(((long)y * TH) / H), TH - 1), tx = min((int)(((long)x * TW) / W), TW - 1);
        Y[idx] = L[idx] + T[(((long)n * TH + ty) * TW + tx) * C + c];
    }
}

// frame uint8 [H][W][3] BGR  ->  Y [PH][PW][4] fp32: ch c<3 = (rgb_c/256 - mean_c)/std_c resized bilinearly
// (align_corners = False, source index (dst + 0.5)/scale - 0.5 clamped at 0) to [RH][RW], zero outside.
__global__ void __launch_bounds__(256) preprocess_frame(const unsigned char *__restrict__ frame, float *__restrict__ Y,
                                                        int H, int W, int RH, int RW, int PH, int PW, float inv_sy,
                                                        float inv_sx, float m0, float m1, float m2, float s0, float s1,
                                                        float s2)
{
    const long n_out = (long)PH * PW;
    const float mean[3] = {m0, m1, m2}, istd[3] = {1.0f / s0, 1.0f / s1, 1.0f / s2};
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n_out; idx += (long)gridDim.x * 256) {
        const int x = idx % PW, y = idx / PW;
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y < RH && x < RW) {
            const float fy = fmaxf(((float)y + 0.5f) * inv_sy - 0.5f, 0.f);
            const float fx = fmaxf(((float)x + 0.5f) * inv_sx - 0.5f, 0.f);
            const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
            const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            float v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int cb = 2 - c;  // BGR -> RGB (cv2.COLOR_BGR2RGB, detector.py:75)
                auto px = [&](int yy, int xx) {
                    return ((float)frame[((long)yy * W + xx) * 3 + cb] / 256.0f - mean[c]) * istd[c];
                };
                const float top = px(y0, x0) * (1.f - lx) + px(y0, x1) * lx;
                const float bot = px(y1, x0) * (1.f - lx) + px(y1, x1) * lx;
                v[c] = top * (1.f - ly) + bot * ly;
            }
            out = make_float4(v[0], v[1], v[2], 0.f);
        }
        ((float4 *)Y)[idx] = out;
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-staged tile kernels: register-staged variant first (stem only), the LDS-DMA production kernel below
// ------------------------------------------------------------------------------------------------
// Workgroup tile BM=128 output pixels x BN (128 or 64) output channels, K walked 16 at a time.
// Per K step the workgroup stages the im2col patch slice A[128 x 16] and the weight slice W[BN x 16]
// through LDS once (each thread: coalesced float4 global loads issued BEFORE the MFMA phase of the current
// step, written to the other LDS buffer after it - the guide's issue-early / write-late split), and each
// wave reads its fragments with conflict-free ds_read_b128 (layout [k/4][row][4]: the 16 lanes of a k-group
// read 16 consecutive rows = one 256-B bank row).  BN=128: 2x2 waves of 64x64 (4x4 fragments, 64 MFMAs per
// K step per wave); BN=64: 4x1 waves of 32x64.  Global traffic per MFMA drops 4x vs conv2d_nhwc.
//
// conv2d_nhwc_tiled is the register-staged variant of the tile kernel for Cin % 16 != 0 (the stem, Cin = 4): a
// 16-wide K step straddles taps, so each lane derives (tap, channel) of its k-quad itself; everything else is
// served by conv2d_nhwc_glds below.  The MFMA is issued with the weight fragment as the A operand: D rows =
// channels, so a lane ends up with 4 consecutive channels of one pixel and the epilogue moves float4s.
#ifndef CONV_XCD
#define CONV_XCD 1
#endif
typedef unsigned conv_u32x4 __attribute__((ext_vector_type(4)));

// 16-byte load through a buffer descriptor: offsets past num_records (0xffffffff = "masked") return zeros, so
// padding taps, rows past M and channels past Cout need no branch - straight-line loads that the compiler can
// keep in flight across K steps with counted waits
__device__ __forceinline__ float4 conv_bload(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    const conv_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// X and the packed weights must each be < 2 GiB (32-bit byte offsets, top bit = masked); the host falls back to
// conv2d_nhwc otherwise.
template <int BN>
__global__ void __launch_bounds__(256) conv2d_nhwc_tiled(const ConvArgs a)
{
    constexpr int BM = 128;
    constexpr int FM = (BN == 128) ? 4 : 2;   // 16-row fragments per wave along M
    constexpr int FN = 4;                     // ... along N
    constexpr int WROWS = BN;                 // weight rows staged per step
    constexpr int WLD = BN / 64;              // weight float4 loads per thread per step (2 or 1)
    __shared__ __attribute__((aligned(16))) float4 As[2][4][BM];
    __shared__ __attribute__((aligned(16))) float4 Ws[2][4][WROWS];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int wm = (BN == 128) ? (w >> 1) : w, wn = (BN == 128) ? (w & 1) : 0;
    const long M = (long)a.N * a.OH * a.OW;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.
    // Give every XCD one contiguous run of tiles, N-tiles of a pixel tile adjacent, so the 3x3 halo rows and the
    // pixel tile shared by the N-tiles are fetched into ONE L2 (and the neighbour's fetch acts as a prefetch).
#if CONV_XCD
    const unsigned nt = gridDim.x * gridDim.y;
    unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    {
        const unsigned per = nt >> 3, rem = nt & 7u, xcd = lin & 7u, slot = lin >> 3;
        lin = xcd * per + min(xcd, rem) + slot;          // XCDs < rem own one extra tile
    }
    const long m0 = (long)(lin / gridDim.y) * BM;
    const int n0 = (lin % gridDim.y) * BN;
#else
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
#endif
    const int K = a.KH * a.KW * a.Cin;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.X, 0, (unsigned)((long)a.N * a.H * a.W * a.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.Wt, 0, (unsigned)((long)a.Cout * a.KP * 4), 0x00020000);

    // loader role: thread -> k-quad (tid & 3) of rows (tid >> 2) and (tid >> 2) + 64, so the 4 lanes of a row
    // read one contiguous 64-byte run (16 rows x 64 B per wave load, like the fragment-shaped direct loads)
    const int lkq = tid & 3, lr0 = tid >> 2;
    int iy0[2], ix0[2];
    int xoff[2];          // byte offset of (image, iy0, ix0, channel 4*lkq); may be negative at the border
    bool prow_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        long p = m0 + lr0 + 64 * j;
        prow_ok[j] = p < M;
        if (!prow_ok[j]) p = M - 1;
        const int ox = p % a.OW;
        const long tq = p / a.OW;
        const int oy = tq % a.OH;
        const int nimg = tq / a.OH;
        iy0[j] = oy * a.stride - a.pad;
        ix0[j] = ox * a.stride - a.pad;
        xoff[j] = (int)((((long)nimg * a.H + iy0[j]) * a.W + ix0[j]) * a.Cin * 4);
    }
    unsigned woff[2];
#pragma unroll
    for (int j = 0; j < WLD; ++j) {
        const int r = n0 + lr0 + 64 * j;
        woff[j] = r < a.Cout ? (unsigned)(((long)r * a.KP + 4 * lkq) * 4) : 0xffffffffu;
    }

    auto load_a2 = [&](int q, float4 (&ra)[2]) {
        const int k4 = 16 * q + 4 * lkq;
        const int tap = k4 / a.Cin, c = k4 - tap * a.Cin;
        const int dy = tap / a.KW, dx = tap - dy * a.KW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int y = iy0[j] + dy, x = ix0[j] + dx;
            const bool ok = prow_ok[j] && k4 < K && y >= 0 && y < a.H && x >= 0 && x < a.W;
            ra[j] = conv_bload(rx, ok ? (unsigned)(xoff[j] + ((dy * a.W + dx) * a.Cin + c) * 4) : 0xffffffffu);
        }
    };
    auto load_w2 = [&](int q, float4 (&rw)[2]) {
#pragma unroll
        for (int j = 0; j < WLD; ++j) rw[j] = conv_bload(rwt, woff[j] == 0xffffffffu ? woff[j] : woff[j] + q * 64);
    };
    auto park = [&](int buf, const float4 (&ra)[2], const float4 (&rw)[2]) {
        As[buf][lkq][lr0] = ra[0];
        As[buf][lkq][lr0 + 64] = ra[1];
#pragma unroll
        for (int j = 0; j < WLD; ++j) Ws[buf][lkq][lr0 + 64 * j] = rw[j];
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int x = 0; x < FM; ++x)
#pragma unroll
        for (int y = 0; y < FN; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nhex = a.KP >> 4;
    float4 ra[2], rw[2];
    load_a2(0, ra);
    load_w2(0, rw);
    park(0, ra, rw);
    __syncthreads();

    for (int q = 0; q < nhex; ++q) {
        const int cur = q & 1;
        const bool more = q + 1 < nhex;
        if (more) {   // issue the next step's global loads before this step's MFMAs
            load_a2(q + 1, ra);
            load_w2(q + 1, rw);
        }
        float4 af[FM], bf[FN];
#pragma unroll
        for (int x = 0; x < FM; ++x) af[x] = As[cur][kk][wm * (FM * 16) + x * 16 + i];
#pragma unroll
        for (int y = 0; y < FN; ++y) bf[y] = Ws[cur][kk][wn * 64 + y * 16 + i];
        // k element outermost: 8..16 independent accumulators between two uses of the same one (the
        // 16x16x4 f32 MFMA has a 40-cycle dependent latency against a 32-cycle issue interval)
#pragma unroll
        for (int x = 0; x < FM; ++x)
#pragma unroll
            for (int y = 0; y < FN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].x, af[x].x, acc[x][y], 0, 0, 0);
#pragma unroll
        for (int x = 0; x < FM; ++x)
#pragma unroll
            for (int y = 0; y < FN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].y, af[x].y, acc[x][y], 0, 0, 0);
#pragma unroll
        for (int x = 0; x < FM; ++x)
#pragma unroll
            for (int y = 0; y < FN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].z, af[x].z, acc[x][y], 0, 0, 0);
#pragma unroll
        for (int x = 0; x < FM; ++x)
#pragma unroll
            for (int y = 0; y < FN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].w, af[x].w, acc[x][y], 0, 0, 0);
        if (more) park(cur ^ 1, ra, rw);   // ... and park them in the other buffer after the MFMAs
        __syncthreads();
    }
    // epilogue: D fragment lane = (pixel column l&15, channel rows 4*(l>>4)+r)
    const bool vec = (a.Cout & 3) == 0;
#pragma unroll
    for (int x = 0; x < FM; ++x) {
        const long pp = m0 + wm * (FM * 16) + x * 16 + i;
        if (pp >= M) continue;
#pragma unroll
        for (int y = 0; y < FN; ++y) {
            const int co = n0 + wn * 64 + y * 16 + 4 * kk;
            if (co >= a.Cout) continue;
            if (vec) {
                float4 v = make_float4(acc[x][y][0], acc[x][y][1], acc[x][y][2], acc[x][y][3]);
                if (a.bias) {
                    const float4 b = *(const float4 *)(a.bias + co);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                if (a.R) {
                    const float4 r = *(const float4 *)(a.R + pp * a.Cout + co);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if (a.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                *(float4 *)(a.Y + pp * a.Cout + co) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (co + r >= a.Cout) break;
                    float v = acc[x][y][r] + (a.bias ? a.bias[co + r] : 0.f);
                    if (a.R) v += a.R[pp * a.Cout + co + r];
                    if (a.relu) v = fmaxf(v, 0.f);
                    a.Y[pp * a.Cout + co + r] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (Cin % 16 == 0): the staged slices go global -> LDS directly (buffer_load_dwordx4 ... lds),
// NS stages deep, with counted vmcnt waits and one raw s_barrier per K step.  No staging VGPRs, no ds_write pass,
// and the prefetch distance (NS-1 steps) is no longer tied to a register ring the compiler will not build.
// LDS image per stage: row-major [row][4 k-quads] float4 with the quad position XOR-swizzled by (row >> 2) & 3 -
// the DMA destination is lane-linear (lane l -> 16 B at M0 + 16 l), so the swizzle is applied on the SOURCE side
// (which k-quad a lane fetches); 4 lanes still fetch one contiguous 64-B run of a row, and the 16 lanes of a
// fragment read (16 consecutive rows, one k-quad) hit 16 distinct 16-B bank groups.
// Masked lanes (padding taps, rows past M, channels past Cout) use an offset past num_records: the DMA writes zeros.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void conv_glds16(conv_u32x4 rsrc, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

// row of the residual operand that output row pp adds: pp itself, or its nearest-neighbour source row in a coarser map
__device__ __forceinline__ long conv_rrow(const ConvArgs &a, long pp)
{
    if (a.RH == 0) return pp;
    const int ox = (int)(pp % a.OW);
    const long t = pp / a.OW;
    const int oy = (int)(t % a.OH);
    const long n = t / a.OH;
    const int ty = min((int)(((long)oy * a.RH) / a.OH), a.RH - 1), tx = min((int)(((long)ox * a.RW) / a.OW), a.RW - 1);
    return (n * a.RH + ty) * a.RW + tx;
}

// C4 = true: Cin == 4 (the stem: 7 x 7 over the 4-channel frame).  A k-quad is then ONE TAP's four channels - still one aligned 16-B
// run in memory - so the same DMA stages it; only the source walk differs: the lane's quad position selects the tap (4 q + quad of
// stage q), each lane advances its own (dy, dx) by four taps per stage, taps past KH x KW are masked (their weights are zero padding).
template <int BN, int NS, bool C4 = false>
__global__ void __launch_bounds__(256) conv2d_nhwc_glds(const ConvArgs a)
{
    constexpr int BM = 128;
    constexpr int FM = (BN == 128) ? 4 : 2;
    constexpr int FN = 4;
    constexpr int WLD = BN / 64;                       // weight DMA instructions per wave per stage
    constexpr int LPS = 2 + WLD;                       // DMA instructions per wave per stage
    constexpr int STAGE_F4 = (BM + BN) * 4;            // float4 slots per stage: A rows then W rows
    __shared__ __attribute__((aligned(1024))) float4 smem[NS * STAGE_F4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int wm = (BN == 128) ? (w >> 1) : w, wn = (BN == 128) ? (w & 1) : 0;
    const long M = (long)a.N * a.OH * a.OW;
    const int XS = a.XS ? a.XS : a.Cin, WS = a.WS ? a.WS : a.KP, YS = a.YS ? a.YS : a.Cout;
    const unsigned nt = gridDim.x * gridDim.y;
    unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    {   // XCD-aware tile order, see conv2d_nhwc_tiled
        const unsigned per = nt >> 3, rem = nt & 7u, xcd = lin & 7u, slot = lin >> 3;
        lin = xcd * per + min(xcd, rem) + slot;
    }
    const long m0 = (long)(lin / gridDim.y) * BM;
    const int n0 = (lin % gridDim.y) * BN;

    const long bz = a.ksplit > 1 ? 0 : (long)blockIdx.z;     // batch index (0 for every launch that is not a batch)
    conv_u32x4 rx, rwt;
    {
        const unsigned long long bx = (unsigned long long)(a.X + bz * a.bsx), bw = (unsigned long long)(a.Wt + bz * a.bsw);
        rx.x = (unsigned)bx; rx.y = (unsigned)(bx >> 32);
        rx.z = (unsigned)((long)a.N * a.H * a.W * XS * 4); rx.w = 0x00020000u;
        rwt.x = (unsigned)bw; rwt.y = (unsigned)(bw >> 32);
        rwt.z = (unsigned)((long)a.Cout * WS * 4); rwt.w = 0x00020000u;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(const void *)smem;   // LDS byte address of the staging area

    // DMA role: wave w, instruction j covers rows 32w+16j .. +15 (A) / (BN/4)w+16j .. (W); lane l -> row + (l >> 2),
    // quad position l & 3, which holds k-quad (l & 3) ^ ((row >> 2) & 3) = (l & 3) ^ ((l >> 4) & 3)
    const int lkq = (lane & 3) ^ ((lane >> 4) & 3), lr = lane >> 2;
    int iy0[2], ix0[2], xoff[2];
    bool prow_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        long p = m0 + 32 * w + 16 * j + lr;
        prow_ok[j] = p < M;
        if (!prow_ok[j]) p = M - 1;
        const int ox = p % a.OW;
        const long tq = p / a.OW;
        const int oy = tq % a.OH;
        const int nimg = tq / a.OH;
        iy0[j] = oy * a.stride - a.pad;
        ix0[j] = ox * a.stride - a.pad;
        xoff[j] = (int)(((((long)nimg * a.H + iy0[j]) * a.W + ix0[j]) * XS + (C4 ? 0 : 4 * lkq)) * 4);
    }
    unsigned woff[WLD];
#pragma unroll
    for (int j = 0; j < WLD; ++j) {
        const int r = n0 + (BN / 4) * w + 16 * j + lr;
        woff[j] = r < a.Cout ? (unsigned)(((long)r * WS + 4 * lkq) * 4) : 0x80000000u;
    }
    // second source (dual product): its descriptor and this lane's row offsets
    conv_u32x4 rx2 = rx;
    unsigned aoff2[2] = {0x80000000u, 0x80000000u};
    if (!C4 && a.X2) {
        const unsigned long long b2 = (unsigned long long)a.X2;
        rx2.x = (unsigned)b2; rx2.y = (unsigned)(b2 >> 32);
        rx2.z = (unsigned)((long)a.N * a.H2 * a.W2 * a.Cin2 * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            long p = m0 + 32 * w + 16 * j + lr;
            if (p >= M) continue;
            const int ox = p % a.OW;
            const long tq = p / a.OW;
            const int oy = tq % a.OH;
            const long nimg = tq / a.OH;
            aoff2[j] = (unsigned)((((nimg * a.H2 + (long)oy * a.stride2) * a.W2 + (long)ox * a.stride2) * a.Cin2 + 4 * lkq) * 4);
        }
    }

    // K steps of this workgroup: all of them, or slice blockIdx.z of a split
    const int nhex_all = a.KP >> 4;
    const int qb = a.ksplit > 1 ? (int)blockIdx.z * a.ksteps : 0;
    const int nhex = a.ksplit > 1 ? min(nhex_all - qb, a.ksteps) : nhex_all;
    int t_c0, t_dx, t_dy;               // wave-uniform tap walk of the next stage to issue
    {
        const int k0 = qb * 16, tap = k0 / a.Cin;
        t_c0 = k0 - tap * a.Cin;
        t_dy = tap / a.KW;
        t_dx = tap - t_dy * a.KW;
    }
    // Per-lane work of a stage issue is ONE add per DMA: the tap's masked base offset (aoff: the bounds tests and the select) is
    // recomputed only when the tap changes - every Cin / 16 steps - and a masked offset (>= 2^31: past num_records, the DMA writes
    // zeros) stays masked under the small per-step additions.  fp32 MFMA runs on the VALU lanes: every VALU instruction of the K
    // loop is taken from the MFMA stream (round 5: ~20 -> 4 per step).
    unsigned aoff[2];
    auto tap_offsets = [&]() {
        const int toff = ((t_dy * a.W + t_dx) * XS) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = prow_ok[j] && (unsigned)(iy0[j] + t_dy) < (unsigned)a.H && (unsigned)(ix0[j] + t_dx) < (unsigned)a.W;
            aoff[j] = ok ? (unsigned)(xoff[j] + toff) : 0x80000000u;
        }
    };
    if constexpr (!C4) tap_offsets();
    int l_dy = (qb * 4 + lkq) / a.KW, l_dx = (qb * 4 + lkq) - l_dy * a.KW;      // C4: this lane's tap of the next stage to issue
    auto issue = [&](int q) {           // q: step within the slice
        const unsigned sbase = lds0 + (unsigned)(q % NS) * (STAGE_F4 * 16);
        const unsigned cb = (unsigned)t_c0 * 4u, wb = (unsigned)(qb + q) * 64u;      // (wave-uniform)
        if constexpr (C4) {
            const int toff = (l_dy * a.W + l_dx) * 16;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = prow_ok[j] && l_dy < a.KH && (unsigned)(iy0[j] + l_dy) < (unsigned)a.H && (unsigned)(ix0[j] + l_dx) < (unsigned)a.W;
                conv_glds16(rx, ok ? (unsigned)(xoff[j] + toff) : 0x80000000u, sbase + (32 * w + 16 * j) * 64);
            }
#pragma unroll
            for (int j = 0; j < WLD; ++j) conv_glds16(rwt, woff[j] + wb, sbase + BM * 64 + ((BN / 4) * w + 16 * j) * 64);
            l_dx += 4;
            while (l_dx >= a.KW) { l_dx -= a.KW; ++l_dy; }
            return;
        }
        if (a.X2 && (qb + q) * 16 >= a.Cin) {            // (wave-uniform) the second source's channels (qb + q) * 16 - Cin ..
            const unsigned cb2 = (unsigned)((qb + q) * 16 - a.Cin) * 4u;
#pragma unroll
            for (int j = 0; j < 2; ++j) conv_glds16(rx2, aoff2[j] + cb2, sbase + (32 * w + 16 * j) * 64);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) conv_glds16(rx, aoff[j] + cb, sbase + (32 * w + 16 * j) * 64);
        }
#pragma unroll
        for (int j = 0; j < WLD; ++j) conv_glds16(rwt, woff[j] + wb, sbase + BM * 64 + ((BN / 4) * w + 16 * j) * 64);
        t_c0 += 16;
        if (t_c0 == a.Cin) {
            t_c0 = 0;
            if (++t_dx == a.KW) { t_dx = 0; ++t_dy; }
            tap_offsets();
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int x = 0; x < FM; ++x)
#pragma unroll
        for (int y = 0; y < FN; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (Measured and not adopted, round 5: fetching the residual operand of the short-K expanding 1 x 1 convs HERE, ahead of the first
    // stage - the narrow tile has the 32 registers - instead of in the epilogue.  It made them 13 - 35 % SLOWER (layer1 conv3 645 ->
    // 870 us, layer2 391 -> 512, layer3 293 -> 343): the first MFMA then waits for 32 KB of residual behind the 12 KB stage, and the
    // other two resident workgroups were already covering the epilogue's latency.)
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nhex) issue(s0);
    // stage 0 landed (this wave's part) when at most the later prologue stages are outstanding
    if (nhex >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int fsw = (i >> 2) & 3;                       // fragment rows are 16-aligned: (row >> 2) & 3 = (i >> 2) & 3
#ifdef CONV_TRACE
    unsigned long long tr_t0 = 0, tr_t1 = 0, tr_t2 = 0, tr_prev2 = 0, tr_a = 0, tr_b = 0, tr_c = 0;
    if (a.trace && tid == 0) {          // census of resident workgroups: [4] now, [5] the most seen at once
        const unsigned long long now = atomicAdd(a.trace + 4, 1ull) + 1;
        atomicMax(a.trace + 5, now);
    }
#endif
    for (int q = 0; q < nhex; ++q) {
#ifdef CONV_TRACE
        asm volatile("s_memtime %0" : "=s"(tr_t0));
#endif
        const bool steady = q + NS - 1 < nhex;
        if (steady) issue(q + NS - 1);                  // into the buffer every wave finished reading at step q-1
        const float4 *As = smem + (q % NS) * STAGE_F4;
        const float4 *Ws = As + BM * 4;
        float4 af[FM], bf[FN];
#pragma unroll
        for (int x = 0; x < FM; ++x) af[x] = As[(wm * (FM * 16) + x * 16 + i) * 4 + (kk ^ fsw)];
#pragma unroll
        for (int y = 0; y < FN; ++y) bf[y] = Ws[(wn * 64 + y * 16 + i) * 4 + (kk ^ fsw)];
#pragma unroll
        for (int x = 0; x < FM; ++x)
#pragma unroll
            for (int y = 0; y < FN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].x, af[x].x, acc[x][y], 0, 0, 0);
#pragma unroll
        for (int x = 0; x < FM; ++x)
#pragma unroll
            for (int y = 0; y < FN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].y, af[x].y, acc[x][y], 0, 0, 0);
#pragma unroll
        for (int x = 0; x < FM; ++x)
#pragma unroll
            for (int y = 0; y < FN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].z, af[x].z, acc[x][y], 0, 0, 0);
#pragma unroll
        for (int x = 0; x < FM; ++x)
#pragma unroll
            for (int y = 0; y < FN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].w, af[x].w, acc[x][y], 0, 0, 0);
        // stage q+1 must have landed before the barrier that lets every wave read it: in steady state NS-2 later
        // stages stay in flight; in the tail (nothing new issued) simply drain
#ifdef CONV_TRACE
        asm volatile("s_memtime %0" : "=s"(tr_t1));
#endif
        if (steady) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef CONV_TRACE
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_t2), "+s"(tr_t0), "+s"(tr_t1) :: "memory");
        if (q > 0) tr_c += tr_t0 - tr_prev2;
        tr_a += tr_t1 - tr_t0; tr_b += tr_t2 - tr_t1; tr_prev2 = tr_t2;
#endif
        __builtin_amdgcn_s_barrier();
    }
#ifdef CONV_TRACE
    if (a.trace && tid == 0) atomicAdd(a.trace + 4, ~0ull);
    if (a.trace && tid == 0 && (blockIdx.x & 31) == 0) {
        atomicAdd(a.trace + 0, (unsigned long long)nhex);
        atomicAdd(a.trace + 1, tr_a); atomicAdd(a.trace + 2, tr_b); atomicAdd(a.trace + 3, tr_c);
    }
#endif
#ifdef CONV_DEBUG                      // probe only: bit 8 of relu = leave without storing (what the epilogue costs)
    if (a.relu & 256) { if (acc[0][0][0] == 12345.678f) a.Y[0] = acc[1][1][1] + acc[2][2][2] + acc[3][3][3]; return; }
#endif
    // epilogue: D fragment lane = (pixel column l&15, channel rows 4*(l>>4)+r)
    if (a.ksplit > 1) {                 // raw sums of this K slice (Cout % 4 == 0, dense rows: the host's split plan)
        float *P = a.P + (long)blockIdx.z * M * a.Cout;
#pragma unroll
        for (int x = 0; x < FM; ++x) {
            const long pp = m0 + wm * (FM * 16) + x * 16 + i;
            if (pp >= M) continue;
#pragma unroll
            for (int y = 0; y < FN; ++y) {
                const int co = n0 + wn * 64 + y * 16 + 4 * kk;
                if (co < a.Cout) *(float4 *)(P + pp * a.Cout + co) = make_float4(acc[x][y][0], acc[x][y][1], acc[x][y][2], acc[x][y][3]);
            }
        }
        return;
    }
    const bool vec = (a.Cout & 3) == 0 && (YS & 3) == 0;
    float *const Yb = a.Y + bz * a.bsy;
#pragma unroll
    for (int x = 0; x < FM; ++x) {
        const long pp = m0 + wm * (FM * 16) + x * 16 + i;
        if (pp >= M) continue;
#pragma unroll
        for (int y = 0; y < FN; ++y) {
            const int co = n0 + wn * 64 + y * 16 + 4 * kk;
            if (co >= a.Cout) continue;
            if (vec) {
                float4 v = make_float4(acc[x][y][0], acc[x][y][1], acc[x][y][2], acc[x][y][3]);
                if (a.bias) {
                    const float4 b = *(const float4 *)(a.bias + co);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                if (a.R) {
                    const float4 r = *(const float4 *)(a.R + conv_rrow(a, pp) * YS + co);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if (a.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                *(float4 *)(Yb + pp * YS + co) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (co + r >= a.Cout) break;
                    float v = acc[x][y][r] + (a.bias ? a.bias[co + r] : 0.f);
                    if (a.R) v += a.R[conv_rrow(a, pp) * YS + co + r];
                    if (a.relu) v = fmaxf(v, 0.f);
                    Yb[pp * YS + co + r] = v;
                }
            }
        }
    }
}

// Y = act( sum_z P[z] + bias (+ R) ): the K slices of a split conv2d_nhwc_glds added in slice order (one float4 per thread)
__global__ void __launch_bounds__(256) conv_splitk_reduce(const float *__restrict__ P, int S, long M, int Cout,
                                                          const float *__restrict__ bias, const float *__restrict__ R,
                                                          float *__restrict__ Y, int relu)
{
    const long n4 = M * Cout / 4, slice4 = n4;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < n4; g += (long)gridDim.x * 256) {
        float4 v = ((const float4 *)P)[g];
        for (int z = 1; z < S; ++z) {
            const float4 t = ((const float4 *)P)[(long)z * slice4 + g];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const int co = (int)((g * 4) % Cout);
        if (bias) {
            const float4 b = *(const float4 *)(bias + co);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (R) {
            const float4 r = ((const float4 *)R)[g];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        ((float4 *)Y)[g] = v;
    }
}
