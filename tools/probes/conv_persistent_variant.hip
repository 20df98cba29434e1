// conv_persistent_variant.hip - a measured-and-not-adopted form of csrc/conv_kernels.hip conv2d_nhwc_glds, kept so that
// tools/probes/gemm_probe.hip reproduces the measurement (profiles/r5_conv_gemm_probe.txt).
// Included by gemm_probe.hip AFTER conv_kernels.hip (same ConvArgs, conv_glds16, f32x4).
// ------------------------------------------------------------------------------------------------
// The same kernel as a PERSISTENT tile loop (round 5).  Measured on the kernel above (tools/probes/gemm_probe.hip): a K step costs
// 2 800 cycles at one wave per SIMD against 2 048 of MFMAs and the pipe is 0.9 busy during K loops at two or three waves per
// SIMD - what is lost is lost per WORKGROUP: every workgroup of a round starts together, walks the same number of K steps and
// reaches its epilogue together, so the chip alternates between "everybody multiplies" and "everybody stores / fetches its
// first stage" (K = 256 GEMMs: 0.58-0.64 of peak with the stores, 0.71-0.76 without; the store pattern by itself writes
// 5.8 TB/s, tools/probes/store_probe.hip).  Here a workgroup stays (grid = what is resident at once) and walks its tiles with
// ONE stage ring across them: while the last NS - 1 K steps of a tile multiply, the first stages of the NEXT tile are already on
// their way into LDS, the epilogue's stores are issued and not waited for (the next tile's first counted vmcnt covers them),
// and nothing ever waits for a prologue again.  Same arithmetic per element as conv2d_nhwc_glds (same K order, same MFMA
// operand roles): bit-identical results.
// Tiles: XCD x (= blockIdx.x & 7: the observed placement) owns a contiguous run of the N-fastest tile order, the workgroups of
// an XCD take its tiles round-robin, so concurrently running workgroups share A rows / weight columns in ONE L2.
// ------------------------------------------------------------------------------------------------
template <int BN, int NS>
__global__ void __launch_bounds__(256, BN == 128 ? 3 : 4) conv2d_nhwc_pglds(const ConvArgs a)
{
    constexpr int BM = 128;
    constexpr int FM = (BN == 128) ? 4 : 2;
    constexpr int FN = 4;
    constexpr int WLD = BN / 64;
    constexpr int LPS = 2 + WLD;
    constexpr int STAGE_F4 = (BM + BN) * 4;
    __shared__ __attribute__((aligned(1024))) float4 smem[NS * STAGE_F4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int wm = (BN == 128) ? (w >> 1) : w, wn = (BN == 128) ? (w & 1) : 0;
    const unsigned M = (unsigned)((long)a.N * a.OH * a.OW);          // (the host keeps every byte offset below 2^31)
    const int XS = a.XS ? a.XS : a.Cin, WS = a.WS ? a.WS : a.KP, YS = a.YS ? a.YS : a.Cout;
    const unsigned tiles_n = (a.Cout + BN - 1) / BN, nt = ((M + BM - 1) / BM) * tiles_n;
    // this workgroup's tiles: first, first + stride, ... below t_end
    unsigned t_first, t_end, t_stride;
    {
        const unsigned G8 = gridDim.x >> 3, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        const unsigned per = nt >> 3, rem = nt & 7u;
        const unsigned t0 = xcd * per + min(xcd, rem);
        t_end = t0 + per + (xcd < rem ? 1u : 0u);
        t_first = t0 + slot;
        t_stride = G8;
    }
    if (t_first >= t_end) return;
#ifdef CONV_DEBUG                      // probe only: a.ksteps = start skew in cycles (workgroups begin spread over that span)
    if (a.ksteps > 0) {
        const long long t0 = clock64(), d = (long long)((blockIdx.x * 2654435761u) >> 22) * a.ksteps / 1024;
        while (clock64() - t0 < d) __builtin_amdgcn_s_sleep(16);
    }
#endif

    conv_u32x4 rx, rwt;
    {
        const unsigned long long bx = (unsigned long long)a.X, bw = (unsigned long long)a.Wt;
        rx.x = (unsigned)bx; rx.y = (unsigned)(bx >> 32);
        rx.z = (unsigned)((long)a.N * a.H * a.W * XS * 4); rx.w = 0x00020000u;
        rwt.x = (unsigned)bw; rwt.y = (unsigned)(bw >> 32);
        rwt.z = (unsigned)((long)a.Cout * WS * 4); rwt.w = 0x00020000u;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(const void *)smem;
    const int lkq = (lane & 3) ^ ((lane >> 4) & 3), lr = lane >> 2;
    const int nhex = a.KP >> 4;
    const bool pointwise = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0;    // output pixel p reads input pixel p

    // ---- the ISSUE side: the tile whose stages are being fetched (runs up to NS - 1 steps ahead of the compute side, across tiles)
    int iy0[2], ix0[2], xoff[2];        // a row past M: iy0 far outside the image - its loads are masked like any padding tap
    unsigned woff[WLD];
    int t_c0 = 0, t_dx = 0, t_dy = 0, iq = 0;
    unsigned it = t_first, gi = 0;      // issue tile, stages issued so far (ring slot = gi % NS)
    bool ihave = true;
    auto setup_issue = [&](unsigned tile) {
        const unsigned m0 = (tile / tiles_n) * BM;
        const int n0 = (int)(tile % tiles_n) * BN;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned p = m0 + 32 * w + 16 * j + lr;
            if (pointwise) {
                iy0[j] = p < M ? 0 : -0x100000;
                ix0[j] = 0;
                xoff[j] = (int)((p * (unsigned)XS + 4 * lkq) * 4);
            } else {
                const unsigned pc = p < M ? p : M - 1;
                const unsigned ox = pc % (unsigned)a.OW, tq = pc / (unsigned)a.OW;
                const unsigned oy = tq % (unsigned)a.OH, nimg = tq / (unsigned)a.OH;
                iy0[j] = p < M ? (int)oy * a.stride - a.pad : -0x100000;
                ix0[j] = (int)ox * a.stride - a.pad;
                xoff[j] = (int)(((((long)nimg * a.H + ((int)oy * a.stride - a.pad)) * a.W + ix0[j]) * XS + 4 * lkq) * 4);
            }
        }
#pragma unroll
        for (int j = 0; j < WLD; ++j) {
            const int r = n0 + (BN / 4) * w + 16 * j + lr;
            woff[j] = r < a.Cout ? (unsigned)(((long)r * WS + 4 * lkq) * 4) : 0x80000000u;
        }
        t_c0 = 0; t_dx = 0; t_dy = 0; iq = 0;
    };
    auto issue_one = [&]() -> bool {
        if (!ihave) return false;
        const unsigned sbase = lds0 + (gi % NS) * (STAGE_F4 * 16);
        const int toff = ((t_dy * a.W + t_dx) * XS + t_c0) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = (unsigned)(iy0[j] + t_dy) < (unsigned)a.H && (unsigned)(ix0[j] + t_dx) < (unsigned)a.W;
            conv_glds16(rx, ok ? (unsigned)(xoff[j] + toff) : 0x80000000u, sbase + (32 * w + 16 * j) * 64);
        }
#pragma unroll
        for (int j = 0; j < WLD; ++j)
            conv_glds16(rwt, woff[j] == 0x80000000u ? woff[j] : woff[j] + iq * 64, sbase + BM * 64 + ((BN / 4) * w + 16 * j) * 64);
        t_c0 += 16;
        if (t_c0 == a.Cin) {
            t_c0 = 0;
            if (++t_dx == a.KW) { t_dx = 0; ++t_dy; }
        }
        ++gi;
        if (++iq == nhex) {
            it += t_stride;
            if (it < t_end) setup_issue(it);
            else ihave = false;
        }
        return true;
    };

    setup_issue(it);
    int primed = 0;
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) primed += issue_one() ? 1 : 0;
    if (primed == NS - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int fsw = (i >> 2) & 3;
    const bool vec = (a.Cout & 3) == 0 && (YS & 3) == 0;
    unsigned gc = 0;                    // stages consumed so far
    for (unsigned ct = t_first; ct < t_end; ct += t_stride) {
        f32x4 acc[FM][FN];
#pragma unroll
        for (int x = 0; x < FM; ++x)
#pragma unroll
            for (int y = 0; y < FN; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < nhex; ++q) {
            const bool issued = issue_one();            // into the buffer every wave finished reading one step ago
            const float4 *As = smem + (gc % NS) * STAGE_F4;
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
#ifdef CONV_DEBUG                      // probe only (relu bit 9): one fragment store per K step inside the loop (partial sums - wrong values, the
                                       // instruction mix of a deferred epilogue): what stores cost when they are spread over the K loop
            if ((a.relu & 512) && q < FM * FN) {
                const unsigned m0d = (ct / tiles_n) * BM;
                const int n0d = (int)(ct % tiles_n) * BN, xd = q / FN, yd = q % FN;
                const unsigned ppd = m0d + wm * (FM * 16) + i;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int x = 0; x < FM; ++x)
#pragma unroll
                    for (int y = 0; y < FN; ++y)
                        if (x == xd && y == yd) v = make_float4(acc[x][y][0], acc[x][y][1], acc[x][y][2], acc[x][y][3]);
                if (ppd + xd * 16 < M) *(float4 *)(a.Y + (long)(ppd + xd * 16) * YS + n0d + wn * 64 + yd * 16 + 4 * kk) = v;
                if (issued) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2) + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else
#endif
            // the stage of the next step (this tile's or the next tile's first) must have landed before the barrier that lets every
            // wave read it; the epilogue's stores of the previous tile are older than it and are covered by the same count
            if (issued) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            ++gc;
        }
#ifdef CONV_DEBUG
        if (a.relu & (256 | 512)) { if (acc[0][0][0] == 12345.678f) a.Y[0] = acc[1][1][1] + acc[2][2][2] + acc[3][3][3]; continue; }
#endif
        // epilogue of tile ct: D fragment lane = (pixel column l&15, channel rows 4*(l>>4)+r); stores are not waited for
        const unsigned m0 = (ct / tiles_n) * BM;
        const int n0 = (int)(ct % tiles_n) * BN;
#pragma unroll
        for (int x = 0; x < FM; ++x) {
            const unsigned pp = m0 + wm * (FM * 16) + x * 16 + i;
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
                        const float4 r = *(const float4 *)(a.R + (long)pp * YS + co);
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                    if (a.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    *(float4 *)(a.Y + (long)pp * YS + co) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (co + r >= a.Cout) break;
                        float v = acc[x][y][r] + (a.bias ? a.bias[co + r] : 0.f);
                        if (a.R) v += a.R[(long)pp * YS + co + r];
                        if (a.relu) v = fmaxf(v, 0.f);
                        a.Y[(long)pp * YS + co + r] = v;
                    }
                }
            }
        }
    }
}

