// ffn_shared_stage_variant.hip - the FIRST form of the fused feed-forward kernel (csrc/ffn_kernels.hip), kept for the record:
// conv2d_nhwc_glds's shared 16-KB stages and one barrier per K step, two workgroups of four waves per CU, BM = 64 or 32 tokens.
// Bit-identical to the two-launch path like the adopted eight-wave form, but no faster than the two launches (M = 76 800:
// 1.54-1.57 ms against 1.55-1.60): its steps are 32 / 64 MFMAs between barriers, and 1 200 tiles on 512 slots are 2.34 -> 3 rounds.
// Include after csrc/conv_kernels.hip and csrc/ffn_kernels.hip (tools/probes/ffn_probe.hip does).
#pragma once
template <int BM>      // tokens per workgroup: 64, or 32 for the tail of a launch shape (same bits)
__global__ void __launch_bounds__(256, 2) ffn_fused_glds(const FfnArgs a)
{
    constexpr int E = 256, HC = 128, K1 = E / 16, K2 = HC / 16, NS = 3, LPS = 4;
    constexpr int FMX = BM / 16;                 // token fragments of a wave (every wave covers all BM tokens)
    constexpr int GX = BM / 16;                  // 16-row DMA groups holding X rows in a first-product stage
    constexpr int STAGE_F4 = 256 * 4;
    constexpr int HROWS_F4 = BM * 4;             // one 16-k slice of H_c: [BM rows][4 quads]
    __shared__ __attribute__((aligned(1024))) float4 smem[NS * STAGE_F4 + K2 * HROWS_F4];
    float4 *const Hc = smem + NS * STAGE_F4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int fsw = (i >> 2) & 3;
    const long m0 = (long)a.m_begin + (long)blockIdx.x * BM;
    const int nchunk = a.F / HC;

    conv_u32x4 rx, rw1, rw2;
    {
        const unsigned long long bx = (unsigned long long)a.X, b1 = (unsigned long long)a.W1, b2 = (unsigned long long)a.W2;
        rx.x = (unsigned)bx; rx.y = (unsigned)(bx >> 32); rx.z = (unsigned)((long)a.M * E * 4); rx.w = 0x00020000u;
        rw1.x = (unsigned)b1; rw1.y = (unsigned)(b1 >> 32); rw1.z = (unsigned)((long)a.F * E * 4); rw1.w = 0x00020000u;
        rw2.x = (unsigned)b2; rw2.y = (unsigned)(b2 >> 32); rw2.z = (unsigned)((long)a.F * E * 4); rw2.w = 0x00020000u;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(const void *)smem;

    // DMA role: slot j of wave w fills the stage's 16-row group g = w + 4 j; lane l -> row 16 g + (l >> 2), quad position l & 3,
    // which holds k-quad (l & 3) ^ ((l >> 4) & 3)
    const int lkq = (lane & 3) ^ ((lane >> 4) & 3), lr = lane >> 2;
    unsigned off1[4], off2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = w + 4 * j;
        if (g < GX) {                                         // X rows
            const long row = m0 + 16 * g + lr;
            off1[j] = row < a.M ? (unsigned)((row * E + 4 * lkq) * 4) : 0x80000000u;
        } else if (g < GX + HC / 16) {                        // W1 rows of the chunk
            off1[j] = (unsigned)(((16 * (g - GX) + lr) * E + 4 * lkq) * 4);
        } else {
            off1[j] = 0x80000000u;
        }
        off2[j] = (unsigned)(((long)(16 * g + lr) * a.F + 4 * lkq) * 4);
    }
    int ic = 0, ir = 0, ibuf = 0;                             // the next stage to issue: chunk, step within the chunk, ring slot
    auto issue = [&]() {
        const unsigned sbase = lds0 + (unsigned)ibuf * (STAGE_F4 * 16);
        if (ir < K1) {
            const unsigned kb = (unsigned)ir * 64u, cb = (unsigned)ic * (HC * E * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = w + 4 * j;
                if (g < GX) conv_glds16(rx, off1[j] + kb, sbase + g * 1024);
                else conv_glds16(rw1, off1[j] + kb + cb, sbase + g * 1024);
            }
        } else {
            const unsigned kb = (unsigned)(ic * HC + (ir - K1) * 16) * 4u;
#pragma unroll
            for (int j = 0; j < 4; ++j) conv_glds16(rw2, off2[j] + kb, sbase + (w + 4 * j) * 1024);
        }
        if (++ir == K1 + K2) { ir = 0; ++ic; }
        if (++ibuf == NS) ibuf = 0;
    };

    f32x4 acc2[FMX][4];
#pragma unroll
    for (int x = 0; x < FMX; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc2[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int total = nchunk * (K1 + K2);
    int s = 0, buf = 0;
    issue();
    if (total > 1) issue();
    if (total > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int c = 0; c < nchunk; ++c) {
        // this chunk's bias quads, fetched behind the compiler's back (a visible load would make it drain vmcnt - and with it the
        // ring - before the first use); issued BEFORE the step's stage, so the step's counted wait covers them
        f32x4 bq[2];
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const float *p = a.b1 + c * HC + 32 * w + 16 * y + 4 * kk;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq[y]) : "v"(p) : "memory");
        }
        f32x4 acc1[FMX][2];
#pragma unroll
        for (int x = 0; x < FMX; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) acc1[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // ---- first product: H_c = X W1_c^T, this wave's 32 hidden units of the chunk for all BM tokens ----
        for (int q = 0; q < K1; ++q, ++s) {
            const bool steady = s + NS - 1 < total;
            if (steady) issue();
            const float4 *S = smem + buf * STAGE_F4;
            float4 af[FMX], bf[2];
#pragma unroll
            for (int x = 0; x < FMX; ++x) af[x] = S[(x * 16 + i) * 4 + (kk ^ fsw)];
#pragma unroll
            for (int y = 0; y < 2; ++y) bf[y] = S[(BM + 32 * w + 16 * y + i) * 4 + (kk ^ fsw)];
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc1[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].x, af[x].x, acc1[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc1[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].y, af[x].y, acc1[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc1[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].z, af[x].z, acc1[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc1[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].w, af[x].w, acc1[x][y], 0, 0, 0);
            // (the bias quads pass through the first step's wait: the compiler must keep them where the load will put them until then)
            if (q == 0) {
                if (steady) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bq[0]), "+v"(bq[1]) : "n"(LPS * (NS - 2)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]) :: "memory");
            } else {
                if (steady) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (q == K1 - 1) {
                // bias + ReLU, then the D fragment (lane = token i, hidden units 4 kk .. + 3 of fragment y) is exactly one k-quad of the
                // second product's operand: one 16-byte LDS store per fragment into slice 2 w + y.  Every wave left the previous
                // chunk's second product through a barrier, and this step's barrier publishes the slices.
#pragma unroll
                for (int x = 0; x < FMX; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) {
                        float4 v = make_float4(acc1[x][y][0] + bq[y][0], acc1[x][y][1] + bq[y][1], acc1[x][y][2] + bq[y][2], acc1[x][y][3] + bq[y][3]);
                        v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                        Hc[(2 * w + y) * HROWS_F4 + (x * 16 + i) * 4 + (kk ^ fsw)] = v;
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (++buf == NS) buf = 0;
        }
        // ---- second product: Y += H_c W2[:, chunk]^T, this wave's 64 output channels for all BM tokens ----
        for (int q = 0; q < K2; ++q, ++s) {
            const bool steady = s + NS - 1 < total;
            if (steady) issue();
            const float4 *S = smem + buf * STAGE_F4;
            const float4 *H = Hc + q * HROWS_F4;
            float4 af[FMX], bf[4];
#pragma unroll
            for (int x = 0; x < FMX; ++x) af[x] = H[(x * 16 + i) * 4 + (kk ^ fsw)];
#pragma unroll
            for (int y = 0; y < 4; ++y) bf[y] = S[(64 * w + 16 * y + i) * 4 + (kk ^ fsw)];
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc2[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].x, af[x].x, acc2[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc2[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].y, af[x].y, acc2[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc2[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].z, af[x].z, acc2[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc2[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[y].w, af[x].w, acc2[x][y], 0, 0, 0);
            if (steady) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (++buf == NS) buf = 0;
        }
    }
    // epilogue: lane = token i, output channels 4 kk .. + 3 of fragment y
#pragma unroll
    for (int x = 0; x < FMX; ++x) {
        const long row = m0 + x * 16 + i;
        if (row >= a.M) continue;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int co = 64 * w + 16 * y + 4 * kk;
            const float4 b = *(const float4 *)(a.b2 + co);
            *(float4 *)(a.Y + row * E + co) = make_float4(acc2[x][y][0] + b.x, acc2[x][y][1] + b.y, acc2[x][y][2] + b.z, acc2[x][y][3] + b.w);
        }
    }
}

