// ffn_kernels.hip - the feed-forward block of a post-LN nn.TransformerEncoderLayer as ONE kernel:
//     Y[m][:] = W2 relu(W1 X[m][:] + b1) + b2          (reference baselines/learned_models.py:166-171, 184: the encoder of
//     TransformerLstm; torch's TransformerEncoderLayer.forward: linear2(dropout(relu(linear1(src)))), eval mode)
// for the throughput form of a served pass (opseq_encoder_layer_batched_f32).  As two launches of conv2d_nhwc_glds the block
// writes and re-reads the [M][ffn] hidden activations - 629 MB per 256 one-clip requests and layer - and a K = 256 product pays
// its store stream in full (compute and the write stream do not overlap on this memory system at 4 B per
// 512 flop).  Here the hidden activations never leave the CU: a workgroup owns 64 tokens whose rows stay in LDS, walks the ffn
// units in chunks of 128, computes the chunk's activations H_c = relu(X W1_c^T + b1_c) [64 x 128] into LDS and at once multiplies
// them into its [64 x 256] output accumulators, Y += H_c W2[:, c]^T.
//
// Arithmetic, element for element, is the two-launch path's: the same v_mfma_f32_16x16x4_f32 fragments (weights as the A
// operand), K walked in ascending 16-steps with lane group kk holding k = 16 q + 4 kk .. + 3, MFMA e of a step taking element e
// of every quad; bias added to the finished sum, then ReLU.  The result is therefore BIT-IDENTICAL to linear1 -> ReLU -> linear2
// on conv2d_nhwc_glds (tests/test_ffn_fused_gpu.py).
//
// One workgroup of EIGHT waves per CU (two per SIMD), and no barrier inside a product: the weight rows a wave multiplies are its
// own - wave w owns hidden units 16 w .. + 15 of every chunk and output channels 32 w .. + 31 - so each wave streams them through a
// PRIVATE three-stage ring of LDS-DMA stages (buffer_load_dwordx4 ... lds, conv2d_nhwc_glds's swizzled [row][4 k-quads] image,
// counted vmcnt waits on its own loads only); what the waves share is read-only while they share it: the token rows (staged once
// per workgroup) and H_c, which is fenced by the chunk's two barriers (all second products done | H_c written).  A first version
// with conv2d_nhwc_glds's shared stages and one barrier per K step (two workgroups of four waves per CU) was bit-identical too
// but no faster than the two launches (tools/probes/ffn_probe.hip).
#pragma once
#include <hip/hip_runtime.h>

struct FfnArgs {
    const float *X;    // [M][256]
    const float *W1;   // [F][256]   linear1.weight
    const float *b1;   // [F]
    const float *W2;   // [256][F]   linear2.weight
    const float *b2;   // [256]
    float *Y;          // [M][256]
    int M, F;
    int m_begin;       // first token row of this launch
    // ffn_fused_w8: workgroups 0 .. n_full - 1 own 64 tokens each; the rest - what the full rounds of 64-token tiles over the CUs
    // leave - is cut into equal tiles of 16 * tail_frags tokens, one per CU, dispatched last (blockIdx >= n_full)
    int n_full, tail_frags;
#ifdef FFN_TRACE       // tools/probes/ffn_probe.hip only: per workgroup (s_memtime ticks, s_memrealtime ticks of 10 ns) of its whole tile
    unsigned long long *clk;
#endif
};

// ------------------------------------------------------------------------------------------------
// the eight-wave form (see the header): one workgroup of 512 threads per CU, 64 tokens, private weight rings, two barriers per chunk.
// The 24 steps of a chunk are unrolled (ring slots, stage kinds and wait counts are then constants: 24 % 3 == 0), and a step's
// fragments are read one step ahead, between the first and the second quarter of the previous step's MFMAs.
// ------------------------------------------------------------------------------------------------
constexpr int FFN_W8_LDS_F4 = 16 * 256 + 8 * 256 + 8 * 3 * 128;      // token rows + H_c + eight private rings: 144 KB

template <int FMX>     // 16-token fragments of the tile: 4, or fewer for the balanced tail tiles (same bits per token)
__device__ __forceinline__ void ffn_tile_w8(const FfnArgs &a, float4 *smem, const long m0)
{
    constexpr int E = 256, HC = 128, BM = 64, K1 = E / 16, K2 = HC / 16, NS = 3;
    constexpr int SLICE_F4 = BM * 4;             // one 16-k slice of 64 rows: [row][4 quads] float4 = 4 KB
    constexpr int RSTAGE_F4 = 32 * 4;            // a wave's ring stage: up to 32 weight rows x 16 k = 2 KB
    static_assert((K1 + K2) % NS == 0, "a chunk's first stage must land in ring slot 0");
    static_assert(K1 * SLICE_F4 + K2 * SLICE_F4 + 8 * NS * RSTAGE_F4 == FFN_W8_LDS_F4, "LDS layout");
    float4 *const Xs = smem;                     // [K1 slices]   the tile's token rows, staged once
    float4 *const Hc = smem + K1 * SLICE_F4;     // [K2 slices]   relu(X W1_c^T + b1_c) of the current chunk

#ifdef FFN_TRACE
    unsigned long long tr_c0, tr_r0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_c0), "=s"(tr_r0) :: "memory");
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int fsw = (i >> 2) & 3;
    const int nchunk = a.F / HC;
    float4 *const ring = smem + (K1 + K2) * SLICE_F4 + w * (NS * RSTAGE_F4);

    conv_u32x4 rx, rw1, rw2;
    {
        const unsigned long long bx = (unsigned long long)a.X, b1 = (unsigned long long)a.W1, b2 = (unsigned long long)a.W2;
        rx.x = (unsigned)bx; rx.y = (unsigned)(bx >> 32); rx.z = (unsigned)((long)a.M * E * 4); rx.w = 0x00020000u;
        rw1.x = (unsigned)b1; rw1.y = (unsigned)(b1 >> 32); rw1.z = (unsigned)((long)a.F * E * 4); rw1.w = 0x00020000u;
        rw2.x = (unsigned)b2; rw2.y = (unsigned)(b2 >> 32); rw2.z = (unsigned)((long)a.F * E * 4); rw2.w = 0x00020000u;
    }
    const unsigned lds_x = (unsigned)(unsigned long long)(const void *)Xs;
    const unsigned lds_ring = (unsigned)(unsigned long long)(const void *)ring;

    // DMA lane role: lane l -> row (l >> 2) of a 16-row group, quad position l & 3, holding k-quad (l & 3) ^ ((l >> 4) & 3)
    const int lkq = (lane & 3) ^ ((lane >> 4) & 3), lr = lane >> 2;
    // the token rows: 16 slices x FMX row groups of DMA instructions, eight per wave (group w & 3, slices 8 (w >> 2) .. + 7)
    if ((w & 3) < FMX) {
        const long row = m0 + 16 * (w & 3) + lr;
        const unsigned off = row < a.M ? (unsigned)((row * E + 4 * lkq) * 4) : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 8 * (w >> 2) + j;
            conv_glds16(rx, off + (unsigned)q * 64u, lds_x + (unsigned)(q * SLICE_F4 + 16 * (w & 3) * 4) * 16u);
        }
    }
    const unsigned off1 = (unsigned)(((16 * w + lr) * E + 4 * lkq) * 4);                  // W1: row 16 w + lr of a chunk
    unsigned off2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) off2[j] = (unsigned)(((long)(32 * w + 16 * j + lr) * a.F + 4 * lkq) * 4);   // W2: row 32 w + 16 j + lr
    // stage r (0 .. 23) of chunk ch into ring slot r % 3: r < 16: W1 rows, k = 16 r .. (one instruction); else W2 rows, k = 128 ch + 16 (r - 16) .. (two)
    auto issue = [&](int ch, int r) {
        const unsigned sbase = lds_ring + (unsigned)(r % NS) * (RSTAGE_F4 * 16);
        if (r < K1) {
            conv_glds16(rw1, off1 + (unsigned)r * 64u + (unsigned)ch * (HC * E * 4), sbase);
        } else {
            const unsigned kb = (unsigned)(ch * HC + (r - K1) * 16) * 4u;
            conv_glds16(rw2, off2[0] + kb, sbase);
            conv_glds16(rw2, off2[1] + kb, sbase + 1024);
        }
    };
    const int frag = i * 4 + (kk ^ fsw);         // this lane's float4 of a 16-row fragment; + 64 per further fragment

    f32x4 acc2[FMX][2];
#pragma unroll
    for (int x = 0; x < FMX; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc2[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    issue(0, 1);
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");      // token rows + this wave's stage 0 landed (stage 1 is one instruction)
    __builtin_amdgcn_s_barrier();                         // ... everybody's token rows

    float4 xa[FMX], wa;                                     // the first product's fragments of the step about to run
#pragma unroll
    for (int x = 0; x < FMX; ++x) xa[x] = Xs[x * 64 + frag];
    wa = ring[frag];

    for (int c = 0; c < nchunk; ++c) {
        const bool last = c + 1 == nchunk;
        // this chunk's bias quad, fetched behind the compiler's back (a visible load would make it drain vmcnt - and the ring -
        // before the first use); issued BEFORE the step's stage, so the step's counted wait covers it
        f32x4 bq;
        {
            const float *p = a.b1 + c * HC + 16 * w + 4 * kk;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq) : "v"(p) : "memory");
        }
        f32x4 acc1[FMX];
#pragma unroll
        for (int x = 0; x < FMX; ++x) acc1[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // ---- first product: this wave's 16 hidden units of the chunk for the 64 tokens ----
#pragma unroll
        for (int q = 0; q < K1; ++q) {
            issue(c, q + 2);                                    // (the first steps of the second product at the end)
#pragma unroll
            for (int x = 0; x < FMX; ++x) acc1[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, xa[x].x, acc1[x], 0, 0, 0);
            // stage q + 1 landed when only the stage just issued is in flight: one instruction (first product) or two (second)
            if (q == 0) asm volatile("s_waitcnt vmcnt(1)" : "+v"(bq) :: "memory");      // (the bias quad passes through: it stays where the load puts it)
            else if (q + 2 < K1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            float4 xn[FMX], wn;
            if (q + 1 < K1) {
#pragma unroll
                for (int x = 0; x < FMX; ++x) xn[x] = Xs[(q + 1) * SLICE_F4 + x * 64 + frag];
                wn = ring[((q + 1) % NS) * RSTAGE_F4 + frag];
            }
            __builtin_amdgcn_sched_barrier(0);                  // (left alone, the scheduler sinks these reads below the step's last MFMA to save registers)
#pragma unroll
            for (int x = 0; x < FMX; ++x) acc1[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, xa[x].y, acc1[x], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x) acc1[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, xa[x].z, acc1[x], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x) acc1[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, xa[x].w, acc1[x], 0, 0, 0);
            if (q + 1 < K1) {
#pragma unroll
                for (int x = 0; x < FMX; ++x) xa[x] = xn[x];
                wa = wn;
            }
        }
        // every wave has left the previous chunk's second product -> H_c may be overwritten; bias + ReLU; the D fragment (lane = token i,
        // hidden units 4 kk .. + 3) is one k-quad of the second product's operand: slice w of H_c
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int x = 0; x < FMX; ++x) {
            float4 v = make_float4(acc1[x][0] + bq[0], acc1[x][1] + bq[1], acc1[x][2] + bq[2], acc1[x][3] + bq[3]);
            v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            Hc[w * SLICE_F4 + x * 64 + frag] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- second product: this wave's 32 output channels for the 64 tokens ----
        float4 ha[FMX], wb[2];
#pragma unroll
        for (int x = 0; x < FMX; ++x) ha[x] = Hc[x * 64 + frag];
#pragma unroll
        for (int y = 0; y < 2; ++y) wb[y] = ring[(K1 % NS) * RSTAGE_F4 + y * 64 + frag];
#pragma unroll
        for (int q = 0; q < K2; ++q) {
            const int r = K1 + q;
            const bool more = !(last && q + 2 >= K2);
            if (more) { if (q + 2 < K2) issue(c, r + 2); else issue(c + 1, r + 2 - (K1 + K2)); }
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc2[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[y].x, ha[x].x, acc2[x][y], 0, 0, 0);
            if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (q + 2 < K2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            float4 hn[FMX], wn[2];
            if (q + 1 < K2) {
#pragma unroll
                for (int x = 0; x < FMX; ++x) hn[x] = Hc[(q + 1) * SLICE_F4 + x * 64 + frag];
#pragma unroll
                for (int y = 0; y < 2; ++y) wn[y] = ring[((r + 1) % NS) * RSTAGE_F4 + y * 64 + frag];
            } else if (!last) {                                 // the next chunk's first step
#pragma unroll
                for (int x = 0; x < FMX; ++x) xa[x] = Xs[x * 64 + frag];
                wa = ring[frag];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc2[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[y].y, ha[x].y, acc2[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc2[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[y].z, ha[x].z, acc2[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc2[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[y].w, ha[x].w, acc2[x][y], 0, 0, 0);
            if (q + 1 < K2) {
#pragma unroll
                for (int x = 0; x < FMX; ++x) ha[x] = hn[x];
#pragma unroll
                for (int y = 0; y < 2; ++y) wb[y] = wn[y];
            }
        }
    }
#ifdef FFN_TRACE
    if (a.clk && tid == 0) {
        unsigned long long tr_c1, tr_r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_c1), "=s"(tr_r1) :: "memory");
        atomicAdd(a.clk + 0, tr_c1 - tr_c0); atomicAdd(a.clk + 1, tr_r1 - tr_r0); atomicAdd(a.clk + 2, 1ull);
    }
#endif
    // epilogue: lane = token i, output channels 4 kk .. + 3 of fragment y
#pragma unroll
    for (int x = 0; x < FMX; ++x) {
        const long row = m0 + x * 16 + i;
        if (row >= a.M) continue;
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int co = 32 * w + 16 * y + 4 * kk;
            const float4 b = *(const float4 *)(a.b2 + co);
            *(float4 *)(a.Y + row * E + co) = make_float4(acc2[x][y][0] + b.x, acc2[x][y][1] + b.y, acc2[x][y][2] + b.z, acc2[x][y][3] + b.w);
        }
    }
}

__global__ void __launch_bounds__(512, 1) ffn_fused_w8(const FfnArgs a)
{
    __shared__ __attribute__((aligned(1024))) float4 smem[FFN_W8_LDS_F4];
    const int b = (int)blockIdx.x;
    if (b < a.n_full) { ffn_tile_w8<4>(a, smem, (long)a.m_begin + (long)b * 64); return; }
    const long m0 = (long)a.m_begin + (long)a.n_full * 64 + (long)(b - a.n_full) * 16 * a.tail_frags;
    if (a.tail_frags == 3) ffn_tile_w8<3>(a, smem, m0);
    else if (a.tail_frags == 2) ffn_tile_w8<2>(a, smem, m0);
    else if (a.tail_frags == 1) ffn_tile_w8<1>(a, smem, m0);
    else ffn_tile_w8<4>(a, smem, m0);
}

// the tile plan of a launch over M token rows on `cus` CUs: full rounds of 64-token tiles, the rest cut evenly
static inline void ffn_w8_plan(long M, int cus, FfnArgs *a, unsigned *grid)
{
    const long per_round = 64L * cus;
    const long full_rounds = M / per_round;
    a->n_full = (int)(full_rounds * cus);
    const long rem = M - full_rounds * per_round;
    if (rem == 0) { a->tail_frags = 4; *grid = (unsigned)a->n_full; return; }
    int frags = (int)(((rem + cus - 1) / cus + 15) / 16);
    if (frags > 4) frags = 4;
    a->tail_frags = frags;
    *grid = (unsigned)(a->n_full + (rem + 16L * frags - 1) / (16L * frags));
}

// ------------------------------------------------------------------------------------------------
// The same tile for ONE product with K = 256:  Y[m][n] = act(X[m][:] . W[n][:] + b[n]),  N a multiple of 128 (the encoder's input
// projection, N = 768, and the hoisted layer-0 input product of the stacked LSTM, N = 2 048; learned_models.py:166-172, 184-192).
// The workgroup's 64 token rows are read from memory ONCE (conv2d_nhwc_glds re-stages them for each of its N / 64 column tiles),
// wave w owns output columns 128 c + 16 w .. + 15 of every 128-column chunk c and streams their weight rows through a private
// four-stage ring; no barrier after the token rows have landed.  Same fragments and K order as conv2d_nhwc_glds: the same bits.
// The chunk's 64 x 16 results per wave are stored as they finish (16 tokens x 64 B per instruction); the counted waits of the
// following steps cover those stores too (a store can only make a wait longer, never let a stage be read early).
// ------------------------------------------------------------------------------------------------
struct Gemm256Args {
    const float *X;    // [M][256]
    const float *W;    // [N][256]
    const float *b;    // [N] or null
    float *Y;          // [M][N]
    int M, N, relu;
    int n_full, tail_frags;      // the tile plan (ffn_w8_plan)
};

constexpr int G256_LDS_F4 = 16 * 256 + 8 * 4 * 64;      // token rows + eight private rings of four 1-KB stages: 96 KB

template <int FMX>
__device__ __forceinline__ void gemm256_tile_w8(const Gemm256Args &a, float4 *smem, const long m0)
{
    constexpr int E = 256, NC = 128, K1 = E / 16, NS = 4, D = NS - 1;
    constexpr int SLICE_F4 = 64 * 4, RSTAGE_F4 = 16 * 4;
    static_assert(K1 % NS == 0, "a chunk's first stage must land in ring slot 0");
    static_assert(K1 * SLICE_F4 + 8 * NS * RSTAGE_F4 == G256_LDS_F4, "LDS layout");
    float4 *const Xs = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int fsw = (i >> 2) & 3;
    const int nchunk = a.N / NC;
    float4 *const ring = smem + K1 * SLICE_F4 + w * (NS * RSTAGE_F4);

    conv_u32x4 rx, rw;
    {
        const unsigned long long bx = (unsigned long long)a.X, bw = (unsigned long long)a.W;
        rx.x = (unsigned)bx; rx.y = (unsigned)(bx >> 32); rx.z = (unsigned)((long)a.M * E * 4); rx.w = 0x00020000u;
        rw.x = (unsigned)bw; rw.y = (unsigned)(bw >> 32); rw.z = (unsigned)((long)a.N * E * 4); rw.w = 0x00020000u;
    }
    const unsigned lds_x = (unsigned)(unsigned long long)(const void *)Xs;
    const unsigned lds_ring = (unsigned)(unsigned long long)(const void *)ring;
    const int lkq = (lane & 3) ^ ((lane >> 4) & 3), lr = lane >> 2;
    if ((w & 3) < FMX) {
        const long row = m0 + 16 * (w & 3) + lr;
        const unsigned off = row < a.M ? (unsigned)((row * E + 4 * lkq) * 4) : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 8 * (w >> 2) + j;
            conv_glds16(rx, off + (unsigned)q * 64u, lds_x + (unsigned)(q * SLICE_F4 + 16 * (w & 3) * 4) * 16u);
        }
    }
    const unsigned offw = (unsigned)(((16 * w + lr) * E + 4 * lkq) * 4);
    auto issue = [&](int ch, int r) {      // k-step r of chunk ch into ring slot r % NS (one instruction)
        conv_glds16(rw, offw + (unsigned)r * 64u + (unsigned)ch * (NC * E * 4), lds_ring + (unsigned)(r % NS) * (RSTAGE_F4 * 16));
    };
    const int frag = i * 4 + (kk ^ fsw);

#pragma unroll
    for (int r = 0; r < D; ++r) issue(0, r);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D - 1) : "memory");      // token rows + stage 0
    __builtin_amdgcn_s_barrier();
    float4 xa[FMX], wa;
#pragma unroll
    for (int x = 0; x < FMX; ++x) xa[x] = Xs[x * 64 + frag];
    wa = ring[frag];

    for (int c = 0; c < nchunk; ++c) {
        const bool last = c + 1 == nchunk;
        f32x4 bq = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (a.b) {
            const float *p = a.b + c * NC + 16 * w + 4 * kk;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq) : "v"(p) : "memory");
        }
        f32x4 acc[FMX];
#pragma unroll
        for (int x = 0; x < FMX; ++x) acc[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < K1; ++q) {
            const bool more = !(last && q + D >= K1);
            if (more) { if (q + D < K1) issue(c, q + D); else issue(c + 1, q + D - K1); }
#pragma unroll
            for (int x = 0; x < FMX; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, xa[x].x, acc[x], 0, 0, 0);
            // stage q + 1 landed when only the D - 1 stages behind it are in flight (fewer at the very end)
            if (q == 0) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(bq) : "n"(D - 1) : "memory");
            else if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float4 xn[FMX], wn;
            if (q + 1 < K1) {
#pragma unroll
                for (int x = 0; x < FMX; ++x) xn[x] = Xs[(q + 1) * SLICE_F4 + x * 64 + frag];
                wn = ring[((q + 1) % NS) * RSTAGE_F4 + frag];
            } else if (!last) {
#pragma unroll
                for (int x = 0; x < FMX; ++x) xn[x] = Xs[x * 64 + frag];
                wn = ring[frag];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < FMX; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, xa[x].y, acc[x], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, xa[x].z, acc[x], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < FMX; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, xa[x].w, acc[x], 0, 0, 0);
            if (q + 1 < K1 || !last) {
#pragma unroll
                for (int x = 0; x < FMX; ++x) xa[x] = xn[x];
                wa = wn;
            }
        }
        // lane = token i of fragment x, output columns 128 c + 16 w + 4 kk .. + 3
#pragma unroll
        for (int x = 0; x < FMX; ++x) {
            const long row = m0 + x * 16 + i;
            float4 v = make_float4(acc[x][0] + bq[0], acc[x][1] + bq[1], acc[x][2] + bq[2], acc[x][3] + bq[3]);
            if (a.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            if (row < a.M) *(float4 *)(a.Y + row * a.N + c * NC + 16 * w + 4 * kk) = v;
        }
    }
}

__global__ void __launch_bounds__(512, 1) gemm_k256_w8(const Gemm256Args a)
{
    __shared__ __attribute__((aligned(1024))) float4 smem[G256_LDS_F4];
    const int b = (int)blockIdx.x;
    if (b < a.n_full) { gemm256_tile_w8<4>(a, smem, (long)b * 64); return; }
    const long m0 = (long)a.n_full * 64 + (long)(b - a.n_full) * 16 * a.tail_frags;
    if (a.tail_frags == 3) gemm256_tile_w8<3>(a, smem, m0);
    else if (a.tail_frags == 2) gemm256_tile_w8<2>(a, smem, m0);
    else if (a.tail_frags == 1) gemm256_tile_w8<1>(a, smem, m0);
    else gemm256_tile_w8<4>(a, smem, m0);
}
