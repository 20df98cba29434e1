// attn_train_kernels.hip - the self-attention of the transformer_lstm encoder in TRAINING, forward and backward, without an
// S x S matrix in memory (reference learned_models.py:166-168, 184: nn.TransformerEncoderLayer -> nn.MultiheadAttention under
// training_main.py:183-217).  fp32 MFMA (v_mfma_f32_16x16x4_f32) for every product.
//
// Round 2-5 evaluated the training attention at GEMM granularity: a [Qc x S] score chunk was written by a GEMM, read and
// rewritten by a softmax kernel, read again by a split-K product against a transposed V, and the backward did the same three
// more times with two transposes of the chunk - ~24 GB of score traffic per 32-clip step against 0.3 GB of activations.  Here
// the scores never leave registers:
//
//   forward   attention_train_fwd   O = dropout(softmax(scale Q K^T)) V  per head, online softmax; keeps (row max, 1 / row sum)
//   backward  attention_bwd_prep    D_q = <dO_q, O_q> per head (= sum_k dP_qk P_qk, dropout included), packed with the stats
//             attention_bwd<.., false>  query-stationary:  dQ  = scale (dS K)
//             attention_bwd<.., true>   key-stationary:    dK  = scale (dS^T Q),  dV = Pd^T dO
//             with  P = exp(scale q.k - m) / l,  Pd = P * M (M = dropout multiplier 0 | 1/(1-p)),  dP = (dO V^T) * M,
//                   dS = P * (dP - D)            (the score gradient before the 1/sqrt(hd) factor)
//   Both backward kernels recompute the 16 x 16 score tile and the dP tile on the matrix pipe (7 products of S^2 hd MACs per head
//   instead of the 5 of a stored-matrix backward): no atomics, no cross-workgroup sums on the data path, bit-reproducible.
//
// One kernel body serves both backward passes.  "Stationary" rows a (dQ pass: queries; dK/dV pass: keys) live in registers as the
// B operands of the two score-shaped products; "streaming" rows b (keys; queries) arrive as 16-row tiles through LDS by DMA
// (buffer_load ... lds, 3 stages, counted vmcnt, one raw barrier per tile - the pipeline of attention_glds):
//   sc[b][a] = Y0 X0^T      (Y0 tile rows as A, X0 fragments as B)        dQ pass: K Q^T      dK/dV pass: Q K^T
//   dp[b][a] = Y1 X1^T                                                    V dO^T              dO V^T
//   a lane then holds 4 streaming rows (4 (l >> 4) + r) of ONE stationary row (l & 15): the elementwise part is lane-local (the
//   softmax statistics are known: no shuffles), and the registers are directly the B operands of
//   acc0^T[d][a] += Y0^T dS        (Y0 read the other way: 16 consecutive 16-byte pieces of one row)    dQ^T += K^T dS^T | dK^T += Q^T dS
//   acc1^T[d][a] += Y1^T Pd        (dK/dV pass only)                                                                     | dV^T += dO^T Pd
// Both tiles are stored with the 16-byte piece index XOR-swizzled on the source side (att_kswz), so the "16 rows x one k-quad"
// fragment read is conflict-free; the row-wise read un-swizzles by the row it reads.
// Dropout multipliers are keyed by the element's GLOBAL index head * S * S + q * S + k exactly as enc_softmax_rows did, so the
// recorded-mask goldens of the reference (tests/golden/transformer_dropout_train.npz) and the generator path both still apply.
// A launch whose workgroup count is a small non-multiple of the CU count splits the streaming sweep over gridDim.z; partial sums
// go to scratch and attention_bwd_reduce adds them in slice order.
#pragma once
#include "attn_kernels.hip"
#include "enc_train_kernels.hip"

struct AttTrainFwdArgs {
    const float *qkv;          // [S][3E]
    float *att;                // [S][E]
    float2 *stats;             // [nhead][S]  (row max of the scaled scores, 1 / sum of exp)
    float *opart;              // key split: [KS][S][E] unnormalised partial outputs
    float2 *ml;                // key split: [KS][S][nhead] (running max, running sum)
    int S, E;
    float scale;
    EncSite ds;                // dropout site 0 (attention weights)
    unsigned thresh;
    float inv_keep;
};

// QF = 1 form of attention_glds with the two things training needs: the dropout multiplier on the probabilities that multiply V
// (the row sum stays the sum of the UNdropped exponentials) and the softmax statistics written out.
// DROP = 0: no dropout; 1: counter generator; 2: test-only mask table (as in attention_bwd below)
template <int HD, int DROP>
__global__ void __launch_bounds__(256, 2) attention_train_fwd(const AttTrainFwdArgs a)
{
    constexpr int NS = 3, PPR = HD / 4, NHEX = HD / 16, NC = (HD + 63) / 64, TI = HD / 16, LPS = (2 * TI + 3) / 4, TILE_F4 = 16 * PPR;
    __shared__ __attribute__((aligned(1024))) float4 smem[NS * 2 * TILE_F4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int head = blockIdx.y, S = a.S, E = a.E;
    const int q0 = ((int)blockIdx.x * 4 + w) * 16;
    const long ld = 3L * E;
    conv_u32x4 rs;
    {
        const unsigned long long b = (unsigned long long)a.qkv;
        rs.x = (unsigned)b; rs.y = (unsigned)(b >> 32);
        rs.z = (unsigned)((long)S * ld * 4); rs.w = 0x00020000u;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(const void *)smem;
    int d_row[LPS], d_col[LPS];
    unsigned d_lds[LPS];
    bool d_on[LPS], d_isv[LPS];
#pragma unroll
    for (int j = 0; j < LPS; ++j) {
        const int n = w + 4 * j;
        d_on[j] = n < 2 * TI;
        d_isv[j] = n >= TI;
        const int m = d_isv[j] ? n - TI : n;
        const int g = 64 * m + lane;
        d_row[j] = g / PPR;
        const int pos = g % PPR;
        d_col[j] = (d_isv[j] ? pos : (pos ^ att_kswz<HD>(d_row[j]))) * 16;
        d_lds[j] = (unsigned)((d_isv[j] ? TILE_F4 : 0) + 64 * m) * 16;
    }
    const unsigned kbase = (unsigned)((E + head * HD) * 4), vbase = (unsigned)((2 * E + head * HD) * 4);
    auto issue = [&](int blk) {
        const unsigned sbase = lds0 + (unsigned)(blk % NS) * (2 * TILE_F4 * 16);
#pragma unroll
        for (int j = 0; j < LPS; ++j)
            if (d_on[j]) {
                const int key = blk * 16 + d_row[j];
                const unsigned off = key < S ? (unsigned)((long)key * ld * 4) + (d_isv[j] ? vbase : kbase) + d_col[j] : 0x80000000u;
                conv_glds16(rs, off, sbase + d_lds[j]);
            }
    };
    const int qi = min(q0 + i, S - 1);
    float4 qf[NHEX];
    {
        const float4 *qp = (const float4 *)(a.qkv + (long)qi * ld + head * HD) + kk;
#pragma unroll
        for (int c = 0; c < NHEX; ++c) {
            const float4 v = qp[c * 4];
            qf[c] = make_float4(v.x * a.scale, v.y * a.scale, v.z * a.scale, v.w * a.scale);
        }
    }
    f32x4 o[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[c][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const unsigned long long idx_q = ((unsigned long long)head * S + (unsigned long long)qi) * (unsigned long long)S;
    const unsigned dkey = DROP == 1 ? enc_key(a.ds.seed, a.ds.site) : 0u;

    const int nall = (S + 15) >> 4;
    const int per = (nall + (int)gridDim.z - 1) / (int)gridDim.z;
    const int blk0 = (int)blockIdx.z * per;
    const int nblk = min(nall, blk0 + per);
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (blk0 + s0 < nblk) issue(blk0 + s0);
    if (nblk - blk0 >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int ksw = att_kswz<HD>(i);
    for (int blk = blk0; blk < nblk; ++blk) {
        const bool steady = blk + NS - 1 < nblk;
        if (steady) issue(blk + NS - 1);
        const float4 *Kt = smem + (blk % NS) * (2 * TILE_F4);
        const float4 *Vt = Kt + TILE_F4;
        const int k0 = blk * 16;
        f32x4 sc[2];
        sc[0] = sc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NHEX; ++c) {
            const float4 kf = Kt[i * PPR + ((4 * c + kk) ^ ksw)];
            sc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[c].x, sc[0], 0, 0, 0);
            sc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[c].y, sc[1], 0, 0, 0);
            sc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[c].z, sc[0], 0, 0, 0);
            sc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[c].w, sc[1], 0, 0, 0);
        }
        float4 vf[4][NC];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                vf[r][c] = (16 * c + i < PPR) ? Vt[(4 * kk + r) * PPR + 16 * c + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float s4[4], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s4[r] = (k0 + 4 * kk + r >= S) ? -INFINITY : sc[0][r] + sc[1][r];
            mx = fmaxf(mx, s4[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float p[4], ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[r] = __expf(s4[r] - m_new);
            ps += p[r];
        }
        ps += __shfl_xor(ps, 16);
        ps += __shfl_xor(ps, 32);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if constexpr (DROP != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + 4 * kk + r;
                const unsigned long long idx = idx_q + (unsigned long long)key;
                if constexpr (DROP == 1) p[r] *= enc_hash_keyed(dkey, idx) >= a.thresh ? a.inv_keep : 0.0f;
                else p[r] *= (key < S && a.ds.mask[idx]) ? a.inv_keep : 0.0f;      // (keys past S: p = 0 already)
            }
        }
        const bool rescale = __any(alpha != 1.0f);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (rescale) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[c][e][0] *= alpha; o[c][e][1] *= alpha; o[c][e][2] *= alpha; o[c][e][3] *= alpha;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].x, p[r], o[c][0], 0, 0, 0);
                o[c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].y, p[r], o[c][1], 0, 0, 0);
                o[c][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].z, p[r], o[c][2], 0, 0, 0);
                o[c][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].w, p[r], o[c][3], 0, 0, 0);
            }
        }
        if (steady) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const int q = q0 + i;
    if (q >= S) return;
    const bool partial = gridDim.z > 1;
    const float inv = partial ? 1.0f : 1.0f / l_run;
    float *op = (partial ? a.opart + ((long)blockIdx.z * S + q) * E : a.att + (long)q * E) + (long)head * HD;
    if (kk == 0) {
        if (partial) a.ml[((long)blockIdx.z * S + q) * gridDim.y + head] = make_float2(m_run, l_run);
        else a.stats[(long)head * S + q] = make_float2(m_run, inv);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 64 * c + 4 * (4 * kk + r);
            if (d < HD) *(float4 *)(op + d) = make_float4(o[c][0][r] * inv, o[c][1][r] * inv, o[c][2][r] * inv, o[c][3][r] * inv);
        }
}

// merge of a key-split training forward: attention_merge's arithmetic + the combined statistics (max_z m_z, 1 / sum_z w_z l_z)
__global__ void __launch_bounds__(256) attention_train_merge(const float *__restrict__ opart, const float2 *__restrict__ ml,
                                                             float *__restrict__ out, float2 *__restrict__ stats, int S, int E,
                                                             int nhead, int KS)
{
    const int hd = E / nhead;
    const long n = (long)S * (E >> 2);
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int e4 = idx % (E >> 2);
        const long q = idx / (E >> 2);
        const int head = (e4 * 4) / hd;
        float m = -INFINITY;
        for (int z = 0; z < KS; ++z) m = fmaxf(m, ml[((long)z * S + q) * nhead + head].x);
        float l = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = 0; z < KS; ++z) {
            const float2 s = ml[((long)z * S + q) * nhead + head];
            if (s.y == 0.f) continue;
            const float wz = __expf(s.x - m);
            const float4 v = *(const float4 *)(opart + ((long)z * S + q) * E + e4 * 4);
            l += wz * s.y;
            acc.x += wz * v.x; acc.y += wz * v.y; acc.z += wz * v.z; acc.w += wz * v.w;
        }
        const float inv = 1.0f / l;
        *(float4 *)(out + q * E + e4 * 4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        if ((e4 * 4) % hd == 0) stats[(long)head * S + q] = make_float2(m, inv);
    }
}

// st4[head][q] = (row max, 1 / row sum, D_q = <dO_q, O_q> over the head's columns, 0): one wave per (q, head)
__global__ void __launch_bounds__(256) attention_bwd_prep(const float *__restrict__ dO, const float *__restrict__ O,
                                                          const float2 *__restrict__ stats, float4 *__restrict__ st4, int S, int E,
                                                          int nhead)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);       // head * S + q
    if (row >= (long)S * nhead) return;
    const int head = (int)(row / S), hd = E / nhead;
    const long q = row - (long)head * S;
    const float *a = dO + q * E + head * hd, *b = O + q * E + head * hd;
    float s = 0.f;
    for (int d = lane; d < hd; d += 64) s = fmaf(a[d], b[d], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const float2 st = stats[row];
        st4[row] = make_float4(st.x, st.y, s, 0.f);
    }
}

struct AttBwdArgs {
    const float *x0, *x1;      // stationary operands (rows a), already at their column offset within the matrix; head h adds h * HD
    const float *y0, *y1;      // streaming operands (rows b), likewise
    int ldx0, ldx1, ldy0, ldy1;   // row strides in floats
    const float4 *st4;         // [nhead][S] (m, 1 / l, D, 0) of the QUERY rows
    float *out0, *out1;        // direct outputs (gridDim.z == 1), at their column offset; row stride ldo
    int ldo;
    float *part;               // split sweep: [z][NOUT][S][E]
    int S, E;
    float scale;
    EncSite ds;
    unsigned thresh;
    float inv_keep;
};

// DKV = false: rows a = queries, rows b = keys, acc0 = dQ.  DKV = true: rows a = keys, rows b = queries, acc0 = dK, acc1 = dV.
// AF = 16-row stationary fragments per wave (the tile reads and the barrier are shared by them).
// DROP = 0: no dropout; 1: multipliers from the counter generator; 2: from the test-only mask table (a.ds.mask) - a template
// parameter so that the per-score code carries no branch on it (the first version: 13 scalar branches per tile).
// Two waves per SIMD (256 registers): the LDS fragments travel through two-deep register rings instead of all at once
// (the first version of the dK / dV pass needed 278 registers: one wave per SIMD, 0.53 of the fp32 MFMA peak against 0.61 for dQ).
template <int HD, bool DKV, int AF, int DROP>
__global__ void __launch_bounds__(256, 2) attention_bwd(const AttBwdArgs a)
{
    constexpr int NS = 3, PPR = HD / 4, NHEX = HD / 16, NC = (HD + 63) / 64, TI = HD / 16, LPS = (2 * TI + 3) / 4, TILE_F4 = 16 * PPR;
    __shared__ __attribute__((aligned(1024))) float4 smem[NS * 2 * TILE_F4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int head = blockIdx.y, S = a.S;
    const int a0 = ((int)blockIdx.x * 4 + w) * (16 * AF);
    conv_u32x4 r0, r1;
    {
        const unsigned long long b0 = (unsigned long long)a.y0, b1 = (unsigned long long)a.y1;
        r0.x = (unsigned)b0; r0.y = (unsigned)(b0 >> 32); r0.z = (unsigned)((long)S * a.ldy0 * 4); r0.w = 0x00020000u;
        r1.x = (unsigned)b1; r1.y = (unsigned)(b1 >> 32); r1.z = (unsigned)((long)S * a.ldy1 * 4); r1.w = 0x00020000u;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(const void *)smem;
    // DMA role: instruction n (0..2 TI) of a stage: n < TI -> tile 0 (Y0), else tile 1 (Y1); pieces 64 m .. 64 m + 63 of the tile
    int d_row[LPS];
    unsigned d_col[LPS], d_lds[LPS];
    bool d_on[LPS], d_t1[LPS];
#pragma unroll
    for (int j = 0; j < LPS; ++j) {
        const int n = w + 4 * j;
        d_on[j] = n < 2 * TI;
        d_t1[j] = n >= TI;
        const int m = d_t1[j] ? n - TI : n;
        const int g = 64 * m + lane;
        d_row[j] = g / PPR;
        const int pos = g % PPR;
        d_col[j] = (unsigned)(((pos ^ att_kswz<HD>(d_row[j])) * 4 + head * HD) * 4);
        d_lds[j] = (unsigned)((d_t1[j] ? TILE_F4 : 0) + 64 * m) * 16;
    }
    auto issue = [&](int blk) {
        const unsigned sbase = lds0 + (unsigned)(blk % NS) * (2 * TILE_F4 * 16);
#pragma unroll
        for (int j = 0; j < LPS; ++j)
            if (d_on[j]) {
                const int row = blk * 16 + d_row[j];
                const unsigned off = row < S ? (unsigned)((long)row * (d_t1[j] ? a.ldy1 : a.ldy0) * 4) + d_col[j] : 0x80000000u;
                conv_glds16(d_t1[j] ? r1 : r0, off, sbase + d_lds[j]);
            }
    };
    // stationary fragments: lane (row i, k-quad kk) holds X[a][16 c + 4 kk .. + 3]; X0 carries the 1 / sqrt(hd) factor
    float4 x0f[AF][NHEX], x1f[AF][NHEX];
    int arow[AF];
#pragma unroll
    for (int f = 0; f < AF; ++f) {
        arow[f] = min(a0 + 16 * f + i, S - 1);
        const float4 *p0 = (const float4 *)(a.x0 + (long)arow[f] * a.ldx0 + head * HD) + kk;
        const float4 *p1 = (const float4 *)(a.x1 + (long)arow[f] * a.ldx1 + head * HD) + kk;
#pragma unroll
        for (int c = 0; c < NHEX; ++c) {
            const float4 v = p0[c * 4];
            x0f[f][c] = make_float4(v.x * a.scale, v.y * a.scale, v.z * a.scale, v.w * a.scale);
            x1f[f][c] = p1[c * 4];
        }
    }
    const float4 *st4h = a.st4 + (long)head * S;
    float4 sta[AF];                          // dQ pass: the statistics of this lane's query
#pragma unroll
    for (int f = 0; f < AF; ++f) sta[f] = DKV ? make_float4(0.f, 0.f, 0.f, 0.f) : st4h[arow[f]];
    f32x4 acc0[AF][NC][4], acc1[DKV ? AF : 1][NC][4];
#pragma unroll
    for (int f = 0; f < AF; ++f)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0[f][c][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (DKV) acc1[f][c][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
    // dropout: element (q, k) of head h has index h S^2 + q S + k.  Per tile one 64-bit base for this lane's first streaming row,
    // per element two 64-bit additions: idx = base + off[f] + r * rstep
    const unsigned long long SS = (unsigned long long)S, idx_h = (unsigned long long)head * SS * SS;
    const unsigned long long rstep = DKV ? SS : 1ull;
    unsigned long long ioff[AF];
#pragma unroll
    for (int f = 0; f < AF; ++f) ioff[f] = DKV ? (unsigned long long)arow[f] : (unsigned long long)arow[f] * SS;
    const unsigned key = DROP == 1 ? enc_key(a.ds.seed, a.ds.site) : 0u;

    const int nall = (S + 15) >> 4;
    const int per = (nall + (int)gridDim.z - 1) / (int)gridDim.z;
    const int blk0 = (int)blockIdx.z * per;
    const int nblk = min(nall, blk0 + per);
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (blk0 + s0 < nblk) issue(blk0 + s0);
    if (nblk - blk0 >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int ksw = att_kswz<HD>(i);
    for (int blk = blk0; blk < nblk; ++blk) {
        const bool steady = blk + NS - 1 < nblk;
        if (steady) issue(blk + NS - 1);
        const float4 *Y0 = smem + (blk % NS) * (2 * TILE_F4);
        const float4 *Y1 = Y0 + TILE_F4;
        const int b0 = blk * 16;
        float4 stb[4];                       // dK/dV pass: the statistics of this lane's four queries (L2-resident, broadcast over i)
        if constexpr (DKV) {
#pragma unroll
            for (int r = 0; r < 4; ++r) stb[r] = st4h[min(b0 + 4 * kk + r, S - 1)];
        }
        // ---- the two score-shaped products: four independent accumulator chains per stationary fragment; the tile's k-quads
        //      through a two-deep register ring ----
        f32x4 sc[AF][2], dp[AF][2];
#pragma unroll
        for (int f = 0; f < AF; ++f) sc[f][0] = sc[f][1] = dp[f][0] = dp[f][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 kfr[2], vfr[2];
        kfr[0] = Y0[i * PPR + (kk ^ ksw)];
        vfr[0] = Y1[i * PPR + (kk ^ ksw)];
#pragma unroll
        for (int c = 0; c < NHEX; ++c) {
            if (c + 1 < NHEX) {
                kfr[(c + 1) & 1] = Y0[i * PPR + ((4 * (c + 1) + kk) ^ ksw)];
                vfr[(c + 1) & 1] = Y1[i * PPR + ((4 * (c + 1) + kk) ^ ksw)];
            }
            __builtin_amdgcn_sched_barrier(0);
            const float4 kf = kfr[c & 1], vf = vfr[c & 1];
#pragma unroll
            for (int f = 0; f < AF; ++f) {
                sc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, x0f[f][c].x, sc[f][0], 0, 0, 0);
                dp[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, x1f[f][c].x, dp[f][0], 0, 0, 0);
                sc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, x0f[f][c].y, sc[f][1], 0, 0, 0);
                dp[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, x1f[f][c].y, dp[f][1], 0, 0, 0);
                sc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, x0f[f][c].z, sc[f][0], 0, 0, 0);
                dp[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, x1f[f][c].z, dp[f][0], 0, 0, 0);
                sc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, x0f[f][c].w, sc[f][1], 0, 0, 0);
                dp[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, x1f[f][c].w, dp[f][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the tiles read row-wise (A operands of the accumulating products): row 4 kk + r, pieces 16 c + i, un-swizzled; row r + 1
        //      is fetched under row r's products, row 0 under the elementwise part ----
        float4 y0r[2][NC], y1r[DKV ? 2 : 1][NC];
        auto rowread = [&](int r, int slot) {
            const int row = 4 * kk + r, rsw = att_kswz<HD>(row);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const bool on = 16 * c + i < PPR;
                y0r[slot][c] = on ? Y0[row * PPR + ((16 * c + i) ^ rsw)] : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (DKV) y1r[slot][c] = on ? Y1[row * PPR + ((16 * c + i) ^ rsw)] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        rowread(0, 0);
        // ---- elementwise: P, Pd, dS of this lane's 4 streaming rows x AF stationary rows ----
        float dsv[AF][4], pdv[AF][4];
        const unsigned long long ibase = idx_h + (unsigned long long)(b0 + 4 * kk) * rstep;
#pragma unroll
        for (int f = 0; f < AF; ++f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = b0 + 4 * kk + r;
                const float4 st = DKV ? stb[r] : sta[f];
                const float s = sc[f][0][r] + sc[f][1][r];
                const float p = b < S ? __expf(s - st.x) * st.y : 0.f;
                const float dpv = dp[f][0][r] + dp[f][1][r];
                float keep = 1.0f;
                if constexpr (DROP != 0) {
                    const unsigned long long idx = ibase + ioff[f] + (unsigned long long)r * rstep;
                    if constexpr (DROP == 1) keep = enc_hash_keyed(key, idx) >= a.thresh ? a.inv_keep : 0.0f;
                    else keep = (b < S && a.ds.mask[idx]) ? a.inv_keep : 0.0f;
                }
                pdv[f][r] = p * keep;
                dsv[f][r] = p * (dpv * keep - st.z);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r + 1 < 4) rowread(r + 1, (r + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < AF; ++f)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float4 y0 = y0r[r & 1][c];
                    acc0[f][c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(y0.x, dsv[f][r], acc0[f][c][0], 0, 0, 0);
                    acc0[f][c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(y0.y, dsv[f][r], acc0[f][c][1], 0, 0, 0);
                    acc0[f][c][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(y0.z, dsv[f][r], acc0[f][c][2], 0, 0, 0);
                    acc0[f][c][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(y0.w, dsv[f][r], acc0[f][c][3], 0, 0, 0);
                    if constexpr (DKV) {
                        const float4 y1 = y1r[r & 1][c];
                        acc1[f][c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(y1.x, pdv[f][r], acc1[f][c][0], 0, 0, 0);
                        acc1[f][c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(y1.y, pdv[f][r], acc1[f][c][1], 0, 0, 0);
                        acc1[f][c][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(y1.z, pdv[f][r], acc1[f][c][2], 0, 0, 0);
                        acc1[f][c][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(y1.w, pdv[f][r], acc1[f][c][3], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (steady) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // ---- store: lane holds stationary row a0 + 16 f + i, d = 64 c + 4 (4 kk + r) + e ----
    const bool partial = gridDim.z > 1;
    constexpr int NOUT = DKV ? 2 : 1;
#pragma unroll
    for (int f = 0; f < AF; ++f) {
        const int ar = a0 + 16 * f + i;
        if (ar >= S) continue;
        float *o0 = partial ? a.part + (((long)blockIdx.z * NOUT + 0) * S + ar) * a.E + head * HD : a.out0 + (long)ar * a.ldo + head * HD;
        float *o1 = nullptr;
        if (DKV) o1 = partial ? a.part + (((long)blockIdx.z * NOUT + 1) * S + ar) * a.E + head * HD : a.out1 + (long)ar * a.ldo + head * HD;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 64 * c + 4 * (4 * kk + r);
                if (d >= HD) continue;
                *(float4 *)(o0 + d) = make_float4(acc0[f][c][0][r] * a.scale, acc0[f][c][1][r] * a.scale, acc0[f][c][2][r] * a.scale,
                                                  acc0[f][c][3][r] * a.scale);
                if constexpr (DKV) *(float4 *)(o1 + d) = make_float4(acc1[f][c][0][r], acc1[f][c][1][r], acc1[f][c][2][r], acc1[f][c][3][r]);
            }
    }
}

// out_w[row][col] = sum_z part[z][w][row][col] in slice order; out0 / out1 at their column offsets, row stride ldo (one float4 per thread)
__global__ void __launch_bounds__(256) attention_bwd_reduce(const float *__restrict__ part, int ZS, int NOUT, float *__restrict__ out0,
                                                            float *__restrict__ out1, int ldo, long S, int E)
{
    const long n4 = S * (E >> 2), plane = S * E;
    for (long g = blockIdx.x * 256L + threadIdx.x; g < n4 * NOUT; g += (long)gridDim.x * 256) {
        const int wch = (int)(g / n4);
        const long r = g - wch * n4;
        const long row = r / (E >> 2);
        const int col = (int)(r - row * (E >> 2)) * 4;
        float4 v = *(const float4 *)(part + (long)wch * plane + row * E + col);
        for (int z = 1; z < ZS; ++z) {
            const float4 t = *(const float4 *)(part + ((long)z * NOUT + wch) * plane + row * E + col);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        *(float4 *)((wch ? out1 : out0) + row * ldo + col) = v;
    }
}
