// attn_kernels.hip - self-attention of the transformer_lstm encoder (reference learned_models.py:166-168,184:
// nn.TransformerEncoderLayer -> nn.MultiheadAttention) as a flash-style kernel with the K / V tiles
// staged in LDS by DMA.  fp32 MFMA (v_mfma_f32_16x16x4_f32) for both products, online softmax in registers.
//
//   O[s][h*hd + d] = sum_k softmax_k(q_s . k_k / sqrt(hd)) v_k[d]      over ONE sequence of S tokens
//
// Workgroup = 4 waves; a wave owns QF x 16 queries of one head for the whole key sweep (Q fragments and the O^T
// accumulators live in registers), the workgroup shares each 16-key K tile and V tile through LDS:
//   * tiles arrive by `buffer_load_dwordx4 ... lds` (conv_glds16), 3 stages deep, counted vmcnt + one raw barrier per
//     tile - the pipeline of conv2d_nhwc_glds (computing the next tile's scores under this tile's softmax, 4 stages,
//     was tried: no gain - with one wave per SIMD the loop is bound by the barrier + LDS round trip per tile); keys past S are buffer offsets past num_records (zeros);
//   * K image: row-major [key][hd] with the 16-byte piece index XOR-swizzled on the SOURCE side so that the fragment
//     read "16 consecutive keys, one k-quad" is bank-conflict free; V image: plain row-major (its fragment read is 16
//     consecutive pieces of one key);
//   * S^T[key][query] = K_tile Q^T (A = K rows from LDS, B = Q registers), two accumulators per query fragment that take turns
//     with every MFMA, so the dependent-MFMA latency (40 cycles vs a 32-cycle issue) is hidden;
//   * a lane holds 4 keys (rows 4*(l>>4)+r) of ONE query (column l&15): row max / sum = 4 registers + two shuffles;
//   * O^T[d][query] += V_tile^T P^T with the score registers themselves as the B operand (MFMA r contracts keys
//     {r, 4+r, 8+r, 12+r}); A = V read as float4 along d, element e -> accumulator e holding d = 64c + 4i + e.
// HD in {16, 32, 64, 128} (a power of two, so a DMA instruction covers whole rows); other head sizes use attention_f32.
#pragma once
#include "conv_kernels.hip"

template <int HD>
__device__ __forceinline__ int att_kswz(int row)
{
    constexpr int PPR = HD / 4;                               // 16-byte pieces per row
    return PPR >= 16 ? (row & 15) : ((row / (16 / PPR)) & (PPR - 1));
}

// Key split (gridDim.z = KS > 1): workgroup z sweeps key tiles [z*per, (z+1)*per) and leaves an UNNORMALISED partial
// (O, running max m, running sum l) per query and head; attention_merge combines them.  Wave-tasks come in units of
// "16 QF queries x all keys"; when their number is a small non-multiple of the SIMD count (S = 9600, 2 heads: 1200
// tasks on 1024 SIMDs) the busiest SIMD carries twice the average - splitting the keys makes the units 1/KS the size.
//
// Segments (nseg > 1; serving independent requests in one pass): qkv / out hold nseg sequences of S tokens back to back and
// every sequence attends to itself only.  blockIdx.x = segment * (gridDim.x / nseg) + query tile; a workgroup rebases its
// pointers to its segment and then runs EXACTLY the single-sequence code on it (same tiles, same sweep, same arithmetic), so a
// segment's output is bit-identical to that sequence run alone.  The key-split partials are indexed by the global row.
template <int HD, int QF>
__global__ void __launch_bounds__(256) attention_glds(const float *__restrict__ qkv, float *__restrict__ out, int S,
                                                      int E, float scale, float *__restrict__ opart,
                                                      float2 *__restrict__ ml, int nseg)
{
    constexpr int NS = 3;                                     // LDS stages
    constexpr int PPR = HD / 4;                               // pieces per row
    constexpr int NHEX = HD / 16;
    constexpr int NC = (HD + 63) / 64;                        // 64-wide d chunks
    constexpr int TI = HD / 16;                               // DMA instructions per 16-key tile (16 * PPR / 64)
    constexpr int LPS = (2 * TI + 3) / 4;                     // DMA instructions per wave per stage (K and V tiles)
    constexpr int TILE_F4 = 16 * PPR;                         // float4 slots per tile
    __shared__ __attribute__((aligned(1024))) float4 smem[NS * 2 * TILE_F4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int head = blockIdx.y;
    const int xtiles = (int)gridDim.x / nseg;                 // query tiles per segment
    const int seg = (int)blockIdx.x / xtiles;
    const int q0 = (((int)blockIdx.x - seg * xtiles) * 4 + w) * (16 * QF);
    const long ld = 3L * E;
    const long row0 = (long)seg * S;                          // first global row of this workgroup's segment
    const long Stot = (long)nseg * S;
    qkv += row0 * ld;
    out += row0 * E;

    conv_u32x4 rs;
    {
        const unsigned long long b = (unsigned long long)qkv;
        rs.x = (unsigned)b; rs.y = (unsigned)(b >> 32);
        rs.z = (unsigned)((long)S * ld * 4); rs.w = 0x00020000u;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(const void *)smem;

    // DMA role: instruction n (0..2 TI) of a stage: n < TI -> K tile, else V tile; pieces 64 m .. 64 m + 63 of the tile
    int d_row[LPS], d_col[LPS];           // key row within the tile, source byte offset within the row's head slice
    unsigned d_lds[LPS];                  // destination within a stage
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

    // Q fragments (B operand of S^T): lane (query i, kk) holds Q[q][16c + 4kk .. +3] * scale
    float4 qf[QF][NHEX];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int qi = min(q0 + 16 * f + i, S - 1);
        const float4 *qp = (const float4 *)(qkv + (long)qi * ld + head * HD) + kk;
#pragma unroll
        for (int c = 0; c < NHEX; ++c) {
            const float4 v = qp[c * 4];
            qf[f][c] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
        }
    }
    f32x4 o[QF][NC][4];   // [query fragment][d chunk c][element e] -> rows i' = 4*(l>>4)+r  <->  d = 64c + 4i' + e
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[f][c][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) { m_run[f] = -INFINITY; l_run[f] = 0.f; }

    const int nall = (S + 15) >> 4;
    const int per = (nall + (int)gridDim.z - 1) / (int)gridDim.z;
    const int blk0 = (int)blockIdx.z * per;
    const int nblk = min(nall, blk0 + per);          // this workgroup sweeps key tiles [blk0, nblk)
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
        // ---- scores^T for 16 keys ----
        f32x4 sc[QF][2];
#pragma unroll
        for (int f = 0; f < QF; ++f) sc[f][0] = sc[f][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NHEX; ++c) {
            const float4 kf = Kt[i * PPR + ((4 * c + kk) ^ ksw)];
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                // the two chains alternate with every MFMA (even / odd k of the hexadecet): back-to-back MFMAs on ONE accumulator issue
                // 40 cycles apart (dependent latency) instead of 32
                sc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[f][c].x, sc[f][0], 0, 0, 0);
                sc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[f][c].y, sc[f][1], 0, 0, 0);
                sc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[f][c].z, sc[f][0], 0, 0, 0);
                sc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[f][c].w, sc[f][1], 0, 0, 0);
            }
        }
        // V fragments of this tile
        float4 vf[4][NC];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                vf[r][c] = (16 * c + i < PPR) ? Vt[(4 * kk + r) * PPR + 16 * c + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        // ---- online softmax: this lane holds keys k0 + 4*kk + r of query q0 + 16 f + i ----
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float s4[4], mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s4[r] = (k0 + 4 * kk + r >= S) ? -INFINITY : sc[f][0][r] + sc[f][1][r];
                mx = fmaxf(mx, s4[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[f], mx);
            const float alpha = __expf(m_run[f] - m_new);   // first block: exp(-inf) = 0
            float p[4], ps = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = __expf(s4[r] - m_new);
                ps += p[r];
            }
            ps += __shfl_xor(ps, 16);
            ps += __shfl_xor(ps, 32);
            l_run[f] = l_run[f] * alpha + ps;
            m_run[f] = m_new;
            // ---- O^T = O^T * alpha + V^T P^T ----
            // the running max of a query stops moving after the first few hundred keys: when no lane of the wave
            // saw a new maximum (alpha == 1 everywhere) the rescale - 16 NC accumulator registers through the
            // VALU - is skipped; bit-identical either way
            const bool rescale = __any(alpha != 1.0f);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (rescale) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[f][c][e][0] *= alpha; o[f][c][e][1] *= alpha; o[f][c][e][2] *= alpha; o[f][c][e][3] *= alpha;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[f][c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].x, p[r], o[f][c][0], 0, 0, 0);
                    o[f][c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].y, p[r], o[f][c][1], 0, 0, 0);
                    o[f][c][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].z, p[r], o[f][c][2], 0, 0, 0);
                    o[f][c][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].w, p[r], o[f][c][3], 0, 0, 0);
                }
            }
        }
        if (steady) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPS * (NS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // ---- normalise and store: lane holds query q0 + 16 f + i, d = 64c + 4*(4*kk + r) + e ----
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int q = q0 + 16 * f + i;
        if (q >= S) continue;
        const bool partial = gridDim.z > 1;
        const float inv = partial ? 1.0f : 1.0f / l_run[f];
        float *op = (partial ? opart + ((long)blockIdx.z * Stot + row0 + q) * E : out + (long)q * E) + (long)head * HD;
        if (partial && kk == 0) ml[((long)blockIdx.z * Stot + row0 + q) * gridDim.y + head] = make_float2(m_run[f], l_run[f]);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 64 * c + 4 * (4 * kk + r);
                if (d < HD)
                    *(float4 *)(op + d) = make_float4(o[f][c][0][r] * inv, o[f][c][1][r] * inv, o[f][c][2][r] * inv, o[f][c][3][r] * inv);
            }
    }
}

// out[q][head*hd + d] = sum_z w_z O_z / sum_z w_z l_z,  w_z = exp(m_z - max_z m_z)   (one thread per float4 of out)
__global__ void __launch_bounds__(256) attention_merge(const float *__restrict__ opart, const float2 *__restrict__ ml,
                                                       float *__restrict__ out, int S, int E, int nhead, int KS)
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
            if (s.y == 0.f) continue;                       // an empty split (no keys)
            const float wz = __expf(s.x - m);
            const float4 v = *(const float4 *)(opart + ((long)z * S + q) * E + e4 * 4);
            l += wz * s.y;
            acc.x += wz * v.x; acc.y += wz * v.y; acc.z += wz * v.z; acc.w += wz * v.w;
        }
        const float inv = 1.0f / l;
        *(float4 *)(out + q * E + e4 * 4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}
