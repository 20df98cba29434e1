// enc_train_kernels.hip - the small kernels around the GEMMs of the transformer encoder layer's TRAINING step
// (reference learned_models.py:166-168 nn.TransformerEncoderLayer under training_main.py:183-217).
// Every matrix product of the layer's forward and backward runs on conv2d_nhwc_glds (strided NT GEMM); what is left
// is row softmax / its backward, LayerNorm forward-with-stats / backward, dropout, ReLU masks, column sums and
// transposes (a product that contracts over tokens is fed as two transposed operands).
//
// Dropout (p = 0.1 in the reference's train mode, four sites per layer) uses a counter-based generator keyed by
// (seed, site, element index): masks are regenerated in the backward pass instead of stored.  The reference's own
// masks come from torch's generator; train-mode parity at p = 0.1 is pinned by FEEDING the layer the masks one reference
// training step drew (tests/golden/transformer_dropout_train.npz) through the test-only mask pointer of EncSite below.
#pragma once
#include <hip/hip_runtime.h>

#define ENC_LN_MAX_PER_LANE 8     // E <= 512

// 32 bits per (seed, site, element).  The key (seed, site) goes through a 64-bit splitmix finaliser - wave-uniform, loop-invariant:
// scalar instructions, once per kernel - and the element index through one multiply, the key, and the two multiply / xor-shift
// rounds of the "lowbias32" integer hash: ~12 VALU instructions per element (the first version put the element through the 64-bit
// finaliser too: ~35, and the flash attention kernels draw one multiplier per score - a fifth of their non-MFMA instructions)
__device__ __forceinline__ unsigned enc_key(unsigned long long seed, unsigned site)
{
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)site + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)((z ^ (z >> 31)) >> 32);
}
__device__ __forceinline__ unsigned enc_hash_keyed(unsigned key, unsigned long long idx)
{
    unsigned x = ((unsigned)idx * 0x9E3779B1u) ^ key;
    x += (unsigned)(idx >> 32) * 0x85EBCA77u;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned enc_hash(unsigned long long seed, unsigned site, unsigned long long idx)
{
    return enc_hash_keyed(enc_key(seed, site), idx);
}

// One dropout site of one layer call: the generator's key (seed, site) - or, TEST-ONLY, a mask buffer (one byte per element, nonzero =
// keep) that the host found registered for this call's seed (opseq_encoder_test_masks_set: how the reference's own draws are fed in,
// forward and backward alike, since both go through enc_keep).  In production `mask` is null: one scalar test of a kernel argument.
struct EncSite { unsigned long long seed; const unsigned char *mask; unsigned site; };

// multiplier of element idx at a dropout site: 0 (dropped) or 1/(1-p); thresh = p * 2^32 (0 => always 1)
__device__ __forceinline__ float enc_keep(const EncSite &ds, unsigned long long idx, unsigned thresh, float inv_keep)
{
    if (thresh == 0u) return 1.0f;
    if (ds.mask) return ds.mask[idx] ? inv_keep : 0.0f;
    return enc_hash(ds.seed, ds.site, idx) >= thresh ? inv_keep : 0.0f;
}

// dst[c][r] = src[r * sld + c]  (r < R, c < C); dst rows have length dld >= R, columns R..dld-1 are zeroed
__global__ void __launch_bounds__(256) enc_transpose(const float *__restrict__ src, long sld, float *__restrict__ dst,
                                                     long dld, int R, int C)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        tile[ty + 8 * j][tx] = (r < R && c < C) ? src[(long)r * sld + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, r = r0 + tx;
        if (c < C && r < dld) dst[(long)c * dld + r] = tile[tx][ty + 8 * j];
    }
}

// rows of scores -> softmax(scale * row) in place (Psoft, saved); optionally Pdrop = Psoft * dropout multiplier.
// One workgroup per row of n valid columns (row stride ld, padding columns zeroed).
// stats (nullable): (row max of scale * score, 1 / sum of exp) per row, so that the backward can rebuild the same
// probabilities in one pass (enc_softmax_from_stats)
__global__ void __launch_bounds__(256) enc_softmax_rows(float *__restrict__ P, float *__restrict__ Pdrop, long ld, int n,
                                                        float scale, const EncSite ds,
                                                        unsigned long long idx0, unsigned thresh, float inv_keep,
                                                        float2 *__restrict__ stats = nullptr)
{
    __shared__ float red[256];
    const long row = blockIdx.x;
    float *p = P + row * ld;
    const int tid = threadIdx.x;
    float m = -INFINITY;
    for (int k = tid; k < n; k += 256) m = fmaxf(m, p[k] * scale);
    red[tid] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    m = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int k = tid; k < n; k += 256) sum += __expf(p[k] * scale - m);
    red[tid] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const float inv = 1.0f / red[0];
    if (stats && tid == 0) stats[row] = make_float2(m, inv);
    for (int k = tid; k < ld; k += 256) {
        const float v = k < n ? __expf(p[k] * scale - m) * inv : 0.f;
        p[k] = v;
        if (Pdrop) Pdrop[row * ld + k] = k < n ? v * enc_keep(ds, idx0 + row * n + k, thresh, inv_keep) : 0.f;
    }
}

// the same probabilities from the scores and the forward's saved (max, 1 / sum): the expression of enc_softmax_rows' last loop,
// hence the same bits, without its two row reductions
__global__ void __launch_bounds__(256) enc_softmax_from_stats(float *__restrict__ P, float *__restrict__ Pdrop, long ld, int n,
                                                              float scale, const EncSite ds,
                                                              unsigned long long idx0, unsigned thresh, float inv_keep,
                                                              const float2 *__restrict__ stats)
{
    const long row = blockIdx.x;
    float *p = P + row * ld;
    const float2 st = stats[row];
    for (int k = threadIdx.x; k < ld; k += 256) {
        const float v = k < n ? __expf(p[k] * scale - st.x) * st.y : 0.f;
        p[k] = v;
        if (Pdrop) Pdrop[row * ld + k] = k < n ? v * enc_keep(ds, idx0 + row * n + k, thresh, inv_keep) : 0.f;
    }
}

// dS = scale * Psoft * (dPs - sum_k dPs Psoft), dPs = dP * dropout multiplier; in place on dP.  One workgroup per row.
__global__ void __launch_bounds__(256) enc_softmax_bwd_rows(const float *__restrict__ P, float *__restrict__ dP, long ld,
                                                            int n, float scale, const EncSite ds,
                                                            unsigned long long idx0, unsigned thresh, float inv_keep)
{
    __shared__ float red[256];
    const long row = blockIdx.x;
    const float *p = P + row * ld;
    float *d = dP + row * ld;
    const int tid = threadIdx.x;
    float dot = 0.f;
    for (int k = tid; k < n; k += 256) dot += d[k] * enc_keep(ds, idx0 + row * n + k, thresh, inv_keep) * p[k];
    red[tid] = dot;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    dot = red[0];
    for (int k = tid; k < ld; k += 256)
        d[k] = k < n ? scale * p[k] * (d[k] * enc_keep(ds, idx0 + row * n + k, thresh, inv_keep) - dot) : 0.f;
}

// u = x + y * dropout multiplier ; out = LayerNorm(u) * g + b ; saves u and (mean, rstd).  One wave per row.
__global__ void __launch_bounds__(256) enc_add_drop_ln(const float *__restrict__ x, const float *__restrict__ y,
                                                       const float *__restrict__ g, const float *__restrict__ b,
                                                       float *__restrict__ u_out, float2 *__restrict__ stats,
                                                       float *__restrict__ out, int rows, int E, float eps,
                                                       const EncSite ds, unsigned thresh, float inv_keep)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[ENC_LN_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < ENC_LN_MAX_PER_LANE; ++j) {
        const int e = j * 64 + lane;
        v[j] = 0.f;
        if (e < E) {
            const long idx = (long)row * E + e;
            v[j] = x[idx] + y[idx] * enc_keep(ds, idx, thresh, inv_keep);
            u_out[idx] = v[j];
            s += v[j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mu = s / (float)E;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < ENC_LN_MAX_PER_LANE; ++j) {
        const int e = j * 64 + lane;
        const float d = e < E ? v[j] - mu : 0.f;
        var += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.0f / sqrtf(var / (float)E + eps);
    if (lane == 0) stats[row] = make_float2(mu, rstd);
#pragma unroll
    for (int j = 0; j < ENC_LN_MAX_PER_LANE; ++j) {
        const int e = j * 64 + lane;
        if (e < E) out[(long)row * E + e] = (v[j] - mu) * rstd * g[e] + b[e];
    }
}

// LayerNorm backward for one row per wave:  du = rstd * (gd - mean(gd) - xhat * mean(gd * xhat)),  gd = dout * g.
// du (+= du_add if given) is written; per-workgroup partial sums of dg = dout * xhat and db = dout go to
// part[blockIdx.x][2][E] (reduced by enc_colsum_final in fixed order).
__global__ void __launch_bounds__(256) enc_ln_bwd(const float *__restrict__ dout, const float *__restrict__ u,
                                                  const float2 *__restrict__ stats, const float *__restrict__ g,
                                                  const float *__restrict__ du_add, float *__restrict__ du,
                                                  float *__restrict__ part, int rows, int E, int rows_per_block)
{
    __shared__ float acc[4][2][ENC_LN_MAX_PER_LANE * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float dg[ENC_LN_MAX_PER_LANE], db[ENC_LN_MAX_PER_LANE];
#pragma unroll
    for (int j = 0; j < ENC_LN_MAX_PER_LANE; ++j) dg[j] = db[j] = 0.f;
    const int r_begin = blockIdx.x * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
    for (int row = r_begin + w; row < r_end; row += 4) {
        const float2 st = stats[row];
        float xh[ENC_LN_MAX_PER_LANE], gd[ENC_LN_MAX_PER_LANE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < ENC_LN_MAX_PER_LANE; ++j) {
            const int e = j * 64 + lane;
            xh[j] = gd[j] = 0.f;
            if (e < E) {
                const long idx = (long)row * E + e;
                const float d = dout[idx];
                xh[j] = (u[idx] - st.x) * st.y;
                gd[j] = d * g[e];
                dg[j] += d * xh[j];
                db[j] += d;
                s1 += gd[j];
                s2 += gd[j] * xh[j];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
        }
        s1 /= (float)E;
        s2 /= (float)E;
#pragma unroll
        for (int j = 0; j < ENC_LN_MAX_PER_LANE; ++j) {
            const int e = j * 64 + lane;
            if (e < E) {
                const long idx = (long)row * E + e;
                const float v = st.y * (gd[j] - s1 - xh[j] * s2);
                du[idx] = du_add ? v + du_add[idx] : v;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < ENC_LN_MAX_PER_LANE; ++j) {
        acc[w][0][j * 64 + lane] = dg[j];
        acc[w][1][j * 64 + lane] = db[j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += 256) {
        part[((long)blockIdx.x * 2 + 0) * E + e] = acc[0][0][e] + acc[1][0][e] + acc[2][0][e] + acc[3][0][e];
        part[((long)blockIdx.x * 2 + 1) * E + e] = acc[0][1][e] + acc[1][1][e] + acc[2][1][e] + acc[3][1][e];
    }
}

// out[j][e] = sum_b part[b][j][e]   (j < J slices of width E), fixed order: a workgroup per 64 columns and slice, wave w adds the
// blocks w, w + 4, ..., the four meet in LDS (150 blocks walked by one thread: 41 us a launch, twelve launches a 32-clip step)
__global__ void __launch_bounds__(256) enc_colsum_final(const float *__restrict__ part, float *__restrict__ out0,
                                                        float *__restrict__ out1, int nblocks, int E)
{
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int j = blockIdx.y;
    float s = 0.f;
    if (e < E) {
#pragma unroll 8
        for (int b = w; b < nblocks; b += 4) s += part[((long)b * gridDim.y + j) * E + e];
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < E) (j == 0 ? out0 : out1)[e] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// partial column sums of X [rows][N] (row stride ld): part[blockIdx.y][n]
__global__ void __launch_bounds__(256) enc_colsum_part(const float *__restrict__ X, long ld, float *__restrict__ part,
                                                       int rows, int N, int rows_per_block)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) s += X[(long)r * ld + n];
    part[(long)blockIdx.y * N + n] = s;
}

// the column sums of a SHORT X [rows][N] in one launch: a workgroup per 64 columns, its four waves take rows w, w + 4, ... and
// meet in LDS (fixed order)
__global__ void __launch_bounds__(256) enc_colsum_small(const float *__restrict__ X, long ld, float *__restrict__ out, int rows, int N)
{
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (n < N) {
#pragma unroll 8
        for (int r = w; r < rows; r += 4) s += X[(long)r * ld + n];
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && n < N) out[n] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// out[row][k] = P[row][k] * attention-dropout multiplier (the operand of P V and of dV = P^T dO when p > 0)
__global__ void __launch_bounds__(256) enc_attn_drop(const float *__restrict__ P, float *__restrict__ out, long ld, int n,
                                                     long rows, const EncSite ds,
                                                     unsigned long long idx0, unsigned thresh, float inv_keep)
{
    const long total = rows * ld;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / ld;
        const int k = (int)(i - row * ld);
        out[i] = k < n ? P[i] * enc_keep(ds, idx0 + row * n + k, thresh, inv_keep) : 0.f;
    }
}

// x *= dropout multiplier (forward sites after ReLU; backward of the residual-branch dropouts)
__global__ void __launch_bounds__(256) enc_dropout(float *__restrict__ x, long n, const EncSite ds,
                                                   unsigned thresh, float inv_keep)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        x[i] *= enc_keep(ds, i, thresh, inv_keep);
}

// out = src * dropout multiplier
__global__ void __launch_bounds__(256) enc_dropout_copy(const float *__restrict__ src, float *__restrict__ out, long n,
                                                        const EncSite ds, unsigned thresh,
                                                        float inv_keep)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = src[i] * enc_keep(ds, i, thresh, inv_keep);
}

// d pre-activation of the FFN: dh * [hid > 0] * dropout multiplier (hid is saved AFTER ReLU and dropout, so hid > 0
// means "active and kept")
__global__ void __launch_bounds__(256) enc_relu_drop_bwd(float *__restrict__ dh, const float *__restrict__ hid, long n,
                                                         float inv_keep)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        dh[i] = hid[i] > 0.f ? dh[i] * inv_keep : 0.f;
}

__global__ void __launch_bounds__(256) enc_add_inplace(float *__restrict__ a, const float *__restrict__ b, long n)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) a[i] += b[i];
}


// ------------------------------------------------------------------------------------------------
// Skinny, long-K products of the chunked attention (O_c = P_c V and dQ_c = dS_c K: [Qc x S] . [S x hd] - a handful of
// 128 x 128 tiles whose K walk is the whole sequence): split K over gridDim.z workgroups, partial tiles to scratch, summed
// in fixed slice order by enc_splitk_reduce (deterministic).  Operands K-contiguous, one float4 per lane per 16-k step
// feeding four MFMAs, straight from L2 / Infinity Cache (the chunk was just written there) - no LDS staging.
// Workgroup = 64 x 64 outputs, wave = 32 x 32.  part [z][M][N].  kc = k per slice (a multiple of 16).
__global__ void __launch_bounds__(256) enc_gemm_nt_splitk(const float *__restrict__ X, long XS, const float *__restrict__ W, long WS,
                                                          float *__restrict__ part, int M, int N, int K, int kc)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 64 + (w >> 1) * 32, n0 = blockIdx.y * 64 + (w & 1) * 32;
    const int i = lane & 15, kk = lane >> 4;
    const int k0 = blockIdx.z * kc;
    const int k1 = min(K, k0 + kc);
    const float4 *a_row[2], *w_row[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int m = min(m0 + f * 16 + i, M - 1), nn = min(n0 + f * 16 + i, N - 1);
        a_row[f] = (const float4 *)(X + (long)m * XS + k0) + kk;
        w_row[f] = (const float4 *)(W + (long)nn * WS + k0) + kk;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nhex = (k1 - k0) >> 4;
#pragma unroll 4
    for (int q = 0; q < nhex; ++q) {
        const float4 av0 = a_row[0][q * 4], av1 = a_row[1][q * 4];
        const float4 wv0 = w_row[0][q * 4], wv1 = w_row[1][q * 4];
        const float ae[2][4] = {{av0.x, av0.y, av0.z, av0.w}, {av1.x, av1.y, av1.z, av1.w}};
        const float we[2][4] = {{wv0.x, wv0.y, wv0.z, wv0.w}, {wv1.x, wv1.y, wv1.z, wv1.w}};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[x][e], we[y][e], acc[x][y], 0, 0, 0);
    }
    float *P = part + (long)blockIdx.z * M * N;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int nn = n0 + y * 16 + i;
            if (nn >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + x * 16 + kk * 4 + r;
                if (m < M) P[(long)m * N + nn] = acc[x][y][r];
            }
        }
}

// Y[m * YS + n] = sum_z part[z][m][n] in ascending z
__global__ void __launch_bounds__(256) enc_splitk_reduce(const float *__restrict__ part, int KS, long MN, int N, float *__restrict__ Y, long YS)
{
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < MN; idx += (long)gridDim.x * 256) {
        float s = part[idx];
        for (int z = 1; z < KS; ++z) s += part[(long)z * MN + idx];
        const long m = idx / N;
        Y[m * YS + (idx - m * N)] = s;
    }
}
