// opnet_kernels.hip - gfx950 (MI355X / CDNA4) kernels for the OPNet reasoner hot path.
//
// What is computed (reference baselines/learned_models.py:35-52, restated in oracle/opnet_oracle.py):
//   LSTM1 90->H1 (no bias, gates i,f,g,o) -> Linear H1->15 -> softmax(15) -> sum_o p[o]*boxes[o,:]
//   -> LSTM2 6->H2 -> Linear H2->4 ; plus the transposed logits.
//
// How: the two recurrences are 300 dependent steps whose per-step work is a
// [32 clips x K] x [K x 4H] product; an in-launch all-to-all exchange of h between CUs costs MORE
// on this chip (2.4-4.2 us, MI355X_MICROARCH.md "allgather") than a dependent kernel boundary
// (1.45-1.9 us, "boundary"), so each time step is ONE launch of `opnet_step`, and the T+3
// launches are replayed from a hipGraph.  Inside a launch four software-pipelined roles run
// side by side on different workgroups:
//     launch s :  LSTM1 step s | selection head step s-1 | LSTM2 step s-2 | output head step s-3
// All four are the same primitive: D[16 rows x 32 clips] = A[16 x K] * h^T[K x 32] on
// v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain), K split across the workgroup's
// waves, partials reduced through LDS in fixed order, then a role-specific epilogue.
//
// Layouts in HBM (fp32):
//   activations  "kq-major row-block":  [row-block of 32 clips][k/4][clip 0..31][4]   so that one
//                wave-wide float4 load is four contiguous 256-B runs and the 4 hidden units a
//                workgroup produces for 32 clips are one contiguous 512-B run;
//   weights      per 16-row tile in MFMA A-fragment order [tile][k/16][lane 0..63][4]: lane l holds
//                row (l&15), k = 16q + 4(l>>4) + e  -> perfectly coalesced 1-KiB wave loads.
//   A 16-row LSTM tile = 4 hidden units x 4 gates, row i <-> (unit i>>2, gate i&3), so after the
//   MFMA lane (clip, quarter) holds i,f,g,o of one (clip, unit) in its 4 accumulator registers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opnet_ctx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define OPNET_NW 4                 // waves per workgroup (K split)
#define OPNET_THREADS (OPNET_NW * 64)

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
// out[tile][q][lane][e] = W(row(tile, lane&15), k = 16q + 4(lane>>4) + e) over the concatenated K
// space [ x part: KX real columns padded to KXP | h part: KH columns ].
// mode 0 (LSTM): row i of tile -> W row (i&3)*H + tile*4 + (i>>2)   (gate-major torch layout)
// mode 1 (head): row i -> W row i if i < nrows, else zero; single tile.
__global__ void opnet_pack_tiles(float *__restrict__ out, const float *__restrict__ wx,
                                 const float *__restrict__ wh, int KX, int KXP, int KH, int H,
                                 int nrows, int mode, int ntiles)
{
    const int nhex = (KXP + KH) / 16;
    const long total = (long)ntiles * nhex * 64 * 4;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int e = idx & 3;
        const int lane = (idx >> 2) & 63;
        const long tq = idx >> 8;
        const int q = tq % nhex;
        const int tile = tq / nhex;
        const int i = lane & 15;
        const int k = 16 * q + 4 * (lane >> 4) + e;
        int row;
        bool valid = true;
        if (mode == 0) {
            row = (i & 3) * H + tile * 4 + (i >> 2);
        } else {
            row = i;
            valid = i < nrows;
        }
        float v = 0.f;
        if (valid) {
            if (k < KXP) {
                if (k < KX) v = wx[(long)row * KX + k];
            } else {
                v = wh[(long)row * KH + (k - KXP)];
            }
        }
        out[idx] = v;
    }
}

// wih2p[unit][gate][8] = w_ih2[gate*H2 + unit][0..5], 0, 0
__global__ void opnet_pack_wih2(float *__restrict__ out, const float *__restrict__ w_ih2, int H2)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H2 * 4 * 8) return;
    const int f = idx & 7;
    const int gate = (idx >> 3) & 3;
    const int unit = idx >> 5;
    out[idx] = f < OPNET_FEATS_ ? w_ih2[(long)(gate * H2 + unit) * OPNET_FEATS_ + f] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// context + input packing
// ------------------------------------------------------------------------------------------------
__global__ void opnet_set_ctx(OpnetCtx *dst, OpnetCtx src) { *dst = src; }

// boxes [B][T][90] -> xp [t][rb][kq 0..23][clip 0..31][4]  (K padded 90 -> 96 with zeros, clips
// beyond B zero).  One workgroup per (t, rb).
__global__ void __launch_bounds__(256) opnet_pack_input(const OpnetCtx *__restrict__ ctx)
{
    const int t = blockIdx.x;
    const int rb = blockIdx.y;
    const int B = ctx->B, T = ctx->T, RB = ctx->RB;
    const float *__restrict__ boxes = ctx->boxes;
    float4 *__restrict__ xp = ctx->xp + ((long)t * RB + rb) * (OPNET_KXQ * 32);
    for (int idx = threadIdx.x; idx < OPNET_KXQ * 32; idx += 256) {
        // read-coalesced mapping: consecutive threads walk k within a clip row
        const int kq = idx % OPNET_KXQ;
        const int clip = idx / OPNET_KXQ;
        const int b = rb * 32 + clip;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B) {
            // rows are 360 B apart: only 8-byte alignment is guaranteed
            const float2 *src = (const float2 *)(boxes + ((long)b * T + t) * OPNET_KX + kq * 4);
            const int k = kq * 4;
            if (k + 1 < OPNET_KX) { float2 a = src[0]; v.x = a.x; v.y = a.y; }
            if (k + 3 < OPNET_KX) { float2 c = src[1]; v.z = c.x; v.w = c.y; }
        }
        xp[kq * 32 + clip] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// the MFMA core shared by all four roles
// ------------------------------------------------------------------------------------------------
// D[16 x 32] = A[16 x K] * Hsrc^T, K = 16*(nh0+nh1) taken from up to two activation segments (each
// [k/4][32][4] float4 for this row block).  Wave w reduces hexadecets [w*nhex/NW, (w+1)*nhex/NW).
// part layout in LDS: [wave][acc reg 0..7][lane]; acc regs 0..3 = clips 0..15, 4..7 = clips 16..31.
__device__ __forceinline__ void gemm16_core(const float4 *__restrict__ A,
                                            const float4 *__restrict__ seg0, int nh0,
                                            const float4 *__restrict__ seg1, int nh1,
                                            float *__restrict__ part)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nhex = nh0 + nh1;
    const int q0 = (w * nhex) / OPNET_NW;
    const int q1 = ((w + 1) * nhex) / OPNET_NW;
    const int boff = (lane >> 4) * 32 + (lane & 15);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int q = q0; q < q1; ++q) {
        const float4 a = A[q * 64 + lane];
        const float4 *src = (q < nh0) ? (seg0 + q * 128) : (seg1 + (q - nh0) * 128);
        const float4 b0 = src[boff];
        const float4 b1 = src[boff + 16];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc1, 0, 0, 0);
    }
    float *p = part + (w * 8) * 64 + lane;
    p[0 * 64] = acc0[0]; p[1 * 64] = acc0[1]; p[2 * 64] = acc0[2]; p[3 * 64] = acc0[3];
    p[4 * 64] = acc1[0]; p[5 * 64] = acc1[1]; p[6 * 64] = acc1[2]; p[7 * 64] = acc1[3];
}

// fixed-order cross-wave reduction of D element (reg, lane)
__device__ __forceinline__ float part_sum(const float *__restrict__ part, int reg, int lane)
{
    float s = part[(0 * 8 + reg) * 64 + lane];
#pragma unroll
    for (int w = 1; w < OPNET_NW; ++w) s += part[(w * 8 + reg) * 64 + lane];
    return s;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// LSTM cell update for one (clip, unit): gates i,f,g,o -> (c, h)
__device__ __forceinline__ float lstm_cell(float gi, float gf, float gg, float go, float *c_io)
{
    const float i = sigmoidf_(gi);
    const float f = sigmoidf_(gf);
    const float g = tanhf(gg);
    const float o = sigmoidf_(go);
    const float c = f * (*c_io) + i * g;
    *c_io = c;
    return o * tanhf(c);
}

// ------------------------------------------------------------------------------------------------
// the step kernel
// ------------------------------------------------------------------------------------------------
// grid.x = n2 (LSTM2 tiles, longest K first) + n1 (LSTM1 tiles) + 2 heads ; grid.y = row blocks.
__global__ void __launch_bounds__(OPNET_THREADS) opnet_step(const OpnetCtx *__restrict__ ctx, int s)
{
    __shared__ __attribute__((aligned(16))) float lds[OPNET_NW * 8 * 64 + 32 * 16];
    float *part = lds;
    float *lg = lds + OPNET_NW * 8 * 64;  // [clip][16] logits / probabilities (selection head)

    const int bx = blockIdx.x;
    const int rb = blockIdx.y;
    const int T = ctx->T, B = ctx->B;
    const int H1 = ctx->H1, H2 = ctx->H2;
    const int n1 = H1 >> 2, n2 = H2 >> 2;
    const int tid = threadIdx.x;

    // epilogue coordinates (threads 0..127): D column = clip, D rows 4*(lane>>4)+r in regs r
    const int el = tid & 63;
    const int half = tid >> 6;
    const int clip = half * 16 + (el & 15);
    const int quarter = el >> 4;

    if (bx < n2) {
        // ---------------- LSTM2 (video_LSTM, learned_models.py:32,46), step t = s-2 -------------
        const int t = s - 2;
        if (t < 0 || t >= T) return;
        const int tile = bx;
        const int nh = H2 >> 4;
        const float4 *hprev = ctx->h2buf + ((long)((t + 1) & 1) * ctx->RB + rb) * (H2 * 8);
        // epilogue operands are fetched before the MFMA phase so their latency hides under it
        const int unit = tile * 4 + quarter;
        float4 xa, xb, wv[8];
        float c_old = 0.f;
        if (tid < 128) {
            const float4 *x2 = ctx->x2buf + ((long)(t & 1) * ctx->RB + rb) * 64 + clip * 2;
            xa = x2[0];
            xb = x2[1];
            const float4 *wi = ctx->wih2p + (long)unit * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = wi[j];
            c_old = ctx->c2[((long)rb * H2 + unit) * 32 + clip];
        }
        gemm16_core(ctx->w2p + (long)tile * nh * 64, hprev, nh, hprev, 0, part);
        __syncthreads();
        if (tid < 128) {
            float g[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // x part: W_ih2[gate r, unit][0..5] . frames_boxes[clip][0..5]
                float xs = wv[2 * r].x * xa.x;
                xs = fmaf(wv[2 * r].y, xa.y, xs);
                xs = fmaf(wv[2 * r].z, xa.z, xs);
                xs = fmaf(wv[2 * r].w, xa.w, xs);
                xs = fmaf(wv[2 * r + 1].x, xb.x, xs);
                xs = fmaf(wv[2 * r + 1].y, xb.y, xs);
                g[r] = part_sum(part, half * 4 + r, el) + xs;
            }
            float c = c_old;
            const float h = lstm_cell(g[0], g[1], g[2], g[3], &c);
            ctx->c2[((long)rb * H2 + unit) * 32 + clip] = c;
            float *hout = (float *)(ctx->h2buf + ((long)(t & 1) * ctx->RB + rb) * (H2 * 8));
            hout[((long)tile * 32 + clip) * 4 + quarter] = h;
        }
    } else if (bx < n2 + n1) {
        // ---------------- LSTM1 (object_to_track_LSTM, learned_models.py:29,39), step t = s -----
        const int t = s;
        if (t >= T) return;
        const int tile = bx - n2;
        const int nhh = H1 >> 4;
        const float4 *xsrc = ctx->xp + ((long)t * ctx->RB + rb) * (OPNET_KXQ * 32);
        const float4 *hprev = ctx->h1buf + ((long)((t + 1) & 1) * ctx->RB + rb) * (H1 * 8);
        const int unit = tile * 4 + quarter;
        float c_old = 0.f;
        if (tid < 128) c_old = ctx->c1[((long)rb * H1 + unit) * 32 + clip];
        gemm16_core(ctx->w1p + (long)tile * (OPNET_KXQ / 4 + nhh) * 64, xsrc, OPNET_KXQ / 4, hprev,
                    nhh, part);
        __syncthreads();
        if (tid < 128) {
            float c = c_old;
            const float h = lstm_cell(part_sum(part, half * 4 + 0, el), part_sum(part, half * 4 + 1, el),
                                      part_sum(part, half * 4 + 2, el), part_sum(part, half * 4 + 3, el), &c);
            ctx->c1[((long)rb * H1 + unit) * 32 + clip] = c;
            float *hout = (float *)(ctx->h1buf + ((long)(t & 1) * ctx->RB + rb) * (H1 * 8));
            hout[((long)tile * 32 + clip) * 4 + quarter] = h;
        }
    } else if (bx == n2 + n1) {
        // ---------------- selection head, step t = s-1 (learned_models.py:40-43,50) -------------
        const int t = s - 1;
        if (t < 0 || t >= T) return;
        const int nh = H1 >> 4;
        const float4 *hcur = ctx->h1buf + ((long)(t & 1) * ctx->RB + rb) * (H1 * 8);
        gemm16_core(ctx->wselp, hcur, nh, hcur, 0, part);
        __syncthreads();
        if (tid < 128) {
            const int b = rb * 32 + clip;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int slot = quarter * 4 + r;
                const float v = part_sum(part, half * 4 + r, el);
                lg[clip * 16 + slot] = v;
                // logits [B][15][T] (the permute(0,2,1).contiguous() of :50)
                if (slot < OPNET_SLOTS_ && b < B) ctx->logits[((long)b * OPNET_SLOTS_ + slot) * T + t] = v;
            }
        }
        __syncthreads();
        if (tid < 32) {
            // F.softmax(dim=-1) over the 15 slots of clip `tid`
            float *row = lg + tid * 16;
            float m = row[0];
#pragma unroll
            for (int j = 1; j < OPNET_SLOTS_; ++j) m = fmaxf(m, row[j]);
            float e[OPNET_SLOTS_];
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < OPNET_SLOTS_; ++j) { e[j] = expf(row[j] - m); sum += e[j]; }
#pragma unroll
            for (int j = 0; j < OPNET_SLOTS_; ++j) row[j] = e[j] / sum;
        }
        __syncthreads();
        {
            // frames_boxes[clip][f] = sum_o boxes[clip][t][o][f] * p[o]   (einsum "bfot,bfo->bft")
            const int c2 = tid >> 3, f = tid & 7;
            const int b = rb * 32 + c2;
            float acc = 0.f;
            if (f < OPNET_FEATS_ && b < B) {
                const float *bx_ = ctx->boxes + ((long)b * T + t) * OPNET_KX + f;
                const float *p = lg + c2 * 16;
#pragma unroll
                for (int o = 0; o < OPNET_SLOTS_; ++o) acc = fmaf(bx_[o * OPNET_FEATS_], p[o], acc);
            }
            float *x2 = (float *)(ctx->x2buf + ((long)(t & 1) * ctx->RB + rb) * 64);
            x2[c2 * 8 + f] = acc;
        }
    } else {
        // ---------------- output head, step t = s-3 (prediction_layer, learned_models.py:33,47) --
        const int t = s - 3;
        if (t < 0 || t >= T) return;
        const int nh = H2 >> 4;
        const float4 *hcur = ctx->h2buf + ((long)(t & 1) * ctx->RB + rb) * (H2 * 8);
        gemm16_core(ctx->woutp, hcur, nh, hcur, 0, part);
        __syncthreads();
        if (tid < 128 && quarter == 0) {
            const int b = rb * 32 + clip;
            if (b < B) {
                float4 v;
                v.x = part_sum(part, half * 4 + 0, el);
                v.y = part_sum(part, half * 4 + 1, el);
                v.z = part_sum(part, half * 4 + 2, el);
                v.w = part_sum(part, half * 4 + 3, el);
                ((float4 *)ctx->y)[(long)b * T + t] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// post-processing + metric (integer arithmetic, bit-exact contract)
// ------------------------------------------------------------------------------------------------
// inference_main.py:219: (float32 * int64 [320,240,320,240]) is a float64 multiply, astype(int32)
// truncates toward zero.  tracking_utils.py:137-159: inclusive-pixel IoU, float64 division.
__device__ __forceinline__ int to_px(float v, int k)
{
    const double scale = (k & 1) ? 240.0 : 320.0;
    return (int)((double)v * scale);
}

__global__ void opnet_postprocess_iou_kernel(const float4 *__restrict__ y, const float4 *__restrict__ lab,
                                             int4 *__restrict__ pred_px, int4 *__restrict__ gt_px,
                                             double *__restrict__ iou, long n)
{
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float4 a = y[idx];
    int4 p = make_int4(to_px(a.x, 0), to_px(a.y, 1), to_px(a.z, 2), to_px(a.w, 3));
    if (pred_px) pred_px[idx] = p;
    if (!lab) return;
    const float4 l = lab[idx];
    int4 g = make_int4(to_px(l.x, 0), to_px(l.y, 1), to_px(l.z, 2), to_px(l.w, 3));
    if (gt_px) gt_px[idx] = g;
    if (iou) {
        // numpy int32 arithmetic (wraps like C int)
        const int xa = max(p.x, g.x), ya = max(p.y, g.y);
        const int xb = min(p.z, g.z), yb = min(p.w, g.w);
        const int inter = max(xb - xa + 1, 0) * max(yb - ya + 1, 0);
        const int a1 = (p.z - p.x + 1) * (p.w - p.y + 1);
        const int a2 = (g.z - g.x + 1) * (g.w - g.y + 1);
        iou[idx] = (double)inter / (double)(a1 + a2 - inter);
    }
}
