// seq_kernels.hip - generic sequence-model kernels for the sibling reasoners of
// reference baselines/learned_models.py:92-197 (BaselineLstm, NonLinearLstm, TransformerLstm):
//   * lstm_stack_step  - L stacked bias-free LSTM layers + a 4-row linear head, one launch per time
//                        step, layers software-pipelined (layer l runs step s-l), same MFMA tile
//                        primitive and layouts as opnet_step (opnet_kernels.hip);
//   * slot_embed_relu  - relu(Linear 5->F) on the per-slot box features (learned_models.py:138,178);
//   * gemm_bias_act    - C = act(A W^T + b), fp32 MFMA, row-major operands (encoder projections/FFN);
//   * attention_f32    - softmax(Q K^T / sqrt(hd)) V over ONE sequence of S tokens, flash-style
//                        (online softmax), fp32 MFMA for both products;
//   * add_layernorm    - LayerNorm(x + y) (post-LN encoder layer);
//   * rows_to_packed   - [B][T][K] row-major -> the kq-major row-block layout the step kernels stream.
// Restated for checking in oracle/opnet_oracle.py (lstm_stack, encoder_layer, ...).
#include "opnet_ctx.h"

#define SEQ_MAX_LAYERS 3

struct StackLayer {
    const float4 *A;       // [H/4 tiles][nhx + H/16][64]  A tiles: K = [x part (nhx hexadecets) | h part]
    int H;                 // hidden size
    int nhx;               // hexadecets of the x part (input width padded to 16)
    float4 *hbuf;          // [2 parity | T+1][RB][H/4][32]
    float *c;              // [1 | T+1][RB][H][32]
    float4 *gsave;         // training: [T][RB][H][32] post-activation gates (i,f,g,o), later overwritten by da
    // input product kept out of the recurrent role: xg [T][RB][H][32] = W_ih x_t (gate float4 per unit and clip).
    // Layer 0 with a wide input: ONE GEMM before the recurrence ("hoisted").  Layers >= 1: the x-projection role of
    // the launch BEFORE the recurrent role's (W_ih h_{l-1,t} only needs the lower layer's h_t), so every role walks
    // 32 hexadecets - one register chunk per wave, no second fetch round trip inside a launch.
    // The recurrent role then walks only the h part of the A tiles (a_skip = x hexadecets to skip).
    // (Giving layer 0's narrow x part its own role as well was measured: the extra workgroups cost more than the
    // second chunk they remove - transformer_lstm B=1 2.06 -> 2.35 ms.)
    float4 *xg;
    int a_skip;
};

struct StackArgs {
    int B, T, RB, L;
    int train;             // 1: full-history buffers (slot t+1 = step t, slot 0 = zero state), gates saved
    const float4 *xp;      // layer 0 input, packed [T][RB][4*nhx0][32]
    StackLayer layer[SEQ_MAX_LAYERS];
    const float4 *headA;   // [H_last/16][64]  4 -> 16 rows
    float4 *ystage;        // [RB*32][T]
};

// grid.x = L * H/4 recurrent tiles + (L-1) * H/4 x-projection tiles + 1 ; grid.y <= RB.
// Launch s: layer l's recurrent role runs step t = s - 2l, its x-projection role (l >= 1) step t = s - (2l - 1) -
// one launch after layer l-1 produced h_t, one before the recurrent role consumes it - and the head t = s - (2L - 1).
// CH = register chunk (see load_a_chunk), NW = waves per workgroup: <4, 8> for one row block (a latency chain: 8 waves
// halve the MFMA chain and need one fetch round trip), <4, 4> above.
template <int CH, int NW = OPNET_NW>
__global__ void __launch_bounds__(NW * 64) lstm_stack_step(const StackArgs a, const int s)
{
    __shared__ __attribute__((aligned(16))) float part[NW * 8 * 64];
    const int tid = threadIdx.x;
    const int el = tid & 63, half = tid >> 6;
    const int clip = half * 16 + (el & 15), quarter = el >> 4;
    float4 a0[CH];

    int bx = blockIdx.x;
    int l = 0;
    for (l = a.L - 1; l >= 0; --l) {
        const int nt = a.layer[l].H >> 2;
        if (bx < nt) break;
        bx -= nt;
    }
    if (l < 0) {
        // ---- x-projection role of an upper layer: xg_l[t] = W_ih_l h_{l-1,t} ----
        int lx = 0;
        for (lx = a.L - 1; lx >= 1; --lx) {
            const int nt = a.layer[lx].H >> 2;
            if (bx < nt) break;
            bx -= nt;
        }
        if (lx >= 1) {
            const StackLayer &ly = a.layer[lx];
            const int t = s - (2 * lx - 1);
            if (t < 0 || t >= a.T) return;
            const int H = ly.H, nhh = H >> 4, nhx = ly.a_skip;
            const int tile = bx;
            const KSlice ks = wave_slice<NW>(nhx);
            const float4 *A = ly.A + (long)tile * (nhx + nhh) * 64;           // the x part leads every tile
            load_a_chunk(a0, A, ks.q0, ks.q1);
            int a_qb = ks.q0;
            const int unit = tile * 4 + quarter;
            for (int rb = blockIdx.y; rb < a.RB; rb += gridDim.y) {
                const long so = a.train ? t + 1 : (t & 1);
                const float4 *xseg = a.layer[lx - 1].hbuf + (so * a.RB + rb) * ((long)a.layer[lx - 1].H * 8);
                gemm16_rb(a0, a_qb, A, xseg, nhx, xseg, ks, part, s, a.B - rb * 32 > 16);
                __syncthreads();
                if (tid < 128)
                    ly.xg[(((long)t * a.RB + rb) * H + unit) * 32 + clip] =
                        make_float4(part_sum<NW>(part, half * 4 + 0, el), part_sum<NW>(part, half * 4 + 1, el),
                                    part_sum<NW>(part, half * 4 + 2, el), part_sum<NW>(part, half * 4 + 3, el));
                if (rb + (int)gridDim.y < a.RB) __syncthreads();
            }
            return;
        }
    }
    if (l >= 0) {
        const StackLayer &ly = a.layer[l];
        const int t = s - 2 * l;
        if (t < 0 || t >= a.T) return;
        const int H = ly.H, nhh = H >> 4, nhx = ly.nhx;
        const int tile = bx;
        const KSlice ks = wave_slice<NW>(nhx + nhh);
        const float4 *A = ly.A + ((long)tile * (ly.a_skip + nhx + nhh) + ly.a_skip) * 64;
        load_a_chunk(a0, A, ks.q0, ks.q1);
        int a_qb = ks.q0;
        const int unit = tile * 4 + quarter;
        for (int rb = blockIdx.y; rb < a.RB; rb += gridDim.y) {
            const long so = a.train ? t + 1 : (t & 1), sp = a.train ? t : ((t + 1) & 1);   // history slots
            const long co = a.train ? t + 1 : 0, cp = a.train ? t : 0;
            const float4 *xseg = a.xp + ((long)t * a.RB + rb) * ((long)nhx * 128);      // layer 0 only (nhx = 0 above it)
            const float4 *hprev = ly.hbuf + (sp * a.RB + rb) * ((long)H * 8);
            float c_old = 0.f;
            float4 xg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid < 128) {
                c_old = ly.c[((cp * a.RB + rb) * H + unit) * 32 + clip];
                if (ly.xg) xg = ly.xg[(((long)t * a.RB + rb) * H + unit) * 32 + clip];
            }
            gemm16_rb(a0, a_qb, A, xseg, nhx, hprev, ks, part, s, a.B - rb * 32 > 16);
            __syncthreads();
            if (tid < 128) {
                float c = c_old;
                float4 gs;
                const float h = lstm_cell_g(part_sum<NW>(part, half * 4 + 0, el) + xg.x, part_sum<NW>(part, half * 4 + 1, el) + xg.y,
                                            part_sum<NW>(part, half * 4 + 2, el) + xg.z, part_sum<NW>(part, half * 4 + 3, el) + xg.w, &c, &gs);
                ly.c[((co * a.RB + rb) * H + unit) * 32 + clip] = c;
                if (a.train) ly.gsave[(((long)t * a.RB + rb) * H + unit) * 32 + clip] = gs;
                float *hout = (float *)(ly.hbuf + (so * a.RB + rb) * ((long)H * 8));
                hout[((long)tile * 32 + clip) * 4 + quarter] = h;
            }
            if (rb + (int)gridDim.y < a.RB) __syncthreads();
        }
    } else {
        // head: predictions_layer (learned_models.py:101,113 / 137,148 / 172,195)
        if (bx != 0) return;   // padding workgroups (grid.x is rounded up to a multiple of the 8 XCDs)
        const int t = s - (2 * a.L - 1);
        if (t < 0 || t >= a.T) return;
        const StackLayer &ly = a.layer[a.L - 1];
        const int nh = ly.H >> 4;
        const KSlice ks = wave_slice<NW>(nh);
        load_a_chunk(a0, a.headA, ks.q0, ks.q1);
        int a_qb = ks.q0;
        for (int rb = blockIdx.y; rb < a.RB; rb += gridDim.y) {
            const float4 *hcur = ly.hbuf + ((a.train ? t + 1 : (long)(t & 1)) * a.RB + rb) * ((long)ly.H * 8);
            gemm16_rb(a0, a_qb, a.headA, hcur, nh, hcur, ks, part, s, a.B - rb * 32 > 16);
            __syncthreads();
            if (tid < 128 && quarter == 0) {
                const long b = rb * 32 + clip;
                float4 v;
                v.x = part_sum<NW>(part, half * 4 + 0, el);
                v.y = part_sum<NW>(part, half * 4 + 1, el);
                v.z = part_sum<NW>(part, half * 4 + 2, el);
                v.w = part_sum<NW>(part, half * 4 + 3, el);
                a.ystage[b * a.T + t] = v;
            }
            if (rb + (int)gridDim.y < a.RB) __syncthreads();
        }
    }
}

// x [B][T][K] row-major -> xp [t][rb][KP/4][clip][4], K padded with zeros to KP (multiple of 16),
// clips beyond B zero.  Also zeroes `state` (the recurrent buffers) - first kernel of a forward.
__global__ void __launch_bounds__(256) rows_to_packed(const float *__restrict__ x, float4 *__restrict__ xp,
                                                      int B, int T, int RB, int K, int KP,
                                                      float4 *__restrict__ state, long state_f4)
{
    const long stride = (long)gridDim.x * 256;
    const long gid = blockIdx.x * 256L + threadIdx.x;
    for (long i = gid; i < state_f4; i += stride) state[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int KQ = KP >> 2;
    const long n = (long)T * RB * KQ * 32;
    for (long idx = gid; idx < n; idx += stride) {
        // consecutive threads walk k within one (t, clip) row: coalesced reads
        const int kq = idx % KQ;
        long r = idx / KQ;
        const int clip = r & 31; r >>= 5;
        const int rb = r % RB;
        const int t = r / RB;
        const int b = rb * 32 + clip;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (b < B) {
            const float *src = x + ((long)b * T + t) * K + kq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (kq * 4 + e < K) v[e] = src[e];
        }
        xp[(((long)t * RB + rb) * KQ + kq) * 32 + clip] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// W_ih [4H][K] (torch gate-major rows: gate*H + unit) -> [4H][KP] with row unit*4 + gate, zero-padded columns:
// the weight operand of the hoisted input GEMM, whose output columns are then (unit, gate) = one float4 per unit
__global__ void __launch_bounds__(256) stack_pack_wih_rows(const float *__restrict__ w, float *__restrict__ out, int H,
                                                           int K, int KP)
{
    const long n = 4L * H * KP;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int k = idx % KP;
        const int row = idx / KP;
        const int unit = row >> 2, gate = row & 3;
        out[idx] = k < K ? w[((long)gate * H + unit) * K + k] : 0.f;
    }
}

// G [B*T][H] float4 (row b*T + t: the GEMM's pixel order over x [B][T][K]) -> xg [T][RB][H][32], clips past B zero
__global__ void __launch_bounds__(256) stack_xg_repack(const float4 *__restrict__ G, float4 *__restrict__ xg, int B, int T,
                                                       int RB, int H)
{
    const long n = (long)T * RB * 32 * H;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int unit = idx % H;            // consecutive threads walk the units of one (t, clip): coalesced reads
        long r = idx / H;
        const int clip = r & 31; r >>= 5;
        const int rb = r % RB;
        const int t = r / RB;
        const int b = rb * 32 + clip;
        xg[(((long)t * RB + rb) * H + unit) * 32 + clip] = b < B ? G[((long)b * T + t) * H + unit] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ystage [RB*32][T] float4 -> y [B][T][4]
__global__ void __launch_bounds__(256) copy_y_out(const float4 *__restrict__ ys, float4 *__restrict__ y, long n)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = ys[i];
}

// ------------------------------------------------------------------------------------------------
// backward through the stacked LSTM (BPTT) - generic counterpart of opnet_bwd_gemm / opnet_bwd_cell
// ------------------------------------------------------------------------------------------------
// Layer l (0 = bottom) lags the top layer by lam_l = L-1-l launch pairs.  Launch pair n:
//   stack_bwd_cell(n): layer l, t = T-1-n+lam_l:   dh = [l == L-1 ? W_head^T dy_t : sum DX_{l+1} partials]
//                                                      + [t < T-1 ? sum R_l partials : 0]  -> cell backward -> da_l(t)
//   stack_bwd_gemm(n): for every layer at its t:   R_l  = W_hh_l^T da_l(t)   (split-K x4) -> dh_l(t-1) partials
//                      for l >= 1:                 DX_l = W_ih_l^T da_l(t)   (split-K x4) -> dh_{l-1}(t) partials
// da overwrites the saved gates in place (k = 4*unit + gate layout), exactly as in opnet_train_kernels.hip.
struct StackBwdLayer {
    const float4 *whh_t;   // [H/16][H/4][64]  W_hh^T tiles, k = 4*unit' + gate
    const float4 *wih_t;   // l >= 1: [H/16][H/4][64]  W_ih^T tiles (rows = units of layer l-1)
    float4 *g;             // [T][RB][H][32]  gates in / da out
    const float *call;     // [T+1][RB][H][32]
    float *rpart;          // [4][RB][H][32]
    float *dxpart;         // l >= 1: [4][RB][H][32] partials of dh_{l-1}
    float *dc;             // [RB][H][32]
};

struct StackBwdArgs {
    int B, T, RB, L, H;
    StackBwdLayer layer[SEQ_MAX_LAYERS];
    const float4 *dyp;     // [T][RB][32]
    const float *whead;    // [4][H] raw
};

// grid.x = L * 4*(H/16) R tiles + (L-1) * 4*(H/16) DX tiles ; grid.y <= RB
__global__ void __launch_bounds__(OPNET_THREADS) stack_bwd_gemm(const StackBwdArgs a, const int n)
{
    __shared__ __attribute__((aligned(16))) float part[OPNET_NW * 8 * 64];
    const int s = n;
    const int H = a.H, per = 4 * (H >> 4);
    int bx = blockIdx.x;
    const int kind = bx >= a.L * per;               // 0: R product, 1: DX product
    if (kind) bx -= a.L * per;
    const int l = kind ? 1 + bx / per : bx / per;
    bx -= (kind ? l - 1 : l) * per;
    const int tile = bx >> 2, ks = bx & 3;
    const int t = a.T - 1 - n + (a.L - 1 - l);
    if (t < 0 || t >= a.T) return;
    const StackBwdLayer &ly = a.layer[l];
    const float4 *A = (kind ? ly.wih_t : ly.whh_t) + ((long)tile * (H >> 2) + ks * (H >> 4)) * 64;
    float *dst = kind ? ly.dxpart : ly.rpart;
    const int tid = threadIdx.x, el = tid & 63, half = tid >> 6;
    const int clip = half * 16 + (el & 15), quarter = el >> 4;
    float4 a0[OPNET_CH];
    const int nh = H >> 4;
    const KSlice ksl = wave_slice(nh);
    load_a_chunk(a0, A, ksl.q0, ksl.q1);
    int a_qb = ksl.q0;
    for (int rb = blockIdx.y; rb < a.RB; rb += gridDim.y) {
        const float4 *seg = ly.g + (((long)t * a.RB + rb) * H + (long)ks * (H >> 2)) * 32;
        gemm16_rb(a0, a_qb, A, seg, nh, seg, ksl, part, s, a.B - rb * 32 > 16);
        __syncthreads();
        if (tid < 128) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tile * 16 + quarter * 4 + r;
                dst[(((long)ks * a.RB + rb) * H + row) * 32 + clip] = part_sum(part, half * 4 + r, el);
            }
        }
        if (rb + (int)gridDim.y < a.RB) __syncthreads();
    }
}

// grid.x = L * H/8 ; grid.y = RB
__global__ void __launch_bounds__(256) stack_bwd_cell(const StackBwdArgs a, const int n)
{
    const int H = a.H, per = H >> 3;
    const int l = blockIdx.x / per, chunk = blockIdx.x - l * per;
    const int rb = blockIdx.y;
    const int t = a.T - 1 - n + (a.L - 1 - l);
    if (t < 0 || t >= a.T) return;
    const StackBwdLayer &ly = a.layer[l];
    const int tid = threadIdx.x, clip = tid & 31;
    const int u = chunk * 8 + (tid >> 5);
    const long e = ((long)rb * H + u) * 32 + clip;
    const long ps = (long)a.RB * H * 32;
    float dh;
    if (l == a.L - 1) {
        const float4 dy = a.dyp[((long)t * a.RB + rb) * 32 + clip];
        dh = a.whead[u] * dy.x;
        dh = fmaf(a.whead[H + u], dy.y, dh);
        dh = fmaf(a.whead[2 * H + u], dy.z, dh);
        dh = fmaf(a.whead[3 * H + u], dy.w, dh);
    } else {
        const float *dx = a.layer[l + 1].dxpart;
        dh = ((dx[e] + dx[ps + e]) + dx[2 * ps + e]) + dx[3 * ps + e];
    }
    float dcc = 0.f;
    if (t < a.T - 1) {
        dh += ((ly.rpart[e] + ly.rpart[ps + e]) + ly.rpart[2 * ps + e]) + ly.rpart[3 * ps + e];
        dcc = ly.dc[e];
    }
    const long ge = (((long)t * a.RB + rb) * H + u) * 32 + clip;
    const float c_t = ly.call[(((long)(t + 1)) * a.RB + rb) * H * 32 + (long)u * 32 + clip];
    const float c_p = ly.call[((long)t * a.RB + rb) * H * 32 + (long)u * 32 + clip];
    float dco;
    ly.g[ge] = cell_backward(dh, dcc, ly.g[ge], c_t, c_p, &dco);
    ly.dc[e] = dco;
}

// da0 in the packed layout [T][RB][H][32] float4 (k = 4*unit + gate) -> row-major [B*T][4H] with the SAME k order,
// the A operand of the input-gradient GEMM  dx0 = da0 . W_ih0  (boxes_linear / encoder backward)
__global__ void __launch_bounds__(256) packed_da_to_rows(const float4 *__restrict__ g, float *__restrict__ out, int B,
                                                         int T, int RB, int H)
{
    const long n = (long)B * T * H;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int u = idx % H;
        const long bt = idx / H;
        const int t = bt % T;
        const int b = bt / T;
        ((float4 *)out)[idx] = g[(((long)t * RB + (b >> 5)) * H + u) * 32 + (b & 31)];
    }
}

// out[(b*T+t)*nslots_out + slot][f] = relu(sum_k W[f][k] * x[b][t][slot][k]), k < 5
// nslots_out = 15 (NonLinearLstm, learned_models.py:138) or 1 = slot 0 only (TransformerLstm's live path).
__global__ void __launch_bounds__(256) slot_embed_relu(const float *__restrict__ x, const float *__restrict__ W,
                                                       float *__restrict__ out, long ntok, int nslots_out, int F)
{
    const long n = ntok * nslots_out * F;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int f = idx % F;
        const long ts = idx / F;
        const int slot = ts % nslots_out;
        const long tok = ts / nslots_out;
        const float *xi = x + (tok * 15 + slot) * 5;
        const float *w = W + (long)f * 5;
        float acc = w[0] * xi[0];
        acc = fmaf(w[1], xi[1], acc);
        acc = fmaf(w[2], xi[2], acc);
        acc = fmaf(w[3], xi[3], acc);
        acc = fmaf(w[4], xi[4], acc);
        out[idx] = fmaxf(acc, 0.f);
    }
}

// gradient of boxes_linear.weight [F][5] through relu: dW[f][k] = sum_{tok,slot} [out > 0] * dout * x[tok][slot][k]
// one workgroup per output feature f; fixed-order block reduction (deterministic)
__global__ void __launch_bounds__(256) slot_embed_relu_bwd(const float *__restrict__ x, const float *__restrict__ out,
                                                           const float *__restrict__ dout, float *__restrict__ dW,
                                                           long ntok, int nslots_out, int F)
{
    __shared__ float red[5][256];
    const int f = blockIdx.x;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const long n = ntok * nslots_out;
    for (long i = threadIdx.x; i < n; i += 256) {
        const long tok = i / nslots_out;
        const int slot = i - tok * nslots_out;
        const long o = i * F + f;
        const float g = out[o] > 0.f ? dout[o] : 0.f;
        const float *xi = x + (tok * 15 + slot) * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] = fmaf(g, xi[k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < 5; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 5) dW[f * 5 + threadIdx.x] = red[threadIdx.x][0];
}

// the same gradient with the rows of out / dout read as they lie (thread = feature: 1 KB per row and array at F = 256; slot_embed_relu_bwd
// walks a feature's column with a stride of F floats): workgroup b adds up rows [b * rpb, (b + 1) * rpb) into part[b][f][5]
// (the row's five inputs are the same for every thread), slot_embed_bwd_final adds the workgroups' partials in order
__global__ void __launch_bounds__(256) slot_embed_relu_bwd_part(const float *__restrict__ x, const float *__restrict__ out,
                                                                const float *__restrict__ dout, float *__restrict__ part,
                                                                long nrows, int nslots_out, int F, long rpb)
{
    const long r0 = blockIdx.x * rpb, r1 = min(nrows, r0 + rpb);
    for (int f = threadIdx.x; f < F; f += 256) {
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        long tok = r0 / nslots_out;
        int slot = (int)(r0 - tok * nslots_out);
        const float *po = out + r0 * F + f, *pd = dout + r0 * F + f;
#pragma unroll 4
        for (long r = r0; r < r1; ++r) {
            const float *xi = x + (tok * 15 + slot) * 5;
            const float g = *po > 0.f ? *pd : 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[k] = fmaf(g, xi[k], acc[k]);
            po += F; pd += F;
            if (++slot == nslots_out) { slot = 0; ++tok; }
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) part[((long)blockIdx.x * F + f) * 5 + k] = acc[k];
    }
}
__global__ void __launch_bounds__(256) slot_embed_bwd_final(const float *__restrict__ part, float *__restrict__ dW, int nblocks, int F)
{
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;                 // (f, k); wave w adds the workgroups w, w + 4, ..., the four meet in LDS
    float s = 0.f;
    if (i < F * 5) {
#pragma unroll 8
        for (int b = w; b < nblocks; b += 4) s += part[(long)b * F * 5 + i];
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < F * 5) dW[i] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// ------------------------------------------------------------------------------------------------
// C[M][N] = act(A[M][K] * W[N][K]^T + bias[N])   (row-major fp32; K % 16 == 0)
// ------------------------------------------------------------------------------------------------
// Workgroup tile 64 x 64, wave (wm, wn) owns 32 x 32 = 2 x 2 fragments of v_mfma_f32_16x16x4_f32.
// Both operands are K-contiguous, so a lane's float4 at k = 16q + 4(l>>4) .. +3 of row (l&15) feeds
// four consecutive MFMAs (the same "hexadecet" trick as the step kernels); no LDS staging - the
// operands are L2-resident at these sizes (cdna guide, common mistake 7).
__global__ void __launch_bounds__(256) gemm_bias_act(const float *__restrict__ A, const float *__restrict__ W,
                                                     const float *__restrict__ bias, float *__restrict__ C,
                                                     int M, int N, int K, int act)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 64 + (w >> 1) * 32, n0 = blockIdx.y * 64 + (w & 1) * 32;
    const int i = lane & 15, kk = lane >> 4;
    const float4 *a_row[2], *w_row[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int m = min(m0 + f * 16 + i, M - 1), nn = min(n0 + f * 16 + i, N - 1);
        a_row[f] = (const float4 *)(A + (long)m * K) + kk;
        w_row[f] = (const float4 *)(W + (long)nn * K) + kk;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nhex = K >> 4;
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
    // D layout: lane holds column j = l&15, rows 4*(l>>4) + r
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int nn = n0 + y * 16 + i;
            if (nn >= N) continue;
            const float b = bias ? bias[nn] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + x * 16 + kk * 4 + r;
                if (m < M) {
                    float v = acc[x][y][r] + b;
                    if (act == 1) v = fmaxf(v, 0.f);
                    C[(long)m * N + nn] = v;
                }
            }
        }
}

// The same product for SMALL M (one clip: S = 300 tokens - config 3): a workgroup owns a 32 x 32 output tile and its four
// waves split K, so that linear2 (K = 2048, N = 256) is 80 workgroups of 32 hexadecets a wave instead of 20 workgroups whose
// waves each walk 128 dependent load -> MFMA rounds (measured 28 us per call on average over the encoder's eight GEMMs at
// S = 300: 222 us of a 0.96 ms forward).  Partials meet in LDS and are summed in fixed wave order (deterministic).  K % 16 == 0.
__global__ void __launch_bounds__(256) gemm_bias_act_ks(const float *__restrict__ A, const float *__restrict__ W,
                                                        const float *__restrict__ bias, float *__restrict__ C,
                                                        int M, int N, int K, int act, const float *__restrict__ R = nullptr)
{
    __shared__ __attribute__((aligned(16))) float4 red[4][4][64];      // [wave][fragment][lane]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int i = lane & 15, kk = lane >> 4;
    // this wave's share of K: hexadecets [w * (K / 16) / 4, (w + 1) * (K / 16) / 4) - a quarter of K when K % 64 == 0
    const int nh16 = K >> 4, q0 = (w * nh16) >> 2, q1 = ((w + 1) * nh16) >> 2;
    const float4 *a_row[2], *w_row[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int m = min(m0 + f * 16 + i, M - 1), nn = min(n0 + f * 16 + i, N - 1);
        a_row[f] = (const float4 *)(A + (long)m * K + (long)q0 * 16) + kk;
        w_row[f] = (const float4 *)(W + (long)nn * K + (long)q0 * 16) + kk;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nhex = q1 - q0;
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
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) red[w][x * 2 + y][lane] = make_float4(acc[x][y][0], acc[x][y][1], acc[x][y][2], acc[x][y][3]);
    __syncthreads();
    // thread (fragment f = wave, lane): column j = lane & 15, rows 4 * (lane >> 4) + r of fragment (x, y) = (f >> 1, f & 1)
    const int x = w >> 1, y = w & 1;
    const float4 p0 = red[0][w][lane], p1 = red[1][w][lane], p2 = red[2][w][lane], p3 = red[3][w][lane];
    const float v[4] = {((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                        ((p0.w + p1.w) + p2.w) + p3.w};
    const int nn = n0 + y * 16 + i;
    if (nn < N) {
        const float b = bias ? bias[nn] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + x * 16 + kk * 4 + r;
            if (m < M) {
                float o = v[r] + b;
                if (R) o += R[(long)m * N + nn];          // (a residual [M][N], added before the activation)
                if (act == 1) o = fmaxf(o, 0.f);
                C[(long)m * N + nn] = o;
            }
        }
    }
}

// The token-contracting product of a SHORT sequence, without transposed copies of its operands:
//   C[n][k] = sum_s A[s * lda + n] B[s * ldb + k]      (dW = dY^T X of a linear layer: A = dY [S][N], B = X [S][K])
// 32 x 32 output tiles, the sequence split over the workgroup's four waves in steps of 16 rows; a lane's operands are scalars
// (16 lanes read 64 consecutive bytes of a row), rows beyond S count as zeros.  Partials meet in LDS, fixed wave order.
__global__ void __launch_bounds__(256) gemm_tn_ks(const float *__restrict__ A, long lda, const float *__restrict__ B, long ldb,
                                                  float *__restrict__ C, int S, int N, int K)
{
    __shared__ __attribute__((aligned(16))) float4 red[4][4][64];      // [wave][fragment][lane]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int i = lane & 15, kk = lane >> 4;
    const int nch = (S + 15) >> 4, c0 = (w * nch) >> 2, c1 = ((w + 1) * nch) >> 2;
    const float *ap[2], *bp[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        ap[f] = A + min(n0 + f * 16 + i, N - 1);
        bp[f] = B + min(k0 + f * 16 + i, K - 1);
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int c = c0; c < c1; ++c) {
        float ae[2][4], be[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int srow = c * 16 + kk * 4 + e;
            const bool in = srow < S;
            const long r = in ? srow : 0;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const float av = ap[f][r * lda], bv = bp[f][r * ldb];
                ae[f][e] = in ? av : 0.f;
                be[f][e] = in ? bv : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[x][e], be[y][e], acc[x][y], 0, 0, 0);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) red[w][x * 2 + y][lane] = make_float4(acc[x][y][0], acc[x][y][1], acc[x][y][2], acc[x][y][3]);
    __syncthreads();
    const int x = w >> 1, y = w & 1;
    const float4 p0 = red[0][w][lane], p1 = red[1][w][lane], p2 = red[2][w][lane], p3 = red[3][w][lane];
    const float v[4] = {((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                        ((p0.w + p1.w) + p2.w) + p3.w};
    const int kc = k0 + y * 16 + i;
    if (kc < K) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + x * 16 + kk * 4 + r;
            if (n < N) C[(long)n * K + kc] = v[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out[S][E] = LayerNorm(x + y) * g + b   (nn.LayerNorm: biased variance, eps inside the sqrt)
// ------------------------------------------------------------------------------------------------
// one wave per row; E <= 64 * LN_MAX_PER_LANE
#define LN_MAX_PER_LANE 8
__global__ void __launch_bounds__(256) add_layernorm(const float *__restrict__ x, const float *__restrict__ y,
                                                     const float *__restrict__ g, const float *__restrict__ b,
                                                     float *__restrict__ out, int S, int E, float eps)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= S) return;
    float v[LN_MAX_PER_LANE];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
        const int e = j * 64 + lane;
        v[j] = e < E ? x[(long)row * E + e] + y[(long)row * E + e] : 0.f;
        sum += v[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mu = sum / (float)E;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
        const int e = j * 64 + lane;
        const float d = e < E ? v[j] - mu : 0.f;
        var += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.0f / sqrtf(var / (float)E + eps);
#pragma unroll
    for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
        const int e = j * 64 + lane;
        if (e < E) out[(long)row * E + e] = (v[j] - mu) * rstd * g[e] + b[e];
    }
}

// ------------------------------------------------------------------------------------------------
// attention over one sequence: O[s][h*hd + d] = sum_k softmax_k(q_s . k_k / sqrt(hd)) v_k[d]
// ------------------------------------------------------------------------------------------------
// qkv [S][3E] (q | k | v), head h uses columns h*hd .. of each third.  hd % 16 == 0, hd <= 128.
// One wave = 16 queries of one head; a workgroup = 4 waves = 64 queries.  Per block of 16 keys:
//   S^T[key][query] = K_blk Q^T          (A = K rows, B = Q rows; hd/4 MFMAs of 16x16x4)
//   online softmax over keys: a lane holds 4 keys (rows 4*(l>>4)+r) of ONE query (column l&15), so the
//     row statistics are a 4-register reduction plus two cross-group shuffles (xor 16, 32)
//   O^T[d][query] += V_blk^T P^T         (B = the score registers themselves: MFMA r contracts keys
//     {r, 4+r, 8+r, 12+r}; A = V read as float4 along d, element e of the float4 -> accumulator e
//     holding d = 64c + 4i + e)
// so neither product needs a transpose or an LDS round trip.
#define ATT_MAX_HEX 8  // hd / 16
__global__ void __launch_bounds__(256) attention_f32(const float *__restrict__ qkv, float *__restrict__ out,
                                                     int S, int E, int hd, float scale)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int head = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + w) * 16;
    if (q0 >= S) return;
    const int i = lane & 15, kk = lane >> 4;
    const int nhex = hd >> 4;
    const int nc = (hd + 63) >> 6;  // 64-wide d chunks
    const long ld = 3L * E;
    const float *Qb = qkv + (long)head * hd;
    const float *Kb = qkv + E + (long)head * hd;
    const float *Vb = qkv + 2L * E + (long)head * hd;

    // Q fragments (B operand of S^T): lane (query j = i, kk) holds Q[q0+j][16c + 4kk .. +3] * scale
    float4 qf[ATT_MAX_HEX];
    {
        const int qi = min(q0 + i, S - 1);
        const float4 *qp = (const float4 *)(Qb + (long)qi * ld) + kk;
#pragma unroll
        for (int c = 0; c < ATT_MAX_HEX; ++c)
            if (c < nhex) {
                float4 v = qp[c * 4];
                qf[c] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
            }
    }
    f32x4 o[2][4];  // [d chunk c][element e] -> rows i' = 4*(l>>4)+r  <->  d = 64c + 4i' + e
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[c][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < S; k0 += 16) {
        // ---- scores^T for 16 keys ----
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        {
            const int ki = min(k0 + i, S - 1);
            const float4 *kp = (const float4 *)(Kb + (long)ki * ld) + kk;
            float4 kf[ATT_MAX_HEX];
#pragma unroll
            for (int c = 0; c < ATT_MAX_HEX; ++c)
                if (c < nhex) kf[c] = kp[c * 4];
#pragma unroll
            for (int c = 0; c < ATT_MAX_HEX; ++c)
                if (c < nhex) {
                    sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c].x, qf[c].x, sc, 0, 0, 0);
                    sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c].y, qf[c].y, sc, 0, 0, 0);
                    sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c].z, qf[c].z, sc, 0, 0, 0);
                    sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c].w, qf[c].w, sc, 0, 0, 0);
                }
        }
        // V fragments for this key block, issued before the softmax arithmetic
        float4 vf[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = min(k0 + 4 * kk + r, S - 1);
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (c < nc) {
                    const int d = 64 * c + 4 * i;
                    vf[r][c] = d < hd ? *(const float4 *)(Vb + (long)key * ld + d) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
        // ---- online softmax: this lane holds keys k0 + 4*kk + r of query q0 + i ----
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (k0 + 4 * kk + r >= S) sc[r] = -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);   // first block: exp(-inf) = 0
        float p[4], ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[r] = __expf(sc[r] - m_new);
            ps += p[r];
        }
        ps += __shfl_xor(ps, 16);
        ps += __shfl_xor(ps, 32);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        // ---- O^T = O^T * alpha + V^T P^T ----
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (c < nc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[c][e][0] *= alpha; o[c][e][1] *= alpha; o[c][e][2] *= alpha; o[c][e][3] *= alpha;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].x, p[r], o[c][0], 0, 0, 0);
                    o[c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].y, p[r], o[c][1], 0, 0, 0);
                    o[c][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].z, p[r], o[c][2], 0, 0, 0);
                    o[c][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c].w, p[r], o[c][3], 0, 0, 0);
                }
            }
    }
    // ---- normalise and store: lane holds query q0 + i, d = 64c + 4*(4*kk + r) + e ----
    if (q0 + i < S) {
        const float inv = 1.0f / l_run;
        float *op = out + (long)(q0 + i) * E + (long)head * hd;
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (c < nc)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 64 * c + 4 * (4 * kk + r);
                    if (d < hd)
                        *(float4 *)(op + d) = make_float4(o[c][0][r] * inv, o[c][1][r] * inv, o[c][2][r] * inv, o[c][3][r] * inv);
                }
    }
}
