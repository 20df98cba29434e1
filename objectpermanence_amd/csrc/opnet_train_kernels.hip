// opnet_train_kernels.hip - backward pass (BPTT), weight-gradient GEMMs, L1 loss and Adam for OPNet.
//
// Replaces what torch autograd + torch.optim.Adam do for the reference's training step
// (reference baselines/training_main.py:150-152 Adam / L1Loss, :183-217 step; model
// baselines/learned_models.py:35-52).  Restated for checking in oracle/torch_port.py.
//
// Backward recurrence.  With a_t the pre-activation gates and (i,f,g,o) their activations
//     dh_t   = [upstream]_t + W_hh^T da_{t+1}                 (all-to-all over 4H gate columns)
//     dc_t   = dc_carry + dh_t * o * (1 - tanh(c_t)^2) ;  dc_carry' = dc_t * f
//     da_t   = ( dc*g*i(1-i), dc*c_{t-1}*f(1-f), dc*i*(1-g^2), dh*tanh(c_t)*o(1-o) )
// The reduction dimension of the recurrent product is 4H (4x the forward's), so a workgroup that
// owned complete dh rows would have to read all of da (256 KB per workgroup at B=32); instead the
// product is split-K over 4 workgroups per 16-unit tile (same per-workgroup traffic as a forward
// tile) and the 4 partials are summed, in fixed order, by the cell kernel of the next launch.
// Per reverse step there are therefore two launches (launch index n = 0, 1, ...):
//     opnet_bwd_cell(n):  LSTM2 cell backward at t = T-1-n   |  head + LSTM1 cell backward at t = T-n
//     opnet_bwd_gemm(n):  W_hh2^T da2_t, W_ih2^T da2_t (t = T-1-n)  |  W_hh1^T da1_t (t = T-n)
// da_t overwrites the saved gates in place ([T][RB][H][32] float4 = (unit, clip) -> 4 gate values,
// which is exactly the kq-major activation layout with k = 4*unit + gate), so that afterwards every
// weight gradient is one GEMM  dW[m][n] = sum_{t,clip} P_t[clip][m] * Q_t[clip][n]  over the saved
// histories (opnet_wgrad).
#include "opnet_ctx.h"

struct BwdArgs {
    int B, T, RB, H1, H2;
    int rb0, rb1;           // opnet_bwd_fused works on row blocks [rb0, rb1): one launch chain per slice of the batch, side by side
    int mlp;                // OPNetLstmMlp: no video LSTM - g2 holds (d hidden, 0, 0, 0), filled by opnet_mlp_dhid
    // saved by the training forward
    const float4 *xp;       // [T][RB][24][32]
    const float4 *h1all;    // [T+1][RB][H1/4][32]   slot t+1 = h1_t, slot 0 = 0
    const float *c1all;     // [T+1][RB][H1][32]
    const float4 *h2all;    // [T+1][RB][H2/4][32]
    const float *c2all;     // [T+1][RB][H2][32]
    const float4 *x2all;    // [T][RB][2][32]
    const float4 *psave;    // [T][RB][4][32]
    float4 *g1;             // [T][RB][H1][32]  gates in, da1 out
    float4 *g2;             // [T][RB][H2][32]  gates in, da2 out
    // backward state
    const float4 *dyp;      // [T][RB][32]   upstream gradient of y_boxes, packed
    float4 *dlall;          // [T][RB][4][32] gradient of the selection logits (15 -> 16)
    float *dhpart2;         // [4][RB][H2][32] split-K partials of W_hh2^T da2
    float *dhpart1;         // [4][RB][H1][32]
    float *dx2part;         // [4][RB][16][32]  split-K partials of W_ih2^T da2 (6 rows valid)
    float *dc2;             // [RB][H2][32] cell-gradient carry
    float *dc1;             // [RB][H1][32]
    // backward weights
    const float4 *w2bt;     // [H2/16][4*H2/16][64]  W_hh2^T tiles, k = 4*unit' + gate
    const float4 *w1bt;     // [H1/16][4*H1/16][64]
    const float4 *wih2t;    // [4*H2/16][64]         W_ih2^T (6 -> 16 rows)
    const float *wsel;      // [15][H1] raw
    const float *wout;      // [4][H2] raw
};

// ------------------------------------------------------------------------------------------------
// transposed weight tiles for the backward recurrence
// ------------------------------------------------------------------------------------------------
// out[tile][q][lane][e], row i = lane&15, k = 16q + 4(lane>>4) + e  with  k = 4*unit' + gate.
// mode 0: A[row u = tile*16+i][k] = W_hh[gate*H + unit'][u]          (W_hh is [4H][H])
// mode 1: A[row f = i][k]         = W_ih[gate*H + unit'][f], f < nf   (W_ih is [4H][nf]); one tile
__global__ void opnet_pack_tiles_t(float *__restrict__ out, const float *__restrict__ w, int H, int nf,
                                   int mode, int ntiles)
{
    const int nhex = (4 * H) / 16;
    const long total = (long)ntiles * nhex * 256;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int e = idx & 3;
        const int lane = (idx >> 2) & 63;
        const long tq = idx >> 8;
        const int q = tq % nhex;
        const int tile = tq / nhex;
        const int i = lane & 15;
        const int k = 16 * q + 4 * (lane >> 4) + e;
        const int unit = k >> 2, gate = k & 3;
        const long wrow = (long)gate * H + unit;
        float v;
        if (mode == 0)
            v = w[wrow * H + tile * 16 + i];
        else
            v = i < nf ? w[wrow * nf + i] : 0.f;
        out[idx] = v;
    }
}

__global__ void opnet_copy_f32(float *__restrict__ dst, const float *__restrict__ src, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// dy [B][T][4] -> dyp [T][RB][32] float4 (zero for clips beyond B); also zeroes the cell-gradient carries
__global__ void __launch_bounds__(256) opnet_pack_dy(const float4 *__restrict__ dy, float4 *__restrict__ dyp,
                                                     float *__restrict__ dc_zero, long n_dc, int B, int T, int RB)
{
    const long n = (long)T * RB * 32;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += stride) {
        const int clip = idx & 31;
        const long trb = idx >> 5;
        const int rb = trb % RB;
        const int t = trb / RB;
        const int b = rb * 32 + clip;
        dyp[idx] = b < B ? dy[(long)b * T + t] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n_dc; idx += stride) dc_zero[idx] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// backward step, launch A: split-K recurrent products
// ------------------------------------------------------------------------------------------------
// grid.x = 4*(H2/16) LSTM2 tiles x K-quarters + 4*(H1/16) LSTM1 + 4 (W_ih2^T) ; grid.y <= RB
__global__ void __launch_bounds__(OPNET_THREADS) opnet_bwd_gemm(const BwdArgs a, const int n)
{
    __shared__ __attribute__((aligned(16))) float part[OPNET_NW * 8 * 64];
    const int s = n;  // (trace hook name used by the shared core)
    const int T = a.T, H1 = a.H1, H2 = a.H2;
    const int n2 = 4 * (H2 >> 4), n1 = 4 * (H1 >> 4);
    const int bx = blockIdx.x;
    const int tid = threadIdx.x;
    const int el = tid & 63, half = tid >> 6;
    const int clip = half * 16 + (el & 15), quarter = el >> 4;
    float4 a0[OPNET_CH];

    int t, H, tile, ks, nrows_out;
    const float4 *A;
    const float4 *da;
    float *dst;
    if (bx < n2) {
        if (a.mlp) return;   // relu(Linear) has no recurrence
        t = T - 1 - n; H = H2; tile = bx >> 2; ks = bx & 3;
        A = a.w2bt + ((long)tile * (H2 >> 2) + ks * (H2 >> 4)) * 64;   // tile has 4H/16 = H/4 hexadecets
        da = a.g2; dst = a.dhpart2; nrows_out = H2;
    } else if (bx < n2 + n1) {
        t = T - n; H = H1; tile = (bx - n2) >> 2; ks = (bx - n2) & 3;
        A = a.w1bt + ((long)tile * (H1 >> 2) + ks * (H1 >> 4)) * 64;
        da = a.g1; dst = a.dhpart1; nrows_out = H1;
    } else {
        t = T - 1 - n; H = H2; tile = 0; ks = bx - n2 - n1;
        A = a.wih2t + (long)ks * (H2 >> 4) * 64;
        da = a.g2; dst = a.dx2part; nrows_out = 16;
    }
    if (t < 0 || t >= T) return;
    const int nh = H >> 4;  // hexadecets in this K quarter ( (4H/16) / 4 )
    const KSlice ksl = wave_slice(nh);
    load_a_chunk(a0, A, ksl.q0, ksl.q1);
    int a_qb = ksl.q0;
    for (int rb = blockIdx.y; rb < a.RB; rb += gridDim.y) {
        // da_t as a kq-major activation segment: kq = unit, this quarter starts at unit ks*H/4
        const float4 *seg = da + (((long)t * a.RB + rb) * H + (long)ks * (H >> 2)) * 32;
        gemm16_rb(a0, a_qb, A, seg, nh, seg, ksl, part, s);
        __syncthreads();
        if (tid < 128) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tile * 16 + quarter * 4 + r;
                dst[(((long)ks * a.RB + rb) * nrows_out + row) * 32 + clip] = part_sum(part, half * 4 + r, el);
            }
        }
        if (rb + (int)gridDim.y < a.RB) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// backward step, launch B: cell backward (+ head backward for LSTM1)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 cell_backward(float dh, float dc_carry, float4 g, float c_t, float c_prev,
                                                float *dc_out)
{
    const float i = g.x, f = g.y, gg = g.z, o = g.w;
    const float tc = fast_tanh(c_t);
    const float dc = dc_carry + dh * o * (1.0f - tc * tc);
    *dc_out = dc * f;
    float4 da;
    da.x = dc * gg * i * (1.0f - i);
    da.y = dc * c_prev * f * (1.0f - f);
    da.z = dc * i * (1.0f - gg * gg);
    da.w = dh * tc * o * (1.0f - o);
    return da;
}

// grid.x = H2/8 (LSTM2, 8 units x 32 clips per workgroup) + H1/8 (LSTM1) ; grid.y = RB
__global__ void __launch_bounds__(256) opnet_bwd_cell(const BwdArgs a, const int n)
{
    __shared__ float dl_s[32][17];
    const int T = a.T, H1 = a.H1, H2 = a.H2;
    const int nc2 = H2 >> 3;
    const int bx = blockIdx.x, rb = blockIdx.y;
    const int tid = threadIdx.x;
    const int clip = tid & 31;
    if (bx < nc2) {
        // ---------------- LSTM2 at t = T-1-n ----------------
        const int t = T - 1 - n;
        if (t < 0 || a.mlp) return;
        const int u = bx * 8 + (tid >> 5);
        const long e = ((long)rb * H2 + u) * 32 + clip;
        // upstream: prediction_layer (learned_models.py:47): dh += W_out^T dy_t
        const float4 dy = a.dyp[((long)t * a.RB + rb) * 32 + clip];
        float dh = a.wout[u] * dy.x;
        dh = fmaf(a.wout[H2 + u], dy.y, dh);
        dh = fmaf(a.wout[2 * H2 + u], dy.z, dh);
        dh = fmaf(a.wout[3 * H2 + u], dy.w, dh);
        float dcc = 0.f;
        if (t < T - 1) {
            const long ps = (long)a.RB * H2 * 32;
            dh += ((a.dhpart2[e] + a.dhpart2[ps + e]) + a.dhpart2[2 * ps + e]) + a.dhpart2[3 * ps + e];
            dcc = a.dc2[e];
        }
        const long ge = (((long)t * a.RB + rb) * H2 + u) * 32 + clip;
        const float c_t = a.c2all[(((long)(t + 1)) * a.RB + rb) * H2 * 32 + (long)u * 32 + clip];
        const float c_p = a.c2all[((long)t * a.RB + rb) * H2 * 32 + (long)u * 32 + clip];
        float dco;
        a.g2[ge] = cell_backward(dh, dcc, a.g2[ge], c_t, c_p, &dco);
        a.dc2[e] = dco;
    } else {
        // ---------------- head backward + LSTM1 at t = T-n ----------------
        const int t = T - n;
        if (t < 0 || t >= T) return;
        if (tid < 32) {
            // d frames_boxes[f] = sum of the split-K partials of W_ih2^T da2_t   (LSTM2 input part)
            float dx[OPNET_FEATS_];
            const long ps = (long)a.RB * 16 * 32;
#pragma unroll
            for (int f = 0; f < OPNET_FEATS_; ++f) {
                const long e = ((long)rb * 16 + f) * 32 + clip;
                dx[f] = ((a.dx2part[e] + a.dx2part[ps + e]) + a.dx2part[2 * ps + e]) + a.dx2part[3 * ps + e];
            }
            // einsum backward: dp[o] = sum_f boxes[o][f] dx[f]; softmax backward: dl = p * (dp - <p, dp>)
            const float *xs = (const float *)(a.xp + ((long)t * a.RB + rb) * (OPNET_KXQ * 32));
            const float4 *pp = a.psave + ((long)t * a.RB + rb) * 128 + clip;
            float p[16], dp[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = pp[q * 32];
                p[4 * q] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
            }
            float dot = 0.f;
#pragma unroll
            for (int o = 0; o < OPNET_SLOTS_; ++o) {
                float acc = 0.f;
#pragma unroll
                for (int f = 0; f < OPNET_FEATS_; ++f) {
                    const int k = o * OPNET_FEATS_ + f;
                    acc = fmaf(xs[((k >> 2) * 32 + clip) * 4 + (k & 3)], dx[f], acc);
                }
                dp[o] = acc;
                dot = fmaf(p[o], acc, dot);
            }
            float dl[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                dl[o] = o < OPNET_SLOTS_ ? p[o] * (dp[o] - dot) : 0.f;
                dl_s[clip][o] = dl[o];
            }
            if (bx == nc2) {
                float4 *dst = a.dlall + ((long)t * a.RB + rb) * 128 + clip;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q * 32] = make_float4(dl[4 * q], dl[4 * q + 1], dl[4 * q + 2], dl[4 * q + 3]);
            }
        }
        __syncthreads();
        const int u = (bx - nc2) * 8 + (tid >> 5);
        const long e = ((long)rb * H1 + u) * 32 + clip;
        // upstream: object_to_track_prediction (learned_models.py:40): dh += W_sel^T dl_t
        float dh = 0.f;
#pragma unroll
        for (int o = 0; o < OPNET_SLOTS_; ++o) dh = fmaf(a.wsel[o * H1 + u], dl_s[clip][o], dh);
        float dcc = 0.f;
        if (t < T - 1) {
            const long ps = (long)a.RB * H1 * 32;
            dh += ((a.dhpart1[e] + a.dhpart1[ps + e]) + a.dhpart1[2 * ps + e]) + a.dhpart1[3 * ps + e];
            dcc = a.dc1[e];
        }
        const long ge = (((long)t * a.RB + rb) * H1 + u) * 32 + clip;
        const float c_t = a.c1all[(((long)(t + 1)) * a.RB + rb) * H1 * 32 + (long)u * 32 + clip];
        const float c_p = a.c1all[((long)t * a.RB + rb) * H1 * 32 + (long)u * 32 + clip];
        float dco;
        a.g1[ge] = cell_backward(dh, dcc, a.g1[ge], c_t, c_p, &dco);
        a.dc1[e] = dco;
    }
}

// ------------------------------------------------------------------------------------------------
// backward step, fused: ONE launch per reverse time step (replaces the opnet_bwd_cell / opnet_bwd_gemm pair)
// ------------------------------------------------------------------------------------------------
// The pair existed because dh_{t-1} = W_hh^T da_t contracts over 4H gate columns: split-K over 4 workgroups kept a
// workgroup's MFMA chain as short as the forward's, and the partials met in the next launch.  Here a workgroup owns
// COMPLETE dh rows instead - 16 units x 16 clips (one clip half), K = 4H walked by its 4 waves in double-buffered
// register chunks with two accumulator chains each - so the cell backward of those (unit, clip) pairs runs in the
// same launch's epilogue and the reverse step costs one kernel boundary, not two.
//   launch n:  LSTM2 cell at t = T-1-n          (recurrent product on da2_{t+1}, written by launch n-1)
//              W_ih2^T da2 + selection-head backward at t = T-n   (da2_t from launch n-1)  -> dl_t
//              LSTM1 cell at t = T+1-n          (product on da1_{t+1}; dl_t from launch n-1)
// grid.x = 2 * (H2/16 + H1/16 + 1) workgroups (tile x clip half), grid.y <= RB.  Deterministic (fixed-order sums).
#define FUSED_NW 8          // waves per workgroup: K = 4H split 8 ways
#define FUSED_CH 16         // hexadecets a wave fetches up front (H2 = 512: its whole slice, one round trip)
#define FUSED_THREADS (64 * FUSED_NW)

// NCH chunks of 4 hexadecets, two register stages: chunk c+1 is in flight while chunk c is multiplied.  Everything is
// unconditional and fully unrolled, so the compiler's waits are counted (vmcnt(8) before a chunk's first MFMA).
// Hexadecet q0 + j feeds accumulator chain j & 1, in order - the same sums as the generic path.
#ifndef FUSED_PC
#define FUSED_PC 2      // hexadecets per pipeline chunk
#endif
#ifndef FUSED_ST
#define FUSED_ST 2      // register stages (chunks in flight + the one being multiplied)
#endif
template <int NCH, int PC, int ST>
__device__ __forceinline__ void fused_pipelined(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rs, int lane, int boff,
                                                int q0, f32x4 &acc0, f32x4 &acc1)
{
    float4 fa[ST][PC], fb[ST][PC];
#pragma unroll
    for (int c = 0; c < ST - 1 && c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < PC; ++j) {
            fa[c][j] = frag_load(ra, lane * 16, (q0 + PC * c + j) * 1024);
            fb[c][j] = frag_load(rs, boff, (q0 + PC * c + j) * 2048);
        }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + ST - 1 < NCH) {
#pragma unroll
            for (int j = 0; j < PC; ++j) {
                fa[(c + ST - 1) % ST][j] = frag_load(ra, lane * 16, (q0 + PC * (c + ST - 1) + j) * 1024);
                fb[(c + ST - 1) % ST][j] = frag_load(rs, boff, (q0 + PC * (c + ST - 1) + j) * 2048);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the later chunks in flight: the scheduler otherwise sinks loads to their uses
#pragma unroll
        for (int j = 0; j < PC; j += 2) {
            const float4 a0 = fa[c % ST][j], b0 = fb[c % ST][j], a1 = fa[c % ST][j + 1], b1 = fb[c % ST][j + 1];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
        }
    }
}

// D[16 rows x 16 clips] = A[16 x 16 nq] . da^T over hexadecets [0, nq) split over the FUSED_NW waves - every wave
// issues ALL its fragment loads before its first MFMA and keeps two accumulator chains - summed in fixed wave order
// through `part` ([FUSED_NW][4][64] floats); returns this thread's element (tid < 256): row tid >> 4, clip tid & 15
__device__ __forceinline__ float fused_product(const float4 *__restrict__ A, const float4 *__restrict__ seg, int nq,
                                               int hf, float *__restrict__ part)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = (w * nq) / FUSED_NW, q1 = ((w + 1) * nq) / FUSED_NW;
    const int boff = ((lane >> 4) * 32 + (lane & 15) + 16 * hf) * 16;
    const __amdgpu_buffer_rsrc_t ra = frag_rsrc(A), rs = frag_rsrc(seg);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // Common sizes (a wave's slice = 16 or 8 hexadecets: H = 512 / 256): a two-stage pipeline of 4-hexadecet chunks, fully
    // unrolled with unconditional loads so that the compiler waits for exactly the chunk it is about to multiply while the
    // next one streams in.  This workgroup's 256 KB of fragments take as long to cross its CU's L1 (64 B/clk) as its 512
    // MFMAs take to issue; fetching everything first and multiplying afterwards (the generic path below) serialises the two.
    const int nslice = q1 - q0;
    if (nslice == 16) fused_pipelined<16 / FUSED_PC, FUSED_PC, FUSED_ST>(ra, rs, lane, boff, q0, acc0, acc1);
    else if (nslice == 8) fused_pipelined<8 / FUSED_PC, FUSED_PC, FUSED_ST>(ra, rs, lane, boff, q0, acc0, acc1);
    else
    for (int qb = q0; qb < q1; qb += FUSED_CH) {
        float4 fa[FUSED_CH], fb[FUSED_CH];
#pragma unroll
        for (int j = 0; j < FUSED_CH; ++j)
            if (qb + j < q1) {
                fa[j] = frag_load(ra, lane * 16, (qb + j) * 1024);
                fb[j] = frag_load(rs, boff, (qb + j) * 2048);
            }
#pragma unroll
        for (int j = 0; j < FUSED_CH; j += 2) {
            if (qb + j < q1) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j].x, fb[j].x, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j].y, fb[j].y, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j].z, fb[j].z, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j].w, fb[j].w, acc0, 0, 0, 0);
            }
            if (qb + j + 1 < q1) {
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j + 1].x, fb[j + 1].x, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j + 1].y, fb[j + 1].y, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j + 1].z, fb[j + 1].z, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j + 1].w, fb[j + 1].w, acc1, 0, 0, 0);
            }
        }
    }
    float *p = part + (w * 4) * 64 + lane;
    p[0 * 64] = acc0[0] + acc1[0]; p[1 * 64] = acc0[1] + acc1[1];
    p[2 * 64] = acc0[2] + acc1[2]; p[3 * 64] = acc0[3] + acc1[3];
    __syncthreads();
    // D element (row, clip) sits in lane clip + 16 (row >> 2), register row & 3
    const int t256 = tid & 255;
    const int row = t256 >> 4, cl = t256 & 15;
    const int src = (row & 3) * 64 + cl + 16 * (row >> 2);
    float sum = part[src];
#pragma unroll
    for (int k = 1; k < FUSED_NW; ++k) sum += part[k * 256 + src];
    return sum;
}

__global__ void __launch_bounds__(FUSED_THREADS) opnet_bwd_fused(const BwdArgs a, const int n)
{
    __shared__ __attribute__((aligned(16))) float part[FUSED_NW * 4 * 64];
    __shared__ float dxs[OPNET_FEATS_][16];
    __shared__ float dps[16][16], pss[16][16], dots[16];
    const int T = a.T, H1 = a.H1, H2 = a.H2;
    const int n2 = 2 * (H2 >> 4), n1 = 2 * (H1 >> 4);
    const int bx = blockIdx.x, tid = threadIdx.x;
    const int row = (tid & 255) >> 4, cl = tid & 15;
    const bool owner = tid < 256;          // one thread per (unit, clip) of the 16 x 16 tile does the cell backward
    if (bx < n2) {
        // ---------------- LSTM2 at t = T-1-n ----------------
        const int t = T - 1 - n;
        if (t < 0 || a.mlp) return;
        const int tile = bx >> 1, hf = bx & 1;
        const int u = tile * 16 + row, clip = hf * 16 + cl;
        for (int rb = a.rb0 + blockIdx.y; rb < a.rb1; rb += gridDim.y) {
            // the cell backward's operands are fetched before the product so their latency hides under it
            const long e = ((long)rb * H2 + u) * 32 + clip;
            const long ge = (((long)t * a.RB + rb) * H2 + u) * 32 + clip;
            float4 dy, gs;
            float wo0, wo1, wo2, wo3, dcc = 0.f, c_t, c_p;
            if (owner) {
                dy = a.dyp[((long)t * a.RB + rb) * 32 + clip];
                wo0 = a.wout[u]; wo1 = a.wout[H2 + u]; wo2 = a.wout[2 * H2 + u]; wo3 = a.wout[3 * H2 + u];
                if (t < T - 1) dcc = a.dc2[e];
                gs = a.g2[ge];
                c_t = a.c2all[(((long)(t + 1)) * a.RB + rb) * H2 * 32 + (long)u * 32 + clip];
                c_p = a.c2all[((long)t * a.RB + rb) * H2 * 32 + (long)u * 32 + clip];
            }
            float rec = 0.f;
            if (t < T - 1)
                rec = fused_product(a.w2bt + (long)tile * (H2 >> 2) * 64, a.g2 + (((long)(t + 1)) * a.RB + rb) * H2 * 32,
                                    H2 >> 2, hf, part);
            if (owner) {
                // upstream: prediction_layer (learned_models.py:47): dh += W_out^T dy_t
                float dh = wo0 * dy.x;
                dh = fmaf(wo1, dy.y, dh);
                dh = fmaf(wo2, dy.z, dh);
                dh = fmaf(wo3, dy.w, dh);
                dh += rec;
                float dco;
                a.g2[ge] = cell_backward(dh, dcc, gs, c_t, c_p, &dco);
                a.dc2[e] = dco;
            }
            if (rb + (int)gridDim.y < a.rb1) __syncthreads();
        }
    } else if (bx < n2 + 2) {
        // ---------------- d frames_boxes = W_ih2^T da2_t, then einsum / softmax backward, t = T-n ----------------
        const int t = T - n;
        if (t < 0 || t >= T) return;
        const int hf = bx - n2;
        // thread (slot o = tid >> 4, clip c = tid & 15) of the first 256 owns one logit gradient; its operands are
        // fetched before the product (the serial 16-thread version of this epilogue was the tail of the launch)
        const int o = (tid >> 4) & 15, c = hf * 16 + cl;
        for (int rb = a.rb0 + blockIdx.y; rb < a.rb1; rb += gridDim.y) {
            float xv[OPNET_FEATS_], pv = 0.f;
            if (owner && o < OPNET_SLOTS_) {
                const float *xs = (const float *)(a.xp + ((long)t * a.RB + rb) * (OPNET_KXQ * 32));
                pv = ((const float *)(a.psave + ((long)t * a.RB + rb) * 128))[((o >> 2) * 32 + c) * 4 + (o & 3)];
#pragma unroll
                for (int f = 0; f < OPNET_FEATS_; ++f) {
                    const int k = o * OPNET_FEATS_ + f;
                    xv[f] = xs[((k >> 2) * 32 + c) * 4 + (k & 3)];
                }
            }
            const float dx = fused_product(a.wih2t, a.g2 + ((long)t * a.RB + rb) * H2 * 32, H2 >> 2, hf, part);
            if (owner && row < OPNET_FEATS_) dxs[row][cl] = dx;
            __syncthreads();
            // einsum backward: dp[o] = sum_f boxes[o][f] dx[f]; softmax backward: dl = p * (dp - <p, dp>)
            float dp = 0.f;
            if (owner && o < OPNET_SLOTS_) {
#pragma unroll
                for (int f = 0; f < OPNET_FEATS_; ++f) dp = fmaf(xv[f], dxs[f][cl], dp);
            }
            if (owner) { dps[o][cl] = dp; pss[o][cl] = pv; }
            __syncthreads();
            if (tid < 16) {
                float dot = 0.f;
#pragma unroll
                for (int q = 0; q < OPNET_SLOTS_; ++q) dot = fmaf(pss[q][tid], dps[q][tid], dot);
                dots[tid] = dot;
            }
            __syncthreads();
            if (owner) {
                float *dst = (float *)(a.dlall + ((long)t * a.RB + rb) * 128);
                dst[((o >> 2) * 32 + c) * 4 + (o & 3)] = o < OPNET_SLOTS_ ? pv * (dp - dots[cl]) : 0.f;
            }
            if (rb + (int)gridDim.y < a.rb1) __syncthreads();
        }
    } else {
        // ---------------- LSTM1 at t = T+1-n ----------------
        const int t = T + 1 - n;
        if (t < 0 || t >= T) return;
        const int b1 = bx - n2 - 2;
        const int tile = b1 >> 1, hf = b1 & 1;
        const int u = tile * 16 + row, clip = hf * 16 + cl;
        for (int rb = a.rb0 + blockIdx.y; rb < a.rb1; rb += gridDim.y) {
            const long e = ((long)rb * H1 + u) * 32 + clip;
            const long ge = (((long)t * a.RB + rb) * H1 + u) * 32 + clip;
            float4 dlv[4], gs;
            float ws[16], dcc = 0.f, c_t, c_p;
            if (owner) {
                const float4 *dlp = a.dlall + ((long)t * a.RB + rb) * 128 + clip;
#pragma unroll
                for (int q = 0; q < 4; ++q) dlv[q] = dlp[q * 32];
#pragma unroll
                for (int o = 0; o < OPNET_SLOTS_; ++o) ws[o] = a.wsel[o * H1 + u];
                if (t < T - 1) dcc = a.dc1[e];
                gs = a.g1[ge];
                c_t = a.c1all[(((long)(t + 1)) * a.RB + rb) * H1 * 32 + (long)u * 32 + clip];
                c_p = a.c1all[((long)t * a.RB + rb) * H1 * 32 + (long)u * 32 + clip];
            }
            float rec = 0.f;
            if (t < T - 1)
                rec = fused_product(a.w1bt + (long)tile * (H1 >> 2) * 64, a.g1 + (((long)(t + 1)) * a.RB + rb) * H1 * 32,
                                    H1 >> 2, hf, part);
            if (owner) {
                // upstream: object_to_track_prediction (learned_models.py:40): dh += W_sel^T dl_t
                float dh = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = dlv[q];
                    dh = fmaf(ws[4 * q], v.x, dh);
                    dh = fmaf(ws[4 * q + 1], v.y, dh);
                    dh = fmaf(ws[4 * q + 2], v.z, dh);
                    if (4 * q + 3 < OPNET_SLOTS_) dh = fmaf(ws[4 * q + 3], v.w, dh);
                }
                dh += rec;
                float dco;
                a.g1[ge] = cell_backward(dh, dcc, gs, c_t, c_p, &dco);
                a.dc1[e] = dco;
            }
            if (rb + (int)gridDim.y < a.rb1) __syncthreads();
        }
    }
}

// OPNetLstmMlp (learned_models.py:83-84): hidden = relu(hidden_layer(frames_boxes)), y = prediction_layer(hidden).
// d hidden = [hidden > 0] * W_out^T dy for every (t, clip, unit) at once (no recurrence), stored as (d hidden, 0,0,0)
// in the gate-gradient layout so the W_ih2^T product, the head backward and the weight-gradient GEMMs of the
// OPNet path apply unchanged (the packed "W_ih2" carries hidden_layer.weight in its gate-0 rows).
__global__ void __launch_bounds__(256) opnet_mlp_dhid(const BwdArgs a)
{
    const int H2 = a.H2;
    const long n = (long)a.T * a.RB * H2 * 32;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int clip = idx & 31;
        long r = idx >> 5;
        const int u = r % H2; r /= H2;
        const int rb = r % a.RB;
        const int t = r / a.RB;
        const float4 dy = a.dyp[((long)t * a.RB + rb) * 32 + clip];
        float dh = a.wout[u] * dy.x;
        dh = fmaf(a.wout[H2 + u], dy.y, dh);
        dh = fmaf(a.wout[2 * H2 + u], dy.z, dh);
        dh = fmaf(a.wout[3 * H2 + u], dy.w, dh);
        // hidden_t sits in the h2 history slot t+1, kq-major: [(u/4)][clip][u%4]
        const float hid = ((const float *)(a.h2all + (((long)(t + 1)) * a.RB + rb) * ((long)H2 * 8)))[((long)(u >> 2) * 32 + clip) * 4 + (u & 3)];
        a.g2[idx] = make_float4(hid > 0.f ? dh : 0.f, 0.f, 0.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradients: dW[m][n] = sum_{t, rb, clip} P[t][rb][m/4][clip][m%4] * Q[t][rb][n/4][clip][n%4]
// ------------------------------------------------------------------------------------------------
// One workgroup = one 64x64 output tile; its 4 waves split the time range and are reduced through LDS
// in fixed order.  v_mfma_f32_16x16x4_f32 contracts over 4 clips: lane (i = l&15, kc = l>>4) loads ONE
// float4 of P (m-quad mq0+i, clip 4cg+kc) and one of Q, and their 4x4 element pairs feed 16
// accumulators (accumulator (e, e') covers rows 4(mq0+i')+e, columns 4(nq0+j)+e').
// rowmode 0: output row = m;  1: m = 4*unit + gate -> torch LSTM row gate*H + unit.
struct WgradArgs {
    const float4 *P; long p_stride; int MQ;   // stride per (t, rb) in float4, number of valid m-quads
    const float4 *Q; long q_stride; int NQ;
    float *out; int ld; int mvalid; int nvalid; int rowmode; int H;
    int T, RB;
    int tiles_m, tile_begin;   // tile grid of this job inside the merged launch
};

// All six weight gradients run as ONE launch (396 workgroups at H1=256/H2=512) so the small GEMMs
// fill the CUs the big one leaves idle; blockIdx.x -> (job, tile) through tile_begin.
#define OPNET_WGRAD_JOBS 6
struct WgradBatch {
    WgradArgs job[OPNET_WGRAD_JOBS];
    const unsigned *abort;     // status word of the persistent recurrences that wrote the histories (null: launch chain);
                               // nonzero = they gave up, the histories are partial -> every dW is NaN, never a plausible number
};

__global__ void __launch_bounds__(256) opnet_wgrad(const WgradBatch batch)
{
    int j = 0;
#pragma unroll
    for (int k = 1; k < OPNET_WGRAD_JOBS; ++k)
        if ((int)blockIdx.x >= batch.job[k].tile_begin) j = k;
    const WgradArgs &g = batch.job[j];
    const int tile = blockIdx.x - g.tile_begin;
    const int tile_x = tile % g.tiles_m, tile_y = tile / g.tiles_m;
    __shared__ float red[4][64][64];  // [wave][acc*4 + r][lane]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mq0 = tile_x * 16, nq0 = tile_y * 16;
    const int i = lane & 15, kc = lane >> 4;
    const bool pa = (mq0 + i) < g.MQ, pb = (nq0 + i) < g.NQ;
    f32x4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int t0 = (w * g.T) / 4, t1 = ((w + 1) * g.T) / 4;
    // the (t, row block) steps of this wave, software-pipelined: the fragments of step it+1 are in flight while the 128
    // MFMAs of step it run (one wave per SIMD has nobody else to hide the fetch behind)
    const long nit = (long)(t1 - t0) * g.RB;
    const float4 *Pb = g.P + (long)t0 * g.RB * g.p_stride + (long)(mq0 + i) * 32 + kc;
    const float4 *Qb = g.Q + (long)t0 * g.RB * g.q_stride + (long)(nq0 + i) * 32 + kc;
    float4 av[8], bv[8], an[8], bn[8];
    if (nit > 0) {
#pragma unroll
        for (int cg = 0; cg < 8; ++cg) {
            av[cg] = pa ? Pb[cg * 4] : make_float4(0.f, 0.f, 0.f, 0.f);
            bv[cg] = pb ? Qb[cg * 4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (long it = 0; it < nit; ++it) {
        if (it + 1 < nit) {
            const float4 *Pp = Pb + (it + 1) * g.p_stride, *Qp = Qb + (it + 1) * g.q_stride;
#pragma unroll
            for (int cg = 0; cg < 8; ++cg) {
                an[cg] = pa ? Pp[cg * 4] : make_float4(0.f, 0.f, 0.f, 0.f);
                bn[cg] = pb ? Qp[cg * 4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int cg = 0; cg < 8; ++cg) {
            const float ae[4] = {av[cg].x, av[cg].y, av[cg].z, av[cg].w};
            const float be[4] = {bv[cg].x, bv[cg].y, bv[cg].z, bv[cg].w};
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[x], be[y], acc[x][y], 0, 0, 0);
        }
#pragma unroll
        for (int cg = 0; cg < 8; ++cg) { av[cg] = an[cg]; bv[cg] = bn[cg]; }
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w][(x * 4 + y) * 4 + r][lane] = acc[x][y][r];
    __syncthreads();
    // 64 x 64 outputs, 16 per thread: element id = (a = x*4+y, r, lane)
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int l = idx & 63, ar = idx >> 6;
        const int r = ar & 3, y = (ar >> 2) & 3, x = ar >> 4;
        float v = ((red[0][ar][l] + red[1][ar][l]) + red[2][ar][l]) + red[3][ar][l];
        if (batch.abort && *batch.abort != 0u) v = NAN;
        // D layout of 16x16x4: lane l holds column j = l&15, rows 4*(l>>4) + r
        const int m = 4 * (mq0 + 4 * (l >> 4) + r) + x;
        const int nn = 4 * (nq0 + (l & 15)) + y;
        if (m < g.mvalid && nn < g.nvalid) {
            const int row = g.rowmode == 1 ? (m & 3) * g.H + (m >> 2) : m;
            g.out[(long)row * g.ld + nn] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradients, second form (round 3): one WAVE per (output tile, time slice), 128 x 128 tiles for the big products
// ------------------------------------------------------------------------------------------------
// opnet_wgrad above gives a workgroup a 64 x 64 tile and lets its four waves split the time range: every wave then loads 16 float4 per
// 128 MFMAs, nothing is shared between waves, 1.9 GB come out of the L2 / Infinity Cache per launch and the matrix pipe is 52 % busy
// (profiles/r3_mfma_util_train.json).  Here a wave owns a 128 x 128 tile (32 m-quads x 32 n-quads: 64 accumulator quads = 256
// registers, in AGPRs - one wave per SIMD has 512) for ONE slice of the time range: 4 float4 per 64 MFMAs, half the operand
// traffic per flop, fetched three clip groups ahead through a ring of four.  The products whose N or M is tiny (W_ih2: N = 6,
// the two heads: M = 15 / 4) keep 64 x 64 tiles.  Every wave writes its partial tile in accumulator order (1 KB per store
// instruction) to the workspace; opnet_wgrad_reduce sums the slices of a tile in slice order (deterministic), maps (m, n) to the
// torch layout and writes the gradient.  The host sizes the slices so that all wave jobs run in ONE round of the 1 024 SIMDs and
// big (512 MFMAs per step) and small (128) jobs end together.
struct Wg2Job {
    WgradArgs g;               // operands and output as in opnet_wgrad (tiles_m / tile_begin unused)
    int big;                   // 1: 128 x 128 tiles, 0: 64 x 64
    int tiles_m, tiles_n;      // tile grid
    int slices;                // time slices per tile
    int wave_begin;            // first wave job of this product: wave job = wave_begin + (slice * tiles_m + tile_m) * tiles_n + tile_n
};
struct Wg2Batch {
    Wg2Job job[OPNET_WGRAD_JOBS];
    int njobs, nwaves;
    int ncg;                   // clip groups (of 4) per row block that can hold a clip: 8, or 1 / 2 / 4 for one ragged row block - the
                               // groups past the batch are all zeros in every operand, adding them changes no sum
    float *partial;            // [wave job][16 384 floats] (small jobs use the first 4 096)
    const unsigned *abort;
};
#define WG2_PART_F 16384

template <int GA, int GB, int NCG>
__device__ __forceinline__ void wg2_tile(const float4 *P, long p_stride, int MQ, const float4 *Q, long q_stride, int NQ, int mq0, int nq0,
                                         long it0, long it1, float *__restrict__ part)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kc = lane >> 4;
    // operands through buffer descriptors based at the slice's first step: a wave-uniform byte offset per (step, clip group), one lane
    // offset per fragment row - and a lane whose row does not exist gets an offset past the end, which reads as zero: no branches, no
    // 64-bit address arithmetic in the loop
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(P + it0 * p_stride), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)(Q + it0 * q_stride), 0, 0x7fffffff, 0x00020000);
    unsigned va[GA], vb[GB];
#pragma unroll
    for (int a = 0; a < GA; ++a) va[a] = (mq0 + 16 * a + i) < MQ ? (unsigned)(((mq0 + 16 * a + i) * 32 + kc) * 16) : 0x80000000u;
#pragma unroll
    for (int b = 0; b < GB; ++b) vb[b] = (nq0 + 16 * b + i) < NQ ? (unsigned)(((nq0 + 16 * b + i) * 32 + kc) * 16) : 0x80000000u;
    const unsigned ps = (unsigned)(p_stride * 16), qs = (unsigned)(q_stride * 16);
    const int nst = (int)(it1 - it0);
    // the accumulators are pinned to AGPRs (constraint "a"): left to the register allocator, a part of the 256 lives in VGPRs and is
    // copied in and out around the loop body - 1.5 v_accvgpr moves per MFMA
    f32x4 acc[GA][GB][4][4];
#pragma unroll
    for (int a = 0; a < GA; ++a)
#pragma unroll
        for (int b = 0; b < GB; ++b)
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    acc[a][b][x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    asm volatile("" : "+a"(acc[a][b][x][y]));
                }
    // ring of four clip groups, three ahead.  The (step, clip group) items of the slice are walked in spans of U = max(NCG, 4) items, so
    // that slot = item & 3 and the clip group = item % NCG are static indices.  Past the slice's last step the ring re-reads that
    // step (never used).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[4][GA], rb[4][GB];
    auto fetch = [&](int st, int cg, int slot) {
        const unsigned so = (unsigned)(st < nst ? st : nst - 1);
#pragma unroll
        for (int a = 0; a < GA; ++a) ra[slot][a] = __builtin_amdgcn_raw_buffer_load_b128(rp, va[a], so * ps + cg * 64, 0);
#pragma unroll
        for (int b = 0; b < GB; ++b) rb[slot][b] = __builtin_amdgcn_raw_buffer_load_b128(rq, vb[b], so * qs + cg * 64, 0);
    };
    constexpr int U = NCG < 4 ? 4 : NCG, SPU = U / NCG;          // items and steps per span
    const int total = nst * NCG;
    if (nst > 0) { fetch(0 / NCG, 0 % NCG, 0); fetch(1 / NCG, 1 % NCG, 1); fetch(2 / NCG, 2 % NCG, 2); }
    for (int s0 = 0, st0 = 0; s0 < total; s0 += U, st0 += SPU) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NCG < 4 && s0 + u >= total) break;               // (a ragged last span: only when a step is less than a span)
            fetch(st0 + (u + 3) / NCG, (u + 3) % NCG, (u + 3) & 3);
            const int sl = u & 3;
#pragma unroll
            for (int a = 0; a < GA; ++a) {
                const float ae[4] = {__uint_as_float(ra[sl][a][0]), __uint_as_float(ra[sl][a][1]), __uint_as_float(ra[sl][a][2]), __uint_as_float(ra[sl][a][3])};
#pragma unroll
                for (int b = 0; b < GB; ++b) {
                    const float be[4] = {__uint_as_float(rb[sl][b][0]), __uint_as_float(rb[sl][b][1]), __uint_as_float(rb[sl][b][2]), __uint_as_float(rb[sl][b][3])};
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 4; ++y)
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[a][b][x][y]) : "v"(ae[x]), "v"(be[y]));
                }
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");     // the last MFMAs' results (8 passes: 11 wait states) before they are read
    float4 *out = (float4 *)part + lane;
#pragma unroll
    for (int a = 0; a < GA; ++a)
#pragma unroll
        for (int b = 0; b < GB; ++b)
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    const f32x4 v = acc[a][b][x][y];
                    out[(((a * GB + b) * 4 + x) * 4 + y) * 64] = make_float4(v[0], v[1], v[2], v[3]);
                }
}

__global__ void __launch_bounds__(256, 1) opnet_wgrad_tiles(const Wg2Batch batch)
{
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wj = blockIdx.x * 4 + w;
    if (wj >= batch.nwaves) return;
    int j = 0;
#pragma unroll
    for (int k = 1; k < OPNET_WGRAD_JOBS; ++k)
        if (k < batch.njobs && wj >= batch.job[k].wave_begin) j = k;
    const Wg2Job &J = batch.job[j];
    const int r = wj - J.wave_begin;
    const int tn = r % J.tiles_n, tm = (r / J.tiles_n) % J.tiles_m, sl = r / (J.tiles_n * J.tiles_m);
    const long nit = (long)J.g.T * J.g.RB;
    const long it0 = (sl * nit) / J.slices, it1 = ((sl + 1) * nit) / J.slices;
    float *part = batch.partial + (long)wj * WG2_PART_F;
    // (the job's fields by value: through the reference every use is a scalar load from the kernarg segment and a wait)
    const float4 *P = J.g.P, *Q = J.g.Q;
    const long p_stride = J.g.p_stride, q_stride = J.g.q_stride;
    const int MQ = J.g.MQ, NQ = J.g.NQ;
    const int ncg = __builtin_amdgcn_readfirstlane(batch.ncg);
#define WG2_RUN(NCG) do { if (J.big) wg2_tile<2, 2, NCG>(P, p_stride, MQ, Q, q_stride, NQ, tm * 32, tn * 32, it0, it1, part); \
                          else wg2_tile<1, 1, NCG>(P, p_stride, MQ, Q, q_stride, NQ, tm * 16, tn * 16, it0, it1, part); } while (0)
    if (ncg == 8) WG2_RUN(8);
    else if (ncg == 4) WG2_RUN(4);
    else if (ncg == 2) WG2_RUN(2);
    else WG2_RUN(1);
#undef WG2_RUN
}

// one thread per accumulator quad (tile, a, b, x, y, lane): the slices of the tile in slice order, then the four rows of the quad
__global__ void __launch_bounds__(256) opnet_wgrad_reduce(const Wg2Batch batch)
{
    const bool bad = batch.abort && *batch.abort != 0u;
    for (int j = 0; j < batch.njobs; ++j) {
        const Wg2Job &J = batch.job[j];
        const WgradArgs &g = J.g;
        const int G = J.big ? 2 : 1;
        const long quads = (long)G * G * 16 * 64;                  // float4 per tile
        const long ntile = (long)J.tiles_m * J.tiles_n;
        for (long idx = blockIdx.x * 256L + threadIdx.x; idx < ntile * quads; idx += (long)gridDim.x * 256) {
            const long tile = idx / quads;
            const int q = (int)(idx - tile * quads);
            const int l = q & 63, y = (q >> 6) & 3, x = (q >> 8) & 3, ab = q >> 10;
            const int b = ab % G, a = ab / G;
            const int tn = (int)(tile % J.tiles_n), tm = (int)(tile / J.tiles_n);
            const float4 *p = (const float4 *)(batch.partial + (long)(J.wave_begin + tile) * WG2_PART_F) + q;
            // slices in slice order, eight loads in flight at a time (one after the other each costs a full memory round trip)
            const long sstride = ntile * (WG2_PART_F / 4);
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s0 = 0; s0 < J.slices; s0 += 8) {
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = s0 + k < J.slices ? p[(long)(s0 + k) * sstride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (s0 + k < J.slices) { sum.x += v[k].x; sum.y += v[k].y; sum.z += v[k].z; sum.w += v[k].w; }
            }
            if (bad) sum = make_float4(NAN, NAN, NAN, NAN);
            // D layout of 16x16x4: lane l holds column j = l & 15, rows 4 (l >> 4) + r
            const int nn = 4 * (tn * 16 * G + 16 * b + (l & 15)) + y;
            const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 4 * (tm * 16 * G + 16 * a + 4 * (l >> 4) + r) + x;
                if (m < g.mvalid && nn < g.nvalid) {
                    const int row = g.rowmode == 1 ? (m & 3) * g.H + (m >> 2) : m;
                    g.out[(long)row * g.ld + nn] = sv[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// loss and optimiser
// ------------------------------------------------------------------------------------------------
// mean(|y - label|) (nn.L1Loss(reduction="none") then torch.mean, training_main.py:152,192,204) and
// its gradient sign(y - label) / n.  Two-stage deterministic reduction: per-block partials, then
// block 0... a second tiny launch sums them in fixed order.
// beta <= 0: L1.  beta > 0: SmoothL1 (torch.nn.SmoothL1Loss, mean reduction): |d| < beta -> 0.5 d^2 / beta, else
// |d| - 0.5 beta - the loss BASELINE.json's config text names; the reference itself trains with L1.
__global__ void __launch_bounds__(256) opnet_l1_partial(const float *__restrict__ y, const float *__restrict__ lab,
                                                        float *__restrict__ dy, float *__restrict__ partial, long n,
                                                        float beta)
{
    __shared__ float red[256];
    const float inv = 1.0f / (float)n;
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = y[i] - lab[i];
        const float ad = fabsf(d);
        if (beta > 0.f && ad < beta) {
            s += 0.5f * d * d / beta;
            if (dy) dy[i] = d / beta * inv;
        } else {
            s += beta > 0.f ? ad - 0.5f * beta : ad;
            if (dy) dy[i] = d > 0.f ? inv : (d < 0.f ? -inv : 0.f);   // torch: sign(0) = 0
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// (one wave: lane l adds partials l, l + 64, ... in double, then a fixed xor tree over the lanes - a single thread walking up to 1 024
// dependent loads took 8 us of a 1.85-ms training step)
__global__ void opnet_l1_final(const float *__restrict__ partial, int nblocks, float *__restrict__ loss, long n)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) s += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) *loss = (float)(s / (double)n);
}

// torch.optim.Adam.step (training_main.py:150,217): defaults betas (0.9, 0.999), eps 1e-8, no weight
// decay, no amsgrad:  m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
//                     p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
__global__ void __launch_bounds__(256) opnet_adam(float *__restrict__ p, const float *__restrict__ gr,
                                                  float *__restrict__ m, float *__restrict__ v, long n,
                                                  float b1, float b2, float eps, float step_size,
                                                  float inv_sqrt_bc2, float grad_scale)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float g = gr[i] * grad_scale;
        const float mi = b1 * m[i] + (1.0f - b1) * g;
        const float vi = b2 * v[i] + (1.0f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}

// the same update for up to OPNET_ADAM_MAX tensors in ONE launch (a model's six weights: six launches of ~5 us each otherwise)
#define OPNET_ADAM_MAX 16
struct AdamBatch {
    float *p[OPNET_ADAM_MAX];
    const float *g[OPNET_ADAM_MAX];
    float *m[OPNET_ADAM_MAX], *v[OPNET_ADAM_MAX];
    long n[OPNET_ADAM_MAX];
    int count;
    // guards (each may be null): the update is SKIPPED - parameters and moments untouched - when the abort word of the
    // persistent launches that made the gradients is nonzero, when the loss is not finite, or when the data-parallel guard
    // (two floats behind the gradient bucket, summed over the ranks by the gradient all-reduce: [0] ranks whose persistent
    // launches gave up, [1] ranks whose loss is not finite - opnet_dp_guard) is nonzero
    const unsigned *abort_u32;
    const float *loss_f32;
    const float *guard_f32;
};

// this rank's contribution to the data-parallel guard (4 floats behind the flat gradient bucket, all-reduced WITH it):
// [0] = 1 when the abort word of this rank's persistent launches is raised, [1] = 1 when this rank's loss is not finite,
// [2] = this rank's share of the global mean loss (loss * n_local / n_global).  After the sum every rank holds the same two
// counts, so every rank's guarded Adam takes the same decision, and the loss of the WHOLE minibatch (what the reference
// prints, training_main.py:212) without a collective of its own.
__global__ void opnet_dp_guard(float *__restrict__ guard, const unsigned *abort_u32, const float *loss_f32, float loss_weight)
{
    if (threadIdx.x == 0) {
        guard[0] = (abort_u32 && *abort_u32 != 0u) ? 1.f : 0.f;
        guard[1] = (loss_f32 && !isfinite(*loss_f32)) ? 1.f : 0.f;
        guard[2] = loss_f32 ? *loss_f32 * loss_weight : 0.f;
        guard[3] = 0.f;
    }
}

// grid (blocks per tensor, tensors)
__global__ void __launch_bounds__(256) opnet_adam_multi(const AdamBatch t, float b1, float b2, float eps, float step_size,
                                                        float inv_sqrt_bc2, float grad_scale)
{
    if (t.abort_u32 && *t.abort_u32 != 0u) return;
    if (t.loss_f32 && !isfinite(*t.loss_f32)) return;
    if (t.guard_f32 && (t.guard_f32[0] != 0.f || t.guard_f32[1] != 0.f)) return;
    const int k = blockIdx.y;
    float *__restrict__ p = t.p[k];
    const float *__restrict__ gr = t.g[k];
    float *__restrict__ m = t.m[k], *__restrict__ v = t.v[k];
    const long n = t.n[k];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float g = gr[i] * grad_scale;
        const float mi = b1 * m[i] + (1.0f - b1) * g;
        const float vi = b2 * v[i] + (1.0f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}
