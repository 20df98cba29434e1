// seq_xcd_kernels.hip - the stacked LSTM of the sibling reasoners as ONE persistent launch per forward.
//
// What is computed: the nn.LSTM(bias=False, batch_first) + Linear head of reference baselines/learned_models.py
//   BaselineLstm   :99-101, 110-115   (1 layer, 75 -> 512)
//   NonLinearLstm  :135-137, 146-148  (2 layers, 3840 -> 512 -> 512; the 3840-wide input product is hoisted into one GEMM)
//   TransformerLstm:170-172, 192-195  (2 layers, 256 -> 512 -> 512)
// - the same function as lstm_stack_step (seq_kernels.hip), which stays the engine for every other shape.
//
// Why: one launch per time step costs 5.7 us of dependent-launch latency per step (T + 2L - 1
// launches, the matrix pipe 6 % busy); config 3 (one clip, S = 300) is 2.0 ms of which 1.75 ms are those launches.
// The scheme of opnet_xcd4_kernels.hip carries over: groups of FOUR clips, every weight of a layer resident in the registers
// of ONE XCD (32 CUs x 4 waves) for all T steps, v_mfma_f32_4x4x1_16b_f32 (16 independent 4 x 4 outer products: block = hidden
// unit, row = gate, column = clip), h exchanged between the 32 CUs through the XCD's L2 with "the data is the flag".
//
// Decomposition.  256 workgroups of 4 waves, one per CU; XCD x = blockIdx.x & 7, CU c = blockIdx.x >> 3 owns hidden units
// 16 c .. 16 c + 15 (64 gate rows) of ITS XCD's layer.
//   L = 1: XCD x runs the layer for the groups G = x, x + 8, ...           (8 groups = 32 clips side by side)
//   L = 2: XCD 2 p runs layer 0 and XCD 2 p + 1 layer 1 of the groups G = p, p + 4, ...   (4 pairs = 16 clips side by side)
// The two layers of a pair are a PIPELINE, not a lock step: layer 1 at step t needs h0[t] and its own h1[t-1]; layer 0 needs
// nothing from layer 1, runs ahead, and the cross-XCD hop (write-through stores, sc1 loads: ~1 us) only delays layer 1's
// start.  There is no ring to overrun: every exchange buffer holds the FULL history [T + 1 slots], each word written once per
// launch (slot 0 = the zero initial state, every other word pre-armed with the sentinel 0xffffffff by seqx_init), so there is
// no re-arming, no flow control, and the top layer's buffer is what the output head reads afterwards.
//
// A phase = (group gi, step t).  The two products of the CU's 64 gate rows are K-split over its four waves (one per SIMD):
//   h side: W_hh . h[t-1]     every wave its K quarter (128 MFMAs); h gathered from the layer's own exchange buffer (XCD-local
//           stores when the placement check of opnet_xcd_kernels.hip passed, write-through otherwise); weights in VGPRs.
//   x side: W_ih . input[t]   waves 1..3 a THIRD of K each (wave 0 is the cell wave and has none); input = the packed x (layer 0,
//           K <= 256) or the lower layer's h[t] (K = 512) gathered from its cross-XCD copy; the weights sit in AGPRs (the
//           MFMA reads an AccVGPR as its A operand directly).  Hoisted layer 0: no x side - the cell adds the GEMM's output.
// Both inputs of a wave are ITS OWN slice of the k range: gathered by that wave into wave-private LDS, read back as B operands
// (one ds_read_b128 = four k of one clip, the same address in all 16 blocks) - the only workgroup barrier of a phase is the one
// between the K-split partials and the cell.  The x side does not depend on the recurrence: waves 1..3 compute it FIRST, while
// wave 0 computes the previous phase's cell and publishes h, so the serial chain of a step is
//   cell -> publish -> [exchange] -> gather -> h side (128 MFMAs) -> barrier
// on every layer (first version: x side split four ways with wave 0 doing its quarter after the cell - 2.17 us a step on the
// second layer against 1.68 us on a single layer).
// Summation order: gate = ((w0 + w1) + w2) + w3 over the waves' partials [+ hoisted pre-activation]; a wave's partial =
// ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7)) over eight interleaved ascending-k chains (k mod 8; chain = 4 * (k-quad
// parity) + k mod 4), the x part of a chain before its h part.
// Every poll is bounded (XCD_SPIN_LIMIT): an abort raises status[0], every poller leaves, seqx_out_head fills y with NaN.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opnet_ctx.h"

#define SX_H 512
#define SX_NGMAX 8             // groups per XCD (pair) one launch carries
#define SX_XT1 128             // k-quads of an upper layer's x side (K = 512: the lower layer's h)

// x side: NXT k-quads split over the last SX_XWAVES waves; x wave xi (= w - (4 - SX_XWAVES)) owns [sx_xstart(NXT, xi),
// sx_xstart(NXT, xi + 1)).  4 = every wave (the cell wave does its share between its publish and the arrival of the
// others' h); 3 = the cell wave has none (measured: the three x waves then carry 300 MFMAs a phase on the upper layer and
// the cell wave idles 2 200 cycles at the barrier: transformer_lstm one clip 0.875 ms against 0.81)
#ifndef SX_XWAVES
#define SX_XWAVES 4
#endif
__host__ __device__ constexpr int sx_xstart(int NXT, int xi) { return NXT * xi / SX_XWAVES; }
__host__ __device__ constexpr int sx_xw(int NXT) { return (NXT + SX_XWAVES - 1) / SX_XWAVES; }   // register quads an x wave holds (padded)

struct SeqXPacked { size_t ah[2], ax[2], total; int nxt[2]; };      // offsets in floats; nxt = x-side k-quads of the layer
__host__ __device__ inline SeqXPacked seqx_packed_layout(int L, int NXT0)
{
    SeqXPacked P;
    size_t o = 0;
    for (int l = 0; l < 2; ++l) {
        P.nxt[l] = l == 0 ? NXT0 : SX_XT1;
        P.ah[l] = o; if (l < L) o += (size_t)32 * 4 * 32 * 256;                     // [cu][wave][q][lane] float4
        P.ax[l] = o; if (l < L) o += (size_t)32 * SX_XWAVES * sx_xw(P.nxt[l]) * 256;   // [cu][x wave][q][lane] float4
    }
    P.total = o;
    return P;
}

struct SeqXArgs {
    int B, T, L, NGT;          // NGT = ceil(B / 4) groups of 4 clips
    int RB, KXQ;               // packed input: row blocks of 32 clips, k-quads per clip (= NXT0)
    const float *pk;           // seqx_packed_layout image
    const float *whead;        // predictions_layer.weight [4][512], the caller's row-major tensor
    char *ws;                  // workspace base; the offsets below are bytes into it (one buffer descriptor)
    unsigned xp_off;           // layer 0 input, direct:  [T][RB][KXQ][32] float4
    unsigned g_off;            // layer 0 input, hoisted: G [B * T][2048] floats, row b * T + t, column 4 * unit + gate
    unsigned hl_off[2];        // per layer: own exchange / history [NGT][T + 1][128][4] float4, slot t + 1 = step t
    unsigned hc_off[2];        // layer l < L - 1: the same data written through for the next layer's XCD
    unsigned *status;          // [0] abort code, [1] block, [2] phase, [3] groups on the write-through path, [8..] XCC ids
    float4 *ystage;            // [B][T] float4 (the caller's y)
    int force_safe, debug;     // debug (tools / tests): bit 2 = no cells (nobody publishes: forces the abort path)
    // training forward: the launch chain's history layouts (stack_train_ws_layout), null for inference
    float *hall[2];            // [T + 1][RB][128][32] float4 as floats
    float *call[2];            // [T + 1][RB][512][32]
    float4 *gsave[2];          // [T][RB][512][32]
    unsigned long long *trace; // tools: [8 XCDs][4 waves][phases][8] s_memtime stamps of CU 0 of every XCD (null = off)
};

typedef float sx_f32x4 __attribute__((ext_vector_type(4)));
// wave-private LDS written by the lanes of a wave and read back by other lanes of the SAME wave: a wave's LDS instructions are
// executed in order, so no s_waitcnt is needed between the write and the read - only the compiler must not reorder them
#ifndef SX_LDS_SYNC
#define SX_LDS_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
#ifndef SX_RING
#define SX_RING 8
#endif
#ifndef SX_AHEAD
#define SX_AHEAD 6
#endif

// one k of 64 gate rows x 4 clips; A in a VGPR / in an AccVGPR
#define SX_MFMA_V(acc, av, bv) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv))
#define SX_MFMA_A(acc, av, bv) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(av), "v"(bv))

// fp32 weights -> the register images (lane = 4 b + i: unit 16 cu + b, gate i).  h side: element e of quad q of wave w is
// k = 4 (32 w + q) + e; x side: element e of quad q of x wave xi is k = 4 (sx_xstart(NXT, xi) + q) + e, zero beyond the wave's share
__global__ void __launch_bounds__(256) seqx_pack(float *__restrict__ out, const float *__restrict__ w_ih0, const float *__restrict__ w_hh0,
                                                 const float *__restrict__ w_ih1, const float *__restrict__ w_hh1, int L, int NXT0, int KX)
{
    const SeqXPacked P = seqx_packed_layout(L, NXT0);
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < P.total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 3, lane = (idx >> 2) & 63, b = lane >> 2, i = lane & 3;
        const int l = (L == 2 && idx >= P.ah[1]) ? 1 : 0;
        const bool xs = idx >= P.ax[l];
        const size_t r = (idx - (xs ? P.ax[l] : P.ah[l])) >> 8;
        float v = 0.f;
        if (!xs) {
            const int q = r % 32, w = (r / 32) % 4, cu = r / 128;
            const int k = 4 * (32 * w + q) + e;
            v = (l == 0 ? w_hh0 : w_hh1)[((size_t)i * SX_H + 16 * cu + b) * SX_H + k];
        } else {
            const int nxt = P.nxt[l], nw = sx_xw(nxt);
            const int q = r % nw, xi = (r / nw) % SX_XWAVES, cu = r / (SX_XWAVES * nw);
            const int kq = sx_xstart(nxt, xi) + q;
            const int k = 4 * kq + e;
            const size_t rr = (size_t)i * SX_H + 16 * cu + b;
            if (kq < sx_xstart(nxt, xi + 1)) v = l == 0 ? (k < KX ? w_ih0[rr * KX + k] : 0.f) : w_ih1[rr * SX_H + k];
        }
        out[idx] = v;
    }
}

// status words, XCC sentinels; every exchange buffer: slot 0 = zeros (h_{-1}), everything else "not published yet"
__global__ void __launch_bounds__(256) seqx_init(SeqXArgs a)
{
    const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x, n = (long)gridDim.x * blockDim.x;
    if (tid < 8) a.status[tid] = 0u;
    for (long i = tid; i < 256; i += n) a.status[8 + i] = 0xffffffffu;
    const xcd_u32x4 z = {0u, 0u, 0u, 0u}, sent = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    const long per = (long)(a.T + 1) * 512, tot = (long)a.NGT * per;       // float4 per group, per buffer
    for (int l = 0; l < a.L; ++l) {
        xcd_u32x4 *hl = (xcd_u32x4 *)(a.ws + a.hl_off[l]);
        for (long i = tid; i < tot; i += n) hl[i] = (i % per) < 512 ? z : sent;
        if (l + 1 < a.L) {
            xcd_u32x4 *hc = (xcd_u32x4 *)(a.ws + a.hc_off[l]);
            for (long i = tid; i < tot; i += n) hc[i] = (i % per) < 512 ? z : sent;
        }
    }
}

__device__ __forceinline__ bool sx_unpublished(xcd_u32x4 r)
{
    return r.x == 0xffffffffu || r.y == 0xffffffffu || r.z == 0xffffffffu || r.w == 0xffffffffu;
}

// NP pieces of 1 KB at src (lane l: 16 B at src + 1024 q + 16 l) -> registers; while any lane still sees a sentinel word all of
// them are loaded again (bounded).  false = abort (wave-uniform).
template <int NP>
__device__ __forceinline__ bool sx_poll(__amdgpu_buffer_rsrc_t rws, unsigned lane16, unsigned src, xcd_u32x4 (&r)[NP],
                                        unsigned *status, int phase)
{
    long long t0 = 0;
    for (unsigned spins = 1;; ++spins) {
        bool bad = false;
#pragma unroll
        for (int q = 0; q < NP; ++q) bad |= sx_unpublished(r[q]);
        if (!__any(bad)) return true;
        if (!x4_keep_polling(spins, t0, status, phase)) return false;
#pragma unroll
        for (int q = 0; q < NP; ++q) r[q] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, src + q * 1024, 16);   // sc1
    }
}

// NQ B fragments (k-quads) out of a wave-private LDS region F (already offset by the lane's clip), four MFMAs each.  EIGHT
// accumulator chains (fragment parity x element): measured with four, a dependent v_mfma_f32_4x4x1 issues every ~52 cycles and
// the 128 MFMAs of the h side took 1 680 cycles (13 a piece) - the chains, not the pipe, set the pace.  The fragments travel
// through a ring of 4 register quads, 3 ahead (opnet_xcd4_kernels.hip: one ds_read in flight per 4 MFMAs exposes every LDS
// latency).  The fragments travel through a ring of SX_RING register quads, SX_AHEAD ahead: with 3 ahead (the 4-clip OPNet
// kernel's ring) a fragment of four MFMAs took 53 cycles = a third of the ds_read_b128 latency instead of 4 x 9.5 - the LDS
// round trip (~160 cycles), not the pipe, set the pace.
// AG: the A operands are AccVGPRs.  `mid` runs once, after fragment MIDQ (the caller's early load issue).
template <int NQ, bool AG, int MIDQ, typename AT, typename MID>
__device__ __forceinline__ void sx_products(sx_f32x4 (&c)[8], const AT &A, const float4 *F, MID mid)
{
    if (NQ == 0) { mid(); return; }
    float4 bf[SX_RING];
#pragma unroll
    for (int i = 0; i < SX_AHEAD; ++i)
        if (i < NQ) bf[i] = F[i * 4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q + SX_AHEAD < NQ) bf[(q + SX_AHEAD) % SX_RING] = F[(q + SX_AHEAD) * 4];
        __builtin_amdgcn_sched_barrier(0);
        const float4 bq = bf[q % SX_RING];
        const int o = (q & 1) * 4;
        if (AG) {
            SX_MFMA_A(c[o + 0], A[4 * q + 0], bq.x);
            SX_MFMA_A(c[o + 1], A[4 * q + 1], bq.y);
            SX_MFMA_A(c[o + 2], A[4 * q + 2], bq.z);
            SX_MFMA_A(c[o + 3], A[4 * q + 3], bq.w);
        } else {
            SX_MFMA_V(c[o + 0], A[4 * q + 0], bq.x);
            SX_MFMA_V(c[o + 1], A[4 * q + 1], bq.y);
            SX_MFMA_V(c[o + 2], A[4 * q + 2], bq.z);
            SX_MFMA_V(c[o + 3], A[4 * q + 3], bq.w);
        }
        if (q == MIDQ) mid();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (MIDQ >= NQ) mid();
}

// NXT0: k-quads of layer 0's direct input (0 = hoisted: the cell adds G); L: layers; TRAIN: keep the backward's histories
template <int NXT0, int L, bool TRAIN>
__global__ void __launch_bounds__(256) seqx_forward(const SeqXArgs a)
{
    constexpr int NXW0 = sx_xw(NXT0), NXW1 = sx_xw(SX_XT1);
    constexpr int NXWMAX = L == 2 ? NXW1 : (NXW0 > 0 ? NXW0 : 1);
    __shared__ __attribute__((aligned(1024))) float4 sH[4][128];         // wave-private: the wave's quarter of h[t-1]
    __shared__ __attribute__((aligned(1024))) float4 sX[SX_XWAVES][4 * NXWMAX];  // x-wave-private: its share of the x-side input
    __shared__ __attribute__((aligned(16))) float4 sP[2][4][64];         // K-split partials by phase parity
    __shared__ float sC[SX_NGMAX][64];
    __shared__ float4 sPad[4096];          // 64 KB never used: > 80 KB of LDS in total keep a second workgroup off the CU (every
                                           // CU must host exactly one of the 256 workgroups, or the exchange waits for a block
                                           // that is not resident)
    __shared__ int sAbort, sLocal;          // through XCD_LDS_LD / XCD_LDS_ST: ds_read / ds_write (a volatile LDS word is a FLAT access + vmcnt wait)

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = blockIdx.x & 7, c = blockIdx.x >> 3;
    constexpr int NPAIR = 8 / L;
    const int pr = x / L, l = __builtin_amdgcn_readfirstlane(x % L);
    const int T = a.T;
    const int ng = a.NGT > pr ? (a.NGT - pr + NPAIR - 1) / NPAIR : 0;
    if (ng == 0) return;
    const int b = lane >> 2, j = lane & 3;
    if (w == 0) {
        const int loc = xcd_group_is_local(a.status, x);
        if (lane == 0) {
            XCD_LDS_ST(sLocal, loc > 0 && a.force_safe == 0);
            XCD_LDS_ST(sAbort, loc < 0);
            if (loc == 0 && c == 0) atomicAdd(a.status + 3, 1u);
        }
    }
    for (int i = tid; i < SX_NGMAX * 64; i += 256) (&sC[0][0])[i] = 0.f;
    if (a.debug & 0x40000000) sPad[tid * 16] = make_float4(0.f, 0.f, 0.f, 0.f);     // (keeps the padding allocated)
    // this XCD's layer: its buffers and histories (selected here once - indexing the kernarg arrays by l costs scratch)
    const unsigned hl_mine = l == 0 ? a.hl_off[0] : a.hl_off[1];
    const unsigned hc_mine = a.hc_off[0];                        // only layer 0 of L = 2 has a reader on another XCD
    float *const hall_mine = l == 0 ? a.hall[0] : a.hall[1];
    float *const call_mine = l == 0 ? a.call[0] : a.call[1];
    float4 *const gsave_mine = l == 0 ? a.gsave[0] : a.gsave[1];
    const size_t pk_ah = l == 0 ? seqx_packed_layout(L, NXT0).ah[0] : seqx_packed_layout(L, NXT0).ah[1];
    const size_t pk_ax = l == 0 ? seqx_packed_layout(L, NXT0).ax[0] : seqx_packed_layout(L, NXT0).ax[1];
    // the x side of this wave: x wave xi = w - 1 owns k-quads [xs0, xs0 + xcnt) of the layer's NXT
    const bool upper = L == 2 && l == 1;
    const int nxt = upper ? SX_XT1 : NXT0;
    constexpr int XW0 = 4 - SX_XWAVES;                           // first x wave
    const bool xwave = w >= XW0;
    const int xi = xwave ? w - XW0 : 0;
    const int xs0 = sx_xstart(nxt, xi), xcnt = xwave ? sx_xstart(nxt, xi + 1) - xs0 : 0;

    // ---- resident weights: h side in VGPRs, x side in AccVGPRs -----------------------------------------------------------
    float ah[128];
    float ax[4 * NXWMAX];
    {
        const float4 *ph = (const float4 *)(a.pk + pk_ah) + ((size_t)(c * 4 + w) * 32) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const float4 v = ph[q * 64];
            ah[4 * q] = v.x; ah[4 * q + 1] = v.y; ah[4 * q + 2] = v.z; ah[4 * q + 3] = v.w;
        }
        const int nw = upper ? NXW1 : NXW0;
        const float4 *px = (const float4 *)(a.pk + pk_ax) + ((size_t)(c * SX_XWAVES + xi) * nw) * 64 + lane;
#pragma unroll
        for (int q = 0; q < NXWMAX; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xwave && q < nw) v = px[q * 64];
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ax[4 * q]) : "v"(v.x));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ax[4 * q + 1]) : "v"(v.y));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ax[4 * q + 2]) : "v"(v.z));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ax[4 * q + 3]) : "v"(v.w));
        }
    }

    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void *)a.ws, 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16;
    __syncthreads();
    if (XCD_LDS_LD(sAbort)) return;
    int abort_seen = 0;                         // the abort word as read behind the PREVIOUS phase's last barrier (see the loop's end)
    const bool local = __builtin_amdgcn_readfirstlane(XCD_LDS_LD(sLocal)) != 0;
    const bool top = l == L - 1;
    const int nph = T * ng;
    constexpr int NLD = (NXWMAX + 15) / 16;                      // 1-KB pieces (16 k-quads x 4 clips) an x wave loads per phase

    int gi = 0, t = 0;              // the phase this iteration computes the products of
    int gp = 0, tp = 0;             // the previous phase (whose cell wave 0 computes now)
    // the x-side input of phase (g, tt): it does not depend on the recurrence, so it is asked for one phase early (under the h
    // side's MFMAs) and only checked / asked again at its use
    xcd_u32x4 xr[NLD];
    unsigned xsrc = 0;
    auto ask_x = [&](int g, int tt) {
        const int GG = g * NPAIR + pr;
#pragma unroll
        for (int r = 0; r < NLD; ++r) xr[r] = (xcd_u32x4){0u, 0u, 0u, 0u};
        if (!xwave) return;
        if (upper) {
            xsrc = a.hc_off[0] + ((unsigned)(GG * (T + 1) + tt + 1) * 128 + xs0) * 64;
#pragma unroll
            for (int r = 0; r < NLD; ++r)
                if ((lane >> 2) + 16 * r < xcnt) xr[r] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, xsrc + r * 1024, 16);
        } else if (NXT0 > 0) {
            const int rb = (4 * GG) >> 5, cb = (4 * GG) & 31;
            const unsigned o = a.xp_off + ((unsigned)((tt * a.RB + rb) * a.KXQ) + xs0) * 512;
#pragma unroll
            for (int r = 0; r < NLD; ++r)
                if ((lane >> 2) + 16 * r < xcnt)
                    xr[r] = __builtin_amdgcn_raw_buffer_load_b128(rws, (((lane >> 2) + 16 * r) * 32 + cb + j) * 16, o, 0);
        }
    };
    ask_x(0, 0);
    unsigned long long *const tr = (a.trace && c == 0 && lane == 0) ? a.trace + ((size_t)(x * 4 + w) * nph) * 8 : nullptr;
#define SX_STAMP(k) do { if (tr && p < nph) tr[(size_t)p * 8 + (k)] = clock64(); } while (0)
    for (int p = 0; p <= nph; ++p) {
        const bool work = p < nph;
        const int G = gi * NPAIR + pr;                           // group of this phase: clips 4 G .. 4 G + 3
        SX_STAMP(0);
        // ---- (1) the x-side input of this phase was asked for during the previous phase's h side (ask_x below) -----------------
        // ---- (2) wave 0: the cell of the previous phase (learned_models.py:110 / 146 / 192), publish h ------------------------
        if (w == 0 && p > 0 && !(a.debug & 4)) {
            const int Gp = gp * NPAIR + pr;
            float4 xg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (NXT0 == 0 && l == 0) {                           // hoisted input product: G[(clip * T + t)][4 unit .. 4 unit + 3]
                int clip = 4 * Gp + j;
                clip = clip < a.B ? clip : a.B - 1;
                xg = *(const float4 *)(a.ws + a.g_off + (((size_t)clip * T + tp) * (4 * SX_H) + (size_t)(16 * c + b) * 4) * 4);
            }
            const float4 *pp = &sP[(p - 1) & 1][0][lane];
            const float4 p0 = pp[0], p1 = pp[64], p2 = pp[128], p3 = pp[192];
            float cc = sC[gp][lane];
            float4 gs;
            const float h = lstm_cell_g((((p0.x + p1.x) + p2.x) + p3.x) + xg.x, (((p0.y + p1.y) + p2.y) + p3.y) + xg.y,
                                        (((p0.z + p1.z) + p2.z) + p3.z) + xg.z, (((p0.w + p1.w) + p2.w) + p3.w) + xg.w, &cc, &gs);
            sC[gp][lane] = cc;
            // float4 = units 4 q .. 4 q + 3 of clip j, assembled by the lanes with (b & 3) == 0; slot tp + 1 = step tp
            const float4 hv = make_float4(h, x4_row_shl<4>(h), x4_row_shl<8>(h), x4_row_shl<12>(h));
            if ((b & 3) == 0) {
                const unsigned vo = ((b >> 2) * 4 + j) * 16;
                const unsigned so = ((unsigned)(Gp * (T + 1) + tp + 1) * 128 + 4 * c) * 64;
                xcd_store16(rws, vo, hl_mine + so, hv, local);
                if (!top) xcd_store16(rws, vo, hc_mine + so, hv, false);
            }
            if (TRAIN) {
                const int clip = 4 * Gp + j, rb = clip >> 5, cl = clip & 31, RB = a.RB;
                const size_t u = 16 * c + b;
                hall_mine[(((size_t)(tp + 1) * RB + rb) * 128 + (u >> 2)) * 128 + cl * 4 + (u & 3)] = h;
                call_mine[(((size_t)(tp + 1) * RB + rb) * SX_H + u) * 32 + cl] = cc;
                gsave_mine[(((size_t)tp * RB + rb) * SX_H + u) * 32 + cl] = gs;
            }
        }
        if (!work) break;
        SX_STAMP(1);                            // (wave 0: the cell is done and published)
        bool ok = true;
        sx_f32x4 acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = (sx_f32x4){0.f, 0.f, 0.f, 0.f};
        // this wave's quarter of h[t-1] (slot t): asked for by the x waves two thirds into their x side (when the cell of the
        // previous phase has usually been published: the load's round trip - ~650 cycles - then hides under the remaining
        // MFMAs), by wave 0 right after its cell; anybody who still sees a sentinel asks again
        xcd_u32x4 hr[2];
        const unsigned hsrc = hl_mine + ((unsigned)(G * (T + 1) + t) * 128 + 32 * w) * 64;
        auto ask_h = [&]() {
            hr[0] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, hsrc, 16);
            hr[1] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, hsrc + 1024, 16);
        };
        // ---- (3) x side (waves 1..3) --------------------------------------------------------------------------------------
        if (xwave && (upper || NXT0 > 0)) {
            if (upper) {
                // the lower layer's h[t] from another XCD: wait until no lane sees the sentinel any more
                long long t0 = 0;
                for (unsigned spins = 1;; ++spins) {
                    bool bad = false;
#pragma unroll
                    for (int r = 0; r < NLD; ++r) bad |= sx_unpublished(xr[r]);
                    if (!__any(bad)) break;
                    if (!x4_keep_polling(spins, t0, a.status, p)) { ok = false; break; }
#pragma unroll
                    for (int r = 0; r < NLD; ++r)
                        if ((lane >> 2) + 16 * r < xcnt) xr[r] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, xsrc + r * 1024, 16);
                }
            }
            float4 *SX = &sX[xi][0];
#pragma unroll
            for (int r = 0; r < NLD; ++r)
                if (64 * r + lane < 4 * NXWMAX) SX[64 * r + lane] = x4_as_float4(xr[r]);      // (zeros beyond the wave's share)
            SX_LDS_SYNC();
            if (upper) sx_products<NXW1, true, (2 * NXW1) / 3>(acc, ax, SX + j, ask_h);
            else sx_products<NXW0, true, (2 * NXW0) / 3>(acc, ax, SX + j, ask_h);
        } else {
            ask_h();
        }
        SX_STAMP(2);                            // (x waves: the x side is done)
        // ---- (4) h side ---------------------------------------------------------------------------------------------------
        {
            if (ok) ok = sx_poll<2>(rws, lane16, hsrc, hr, a.status, p);
            SX_STAMP(3);                        // h[t-1] has arrived
            sH[w][lane] = x4_as_float4(hr[0]);
            sH[w][64 + lane] = x4_as_float4(hr[1]);
            {                                   // the next phase's x-side input: lands under the h side's MFMAs
                int gn = gi + 1, tn = t;
                if (gn == ng) { gn = 0; ++tn; }
                if (tn < T) ask_x(gn, tn);
            }
            SX_LDS_SYNC();
            sx_products<32, false, 99>(acc, ah, &sH[w][0] + j, []() {});
        }
        {
            sx_f32x4 s4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s4[r] = ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) + ((acc[4][r] + acc[5][r]) + (acc[6][r] + acc[7][r]));
            sP[p & 1][w][lane] = make_float4(s4[0], s4[1], s4[2], s4[3]);
        }
        if (!ok) XCD_LDS_ST(sAbort, 1);
        SX_STAMP(4);                            // products done
        __syncthreads();                        // the phase's partials are in sP
        SX_STAMP(5);
        // the abort word is looked at one phase late: read here, behind the barrier, but only tested a phase on, so that the LDS round
        // trip stays off the step's critical chain (an aborted launch's outputs are NaN whatever this block still stores)
        if (abort_seen) return;
        abort_seen = XCD_LDS_LD(sAbort);
        gp = gi; tp = t;
        if (++gi == ng) { gi = 0; ++t; }
    }
}

// y[clip][t] = predictions_layer.weight . h_top[t] (learned_models.py:113 / 148 / 195) from the top layer's exchange history;
// one workgroup of 64 threads per (t, group): thread (r, clip j) walks k-quads r, r + 16, ..., the 16 partials are summed in
// fixed order.  An aborted launch (status[0] != 0) poisons y with NaN.
__global__ void __launch_bounds__(64) seqx_out_head(const SeqXArgs a)
{
    __shared__ __attribute__((aligned(16))) float4 red[16][4];
    const int t = blockIdx.x, G = blockIdx.y, tid = threadIdx.x, r = tid >> 2, j = tid & 3;
    const float4 *h = (const float4 *)(a.ws + a.hl_off[a.L - 1]) + ((size_t)(G * (a.T + 1) + t + 1) * 128) * 4 + j;
    const float4 *w4 = (const float4 *)a.whead;                     // [4][128] float4
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q = r; q < 128; q += 16) {
        const float4 hv = h[q * 4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float4 wv = w4[o * 128 + q];
            acc[o] = fmaf(hv.x, wv.x, acc[o]);
            acc[o] = fmaf(hv.y, wv.y, acc[o]);
            acc[o] = fmaf(hv.z, wv.z, acc[o]);
            acc[o] = fmaf(hv.w, wv.w, acc[o]);
        }
    }
    red[r][j] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (tid < 4) {
        float4 s = red[0][tid];
        for (int k = 1; k < 16; ++k) {
            const float4 v = red[k][tid];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (a.status[0] != 0u) s = make_float4(NAN, NAN, NAN, NAN);
        const int clip = 4 * G + tid;
        if (clip < a.B) a.ystage[(size_t)clip * a.T + t] = s;
    }
}
