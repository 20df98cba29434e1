// NOT BUILT - kept as the record of an experiment (profiles/r4_xcd4_experiments.txt, item 3): the one-barrier form of the 4-clip
// persistent forward; it was measured slower than opnet_xcd4_forward and fails the T = 1 case of tests/test_opnet_xcd4_gpu.py.
// It was compiled by including it from opnet_abi.hip behind seq_xcd_kernels.hip and launching it in place of opnet_xcd4_forward.
// opnet_xcd4s_kernels.hip - the 4-clip persistent OPNet forward in single-barrier form (opnet_xcd4_forward2): what
// opnet_xcd4_forward (opnet_xcd4_kernels.hip) computes - reference baselines/learned_models.py:35-52 for groups of four clips, one
// group per XCD and row block, every weight resident in registers for all T steps, the launch chain's history layouts - with the
// phase structure of seqx_forward (seq_xcd_kernels.hip).
//
// Why.  opnet_xcd4_forward's phase is a chain in which nothing overlaps (profiles/r4_xcd4_phase_timeline.txt, 5 336 cycles a step):
//     products (188 MFMAs a wave, 2 340) -> barrier -> cells (740) -> history stores (300) -> gather of the NEXT phase's inputs by
//     three waves for all four (1 190) -> drain (290) -> barrier -> loop (260)
// Here a phase has ONE workgroup barrier (K-split partials -> cells) and every wave feeds itself:
//   * the finishes of phase p - 1 run at the START of phase p, each on its own wave and SIMD: wave 0 the LSTM2 cell (publishes h2),
//     wave 3 the LSTM1 cell (publishes h1), wave 2 the head's softmax + einsum (frames_boxes -> LDS); meanwhile waves 0 / 1 multiply
//     the fragments that read only x (LSTM1's x part: all of wave 0's LSTM1 share) or frames_boxes (LSTM2's input part);
//   * every wave gathers only the k-quads ITS fragments read - its K quarter of h2 (two 1-KB pieces) and one or two pieces of h1 -
//     into wave-private LDS, polls them itself and starts multiplying as soon as ITS data is there: LSTM2's 128 MFMAs first (h1
//     lands under them), then LSTM1's recurrent part and the head; no second barrier, no wave waits for another wave's gather;
//   * eight accumulator chains per product and a 6-deep fragment ring (seqx_forward measured: four chains = 13 cycles per 4x4x1
//     MFMA, eight = ~10);
//   * the abort word, the placement check, the exchange rings ("the data is the flag", slot (t + 1) & 3 = step t, re-armed two steps
//     on) and every history layout are opnet_xcd4_forward's.
// Phase (gi, s), T + 3 steps:  LSTM1 step s | head products of step s-1 | LSTM2 step s-3  (LSTM2 is one step further behind than in
// opnet_xcd4_forward: frames_boxes[s-3] was finished at the start of phase s-1 and crossed that phase's barrier).
//
// Tried on the way and dropped (measured on the box, tools/xcd4s_probe.py): the two recurrences as two CO-RESIDENT workgroups per
// CU (512 workgroups, two waves per SIMD, LSTM2 on one, LSTM1 + head on the other, frames_boxes through a full-history buffer).
// Alone the LSTM2 workgroups ran a step in 3 650 cycles and the LSTM1 ones in 4 300-4 700; together 5 400-5 700 - no better than one
// workgroup: an fp32 MFMA stream holds its SIMD's issue port, so each role's cells and polls (VALU) starve exactly while the other
// role multiplies, whatever s_setprio says, and LSTM1's chain - the one LSTM2 waits for - suffers most.
//
// Summation order (differs from opnet_xcd4_forward's in the last bits; both are held to the reference's goldens and the fp64 port):
// LSTM2 gate = ((w0 + w1) + w2) + w3 over the waves' K quarters, a quarter = ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7)) over
// eight interleaved ascending-k chains (chain = 4 * (k-quad parity) + k mod 4), wave 1's chains started by the W_ih2 part; LSTM1
// gate = sum over waves of (low k half + high k half), a wave's partial the same eight-chain sum over its fragments (chain = 4 *
// (fragment parity) + k mod 4); logits = sum over waves, over the k quarters of the wave's slice, eight chains likewise.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opnet_ctx.h"

#define X4S_XSLOTS 6           // x of the last phases kept for the head's einsum: >= X4_NGMAX + 2
// the fragment ring of the LSTM1 / head products (register quads / fragments ahead): the LDS round trip is ~160 cycles, a fragment's
// four MFMAs ~40
#ifndef X4S_RING
#define X4S_RING 8
#endif
#ifndef X4S_AHEAD
#define X4S_AHEAD 6
#endif

// status words and XCC sentinels, the two exchange rings (slot 0 = the zero initial state, the others unpublished): x4_init_body
// (opnet_xcd4_kernels.hip) serves both forms.

// fragments M0 .. M1 - 1 of a wave's LSTM1 / head share (0..10: LSTM1, two k-quads each; 11..14: the head, four k-quads each), B
// operands through a ring of X4S_RING register quads, X4S_AHEAD ahead; EIGHT accumulator chains per product (fragment parity x
// element); the weights are AccVGPRs
template <int M0, int M1>
__device__ __forceinline__ void x4s_products(x4_f32x4 (&c1)[8], x4_f32x4 (&cH)[8], const float (&a1)[44], const float (&as_)[16],
                                             const float4 *F1, const float4 *FH)
{
    constexpr int N = M1 - M0;
    if (N <= 0) return;
    auto frag = [&](int k) -> const float4 * { return M0 + k < 11 ? F1 + (M0 + k) * 8 : FH + (M0 + k - 11) * 16; };
    float4 bf[X4S_RING];
#pragma unroll
    for (int i = 0; i < X4S_AHEAD; ++i)
        if (i < N) bf[i] = *frag(i);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k + X4S_AHEAD < N) bf[(k + X4S_AHEAD) % X4S_RING] = *frag(k + X4S_AHEAD);
        __builtin_amdgcn_sched_barrier(0);
        const float4 bq = bf[k % X4S_RING];
        const int o = ((M0 + k) & 1) * 4;
        if (M0 + k < 11) {
            const int m = M0 + k < 11 ? M0 + k : 0;
            SX_MFMA_A(c1[o + 0], a1[4 * m], bq.x);
            SX_MFMA_A(c1[o + 1], a1[4 * m + 1], bq.y);
            SX_MFMA_A(c1[o + 2], a1[4 * m + 2], bq.z);
            SX_MFMA_A(c1[o + 3], a1[4 * m + 3], bq.w);
        } else {
            const int m = M0 + k - 11 >= 0 ? M0 + k - 11 : 0;
            SX_MFMA_A(cH[o + 0], as_[4 * m], bq.x);
            SX_MFMA_A(cH[o + 1], as_[4 * m + 1], bq.y);
            SX_MFMA_A(cH[o + 2], as_[4 * m + 2], bq.z);
            SX_MFMA_A(cH[o + 3], as_[4 * m + 3], bq.w);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// lane l of every row of 16 lanes receives lane l ^ 8 / l ^ 4 of its row through DPP (row_ror) instead of ds_bpermute: the head's
// softmax and einsum are chains of ~16 such exchanges, ~120 cycles each through the LDS crossbar (measured: 2 200 cycles for the
// head's finish, twice a cell), a few cycles each as DPP moves.  Same partners as __shfl_xor, so the same sums bit for bit.
__device__ __forceinline__ float x4s_xor8(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));      // row_ror:8
}
__device__ __forceinline__ float x4s_xor4(float v)
{
    const int vi = __float_as_int(v);
    int t = __builtin_amdgcn_update_dpp(vi, vi, 0x12c, 0xf, 0x5, false);   // row_ror:12 (lane i <- i + 4): banks 0, 2
    t = __builtin_amdgcn_update_dpp(t, vi, 0x124, 0xf, 0xa, false);        // row_ror:4  (lane i <- i - 4): banks 1, 3
    return __int_as_float(t);
}

// a wave's 1-KB pieces at src (lane l: 16 B at src + 1024 q + 16 l): while any lane still sees a sentinel word, all NP are asked for
// again (bounded).  false = abort (wave-uniform).  np < NP: only the first np pieces exist.
template <int NP>
__device__ __forceinline__ bool x4s_poll(__amdgpu_buffer_rsrc_t rws, unsigned lane16, unsigned src, int np, xcd_u32x4 (&r)[NP],
                                         unsigned *status, int phase)
{
    long long t0 = 0;
    for (unsigned spins = 1;; ++spins) {
        bool bad = false;
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (q < np) bad |= x4_unpublished(r[q]);
        if (!__any(bad)) return true;
        if (!x4_keep_polling(spins, t0, status, phase)) return false;
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (q < np) r[q] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, src + q * 1024, 16);   // sc1
    }
}

// trace (tools/xcd4s_probe.py): [wave][phase][8] s_memtime stamps of CU 0 of XCD 0
#define X4S_STAMP(k) do { if (tr && p < nph) tr[(size_t)p * 8 + (k)] = clock64(); } while (0)

template <bool TRAIN>
__global__ void __launch_bounds__(256) opnet_xcd4_forward2(const Xcd4Args a)
{
    __shared__ __attribute__((aligned(1024))) float4 sH[4][128];        // wave-private: the wave's K quarter of h2[s-4]
    __shared__ __attribute__((aligned(1024))) float4 sBW[4][352];       // wave-private: [x 24 k-quads | h1 64 k-quads] x 4 clips (the k-quads the wave reads)
    __shared__ __attribute__((aligned(16))) float4 sXS[X4S_XSLOTS][96]; // x of the last phases (by phase % X4S_XSLOTS): the head's einsum reads it ng + 1 phases later
    __shared__ __attribute__((aligned(16))) float4 sFB[2][X4_NGMAX][8]; // frames_boxes by step parity: [k-quad][clip]
    __shared__ __attribute__((aligned(16))) float4 sP2[2][4][64];       // LSTM2 K-split partials by phase parity
    __shared__ __attribute__((aligned(16))) float4 sPB[2][4][2][64];    // LSTM1 | head partials by phase parity
    __shared__ float sC2[X4_NGMAX][64];
    __shared__ float sC1[X4_NGMAX][32];
    __shared__ float4 sPad[2560];           // 40 KB never used: > 80 KB of LDS in total keep a second workgroup off the CU
    __shared__ int sAbort, sLocal;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = blockIdx.x & 7, c = blockIdx.x >> 3;
    const int T = a.T, RB = a.RB, ng = a.RB;
    const int b = lane >> 2, j = lane & 3;
    if (w == 0) {
        const int loc = xcd_group_is_local(a.status, x);
        if (lane == 0) {
            XCD_LDS_ST(sLocal, loc > 0 && a.force_safe == 0);
            XCD_LDS_ST(sAbort, loc < 0);
            if (loc == 0 && c == 0) atomicAdd(a.status + 3, 1u);
        }
    }
    for (int i = tid; i < X4_NGMAX * 64; i += 256) (&sC2[0][0])[i] = 0.f;
    for (int i = tid; i < X4_NGMAX * 32; i += 256) (&sC1[0][0])[i] = 0.f;
    if (tid < 2 * X4_NGMAX * 8) (&sFB[0][0][0])[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.debug & 0x40000000) sPad[tid * 10] = make_float4(0.f, 0.f, 0.f, 0.f);     // (keeps the padding allocated)

    // ---- resident weights: AccVGPRs (MFMA A operands) -------------------------------------------------------------------------
    const X4Packed P = x4_packed_layout();
    float ah[128];                                              // W_hh2, the wave's K quarter
    float a1[44], as_[16];                                      // LSTM1 (x | h part), the selection head
    float4 ax0 = make_float4(0.f, 0.f, 0.f, 0.f), ax1 = ax0;    // wave 1: W_ih2 (K = 6 -> 8)
    {
        const float4 *p2 = (const float4 *)(a.pk + P.a2) + ((size_t)(c * 4 + w) * 32) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const float4 v = p2[q * 64];
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ah[4 * q]) : "v"(v.x));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ah[4 * q + 1]) : "v"(v.y));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ah[4 * q + 2]) : "v"(v.z));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ah[4 * q + 3]) : "v"(v.w));
        }
        const float4 *p1 = (const float4 *)(a.pk + P.a1) + ((size_t)(c * 4 + w) * 11) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            const float4 v = p1[q * 64];
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a1[4 * q]) : "v"(v.x));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a1[4 * q + 1]) : "v"(v.y));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a1[4 * q + 2]) : "v"(v.z));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a1[4 * q + 3]) : "v"(v.w));
        }
        const float4 *ps = (const float4 *)(a.pk + P.as) + (size_t)(w * 4) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = ps[q * 64];
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(as_[4 * q]) : "v"(v.x));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(as_[4 * q + 1]) : "v"(v.y));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(as_[4 * q + 2]) : "v"(v.z));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(as_[4 * q + 3]) : "v"(v.w));
        }
        if (w == 1) {
            const float4 *px = (const float4 *)(a.pk + P.ax) + (size_t)(c * 2) * 64 + lane;
            ax0 = px[0];
            ax1 = px[64];
        }
    }

    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void *)a.ws, 0, 0x7fffffff, 0x00020000);
    const unsigned cb = 4 * x;                                  // first clip of this XCD's groups within a row block
    const unsigned lane16 = lane * 16;
    const xcd_u32x4 sent4 = {X4_SENT, X4_SENT, X4_SENT, X4_SENT};
    const float4 sentf = x4_as_float4(sent4);
    __syncthreads();
    if (XCD_LDS_LD(sAbort)) return;
    int abort_seen = 0;
    const bool local = __builtin_amdgcn_readfirstlane(XCD_LDS_LD(sLocal)) != 0;
    const int nph = (T + 3) * ng;
    unsigned long long *const tr = (a.trace && c == 0 && x == 0 && lane == 0) ? a.trace + ((size_t)w * nph) * 8 : nullptr;
    // the two 1-KB pieces (16 k-quads each) of h1[s-1] this wave gathers: wave 0: 0, 1 (its head fragments read piece 0) | 1: 0, 1 |
    // 2: 1, 2 | 3: 2, 3
    const int pc0 = w == 0 ? 0 : w - 1;
    float4 *const SW = &sBW[w][0];
    // x[s] (packed input: read-only, no sentinel), asked for one phase early.  Wave 0: all 24 k-quads (its LSTM1 fragments, and the
    // head's einsum later through sXS); wave 1: k-quads 22, 23.
    const unsigned xlane = ((lane >> 2) * 32 + cb + (lane & 3)) * 16;   // [k-quad][32 clips] float4: k-quad lane >> 2, clip cb + (lane & 3)
    // (every wave and lane issues the same two loads - what a lane does not need is read and dropped: a load under a condition
    // makes the compiler merge registers behind it and wait for the load on the spot, vmcnt(0), in the middle of the phase)
    const unsigned xoff = w == 1 ? 22 * 512 : 0;
    xcd_u32x4 xa = __builtin_amdgcn_raw_buffer_load_b128(rws, xlane, a.xp_off + xoff, 0);       // (gi, s) = (0, 0)
    xcd_u32x4 xb = __builtin_amdgcn_raw_buffer_load_b128(rws, xlane, a.xp_off + 16 * 512, 0);

    int gi = 0, s = 0, gp = 0, sp = 0;
    for (int p = 0; p <= nph; ++p) {
        const bool work = p < nph;
        const unsigned gg = gi * 8 + x;
        X4S_STAMP(0);
        // ================================ the finishes of the previous phase (gp, sp) ===========================================
        if (p > 0) {
            const unsigned ggp = gp * 8 + x;
            const int par = (p - 1) & 1, rb = gp;
            if (w == 0) {
                // ---- LSTM2 cell of step t = sp - 3 (learned_models.py:46): lane = (unit 16 c + b, clip j) ------------------------
                const int t = sp - 3;
                if (t >= 0 && t < T && !(a.debug & 4)) {
                    const float4 *pp = &sP2[par][0][lane];
                    const float4 p0 = pp[0], p1 = pp[64], p2 = pp[128], p3 = pp[192];
                    float cc = sC2[gp][lane];
                    float4 gs;
                    const float h = lstm_cell_g(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                                                ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w, &cc, &gs);
                    sC2[gp][lane] = cc;
                    // exchange: float4 = units 4 q .. 4 q + 3 of clip j, by the lanes with (b & 3) == 0; ring slot (t + 1) & 3 = step t;
                    // the slot two steps on is re-armed (this wave's stores of the phase before have long been acknowledged)
                    const float4 hv = make_float4(h, x4_row_shl<4>(h), x4_row_shl<8>(h), x4_row_shl<12>(h));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if ((b & 3) == 0) {
                        const unsigned vo = ((b >> 2) * 4 + j) * 16;
                        xcd_store16(rws, vo, a.h2x_off + ((ggp * X4_SLOTS + ((t + 1) & 3)) * 128 + 4 * c) * 64, hv, local);
                        xcd_store16(rws, vo, a.h2x_off + ((ggp * X4_SLOTS + ((t + 3) & 3)) * 128 + 4 * c) * 64, sentf, local);
                    }
                    // the histories of the backward pass / the output head
                    const size_t u = 16 * c + b;
                    if (!(a.debug & 2))
                    ((float *)(a.ws + a.h2_off))[(((size_t)(t + 1) * RB + rb) * 128 + (u >> 2)) * 128 + (cb + j) * 4 + (u & 3)] = h;
                    if (TRAIN && !(a.debug & 2)) {
                        ((float *)(a.ws + a.c2_off))[(((size_t)(t + 1) * RB + rb) * 512 + u) * 32 + cb + j] = cc;
                        ((float4 *)(a.ws + a.g2_off))[(((size_t)t * RB + rb) * 512 + u) * 32 + cb + j] = gs;
                    }
                }
            } else if (w == 3) {
                // ---- LSTM1 cell of step t = sp (learned_models.py:39): lanes 0..31 = (unit 8 c + b, clip j) ----------------------
                const int t = sp;
                if (t < T && lane < 32 && !(a.debug & 4)) {
                    float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 lo = sPB[par][q][0][lane], hi = sPB[par][q][0][lane + 32];
                        g[0] += lo.x + hi.x; g[1] += lo.y + hi.y; g[2] += lo.z + hi.z; g[3] += lo.w + hi.w;
                    }
                    float cc = sC1[gp][lane];
                    float4 gs;
                    const float h = lstm_cell_g(g[0], g[1], g[2], g[3], &cc, &gs);
                    sC1[gp][lane] = cc;
                    const float4 hv = make_float4(h, x4_row_shl<4>(h), x4_row_shl<8>(h), x4_row_shl<12>(h));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if ((b & 3) == 0) {     // ring slot (t + 1) & 3; the slot two steps on is re-armed
                        const unsigned vo = ((b >> 2) * 4 + j) * 16;
                        xcd_store16(rws, vo, a.h1x_off + ((ggp * X4_SLOTS + ((t + 1) & 3)) * 64 + 2 * c) * 64, hv, local);
                        xcd_store16(rws, vo, a.h1x_off + ((ggp * X4_SLOTS + ((t + 3) & 3)) * 64 + 2 * c) * 64, sentf, local);
                    }
                    if (TRAIN && !(a.debug & 2)) {
                        const size_t u = 8 * c + b;
                        ((float *)(a.ws + a.h1_off))[(((size_t)(t + 1) * RB + rb) * 64 + (u >> 2)) * 128 + (cb + j) * 4 + (u & 3)] = h;
                        ((float *)(a.ws + a.c1_off))[(((size_t)(t + 1) * RB + rb) * 256 + u) * 32 + cb + j] = cc;
                        ((float4 *)(a.ws + a.g1_off))[(((size_t)t * RB + rb) * 256 + u) * 32 + cb + j] = gs;
                    }
                }
            } else if (w == 2) {
                // ---- selection head of step t = sp - 1 (learned_models.py:40-43,50): lanes 0..15 = (slot quad rg, clip j) ---------
                const int t = sp - 1;
                if (t >= 0 && t < T && lane < 16 && !(a.debug & 8)) {
                    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const float4 pv = sPB[par][q][1][lane + 16 * kk];
                            v[0] += pv.x; v[1] += pv.y; v[2] += pv.z; v[3] += pv.w;
                        }
                    const int rg = lane >> 2;
                    float m = fmaxf(fmaxf(v[0], v[1]), v[2]);
                    if (rg < 3) m = fmaxf(m, v[3]);                 // slot 15 does not exist
                    m = fmaxf(m, x4s_xor4(m));
                    m = fmaxf(m, x4s_xor8(m));
                    float e[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[r] = __expf(v[r] - m);
                    if (rg == 3) e[3] = 0.f;
                    float sum = (e[0] + e[1]) + (e[2] + e[3]);
                    sum += x4s_xor4(sum);
                    sum += x4s_xor8(sum);
                    const float inv = 1.0f / sum;
                    const float pr[4] = {e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv};
                    // frames_boxes[j][f] = sum_o boxes[j][t][o][f] p[o] (einsum "bfot,bfo->bft"): this lane's slots 4 rg .. 4 rg + 3
                    // = k-quads 6 rg .. 6 rg + 5 of x[t], left in sXS by wave 0 in the phase that computed LSTM1 step t (ng + 1 phases ago)
                    const float4 *X = &sXS[(p - 1 - ng + 2 * X4S_XSLOTS) % X4S_XSLOTS][0] + (6 * rg) * 4 + j;
                    float xf[24];
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const float4 xv = X[q * 4];
                        xf[4 * q] = xv.x; xf[4 * q + 1] = xv.y; xf[4 * q + 2] = xv.z; xf[4 * q + 3] = xv.w;
                    }
                    float fbv[8];
#pragma unroll
                    for (int f = 0; f < OPNET_FEATS_; ++f) {
                        float acc = pr[0] * xf[f];
                        acc = fmaf(pr[1], xf[6 + f], acc);
                        acc = fmaf(pr[2], xf[12 + f], acc);
                        acc = fmaf(pr[3], xf[18 + f], acc);
                        acc += x4s_xor4(acc);
                        acc += x4s_xor8(acc);
                        fbv[f] = acc;
                    }
                    const float4 fb0 = make_float4(fbv[0], fbv[1], fbv[2], fbv[3]), fb1 = make_float4(fbv[4], fbv[5], 0.f, 0.f);
                    if (rg == 0) {          // [k-quad][clip]: LSTM2's input part of step t, read by wave 1 two phases on
                        sFB[t & 1][gp][j] = fb0;
                        sFB[t & 1][gp][4 + j] = fb1;
                    }
                    if (c == (sp & 31)) {           // every CU computes the head; one of them records it
                        const unsigned clip = rb * 32 + cb + j;
                        float *lg = (float *)(a.ws + a.lg_off);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (4 * rg + r < OPNET_SLOTS_) lg[((size_t)clip * OPNET_SLOTS_ + 4 * rg + r) * T + t] = v[r];
                        if (TRAIN) {
                            float4 *ps = (float4 *)(a.ws + a.ps_off);
                            ps[((size_t)(t * RB + rb) * 4 + rg) * 32 + cb + j] = make_float4(pr[0], pr[1], pr[2], pr[3]);
                            float4 *x2 = (float4 *)(a.ws + a.x2_off) + ((size_t)(t * RB + rb) * 2) * 32 + cb + j;
                            if (rg == 0) x2[0] = fb0;          // (two stores: a select between two float4 goes through scratch)
                            if (rg == 1) x2[32] = fb1;
                        }
                    }
                }
            }
        }
        if (!work) break;
        X4S_STAMP(1);                           // (the wave's finish is done, its h published)
        const bool do1 = s <= T;                // h1[s-1] is read (LSTM1 step s, head step s-1): somebody still publishes it
        const bool do2 = s >= 3 && s - 3 < T;   // LSTM2 step s-3 exists
        // ---- the x side: this phase's x (asked for a phase ago) into the wave's input copy; the fragments that wait for nothing ----
        sx_f32x4 acc[8];
        x4_f32x4 c1[8], cH[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { acc[q] = (sx_f32x4){0.f, 0.f, 0.f, 0.f}; c1[q] = cH[q] = (x4_f32x4){0.f, 0.f, 0.f, 0.f}; }
        const float4 *F1 = SW + (22 * w + (b >> 3)) * 4 + j;            // LSTM1 fragment m: + 8 m float4 (two k-quads: the block's k half)
        const float4 *FH = SW + 96 + (16 * w + (b >> 2)) * 4 + j;      // head fragment m: + 16 m
        if (w == 0) {
            SW[lane] = x4_as_float4(xa);
            float4 *XS = &sXS[p % X4S_XSLOTS][0];
            XS[lane] = x4_as_float4(xa);
            if (lane < 32) { SW[64 + lane] = x4_as_float4(xb); XS[64 + lane] = x4_as_float4(xb); }
        } else if (w == 1 && lane < 8) {
            SW[88 + lane] = x4_as_float4(xa);
        }
        // ---- ask for this wave's pieces (behind the x registers' last use: the counter the waits go by is in order): its K quarter of h2[s-4] (ring slot (s - 3) & 3), its pieces of h1[s-1] (slot s & 3) ------
        xcd_u32x4 h2r[2], h1r[2];
        const unsigned h2src = a.h2x_off + ((gg * X4_SLOTS + ((s + 1) & 3)) * 128 + 32 * w) * 64;
        const unsigned h1src = a.h1x_off + (gg * X4_SLOTS + (s & 3)) * 4096 + pc0 * 1024;
        h2r[0] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, h2src, 16);          // (unconditional, see the x loads; a phase
        h2r[1] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, h2src + 1024, 16);   // without LSTM2 / LSTM1 drops them unread)
        h1r[0] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, h1src, 16);
        h1r[1] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, h1src + 1024, 16);
        SX_LDS_SYNC();
        if (!(a.debug & 16)) {
            // all of wave 0's LSTM1 share (k-quads 0..21 are x), fragment 0 of wave 1 (k-quads 22, 23); wave 1 also LSTM2's input
            // part, W_ih2 . frames_boxes[s-3] (K = 6 -> 8): its 6 MFMAs start the wave's LSTM2 chains
            if (w == 0) x4s_products<0, 11>(c1, cH, a1, as_, F1, FH);
            else if (w == 1) {
                x4s_products<0, 1>(c1, cH, a1, as_, F1, FH);
                if (do2) {
                    const float4 f0 = sFB[(s - 3) & 1][gi][j], f1 = sFB[(s - 3) & 1][gi][4 + j];
                    SX_MFMA_V(acc[0], ax0.x, f0.x);
                    SX_MFMA_V(acc[1], ax0.y, f0.y);
                    SX_MFMA_V(acc[2], ax0.z, f0.z);
                    SX_MFMA_V(acc[3], ax0.w, f0.w);
                    SX_MFMA_V(acc[4], ax1.x, f1.x);
                    SX_MFMA_V(acc[5], ax1.y, f1.y);
                }
            }
        }
        X4S_STAMP(2);
        // ---- LSTM2's recurrent part: the wave's K quarter of h2[s-4] ------------------------------------------------------------
        bool ok = true;
        // both sets are polled here, in front of all the products: they were published at about the same time (the two cells run
        // side by side), and a set that is only looked at behind LSTM2's MFMAs pays its retry round trip there, in the open
        if (do1) ok = x4s_poll<2>(rws, lane16, h1src, 2, h1r, a.status, p);
        if (do2) {
            if (ok) ok = x4s_poll<2>(rws, lane16, h2src, 2, h2r, a.status, p);
            X4S_STAMP(3);                       // h2 (and h1) have arrived
            sH[w][lane] = x4_as_float4(h2r[0]);
            sH[w][64 + lane] = x4_as_float4(h2r[1]);
            SX_LDS_SYNC();
        }
        {                                       // the next phase's x: asked for behind the h2 poll, lands under the products
            int gn = gi + 1, sn = s;
            if (gn == ng) { gn = 0; ++sn; }
            const int t0 = sn < T ? sn : T - 1;
            const unsigned o0 = a.xp_off + (unsigned)((t0 * RB + gn) * OPNET_KXQ) * 512;
            xa = __builtin_amdgcn_raw_buffer_load_b128(rws, xlane, o0 + xoff, 0);
            xb = __builtin_amdgcn_raw_buffer_load_b128(rws, xlane, o0 + 16 * 512, 0);
        }
        if (do2) {
            if (!(a.debug & 16)) sx_products<32, true, 99>(acc, ah, &sH[w][0] + j, []() {});
        }
        X4S_STAMP(4);
        // ---- LSTM1's recurrent part and the head: the wave's pieces of h1[s-1] (they landed under LSTM2's MFMAs) --------------------
        if (do1) {
            SW[96 + pc0 * 64 + lane] = x4_as_float4(h1r[0]);
            SW[96 + (pc0 + 1) * 64 + lane] = x4_as_float4(h1r[1]);
            SX_LDS_SYNC();
            if (!(a.debug & 16)) {
                if (w == 0) x4s_products<11, 15>(c1, cH, a1, as_, F1, FH);
                else if (w == 1) x4s_products<1, 15>(c1, cH, a1, as_, F1, FH);
                else x4s_products<0, 15>(c1, cH, a1, as_, F1, FH);
            }
        }
        X4_MFMA_DRAIN8(c1[0], c1[1], c1[2], c1[3], c1[4], c1[5], c1[6], c1[7]);
        X4_MFMA_DRAIN8(cH[0], cH[1], cH[2], cH[3], cH[4], cH[5], cH[6], cH[7]);
        {
            sx_f32x4 s4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s4[r] = ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) + ((acc[4][r] + acc[5][r]) + (acc[6][r] + acc[7][r]));
            sP2[p & 1][w][lane] = make_float4(s4[0], s4[1], s4[2], s4[3]);
            const x4_f32x4 s1 = ((c1[0] + c1[1]) + (c1[2] + c1[3])) + ((c1[4] + c1[5]) + (c1[6] + c1[7]));
            const x4_f32x4 sHd = ((cH[0] + cH[1]) + (cH[2] + cH[3])) + ((cH[4] + cH[5]) + (cH[6] + cH[7]));
            sPB[p & 1][w][0][lane] = make_float4(s1[0], s1[1], s1[2], s1[3]);
            sPB[p & 1][w][1][lane] = make_float4(sHd[0], sHd[1], sHd[2], sHd[3]);
        }
        if (!ok) XCD_LDS_ST(sAbort, 1);
        X4S_STAMP(5);                           // products done
        __syncthreads();                        // the phase's partials are in sP2 / sPB
        X4S_STAMP(6);
        // the abort word is looked at one phase late (opnet_xcd4_forward)
        if (abort_seen) return;
        abort_seen = XCD_LDS_LD(sAbort);
        gp = gi; sp = s;
        if (++gi == ng) { gi = 0; ++s; }
    }
}
