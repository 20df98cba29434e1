// seq_xcdt_kernels.hip - the stacked LSTM of the sibling reasoners, THROUGHPUT form: one persistent launch per forward on
// 16-clip column groups (v_mfma_f32_16x16x4_f32), the scheme of opnet_xcd_kernels.hip.
//
// What is computed: the nn.LSTM(bias=False, batch_first) of reference baselines/learned_models.py
//   BaselineLstm   :99-101, 110-115   (1 layer, 75 -> 512)
//   NonLinearLstm  :135-137, 146-148  (2 layers, 3840 -> 512 -> 512)
//   TransformerLstm:170-172, 192-195  (2 layers, 256 -> 512 -> 512)
// - the same function as seqx_forward (seq_xcd_kernels.hip: 4-clip groups, the LATENCY form, which stays the engine for lone /
// small requests and for the training forward) and lstm_stack_step (seq_kernels.hip: every other shape).
//
// Why: the 4-clip form fills 4 of 16 MFMA columns and its phases are barrier-serialised; beyond 16
// clips (L = 2) / 32 (L = 1) a launch grows linearly at <= 0.25 of the fp32 MFMA peak, while opnet_xcd_forward runs the same
// recurrence at 0.74.  Here every weight of the stack is resident in the registers of the PRODUCT waves for all T steps, and a
// second wave per SIMD (the FINISH wave) does the cells, the publish and the gather of the phase after next under the next
// phase's MFMAs - exactly opnet_xcd_forward's two roles, one s_barrier per phase.
//
// Decomposition.  256 workgroups of 8 waves, one per CU; XCD x = blockIdx.x & 7, CU c = blockIdx.x >> 3; the product wave of
// SIMD w owns TILE t2 = 4 c + w = hidden units 4 t2 .. 4 t2 + 3 (16 gate rows: row i <-> unit 4 t2 + (i >> 2), gate i & 3).
//   MODE 1 (L = 1, direct input, KX <= 80):  XCD x runs the layer for ITS groups (contiguous share of ceil(B / 16)):
//       A image per tile: [W_ih 5 hexadecets (K padded to 80) | W_hh 32] = 37 float4 per lane (148 VGPRs)
//       phase (g, s), s = 0 .. T-1: gates[s] = [x[s] | h[s-1]]
//   MODE 2 (L = 2, the layer-0 input product HOISTED into one GEMM G [B T][2048] before the launch - also for the 256-wide
//       input of TransformerLstm: two layers need 112 hexadecets per tile, a pair of XCDs holds 2 x ~50):
//       XCD 2p = role A, XCD 2p + 1 = role B, 48 hexadecets (192 VGPRs) each:
//       A: [W_hh0 32 | W_ih1[:, 0:256] 16]    phase (g, s), s = 0 .. T:   gates0[s] = W_hh0 h0[s-1] (+ G[s] in the cell),
//                                             and, ON THE SAME B FRAGMENTS, P1[s-1] = W_ih1[:, 0:256] h0[s-1][0:256]
//                                             (two more accumulator chains), handed to role B through memory
//       B: [W_ih1[:, 256:512] 16 | W_hh1 32]  phase (g, t), t = 0 .. T-1: gates1[t] = [h0[t][256:512] | h1[t-1]] (+ P1[t] in the cell)
//       Layer 0 needs nothing from layer 1 and runs ahead; every buffer holds the FULL history (each word written once per
//       launch and read only behind its flag), so there is no ring to overrun and no flow control between the two XCDs.
// Exchange (cdna_hip_programming.md Guideline 16 recipe R1, as opnet_xcd_forward): h leaves a finish wave as 16-byte stores
// into the history (slot t + 1 = step t, slot 0 = zeros), every storing wave drains vmcnt(0), the last of the CU's four finish
// waves (LDS arrival counter) stores ONE flag per CU; consumers poll the 32 flags with sc1 loads and gather with sc1 LDS-DMA.
// Stores that stay inside the XCD are plain (the line stays in this XCD's L2) when the placement check passed; everything role
// B reads from role A - h0's upper half (a second, written-through copy), P1, A's flags (a second copy) - is written through.
// Summation order: a gate pre-activation = (sum over even hexadecets) + (sum over odd hexadecets) [+ G | + P1], P1 likewise
// (even) + (odd); each partial an ascending-k fmaf chain per MFMA lane group.  Differs from the 4-clip form's K quarters and
// from the launch chain's K split: the three agree to rounding and are all held to the reference goldens / the fp64 oracle.
// y = predictions_layer . h_top is not on the recurrence: seqt_out_head computes it from the top layer's history afterwards.
// Every spin is bounded (XCD_SPIN_LIMIT); an aborted launch leaves status[0] != 0 and y = NaN.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "opnet_ctx.h"

#define ST_H 512
#define ST_NGMAX 8             // 16-clip groups one XCD (pair) carries in one launch
#define ST_NXH1 5              // MODE 1: hexadecets of the direct input (KX <= 80)
#define ST_NH1 (ST_NXH1 + 32)  // MODE 1: hexadecets per tile
#define ST_NH2 48              // MODE 2: hexadecets per tile and role

struct SeqTArgs {
    int B, T, mode, NGT;       // NGT = ceil(B / 16)
    int KX;                    // MODE 1: input features per frame
    const float *pk;           // seqt_pack image [role][128 tiles][NH][64 lanes] float4
    const float *whead;        // predictions_layer.weight [4][512], the caller's row-major tensor
    char *ws;                  // workspace base; the *_off below are bytes into it (one buffer descriptor, < 2 GiB)
    unsigned xp_off;           // MODE 1: [NGT][T][20][16] float4, slot t = x[t]
    unsigned hl_off[2];        // layer l: [NGT][T + 1][128][16] float4, slot t + 1 = step t, slot 0 zero
    unsigned hc_off;           // MODE 2: units 256..511 of layer 0 written through for role B: [NGT][T + 1][64][16] float4
    unsigned flags_off;        // [NGT][3][32]: 0 = layer 0 (or the single layer), 1 = layer 0 written through, 2 = layer 1
    unsigned status_off;       // [0] abort code, [1] block, [2] phase, [3] groups on the write-through path, [8 + b] XCC id of block b
    unsigned *status;
    const float *G;            // MODE 2: hoisted layer-0 input product [B T][2048], row b T + t, column 4 unit + gate
    float4 *P1;                // MODE 2: [NGT][T][128 tiles][64 lanes]: role A's partial of layer 1's gates
    float4 *y;                 // caller's [B][T] float4
    int force_safe, debug;     // debug (tools / tests): bit 3 = nobody publishes (forces the abort path)
    unsigned long long *trace; // tools: [2 blocks][phases][8] s_memtime stamps of blocks 0 and 1 (null = off)
};

// groups of set `set` of `nsets` (8 XCDs / 4 XCD pairs): a contiguous share
__host__ __device__ inline void st_groups(int NGT, int nsets, int set, int *g0, int *ng)
{
    const int base = NGT / nsets, rem = NGT % nsets;
    *ng = base + (set < rem ? 1 : 0);
    *g0 = set * base + (set < rem ? set : rem);
}

// fp32 weights -> the register images.  Lane (i = lane & 15, u = lane >> 4), element e of hexadecet q = W[row(i)][16 q + 4 u + e]
// of the role's concatenated K (see the header); row(i) = gate (i & 3) x 512 + unit 4 tile + (i >> 2) (torch's gate-major rows)
__global__ void __launch_bounds__(256) seqt_pack(float *__restrict__ out, const float *__restrict__ w_ih0, const float *__restrict__ w_hh0,
                                                 const float *__restrict__ w_ih1, const float *__restrict__ w_hh1, int mode, int KX)
{
    const int NH = mode == 1 ? ST_NH1 : ST_NH2;
    const size_t total = (size_t)(mode == 1 ? 1 : 2) * 128 * NH * 256;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 3, lane = (idx >> 2) & 63, i = lane & 15, u = lane >> 4;
        const size_t r = idx >> 8;
        const int q = r % NH, tile = (r / NH) % 128, role = (int)(r / ((size_t)NH * 128));
        const size_t row = (size_t)(i & 3) * ST_H + 4 * tile + (i >> 2);
        const int kk = 16 * q + 4 * u + e;
        float v;
        if (mode == 1) {
            if (q < ST_NXH1) v = kk < KX ? w_ih0[row * KX + kk] : 0.f;
            else v = w_hh0[row * ST_H + kk - 16 * ST_NXH1];
        } else if (role == 0) {
            v = q < 32 ? w_hh0[row * ST_H + kk] : w_ih1[row * ST_H + kk - 512];
        } else {
            v = q < 16 ? w_ih1[row * ST_H + 256 + kk] : w_hh1[row * ST_H + kk - 256];
        }
        out[idx] = v;
    }
}

// status words, XCC sentinels, flags, slot 0 of the histories; MODE 1: x [B][T][KX] -> xp [NGT][T][20][16] float4 (zeros beyond KX / B)
__global__ void __launch_bounds__(256) seqt_init(const SeqTArgs a, const float *__restrict__ x)
{
    const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x, n = (long)gridDim.x * blockDim.x;
    if (tid < 8) a.status[tid] = 0u;
    for (long i = tid; i < 256; i += n) a.status[8 + i] = 0xffffffffu;
    unsigned *flags = (unsigned *)(a.ws + a.flags_off);
    for (long i = tid; i < (long)a.NGT * 96; i += n) flags[i] = 0u;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const long per = (long)(a.T + 1) * 2048;                           // float4 per group and layer
    for (int l = 0; l < (a.mode == 1 ? 1 : 2); ++l) {
        float4 *hl = (float4 *)(a.ws + a.hl_off[l]);
        for (long i = tid; i < (long)a.NGT * 2048; i += n) hl[(i / 2048) * per + (i % 2048)] = z;
    }
    if (a.mode == 1) {
        float4 *xp = (float4 *)(a.ws + a.xp_off);
        const long tot = (long)a.NGT * a.T * 320;
        for (long i = tid; i < tot; i += n) {
            const int clip = i & 15, kq = (i >> 4) % 20;
            const long gt = i / 320, gg = gt / a.T, t = gt - gg * a.T;
            const long b = gg * 16 + clip;
            float4 v = z;
            if (b < a.B) {
                const float *s = x + (b * a.T + t) * a.KX + 4 * kq;
                if (4 * kq + 0 < a.KX) v.x = s[0];
                if (4 * kq + 1 < a.KX) v.y = s[1];
                if (4 * kq + 2 < a.KX) v.z = s[2];
                if (4 * kq + 3 < a.KX) v.w = s[3];
            }
            xp[i] = v;
        }
    }
}

// bounded wait until every lane's flag (sc1 load at flags_soff + fvoff) has reached the lane's `need`; false = abort (wave-uniform)
__device__ __forceinline__ bool st_wait_flags(__amdgpu_buffer_rsrc_t rws, unsigned fvoff, unsigned flags_soff, unsigned need,
                                              unsigned status_soff, int phase)
{
    const unsigned lane = threadIdx.x & 63;
    long long t0 = 0;
    for (unsigned spins = 1;; ++spins) {
        __builtin_amdgcn_s_sleep(4);
        if (__all(__builtin_amdgcn_raw_buffer_load_b32(rws, fvoff, flags_soff, 16) >= need)) return true;
        if ((spins & 63u) == 0) {
            if (__builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(rws, 0, status_soff, 16)) != 0u) return false;
            const long long now = (long long)wall_clock64();
            if (spins == 64u) t0 = now;
            if (now - t0 > XCD_SPIN_LIMIT) {
                if (lane == 0) {
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned)blockIdx.x, rws, 4, status_soff, 16);
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned)phase, rws, 8, status_soff, 16);
                    __builtin_amdgcn_raw_buffer_store_b32(1u, rws, 0, status_soff, 16);
                }
                return false;
            }
        }
    }
}

// Yielding (opnet_xcd_kernels.hip "Yielding"): fp32 MFMA runs on the SIMD's fp32 lanes, so the finish wave beside a product wave
// only gets VALU cycles the product wave GIVES it - an issue gap of 36 cycles (XCD_GAP) after the MFMAs of the first YM rows of a
// phase and after the first MFMA of every ST_TAIL-th row of the rest.  YM by groups per XCD: -1 (no gap anywhere) for one group -
// the finish runs while this wave waits for the exchange anyway.
// Measured (tools/seqt_gap_ab.sh, kernel ms at 256 / 384 / 512 clips for MODE 1 and 128 / 192 / 256 for MODE 2): no gaps at all 2.21 / 3.36 / 4.18
// and 3.05 / 4.35 / 5.59 against 1.71 / 2.49 / 3.24 and 2.15 / 2.94 / 3.87 with them; a gap on every 3rd row of the tail is ~1 % ahead
// of every 2nd for MODE 1 (1.69 / 2.46 / 3.20) and behind for MODE 2; 16 leading gapped rows instead of 8 with two groups help MODE 2
// (2.08 at 128 clips) and not MODE 1; leading gaps from three groups on help nobody.
#ifndef ST_TAIL
#define ST_TAIL (MODE == 1 ? 3 : 2)
#endif
#ifndef ST_NY2
#define ST_NY2 (MODE == 1 ? 8 : 16)   // two groups per XCD: the exchange is on the critical path, the finish must be quick
#endif
#ifndef ST_NY3
#define ST_NY3 0
#endif
#define ST_GAPPED(ROW) ((ROW) < YM || (ST_TAIL > 0 && YM >= 0 && ((ROW) % (ST_TAIL > 0 ? ST_TAIL : 1)) == 0))
#define ST_MFMA0(acc, av, bv) do { if (ROW < YM) XCD_MFMA0_G(acc, av, bv, XCD_GAP); else XCD_MFMA0_G(acc, av, bv, 0); } while (0)
#define ST_MFMA(acc, av, bv) do { if (ROW < YM) XCD_MFMA_G(acc, av, bv, XCD_GAP); else XCD_MFMA_G(acc, av, bv, 0); } while (0)
#define ST_MFMA_LEAD(acc, av, bv) do { if (ST_GAPPED(ROW)) XCD_MFMA_G(acc, av, bv, XCD_GAP); else XCD_MFMA_G(acc, av, bv, 0); } while (0)

#define ST_CF_NG4 1u
#define ST_CF_NG2 2u
#define ST_CF_NG1 4u
#define ST_CF_NOPUB 8u
#define ST_CF_LOCAL 16u
#define ST_CF_TRACE 32u

template <int MODE>
__global__ void __launch_bounds__(512, 2) seqt_forward(const SeqTArgs a)
{
    constexpr int NH = MODE == 1 ? ST_NH1 : ST_NH2;              // hexadecets per tile = 1-KB chunks of a gather buffer (role A: 32)
    __shared__ __attribute__((aligned(1024))) float4 sbuf[2][NH * 64];
    __shared__ __attribute__((aligned(16))) float4 sHAND[2][4][128];  // per product wave: gates | role A: P1
    __shared__ __attribute__((aligned(16))) float sTR[4][64];         // (clip, unit) -> float4-per-clip transposes
    __shared__ float sC[ST_NGMAX][4][64];
    __shared__ int sAbort, sLocal;
    __shared__ unsigned sArrive[2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv & 3;
    const int x = blockIdx.x & (XCD_COUNT - 1), c = blockIdx.x >> 3;
    const int T = a.T;
    const int role = MODE == 1 ? 0 : (x & 1);
    int g0, ng;
    if (MODE == 1) st_groups(a.NGT, 8, x, &g0, &ng);
    else st_groups(a.NGT, 4, x >> 1, &g0, &ng);
    if (ng > ST_NGMAX) ng = ST_NGMAX;       // the host never asks for more
    const int n = lane & 15, u = lane >> 4;
    const int t2 = 4 * c + w;
    const int nsteps = (MODE == 2 && role == 0) ? T + 1 : T;
    const int nph = nsteps * ng;

    if (wv == XCD_FW0) {
        // placement check by the first finish wave (XCDs with no work still publish their id and leave)
        const int loc = ng > 0 ? xcd_group_is_local(a.status, x) : 0;
        if (ng == 0) {
            if (lane == 0) __hip_atomic_store(a.status + 8 + blockIdx.x, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lane == 0) {
            XCD_LDS_ST(sLocal, loc > 0 && a.force_safe == 0);
            XCD_LDS_ST(sAbort, loc < 0);
            sArrive[0] = 0u; sArrive[1] = 0u;
            if (loc == 0 && c == 0) atomicAdd(a.status + 3, 1u);
        }
    }
    if (ng == 0) return;
    for (int i = tid; i < ST_NGMAX * 4 * 64; i += 512) (&sC[0][0][0])[i] = 0.f;

    if ((wv >= 4) == (XCD_FW0 == 0)) {
        // =========================================== product waves ===================================================
        float4 A[NH];
        {
            const float4 *pa = (const float4 *)a.pk + ((size_t)(role * 128 + t2) * NH) * 64 + lane;
#pragma unroll
            for (int q = 0; q < NH; ++q) A[q] = pa[q * 64];
        }
        const bool mtracer = a.trace && blockIdx.x < 2 && tid == (4 - XCD_FW0) * 64;
        unsigned long long *const tr = a.trace ? a.trace + (size_t)blockIdx.x * (size_t)(T + 1) * ST_NGMAX * 8 : nullptr;
        __syncthreads();                        // phase 0's gather has landed
        if (XCD_LDS_LD(sAbort)) return;
        for (int p = 0; p < nph; ++p) {
            const float4 *F = &sbuf[p & 1][0] + lane;
            if (mtracer) tr[(long)p * 8 + 0] = clock64();
            f32x4 ga, gb, pa_ = {0.f, 0.f, 0.f, 0.f}, pb_ = {0.f, 0.f, 0.f, 0.f};
            // NHM hexadecets on two chains (even / odd), B fragments by ds_read_b128 one pair ahead; role A (AUX): the first 16
            // fragments also feed the P1 chains.  The issue order is pinned with sched_barrier after every row of independent
            // MFMAs (left alone, the scheduler clusters a hexadecet's four MFMAs on one accumulator: 40-cycle dependent latency
            // against a 32-cycle issue)
            auto products = [&](auto ym, auto aux) {
                constexpr int YM = decltype(ym)::value;
                constexpr bool AUX = decltype(aux)::value;
                constexpr int NHM = AUX ? 32 : NH, NP = NHM / 2;
                float4 fa[2], fb[2];
                fa[0] = F[0]; fb[0] = F[64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const int cur = j & 1, nxt = cur ^ 1;
                    const bool ax = AUX && j < 8;
                    int ROW = 4 * j;
                    if (j == 0) {
                        ST_MFMA0(ga, A[0].x, fa[cur].x); ST_MFMA0(gb, A[1].x, fb[cur].x);
                        if (ax) { ST_MFMA0(pa_, A[AUX ? 32 : 0].x, fa[cur].x); ST_MFMA0(pb_, A[AUX ? 33 : 0].x, fb[cur].x); }
                    } else {
                        ST_MFMA_LEAD(ga, A[2 * j].x, fa[cur].x); ST_MFMA(gb, A[2 * j + 1].x, fb[cur].x);
                        if (ax) { ST_MFMA(pa_, A[AUX ? 32 + 2 * j : 0].x, fa[cur].x); ST_MFMA(pb_, A[AUX ? 33 + 2 * j : 0].x, fb[cur].x); }
                    }
                    if (2 * j + 2 < NHM) fa[nxt] = F[(2 * j + 2) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    ROW = 4 * j + 1;
                    ST_MFMA_LEAD(ga, A[2 * j].y, fa[cur].y); ST_MFMA(gb, A[2 * j + 1].y, fb[cur].y);
                    if (ax) { ST_MFMA(pa_, A[AUX ? 32 + 2 * j : 0].y, fa[cur].y); ST_MFMA(pb_, A[AUX ? 33 + 2 * j : 0].y, fb[cur].y); }
                    if (2 * j + 3 < NHM) fb[nxt] = F[(2 * j + 3) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    ROW = 4 * j + 2;
                    ST_MFMA_LEAD(ga, A[2 * j].z, fa[cur].z); ST_MFMA(gb, A[2 * j + 1].z, fb[cur].z);
                    if (ax) { ST_MFMA(pa_, A[AUX ? 32 + 2 * j : 0].z, fa[cur].z); ST_MFMA(pb_, A[AUX ? 33 + 2 * j : 0].z, fb[cur].z); }
                    __builtin_amdgcn_sched_barrier(0);
                    ROW = 4 * j + 3;
                    ST_MFMA_LEAD(ga, A[2 * j].w, fa[cur].w); ST_MFMA(gb, A[2 * j + 1].w, fb[cur].w);
                    if (ax) { ST_MFMA(pa_, A[AUX ? 32 + 2 * j : 0].w, fa[cur].w); ST_MFMA(pb_, A[AUX ? 33 + 2 * j : 0].w, fb[cur].w); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (NHM & 1) {                  // the odd last hexadecet (MODE 1: 37): on the even chain alone
                    constexpr int q = NHM - 1;
                    const int ROW = 4 * NP;
                    const float4 fl = fa[NP & 1];
                    ST_MFMA_LEAD(ga, A[q].x, fl.x); ST_MFMA(ga, A[q].y, fl.y); ST_MFMA(ga, A[q].z, fl.z); ST_MFMA(ga, A[q].w, fl.w);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            const std::integral_constant<bool, MODE == 2> auxT{};
            const std::false_type auxF{};
            if (MODE == 2 && role == 0) {
                if (ng == 1) products(std::integral_constant<int, -1>{}, auxT);
                else if (ng == 2) products(std::integral_constant<int, ST_NY2>{}, auxT);
                else products(std::integral_constant<int, ST_NY3>{}, auxT);
            } else {
                if (ng == 1) products(std::integral_constant<int, -1>{}, auxF);
                else if (ng == 2) products(std::integral_constant<int, ST_NY2>{}, auxF);
                else products(std::integral_constant<int, ST_NY3>{}, auxF);
            }
            XCD_MFMA_DRAIN(ga, gb, pa_, pb_);
            if (mtracer) tr[(long)p * 8 + 1] = clock64();
            float4 *hd_ = &sHAND[p & 1][w][0] + lane;
            hd_[0] = make_float4(ga[0] + gb[0], ga[1] + gb[1], ga[2] + gb[2], ga[3] + gb[3]);
            if (MODE == 2 && role == 0) hd_[64] = make_float4(pa_[0] + pb_[0], pa_[1] + pb_[1], pa_[2] + pb_[2], pa_[3] + pb_[3]);
            // barrier p (no look at the abort word: when the finish waves leave on an abort the barrier only counts the waves still
            // alive, and this wave runs its remaining phases on whatever is in LDS - it stores nothing to memory - and ends)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ng == 1) {                      // exposed exchange: wait for this phase's finish + the next gather
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        return;
    }

    // ============================================== finish waves ======================================================
    // every VALU instruction of this wave takes its cycles from the product wave of the same SIMD: one buffer descriptor per
    // region with 32-bit offsets (wave-uniform part on the scalar unit), nothing spilled (opnet_xcd_kernels.hip, finish waves)
    const unsigned NS = (unsigned)(T + 1);
    const unsigned lds0 = (unsigned)(unsigned long long)(const void *)&sbuf[0][0];
    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void *)a.ws, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)a.G, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)a.P1, 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16;
    // flags this lane polls: own layer's CU (lane & 31); role B: lanes 32..63 poll role A's written-through flags, two steps further
    const unsigned fl_own = (MODE == 2 && role == 1) ? 2u : 0u;
    const unsigned flag_voff = ((MODE == 2 && role == 1 && lane >= 32) ? 32u + (lane & 31) : fl_own * 32u + (lane & 31)) * 4;
    const unsigned need_add = (MODE == 2 && role == 1 && lane >= 32) ? 2u : 0u;
    bool alive = true;

    // this wave's share (chunks w, w + 4, ...) of the gather of phase (group gi, step s) into LDS buffer `buf`:
    //   MODE 1: x[s] (5 chunks) | h[s-1] (slot s, 32);  role A: h0[s-1] (slot s, 32);  role B: h0[s][256:512] (hc slot s + 1, 16) | h1[s-1] (slot s, 32)
    auto gather = [&](int gi, int s, int buf, int w) {
        const unsigned gg = g0 + gi;
        const unsigned dst = lds0 + (unsigned)buf * (NH * 1024) + w * 1024;
        unsigned o0, o1;
        int n0;
        if (MODE == 1) {
            n0 = ST_NXH1;
            o0 = a.xp_off + (gg * (unsigned)T + (unsigned)s) * (20 * 256);
            o1 = a.hl_off[0] + (gg * NS + (unsigned)s) * (128 * 256);
        } else if (role == 0) {
            n0 = 0; o0 = 0;
            o1 = a.hl_off[0] + (gg * NS + (unsigned)s) * (128 * 256);
        } else {
            n0 = 16;
            o0 = a.hc_off + (gg * NS + (unsigned)s + 1) * (64 * 256);
            o1 = a.hl_off[1] + (gg * NS + (unsigned)s) * (128 * 256);
        }
        const int nch = n0 + 32;
        o0 += w * 1024;
        o1 += (w - n0) * 1024;
#pragma unroll
        for (int j = 0; j < (NH + 3) / 4; ++j) {
            const int ch = 4 * j + w;           // wave-uniform
            if (ch < n0) xcd_glds16(rws, lane16, o0 + j * 4096, dst + j * 4096);
            else if (ch < nch) xcd_glds16(rws, lane16, o1 + j * 4096, dst + j * 4096);
        }
    };
    auto flags_ready = [&](int gn, unsigned need) -> bool {
        const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rws, flag_voff, a.flags_off + (g0 + gn) * (96 * 4), 16);   // sc1
        return __all(v >= need + need_add);
    };
    // wait until the inputs of phase (gn, sn) are published, then gather it
    auto poll_gather = [&](int gn, int sn, int buf, int phase, int w) {
        if (!alive) return;
        const bool waits = (MODE == 2 && role == 1) || sn > 0;
        if (waits && !flags_ready(gn, (unsigned)sn))
            alive = st_wait_flags(rws, flag_voff, a.flags_off + (g0 + gn) * (96 * 4), (unsigned)sn + need_add, a.status_off, phase);
        if (alive) gather(gn, sn, buf, w);
        else XCD_LDS_ST(sAbort, 1);
    };

    poll_gather(0, 0, 0, 0, w);                 // (role B waits for layer 0's first two steps here)
    if (ng >= 2 && nph > 1) poll_gather(1, 0, 1, 0, w);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (XCD_LDS_LD(sAbort)) return;
    const unsigned cfbits = (ng >= 4 ? ST_CF_NG4 : 0u) | (ng >= 2 ? ST_CF_NG2 : 0u) | (ng == 1 ? ST_CF_NG1 : 0u)
                            | ((a.debug & 8) ? ST_CF_NOPUB : 0u)
                            | (__builtin_amdgcn_readfirstlane(XCD_LDS_LD(sLocal)) != 0 ? ST_CF_LOCAL : 0u)
                            | ((a.trace && blockIdx.x < 2 && wv == XCD_FW0) ? ST_CF_TRACE : 0u);
    unsigned long long *const tr = a.trace ? a.trace + (size_t)blockIdx.x * (size_t)(T + 1) * ST_NGMAX * 8 : nullptr;

    int gi = 0, s = 0;                          // phase fp = s * ng + gi
    for (int fp = 0; fp < nph; ++fp) {
        unsigned cf = __builtin_amdgcn_readfirstlane(cfbits);
        int wq = w;
        asm volatile("" : "+s"(cf), "+s"(wq));  // the loop-invariant conditions as bits of one opaque scalar (no hoisted lane masks)
        const bool local = (cf & ST_CF_LOCAL) != 0, tracer = (cf & ST_CF_TRACE) != 0 && lane == 0;
        const unsigned gg = g0 + gi;
        const int ahead = (cf & ST_CF_NG2) ? 2 : 1;
        int gn = gi + ahead, sn = s;
        while (gn >= ng) { gn -= ng; ++sn; }
        // what the cell adds to the MFMA sums, asked for before the barrier (which must therefore not drain vmcnt):
        // role A: G[clip][s][unit's 4 gates] (read-only, written before the launch); role B: P1[s] of this tile (published with the
        // flags the gather of this phase waited for; sc1: never this CU's L1)
        xcd_u32x4 addv = {0u, 0u, 0u, 0u};
        const bool cell = (MODE == 2 && role == 0) ? s < T : true;
        if (MODE == 2 && role == 0 && cell) {
            long clip = (long)gg * 16 + n;
            clip = clip < a.B ? clip : a.B - 1;
            addv = __builtin_amdgcn_raw_buffer_load_b128(rg, (unsigned)(((clip * T + s) * 2048 + 4 * (4 * t2 + u)) * 4), 0, 0);
        } else if (MODE == 2 && role == 1) {
            addv = __builtin_amdgcn_raw_buffer_load_b128(rp, lane16, ((gg * (unsigned)T + (unsigned)s) * 128 + (unsigned)t2) * 1024, 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // barrier fp: the phase's accumulators are in sHAND[fp & 1]
        asm volatile("" ::: "memory");
        if (tracer) tr[(long)fp * 8 + 2] = clock64();
        if (XCD_LDS_LD(sAbort)) return;
        // four or more groups: the phase after next has usually been published more than a window ago - its gather goes first
        // and lands under the finish
        bool early = false;
        if ((cf & ST_CF_NG4) && fp + 2 < nph && alive && flags_ready(gn, (unsigned)sn)) {
            gather(gn, sn, fp & 1, wq);
            early = true;
        }
        if (tracer) tr[(long)fp * 8 + 3] = clock64();
        const float4 *H = &sHAND[fp & 1][w][0] + lane;
        // ---- the cell (learned_models.py:110 / 146 / 192): lane (clip n, unit 4 t2 + u) ---------------------------------------
        if (cell && alive) {
            const float4 g4 = H[0];
            float cc = sC[gi][w][lane];
            const float h = lstm_cell(g4.x + __uint_as_float(addv.x), g4.y + __uint_as_float(addv.y), g4.z + __uint_as_float(addv.z),
                                      g4.w + __uint_as_float(addv.w), &cc);
            sC[gi][w][lane] = cc;
            sTR[w][n * 4 + u] = h;
            XCD_WAVE_LDS_SYNC();
            if (lane < 16) {
                const float4 hv = *(const float4 *)&sTR[w][lane * 4];
                const unsigned hl = (MODE == 2 && role == 1) ? a.hl_off[1] : a.hl_off[0];
                xcd_store16(rws, lane16, hl + ((gg * NS + (unsigned)s + 1) * 128 + (unsigned)t2) * 256, hv, local);
                if (MODE == 2 && role == 0 && t2 >= 64)
                    xcd_store16(rws, lane16, a.hc_off + ((gg * NS + (unsigned)s + 1) * 64 + (unsigned)(t2 - 64)) * 256, hv, false);
            }
        }
        // ---- role A: this tile's partial of layer 1's gates of step s - 1, written through ---------------------------------
        if (MODE == 2 && role == 0 && s >= 1 && alive) {
            const float4 p4 = H[64];
            xcd_u32x4 pv;
            pv.x = __float_as_uint(p4.x); pv.y = __float_as_uint(p4.y); pv.z = __float_as_uint(p4.z); pv.w = __float_as_uint(p4.w);
            __builtin_amdgcn_raw_buffer_store_b128(pv, rp, lane16, ((gg * (unsigned)T + (unsigned)(s - 1)) * 128 + (unsigned)t2) * 1024, 16);
        }
        if (tracer) tr[(long)fp * 8 + 4] = clock64();
        // ---- publish: every finish wave drains its stores (and DMA) and arrives at an LDS counter; the last one stores the CU's flag(s)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (alive && !(cf & ST_CF_NOPUB)) {
            const bool last = (xcd_lds_add_lane0((unsigned)(unsigned long long)(const void *)&sArrive[fp & 1], 1u) & 3u) == 3u;
            if (last && lane == 0) {
                const unsigned fo = a.flags_off + ((gg * 3 + fl_own) * 32 + (unsigned)c) * 4;
                if (local) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(s + 1), rws, 0, fo, 0);
                else __builtin_amdgcn_raw_buffer_store_b32((unsigned)(s + 1), rws, 0, fo, 16);
                if (MODE == 2 && role == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(s + 1), rws, 0, fo + 128, 16);
            }
        }
        if (tracer) tr[(long)fp * 8 + 5] = clock64();
        if (!early && fp + ahead < nph) {
            poll_gather(gn, sn, (fp + ahead) & 1, fp, wq);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (tracer) tr[(long)fp * 8 + 6] = clock64();
        if (cf & ST_CF_NG1) {
            __syncthreads();
            if (XCD_LDS_LD(sAbort)) return;
        }
        if (++gi == ng) { gi = 0; ++s; }
    }
}

// y[b][t][0..3] = predictions_layer.weight . h_top[t][b] (learned_models.py:113 / 148 / 195) from the top layer's history; one
// workgroup per (t, group): thread (r, n) walks k-quads r, r + 16, ... of clip n, the 16 partials are summed in fixed order.
// An aborted launch (status[0] != 0) poisons y with NaN.
__global__ void __launch_bounds__(256) seqt_out_head(const SeqTArgs a)
{
    __shared__ float sw[4][ST_H];
    __shared__ __attribute__((aligned(16))) float4 red[16][16];
    const int t = blockIdx.x, gg = blockIdx.y, tid = threadIdx.x, T = a.T;
    for (int i = tid; i < 4 * ST_H; i += 256) (&sw[0][0])[i] = a.whead[i];
    __syncthreads();
    const int r = tid >> 4, n = tid & 15;
    const float4 *h = (const float4 *)(a.ws + a.hl_off[a.mode == 1 ? 0 : 1]) + ((long)gg * (T + 1) + t + 1) * 2048;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kq = r; kq < ST_H / 4; kq += 16) {
        const float4 hv = h[kq * 16 + n];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            acc[o] = fmaf(sw[o][4 * kq + 0], hv.x, acc[o]);
            acc[o] = fmaf(sw[o][4 * kq + 1], hv.y, acc[o]);
            acc[o] = fmaf(sw[o][4 * kq + 2], hv.z, acc[o]);
            acc[o] = fmaf(sw[o][4 * kq + 3], hv.w, acc[o]);
        }
    }
    red[r][n] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (tid < 16) {
        float4 sum = red[0][tid];
        for (int i = 1; i < 16; ++i) {
            const float4 v = red[i][tid];
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        const long b = (long)gg * 16 + tid;
        if (a.status[0] != 0u) sum = make_float4(NAN, NAN, NAN, NAN);
        if (b < a.B) a.y[b * T + t] = sum;
    }
}
