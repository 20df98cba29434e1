// opnet_xcd_kernels.hip - OPNet forward as ONE persistent launch: eight independent forwards, one per XCD,
// every weight resident in registers for all T steps.
//
// What is computed: reference baselines/learned_models.py:35-52 (the same function as opnet_kernels.hip).
//
// Why: the launch-per-step chain re-fetches the 5.68 MB weight set 303 times per forward and pays a
// dependent kernel boundary per step; a chip-wide persistent kernel pays a chip-wide exchange per step instead and lost.
// Here the chip is cut along its XCDs.  One XCD = 32 CUs = 128 SIMDs, and 128 one-wave-per-SIMD register files hold the
// whole weight set:
//     wave (CU c, SIMD w)   LSTM2 tile 4c+w   : 16 gate rows (4 units x i,f,g,o) x K = 512        128 VGPRs
//                           LSTM1 tile 2c+w/2 : 16 gate rows x half of K = 96 + 256 (w&1 picks)     44 VGPRs
//                           selection head    : 16 (15) rows x K quarter w of 256                   16 VGPRs
//                           W_ih2 rows of the lane's own unit (6 -> 8 wide)                         32 VGPRs
// Each XCD runs ITS OWN clips (groups of 16 = one MFMA column block), so nothing crosses an XCD boundary on the data
// path and the only exchange is h1 / h2 between the 32 CUs of one XCD, once per (group, step):
//     phase (group g, step s):  LSTM1 step s | selection head + einsum + LSTM2 step s-1        (T+1 steps per forward)
//   * barrier; the phase's activations x[s], x[s-1], h1[s-1], h2[s-2] (60 KB, "kq-major" [k/4][16 clips][4] so that
//     1 KB = one MFMA B fragment set of a 16-k step) are in LDS, gathered by LDS-DMA during the PREVIOUS phase;
//   * 188 MFMAs a wave (v_mfma_f32_16x16x4_f32, exact fp32) on 4 independent accumulator chains, B operands by
//     ds_read_b128, A operands = the resident registers;
//   * half way through, the wave polls the 32 per-CU flags of the NEXT phase's group (published one phase ago) and
//     issues its share of that phase's gather (15 x buffer_load_dwordx4 ... sc1 lds) under the remaining MFMAs:
//     with >= 2 groups per XCD the exchange latency hides under the other group's compute;
//   * LDS: K-split partials of LSTM1 (2 waves) and of the head (4 waves), barrier, then every wave finishes the head
//     redundantly (softmax, einsum - every CU needs frames_boxes), its LSTM2 cell and (odd waves) its LSTM1 cell;
//   * h is published with write-through (sc1) 16-byte stores into FULL-HISTORY buffers (slot t+1 = step t; every word
//     is written once per launch and only read after its flag), then vmcnt(0), barrier, ONE flag store per CU
//     (cdna_hip_programming.md Guideline 16 recipe R1; the consumer side reads with sc1 loads, so no acquire fence and
//     nothing depends on which XCD a workgroup landed on - placement is for speed only: block b runs on XCD b % 8).
// y = W_out h2 is not on the recurrence: it is computed from the h2 history by opnet_xcd_out_head afterwards.
//
// Every spin is bounded (XCD_SPIN_LIMIT cycles): a workgroup that cannot see its producers raises the abort word, every
// other poller sees it and leaves, and opnet_xcd_out_head poisons y with NaN - nothing can hang the device.
//
// Summation order (differs from the launch chain's K-split, agrees to rounding; tests hold both to the oracle):
//   LSTM2 gate = (sum over even 16-k steps) + (sum over odd 16-k steps) + x part; LSTM1 gate = lower K half + upper K
//   half; logits = ((w0 + w1) + w2) + w3 over K quarters; each partial an ascending-k fmaf chain per MFMA lane group.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opnet_ctx.h"

#define XCD_COUNT 8
#define XCD_CUS 32
#define XCD_NGMAX 8            // 16-clip groups one XCD carries in one launch (8 x 16 x 8 XCDs = 1024 clips)
#define XCD_H1 256
#define XCD_H2 512
#define XCD_SPIN_LIMIT (3ll << 30)   // shader cycles (~1.5 s) before a poller gives up
#ifndef XCD_PF
#define XCD_PF 5               // LSTM2 hexadecet pair after which the next phase's gather is issued
#endif

// LDS gather buffer of one phase, in float4 units: X0 = x[s] | X1 = x[s-1] | H1 = h1[s-1] | H2 = h2[s-2]
#define XB_X0 0
#define XB_X1 (6 * 64)
#define XB_H1 (12 * 64)
#define XB_H2 (28 * 64)
#define XB_F4 (60 * 64)        // 60 KB
#define XB_CHUNKS 60

struct XcdArgs {
    int B, T, NGT;             // clips, frames, 16-clip groups = ceil(B / 16)
    const float *packed;       // opnet_pack_weights_f32 image for H1 = 256, H2 = 512
    const float4 *xp;          // [NGT][T+2][24][16]   slot t+1 = x[t]; slots 0 and T+1 zero
    float4 *h1h;               // [NGT][T+1][64][16]   slot t+1 = h1[t]; slot 0 zero
    float4 *h2h;               // [NGT][T+1][128][16]  slot t+1 = h2[t]; slot 0 zero
    unsigned *flags;           // [NGT][32]            steps published by CU c of the group's XCD
    unsigned *status;          // [0] abort code (0 = ok), [1] first failing block, [2] phase, [8 + b] XCC_ID of block b
    float *logits;             // caller's [B][15][T]
    unsigned long long *trace; // optional [phases][4] s_memtime stamps of block 0 wave 0 (tools), or null
};

__host__ __device__ inline void xcd_groups(int NGT, int x, int *g0, int *ng)
{
    const int base = NGT / XCD_COUNT, rem = NGT % XCD_COUNT;
    *ng = base + (x < rem ? 1 : 0);
    *g0 = x * base + (x < rem ? x : rem);
}

typedef unsigned xcd_u32x4 __attribute__((ext_vector_type(4)));

// one 1-KiB LDS-DMA piece: lane l -> 16 B from (rsrc base + soff + 16 l) to LDS byte address lds_dst + 16 l.
// sc1: served by the L2 / fabric, never by this CU's L1 (which other CUs' stores do not refresh).
__device__ __forceinline__ void xcd_glds16(xcd_u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ xcd_u32x4 xcd_rsrc(const void *base)
{
    const unsigned long long b = (unsigned long long)base;
    xcd_u32x4 r;
    r.x = (unsigned)b; r.y = (unsigned)(b >> 32); r.z = 0x7fffffffu; r.w = 0x00020000u;
    return r;
}

// 16-byte write-through store (aux 16 = sc1), counted by the compiler's vmcnt bookkeeping
__device__ __forceinline__ void xcd_store16_sc1(float4 *p, float4 v)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 16, 0x00020000);
    xcd_u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, 0, 0, 16);
}

// boxes [B][T][90] -> xp; zero slot 0 / T+1 of xp, slot 0 of the histories, the flags and the status words.
// grid (T + 2, NGT), 384 threads (24 k-quads x 16 clips)
__global__ void __launch_bounds__(384) opnet_xcd_pack_input(const float *__restrict__ boxes, XcdArgs a)
{
    const int slot = blockIdx.x, gg = blockIdx.y;
    const int T = a.T, tid = threadIdx.x;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int kq = tid % OPNET_KXQ, clip = tid / OPNET_KXQ;   // consecutive threads walk k within a clip row
        const int b = gg * 16 + clip, t = slot - 1;
        float4 v = z;
        if (b < a.B && t >= 0 && t < T) {
            const float2 *src = (const float2 *)(boxes + ((long)b * T + t) * OPNET_KX + kq * 4);   // rows are 360 B apart
            const int k = kq * 4;
            if (k + 1 < OPNET_KX) { float2 p = src[0]; v.x = p.x; v.y = p.y; }
            if (k + 3 < OPNET_KX) { float2 q = src[1]; v.z = q.x; v.w = q.y; }
        }
        ((float4 *)a.xp)[(((long)gg * (T + 2) + slot) * OPNET_KXQ + kq) * 16 + clip] = v;
    }
    if (slot == 0) {
        float4 *h1 = a.h1h + (long)gg * (T + 1) * (XCD_H1 * 4);
        float4 *h2 = a.h2h + (long)gg * (T + 1) * (XCD_H2 * 4);
        for (int i = tid; i < XCD_H1 * 4; i += 384) h1[i] = z;
        for (int i = tid; i < XCD_H2 * 4; i += 384) h2[i] = z;
        if (tid < XCD_CUS) a.flags[gg * XCD_CUS + tid] = 0u;
        if (gg == 0 && tid < 8) a.status[tid] = 0u;
    }
}

// bounded wait until all 32 CUs of the group have published `need` steps; false = abort (wave-uniform)
__device__ __forceinline__ bool xcd_wait_flags(const unsigned *flags, unsigned need, unsigned *status, int phase)
{
    const int lane = threadIdx.x & 63;
    const unsigned *f = flags + (lane & 31);
    unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__all(v >= need)) return true;
    const long long t0 = clock64();
    for (unsigned spins = 1;; ++spins) {
        __builtin_amdgcn_s_sleep(2);
        v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= need)) return true;
        if ((spins & 63u) == 0) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (clock64() - t0 > XCD_SPIN_LIMIT) {
                if (lane == 0) {
                    __hip_atomic_store(status + 1, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(status + 2, (unsigned)phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return false;
            }
        }
    }
}

// wave-private LDS scratch written by some lanes and read by others of the SAME wave: the LDS queue is in order per
// wave, so only the compiler has to be kept from moving the read above the write
#define XCD_WAVE_LDS_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
#define XCD_MFMA(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0)

// PF = the LSTM2 hexadecet pair after which the next phase's gather is issued (tuning knob)
template <int PF>
__global__ void __launch_bounds__(256, 1) opnet_xcd_forward(const XcdArgs a)
{
    __shared__ __attribute__((aligned(1024))) float4 sbuf[2][XB_F4];
    __shared__ __attribute__((aligned(16))) float4 sPH[4][64];       // head partials (K quarters)
    __shared__ __attribute__((aligned(16))) float4 sP1[2][64];       // LSTM1 partial of the lower-K wave of each pair
    __shared__ __attribute__((aligned(16))) float sSP[4][16][16];    // slot probabilities, wave-private
    __shared__ __attribute__((aligned(16))) float sFB[4][16][8];     // frames_boxes (6 -> 8), wave-private
    __shared__ __attribute__((aligned(16))) float sTR[4][2][64];     // (clip, unit) -> float4-per-clip transposes
    __shared__ float sC2[XCD_NGMAX][4][64];
    __shared__ float sC1[XCD_NGMAX][2][64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = blockIdx.x & (XCD_COUNT - 1), c = blockIdx.x >> 3;
    const int T = a.T;
    int g0, ng;
    xcd_groups(a.NGT, x, &g0, &ng);
    if (tid == 0) a.status[8 + blockIdx.x] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf;  // HW_REG_XCC_ID[3:0]
    if (ng == 0) return;
    if (ng > XCD_NGMAX) ng = XCD_NGMAX;   // the host never asks for more

    const int n = lane & 15, u = lane >> 4;
    const int t2 = 4 * c + w;                 // LSTM2 tile
    const int t1 = 2 * c + (w >> 1), kh = w & 1;

    // ---- resident weights ------------------------------------------------------------------------------------------
    const PackedLayout P = packed_layout(XCD_H1, XCD_H2);
    float4 a2[32], a1[11], as_[4], wx[8];
    {
        const float4 *p2 = (const float4 *)(a.packed + P.w2p) + (long)t2 * 32 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 32; ++q) a2[q] = p2[q * 64];
        const float4 *p1 = (const float4 *)(a.packed + P.w1p) + ((long)t1 * 22 + 11 * kh) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 11; ++q) a1[q] = p1[q * 64];
        const float4 *ps = (const float4 *)(a.packed + P.wselp) + (4 * w) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) as_[q] = ps[q * 64];
        const float4 *px = (const float4 *)(a.packed + P.wih2p) + (long)(4 * t2 + u) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) wx[q] = px[q];
    }
    for (int i = tid; i < XCD_NGMAX * 4 * 64; i += 256) (&sC2[0][0][0])[i] = 0.f;
    for (int i = tid; i < XCD_NGMAX * 2 * 64; i += 256) (&sC1[0][0][0])[i] = 0.f;

    const unsigned lds0 = (unsigned)(unsigned long long)(const void *)&sbuf[0][0];
    const xcd_u32x4 rx = xcd_rsrc(a.xp), rh1 = xcd_rsrc(a.h1h), rh2 = xcd_rsrc(a.h2h);

    // this wave's share (chunks w, w+4, ...) of the gather of phase (group gi, step s) into LDS buffer `buf`
    auto gather = [&](int gi, int s, int buf) {
        const long gg = g0 + gi;
        const unsigned dst = lds0 + (unsigned)buf * (XB_F4 * 16);
        const unsigned ox0 = (unsigned)(((gg * (T + 2) + s + 1) * OPNET_KXQ) * 256);
        const unsigned ox1 = (unsigned)(((gg * (T + 2) + s) * OPNET_KXQ) * 256);
        const unsigned oh1 = (unsigned)(((gg * (T + 1) + s) * (XCD_H1 / 4)) * 256);
        const unsigned oh2 = (unsigned)(((gg * (T + 1) + (s > 0 ? s - 1 : 0)) * (XCD_H2 / 4)) * 256);
#pragma unroll
        for (int j = 0; j < XB_CHUNKS / 4; ++j) {
            const int ch = 4 * j + w;          // wave-uniform; the section a chunk falls in depends on j only up to w
            if (4 * j + 3 < 6) xcd_glds16(rx, lane * 16, ox0 + ch * 1024, dst + ch * 1024);
            else if (4 * j >= 6 && 4 * j + 3 < 12) xcd_glds16(rx, lane * 16, ox1 + (ch - 6) * 1024, dst + ch * 1024);
            else if (4 * j >= 12 && 4 * j + 3 < 28) xcd_glds16(rh1, lane * 16, oh1 + (ch - 12) * 1024, dst + ch * 1024);
            else if (4 * j >= 28) xcd_glds16(rh2, lane * 16, oh2 + (ch - 28) * 1024, dst + ch * 1024);
            else {                              // a j whose four chunks straddle two sections
                if (ch < 6) xcd_glds16(rx, lane * 16, ox0 + ch * 1024, dst + ch * 1024);
                else if (ch < 12) xcd_glds16(rx, lane * 16, ox1 + (ch - 6) * 1024, dst + ch * 1024);
                else xcd_glds16(rh1, lane * 16, oh1 + (ch - 12) * 1024, dst + ch * 1024);
            }
        }
    };

    const int nph = (T + 1) * ng;
    bool alive = true;
    gather(0, 0, 0);                        // phase 0 reads only zero slots and x[0]: nothing to wait for
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int gi = 0, s = 0;                      // phase p = s * ng + gi
    for (int p = 0; p < nph; ++p) {
        const int buf = p & 1;
        int gn = gi + 1, sn = s;            // the next phase
        if (gn == ng) { gn = 0; sn = s + 1; }
        const float4 *F = &sbuf[buf][0] + lane;
        // B fragment of LSTM1 hexadecet 11 kh + j of [x 0..5 | h1 6..21]: X0 and H1 are 6 KB apart in the buffer (X1 sits
        // between them), so the lower-K wave reads F[j] for j < 6 and F[6 + j] from there on, the upper-K wave F[17 + j]
        const float4 *FLa = F + (kh ? 17 * 64 : 0), *FLb = F + (kh ? 17 * 64 : 6 * 64);
        const float4 *FH = F + XB_H1 + 4 * w * 64;     // head: K quarter w of h1
        if (a.trace && blockIdx.x == 0 && tid == 0) a.trace[(long)p * 4 + 0] = clock64();

        // ---- products ------------------------------------------------------------------------------------------
        // 188 MFMAs on four accumulator chains (LSTM2 even / odd hexadecets, LSTM1, head).  The issue order is pinned with
        // sched_barrier after every row of independent MFMAs: left alone, the scheduler clusters the four MFMAs of one
        // hexadecet on the same accumulator (40-cycle dependent latency against a 32-cycle issue) and keeps only one or
        // two B fragments in flight.  Fragments are fetched one j-step (12 MFMAs ~ 400 cycles) ahead of their use.
        f32x4 accH = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc2a = {0.f, 0.f, 0.f, 0.f}, acc2b = {0.f, 0.f, 0.f, 0.f};
        float4 fa[2], fb[2], fl[2];
        fa[0] = F[XB_H2]; fb[0] = F[XB_H2 + 64]; fl[0] = FLa[0];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int cur = j & 1, nxt = cur ^ 1;
            const bool l1 = j < 11, hd = j >= 11 && j < 15;
            XCD_MFMA(acc2a, a2[2 * j].x, fa[cur].x); XCD_MFMA(acc2b, a2[2 * j + 1].x, fb[cur].x);
            if (l1) XCD_MFMA(acc1, a1[j].x, fl[cur].x);
            if (hd) XCD_MFMA(accH, as_[j - 11].x, fl[cur].x);
            if (j + 1 < 16) fa[nxt] = F[XB_H2 + (2 * j + 2) * 64];
            __builtin_amdgcn_sched_barrier(0);
            XCD_MFMA(acc2a, a2[2 * j].y, fa[cur].y); XCD_MFMA(acc2b, a2[2 * j + 1].y, fb[cur].y);
            if (l1) XCD_MFMA(acc1, a1[j].y, fl[cur].y);
            if (hd) XCD_MFMA(accH, as_[j - 11].y, fl[cur].y);
            if (j + 1 < 16) fb[nxt] = F[XB_H2 + (2 * j + 3) * 64];
            __builtin_amdgcn_sched_barrier(0);
            XCD_MFMA(acc2a, a2[2 * j].z, fa[cur].z); XCD_MFMA(acc2b, a2[2 * j + 1].z, fb[cur].z);
            if (l1) XCD_MFMA(acc1, a1[j].z, fl[cur].z);
            if (hd) XCD_MFMA(accH, as_[j - 11].z, fl[cur].z);
            if (j + 1 < 11) fl[nxt] = j + 1 < 6 ? FLa[(j + 1) * 64] : FLb[(j + 1) * 64];
            else if (j + 1 < 15) fl[nxt] = FH[(j + 1 - 11) * 64];
            __builtin_amdgcn_sched_barrier(0);
            XCD_MFMA(acc2a, a2[2 * j].w, fa[cur].w); XCD_MFMA(acc2b, a2[2 * j + 1].w, fb[cur].w);
            if (l1) XCD_MFMA(acc1, a1[j].w, fl[cur].w);
            if (hd) XCD_MFMA(accH, as_[j - 11].w, fl[cur].w);
            __builtin_amdgcn_sched_barrier(0);
            if (j == PF && ng > 1 && p + 1 < nph && alive) {
                // the next phase's group published its previous step one phase ago: poll, then gather under the
                // remaining MFMAs (the buffer it fills was last read in the previous phase)
                if (sn > 0) alive = xcd_wait_flags(a.flags + (g0 + gn) * XCD_CUS, (unsigned)sn, a.status, p);
                if (alive) gather(gn, sn, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (a.trace && blockIdx.x == 0 && tid == 0) a.trace[(long)p * 4 + 1] = clock64();
        sPH[w][lane] = make_float4(accH[0], accH[1], accH[2], accH[3]);
        if (!kh) sP1[w >> 1][lane] = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
        __syncthreads();
        if (!alive) return;                 // wave-uniform; the others leave at their own poll

        const long gg = g0 + gi;
        // ---- selection head of step s-1, finished by every wave (learned_models.py:40-43,50) -------------------
        float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
        if (s > 0) {
            const float4 h0 = sPH[0][lane], h1 = sPH[1][lane], h2 = sPH[2][lane], h3 = sPH[3][lane];
            float v[4] = {((h0.x + h1.x) + h2.x) + h3.x, ((h0.y + h1.y) + h2.y) + h3.y,
                          ((h0.z + h1.z) + h2.z) + h3.z, ((h0.w + h1.w) + h2.w) + h3.w};
            const long b = gg * 16 + n;
            const bool writer = (c == ((s - 1) & (XCD_CUS - 1))) && w == 0 && b < a.B;
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int slot = 4 * u + r;
                if (slot < OPNET_SLOTS_) {
                    if (writer) a.logits[(b * OPNET_SLOTS_ + slot) * T + (s - 1)] = v[r];
                    m = fmaxf(m, v[r]);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            float e[4], sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = (4 * u + r < OPNET_SLOTS_) ? __expf(v[r] - m) : 0.f;
                sum += e[r];
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            *(float4 *)&sSP[w][n][4 * u] = make_float4(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv);
            XCD_WAVE_LDS_SYNC();
            // frames_boxes[n][f] = sum_o boxes[n][s-1][o][f] * p[o], one ascending-o fmaf chain per (clip, feature)
            // (einsum "bfot,bfo->bft"); lane (n, u) carries features u and u + 4
            const float *xs = (const float *)&sbuf[buf][XB_X1];
            const float *pp = &sSP[w][n][0];
            float f0 = 0.f, f1 = 0.f;
            const int fb1 = u + 4 < OPNET_FEATS_ ? u + 4 : 0;
#pragma unroll
            for (int o = 0; o < OPNET_SLOTS_; ++o) {
                const int k0 = o * OPNET_FEATS_ + u, k1 = o * OPNET_FEATS_ + fb1;
                const float pv = pp[o];
                f0 = fmaf(xs[((k0 >> 2) * 16 + n) * 4 + (k0 & 3)], pv, f0);
                f1 = fmaf(xs[((k1 >> 2) * 16 + n) * 4 + (k1 & 3)], pv, f1);
            }
            sFB[w][n][u] = f0;
            sFB[w][n][u + 4] = u + 4 < OPNET_FEATS_ ? f1 : 0.f;
            XCD_WAVE_LDS_SYNC();
            xa = *(const float4 *)&sFB[w][n][0];
            xb = *(const float4 *)&sFB[w][n][4];
        }
        // ---- LSTM2 cell of step s-1 (learned_models.py:46): lane (clip n, unit 4 t2 + u) -----------------------
        if (s > 0) {
            float g[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 w0 = wx[2 * r], w1 = wx[2 * r + 1];
                float xsum = w0.x * xa.x;
                xsum = fmaf(w0.y, xa.y, xsum);
                xsum = fmaf(w0.z, xa.z, xsum);
                xsum = fmaf(w0.w, xa.w, xsum);
                xsum = fmaf(w1.x, xb.x, xsum);
                xsum = fmaf(w1.y, xb.y, xsum);
                g[r] = (acc2a[r] + acc2b[r]) + xsum;
            }
            float cc = sC2[gi][w][lane];
            const float h = lstm_cell(g[0], g[1], g[2], g[3], &cc);
            sC2[gi][w][lane] = cc;
            sTR[w][0][n * 4 + u] = h;
            XCD_WAVE_LDS_SYNC();
            if (lane < 16) {
                const float4 hv = *(const float4 *)&sTR[w][0][lane * 4];
                xcd_store16_sc1(a.h2h + ((gg * (T + 1) + s) * (XCD_H2 / 4) + t2) * 16 + lane, hv);
            }
        }
        // ---- LSTM1 cell of step s (learned_models.py:39), by the upper-K wave of each pair ---------------------
        if (kh && s < T) {
            const float4 lo = sP1[w >> 1][lane];
            float cc = sC1[gi][w >> 1][lane];
            const float h = lstm_cell(lo.x + acc1[0], lo.y + acc1[1], lo.z + acc1[2], lo.w + acc1[3], &cc);
            sC1[gi][w >> 1][lane] = cc;
            sTR[w][1][n * 4 + u] = h;
            XCD_WAVE_LDS_SYNC();
            if (lane < 16) {
                const float4 hv = *(const float4 *)&sTR[w][1][lane * 4];
                xcd_store16_sc1(a.h1h + ((gg * (T + 1) + s + 1) * (XCD_H1 / 4) + t1) * 16 + lane, hv);
            }
        }
        if (a.trace && blockIdx.x == 0 && tid == 0) a.trace[(long)p * 4 + 2] = clock64();
        // ---- publish: every wave drains its stores (and its share of the gather), barrier, ONE flag per CU ------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.flags + gg * XCD_CUS + c, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ng == 1 && p + 1 < nph) {
            // one group per XCD: its own next step needs what was just published - the exchange is exposed
            alive = xcd_wait_flags(a.flags + (g0 + gn) * XCD_CUS, (unsigned)sn, a.status, p);
            if (!alive) return;
            gather(gn, sn, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (a.trace && blockIdx.x == 0 && tid == 0) a.trace[(long)p * 4 + 3] = clock64();
        gi = gn; s = sn;
    }
}

// y[b][t][0..3] = W_out h2[t][b] (prediction_layer, learned_models.py:33,47) from the h2 history; one workgroup per
// (group, t): thread (r, n) walks k-quads r, r + 16, ... of clip n, the 16 partials are summed in fixed order.
// An aborted persistent launch (status[0] != 0) poisons y with NaN.
__global__ void __launch_bounds__(256) opnet_xcd_out_head(const XcdArgs a, float *__restrict__ y)
{
    __shared__ float sw[4][XCD_H2];
    __shared__ __attribute__((aligned(16))) float4 red[16][16];
    const int t = blockIdx.x, gg = blockIdx.y, tid = threadIdx.x, T = a.T;
    const PackedLayout P = packed_layout(XCD_H1, XCD_H2);
    const float *wo = a.packed + P.woutp;      // [H2/16][64][4]: lane l = row l & 15, k = 16 q + 4 (l >> 4) + e
    for (int i = tid; i < 4 * XCD_H2; i += 256) {
        const int o = i / XCD_H2, k = i % XCD_H2;
        sw[o][k] = wo[(((k >> 4) * 64) + o + 16 * ((k & 15) >> 2)) * 4 + (k & 3)];
    }
    __syncthreads();
    const int r = tid >> 4, n = tid & 15;
    const float4 *h = a.h2h + ((long)gg * (T + 1) + t + 1) * (XCD_H2 * 4);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kq = r; kq < XCD_H2 / 4; kq += 16) {
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
        if (b < a.B) ((float4 *)y)[b * T + t] = sum;
    }
}
