// opnet_xcd4_kernels.hip - the per-XCD persistent OPNet step for SMALL batches: the training step of the reference's 32-clip
// batch (forward and reverse recurrence as one launch each) and the inference forward of one small request (its shipped
// inference batch is 16).  Groups of FOUR clips, one group per XCD and row block, every weight resident in registers for all
// T steps.
//
// What is computed: reference baselines/learned_models.py:35-52 (forward) and torch autograd through it under
// training_main.py:216 (backward; restated in oracle/torch_port.py) - the same functions as opnet_kernels.hip /
// opnet_train_kernels.hip, on the launch chain's own history layouts (opnet_ctx.h StepArgs "train", BwdArgs), so that the loss,
// the weight-gradient GEMMs (opnet_wgrad) and Adam run on the result unchanged.
//
// Why a third form.  The 16-clip persistent kernel (opnet_xcd_kernels.hip) needs >= 3 groups per XCD
// (384 clips) to hide its exchange; a 32-clip batch is 2 groups - 2 XCDs busy, 5.5 us per step - and the launch chain pays
// a kernel boundary + a 5.68 MB weight fetch per step (4.9 us forward, 5.1 us backward).  A 32-clip batch cut into EIGHT
// groups of 4 clips puts every XCD to work with a quarter of the matrix work per step: v_mfma_f32_4x4x1_16b_f32 = 16
// independent 4x4 outer products per instruction (measured 9.5 cycles per instruction, tools/probes/mfma4x4_probe.hip), so
// the ~190 MFMAs of a step cost ~2 000 cycles instead of 6 000 and the step is that + the exchange latency.
//
// One workgroup = 4 waves on one CU (one per SIMD), 256 workgroups; XCD x = blockIdx.x & 7 runs clips 4x .. 4x+3 of every
// row block (group gi = row block gi), CU c = blockIdx.x >> 3 owns LSTM2 units 16c .. 16c+15 and LSTM1 units 8c .. 8c+7.
// MFMA operand layout (probe): lane = 4 b + i holds A[block b][row i], lane 4 b + j holds B[block b][col j], D[b][i][j] sits in
// lane 4 b + j, register i; the clip is always the column j.
//
// The exchange between the 32 CUs of an XCD (h forward, da backward) - "the data is the flag":
//   * ring buffers of 4 steps, [k/4 or unit][4 clips] float4, a CU's piece = whole 128-B lines; every word holds the
//     SENTINEL 0xffffffff (a NaN no cell produces) until its step is published;
//   * a producer just stores its piece (plain stores that stay in this XCD's L2 when the placement check passed, write-through
//     otherwise - as in opnet_xcd_kernels.hip) and, with it, re-arms its piece of the slot two steps on with the sentinel;
//     nothing is drained, no flag is written;
//   * a consumer loads its pieces with sc1 (L2-served) loads into registers, and any lane that still sees a sentinel word
//     makes the wave load its pieces again (bounded: XCD_SPIN_LIMIT, then the abort word and NaN in y); then ds_write.
//     (A lane would also wait for a genuine value with that bit pattern - a NaN with an all-ones payload can only come from
//     an input or weight that already holds it - and the launch would end in the same abort, NaN in y.)
//   Against payload + drain + flag + poll + gather (Guideline 16 recipe R1, the first version of this file) this takes a
//   store-acknowledge and a flag round trip out of every step: measured 0.91 -> 0.72 ms per 32-clip forward.
//   Re-arming is safe with the slot two steps ahead: a CU publishes step k only after it has gathered every CU's step k-1, i.e.
//   after every CU has finished READING step k-2 (its gather precedes its publish), and the workgroup's end-of-phase
//   s_waitcnt vmcnt(0) orders a CU's re-arm before its next publish, which every reader of the re-armed slot has to see first.
//   (LDS-DMA gathers cost ~180 cycles per 1-KB piece here - each rewrites M0 - so the gather goes through registers.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opnet_ctx.h"

#define X4_NGMAX 4             // row blocks (groups per XCD) one launch carries: B <= 128
#define X4_SENT 0xffffffffu    // "not published yet"
#define X4_SLOTS 4             // exchange ring depth (steps)
// LDS gather buffer of a forward phase, float4 units: [k-quad][4 clips]
#define X4B_X0 0               // x[s]      24 k-quads
#define X4B_H1 96              // h1[s-1]   64
#define X4B_H2 352             // h2[s-3]  128
#define X4B_X1 864             // x[s-1]    24 (the head's einsum)
#define X4B_F4 1024            // 16 KB
#define X4_NBUF 6              // buffers allocated (two are used): 96 KB keep a second workgroup off the CU
#ifndef X4_RING
#define X4_RING 4
#endif
#ifndef X4_AHEAD
#define X4_AHEAD 3
#endif

struct X4Packed { size_t a2, a1, as, ax, total; };     // offsets in floats
__host__ __device__ inline X4Packed x4_packed_layout()
{
    X4Packed P;
    size_t o = 0;
    P.a2 = o; o += (size_t)32 * 4 * 32 * 256;    // [cu][wave][q][lane] float4
    P.a1 = o; o += (size_t)32 * 4 * 11 * 256;    // [cu][wave][m][lane]
    P.as = o; o += (size_t)4 * 4 * 256;          // [wave][m][lane]
    P.ax = o; o += (size_t)32 * 2 * 256;         // [cu][q][lane]
    P.total = o;
    return P;
}

struct Xcd4Args {
    int B, T, RB;
    const float *pk;           // x4_packed_layout image
    const float *woutp;        // output head tiles of the inference layout (opnet_xcd4_out_head)
    char *ws;                  // training workspace base; everything below is a byte offset into it (one buffer descriptor)
    unsigned xp_off;           // [T][RB][24][32] float4
    unsigned h1_off, h2_off;   // [T+1][RB][H/4][32] float4, slot t+1 = step t
    unsigned c1_off, c2_off;   // [T+1][RB][H][32] float
    unsigned g1_off, g2_off;   // [T][RB][H][32] float4 post-activation gates
    unsigned ps_off, x2_off;   // [T][RB][4][32], [T][RB][2][32] float4
    unsigned lg_off, ys_off;   // logits / y staging
    unsigned h1x_off, h2x_off; // exchange rings: [RB*8 groups][4][H/4][4] float4, slot (t + 1) & 3 = step t
    unsigned *status;          // as XcdArgs.status
    int force_safe;
    int debug;                 // tools only (wrong results): bit 0 no gather, 1 no history stores, 2 no cells, 3 no head, 4 no products
    unsigned long long *trace; // optional [phases][8] s_memtime stamps of block 0, wave 0
};

typedef float x4_f32x4 __attribute__((ext_vector_type(4)));
#define X4_MFMA(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc, 0, 0, 0)
// The forward's product loop as asm statements, LSTM2's weights (128 of the wave's registers) as AccVGPR A operands: with all the
// weights in VGPRs the kernel needs ~310 registers, the allocator parks 54 of them in AGPRs and moves them back and forth around the
// products - ~80 v_accvgpr moves per phase, VALU instructions on the step's critical path.  An MFMA reads an A operand out of an
// AGPR at the same rate (measured in the stacked-LSTM launch), so the weights can LIVE there.  What the compiler no longer does for
// asm MFMAs: chains accumulate in place (D = C: forwarded, no wait states), B operands come from counted ds_reads, the A operands
// were written at kernel start, X4_MFMA_DRAIN pads the last results (2 passes: 5 wait states) before the compiler's code reads them.
#ifndef X4_AG
#define X4_AG 1
#endif
#if X4_AG
#define X4_MFMA_W(acc, av, bv) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(av), "v"(bv))
#else
#define X4_MFMA_W(acc, av, bv) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv))
#endif
#define X4_MFMA_V(acc, av, bv) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv))
// the FIRST MFMA of a chain takes the constant 0 as C: no accumulator is cleared beforehand (a wave64 v_mov occupies the SIMD for 4
// cycles; twelve accumulator quads a phase were 48 of them = ~190 cycles of the step's critical chain, profiles/r4_xcd4_experiments.txt)
#if X4_AG
#define X4_MFMA0_W(acc, av, bv) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=&v"(acc) : "a"(av), "v"(bv))
#else
#define X4_MFMA0_W(acc, av, bv) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(av), "v"(bv))
#endif
#define X4_MFMA0_V(acc, av, bv) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(av), "v"(bv))
// lane l of every row of 16 lanes receives lane l ^ 8 / l ^ 4 of its row through DPP (row_ror) instead of ds_bpermute (what
// __shfl_xor compiles to: a trip through the LDS crossbar, ~120 cycles each, and the head's softmax + einsum chains 16 of them).
// Same partners as __shfl_xor, so the same sums bit for bit.
__device__ __forceinline__ float x4_xor8(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));      // row_ror:8
}
__device__ __forceinline__ float x4_xor4(float v)
{
    const int vi = __float_as_int(v);
    int t = __builtin_amdgcn_update_dpp(vi, vi, 0x12c, 0xf, 0x5, false);   // row_ror:12 (lane i <- i + 4): banks 0, 2
    t = __builtin_amdgcn_update_dpp(t, vi, 0x124, 0xf, 0xa, false);        // row_ror:4  (lane i <- i - 4): banks 1, 3
    return __int_as_float(t);
}
#define X4_MFMA_DRAIN8(c0, c1, c2, c3, c4, c5, c6, c7) \
    asm volatile("s_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7))

// history accesses through the workspace's buffer descriptor: lane offset (loop-invariant, computed once) + wave-uniform scalar
// offset.  As pointer arithmetic every one of them cost ~10 VALU instructions of 64-bit address math per phase - 4 cycles each on
// a wave64 - on the cell waves, i.e. on the step's critical chain (profiles/r4_xcd4_experiments.txt).
__device__ __forceinline__ void x4_st1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
__device__ __forceinline__ void x4_st4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float4 v)
{
    xcd_u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 0);
}
__device__ __forceinline__ float x4_ld1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

__device__ __forceinline__ float4 x4_as_float4(xcd_u32x4 r)
{
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}
__device__ __forceinline__ bool x4_unpublished(xcd_u32x4 r) { return r.x == X4_SENT || r.y == X4_SENT || r.z == X4_SENT || r.w == X4_SENT; }

// every 64th round of a sentinel poll: has somebody raised the abort word, is the wait over the limit?  false = give up.
// The wall clock (s_memrealtime: hundreds of cycles) is first read at round 64 - a poll that succeeds never pays for it.
__device__ __forceinline__ bool x4_keep_polling(unsigned spins, long long &t0, unsigned *status, int phase)
{
    if ((spins & 63u) != 0) return true;
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    const long long now = (long long)wall_clock64();
    if (spins == 64u) t0 = now;
    if (now - t0 > XCD_SPIN_LIMIT) {
        if ((threadIdx.x & 63) == 0) {
            __hip_atomic_store(status + 1, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(status + 2, (unsigned)phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return false;
    }
    return true;
}

// NP pieces of 1 KB (lane l: 16 B at src + 1024 q + 16 l) -> registers -> LDS at dst + 64 q float4; while any lane of the wave still
// sees a sentinel word, all of them are loaded again (straight-line code: per-piece bookkeeping compiled to ~30 scalar branches a
// call).  wait = false: this phase does not use the pieces (nobody publishes them any more / yet): nothing is loaded.
// false = abort (wave-uniform).
template <int NP>
__device__ __forceinline__ bool x4_gather(__amdgpu_buffer_rsrc_t rws, unsigned lane16, unsigned src, bool wait, float4 *dst,
                                          unsigned *status, int phase)
{
    if (!wait) return true;
    xcd_u32x4 r[NP];
    long long t0 = 0;
    for (unsigned spins = 1;; ++spins) {
#pragma unroll
        for (int q = 0; q < NP; ++q) r[q] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, src + q * 1024, 16);   // sc1
        bool bad = false;
#pragma unroll
        for (int q = 0; q < NP; ++q) bad |= x4_unpublished(r[q]);
        if (!__any(bad)) break;
        if (!x4_keep_polling(spins, t0, status, phase)) return false;
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) dst[q * 64] = x4_as_float4(r[q]);
    return true;
}

// The same for four pieces that are only ever needed as their SUM (the backward's partial dh rows): (r0 + r1) + (r2 + r3) per lane
// goes to LDS as one piece - the consumer then adds 8 values instead of 32.
__device__ __forceinline__ bool x4_gather_sum4(__amdgpu_buffer_rsrc_t rws, unsigned lane16, unsigned src, bool wait, float4 *dst,
                                               unsigned *status, int phase)
{
    if (!wait) return true;
    xcd_u32x4 r[4];
    long long t0 = 0;
    for (unsigned spins = 1;; ++spins) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, src + q * 1024, 16);   // sc1
        bool bad = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) bad |= x4_unpublished(r[q]);
        if (!__any(bad)) break;
        if (!x4_keep_polling(spins, t0, status, phase)) return false;
    }
    const float4 a = x4_as_float4(r[0]), b = x4_as_float4(r[1]), c = x4_as_float4(r[2]), d = x4_as_float4(r[3]);
    *dst = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
    return true;
}

// fp32 weights -> the register images of opnet_xcd4_forward (see the layout notes above)
__device__ __forceinline__ void x4_pack_fwd_body(float *__restrict__ out, const float *__restrict__ w_ih1,
                                                 const float *__restrict__ w_hh1, const float *__restrict__ w_sel,
                                                 const float *__restrict__ w_ih2, const float *__restrict__ w_hh2, unsigned bid, unsigned nblk)
{
    const X4Packed P = x4_packed_layout();
    for (size_t idx = bid * (size_t)blockDim.x + threadIdx.x; idx < P.total; idx += (size_t)nblk * blockDim.x) {
        const int e = idx & 3, lane = (idx >> 2) & 63, b = lane >> 2, i = lane & 3;
        float v = 0.f;
        if (idx < P.a1) {
            const size_t r = idx >> 8;
            const int q = r % 32, w = (r / 32) % 4, cu = r / 128;
            const int k = 4 * (32 * w + q) + e;
            v = w_hh2[(size_t)(i * 512 + 16 * cu + b) * 512 + k];
        } else if (idx < P.as) {
            const size_t r = (idx - P.a1) >> 8;
            const int m = r % 11, w = (r / 11) % 4, cu = r / 44;
            const int k = 4 * (22 * w + 2 * m + (b >> 3)) + e;
            const size_t row = i * 256 + 8 * cu + (b & 7);
            v = k < 96 ? (k < OPNET_KX ? w_ih1[row * OPNET_KX + k] : 0.f) : w_hh1[row * 256 + (k - 96)];
        } else if (idx < P.ax) {
            const size_t r = (idx - P.as) >> 8;
            const int m = r % 4, w = r / 4;
            const int slot = 4 * (b & 3) + i;
            const int k = 4 * (16 * w + 4 * m + (b >> 2)) + e;
            v = slot < OPNET_SLOTS_ ? w_sel[(size_t)slot * 256 + k] : 0.f;
        } else {
            const size_t r = (idx - P.ax) >> 8;
            const int q = r % 2, cu = r / 2;
            const int k = 4 * q + e;
            v = k < OPNET_FEATS_ ? w_ih2[(size_t)(i * 512 + 16 * cu + b) * OPNET_FEATS_ + k] : 0.f;
        }
        out[idx] = v;
    }
}

// status words and the XCC sentinels; the exchange rings: slot 0 = the zero initial state, the others unpublished
__device__ __forceinline__ void x4_init_body(const Xcd4Args &a, int tid, int n)
{
    const int NG = a.RB * 8;
    if (tid < 8) a.status[tid] = 0u;
    for (int i = tid; i < 256; i += n) a.status[8 + i] = 0xffffffffu;
    const xcd_u32x4 z = {0u, 0u, 0u, 0u}, sent = {X4_SENT, X4_SENT, X4_SENT, X4_SENT};
    xcd_u32x4 *h1x = (xcd_u32x4 *)(a.ws + a.h1x_off), *h2x = (xcd_u32x4 *)(a.ws + a.h2x_off);
    for (int i = tid; i < NG * X4_SLOTS * 256; i += n) h1x[i] = ((i >> 8) & (X4_SLOTS - 1)) == 0 ? z : sent;
    for (int i = tid; i < NG * X4_SLOTS * 512; i += n) h2x[i] = ((i >> 9) & (X4_SLOTS - 1)) == 0 ? z : sent;
}

__global__ void __launch_bounds__(256) opnet_xcd4_init(Xcd4Args a)
{
    x4_init_body(a, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// The three launches in front of the training forward as one (each of them is a few microseconds of work behind ~4 us of launch):
// the caller's pointers into the device-side OpnetIO (opnet_copy_out reads it), boxes -> xp + zeroed state (opnet_pack_input),
// status words and exchange rings (opnet_xcd4_init).  Grid (T, RB + 1): row RB initialises, the rest pack; the pack reads the
// pointers out of the kernarg copy, not out of the struct this launch is writing.
__global__ void __launch_bounds__(256) opnet_x4_train_prologue(OpnetIO *dio, const OpnetIO io, const Xcd4Args a)
{
    if ((int)blockIdx.y < io.RB) { pack_input_body(&io, blockIdx.x, blockIdx.y, io.RB); return; }
    if (blockIdx.x == 0 && threadIdx.x == 0) *dio = io;
    const int tid = blockIdx.x * 256 + threadIdx.x, n = gridDim.x * 256;
    x4_init_body(a, tid, n);
    // slot 0 of the four state histories (h1, c1, h2, c2 of "step -1": the backward and the weight gradients read them); the
    // launch chain zeroes its whole state region (59 MB at 32 x 300) because its step kernels read what they have not written yet
    // - here every other slot is written by the recurrence before anything reads it (the host passes io.state_f4 = 0)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 *h1 = (float4 *)(a.ws + a.h1_off), *c1 = (float4 *)(a.ws + a.c1_off), *h2 = (float4 *)(a.ws + a.h2_off), *c2 = (float4 *)(a.ws + a.c2_off);
    for (int i = tid; i < a.RB * 2048; i += n) { h1[i] = z; c1[i] = z; }
    for (int i = tid; i < a.RB * 4096; i += n) { h2[i] = z; c2[i] = z; }
}

// lane i of every row of 16 lanes receives lane i + N of its row
template <int N>
__device__ __forceinline__ float x4_row_shl(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true));
}

template <bool TRAIN>
__global__ void __launch_bounds__(256) opnet_xcd4_forward(const Xcd4Args a)
{
    // Phase (row block gi, step s), T + 2 steps:  LSTM1 step s | selection head step s-1 | LSTM2 step s-2.
    //   CU c: LSTM2 64 gate rows x K = 512 split over the 4 waves (128 VGPRs a wave), LSTM1 32 gate rows x K = 96 + 256 (44),
    //   the selection head (every CU, redundantly) 16 rows x K = 256 (16), wave 0: W_ih2 of its 64 rows (8).
    //   LSTM2: block = unit, row = gate, one k per instruction, B = h2[k][clip j] in every block (a ds_read_b128 of the
    //   [k/4][clip] float4 buffer, the same address in all 16 blocks); LSTM1: block = (unit, k half); head: (slot quad, k quarter).
    //   1. products out of the LDS buffer of the phase (x[s], h1[s-1], h2[s-3], x[s-1]) and the CU's own frames_boxes[s-2]
    //      (LDS) -> K-split partials to LDS, barrier;
    //   2. wave 0: LSTM2 cell (lane = unit b, clip j), wave 1: LSTM1 cell, wave 2: head sum, softmax, einsum -> frames_boxes[s-1]
    //      to LDS, wave 3: the next phase's x and h1; the cell waves store h into the exchange ring, then the histories the
    //      backward needs (gates, c, h; wave 2: p, frames_boxes, logits), then gather their pieces of the next phase's h2;
    //   3. barrier.
    // Summation order: LSTM2 gate = ((w0 + w1) + w2) + w3 over the K quarters, each quarter (c0 + c1) + (c2 + c3) over four
    // interleaved ascending-k chains (k mod 4), wave 0's chains continued by the W_ih2 part; LSTM1 gate = sum over waves of
    // (low k half + high k half), each half two chains; logits = sum over waves, over k quarters of the wave's slice.
    __shared__ __attribute__((aligned(1024))) float4 sbuf[X4_NBUF][X4B_F4];
    __shared__ __attribute__((aligned(16))) float4 sP[4][3][64];       // K-split partials: LSTM2 | LSTM1 | head
    __shared__ __attribute__((aligned(16))) float sFB[X4_NGMAX][4][8]; // frames_boxes of the group's previous head step
    __shared__ float sC2[X4_NGMAX][64];
    __shared__ float sC1[X4_NGMAX][32];
    __shared__ int sAbort, sLocal;          // through XCD_LDS_LD / XCD_LDS_ST: ds_read / ds_write (a volatile LDS word is a FLAT access + vmcnt wait)

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
    for (int i = tid; i < X4_NGMAX * 32; i += 256) { (&sC1[0][0])[i] = 0.f; (&sFB[0][0][0])[i] = 0.f; }

    // ---- resident weights ------------------------------------------------------------------------------------------------
    const X4Packed P = x4_packed_layout();
    float4 a2[32], a1[11], as_[4], ax[2];
    {
        const float4 *p2 = (const float4 *)(a.pk + P.a2) + ((size_t)(c * 4 + w) * 32) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 32; ++q) a2[q] = p2[q * 64];
        const float4 *p1 = (const float4 *)(a.pk + P.a1) + ((size_t)(c * 4 + w) * 11) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 11; ++q) a1[q] = p1[q * 64];
        const float4 *ps = (const float4 *)(a.pk + P.as) + (size_t)(w * 4) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) as_[q] = ps[q * 64];
        const float4 *px = (const float4 *)(a.pk + P.ax) + (size_t)(c * 2) * 64 + lane;
        ax[0] = px[0];
        ax[1] = px[64];
    }

    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void *)a.ws, 0, 0x7fffffff, 0x00020000);
    const unsigned cb = 4 * x;                                  // first clip of this XCD's groups within a row block
    const unsigned lane16 = lane * 16;
    const unsigned xlane = ((lane >> 2) * 32 + cb + (lane & 3)) * 16;   // [k-quad][32 clips] float4: k-quad lane >> 2, clip cb + (lane & 3)
    const xcd_u32x4 sent4 = {X4_SENT, X4_SENT, X4_SENT, X4_SENT};
    const float4 sentf = x4_as_float4(sent4);
    bool alive = true;

    // The inputs of phase (gi, s) into LDS buffer `buf`, by wave: wave 3 x[s], x[s-1] (read-only input, no sentinel) and h1[s-1]
    // (ring slot s & 3, 4 pieces); waves 0 / 1: the halves of h2[s-3] (ring slot (s - 2) & 3, 4 pieces each); wave 2 (the head) none.
    // h1 is only waited for while somebody still publishes it (s <= T), h2 from s = 2 on.
    auto gather = [&](int gi, int s, int buf, int phase) -> bool {
        float4 *S = &sbuf[buf][0] + lane;
        const unsigned gg = gi * 8 + x;
        if (w == 3) {
            const int t0 = s < T ? s : T - 1, t1 = s > 0 ? (s - 1 < T ? s - 1 : T - 1) : 0;
            const unsigned o0 = a.xp_off + (unsigned)((t0 * RB + gi) * OPNET_KXQ) * 512;
            const unsigned o1 = a.xp_off + (unsigned)((t1 * RB + gi) * OPNET_KXQ) * 512;
            const xcd_u32x4 xa = __builtin_amdgcn_raw_buffer_load_b128(rws, xlane, o0, 0);
            const xcd_u32x4 xb = __builtin_amdgcn_raw_buffer_load_b128(rws, xlane, o1, 0);
            xcd_u32x4 xc = sent4, xd = sent4;
            if (lane < 32) {                                    // k-quads 16 .. 23
                xc = __builtin_amdgcn_raw_buffer_load_b128(rws, xlane, o0 + 16 * 512, 0);
                xd = __builtin_amdgcn_raw_buffer_load_b128(rws, xlane, o1 + 16 * 512, 0);
            }
            const bool ok = x4_gather<4>(rws, lane16, a.h1x_off + (gg * X4_SLOTS + (s & 3)) * 4096, s <= T,
                                         S + X4B_H1, a.status, phase);
            S[X4B_X0] = x4_as_float4(xa);
            S[X4B_X1] = x4_as_float4(xb);
            if (lane < 32) { S[X4B_X0 + 64] = x4_as_float4(xc); S[X4B_X1 + 64] = x4_as_float4(xd); }
            return ok;
        }
        if (w == 2) return true;
        return x4_gather<4>(rws, lane16, a.h2x_off + (gg * X4_SLOTS + ((s + 2) & 3)) * 8192 + w * 4096, s >= 2,
                            S + X4B_H2 + w * 256, a.status, phase);
    };

    // lane offsets (bytes) of the history stores: LSTM2 unit u2 = 16 c + b, LSTM1 unit u1 = 8 c + b, clip cb + j
    const unsigned u2 = 16 * c + b, u1 = 8 * c + b;
    const unsigned vh2 = ((u2 >> 2) * 128 + (cb + j) * 4 + (u2 & 3)) * 4, vc2 = (u2 * 32 + cb + j) * 4, vg2 = (u2 * 32 + cb + j) * 16;
    const unsigned vh1 = ((u1 >> 2) * 128 + (cb + j) * 4 + (u1 & 3)) * 4, vc1 = (u1 * 32 + cb + j) * 4, vg1 = (u1 * 32 + cb + j) * 16;

    if (!gather(0, 0, 0, 0)) XCD_LDS_ST(sAbort, 1);
    __syncthreads();
    if (XCD_LDS_LD(sAbort)) return;
    int abort_seen = 0;                         // the abort word as read behind the PREVIOUS phase's last barrier (see the loop's end)
    const bool local = __builtin_amdgcn_readfirstlane(XCD_LDS_LD(sLocal)) != 0;
    const bool tracer = a.trace && blockIdx.x == 0 && tid == 0;
    const int nph = (T + 2) * ng;

    int gi = 0, s = 0;
    for (int p = 0; p < nph; ++p) {
        const int buf = p & 1;
        if (tracer) a.trace[(long)p * 8 + 0] = clock64();
        // ================================ products =======================================================================
        if (!(a.debug & 16)) {
            const float4 *S = &sbuf[buf][0];
            const float4 *F2 = S + X4B_H2 + (32 * w) * 4 + j;
            const float4 *F1 = S + X4B_X0 + (22 * w + (b >> 3)) * 4 + j;
            const float4 *FH = S + X4B_H1 + (16 * w + (b >> 2)) * 4 + j;
            // (four chains per product: with two, LSTM1's 44 and the head's 16 MFMAs wait for each other's results - a dependent
            // v_mfma_f32_4x4x1 issues every ~30 cycles, the pipe takes one every 9.5)
            x4_f32x4 c2[4], c1[4], cH[4];      // (not cleared: the first MFMA of every chain has C = 0)
            // 47 B fragments (32 LSTM2 | 11 LSTM1 | 4 head) through a ring of X4_RING registers quads, fetched X4_AHEAD
            // fragments (~40 cycles of MFMA each) ahead: with one ds_read in flight per 4 MFMAs (what the compiler schedules
            // when left alone) every fragment's LDS latency is exposed (measured 2 750 cycles for the 194 MFMAs instead of
            // 1 850); a bigger ring pushes weights into AGPRs, and every MFMA operand then costs a v_accvgpr_read (2 470)
            auto frag = [&](int idx) -> const float4 * {
                return idx < 32 ? F2 + idx * 4 : idx < 43 ? F1 + (idx - 32) * 8 : FH + (idx - 43) * 16;
            };
            float4 bf[X4_RING];
            float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
            if (w == 0) { f0 = *(const float4 *)&sFB[gi][j][0]; f1 = *(const float4 *)&sFB[gi][j][4]; }
#pragma unroll
            for (int i = 0; i < X4_AHEAD; ++i) bf[i] = *frag(i);
#pragma unroll
            for (int idx = 0; idx < 47; ++idx) {
                if (idx + X4_AHEAD < 47) bf[(idx + X4_AHEAD) % X4_RING] = *frag(idx + X4_AHEAD);
                __builtin_amdgcn_sched_barrier(0);
                const float4 bq = bf[idx % X4_RING];
                if (idx == 0) {
                    X4_MFMA0_W(c2[0], a2[0].x, bq.x);
                    X4_MFMA0_W(c2[1], a2[0].y, bq.y);
                    X4_MFMA0_W(c2[2], a2[0].z, bq.z);
                    X4_MFMA0_W(c2[3], a2[0].w, bq.w);
                } else if (idx < 32) {
                    X4_MFMA_W(c2[0], a2[idx < 32 ? idx : 0].x, bq.x);
                    X4_MFMA_W(c2[1], a2[idx < 32 ? idx : 0].y, bq.y);
                    X4_MFMA_W(c2[2], a2[idx < 32 ? idx : 0].z, bq.z);
                    X4_MFMA_W(c2[3], a2[idx < 32 ? idx : 0].w, bq.w);
                } else if (idx == 32) {
                    X4_MFMA0_V(c1[0], a1[0].x, bq.x);
                    X4_MFMA0_V(c1[1], a1[0].y, bq.y);
                    X4_MFMA0_V(c1[2], a1[0].z, bq.z);
                    X4_MFMA0_V(c1[3], a1[0].w, bq.w);
                } else if (idx < 43) {
                    const int m = idx < 43 ? idx - 32 : 0;
                    X4_MFMA_V(c1[0], a1[m].x, bq.x);
                    X4_MFMA_V(c1[1], a1[m].y, bq.y);
                    X4_MFMA_V(c1[2], a1[m].z, bq.z);
                    X4_MFMA_V(c1[3], a1[m].w, bq.w);
                } else if (idx == 43) {
                    X4_MFMA0_V(cH[0], as_[0].x, bq.x);
                    X4_MFMA0_V(cH[1], as_[0].y, bq.y);
                    X4_MFMA0_V(cH[2], as_[0].z, bq.z);
                    X4_MFMA0_V(cH[3], as_[0].w, bq.w);
                } else {
                    const int m = idx - 43;
                    X4_MFMA_V(cH[0], as_[m].x, bq.x);
                    X4_MFMA_V(cH[1], as_[m].y, bq.y);
                    X4_MFMA_V(cH[2], as_[m].z, bq.z);
                    X4_MFMA_V(cH[3], as_[m].w, bq.w);
                }
                if (idx == 31 && w == 0) {      // LSTM2's input part: W_ih2 . frames_boxes[s-2] (K = 6)
                    X4_MFMA_V(c2[0], ax[0].x, f0.x);
                    X4_MFMA_V(c2[1], ax[0].y, f0.y);
                    X4_MFMA_V(c2[2], ax[0].z, f0.z);
                    X4_MFMA_V(c2[3], ax[0].w, f0.w);
                    X4_MFMA_V(c2[0], ax[1].x, f1.x);
                    X4_MFMA_V(c2[1], ax[1].y, f1.y);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            X4_MFMA_DRAIN8(c2[0], c2[1], c2[2], c2[3], c1[0], c1[1], c1[2], c1[3]);
            asm volatile("" : "+v"(cH[0]), "+v"(cH[1]), "+v"(cH[2]), "+v"(cH[3]));
            float4 *pp = &sP[w][0][lane];
            pp[0] = make_float4((c2[0][0] + c2[1][0]) + (c2[2][0] + c2[3][0]), (c2[0][1] + c2[1][1]) + (c2[2][1] + c2[3][1]),
                                (c2[0][2] + c2[1][2]) + (c2[2][2] + c2[3][2]), (c2[0][3] + c2[1][3]) + (c2[2][3] + c2[3][3]));
            const x4_f32x4 s1 = (c1[0] + c1[1]) + (c1[2] + c1[3]), sH = (cH[0] + cH[1]) + (cH[2] + cH[3]);
            pp[64] = make_float4(s1[0], s1[1], s1[2], s1[3]);
            pp[128] = make_float4(sH[0], sH[1], sH[2], sH[3]);
        }
        if (tracer) a.trace[(long)p * 8 + 1] = clock64();
        __syncthreads();                        // barrier 1: the phase's partials are in sP
        if (tracer) a.trace[(long)p * 8 + 2] = clock64();
        // (an abort can only be raised in the second half of a phase: it is looked at after barrier 2)
        // the next phase
        int gn = gi + 1, sn = s;
        if (gn == ng) { gn = 0; ++sn; }
        const bool more = p + 1 < nph;
        const unsigned gg = gi * 8 + x;
        const int rb = gi;

        // ================================ finish, by wave =================================================================
        if (w == 0) {
            // ---- LSTM2 cell of step t = s - 2 (learned_models.py:46): lane = (unit 16 c + b, clip j) ----------------------
            const int t = s - 2;
            if (t >= 0 && t < T && !(a.debug & 4)) {
                const float4 p0 = sP[0][0][lane], p1 = sP[1][0][lane], p2 = sP[2][0][lane], p3 = sP[3][0][lane];
                float cc = sC2[gi][lane];
                float4 gs;
                const float h = lstm_cell_g(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                                            ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w, &cc, &gs);
                sC2[gi][lane] = cc;
                // exchange: float4 = units 4 q .. 4 q + 3 of clip j, by the lanes with (b & 3) == 0; ring slot (t + 1) & 3 = (s - 1) & 3,
                // and the slot two steps on is re-armed
                const float4 hv = make_float4(h, x4_row_shl<4>(h), x4_row_shl<8>(h), x4_row_shl<12>(h));
                if ((b & 3) == 0) {
                    const unsigned vo = ((b >> 2) * 4 + j) * 16;
                    xcd_store16(rws, vo, a.h2x_off + ((gg * X4_SLOTS + ((s - 1) & 3)) * 128 + 4 * c) * 64, hv, local);
                    xcd_store16(rws, vo, a.h2x_off + ((gg * X4_SLOTS + ((s + 1) & 3)) * 128 + 4 * c) * 64, sentf, local);
                }
                if (tracer) a.trace[(long)p * 8 + 3] = clock64();
                // the histories of the backward pass / the output head
                const unsigned row1 = (unsigned)((t + 1) * RB + rb), row0 = (unsigned)(t * RB + rb);
                if (!(a.debug & 2)) x4_st1(rws, vh2, a.h2_off + row1 * (128 * 128 * 4), h);
                if (TRAIN && !(a.debug & 2)) {
                    x4_st1(rws, vc2, a.c2_off + row1 * (512 * 32 * 4), cc);
                    x4_st4(rws, vg2, a.g2_off + row0 * (512 * 32 * 16), gs);
                }
            }
        } else if (w == 1) {
            // ---- LSTM1 cell of step t = s (learned_models.py:39): lanes 0..31 = (unit 8 c + b, clip j) --------------------
            const int t = s;
            if (t < T && lane < 32 && !(a.debug & 4)) {
                float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 lo = sP[q][1][lane], hi = sP[q][1][lane + 32];
                    g[0] += lo.x + hi.x; g[1] += lo.y + hi.y; g[2] += lo.z + hi.z; g[3] += lo.w + hi.w;
                }
                float cc = sC1[gi][lane];
                float4 gs;
                const float h = lstm_cell_g(g[0], g[1], g[2], g[3], &cc, &gs);
                sC1[gi][lane] = cc;
                const float4 hv = make_float4(h, x4_row_shl<4>(h), x4_row_shl<8>(h), x4_row_shl<12>(h));
                if ((b & 3) == 0) {     // ring slot (t + 1) & 3
                    const unsigned vo = ((b >> 2) * 4 + j) * 16;
                    xcd_store16(rws, vo, a.h1x_off + ((gg * X4_SLOTS + ((s + 1) & 3)) * 64 + 2 * c) * 64, hv, local);
                    xcd_store16(rws, vo, a.h1x_off + ((gg * X4_SLOTS + ((s + 3) & 3)) * 64 + 2 * c) * 64, sentf, local);
                }
                if (TRAIN && !(a.debug & 2)) {
                    const unsigned row1 = (unsigned)((t + 1) * RB + rb), row0 = (unsigned)(t * RB + rb);
                    x4_st1(rws, vh1, a.h1_off + row1 * (64 * 128 * 4), h);
                    x4_st1(rws, vc1, a.c1_off + row1 * (256 * 32 * 4), cc);
                    x4_st4(rws, vg1, a.g1_off + row0 * (256 * 32 * 16), gs);
                }
            }
        } else if (w == 2) {
            // ---- selection head of step t = s - 1 (learned_models.py:40-43,50): lanes 0..15 = (slot quad rg, clip j) ------
            const int t = s - 1;
            if (t >= 0 && t < T && lane < 16 && !(a.debug & 8)) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const float4 pv = sP[q][2][lane + 16 * kk];
                        v[0] += pv.x; v[1] += pv.y; v[2] += pv.z; v[3] += pv.w;
                    }
                const int rg = lane >> 2;
                float m = fmaxf(fmaxf(v[0], v[1]), v[2]);
                if (rg < 3) m = fmaxf(m, v[3]);                 // slot 15 does not exist
                m = fmaxf(m, x4_xor4(m));
                m = fmaxf(m, x4_xor8(m));
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = __expf(v[r] - m);
                if (rg == 3) e[3] = 0.f;
                float sum = (e[0] + e[1]) + (e[2] + e[3]);
                sum += x4_xor4(sum);
                sum += x4_xor8(sum);
                const float inv = 1.0f / sum;
                const float pr[4] = {e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv};
                // frames_boxes[j][f] = sum_o boxes[j][t][o][f] p[o] (einsum "bfot,bfo->bft"): this lane's slots 4 rg .. 4 rg + 3
                // = k 24 rg .. 24 rg + 23 of x[t] = k-quads 6 rg .. 6 rg + 5 of the X1 piece
                const float4 *X = &sbuf[buf][X4B_X1] + (6 * rg) * 4 + j;
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
                    acc += x4_xor4(acc);
                    acc += x4_xor8(acc);
                    fbv[f] = acc;
                }
                fbv[6] = fbv[7] = 0.f;
                if (rg == 0) {
                    *(float4 *)&sFB[gi][j][0] = make_float4(fbv[0], fbv[1], fbv[2], fbv[3]);
                    *(float4 *)&sFB[gi][j][4] = make_float4(fbv[4], fbv[5], 0.f, 0.f);
                }
                if (c == (s & 31)) {            // every CU computes the head; one of them records it
                    const unsigned clip = rb * 32 + cb + j;
                    float *lg = (float *)(a.ws + a.lg_off);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * rg + r < OPNET_SLOTS_) lg[((size_t)clip * OPNET_SLOTS_ + 4 * rg + r) * T + t] = v[r];
                    if (TRAIN) {
                        float4 *ps = (float4 *)(a.ws + a.ps_off);
                        ps[((size_t)(t * RB + rb) * 4 + rg) * 32 + cb + j] = make_float4(pr[0], pr[1], pr[2], pr[3]);
                        if (rg < 2) {
                            float4 *x2 = (float4 *)(a.ws + a.x2_off);
                            x2[((size_t)(t * RB + rb) * 2 + rg) * 32 + cb + j] =
                                rg == 0 ? make_float4(fbv[0], fbv[1], fbv[2], fbv[3]) : make_float4(fbv[4], fbv[5], 0.f, 0.f);
                        }
                    }
                }
            }
        }
        if (tracer) a.trace[(long)p * 8 + 4] = clock64();
        // ================================ the next phase's inputs ==========================================================
        if (more && alive && !(a.debug & 1)) {
            alive = gather(gn, sn, buf ^ 1, p);
            if (!alive) XCD_LDS_ST(sAbort, 1);
        }
        if (tracer) a.trace[(long)p * 8 + 5] = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // orders this phase's re-arm stores before the next publish
        if (tracer) a.trace[(long)p * 8 + 6] = clock64();
        __syncthreads();                        // barrier 2: the next phase's inputs have landed
        if (tracer) a.trace[(long)p * 8 + 7] = clock64();
        // the abort word is looked at one phase late: read here, behind the barrier, but only tested a phase on, so that the LDS round
        // trip stays off the step's critical chain (an aborted launch's outputs are NaN whatever this block still stores)
        if (abort_seen) return;
        abort_seen = XCD_LDS_LD(sAbort);
        gi = gn;
        s = sn;
    }
}

// y staging [RB*32][T] float4 = W_out h2[t] (prediction_layer, learned_models.py:33,47) from the h2 history; one workgroup per
// (t, row block): thread (r, clip) walks k-quads r, r + 8, ...; the 8 partials are summed in fixed order.  An aborted persistent
// launch (status[0] != 0) poisons y with NaN.
// y_out / lg_out (training forward; null otherwise): the caller's y [B][T][4] is written here as well, and one more row of
// workgroups (blockIdx.y == RB) copies the staged logits to the caller's [B][15][T] - what opnet_copy_out did in a launch of its own
__global__ void __launch_bounds__(256) opnet_xcd4_out_head(const Xcd4Args a, float4 *__restrict__ y_out, float *__restrict__ lg_out)
{
    __shared__ float sw[4][512];
    __shared__ __attribute__((aligned(16))) float4 red[8][32];
    const int t = blockIdx.x, rb = blockIdx.y, tid = threadIdx.x, T = a.T;
    if (rb == a.RB) {
        const float *__restrict__ ls = (const float *)(a.ws + a.lg_off);
        const long nl = (long)a.B * OPNET_SLOTS_ * T;
        for (long i = (long)t * 256 + tid; i < nl; i += (long)gridDim.x * 256) lg_out[i] = ls[i];
        return;
    }
    const float *wo = a.woutp;                 // [H2/16][64][4]: lane l = row l & 15, k = 16 q + 4 (l >> 4) + e
    for (int i = tid; i < 4 * 512; i += 256) {
        const int o = i / 512, k = i % 512;
        sw[o][k] = wo[(((k >> 4) * 64) + o + 16 * ((k & 15) >> 2)) * 4 + (k & 3)];
    }
    __syncthreads();
    const int r = tid >> 5, clip = tid & 31;
    const float4 *h = (const float4 *)(a.ws + a.h2_off) + ((size_t)(t + 1) * a.RB + rb) * (128 * 32);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kq = r; kq < 128; kq += 8) {
        const float4 hv = h[kq * 32 + clip];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            acc[o] = fmaf(sw[o][4 * kq + 0], hv.x, acc[o]);
            acc[o] = fmaf(sw[o][4 * kq + 1], hv.y, acc[o]);
            acc[o] = fmaf(sw[o][4 * kq + 2], hv.z, acc[o]);
            acc[o] = fmaf(sw[o][4 * kq + 3], hv.w, acc[o]);
        }
    }
    red[r][clip] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (tid < 32) {
        float4 sum = red[0][tid];
        for (int i = 1; i < 8; ++i) {
            const float4 v = red[i][tid];
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        if (a.status[0] != 0u) sum = make_float4(NAN, NAN, NAN, NAN);
        ((float4 *)(a.ws + a.ys_off))[((size_t)rb * 32 + tid) * T + t] = sum;
        if (y_out && rb * 32 + tid < a.B) y_out[((size_t)rb * 32 + tid) * T + t] = sum;
    }
}

// ====================================================================================================================
// backward: the reverse recurrence of opnet_train_backward_f32, same placement and exchange protocol as the forward above.
//     dh2_t = W_out^T dy_t + W_hh2^T da2_{t+1};   (da2_t, dc2) = cell backward           (opnet_train_kernels.hip cell_backward)
//     dfb_t = W_ih2^T da2_t;  dp = boxes_t . dfb_t;  dl_t = p_t * (dp - <p_t, dp>)          (einsum + softmax backward)
//     dh1_t = W_sel^T dl_t + W_hh1^T da1_{t+1};   (da1_t, dc1) = cell backward
// da_t overwrites the saved gates (g2 / g1, the launch chain's layout) and dl_t goes to dlall, so that opnet_wgrad runs on
// the result unchanged.
// The recurrent product contracts over the 4H gate columns.  A CU that owned complete dh rows would have to gather ALL of da
// (48 KB a step: the first version of this kernel, 14 000 cycles a step); instead CU c keeps the gate columns of ITS units - the
// same 64 + 32 rows of W_hh it holds in the forward, read the other way - multiplies them by ITS OWN da (from LDS: no exchange on
// the way in) into partial dh rows of EVERY unit, and the exchange is a reduce-scatter: every CU stores 32 x 384 B of partials,
// one chunk per owner, and gathers the 32 chunks of its own 16 + 8 units (12 KB, as in the forward); they are summed in fixed
// order - four by four by the gathering waves (registers), the rest by the cell waves.
// MFMA blocks = row quads, one k per instruction (64 rows x 1 k x 4 clips), B = the (unit', clip) float4 of da - its four gates
// are four consecutive k - the same address in all 16 blocks.
// Phase (row block gi, n), T + 3 of them:  LSTM2 cell at t2 = T-1-n | head backward at th = T+1-n | LSTM1 cell at t1 = T+2-n.
//     1. wave 0: dh2 = sum of the 32 partial chunks + W_out^T dy -> cell -> da2_{t2} -> LDS, g2;
//        wave 3: the CU's part of dfb of the PREVIOUS phase's da2 (8 MFMAs, summed over the CU's units through LDS) -> its exchange
//                ring - beside wave 0's cell instead of after it, which is why the head runs two steps behind LSTM2;
//        wave 1: dh1 = sum of the chunks + W_sel^T dl_{t1} (dl from LDS, left by wave 2 one phase earlier) -> cell -> da1 -> LDS, g1;
//        wave 3: the next head step's p and boxes (HBM) -> LDS;      barrier;
//     2. every wave: its 128 + 64 rows of the partial products out of the CU's da, stored to the owners' chunks (ring slot t & 3;
//        the slot two steps on re-armed);
//     3. waves 0, 1, 3: four 1-KB pieces each of the next phase's chunks (sentinel-polled); wave 2 meanwhile: the head backward
//        of step th - sums the 32 CUs' dfb parts (published a phase ago), dp, dl_{th} -> LDS (+ dlall by one CU);      barrier.
// ====================================================================================================================
// LDS buffer of a backward phase, float4 units; the gathering waves add their four 1-KB pieces before they store them:
#define X4D_P2 0               // dh2: [wave 0 | 1][lane]: the sum over q of the chunk of producer 16 wave + 4 q + (lane >> 4), float4 lane & 15
#define X4D_P1 128             // dh1: [lane]: the sum over q of the chunk of producer 8 q + (lane >> 3), float4 lane & 7
#define X4D_F4 768             // 12 KB
#define X4D_NBUF 8             // buffers allocated (two are used): 96 KB keep a second workgroup off the CU

struct X4BPacked { size_t b2, b1, bx, bo, bs, total; };    // offsets in floats
__host__ __device__ inline X4BPacked x4b_packed_layout()
{
    X4BPacked P;
    size_t o = 0;
    P.b2 = o; o += (size_t)32 * 4 * 32 * 256;    // [cu][wave][set 2 x unit' 16][lane] float4   W_hh2 columns
    P.b1 = o; o += (size_t)32 * 4 * 8 * 256;     // [cu][wave][unit' 8][lane]                   W_hh1 columns
    P.bx = o; o += (size_t)32 * 2 * 256;         // [cu][f quad][lane]           W_ih2^T of the CU's units
    P.bo = o; o += (size_t)32 * 256;             // [cu][lane]                   W_out^T of the lane's unit
    P.bs = o; o += (size_t)32 * 4 * 256;         // [cu][slot quad][lane]        W_sel^T of the lane's unit
    P.total = o;
    return P;
}

struct Xcd4BArgs {
    int B, T, RB;
    const float *pk;           // x4b_packed_layout image
    char *ws;
    unsigned xp_off;           // [T][RB][24][32] float4
    unsigned c1_off, c2_off;   // [T+1][RB][H][32] float
    unsigned g1_off, g2_off;   // [T][RB][H][32] float4: gates in, da out
    unsigned ps_off;           // [T][RB][4][32] float4
    unsigned dy_off;           // [T][RB][32] float4
    unsigned dl_off;           // [T][RB][4][32] float4 out
    unsigned p1x_off, p2x_off; // exchange rings of partial dh: [RB*8][4][32 owners][32 producers][2 | 4 unit quads][4 clips] float4;
                               // slot t & 3 = the products of da_t (slot T & 3 starts as zeros: da_T = 0)
    unsigned dfx_off;          // exchange ring [RB*8][4][32 CUs][8 features][4 clips] float
    unsigned *status;
    int force_safe;
    int debug;                 // tools only (wrong results): bit 0 no gather, 1 no history stores, 2 no cells, 3 no head, 4 no products
    unsigned long long *trace;
};

__device__ __forceinline__ void x4_pack_bwd_body(float *__restrict__ out, const float *__restrict__ w_hh1,
                                                 const float *__restrict__ w_sel, const float *__restrict__ w_ih2,
                                                 const float *__restrict__ w_hh2, const float *__restrict__ w_out, unsigned bid, unsigned nblk)
{
    const X4BPacked P = x4b_packed_layout();
    for (size_t idx = bid * (size_t)blockDim.x + threadIdx.x; idx < P.total; idx += (size_t)nblk * blockDim.x) {
        const int e = idx & 3, lane = (idx >> 2) & 63, b = lane >> 2, i = lane & 3;
        float v = 0.f;
        if (idx < P.b1) {
            // lane (block b, row i) of (set, unit' m): dh row 128 w + 64 set + 4 b + i, k = (unit 16 cu + m, gate e)
            const size_t r = idx >> 8;
            const int m = r % 16, set = (r / 16) % 2, w = (r / 32) % 4, cu = r / 128;
            v = w_hh2[(size_t)(e * 512 + 16 * cu + m) * 512 + 128 * w + 64 * set + 4 * b + i];
        } else if (idx < P.bx) {
            const size_t r = (idx - P.b1) >> 8;
            const int m = r % 8, w = (r / 8) % 4, cu = r / 32;
            v = w_hh1[(size_t)(e * 256 + 8 * cu + m) * 256 + 64 * w + 4 * b + i];
        } else if (idx < P.bo) {
            const size_t r = (idx - P.bx) >> 8;
            const int q = r % 2, cu = r / 2;
            const int f = 4 * q + i;                // row of the MFMA block of unit 16 cu + b; k = gate e
            v = f < OPNET_FEATS_ ? w_ih2[(size_t)(e * 512 + 16 * cu + b) * OPNET_FEATS_ + f] : 0.f;
        } else if (idx < P.bs) {
            const size_t cu = (idx - P.bo) >> 8;
            v = w_out[(size_t)e * 512 + 16 * cu + b];            // lane (b, any i): W_out[e][unit]
        } else {
            const size_t r = (idx - P.bs) >> 8;
            const int q = r % 4, cu = r / 4;
            const int slot = 4 * q + e;
            v = slot < OPNET_SLOTS_ ? w_sel[(size_t)slot * 256 + 8 * cu + (b & 7)] : 0.f;
        }
        out[idx] = v;
    }
}

// status, the exchange rings: slot T & 3 of the partials = zeros (da_T = 0), everything else unpublished
__global__ void __launch_bounds__(256) opnet_xcd4_pack_fwd(float *__restrict__ out, const float *__restrict__ w_ih1,
                                                           const float *__restrict__ w_hh1, const float *__restrict__ w_sel,
                                                           const float *__restrict__ w_ih2, const float *__restrict__ w_hh2)
{
    x4_pack_fwd_body(out, w_ih1, w_hh1, w_sel, w_ih2, w_hh2, blockIdx.x, gridDim.x);
}
// both register images of a training step's weights in one launch (they are re-packed after every optimiser step): the first half of
// the grid writes the forward's, the second the backward's
__global__ void __launch_bounds__(256) opnet_xcd4_pack_both(float *__restrict__ out_f, float *__restrict__ out_b, const float *__restrict__ w_ih1,
                                                            const float *__restrict__ w_hh1, const float *__restrict__ w_sel,
                                                            const float *__restrict__ w_ih2, const float *__restrict__ w_hh2,
                                                            const float *__restrict__ w_out, float *__restrict__ out_head)
{
    const unsigned half = gridDim.x / 2;
    if (blockIdx.x < 32 && out_head) {
        // the output head's A-fragment tile (opnet_pack_tiles, mode 1: 4 rows of a 16-row tile, K = 512): 8 192 floats
        const int idx = blockIdx.x * 256 + threadIdx.x;
        const int e = idx & 3, lane = (idx >> 2) & 63, q = idx >> 8, i = lane & 15;
        const int k = 16 * q + 4 * (lane >> 4) + e;
        out_head[idx] = i < 4 ? w_out[i * 512 + k] : 0.f;
    }
    if (blockIdx.x < half) x4_pack_fwd_body(out_f, w_ih1, w_hh1, w_sel, w_ih2, w_hh2, blockIdx.x, half);
    else x4_pack_bwd_body(out_b, w_hh1, w_sel, w_ih2, w_hh2, w_out, blockIdx.x - half, half);
}

// (+ opnet_pack_dy's work when dy is given: the caller's dy [B][T][4] -> dyp [t][rb][clip], zeroed cell-gradient carry - one launch
// in front of the reverse recurrence instead of two)
__global__ void __launch_bounds__(256) opnet_xcd4_init_bwd(Xcd4BArgs a, const float4 *__restrict__ dy, float4 *__restrict__ dyp,
                                                           float *__restrict__ dc_zero, long n_dc, int B)
{
    const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x, n = (long)gridDim.x * blockDim.x;
    const long NG = a.RB * 8;
    if (dy) {
        const long nd = (long)a.T * a.RB * 32;
        for (long idx = tid; idx < nd; idx += n) {
            const int clip = idx & 31;
            const long trb = idx >> 5;
            const int rb = trb % a.RB;
            const int t = trb / a.RB;
            const int b = rb * 32 + clip;
            dyp[idx] = b < B ? dy[(long)b * a.T + t] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (long idx = tid; idx < n_dc; idx += n) dc_zero[idx] = 0.f;
    }
    // status[0..2] (abort code, block, phase) are STICKY from the forward of this step: an aborted forward makes the
    // reverse recurrence leave at once, and the weight-gradient launch and the optimiser's guard still see the abort word
    if (tid >= 3 && tid < 8) a.status[tid] = 0u;
    for (long i = tid; i < 256; i += n) a.status[8 + i] = 0xffffffffu;
    const xcd_u32x4 z = {0u, 0u, 0u, 0u}, sent = {X4_SENT, X4_SENT, X4_SENT, X4_SENT};
    const long zs = a.T & (X4_SLOTS - 1);
    xcd_u32x4 *d1 = (xcd_u32x4 *)(a.ws + a.p1x_off), *d2 = (xcd_u32x4 *)(a.ws + a.p2x_off), *df = (xcd_u32x4 *)(a.ws + a.dfx_off);
    for (long i = tid; i < NG * X4_SLOTS * 8192; i += n) d1[i] = ((i >> 13) & (X4_SLOTS - 1)) == zs ? z : sent;
    for (long i = tid; i < NG * X4_SLOTS * 16384; i += n) d2[i] = ((i >> 14) & (X4_SLOTS - 1)) == zs ? z : sent;
    for (long i = tid; i < NG * X4_SLOTS * 256; i += n) df[i] = sent;
}

__global__ void __launch_bounds__(256) opnet_xcd4_backward(const Xcd4BArgs a)
{
    __shared__ __attribute__((aligned(1024))) float4 sbuf[X4D_NBUF][X4D_F4];
    __shared__ __attribute__((aligned(16))) float4 sDA2[2][16][4];     // the CU's da2 of the phase (by phase parity: wave 3 turns the previous
                                                                       // phase's into its dfb part): [unit'][clip] -> (i, f, g, o)
    __shared__ __attribute__((aligned(16))) float4 sDA1[8][4];
    __shared__ long long sArrT[4];                                     // tools (trace): arrival of each wave at barrier 2
    __shared__ __attribute__((aligned(16))) float4 sX[2][64];          // per-unit parts of dfb: features 0..3 | 4..7
    __shared__ float sDC2[X4_NGMAX][64];
    __shared__ float sDC1[X4_NGMAX][32];
    __shared__ __attribute__((aligned(16))) float sDL[X4_NGMAX][2][4][16];   // dl of the group's head steps, by phase parity: [clip][slot]
    __shared__ float sDFB[2][32];
    __shared__ __attribute__((aligned(16))) float4 sHX[X4_NGMAX][2][28][4];   // the head step's boxes (24 k-quads) and p (4 slot quads) x 4 clips, by step parity
    __shared__ int sAbort, sLocal;          // through XCD_LDS_LD / XCD_LDS_ST: ds_read / ds_write (a volatile LDS word is a FLAT access + vmcnt wait)

    // the forward of this step gave up (sticky abort word, see opnet_xcd4_init_bwd): its histories are partial - leave
    if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
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
    for (int i = tid; i < X4_NGMAX * 64; i += 256) { (&sDC2[0][0])[i] = 0.f; (&sDL[0][0][0][0])[i] = 0.f; (&sDL[0][0][0][0])[X4_NGMAX * 64 + i] = 0.f; }
    for (int i = tid; i < X4_NGMAX * 32; i += 256) (&sDC1[0][0])[i] = 0.f;
    if (tid < 128) (&sDA2[0][0][0])[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 32) (&sDA1[0][0])[tid] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- resident weights ------------------------------------------------------------------------------------------------
    const X4BPacked P = x4b_packed_layout();
    float4 b2[32], b1[8], bxa, bxb, wo, wsl[4];
    {
        const float4 *p2 = (const float4 *)(a.pk + P.b2) + ((size_t)(c * 4 + w) * 32) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 32; ++q) b2[q] = p2[q * 64];
        const float4 *p1 = (const float4 *)(a.pk + P.b1) + ((size_t)(c * 4 + w) * 8) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) b1[q] = p1[q * 64];
        const float4 *px = (const float4 *)(a.pk + P.bx) + (size_t)(c * 2) * 64 + lane;
        bxa = px[0];
        bxb = px[64];
        wo = ((const float4 *)(a.pk + P.bo))[(size_t)c * 64 + lane];
        const float4 *ps = (const float4 *)(a.pk + P.bs) + (size_t)(c * 4) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) wsl[q] = ps[q * 64];
    }

    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void *)a.ws, 0, 0x7fffffff, 0x00020000);
    const unsigned cb = 4 * x;
    const unsigned lane16 = lane * 16;
    const xcd_u32x4 sent4 = {X4_SENT, X4_SENT, X4_SENT, X4_SENT};
    const float4 sentf = x4_as_float4(sent4);
    bool alive = true;
    float hsum = 0.f;                           // wave 2: its lane's share of the next head step's dfb

    // The inputs of phase (gi, n): this CU's chunks of the partial products of da2_{T-n} (8 KB; waited for while LSTM2 still runs:
    // n <= T - 1) and of da1_{T+3-n} (4 KB; n >= 3): twelve 1-KB pieces, four each by waves 0, 1 (dh2) and 3 (dh1)
    auto gather = [&](int gi, int n, int buf, int phase) -> bool {
        if (w == 2) return true;
        float4 *S = &sbuf[buf][0] + lane;
        const unsigned gg = gi * 8 + x;
        if (w == 3)
            return x4_gather_sum4(rws, lane16, a.p1x_off + ((gg * X4_SLOTS + ((T + 3 - n) & 3)) * 32 + c) * 4096, n >= 3,
                                  S + X4D_P1, a.status, phase);
        return x4_gather_sum4(rws, lane16, a.p2x_off + ((gg * X4_SLOTS + ((T - n) & 3)) * 32 + c) * 8192 + w * 4096,
                              n <= T - 1, S + X4D_P2 + w * 64, a.status, phase);
    };

    // the saved activations a cell needs (gates, c_t, c_{t-1}, wave 0: dy), fetched one phase ahead - they come from HBM / the
    // Infinity Cache
    float4 cg = make_float4(0.f, 0.f, 0.f, 0.f), cdy = cg, ng_ = cg, ndy = cg;      // gates, dy of the current / next phase
    float cct = 0.f, ccp = 0.f, nct = 0.f, ncp = 0.f;                                   // c_t, c_{t-1}
    // (pointer loads: as buffer loads through the workspace's descriptor - the forward's history stores went that way - the cell
    // phase grew from 1 740 to 2 100 cycles: the compiler then orders them against the descriptor's stores and waits)
    auto fetch = [&](int gi, int n, float4 &g, float4 &dy, float &ct, float &cp) {
        const int rb = gi;
        if (w == 0) {
            const int t = T - 1 - n;
            if (t >= 0 && t < T) {
                const size_t u = 16 * c + b;
                g = ((const float4 *)(a.ws + a.g2_off))[(((size_t)t * RB + rb) * 512 + u) * 32 + cb + j];
                dy = ((const float4 *)(a.ws + a.dy_off))[((size_t)t * RB + rb) * 32 + cb + j];
                ct = ((const float *)(a.ws + a.c2_off))[(((size_t)(t + 1) * RB + rb) * 512 + u) * 32 + cb + j];
                cp = ((const float *)(a.ws + a.c2_off))[(((size_t)t * RB + rb) * 512 + u) * 32 + cb + j];
            }
        } else if (w == 1) {
            const int t = T + 2 - n;
            if (t >= 0 && t < T && lane < 32) {
                const size_t u = 8 * c + b;
                g = ((const float4 *)(a.ws + a.g1_off))[(((size_t)t * RB + rb) * 256 + u) * 32 + cb + j];
                ct = ((const float *)(a.ws + a.c1_off))[(((size_t)(t + 1) * RB + rb) * 256 + u) * 32 + cb + j];
                cp = ((const float *)(a.ws + a.c1_off))[(((size_t)t * RB + rb) * 256 + u) * 32 + cb + j];
            }
        }
    };

    if (!gather(0, 0, 0, 0)) XCD_LDS_ST(sAbort, 1);
    fetch(0, 0, cg, cdy, cct, ccp);
    __syncthreads();
    if (XCD_LDS_LD(sAbort)) return;
    int abort_seen = 0;                         // the abort word as read behind the PREVIOUS phase's last barrier (see the loop's end)
    // the loop's wave-uniform, loop-invariant conditions (debug switches, placement, tracing, the wave's role) as bits of one scalar that
    // is made opaque at the top of every phase, w likewise: hoisted out of the loop as 64-bit lane masks they - and the workspace's
    // buffer descriptor with them - were spilled into VGPR lanes and fetched back with ~35 v_readlane per phase (opnet_xcd_kernels.hip)
    const unsigned cfbits = ((unsigned)a.debug & 0xffffu) | (__builtin_amdgcn_readfirstlane(XCD_LDS_LD(sLocal)) != 0 ? 0x10000u : 0u)
                            | ((a.trace && blockIdx.x == 0) ? 0x20000u : 0u);
    const int nph = (T + 3) * ng;

    int gi = 0, n = 0, gprev = 0, nprev = -1;      // (gprev, nprev): the previous phase
    for (int p = 0; p < nph; ++p) {
        unsigned cfq = __builtin_amdgcn_readfirstlane(cfbits);
        int wq = w;
        asm volatile("" : "+s"(cfq), "+s"(wq));
        const bool local = (cfq & 0x10000u) != 0, tracer = (cfq & 0x20000u) != 0 && tid == 0;
        const int buf = p & 1;
        int gn = gi + 1, nn = n;
        if (gn == ng) { gn = 0; ++nn; }
        const bool more = p + 1 < nph;
        const unsigned gg = gi * 8 + x;
        const int rb = gi;
        if (tracer) a.trace[(long)p * 8 + 0] = clock64();
        if (more) fetch(gn, nn, ng_, ndy, nct, ncp);
        // ================================ the cells (and the head backward), by wave ========================================
        const float *PF = (const float *)&sbuf[buf][0];
        if (wq == 0) {
            // ---- LSTM2 cell backward at t = T-1-n: lane = (unit 16 c + b, clip j): component b & 3 of the float4 (unit quad b >> 2,
            //      clip j) of every producer's chunk ---------------------------------------------------------------------------
            const int t = T - 1 - n;
            float4 da = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0 && t < T && !(cfq & 4)) {
                // the 32 producers' partials: the gathering waves have added them four by four - 2 waves x 4 lane groups are left
                float r0 = 0.f, r1 = 0.f;
                const float *pr = PF + X4D_P2 * 4 + ((b >> 2) * 4 + j) * 4 + (b & 3);
#pragma unroll
                for (int q = 0; q < 4; ++q) { r0 += pr[q * 64]; r1 += pr[256 + q * 64]; }
                // upstream: prediction_layer (learned_models.py:47): dh += W_out^T dy_t
                float dh = wo.x * cdy.x;
                dh = fmaf(wo.y, cdy.y, dh);
                dh = fmaf(wo.z, cdy.z, dh);
                dh = fmaf(wo.w, cdy.w, dh);
                dh += r0 + r1;
                float dco;
                da = cell_backward(dh, sDC2[gi][lane], cg, cct, ccp, &dco);
                sDC2[gi][lane] = dco;
                sDA2[buf][b][j] = da;
                // da replaces the saved gates (the weight-gradient GEMMs read it there)
                if (!(cfq & 2))
                ((float4 *)(a.ws + a.g2_off))[(((size_t)t * RB + rb) * 512 + 16 * c + b) * 32 + cb + j] = da;
            } else sDA2[buf][b][j] = da;
        } else if (wq == 1) {
            // ---- LSTM1 cell backward at t = T+2-n: lanes 0..31 = (unit 8 c + b, clip j) ------------------------------------------
            const int t = T + 2 - n;
            float4 da = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0 && t < T && !(cfq & 4)) {
                float rec = 0.f;
                const int lb = (lane & 31) >> 2;
                const float *pr = PF + X4D_P1 * 4 + ((lb >> 2) * 4 + j) * 4 + (lb & 3);       // 8 lane groups of the summed piece
#pragma unroll
                for (int q = 0; q < 8; ++q) rec += pr[q * 32];
                if (lane < 32) {
                    // upstream: object_to_track_prediction (learned_models.py:40): dh += W_sel^T dl_t
                    const float4 *dl = (const float4 *)&sDL[gi][(n + 1) & 1][j][0];   // written by wave 2 one phase ago
                    float dh = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 d = dl[q];
                        dh = fmaf(wsl[q].x, d.x, dh);
                        dh = fmaf(wsl[q].y, d.y, dh);
                        dh = fmaf(wsl[q].z, d.z, dh);
                        dh = fmaf(wsl[q].w, d.w, dh);
                    }
                    dh += rec;
                    float dco;
                    da = cell_backward(dh, sDC1[gi][lane], cg, cct, ccp, &dco);
                    sDC1[gi][lane] = dco;
                    if (!(cfq & 2))
                    ((float4 *)(a.ws + a.g1_off))[(((size_t)t * RB + rb) * 256 + 8 * c + b) * 32 + cb + j] = da;
                }
            }
            if (lane < 32) sDA1[b][j] = da;
        }
        if (wq == 2) {
            // ---- head backward at t = T+1-n (dfb_t was summed at the end of the previous phase): dp, dl_t -> LDS for wave 1 one phase on --
            const int t = T + 1 - n;
            if (t >= 0 && t < T && !(cfq & 8)) {
                const float sum = hsum;
                    // this step's slot probabilities and boxes (lanes 0..15 = (slot quad rg, clip j)): left in LDS by wave 3 one
                    // phase ago (they come from HBM)
                    float4 hp = make_float4(0.f, 0.f, 0.f, 0.f), hx[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) hx[q] = hp;
                    if (lane < 16) {
                        const int rg = lane >> 2;
                        hp = sHX[gi][n & 1][24 + rg][j];
#pragma unroll
                        for (int q = 0; q < 6; ++q) hx[q] = sHX[gi][n & 1][6 * rg + q][j];
                    }
                    sDFB[lane >> 5][lane & 31] = sum;
                    XCD_WAVE_LDS_SYNC();
                    if (lane < 16) {
                        const int rg = lane >> 2;
                        float dfb[OPNET_FEATS_];
#pragma unroll
                        for (int f = 0; f < OPNET_FEATS_; ++f) dfb[f] = sDFB[0][f * 4 + j] + sDFB[1][f * 4 + j];
                        // einsum backward: dp[o] = sum_f boxes[o][f] dfb[f]; softmax backward: dl = p * (dp - <p, dp>)
                        const float xf[24] = {hx[0].x, hx[0].y, hx[0].z, hx[0].w, hx[1].x, hx[1].y, hx[1].z, hx[1].w,
                                              hx[2].x, hx[2].y, hx[2].z, hx[2].w, hx[3].x, hx[3].y, hx[3].z, hx[3].w,
                                              hx[4].x, hx[4].y, hx[4].z, hx[4].w, hx[5].x, hx[5].y, hx[5].z, hx[5].w};
                        const float pv[4] = {hp.x, hp.y, hp.z, hp.w};
                        float dp[4], dot = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float acc = 0.f;
#pragma unroll
                            for (int f = 0; f < OPNET_FEATS_; ++f) acc = fmaf(xf[6 * r + f], dfb[f], acc);
                            dp[r] = acc;
                            dot = fmaf(pv[r], acc, dot);          // p of the non-existent slot 15 is 0
                        }
                        dot += x4_xor4(dot);
                        dot += x4_xor8(dot);
                        float4 dl;
                        dl.x = pv[0] * (dp[0] - dot);
                        dl.y = pv[1] * (dp[1] - dot);
                        dl.z = pv[2] * (dp[2] - dot);
                        dl.w = rg < 3 ? pv[3] * (dp[3] - dot) : 0.f;
                        *(float4 *)&sDL[gi][n & 1][j][4 * rg] = dl;
                        if (c == (n & 31))
                            ((float4 *)(a.ws + a.dl_off))[(((size_t)t * RB + rb) * 4 + rg) * 32 + cb + j] = dl;
                    }
            }
        }
        if (wq == 3 && nprev >= 0) {
            // ---- the CU's part of dfb_t = W_ih2^T da2_t of the PREVIOUS phase's LSTM2 step (its da2 is still in LDS): MFMA block = unit,
            //      k = gate, B = da2; D[unit b][feature][clip j], then the sum over the 16 units through LDS.  Kept off wave 0's
            //      path to the barrier (-700 cycles there), which is why the head runs two steps behind LSTM2.  Re-armed THREE steps
            //      on (its reader, wave 2, runs beside the publishing waves of its phase).
            const int t = T - 1 - nprev;
            if (t >= 0 && t < T && !(cfq & 4)) {
                const unsigned gp = gprev * 8 + x;
                const float4 da = sDA2[buf ^ 1][b][j];
                x4_f32x4 d1 = {0.f, 0.f, 0.f, 0.f}, d2 = {0.f, 0.f, 0.f, 0.f};
                X4_MFMA(d1, bxa.x, da.x); X4_MFMA(d2, bxb.x, da.x);
                X4_MFMA(d1, bxa.y, da.y); X4_MFMA(d2, bxb.y, da.y);
                X4_MFMA(d1, bxa.z, da.z); X4_MFMA(d2, bxb.z, da.z);
                X4_MFMA(d1, bxa.w, da.w); X4_MFMA(d2, bxb.w, da.w);
                sX[0][lane] = make_float4(d1[0], d1[1], d1[2], d1[3]);
                sX[1][lane] = make_float4(d2[0], d2[1], d2[2], d2[3]);
                XCD_WAVE_LDS_SYNC();
                if (lane < 32) {                // lane = (feature f = lane >> 2, clip j)
                    const float *px = (const float *)&sX[lane >> 4][0] + j * 4 + ((lane >> 2) & 3);
                    float sum = 0.f;
#pragma unroll
                    for (int u = 0; u < 16; ++u) sum += px[u * 16];
                    xcd_store4(rws, lane * 4, a.dfx_off + ((gp * X4_SLOTS + (t & 3)) * 32 + c) * 128, sum, local);
                    xcd_store4(rws, lane * 4, a.dfx_off + ((gp * X4_SLOTS + ((t + 3) & 3)) * 32 + c) * 128, sentf.x, local);
                }
            }
        }
        // wave 3: the next head step's boxes and p, from HBM into registers now, into LDS at the end of the phase
        float4 hxa = make_float4(0.f, 0.f, 0.f, 0.f), hxb = hxa;
        const int tn = T + 1 - nn;
        const bool hfetch = wq == 3 && more && tn >= 0 && tn < T;
        if (hfetch) {
            // lane = (k-quad or slot quad q, clip j): boxes k-quads 0..15 | boxes k-quads 16..23 and p
            const float4 *xs = (const float4 *)(a.ws + a.xp_off) + ((size_t)tn * RB + gn) * (OPNET_KXQ * 32) + cb + j;
            hxa = xs[b * 32];
            if (b < 8) hxb = xs[(16 + b) * 32];
            else if (b < 12) hxb = ((const float4 *)(a.ws + a.ps_off))[(((size_t)tn * RB + gn) * 4 + (b - 8)) * 32 + cb + j];
        }
        if (tracer) a.trace[(long)p * 8 + 1] = clock64();
        __syncthreads();                        // barrier 1: the CU's da of the phase is in LDS
        if (tracer) a.trace[(long)p * 8 + 2] = clock64();
        // (an abort can only be raised in the second half of a phase: it is looked at after barrier 2)
        // ================================ products: the CU's gate columns x its da -> partial dh rows of every unit ==========
        if (!(cfq & 16)) {
            const float4 *F2 = &sDA2[buf][0][0] + j, *F1 = &sDA1[0][0] + j;
            // one accumulator chain per (row set, fragment element): a dependent v_mfma_f32_4x4x1 can issue every ~30 cycles, so with the
            // three chains of the first version (d2a / d2b alternating, d1 alone) the 160 MFMAs took ~2 900 cycles instead of the
            // pipe's 1 500.  (W_hh2's columns are AccVGPR operands, as in the forward.)
            x4_f32x4 e2a[4], e2b[4], e1[4];     // (not cleared: the first MFMA of every chain has C = 0)
            float4 bf[X4_RING];
            // 24 B fragments: 16 of da2 (each feeds set 0 and set 1: 8 MFMAs) | 8 of da1 (4 MFMAs), X4_AHEAD ahead
            auto frag = [&](int idx) -> const float4 * { return idx < 16 ? F2 + idx * 4 : F1 + (idx - 16) * 4; };
#pragma unroll
            for (int i = 0; i < X4_AHEAD; ++i) bf[i] = *frag(i);
#pragma unroll
            for (int idx = 0; idx < 24; ++idx) {
                if (idx + X4_AHEAD < 24) bf[(idx + X4_AHEAD) % X4_RING] = *frag(idx + X4_AHEAD);
                __builtin_amdgcn_sched_barrier(0);
                const float4 bq = bf[idx % X4_RING];
                if (idx == 0) {
                    const float4 wa = b2[0], wb = b2[16];
                    X4_MFMA0_W(e2a[0], wa.x, bq.x); X4_MFMA0_W(e2b[0], wb.x, bq.x);
                    X4_MFMA0_W(e2a[1], wa.y, bq.y); X4_MFMA0_W(e2b[1], wb.y, bq.y);
                    X4_MFMA0_W(e2a[2], wa.z, bq.z); X4_MFMA0_W(e2b[2], wb.z, bq.z);
                    X4_MFMA0_W(e2a[3], wa.w, bq.w); X4_MFMA0_W(e2b[3], wb.w, bq.w);
                } else if (idx < 16) {
                    const float4 wa = b2[idx < 16 ? idx : 0], wb = b2[idx < 16 ? 16 + idx : 16];
                    X4_MFMA_W(e2a[0], wa.x, bq.x); X4_MFMA_W(e2b[0], wb.x, bq.x);
                    X4_MFMA_W(e2a[1], wa.y, bq.y); X4_MFMA_W(e2b[1], wb.y, bq.y);
                    X4_MFMA_W(e2a[2], wa.z, bq.z); X4_MFMA_W(e2b[2], wb.z, bq.z);
                    X4_MFMA_W(e2a[3], wa.w, bq.w); X4_MFMA_W(e2b[3], wb.w, bq.w);
                } else if (idx == 16) {
                    const float4 wc = b1[0];
                    X4_MFMA0_V(e1[0], wc.x, bq.x);
                    X4_MFMA0_V(e1[1], wc.y, bq.y);
                    X4_MFMA0_V(e1[2], wc.z, bq.z);
                    X4_MFMA0_V(e1[3], wc.w, bq.w);
                } else {
                    const float4 wc = b1[idx - 16];
                    X4_MFMA_V(e1[0], wc.x, bq.x);
                    X4_MFMA_V(e1[1], wc.y, bq.y);
                    X4_MFMA_V(e1[2], wc.z, bq.z);
                    X4_MFMA_V(e1[3], wc.w, bq.w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            X4_MFMA_DRAIN8(e2a[0], e2a[1], e2a[2], e2a[3], e2b[0], e2b[1], e2b[2], e2b[3]);
            asm volatile("" : "+v"(e1[0]), "+v"(e1[1]), "+v"(e1[2]), "+v"(e1[3]));
            const x4_f32x4 d2a = (e2a[0] + e2a[1]) + (e2a[2] + e2a[3]), d2b = (e2b[0] + e2b[1]) + (e2b[2] + e2b[3]),
                           d1 = (e1[0] + e1[1]) + (e1[2] + e1[3]);
            // lane (block bb, clip j) holds rows 4 bb .. 4 bb + 3 of its row set = one float4 of the owner's chunk:
            //   dh2 row 128 w + 64 set + 4 bb + i -> owner 8 w + 4 set + (bb >> 2), unit quad bb & 3
            //   dh1 row  64 w + 4 bb + i          -> owner 8 w + (bb >> 1),         unit quad bb & 1
            const int t2 = T - 1 - n, t1 = T + 2 - n;
            if (t2 >= 0 && t2 < T) {
                const unsigned vo = (((b >> 2) * 32) * 16 + (b & 3) * 4 + j) * 16;       // owner stride 32 producers x 16 float4
                const unsigned so = a.p2x_off + (((gg * X4_SLOTS + (t2 & 3)) * 32 + 8 * w) * 32 + c) * 256;
                const unsigned sr = a.p2x_off + (((gg * X4_SLOTS + ((t2 + 2) & 3)) * 32 + 8 * w) * 32 + c) * 256;
                xcd_store16(rws, vo, so, make_float4(d2a[0], d2a[1], d2a[2], d2a[3]), local);
                xcd_store16(rws, vo, so + 4 * 32 * 256, make_float4(d2b[0], d2b[1], d2b[2], d2b[3]), local);
                xcd_store16(rws, vo, sr, sentf, local);
                xcd_store16(rws, vo, sr + 4 * 32 * 256, sentf, local);
            }
            if (t1 >= 0 && t1 < T) {
                const unsigned vo = (((b >> 1) * 32) * 8 + (b & 1) * 4 + j) * 16;        // owner stride 32 producers x 8 float4
                const unsigned so = a.p1x_off + (((gg * X4_SLOTS + (t1 & 3)) * 32 + 8 * w) * 32 + c) * 128;
                const unsigned sr = a.p1x_off + (((gg * X4_SLOTS + ((t1 + 2) & 3)) * 32 + 8 * w) * 32 + c) * 128;
                xcd_store16(rws, vo, so, make_float4(d1[0], d1[1], d1[2], d1[3]), local);
                xcd_store16(rws, vo, sr, sentf, local);
            }
        }
        if (tracer) a.trace[(long)p * 8 + 3] = clock64();
        // ================================ wave 2: the next head step's dfb = sum of the 32 CUs' parts (published in this phase's
        //                                  first half, i.e. long ago), fetched now so that the head backward is pure arithmetic ====
        if (wq == 2 && more && alive && !(cfq & 8)) {
            const int t = T + 1 - nn;
            hsum = 0.f;
            if (t >= 0 && t < T) {
                // lane = (CU half h, feature f, clip j): 16 CUs each
                const unsigned base = a.dfx_off + (((gn * 8 + x) * X4_SLOTS + (t & 3)) * 32) * 128;   // wave-uniform: the lane half goes into the lane offset
                const unsigned hvo = (lane & 31) * 4 + (lane >> 5) * (16 * 128);
                long long t0 = 0;
                for (unsigned spins = 1;; ++spins) {
                    unsigned v[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b32(rws, hvo, base + q * 128, 16);   // sc1
                    bool bad = false;
                    hsum = 0.f;
#pragma unroll
                    for (int q = 0; q < 16; ++q) { bad |= v[q] == X4_SENT; hsum += __uint_as_float(v[q]); }
                    if (!__any(bad)) break;
                    if (!x4_keep_polling(spins, t0, a.status, p)) { alive = false; XCD_LDS_ST(sAbort, 1); break; }
                }
            }
        }
        // ================================ the next phase's inputs ==========================================================
        if (more && alive && !(cfq & 1)) {
            alive = gather(gn, nn, buf ^ 1, p);
            if (!alive) XCD_LDS_ST(sAbort, 1);
        }
        if (tracer) a.trace[(long)p * 8 + 4] = clock64();
        if (hfetch) {
            sHX[gn][nn & 1][b][j] = hxa;
            if (b < 12) sHX[gn][nn & 1][16 + b][j] = hxb;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // orders this phase's re-arm stores before the next publish
        if (tracer) a.trace[(long)p * 8 + 5] = clock64();
        if (a.trace && blockIdx.x == 0 && lane == 0) sArrT[w] = clock64();      // tools: when each wave reaches barrier 2
        __syncthreads();                        // barrier 2: the next phase's chunks have landed
        if (tracer) {
            a.trace[(long)p * 8 + 6] = clock64();
            const long long t0 = sArrT[0];      // waves 1..3 relative to wave 0, 16 bits each
            a.trace[(long)p * 8 + 7] = (unsigned long long)(((sArrT[1] - t0) & 0xffff) | (((sArrT[2] - t0) & 0xffff) << 16)
                                                            | (((sArrT[3] - t0) & 0xffff) << 32));
        }
        // the abort word is looked at one phase late: read here, behind the barrier, but only tested a phase on, so that the LDS round
        // trip stays off the step's critical chain (an aborted launch's outputs are NaN whatever this block still stores)
        if (abort_seen) return;
        abort_seen = XCD_LDS_LD(sAbort);
        gprev = gi;
        nprev = n;
        gi = gn;
        n = nn;
        cg = ng_; cdy = ndy; cct = nct; ccp = ncp;
    }
}
