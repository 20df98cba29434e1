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

// Optional in-kernel timeline (tools/step_probe.hip builds with -DOPNET_TRACE): wave 0 of every
// workgroup stamps s_memtime at fixed points of each step.  Compiled out of the product library.
#ifdef OPNET_TRACE
__device__ unsigned long long *g_trace = nullptr;  // [step][wg][8]
#define TRACE_STAMP(slot)                                                                         \
    do {                                                                                          \
        if (g_trace && threadIdx.x == 0)                                                          \
            g_trace[(((long)s * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y) * 8 + (slot)] = \
                (slot) == 0 ? wall_clock64() : clock64();                                         \
    } while (0)
#ifndef OPNET_VARIANT
#define OPNET_VARIANT 0
#endif
#if OPNET_VARIANT == 3   /* stamps only: do not perturb the load/MFMA overlap */
#define TRACE_WAIT_LOADS() do { } while (0)
#else
#define TRACE_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#else
#define TRACE_STAMP(slot) do { } while (0)
#define TRACE_WAIT_LOADS() do { } while (0)
#endif

#ifndef OPNET_NW
#define OPNET_NW 4                 // waves per workgroup (K split)
#endif
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
// boundary kernels: caller's tensors <-> workspace
// ------------------------------------------------------------------------------------------------
__global__ void opnet_set_io(OpnetIO *dst, OpnetIO src) { *dst = src; }

// boxes [B][T][90] -> xp [t][rb][kq 0..23][clip 0..31][4]  (K padded 90 -> 96 with zeros, clips
// beyond B zero).  One workgroup per (t, rb).  The same launch zeroes the recurrent state (h0 = c0 =
// 0, learned_models.py:39,46 pass no initial state): it is the first kernel node of the forward, so
// the zeroing is ordered like every other kernel of the chain (a hipGraph memset root node is not
// ordered against the tail of a previous replay of the same graph on ROCm 7.2 - measured).
// (the body, for the workgroup (t, rb) of a T x nrb grid; also part of opnet_x4_train_prologue)
__device__ __forceinline__ void pack_input_body(const OpnetIO *__restrict__ io, int t, int rb, int nrb)
{
    const int B = io->B, T = io->T, RB = io->RB;
    {
        float4 *__restrict__ st = io->state;
        const long n = io->state_f4;
        const long nthreads = (long)T * nrb * 256;
        for (long i = ((long)rb * T + t) * 256 + threadIdx.x; i < n; i += nthreads)
            st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float *__restrict__ boxes = io->boxes;
    float4 *__restrict__ xp = io->xp + ((long)t * RB + rb) * (OPNET_KXQ * 32);
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

__global__ void __launch_bounds__(256) opnet_pack_input(const OpnetIO *__restrict__ io)
{
    pack_input_body(io, blockIdx.x, blockIdx.y, gridDim.y);
}

// staging -> caller's y [B][T][4] and logits [B][15][T] (both staging buffers use the same layouts
// with the clip count padded to whole row blocks, so this is two straight copies)
__global__ void __launch_bounds__(256) opnet_copy_out(const OpnetIO *__restrict__ io)
{
    const long ny = (long)io->B * io->T;               // float4 units
    const long nl = (long)io->B * OPNET_SLOTS_ * io->T;  // floats
    float4 *__restrict__ y = (float4 *)io->y;
    float *__restrict__ lg = io->logits;
    const float4 *__restrict__ ys = io->ystage;
    const float *__restrict__ ls = io->lgstage;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < ny; i += stride) y[i] = ys[i];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nl; i += stride) lg[i] = ls[i];
}

// ------------------------------------------------------------------------------------------------
// the MFMA core shared by all four roles
// ------------------------------------------------------------------------------------------------
// D[16 x 32] = A[16 x K] * Hsrc^T, K = 16*(nh0+nh1) taken from up to two activation segments (each
// [k/4][32][4] float4 for one row block).  Wave w reduces hexadecets [q0, q1) = its quarter of K.
// All fragment loads of a chunk (up to CH hexadecets = 3*CH wave-wide 1-KiB loads) are issued
// before the first MFMA so the memory round trips overlap instead of serialising per hexadecet.
// A workgroup that serves several row blocks keeps its A fragments (the weights) in registers
// across them when its K slice fits one chunk (true for H1=256/H2=512: 4..8 hexadecets a wave).
// part layout in LDS: [wave][clip half][lane][acc reg 0..3]; half 0 = clips 0..15, half 1 = clips 16..31.
#ifndef OPNET_CH
#define OPNET_CH 8
#endif

struct KSlice {
    int q0, q1;  // this wave's hexadecet range (wave-uniform)
};

template <int NW = OPNET_NW>
__device__ __forceinline__ KSlice wave_slice(int nhex)
{
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    KSlice k;
    k.q0 = (w * nhex) / NW;
    k.q1 = ((w + 1) * nhex) / NW;
    return k;
}

// CH = hexadecets of a register chunk, deduced from the caller's fragment array.  8 covers a wave's whole K slice of the
// OPNet roles in one fetch round trip (best at B <= 32); 4 keeps the step kernel at 140 instead of 204 VGPRs, i.e. 3
// resident waves per SIMD - which wins once several row blocks make the launch compute-bound (B >= 64).
// Fragment loads go through buffer descriptors: the wave-uniform part of every address (tile base, hexadecet) rides in
// SGPRs (descriptor + soffset) and ONE VGPR holds the lane's byte offset, instead of a 64-bit VGPR address pair per
// load that the compiler precomputes - and keeps live - for each of the up to 24 loads of a chunk.
typedef unsigned opnet_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frag_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffffu, 0x00020000);
}
__device__ __forceinline__ float4 frag_load(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const opnet_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int CH>
__device__ __forceinline__ void load_a_chunk(float4 (&a)[CH], const float4 *__restrict__ A, int qb, int q1)
{
    const int lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t ra = frag_rsrc(A);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        if (qb + j < q1) {
#if (defined(OPNET_TRACE) && OPNET_VARIANT == 2) || defined(OPNET_ABLATE_A)   /* probe: no weight loads */
            a[j] = make_float4(1.f, 2.f, 3.f, (float)(qb + j));
#else
            a[j] = frag_load(ra, lane * 16, (qb + j) * 1024);
#endif
        }
    }
}

// two_halves = false: the row block holds <= 16 clips, so the second accumulator chain (clips 16..31,
// all padding) is skipped - half the loads and MFMAs for small batches (B <= 16, e.g. one 300-frame clip).
template <int CH>
__device__ __forceinline__ void mma_chunk(const float4 (&a)[CH], const float4 *__restrict__ seg0, int nh0,
                                          const float4 *__restrict__ seg1, int qb, int q1,
                                          f32x4 &acc0, f32x4 &acc1, const int s, const bool two_halves)
{
    const int lane = threadIdx.x & 63;
    const int boff = ((lane >> 4) * 32 + (lane & 15)) * 16;
    const __amdgpu_buffer_rsrc_t r0 = frag_rsrc(seg0), r1 = frag_rsrc(seg1);
    float4 b0[CH], b1[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int q = qb + j;
        if (q < q1) {  // wave-uniform
            const bool first = q < nh0;
            const int soff = (first ? q : q - nh0) * 2048;
#if (defined(OPNET_TRACE) && OPNET_VARIANT == 1) || defined(OPNET_ABLATE_B)   /* probe: no activation loads */
            b0[j] = make_float4(1.f, 2.f, 3.f, (float)q);
            b1[j] = make_float4(1.f, 2.f, 3.f, (float)lane);
#else
            b0[j] = first ? frag_load(r0, boff, soff) : frag_load(r1, boff, soff);
            if (two_halves) b1[j] = first ? frag_load(r0, boff + 256, soff) : frag_load(r1, boff + 256, soff);
#endif
        }
    }
    TRACE_WAIT_LOADS();
    TRACE_STAMP(2);
#ifdef OPNET_ABLATE_MFMA   /* probe: loads only (kept alive), no matrix work */
#pragma unroll
    for (int j = 0; j < CH; ++j)
        if (qb + j < q1) {
            asm volatile("" ::"v"(a[j].x), "v"(b0[j].x), "v"(b1[j].x), "v"(a[j].w), "v"(b0[j].w), "v"(b1[j].w));
            acc0[0] += a[j].x + b0[j].y; acc1[0] += b1[j].z;
        }
    return;
#endif
    if (!two_halves) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (qb + j < q1) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b0[j].x, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b0[j].y, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b0[j].z, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b0[j].w, acc0, 0, 0, 0);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        if (qb + j < q1) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b0[j].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b1[j].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b0[j].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b1[j].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b0[j].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b1[j].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b0[j].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b1[j].w, acc1, 0, 0, 0);
        }
    }
}

// One row block.  `a0` holds the A chunk that starts at hexadecet `a_qb` (the caller loads the wave's first chunk
// before its row-block loop).  When the K slice fits one chunk the weights simply stay in `a0` across row blocks; when
// it does not, the further chunks are streamed through the SAME registers (and the first one is fetched again for the
// next row block) - a second chunk array next to a persistent first one made this rarely-taken path, not the
// one-chunk path that real OPNet sizes use, set the kernel's register allocation (191 VGPRs).
template <int CH>
__device__ __forceinline__ void gemm16_rb(float4 (&a0)[CH], int &a_qb, const float4 *__restrict__ A,
                                          const float4 *__restrict__ seg0, int nh0,
                                          const float4 *__restrict__ seg1, const KSlice ks,
                                          float *__restrict__ part, const int s, const bool two_halves = true)
{
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
    if (a_qb != ks.q0) {  // wave-uniform
        load_a_chunk(a0, A, ks.q0, ks.q1);
        a_qb = ks.q0;
    }
    mma_chunk(a0, seg0, nh0, seg1, ks.q0, ks.q1, acc0, acc1, s, two_halves);
    for (int qb = ks.q0 + CH; qb < ks.q1; qb += CH) {
        load_a_chunk(a0, A, qb, ks.q1);
        a_qb = qb;
        mma_chunk(a0, seg0, nh0, seg1, qb, ks.q1, acc0, acc1, s, two_halves);
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // [wave][clip half][lane][4 acc regs]: one 16-byte LDS access per accumulator on both sides, conflict-free
    float4 *p = (float4 *)part + (w * 2) * 64 + lane;
    p[0] = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
    p[64] = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
    TRACE_STAMP(3);
}

// fixed-order cross-wave reduction of D element (reg, lane)
template <int NW = OPNET_NW>
__device__ __forceinline__ float part_sum(const float *__restrict__ part, int reg, int lane)
{
    const int o = (((reg >> 2) * 64) + lane) * 4 + (reg & 3);
    float s = part[o];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += part[w * 512 + o];
    return s;
}

// Gate non-linearities on the hardware transcendental path: v_exp_f32 / v_rcp_f32 are 1-ulp
// class, giving |error| ~1e-7 on values in (-1, 1) - the same order as the fp32 rounding of the
// recurrence itself (measured drift vs the fp64 oracle is unchanged to the digit).
#ifndef OPNET_FAST_GATES
#define OPNET_FAST_GATES 1
#endif
#if OPNET_FAST_GATES
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
#else
__device__ __forceinline__ float fast_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return tanhf(x); }
#endif

// LSTM cell update for one (clip, unit): gates i,f,g,o -> (c, h)
__device__ __forceinline__ float lstm_cell(float gi, float gf, float gg, float go, float *c_io)
{
    const float i = fast_sigmoid(gi);
    const float f = fast_sigmoid(gf);
    const float g = fast_tanh(gg);
    const float o = fast_sigmoid(go);
    const float c = f * (*c_io) + i * g;
    *c_io = c;
    return o * fast_tanh(c);
}

// history slots: inference keeps two parity slots per state buffer, training keeps every step
__device__ __forceinline__ long slot_prev(const StepArgs &a, int t) { return a.train ? t : ((t + 1) & 1); }
__device__ __forceinline__ long slot_out(const StepArgs &a, int t) { return a.train ? t + 1 : (t & 1); }
__device__ __forceinline__ long slot_x2(const StepArgs &a, int t) { return a.train ? t : (t & 1); }
__device__ __forceinline__ long cslot_prev(const StepArgs &a, int t) { return a.train ? t : 0; }
__device__ __forceinline__ long cslot_out(const StepArgs &a, int t) { return a.train ? t + 1 : 0; }

// LSTM cell update that also returns the post-activation gates (saved for the backward pass)
__device__ __forceinline__ float lstm_cell_g(float gi, float gf, float gg, float go, float *c_io, float4 *gates)
{
    const float i = fast_sigmoid(gi);
    const float f = fast_sigmoid(gf);
    const float g = fast_tanh(gg);
    const float o = fast_sigmoid(go);
    const float c = f * (*c_io) + i * g;
    *c_io = c;
    *gates = make_float4(i, f, g, o);
    return o * fast_tanh(c);
}

// ---------------- selection head, step t = s-1 (learned_models.py:40-43,50) -------------
template <int CH, int NW = OPNET_NW>
__device__ __forceinline__ void role_selection_head(const StepArgs &a, const int s, float *part, float *lg)
{
    const int T = a.T, H1 = a.H1;
    const int tid = threadIdx.x;
    const int el = tid & 63, half = tid >> 6;
    const int clip = half * 16 + (el & 15), quarter = el >> 4;
    float4 a0[CH];
    const int t = s - 1;
    if (t < 0 || t >= T) return;
    const int nh = H1 >> 4;
    const KSlice ks = wave_slice<NW>(nh);
    load_a_chunk(a0, a.wselp, ks.q0, ks.q1);
    int a_qb = ks.q0;
    const int mc = tid >> 3, mf = tid & 7;
    for (int rb = blockIdx.y; rb < a.RB; rb += gridDim.y) {
        const float4 *hcur = a.h1buf + (slot_out(a, t) * a.RB + rb) * (H1 * 8);
        // this frame's boxes, for the mix below: feature f of slot o of clip c sits at k = 6*o + f
        // of the packed LSTM1 input.  Issued before the MFMA phase.
        float bxv[OPNET_SLOTS_];
        if (tid < 256) {
            const float *xs = (const float *)(a.xp + ((long)t * a.RB + rb) * (OPNET_KXQ * 32));
#pragma unroll
            for (int o = 0; o < OPNET_SLOTS_; ++o) {
                const int k = o * OPNET_FEATS_ + (mf < OPNET_FEATS_ ? mf : 0);
                bxv[o] = xs[((k >> 2) * 32 + mc) * 4 + (k & 3)];
            }
        }
        gemm16_rb(a0, a_qb, a.wselp, hcur, nh, hcur, ks, part, s, a.B - rb * 32 > 16);
        __syncthreads();
        TRACE_STAMP(4);
        if (tid < 128) {
            // thread (clip, quarter) holds logits of slots 4*quarter .. +3; the 15-way softmax
            // (F.softmax(dim=-1), :41) spans the four lanes el, el^16, el^32, el^48
            const long b = rb * 32 + clip;
            float v[4];
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int slot = quarter * 4 + r;
                v[r] = part_sum<NW>(part, half * 4 + r, el);
                // logits [B][15][T] (the permute(0,2,1).contiguous() of :50)
                if (slot < OPNET_SLOTS_) {
                    a.lgstage[(b * OPNET_SLOTS_ + slot) * T + t] = v[r];
                    m = fmaxf(m, v[r]);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            float e[4], sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = (quarter * 4 + r < OPNET_SLOTS_) ? __expf(v[r] - m) : 0.f;
                sum += e[r];
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            float4 pv = make_float4(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv);
            ((float4 *)lg)[clip * 4 + quarter] = pv;
            if (a.train) a.psave[(((long)t * a.RB + rb) * 4 + quarter) * 32 + clip] = pv;
        }
        __syncthreads();
        if (tid < 256) {
            // frames_boxes[clip][f] = sum_o boxes[clip][t][o][f] * p[o]   (einsum "bfot,bfo->bft")
            float acc = 0.f;
            const float *p = lg + mc * 16;
#pragma unroll
            for (int o = 0; o < OPNET_SLOTS_; ++o) acc = fmaf(bxv[o], p[o], acc);
            float *x2 = (float *)(a.x2buf + (slot_x2(a, t) * a.RB + rb) * 64);
            x2[((mf >> 2) * 32 + mc) * 4 + (mf & 3)] = mf < OPNET_FEATS_ ? acc : 0.f;
        }
        if (rb + (int)gridDim.y < a.RB) __syncthreads();
    }
}

// ---------------- output head, step t = s-3 (prediction_layer, learned_models.py:33,47) --
template <int CH, int NW = OPNET_NW>
__device__ __forceinline__ void role_output_head(const StepArgs &a, const int s, float *part)
{
    const int T = a.T, H2 = a.H2;
    const int tid = threadIdx.x;
    const int el = tid & 63, half = tid >> 6;
    const int clip = half * 16 + (el & 15), quarter = el >> 4;
    float4 a0[CH];
    const int t = s - 3;
    if (t < 0 || t >= T) return;
    const int nh = H2 >> 4;
    const KSlice ks = wave_slice<NW>(nh);
    load_a_chunk(a0, a.woutp, ks.q0, ks.q1);
    int a_qb = ks.q0;
    for (int rb = blockIdx.y; rb < a.RB; rb += gridDim.y) {
        const float4 *hcur = a.h2buf + (slot_out(a, t) * a.RB + rb) * (H2 * 8);
        gemm16_rb(a0, a_qb, a.woutp, hcur, nh, hcur, ks, part, s, a.B - rb * 32 > 16);
        __syncthreads();
        TRACE_STAMP(4);
        if (tid < 128 && quarter == 0) {
            const long b = rb * 32 + clip;
            float4 v;
            v.x = part_sum<NW>(part, half * 4 + 0, el);
            v.y = part_sum<NW>(part, half * 4 + 1, el);
            v.z = part_sum<NW>(part, half * 4 + 2, el);
            v.w = part_sum<NW>(part, half * 4 + 3, el);
            a.ystage[b * T + t] = v;
        }
        if (rb + (int)gridDim.y < a.RB) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// the step kernel
// ------------------------------------------------------------------------------------------------
// grid.x = n2 (LSTM2 tiles, longest K first) + n1 (LSTM1 tiles) + 2 heads ; grid.y <= row blocks
// (a workgroup walks row blocks rb = blockIdx.y, blockIdx.y + gridDim.y, ... with its weights held
// in registers).
template <int CH, int NW>
__device__ __forceinline__ void opnet_step_body(const StepArgs &a, const int s)
{
    __shared__ __attribute__((aligned(16))) float lds[NW * 8 * 64 + 32 * 16];
    float *part = lds;
    float *lg = lds + NW * 8 * 64;  // [clip][16] slot probabilities (selection head)

    const int bx = blockIdx.x;
    const int T = a.T;
    const int H1 = a.H1, H2 = a.H2;
    const int n1 = H1 >> 2, n2 = H2 >> 2;
    const int tid = threadIdx.x;
    TRACE_STAMP(0);
    TRACE_STAMP(1);

    // epilogue coordinates (threads 0..127): D column = clip, D rows 4*(lane>>4)+r in regs r
    const int el = tid & 63;
    const int half = tid >> 6;
    const int clip = half * 16 + (el & 15);
    const int quarter = el >> 4;
    float4 a0[CH];

    if (bx < n2) {
        // ---------------- LSTM2 (video_LSTM, learned_models.py:32,46), step t = s-2 -------------
        const int t = s - 2;
        if (t < 0 || t >= T) return;
        const int tile = bx;
        const int nh = H2 >> 4;
        const KSlice ks = wave_slice<NW>(nh);
        const float4 *A = a.w2p + (long)tile * nh * 64;
        if (!a.mlp) load_a_chunk(a0, A, ks.q0, ks.q1);
        int a_qb = ks.q0;
        const int unit = tile * 4 + quarter;
        // the tile's 4 units x 4 gates x 8 input weights (512 B) wait in LDS for the epilogue instead of in 32
        // registers of every lane across the MFMA phase
        float4 *wl = (float4 *)lg;
        if (tid < 32) wl[tid] = a.wih2p[(long)tile * 32 + tid];
        for (int rb = blockIdx.y; rb < a.RB; rb += gridDim.y) {
            const float4 *hprev = a.h2buf + (slot_prev(a, t) * a.RB + rb) * (H2 * 8);
            // epilogue operands are fetched before the MFMA phase so their latency hides under it
            float4 xa, xb;
            float c_old = 0.f;
            if (tid < 128) {
                const float4 *x2 = a.x2buf + (slot_x2(a, t) * a.RB + rb) * 64 + clip;
                xa = x2[0];
                xb = x2[32];
                c_old = a.c2[((cslot_prev(a, t) * a.RB + rb) * H2 + unit) * 32 + clip];
            }
            if (a.mlp) {
                // hidden = relu(hidden_layer(frames_boxes)) (learned_models.py:83): no recurrence, the
                // weight rows sit in the "gate 0" slot of the packed x-part
                if (tid < 128) {
                    const float4 w0 = a.wih2p[(long)unit * 8], w1 = a.wih2p[(long)unit * 8 + 1];
                    float xs = w0.x * xa.x;
                    xs = fmaf(w0.y, xa.y, xs);
                    xs = fmaf(w0.z, xa.z, xs);
                    xs = fmaf(w0.w, xa.w, xs);
                    xs = fmaf(w1.x, xb.x, xs);
                    xs = fmaf(w1.y, xb.y, xs);
                    float *hout = (float *)(a.h2buf + (slot_out(a, t) * a.RB + rb) * (H2 * 8));
                    hout[((long)tile * 32 + clip) * 4 + quarter] = fmaxf(xs, 0.f);
                }
                continue;
            }
            gemm16_rb(a0, a_qb, A, hprev, nh, hprev, ks, part, s, a.B - rb * 32 > 16);
            __syncthreads();
            TRACE_STAMP(4);
            if (tid < 128) {
                float g[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // x part: W_ih2[gate r, unit][0..5] . frames_boxes[clip][0..5]
                    const float4 w0 = wl[quarter * 8 + 2 * r], w1 = wl[quarter * 8 + 2 * r + 1];
                    float xs = w0.x * xa.x;
                    xs = fmaf(w0.y, xa.y, xs);
                    xs = fmaf(w0.z, xa.z, xs);
                    xs = fmaf(w0.w, xa.w, xs);
                    xs = fmaf(w1.x, xb.x, xs);
                    xs = fmaf(w1.y, xb.y, xs);
                    g[r] = part_sum<NW>(part, half * 4 + r, el) + xs;
                }
                float c = c_old;
                float4 gs;
                const float h = lstm_cell_g(g[0], g[1], g[2], g[3], &c, &gs);
                a.c2[((cslot_out(a, t) * a.RB + rb) * H2 + unit) * 32 + clip] = c;
                if (a.train) a.g2save[(((long)t * a.RB + rb) * H2 + unit) * 32 + clip] = gs;
                float *hout = (float *)(a.h2buf + (slot_out(a, t) * a.RB + rb) * (H2 * 8));
                hout[((long)tile * 32 + clip) * 4 + quarter] = h;
            }
            if (rb + (int)gridDim.y < a.RB) __syncthreads();  // partials are rewritten next round
        }
    } else if (bx < n2 + n1) {
        // ---------------- LSTM1 (object_to_track_LSTM, learned_models.py:29,39), step t = s -----
        const int t = s;
        if (t >= T) return;
        const int tile = bx - n2;
        const int nhh = H1 >> 4;
        const KSlice ks = wave_slice<NW>(OPNET_KXQ / 4 + nhh);
        const float4 *A = a.w1p + (long)tile * (OPNET_KXQ / 4 + nhh) * 64;
        load_a_chunk(a0, A, ks.q0, ks.q1);
        int a_qb = ks.q0;
        const int unit = tile * 4 + quarter;
        for (int rb = blockIdx.y; rb < a.RB; rb += gridDim.y) {
            const float4 *xsrc = a.xp + ((long)t * a.RB + rb) * (OPNET_KXQ * 32);
            const float4 *hprev = a.h1buf + (slot_prev(a, t) * a.RB + rb) * (H1 * 8);
            float c_old = 0.f;
            if (tid < 128) c_old = a.c1[((cslot_prev(a, t) * a.RB + rb) * H1 + unit) * 32 + clip];
            gemm16_rb(a0, a_qb, A, xsrc, OPNET_KXQ / 4, hprev, ks, part, s, a.B - rb * 32 > 16);
            __syncthreads();
            TRACE_STAMP(4);
            if (tid < 128) {
                float c = c_old;
                float4 gs;
                const float h = lstm_cell_g(part_sum<NW>(part, half * 4 + 0, el), part_sum<NW>(part, half * 4 + 1, el),
                                            part_sum<NW>(part, half * 4 + 2, el), part_sum<NW>(part, half * 4 + 3, el), &c, &gs);
                a.c1[((cslot_out(a, t) * a.RB + rb) * H1 + unit) * 32 + clip] = c;
                if (a.train) a.g1save[(((long)t * a.RB + rb) * H1 + unit) * 32 + clip] = gs;
                float *hout = (float *)(a.h1buf + (slot_out(a, t) * a.RB + rb) * (H1 * 8));
                hout[((long)tile * 32 + clip) * 4 + quarter] = h;
            }
            if (rb + (int)gridDim.y < a.RB) __syncthreads();
        }
    } else if (bx == n2 + n1) {
        role_selection_head<CH, NW>(a, s, part, lg);
    } else if (bx == n2 + n1 + 1) {
        role_output_head<CH, NW>(a, s, part);
    }   // beyond: padding workgroups (the host rounds grid.x up to a multiple of the 8 XCDs)
    TRACE_STAMP(5);
}

template <int CH, int NW = OPNET_NW>
__global__ void __launch_bounds__(NW * 64) opnet_step(const StepArgs a, const int s)
{
    opnet_step_body<CH, NW>(a, s);
}

// The same step with SCALAR arguments only (two base pointers + the shape), so that the compiler's kernarg preloading
// (-mllvm -amdgpu-kernarg-preload-count) has the CP drop them into user SGPRs at dispatch: the by-value StepArgs costs a
// scalar-load round trip to memory (the L2s were just invalidated) before the first fragment load can even be issued.
// The buffer carving is recomputed on the scalar unit (opnet_ctx.h).  Inference forward only.
template <int CH, int NW>
__global__ void __launch_bounds__(NW * 64) opnet_step_pl(char *ws, const float *packed, int B, int T, int H1, int H2, int s)
{
    StepArgs a;
    step_args_inference(&a, ws, packed, B, T, H1, H2);
    opnet_step_body<CH, NW>(a, s);
}

// ------------------------------------------------------------------------------------------------
// the step kernel for several row blocks: two 16-row tiles per workgroup
// ------------------------------------------------------------------------------------------------
// With B >= 128 the narrow kernel is bound by every 16-row workgroup re-reading the whole activation block of each row
// block through its L1 (8 flop per byte filled).  Here a workgroup owns TWO adjacent tiles (32 gate
// rows = 8 hidden units): each wave's activation fragments feed both tiles' MFMAs, halving the fill traffic per flop, and
// the second tile's cell update runs on the 128 threads the narrow epilogue leaves idle.  Every accumulator sees exactly
// the narrow kernel's operation sequence, so the two kernels agree bit for bit.
// part layout in LDS: [wave][acc reg 0..15][lane]; regs 8*tile + 4*half + r.
template <int CH>
__device__ __forceinline__ void mma_chunk2(const float4 (&a0)[CH], const float4 (&a1)[CH], const float4 *__restrict__ seg0,
                                           int nh0, const float4 *__restrict__ seg1, int qb, int q1, f32x4 (&acc)[4],
                                           const bool two_halves)
{
    const int lane = threadIdx.x & 63;
    const int boff = ((lane >> 4) * 32 + (lane & 15)) * 16;
    const __amdgpu_buffer_rsrc_t r0 = frag_rsrc(seg0), r1 = frag_rsrc(seg1);
    float4 b0[CH], b1[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int q = qb + j;
        if (q < q1) {  // wave-uniform
            const bool first = q < nh0;
            const int soff = (first ? q : q - nh0) * 2048;
            b0[j] = first ? frag_load(r0, boff, soff) : frag_load(r1, boff, soff);
            if (two_halves) b1[j] = first ? frag_load(r0, boff + 256, soff) : frag_load(r1, boff + 256, soff);
        }
    }
#define OPNET_MMA2(c)                                                                            \
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j].c, b0[j].c, acc[0], 0, 0, 0);           \
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j].c, b0[j].c, acc[2], 0, 0, 0);           \
    if (two_halves) {                                                                           \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j].c, b1[j].c, acc[1], 0, 0, 0);       \
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j].c, b1[j].c, acc[3], 0, 0, 0);       \
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        if (qb + j < q1) {
            OPNET_MMA2(x) OPNET_MMA2(y) OPNET_MMA2(z) OPNET_MMA2(w)
        }
    }
#undef OPNET_MMA2
}

template <int CH>
__device__ __forceinline__ void gemm32_rb(float4 (&a0)[CH], float4 (&a1)[CH], int &a_qb, const float4 *__restrict__ A0,
                                          const float4 *__restrict__ A1, const float4 *__restrict__ seg0, int nh0,
                                          const float4 *__restrict__ seg1, const KSlice ks, float *__restrict__ part,
                                          const bool two_halves)
{
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a_qb != ks.q0) {  // wave-uniform
        load_a_chunk(a0, A0, ks.q0, ks.q1);
        load_a_chunk(a1, A1, ks.q0, ks.q1);
        a_qb = ks.q0;
    }
    mma_chunk2(a0, a1, seg0, nh0, seg1, ks.q0, ks.q1, acc, two_halves);
    for (int qb = ks.q0 + CH; qb < ks.q1; qb += CH) {
        load_a_chunk(a0, A0, qb, ks.q1);
        load_a_chunk(a1, A1, qb, ks.q1);
        a_qb = qb;
        mma_chunk2(a0, a1, seg0, nh0, seg1, qb, ks.q1, acc, two_halves);
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *p = part + (w * 16) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[(4 * i + 0) * 64] = acc[i][0]; p[(4 * i + 1) * 64] = acc[i][1];
        p[(4 * i + 2) * 64] = acc[i][2]; p[(4 * i + 3) * 64] = acc[i][3];
    }
}

__device__ __forceinline__ float part_sum2(const float *__restrict__ part, int reg, int lane)
{
    float s = part[(0 * 16 + reg) * 64 + lane];
#pragma unroll
    for (int w = 1; w < OPNET_NW; ++w) s += part[(w * 16 + reg) * 64 + lane];
    return s;
}

// grid.x = n2/2 (LSTM2 tile pairs) + n1/2 (LSTM1 tile pairs) + 2 heads ; requires H1 % 8 == 0, H2 % 8 == 0, !mlp
template <int CH>
__global__ void __launch_bounds__(OPNET_THREADS) opnet_step_wide(const StepArgs a, const int s)
{
    __shared__ __attribute__((aligned(16))) float lds[OPNET_NW * 16 * 64 + 32 * 16];
    float *part = lds;
    float *lg = lds + OPNET_NW * 16 * 64;

    const int bx = blockIdx.x;
    const int T = a.T;
    const int H1 = a.H1, H2 = a.H2;
    const int p1 = H1 >> 3, p2 = H2 >> 3;
    const int tid = threadIdx.x;
    // epilogue coordinates: threads 0..127 own tile 0, threads 128..255 tile 1
    const int el = tid & 63;
    const int half = (tid >> 6) & 1, tsel = tid >> 7;
    const int clip = half * 16 + (el & 15);
    const int quarter = el >> 4;
    const int ereg = tsel * 8 + half * 4;

    if (bx < p2) {
        // ---------------- LSTM2 (video_LSTM, learned_models.py:32,46), step t = s-2 -------------
        const int t = s - 2;
        if (t < 0 || t >= T) return;
        const int nh = H2 >> 4;
        const KSlice ks = wave_slice(nh);
        const float4 *A0 = a.w2p + (long)(2 * bx) * nh * 64, *A1 = A0 + (long)nh * 64;
        float4 a0[CH], a1[CH];
        load_a_chunk(a0, A0, ks.q0, ks.q1);
        load_a_chunk(a1, A1, ks.q0, ks.q1);
        int a_qb = ks.q0;
        const int unit = (2 * bx + tsel) * 4 + quarter;
        float4 *wl = (float4 *)lg;          // both tiles' input weights: 2 x 4 units x 8 float4
        if (tid < 64) wl[tid] = a.wih2p[(long)(2 * bx) * 32 + tid];
        for (int rb = blockIdx.y; rb < a.RB; rb += gridDim.y) {
            const float4 *hprev = a.h2buf + (slot_prev(a, t) * a.RB + rb) * (H2 * 8);
            const float4 *x2 = a.x2buf + (slot_x2(a, t) * a.RB + rb) * 64 + clip;
            const float4 xa = x2[0], xb = x2[32];
            const float c_old = a.c2[((cslot_prev(a, t) * a.RB + rb) * H2 + unit) * 32 + clip];
            gemm32_rb(a0, a1, a_qb, A0, A1, hprev, nh, hprev, ks, part, a.B - rb * 32 > 16);
            __syncthreads();
            float g[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 w0 = wl[(tsel * 4 + quarter) * 8 + 2 * r], w1 = wl[(tsel * 4 + quarter) * 8 + 2 * r + 1];
                float xs = w0.x * xa.x;
                xs = fmaf(w0.y, xa.y, xs);
                xs = fmaf(w0.z, xa.z, xs);
                xs = fmaf(w0.w, xa.w, xs);
                xs = fmaf(w1.x, xb.x, xs);
                xs = fmaf(w1.y, xb.y, xs);
                g[r] = part_sum2(part, ereg + r, el) + xs;
            }
            float c = c_old;
            float4 gs;
            const float h = lstm_cell_g(g[0], g[1], g[2], g[3], &c, &gs);
            a.c2[((cslot_out(a, t) * a.RB + rb) * H2 + unit) * 32 + clip] = c;
            if (a.train) a.g2save[(((long)t * a.RB + rb) * H2 + unit) * 32 + clip] = gs;
            float *hout = (float *)(a.h2buf + (slot_out(a, t) * a.RB + rb) * (H2 * 8));
            hout[((long)(2 * bx + tsel) * 32 + clip) * 4 + quarter] = h;
            if (rb + (int)gridDim.y < a.RB) __syncthreads();  // partials are rewritten next round
        }
    } else if (bx < p2 + p1) {
        // ---------------- LSTM1 (object_to_track_LSTM, learned_models.py:29,39), step t = s -----
        const int t = s;
        if (t >= T) return;
        const int pr = bx - p2;
        const int nhh = H1 >> 4;
        const int nhex = OPNET_KXQ / 4 + nhh;
        const KSlice ks = wave_slice(nhex);
        const float4 *A0 = a.w1p + (long)(2 * pr) * nhex * 64, *A1 = A0 + (long)nhex * 64;
        float4 a0[CH], a1[CH];
        load_a_chunk(a0, A0, ks.q0, ks.q1);
        load_a_chunk(a1, A1, ks.q0, ks.q1);
        int a_qb = ks.q0;
        const int unit = (2 * pr + tsel) * 4 + quarter;
        for (int rb = blockIdx.y; rb < a.RB; rb += gridDim.y) {
            const float4 *xsrc = a.xp + ((long)t * a.RB + rb) * (OPNET_KXQ * 32);
            const float4 *hprev = a.h1buf + (slot_prev(a, t) * a.RB + rb) * (H1 * 8);
            const float c_old = a.c1[((cslot_prev(a, t) * a.RB + rb) * H1 + unit) * 32 + clip];
            gemm32_rb(a0, a1, a_qb, A0, A1, xsrc, OPNET_KXQ / 4, hprev, ks, part, a.B - rb * 32 > 16);
            __syncthreads();
            float c = c_old;
            float4 gs;
            const float h = lstm_cell_g(part_sum2(part, ereg + 0, el), part_sum2(part, ereg + 1, el),
                                        part_sum2(part, ereg + 2, el), part_sum2(part, ereg + 3, el), &c, &gs);
            a.c1[((cslot_out(a, t) * a.RB + rb) * H1 + unit) * 32 + clip] = c;
            if (a.train) a.g1save[(((long)t * a.RB + rb) * H1 + unit) * 32 + clip] = gs;
            float *hout = (float *)(a.h1buf + (slot_out(a, t) * a.RB + rb) * (H1 * 8));
            hout[((long)(2 * pr + tsel) * 32 + clip) * 4 + quarter] = h;
            if (rb + (int)gridDim.y < a.RB) __syncthreads();
        }
    } else if (bx == p2 + p1) {
        role_selection_head<CH>(a, s, part, lg);
    } else if (bx == p2 + p1 + 1) {
        role_output_head<CH>(a, s, part);
    }   // beyond: padding workgroups
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
