// seq_xcd_bwd_kernels.hip - the reverse recurrence (BPTT) of the sibling reasoners' stacked LSTM as ONE persistent launch.
//
// What is computed: the backward of nn.LSTM(bias=False, batch_first) + Linear head of reference baselines/learned_models.py
// (BaselineLstm :99-101, NonLinearLstm :135-137, TransformerLstm :170-172) under training_main.py:183-217 - the same function as
// the launch chain stack_bwd_cell / stack_bwd_gemm (seq_kernels.hip), which stays the engine for every other shape:
//     dh_l(t) = [l == L-1 ? W_head^T dy_t : W_ih_{l+1}^T da_{l+1}(t)] + W_hh_l^T da_l(t+1)
//     (da_l(t), dc_l(t-1)) = cell backward(dh_l(t), dc_l(t), gates_l(t), c_l(t), c_l(t-1))
// da_l(t) overwrites the saved gates in place (the launch chain's layout), so the merged weight-gradient launch and the
// input-gradient GEMM run on the result unchanged.
//
// Why: the chain is two launches per reverse step - ~600 dependent launches for T = 300, 2.8 ms of a
// one-clip transformer_lstm training step with the GPU idle between them.
//
// Placement = seqx_forward's (seq_xcd_kernels.hip): groups of FOUR clips, 256 workgroups of 4 waves, XCD x = blockIdx.x & 7,
// CU c = blockIdx.x >> 3 owns hidden units 16 c .. 16 c + 15 of ITS XCD's layer; L = 1: XCD x runs groups x, x + 8, ...;
// L = 2: XCD 2 p runs layer 0 and XCD 2 p + 1 layer 1 of groups p, p + 4, ...
// The recurrent product contracts over the 4H gate columns.  As in opnet_xcd4_backward, CU c keeps the 64 gate ROWS of its own
// units - the rows of W_hh it holds in the forward, read the other way - multiplies them by ITS OWN da (LDS, no exchange on the
// way in) into partial dh rows of EVERY unit (v_mfma_f32_4x4x1: block = row quad, one (unit', gate) per instruction, B = the
// (unit', clip) float4 of da), and the exchange is a reduce-scatter: every CU stores one 256-B chunk per owner, and gathers the 32
// chunks of its own 16 units (8 KB), summed in fixed order: four by four in the registers of the gathering waves, the remaining 8
// values per lane by the cell wave.  The chunks travel through rings of 4 steps whose words hold a sentinel until published
// ("the data is the flag", re-armed two steps on: opnet_xcd4_kernels.hip has the safety argument).
// L = 2: the top layer's XCD also multiplies W_ih1^T (AccVGPR operands) by its da into partial dh rows of the LOWER layer,
// reduces them among its own CUs (waves 2, 3 gather beside waves 0, 1, one phase late; wave 1 finishes the sum a phase after
// that, off the recurrence's critical chain) and writes the 256 B per CU and step into a FULL-history buffer for the lower layer's XCD: every
// word written once per launch, so the two layers are a pipeline without flow control, as in the forward.
// Phase (group gi, n), t = T - 1 - n:
//   A. wave 0: dh = sum of the 32 chunks + upstream -> cell backward -> da -> LDS and the gate history; barrier
//   B. every wave: its 128 output rows x the CU's 64 gate rows (128 MFMAs; + 128 for W_ih1^T) -> the owners' chunks
//   C. waves 0, 1: the next phase's chunks (sentinel-polled); top of L = 2, waves 2, 3: the PREVIOUS phase's lower-layer chunks
//      (published a phase ago: with two or more groups per XCD pair nothing in a phase waits for that phase's products); barrier
// Every poll is bounded (XCD_SPIN_LIMIT): an abort raises status[0] (sticky: the optimiser's guard and the weight-gradient
// launch see it, the gradients are NaN), every poller leaves.
// (included by opnet_abi.hip after seq_xcd_kernels.hip, opnet_xcd4_kernels.hip and opnet_train_kernels.hip: it uses their helpers)
#pragma once

#define SXB_SLOTS 4

struct SeqXBPacked { size_t bh[2], bx, bo, total; };      // offsets in floats
__host__ __device__ inline SeqXBPacked seqxb_packed_layout(int L)
{
    SeqXBPacked P;
    size_t o = 0;
    for (int l = 0; l < 2; ++l) { P.bh[l] = o; if (l < L) o += (size_t)32 * 4 * 32 * 256; }   // [cu][wave][set 2 x unit' 16][lane] float4
    P.bx = o; if (L == 2) o += (size_t)32 * 4 * 32 * 256;                                       // W_ih1, the same shape
    P.bo = o; o += (size_t)32 * 256;                                                            // [cu][lane] float4: W_head of the lane's unit
    P.total = o;
    return P;
}

struct SeqXBArgs {
    int B, T, L, NGT, RB;
    const float *pk;           // seqxb_packed_layout image
    char *ws;                  // workspace base; offsets below in bytes
    unsigned g_off[2];         // per layer [T][RB][512][32] float4: gates in, da out
    unsigned c_off[2];         // per layer [T + 1][RB][512][32] float
    unsigned dy_off;           // [T][RB][32] float4
    unsigned ring_off[2];      // per layer: partial dh rows [NGT][SXB_SLOTS][32 owners][32 producers][16] float4
    unsigned dxring_off;       // L = 2: the top layer's partials of the lower layer's dh, the same shape (read on the top layer's XCD)
    unsigned dxh_off;          // L = 2: their sums, full history [NGT][T][32 CUs][64 lanes] float
    unsigned *status;          // the training forward's status words (sticky abort)
    int force_safe, debug;
};

// lane (block bb, row i) of (set, unit' m) of wave w of CU cu: output row 128 w + 64 set + 4 bb + i, k = (unit 16 cu + m, gate e)
__global__ void __launch_bounds__(256) seqxb_pack(float *__restrict__ out, const float *__restrict__ w_hh0, const float *__restrict__ w_hh1,
                                                  const float *__restrict__ w_ih1, const float *__restrict__ w_head, int L)
{
    const SeqXBPacked P = seqxb_packed_layout(L);
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < P.total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 3, lane = (idx >> 2) & 63, bb = lane >> 2, i = lane & 3;
        float v;
        if (idx >= P.bo) {
            const size_t cu = (idx - P.bo) >> 8;
            v = w_head[(size_t)e * SX_H + 16 * cu + bb];
        } else {
            const float *W = idx >= P.bx ? w_ih1 : (L == 2 && idx >= P.bh[1]) ? w_hh1 : w_hh0;
            const size_t base = idx >= P.bx ? P.bx : (L == 2 && idx >= P.bh[1]) ? P.bh[1] : P.bh[0];
            const size_t r = (idx - base) >> 8;
            const int m = r % 16, set = (r / 16) % 2, w = (r / 32) % 4, cu = r / 128;
            v = W[(size_t)(e * SX_H + 16 * cu + m) * SX_H + 128 * w + 64 * set + 4 * bb + i];
        }
        out[idx] = v;
    }
}

// status words 3..7 and the XCC sentinels (0..2 stay: the forward's abort is sticky); every exchange word "not published yet"
__global__ void __launch_bounds__(256) seqxb_init(SeqXBArgs a)
{
    const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x, n = (long)gridDim.x * blockDim.x;
    if (tid >= 3 && tid < 8) a.status[tid] = 0u;
    for (long i = tid; i < 256; i += n) a.status[8 + i] = 0xffffffffu;
    const xcd_u32x4 sent = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    const long ring4 = (long)a.NGT * SXB_SLOTS * 32 * 32 * 16;           // float4 per ring
    for (int l = 0; l < a.L; ++l) {
        xcd_u32x4 *r = (xcd_u32x4 *)(a.ws + a.ring_off[l]);
        for (long i = tid; i < ring4; i += n) r[i] = sent;
    }
    if (a.L == 2) {
        xcd_u32x4 *r = (xcd_u32x4 *)(a.ws + a.dxring_off);
        for (long i = tid; i < ring4; i += n) r[i] = sent;
        xcd_u32x4 *h = (xcd_u32x4 *)(a.ws + a.dxh_off);
        const long nh = (long)a.NGT * a.T * 32 * 16;
        for (long i = tid; i < nh; i += n) h[i] = sent;
    }
}

template <int L>
__global__ void __launch_bounds__(256) seqx_backward(const SeqXBArgs a)
{
    __shared__ __attribute__((aligned(1024))) float4 sR[2][2][64];       // by phase parity: [gathering wave][lane]: four chunks of dh summed
    __shared__ __attribute__((aligned(1024))) float4 sDX[2][2][64];      // top of L = 2: the same for the lower layer's dh
    __shared__ __attribute__((aligned(16))) float4 sDA[16][4];           // the CU's da of the phase: [unit'][clip] -> (i, f, g, o)
    __shared__ float sDC[SX_NGMAX][64];
    __shared__ float4 sPad[5120];          // 80 KB never used: more than half a CU's 160 KB of LDS keeps a second workgroup off it (seqx_forward)
    __shared__ int sAbort, sLocal;

    // the forward of this step gave up (sticky abort word): its histories are partial - leave
    if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = blockIdx.x & 7, c = blockIdx.x >> 3;
    constexpr int NPAIR = 8 / L;
    const int pr = x / L, l = __builtin_amdgcn_readfirstlane(x % L);
    const int T = a.T, RB = a.RB;
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
    for (int i = tid; i < SX_NGMAX * 64; i += 256) (&sDC[0][0])[i] = 0.f;
    if (a.debug & 0x40000000) sPad[tid * 20] = make_float4(0.f, 0.f, 0.f, 0.f);     // (keeps the padding allocated)
    const bool top = l == L - 1;
    const bool dxon = L == 2 && top;          // this XCD also produces the lower layer's dh
    const bool lower = L == 2 && !top;        // ... this one consumes it
    const unsigned g_mine = l == 0 ? a.g_off[0] : a.g_off[1];
    const unsigned c_mine = l == 0 ? a.c_off[0] : a.c_off[1];
    const unsigned ring_mine = l == 0 ? a.ring_off[0] : a.ring_off[1];

    // ---- resident weights: W_hh^T rows in VGPRs, W_ih1^T rows in AccVGPRs -----------------------------------------------------
    const SeqXBPacked P = seqxb_packed_layout(L);
    float bh[128], bx[128];
    float4 wo;
    {
        const float4 *ph = (const float4 *)(a.pk + (l == 0 ? P.bh[0] : P.bh[1])) + ((size_t)(c * 4 + w) * 32) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const float4 v = ph[q * 64];
            bh[4 * q] = v.x; bh[4 * q + 1] = v.y; bh[4 * q + 2] = v.z; bh[4 * q + 3] = v.w;
        }
        const float4 *px = (const float4 *)(a.pk + P.bx) + ((size_t)(c * 4 + w) * 32) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dxon) v = px[q * 64];
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(bx[4 * q]) : "v"(v.x));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(bx[4 * q + 1]) : "v"(v.y));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(bx[4 * q + 2]) : "v"(v.z));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(bx[4 * q + 3]) : "v"(v.w));
        }
        wo = ((const float4 *)(a.pk + P.bo))[(size_t)c * 64 + lane];
    }

    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void *)a.ws, 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16, lane4 = lane * 4;
    const xcd_u32x4 sent4 = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    const float4 sentf = x4_as_float4(sent4);
    __syncthreads();
    if (XCD_LDS_LD(sAbort)) return;
    int abort_seen = 0;
    const bool local = __builtin_amdgcn_readfirstlane(XCD_LDS_LD(sLocal)) != 0;
    const int nph = T * ng;

    // the saved activations the cell needs (gates, c_t, c_{t-1}, top layer: dy), fetched one phase ahead (HBM / Infinity Cache)
    float4 cg = make_float4(0.f, 0.f, 0.f, 0.f), cdy = cg, ng_ = cg, ndy = cg;
    float cct = 0.f, ccp = 0.f, nct = 0.f, ncp = 0.f;
    auto fetch = [&](int gi, int n, float4 &g, float4 &dy, float &ct, float &cp) {
        if (w != 0) return;
        const int G = gi * NPAIR + pr, t = T - 1 - n;
        const int rb = (4 * G) >> 5, cl = ((4 * G) & 31) + j;
        const size_t u = 16 * c + b;
        g = ((const float4 *)(a.ws + g_mine))[(((size_t)t * RB + rb) * SX_H + u) * 32 + cl];
        ct = ((const float *)(a.ws + c_mine))[(((size_t)(t + 1) * RB + rb) * SX_H + u) * 32 + cl];
        cp = ((const float *)(a.ws + c_mine))[(((size_t)t * RB + rb) * SX_H + u) * 32 + cl];
        if (top) dy = ((const float4 *)(a.ws + a.dy_off))[((size_t)t * RB + rb) * 32 + cl];
    };
    // lower layer of L = 2: the upstream dh of (group, step) from the other XCD: asked for a phase early, polled at its use
    unsigned dxr = 0xffffffffu, dxsrc = 0;
    auto ask_dx = [&](int gi, int n) {
        if (!lower || w != 0) return;
        const int G = gi * NPAIR + pr, t = T - 1 - n;
        dxsrc = a.dxh_off + ((unsigned)(G * T + t) * 32 + c) * 256;
        dxr = __builtin_amdgcn_raw_buffer_load_b32(rws, lane4, dxsrc, 16);      // sc1
    };
    fetch(0, 0, cg, cdy, cct, ccp);
    ask_dx(0, 0);

    // (top of L = 2) the lower layer's dh lags the recurrence: the chunks of phase p are gathered in phase p + 1 - they were published a
    // phase ago, so with two or more groups per XCD pair no gather of a phase waits for this phase's products - and summed and
    // handed over in phase p + 2
    int gi = 0, n = 0, g1 = 0, n1 = -1, g2 = 0, n2 = -1;       // (g1, n1): the previous phase, (g2, n2): the one before
    for (int p = 0; p <= nph + 1; ++p) {
        const bool work = p < nph;
        const int buf = p & 1;
        const int G = gi * NPAIR + pr, t = T - 1 - n;
        int gn = gi + 1, nn = n;
        if (gn == ng) { gn = 0; ++nn; }
        const bool more = p + 1 < nph;
        bool ok = true;
        if (work && more) fetch(gn, nn, ng_, ndy, nct, ncp);
        // ================================ A. the cell of this phase (wave 0); top of L = 2, wave 1: the lower layer's dh of the phase
        //                                     before the previous one (its chunks were summed four by four at the end of the last phase)
        if (w == 0 && work) {
            float rec = 0.f;
            if (n > 0) {
                const float *prp = (const float *)&sR[buf][0][0] + ((b >> 2) * 4 + j) * 4 + (b & 3);
                float r0 = 0.f, r1 = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) { r0 += prp[q * 64]; r1 += prp[256 + q * 64]; }
                rec = r0 + r1;
            }
            float dh;
            if (top) {
                // upstream: predictions_layer (learned_models.py:113 / 148 / 195): dh = W_head^T dy_t
                dh = wo.x * cdy.x;
                dh = fmaf(wo.y, cdy.y, dh);
                dh = fmaf(wo.z, cdy.z, dh);
                dh = fmaf(wo.w, cdy.w, dh);
            } else {
                long long t0 = 0;
                for (unsigned spins = 1;; ++spins) {
                    if (!__any(dxr == 0xffffffffu)) break;
                    if (!x4_keep_polling(spins, t0, a.status, p)) { ok = false; break; }
                    dxr = __builtin_amdgcn_raw_buffer_load_b32(rws, lane4, dxsrc, 16);
                }
                dh = __uint_as_float(dxr);
            }
            dh += rec;
            float dco;
            const float4 da = cell_backward(dh, sDC[gi][lane], cg, cct, ccp, &dco);
            sDC[gi][lane] = dco;
            sDA[b][j] = da;
            // da replaces the saved gates (the weight-gradient GEMMs and the input-gradient GEMM read it there)
            const int rb = (4 * G) >> 5, cl = ((4 * G) & 31) + j;
            ((float4 *)(a.ws + g_mine))[(((size_t)t * RB + rb) * SX_H + 16 * c + b) * 32 + cl] = da;
        }
        if (w == 1 && dxon && n2 >= 0) {
            const float *prp = (const float *)&sDX[buf ^ 1][0][0] + ((b >> 2) * 4 + j) * 4 + (b & 3);
            float r0 = 0.f, r1 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { r0 += prp[q * 64]; r1 += prp[256 + q * 64]; }
            const int Gp = g2 * NPAIR + pr, tp = T - 1 - n2;
            xcd_store4(rws, lane4, a.dxh_off + ((unsigned)(Gp * T + tp) * 32 + c) * 256, r0 + r1, false);   // write-through: read on another XCD
        }
        if (p > nph) break;
        if (!ok) XCD_LDS_ST(sAbort, 1);
        if (work) {
        __syncthreads();                        // barrier 1: the CU's da of the phase is in LDS
        // ================================ B. products: the CU's 64 gate rows x its da -> partial dh rows of every unit =========
        {
            const float4 *F = &sDA[0][0] + j;
            sx_f32x4 e2a[4], e2b[4], f2a[4], f2b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) e2a[q] = e2b[q] = f2a[q] = f2b[q] = (sx_f32x4){0.f, 0.f, 0.f, 0.f};
            float4 bf[SX_RING];
#pragma unroll
            for (int i = 0; i < SX_AHEAD; ++i) bf[i] = F[i * 4];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (m + SX_AHEAD < 16) bf[(m + SX_AHEAD) % SX_RING] = F[(m + SX_AHEAD) * 4];
                __builtin_amdgcn_sched_barrier(0);
                const float4 bq = bf[m % SX_RING];
                SX_MFMA_V(e2a[0], bh[4 * m + 0], bq.x); SX_MFMA_V(e2b[0], bh[64 + 4 * m + 0], bq.x);
                SX_MFMA_V(e2a[1], bh[4 * m + 1], bq.y); SX_MFMA_V(e2b[1], bh[64 + 4 * m + 1], bq.y);
                SX_MFMA_V(e2a[2], bh[4 * m + 2], bq.z); SX_MFMA_V(e2b[2], bh[64 + 4 * m + 2], bq.z);
                SX_MFMA_V(e2a[3], bh[4 * m + 3], bq.w); SX_MFMA_V(e2b[3], bh[64 + 4 * m + 3], bq.w);
                if (dxon) {
                    SX_MFMA_A(f2a[0], bx[4 * m + 0], bq.x); SX_MFMA_A(f2b[0], bx[64 + 4 * m + 0], bq.x);
                    SX_MFMA_A(f2a[1], bx[4 * m + 1], bq.y); SX_MFMA_A(f2b[1], bx[64 + 4 * m + 1], bq.y);
                    SX_MFMA_A(f2a[2], bx[4 * m + 2], bq.z); SX_MFMA_A(f2b[2], bx[64 + 4 * m + 2], bq.z);
                    SX_MFMA_A(f2a[3], bx[4 * m + 3], bq.w); SX_MFMA_A(f2b[3], bx[64 + 4 * m + 3], bq.w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // lane (block bb = b, clip j) holds rows 4 b .. 4 b + 3 of its row set = one float4 of the owner's chunk:
            //   row 128 w + 64 set + 4 b + i -> owner 8 w + 4 set + (b >> 2), unit quad b & 3
            const unsigned vo = (((b >> 2) * 32) * 16 + (b & 3) * 4 + j) * 16;           // owner stride: 32 producers x 16 float4
            const unsigned slot = (unsigned)(G * SXB_SLOTS + (t & 3));
            if (t > 0) {                        // (the products of da_0 would feed dh_{-1}); re-armed two steps on
                const sx_f32x4 d2a = (e2a[0] + e2a[1]) + (e2a[2] + e2a[3]), d2b = (e2b[0] + e2b[1]) + (e2b[2] + e2b[3]);
                const unsigned so = ring_mine + ((slot * 32 + 8 * w) * 32 + c) * 256;
                const unsigned sr = ring_mine + (((unsigned)(G * SXB_SLOTS + ((t + 2) & 3)) * 32 + 8 * w) * 32 + c) * 256;
                xcd_store16(rws, vo, so, make_float4(d2a[0], d2a[1], d2a[2], d2a[3]), local);
                xcd_store16(rws, vo, so + 4 * 32 * 256, make_float4(d2b[0], d2b[1], d2b[2], d2b[3]), local);
                xcd_store16(rws, vo, sr, sentf, local);
                xcd_store16(rws, vo, sr + 4 * 32 * 256, sentf, local);
            }
            if (dxon) {                         // re-armed THREE steps on: its readers gather a phase late
                const sx_f32x4 d2a = (f2a[0] + f2a[1]) + (f2a[2] + f2a[3]), d2b = (f2b[0] + f2b[1]) + (f2b[2] + f2b[3]);
                const unsigned so = a.dxring_off + ((slot * 32 + 8 * w) * 32 + c) * 256;
                const unsigned sr = a.dxring_off + (((unsigned)(G * SXB_SLOTS + ((t + 3) & 3)) * 32 + 8 * w) * 32 + c) * 256;
                xcd_store16(rws, vo, so, make_float4(d2a[0], d2a[1], d2a[2], d2a[3]), local);
                xcd_store16(rws, vo, so + 4 * 32 * 256, make_float4(d2b[0], d2b[1], d2b[2], d2b[3]), local);
                xcd_store16(rws, vo, sr, sentf, local);
                xcd_store16(rws, vo, sr + 4 * 32 * 256, sentf, local);
            }
        }
        }
        // ================================ C. the next phase's inputs =========================================================
        if (w < 2) {
            if (more && nn > 0) {
                const unsigned Gn = (unsigned)(gn * NPAIR + pr);
                const unsigned src = ring_mine + ((Gn * SXB_SLOTS + ((T - nn) & 3)) * 32 + c) * 8192 + w * 4096;
                if (!x4_gather_sum4(rws, lane16, src, true, &sR[buf ^ 1][w][lane], a.status, p)) ok = false;
            }
        } else if (dxon && n1 >= 0) {
            const unsigned G1 = (unsigned)(g1 * NPAIR + pr);
            const unsigned src = a.dxring_off + ((G1 * SXB_SLOTS + ((T - 1 - n1) & 3)) * 32 + c) * 8192 + (w - 2) * 4096;
            if (!x4_gather_sum4(rws, lane16, src, true, &sDX[buf][w - 2][lane], a.status, p)) ok = false;
        }
        if (more) ask_dx(gn, nn);
        if (!ok) XCD_LDS_ST(sAbort, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // orders this phase's re-arm stores before the next publish
        __syncthreads();                        // barrier 2: the next phase's chunks have landed
        if (abort_seen) return;
        abort_seen = XCD_LDS_LD(sAbort);
        g2 = g1; n2 = n1;
        g1 = gi; n1 = work ? n : -1;
        gi = gn; n = nn;
        cg = ng_; cdy = ndy; cct = nct; ccp = ncp;
    }
}
