// opnet_xcd_kernels.hip - OPNet forward as ONE persistent launch: eight independent forwards, one per XCD,
// every weight resident in registers for all T steps.
//
// What is computed: reference baselines/learned_models.py:35-52 (the same function as opnet_kernels.hip).
//
// Why (DESIGN.md sections 3, 3a, 7): the launch-per-step chain re-fetches the 5.68 MB weight set 303 times per forward and
// pays a dependent kernel boundary per step; a chip-wide persistent kernel pays a chip-wide exchange per step instead and
// lost.  Here the chip is cut along its XCDs.  One XCD = 32 CUs = 128 SIMDs, and 128 register files hold the whole weight
// set.  A workgroup = 8 waves on one CU; SIMD w carries a PRODUCT wave and a FINISH wave:
//     product wave (CU c, SIMD w)  LSTM2 tile 4c+w   : 16 gate rows (4 units x i,f,g,o) x K = 512        128 VGPRs
//                                  LSTM1 tile 2c+w/2 : 16 gate rows x half of K = 96 + 256 (w&1 picks)     44 VGPRs
//                                  selection head    : 16 (15) rows x K quarter w of 256                   16 VGPRs
//     finish wave                  W_ih2 rows of the lane's own unit (6 -> 8 wide)                         32 VGPRs
// Each XCD runs ITS OWN clips (groups of 16 = one MFMA column block, 1..8 groups per XCD), so nothing crosses an XCD
// boundary on the data path and the only exchange is h1 / h2 between the 32 CUs of one XCD, once per (group, step):
//     phase (group g, step s):  LSTM1 step s | selection head + einsum + LSTM2 step s-1        (T+1 steps per forward)
//   * the phase's activations x[s], h1[s-1], h2[s-2] (54 KB, "kq-major" [k/4][16 clips][4] so that 1 KB = one MFMA B
//     fragment set of a 16-k step) sit in one of two LDS buffers, gathered by LDS-DMA while the previous phase computes;
//   * the product wave runs the phase's 188 MFMAs (v_mfma_f32_16x16x4_f32, exact fp32) on 4 independent accumulator
//     chains, B operands by ds_read_b128 one step ahead, A operands = the resident registers, hands its accumulators to
//     LDS and meets the one barrier of the phase; it issues some of its MFMAs with an idle gap, because fp32 MFMA runs on
//     the SIMD's fp32 lanes and the finish wave beside it otherwise gets no VALU cycle at all (see "Yielding" below);
//   * the finish wave, in the window of the NEXT phase's products: sums the K-split partials (head 4 waves, LSTM1 2),
//     softmax, einsum (every CU redundantly - every LSTM2 tile needs frames_boxes), its LSTM2 cell and (odd waves) its LSTM1
//     cell, publishes h, then polls the flags of the phase after next and issues its quarter of that gather;
//   * h is published with 16-byte stores into FULL-HISTORY buffers (slot t+1 = step t; every word is written once per
//     launch and only read after its flag), every storing wave drains vmcnt(0), the last of the CU's four finish waves
//     (LDS arrival counter) stores ONE flag per CU (cdna_hip_programming.md Guideline 16 recipe R1; consumers read with
//     sc1 loads, so no acquire fence).  Stores are plain - the line stays in this XCD's L2 - when the kernel has verified at
//     start that the group's 32 workgroups share an XCD (block b runs on XCD b % 8: observed, not promised), write-through
//     (sc1) otherwise: placement is for speed only.
// y = W_out h2 is not on the recurrence: it is computed from the h2 history by opnet_xcd_out_head afterwards.
//
// Every spin is bounded (XCD_SPIN_LIMIT = 1.5 s): a workgroup that cannot see its producers raises the abort word, every
// other poller sees it and leaves, and opnet_xcd_out_head poisons y with NaN - nothing can hang the device.
//
// Summation order (differs from the launch chain's K-split, agrees to rounding; tests hold both to the oracle):
//   LSTM2 gate = (sum over even 16-k steps) + (sum over odd 16-k steps) + x part; LSTM1 gate = lower K half + upper K
//   half; logits = ((w0 + w1) + w2) + w3 over K quarters; each partial an ascending-k fmaf chain per MFMA lane group;
//   frames_boxes = per lane group an ascending-slot fmaf chain over its 4 slots, then (g0 + g1) + (g2 + g3).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "opnet_ctx.h"

#define XCD_COUNT 8
#define XCD_CUS 32
#define XCD_NGMAX 8            // 16-clip groups one XCD carries in one launch (8 x 16 x 8 XCDs = 1024 clips)
#define XCD_H1 256
#define XCD_H2 512
#define XCD_RING 4             // slots of the h1 / h2 / frames_boxes rings (a.ring): a CU publishes step t only after it has gathered every
                               // CU's step t - 1, i.e. after every CU has finished reading step t - 2 and older: 3 slots are enough
#define XCD_SPIN_LIMIT 150000000ll    // wall_clock64 ticks (100 MHz: 1.5 s) before a poller gives up

// LDS gather buffer of one phase, in float4 units: X0 = x[s] | H1 = h1[s-1] | H2 = h2[s-3] | FB = frames_boxes[s-2]
#define XB_X0 0
#define XB_H1 (6 * 64)
#define XB_H2 (22 * 64)
#define XB_FB (54 * 64)        // [2 k-quads][4][16 clips] floats = 512 B of the 1-KB piece
#define XB_F4 (55 * 64)        // 55 KB
#define XB_CHUNKS 55

struct XcdArgs {
    int B, T, NGT;             // clips, frames, 16-clip groups = ceil(B / 16)
    const float *packed;       // opnet_pack_weights_f32 image for H1 = 256, H2 = 512
    const float4 *xp;          // [NGT][T+2][24][16]   slot t+1 = x[t]; slots 0 and T+1 zero
    float4 *h1h;               // [NGT][T+1][64][16]   slot t+1 = h1[t]; slot 0 zero
    float4 *h2h;               // [NGT][T+1][128][16]  slot t+1 = h2[t]; slot 0 zero
    float4 *fbh;               // [NGT][T+1][64]       slot t+1 = frames_boxes[t] as [8 features][16 clips] floats in the first 512 B; slot 0 zero
    unsigned *flags;           // [NGT][32]            steps published by CU c of the group's XCD
    unsigned *status;          // [0] abort code (0 = ok), [1] first failing block, [2] phase, [3] groups not XCD-local,
                               // [8 + b] XCC_ID of block b
    float *logits;             // caller's [B][15][T]
    char *ws;                  // workspace base and the byte offsets of xp / h1h / h2h / flags in it (one buffer descriptor)
    unsigned xp_off, h1_off, h2_off, fb_off, flags_off, status_off;
    int ring;                  // 0: h1h / h2h / fbh hold the full history ([T+1] slots per group) and opnet_xcd_out_head reads h2 back;
                               // 1: they are rings of XCD_RING slots (slot = step & 3) and the output head is computed in the launch:
                               //    head-once form: y[s-3] = W_out h2[s-3] as 32 more MFMAs per product wave of ONE CU of the XCD per
                               //    phase, on the B fragments its LSTM2 products stream anyway, stored to y by that CU (one more phase
                               //    per group: s = T+2);  otherwise per-CU partials ypart [NGT][T][32 CUs][16 clips] float4, summed by
                               //    opnet_xcd_y_reduce
    unsigned yp_off;
    float *y;                  // caller's [B][T][4] (ring mode, head-once form)
    int force_safe;            // 1: always use the placement-independent write-through protocol (tests)
    int debug;                 // tools only (wrong results): bit 0 no poll/gather, 1 no head, 2 no cells, 3 no publish, 4 no x fetch
    unsigned long long *trace; // optional [phases][8] s_memtime stamps of block 0 (product wave 0: 0-1, finish wave 4: 2-7), or null
    // TRAIN instantiation (the training forward of 97+ clips, opnet_train_forward_f32): byte offsets in ws of the launch chain's
    // history buffers (train_workspace_layout: slot t + 1 of h / c = step t; row blocks of 32 clips = two 16-clip groups), which the
    // finish waves fill beside the exchange so that the chain's reverse recurrence and the weight-gradient launch run on them
    unsigned tr_h1, tr_c1, tr_h2, tr_c2, tr_g1, tr_g2, tr_ps, tr_x2;
    int RB;
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
__device__ __forceinline__ void xcd_glds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}

// 16-byte store to (wave-uniform descriptor, wave-uniform byte offset soff) + lane offset voff, counted by the compiler's
// vmcnt bookkeeping.  local = false: write-through (aux 16 = sc1; the line leaves the L2, every reader - any XCD - then
// fetches it across the fabric: placement-independent, Guideline 16 R1).  local = true: plain store, the line stays in
// THIS XCD's L2, where the group's other CUs read it with sc1 (L1-bypassing) loads - valid only when every workgroup of
// the group sits on the same XCD, which the kernel verifies at start (xcd_group_is_local).
__device__ __forceinline__ void xcd_store16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float4 v, bool local)
{
    xcd_u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    if (local) __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 16);
}
// plain history stores (nobody reads them inside the launch): lane offset + wave-uniform scalar offset through the descriptor
__device__ __forceinline__ void xcd_st4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float4 v)
{
    xcd_u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 0);
}
__device__ __forceinline__ void xcd_st1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
__device__ __forceinline__ void xcd_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v, bool local)
{
    if (local) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 16);
}
__device__ __forceinline__ void xcd_store_flag(unsigned *p, unsigned v, bool local)
{
    if (local) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // plain store: stays in the L2
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Where the clips of a launch come from: up to XCD_MAX_SOURCES request tensors [n_r][T][90], clip b of the launch = clip
// b - start[r] of request r (start[r] <= b < start[r + 1]); by value in the kernarg segment, so that a server can run its
// pending requests as one launch without concatenating them first.
#define XCD_MAX_SOURCES 64
struct XcdSources {
    const float *p[XCD_MAX_SOURCES];
    int start[XCD_MAX_SOURCES + 1];
    int n;
};

// boxes -> xp; zero slot 0 / T+1 of xp, slot 0 of the histories, the flags and the status words.
// grid (T + 2, NGT), 384 threads (24 k-quads x 16 clips)
__global__ void __launch_bounds__(384) opnet_xcd_pack_input(const XcdSources src, XcdArgs a)
{
    const int slot = blockIdx.x, gg = blockIdx.y;
    const int T = a.T, tid = threadIdx.x;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int kq = tid % OPNET_KXQ, clip = tid / OPNET_KXQ;   // consecutive threads walk k within a clip row
        const int b = gg * 16 + clip, t = slot - 1;
        float4 v = z;
        if (b < a.B && t >= 0 && t < T) {
            int r = 0;
            while (r + 1 < src.n && b >= src.start[r + 1]) ++r;
            const float2 *s2 = (const float2 *)(src.p[r] + ((long)(b - src.start[r]) * T + t) * OPNET_KX + kq * 4);   // rows are 360 B apart
            const int k = kq * 4;
            if (k + 1 < OPNET_KX) { float2 p = s2[0]; v.x = p.x; v.y = p.y; }
            if (k + 3 < OPNET_KX) { float2 q = s2[1]; v.z = q.x; v.w = q.y; }
        }
        ((float4 *)a.xp)[(((long)gg * (T + 2) + slot) * OPNET_KXQ + kq) * 16 + clip] = v;
    }
    if (slot == 0) {
        const long NS = a.ring ? XCD_RING : T + 1;             // slots per group
        float4 *h1 = a.h1h + (long)gg * NS * (XCD_H1 * 4);
        float4 *h2 = a.h2h + (long)gg * NS * (XCD_H2 * 4);
        for (int i = tid; i < XCD_H1 * 4; i += 384) h1[i] = z;
        for (int i = tid; i < XCD_H2 * 4; i += 384) h2[i] = z;
        if (tid < 64) a.fbh[(long)gg * NS * 64 + tid] = z;
        if (tid < XCD_CUS) a.flags[gg * XCD_CUS + tid] = 0u;
        if (gg == 0 && tid < 8) a.status[tid] = 0u;
        if (gg == 0 && tid < XCD_COUNT * XCD_CUS) a.status[8 + tid] = 0xffffffffu;
    }
}

// have all 32 CUs of the group's XCD published `need` steps?  (one sc1 load per lane: L2-served, never this CU's L1)
__device__ __forceinline__ bool xcd_flags_ready(const unsigned *flags, unsigned need)
{
    const unsigned v = __hip_atomic_load(flags + (threadIdx.x & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __all(v >= need);
}

// bounded wait for the same; false = abort (wave-uniform).  The wall clock (s_memrealtime: ~600 cycles) is first read at round 64:
// a wait that ends within a few polls - the common case - never pays for it.
__device__ __forceinline__ bool xcd_wait_flags(const unsigned *flags, unsigned need, unsigned *status, int phase)
{
    if (xcd_flags_ready(flags, need)) return true;
    const int lane = threadIdx.x & 63;
    long long t0 = 0;
    for (unsigned spins = 1;; ++spins) {
        __builtin_amdgcn_s_sleep(4);
        if (xcd_flags_ready(flags, need)) return true;
        if ((spins & 63u) == 0) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            const long long now = (long long)wall_clock64();
            if (spins == 64u) t0 = now;
            if (now - t0 > XCD_SPIN_LIMIT) {
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

// The same through the workspace's buffer descriptor (wave-uniform byte offsets of the group's flags and of the status words): the
// finish waves' loop keeps no 64-bit pointer alive for it.  Every SGPR the loop does not hold is one the compiler does not spill into
// VGPR lanes - and a v_readlane in that loop is a VALU instruction the SIMD's MFMA stream pays for (section 3a of DESIGN.md).
__device__ __forceinline__ bool xcd_wait_flags_ws(__amdgpu_buffer_rsrc_t rws, unsigned flags_soff, unsigned need, unsigned status_soff, int phase)
{
    const unsigned lane = threadIdx.x & 63, fv = (lane & 31) * 4;
    if (__all(__builtin_amdgcn_raw_buffer_load_b32(rws, fv, flags_soff, 16) >= need)) return true;
    long long t0 = 0;
    for (unsigned spins = 1;; ++spins) {
        __builtin_amdgcn_s_sleep(4);
        if (__all(__builtin_amdgcn_raw_buffer_load_b32(rws, fv, flags_soff, 16) >= need)) return true;
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

#define XCD_LDS_LD(x) __hip_atomic_load(&(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define XCD_LDS_ST(x, v) __hip_atomic_store(&(x), (int)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

// One LDS add by lane 0 alone (exec = 1 for the one instruction), the old value to the whole wave.  What the compiler makes of
// `if (lane == 0) atomic_add(...)` is a dozen VALU instructions (mbcnt, compares, moves) - in the finish waves' loop every one of
// them is taken from the MFMA stream.
__device__ __forceinline__ unsigned xcd_lds_add_lane0(unsigned lds_addr, unsigned val)
{
    unsigned old;
    unsigned long long keep;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_mov_b64 exec, %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(old), "=&s"(keep) : "v"(lds_addr), "v"(val) : "memory");
    return __builtin_amdgcn_readfirstlane(old);
}

// A kernel argument fetched where it is used (one s_load from the kernarg segment; the struct is the kernel's only argument, at
// offset 0) instead of living in SGPRs across the finish waves' loop: for pointers only one wave in 128 needs per phase.
template <unsigned OFF>
__device__ __forceinline__ unsigned long long xcd_karg_u64()
{
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(OFF) : "memory");
    return v;
}
#define XCD_KARG(type, field) ((type)xcd_karg_u64<(unsigned)__builtin_offsetof(XcdArgs, field)>())

// Do the 32 workgroups of group x (blocks x, x + 8, ...) sit on one XCD?  Every block publishes its XCC_ID (write-through)
// and reads the 32 ids of its group (sc1 loads), so all of them reach the same verdict.  -1 = gave up (abort raised).
__device__ __forceinline__ int xcd_group_is_local(unsigned *status, int x)
{
    const int lane = threadIdx.x & 63;
    const unsigned mine = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf;   // HW_REG_XCC_ID[3:0]
    if (lane == 0) __hip_atomic_store(status + 8 + blockIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned *p = status + 8 + x + XCD_COUNT * (lane & 31);
    const long long t0 = wall_clock64();
    for (unsigned spins = 1;; ++spins) {
        const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v != 0xffffffffu)) return __all(v == mine) ? 1 : 0;
        __builtin_amdgcn_s_sleep(4);
        if ((spins & 63u) == 0) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return -1;
            if ((long long)wall_clock64() - t0 > XCD_SPIN_LIMIT) {
                if (lane == 0) {
                    __hip_atomic_store(status + 1, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(status + 2, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return -1;
            }
        }
    }
}

// wave-private LDS scratch written by some lanes and read by others of the SAME wave: the LDS queue is in order per
// wave, so only the compiler has to be kept from moving the read above the write
#define XCD_WAVE_LDS_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
// XCD_GAP: idle issue states the product wave spends after every MFMA.  A wave whose next instruction is an MFMA that the
// (busy) matrix pipe cannot accept yet holds its SIMD's VALU issue port, and with four independent accumulator chains the
// product wave ALWAYS has one pending: the finish wave on the same SIMD then does not get a single VALU instruction
// issued until the product phase is over (measured: frozen for exactly the 6 100 cycles of the products, whatever the
// s_setprio / wave age).  An s_nop shorter than the MFMA's 32-cycle occupancy costs the MFMA stream nothing and leaves
// the port to the other wave in the meantime.
#ifndef XCD_GAP
#define XCD_GAP 8
#endif
// The MFMA and its idle states are ONE asm statement (volatile asm statements keep their order; a builtin MFMA next to an
// asm s_nop gets re-paired by the scheduler).  What hipcc then no longer does for these MFMAs (cdna_hip_programming.md
// section 5.7): the first MFMA of a chain takes the constant 0 as C (no VALU-written accumulator is read), chains
// accumulate in place (D = C: no wait states needed), the B operands come from counted ds_reads, the A operands were
// written long ago, and XCD_MFMA_DRAIN pads the last MFMAs' results before the compiler's code reads them.
#define XCD_STR2(x) #x
#define XCD_STR(x) XCD_STR2(x)
#define XCD_MFMA0_G(acc, av, bv, g) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0\n\ts_nop " XCD_STR(g) : "=&v"(acc) : "v"(av), "v"(bv))
#define XCD_MFMA_G(acc, av, bv, g) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop " XCD_STR(g) : "+v"(acc) : "v"(av), "v"(bv))
// inside the product loop: rows below YM (a compile-time constant there) issue with the yielding gap, the rest dense
#define XCD_MFMA0(acc, av, bv) do { if (ROW < YM) XCD_MFMA0_G(acc, av, bv, XCD_GAP); else XCD_MFMA0_G(acc, av, bv, 0); } while (0)
#define XCD_MFMA(acc, av, bv) do { if (ROW < YM) XCD_MFMA_G(acc, av, bv, XCD_GAP); else XCD_MFMA_G(acc, av, bv, 0); } while (0)
// the first MFMA of a row: in the dense tail (rows >= YM) it still carries the gap when XCD_TAIL is set, so that the few VALU
// instructions the finish wave has left (flag compare of the poll, address moves of the gather) are not frozen until the
// product phase ends
#ifndef XCD_TAIL
#define XCD_TAIL 2             // every-CU-head form, round 3 (lighter finish loop): every 2nd row 2.24 ms at 256 clips / 2.16 at 192 against 2.31 / 2.22 with
                               // every row; with XCD_NY2 = 8 on top 2.55 / 2.50
#endif
#ifndef XCD_TAIL_HO
#define XCD_TAIL_HO 2          // head-once form, every 2nd row: 131 / 133.5 / 134.8 / 136.6 k clips/s at 384 / 512 / 640 / 1024 clips (every row: 128 / 131 /
                               // 131 / 132, every 3rd: 130 / 133.4 / 135.9 / -, every 4th: 128.7 at 640).  Round 2 measured the opposite (every row 124 / 126 / 129,
                               // every 2nd 117 / 126 / 129): the finish waves then executed ~25 more VALU instructions per phase (SGPR spill reloads,
                               // the compiler's lane-0 atomic) and needed every gap to get through them
#endif
#define XCD_TAILV (HO ? XCD_TAIL_HO : XCD_TAIL)
#define XCD_MFMA_LEAD(acc, av, bv) do { if (ROW < YM || (XCD_TAILV > 0 && YM >= 0 && (ROW % (XCD_TAILV > 0 ? XCD_TAILV : 1)) == 0)) XCD_MFMA_G(acc, av, bv, XCD_GAP); else XCD_MFMA_G(acc, av, bv, 0); } while (0)
// Yielding.  fp32 MFMA executes on the SIMD's fp32 lanes - it runs at exactly the VALU rate - so while the product wave
// keeps the matrix queue full the finish wave on the same SIMD gets no VALU cycle at all (measured: frozen for the whole
// product phase, whatever s_setprio or the wave age).  The product wave therefore issues the MFMAs of
// the first YM rows of a phase (of 64) with an idle gap LONGER than the MFMA's own 32 cycles (XCD_GAP = 8: 36 cycles of
// s_nop, i.e. a bubble of a few cycles after every MFMA in which the finish wave's next VALU instruction issues), and the
// rest dense.  The finish wave needs its VALU cycles at the START of its window (head, cells; afterwards it only drains,
// polls and issues DMA) and with a bubble after every MFMA it runs at close to its stand-alone latency; its latencies
// (LDS, shuffles, memory, the hand-off) overlap the MFMA stream instead of following it.  YM per groups per XCD:
// -1 (no gap anywhere) for one group: the finish runs while this wave waits for the exchange anyway.
// Tried before: s_sleep 1 after a row idles the VALU ~32 of ~64 cycles - coarser, 8 300-9 200 cycles
// per phase; a gap <= 28 cycles gives the finish wave nothing; gaps of 40-52 cycles on every row 9 500-12 300 cycles.
// Measured (clips/s at 256 / 384 / 512 / 1024 clips, tail gap on): YM 16: 106-107 k at 256; YM 8: 113.8 k at 384; YM 0:
// 115.8 k / 118.5 k at 512 / 1024 (YM 8 there: 115.4 / 116.8, YM 16: 114.1 / 114.3).
#ifndef XCD_NY2
#define XCD_NY2 16             // two groups per XCD: the exchange is on the critical path, the finish must be quick
#endif
#ifndef XCD_NY3
#define XCD_NY3 8              // three groups: the finish has some slack
#endif
#ifndef XCD_NY4
#define XCD_NY4 0              // four or more: a window of slack - the tail gap alone is enough
#endif
// an 8-pass MFMA's result may be read by other instructions 11 wait states after it issued (16-pass: 19)
#ifndef XCD_DRAIN_NOP
#define XCD_DRAIN_NOP "s_nop 13"      // 14 wait states (11 required; the 20 of round 2 cost ~25 cycles per phase)
#endif
#define XCD_MFMA_DRAIN(a0, a1, a2, a3) asm volatile(XCD_DRAIN_NOP : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3))

// ------------------------------------------------------------------------------------------------------------------
// the persistent kernel: 8 waves per CU, two roles
//   waves 0..3 ("product" waves, one per SIMD): hold the weights, run the 188 MFMAs of a phase out of the LDS gather
//                buffer, hand their accumulators to LDS (sHAND), barrier, next phase - nothing else;
//   waves 4..7 ("finish" waves, the second wave of each SIMD): in the window of the NEXT phase's products they finish the
//                phase - head sum, softmax, einsum, the LSTM cells, write-through stores of h, drain, one flag per wave -
//                and poll + issue the LDS-DMA gather of the phase after next.  Their LDS / transcendental / memory
//                latencies run under the other wave's MFMA stream instead of in series with it.
// One s_barrier per phase (every wave): it says "accumulators of phase p are in sHAND, the gather of phase p+1 has
// landed".  Timeline of group A with three groups per XCD:  products A(s) | finish A(s) + publish | gather A(s+1) |
// products A(s+1): the product waves never wait.  With two groups the gather of A(s+1) can only be issued once every CU
// has finished A(s), ~0.15 phase too late (measured stall); with one group the exchange is fully exposed (second
// barrier per phase).
// ------------------------------------------------------------------------------------------------------------------
// first finish wave: 0 = the finish waves are the OLDER wave of each SIMD (waves 0..3), 4 = the younger one
#ifndef XCD_FW0
#define XCD_FW0 0
#endif
// which CU of the XCD computes the selection head of (step s's phase, group gi): rotates so that the extra work (16 MFMAs a
// wave + softmax / einsum in one finish wave) lands on every CU once in 32 phases
__host__ __device__ inline int xcd_head_cu(int s, int gi) { return (s + 11 * gi) & (XCD_CUS - 1); }
// ... and which one the output head y[s-3] (ring mode): half an XCD away from the selection head's
__host__ __device__ inline int xcd_y_cu(int s, int gi) { return (s + 11 * gi + 16) & (XCD_CUS - 1); }

// wave-uniform condition bits of the finish waves' loop (see XCD_CF there)
#define XCD_CF_NG4 1u
#define XCD_CF_NG2 2u
#define XCD_CF_NG1 4u
#define XCD_CF_DBG1 8u
#define XCD_CF_DBG2 16u
#define XCD_CF_DBG4 32u
#define XCD_CF_DBG8 64u
#define XCD_CF_RING 128u
#define XCD_CF_LOCAL 256u
#define XCD_CF_TRACE 512u

#define XH_F4 (3 * 64)         // sHAND per product wave: LSTM2 gates | LSTM1 partial | head partial

template <bool HO, bool TRAIN = false>
__global__ void __launch_bounds__(512, 2) opnet_xcd_forward(const XcdArgs a)
{
    __shared__ __attribute__((aligned(1024))) float4 sbuf[2][XB_F4];
    __shared__ __attribute__((aligned(16))) float4 sHAND[2][4][XH_F4];
    __shared__ __attribute__((aligned(16))) float sTR[4][2][64];     // (clip, unit) -> float4-per-clip transposes
    __shared__ float sC2[XCD_NGMAX][4][64];
    __shared__ float sC1[XCD_NGMAX][2][64];
    __shared__ int sAbort, sLocal, sH1done;      // read / written through XCD_LDS_LD / XCD_LDS_ST (a `volatile` LDS word is accessed with
                                                 // FLAT instructions and a vmcnt wait; these are plain ds_read / ds_write)
    __shared__ unsigned sArrive[2];
    __shared__ __attribute__((aligned(16))) float4 sY[2][4][64];     // ring mode: W_out[0..3][unit] * h per (wave, lane = clip + 16 unit), by phase parity

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv & 3;
    const int x = blockIdx.x & (XCD_COUNT - 1), c = blockIdx.x >> 3;
    const int T = a.T;
    int g0, ng;
    xcd_groups(a.NGT, x, &g0, &ng);
    if (ng > XCD_NGMAX) ng = XCD_NGMAX;   // the host never asks for more

    const int n = lane & 15, u = lane >> 4;
    const int t2 = 4 * c + w;                 // LSTM2 tile of this SIMD
    const int t1 = 2 * c + (w >> 1), kh = w & 1;
    // HO ("head once"): steps 0 .. T+1 = LSTM1 step s | head step s-1 on ONE wave of the XCD | LSTM2 step s-2;
    // otherwise steps 0 .. T = LSTM1 step s | head + LSTM2 step s-1 on every CU
    // (ring mode, HO: one more step, in which only the output head's CU computes: y[T-1] needs h2[T-1] in a gather buffer)
    const bool YH = HO && a.ring != 0;
    const int nph = (HO ? (YH ? T + 3 : T + 2) : T + 1) * ng;
    const PackedLayout P = packed_layout(XCD_H1, XCD_H2);
    if (wv == XCD_FW0) {
        // placement check by the first finish wave (groups with no work still publish their id and leave)
        const int loc = ng > 0 ? xcd_group_is_local(a.status, x) : 0;
        if (ng == 0) {
            if (lane == 0) __hip_atomic_store(a.status + 8 + blockIdx.x, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lane == 0) {
            XCD_LDS_ST(sLocal, loc > 0 && a.force_safe == 0);
            XCD_LDS_ST(sAbort, loc < 0);
            sArrive[0] = 0u; sArrive[1] = 0u;
            XCD_LDS_ST(sH1done, 0);
            if (loc == 0 && c == 0) atomicAdd(a.status + 3, 1u);
        }
    }
    if (ng == 0) return;
    for (int i = tid; i < XCD_NGMAX * 4 * 64; i += 512) (&sC2[0][0][0])[i] = 0.f;
    for (int i = tid; i < XCD_NGMAX * 2 * 64; i += 512) (&sC1[0][0][0])[i] = 0.f;

    if ((wv >= 4) == (XCD_FW0 == 0)) {
        // =========================================== product waves ===================================================
        float4 a2[32], a1[11], as_[HO ? 1 : 4];
        {
            const float4 *p2 = (const float4 *)(a.packed + P.w2p) + (long)t2 * 32 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 32; ++q) a2[q] = p2[q * 64];
            const float4 *p1 = (const float4 *)(a.packed + P.w1p) + ((long)t1 * 22 + 11 * kh) * 64 + lane;
#pragma unroll
            for (int q = 0; q < 11; ++q) a1[q] = p1[q * 64];
            if (!HO) {          // every CU computes the head: K quarter w of W_sel, resident
                const float4 *ps = (const float4 *)(a.packed + P.wselp) + (4 * w) * 64 + lane;
#pragma unroll
                for (int q = 0; q < (HO ? 1 : 4); ++q) as_[q] = ps[q * 64];
            }
        }
        // HO: W_ih2 of the tile as two more A fragments (K = 6 -> 8): lane (row i, k-quad position kq) holds W_ih2[row(i)][4 j + kq],
        // row i <-> (unit 4 t2 + (i >> 2), gate i & 3); packed as wih2p[unit][gate][8]
        float ax[2] = {0.f, 0.f};
        if (HO) {
            const float *px = a.packed + P.wih2p + ((long)(4 * t2 + (n >> 2)) * 4 + (n & 3)) * 8 + u;
            ax[0] = px[0];
            ax[1] = px[4];
        }
        const bool mtracer = a.trace && blockIdx.x == 0 && tid == (4 - XCD_FW0) * 64;
        // output head (YH): A fragments of W_out (rows 0..3 of a 16-row tile, the rest zero) - not resident: the CU needs them
        // once in 32 phases, two at a time straight from the packed image (L2-resident)
        // (a buffer descriptor + one lane offset: with pointers the compiler keeps 32 per-lane addresses and spills them)
        const __amdgpu_buffer_rsrc_t rwy = __builtin_amdgcn_make_buffer_rsrc((void *)(a.packed + P.woutp), 0, (XCD_H2 / 16) * 1024, 0x00020000);
        const unsigned ylane = lane * 16;
        auto yfrag = [&](int q) -> float4 {     // fragment of hexadecet q (wave-uniform)
            const xcd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rwy, ylane, (unsigned)q * 1024u, 0);
            return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        };
        __syncthreads();                        // phase 0's gather has landed
        if (XCD_LDS_LD(sAbort)) return;
        int pgi = 0, ps = 0;                    // phase p = ps * ng + pgi
        for (int p = 0; p < nph; ++p) {
            const bool ycu = YH && ps >= 3 && xcd_y_cu(ps, pgi) == c;
            const float4 *F = &sbuf[p & 1][0] + lane;
            const float *FBp = (const float *)(&sbuf[p & 1][XB_FB]) + u * 16 + n;   // frames_boxes[s-2][4 j + u][clip n]
            // B fragment of LSTM1 hexadecet 11 kh + j of [x 0..5 | h1 6..21]: the buffer is X0 | H1 | H2, so the lower-K
            // wave reads fragment j of the buffer, the upper-K wave fragment 11 + j
            const float4 *FL = F + (kh ? 11 * 64 : 0);
            const float4 *FH = F + XB_H1 + 4 * w * 64;     // head: K quarter w of h1
            if (mtracer) a.trace[(long)p * 8 + 0] = clock64();
            // 188 MFMAs on four accumulator chains (LSTM2 even / odd hexadecets, LSTM1, head).  The issue order is pinned
            // with sched_barrier after every row of independent MFMAs: left alone, the scheduler clusters the four MFMAs
            // of one hexadecet on the same accumulator (40-cycle dependent latency against a 32-cycle issue) and keeps
            // only one or two B fragments in flight.  Fragments are fetched one j-step (12 MFMAs ~ 400 cycles) ahead.
            f32x4 accH = {0.f, 0.f, 0.f, 0.f}, acc1, acc2a, acc2b;   // accH: !HO only
            auto products = [&](auto ym, auto yk) {
                constexpr int YM = decltype(ym)::value;
                constexpr bool YK = decltype(yk)::value;    // this CU computes the output head in this phase
                float4 fa[2], fb[2], fl[2];
                float4 ay0, ay1;
                f32x4 accYa, accYb;             // (local to this variant: no copies on the other CUs' path)
                if (YK) { ay0 = yfrag(2 * w); ay1 = yfrag(2 * w + 1); }
                fa[0] = F[XB_H2]; fb[0] = F[XB_H2 + 64]; fl[0] = FL[0];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int cur = j & 1, nxt = cur ^ 1;
                    const bool l1 = j < 11, hd = j >= 11 && j < 15;
                    int ROW = 4 * j;            // compile-time after unrolling: picks the MFMA form of the row
                    if (j == 0) {
                        XCD_MFMA0(acc2a, a2[0].x, fa[cur].x); XCD_MFMA0(acc2b, a2[1].x, fb[cur].x);
                        XCD_MFMA0(acc1, a1[0].x, fl[cur].x);
                    } else {
                        XCD_MFMA_LEAD(acc2a, a2[2 * j].x, fa[cur].x); XCD_MFMA(acc2b, a2[2 * j + 1].x, fb[cur].x);
                        if (l1) XCD_MFMA(acc1, a1[j].x, fl[cur].x);
                        if (!HO) {
                            if (j == 11) XCD_MFMA0(accH, as_[0].x, fl[cur].x);
                            else if (hd) XCD_MFMA(accH, as_[HO ? 0 : j - 11].x, fl[cur].x);
                        }
                    }
                    if (j + 1 < 16) fa[nxt] = F[XB_H2 + (2 * j + 2) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    ROW = 4 * j + 1;
                    XCD_MFMA_LEAD(acc2a, a2[2 * j].y, fa[cur].y); XCD_MFMA(acc2b, a2[2 * j + 1].y, fb[cur].y);
                    if (l1) XCD_MFMA(acc1, a1[j].y, fl[cur].y);
                    if (hd && !HO) XCD_MFMA(accH, as_[HO ? 0 : j - 11].y, fl[cur].y);
                    if (j + 1 < 16) fb[nxt] = F[XB_H2 + (2 * j + 3) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    ROW = 4 * j + 2;
                    XCD_MFMA_LEAD(acc2a, a2[2 * j].z, fa[cur].z); XCD_MFMA(acc2b, a2[2 * j + 1].z, fb[cur].z);
                    if (l1) XCD_MFMA(acc1, a1[j].z, fl[cur].z);
                    if (hd && !HO) XCD_MFMA(accH, as_[HO ? 0 : j - 11].z, fl[cur].z);
                    if (j + 1 < 11) fl[nxt] = FL[(j + 1) * 64];
                    else if (j + 1 < 15 && !HO) fl[nxt] = FH[(j + 1 - 11) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    ROW = 4 * j + 3;
                    XCD_MFMA_LEAD(acc2a, a2[2 * j].w, fa[cur].w); XCD_MFMA(acc2b, a2[2 * j + 1].w, fb[cur].w);
                    if (l1) XCD_MFMA(acc1, a1[j].w, fl[cur].w);
                    if (hd && !HO) XCD_MFMA(accH, as_[HO ? 0 : j - 11].w, fl[cur].w);
                    __builtin_amdgcn_sched_barrier(0);
                    // output head: K is split over the four product waves - wave w takes the hexadecet pairs j = w, w+4, w+8,
                    // w+12 (the B fragments this iteration holds), two chains, every other MFMA with the yielding gap
                    if (YK && (j & 3) == w) {
                        if (j < 4) {
                            XCD_MFMA0_G(accYa, ay0.x, fa[cur].x, XCD_GAP); XCD_MFMA0_G(accYb, ay1.x, fb[cur].x, 0);
                        } else {
                            XCD_MFMA_G(accYa, ay0.x, fa[cur].x, XCD_GAP); XCD_MFMA_G(accYb, ay1.x, fb[cur].x, 0);
                        }
                        XCD_MFMA_G(accYa, ay0.y, fa[cur].y, XCD_GAP); XCD_MFMA_G(accYb, ay1.y, fb[cur].y, 0);
                        XCD_MFMA_G(accYa, ay0.z, fa[cur].z, XCD_GAP); XCD_MFMA_G(accYb, ay1.z, fb[cur].z, 0);
                        XCD_MFMA_G(accYa, ay0.w, fa[cur].w, XCD_GAP); XCD_MFMA_G(accYb, ay1.w, fb[cur].w, 0);
                        if (j + 4 < 16) { ay0 = yfrag(2 * j + 8); ay1 = yfrag(2 * j + 9); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // HO: LSTM2's input part W_ih2 . frames_boxes[s-2] (K = 6 -> 8): two more MFMAs on the two LSTM2 chains
                if (HO) {
                    const int ROW = 64;
                    const float b0 = FBp[0], b1 = FBp[64];
                    XCD_MFMA(acc2a, ax[0], b0);
                    XCD_MFMA(acc2b, ax[1], b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (YK) {                       // this wave's K quarter of the output head -> the (otherwise unused) head slot
                    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(accYa), "+v"(accYb));
                    (&sHAND[p & 1][w][0] + lane)[128] = make_float4(accYa[0] + accYb[0], accYa[1] + accYb[1], accYa[2] + accYb[2], accYa[3] + accYb[3]);
                }
            };
            const std::false_type y0{};
            const std::integral_constant<bool, HO> y1{};
            if (ycu) {                          // (the gap pattern of three or more groups: the phase is longer anyway)
                products(std::integral_constant<int, HO ? 0 : XCD_NY3>{}, y1);
            } else if (YH && ps == T + 2) {     // the extra step of the output head: nothing to do on the other CUs
            } else if (ng == 1) products(std::integral_constant<int, -1>{}, y0);
            else if (ng == 2) products(std::integral_constant<int, XCD_NY2>{}, y0);
            else if (ng == 3) products(std::integral_constant<int, HO ? 0 : XCD_NY3>{}, y0);
            else products(std::integral_constant<int, XCD_NY4>{}, y0);
            XCD_MFMA_DRAIN(acc2a, acc2b, acc1, accH);
            if (mtracer) a.trace[(long)p * 8 + 1] = clock64();
            float4 *hd_ = &sHAND[p & 1][w][0] + lane;
            hd_[0] = make_float4(acc2a[0] + acc2b[0], acc2a[1] + acc2b[1], acc2a[2] + acc2b[2], acc2a[3] + acc2b[3]);
            hd_[64] = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
            if (!HO) hd_[128] = make_float4(accH[0], accH[1], accH[2], accH[3]);
            // barrier p.  No look at the abort word here (an LDS read and its wait per phase): when the finish waves leave
            // on an abort, the barrier only counts the waves that are still alive, and this wave runs its remaining phases
            // on whatever is in LDS - it stores nothing to memory - and ends.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ng == 1) {                      // exposed exchange: wait for this phase's finish + the next gather
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (++pgi == ng) { pgi = 0; ++ps; }
        }
        return;
    }

    // ============================================== finish waves ======================================================
    // fp32 MFMA runs at the VALU rate, so every VALU instruction of this wave takes its cycles from the product wave of the
    // same SIMD (measured: DESIGN.md section 7) - what this role hides is LATENCY (LDS, memory, hand-off), and its
    // instruction count is kept down: ONE buffer descriptor over the whole workspace with 32-bit offsets (wave-uniform part
    // on the scalar unit, one constant VGPR of lane offset) instead of 64-bit per-lane address arithmetic, nothing spilled.
    float4 wx[HO ? 1 : 8];                      // !HO: W_ih2 rows of the lane's own unit (the input part is added in the cell)
    if (!HO) {
        const float4 *px = (const float4 *)(a.packed + P.wih2p) + (long)(4 * t2 + u) * 8;
#pragma unroll
        for (int q = 0; q < (HO ? 1 : 8); ++q) wx[q] = px[q];
    }
    // history addressing: full history (slot = step) or rings of XCD_RING slots
    const unsigned NS = a.ring ? (unsigned)XCD_RING : (unsigned)(T + 1);
    const unsigned smask = a.ring ? (unsigned)(XCD_RING - 1) : 0xffffffffu;
    // ring mode: W_out[o][unit of this lane] (prediction_layer, learned_models.py:33,47): the cell's h leaves as a partial of y
    float wo[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.ring && !HO) {
        const int k = 4 * t2 + u;
        const float *pw = a.packed + P.woutp + (((k >> 4) * 64) + 16 * ((k & 15) >> 2)) * 4 + (k & 3);   // row o at lane offset o
#pragma unroll
        for (int o = 0; o < 4; ++o) wo[o] = pw[o * 4];
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(const void *)&sbuf[0][0];
    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void *)a.ws, 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = lane * 16;
    const unsigned xq_voff = ((6 * u) * 16 + n) * 16;            // this lane's first k-quad of the packed input
    const unsigned flag_voff = (lane & 31) * 4;
    // logits [B][15][T]: byte offset of (clip n of the group, slot 4u + r, frame 0), 0xffffffff = not stored
    unsigned lg_voff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) lg_voff[r] = (4 * u + r < OPNET_SLOTS_) ? (unsigned)(((n * OPNET_SLOTS_ + 4 * u + r) * T) * 4) : 0xffffffffu;
    bool alive = true;

    // this wave's share (chunks w, w+4, ...) of the gather of phase (group gi, step s) into LDS buffer `buf`
    // (w arrives as a parameter: the loop hands in a copy the compiler cannot see through, see XCD_CF below)
    auto gather = [&](int gi, int s, int buf, int w) {
        const unsigned gg = g0 + gi;
        const unsigned dst = lds0 + (unsigned)buf * (XB_F4 * 16) + w * 1024;
        // x[s] (slot s+1; past the end: the zero slot T+1), h1[s-1] (slot s), and HO: h2[s-3] (slot s-2), frames_boxes[s-2]
        // (slot s-1) / !HO: h2[s-2] (slot s-1)
        const unsigned ox0 = a.xp_off + ((gg * (T + 2) + (s < T ? s + 1 : T + 1)) * OPNET_KXQ) * 256 + w * 1024;
        const unsigned oh1 = a.h1_off + ((gg * NS + ((unsigned)(s <= T ? s : T) & smask)) * (XCD_H1 / 4)) * 256 + (w - 6) * 1024;
        const unsigned oh2 = a.h2_off + ((gg * NS + ((unsigned)(HO ? (s > 2 ? s - 2 : 0) : (s > 0 ? s - 1 : 0)) & smask)) * (XCD_H2 / 4)) * 256 + (w - 22) * 1024;
        const unsigned ofb = a.fb_off + (gg * NS + ((unsigned)(s > 1 ? s - 1 : 0) & smask)) * 1024;
#pragma unroll
        for (int j = 0; j < (XB_CHUNKS + 3) / 4; ++j) {
            // chunk 4j + w (wave-uniform): X0 = chunks 0..5, H1 = 6..21, H2 = 22..53, FB = 54
            if (4 * j + 3 < 6) xcd_glds16(rws, lane16, ox0 + j * 4096, dst + j * 4096);
            else if (4 * j >= 6 && 4 * j + 3 < 22) xcd_glds16(rws, lane16, oh1 + j * 4096, dst + j * 4096);
            else if (4 * j >= 22 && 4 * j + 3 < 54) xcd_glds16(rws, lane16, oh2 + j * 4096, dst + j * 4096);
            else if (4 * j + w < 6) xcd_glds16(rws, lane16, ox0 + j * 4096, dst + j * 4096);
            else if (4 * j + w < 22) xcd_glds16(rws, lane16, oh1 + j * 4096, dst + j * 4096);
            else if (4 * j + w < 54) xcd_glds16(rws, lane16, oh2 + j * 4096, dst + j * 4096);
            else if (HO && 4 * j + w == 54) xcd_glds16(rws, lane16, ofb, dst - w * 1024 + 54 * 1024);
        }
    };
    auto flags_ready = [&](int gn, unsigned need) -> bool {
        const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rws, flag_voff, a.flags_off + (g0 + gn) * (XCD_CUS * 4), 16);   // sc1
        return __all(v >= need);
    };
    // wait until every CU of the group's XCD has published step sn - 1, then gather phase (gn, sn)
    auto poll_gather = [&](int gn, int sn, int buf, int phase, int w, unsigned cf) {
        if (!alive || (cf & XCD_CF_DBG1)) return;
        if (sn > 0 && !flags_ready(gn, (unsigned)sn))
            alive = xcd_wait_flags_ws(rws, a.flags_off + (g0 + gn) * (XCD_CUS * 4), (unsigned)sn, a.status_off, phase);
        if (alive) gather(gn, sn, buf, w);
        else XCD_LDS_ST(sAbort, 1);
    };

    gather(0, 0, 0, w);                         // phase 0 reads only zero slots and x[0]: nothing to wait for
    if (ng >= 2 && nph > 1) gather(1, 0, 1, w); // phase 1 = group 1, step 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (XCD_LDS_LD(sAbort)) return;
    // XCD_CF: the loop's wave-uniform, loop-invariant conditions as bits of ONE scalar that is made opaque at the top of every phase
    // (and w with it).  Left to itself the compiler hoists each such condition out of the loop as a 64-bit lane mask - 15 of them
    // here, 30 SGPRs - runs out of SGPRs, spills them into VGPR lanes, and every v_readlane that fetches one back inside the loop is a
    // VALU instruction taken from the MFMA stream of this SIMD (a dozen per phase, measured: ~200 cycles of a 7 300-cycle phase).
    // Tested at the point of use, a bit costs one scalar instruction and no register.
    const unsigned cfbits = (ng >= 4 ? XCD_CF_NG4 : 0u) | (ng >= 2 ? XCD_CF_NG2 : 0u) | (ng == 1 ? XCD_CF_NG1 : 0u)
                            | ((a.debug & 1) ? XCD_CF_DBG1 : 0u) | ((a.debug & 2) ? XCD_CF_DBG2 : 0u) | ((a.debug & 4) ? XCD_CF_DBG4 : 0u)
                            | ((a.debug & 8) ? XCD_CF_DBG8 : 0u) | (a.ring ? XCD_CF_RING : 0u)
                            | (__builtin_amdgcn_readfirstlane(XCD_LDS_LD(sLocal)) != 0 ? XCD_CF_LOCAL : 0u)
                            | ((a.trace && blockIdx.x == 0 && wv == XCD_FW0) ? XCD_CF_TRACE : 0u);

    int gi = 0, s = 0;                          // phase fp = s * ng + gi
    for (int fp = 0; fp < nph; ++fp) {
        unsigned cf = __builtin_amdgcn_readfirstlane(cfbits);
        int wq = w;
        asm volatile("" : "+s"(cf), "+s"(wq));
        const bool local = (cf & XCD_CF_LOCAL) != 0, tracer = (cf & XCD_CF_TRACE) != 0 && lane == 0;
        const unsigned gg = g0 + gi;
        // the phase after next (two or more groups) / the next phase (one group)
        const int ahead = (cf & XCD_CF_NG2) ? 2 : 1;
        int gn = gi + ahead, sn = s;
        while (gn >= ng) { gn -= ng; ++sn; }
        // Who computes the selection head of step s-1: !HO every finish wave (every CU needs frames_boxes at once); HO wave 0
        // of the step's head CU alone.  It needs boxes[s-1] - this lane's clip n and slots 4u .. 4u+3 (24 consecutive k = 6
        // float4 of the packed input, each a coalesced 256-B run per 16 clips), straight from global memory (read-only here,
        // L2-resident) - and, HO, the 16 A fragments of W_sel (not resident anywhere: one wave in 128 needs them per phase);
        // all issued before the barrier - which must therefore not drain vmcnt - so that the round trips are over when the
        // phase's sums arrive
        const bool head_cu = HO && xcd_head_cu(s, gi) == c && s >= 1 && s <= T;
        const bool head_wave = HO ? (head_cu && wq == 0) : (s >= 1);
        xcd_u32x4 xq[6];
        float4 wsel[HO ? 16 : 1];
        if (head_wave) {
            const unsigned xs = a.xp_off + ((gg * (T + 2) + s) * OPNET_KXQ) * 256;
#pragma unroll
            for (int j = 0; j < 6; ++j) xq[j] = __builtin_amdgcn_raw_buffer_load_b128(rws, xq_voff + j * 256, xs, 0);
            if (HO) {
                const float4 *ps = (const float4 *)(XCD_KARG(const float *, packed) + P.wselp) + lane;
#pragma unroll
                for (int q = 0; q < (HO ? 16 : 1); ++q) wsel[q] = ps[q * 64];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // barrier fp: the phase's accumulators are in sHAND[fp & 1]
        asm volatile("" ::: "memory");
        if (tracer) XCD_KARG(unsigned long long *, trace)[(long)fp * 8 + 2] = clock64();
        if (XCD_LDS_LD(sAbort)) return;
        // If the phase after next already has its inputs published (four or more groups per XCD: its group finished its
        // previous step more than a window ago) the gather goes first and lands under the finish; otherwise the finish
        // goes first and the gather follows it (three groups: published by then; two: it is THIS finish - the product
        // waves wait for the exchange)
        bool early = false;
        if ((cf & XCD_CF_NG4) && !head_cu && fp + 2 < nph && alive && !(cf & XCD_CF_DBG1) && (sn == 0 || flags_ready(gn, (unsigned)sn))) {
            gather(gn, sn, fp & 1, wq);
            early = true;
        }
        if (tracer) XCD_KARG(unsigned long long *, trace)[(long)fp * 8 + 3] = clock64();

        const float4 *H = &sHAND[fp & 1][0][0] + lane;
        // ---- selection head of step s-1 (learned_models.py:40-43,50): logits, softmax, einsum -> frames_boxes[s-1].  !HO: it
        //      feeds this wave's LSTM2 cell right away; HO: it travels to every CU with this phase's publish and enters LSTM2
        //      as two MFMAs one step later -----------------------------------------------------------------------------
        float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
        if (HO && head_wave && !(alive && !(cf & XCD_CF_DBG2)) && lane == 0) XCD_LDS_ST(sH1done, fp + 1);   // skipped head: still release the buffer
        if (head_wave && alive && !(cf & XCD_CF_DBG2)) {
            float v[4];
            if (HO) {
                // the whole 16 x 256 x 16-clip product on this one wave: B fragments = h1[s-1] out of the phase's gather buffer
                // (which the gather of the phase after next will overwrite: sH1done releases it), two accumulator chains
                const float4 *Fh = &sbuf[fp & 1][XB_H1] + lane;
                float4 bh[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) bh[q] = Fh[q * 64];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) XCD_LDS_ST(sH1done, fp + 1);
                f32x4 ha = {0.f, 0.f, 0.f, 0.f}, hb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    ha = __builtin_amdgcn_mfma_f32_16x16x4f32(wsel[HO ? q : 0].x, bh[q].x, ha, 0, 0, 0);
                    hb = __builtin_amdgcn_mfma_f32_16x16x4f32(wsel[HO ? q + 1 : 0].x, bh[q + 1].x, hb, 0, 0, 0);
                    ha = __builtin_amdgcn_mfma_f32_16x16x4f32(wsel[HO ? q : 0].y, bh[q].y, ha, 0, 0, 0);
                    hb = __builtin_amdgcn_mfma_f32_16x16x4f32(wsel[HO ? q + 1 : 0].y, bh[q + 1].y, hb, 0, 0, 0);
                    ha = __builtin_amdgcn_mfma_f32_16x16x4f32(wsel[HO ? q : 0].z, bh[q].z, ha, 0, 0, 0);
                    hb = __builtin_amdgcn_mfma_f32_16x16x4f32(wsel[HO ? q + 1 : 0].z, bh[q + 1].z, hb, 0, 0, 0);
                    ha = __builtin_amdgcn_mfma_f32_16x16x4f32(wsel[HO ? q : 0].w, bh[q].w, ha, 0, 0, 0);
                    hb = __builtin_amdgcn_mfma_f32_16x16x4f32(wsel[HO ? q + 1 : 0].w, bh[q + 1].w, hb, 0, 0, 0);
                }
                v[0] = ha[0] + hb[0]; v[1] = ha[1] + hb[1]; v[2] = ha[2] + hb[2]; v[3] = ha[3] + hb[3];
            } else {
                const float4 h0 = H[0 * XH_F4 + 128], h1 = H[1 * XH_F4 + 128], h2 = H[2 * XH_F4 + 128], h3 = H[3 * XH_F4 + 128];
                v[0] = ((h0.x + h1.x) + h2.x) + h3.x; v[1] = ((h0.y + h1.y) + h2.y) + h3.y;
                v[2] = ((h0.z + h1.z) + h2.z) + h3.z; v[3] = ((h0.w + h1.w) + h2.w) + h3.w;
            }
            if (HO || (c == ((s - 1) & (XCD_CUS - 1)) && w == 0)) {   // wave-uniform: this wave writes the step's logits
                float *lg = XCD_KARG(float *, logits) + ((long)gg * 16 * OPNET_SLOTS_) * T + (s - 1);
                const bool clip_ok = (int)(gg * 16 + n) < a.B;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (clip_ok && lg_voff[r] != 0xffffffffu) *(float *)((char *)lg + lg_voff[r]) = v[r];
            }
            float m = fmaxf(fmaxf(v[0], v[1]), v[2]);
            if (u < 3) m = fmaxf(m, v[3]);                        // slot 15 does not exist
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = __expf(v[r] - m);
            if (u == 3) e[3] = 0.f;
            float sum = (e[0] + e[1]) + (e[2] + e[3]);
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = __builtin_amdgcn_rcpf(sum);
            const float pr[4] = {e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv};
            // frames_boxes[n][f] = sum_o boxes[n][s-1][o][f] * p[o] (einsum "bfot,bfo->bft"): this lane's four slots as one
            // ascending fmaf chain per feature, then the four lane groups of the clip are added (0+1) + (2+3)
            const float xf[24] = {__uint_as_float(xq[0].x), __uint_as_float(xq[0].y), __uint_as_float(xq[0].z), __uint_as_float(xq[0].w),
                                  __uint_as_float(xq[1].x), __uint_as_float(xq[1].y), __uint_as_float(xq[1].z), __uint_as_float(xq[1].w),
                                  __uint_as_float(xq[2].x), __uint_as_float(xq[2].y), __uint_as_float(xq[2].z), __uint_as_float(xq[2].w),
                                  __uint_as_float(xq[3].x), __uint_as_float(xq[3].y), __uint_as_float(xq[3].z), __uint_as_float(xq[3].w),
                                  __uint_as_float(xq[4].x), __uint_as_float(xq[4].y), __uint_as_float(xq[4].z), __uint_as_float(xq[4].w),
                                  __uint_as_float(xq[5].x), __uint_as_float(xq[5].y), __uint_as_float(xq[5].z), __uint_as_float(xq[5].w)};
            float fbv[OPNET_FEATS_];
#pragma unroll
            for (int f = 0; f < OPNET_FEATS_; ++f) {
                float t = pr[0] * xf[f];
                t = fmaf(pr[1], xf[6 + f], t);
                t = fmaf(pr[2], xf[12 + f], t);
                t = fmaf(pr[3], xf[18 + f], t);
                t += __shfl_xor(t, 16);
                t += __shfl_xor(t, 32);
                fbv[f] = t;
            }
            xa = make_float4(fbv[0], fbv[1], fbv[2], fbv[3]);
            xb = make_float4(fbv[4], fbv[5], 0.f, 0.f);
          if (TRAIN && (HO || (c == ((s - 1) & (XCD_CUS - 1)) && w == 0))) {
            // the step's slot probabilities [4 slot quads][32 clips] and frames_boxes [2 k-quads][32 clips] in the chain's layouts
            const unsigned trb = (unsigned)(s - 1) * (unsigned)a.RB + (gg >> 1), half = (gg & 1u) * 256u;
            xcd_st4(rws, (u * 32 + n) * 16, a.tr_ps + trb * 2048 + half, make_float4(pr[0], pr[1], pr[2], pr[3]));
            if (u == 0) {
                xcd_st4(rws, n * 16, a.tr_x2 + trb * 1024 + half, xa);
                xcd_st4(rws, n * 16, a.tr_x2 + trb * 1024 + 512 + half, xb);
            }
          }
          if (HO) {
            // every lane now holds the clip's six values: lane (n, u) stores features u and u + 4 (6, 7 = padding zeros) of
            // slot s (= frames_boxes[s-1]) as [feature][clip]
            const float f_lo = u == 0 ? fbv[0] : u == 1 ? fbv[1] : u == 2 ? fbv[2] : fbv[3];
            const float f_hi = u == 0 ? fbv[4] : u == 1 ? fbv[5] : 0.f;
            const unsigned fo = a.fb_off + (gg * NS + ((unsigned)s & smask)) * 1024;
            xcd_store4(rws, (u * 16 + n) * 4, fo, f_lo, local);
            xcd_store4(rws, ((u + 4) * 16 + n) * 4, fo, f_hi, local);
          }
        }
        if (tracer) XCD_KARG(unsigned long long *, trace)[(long)fp * 8 + 4] = clock64();
        // ---- LSTM2 cell (learned_models.py:46): lane (clip n, unit 4 t2 + u).  HO: step s-2, the gates arrive complete (the
        //      input part W_ih2 . frames_boxes[s-2] rode the MFMA stream); !HO: step s-1, the input part is added here -------
        if ((HO ? s >= 2 && s <= T + 1 : s >= 1) && alive && !(cf & XCD_CF_DBG4)) {
            const float4 g2 = H[w * XH_F4];
            float g[4] = {g2.x, g2.y, g2.z, g2.w};
            if (!HO) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 w0 = wx[HO ? 0 : 2 * r], w1 = wx[HO ? 0 : 2 * r + 1];
                    float xsum = w0.x * xa.x;
                    xsum = fmaf(w0.y, xa.y, xsum);
                    xsum = fmaf(w0.z, xa.z, xsum);
                    xsum = fmaf(w0.w, xa.w, xsum);
                    xsum = fmaf(w1.x, xb.x, xsum);
                    xsum = fmaf(w1.y, xb.y, xsum);
                    g[r] += xsum;
                }
            }
            float cc = sC2[gi][w][lane];
            float4 gs;
            const float h = TRAIN ? lstm_cell_g(g[0], g[1], g[2], g[3], &cc, &gs) : lstm_cell(g[0], g[1], g[2], g[3], &cc);
            sC2[gi][w][lane] = cc;
            sTR[w][0][n * 4 + u] = h;
            XCD_WAVE_LDS_SYNC();
            if (lane < 16) {
                const float4 hv = *(const float4 *)&sTR[w][0][lane * 4];
                xcd_store16(rws, lane16, a.h2_off + (((gg * NS + ((unsigned)(HO ? s - 1 : s) & smask)) * (XCD_H2 / 4) + t2) * 16) * 16, hv, local);
                if (TRAIN)      // h2[t] in the chain's history: [slot t + 1][row block][unit quad][32 clips] float4
                    xcd_st4(rws, lane16, a.tr_h2 + ((((unsigned)(HO ? s - 1 : s) * (unsigned)a.RB + (gg >> 1)) * (XCD_H2 / 4) + t2) * 32 + (gg & 1u) * 16) * 16, hv);
            }
            if (TRAIN) {        // gates (post-activation) of step t and c[t] (slot t + 1): lane (clip n, unit 4 t2 + u)
                const unsigned tstep = (unsigned)(HO ? s - 2 : s - 1), half = (gg & 1u) * 16;
                xcd_st4(rws, (u * 32 + n) * 16, a.tr_g2 + (((tstep * (unsigned)a.RB + (gg >> 1)) * XCD_H2 + 4 * t2) * 32 + half) * 16, gs);
                xcd_st1(rws, (u * 32 + n) * 4, a.tr_c2 + ((((tstep + 1) * (unsigned)a.RB + (gg >> 1)) * XCD_H2 + 4 * t2) * 32 + half) * 4, cc);
            }
            if ((cf & XCD_CF_RING) && !HO)   // this lane's terms of y = W_out h2: four products, summed by the CU's last-arriving wave (no shuffles
                                 // here: every VALU instruction of a finish wave is paid for by the MFMA stream of its SIMD)
                sY[fp & 1][w][lane] = make_float4(wo[0] * h, wo[1] * h, wo[2] * h, wo[3] * h);
        }
        // ---- LSTM1 cell of step s (learned_models.py:39), by the upper-K wave of each pair ---------------------
        if ((wq & 1) && s < T && alive && !(cf & XCD_CF_DBG4)) {
            const float4 lo = H[(w - 1) * XH_F4 + 64], hi = H[w * XH_F4 + 64];
            float cc = sC1[gi][w >> 1][lane];
            float4 gs;
            const float h = TRAIN ? lstm_cell_g(lo.x + hi.x, lo.y + hi.y, lo.z + hi.z, lo.w + hi.w, &cc, &gs)
                                  : lstm_cell(lo.x + hi.x, lo.y + hi.y, lo.z + hi.z, lo.w + hi.w, &cc);
            sC1[gi][w >> 1][lane] = cc;
            sTR[w][1][n * 4 + u] = h;
            XCD_WAVE_LDS_SYNC();
            if (lane < 16) {
                const float4 hv = *(const float4 *)&sTR[w][1][lane * 4];
                xcd_store16(rws, lane16, a.h1_off + (((gg * NS + ((unsigned)(s + 1) & smask)) * (XCD_H1 / 4) + t1) * 16) * 16, hv, local);
                if (TRAIN)
                    xcd_st4(rws, lane16, a.tr_h1 + ((((unsigned)(s + 1) * (unsigned)a.RB + (gg >> 1)) * (XCD_H1 / 4) + t1) * 32 + (gg & 1u) * 16) * 16, hv);
            }
            if (TRAIN) {
                const unsigned half = (gg & 1u) * 16;
                xcd_st4(rws, (u * 32 + n) * 16, a.tr_g1 + ((((unsigned)s * (unsigned)a.RB + (gg >> 1)) * XCD_H1 + 4 * t1) * 32 + half) * 16, gs);
                xcd_st1(rws, (u * 32 + n) * 4, a.tr_c1 + ((((unsigned)(s + 1) * (unsigned)a.RB + (gg >> 1)) * XCD_H1 + 4 * t1) * 32 + half) * 4, cc);
            }
        }
        if (tracer) XCD_KARG(unsigned long long *, trace)[(long)fp * 8 + 5] = clock64();
        // ---- publish: every finish wave drains its stores (and DMA) and arrives at an LDS counter; the last one stores
        //      the CU's flag -------------------------------------------------------------------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (the counter only grows - four arrivals per use, two phases apart behind a barrier: the fourth arriver sees 3 mod 4)
        bool last = false;
        if (alive && !(cf & XCD_CF_DBG8)) {
            last = (xcd_lds_add_lane0((unsigned)(unsigned long long)(const void *)&sArrive[fp & 1], 1u) & 3u) == 3u;
            if (last && lane == 0) {
                // (plain store: the line stays in this XCD's L2; otherwise write-through, as xcd_store_flag)
                if (local) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(s + 1), rws, 0, a.flags_off + (gg * XCD_CUS + c) * 4, 0);
                else __builtin_amdgcn_raw_buffer_store_b32((unsigned)(s + 1), rws, 0, a.flags_off + (gg * XCD_CUS + c) * 4, 16);
            }
        }
        if ((cf & XCD_CF_RING) && !HO && s >= 1 && !(cf & XCD_CF_DBG4)) {
            // the CU's last-arriving finish wave (the others' LDS writes precede their arrival in their own LDS queues): the CU's
            // partial of y[t] = the four waves' parts in wave order -> ypart [group][t][CU][clip]; read after the launch, so it is
            // a plain store behind the flag, off the hand-off's critical path
            if (last) {
                // lane (clip n, output o = u): the CU's 16 units in (wave, unit) order, one sequential chain
                const float *py = (const float *)&sY[fp & 1][0][0] + n * 4 + u;
                float sum = py[0];
#pragma unroll
                for (int k = 1; k < 16; ++k) sum += py[((k >> 2) * 64 + (k & 3) * 16) * 4];
                const unsigned t_y = (unsigned)(s - 1);
                xcd_store4(rws, (n * 4 + u) * 4, a.yp_off + ((gg * (unsigned)T + t_y) * XCD_CUS + c) * 256, sum, true);
            }
        }
        if (HO && (cf & XCD_CF_RING) && wq == 2 && s >= 3 && xcd_y_cu(s, gi) == c && alive && !(cf & XCD_CF_DBG4)) {
            // output head of step s-3 (prediction_layer, learned_models.py:33,47): the four product waves' K quarters in wave order;
            // rows 0..3 of the tile = the registers of lanes 0..15 (clip = lane).  Behind the publish: nothing in the launch reads y
            const float4 p0 = H[0 * XH_F4 + 128], p1 = H[1 * XH_F4 + 128], p2 = H[2 * XH_F4 + 128], p3 = H[3 * XH_F4 + 128];
            const long b = (long)gg * 16 + lane;
            if (lane < 16 && b < a.B)
                XCD_KARG(float4 *, y)[b * T + (s - 3)] = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                                                               ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
        }
        if (tracer) XCD_KARG(unsigned long long *, trace)[(long)fp * 8 + 6] = clock64();
        if (!early && fp + ahead < nph) {
            if (head_cu && (cf & XCD_CF_NG2)) {           // the gather fills the buffer this CU's head wave read h1 from
                while (XCD_LDS_LD(sH1done) < fp + 1 && !XCD_LDS_LD(sAbort)) __builtin_amdgcn_s_sleep(1);
            }
            poll_gather(gn, sn, (fp + ahead) & 1, fp, wq, cf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (tracer) XCD_KARG(unsigned long long *, trace)[(long)fp * 8 + 7] = clock64();
        if (cf & XCD_CF_NG1) {
            __syncthreads();
            if (XCD_LDS_LD(sAbort)) return;
        }
        if (++gi == ng) { gi = 0; ++s; }
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


// ring mode: y[b][t][0..3] = the 32 CUs' partials of W_out h2[t] in CU order (each one sequential chain over the CU's 16 units in
// (wave, unit) order); one thread per (group, t, clip).  An aborted launch poisons y with NaN.
__global__ void __launch_bounds__(256) opnet_xcd_y_reduce(const XcdArgs a, float *__restrict__ y)
{
    const long total = (long)a.NGT * a.T * 16;
    const float4 *yp = (const float4 *)(a.ws + a.yp_off);
    const bool bad = a.status[0] != 0u;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i & 15);
        const long gt = i >> 4;                                  // gg * T + t
        const float4 *p = yp + gt * (XCD_CUS * 16) + n;
        float4 sum = p[0];
        for (int cu = 1; cu < XCD_CUS; ++cu) {
            const float4 v = p[cu * 16];
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        if (bad) sum = make_float4(NAN, NAN, NAN, NAN);
        const long gg = gt / a.T, t = gt - gg * a.T;
        const long b = gg * 16 + n;
        if (b < a.B) ((float4 *)y)[b * a.T + t] = sum;
    }
}


// ring mode, head-once form: y was written inside the launch; an aborted launch poisons it with NaN like the other tails do
__global__ void __launch_bounds__(256) opnet_xcd_y_poison(const XcdArgs a, float *__restrict__ y)
{
    if (a.status[0] == 0u) return;
    const long total = (long)a.B * a.T;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256)
        ((float4 *)y)[i] = make_float4(NAN, NAN, NAN, NAN);
}
