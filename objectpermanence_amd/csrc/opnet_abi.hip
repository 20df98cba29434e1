// opnet_abi.hip - host side of libopnet_hip.so: the C ABI declared in include/opnet_hip.h.
// Plain pointers and sizes in, HIP launches on the caller's stream out; no torch types.
#include "opnet_kernels.hip"
#include "opnet_xcd_kernels.hip"
#include "opnet_train_kernels.hip"
#include "opnet_xcd4_kernels.hip"
#include "seq_kernels.hip"
#include "seq_xcd_kernels.hip"
#include "seq_xcdt_kernels.hip"
#include "seq_xcd_bwd_kernels.hip"
#include "conv_kernels.hip"
#include "wino_kernels.hip"
#include "ffn_kernels.hip"
#include "attn_kernels.hip"
#include "enc_train_kernels.hip"
#include "attn_train_kernels.hip"

#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/opnet_hip.h"

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// error text set from the detector-head translation unit (opdet_abi.hip)
__attribute__((visibility("hidden"))) int opnet_set_error(int code, const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(OPNET_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

static int env_int(const char *name, int dflt);
static int launch_gemm_k256(const ConvArgs &c, long M, hipStream_t st);   // K = 256 products: resident-token tile or the tiled conv kernel

// 128 x 64 instead of 128 x 128 tiles for the LDS-DMA kernel?  Same arithmetic per output element (one K-ordered chain per element
// whatever the tile), so the choice is free; measured per shape on the MI355X (tools/probes/gemm_probe.hip, profiles/r5_conv_gemm_probe.txt):
// the narrow tiles (four resident workgroups per CU, twice the tiles) win by 6-11 % on short K (K <= 512: the K = 256 token-wise
// products, the 1 x 1 expand convs) and when the wide tiles would leave a nearly empty second round (1.0-1.4 rounds of 768: the
// stride-16 layers of a 16-frame pass, fc6); the wide tiles win by 3-8 % on the long-K 3 x 3 layers with many or fewer-than-one rounds.
static bool conv_prefers_bn64(const ConvArgs &c, long M)
{
    if (c.Cout <= 64) return true;
    const int K = c.KH * c.KW * c.Cin;
    const double rounds128 = (double)((M + 127) / 128) * ((c.Cout + 127) / 128) / 768.0;
    // short K: fewer, fatter workgroups keep their K loop fed in small launches - except where the wide tiles would leave a nearly
    // empty second round (one frame's layer1 / P2 lateral 1 x 1 convs: 850 wide tiles; the one-frame call 4.91 -> 4.80 ms)
    if (K <= 512) return rounds128 >= 2.0 || (rounds128 > 1.0 && rounds128 < 1.4);
    return rounds128 > 1.0 && rounds128 < 1.4;
}

// the LDS-staged conv / GEMM kernels: 128 x {128, 64} tiles; LDS-DMA staging (3 stages) needs Cin % 16 == 0
static void launch_conv_tiled(const ConvArgs &c, long M, hipStream_t st)
{
    // the tiled kernel addresses X and W with 32-bit byte offsets through buffer descriptors
    if ((long)c.N * c.H * c.W * c.Cin * 4 >= (1L << 31) || (long)c.Cout * c.KP * 4 >= (1L << 31)) {
        conv2d_nhwc<<<dim3((unsigned)((M + 63) / 64), (c.Cout + 63) / 64, 1), 256, 0, st>>>(c);
        return;
    }
    const bool al = (c.Cin & 15) == 0;
    const unsigned gx = (unsigned)((M + 127) / 128);
    if (c.Cin == 4 && c.XS == 0 && c.KW >= 4 && c.ksplit <= 1) {     // the stem: a tap per k-quad through the same LDS-DMA stages
        conv2d_nhwc_glds<64, 3, true><<<dim3(gx, (c.Cout + 63) / 64, 1), 256, 0, st>>>(c);
        return;
    }
    if (c.Cout > 64 && !(al && env_int("OPNET_CONV_BN64", 1) && conv_prefers_bn64(c, M))) {
        const dim3 g(gx, (c.Cout + 127) / 128, 1);
        if (al) conv2d_nhwc_glds<128, 3><<<g, 256, 0, st>>>(c);
        else conv2d_nhwc_tiled<128><<<g, 256, 0, st>>>(c);
    } else {
        const dim3 g(gx, (c.Cout + 63) / 64, 1);
        if (al) conv2d_nhwc_glds<64, 3><<<g, 256, 0, st>>>(c);
        else conv2d_nhwc_tiled<64><<<g, 256, 0, st>>>(c);
    }
}

// self-attention over one sequence of S tokens: LDS-DMA flash kernel for power-of-two head sizes; two query
// fragments per wave (K/V fragment reuse) once that still leaves >= one workgroup per CU; a key split when the
// workgroup count is a small non-multiple of the CU count (needs scratch: KS partial outputs + their softmax stats)
static size_t attention_scratch_bytes(long S, int E, int nhead, int KS)
{
    return KS <= 1 ? 0 : (size_t)KS * S * E * 4 + (size_t)KS * S * nhead * 8;
}

static int attention_key_split(long S, int nhead, int QF)
{
    const long W = ((S + 64 * QF - 1) / (64 * QF)) * nhead;
    if (W >= 8 * 256 || (S + 15) / 16 < 64) return 1;         // enough workgroups / too few key tiles to split
    int best = 1;
    double best_cost = (double)((W + 255) / 256);              // rounds of 256 CUs, in units of one full sweep
    for (int ks = 2; ks <= 4; ++ks) {
        const double cost = (double)((W * ks + 255) / 256) / ks * 1.03;    // +3 % per split for the merge pass
        if (cost < best_cost - 1e-9) { best_cost = cost; best = ks; }
    }
    return best;
}

// nseg sequences of S tokens back to back, each attending to itself: the key split (which changes the order of a query's sums)
// is chosen from ONE segment's shape, so a segment is computed exactly as that sequence alone would be (bit-identical); the
// segments only add workgroups (grid.x).  The key split's scratch covers all nseg * S rows.
template <int HD>
static void launch_attention_hd(const float *qkv, float *att, int S, int nseg, int E, int nhead, void *scratch,
                                size_t scratch_bytes, hipStream_t st)
{
    const float scale = 1.0f / sqrtf((float)HD);
    // (two query fragments per wave once ONE segment alone leaves a workgroup per CU.  Counting all segments of a pass instead is
    // bit-identical - which queries share a wave enters no query's arithmetic; tested in round 5 - and slower: at head size 128 the
    // QF = 2 form needs 206 registers, one wave per SIMD: sixteen 16-clip requests per pass 16.8 ms against 15.7)
    const int QF = (long)((S + 127) / 128) * nhead >= 256 ? 2 : 1;
    const long Stot = (long)S * nseg;
    int KS = attention_key_split(S, nhead, QF);                   // from the segment's shape alone, like QF: same arithmetic as alone
    while (KS > 1 && (!scratch || attention_scratch_bytes(nseg > 1 ? Stot : S, E, nhead, KS) > scratch_bytes)) --KS;
    float *opart = KS > 1 ? (float *)scratch : nullptr;
    float2 *ml = KS > 1 ? (float2 *)((char *)scratch + (size_t)KS * Stot * E * 4) : nullptr;
    if (QF == 2)
        attention_glds<HD, 2><<<dim3((unsigned)((S + 127) / 128) * nseg, nhead, KS), 256, 0, st>>>(qkv, att, S, E, scale, opart, ml, nseg);
    else
        attention_glds<HD, 1><<<dim3((unsigned)((S + 63) / 64) * nseg, nhead, KS), 256, 0, st>>>(qkv, att, S, E, scale, opart, ml, nseg);
    if (KS > 1) {
        const long n = Stot * (E / 4);
        attention_merge<<<(unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256), 256, 0, st>>>(opart, ml, att, (int)Stot, E, nhead, KS);
    }
}

static void launch_attention(const float *qkv, float *att, int S, int nseg, int E, int nhead, int hd, void *scratch,
                             size_t scratch_bytes, hipStream_t st)
{
    if ((long)S * 3 * E * 4 >= (1L << 31)) hd = 0;      // 32-bit buffer offsets: fall through to the direct kernel
    switch (hd) {
    case 16: launch_attention_hd<16>(qkv, att, S, nseg, E, nhead, scratch, scratch_bytes, st); break;
    case 32: launch_attention_hd<32>(qkv, att, S, nseg, E, nhead, scratch, scratch_bytes, st); break;
    case 64: launch_attention_hd<64>(qkv, att, S, nseg, E, nhead, scratch, scratch_bytes, st); break;
    case 128: launch_attention_hd<128>(qkv, att, S, nseg, E, nhead, scratch, scratch_bytes, st); break;
    default:
        for (int g = 0; g < nseg; ++g)
            attention_f32<<<dim3((S + 63) / 64, nhead, 1), 256, 0, st>>>(qkv + (size_t)g * S * 3 * E, att + (size_t)g * S * E, S, E,
                                                                          E / nhead, 1.0f / sqrtf((float)(E / nhead)));
    }
}

static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

extern "C" int opnet_hip_abi_version(void) { return OPNET_HIP_ABI_VERSION; }
extern "C" const char *opnet_last_error(void) { return g_err; }

// the step kernel comes in two register-chunk sizes (opnet_kernels.hip, load_a_chunk): 8 for one row block, 4 beyond;
// from OPNET_WIDE_MIN (3) row blocks on, the two-tiles-per-workgroup form (opnet_step_wide) takes over
typedef void (*opnet_step_fn)(const StepArgs, const int);
static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}
static bool step_is_wide(const StepArgs &a) { return !a.mlp && a.RB >= env_int("OPNET_WIDE_MIN", 3); }
// one row block: K split over 8 waves instead of 4 - a wave's whole slice is then one 4-hexadecet chunk (one fetch round
// trip, 32 instead of 64 chained MFMAs) at 119 VGPRs; the launch is a latency chain there
static bool step_is_nw8(const StepArgs &a)
{
    return !step_is_wide(a) && !getenv("OPNET_STEP_CH") && a.RB <= env_int("OPNET_NW8_MAX_RB", 1);
}
// inference at one row block: the scalar-argument (kernarg-preload) form of the same kernel
static bool step_is_preload(const StepArgs &a)
{
    return step_is_nw8(a) && !a.mlp && !a.train && env_int("OPNET_PRELOAD", 1) != 0;
}
static int step_threads(const StepArgs &a) { return step_is_nw8(a) ? 8 * 64 : OPNET_THREADS; }
static opnet_step_fn step_kernel(const StepArgs &a)
{
    const char *force = getenv("OPNET_STEP_CH");          // "4" / "8": measurement override (4-wave kernels)
    const bool small_chunks = force ? atoi(force) == 4 : a.RB >= 2;
    if (step_is_wide(a)) return small_chunks ? opnet_step_wide<4> : opnet_step_wide<8>;
    if (step_is_nw8(a)) return opnet_step<4, 8>;
    return small_chunks ? opnet_step<4> : opnet_step<8>;
}

// ------------------------------------------------------------------------------------------------
// shapes and buffer carving
// ------------------------------------------------------------------------------------------------
static int check_dims(int B, int T, int H1, int H2)
{
    if (B <= 0 || T <= 0) return fail(OPNET_ESHAPE, "B=%d T=%d must be positive", B, T);
    if (H1 <= 0 || H2 <= 0 || (H1 & 15) || (H2 & 15))
        return fail(OPNET_ESHAPE, "hidden sizes must be positive multiples of 16 (H1=%d H2=%d)", H1, H2);
    return OPNET_OK;
}

extern "C" size_t opnet_packed_weights_bytes(int H1, int H2)
{
    if (check_dims(1, 1, H1, H2)) return 0;
    return packed_layout(H1, H2).total * sizeof(float);
}

extern "C" size_t opnet_workspace_bytes(int B, int T, int H1, int H2)
{
    if (check_dims(B, T, H1, H2)) return 0;
    return workspace_layout(B, T, H1, H2).total;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
extern "C" int opnet_pack_weights_f32(const float *w_ih1, const float *w_hh1, const float *w_sel,
                                      const float *w_ih2, const float *w_hh2, const float *w_out,
                                      float *packed, size_t packed_bytes, int H1, int H2, void *stream)
{
    if (int rc = check_dims(1, 1, H1, H2)) return rc;
    if (!w_ih1 || !w_hh1 || !w_sel || !w_ih2 || !w_hh2 || !w_out || !packed)
        return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed)) return fail(OPNET_EINVAL, "packed must be 16-byte aligned");
    const PackedLayout L = packed_layout(H1, H2);
    if (packed_bytes < L.total * sizeof(float))
        return fail(OPNET_EWORKSPACE, "packed buffer %zu B < %zu B", packed_bytes, L.total * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    const int n1 = H1 / 4, n2 = H2 / 4;
    auto blocks = [](size_t n) { return (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256); };
    // LSTM1: K = [x 90 -> 96 | h H1]
    opnet_pack_tiles<<<blocks((size_t)n1 * ((96 + H1) / 16) * 256), 256, 0, st>>>(
        packed + L.w1p, w_ih1, w_hh1, OPNET_KX, OPNET_KXQ * 4, H1, H1, 0, 0, n1);
    // LSTM2: K = [h H2]; its 6-wide x part is applied in the epilogue
    opnet_pack_tiles<<<blocks((size_t)n2 * (H2 / 16) * 256), 256, 0, st>>>(
        packed + L.w2p, nullptr, w_hh2, 0, 0, H2, H2, 0, 0, n2);
    opnet_pack_wih2<<<(H2 * 32 + 255) / 256, 256, 0, st>>>(packed + L.wih2p, w_ih2, H2);
    opnet_pack_tiles<<<blocks((size_t)(H1 / 16) * 256), 256, 0, st>>>(
        packed + L.wselp, nullptr, w_sel, 0, 0, H1, 0, OPNET_SLOTS, 1, 1);
    opnet_pack_tiles<<<blocks((size_t)(H2 / 16) * 256), 256, 0, st>>>(
        packed + L.woutp, nullptr, w_out, 0, 0, H2, 0, 4, 1, 1);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
static int make_args(StepArgs *a, OpnetIO *io, const float *boxes, const float *packed, float *y,
                     float *logits, void *ws, size_t ws_bytes, int B, int T, int H1, int H2)
{
    if (int rc = check_dims(B, T, H1, H2)) return rc;
    if (!boxes || !packed || !y || !logits || !ws) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed) || !aligned16(y) || !aligned16(ws) || (((uintptr_t)boxes) & 7u))
        return fail(OPNET_EINVAL, "packed/y/workspace must be 16-byte and boxes 8-byte aligned");
    const WorkspaceLayout W = workspace_layout(B, T, H1, H2);
    if (ws_bytes < W.total) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", ws_bytes, W.total);
    char *w = (char *)ws;
    step_args_inference(a, w, packed, B, T, H1, H2);
    memset(io, 0, sizeof(*io));
    io->B = B; io->T = T; io->RB = a->RB;
    io->boxes = boxes; io->y = y; io->logits = logits;
    io->xp = (float4 *)(w + W.xp);
    io->ystage = a->ystage;
    io->lgstage = a->lgstage;
    io->state = (float4 *)(w + W.state);
    io->state_f4 = (long)((W.state_end - W.state) / 16);
    return OPNET_OK;
}

// grid.y: up to OPNET_MAX_GY row blocks side by side (about 3 workgroups a CU at H1=256/H2=512);
// beyond that a workgroup walks its row blocks with the weights held in registers
#define OPNET_MAX_GY 4
static dim3 step_grid(const StepArgs &a)
{
    // the wide form has half the workgroups per row block, so it spreads twice as many row blocks side by side
    const bool wide = step_is_wide(a);
    const int gy = env_int("OPNET_MAX_GY", wide ? 2 * OPNET_MAX_GY : OPNET_MAX_GY);
    const int tiles = wide ? a.H2 / 8 + a.H1 / 8 : a.H2 / 4 + a.H1 / 4;
    // grid.x is rounded up to a multiple of 8: workgroups are dealt round-robin over the 8 XCDs in linear order, so with
    // a multiple of 8 per grid row the workgroups (bx, 0), (bx, 1), ... of one tile land on the SAME XCD and the second to
    // fourth row block's fetch of the tile's weights hits that XCD's L2 instead of crossing the fabric again
    const int gx = env_int("OPNET_XCD_ALIGN", 1) ? (tiles + 2 + 7) / 8 * 8 : tiles + 2;
    return dim3(gx, a.RB < gy ? a.RB : gy, 1);
}
static dim3 copy_grid(int B, int T)
{
    const long n = (long)B * T * OPNET_SLOTS;
    return dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), 1, 1);
}

extern "C" int opnet_forward_f32(const float *boxes, const float *packed, float *y, float *logits,
                                 void *workspace, size_t workspace_bytes, int B, int T, int H1, int H2,
                                 void *stream)
{
    StepArgs a;
    OpnetIO io;
    if (int rc = make_args(&a, &io, boxes, packed, y, logits, workspace, workspace_bytes, B, T, H1, H2))
        return rc;
    const WorkspaceLayout W = workspace_layout(B, T, H1, H2);
    hipStream_t st = (hipStream_t)stream;
    OpnetIO *dio = (OpnetIO *)((char *)workspace + W.io);
    opnet_set_io<<<1, 1, 0, st>>>(dio, io);
    opnet_pack_input<<<dim3(T, a.RB), 256, 0, st>>>(dio);
    const dim3 grid = step_grid(a);
    if (step_is_preload(a)) {
        for (int s = 0; s < T + 3; ++s)
            opnet_step_pl<4, 8><<<grid, 8 * 64, 0, st>>>((char *)workspace, packed, B, T, H1, H2, s);
    } else {
        const opnet_step_fn stepk = step_kernel(a);
        for (int s = 0; s < T + 3; ++s) stepk<<<grid, step_threads(a), 0, st>>>(a, s);
    }
    opnet_copy_out<<<copy_grid(B, T), 256, 0, st>>>(dio);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// ------------------------------------------------------------------------------------------------
// forward, per-XCD persistent form (opnet_xcd_kernels.hip): ONE launch, weights resident in registers
// ------------------------------------------------------------------------------------------------
struct XcdWorkspaceLayout {  // offsets in bytes
    size_t status, flags, xp, h1h, h2h, fbh, yp, total;
    int NGT, ring, ho;
};

// OPNET_XCD_RING (default 1): h1 / h2 / frames_boxes as rings of XCD_RING steps and the output head inside the launch (head-once
// form: 32 more MFMAs per product wave of one CU per phase; otherwise per-CU partials, 8 KB per step and group, summed by
// opnet_xcd_y_reduce) instead of full histories (48.25 KB per step and group) that a second kernel reads back (DESIGN.md
// section 5a); 0 = round 2's layout.
static int xcd_ring_mode() { return env_int("OPNET_XCD_RING", 1) != 0; }
// three or more groups on the fullest XCD (they set the launch's duration): the "head once" form (the selection head on
// one wave per XCD and phase, LSTM2 one more step behind); fewer: every CU computes the head (the exchange is on the
// critical path there and a lone head wave lengthens it: 88 k against 106 k clips/s at 256 clips, 68 k against 75 k at
// 128; at 320 clips - XCDs with 3 and with 2 groups - 106 k against 93 k the other way).  OPNET_XCD_HO = 0 / 1 overrides.
static int xcd_head_once(int NGT) { return env_int("OPNET_XCD_HO", (NGT + XCD_COUNT - 1) / XCD_COUNT >= 3 ? 1 : 0) != 0; }

static XcdWorkspaceLayout xcd_workspace_layout(int B, int T)
{
    XcdWorkspaceLayout L;
    const size_t NGT = (size_t)(B + 15) / 16;
    size_t o = 0;
    L.NGT = (int)NGT;
    L.ring = xcd_ring_mode();
    L.ho = xcd_head_once(L.NGT);
    const size_t NS = L.ring ? (size_t)XCD_RING : (size_t)(T + 1);
    L.status = o; o += 2048;                                   // 8 control words + 256 XCC ids
    L.flags = o;  o += align_up(NGT * XCD_CUS * 4, 256);
    L.xp = o;     o += NGT * (size_t)(T + 2) * OPNET_KXQ * 256;
    L.h1h = o;    o += NGT * NS * (XCD_H1 / 4) * 256;
    L.h2h = o;    o += NGT * NS * (XCD_H2 / 4) * 256;
    L.fbh = o;    o += NGT * NS * 1024;
    L.yp = o;     if (L.ring && !L.ho) o += NGT * (size_t)T * XCD_CUS * 256;
    L.total = align_up(o, 256);
    return L;
}

// largest batch one launch carries: XCD_NGMAX groups of 16 clips on each of the 8 XCDs
#define XCD_MAX_B (XCD_COUNT * XCD_NGMAX * 16)

static int check_xcd(int B, int T, int H1, int H2)
{
    if (B <= 0 || T <= 0) return fail(OPNET_ESHAPE, "B=%d T=%d must be positive", B, T);
    if (H1 != XCD_H1 || H2 != XCD_H2)
        return fail(OPNET_ESHAPE, "the per-XCD persistent forward is built for H1=%d, H2=%d (got %d, %d); use opnet_forward_f32",
                    XCD_H1, XCD_H2, H1, H2);
    if (B > XCD_MAX_B) return fail(OPNET_ESHAPE, "B=%d > %d clips per launch: split the batch", B, XCD_MAX_B);
    const XcdWorkspaceLayout L = xcd_workspace_layout(B, T);
    if (L.total >= ((size_t)1 << 31))
        return fail(OPNET_ESHAPE, "B=%d x T=%d: the workspace exceeds the 2 GiB one buffer descriptor addresses; split the batch", B, T);
    return OPNET_OK;
}

extern "C" int opnet_xcd_max_batch(void) { return XCD_MAX_B; }

extern "C" size_t opnet_xcd_workspace_bytes(int B, int T, int H1, int H2)
{
    if (check_xcd(B, T, H1, H2)) return 0;
    return xcd_workspace_layout(B, T).total;
}

// tools: in-kernel timeline of block 0 (device buffer of >= (T+1) * groups-per-XCD * 4 u64), null = off
static unsigned long long *g_xcd_trace = nullptr;
extern "C" void opnet_xcd_set_trace(void *device_buffer) { g_xcd_trace = (unsigned long long *)device_buffer; }

// Two persistent launches must never be co-resident (each needs every CU of its XCDs: two half-resident grids would
// wait for each other until the spin limit), so launches are chained per device through an event whatever streams
// the callers use.
static std::mutex g_xcd_mu;
static hipEvent_t g_xcd_done[64] = {};
// side streams of the sliced reverse recurrence (opnet_train_backward_f32), per device, created on first use
#define BWD_MAX_SLICES 4
struct BwdSide {
    hipStream_t s[BWD_MAX_SLICES - 1];
    hipEvent_t fork, join[BWD_MAX_SLICES - 1];
};
static BwdSide g_bwd_side[64] = {};
static int g_xcd_cus[64] = {};

// Measurement (bench.py): with profiling on, every launch of a profiled kernel (tag 0: opnet_xcd_forward, 1: seqx_forward,
// 2: the attention kernel(s) of one attention call, 3: seqt_forward) is bracketed by a pair of HIP events on the caller's stream;
// opnet_kernel_profile_read waits for them and returns the summed kernel time of a tag.
#define PROF_XCD 0
#define PROF_SEQX 1
#define PROF_ATTN 2
#define PROF_SEQT 3
#define PROF_FFN 4          // the fused feed-forward kernel of an encoder layer (csrc/ffn_kernels.hip)
#define PROF_ATTN_TF 5      // training attention, forward: attention_train_fwd (+ merge) of one layer call (csrc/attn_train_kernels.hip)
#define PROF_ATTN_TB 6      // training attention, backward: prep + the dQ pass + the dK / dV pass (+ reduces) of one layer call
#define PROF_SEQXB 7        // seqx_backward: the stacked LSTM's reverse recurrence as one launch (csrc/seq_xcd_bwd_kernels.hip)
#define PROF_TAGS 8
typedef std::pair<hipEvent_t, hipEvent_t> ProfPair;
static bool g_xcd_prof = false;
static std::vector<ProfPair> g_prof_ev[PROF_TAGS];
static std::vector<ProfPair> g_prof_pool;
static std::mutex g_prof_mu;

extern "C" int opnet_xcd_profile(int enable)
{
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_xcd_prof = enable != 0;
    for (auto &v : g_prof_ev) {
        for (auto &e : v) g_prof_pool.push_back(e);
        v.clear();
    }
    return OPNET_OK;
}

extern "C" int opnet_kernel_profile_read(int tag, double *kernel_ms_total, int *launches)
{
    if (!kernel_ms_total || !launches) return fail(OPNET_EINVAL, "null pointer");
    if (tag < 0 || tag >= PROF_TAGS) return fail(OPNET_EINVAL, "profile tag %d out of range", tag);
    std::lock_guard<std::mutex> lock(g_prof_mu);
    double total = 0.0;
    for (auto &e : g_prof_ev[tag]) {
        HIP_TRY(hipEventSynchronize(e.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
        total += ms;
    }
    *kernel_ms_total = total;
    *launches = (int)g_prof_ev[tag].size();
    for (auto &e : g_prof_ev[tag]) g_prof_pool.push_back(e);
    g_prof_ev[tag].clear();
    return OPNET_OK;
}
extern "C" int opnet_xcd_profile_read(double *kernel_ms_total, int *launches)
{
    return opnet_kernel_profile_read(PROF_XCD, kernel_ms_total, launches);
}

// begin / end of one profiled launch (no-ops unless profiling is on)
static bool prof_begin(hipStream_t st, ProfPair *pe)
{
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_xcd_prof) return false;
    if (!g_prof_pool.empty()) { *pe = g_prof_pool.back(); g_prof_pool.pop_back(); }
    else if (hipEventCreate(&pe->first) != hipSuccess || hipEventCreate(&pe->second) != hipSuccess) return false;
    return hipEventRecord(pe->first, st) == hipSuccess;
}
static void prof_end(int tag, hipStream_t st, const ProfPair &pe)
{
    (void)hipEventRecord(pe.second, st);
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof_ev[tag].push_back(pe);
}

static int xcd_device_cus(int dev)
{
    if (dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(g_xcd_mu);
    if (!g_xcd_cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        g_xcd_cus[dev] = prop.multiProcessorCount;
    }
    return g_xcd_cus[dev];
}

// 1 when the persistent form can run on the CURRENT device for these sizes (reference hidden sizes, all 8 XCDs x 32 CUs
// visible - not a compute partition), else 0: callers then use opnet_forward_f32 / opnet_plan_forward
extern "C" int opnet_xcd_supported(int H1, int H2)
{
    if (H1 != XCD_H1 || H2 != XCD_H2) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    return xcd_device_cus(dev) >= XCD_COUNT * XCD_CUS ? 1 : 0;
}

static int xcd_forward_impl(const XcdSources &src, const float *packed, float *y, float *logits,
                            void *workspace, size_t workspace_bytes, int B, int T, int H1, int H2,
                            void *stream)
{
    if (int rc = check_xcd(B, T, H1, H2)) return rc;
    if (!packed || !y || !logits || !workspace) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed) || !aligned16(y) || !aligned16(workspace))
        return fail(OPNET_EINVAL, "packed/y/workspace must be 16-byte aligned");
    const XcdWorkspaceLayout L = xcd_workspace_layout(B, T);
    if (workspace_bytes < L.total) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", workspace_bytes, L.total);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(OPNET_EINVAL, "device index %d out of range", dev);
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)workspace;
    XcdArgs a;
    a.B = B; a.T = T; a.NGT = L.NGT;
    a.packed = packed;
    a.xp = (const float4 *)(w + L.xp);
    a.h1h = (float4 *)(w + L.h1h);
    a.h2h = (float4 *)(w + L.h2h);
    a.fbh = (float4 *)(w + L.fbh);
    a.flags = (unsigned *)(w + L.flags);
    a.status = (unsigned *)(w + L.status);
    a.logits = logits;
    a.ws = w;
    a.xp_off = (unsigned)L.xp; a.h1_off = (unsigned)L.h1h; a.h2_off = (unsigned)L.h2h; a.fb_off = (unsigned)L.fbh; a.flags_off = (unsigned)L.flags; a.status_off = (unsigned)L.status;
    a.ring = L.ring; a.yp_off = (unsigned)L.yp; a.y = y;
    a.trace = g_xcd_trace;
    a.force_safe = env_int("OPNET_XCD_SAFE", 0);
    a.debug = env_int("OPNET_XCD_DEBUG", 0);
    const int cus = xcd_device_cus(dev);
    if (cus < XCD_COUNT * XCD_CUS)
        return fail(OPNET_ESHAPE, "device %d exposes %d CUs; the persistent forward needs %d resident workgroups", dev,
                    cus, XCD_COUNT * XCD_CUS);
    std::lock_guard<std::mutex> lock(g_xcd_mu);
    opnet_xcd_pack_input<<<dim3(T + 2, L.NGT), 384, 0, st>>>(src, a);
    if (!g_xcd_done[dev]) HIP_TRY(hipEventCreateWithFlags(&g_xcd_done[dev], hipEventDisableTiming));
    else HIP_TRY(hipStreamWaitEvent(st, g_xcd_done[dev], 0));
    ProfPair pe{};
    const bool prof = prof_begin(st, &pe);
    const int ho = L.ho;                    // xcd_head_once
    if (ho) opnet_xcd_forward<true><<<XCD_COUNT * XCD_CUS, 512, 0, st>>>(a);
    else opnet_xcd_forward<false><<<XCD_COUNT * XCD_CUS, 512, 0, st>>>(a);
    if (prof) prof_end(PROF_XCD, st, pe);
    HIP_TRY(hipEventRecord(g_xcd_done[dev], st));
    if (a.ring && ho) {
        opnet_xcd_y_poison<<<64, 256, 0, st>>>(a, y);      // y left the launch complete; NaN only if the launch gave up
    } else if (a.ring) {
        const long ny = (long)L.NGT * T * 16;
        opnet_xcd_y_reduce<<<(unsigned)((ny + 255) / 256 > 2048 ? 2048 : (ny + 255) / 256), 256, 0, st>>>(a, y);
    } else {
        opnet_xcd_out_head<<<dim3(T, L.NGT), 256, 0, st>>>(a, y);
    }
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opnet_xcd_forward_f32(const float *boxes, const float *packed, float *y, float *logits,
                                     void *workspace, size_t workspace_bytes, int B, int T, int H1, int H2,
                                     void *stream)
{
    if (!boxes) return fail(OPNET_EINVAL, "null pointer");
    if (((uintptr_t)boxes) & 7u) return fail(OPNET_EINVAL, "boxes must be 8-byte aligned");
    XcdSources src;
    memset(&src, 0, sizeof(src));
    src.p[0] = boxes; src.start[0] = 0; src.start[1] = B; src.n = 1;
    return xcd_forward_impl(src, packed, y, logits, workspace, workspace_bytes, B, T, H1, H2, stream);
}

/* The same launch over `nreq` (<= 64) request tensors boxes[r] = [counts[r]][T][90] (device pointers in a HOST array): the
 * launch's clips are the requests' clips in order, y / logits are [sum counts][...] - a server's pending requests without a
 * concatenation copy (serving.ReasonerServer). */
extern "C" int opnet_xcd_forward_multi_f32(const float *const *boxes, const int *counts, int nreq, const float *packed, float *y,
                                           float *logits, void *workspace, size_t workspace_bytes, int T, int H1, int H2,
                                           void *stream)
{
    if (!boxes || !counts) return fail(OPNET_EINVAL, "null pointer");
    if (nreq < 1 || nreq > XCD_MAX_SOURCES) return fail(OPNET_ESHAPE, "1..%d requests per launch (nreq=%d)", XCD_MAX_SOURCES, nreq);
    XcdSources src;
    memset(&src, 0, sizeof(src));
    int B = 0;
    for (int r = 0; r < nreq; ++r) {
        if (!boxes[r]) return fail(OPNET_EINVAL, "null pointer (request %d)", r);
        if (((uintptr_t)boxes[r]) & 7u) return fail(OPNET_EINVAL, "boxes must be 8-byte aligned (request %d)", r);
        if (counts[r] <= 0) return fail(OPNET_ESHAPE, "request %d: %d clips", r, counts[r]);
        src.p[r] = boxes[r];
        src.start[r] = B;
        B += counts[r];
    }
    src.start[nreq] = B;
    src.n = nreq;
    return xcd_forward_impl(src, packed, y, logits, workspace, workspace_bytes, B, T, H1, H2, stream);
}

// OPNetLstmMlp (learned_models.py:55-89): same packed layout; w_hidden [H2][6] takes the place of the
// video LSTM's input weights (gate-0 slot), the recurrent tiles stay unused.
__global__ void opnet_pack_hidden(float *__restrict__ out, const float *__restrict__ w_hidden, int H2)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H2 * 4 * 8) return;
    const int f = idx & 7, gate = (idx >> 3) & 3, unit = idx >> 5;
    out[idx] = (gate == 0 && f < OPNET_FEATS_) ? w_hidden[(long)unit * OPNET_FEATS_ + f] : 0.f;
}

extern "C" int opnet_mlp_pack_weights_f32(const float *w_ih1, const float *w_hh1, const float *w_sel,
                                          const float *w_hidden, const float *w_out, float *packed,
                                          size_t packed_bytes, int H1, int H2, void *stream)
{
    if (int rc = check_dims(1, 1, H1, H2)) return rc;
    if (!w_ih1 || !w_hh1 || !w_sel || !w_hidden || !w_out || !packed) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed)) return fail(OPNET_EINVAL, "packed must be 16-byte aligned");
    const PackedLayout L = packed_layout(H1, H2);
    if (packed_bytes < L.total * sizeof(float)) return fail(OPNET_EWORKSPACE, "packed buffer too small");
    hipStream_t st = (hipStream_t)stream;
    auto blocks = [](size_t n) { return (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256); };
    opnet_pack_tiles<<<blocks((size_t)(H1 / 4) * ((96 + H1) / 16) * 256), 256, 0, st>>>(
        packed + L.w1p, w_ih1, w_hh1, OPNET_KX, OPNET_KXQ * 4, H1, H1, 0, 0, H1 / 4);
    opnet_pack_hidden<<<(H2 * 32 + 255) / 256, 256, 0, st>>>(packed + L.wih2p, w_hidden, H2);
    opnet_pack_tiles<<<blocks((size_t)(H1 / 16) * 256), 256, 0, st>>>(packed + L.wselp, nullptr, w_sel, 0, 0, H1, 0, OPNET_SLOTS, 1, 1);
    opnet_pack_tiles<<<blocks((size_t)(H2 / 16) * 256), 256, 0, st>>>(packed + L.woutp, nullptr, w_out, 0, 0, H2, 0, 4, 1, 1);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opnet_mlp_forward_f32(const float *boxes, const float *packed, float *y, float *logits,
                                     void *workspace, size_t workspace_bytes, int B, int T, int H1, int H2,
                                     void *stream)
{
    StepArgs a;
    OpnetIO io;
    if (int rc = make_args(&a, &io, boxes, packed, y, logits, workspace, workspace_bytes, B, T, H1, H2))
        return rc;
    a.mlp = 1;
    const WorkspaceLayout W = workspace_layout(B, T, H1, H2);
    hipStream_t st = (hipStream_t)stream;
    OpnetIO *dio = (OpnetIO *)((char *)workspace + W.io);
    opnet_set_io<<<1, 1, 0, st>>>(dio, io);
    opnet_pack_input<<<dim3(T, a.RB), 256, 0, st>>>(dio);
    const dim3 grid = step_grid(a);
    const opnet_step_fn stepk = step_kernel(a);
    for (int s = 0; s < T + 3; ++s) stepk<<<grid, step_threads(a), 0, st>>>(a, s);
    opnet_copy_out<<<copy_grid(B, T), 256, 0, st>>>(dio);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// ------------------------------------------------------------------------------------------------
// graph plan
// ------------------------------------------------------------------------------------------------
struct opnet_plan {
    int B, T, H1, H2;
    void *ws;             // workspace and packed weights the graph was built for
    const float *packed;
    hipGraph_t graph;
    hipGraphExec_t exec;
};

extern "C" int opnet_plan_create(opnet_plan **plan, int B, int T, int H1, int H2)
{
    if (!plan) return fail(OPNET_EINVAL, "null plan pointer");
    if (int rc = check_dims(B, T, H1, H2)) return rc;
    opnet_plan *p = new (std::nothrow) opnet_plan();
    if (!p) return fail(OPNET_EINVAL, "out of host memory");
    p->B = B; p->T = T; p->H1 = H1; p->H2 = H2;
    p->ws = nullptr; p->packed = nullptr; p->graph = nullptr; p->exec = nullptr;
    *plan = p;
    return OPNET_OK;
}

static void plan_drop_graph(opnet_plan *p)
{
    if (p->exec) { (void)hipGraphExecDestroy(p->exec); p->exec = nullptr; }
    if (p->graph) { (void)hipGraphDestroy(p->graph); p->graph = nullptr; }
    p->ws = nullptr;
    p->packed = nullptr;
}

extern "C" void opnet_plan_destroy(opnet_plan *p)
{
    if (!p) return;
    plan_drop_graph(p);
    delete p;
}

// pack_input (+ state zeroing) -> step 0 -> ... -> step T+2 -> copy_out, one linear dependency chain
static int plan_build(opnet_plan *p, const StepArgs &a, void *ws)
{
    plan_drop_graph(p);
    const WorkspaceLayout W = workspace_layout(p->B, p->T, p->H1, p->H2);
    OpnetIO *dio = (OpnetIO *)((char *)ws + W.io);
    HIP_TRY(hipGraphCreate(&p->graph, 0));
    hipGraphNode_t prev = nullptr, node = nullptr;

    auto add_io_kernel = [&](void *func, dim3 grid) -> hipError_t {
        void *args[] = {(void *)&dio};
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        kp.func = func;
        kp.gridDim = grid;
        kp.blockDim = dim3(256, 1, 1);
        kp.kernelParams = args;
        hipError_t e = hipGraphAddKernelNode(&node, p->graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp);
        prev = node;
        return e;
    };
    HIP_TRY(add_io_kernel((void *)opnet_pack_input, dim3(p->T, a.RB, 1)));
    for (int s = 0; s < p->T + 3; ++s) {
        StepArgs av = a;
        int step = s;
        void *args[] = {(void *)&av, (void *)&step};
        char *wsb = (char *)ws;
        const float *pk = (const float *)a.w1p - packed_layout(p->H1, p->H2).w1p;
        int sB = p->B, sT = p->T, sH1 = p->H1, sH2 = p->H2;
        void *args_pl[] = {(void *)&wsb, (void *)&pk, (void *)&sB, (void *)&sT, (void *)&sH1, (void *)&sH2, (void *)&step};
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        const bool pl = step_is_preload(a);
        kp.func = pl ? (void *)opnet_step_pl<4, 8> : (void *)step_kernel(a);
        kp.gridDim = step_grid(a);
        kp.blockDim = dim3(step_threads(a), 1, 1);
        kp.kernelParams = pl ? args_pl : args;
        HIP_TRY(hipGraphAddKernelNode(&node, p->graph, &prev, 1, &kp));
        prev = node;
    }
    HIP_TRY(add_io_kernel((void *)opnet_copy_out, copy_grid(p->B, p->T)));
    HIP_TRY(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
    p->ws = ws;
    p->packed = (const float *)a.w1p - packed_layout(p->H1, p->H2).w1p;
    return OPNET_OK;
}

extern "C" int opnet_plan_forward(opnet_plan *p, const float *boxes, const float *packed, float *y,
                                  float *logits, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!p) return fail(OPNET_EINVAL, "null plan");
    StepArgs a;
    OpnetIO io;
    if (int rc = make_args(&a, &io, boxes, packed, y, logits, workspace, workspace_bytes, p->B, p->T,
                           p->H1, p->H2))
        return rc;
    if (p->ws != workspace || p->packed != packed || !p->exec) {
        if (int rc = plan_build(p, a, workspace)) { plan_drop_graph(p); return rc; }
    }
    const WorkspaceLayout W = workspace_layout(p->B, p->T, p->H1, p->H2);
    hipStream_t st = (hipStream_t)stream;
    opnet_set_io<<<1, 1, 0, st>>>((OpnetIO *)((char *)workspace + W.io), io);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipGraphLaunch(p->exec, st));
    return OPNET_OK;
}

// ------------------------------------------------------------------------------------------------
// training: forward with saved history, backward, loss, optimiser
// ------------------------------------------------------------------------------------------------
struct TrainPackedLayout {  // offsets in floats; the inference layout comes first
    size_t fwd_total, w2bt, w1bt, wih2t, wsel, wout, x4fwd, x4bwd, total;   // x4*: register images of the 4-clip persistent kernels (reference sizes only)
};

static bool x4_dims(int H1, int H2) { return H1 == XCD_H1 && H2 == XCD_H2; }

static TrainPackedLayout train_packed_layout(int H1, int H2)
{
    TrainPackedLayout L;
    size_t o = packed_layout(H1, H2).total;
    L.fwd_total = o;
    o = align_up(o, 4);
    L.w2bt = o;  o += (size_t)(H2 / 16) * (H2 / 4) * 256;
    L.w1bt = o;  o += (size_t)(H1 / 16) * (H1 / 4) * 256;
    L.wih2t = o; o += (size_t)(H2 / 4) * 256;
    L.wsel = o;  o += align_up((size_t)OPNET_SLOTS * H1, 4);
    L.wout = o;  o += align_up((size_t)4 * H2, 4);
    L.x4fwd = o; o += x4_dims(H1, H2) ? x4_packed_layout().total : 0;
    L.x4bwd = o; o += x4_dims(H1, H2) ? x4b_packed_layout().total : 0;
    L.total = o;
    return L;
}

struct TrainWorkspaceLayout {  // offsets in bytes
    size_t io, xp, state, h1all, c1all, h2all, c2all, state_end, x2all, g1, g2, psave, ystage, lgstage,
        dyp, dlall, dhpart2, dhpart1, dx2part, dcz, dc2, dc1, dcz_end, l1part, x4h1x, x4h2x, x4status, x4da1x, x4da2x, x4dfx, wgpart,
        xcdws, total;       // xcdws: workspace of the 16-clip persistent forward when it carries the training forward (section 9f)
};
#define WG2_MAX_WAVES 1024      // one round of the chip's SIMDs (opnet_wgrad_tiles)

// the 4-clip persistent step carries up to X4_NGMAX row blocks; its exchange buffers exist only for such batches
static bool x4_batch(int B, int H1, int H2) { return x4_dims(H1, H2) && (B + 31) / 32 <= X4_NGMAX; }
// batches whose TRAINING forward may run as the 16-clip persistent launch (opnet_xcd_forward<HO, true>): above the 4-clip forward's
// range, one launch's worth of clips
static bool xcdt_batch(int B, int H1, int H2) { return x4_dims(H1, H2) && B > 96 && B <= XCD_MAX_B; }

static TrainWorkspaceLayout train_workspace_layout(int B, int T, int H1, int H2)
{
    const size_t RB = (B + 31) / 32, TT = T;
    TrainWorkspaceLayout L;
    size_t o = 0;
    L.io = o;      o += align_up(sizeof(OpnetIO), 256);
    L.xp = o;      o += TT * RB * OPNET_KXQ * 32 * 16;
    L.state = o;   // zeroed at the start of every forward (slot 0 = initial state; the rest is overwritten)
    L.h1all = o;   o += (TT + 1) * RB * (size_t)H1 * 32 * 4;
    L.c1all = o;   o += (TT + 1) * RB * (size_t)H1 * 32 * 4;
    L.h2all = o;   o += (TT + 1) * RB * (size_t)H2 * 32 * 4;
    L.c2all = o;   o += (TT + 1) * RB * (size_t)H2 * 32 * 4;
    L.state_end = o;
    L.x2all = o;   o += TT * RB * 64 * 16;
    L.g1 = o;      o += TT * RB * (size_t)H1 * 32 * 16;
    L.g2 = o;      o += TT * RB * (size_t)H2 * 32 * 16;
    L.psave = o;   o += TT * RB * 128 * 16;
    L.ystage = o;  o += RB * 32 * TT * 16;
    L.lgstage = o; o += RB * 32 * TT * OPNET_SLOTS * 4;
    L.dyp = o;     o += TT * RB * 32 * 16;
    L.dlall = o;   o += TT * RB * 128 * 16;
    L.dhpart2 = o; o += 4 * RB * (size_t)H2 * 32 * 4;
    L.dhpart1 = o; o += 4 * RB * (size_t)H1 * 32 * 4;
    L.dx2part = o; o += 4 * RB * 16 * 32 * 4;
    L.dcz = o;
    L.dc2 = o;     o += RB * (size_t)H2 * 32 * 4;
    L.dc1 = o;     o += RB * (size_t)H1 * 32 * 4;
    L.dcz_end = o;
    L.l1part = o;  o += 1024 * 4;
    o = align_up(o, 4096);
    const size_t NG = x4_batch(B, H1, H2) ? RB * 8 : 0;     // 4-clip groups; exchange rings of X4_SLOTS steps
    L.x4h1x = o;   o += NG * X4_SLOTS * 4096;
    L.x4h2x = o;   o += NG * X4_SLOTS * 8192;
    L.x4status = o; o += NG ? 2048 : 0;
    L.x4da1x = o;  o += NG * X4_SLOTS * 131072;   // partial dh1: [32 owners][32 producers][128 B]
    L.x4da2x = o;  o += NG * X4_SLOTS * 262144;   // partial dh2: [32 owners][32 producers][256 B]
    L.x4dfx = o;   o += NG * X4_SLOTS * 4096;
    o = align_up(o, 4096);
    L.wgpart = o;  o += (size_t)WG2_MAX_WAVES * WG2_PART_F * 4;     // partial tiles of the weight-gradient waves (64 MB)
    o = align_up(o, 4096);
    L.xcdws = o;   if (xcdt_batch(B, H1, H2)) o += xcd_workspace_layout(B, T).total;
    L.total = align_up(o, 256);
    return L;
}
// the status words of a training step's persistent launches (abort code, block, phase, ...): the 4-clip kernels' where they exist,
// else the 16-clip forward's; (size_t)-1: this batch never runs a persistent launch
static size_t train_status_offset(const TrainWorkspaceLayout &W, int B, int T, int H1, int H2)
{
    if (x4_batch(B, H1, H2)) return W.x4status;
    if (xcdt_batch(B, H1, H2)) return W.xcdws + xcd_workspace_layout(B, T).status;
    return (size_t)-1;
}

extern "C" size_t opnet_train_packed_weights_bytes(int H1, int H2)
{
    if (check_dims(1, 1, H1, H2)) return 0;
    return train_packed_layout(H1, H2).total * sizeof(float);
}

extern "C" size_t opnet_train_workspace_bytes(int B, int T, int H1, int H2)
{
    if (check_dims(B, T, H1, H2)) return 0;
    return train_workspace_layout(B, T, H1, H2).total;
}

// The launch chain's training layouts (the inference layout + W^T tiles + raw heads: 9 small launches) are only read by the
// launch-chain step; a batch that trains on the 4-clip persistent kernels reads the x4 images and the output-head tiles.
// opnet_train_pack_weights_f32 therefore packs those three eagerly and leaves the rest to the first launch-chain call that
// follows (train_chain_layouts) - the caller's weight pointers are remembered per packed buffer until then.
// g_tp holds ONLY buffers whose chain layouts are still owed (an entry is dropped when they are packed, when the buffer is packed
// again, or when it becomes an OPNetLstmMlp image); it is never pruned - a pruned entry would make the next launch-chain step read
// layouts that were never written - and when it is full (entries of modules that died unpacked linger: their addresses are only
// ever looked up again after a new pack has replaced the entry) a new buffer is simply packed in full.
struct TrainPackState { const float *w[6]; int H1, H2; };
static std::mutex g_tp_mu;
static std::map<const float *, TrainPackState> g_tp;
static const size_t kTrainPackPending = 4096;

static int pack_chain_train_layouts(const float *w_ih1, const float *w_hh1, const float *w_sel, const float *w_ih2,
                                    const float *w_hh2, const float *w_out, float *packed, int H1, int H2, hipStream_t st)
{
    const TrainPackedLayout L = train_packed_layout(H1, H2);
    if (int rc = opnet_pack_weights_f32(w_ih1, w_hh1, w_sel, w_ih2, w_hh2, w_out, packed,
                                        L.fwd_total * sizeof(float), H1, H2, st))
        return rc;
    auto blocks = [](size_t n) { return (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256); };
    opnet_pack_tiles_t<<<blocks((size_t)(H2 / 16) * (H2 / 4) * 256), 256, 0, st>>>(packed + L.w2bt, w_hh2, H2, 0, 0, H2 / 16);
    opnet_pack_tiles_t<<<blocks((size_t)(H1 / 16) * (H1 / 4) * 256), 256, 0, st>>>(packed + L.w1bt, w_hh1, H1, 0, 0, H1 / 16);
    opnet_pack_tiles_t<<<blocks((size_t)(H2 / 4) * 256), 256, 0, st>>>(packed + L.wih2t, w_ih2, H2, OPNET_FEATS, 1, 1);
    opnet_copy_f32<<<blocks((size_t)OPNET_SLOTS * H1), 256, 0, st>>>(packed + L.wsel, w_sel, (long)OPNET_SLOTS * H1);
    opnet_copy_f32<<<blocks((size_t)4 * H2), 256, 0, st>>>(packed + L.wout, w_out, (long)4 * H2);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// before a launch-chain training launch: pack what opnet_train_pack_weights_f32 left out
static int train_chain_layouts(const float *packed, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_tp_mu);
    auto it = g_tp.find(packed);
    if (it == g_tp.end()) return OPNET_OK;
    const TrainPackState t = it->second;
    if (int rc = pack_chain_train_layouts(t.w[0], t.w[1], t.w[2], t.w[3], t.w[4], t.w[5], (float *)packed, t.H1, t.H2, st)) return rc;
    g_tp.erase(it);
    return OPNET_OK;
}

// Run-time switch of the 4-clip persistent kernels (host callers fall back to the launch chain after an aborted launch)
static std::atomic<int> g_x4_enabled{1};
extern "C" void opnet_xcd4_enable(int on) { g_x4_enabled.store(on ? 1 : 0); }
extern "C" int opnet_xcd4_enabled(void) { return g_x4_enabled.load(); }
static bool x4_on() { return g_x4_enabled.load() != 0 && env_int("OPNET_XCD4", 1) != 0; }

static bool x4_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    return xcd_device_cus(dev) >= XCD_COUNT * XCD_CUS;
}

extern "C" int opnet_train_pack_weights_f32(const float *w_ih1, const float *w_hh1, const float *w_sel,
                                            const float *w_ih2, const float *w_hh2, const float *w_out,
                                            float *packed, size_t packed_bytes, int H1, int H2, void *stream)
{
    if (int rc = check_dims(1, 1, H1, H2)) return rc;
    if (!w_ih1 || !w_hh1 || !w_sel || !w_ih2 || !w_hh2 || !w_out || !packed) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed)) return fail(OPNET_EINVAL, "packed must be 16-byte aligned");
    const TrainPackedLayout L = train_packed_layout(H1, H2);
    if (packed_bytes < L.total * sizeof(float))
        return fail(OPNET_EWORKSPACE, "packed buffer %zu B < %zu B", packed_bytes, L.total * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    bool lazy = x4_dims(H1, H2) && x4_on() && x4_device();
    if (lazy) {
        std::lock_guard<std::mutex> lock(g_tp_mu);
        if (g_tp.find(packed) == g_tp.end() && g_tp.size() >= kTrainPackPending) lazy = false;      // no room to owe: pack in full
    }
    if (!lazy) {
        if (int rc = pack_chain_train_layouts(w_ih1, w_hh1, w_sel, w_ih2, w_hh2, w_out, packed, H1, H2, st)) return rc;
    }
    if (x4_dims(H1, H2)) {
        // (lazy: the output head's tiles - all opnet_xcd4_out_head reads of the chain's layouts - ride in the same launch)
        const PackedLayout P = packed_layout(H1, H2);
        opnet_xcd4_pack_both<<<2048, 256, 0, st>>>(packed + L.x4fwd, packed + L.x4bwd, w_ih1, w_hh1, w_sel, w_ih2, w_hh2, w_out,
                                                   lazy ? packed + P.woutp : nullptr);
    }
    HIP_TRY(hipGetLastError());
    {
        std::lock_guard<std::mutex> lock(g_tp_mu);
        if (lazy) g_tp[packed] = TrainPackState{{w_ih1, w_hh1, w_sel, w_ih2, w_hh2, w_out}, H1, H2};
        else g_tp.erase(packed);
    }
    return OPNET_OK;
}

static int make_train_args(StepArgs *a, OpnetIO *io, BwdArgs *bw, const float *boxes, const float *packed,
                           float *y, float *logits, void *ws, size_t ws_bytes, int B, int T, int H1, int H2)
{
    if (int rc = check_dims(B, T, H1, H2)) return rc;
    if (!packed || !ws) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed) || !aligned16(ws)) return fail(OPNET_EINVAL, "packed/workspace must be 16-byte aligned");
    const TrainWorkspaceLayout W = train_workspace_layout(B, T, H1, H2);
    if (ws_bytes < W.total) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", ws_bytes, W.total);
    const PackedLayout P = packed_layout(H1, H2);
    const TrainPackedLayout TP = train_packed_layout(H1, H2);
    char *w = (char *)ws;
    const int RB = (B + 31) / 32;
    memset(a, 0, sizeof(*a));
    a->B = B; a->T = T; a->RB = RB; a->H1 = H1; a->H2 = H2; a->train = 1;
    a->xp = (const float4 *)(w + W.xp);
    a->w1p = (const float4 *)(packed + P.w1p);
    a->w2p = (const float4 *)(packed + P.w2p);
    a->wih2p = (const float4 *)(packed + P.wih2p);
    a->wselp = (const float4 *)(packed + P.wselp);
    a->woutp = (const float4 *)(packed + P.woutp);
    a->h1buf = (float4 *)(w + W.h1all);
    a->c1 = (float *)(w + W.c1all);
    a->h2buf = (float4 *)(w + W.h2all);
    a->c2 = (float *)(w + W.c2all);
    a->x2buf = (float4 *)(w + W.x2all);
    a->g1save = (float4 *)(w + W.g1);
    a->g2save = (float4 *)(w + W.g2);
    a->psave = (float4 *)(w + W.psave);
    a->ystage = (float4 *)(w + W.ystage);
    a->lgstage = (float *)(w + W.lgstage);
    memset(io, 0, sizeof(*io));
    io->B = B; io->T = T; io->RB = RB;
    io->boxes = boxes; io->y = y; io->logits = logits;
    io->xp = (float4 *)(w + W.xp);
    io->ystage = a->ystage;
    io->lgstage = a->lgstage;
    io->state = (float4 *)(w + W.state);
    io->state_f4 = (long)((W.state_end - W.state) / 16);
    memset(bw, 0, sizeof(*bw));
    bw->B = B; bw->T = T; bw->RB = RB; bw->H1 = H1; bw->H2 = H2;
    bw->rb0 = 0; bw->rb1 = RB;
    bw->xp = a->xp; bw->h1all = a->h1buf; bw->c1all = a->c1; bw->h2all = a->h2buf; bw->c2all = a->c2;
    bw->x2all = a->x2buf; bw->psave = a->psave; bw->g1 = a->g1save; bw->g2 = a->g2save;
    bw->dyp = (const float4 *)(w + W.dyp);
    bw->dlall = (float4 *)(w + W.dlall);
    bw->dhpart2 = (float *)(w + W.dhpart2);
    bw->dhpart1 = (float *)(w + W.dhpart1);
    bw->dx2part = (float *)(w + W.dx2part);
    bw->dc2 = (float *)(w + W.dc2);
    bw->dc1 = (float *)(w + W.dc1);
    bw->w2bt = (const float4 *)(packed + TP.w2bt);
    bw->w1bt = (const float4 *)(packed + TP.w1bt);
    bw->wih2t = (const float4 *)(packed + TP.wih2t);
    bw->wsel = packed + TP.wsel;
    bw->wout = packed + TP.wout;
    return OPNET_OK;
}

// tools: in-kernel timeline of block 0 of the 4-clip persistent kernels (device buffer of >= (T + 2) * row blocks * 8 u64)
static unsigned long long *g_x4_trace = nullptr;
extern "C" void opnet_xcd4_set_trace(void *device_buffer) { g_x4_trace = (unsigned long long *)device_buffer; }
// tools / tests: the status words of the most recent 4-clip persistent launch of this process (synchronises the device):
// [0] abort code (0 = ok), [1] first failing block, [2] phase, [3] groups that were not XCD-local (write-through protocol)
static unsigned *g_x4_last_status = nullptr;
extern "C" int opnet_xcd4_last_status(unsigned *out4)
{
    if (!out4) return fail(OPNET_EINVAL, "null pointer");
    if (!g_x4_last_status) { out4[0] = out4[1] = out4[2] = out4[3] = 0u; return OPNET_OK; }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out4, g_x4_last_status, 4 * sizeof(unsigned), hipMemcpyDeviceToHost));
    return OPNET_OK;
}

// ------------------------------------------------------------------------------------------------
// inference on the 4-clip persistent forward (one small request: the reference's inference batch_size is 16,
// configs/inference_config.json:2): opnet_xcd4_forward<false> = the training forward without the backward's histories
// ------------------------------------------------------------------------------------------------
struct X4InferPacked { size_t x4fwd, woutp, total; };      // floats
static X4InferPacked x4_infer_packed_layout()
{
    X4InferPacked L;
    L.x4fwd = 0;
    L.woutp = x4_packed_layout().total;
    L.total = L.woutp + (size_t)(XCD_H2 / 16) * 256;
    return L;
}
struct X4InferLayout { size_t io, xp, h2all, ystage, lgstage, h1x, h2x, status, total; };   // bytes
static X4InferLayout x4_infer_layout(int B, int T)
{
    const size_t RB = (B + 31) / 32, TT = T, NG = RB * 8;
    X4InferLayout L;
    size_t o = 0;
    L.io = o;      o += align_up(sizeof(OpnetIO), 256);
    L.xp = o;      o += TT * RB * OPNET_KXQ * 32 * 16;
    L.h2all = o;   o += (TT + 1) * RB * (size_t)XCD_H2 * 32 * 4;
    L.ystage = o;  o += RB * 32 * TT * 16;
    L.lgstage = o; o += RB * 32 * TT * OPNET_SLOTS * 4;
    o = align_up(o, 4096);
    L.h1x = o;     o += NG * X4_SLOTS * 4096;
    L.h2x = o;     o += NG * X4_SLOTS * 8192;
    L.status = o;  o += 2048;
    L.total = align_up(o, 256);
    return L;
}
static int check_x4_infer(int B, int T, int H1, int H2)
{
    if (B <= 0 || T <= 0) return fail(OPNET_ESHAPE, "B=%d T=%d must be positive", B, T);
    if (!x4_dims(H1, H2))
        return fail(OPNET_ESHAPE, "the 4-clip persistent forward is built for H1=%d, H2=%d (got %d, %d); use opnet_forward_f32",
                    XCD_H1, XCD_H2, H1, H2);
    if ((B + 31) / 32 > X4_NGMAX) return fail(OPNET_ESHAPE, "B=%d > %d clips per launch: use opnet_xcd_forward_f32", B, 32 * X4_NGMAX);
    if (x4_infer_layout(B, T).total >= ((size_t)1 << 31))
        return fail(OPNET_ESHAPE, "B=%d x T=%d: the workspace exceeds the 2 GiB one buffer descriptor addresses", B, T);
    return OPNET_OK;
}
extern "C" int opnet_xcd4_max_batch(void) { return 32 * X4_NGMAX; }
/* Byte offset of the 4 status words {abort code, block, phase, groups on the write-through path} of a persistent launch
 * inside its workspace, for callers that mirror them to the host behind the launch; (size_t)-1: this shape never runs a
 * persistent kernel.  (opnet_xcd_forward_f32: offset 0 of its workspace.) */
extern "C" size_t opnet_xcd4_status_offset(int B, int T, int H1, int H2)
{
    if (check_x4_infer(B, T, H1, H2)) return (size_t)-1;
    return x4_infer_layout(B, T).status;
}
extern "C" size_t opnet_train_status_offset(int B, int T, int H1, int H2)
{
    if (check_dims(B, T, H1, H2)) return (size_t)-1;
    return train_status_offset(train_workspace_layout(B, T, H1, H2), B, T, H1, H2);
}
extern "C" size_t opnet_xcd4_packed_weights_bytes(int H1, int H2)
{
    return x4_dims(H1, H2) ? x4_infer_packed_layout().total * sizeof(float) : 0;
}
extern "C" size_t opnet_xcd4_workspace_bytes(int B, int T, int H1, int H2)
{
    if (check_x4_infer(B, T, H1, H2)) return 0;
    return x4_infer_layout(B, T).total;
}
extern "C" int opnet_xcd4_pack_weights_f32(const float *w_ih1, const float *w_hh1, const float *w_sel, const float *w_ih2,
                                           const float *w_hh2, const float *w_out, float *packed, size_t packed_bytes,
                                           int H1, int H2, void *stream)
{
    if (!x4_dims(H1, H2)) return fail(OPNET_ESHAPE, "the 4-clip persistent forward is built for H1=%d, H2=%d", XCD_H1, XCD_H2);
    if (!w_ih1 || !w_hh1 || !w_sel || !w_ih2 || !w_hh2 || !w_out || !packed) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed)) return fail(OPNET_EINVAL, "packed must be 16-byte aligned");
    const X4InferPacked L = x4_infer_packed_layout();
    if (packed_bytes < L.total * sizeof(float)) return fail(OPNET_EWORKSPACE, "packed buffer %zu B < %zu B", packed_bytes, L.total * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    opnet_xcd4_pack_fwd<<<1024, 256, 0, st>>>(packed + L.x4fwd, w_ih1, w_hh1, w_sel, w_ih2, w_hh2);
    opnet_pack_tiles<<<(unsigned)(((size_t)(H2 / 16) * 256 + 255) / 256), 256, 0, st>>>(packed + L.woutp, nullptr, w_out, 0, 0, H2, 0, 4, 1, 1);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

/* y [B][T][4], logits [B][15][T] for B <= opnet_xcd4_max_batch() clips as ONE persistent launch on a whole MI355X
 * (opnet_xcd_supported); packed: opnet_xcd4_pack_weights_f32 image */
extern "C" int opnet_xcd4_forward_f32(const float *boxes, const float *packed, float *y, float *logits, void *workspace,
                                      size_t workspace_bytes, int B, int T, int H1, int H2, void *stream)
{
    if (int rc = check_x4_infer(B, T, H1, H2)) return rc;
    if (!boxes || !packed || !y || !logits || !workspace) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed) || !aligned16(y) || !aligned16(workspace) || (((uintptr_t)boxes) & 7u))
        return fail(OPNET_EINVAL, "packed/y/workspace must be 16-byte and boxes 8-byte aligned");
    const X4InferLayout L = x4_infer_layout(B, T);
    if (workspace_bytes < L.total) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", workspace_bytes, L.total);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (xcd_device_cus(dev) < XCD_COUNT * XCD_CUS)
        return fail(OPNET_ESHAPE, "device %d exposes %d CUs; the persistent forward needs %d resident workgroups", dev,
                    xcd_device_cus(dev), XCD_COUNT * XCD_CUS);
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)workspace;
    const X4InferPacked PK = x4_infer_packed_layout();
    const int RB = (B + 31) / 32;
    OpnetIO io;
    memset(&io, 0, sizeof(io));
    io.B = B; io.T = T; io.RB = RB;
    io.boxes = boxes; io.y = y; io.logits = logits;
    io.xp = (float4 *)(w + L.xp);
    io.ystage = (const float4 *)(w + L.ystage);
    io.lgstage = (const float *)(w + L.lgstage);
    io.state = (float4 *)(w + L.h2all);
    io.state_f4 = 0;                                    // nothing to zero: the initial state lives in the exchange rings
    Xcd4Args x;
    memset(&x, 0, sizeof(x));
    x.B = B; x.T = T; x.RB = RB;
    x.pk = packed + PK.x4fwd;
    x.woutp = packed + PK.woutp;
    x.ws = w;
    x.xp_off = (unsigned)L.xp;
    x.h2_off = (unsigned)L.h2all;
    x.lg_off = (unsigned)L.lgstage; x.ys_off = (unsigned)L.ystage;
    x.h1x_off = (unsigned)L.h1x; x.h2x_off = (unsigned)L.h2x;
    x.status = (unsigned *)(w + L.status);
    x.force_safe = env_int("OPNET_XCD_SAFE", 0);
    x.debug = env_int("OPNET_X4_DEBUG", 0);           // tools / the abort test only
    x.trace = g_x4_trace;
    g_x4_last_status = x.status;
    OpnetIO *dio = (OpnetIO *)(w + L.io);
    opnet_set_io<<<1, 1, 0, st>>>(dio, io);
    opnet_pack_input<<<dim3(T, RB), 256, 0, st>>>(dio);
    opnet_xcd4_init<<<8, 256, 0, st>>>(x);
    std::lock_guard<std::mutex> lock(g_xcd_mu);
    if (!g_xcd_done[dev]) HIP_TRY(hipEventCreateWithFlags(&g_xcd_done[dev], hipEventDisableTiming));
    else HIP_TRY(hipStreamWaitEvent(st, g_xcd_done[dev], 0));
    opnet_xcd4_forward<false><<<XCD_COUNT * XCD_CUS, 256, 0, st>>>(x);
    HIP_TRY(hipEventRecord(g_xcd_done[dev], st));
    opnet_xcd4_out_head<<<dim3(T, RB), 256, 0, st>>>(x, nullptr, nullptr);
    opnet_copy_out<<<copy_grid(B, T), 256, 0, st>>>(dio);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// Does the training step of this batch run on the 4-clip persistent kernels?  Reference hidden sizes, a whole device (8 XCDs
// x 32 CUs), at most OPNET_XCD4_MAX_B clips (default 32: one group per XCD - larger batches serialise their row blocks and
// the launch chain's wide step wins again), OPNET_XCD4 = 0 switches it off.
// (Round 6: forward and reverse recurrence are routed separately.  Both kernels read and write the launch chain's own history
// layouts, so a step may run its forward on the persistent kernel and its backward on the chain.  With two or three row blocks per
// XCD the persistent FORWARD still wins - 64 clips 1.31 against 1.78 ms, 96 clips 2.01 against 2.66 - while the reverse recurrence,
// whose phase is longer, loses from 33 clips on: 64 clips 2.32 against 2.22 ms.  OPNET_XCD4_MAX_B sets both limits.)
static bool x4_use(int B, int T, int H1, int H2, bool forward = false)
{
    if (!x4_batch(B, H1, H2) || !x4_on()) return false;
    const int cap = getenv("OPNET_XCD4_MAX_B") ? env_int("OPNET_XCD4_MAX_B", 32) : (forward ? env_int("OPNET_XCD4_FWD_MAX_B", 96) : 32);
    if (B > cap) return false;
    if (train_workspace_layout(B, T, H1, H2).total >= ((size_t)1 << 31)) return false;   // one buffer descriptor
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    return xcd_device_cus(dev) >= XCD_COUNT * XCD_CUS;
}

// Does the training forward of this batch run as the 16-clip persistent launch?  (OPNET_XCD_TRAIN=0 keeps the chain; the switch of
// the 4-clip kernels - thrown by the host after an aborted launch - covers it too.)
static bool xcdt_use(int B, int T, int H1, int H2)
{
    if (!xcdt_batch(B, H1, H2) || !x4_on() || env_int("OPNET_XCD_TRAIN", 1) == 0) return false;
    if (B < env_int("OPNET_XCD_TRAIN_MIN_B", 97)) return false;
    if (train_workspace_layout(B, T, H1, H2).total >= ((size_t)1 << 31)) return false;   // one buffer descriptor, 32-bit offsets
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    return xcd_device_cus(dev) >= XCD_COUNT * XCD_CUS;
}

static int make_x4_args(Xcd4Args *x, const float *packed, void *ws, int B, int T, int H1, int H2)
{
    const TrainWorkspaceLayout W = train_workspace_layout(B, T, H1, H2);
    const TrainPackedLayout TP = train_packed_layout(H1, H2);
    const PackedLayout P = packed_layout(H1, H2);
    memset(x, 0, sizeof(*x));
    x->B = B; x->T = T; x->RB = (B + 31) / 32;
    x->pk = packed + TP.x4fwd;
    x->woutp = packed + P.woutp;
    x->ws = (char *)ws;
    x->xp_off = (unsigned)W.xp;
    x->h1_off = (unsigned)W.h1all; x->h2_off = (unsigned)W.h2all;
    x->c1_off = (unsigned)W.c1all; x->c2_off = (unsigned)W.c2all;
    x->g1_off = (unsigned)W.g1; x->g2_off = (unsigned)W.g2;
    x->ps_off = (unsigned)W.psave; x->x2_off = (unsigned)W.x2all;
    x->lg_off = (unsigned)W.lgstage; x->ys_off = (unsigned)W.ystage;
    x->h1x_off = (unsigned)W.x4h1x; x->h2x_off = (unsigned)W.x4h2x;
    x->status = (unsigned *)((char *)ws + W.x4status);
    x->force_safe = env_int("OPNET_XCD_SAFE", 0);
    x->debug = env_int("OPNET_X4_DEBUG", 0);
    g_x4_last_status = x->status;
    x->trace = g_x4_trace;
    return OPNET_OK;
}

static void make_x4b_args(Xcd4BArgs *x, const float *packed, void *ws, int B, int T, int H1, int H2)
{
    const TrainWorkspaceLayout W = train_workspace_layout(B, T, H1, H2);
    const TrainPackedLayout TP = train_packed_layout(H1, H2);
    memset(x, 0, sizeof(*x));
    x->B = B; x->T = T; x->RB = (B + 31) / 32;
    x->pk = packed + TP.x4bwd;
    x->ws = (char *)ws;
    x->xp_off = (unsigned)W.xp;
    x->c1_off = (unsigned)W.c1all; x->c2_off = (unsigned)W.c2all;
    x->g1_off = (unsigned)W.g1; x->g2_off = (unsigned)W.g2;
    x->ps_off = (unsigned)W.psave; x->dy_off = (unsigned)W.dyp; x->dl_off = (unsigned)W.dlall;
    x->p1x_off = (unsigned)W.x4da1x; x->p2x_off = (unsigned)W.x4da2x; x->dfx_off = (unsigned)W.x4dfx;
    x->debug = env_int("OPNET_X4_DEBUG_BWD", 0);
    x->status = (unsigned *)((char *)ws + W.x4status);
    x->force_safe = env_int("OPNET_XCD_SAFE", 0);
    x->trace = g_x4_trace;
    g_x4_last_status = x->status;
}

extern "C" int opnet_train_forward_f32(const float *boxes, const float *packed, float *y, float *logits,
                                       void *workspace, size_t workspace_bytes, int B, int T, int H1, int H2,
                                       void *stream)
{
    StepArgs a; OpnetIO io; BwdArgs bw;
    if (!boxes || !y || !logits) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(y) || (((uintptr_t)boxes) & 7u)) return fail(OPNET_EINVAL, "y must be 16-byte and boxes 8-byte aligned");
    if (int rc = make_train_args(&a, &io, &bw, boxes, packed, y, logits, workspace, workspace_bytes, B, T, H1, H2))
        return rc;
    const TrainWorkspaceLayout W = train_workspace_layout(B, T, H1, H2);
    hipStream_t st = (hipStream_t)stream;
    OpnetIO *dio = (OpnetIO *)((char *)workspace + W.io);
    if (x4_use(B, T, H1, H2, true)) {
        // small batch on a whole device: the 4-clip persistent step (opnet_xcd4_kernels.hip) writes the same histories.  Three
        // launches per forward: prologue (device-side io, input pack, rings), the recurrence, output head + copies to the caller
        Xcd4Args x;
        if (int rc = make_x4_args(&x, packed, workspace, B, T, H1, H2)) return rc;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        OpnetIO io_nz = io;
        io_nz.state_f4 = 0;                     // (the prologue zeroes slot 0 of the histories itself)
        opnet_x4_train_prologue<<<dim3(T, a.RB + 1), 256, 0, st>>>(dio, io_nz, x);
        std::lock_guard<std::mutex> lock(g_xcd_mu);
        if (!g_xcd_done[dev]) HIP_TRY(hipEventCreateWithFlags(&g_xcd_done[dev], hipEventDisableTiming));
        else HIP_TRY(hipStreamWaitEvent(st, g_xcd_done[dev], 0));
        opnet_xcd4_forward<true><<<XCD_COUNT * XCD_CUS, 256, 0, st>>>(x);
        HIP_TRY(hipEventRecord(g_xcd_done[dev], st));
        opnet_xcd4_out_head<<<dim3(T, a.RB + 1), 256, 0, st>>>(x, (float4 *)y, logits);
        HIP_TRY(hipGetLastError());
        return OPNET_OK;
    }
    opnet_set_io<<<1, 1, 0, st>>>(dio, io);
    opnet_pack_input<<<dim3(T, a.RB), 256, 0, st>>>(dio);
    if (int rc = train_chain_layouts(packed, st)) return rc;
    if (xcdt_use(B, T, H1, H2)) {
        // 97+ clips: the forward as ONE persistent launch of 16-clip groups (opnet_xcd_forward<HO, true>, section 9f) whose finish
        // waves write the chain's histories beside the exchange; the chain's packed input (above) is what the backward reads
        const XcdWorkspaceLayout L = xcd_workspace_layout(B, T);
        char *w = (char *)workspace;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        XcdArgs xa;
        memset(&xa, 0, sizeof(xa));
        xa.B = B; xa.T = T; xa.NGT = L.NGT;
        xa.packed = packed;                     // (the inference image comes first in the training image)
        xa.xp = (const float4 *)(w + W.xcdws + L.xp);
        xa.h1h = (float4 *)(w + W.xcdws + L.h1h);
        xa.h2h = (float4 *)(w + W.xcdws + L.h2h);
        xa.fbh = (float4 *)(w + W.xcdws + L.fbh);
        xa.flags = (unsigned *)(w + W.xcdws + L.flags);
        xa.status = (unsigned *)(w + train_status_offset(W, B, T, H1, H2));
        xa.logits = logits;
        xa.ws = w;
        xa.xp_off = (unsigned)(W.xcdws + L.xp); xa.h1_off = (unsigned)(W.xcdws + L.h1h); xa.h2_off = (unsigned)(W.xcdws + L.h2h);
        xa.fb_off = (unsigned)(W.xcdws + L.fbh); xa.flags_off = (unsigned)(W.xcdws + L.flags);
        xa.status_off = (unsigned)train_status_offset(W, B, T, H1, H2);
        xa.ring = L.ring; xa.yp_off = (unsigned)(W.xcdws + L.yp); xa.y = y;
        xa.trace = nullptr;
        xa.force_safe = env_int("OPNET_XCD_SAFE", 0);
        xa.debug = env_int("OPNET_XCD_DEBUG", 0);
        xa.tr_h1 = (unsigned)W.h1all; xa.tr_c1 = (unsigned)W.c1all; xa.tr_h2 = (unsigned)W.h2all; xa.tr_c2 = (unsigned)W.c2all;
        xa.tr_g1 = (unsigned)W.g1; xa.tr_g2 = (unsigned)W.g2; xa.tr_ps = (unsigned)W.psave; xa.tr_x2 = (unsigned)W.x2all;
        xa.RB = a.RB;
        XcdSources src;
        memset(&src, 0, sizeof(src));
        src.p[0] = boxes; src.start[0] = 0; src.start[1] = B; src.n = 1;
        // a row block whose second 16-clip group does not exist keeps whatever the workspace held in its gate histories: zeros there
        // (0 x NaN would poison the gradients; h / c of those clips are zeroed with the state above)
        if (L.NGT * 16 < a.RB * 32) {
            HIP_TRY(hipMemsetAsync(w + W.g1, 0, (size_t)T * a.RB * (size_t)H1 * 32 * 16, st));
            HIP_TRY(hipMemsetAsync(w + W.g2, 0, (size_t)T * a.RB * (size_t)H2 * 32 * 16, st));
            HIP_TRY(hipMemsetAsync(w + W.psave, 0, (size_t)T * a.RB * 128 * 16, st));
            HIP_TRY(hipMemsetAsync(w + W.x2all, 0, (size_t)T * a.RB * 64 * 16, st));
        }
        std::lock_guard<std::mutex> lock(g_xcd_mu);
        opnet_xcd_pack_input<<<dim3(T + 2, L.NGT), 384, 0, st>>>(src, xa);
        if (!g_xcd_done[dev]) HIP_TRY(hipEventCreateWithFlags(&g_xcd_done[dev], hipEventDisableTiming));
        else HIP_TRY(hipStreamWaitEvent(st, g_xcd_done[dev], 0));
        ProfPair pe{};
        const bool prof = prof_begin(st, &pe);
        if (L.ho) opnet_xcd_forward<true, true><<<XCD_COUNT * XCD_CUS, 512, 0, st>>>(xa);
        else opnet_xcd_forward<false, true><<<XCD_COUNT * XCD_CUS, 512, 0, st>>>(xa);
        if (prof) prof_end(PROF_XCD, st, pe);
        HIP_TRY(hipEventRecord(g_xcd_done[dev], st));
        if (xa.ring && L.ho) {
            opnet_xcd_y_poison<<<64, 256, 0, st>>>(xa, y);
        } else if (xa.ring) {
            const long ny = (long)L.NGT * T * 16;
            opnet_xcd_y_reduce<<<(unsigned)((ny + 255) / 256 > 2048 ? 2048 : (ny + 255) / 256), 256, 0, st>>>(xa, y);
        } else {
            opnet_xcd_out_head<<<dim3(T, L.NGT), 256, 0, st>>>(xa, y);
        }
        HIP_TRY(hipGetLastError());
        return OPNET_OK;
    }
    // the status words of the persistent kernels are sticky from the forward to the weight-gradient launch and the optimiser's
    // guard: a forward on the launch chain has to say "nothing aborted" itself
    if (train_status_offset(W, B, T, H1, H2) != (size_t)-1) HIP_TRY(hipMemsetAsync((char *)workspace + train_status_offset(W, B, T, H1, H2), 0, 32, st));
    const dim3 grid = step_grid(a);
    const opnet_step_fn stepk = step_kernel(a);
    for (int s = 0; s < T + 3; ++s) stepk<<<grid, step_threads(a), 0, st>>>(a, s);
    opnet_copy_out<<<copy_grid(B, T), 256, 0, st>>>(dio);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// The weight gradients of up to OPNET_WGRAD_JOBS products as one wave per (tile, time slice) (opnet_wgrad_tiles + opnet_wgrad_reduce,
// DESIGN.md 9c): 128 x 128 tiles for the products with both dimensions large, 64 x 64 for the rest; the slices sized so that every
// wave job fits ONE round of the SIMDs and big (4 units of work per step) and small (1) end together.  `partial`: WG2_MAX_WAVES x
// WG2_PART_F floats of workspace.  false = not applicable (OPNET_WGRAD2=0, or a slice beyond a buffer descriptor's 2 GiB): the caller
// runs opnet_wgrad.
static bool wgrad_wave_tiles(const WgradArgs *jobs, int njobs, int T, int RB, int B, float *partial, const unsigned *abort, hipStream_t st,
                             int max_waves = WG2_MAX_WAVES)
{
    if (env_int("OPNET_WGRAD2", 1) == 0 || njobs < 1 || njobs > OPNET_WGRAD_JOBS || !partial) return false;
    Wg2Batch tb;
    memset(&tb, 0, sizeof(tb));
    int order[OPNET_WGRAD_JOBS], nb = 0, ns = 0, k = 0;
    for (int pass = 1; pass >= 0; --pass)
        for (int j = 0; j < njobs; ++j) {
            const int big = jobs[j].MQ >= 32 && jobs[j].NQ >= 17;
            if (big == pass) order[k++] = j;
        }
    for (int q = 0; q < njobs; ++q) {
        Wg2Job &J = tb.job[q];
        J.g = jobs[order[q]];
        J.big = J.g.MQ >= 32 && J.g.NQ >= 17;
        const int tq = J.big ? 32 : 16;
        J.tiles_m = (J.g.MQ + tq - 1) / tq;
        J.tiles_n = (J.g.NQ + tq - 1) / tq;
        (J.big ? nb : ns) += J.tiles_m * J.tiles_n;
    }
    // slices per big / small tile: the wave jobs run in rounds of WG2_MAX_WAVES (one wave per SIMD), a round lasts as long as its
    // longest job (a big tile's step is 4 units of work) - the plan with the smallest rounds x length that the partial buffer holds.
    // With few tiles that is one round of many slices (OPNet: 80 big tiles, 12 slices); with many (a 3840-wide W_ih0: 672 big tiles)
    // one slice each would leave a third of the SIMDs idle for the whole launch, three slices run two full rounds of a third the length.
    const long nit = (long)T * RB;
    long best = -1;
    int sb_best = 1, ss_best = 1;
    for (int ss = 1; ss <= 32; ++ss) {
        if ((long)ns * ss > max_waves || ss > nit) break;
        for (int sb = 1; sb <= 32; ++sb) {
            const long nw_ = (long)ns * ss + (long)nb * sb;
            if (nw_ > max_waves || sb > nit) break;
            const long cost_b = nb ? (nit + sb - 1) / sb * 4 : 0, cost_s = ns ? (nit + ss - 1) / ss : 0;
            const long rounds = (nw_ + WG2_MAX_WAVES - 1) / WG2_MAX_WAVES;
            const long cost = rounds * (cost_b > cost_s ? cost_b : cost_s);
            // (ties: the most slices of the big tiles for the same small-tile slices - what the one-round planner of rounds 3-5 chose)
            if (best < 0 || cost < best || (cost == best && ss == ss_best)) { best = cost; sb_best = sb; ss_best = ss; }
            if (!nb) break;
        }
        if (!ns) break;
    }
    if (best < 0) return false;
    // (a wave addresses its slice through a buffer descriptor: 2 GiB from the slice's first step)
    long span = 0;
    for (int q = 0; q < njobs; ++q) {
        const long steps = nit / (tb.job[q].big ? sb_best : ss_best) + 2;
        const long ps = tb.job[q].g.p_stride > tb.job[q].g.q_stride ? tb.job[q].g.p_stride : tb.job[q].g.q_stride;
        if (steps * ps * 16 > span) span = steps * ps * 16;
    }
    if (span >= (1L << 31)) return false;
    int nw = 0;
    for (int q = 0; q < njobs; ++q) {
        Wg2Job &J = tb.job[q];
        J.slices = J.big ? sb_best : ss_best;
        J.wave_begin = nw;
        nw += J.slices * J.tiles_m * J.tiles_n;
    }
    if (nw > max_waves) return false;
    tb.njobs = njobs; tb.nwaves = nw;
    tb.partial = partial;
    tb.abort = abort;
    // one ragged row block: the clip groups past the batch hold zeros in every operand (histories zeroed, da = 0 where dy = 0)
    tb.ncg = 8;
    if (RB == 1 && env_int("OPNET_WGRAD_NCG", 1) != 0) tb.ncg = B <= 4 ? 1 : B <= 8 ? 2 : B <= 16 ? 4 : 8;
    opnet_wgrad_tiles<<<(nw + 3) / 4, 256, 0, st>>>(tb);
    opnet_wgrad_reduce<<<1024, 256, 0, st>>>(tb);
    return true;
}

static int train_backward_impl(const float *dy, const float *packed, void *workspace, size_t workspace_bytes,
                               float *g_ih1, float *g_hh1, float *g_sel, float *g_ih2, float *g_hh2, float *g_out,
                               int B, int T, int H1, int H2, void *stream, int mlp);

extern "C" int opnet_train_backward_f32(const float *dy, const float *packed, void *workspace,
                                        size_t workspace_bytes, float *g_ih1, float *g_hh1, float *g_sel,
                                        float *g_ih2, float *g_hh2, float *g_out, int B, int T, int H1, int H2,
                                        void *stream)
{
    if (!g_hh2) return fail(OPNET_EINVAL, "null pointer");
    return train_backward_impl(dy, packed, workspace, workspace_bytes, g_ih1, g_hh1, g_sel, g_ih2, g_hh2, g_out, B, T,
                               H1, H2, stream, 0);
}

/* OPNetLstmMlp: g_hidden [H2,6] takes the place of g_ih2; there is no g_hh2.  g_ih2_scratch: >= 4*H2*6 floats. */
extern "C" int opnet_mlp_train_backward_f32(const float *dy, const float *packed, void *workspace,
                                            size_t workspace_bytes, float *g_ih1, float *g_hh1, float *g_sel,
                                            float *g_hidden_scratch, float *g_out, int B, int T, int H1, int H2,
                                            void *stream)
{
    return train_backward_impl(dy, packed, workspace, workspace_bytes, g_ih1, g_hh1, g_sel, g_hidden_scratch, nullptr,
                               g_out, B, T, H1, H2, stream, 1);
}

static int train_backward_impl(const float *dy, const float *packed, void *workspace, size_t workspace_bytes,
                               float *g_ih1, float *g_hh1, float *g_sel, float *g_ih2, float *g_hh2, float *g_out,
                               int B, int T, int H1, int H2, void *stream, int mlp)
{
    StepArgs a; OpnetIO io; BwdArgs bw;
    if (!dy || !g_ih1 || !g_hh1 || !g_sel || !g_ih2 || !g_out) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(dy)) return fail(OPNET_EINVAL, "dy must be 16-byte aligned");
    if (int rc = make_train_args(&a, &io, &bw, nullptr, packed, nullptr, nullptr, workspace, workspace_bytes, B, T, H1, H2))
        return rc;
    const TrainWorkspaceLayout W = train_workspace_layout(B, T, H1, H2);
    hipStream_t st = (hipStream_t)stream;
    const int RB = a.RB;
    char *w = (char *)workspace;
    // (33 .. 64 clips as one 4-clip persistent launch per row block, one after the other: measured 2.30 ms against 2.19 for the
    // two chains of fused steps below - not adopted)
    const bool x4_bwd = !mlp && x4_use(B, T, H1, H2) && env_int("OPNET_XCD4_BWD", 1) != 0;
    if (!x4_bwd)        // (the 4-clip persistent form packs dy in its own initialisation launch)
        opnet_pack_dy<<<256, 256, 0, st>>>((const float4 *)dy, (float4 *)(w + W.dyp), (float *)(w + W.dcz),
                                            (long)((W.dcz_end - W.dcz) / 4), B, T, RB);
    bw.mlp = mlp;
    if (mlp) opnet_mlp_dhid<<<4096, 256, 0, st>>>(bw);
    // Reverse recurrence.  Up to four row blocks: ONE fused launch per step - a workgroup owns complete dh rows, so the
    // cell backward rides the product's epilogue.  Larger batches: the split-K pair (4x the workgroups per product,
    // partials met by the next launch's cell kernel) - with many row blocks per tile the fused form's 98 workgroups
    // leave most of the chip idle (measured per step, fused / split: B=96 6.72 / 7.17 ms, B=128 7.79 / 7.92, B=256 12.97 / 12.38).
    const char *mode = getenv("OPNET_BWD_MODE");          // "fused" / "split": measurement override
    const bool fused = mode ? strcmp(mode, "fused") == 0 : RB <= 4;
    // From 3 row blocks on a launch's time grows with the row blocks (6.3 us at 2, 11.4 at 4, 12.6 + 5.3 for the pair at 8) while it
    // keeps a fifth of the matrix pipe busy (at 8 row blocks: 2.2 us of launch, 6 of fragment loads, 2.5 of MFMAs, 1.6 of epilogue,
    // one after the other): slices of the batch as separate chains on separate streams overlap instead (measured
    // 96 / 128 / 192 / 256 / 384 clips: 2 / 2 / 3 / 2-4 / 2 slices best).
    int nsl = env_int("OPNET_BWD_SLICES", -1);
    if (nsl < 0) nsl = RB < 2 ? 1 : (RB == 5 || RB == 6) ? 3 : RB > 12 ? 4 : 2;
    if (nsl > BWD_MAX_SLICES) nsl = BWD_MAX_SLICES;
    if (nsl > RB) nsl = RB;
    const bool sliced = !mode && !mlp && !x4_bwd && nsl > 1 && (RB + nsl - 1) / nsl <= 4 * OPNET_MAX_GY;
    if (x4_bwd) {
        // small batch on a whole device: the 4-clip persistent reverse recurrence (opnet_xcd4_kernels.hip)
        Xcd4BArgs x;
        make_x4b_args(&x, packed, workspace, B, T, H1, H2);
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        opnet_xcd4_init_bwd<<<256, 256, 0, st>>>(x, (const float4 *)dy, (float4 *)(w + W.dyp), (float *)(w + W.dcz),
                                                 (long)((W.dcz_end - W.dcz) / 4), B);
        std::lock_guard<std::mutex> lock(g_xcd_mu);
        if (!g_xcd_done[dev]) HIP_TRY(hipEventCreateWithFlags(&g_xcd_done[dev], hipEventDisableTiming));
        else HIP_TRY(hipStreamWaitEvent(st, g_xcd_done[dev], 0));
        opnet_xcd4_backward<<<XCD_COUNT * XCD_CUS, 256, 0, st>>>(x);
        HIP_TRY(hipEventRecord(g_xcd_done[dev], st));
    } else if (int rc = mlp ? OPNET_OK : train_chain_layouts(packed, st)) {      // (the mlp image is always packed eagerly)
        return rc;
    } else if (sliced) {
        // slices of the batch side by side: S launch chains of the fused step, each over its own row blocks on its own stream
        // (every buffer of the recurrence is indexed by row block, so the chains share nothing but the weights)
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(g_xcd_mu);
        BwdSide &sd = g_bwd_side[dev & 63];
        for (int i = 0; i < nsl - 1; ++i) {
            if (!sd.s[i]) HIP_TRY(hipStreamCreateWithFlags(&sd.s[i], hipStreamNonBlocking));
            if (!sd.join[i]) HIP_TRY(hipEventCreateWithFlags(&sd.join[i], hipEventDisableTiming));
        }
        if (!sd.fork) HIP_TRY(hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(sd.fork, st));
        BwdArgs sl[BWD_MAX_SLICES];
        hipStream_t ss[BWD_MAX_SLICES];
        for (int i = 0; i < nsl; ++i) {
            sl[i] = bw;
            sl[i].rb0 = (int)((long)RB * i / nsl);
            sl[i].rb1 = (int)((long)RB * (i + 1) / nsl);
            ss[i] = i == 0 ? st : sd.s[i - 1];
            if (i) HIP_TRY(hipStreamWaitEvent(ss[i], sd.fork, 0));
        }
        for (int n = 0; n <= T + 1; ++n)
            for (int i = 0; i < nsl; ++i) {
                const int nrb = sl[i].rb1 - sl[i].rb0;
                opnet_bwd_fused<<<dim3(2 * (H2 / 16 + H1 / 16 + 1), nrb < OPNET_MAX_GY ? nrb : OPNET_MAX_GY, 1), FUSED_THREADS, 0, ss[i]>>>(sl[i], n);
            }
        for (int i = 1; i < nsl; ++i) {
            HIP_TRY(hipEventRecord(sd.join[i - 1], ss[i]));
            HIP_TRY(hipStreamWaitEvent(st, sd.join[i - 1], 0));
        }
    } else if (fused) {
        const dim3 gfused(2 * (H2 / 16 + H1 / 16 + 1), RB < OPNET_MAX_GY ? RB : OPNET_MAX_GY, 1);
        for (int n = 0; n <= T + 1; ++n) opnet_bwd_fused<<<gfused, FUSED_THREADS, 0, st>>>(bw, n);
    } else {
        const dim3 gcell(H2 / 8 + H1 / 8, RB, 1);
        const dim3 ggemm(4 * (H2 / 16) + 4 * (H1 / 16) + 4, RB < OPNET_MAX_GY ? RB : OPNET_MAX_GY, 1);
        for (int n = 0; n <= T; ++n) {
            opnet_bwd_cell<<<gcell, 256, 0, st>>>(bw, n);
            if (n < T) opnet_bwd_gemm<<<ggemm, OPNET_THREADS, 0, st>>>(bw, n);
        }
    }
    // weight gradients over the saved histories
    WgradBatch wb;
    int njobs = 0, ntiles = 0;
    auto wgrad = [&](const float4 *P, long ps, int MQ, const float4 *Q, long qs, int NQ, float *out, int ld,
                     int mvalid, int nvalid, int rowmode, int H) {
        WgradArgs &g = wb.job[njobs++];
        g.P = P; g.p_stride = ps; g.MQ = MQ; g.Q = Q; g.q_stride = qs; g.NQ = NQ;
        g.out = out; g.ld = ld; g.mvalid = mvalid; g.nvalid = nvalid; g.rowmode = rowmode; g.H = H;
        g.T = T; g.RB = RB;
        g.tiles_m = (MQ + 15) / 16;
        g.tile_begin = ntiles;
        ntiles += g.tiles_m * ((NQ + 15) / 16);
    };
    const long h1s = (long)(H1 / 4) * 32, h2s = (long)(H2 / 4) * 32;
    // video_LSTM.weight_hh_l0 [4H2][H2]: da2_t x h2_{t-1} (slot t) ; weight_ih_l0 [4H2][6]: da2_t x frames_boxes_t
    if (!mlp) wgrad(bw.g2, (long)H2 * 32, H2, bw.h2all, h2s, H2 / 4, g_hh2, H2, 4 * H2, H2, 1, H2);
    wgrad(bw.g2, (long)H2 * 32, H2, bw.x2all, 64, 2, g_ih2, OPNET_FEATS, 4 * H2, OPNET_FEATS, 1, H2);
    // object_to_track_LSTM.weight_hh_l0 [4H1][H1], weight_ih_l0 [4H1][90]
    wgrad(bw.g1, (long)H1 * 32, H1, bw.h1all, h1s, H1 / 4, g_hh1, H1, 4 * H1, H1, 1, H1);
    wgrad(bw.g1, (long)H1 * 32, H1, bw.xp, OPNET_KXQ * 32, OPNET_KXQ, g_ih1, OPNET_KX, 4 * H1, OPNET_KX, 1, H1);
    // object_to_track_prediction.weight [15][H1]: dl_t x h1_t (slot t+1)
    wgrad(bw.dlall, 128, 4, bw.h1all + (long)RB * h1s, h1s, H1 / 4, g_sel, H1, OPNET_SLOTS, H1, 0, 0);
    // prediction_layer.weight [4][H2]: dy_t x h2_t (slot t+1)
    wgrad(bw.dyp, 32, 1, bw.h2all + (long)RB * h2s, h2s, H2 / 4, g_out, H2, 4, H2, 0, 0);
    for (int j = njobs; j < OPNET_WGRAD_JOBS; ++j) {   // unused slots never match a block
        memset(&wb.job[j], 0, sizeof(WgradArgs));
        wb.job[j].tile_begin = 0x7fffffff;
        wb.job[j].tiles_m = 1;
    }
    // a forward or reverse recurrence that gave up (4-clip persistent kernels) left partial histories: every dW becomes NaN
    wb.abort = train_status_offset(W, B, T, H1, H2) != (size_t)-1 ? (const unsigned *)(w + train_status_offset(W, B, T, H1, H2)) : nullptr;
    if (wgrad_wave_tiles(wb.job, njobs, T, RB, B, (float *)(w + W.wgpart), wb.abort, st)) {
        HIP_TRY(hipGetLastError());
        return OPNET_OK;
    }
    opnet_wgrad<<<ntiles, 256, 0, st>>>(wb);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

/* OPNetLstmMlp training weights: the OPNet training layout with hidden_layer.weight [H2,6] in the gate-0 rows of
 * "W_ih2" and no recurrent video weights.  scratch4h2x6: caller scratch of 4*H2*6 floats. */
extern "C" int opnet_mlp_train_pack_weights_f32(const float *w_ih1, const float *w_hh1, const float *w_sel,
                                                const float *w_hidden, const float *w_out, float *packed,
                                                size_t packed_bytes, float *scratch4h2x6, int H1, int H2, void *stream)
{
    if (int rc = check_dims(1, 1, H1, H2)) return rc;
    if (!scratch4h2x6) return fail(OPNET_EINVAL, "null pointer");
    const TrainPackedLayout L = train_packed_layout(H1, H2);
    if (packed_bytes < L.total * sizeof(float)) return fail(OPNET_EWORKSPACE, "packed buffer too small");
    if (int rc = opnet_mlp_pack_weights_f32(w_ih1, w_hh1, w_sel, w_hidden, w_out, packed, L.fwd_total * sizeof(float), H1, H2, stream))
        return rc;
    {
        // this buffer now holds an OPNetLstmMlp image: a deferred OPNet chain pack remembered for the same address (a freed
        // module's buffer handed out again by the caching allocator) must never run over it
        std::lock_guard<std::mutex> lock(g_tp_mu);
        g_tp.erase(packed);
    }
    hipStream_t st = (hipStream_t)stream;
    auto blocks = [](size_t n) { return (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256); };
    // "W_ih2" [4H2][6] = [hidden_layer.weight ; 0 ; 0 ; 0]
    HIP_TRY(hipMemsetAsync(scratch4h2x6, 0, (size_t)4 * H2 * OPNET_FEATS * sizeof(float), st));
    opnet_copy_f32<<<blocks((size_t)H2 * OPNET_FEATS), 256, 0, st>>>(scratch4h2x6, w_hidden, (long)H2 * OPNET_FEATS);
    opnet_pack_tiles_t<<<blocks((size_t)(H1 / 16) * (H1 / 4) * 256), 256, 0, st>>>(packed + L.w1bt, w_hh1, H1, 0, 0, H1 / 16);
    opnet_pack_tiles_t<<<blocks((size_t)(H2 / 4) * 256), 256, 0, st>>>(packed + L.wih2t, scratch4h2x6, H2, OPNET_FEATS, 1, 1);
    opnet_copy_f32<<<blocks((size_t)OPNET_SLOTS * H1), 256, 0, st>>>(packed + L.wsel, w_sel, (long)OPNET_SLOTS * H1);
    opnet_copy_f32<<<blocks((size_t)4 * H2), 256, 0, st>>>(packed + L.wout, w_out, (long)4 * H2);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opnet_mlp_train_forward_f32(const float *boxes, const float *packed, float *y, float *logits,
                                           void *workspace, size_t workspace_bytes, int B, int T, int H1, int H2,
                                           void *stream)
{
    StepArgs a; OpnetIO io; BwdArgs bw;
    if (!boxes || !y || !logits) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(y) || (((uintptr_t)boxes) & 7u)) return fail(OPNET_EINVAL, "y must be 16-byte and boxes 8-byte aligned");
    if (int rc = make_train_args(&a, &io, &bw, boxes, packed, y, logits, workspace, workspace_bytes, B, T, H1, H2))
        return rc;
    a.mlp = 1;
    const TrainWorkspaceLayout W = train_workspace_layout(B, T, H1, H2);
    hipStream_t st = (hipStream_t)stream;
    OpnetIO *dio = (OpnetIO *)((char *)workspace + W.io);
    opnet_set_io<<<1, 1, 0, st>>>(dio, io);
    opnet_pack_input<<<dim3(T, a.RB), 256, 0, st>>>(dio);
    if (x4_batch(B, H1, H2)) HIP_TRY(hipMemsetAsync((char *)workspace + W.x4status, 0, 32, st));   // (see opnet_train_forward_f32)
    const dim3 grid = step_grid(a);
    const opnet_step_fn stepk = step_kernel(a);
    for (int s = 0; s < T + 3; ++s) stepk<<<grid, step_threads(a), 0, st>>>(a, s);
    opnet_copy_out<<<copy_grid(B, T), 256, 0, st>>>(dio);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

static int l1_family(const float *y, const float *labels, float *loss, float *dy, long n, void *scratch,
                     size_t scratch_bytes, float beta, void *stream);

extern "C" int opnet_l1_loss_f32(const float *y, const float *labels, float *loss, float *dy, long n,
                                 void *scratch, size_t scratch_bytes, void *stream)
{
    return l1_family(y, labels, loss, dy, n, scratch, scratch_bytes, 0.f, stream);
}

extern "C" int opnet_smooth_l1_loss_f32(const float *y, const float *labels, float *loss, float *dy, long n,
                                        float beta, void *scratch, size_t scratch_bytes, void *stream)
{
    if (!(beta > 0.f)) return fail(OPNET_EINVAL, "beta must be positive");
    return l1_family(y, labels, loss, dy, n, scratch, scratch_bytes, beta, stream);
}

static int l1_family(const float *y, const float *labels, float *loss, float *dy, long n, void *scratch,
                     size_t scratch_bytes, float beta, void *stream)
{
    if (!y || !labels || !loss || !scratch) return fail(OPNET_EINVAL, "null pointer");
    if (n <= 0) return fail(OPNET_ESHAPE, "n must be positive");
    int nb = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    if (scratch_bytes < (size_t)nb * 4) return fail(OPNET_EWORKSPACE, "scratch %zu B < %d B", scratch_bytes, nb * 4);
    hipStream_t st = (hipStream_t)stream;
    opnet_l1_partial<<<nb, 256, 0, st>>>(y, labels, dy, (float *)scratch, n, beta);
    opnet_l1_final<<<1, 64, 0, st>>>((const float *)scratch, nb, loss, n);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opnet_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n,
                                   float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                                   void *stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq) return fail(OPNET_EINVAL, "null pointer");
    if (n <= 0 || step <= 0) return fail(OPNET_ESHAPE, "n and step must be positive");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const unsigned nb = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    opnet_adam<<<nb, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps,
                                                    (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), grad_scale);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

/* opnet_adam_step_f32 for `count` (<= 16) tensors that share the hyper-parameters and the step number, as ONE launch.
 * The guarded form skips the whole update on device (no host round trip) when *abort_u32 != 0 (a persistent recurrence gave
 * up: its gradients are NaN), when *loss_f32 is not finite, or when *guard_f32 != 0 (data parallel: some rank aborted). */
extern "C" int opnet_adam_multi_step_guarded_f32(int count, float *const *params, const float *const *grads, float *const *exp_avgs,
                                                 float *const *exp_avg_sqs, const long *numels, float lr, float beta1, float beta2,
                                                 float eps, int step, float grad_scale, const unsigned *abort_u32,
                                                 const float *loss_f32, const float *guard_f32, void *stream);
extern "C" int opnet_adam_multi_step_f32(int count, float *const *params, const float *const *grads, float *const *exp_avgs,
                                         float *const *exp_avg_sqs, const long *numels, float lr, float beta1, float beta2,
                                         float eps, int step, float grad_scale, void *stream)
{
    return opnet_adam_multi_step_guarded_f32(count, params, grads, exp_avgs, exp_avg_sqs, numels, lr, beta1, beta2, eps, step,
                                             grad_scale, nullptr, nullptr, nullptr, stream);
}
extern "C" int opnet_adam_multi_step_guarded_f32(int count, float *const *params, const float *const *grads, float *const *exp_avgs,
                                                 float *const *exp_avg_sqs, const long *numels, float lr, float beta1, float beta2,
                                                 float eps, int step, float grad_scale, const unsigned *abort_u32,
                                                 const float *loss_f32, const float *guard_f32, void *stream)
{
    if (!params || !grads || !exp_avgs || !exp_avg_sqs || !numels) return fail(OPNET_EINVAL, "null pointer");
    if (count <= 0 || count > OPNET_ADAM_MAX) return fail(OPNET_ESHAPE, "1..%d tensors per call (count=%d)", OPNET_ADAM_MAX, count);
    if (step <= 0) return fail(OPNET_ESHAPE, "step must be positive");
    AdamBatch t;
    memset(&t, 0, sizeof(t));
    long nmax = 0;
    for (int k = 0; k < count; ++k) {
        if (!params[k] || !grads[k] || !exp_avgs[k] || !exp_avg_sqs[k]) return fail(OPNET_EINVAL, "null pointer (tensor %d)", k);
        if (numels[k] <= 0) return fail(OPNET_ESHAPE, "tensor %d: n must be positive", k);
        t.p[k] = params[k]; t.g[k] = grads[k]; t.m[k] = exp_avgs[k]; t.v[k] = exp_avg_sqs[k]; t.n[k] = numels[k];
        if (numels[k] > nmax) nmax = numels[k];
    }
    t.count = count;
    t.abort_u32 = abort_u32; t.loss_f32 = loss_f32; t.guard_f32 = guard_f32;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const unsigned nb = (unsigned)((nmax + 255) / 256 > 1024 ? 1024 : (nmax + 255) / 256);
    opnet_adam_multi<<<dim3(nb, count), 256, 0, (hipStream_t)stream>>>(t, beta1, beta2, eps, (float)((double)lr / bc1),
                                                                      (float)(1.0 / sqrt(bc2)), grad_scale);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

/* this rank's data-parallel guard words (see opnet_dp_guard): guard4_f32 = the 4 floats behind the flat gradient bucket */
extern "C" int opnet_dp_guard_f32(float *guard4_f32, const unsigned *abort_u32, const float *loss_f32, float loss_weight,
                                  void *stream)
{
    if (!guard4_f32) return fail(OPNET_EINVAL, "null pointer");
    opnet_dp_guard<<<1, 64, 0, (hipStream_t)stream>>>(guard4_f32, abort_u32, loss_f32, loss_weight);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// ------------------------------------------------------------------------------------------------
// sibling reasoners: stacked LSTM + head, slot embedding, transformer encoder layer
// ------------------------------------------------------------------------------------------------
static int check_stack(int B, int T, int L, int KX, int H)
{
    if (B <= 0 || T <= 0) return fail(OPNET_ESHAPE, "B=%d T=%d must be positive", B, T);
    if (L < 1 || L > SEQ_MAX_LAYERS) return fail(OPNET_ESHAPE, "1..%d LSTM layers supported (L=%d)", SEQ_MAX_LAYERS, L);
    if (KX <= 0 || H <= 0 || (H & 15)) return fail(OPNET_ESHAPE, "H must be a positive multiple of 16 (KX=%d H=%d)", KX, H);
    return OPNET_OK;
}

// A wide layer-0 input (NonLinearLstm: 15 x 256 = 3840 features against H = 512) makes W_ih x_t 88 % of every step's
// weight stream; it does not depend on the recurrence, so inference computes it for all t in ONE GEMM on the conv kernel
// ("hoisting") and the steps walk only W_hh.
static bool stack_hoists_input(int KX, int H) { return (KX & 15) == 0 && KX >= 2 * H; }

typedef void (*stack_step_fn)(const StackArgs, const int);
static bool stack_is_nw8(int RB) { return !getenv("OPNET_STEP_CH") && RB <= env_int("OPNET_NW8_MAX_RB", 1); }
static int stack_step_threads(int RB) { return stack_is_nw8(RB) ? 8 * 64 : OPNET_THREADS; }
static stack_step_fn stack_step_kernel(int RB)
{
    const char *force = getenv("OPNET_STEP_CH");          // "4" / "8": measurement override (4-wave kernels)
    if (stack_is_nw8(RB)) return lstm_stack_step<4, 8>;
    return (force ? atoi(force) == 4 : RB >= 2) ? lstm_stack_step<4> : lstm_stack_step<8>;
}

struct StackPackedLayout { size_t layer[SEQ_MAX_LAYERS], head, wih0g, total; int nhx[SEQ_MAX_LAYERS]; };

static StackPackedLayout stack_packed_layout(int L, int KX, int H)
{
    StackPackedLayout P;
    size_t o = 0;
    for (int l = 0; l < L; ++l) {
        P.nhx[l] = l == 0 ? (KX + 15) / 16 : H / 16;
        P.layer[l] = o;
        o += (size_t)(H / 4) * (P.nhx[l] + H / 16) * 256;
    }
    P.head = o; o += (size_t)(H / 16) * 256;
    P.wih0g = o;                                   // [4H][KX] rows unit*4+gate: weight operand of the hoisted GEMM
    if (stack_hoists_input(KX, H)) o += (size_t)4 * H * KX;
    P.total = o;
    return P;
}

struct StackWorkspaceLayout { size_t xp, state, hbuf[SEQ_MAX_LAYERS], c[SEQ_MAX_LAYERS], state_end, ystage, gemm, xg, xgl[SEQ_MAX_LAYERS], total; };

static StackWorkspaceLayout stack_workspace_layout(int B, int T, int L, int KX, int H)
{
    const size_t RB = (B + 31) / 32, KXP = (size_t)((KX + 15) / 16) * 16;
    StackWorkspaceLayout W;
    size_t o = 0;
    W.xp = o;
    if (!stack_hoists_input(KX, H)) o += (size_t)T * RB * (KXP / 4) * 32 * 16;     // packed x: only when the steps read it
    W.state = o;
    for (int l = 0; l < L; ++l) {
        W.hbuf[l] = o; o += 2 * RB * (size_t)H * 32 * 4;
        W.c[l] = o;    o += RB * (size_t)H * 32 * 4;
    }
    W.state_end = o;
    W.ystage = o; o += RB * 32 * (size_t)T * 16;
    o = align_up(o, 256);
    W.gemm = W.xg = o;
    if (stack_hoists_input(KX, H)) {
        W.gemm = o; o += align_up((size_t)B * T * 4 * H * 4, 256);       // G [B*T][4H]
        W.xg = o;   o += (size_t)T * RB * H * 32 * 16;                     // xg [T][RB][H][32] float4
    }
    for (int l = 0; l < L; ++l) {                                          // x-projection of layers >= 1
        W.xgl[l] = o;
        if (l >= 1) o += (size_t)T * RB * H * 32 * 16;
    }
    W.total = align_up(o, 256);
    return W;
}

extern "C" size_t opseq_lstm_stack_packed_bytes(int L, int KX, int H)
{
    if (check_stack(1, 1, L, KX, H)) return 0;
    return stack_packed_layout(L, KX, H).total * sizeof(float);
}

extern "C" size_t opseq_lstm_stack_workspace_bytes(int B, int T, int L, int KX, int H)
{
    if (check_stack(B, T, L, KX, H)) return 0;
    return stack_workspace_layout(B, T, L, KX, H).total;
}

extern "C" int opseq_lstm_stack_pack_weights_f32(const float *const *w_ih, const float *const *w_hh,
                                                 const float *w_head, float *packed, size_t packed_bytes,
                                                 int L, int KX, int H, void *stream)
{
    if (int rc = check_stack(1, 1, L, KX, H)) return rc;
    if (!w_ih || !w_hh || !w_head || !packed) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed)) return fail(OPNET_EINVAL, "packed must be 16-byte aligned");
    const StackPackedLayout P = stack_packed_layout(L, KX, H);
    if (packed_bytes < P.total * sizeof(float)) return fail(OPNET_EWORKSPACE, "packed buffer too small");
    hipStream_t st = (hipStream_t)stream;
    auto blocks = [](size_t n) { return (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256); };
    for (int l = 0; l < L; ++l) {
        if (!w_ih[l] || !w_hh[l]) return fail(OPNET_EINVAL, "null weight pointer (layer %d)", l);
        const int kx = l == 0 ? KX : H;
        opnet_pack_tiles<<<blocks((size_t)(H / 4) * (P.nhx[l] + H / 16) * 256), 256, 0, st>>>(
            packed + P.layer[l], w_ih[l], w_hh[l], kx, P.nhx[l] * 16, H, H, 0, 0, H / 4);
    }
    opnet_pack_tiles<<<blocks((size_t)(H / 16) * 256), 256, 0, st>>>(packed + P.head, nullptr, w_head, 0, 0, H, 0, 4, 1, 1);
    if (stack_hoists_input(KX, H))
        stack_pack_wih_rows<<<blocks((size_t)4 * H * KX), 256, 0, st>>>(w_ih[0], packed + P.wih0g, H, KX, KX);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// The T+L step launches of a stacked-LSTM forward as one cached hipGraph per (workspace, packed, shape):
// every argument of the step kernel is fixed for that key, the caller-dependent boundary kernels
// (rows_to_packed, copy_y_out) stay eager on the same stream.
struct StackGraphKey {
    const void *ws, *packed;
    int B, T, L, KX, H;
    bool operator<(const StackGraphKey &o) const
    {
        return memcmp(this, &o, sizeof(*this)) < 0;
    }
};
static std::map<StackGraphKey, hipGraphExec_t> g_stack_graphs;
static std::mutex g_stack_graphs_mu;

static int stack_forward_impl(const float *x, const float *packed, float *y, void *workspace,
                              size_t workspace_bytes, int B, int T, int L, int KX, int H, void *stream, bool graph);

extern "C" int opseq_lstm_stack_forward_f32(const float *x, const float *packed, float *y, void *workspace,
                                            size_t workspace_bytes, int B, int T, int L, int KX, int H, void *stream)
{
    return stack_forward_impl(x, packed, y, workspace, workspace_bytes, B, T, L, KX, H, stream, false);
}

extern "C" int opseq_lstm_stack_forward_graph_f32(const float *x, const float *packed, float *y, void *workspace,
                                                  size_t workspace_bytes, int B, int T, int L, int KX, int H,
                                                  void *stream)
{
    return stack_forward_impl(x, packed, y, workspace, workspace_bytes, B, T, L, KX, H, stream, true);
}

extern "C" void opseq_graph_cache_clear(void)
{
    std::lock_guard<std::mutex> lk(g_stack_graphs_mu);
    for (auto &kv : g_stack_graphs) (void)hipGraphExecDestroy(kv.second);
    g_stack_graphs.clear();
}

static int stack_forward_impl(const float *x, const float *packed, float *y, void *workspace,
                              size_t workspace_bytes, int B, int T, int L, int KX, int H, void *stream, bool graph)
{
    if (int rc = check_stack(B, T, L, KX, H)) return rc;
    if (!x || !packed || !y || !workspace) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed) || !aligned16(y) || !aligned16(workspace))
        return fail(OPNET_EINVAL, "packed/y/workspace must be 16-byte aligned");
    const StackWorkspaceLayout W = stack_workspace_layout(B, T, L, KX, H);
    if (workspace_bytes < W.total) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", workspace_bytes, W.total);
    const StackPackedLayout P = stack_packed_layout(L, KX, H);
    char *w = (char *)workspace;
    const int RB = (B + 31) / 32;
    StackArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.RB = RB; a.L = L;
    a.xp = (const float4 *)(w + W.xp);
    int ntiles = 1;
    for (int l = 0; l < L; ++l) {
        a.layer[l].A = (const float4 *)(packed + P.layer[l]);
        a.layer[l].H = H;
        a.layer[l].nhx = P.nhx[l];
        a.layer[l].hbuf = (float4 *)(w + W.hbuf[l]);
        a.layer[l].c = (float *)(w + W.c[l]);
        ntiles += H / 4;
        if (l >= 1) {   // upper layers: the input product runs in its own role one launch earlier
            a.layer[l].xg = (float4 *)(w + W.xgl[l]);
            a.layer[l].a_skip = P.nhx[l];
            a.layer[l].nhx = 0;
            ntiles += H / 4;
        }
    }
    const int nlaunch = T + 2 * L - 1;
    a.headA = (const float4 *)(packed + P.head);
    a.ystage = (float4 *)(w + W.ystage);
    hipStream_t st = (hipStream_t)stream;
    if (stack_hoists_input(KX, H)) {
        if (!aligned16(x)) return fail(OPNET_EINVAL, "x must be 16-byte aligned");
        // G [B*T][4H] = x [B*T][KX] . W_ih0^T  (a 1x1 "conv" over B*T pixels), then into the step kernel's layout
        ConvArgs c = {};
        c.X = x; c.Wt = packed + P.wih0g; c.bias = nullptr; c.R = nullptr; c.Y = (float *)(w + W.gemm);
        c.N = 1; c.H = 1; c.W = B * T; c.Cin = KX; c.Cout = 4 * H; c.KH = 1; c.KW = 1; c.stride = 1; c.pad = 0;
        c.OH = 1; c.OW = B * T; c.KP = KX; c.relu = 0;
        launch_conv_tiled(c, (long)B * T, st);
        const long nx = (long)T * RB * 32 * H;
        stack_xg_repack<<<(unsigned)((nx + 255) / 256 > 8192 ? 8192 : (nx + 255) / 256), 256, 0, st>>>(
            (const float4 *)(w + W.gemm), (float4 *)(w + W.xg), B, T, RB, H);
        a.layer[0].xg = (float4 *)(w + W.xg);
        a.layer[0].a_skip = P.nhx[0];
        a.layer[0].nhx = 0;
        rows_to_packed<<<2048, 256, 0, st>>>(x, (float4 *)(w + W.xp), B, T, RB, 0, 0, (float4 *)(w + W.state),
                                              (long)((W.state_end - W.state) / 16));      // zero the state only
    } else {
        rows_to_packed<<<2048, 256, 0, st>>>(x, (float4 *)(w + W.xp), B, T, RB, KX, P.nhx[0] * 16,
                                              (float4 *)(w + W.state), (long)((W.state_end - W.state) / 16));
    }
    const dim3 grid((ntiles + 7) / 8 * 8, RB < OPNET_MAX_GY ? RB : OPNET_MAX_GY, 1);   // XCD-aligned, see step_grid
    if (!graph) {
        const stack_step_fn stepk = stack_step_kernel(RB);
        for (int s = 0; s < nlaunch; ++s) stepk<<<grid, stack_step_threads(RB), 0, st>>>(a, s);
    } else {
        StackGraphKey key;
        memset(&key, 0, sizeof(key));
        key.ws = workspace; key.packed = packed; key.B = B; key.T = T; key.L = L; key.KX = KX; key.H = H;
        hipGraphExec_t exec = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_stack_graphs_mu);
            auto it = g_stack_graphs.find(key);
            if (it != g_stack_graphs.end()) exec = it->second;
        }
        if (!exec) {
            hipGraph_t g = nullptr;
            HIP_TRY(hipGraphCreate(&g, 0));
            hipGraphNode_t prev = nullptr, node = nullptr;
            for (int s = 0; s < nlaunch; ++s) {
                StackArgs av = a;
                int step = s;
                void *args[] = {(void *)&av, (void *)&step};
                hipKernelNodeParams kp;
                memset(&kp, 0, sizeof(kp));
                kp.func = (void *)stack_step_kernel(RB);
                kp.gridDim = grid;
                kp.blockDim = dim3(stack_step_threads(RB), 1, 1);
                kp.kernelParams = args;
                hipError_t e = hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp);
                if (e != hipSuccess) { (void)hipGraphDestroy(g); return fail(OPNET_EHIP, "hipGraphAddKernelNode: %s", hipGetErrorString(e)); }
                prev = node;
            }
            hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) return fail(OPNET_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
            std::lock_guard<std::mutex> lk(g_stack_graphs_mu);
            g_stack_graphs[key] = exec;
        }
        HIP_TRY(hipGraphLaunch(exec, st));
    }
    const long ny = (long)B * T;
    copy_y_out<<<(unsigned)((ny + 255) / 256 > 1024 ? 1024 : (ny + 255) / 256), 256, 0, st>>>(a.ystage, (float4 *)y, ny);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// ---- the stacked LSTM as ONE persistent launch (seq_xcd_kernels.hip) -------------------------------------------------
// Shapes it is built for: H = 512, L = 1 or 2, layer-0 input either hoisted (KX % 16 == 0 and KX >= 2 H: NonLinearLstm) or
// direct with KX <= 256 in the instantiated quarter widths (BaselineLstm 75 -> 5 k-quads a wave, TransformerLstm 256 -> 16).
// k-quads of layer 0's direct input (K padded to 16): 20 for KX = 65..80, 64 for KX = 241..256; 0 = hoisted
static int seqx_nxq0(int KX, int H) { return stack_hoists_input(KX, H) ? 0 : 4 * ((KX + 15) / 16); }
static bool seqx_dims(int L, int KX, int H)
{
    if (H != SX_H || L < 1 || L > 2 || KX <= 0) return false;
    const int nxt0 = seqx_nxq0(KX, H);
    return (L == 1 && nxt0 == 20) || (L == 2 && (nxt0 == 64 || nxt0 == 0));
}
static std::atomic<int> g_seqx_enabled{1};
// tools: in-kernel timeline of CU 0 of every XCD (device buffer of >= 8 * 4 * phases * 8 uint64; NULL = off)
static unsigned long long *g_seqx_trace = nullptr;
extern "C" void opseq_xcd_set_trace(void *device_buffer) { g_seqx_trace = (unsigned long long *)device_buffer; }
extern "C" void opseq_xcd_enable(int on) { g_seqx_enabled.store(on ? 1 : 0); }
extern "C" int opseq_xcd_supported(int L, int KX, int H)
{
    if (!seqx_dims(L, KX, H) || g_seqx_enabled.load() == 0 || env_int("OPSEQ_XCD", 1) == 0) return 0;
    return x4_device() ? 1 : 0;
}
extern "C" int opseq_xcd_max_batch(int L) { return L == 2 ? 4 * SX_NGMAX * 4 : 4 * SX_NGMAX * 8; }

struct SeqXHostPacked { size_t regs, wih0g, total; };          // floats
static SeqXHostPacked seqx_host_packed(int L, int KX, int H)
{
    SeqXHostPacked P;
    P.regs = 0;
    P.wih0g = align_up(seqx_packed_layout(L, seqx_nxq0(KX, H)).total, 64);
    P.total = P.wih0g + (stack_hoists_input(KX, H) ? (size_t)4 * H * KX : 0);
    return P;
}
struct SeqXWs { size_t status, xp, gemm, hl[2], hc[2], total; };  // bytes
static SeqXWs seqx_ws_layout(int B, int T, int L, int KX, int H)
{
    const size_t RB = (B + 31) / 32, NGT = (B + 3) / 4, KXP = (size_t)((KX + 15) / 16) * 16;
    SeqXWs W;
    size_t o = 0;
    W.status = o; o += 2048;
    W.xp = o;   if (!stack_hoists_input(KX, H)) o += (size_t)T * RB * (KXP / 4) * 32 * 16;
    o = align_up(o, 4096);
    W.gemm = o; if (stack_hoists_input(KX, H)) o += align_up((size_t)B * T * 4 * H * 4, 4096);
    for (int l = 0; l < 2; ++l) {
        W.hl[l] = o; if (l < L) o += NGT * (size_t)(T + 1) * 8192;
        W.hc[l] = o; if (l + 1 < L) o += NGT * (size_t)(T + 1) * 8192;
    }
    W.total = align_up(o, 4096);
    return W;
}
static int check_seqx(int B, int T, int L, int KX, int H)
{
    if (B <= 0 || T <= 0) return fail(OPNET_ESHAPE, "B=%d T=%d must be positive", B, T);
    if (!seqx_dims(L, KX, H))
        return fail(OPNET_ESHAPE, "the persistent stacked LSTM is built for H=512, (L=1, KX<=80) or (L=2, KX=241..256 or a hoisted "
                                  "input); got L=%d KX=%d H=%d - use opseq_lstm_stack_forward_f32", L, KX, H);
    if (B > opseq_xcd_max_batch(L)) return fail(OPNET_ESHAPE, "B=%d > %d clips per launch", B, opseq_xcd_max_batch(L));
    if (seqx_ws_layout(B, T, L, KX, H).total >= ((size_t)1 << 31))
        return fail(OPNET_ESHAPE, "B=%d x T=%d: the workspace exceeds the 2 GiB one buffer descriptor addresses", B, T);
    return OPNET_OK;
}
extern "C" size_t opseq_xcd_packed_bytes(int L, int KX, int H)
{
    return seqx_dims(L, KX, H) ? seqx_host_packed(L, KX, H).total * sizeof(float) : 0;
}
extern "C" size_t opseq_xcd_workspace_bytes(int B, int T, int L, int KX, int H)
{
    if (check_seqx(B, T, L, KX, H)) return 0;
    return seqx_ws_layout(B, T, L, KX, H).total;
}
/* byte offset of the launch's 4 status words in its workspace (see opnet_xcd4_status_offset) */
extern "C" size_t opseq_xcd_status_offset(int B, int T, int L, int KX, int H)
{
    if (check_seqx(B, T, L, KX, H)) return (size_t)-1;
    return seqx_ws_layout(B, T, L, KX, H).status;
}
extern "C" int opseq_xcd_pack_weights_f32(const float *const *w_ih, const float *const *w_hh, float *packed, size_t packed_bytes,
                                          int L, int KX, int H, void *stream)
{
    if (!seqx_dims(L, KX, H)) return fail(OPNET_ESHAPE, "unsupported shape for the persistent stacked LSTM (L=%d KX=%d H=%d)", L, KX, H);
    if (!w_ih || !w_hh || !packed) return fail(OPNET_EINVAL, "null pointer");
    for (int l = 0; l < L; ++l)
        if (!w_ih[l] || !w_hh[l]) return fail(OPNET_EINVAL, "null weight pointer (layer %d)", l);
    if (!aligned16(packed)) return fail(OPNET_EINVAL, "packed must be 16-byte aligned");
    const SeqXHostPacked P = seqx_host_packed(L, KX, H);
    if (packed_bytes < P.total * sizeof(float)) return fail(OPNET_EWORKSPACE, "packed buffer too small");
    hipStream_t st = (hipStream_t)stream;
    seqx_pack<<<2048, 256, 0, st>>>(packed + P.regs, w_ih[0], w_hh[0], L == 2 ? w_ih[1] : nullptr, L == 2 ? w_hh[1] : nullptr, L,
                                    seqx_nxq0(KX, H), KX);
    if (stack_hoists_input(KX, H)) {
        const size_t n = (size_t)4 * H * KX;
        stack_pack_wih_rows<<<(unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256), 256, 0, st>>>(w_ih[0], packed + P.wih0g, H, KX, KX);
    }
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

/* y [B][T][4] = head(LSTM stack(x [B][T][KX])) as ONE persistent launch (+ input pack / hoisted GEMM before, the 4-row head
 * after).  packed: opseq_xcd_pack_weights_f32 image; w_head: predictions_layer.weight [4][H] as the caller holds it. */
extern "C" int opseq_xcd_forward_f32(const float *x, const float *packed, const float *w_head, float *y, void *workspace,
                                     size_t workspace_bytes, int B, int T, int L, int KX, int H, void *stream)
{
    if (int rc = check_seqx(B, T, L, KX, H)) return rc;
    if (!x || !packed || !w_head || !y || !workspace) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed) || !aligned16(y) || !aligned16(workspace) || !aligned16(w_head))
        return fail(OPNET_EINVAL, "packed/w_head/y/workspace must be 16-byte aligned");
    const SeqXWs W = seqx_ws_layout(B, T, L, KX, H);
    if (workspace_bytes < W.total) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", workspace_bytes, W.total);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (xcd_device_cus(dev) < XCD_COUNT * XCD_CUS)
        return fail(OPNET_ESHAPE, "device %d exposes %d CUs; the persistent launch needs %d resident workgroups", dev,
                    xcd_device_cus(dev), XCD_COUNT * XCD_CUS);
    const SeqXHostPacked PK = seqx_host_packed(L, KX, H);
    const int nxq0 = seqx_nxq0(KX, H);
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)workspace;
    const int RB = (B + 31) / 32;
    SeqXArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.L = L; a.NGT = (B + 3) / 4; a.RB = RB; a.KXQ = nxq0;
    a.pk = packed + PK.regs;
    a.whead = w_head;
    a.ws = w;
    a.xp_off = (unsigned)W.xp; a.g_off = (unsigned)W.gemm;
    for (int l = 0; l < 2; ++l) { a.hl_off[l] = (unsigned)W.hl[l]; a.hc_off[l] = (unsigned)W.hc[l]; }
    a.status = (unsigned *)(w + W.status);
    a.ystage = (float4 *)y;
    a.force_safe = env_int("OPNET_XCD_SAFE", 0);
    a.debug = env_int("OPSEQ_XCD_DEBUG", 0);
    a.trace = g_seqx_trace;
    if (nxq0 == 0) {
        if (!aligned16(x)) return fail(OPNET_EINVAL, "x must be 16-byte aligned");
        // G [B*T][4H] = x [B*T][KX] . W_ih0^T (a 1 x 1 "conv" over B*T pixels); the cell reads it where it lies
        ConvArgs c = {};
        c.X = x; c.Wt = packed + PK.wih0g; c.bias = nullptr; c.R = nullptr; c.Y = (float *)(w + W.gemm);
        c.N = 1; c.H = 1; c.W = B * T; c.Cin = KX; c.Cout = 4 * H; c.KH = 1; c.KW = 1; c.stride = 1; c.pad = 0;
        c.OH = 1; c.OW = B * T; c.KP = KX; c.relu = 0;
        launch_conv_tiled(c, (long)B * T, st);
    } else {
        rows_to_packed<<<1024, 256, 0, st>>>(x, (float4 *)(w + W.xp), B, T, RB, KX, 4 * nxq0, nullptr, 0);
    }
    seqx_init<<<512, 256, 0, st>>>(a);
    {
        std::lock_guard<std::mutex> lock(g_xcd_mu);           // two persistent grids must never be co-resident
        if (!g_xcd_done[dev]) HIP_TRY(hipEventCreateWithFlags(&g_xcd_done[dev], hipEventDisableTiming));
        else HIP_TRY(hipStreamWaitEvent(st, g_xcd_done[dev], 0));
        ProfPair pe{};
        const bool prof = prof_begin(st, &pe);
        if (L == 1) seqx_forward<20, 1, false><<<XCD_COUNT * XCD_CUS, 256, 0, st>>>(a);
        else if (nxq0 == 64) seqx_forward<64, 2, false><<<XCD_COUNT * XCD_CUS, 256, 0, st>>>(a);
        else seqx_forward<0, 2, false><<<XCD_COUNT * XCD_CUS, 256, 0, st>>>(a);
        if (prof) prof_end(PROF_SEQX, st, pe);
        HIP_TRY(hipEventRecord(g_xcd_done[dev], st));
    }
    seqx_out_head<<<dim3(T, a.NGT), 64, 0, st>>>(a);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// ---- training of the stacked LSTM ---------------------------------------------------------------------
// ---- the stacked LSTM as ONE persistent launch, throughput form (seq_xcdt_kernels.hip) ---------------------------------
// 16-clip column groups on v_mfma_f32_16x16x4_f32, the two-role scheme of opnet_xcd_forward.  Shapes: H = 512 and
//   L = 1 with a direct input of KX <= 80 (BaselineLstm): one copy of the layer per XCD;
//   L = 2 with KX % 16 == 0 (TransformerLstm 256, NonLinearLstm 3840): the layer-0 input product is hoisted into ONE GEMM
//         (G [B T][2048]) and a pair of XCDs carries the two layers (layer 0 + the lower K half of W_ih1 | the rest of layer 1).
static int seqt_mode(int L, int KX, int H)
{
    if (H != ST_H || KX <= 0) return 0;
    if (L == 1 && KX <= 16 * ST_NXH1) return 1;
    if (L == 2 && (KX & 15) == 0) return 2;
    return 0;
}
static std::atomic<int> g_seqt_enabled{1};
static unsigned long long *g_seqt_trace = nullptr;
extern "C" void opseq_xcdt_set_trace(void *device_buffer) { g_seqt_trace = (unsigned long long *)device_buffer; }
extern "C" void opseq_xcdt_enable(int on) { g_seqt_enabled.store(on ? 1 : 0); }
extern "C" int opseq_xcdt_supported(int L, int KX, int H)
{
    if (!seqt_mode(L, KX, H) || g_seqt_enabled.load() == 0 || env_int("OPSEQ_XCDT", 1) == 0) return 0;
    return x4_device() ? 1 : 0;
}
struct SeqTHostPacked { size_t regs, wih0g, total; };          // floats
static SeqTHostPacked seqt_host_packed(int mode, int KX)
{
    SeqTHostPacked P;
    P.regs = 0;
    P.wih0g = align_up((size_t)(mode == 1 ? 1 : 2) * 128 * (mode == 1 ? ST_NH1 : ST_NH2) * 256, 64);
    P.total = P.wih0g + (mode == 2 ? (size_t)4 * ST_H * KX : 0);
    return P;
}
// one buffer: [status | flags | xp | histories] addressed through one descriptor (< 2 GiB), then G and P1 behind their own
struct SeqTWs { size_t status, flags, xp, hl[2], hc, ws1, g, p1, total; int NGT; };  // bytes
static SeqTWs seqt_ws_layout(int B, int T, int mode, int KX)
{
    (void)KX;
    SeqTWs W;
    const size_t NGT = (B + 15) / 16;
    W.NGT = (int)NGT;
    size_t o = 0;
    W.status = o; o += 2048;
    W.flags = o;  o += align_up(NGT * 96 * 4, 256);
    W.xp = o;     if (mode == 1) o += NGT * (size_t)T * 20 * 256;
    o = align_up(o, 4096);
    for (int l = 0; l < 2; ++l) { W.hl[l] = o; if (l < (mode == 1 ? 1 : 2)) o += NGT * (size_t)(T + 1) * 128 * 256; }
    W.hc = o;     if (mode == 2) o += NGT * (size_t)(T + 1) * 64 * 256;
    W.ws1 = o = align_up(o, 4096);
    W.g = o;      if (mode == 2) o += align_up((size_t)B * T * 4 * ST_H * 4, 4096);
    W.p1 = o;     if (mode == 2) o += NGT * (size_t)T * 128 * 1024;
    W.total = align_up(o, 4096);
    return W;
}
/* clips one launch carries at T frames (0 = unsupported shape): ST_NGMAX groups per XCD (pair), and every descriptor < 2 GiB */
extern "C" int opseq_xcdt_max_batch(int T, int L, int KX, int H)
{
    const int mode = seqt_mode(L, KX, H);
    if (!mode || T <= 0) return 0;
    long ng = (long)ST_NGMAX * (mode == 1 ? 8 : 4);
    const long lim = ((long)1 << 31) - (1 << 20);
    while (ng > 0) {
        const SeqTWs W = seqt_ws_layout((int)(ng * 16), T, mode, KX);
        if ((long)W.ws1 < lim && (long)(W.p1 - W.g) < lim && (long)(W.total - W.p1) < lim) break;
        --ng;
    }
    return (int)(ng * 16);
}
static int check_seqt(int B, int T, int L, int KX, int H)
{
    if (B <= 0 || T <= 0) return fail(OPNET_ESHAPE, "B=%d T=%d must be positive", B, T);
    if (!seqt_mode(L, KX, H))
        return fail(OPNET_ESHAPE, "the throughput form of the persistent stacked LSTM is built for H=512, (L=1, KX<=80) or (L=2, KX a "
                                  "multiple of 16); got L=%d KX=%d H=%d", L, KX, H);
    if (B > opseq_xcdt_max_batch(T, L, KX, H))
        return fail(OPNET_ESHAPE, "B=%d > %d clips per launch at T=%d", B, opseq_xcdt_max_batch(T, L, KX, H), T);
    return OPNET_OK;
}
extern "C" size_t opseq_xcdt_packed_bytes(int L, int KX, int H)
{
    const int mode = seqt_mode(L, KX, H);
    return mode ? seqt_host_packed(mode, KX).total * sizeof(float) : 0;
}
extern "C" size_t opseq_xcdt_workspace_bytes(int B, int T, int L, int KX, int H)
{
    if (check_seqt(B, T, L, KX, H)) return 0;
    return seqt_ws_layout(B, T, seqt_mode(L, KX, H), KX).total;
}
extern "C" size_t opseq_xcdt_status_offset(int B, int T, int L, int KX, int H)
{
    if (check_seqt(B, T, L, KX, H)) return (size_t)-1;
    return seqt_ws_layout(B, T, seqt_mode(L, KX, H), KX).status;
}
extern "C" int opseq_xcdt_pack_weights_f32(const float *const *w_ih, const float *const *w_hh, float *packed, size_t packed_bytes,
                                           int L, int KX, int H, void *stream)
{
    const int mode = seqt_mode(L, KX, H);
    if (!mode) return fail(OPNET_ESHAPE, "unsupported shape for the throughput form of the persistent stacked LSTM (L=%d KX=%d H=%d)", L, KX, H);
    if (!w_ih || !w_hh || !packed) return fail(OPNET_EINVAL, "null pointer");
    for (int l = 0; l < L; ++l)
        if (!w_ih[l] || !w_hh[l]) return fail(OPNET_EINVAL, "null weight pointer (layer %d)", l);
    if (!aligned16(packed)) return fail(OPNET_EINVAL, "packed must be 16-byte aligned");
    const SeqTHostPacked P = seqt_host_packed(mode, KX);
    if (packed_bytes < P.total * sizeof(float)) return fail(OPNET_EWORKSPACE, "packed buffer too small");
    hipStream_t st = (hipStream_t)stream;
    seqt_pack<<<2048, 256, 0, st>>>(packed + P.regs, w_ih[0], w_hh[0], L == 2 ? w_ih[1] : nullptr, L == 2 ? w_hh[1] : nullptr, mode, KX);
    if (mode == 2) {
        const size_t n = (size_t)4 * H * KX;
        stack_pack_wih_rows<<<(unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256), 256, 0, st>>>(w_ih[0], packed + P.wih0g, H, KX, KX);
    }
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

/* y [B][T][4] = head(LSTM stack(x [B][T][KX])) as ONE persistent launch of 16-clip groups (+ input pack / hoisted GEMM before,
 * the 4-row head after).  packed: opseq_xcdt_pack_weights_f32 image; w_head: predictions_layer.weight [4][H]. */
extern "C" int opseq_xcdt_forward_f32(const float *x, const float *packed, const float *w_head, float *y, void *workspace,
                                      size_t workspace_bytes, int B, int T, int L, int KX, int H, void *stream)
{
    if (int rc = check_seqt(B, T, L, KX, H)) return rc;
    if (!x || !packed || !w_head || !y || !workspace) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed) || !aligned16(y) || !aligned16(workspace) || !aligned16(w_head))
        return fail(OPNET_EINVAL, "packed/w_head/y/workspace must be 16-byte aligned");
    const int mode = seqt_mode(L, KX, H);
    const SeqTWs W = seqt_ws_layout(B, T, mode, KX);
    if (workspace_bytes < W.total) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", workspace_bytes, W.total);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (xcd_device_cus(dev) < XCD_COUNT * XCD_CUS)
        return fail(OPNET_ESHAPE, "device %d exposes %d CUs; the persistent launch needs %d resident workgroups", dev,
                    xcd_device_cus(dev), XCD_COUNT * XCD_CUS);
    const SeqTHostPacked PK = seqt_host_packed(mode, KX);
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)workspace;
    SeqTArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.mode = mode; a.NGT = W.NGT; a.KX = KX;
    a.pk = packed + PK.regs;
    a.whead = w_head;
    a.ws = w;
    a.xp_off = (unsigned)W.xp; a.hl_off[0] = (unsigned)W.hl[0]; a.hl_off[1] = (unsigned)W.hl[1]; a.hc_off = (unsigned)W.hc;
    a.flags_off = (unsigned)W.flags; a.status_off = (unsigned)W.status;
    a.status = (unsigned *)(w + W.status);
    a.G = mode == 2 ? (const float *)(w + W.g) : nullptr;
    a.P1 = mode == 2 ? (float4 *)(w + W.p1) : nullptr;
    a.y = (float4 *)y;
    a.force_safe = env_int("OPNET_XCD_SAFE", 0);
    a.debug = env_int("OPSEQ_XCDT_DEBUG", 0);
    a.trace = g_seqt_trace;
    if (mode == 2) {
        if (!aligned16(x)) return fail(OPNET_EINVAL, "x must be 16-byte aligned");
        // G [B*T][4H] = x [B*T][KX] . W_ih0^T (a 1 x 1 "conv" over B*T pixels); the layer-0 cell reads it where it lies
        ConvArgs c = {};
        c.X = x; c.Wt = packed + PK.wih0g; c.bias = nullptr; c.R = nullptr; c.Y = (float *)(w + W.g);
        c.N = 1; c.H = 1; c.W = B * T; c.Cin = KX; c.Cout = 4 * H; c.KH = 1; c.KW = 1; c.stride = 1; c.pad = 0;
        c.OH = 1; c.OW = B * T; c.KP = KX; c.relu = 0;
        if (int rc = launch_gemm_k256(c, (long)B * T, st)) return rc;
    }
    seqt_init<<<1024, 256, 0, st>>>(a, x);
    {
        std::lock_guard<std::mutex> lock(g_xcd_mu);           // two persistent grids must never be co-resident
        if (!g_xcd_done[dev]) HIP_TRY(hipEventCreateWithFlags(&g_xcd_done[dev], hipEventDisableTiming));
        else HIP_TRY(hipStreamWaitEvent(st, g_xcd_done[dev], 0));
        ProfPair pe{};
        const bool prof = prof_begin(st, &pe);
        if (mode == 1) seqt_forward<1><<<XCD_COUNT * XCD_CUS, 512, 0, st>>>(a);
        else seqt_forward<2><<<XCD_COUNT * XCD_CUS, 512, 0, st>>>(a);
        if (prof) prof_end(PROF_SEQT, st, pe);
        HIP_TRY(hipEventRecord(g_xcd_done[dev], st));
    }
    seqt_out_head<<<dim3(T, a.NGT), 256, 0, st>>>(a);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

struct StackTrainPacked { size_t fwd_total, whh_t[SEQ_MAX_LAYERS], wih_t[SEQ_MAX_LAYERS], whead, wih0_t, seqx, seqxb, total; };

static StackTrainPacked stack_train_packed_layout(int L, int KX, int H)
{
    StackTrainPacked P;
    size_t o = align_up(stack_packed_layout(L, KX, H).total, 4);
    P.fwd_total = stack_packed_layout(L, KX, H).total;
    for (int l = 0; l < L; ++l) {
        P.whh_t[l] = o; o += (size_t)(H / 16) * (H / 4) * 256;
        P.wih_t[l] = o; if (l >= 1) o += (size_t)(H / 16) * (H / 4) * 256;
    }
    P.whead = o; o += align_up((size_t)4 * H, 4);
    P.wih0_t = o; o += align_up((size_t)(((KX + 15) / 16) * 16) * 4 * H, 4);   // W_ih0^T [KXP][4H], k = 4*unit + gate
    o = align_up(o, 64);
    P.seqx = o;                                    // register image of the persistent forward (seq_xcd_kernels.hip)
    if (seqx_dims(L, KX, H)) o += seqx_packed_layout(L, seqx_nxq0(KX, H)).total;
    o = align_up(o, 64);
    P.seqxb = o;                                   // register image of the persistent reverse recurrence (seq_xcd_bwd_kernels.hip)
    if (seqx_dims(L, KX, H)) o += seqxb_packed_layout(L).total;
    P.total = o;
    return P;
}

struct StackTrainWs {
    size_t xp, state, hall[SEQ_MAX_LAYERS], call[SEQ_MAX_LAYERS], state_end, g[SEQ_MAX_LAYERS], ystage, dyp,
        rpart[SEQ_MAX_LAYERS], dxpart[SEQ_MAX_LAYERS], dcz, dc[SEQ_MAX_LAYERS], dcz_end, darows, wgpart,
        sx_status, sx_hl[2], sx_hc[2],             // sx_*: exchange histories + status words of the persistent forward
        sxb_ring[2], sxb_dxring, sxb_dxh, total;   // sxb_*: exchange rings of the persistent reverse recurrence
};
static bool seqx_train_shape(int B, int L, int KX, int H) { return seqx_dims(L, KX, H) && B <= opseq_xcd_max_batch(L); }

// wave jobs the stack's weight-gradient launch may plan (the capacity of its partial-tile buffer): one round of the SIMDs, or two when
// the products have so many 128 x 128 tiles (a wide layer-0 input: 480 of them at KX = 3840) that one round could not even hold two
// time slices per tile
static int stack_wgrad_max_waves(int L, int KX, int H)
{
    const int tm = (H + 31) / 32;                                   // 4H rows = H m-quads
    const int kxq = ((KX + 15) / 16) * 4;                           // input quads (padded to 16)
    long big = (long)(2 * L - 1) * tm * ((H / 4 + 31) / 32);       // W_hh of every layer, W_ih of the upper layers
    if (H >= 32 && kxq >= 17) big += (long)tm * ((kxq + 31) / 32);  // W_ih0
    return big * 2 > WG2_MAX_WAVES ? 2 * WG2_MAX_WAVES : WG2_MAX_WAVES;
}

static StackTrainWs stack_train_ws_layout(int B, int T, int L, int KX, int H)
{
    const size_t RB = (B + 31) / 32, KXP = (size_t)((KX + 15) / 16) * 16, TT = T;
    StackTrainWs W;
    size_t o = 0;
    W.xp = o; o += TT * RB * (KXP / 4) * 32 * 16;
    W.state = o;
    for (int l = 0; l < L; ++l) {
        W.hall[l] = o; o += (TT + 1) * RB * (size_t)H * 32 * 4;
        W.call[l] = o; o += (TT + 1) * RB * (size_t)H * 32 * 4;
    }
    W.state_end = o;
    for (int l = 0; l < L; ++l) { W.g[l] = o; o += TT * RB * (size_t)H * 32 * 16; }
    W.ystage = o; o += RB * 32 * TT * 16;
    W.dyp = o;    o += TT * RB * 32 * 16;
    for (int l = 0; l < L; ++l) {
        W.rpart[l] = o;  o += 4 * RB * (size_t)H * 32 * 4;
        W.dxpart[l] = o; o += 4 * RB * (size_t)H * 32 * 4;
    }
    W.dcz = o;
    for (int l = 0; l < L; ++l) { W.dc[l] = o; o += RB * (size_t)H * 32 * 4; }
    W.dcz_end = o;
    W.darows = o; o += (size_t)B * TT * 4 * H * 4;      // da0 as rows, for the input-gradient GEMM
    o = align_up(o, 4096);
    W.wgpart = o; o += (size_t)stack_wgrad_max_waves(L, KX, H) * WG2_PART_F * 4;     // partial tiles of the weight-gradient waves (64 / 128 MB)
    W.sx_status = o;
    for (int l = 0; l < 2; ++l) W.sx_hl[l] = W.sx_hc[l] = o;
    if (seqx_train_shape(B, L, KX, H)) {
        const size_t NGT = (B + 3) / 4;
        o += 4096;
        for (int l = 0; l < L; ++l) {
            W.sx_hl[l] = o; o += NGT * (TT + 1) * 8192;
            W.sx_hc[l] = o; if (l + 1 < L) o += NGT * (TT + 1) * 8192;
        }
    }
    // the persistent reverse recurrence: a ring of partial dh rows per layer (+ the top layer's partials of the lower layer's dh);
    // their sums for the lower layer's XCD reuse the forward's cross-XCD history (sx_hc[0]: idle once the forward is over)
    for (int l = 0; l < 2; ++l) W.sxb_ring[l] = o;
    W.sxb_dxring = o; W.sxb_dxh = W.sx_hc[0];
    if (seqx_train_shape(B, L, KX, H)) {
        const size_t NGT = (B + 3) / 4, ring = NGT * SXB_SLOTS * 32 * 32 * 256;
        for (int l = 0; l < L; ++l) { W.sxb_ring[l] = o; o += ring; }
        if (L == 2) { W.sxb_dxring = o; o += ring; }
    }
    W.total = align_up(o, 256);
    return W;
}
/* byte offset of the training forward's status words in its workspace ((size_t)-1: the shape never runs the persistent kernel) */
extern "C" size_t opseq_lstm_stack_train_status_offset(int B, int T, int L, int KX, int H)
{
    if (check_stack(B, T, L, KX, H) || !seqx_train_shape(B, L, KX, H)) return (size_t)-1;
    const auto W = stack_train_ws_layout(B, T, L, KX, H);
    // the same predicate as the training forward's use_sx: a workspace of 2 GiB or more runs the launch chain (32-bit offsets
    // in the persistent kernel), which has no status words
    if (W.total >= ((size_t)1 << 31)) return (size_t)-1;
    return W.sx_status;
}

extern "C" size_t opseq_lstm_stack_train_packed_bytes(int L, int KX, int H)
{
    if (check_stack(1, 1, L, KX, H)) return 0;
    return stack_train_packed_layout(L, KX, H).total * sizeof(float);
}

extern "C" size_t opseq_lstm_stack_train_workspace_bytes(int B, int T, int L, int KX, int H)
{
    if (check_stack(B, T, L, KX, H)) return 0;
    return stack_train_ws_layout(B, T, L, KX, H).total;
}

// W_ih0 [4H][KX] (torch gate-major rows) -> [KXP][4H] with column k = 4*unit + gate, zero rows beyond KX
__global__ void pack_wih0_t(float *__restrict__ out, const float *__restrict__ w, int H, int KX, int KXP)
{
    const long n = (long)KXP * 4 * H;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int k = idx % (4 * H);
        const int x = idx / (4 * H);
        const int unit = k >> 2, gate = k & 3;
        out[idx] = x < KX ? w[((long)gate * H + unit) * KX + x] : 0.f;
    }
}

extern "C" int opseq_lstm_stack_train_pack_weights_f32(const float *const *w_ih, const float *const *w_hh,
                                                       const float *w_head, float *packed, size_t packed_bytes,
                                                       int L, int KX, int H, void *stream)
{
    if (int rc = check_stack(1, 1, L, KX, H)) return rc;
    const StackTrainPacked P = stack_train_packed_layout(L, KX, H);
    if (packed_bytes < P.total * sizeof(float)) return fail(OPNET_EWORKSPACE, "packed buffer too small");
    if (int rc = opseq_lstm_stack_pack_weights_f32(w_ih, w_hh, w_head, packed, P.fwd_total * sizeof(float), L, KX, H, stream))
        return rc;
    hipStream_t st = (hipStream_t)stream;
    auto blocks = [](size_t n) { return (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256); };
    const size_t tile_f = (size_t)(H / 16) * (H / 4) * 256;
    for (int l = 0; l < L; ++l) {
        opnet_pack_tiles_t<<<blocks(tile_f), 256, 0, st>>>(packed + P.whh_t[l], w_hh[l], H, 0, 0, H / 16);
        if (l >= 1) opnet_pack_tiles_t<<<blocks(tile_f), 256, 0, st>>>(packed + P.wih_t[l], w_ih[l], H, 0, 0, H / 16);
    }
    opnet_copy_f32<<<blocks((size_t)4 * H), 256, 0, st>>>(packed + P.whead, w_head, (long)4 * H);
    const int KXP = ((KX + 15) / 16) * 16;
    pack_wih0_t<<<blocks((size_t)KXP * 4 * H), 256, 0, st>>>(packed + P.wih0_t, w_ih[0], H, KX, KXP);
    if (seqx_dims(L, KX, H))
        seqx_pack<<<2048, 256, 0, st>>>(packed + P.seqx, w_ih[0], w_hh[0], L == 2 ? w_ih[1] : nullptr, L == 2 ? w_hh[1] : nullptr, L,
                                        seqx_nxq0(KX, H), KX);
    if (seqx_dims(L, KX, H))
        seqxb_pack<<<2048, 256, 0, st>>>(packed + P.seqxb, w_hh[0], L == 2 ? w_hh[1] : nullptr, L == 2 ? w_ih[1] : nullptr, w_head, L);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

static int make_stack_train_args(StackArgs *a, StackBwdArgs *b, const float *packed, void *workspace,
                                 size_t workspace_bytes, int B, int T, int L, int KX, int H)
{
    if (int rc = check_stack(B, T, L, KX, H)) return rc;
    if (!packed || !workspace) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(packed) || !aligned16(workspace)) return fail(OPNET_EINVAL, "packed/workspace must be 16-byte aligned");
    const StackTrainWs W = stack_train_ws_layout(B, T, L, KX, H);
    if (workspace_bytes < W.total) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", workspace_bytes, W.total);
    const StackPackedLayout P = stack_packed_layout(L, KX, H);
    const StackTrainPacked TP = stack_train_packed_layout(L, KX, H);
    char *w = (char *)workspace;
    const int RB = (B + 31) / 32;
    memset(a, 0, sizeof(*a));
    memset(b, 0, sizeof(*b));
    a->B = B; a->T = T; a->RB = RB; a->L = L; a->train = 1;
    a->xp = (const float4 *)(w + W.xp);
    b->B = B; b->T = T; b->RB = RB; b->L = L; b->H = H;
    for (int l = 0; l < L; ++l) {
        a->layer[l].A = (const float4 *)(packed + P.layer[l]);
        a->layer[l].H = H;
        a->layer[l].nhx = P.nhx[l];
        a->layer[l].hbuf = (float4 *)(w + W.hall[l]);
        a->layer[l].c = (float *)(w + W.call[l]);
        a->layer[l].gsave = (float4 *)(w + W.g[l]);
        if (l >= 1) {   // x-projection role; its output shares the gate-save buffer (read before overwritten, same thread)
            a->layer[l].xg = (float4 *)(w + W.g[l]);
            a->layer[l].a_skip = P.nhx[l];
            a->layer[l].nhx = 0;
        }
        b->layer[l].whh_t = (const float4 *)(packed + TP.whh_t[l]);
        b->layer[l].wih_t = l >= 1 ? (const float4 *)(packed + TP.wih_t[l]) : nullptr;
        b->layer[l].g = (float4 *)(w + W.g[l]);
        b->layer[l].call = (const float *)(w + W.call[l]);
        b->layer[l].rpart = (float *)(w + W.rpart[l]);
        b->layer[l].dxpart = (float *)(w + W.dxpart[l]);
        b->layer[l].dc = (float *)(w + W.dc[l]);
    }
    a->headA = (const float4 *)(packed + P.head);
    a->ystage = (float4 *)(w + W.ystage);
    b->dyp = (const float4 *)(w + W.dyp);
    b->whead = packed + TP.whead;
    return OPNET_OK;
}

extern "C" int opseq_lstm_stack_train_forward_f32(const float *x, const float *packed, float *y, void *workspace,
                                                  size_t workspace_bytes, int B, int T, int L, int KX, int H,
                                                  void *stream)
{
    StackArgs a; StackBwdArgs b;
    if (!x || !y) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(y)) return fail(OPNET_EINVAL, "y must be 16-byte aligned");
    if (int rc = make_stack_train_args(&a, &b, packed, workspace, workspace_bytes, B, T, L, KX, H)) return rc;
    const StackTrainWs W = stack_train_ws_layout(B, T, L, KX, H);
    const StackPackedLayout P = stack_packed_layout(L, KX, H);
    char *w = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    rows_to_packed<<<2048, 256, 0, st>>>(x, (float4 *)(w + W.xp), B, T, a.RB, KX, P.nhx[0] * 16,
                                          (float4 *)(w + W.state), (long)((W.state_end - W.state) / 16));
    const bool use_sx = seqx_train_shape(B, L, KX, H) && opseq_xcd_supported(L, KX, H) && W.total < ((size_t)1 << 31);
    if (stack_hoists_input(KX, H)) {
        // same hoisted input product as the inference forward (bit-identical y); the packed x above is still needed
        // by the weight-gradient GEMM.  Scratch: G lives in the da-rows buffer (used by backward only), xg in layer
        // 0's gate-save buffer - every thread reads its xg element before it overwrites it with the saved gates.
        if (!aligned16(x)) return fail(OPNET_EINVAL, "x must be 16-byte aligned");
        ConvArgs c = {};
        c.X = x; c.Wt = packed + P.wih0g; c.bias = nullptr; c.R = nullptr; c.Y = (float *)(w + W.darows);
        c.N = 1; c.H = 1; c.W = B * T; c.Cin = KX; c.Cout = 4 * H; c.KH = 1; c.KW = 1; c.stride = 1; c.pad = 0;
        c.OH = 1; c.OW = B * T; c.KP = KX; c.relu = 0;
        launch_conv_tiled(c, (long)B * T, st);
        const long nx = (long)T * a.RB * 32 * H;
        if (!use_sx)
            stack_xg_repack<<<(unsigned)((nx + 255) / 256 > 8192 ? 8192 : (nx + 255) / 256), 256, 0, st>>>(
                (const float4 *)(w + W.darows), (float4 *)(w + W.g[0]), B, T, a.RB, H);
        a.layer[0].xg = (float4 *)(w + W.g[0]);
        a.layer[0].a_skip = P.nhx[0];
        a.layer[0].nhx = 0;
    }
    if (use_sx) {
        // the whole recurrence as ONE persistent launch (seq_xcd_kernels.hip, TRAIN = true): the same arithmetic as the
        // inference forward of this shape (bit-identical y) + the h / c / gate histories the backward below reads, written in
        // the launch chain's own layouts.  (The hoisted input product: the cell reads G where the GEMM left it - W.darows.)
        const StackTrainPacked TP = stack_train_packed_layout(L, KX, H);
        const int nxq0 = seqx_nxq0(KX, H);
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        SeqXArgs sx;
        memset(&sx, 0, sizeof(sx));
        sx.B = B; sx.T = T; sx.L = L; sx.NGT = (B + 3) / 4; sx.RB = a.RB; sx.KXQ = nxq0;
        sx.pk = packed + TP.seqx;
        sx.whead = packed + TP.whead;
        sx.ws = w;
        sx.xp_off = (unsigned)W.xp; sx.g_off = (unsigned)W.darows;
        for (int l = 0; l < 2; ++l) { sx.hl_off[l] = (unsigned)W.sx_hl[l]; sx.hc_off[l] = (unsigned)W.sx_hc[l]; }
        sx.status = (unsigned *)(w + W.sx_status);
        sx.ystage = (float4 *)y;
        sx.force_safe = env_int("OPNET_XCD_SAFE", 0);
        sx.debug = env_int("OPSEQ_XCD_DEBUG", 0);
        for (int l = 0; l < L; ++l) {
            sx.hall[l] = (float *)(w + W.hall[l]);
            sx.call[l] = (float *)(w + W.call[l]);
            sx.gsave[l] = (float4 *)(w + W.g[l]);
        }
        seqx_init<<<512, 256, 0, st>>>(sx);
        // The launch works on groups of 4 clips; the backward and the weight-gradient GEMM on whole row blocks of 32.  Clips of the
        // last row block that no group covers keep whatever the caller's workspace held in the gate histories - and a NaN there turns
        // 0 x NaN into a NaN gradient (seen with a workspace that an aborted launch's poisoned tensors had occupied before).  Zeros
        // there (their h / c histories are zeroed with the state above): only for ragged batches, ~25 us per 64 MB.
        if (sx.NGT * 4 < a.RB * 32)
            for (int l = 0; l < L; ++l)
                HIP_TRY(hipMemsetAsync(w + W.g[l], 0, (size_t)T * a.RB * (size_t)H * 32 * 16, st));
        {
            std::lock_guard<std::mutex> lock(g_xcd_mu);
            if (!g_xcd_done[dev]) HIP_TRY(hipEventCreateWithFlags(&g_xcd_done[dev], hipEventDisableTiming));
            else HIP_TRY(hipStreamWaitEvent(st, g_xcd_done[dev], 0));
            ProfPair pe{};
            const bool prof = prof_begin(st, &pe);
            if (L == 1) seqx_forward<20, 1, true><<<XCD_COUNT * XCD_CUS, 256, 0, st>>>(sx);
            else if (nxq0 == 64) seqx_forward<64, 2, true><<<XCD_COUNT * XCD_CUS, 256, 0, st>>>(sx);
            else seqx_forward<0, 2, true><<<XCD_COUNT * XCD_CUS, 256, 0, st>>>(sx);
            if (prof) prof_end(PROF_SEQX, st, pe);
            HIP_TRY(hipEventRecord(g_xcd_done[dev], st));
        }
        seqx_out_head<<<dim3(T, sx.NGT), 64, 0, st>>>(sx);
        HIP_TRY(hipGetLastError());
        return OPNET_OK;
    }
    // (the status words a caller's optimiser guard reads: a forward on the launch chain says "nothing aborted" itself)
    if (seqx_train_shape(B, L, KX, H)) HIP_TRY(hipMemsetAsync(w + W.sx_status, 0, 32, st));
    const dim3 grid(((2 * L - 1) * (H / 4) + 1 + 7) / 8 * 8, a.RB < OPNET_MAX_GY ? a.RB : OPNET_MAX_GY, 1);
    const stack_step_fn stepk = stack_step_kernel(a.RB);
    for (int s = 0; s < T + 2 * L - 1; ++s) stepk<<<grid, stack_step_threads(a.RB), 0, st>>>(a, s);
    const long ny = (long)B * T;
    copy_y_out<<<(unsigned)((ny + 255) / 256 > 1024 ? 1024 : (ny + 255) / 256), 256, 0, st>>>(a.ystage, (float4 *)y, ny);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

/* g_ih / g_hh: HOST arrays of L device pointers (gradients in the state_dict layouts); dx0 (nullable):
 * gradient of the stack's input x [B,T,KX] (needed by models with trainable layers below the LSTM). */
extern "C" int opseq_lstm_stack_train_backward_f32(const float *dy, const float *packed, void *workspace,
                                                   size_t workspace_bytes, float *const *g_ih, float *const *g_hh,
                                                   float *g_head, float *dx0, int B, int T, int L, int KX, int H,
                                                   void *stream)
{
    StackArgs a; StackBwdArgs b;
    if (!dy || !g_ih || !g_hh || !g_head) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(dy)) return fail(OPNET_EINVAL, "dy must be 16-byte aligned");
    if (int rc = make_stack_train_args(&a, &b, packed, workspace, workspace_bytes, B, T, L, KX, H)) return rc;
    const StackTrainWs W = stack_train_ws_layout(B, T, L, KX, H);
    const StackTrainPacked TP = stack_train_packed_layout(L, KX, H);
    const StackPackedLayout P = stack_packed_layout(L, KX, H);
    char *w = (char *)workspace;
    const int RB = a.RB;
    hipStream_t st = (hipStream_t)stream;
    opnet_pack_dy<<<256, 256, 0, st>>>((const float4 *)dy, (float4 *)(w + W.dyp), (float *)(w + W.dcz),
                                        (long)((W.dcz_end - W.dcz) / 4), B, T, RB);
    // the whole reverse recurrence as ONE persistent launch (seq_xcd_bwd_kernels.hip) for the shapes whose forward runs as one
    // (OPSEQ_XCD_BWD=0 keeps the launch chain: two launches per reverse step)
    const bool use_sxb = seqx_train_shape(B, L, KX, H) && opseq_xcd_supported(L, KX, H) && W.total < ((size_t)1 << 31) &&
                         env_int("OPSEQ_XCD_BWD", 1) != 0;
    unsigned *sxb_status = nullptr;
    if (use_sxb) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        SeqXBArgs sb;
        memset(&sb, 0, sizeof(sb));
        sb.B = B; sb.T = T; sb.L = L; sb.NGT = (B + 3) / 4; sb.RB = RB;
        sb.pk = packed + TP.seqxb;
        sb.ws = w;
        for (int l = 0; l < 2; ++l) {
            sb.g_off[l] = (unsigned)W.g[l < L ? l : 0]; sb.c_off[l] = (unsigned)W.call[l < L ? l : 0];
            sb.ring_off[l] = (unsigned)W.sxb_ring[l];
        }
        sb.dy_off = (unsigned)W.dyp; sb.dxring_off = (unsigned)W.sxb_dxring; sb.dxh_off = (unsigned)W.sxb_dxh;
        sb.status = sxb_status = (unsigned *)(w + W.sx_status);
        sb.force_safe = env_int("OPNET_XCD_SAFE", 0);
        sb.debug = env_int("OPSEQ_XCD_DEBUG", 0);
        seqxb_init<<<512, 256, 0, st>>>(sb);
        {
            std::lock_guard<std::mutex> lock(g_xcd_mu);
            if (!g_xcd_done[dev]) HIP_TRY(hipEventCreateWithFlags(&g_xcd_done[dev], hipEventDisableTiming));
            else HIP_TRY(hipStreamWaitEvent(st, g_xcd_done[dev], 0));
            ProfPair pe{};
            const bool prof = prof_begin(st, &pe);
            if (L == 1) seqx_backward<1><<<XCD_COUNT * XCD_CUS, 256, 0, st>>>(sb);
            else seqx_backward<2><<<XCD_COUNT * XCD_CUS, 256, 0, st>>>(sb);
            if (prof) prof_end(PROF_SEQXB, st, pe);
            HIP_TRY(hipEventRecord(g_xcd_done[dev], st));
        }
    } else {
        const int per = 4 * (H / 16);
        const dim3 ggemm((2 * L - 1) * per, RB < OPNET_MAX_GY ? RB : OPNET_MAX_GY, 1);
        const dim3 gcell(L * (H / 8), RB, 1);
        for (int n = 0; n < T + L - 1; ++n) {
            stack_bwd_cell<<<gcell, 256, 0, st>>>(b, n);
            stack_bwd_gemm<<<ggemm, OPNET_THREADS, 0, st>>>(b, n);
        }
    }
    // weight gradients over the saved histories (same merged GEMM launch as OPNet's)
    const long hs = (long)(H / 4) * 32;
    std::vector<WgradArgs> jobs;
    auto add = [&](const float4 *Pp, long ps, int MQ, const float4 *Q, long qs, int NQ, float *out, int ld,
                   int mvalid, int nvalid, int rowmode) {
        WgradArgs g;
        memset(&g, 0, sizeof(g));
        g.P = Pp; g.p_stride = ps; g.MQ = MQ; g.Q = Q; g.q_stride = qs; g.NQ = NQ;
        g.out = out; g.ld = ld; g.mvalid = mvalid; g.nvalid = nvalid; g.rowmode = rowmode; g.H = H;
        g.T = T; g.RB = RB;
        g.tiles_m = (MQ + 15) / 16;
        jobs.push_back(g);
    };
    for (int l = 0; l < L; ++l) {
        if (!g_ih[l] || !g_hh[l]) return fail(OPNET_EINVAL, "null gradient pointer (layer %d)", l);
        const float4 *hall = (const float4 *)(w + W.hall[l]);
        add(b.layer[l].g, (long)H * 32, H, hall, hs, H / 4, g_hh[l], H, 4 * H, H, 1);                 // da_l x h_l(t-1)
        if (l == 0)
            add(b.layer[l].g, (long)H * 32, H, a.xp, (long)P.nhx[0] * 128, P.nhx[0] * 4, g_ih[0], KX, 4 * H, KX, 1);
        else
            add(b.layer[l].g, (long)H * 32, H, (const float4 *)(w + W.hall[l - 1]) + (long)RB * hs, hs, H / 4,
                g_ih[l], H, 4 * H, H, 1);                                                                // da_l x h_{l-1}(t)
    }
    add(b.dyp, 32, 1, (const float4 *)(w + W.hall[L - 1]) + (long)RB * hs, hs, H / 4, g_head, H, 4, H, 0);
    for (size_t j0 = 0; j0 < jobs.size(); j0 += OPNET_WGRAD_JOBS) {
        WgradBatch wb;
        int ntiles = 0;
        for (int j = 0; j < OPNET_WGRAD_JOBS; ++j) {
            if (j0 + j < jobs.size()) {
                wb.job[j] = jobs[j0 + j];
                wb.job[j].tile_begin = ntiles;
                ntiles += wb.job[j].tiles_m * ((wb.job[j].NQ + 15) / 16);
            } else {
                memset(&wb.job[j], 0, sizeof(WgradArgs));
                wb.job[j].tile_begin = 0x7fffffff;
                wb.job[j].tiles_m = 1;
            }
        }
        wb.abort = sxb_status;    // (null on the launch chain: nothing can give up there)
        const int nj = (int)(jobs.size() - j0 < OPNET_WGRAD_JOBS ? jobs.size() - j0 : OPNET_WGRAD_JOBS);
        if (!wgrad_wave_tiles(wb.job, nj, T, RB, B, (float *)(w + W.wgpart), sxb_status, st, stack_wgrad_max_waves(L, KX, H)))
            opnet_wgrad<<<ntiles, 256, 0, st>>>(wb);
    }
    if (dx0) {
        // dx0 [B*T][KX] = da0 [B*T][4H] . W_ih0 [4H][KX]  (k = 4*unit + gate on both sides) via the tiled GEMM
        float *rows = (float *)(w + W.darows);
        const long nb = (long)B * T * H;
        packed_da_to_rows<<<(unsigned)((nb + 255) / 256 > 8192 ? 8192 : (nb + 255) / 256), 256, 0, st>>>(
            b.layer[0].g, rows, B, T, RB, H);
        ConvArgs c = {};
        c.X = rows; c.Wt = packed + TP.wih0_t; c.bias = nullptr; c.R = nullptr; c.Y = dx0;
        c.N = 1; c.H = 1; c.W = B * T; c.Cin = 4 * H; c.Cout = KX; c.KH = 1; c.KW = 1; c.stride = 1; c.pad = 0;
        c.OH = 1; c.OW = B * T; c.KP = 4 * H; c.relu = 0;
        const int M = B * T;
        // a short sequence (one clip: 3 x 2 tiles of 128 walking K = 4H alone, 160 us): 32 x 32 tiles, K split over the waves
        if (((M + 63) / 64) * ((KX + 63) / 64) < 256 && ((4 * H) & 15) == 0)
            gemm_bias_act_ks<<<dim3((M + 31) / 32, (KX + 31) / 32, 1), 256, 0, st>>>(rows, packed + TP.wih0_t, nullptr, dx0, M, KX, 4 * H, 0);
        else
            launch_conv_tiled(c, M, st);
    }
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opseq_slot_embed_relu_f32(const float *x, const float *W, float *out, long ntok, int nslots_out,
                                         int F, void *stream)
{
    if (!x || !W || !out) return fail(OPNET_EINVAL, "null pointer");
    if (ntok <= 0 || F <= 0 || (nslots_out != 1 && nslots_out != 15))
        return fail(OPNET_ESHAPE, "ntok=%ld F=%d nslots_out=%d", ntok, F, nslots_out);
    const long n = ntok * nslots_out * F;
    slot_embed_relu<<<(unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        x, W, out, ntok, nslots_out, F);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

static int slot_embed_bwd_blocks(long nrows)
{
    long nb = (nrows + 15) / 16;                        // >= 16 rows a workgroup
    return (int)(nb > 1024 ? 1024 : nb < 1 ? 1 : nb);
}
extern "C" size_t opseq_slot_embed_bwd_workspace_bytes(long ntok, int nslots_out, int F)
{
    if (ntok <= 0 || F <= 0 || (nslots_out != 1 && nslots_out != 15)) return 0;
    return (size_t)slot_embed_bwd_blocks(ntok * nslots_out) * F * 5 * sizeof(float);
}
extern "C" int opseq_slot_embed_relu_bwd_ws_f32(const float *x, const float *out, const float *dout, float *dW, long ntok,
                                                int nslots_out, int F, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !out || !dout || !dW || !workspace) return fail(OPNET_EINVAL, "null pointer");
    if (ntok <= 0 || F <= 0 || (nslots_out != 1 && nslots_out != 15)) return fail(OPNET_ESHAPE, "bad shape");
    if (workspace_bytes < opseq_slot_embed_bwd_workspace_bytes(ntok, nslots_out, F)) return fail(OPNET_EWORKSPACE, "workspace too small");
    const long nrows = ntok * nslots_out;
    const int nb = slot_embed_bwd_blocks(nrows);
    const long rpb = (nrows + nb - 1) / nb;
    slot_embed_relu_bwd_part<<<nb, 256, 0, (hipStream_t)stream>>>(x, out, dout, (float *)workspace, nrows, nslots_out, F, rpb);
    slot_embed_bwd_final<<<(F * 5 + 63) / 64, 256, 0, (hipStream_t)stream>>>((const float *)workspace, dW, nb, F);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opseq_slot_embed_relu_bwd_f32(const float *x, const float *out, const float *dout, float *dW, long ntok,
                                             int nslots_out, int F, void *stream)
{
    if (!x || !out || !dout || !dW) return fail(OPNET_EINVAL, "null pointer");
    if (ntok <= 0 || F <= 0 || (nslots_out != 1 && nslots_out != 15)) return fail(OPNET_ESHAPE, "bad shape");
    slot_embed_relu_bwd<<<F, 256, 0, (hipStream_t)stream>>>(x, out, dout, dW, ntok, nslots_out, F);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

static int check_encoder(long S, int E, int nhead, int ffn)
{
    if (S <= 0 || S > 0x7fffffffL) return fail(OPNET_ESHAPE, "S=%ld out of range", S);
    if (E <= 0 || (E & 15) || E > 64 * LN_MAX_PER_LANE) return fail(OPNET_ESHAPE, "E=%d must be a multiple of 16, <= %d", E, 64 * LN_MAX_PER_LANE);
    if (nhead <= 0 || E % nhead) return fail(OPNET_ESHAPE, "E=%d not divisible by nhead=%d", E, nhead);
    const int hd = E / nhead;
    if ((hd & 15) || hd > 16 * ATT_MAX_HEX) return fail(OPNET_ESHAPE, "head dim %d must be a multiple of 16, <= %d", hd, 16 * ATT_MAX_HEX);
    if (ffn <= 0 || (ffn & 15)) return fail(OPNET_ESHAPE, "ffn=%d must be a multiple of 16", ffn);
    return OPNET_OK;
}

extern "C" size_t opseq_encoder_workspace_bytes(long S, int E, int nhead, int ffn)
{
    if (check_encoder(S, E, nhead, ffn)) return 0;
    return align_up((size_t)S * (3 * E + 3 * E + ffn) * sizeof(float), 256);
}

/* the attention core of nn.MultiheadAttention over ONE sequence: qkv [S][3E] (q | k | v, already projected) ->
 * out [S][E] = concat_h softmax(q_h k_h^T / sqrt(hd)) v_h */
extern "C" size_t opseq_attention_workspace_bytes(long S, int E, int nhead)
{
    if (S <= 0 || E <= 0 || nhead <= 0) return 0;
    return align_up(attention_scratch_bytes(S, E, nhead, 4), 256);
}

extern "C" int opseq_attention_f32(const float *qkv, float *out, long S, int E, int nhead, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    if (!qkv || !out) return fail(OPNET_EINVAL, "null pointer");
    if (workspace && !aligned16(workspace)) return fail(OPNET_EINVAL, "workspace must be 16-byte aligned");
    if (!aligned16(qkv) || !aligned16(out)) return fail(OPNET_EINVAL, "qkv / out must be 16-byte aligned");
    if (S <= 0 || S > 0x7fffffffL || E <= 0 || nhead <= 0 || E % nhead) return fail(OPNET_ESHAPE, "bad attention shape");
    const int hd = E / nhead;
    if ((hd & 15) || hd > 128) return fail(OPNET_ESHAPE, "head size %d: must be a multiple of 16, <= 128", hd);
    launch_attention(qkv, out, (int)S, 1, E, nhead, hd, workspace, workspace ? workspace_bytes : 0, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

/* linear1 -> ReLU -> linear2 as ONE kernel (csrc/ffn_kernels.hip): y [M][E] = relu(x W1^T + b1) W2^T + b2, bit-identical to the two
 * products on conv2d_nhwc_glds.  E == 256, ffn a multiple of 128, x / y / weights 16-byte aligned, x below 2 GiB. */
// the eight-wave tiles of csrc/ffn_kernels.hip need 144 KB of LDS in one workgroup (gfx950: 160 KB per CU)
static bool w8_device_ok()
{
    static std::atomic<int> cached[64];          // 0 = unknown, 1 = yes, 2 = no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    int c = cached[dev].load();
    if (c == 0) {
        int lds = 0;
        c = (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess &&
             (size_t)lds >= (size_t)FFN_W8_LDS_F4 * sizeof(float4)) ? 1 : 2;
        cached[dev].store(c);
    }
    return c == 1;
}

static bool ffn_fused_shape(long M, int E, int ffn)
{
    if (!w8_device_ok()) return false;
    return E == 256 && ffn > 0 && (ffn & 127) == 0 && M > 0 && M * (long)E * 4 < (1L << 31) && (long)ffn * E * 4 < (1L << 31);
}

static int launch_ffn_fused(const float *x, const float *w1, const float *b1, const float *w2, const float *b2, float *y, long M,
                            int ffn, hipStream_t st)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const int cus = xcd_device_cus(dev);
    if (cus <= 0) return fail(OPNET_EHIP, "no CU count for device %d", dev);
    FfnArgs a = {};
    a.X = x; a.W1 = w1; a.b1 = b1; a.W2 = w2; a.b2 = b2; a.Y = y; a.M = (int)M; a.F = ffn; a.m_begin = 0;
    unsigned grid = 0;
    ffn_w8_plan(M, cus, &a, &grid);
    ffn_fused_w8<<<grid, 512, 0, st>>>(a);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

/* a plain K = 256 product Y [M][N] = act(X W^T + b) on the resident-token tile of csrc/ffn_kernels.hip (gemm_k256_w8: the token rows
 * read once, private weight rings; the same bits as conv2d_nhwc_glds, 9-20 % less time from 8 192 rows on), else the tiled conv kernel */
static int launch_gemm_k256(const ConvArgs &c, long M, hipStream_t st)
{
    const bool plain = c.KH == 1 && c.KW == 1 && c.stride == 1 && c.pad == 0 && !c.R && !c.X2 && c.ksplit <= 1 && c.XS == 0 && c.WS == 0 &&
                       c.YS == 0 && c.RH == 0 && c.N == 1 && c.H == 1;
    if (plain && c.Cin == 256 && c.KP == 256 && (c.Cout & 127) == 0 && M >= 8192 && M * 256L * 4 < (1L << 31) &&
        (!c.bias || aligned16(c.bias)) && aligned16(c.X) && aligned16(c.Wt) && aligned16(c.Y) && env_int("OPSEQ_GEMM_W8", 1) && w8_device_ok()) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        const int cus = xcd_device_cus(dev);
        if (cus > 0) {
            FfnArgs plan = {};
            unsigned grid = 0;
            ffn_w8_plan(M, cus, &plan, &grid);
            Gemm256Args g = {c.X, c.Wt, c.bias, c.Y, (int)M, c.Cout, c.relu, plan.n_full, plan.tail_frags};
            gemm_k256_w8<<<grid, 512, 0, st>>>(g);
            HIP_TRY(hipGetLastError());
            return OPNET_OK;
        }
    }
    launch_conv_tiled(c, M, st);
    return OPNET_OK;
}

extern "C" int opseq_ffn_fused_supported(long M, int E, int ffn) { return ffn_fused_shape(M, E, ffn) ? 1 : 0; }

/* host logic only (no device call): the tile plan ffn_fused_w8 / gemm_k256_w8 run M token rows with on `cus` compute units -
 * workgroups 0 .. n_full - 1 own 64 rows each, the remaining grid - n_full workgroups 16 * tail_frags rows each */
extern "C" int opseq_ffn_fused_plan(long M, int cus, int *n_full, int *tail_frags, unsigned *grid)
{
    if (M <= 0 || cus <= 0 || !n_full || !tail_frags || !grid) return fail(OPNET_EINVAL, "bad plan request");
    FfnArgs a = {};
    ffn_w8_plan(M, cus, &a, grid);
    *n_full = a.n_full;
    *tail_frags = a.tail_frags;
    return OPNET_OK;
}

extern "C" int opseq_ffn_fused_f32(const float *x, const float *l1_w, const float *l1_b, const float *l2_w, const float *l2_b,
                                   float *y, long M, int E, int ffn, void *stream)
{
    if (!x || !l1_w || !l1_b || !l2_w || !l2_b || !y) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(x) || !aligned16(y) || !aligned16(l1_w) || !aligned16(l2_w) || !aligned16(l1_b) || !aligned16(l2_b))
        return fail(OPNET_EINVAL, "x / y / weights / biases must be 16-byte aligned");
    if (!ffn_fused_shape(M, E, ffn)) return fail(OPNET_ESHAPE, "fused feed-forward: M=%ld E=%d ffn=%d (E == 256, ffn %% 128 == 0, x < 2 GiB)", M, E, ffn);
    return launch_ffn_fused(x, l1_w, l1_b, l2_w, l2_b, y, M, ffn, (hipStream_t)stream);
}

/* one post-LN nn.TransformerEncoderLayer (eval), in place on z [nseg * S][E]: nseg independent sequences of S tokens each.
 * Token-wise stages (the four products, the two layer norms) run over all nseg * S rows at once; attention stays inside a
 * sequence.  EVERY kernel choice is made from ONE sequence's shape (S), never from nseg * S, and every kernel computes an
 * output row from its own input row in an order that does not depend on where the row sits - so each sequence comes out
 * bit-identical to that sequence run alone (nseg = 1), which is the contract serving.ReasonerServer relies on. */
static int encoder_layer(float *z, const float *in_w, const float *in_b, const float *out_w, const float *out_b, const float *l1_w,
                         const float *l1_b, const float *l2_w, const float *l2_b, const float *n1_w, const float *n1_b,
                         const float *n2_w, const float *n2_b, void *workspace, size_t workspace_bytes, long S, int nseg, int E,
                         int nhead, int ffn, void *stream, bool batched = false)
{
    if (int rc = check_encoder(S, E, nhead, ffn)) return rc;
    if (nseg <= 0 || S * nseg > 0x7fffffffL) return fail(OPNET_ESHAPE, "n_seg=%d x S=%ld out of range", nseg, S);
    if (!z || !in_w || !in_b || !out_w || !out_b || !l1_w || !l1_b || !l2_w || !l2_b || !n1_w || !n1_b || !n2_w ||
        !n2_b || !workspace)
        return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(z) || !aligned16(workspace) || !aligned16(in_w) || !aligned16(out_w) || !aligned16(l1_w) || !aligned16(l2_w))
        return fail(OPNET_EINVAL, "z / workspace / weight matrices must be 16-byte aligned");
    const long St = S * nseg;
    // one pass addresses its buffers with the offsets a single sequence's kernels use (32-bit in the tiled GEMM): keep the
    // largest (hid [St][ffn]) below 2 GiB so that the kernel choice cannot differ from the lone sequence's
    if (nseg > 1 && (St * (long)ffn * 4 >= (1L << 31) || St * 3L * E * 4 >= (1L << 31)))
        return fail(OPNET_ESHAPE, "n_seg=%d x S=%ld tokens exceed one pass (2 GiB buffers): split the requests", nseg, S);
    if (workspace_bytes < opseq_encoder_workspace_bytes(St, E, nhead, ffn)) return fail(OPNET_EWORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *qkv = (float *)workspace;
    float *att = qkv + (size_t)St * 3 * E;
    float *proj = att + (size_t)St * E;
    float *z1 = proj + (size_t)St * E;
    float *hid = z1 + (size_t)St * E;
    const int M = (int)St, hd = E / nhead;
    // M rows are computed; Ms picks the token-wise kernels: ONE sequence's rows (each sequence then comes out bit-identical to its
    // lone forward), or - batched: the throughput form of a served pass - all of them (128 x 128 LDS-DMA tiles once they fill the
    // chip: another K order, results agree with the lone forward to rounding)
    const int Ms = batched ? M : (int)S;
    // C[M][N] = act(A[M][K] W[N][K]^T + b) = a 1x1 "convolution" over M pixels: the LDS-staged tiled kernel
    auto gemm = [&](const float *A, const float *Wt, const float *b, float *C, int N, int K, int act) -> int {
        if ((long)((Ms + 127) / 128) * ((N + 127) / 128) < 256) {   // too few 128-tiles to fill the chip
            // ... and with fewer than 256 64-tiles (one clip: S = 300) even those leave CUs idle behind long serial K walks:
            // 32 x 32 tiles with K split over the workgroup's waves
            if ((K & 63) == 0 && (long)((Ms + 63) / 64) * ((N + 63) / 64) < 256 && env_int("OPSEQ_GEMM_KS", 1))
                gemm_bias_act_ks<<<dim3((M + 31) / 32, (N + 31) / 32, 1), 256, 0, st>>>(A, Wt, b, C, M, N, K, act);
            else
                gemm_bias_act<<<dim3((M + 63) / 64, (N + 63) / 64, 1), 256, 0, st>>>(A, Wt, b, C, M, N, K, act);
            return OPNET_OK;
        }
        ConvArgs c = {};
        c.X = A; c.Wt = Wt; c.bias = b; c.R = nullptr; c.Y = C;
        c.N = 1; c.H = 1; c.W = M; c.Cin = K; c.Cout = N; c.KH = 1; c.KW = 1; c.stride = 1; c.pad = 0;
        c.OH = 1; c.OW = M; c.KP = K; c.relu = act;
        if (batched) return launch_gemm_k256(c, M, st);     // (K = 256 products of a throughput pass: the resident-token tile, same bits)
        launch_conv_tiled(c, M, st);
        return OPNET_OK;
    };
    if (int rc = gemm(z, in_w, in_b, qkv, 3 * E, E, 0)) return rc;
    {
        ProfPair pe{};
        const bool prof = prof_begin(st, &pe);
        launch_attention(qkv, att, (int)S, nseg, E, nhead, hd, hid, (size_t)St * ffn * sizeof(float), st);   // hid is free until the FFN
        if (prof) prof_end(PROF_ATTN, st, pe);
    }
    if (int rc = gemm(att, out_w, out_b, proj, E, E, 0)) return rc;
    add_layernorm<<<(M + 3) / 4, 256, 0, st>>>(z, proj, n1_w, n1_b, z1, M, E, 1e-5f);
    // the throughput form runs the feed-forward block as one kernel (csrc/ffn_kernels.hip: the [M][ffn] activations never leave the
    // CU; the same bits as the two products below); the exact form keeps the lone request's K-split products
    if (batched && ffn_fused_shape(M, E, ffn) && aligned16(l1_b) && aligned16(l2_b) && env_int("OPSEQ_FFN_FUSED", 1)) {
        ProfPair pe{};
        const bool prof = prof_begin(st, &pe);
        if (int rc = launch_ffn_fused(z1, l1_w, l1_b, l2_w, l2_b, proj, M, ffn, st)) return rc;
        if (prof) prof_end(PROF_FFN, st, pe);
    } else {
        if (int rc = gemm(z1, l1_w, l1_b, hid, ffn, E, 1)) return rc;
        if (int rc = gemm(hid, l2_w, l2_b, proj, E, ffn, 0)) return rc;
    }
    add_layernorm<<<(M + 3) / 4, 256, 0, st>>>(z1, proj, n2_w, n2_b, z, M, E, 1e-5f);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opseq_encoder_layer_f32(float *z, const float *in_w, const float *in_b, const float *out_w,
                                       const float *out_b, const float *l1_w, const float *l1_b, const float *l2_w,
                                       const float *l2_b, const float *n1_w, const float *n1_b, const float *n2_w,
                                       const float *n2_b, void *workspace, size_t workspace_bytes, long S, int E,
                                       int nhead, int ffn, void *stream)
{
    return encoder_layer(z, in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b, workspace, workspace_bytes, S, 1,
                         E, nhead, ffn, stream);
}

/* the same layer over n_seg independent sequences of S tokens each, z [n_seg * S][E] (requests served in one pass: each
 * sequence attends only to itself and comes out bit-identical to opseq_encoder_layer_f32 on it alone); workspace >=
 * opseq_encoder_workspace_bytes(n_seg * S, ...) */
extern "C" int opseq_encoder_layer_segmented_f32(float *z, const float *in_w, const float *in_b, const float *out_w,
                                                 const float *out_b, const float *l1_w, const float *l1_b, const float *l2_w,
                                                 const float *l2_b, const float *n1_w, const float *n1_b, const float *n2_w,
                                                 const float *n2_b, void *workspace, size_t workspace_bytes, long S, int n_seg,
                                                 int E, int nhead, int ffn, void *stream)
{
    return encoder_layer(z, in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b, workspace, workspace_bytes, S,
                         n_seg, E, nhead, ffn, stream);
}

/* the same n_seg independent sequences with the token-wise products chosen by ALL n_seg * S rows (the throughput form of a served
 * pass: 128 x 128 LDS-DMA tiles once they fill the chip instead of one sequence's K-split 32 x 32 tiles); attention as above.
 * Each sequence agrees with opseq_encoder_layer_f32 on it alone to rounding (another summation order), not bit for bit. */
extern "C" int opseq_encoder_layer_batched_f32(float *z, const float *in_w, const float *in_b, const float *out_w,
                                               const float *out_b, const float *l1_w, const float *l1_b, const float *l2_w,
                                               const float *l2_b, const float *n1_w, const float *n1_b, const float *n2_w,
                                               const float *n2_b, void *workspace, size_t workspace_bytes, long S, int n_seg,
                                               int E, int nhead, int ffn, void *stream)
{
    return encoder_layer(z, in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b, workspace, workspace_bytes, S,
                         n_seg, E, nhead, ffn, stream, true);
}

#include "enc_train_abi.hip"

// ------------------------------------------------------------------------------------------------
// detector backbone primitives (NHWC fp32)
// ------------------------------------------------------------------------------------------------
static unsigned ew_blocks(long n) { return (unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256); }

static int conv_args(ConvArgs *a, const float *x, const float *w, const float *bias, const float *residual, float *y, int N, int H, int W,
                     int Cin, int Cout, int KH, int KW, int stride, int pad, int KP, int relu)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
        return fail(OPNET_ESHAPE, "bad conv shape (Cin must be a multiple of 4)");
    if ((KP & 15) || KP < KH * KW * Cin) return fail(OPNET_ESHAPE, "KP=%d must be a multiple of 16 >= KH*KW*Cin=%d", KP, KH * KW * Cin);
    *a = ConvArgs{};
    a->X = x; a->Wt = w; a->bias = bias; a->R = residual; a->Y = y;
    a->N = N; a->H = H; a->W = W; a->Cin = Cin; a->Cout = Cout; a->KH = KH; a->KW = KW; a->stride = stride; a->pad = pad;
    a->OH = (H + 2 * pad - KH) / stride + 1;
    a->OW = (W + 2 * pad - KW) / stride + 1;
    a->KP = KP; a->relu = relu;
    if (a->OH <= 0 || a->OW <= 0) return fail(OPNET_ESHAPE, "empty conv output");
    return OPNET_OK;
}

// LDS-staged 128 x {128, 64} tiles when they give >= one workgroup per CU
static bool conv_fills_chip(const ConvArgs &a, long M)
{
    const long t128 = ((M + 127) / 128) * ((a.Cout + 127) / 128), t64 = ((M + 127) / 128) * ((a.Cout + 63) / 64);
    return (a.Cout > 64 && t128 >= 256) || (a.Cout <= 64 && t64 >= 256);
}

// The deep, spatially small layers of ONE frame (25x34 .. 50x68 maps; the 1000-row FCs) give 28 .. 216 of those tiles: with scratch
// for the partial sums their K is split over blockIdx.z so that ~2.5 workgroups per CU run the LDS-DMA kernel (>= 16 K steps each).
// -> number of slices (1 = no split) and K steps per slice
static int conv_split_plan(const ConvArgs &a, long M, int *ksteps)
{
    *ksteps = a.KP >> 4;
    if (conv_fills_chip(a, M) || (a.Cin & 15) || (a.Cout & 3) || a.KP != a.KH * a.KW * a.Cin) return 1;
    if ((long)a.N * a.H * a.W * a.Cin * 4 >= (1L << 31) || (long)a.Cout * a.KP * 4 >= (1L << 31)) return 1;
    const int nhex = a.KP >> 4;
    // split layers run on 128 x 64 tiles (round 5: twice the tiles = half the slices, i.e. half the partial-sum traffic and fewer
    // K-loop prologues: the one-frame call 240 -> 246 frames/s with two in flight; OPNET_SPLIT_BN128=1 = the wide tiles of round 4)
    const bool bn64 = a.Cout <= 64 || env_int("OPNET_SPLIT_BN128", 0) == 0;
    const long tiles = ((M + 127) / 128) * (bn64 ? (a.Cout + 63) / 64 : (a.Cout + 127) / 128);
    long S = (640 + tiles - 1) / tiles;      // (swept on the one-frame call: 400 .. 1024 workgroups x >= 8 .. 32 steps; this is the minimum)
    if (S > nhex / 16) S = nhex / 16;
    if (S > 32) S = 32;
    if (S < 2) return 1;
    const int per = (int)((nhex + S - 1) / S);
    *ksteps = per;
    return (nhex + per - 1) / per;          // no empty slice
}

static void launch_conv(const ConvArgs &a, long M, hipStream_t st)
{
    // Fewer tiles than CUs: the LDS-DMA kernel still wins from ~100 tiles up (measured on one frame's layer3 1x1s, 216 tiles x 16 K
    // steps: 25 us against 40); below that the un-staged 64 x 64 tiles keep more of the chip busy
    const long tiles = ((M + 127) / 128) * (a.Cout > 64 ? (a.Cout + 127) / 128 : (a.Cout + 63) / 64);
    if (conv_fills_chip(a, M) || (!(a.Cin & 15) && tiles >= 100)) launch_conv_tiled(a, M, st);
    else conv2d_nhwc<<<dim3((unsigned)((M + 63) / 64), (a.Cout + 63) / 64, 1), 256, 0, st>>>(a);
}

extern "C" int opdet_conv2d_f32(const float *x, const float *w, const float *bias, const float *residual, float *y,
                                int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int KP,
                                int relu, void *stream)
{
    if (!x || !w || !y) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(x) || !aligned16(w) || !aligned16(y)) return fail(OPNET_EINVAL, "x / w / y must be 16-byte aligned");
    ConvArgs a;
    if (int rc = conv_args(&a, x, w, bias, residual, y, N, H, W, Cin, Cout, KH, KW, stride, pad, KP, relu)) return rc;
    launch_conv(a, (long)N * a.OH * a.OW, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" size_t opdet_conv2d_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int KP)
{
    ConvArgs a;
    if (conv_args(&a, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, KH, KW, stride, pad, KP, 0)) return 0;
    const long M = (long)N * a.OH * a.OW;
    int ksteps;
    const int S = conv_split_plan(a, M, &ksteps);
    return S > 1 ? (size_t)S * M * Cout * 4 : 0;
}

extern "C" int opdet_conv2d_ws_f32(const float *x, const float *w, const float *bias, const float *residual, float *y,
                                   int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int KP,
                                   int relu, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !w || !y) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || !aligned16(workspace) || !aligned16(bias) || !aligned16(residual))
        return fail(OPNET_EINVAL, "x / w / y / bias / residual / workspace must be 16-byte aligned");
    ConvArgs a;
    if (int rc = conv_args(&a, x, w, bias, residual, y, N, H, W, Cin, Cout, KH, KW, stride, pad, KP, relu)) return rc;
    const long M = (long)N * a.OH * a.OW;
    int ksteps;
    const int S = conv_split_plan(a, M, &ksteps);
    hipStream_t st = (hipStream_t)stream;
    if (S <= 1) {
        launch_conv(a, M, st);
    } else {
        const size_t need = (size_t)S * M * Cout * 4;
        if (!workspace || workspace_bytes < need) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", workspace_bytes, need);
        a.P = (float *)workspace; a.ksplit = S; a.ksteps = ksteps;
        const unsigned gx = (unsigned)((M + 127) / 128);
        if (Cout > 64 && env_int("OPNET_SPLIT_BN128", 0) != 0) conv2d_nhwc_glds<128, 3><<<dim3(gx, (Cout + 127) / 128, S), 256, 0, st>>>(a);
        else conv2d_nhwc_glds<64, 3><<<dim3(gx, (Cout + 63) / 64, S), 256, 0, st>>>(a);
        const long n4 = M * Cout / 4;
        conv_splitk_reduce<<<(unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256), 256, 0, st>>>(a.P, S, M, Cout, bias, residual, y,
                                                                                                    relu);
    }
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// ---- Winograd F(2 x 2, 3 x 3) for stride-1 3 x 3 convs (csrc/wino_kernels.hip) ------------------------------------------------
// workspace: the 16 position planes of the transformed input and of the products for a chunk of TILES (a tile = 2 x 2 output pixels
// of one image; any contiguous range of the pass's tiles is a chunk: the input is only read, the output written tile by tile); a
// pass whose planes exceed OPDET_WINO_WS_MB runs in chunks
static long wino_chunk_tiles(int N, int H, int W, int Cin, int Cout)
{
    const long tiles = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
    const size_t per = (size_t)16 * (size_t)(Cin + Cout) * 4;                 // bytes of the planes per tile
    const size_t cap = (size_t)env_int("OPDET_WINO_WS_MB", 256) << 20;
    long n = (long)(cap / per);
    // (one GEMM batch addresses a position plane through a 2 GiB buffer descriptor)
    const long lim_x = ((1L << 31) - 1) / ((long)(Cin > Cout ? Cin : Cout) * 4);
    if (n > lim_x) n = lim_x;
    n &= ~127L;                                                                // whole 128-row GEMM tiles
    if (n < 128) n = 128;
    return n > tiles ? tiles : n;
}
static int check_wino(int N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 15) || (Cout & 3))
        return fail(OPNET_ESHAPE, "Winograd conv: Cin must be a multiple of 16 and Cout of 4");
    return OPNET_OK;
}
extern "C" size_t opdet_wino_weights_bytes(int Cin, int Cout) { return (size_t)16 * Cin * Cout * 4; }
/* the transformed weights U [16][Cout][Cin] of a 3 x 3 conv from its packed weight w [Cout][KP] (k = (ky * 3 + kx) * Cin + ci); once per weight set */
extern "C" int opdet_wino_weights_f32(const float *w, float *u, int Cin, int Cout, int KP, void *stream)
{
    if (!w || !u) return fail(OPNET_EINVAL, "null pointer");
    if (KP < 9 * Cin) return fail(OPNET_ESHAPE, "KP=%d < 9 * Cin", KP);
    const long n = (long)Cin * Cout;
    wino_weights<<<(unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, u, Cin, Cout, KP);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}
extern "C" size_t opdet_conv2d_wino_workspace_bytes(int N, int H, int W, int Cin, int Cout)
{
    if (check_wino(N, H, W, Cin, Cout)) return 0;
    return (size_t)16 * wino_chunk_tiles(N, H, W, Cin, Cout) * (size_t)(Cin + Cout) * 4;
}
/* y [N, H, W, Cout] = act(conv3x3(x [N, H, W, Cin], stride 1, pad 1) + bias) through Winograd F(2 x 2, 3 x 3): u from opdet_wino_weights_f32 */
extern "C" int opdet_conv2d_wino_f32(const float *x, const float *u, const float *bias, float *y, int N, int H, int W, int Cin, int Cout,
                                     int relu, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_wino(N, H, W, Cin, Cout)) return rc;
    if (!x || !u || !y || !workspace) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(x) || !aligned16(u) || !aligned16(y) || !aligned16(workspace) || !aligned16(bias))
        return fail(OPNET_EINVAL, "x / u / y / bias / workspace must be 16-byte aligned");
    if (workspace_bytes < opdet_conv2d_wino_workspace_bytes(N, H, W, Cin, Cout)) return fail(OPNET_EWORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const long tiles = (long)N * ((H + 1) / 2) * ((W + 1) / 2), ct = wino_chunk_tiles(N, H, W, Cin, Cout);
    float *V = (float *)workspace, *Mm = V + (size_t)16 * ct * Cin;
    for (long t0 = 0; t0 < tiles; t0 += ct) {
        const long NT = tiles - t0 < ct ? tiles - t0 : ct;
        const long ni = NT * (Cin / 4), no = NT * (Cout / 4);
        wino_input<<<(unsigned)((ni + 255) / 256 > 16384 ? 16384 : (ni + 255) / 256), 256, 0, st>>>(x, V, t0, NT, H, W, Cin);
        ConvArgs g = {};
        g.X = V; g.Wt = u; g.Y = Mm;
        g.N = 1; g.H = 1; g.W = (int)NT; g.Cin = Cin; g.Cout = Cout; g.KH = 1; g.KW = 1; g.stride = 1; g.pad = 0; g.OH = 1; g.OW = (int)NT; g.KP = Cin;
        g.bsx = NT * Cin; g.bsw = (long)Cout * Cin; g.bsy = NT * Cout;
        conv2d_nhwc_glds<64, 3><<<dim3((unsigned)((NT + 127) / 128), (Cout + 63) / 64, 16), 256, 0, st>>>(g);
        wino_output<<<(unsigned)((no + 255) / 256 > 16384 ? 16384 : (no + 255) / 256), 256, 0, st>>>(Mm, bias, y, t0, NT, H, W, Cout, relu);
    }
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// does launch_conv() run this shape on conv2d_nhwc_glds (the only kernel that honours ConvArgs.RH / RW)?
static bool conv_uses_glds(const ConvArgs &a, long M)
{
    const long tiles = ((M + 127) / 128) * (a.Cout > 64 ? (a.Cout + 127) / 128 : (a.Cout + 63) / 64);
    if (!(conv_fills_chip(a, M) || (!(a.Cin & 15) && tiles >= 100))) return false;
    if ((long)a.N * a.H * a.W * a.Cin * 4 >= (1L << 31) || (long)a.Cout * a.KP * 4 >= (1L << 31)) return false;
    return (a.Cin & 15) == 0;
}

/* FeaturePyramidNetwork's top-down step in one call:  y = conv(x) + bias + nearest_upsample(top)  with top [N, TH, TW, Cout] and y the
 * conv's [N, OH, OW, Cout] (F.interpolate(top, size = (OH, OW), mode "nearest"): source index floor(dst * T / O)).  Where the shape runs on
 * the LDS-DMA kernel un-split, the addition happens in that kernel's epilogue (one pass over y instead of three); otherwise the conv
 * is followed by the in-place upsample_add launch.  Same fp32 operations in the same order either way. */
extern "C" int opdet_conv2d_up_f32(const float *x, const float *w, const float *bias, const float *top, float *y, int N, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int stride, int pad, int KP, int TH, int TW, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    if (!x || !w || !y || !top) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || !aligned16(workspace) || !aligned16(bias) || !aligned16(top))
        return fail(OPNET_EINVAL, "x / w / y / bias / top / workspace must be 16-byte aligned");
    if (TH <= 0 || TW <= 0) return fail(OPNET_ESHAPE, "bad top map size");
    ConvArgs a;
    if (int rc = conv_args(&a, x, w, bias, nullptr, y, N, H, W, Cin, Cout, KH, KW, stride, pad, KP, 0)) return rc;
    const long M = (long)N * a.OH * a.OW;
    int ksteps;
    hipStream_t st = (hipStream_t)stream;
    if (conv_split_plan(a, M, &ksteps) <= 1 && conv_uses_glds(a, M) && (Cout & 3) == 0) {
        a.R = top; a.RH = TH; a.RW = TW;
        launch_conv(a, M, st);
        HIP_TRY(hipGetLastError());
        return OPNET_OK;
    }
    if (int rc = opdet_conv2d_ws_f32(x, w, bias, nullptr, y, N, H, W, Cin, Cout, KH, KW, stride, pad, KP, 0, workspace, workspace_bytes, stream))
        return rc;
    upsample_add<<<ew_blocks(M * Cout), 256, 0, st>>>(y, top, y, N, a.OH, a.OW, Cout, TH, TW);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

/* conv3 and the downsample branch of a stage's first bottleneck as ONE product (torchvision Bottleneck.forward:
 * relu(bn3(conv3(out)) + downsample(x))):  y = act(W[:, :Cin] . x + W[:, Cin:] . x2[sampled at stride2] + bias), both 1 x 1, x [N, H, W, Cin]
 * at stride 1, x2 [N, H2, W2, Cin2] with (H2 - 1) / stride2 + 1 == H (same for W), w [Cout][Cin + Cin2], bias = the two folded biases
 * added.  Saves writing the downsample's output and reading it back as the residual.  Cin, Cin2 multiples of 16 and the shape on the
 * LDS-DMA kernel (un-split or K-split: _dual_workspace_bytes); otherwise OPNET_ESHAPE - the caller then runs the two convs. */
static int conv_dual_args(ConvArgs *a, const float *x, const float *x2, const float *w, const float *bias, float *y, int N, int H, int W,
                          int Cin, int H2, int W2, int Cin2, int stride2, int Cout, int relu)
{
    if (stride2 <= 0 || Cin2 <= 0 || (Cin & 15) || (Cin2 & 15)) return fail(OPNET_ESHAPE, "dual conv: Cin, Cin2 multiples of 16");
    if (H2 <= 0 || W2 <= 0 || (H2 - 1) / stride2 + 1 != H || (W2 - 1) / stride2 + 1 != W)
        return fail(OPNET_ESHAPE, "dual conv: the second source sampled at stride %d does not give %d x %d", stride2, H, W);
    if (int rc = conv_args(a, x, w, bias, nullptr, y, N, H, W, Cin, Cout, 1, 1, 1, 0, Cin + Cin2, relu)) return rc;
    if ((long)N * H2 * W2 * Cin2 * 4 >= (1L << 31)) return fail(OPNET_ESHAPE, "dual conv: second source of 2 GiB or more");
    a->X2 = x2; a->Cin2 = Cin2; a->H2 = H2; a->W2 = W2; a->stride2 = stride2;
    return OPNET_OK;
}

// K-split plan of a dual product: conv_split_plan() with the K of both sources (it tests KP == KH KW Cin)
static int conv_dual_split_plan(const ConvArgs &a, long M, int *ksteps)
{
    ConvArgs t = a;
    t.Cin = a.Cin + a.Cin2;
    t.X2 = nullptr;
    return conv_split_plan(t, M, ksteps);
}

extern "C" long long opdet_conv2d_dual_workspace_bytes(int N, int H, int W, int Cin, int H2, int W2, int Cin2, int stride2, int Cout)
{
    ConvArgs a;
    if (conv_dual_args(&a, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, H2, W2, Cin2, stride2, Cout, 0)) return -1;
    const long M = (long)N * H * W;
    ConvArgs t = a;
    t.Cin = Cin + Cin2;
    if (!conv_uses_glds(t, M)) { fail(OPNET_ESHAPE, "dual conv: shape does not run on the LDS-DMA kernel"); return -1; }
    int ksteps;
    const int S = conv_dual_split_plan(a, M, &ksteps);
    return S > 1 ? (long long)S * M * Cout * 4 : 0;
}

extern "C" int opdet_conv2d_dual_f32(const float *x, const float *x2, const float *w, const float *bias, float *y, int N, int H, int W,
                                     int Cin, int H2, int W2, int Cin2, int stride2, int Cout, int relu, void *workspace,
                                     size_t workspace_bytes, void *stream)
{
    if (!x || !x2 || !w || !y) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(x) || !aligned16(x2) || !aligned16(w) || !aligned16(y) || !aligned16(workspace) || !aligned16(bias))
        return fail(OPNET_EINVAL, "x / x2 / w / y / bias / workspace must be 16-byte aligned");
    ConvArgs a;
    if (int rc = conv_dual_args(&a, x, x2, w, bias, y, N, H, W, Cin, H2, W2, Cin2, stride2, Cout, relu)) return rc;
    const long M = (long)N * H * W;
    ConvArgs t = a;
    t.Cin = Cin + Cin2;
    if (!conv_uses_glds(t, M)) return fail(OPNET_ESHAPE, "dual conv: shape does not run on the LDS-DMA kernel");
    int ksteps;
    const int S = conv_dual_split_plan(a, M, &ksteps);
    hipStream_t st = (hipStream_t)stream;
    const unsigned gx = (unsigned)((M + 127) / 128);
    if (S <= 1) {
        if (Cout > 64 && !conv_prefers_bn64(t, M)) conv2d_nhwc_glds<128, 3><<<dim3(gx, (Cout + 127) / 128, 1), 256, 0, st>>>(a);
        else conv2d_nhwc_glds<64, 3><<<dim3(gx, (Cout + 63) / 64, 1), 256, 0, st>>>(a);
    } else {
        const size_t need = (size_t)S * M * Cout * 4;
        if (!workspace || workspace_bytes < need) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", workspace_bytes, need);
        a.P = (float *)workspace; a.ksplit = S; a.ksteps = ksteps;
        conv2d_nhwc_glds<64, 3><<<dim3(gx, (Cout + 63) / 64, S), 256, 0, st>>>(a);
        const long n4 = M * Cout / 4;
        conv_splitk_reduce<<<(unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256), 256, 0, st>>>(a.P, S, M, Cout, bias, nullptr, y, relu);
    }
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opdet_maxpool3x3s2_f32(const float *x, float *y, int N, int H, int W, int C, void *stream)
{
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(OPNET_EINVAL, "bad argument");
    if ((C & 3) || !aligned16(x) || !aligned16(y)) return fail(OPNET_EINVAL, "maxpool: C %% 4 == 0, x / y 16-byte aligned");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    maxpool3x3s2<<<ew_blocks((long)N * OH * OW * (C / 4)), 256, 0, (hipStream_t)stream>>>(x, y, N, H, W, C, OH, OW);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opdet_subsample2_f32(const float *x, float *y, int N, int H, int W, int C, void *stream)
{
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(OPNET_EINVAL, "bad argument");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    subsample2<<<ew_blocks((long)N * OH * OW * C), 256, 0, (hipStream_t)stream>>>(x, y, N, H, W, C, OH, OW);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opdet_upsample_add_f32(const float *lateral, const float *top, float *y, int N, int H, int W, int C,
                                      int TH, int TW, void *stream)
{
    if (!lateral || !top || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || TH <= 0 || TW <= 0)
        return fail(OPNET_EINVAL, "bad argument");
    upsample_add<<<ew_blocks((long)N * H * W * C), 256, 0, (hipStream_t)stream>>>(lateral, top, y, N, H, W, C, TH, TW);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opdet_preprocess_frame_f32(const unsigned char *frame_bgr, float *y, int H, int W, int RH, int RW,
                                          int PH, int PW, const float *mean3_host, const float *std3_host, void *stream)
{
    if (!frame_bgr || !y || !mean3_host || !std3_host) return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(y)) return fail(OPNET_EINVAL, "y must be 16-byte aligned");
    if (H <= 0 || W <= 0 || RH <= 0 || RW <= 0 || PH < RH || PW < RW) return fail(OPNET_ESHAPE, "bad sizes");
    preprocess_frame<<<ew_blocks((long)PH * PW), 256, 0, (hipStream_t)stream>>>(
        frame_bgr, y, H, W, RH, RW, PH, PW, (float)H / (float)RH, (float)W / (float)RW, mean3_host[0], mean3_host[1],
        mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

// ------------------------------------------------------------------------------------------------
// post-processing + metric
// ------------------------------------------------------------------------------------------------
extern "C" int opnet_postprocess_iou(const float *y, const float *labels, int *pred_px, int *gt_px,
                                     double *iou, int N, int T, void *stream)
{
    if (!y) return fail(OPNET_EINVAL, "null y");
    if (N <= 0 || T <= 0) return fail(OPNET_ESHAPE, "N=%d T=%d must be positive", N, T);
    if ((gt_px || iou) && !labels) return fail(OPNET_EINVAL, "labels required for gt_px / iou");
    if (!aligned16(y) || !aligned16(labels) || !aligned16(pred_px) || !aligned16(gt_px))
        return fail(OPNET_EINVAL, "buffers must be 16-byte aligned");
    const long n = (long)N * T;
    opnet_postprocess_iou_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const float4 *)y, (const float4 *)labels, (int4 *)pred_px, (int4 *)gt_px, iou, n);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}
