// opnet_abi.hip - host side of libopnet_hip.so: the C ABI declared in include/opnet_hip.h.
// Plain pointers and sizes in, HIP launches on the caller's stream out; no torch types.
#include "opnet_kernels.hip"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

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

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(OPNET_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" int opnet_hip_abi_version(void) { return 1; }
extern "C" const char *opnet_last_error(void) { return g_err; }

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

struct PackedLayout {  // offsets in floats
    size_t w1p, w2p, wih2p, wselp, woutp, total;
};

static PackedLayout packed_layout(int H1, int H2)
{
    PackedLayout L;
    size_t o = 0;
    L.w1p = o;   o += (size_t)(H1 / 4) * ((OPNET_KXQ * 4 + H1) / 16) * 256;
    L.w2p = o;   o += (size_t)(H2 / 4) * (H2 / 16) * 256;
    L.wih2p = o; o += (size_t)H2 * 32;
    L.wselp = o; o += (size_t)(H1 / 16) * 256;
    L.woutp = o; o += (size_t)(H2 / 16) * 256;
    L.total = o;
    return L;
}

struct WorkspaceLayout {  // offsets in bytes
    size_t io, xp, state, h1buf, c1, h2buf, c2, x2buf, state_end, ystage, lgstage, total;
};

static WorkspaceLayout workspace_layout(int B, int T, int H1, int H2)
{
    const size_t RB = (B + 31) / 32;
    WorkspaceLayout L;
    size_t o = 0;
    L.io = o;    o += align_up(sizeof(OpnetIO), 256);
    L.xp = o;    o += (size_t)T * RB * OPNET_KXQ * 32 * 16;
    L.state = o;
    L.h1buf = o; o += 2 * RB * (size_t)H1 * 32 * 4;
    L.c1 = o;    o += RB * (size_t)H1 * 32 * 4;
    L.h2buf = o; o += 2 * RB * (size_t)H2 * 32 * 4;
    L.c2 = o;    o += RB * (size_t)H2 * 32 * 4;
    L.x2buf = o; o += 2 * RB * 32 * 8 * 4;
    L.state_end = o;
    L.ystage = o;  o += RB * 32 * (size_t)T * 16;
    L.lgstage = o; o += RB * 32 * (size_t)T * OPNET_SLOTS * 4;
    L.total = align_up(o, 256);
    return L;
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
    const PackedLayout P = packed_layout(H1, H2);
    char *w = (char *)ws;
    memset(a, 0, sizeof(*a));
    a->B = B; a->T = T; a->RB = (B + 31) / 32; a->H1 = H1; a->H2 = H2;
    a->xp = (const float4 *)(w + W.xp);
    a->w1p = (const float4 *)(packed + P.w1p);
    a->w2p = (const float4 *)(packed + P.w2p);
    a->wih2p = (const float4 *)(packed + P.wih2p);
    a->wselp = (const float4 *)(packed + P.wselp);
    a->woutp = (const float4 *)(packed + P.woutp);
    a->h1buf = (float4 *)(w + W.h1buf);
    a->c1 = (float *)(w + W.c1);
    a->h2buf = (float4 *)(w + W.h2buf);
    a->c2 = (float *)(w + W.c2);
    a->x2buf = (float4 *)(w + W.x2buf);
    a->ystage = (float4 *)(w + W.ystage);
    a->lgstage = (float *)(w + W.lgstage);
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
static dim3 step_grid(int RB, int H1, int H2)
{
    return dim3(H2 / 4 + H1 / 4 + 2, RB < OPNET_MAX_GY ? RB : OPNET_MAX_GY, 1);
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
    const dim3 grid = step_grid(a.RB, H1, H2);
    for (int s = 0; s < T + 3; ++s) opnet_step<<<grid, OPNET_THREADS, 0, st>>>(a, s);
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
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        kp.func = (void *)opnet_step;
        kp.gridDim = step_grid(a.RB, p->H1, p->H2);
        kp.blockDim = dim3(OPNET_THREADS, 1, 1);
        kp.kernelParams = args;
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
