// opnet_ctx.h - launch arguments shared by the OPNet kernels and the C-ABI host code.
//
// The T+3 step launches of one forward are replayed from a hipGraph.  Everything the step kernel
// needs is fixed for a (plan, workspace, packed-weights) triple and travels BY VALUE in the kernarg
// segment (StepArgs: one scalar-load round trip, no pointer chasing on the per-step critical
// path).  What changes between calls - the caller's boxes / y / logits pointers - lives in
// OpnetIO at the head of the workspace, is rewritten by a by-value kernel argument before each
// replay, and is only touched by the two boundary kernels (pack_input, copy_out).
#pragma once
#include <hip/hip_runtime.h>

#define OPNET_SLOTS_ 15
#define OPNET_FEATS_ 6
#define OPNET_KX 90    // LSTM1 input width = 15 slots x 6 features (learned_models.py:24)
#define OPNET_KXQ 24   // ... padded to 96 = 24 float4 k-quads = 6 MFMA hexadecets

struct StepArgs {
    int B, T, RB, H1, H2;
    int mlp;               // 1: OPNetLstmMlp (learned_models.py:55-89) - the video LSTM is replaced by
                           //    relu(Linear 6->H2): the LSTM2 tiles skip the recurrent product
    int train;             // 0: two-deep parity buffers; 1: full-history buffers (slot t+1 holds step t,
                           //    slot 0 the zero initial state) so the backward pass can read every step
    const float4 *xp;      // [T][RB][24][32]        packed LSTM1 input (also read by the selection head)
    const float4 *w1p;     // [H1/4][(96+H1)/16][64] LSTM1 A tiles  (x part | h part)
    const float4 *w2p;     // [H2/4][H2/16][64]      LSTM2 A tiles  (h part)
    const float4 *wih2p;   // [H2][4 gates][2]       LSTM2 x part (6 -> 8 floats)
    const float4 *wselp;   // [H1/16][64]            selection head (15 -> 16 rows)
    const float4 *woutp;   // [H2/16][64]            output head    (4 -> 16 rows)
    float4 *h1buf;         // [2 parity | T+1][RB][H1/4][32]
    float *c1;             // [1 | T+1][RB][H1][32]
    float4 *h2buf;         // [2 parity | T+1][RB][H2/4][32]
    float *c2;             // [1 | T+1][RB][H2][32]
    float4 *x2buf;         // [2 parity | T][RB][2 kq][32]  frames_boxes (6 -> 8 floats), kq-major
    float4 *g1save;        // train only: [T][RB][H1][32] post-activation gates (i,f,g,o) per (unit, clip)
    float4 *g2save;        // train only: [T][RB][H2][32]
    float4 *psave;         // train only: [T][RB][4 kq][32] slot probabilities (15 -> 16), kq-major
    float4 *ystage;        // [RB*32][T]             y_boxes staging
    float *lgstage;        // [RB*32][15][T]         logits staging
};

struct OpnetIO {
    int B, T, RB, pad_;
    const float *boxes;    // [B][T][90]   caller's input
    float *y;              // [B][T][4]    caller's outputs
    float *logits;         // [B][15][T]
    float4 *xp;
    const float4 *ystage;
    const float *lgstage;
    float4 *state;         // recurrent state (h1buf .. x2buf), zeroed by pack_input at the start of a forward
    long state_f4;         // its size in float4 units
};

// ------------------------------------------------------------------------------------------------
// buffer carving (host and device: the step kernel with preloaded scalar arguments rebuilds its StepArgs from the two
// base pointers and the shape, see opnet_step_pl)
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct PackedLayout {  // offsets in floats
    size_t w1p, w2p, wih2p, wselp, woutp, total;
};

__host__ __device__ inline PackedLayout packed_layout(int H1, int H2)
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

__host__ __device__ inline WorkspaceLayout workspace_layout(int B, int T, int H1, int H2)
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
    L.lgstage = o; o += RB * 32 * (size_t)T * OPNET_SLOTS_ * 4;
    L.total = align_up(o, 256);
    return L;
}


// StepArgs of an inference forward from its workspace and packed-weights bases
__host__ __device__ inline void step_args_inference(StepArgs *a, char *w, const float *packed, int B, int T, int H1, int H2)
{
    const WorkspaceLayout W = workspace_layout(B, T, H1, H2);
    const PackedLayout P = packed_layout(H1, H2);
    *a = StepArgs{};
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
}
