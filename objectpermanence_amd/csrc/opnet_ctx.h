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
