// opnet_ctx.h - device-resident launch context shared by the OPNet kernels and the C-ABI host code.
//
// The T+3 step launches of one forward are replayed from a hipGraph whose kernel nodes carry only
// (ctx pointer, step index); everything that can change between calls (tensor pointers) lives in
// this struct, which sits at the head of the caller's workspace and is rewritten by a by-value
// kernel argument (opnet_set_ctx) before each replay.
#pragma once
#include <hip/hip_runtime.h>

#define OPNET_SLOTS_ 15
#define OPNET_FEATS_ 6
#define OPNET_KX 90    // LSTM1 input width = 15 slots x 6 features (learned_models.py:24)
#define OPNET_KXQ 24   // ... padded to 96 = 24 float4 k-quads = 6 MFMA hexadecets

struct OpnetCtx {
    int B, T, RB, H1, H2;
    int pad_[3];
    const float *boxes;    // [B][T][90]   caller's input
    float4 *xp;            // [T][RB][24][32]        packed LSTM1 input
    const float4 *w1p;     // [H1/4][(96+H1)/16][64] LSTM1 A tiles  (x part | h part)
    const float4 *w2p;     // [H2/4][H2/16][64]      LSTM2 A tiles  (h part)
    const float4 *wih2p;   // [H2][4 gates][2]       LSTM2 x part (6 -> 8 floats)
    const float4 *wselp;   // [H1/16][64]            selection head (15 -> 16 rows)
    const float4 *woutp;   // [H2/16][64]            output head    (4 -> 16 rows)
    float4 *h1buf;         // [2 parity][RB][H1/4][32]
    float *c1;             // [RB][H1][32]
    float4 *h2buf;         // [2 parity][RB][H2/4][32]
    float *c2;             // [RB][H2][32]
    float4 *x2buf;         // [2 parity][RB][32][2]  frames_boxes (6 -> 8 floats)
    float *y;              // [B][T][4]
    float *logits;         // [B][15][T]
};
