/*
 * opnet_oracle.c - plain-C fp32 restatement of the reference's OPNet forward.
 *
 * TEST INFRASTRUCTURE ONLY (parity checker at full size + the "port" CPU baseline that bench.py
 * times on the GPU box's host cores).  Never linked into or called by the product library.
 *
 * Follows reference baselines/learned_models.py:35-52 (OPNet.forward) with torch.nn.LSTM's
 * published cell equations (bias-free, gate rows i,f,g,o; torch==1.4.0, environment.yml:97):
 *     g = x_t W_ih^T + h W_hh^T ; c = sig(f) c + sig(i) tanh(g) ; h = sig(o) tanh(c)
 * Pinned against the reference-generated goldens by tests/test_c_oracle.py.
 *
 * Threading mirrors what a CPU BLAS-backed LSTM does: all B clips advance together, one time
 * step at a time, and the OpenMP threads split the HIDDEN UNITS (each thread owns the i,f,g,o
 * columns of its units for every clip, so its weight slice - 0.5 MB at H2=512 on 8 threads -
 * stays in its private L2 across the 300 steps).  Per step the gates are accumulated in "axpy"
 * order (acc[c][:] += h[c][k] * W^T[k][:], k ascending), which vectorises over the gate dimension
 * without reassociating any sum, so results do not depend on the thread count.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SLOTS 15
#define FEATS 6
#define KX (SLOTS * FEATS)

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* dst[k][j] = src[j][k] */
static void transpose(const float *src, float *dst, int rows, int cols)
{
    for (int j = 0; j < rows; ++j)
        for (int k = 0; k < cols; ++k) dst[(size_t)k * rows + j] = src[(size_t)j * cols + k];
}

/* One LSTM step for units [u0, u1) of every clip.
 * xin  : [B][xstride] input rows (first KXn entries used), wx_t [KXn][4H]
 * hprev: [B][H], wh_t [H][4H];  c [B][H] updated in place;  hnext [B][H] written for own units.
 * Register blocking: a tile of CBLK clips x 4 gates x VW units is accumulated over all k (ascending:
 * x part then h part) before the cell update, so every output is one left-to-right fp32 sum. */
#define VW 16
#define CBLK 4
static void lstm_step_units(const float *xin, size_t xstride, int KXn, const float *wx_t,
                            const float *hprev, const float *wh_t, float *c, float *hnext, int B, int H,
                            int u0, int u1)
{
    const int G = 4 * H;
    for (int j0 = u0; j0 < u1; j0 += VW) {
        const int vw = (u1 - j0) < VW ? (u1 - j0) : VW;
        for (int b0 = 0; b0 < B; b0 += CBLK) {
            const int nc = (B - b0) < CBLK ? (B - b0) : CBLK;
            float acc[CBLK][4][VW];
            memset(acc, 0, sizeof(acc));
            for (int part = 0; part < 2; ++part) {
                const int K = part == 0 ? KXn : H;
                const float *w_t = part == 0 ? wx_t : wh_t;
                for (int k = 0; k < K; ++k) {
                    const float *wrow = w_t + (size_t)k * G + j0;
                    float sv[CBLK];
                    for (int cc = 0; cc < CBLK; ++cc) {
                        const int b = b0 + (cc < nc ? cc : 0);
                        sv[cc] = part == 0 ? xin[(size_t)b * xstride + k] : hprev[(size_t)b * H + k];
                    }
                    if (vw == VW) {
                        for (int g = 0; g < 4; ++g) {
                            const float *restrict wr = wrow + (size_t)g * H;
                            for (int cc = 0; cc < CBLK; ++cc) {
#pragma omp simd
                                for (int j = 0; j < VW; ++j) acc[cc][g][j] += sv[cc] * wr[j];
                            }
                        }
                    } else {
                        for (int g = 0; g < 4; ++g)
                            for (int cc = 0; cc < CBLK; ++cc)
                                for (int j = 0; j < vw; ++j) acc[cc][g][j] += sv[cc] * wrow[(size_t)g * H + j];
                    }
                }
            }
            for (int cc = 0; cc < nc; ++cc) {
                const int b = b0 + cc;
                for (int j = 0; j < vw; ++j) {
                    const int u = j0 + j;
                    const float i = sigmoidf_(acc[cc][0][j]);
                    const float f = sigmoidf_(acc[cc][1][j]);
                    const float gg = tanhf(acc[cc][2][j]);
                    const float o = sigmoidf_(acc[cc][3][j]);
                    const float cn = f * c[(size_t)b * H + u] + i * gg;
                    c[(size_t)b * H + u] = cn;
                    hnext[(size_t)b * H + u] = o * tanhf(cn);
                }
            }
        }
    }
}

/* returns 0 on success */
int opnet_oracle_forward_f32(const float *boxes, const float *w_ih1, const float *w_hh1, const float *w_sel,
                             const float *w_ih2, const float *w_hh2, const float *w_out, float *y,
                             float *logits_bct, int B, int T, int H1, int H2, int nthreads)
{
    const int G1 = 4 * H1, G2 = 4 * H2;
    float *w_ih1_t = malloc(sizeof(float) * (size_t)KX * G1);
    float *w_hh1_t = malloc(sizeof(float) * (size_t)H1 * G1);
    float *w_ih2_t = malloc(sizeof(float) * (size_t)FEATS * G2);
    float *w_hh2_t = malloc(sizeof(float) * (size_t)H2 * G2);
    float *h1 = calloc((size_t)2 * B * H1, sizeof(float)), *c1 = calloc((size_t)B * H1, sizeof(float));
    float *h2 = calloc((size_t)2 * B * H2, sizeof(float)), *c2 = calloc((size_t)B * H2, sizeof(float));
    float *fb = calloc((size_t)B * 8, sizeof(float));
    int fail = !w_ih1_t || !w_hh1_t || !w_ih2_t || !w_hh2_t || !h1 || !c1 || !h2 || !c2 || !fb;
    if (!fail) {
        transpose(w_ih1, w_ih1_t, G1, KX);
        transpose(w_hh1, w_hh1_t, G1, H1);
        transpose(w_ih2, w_ih2_t, G2, FEATS);
        transpose(w_hh2, w_hh2_t, G2, H2);
#ifdef _OPENMP
        if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
        {
#ifdef _OPENMP
            const int nt = omp_get_num_threads(), tid = omp_get_thread_num();
#else
            const int nt = 1, tid = 0;
#endif
            const int u1a = (int)((long)tid * H1 / nt), u1b = (int)((long)(tid + 1) * H1 / nt);
            const int u2a = (int)((long)tid * H2 / nt), u2b = (int)((long)(tid + 1) * H2 / nt);
            {
                for (int t = 0; t < T; ++t) {
                    const int pi = (t + 1) & 1, po = t & 1; /* state parity: read [pi], write [po] */
                    /* ---- LSTM1 over the flattened scene (learned_models.py:36-39) ---- */
                    lstm_step_units(boxes + (size_t)t * KX, (size_t)T * KX, KX, w_ih1_t, h1 + (size_t)pi * B * H1,
                                    w_hh1_t, c1, h1 + (size_t)po * B * H1, B, H1, u1a, u1b);
#pragma omp barrier
                    /* ---- Linear H1->15 + softmax + einsum (:40-43), one clip per iteration ---- */
#pragma omp for schedule(static)
                    for (int b = 0; b < B; ++b) {
                        const float *hh = h1 + (size_t)po * B * H1 + (size_t)b * H1;
                        float lg[SLOTS], p[SLOTS], m = -INFINITY, sum = 0.f;
                        for (int s = 0; s < SLOTS; ++s) {
                            float a = 0.f;
                            for (int k = 0; k < H1; ++k) a += hh[k] * w_sel[(size_t)s * H1 + k];
                            lg[s] = a;
                            logits_bct[((size_t)b * SLOTS + s) * T + t] = a; /* permute(0,2,1) (:50) */
                            if (a > m) m = a;
                        }
                        for (int s = 0; s < SLOTS; ++s) { p[s] = expf(lg[s] - m); sum += p[s]; }
                        for (int s = 0; s < SLOTS; ++s) p[s] /= sum;
                        const float *bx = boxes + ((size_t)b * T + t) * KX;
                        for (int f = 0; f < FEATS; ++f) {
                            float a = 0.f;
                            for (int o = 0; o < SLOTS; ++o) a += bx[o * FEATS + f] * p[o];
                            fb[(size_t)b * 8 + f] = a;
                        }
                    } /* implicit barrier */
                    /* ---- LSTM2 (:46) ---- */
                    lstm_step_units(fb, 8, FEATS, w_ih2_t, h2 + (size_t)pi * B * H2, w_hh2_t, c2,
                                    h2 + (size_t)po * B * H2, B, H2, u2a, u2b);
#pragma omp barrier
                    /* ---- Linear H2->4 (:47); the next step only writes the other parity ---- */
#pragma omp for schedule(static) nowait
                    for (int b = 0; b < B; ++b) {
                        const float *hh = h2 + (size_t)po * B * H2 + (size_t)b * H2;
                        for (int j = 0; j < 4; ++j) {
                            float a = 0.f;
                            for (int k = 0; k < H2; ++k) a += hh[k] * w_out[(size_t)j * H2 + k];
                            y[((size_t)b * T + t) * 4 + j] = a;
                        }
                    }
                }
            }
        }
    }
    free(w_ih1_t); free(w_hh1_t); free(w_ih2_t); free(w_hh2_t);
    free(h1); free(c1); free(h2); free(c2); free(fb);
    return fail ? -1 : 0;
}

int opnet_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
