// enc_train_abi.hip - host side of the transformer encoder layer's TRAINING step (C ABI: opseq_encoder_layer_train_*).
// Included by opnet_abi.hip (one translation unit: it uses that file's fail / HIP_TRY / env_int / aligned16 / check_encoder helpers).
#pragma once

// ------------------------------------------------------------------------------------------------
// transformer encoder layer: training forward (saves activations) and backward.  Every product is an NT GEMM on
// conv2d_nhwc_glds with row strides; products that contract over tokens get both operands transposed first.
// ------------------------------------------------------------------------------------------------
// Y[m * YS + n] = act(sum_k X[m * XS + k] W[n * WS + k] + bias[n] (+ R[m * YS + n]));  K % 16 == 0
static void gemm_nt(const float *X, long XS, const float *Wt, long WS, const float *bias, const float *R, float *Y, long YS,
                    long M, int N, int K, int relu, hipStream_t st)
{
    // a SHORT sequence (one clip: S = 300 rows = 3 tiles of 128) leaves a dense product with a handful of workgroups walking K alone:
    // the inference path's 32 x 32 tiles with K split over the workgroup's waves (gemm_bias_act_ks) - 35 -> ~10 us a launch
    if (!R && XS == K && WS == K && YS == N && (K & 15) == 0 && ((M + 63) / 64) * ((N + 63) / 64) < 256 && env_int("OPSEQ_GEMM_KS", 1)) {
        gemm_bias_act_ks<<<dim3((unsigned)((M + 31) / 32), (N + 31) / 32, 1), 256, 0, st>>>(X, Wt, bias, Y, (int)M, N, K, relu);
        return;
    }
    ConvArgs c = {};
    c.X = X; c.Wt = Wt; c.bias = bias; c.R = R; c.Y = Y;
    c.N = 1; c.H = 1; c.W = (int)M; c.Cin = K; c.Cout = N; c.KH = 1; c.KW = 1; c.stride = 1; c.pad = 0;
    c.OH = 1; c.OW = (int)M; c.KP = K; c.relu = relu;
    c.XS = (int)XS; c.WS = (int)WS; c.YS = (int)YS;
    const unsigned gx = (unsigned)((M + 127) / 128);
    // 128 x 128 tiles need well over one round of the chip's resident workgroups (3 a CU = 768) to even out: below ~1 300 of them the
    // 64-wide tile (twice the workgroups, half the work each) balances better - measured on the training step: 16 clips (608 tiles in
    // the FFN products) 6.50 -> 6.10 ms, 8 clips 4.04 -> 3.99, 32 clips (1 200 tiles) 14.65 -> 14.52
    const bool wide = N > 64 && (long)gx * ((N + 127) / 128) >= env_int("OPSEQ_GEMM_WIDE_MIN_TILES", 1300);
    if (wide) conv2d_nhwc_glds<128, 3><<<dim3(gx, (N + 127) / 128, 1), 256, 0, st>>>(c);
    else conv2d_nhwc_glds<64, 3><<<dim3(gx, (N + 63) / 64, 1), 256, 0, st>>>(c);
}

// the same product for a SKINNY output and a long K (the chunked attention's P_c V and dS_c K): K split over up to 32
// workgroups per 64 x 64 tile, partials in `part` (>= KS * M * N floats), reduced in fixed order.  K % 16 == 0.
static int splitk_slices(long M, int N, int K)
{
    const long tiles = ((M + 63) / 64) * ((N + 63) / 64);
    long ks = 512 / (tiles > 0 ? tiles : 1);                 // ~2 workgroups per CU
    const long kmax = K / 128 > 0 ? K / 128 : 1;             // at least 128 k per slice
    if (ks > kmax) ks = kmax;
    if (ks > 32) ks = 32;
    return ks < 1 ? 1 : (int)ks;
}
static void gemm_nt_splitk(const float *X, long XS, const float *Wt, long WS, float *Y, long YS, long M, int N, int K, float *part,
                           hipStream_t st)
{
    const int KS = splitk_slices(M, N, K);
    const int kc = ((K + KS - 1) / KS + 15) / 16 * 16;
    const int ks = (K + kc - 1) / kc;
    enc_gemm_nt_splitk<<<dim3((unsigned)((M + 63) / 64), (N + 63) / 64, ks), 256, 0, st>>>(X, XS, Wt, WS, part, (int)M, N, K, kc);
    const long MN = M * N;
    enc_splitk_reduce<<<(unsigned)((MN + 255) / 256 > 4096 ? 4096 : (MN + 255) / 256), 256, 0, st>>>(part, ks, MN, N, Y, YS);
}

// the token-contracting products of the backward (dW = A^T B: a [N x K] output of a few dozen 128-wide tiles with the whole sequence
// as the contraction): K split over blockIdx.z of the LDS-DMA kernel, raw slices to `part`, summed in slice order (deterministic).
// Without the split these launches were 32 workgroups on 256 CUs: 0.66 ms each at S = 9600, 5.3 ms of a 31-ms step (round 6 profile).
static int gemm_split_slices(long M, int N, int K)
{
    const long tiles = ((M + 127) / 128) * ((N + 63) / 64);
    const int nhex = K >> 4;
    long ks = (512 + tiles - 1) / tiles;
    if (ks > nhex / 16) ks = nhex / 16;              // >= 16 K steps per slice
    if (ks > 16) ks = 16;
    return ks < 2 ? 1 : (int)ks;
}
static void gemm_nt_ks(const float *X, long XS, const float *Wt, long WS, float *Y, long M, int N, int K, float *part, hipStream_t st)
{
    const int ks = (N & 3) ? 1 : gemm_split_slices(M, N, K);
    if (ks <= 1) { gemm_nt(X, XS, Wt, WS, nullptr, nullptr, Y, N, M, N, K, 0, st); return; }
    ConvArgs c = {};
    c.X = X; c.Wt = Wt; c.Y = Y;
    c.N = 1; c.H = 1; c.W = (int)M; c.Cin = K; c.Cout = N; c.KH = 1; c.KW = 1; c.stride = 1; c.pad = 0;
    c.OH = 1; c.OW = (int)M; c.KP = K;
    c.XS = (int)XS; c.WS = (int)WS; c.YS = N;
    const int nhex = K >> 4, per = (nhex + ks - 1) / ks;
    c.P = part; c.ksteps = per; c.ksplit = (nhex + per - 1) / per;
    conv2d_nhwc_glds<64, 3><<<dim3((unsigned)((M + 127) / 128), (N + 63) / 64, c.ksplit), 256, 0, st>>>(c);
    const long n4 = M * N / 4;
    conv_splitk_reduce<<<(unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256), 256, 0, st>>>(part, c.ksplit, M, N, nullptr, nullptr, Y, 0);
}
static const size_t kGemmSplitMax = 16;

// Y = act(X W^T + bias (+ R)) for a SHORT sequence (a handful of 128-row tiles) and a long K: the same split, with the epilogue in
// the reduce.  Dense output rows (YS = N).  A one-clip training step (S = 300) had eleven such launches of 6 workgroups walking
// K = 256 .. 2048 alone: 90 us each.
static void gemm_nt_small(const float *X, long XS, const float *Wt, long WS, const float *bias, const float *R, float *Y, long M, int N,
                          int K, int relu, float *part, size_t part_floats, hipStream_t st)
{
    // a handful of tiles: the 32 x 32 K-split tile kernel with the epilogue inside (no partials, no reduce launch: 21 -> 9 us)
    if (XS == K && WS == K && (K & 15) == 0 && ((M + 63) / 64) * ((N + 63) / 64) < 64 && env_int("OPSEQ_GEMM_KS", 1)) {
        gemm_bias_act_ks<<<dim3((unsigned)((M + 31) / 32), (N + 31) / 32, 1), 256, 0, st>>>(X, Wt, bias, Y, (int)M, N, K, relu, R);
        return;
    }
    const long tiles = ((M + 127) / 128) * ((N + 63) / 64);
    int ks = (N & 3) || tiles >= 128 ? 1 : gemm_split_slices(M, N, K);
    while (ks > 1 && (size_t)ks * M * N > part_floats) --ks;
    if (ks <= 1) { gemm_nt(X, XS, Wt, WS, bias, R, Y, N, M, N, K, relu, st); return; }
    ConvArgs c = {};
    c.X = X; c.Wt = Wt; c.Y = Y;
    c.N = 1; c.H = 1; c.W = (int)M; c.Cin = K; c.Cout = N; c.KH = 1; c.KW = 1; c.stride = 1; c.pad = 0;
    c.OH = 1; c.OW = (int)M; c.KP = K;
    c.XS = (int)XS; c.WS = (int)WS; c.YS = N;
    const int nhex = K >> 4, per = (nhex + ks - 1) / ks;
    c.P = part; c.ksteps = per; c.ksplit = (nhex + per - 1) / per;
    conv2d_nhwc_glds<64, 3><<<dim3((unsigned)((M + 127) / 128), (N + 63) / 64, c.ksplit), 256, 0, st>>>(c);
    const long n4 = M * N / 4;
    conv_splitk_reduce<<<(unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256), 256, 0, st>>>(part, c.ksplit, M, N, bias, R, Y, relu);
}

// ---- flash training attention (attn_train_kernels.hip) ---------------------------------------------------------------------
// head sizes the LDS-DMA tiles are built for; anything else (and OPSEQ_ATTN_FLASH=0) keeps the chunked GEMM form below
static bool attn_flash_shape(long S, int E, int nhead)
{
    const int hd = E / nhead;
    if (!(hd == 16 || hd == 32 || hd == 64 || hd == 128)) return false;
    if ((double)S * 3 * E * 4 >= 2147483648.0) return false;      // 32-bit buffer offsets
    return env_int("OPSEQ_ATTN_FLASH", 1) != 0;
}
// stationary fragments per wave of the backward passes: two for the small heads once that still leaves a workgroup per CU
static int attn_bwd_af(long S, int nhead, int hd)
{
    const int forced = env_int("OPSEQ_ATTN_AF", 0);
    if (forced == 1 || forced == 2) return hd > 64 ? 1 : forced;
    return (hd <= 64 && ((S + 127) / 128) * nhead >= 256) ? 2 : 1;
}
// slices of the streaming sweep (gridDim.z): the one that wastes the least of the last round of workgroups
static int attn_sweep_split(long W, long ntiles, int zmax)
{
    const int forced = env_int("OPSEQ_ATTN_ZS", 0);
    if (forced > 0) return forced < zmax ? forced : (zmax < 1 ? 1 : zmax);
    if (W >= 8 * 256) return 1;
    if (ntiles < 64) {
        // a short sequence (one clip: 10 workgroups sweeping 19 steps on a 256-CU device): up to zmax slices of >= 4 steps
        int z = (int)(ntiles / 4);
        if (z > zmax) z = zmax;
        if (W > 0 && z > 256 / W) z = (int)(256 / W);
        return z < 1 ? 1 : z;
    }
    int best = 1;
    double best_cost = (double)((W + 255) / 256);
    for (int z = 2; z <= zmax && ntiles / z >= 32; ++z) {
        const double cost = (double)((W * z + 255) / 256) / z * (1.0 + 0.01 * z);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = z; }
    }
    // just under one workgroup per CU (S = 7 200: 226): the model above sees one full round and leaves it alone, but two workgroups a CU
    // are resident - two slices of half the length hide each other's latency (measured 11.68 -> 11.0 ms a 24-clip step, any split >= 2)
    if (best == 1 && W < 512 && zmax >= 2 && ntiles / 2 >= 32) best = 2;
    return best;
}

template <int HD>
static void launch_attn_train_fwd(const AttTrainFwdArgs &a, int nhead, int KS, hipStream_t st)
{
    const dim3 g((unsigned)((a.S + 63) / 64), nhead, KS);
    if (a.thresh == 0u) attention_train_fwd<HD, 0><<<g, 256, 0, st>>>(a);
    else if (a.ds.mask) attention_train_fwd<HD, 2><<<g, 256, 0, st>>>(a);
    else attention_train_fwd<HD, 1><<<g, 256, 0, st>>>(a);
}
template <int HD, int DROP>
static void launch_attn_bwd_d(const AttBwdArgs &a, bool dkv, int AF, int nhead, int ZS, hipStream_t st)
{
    const dim3 g((unsigned)((a.S + 64 * AF - 1) / (64 * AF)), nhead, ZS);
    if constexpr (HD <= 64) {
        if (AF == 2) {
            if (dkv) attention_bwd<HD, true, 2, DROP><<<g, 256, 0, st>>>(a);
            else attention_bwd<HD, false, 2, DROP><<<g, 256, 0, st>>>(a);
            return;
        }
    }
    if (dkv) attention_bwd<HD, true, 1, DROP><<<g, 256, 0, st>>>(a);
    else attention_bwd<HD, false, 1, DROP><<<g, 256, 0, st>>>(a);
}
template <int HD>
static void launch_attn_bwd(const AttBwdArgs &a, bool dkv, int AF, int nhead, int ZS, hipStream_t st)
{
    if (a.thresh == 0u) launch_attn_bwd_d<HD, 0>(a, dkv, AF, nhead, ZS, st);
    else if (a.ds.mask) launch_attn_bwd_d<HD, 2>(a, dkv, AF, nhead, ZS, st);       // (tests: the reference's recorded masks)
    else launch_attn_bwd_d<HD, 1>(a, dkv, AF, nhead, ZS, st);
}

static void transpose_to(const float *src, long sld, float *dst, long dld, long R, int C, hipStream_t st)
{
    enc_transpose<<<dim3((unsigned)((dld + 31) / 32), (C + 31) / 32, 1), 256, 0, st>>>(src, sld, dst, dld, (int)R, C);
}

static unsigned ew_grid(long n) { return (unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256); }

struct EncSaved { size_t z_in, qkv, att, astat, u1, st1, x1, hid, u2, st2, total; long Sp; };   // offsets in floats

// The attention of the training step is evaluated in CHUNKS of Qc query rows (flash-style tiling at GEMM granularity): the
// scores of a chunk [Qc][S] are computed, soft-maxed, used and dropped - in the forward AND again in the backward, which
// recomputes them from q and k instead of reading a saved S x S matrix per head (737 MB per layer at S = 9600, 3.1 GiB per
// step, S <= 23 000: round 2).  A chunk is sized to stay in the 256 MB Infinity Cache with its companions (~24 MB each).
static long enc_chunk_rows(long S)
{
    const long Sp = (S + 15) / 16 * 16;
    // 128-row blocks per chunk: up to ~48 MB of fp32 per chunk buffer, and among those sizes the one whose score GEMM
    // (r x ceil(S / 128) tiles of 128 x 128) fills whole rounds of the 256 CUs best (S = 9600: 10 blocks = 750 tiles = 2.93
    // rounds; 5 blocks = 375 tiles = 1.46 rounds would idle a quarter of the chip in every launch)
    const long colt = (Sp + 127) / 128;
    long rmax = (12L << 20) / Sp / 128;
    if (rmax < 1) rmax = 1;
    long best = rmax;
    double beff = 0.0;
    for (long r = rmax; r >= (rmax + 1) / 2; --r) {
        const long tiles = r * colt, rounds = (tiles + 255) / 256;
        const double eff = (double)tiles / (double)(rounds * 256);
        if (eff > beff + 1e-9) { beff = eff; best = r; }
    }
    long qc = best * 128;
    const int forced = env_int("OPSEQ_ATTN_CHUNK", 0);          // tests: several chunks on a short sequence (a multiple of 16)
    if (forced > 0) qc = (forced + 15) / 16 * 16;
    return qc < S ? qc : S;
}

static EncSaved enc_saved_layout(long S, int E, int nhead, int ffn)
{
    EncSaved L;
    L.Sp = (S + 15) / 16 * 16;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += (n + 63) / 64 * 64; return at; };
    L.z_in = take((size_t)S * E);  L.qkv = take((size_t)S * 3 * E);  L.astat = take((size_t)nhead * S * 2);   // softmax (max, 1/sum) per head and row
    L.att = take((size_t)S * E);   L.u1 = take((size_t)S * E);        L.st1 = take((size_t)S * 2);
    L.x1 = take((size_t)S * E);    L.hid = take((size_t)S * ffn);     L.u2 = take((size_t)S * E);
    L.st2 = take((size_t)S * 2);
    L.total = o;
    return L;
}

struct EncScratch { size_t wt_in, wt_out, wt_l1, wt_l2, pc, sq0, sq1, sq2, hT, t0, tA, tB, dqkv, e0, e1, e2, part, st4, wgp, total, tAB; };

static EncScratch enc_scratch_layout(long S, int E, int nhead, int ffn)
{
    EncScratch L;
    const long Sp = (S + 15) / 16 * 16;
    const int hd = E / nhead;
    const size_t wide = (size_t)(ffn > 3 * E ? ffn : 3 * E);
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += (n + 63) / 64 * 64; return at; };
    L.wt_in = take((size_t)E * 3 * E);  L.wt_out = take((size_t)E * E);
    L.wt_l1 = take((size_t)E * ffn);    L.wt_l2 = take((size_t)ffn * E);
    // the chunked GEMM form of the attention (head sizes the flash kernels are not built for, OPSEQ_ATTN_FLASH=0) needs four [Qc][S]
    // chunk buffers and the per-head transposes; the flash form needs neither (its split-sweep partials alias tA / tB)
    const bool flash = attn_flash_shape(S, E, nhead);
    const size_t chunk = flash ? 0 : (size_t)((enc_chunk_rows(S) + 15) / 16 * 16) * Sp;    // one [Qc][S] / [S][Qc] attention chunk
    L.pc = take(chunk);  L.sq0 = take(chunk);  L.sq1 = take(chunk);  L.sq2 = take(chunk);
    L.hT = take(flash ? 0 : (size_t)4 * hd * Sp);   L.t0 = take((size_t)S * ffn);
    L.tA = take(wide * Sp);             L.tB = take(wide * Sp);
    L.tAB = o - L.tA;                   // floats from tA to the end of tB (contiguous)
    L.dqkv = take((size_t)S * 3 * E);
    L.e0 = take((size_t)S * E);  L.e1 = take((size_t)S * E);  L.e2 = take((size_t)S * E);
    L.part = take((size_t)((S + 63) / 64) * 2 * wide);
    L.st4 = take(flash ? (size_t)nhead * S * 4 : 0);
    L.wgp = take(kGemmSplitMax * (size_t)E * wide);        // K slices of the split weight-gradient products
    L.total = o;
    return L;
}

static int check_encoder_train(long S, int E, int nhead, int ffn, float p_drop)
{
    if (int rc = check_encoder(S, E, nhead, ffn)) return rc;
    if ((E & 15) || (ffn & 15) || E > 64 * ENC_LN_MAX_PER_LANE) return fail(OPNET_ESHAPE, "E and ffn must be multiples of 16, E <= %d", 64 * ENC_LN_MAX_PER_LANE);
    if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(OPNET_EINVAL, "dropout probability must be in [0, 1)");
    const long wide = ffn > 3 * E ? ffn : 3 * E;
    if ((double)S * wide * 4 >= 2147483648.0)
        return fail(OPNET_ESHAPE, "S=%ld: an activation matrix exceeds the GEMM kernel's 2 GiB operand limit", S);
    return OPNET_OK;
}

extern "C" size_t opseq_encoder_train_saved_bytes(long S, int E, int nhead, int ffn)
{
    if (check_encoder_train(S, E, nhead, ffn, 0.f)) return 0;
    return enc_saved_layout(S, E, nhead, ffn).total * sizeof(float);
}

extern "C" size_t opseq_encoder_train_scratch_bytes(long S, int E, int nhead, int ffn)
{
    if (check_encoder_train(S, E, nhead, ffn, 0.f)) return 0;
    return enc_scratch_layout(S, E, nhead, ffn).total * sizeof(float);
}

/* TEST-ONLY (tests/test_siblings_train.py): the layer calls made with `seed` take their four dropout masks (device pointers,
 * one byte per element, nonzero = keep; sites 0 attention weights [nhead][S][S], 1 [S][E], 2 [S][ffn], 3 [S][E]) from these buffers
 * instead of the counter generator - how the reference's own masks are fed in.  slot 0..7; *_clear() empties the table.  The table
 * lives on the HOST: a layer call looks its seed up when it builds its launches and hands the kernels a mask pointer (null in
 * production) - no device state, nothing for a production kernel to scan. */
#define ENC_TEST_MASK_SLOTS 8
struct EncTestMasks { unsigned long long seed; const unsigned char *m[4]; bool used; };
static EncTestMasks g_enc_test[ENC_TEST_MASK_SLOTS] = {};
static std::mutex g_enc_test_mu;

extern "C" int opseq_encoder_test_masks_set(int slot, unsigned long long seed, const unsigned char *m0, const unsigned char *m1,
                                            const unsigned char *m2, const unsigned char *m3)
{
    if (slot < 0 || slot >= ENC_TEST_MASK_SLOTS || !m0 || !m1 || !m2 || !m3) return fail(OPNET_EINVAL, "bad test-mask slot / null mask");
    std::lock_guard<std::mutex> lock(g_enc_test_mu);
    g_enc_test[slot] = EncTestMasks{seed, {m0, m1, m2, m3}, true};
    return OPNET_OK;
}
extern "C" int opseq_encoder_test_masks_clear(void)
{
    std::lock_guard<std::mutex> lock(g_enc_test_mu);
    for (auto &e : g_enc_test) e.used = false;
    return OPNET_OK;
}

struct EncDrop {
    unsigned thresh; float inv_keep; unsigned long long seed;
    const unsigned char *mask[4];
    EncSite at(unsigned site) const { return EncSite{seed, mask[site & 3], site}; }
};

static EncDrop enc_drop(float p, unsigned long long seed)
{
    EncDrop d;
    d.thresh = p > 0.f ? (unsigned)((double)p * 4294967296.0) : 0u;
    d.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    d.seed = seed;
    for (auto &m : d.mask) m = nullptr;
    std::lock_guard<std::mutex> lock(g_enc_test_mu);
    for (const auto &e : g_enc_test)
        if (e.used && e.seed == seed)
            for (int k = 0; k < 4; ++k) d.mask[k] = e.m[k];
    return d;
}

/* training-mode forward of one post-LN nn.TransformerEncoderLayer: z_out = layer(z_in), activations kept in `saved`
 * for opseq_encoder_layer_train_backward_f32.  Dropout sites: 0 attention weights, 1 after out_proj, 2 after ReLU,
 * 3 after linear2 (reference learned_models.py:166: default dropout 0.1); p_drop = 0 disables them. */
extern "C" int opseq_encoder_layer_train_forward_f32(const float *z_in, float *z_out, const float *in_w, const float *in_b,
                                                     const float *out_w, const float *out_b, const float *l1_w,
                                                     const float *l1_b, const float *l2_w, const float *l2_b,
                                                     const float *n1_w, const float *n1_b, const float *n2_w,
                                                     const float *n2_b, void *saved, size_t saved_bytes, void *scratch,
                                                     size_t scratch_bytes, long S, int E, int nhead, int ffn, float p_drop,
                                                     unsigned long long seed, void *stream)
{
    if (int rc = check_encoder_train(S, E, nhead, ffn, p_drop)) return rc;
    if (!z_in || !z_out || !in_w || !in_b || !out_w || !out_b || !l1_w || !l1_b || !l2_w || !l2_b || !n1_w || !n1_b ||
        !n2_w || !n2_b || !saved || !scratch)
        return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(z_in) || !aligned16(z_out) || !aligned16(saved) || !aligned16(scratch) || !aligned16(in_w) ||
        !aligned16(out_w) || !aligned16(l1_w) || !aligned16(l2_w))
        return fail(OPNET_EINVAL, "activations / workspaces / weight matrices must be 16-byte aligned");
    const EncSaved SV = enc_saved_layout(S, E, nhead, ffn);
    const EncScratch SC = enc_scratch_layout(S, E, nhead, ffn);
    if (saved_bytes < SV.total * 4 || scratch_bytes < SC.total * 4) return fail(OPNET_EWORKSPACE, "saved / scratch buffer too small");
    const size_t wgp_floats = SC.total - SC.wgp;
    hipStream_t st = (hipStream_t)stream;
    float *sv = (float *)saved, *sc = (float *)scratch;
    const int hd = E / nhead;
    const long Sp = SV.Sp;
    const EncDrop D = enc_drop(p_drop, seed);
    const float scale = 1.0f / sqrtf((float)hd);
    float *zs = sv + SV.z_in, *qkv = sv + SV.qkv, *att = sv + SV.att;
    HIP_TRY(hipMemcpyAsync(zs, z_in, (size_t)S * E * 4, hipMemcpyDeviceToDevice, st));
    gemm_nt(zs, E, in_w, E, in_b, nullptr, qkv, 3 * E, S, 3 * E, E, 0, st);
    const long QC = enc_chunk_rows(S);
    const bool flash = attn_flash_shape(S, E, nhead);
    if (flash) {
        // all heads in ONE launch, the scores in registers (attn_train_kernels.hip); a key split when the workgroup count is a
        // small non-multiple of the CU count (partials in tA / tB, idle in the forward)
        ProfPair pe{};
        const bool prof = prof_begin(st, &pe);
        AttTrainFwdArgs fa = {};
        fa.qkv = qkv; fa.att = att; fa.stats = (float2 *)(sv + SV.astat);
        fa.S = (int)S; fa.E = E; fa.scale = scale; fa.ds = D.at(0u); fa.thresh = D.thresh; fa.inv_keep = D.inv_keep;
        int KS = attn_sweep_split(((S + 63) / 64) * nhead, (S + 15) / 16, 8);
        while (KS > 1 && (size_t)KS * S * E + (size_t)KS * S * nhead * 2 > SC.tAB) --KS;
        fa.opart = sc + SC.tA;
        fa.ml = (float2 *)(sc + SC.tA + (size_t)KS * S * E);
        switch (hd) {
        case 16: launch_attn_train_fwd<16>(fa, nhead, KS, st); break;
        case 32: launch_attn_train_fwd<32>(fa, nhead, KS, st); break;
        case 64: launch_attn_train_fwd<64>(fa, nhead, KS, st); break;
        default: launch_attn_train_fwd<128>(fa, nhead, KS, st); break;
        }
        if (KS > 1) {
            const long n = S * (E / 4);
            attention_train_merge<<<(unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256), 256, 0, st>>>(fa.opart, fa.ml, att, fa.stats,
                                                                                                           (int)S, E, nhead, KS);
        }
        if (prof) prof_end(PROF_ATTN_TF, st, pe);
    }
    for (int h = 0; h < nhead && !flash; ++h) {
        float *Vt = sc + SC.hT;
        transpose_to(qkv + 2 * E + h * hd, 3 * E, Vt, Sp, S, hd, st);
        for (long q0 = 0; q0 < S; q0 += QC) {
            const long qn = S - q0 < QC ? S - q0 : QC;                   // query rows of this chunk
            float *P = sc + SC.pc;                                       // [qn][Sp] scores -> probabilities, not kept
            gemm_nt(qkv + q0 * 3 * E + h * hd, 3 * E, qkv + E + h * hd, 3 * E, nullptr, nullptr, P, Sp, qn, (int)S, hd, 0, st);
            float *Pd = D.thresh ? sc + SC.sq0 : nullptr;
            enc_softmax_rows<<<(unsigned)qn, 256, 0, st>>>(P, Pd, Sp, (int)S, scale, D.at(0u),
                                                           (unsigned long long)h * S * S + (unsigned long long)q0 * S, D.thresh, D.inv_keep,
                                                           (float2 *)(sv + SV.astat) + (size_t)h * S + q0);
            if (S > QC) gemm_nt_splitk(Pd ? Pd : P, Sp, Vt, Sp, att + q0 * E + h * hd, E, qn, hd, (int)Sp, sc + SC.sq1, st);
            else gemm_nt(Pd ? Pd : P, Sp, Vt, Sp, nullptr, nullptr, att + q0 * E + h * hd, E, qn, hd, (int)Sp, 0, st);
        }
    }
    float *e0 = sc + SC.e0;
    gemm_nt(att, E, out_w, E, out_b, nullptr, e0, E, S, E, E, 0, st);
    enc_add_drop_ln<<<(unsigned)((S + 3) / 4), 256, 0, st>>>(zs, e0, n1_w, n1_b, sv + SV.u1, (float2 *)(sv + SV.st1),
                                                            sv + SV.x1, (int)S, E, 1e-5f, D.at(1u), D.thresh, D.inv_keep);
    gemm_nt(sv + SV.x1, E, l1_w, E, l1_b, nullptr, sv + SV.hid, ffn, S, ffn, E, 1, st);
    if (D.thresh) enc_dropout<<<ew_grid((long)S * ffn), 256, 0, st>>>(sv + SV.hid, (long)S * ffn, D.at(2u), D.thresh, D.inv_keep);
    gemm_nt_small(sv + SV.hid, ffn, l2_w, ffn, l2_b, nullptr, e0, S, E, ffn, 0, sc + SC.wgp, wgp_floats, st);
    enc_add_drop_ln<<<(unsigned)((S + 3) / 4), 256, 0, st>>>(sv + SV.x1, e0, n2_w, n2_b, sv + SV.u2, (float2 *)(sv + SV.st2),
                                                            z_out, (int)S, E, 1e-5f, D.at(3u), D.thresh, D.inv_keep);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

/* backward of the layer: dz_out [S][E] -> dz_in [S][E] and the 12 parameter gradients (state_dict layouts; all
 * OVERWRITTEN).  Same p_drop / seed as the forward call that filled `saved`. */
extern "C" int opseq_encoder_layer_train_backward_f32(const float *dz_out, float *dz_in, const float *in_w,
                                                      const float *out_w, const float *l1_w, const float *l2_w,
                                                      const float *n1_w, const float *n2_w, float *g_in_w, float *g_in_b,
                                                      float *g_out_w, float *g_out_b, float *g_l1_w, float *g_l1_b,
                                                      float *g_l2_w, float *g_l2_b, float *g_n1_w, float *g_n1_b,
                                                      float *g_n2_w, float *g_n2_b, const void *saved, size_t saved_bytes,
                                                      void *scratch, size_t scratch_bytes, long S, int E, int nhead,
                                                      int ffn, float p_drop, unsigned long long seed, void *stream)
{
    if (int rc = check_encoder_train(S, E, nhead, ffn, p_drop)) return rc;
    if (!dz_out || !dz_in || !in_w || !out_w || !l1_w || !l2_w || !n1_w || !n2_w || !g_in_w || !g_in_b || !g_out_w ||
        !g_out_b || !g_l1_w || !g_l1_b || !g_l2_w || !g_l2_b || !g_n1_w || !g_n1_b || !g_n2_w || !g_n2_b || !saved || !scratch)
        return fail(OPNET_EINVAL, "null pointer");
    if (!aligned16(dz_out) || !aligned16(dz_in) || !aligned16(saved) || !aligned16(scratch) || !aligned16(g_in_w) ||
        !aligned16(g_out_w) || !aligned16(g_l1_w) || !aligned16(g_l2_w))
        return fail(OPNET_EINVAL, "activations / workspaces / weight-gradient matrices must be 16-byte aligned");
    const EncSaved SV = enc_saved_layout(S, E, nhead, ffn);
    const EncScratch SC = enc_scratch_layout(S, E, nhead, ffn);
    if (saved_bytes < SV.total * 4 || scratch_bytes < SC.total * 4) return fail(OPNET_EWORKSPACE, "saved / scratch buffer too small");
    const size_t wgp_floats = SC.total - SC.wgp;
    hipStream_t st = (hipStream_t)stream;
    const float *sv = (const float *)saved;
    float *sc = (float *)scratch;
    const int hd = E / nhead;
    const long Sp = SV.Sp;
    const EncDrop D = enc_drop(p_drop, seed);
    const float scale = 1.0f / sqrtf((float)hd);
    const float *zs = sv + SV.z_in, *qkv = sv + SV.qkv, *att = sv + SV.att, *hid = sv + SV.hid, *x1 = sv + SV.x1;
    float *e0 = sc + SC.e0, *e1 = sc + SC.e1, *e2 = sc + SC.e2, *t0 = sc + SC.t0, *tA = sc + SC.tA, *tB = sc + SC.tB;
    float *dqkv = sc + SC.dqkv, *part = sc + SC.part, *sq0 = sc + SC.sq0, *sq1 = sc + SC.sq1;
    // LayerNorm backward: rows per workgroup as few as the partial-sum buffer ((S + 63) / 64 x 2 x wide floats) allows - 64 rows a
    // workgroup left a one-clip step with 5 workgroups walking 16 rows per wave (37 us a launch)
    const long wide_ = ffn > 3 * E ? ffn : 3 * E;
    long nb_cap = ((S + 63) / 64) * (wide_ / E);
    if (nb_cap > 256) nb_cap = 256;                  // (enc_colsum_final walks the workgroups' partial sums one by one)
    const int ln_rows = (int)(((S + nb_cap - 1) / nb_cap + 3) / 4 * 4);
    const int cs_rows = 64;
    const int nb_ln = (int)((S + ln_rows - 1) / ln_rows), nb_cs = (int)((S + cs_rows - 1) / cs_rows);
    auto colsum = [&](const float *X, long ld, int N, float *out) {
        if (S < 4096) {         // a short sequence: one launch, a workgroup per 64 columns
            enc_colsum_small<<<(N + 63) / 64, 256, 0, st>>>(X, ld, out, (int)S, N);
            return;
        }
        enc_colsum_part<<<dim3((N + 255) / 256, nb_cs, 1), 256, 0, st>>>(X, ld, part, (int)S, N, cs_rows);
        enc_colsum_final<<<dim3((N + 63) / 64, 1, 1), 256, 0, st>>>(part, out, out, nb_cs, N);
    };
    // dW[n][k] = sum_s A[s][n] B[s][k]  (A [S][N], B [S][K])
    auto gemm_tn = [&](const float *A, long lda, int N, const float *B, long ldb, int K, float *dW) {
        if (S <= 1024 && env_int("OPSEQ_GEMM_TN", 1)) {       // a short sequence: straight from the row-major operands (no transposes)
            gemm_tn_ks<<<dim3((N + 31) / 32, (K + 31) / 32, 1), 256, 0, st>>>(A, lda, B, ldb, dW, (int)S, N, K);
            return;
        }
        transpose_to(A, lda, tA, Sp, S, N, st);
        transpose_to(B, ldb, tB, Sp, S, K, st);
        gemm_nt_ks(tA, Sp, tB, Sp, dW, N, K, (int)Sp, sc + SC.wgp, st);
    };
    // transposed weights (the "dX = dY W" products contract over the weight's ROW index)
    transpose_to(in_w, E, sc + SC.wt_in, 3 * E, 3 * E, E, st);       // [E][3E]
    transpose_to(out_w, E, sc + SC.wt_out, E, E, E, st);             // [E][E]
    transpose_to(l1_w, E, sc + SC.wt_l1, ffn, ffn, E, st);           // [E][ffn]
    transpose_to(l2_w, ffn, sc + SC.wt_l2, E, E, ffn, st);           // [ffn][E]

    // ---- norm2, dropout2, linear2, ReLU/dropout, linear1 ----
    enc_ln_bwd<<<nb_ln, 256, 0, st>>>(dz_out, sv + SV.u2, (const float2 *)(sv + SV.st2), n2_w, nullptr, e1, part, (int)S, E, ln_rows);
    enc_colsum_final<<<dim3((E + 63) / 64, 2, 1), 256, 0, st>>>(part, g_n2_w, g_n2_b, nb_ln, E);
    enc_dropout_copy<<<ew_grid((long)S * E), 256, 0, st>>>(e1, e0, (long)S * E, D.at(3u), D.thresh, D.inv_keep);   // d f
    colsum(e0, E, E, g_l2_b);
    gemm_tn(e0, E, E, hid, ffn, ffn, g_l2_w);                                               // [E][ffn]
    gemm_nt(e0, E, sc + SC.wt_l2, E, nullptr, nullptr, t0, ffn, S, ffn, E, 0, st);          // d hid
    enc_relu_drop_bwd<<<ew_grid((long)S * ffn), 256, 0, st>>>(t0, hid, (long)S * ffn, D.inv_keep);
    colsum(t0, ffn, ffn, g_l1_b);
    gemm_tn(t0, ffn, ffn, x1, E, E, g_l1_w);                                                // [ffn][E]
    gemm_nt_small(t0, ffn, sc + SC.wt_l1, ffn, nullptr, e1, e2, S, E, ffn, 0, sc + SC.wgp, wgp_floats, st);   // d x1 = d pre W1 + d u2
    // ---- norm1, dropout1, out_proj ----
    enc_ln_bwd<<<nb_ln, 256, 0, st>>>(e2, sv + SV.u1, (const float2 *)(sv + SV.st1), n1_w, nullptr, e1, part, (int)S, E, ln_rows);
    enc_colsum_final<<<dim3((E + 63) / 64, 2, 1), 256, 0, st>>>(part, g_n1_w, g_n1_b, nb_ln, E);
    enc_dropout_copy<<<ew_grid((long)S * E), 256, 0, st>>>(e1, e0, (long)S * E, D.at(1u), D.thresh, D.inv_keep);   // d proj
    colsum(e0, E, E, g_out_b);
    gemm_tn(e0, E, E, att, E, E, g_out_w);
    gemm_nt(e0, E, sc + SC.wt_out, E, nullptr, nullptr, e2, E, S, E, E, 0, st);             // d att
    // ---- attention, head by head ----
    float *Vt_unused = sc + SC.hT, *Kt = sc + SC.hT + (size_t)hd * Sp, *Qt = sc + SC.hT + (size_t)2 * hd * Sp,
          *dOt = sc + SC.hT + (size_t)3 * hd * Sp;
    (void)Vt_unused;
    float *pc = sc + SC.pc, *sq2 = sc + SC.sq2;
    const long QC = enc_chunk_rows(S);
    const bool flash = attn_flash_shape(S, E, nhead);
    if (flash) {
        // two launches for all heads: the query-stationary pass (dQ) and the key-stationary pass (dK, dV), each recomputing its score
        // and dP tiles in registers from q, k, v, dO and the forward's row statistics (attn_train_kernels.hip)
        ProfPair pe{};
        const bool prof = prof_begin(st, &pe);
        float4 *st4 = (float4 *)(sc + SC.st4);
        attention_bwd_prep<<<(unsigned)((S * nhead + 3) / 4), 256, 0, st>>>(e2, att, (const float2 *)(sv + SV.astat), st4, (int)S, E, nhead);
        AttBwdArgs ba = {};
        ba.st4 = st4; ba.S = (int)S; ba.E = E; ba.scale = scale; ba.ds = D.at(0u); ba.thresh = D.thresh; ba.inv_keep = D.inv_keep;
        ba.ldo = 3 * E; ba.part = sc + SC.tA;
        const int AF = attn_bwd_af(S, nhead, hd);
        const long W = ((S + 64 * AF - 1) / (64 * AF)) * nhead;
        for (int pass = 0; pass < 2; ++pass) {
            const int NOUT = pass ? 2 : 1;
            int zmax = (int)(SC.tAB / ((size_t)NOUT * S * E));
            if (zmax > 8) zmax = 8;
            const int ZS = attn_sweep_split(W, (S + 15) / 16, zmax);
            if (pass == 0) {
                ba.x0 = qkv; ba.ldx0 = 3 * E; ba.x1 = e2; ba.ldx1 = E;
                ba.y0 = qkv + E; ba.y1 = qkv + 2 * E; ba.ldy0 = ba.ldy1 = 3 * E;
                ba.out0 = dqkv; ba.out1 = nullptr;
            } else {
                ba.x0 = qkv + E; ba.x1 = qkv + 2 * E; ba.ldx0 = ba.ldx1 = 3 * E;
                ba.y0 = qkv; ba.ldy0 = 3 * E; ba.y1 = e2; ba.ldy1 = E;
                ba.out0 = dqkv + E; ba.out1 = dqkv + 2 * E;
            }
            switch (hd) {
            case 16: launch_attn_bwd<16>(ba, pass != 0, AF, nhead, ZS, st); break;
            case 32: launch_attn_bwd<32>(ba, pass != 0, AF, nhead, ZS, st); break;
            case 64: launch_attn_bwd<64>(ba, pass != 0, AF, nhead, ZS, st); break;
            default: launch_attn_bwd<128>(ba, pass != 0, AF, nhead, ZS, st); break;
            }
            if (ZS > 1) {
                const long n4 = S * (E / 4) * NOUT;
                attention_bwd_reduce<<<(unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256), 256, 0, st>>>(ba.part, ZS, NOUT, ba.out0,
                                                                                                              ba.out1, 3 * E, S, E);
            }
        }
        if (prof) prof_end(PROF_ATTN_TB, st, pe);
    }
    for (int h = 0; h < nhead && !flash; ++h) {
        transpose_to(e2 + h * hd, E, dOt, Sp, S, hd, st);                                     // dO^T [hd][query]
        transpose_to(qkv + E + h * hd, 3 * E, Kt, Sp, S, hd, st);
        transpose_to(qkv + h * hd, 3 * E, Qt, Sp, S, hd, st);
        float *dQ = dqkv + h * hd, *dK = dqkv + E + h * hd, *dV = dqkv + 2 * E + h * hd;
        for (long q0 = 0; q0 < S; q0 += QC) {
            const long qn = S - q0 < QC ? S - q0 : QC;
            const long qp = (qn + 15) / 16 * 16;                                              // the chunk as a K dimension
            const unsigned long long idx0 = (unsigned long long)h * S * S + (unsigned long long)q0 * S;
            // the chunk's probabilities again: the forward's score GEMM and the last pass of its softmax (the same bits)
            gemm_nt(qkv + q0 * 3 * E + h * hd, 3 * E, qkv + E + h * hd, 3 * E, nullptr, nullptr, pc, Sp, qn, (int)S, hd, 0, st);
            const float *Pu = pc;                                        // the matrix that multiplied V in the forward
            enc_softmax_from_stats<<<(unsigned)qn, 256, 0, st>>>(pc, D.thresh ? sq2 : nullptr, Sp, (int)S, scale, D.at(0u), idx0,
                                                                 D.thresh, D.inv_keep, (const float2 *)(sv + SV.astat) + (size_t)h * S + q0);
            if (D.thresh) Pu = sq2;
            const float *acc_v = q0 ? dV : nullptr, *acc_k = q0 ? dK : nullptr;               // later chunks accumulate
            transpose_to(Pu, Sp, sq1, qp, qn, (int)S, st);                                    // P^T [key][query of the chunk]
            gemm_nt(sq1, qp, dOt + q0, Sp, nullptr, acc_v, dV, 3 * E, S, hd, (int)qp, 0, st);             // dV += P^T dO
            gemm_nt(e2 + q0 * E + h * hd, E, qkv + 2 * E + h * hd, 3 * E, nullptr, nullptr, sq0, Sp, qn, (int)S, hd, 0, st);   // dP
            enc_softmax_bwd_rows<<<(unsigned)qn, 256, 0, st>>>(pc, sq0, Sp, (int)S, scale, D.at(0u), idx0, D.thresh, D.inv_keep);
            if (S > QC) gemm_nt_splitk(sq0, Sp, Kt, Sp, dQ + q0 * 3 * E, 3 * E, qn, hd, (int)Sp, sq1, st);   // dQ rows of the chunk
            else gemm_nt(sq0, Sp, Kt, Sp, nullptr, nullptr, dQ + q0 * 3 * E, 3 * E, qn, hd, (int)Sp, 0, st);
            transpose_to(sq0, Sp, sq1, qp, qn, (int)S, st);                                   // dS^T
            gemm_nt(sq1, qp, Qt + q0, Sp, nullptr, acc_k, dK, 3 * E, S, hd, (int)qp, 0, st);              // dK += dS^T Q
        }
    }
    // ---- in_proj ----
    colsum(dqkv, 3 * E, 3 * E, g_in_b);
    gemm_tn(dqkv, 3 * E, 3 * E, zs, E, E, g_in_w);                                          // [3E][E]
    gemm_nt_small(dqkv, 3 * E, sc + SC.wt_in, 3 * E, nullptr, e1, dz_in, S, E, 3 * E, 0, sc + SC.wgp, wgp_floats, st);  // + d u1 (residual)
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

