// opdet_abi.hip - host side of the detector's proposal / RoI / detection stages (second translation unit of
// libopnet_hip.so; kernels: det_head_kernels.hip, the radix sort of det_sort_kernels.hip).
#include "det_head_kernels.hip"
#include "det_sort_kernels.hip"

#include <string.h>
#include <atomic>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/opnet_hip.h"

int opnet_set_error(int code, const char *msg);   // opnet_abi.hip

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return opnet_set_error(code, buf);
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(OPNET_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- stable radix sort of (key, value) pairs (det_sort_kernels.hip) ----------------------------------------------------------------
// scratch: the digit histograms of the tiles ([tiles][512] words at most)
static size_t sort_scratch_bytes(size_t n) { return up256(512 * ((n + RS_TILE - 1) / RS_TILE) * 4); }
// Sorts ascending by the low `bits` bits of the keys.  (kin, vin) and (kout, vout) are both overwritten (ping-pong); the result is in
// (kout, vout) when the return value is 1 and in (kin, vin) when it is 0 (an even number of passes).  4-byte keys, n <= RS_SMALL_MAX:
// one workgroup, result always in (kout, vout); otherwise two launches per pass of 8 (4-byte keys) or 9 bits.
// nseg arrays of n pairs each are sorted by the same launches; every buffer (scratch included) of segment i lies i * seg BYTES after
// segment 0's.
template <typename K>
static int sort_pairs(K *kin, unsigned *vin, K *kout, unsigned *vout, size_t n, int bits, void *scratch, hipStream_t st, int nseg = 1,
                      size_t seg = 0)
{
    if constexpr (sizeof(K) == 4) {
        if (n <= RS_SMALL_MAX) {
            const size_t lds = (size_t)RS_SMALL_MAX * 16;
            static std::atomic<unsigned long long> raised{0};      // (> 64 KB of dynamic LDS has to be asked for once per device)
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (dev < 0 || dev >= 64 || !((raised.load(std::memory_order_relaxed) >> dev) & 1ull)) {
                (void)hipFuncSetAttribute((const void *)rs_sort_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (dev >= 0 && dev < 64) raised.fetch_or(1ull << dev, std::memory_order_relaxed);
            }
            rs_sort_small<<<nseg, 1024, lds, st>>>(kin, vin, kout, vout, (int)n, bits, seg);
            return 1;
        }
    }
    constexpr int RB = sizeof(K) == 8 ? 9 : 8;
    const size_t ntiles = (n + RS_TILE - 1) / RS_TILE;
    unsigned *hist = (unsigned *)scratch;
    int where = 0;
    for (int shift = 0; shift < bits; shift += RB) {
        rs_pass_count<K, RB><<<dim3((unsigned)ntiles, nseg), 256, 0, st>>>(kin, vin, (long)n, bits, shift, hist, seg);
        rs_pass_scatter<K, RB><<<dim3((unsigned)ntiles, nseg), 256, 0, st>>>(kin, vin, kout, vout, (long)n, bits, shift, hist, seg);
        K *tk = kin; kin = kout; kout = tk;
        unsigned *tv = vin; vin = vout; vout = tv;
        where ^= 1;
    }
    return where;
}

/* TEST-ONLY (tests/test_detector_sort_gpu.py): the sort by itself.  keys: 8-byte (key64 != 0) or 4-byte; both buffer pairs hold n
 * elements; *result_in_out = 1: sorted pairs in (keys_out, vals_out), 0: in (keys_in, vals_in). */
extern "C" int opdet_test_sort_pairs(void *keys_in, unsigned *vals_in, void *keys_out, unsigned *vals_out, long n, int bits, int key64,
                                     void *scratch, size_t scratch_bytes, int *result_in_out, void *stream)
{
    if (!keys_in || !vals_in || !keys_out || !vals_out || !scratch || !result_in_out) return fail(OPNET_EINVAL, "null pointer");
    if (n <= 0 || n > 0x7fffffffL || bits <= 0 || bits > (key64 ? 64 : 32)) return fail(OPNET_ESHAPE, "bad sort sizes");
    if (scratch_bytes < sort_scratch_bytes((size_t)n)) return fail(OPNET_EWORKSPACE, "scratch too small");
    *result_in_out = key64 ? sort_pairs<unsigned long long>((unsigned long long *)keys_in, vals_in, (unsigned long long *)keys_out, vals_out,
                                                             (size_t)n, bits, scratch, (hipStream_t)stream)
                           : sort_pairs<unsigned>((unsigned *)keys_in, vals_in, (unsigned *)keys_out, vals_out, (size_t)n, bits, scratch,
                                                  (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}
extern "C" size_t opdet_test_sort_scratch_bytes(long n) { return n > 0 ? sort_scratch_bytes((size_t)n) : 0; }
static const float kBoxClip = 4.135166556742356f;   // log(1000 / 16), BoxCoder.bbox_xform_clip
#define DET_MAX_IMAGES 64                           // images of one batched call (grid.y / grid.z of the stage launches)

// ---- shared NMS stage: sorted boxes -> kept indices -------------------------------------------------
struct NmsBuffers { float4 *sbox; int *sgroup; float *sscore; unsigned long long *mask; int *kept; int *n_kept; };

static size_t nms_bytes(int cap)
{
    const size_t nw = (cap + 63) / 64;
    return up256((size_t)cap * 16) + up256((size_t)cap * 4) * 2 + up256((size_t)cap * nw * 8) + up256((size_t)cap * 4) + 256;
}

static char *carve_nms(char *p, int cap, NmsBuffers *b)
{
    const size_t nw = (cap + 63) / 64;
    b->sbox = (float4 *)p;  p += up256((size_t)cap * 16);
    b->sgroup = (int *)p;   p += up256((size_t)cap * 4);
    b->sscore = (float *)p; p += up256((size_t)cap * 4);
    b->mask = (unsigned long long *)p; p += up256((size_t)cap * nw * 8);
    b->kept = (int *)p;     p += up256((size_t)cap * 4);
    b->n_kept = (int *)p;   p += 256;
    return p;
}

static int run_nms(const NmsBuffers &b, const int *n_valid, int cap, float thresh, int max_keep, int n_images, size_t wsb, hipStream_t st)
{
    const int nb = (cap + 63) / 64;
    if (nb > 512) return fail(OPNET_ESHAPE, "NMS over more than 32768 candidates is not supported (%d)", cap);
    // one wave per workgroup; ~8 192 of them walk the block pairs that exist (up to nb (nb + 1) / 2 = 44 000 per image at the capacity)
    const int per_image = 8192 / n_images > 256 ? 8192 / n_images : 256;
    nms_mask<<<dim3(per_image, n_images), 64, 0, st>>>(b.sbox, b.sgroup, n_valid, cap, thresh, b.mask, nb, wsb);
    const int threads = ((nb + 63) / 64) * 64;
    nms_scan<<<n_images, threads, 0, st>>>(b.mask, nb, n_valid, cap, max_keep, b.kept, b.n_kept, wsb);
    return OPNET_OK;
}

// ---- RPN proposals ---------------------------------------------------------------------------------
struct RpnPlan { RpnLevels L; int total, ncand, nb; size_t sort1, sort2; };

static int rpn_plan(RpnPlan *P, int n_levels, const int *gh, const int *gw, const int *anchor_sizes, int padded_h,
                    int padded_w, int pre_nms_top_n)
{
    if (n_levels < 1 || n_levels > DET_MAX_LEVELS || !gh || !gw || !anchor_sizes)
        return fail(OPNET_ESHAPE, "1..%d feature levels", DET_MAX_LEVELS);
    if (pre_nms_top_n <= 0 || padded_h <= 0 || padded_w <= 0) return fail(OPNET_ESHAPE, "bad RPN sizes");
    RpnLevels &L = P->L;
    memset(&L, 0, sizeof(L));
    L.n_levels = n_levels;
    int off = 0, coff = 0;
    for (int l = 0; l < n_levels; ++l) {
        if (gh[l] <= 0 || gw[l] <= 0) return fail(OPNET_ESHAPE, "empty feature level %d", l);
        L.gh[l] = gh[l]; L.gw[l] = gw[l];
        L.sh[l] = padded_h / gh[l];       // int(image / grid), AnchorGenerator.forward
        L.sw[l] = padded_w / gw[l];
        L.off[l] = off; L.coff[l] = coff;
        const int na = gh[l] * gw[l] * DET_ANCHORS;
        off += na;
        coff += na < pre_nms_top_n ? na : pre_nms_top_n;
        // AnchorGenerator.generate_anchors in fp32, aspect ratios (0.5, 1, 2), round half to even
        const float ratios[DET_ANCHORS] = {0.5f, 1.0f, 2.0f};
        for (int a = 0; a < DET_ANCHORS; ++a) {
            const float hr = sqrtf(ratios[a]), wr = 1.0f / hr;
            const float ws = wr * (float)anchor_sizes[l], hs = hr * (float)anchor_sizes[l];
            L.base[l][a][0] = rintf(-ws / 2.0f); L.base[l][a][1] = rintf(-hs / 2.0f);
            L.base[l][a][2] = rintf(ws / 2.0f);  L.base[l][a][3] = rintf(hs / 2.0f);
        }
    }
    L.off[n_levels] = off; L.coff[n_levels] = coff;
    P->total = off; P->ncand = coff;
    int nmax = 0;
    for (int l = 0; l < n_levels; ++l) nmax = L.coff[l + 1] - L.coff[l] > nmax ? L.coff[l + 1] - L.coff[l] : nmax;
    P->nb = (nmax + 63) / 64;
    if (P->nb > 64) return fail(OPNET_ESHAPE, "pre_nms_top_n above 4096 is not supported");
    P->sort1 = sort_scratch_bytes((size_t)off); P->sort2 = sort_scratch_bytes((size_t)coff);
    return OPNET_OK;
}

static size_t rpn_bytes(const RpnPlan &P)
{
    const size_t n = P.total, c = P.ncand;
    return up256(n * 8) * 2 + up256(n * 4) * 2 + up256(P.sort1) + up256(c * 16) + up256(c * 4) * 6 + up256(P.sort2) + 256 +
           up256(c * P.nb * 8);
}

extern "C" size_t opdet_rpn_workspace_bytes_batch(int n_images, int n_levels, const int *gh, const int *gw, const int *anchor_sizes,
                                                  int padded_h, int padded_w, int pre_nms_top_n)
{
    RpnPlan P;
    if (n_images < 1 || n_images > DET_MAX_IMAGES) { fail(OPNET_ESHAPE, "1..%d images per call", DET_MAX_IMAGES); return 0; }
    if (rpn_plan(&P, n_levels, gh, gw, anchor_sizes, padded_h, padded_w, pre_nms_top_n)) return 0;
    return (size_t)n_images * up256(rpn_bytes(P));
}
extern "C" size_t opdet_rpn_workspace_bytes(int n_levels, const int *gh, const int *gw, const int *anchor_sizes,
                                            int padded_h, int padded_w, int pre_nms_top_n)
{
    return opdet_rpn_workspace_bytes_batch(1, n_levels, gh, gw, anchor_sizes, padded_h, padded_w, pre_nms_top_n);
}

extern "C" int opdet_rpn_proposals_batch_f32(const float *const *head_out, int n_images, int n_levels, const int *gh, const int *gw,
                                             const int *anchor_sizes, int image_h, int image_w, int padded_h, int padded_w,
                                             int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float min_size,
                                             float *proposals, float *scores, int *count, void *workspace,
                                             size_t workspace_bytes, void *stream)
{
    if (!head_out || !proposals || !count || !workspace) return fail(OPNET_EINVAL, "null pointer");
    if (n_images < 1 || n_images > DET_MAX_IMAGES) return fail(OPNET_ESHAPE, "1..%d images per call", DET_MAX_IMAGES);
    if ((((uintptr_t)proposals) & 15u) || (((uintptr_t)workspace) & 255u))
        return fail(OPNET_EINVAL, "proposals must be 16-byte and workspace 256-byte aligned");
    if (post_nms_top_n <= 0 || image_h <= 0 || image_w <= 0) return fail(OPNET_ESHAPE, "bad RPN sizes");
    RpnPlan P;
    if (int rc = rpn_plan(&P, n_levels, gh, gw, anchor_sizes, padded_h, padded_w, pre_nms_top_n)) return rc;
    const size_t wsb = up256(rpn_bytes(P));          // one workspace block per image
    if (workspace_bytes < n_images * wsb) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", workspace_bytes, n_images * wsb);
    for (int l = 0; l < n_levels; ++l) {
        if (!head_out[l]) return fail(OPNET_EINVAL, "null pointer");
        P.L.head[l] = head_out[l];
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t n = P.total, c = P.ncand;
    char *p = (char *)workspace;
    unsigned long long *k_in = (unsigned long long *)p;  p += up256(n * 8);
    unsigned long long *k_out = (unsigned long long *)p; p += up256(n * 8);
    unsigned *v_in = (unsigned *)p;  p += up256(n * 4);
    unsigned *v_out = (unsigned *)p; p += up256(n * 4);
    void *tmp1 = p; p += up256(P.sort1);
    float4 *cbox = (float4 *)p; p += up256(c * 16);
    float *cscore = (float *)p; p += up256(c * 4);
    unsigned *ck_in = (unsigned *)p;  p += up256(c * 4);
    unsigned *ck_out = (unsigned *)p; p += up256(c * 4);
    unsigned *cv_in = (unsigned *)p;  p += up256(c * 4);
    unsigned *cv_out = (unsigned *)p; p += up256(c * 4);
    void *tmp2 = p; p += up256(P.sort2);
    unsigned *fkeys = (unsigned *)p; p += up256(c * 4);
    int *counters = (int *)p; p += 256;      // [0] boxes that pass the size test, [1] boxes kept by NMS
    unsigned long long *mask = (unsigned long long *)p;

    const unsigned ni = (unsigned)n_images;
    det_stage_init<<<dim3((unsigned)((c + 255) / 256), ni), 256, 0, st>>>(counters, 2, fkeys, (int)c, wsb);
    rpn_make_keys<<<dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256), ni), 256, 0, st>>>(P.L, k_in, v_in, wsb);
    const unsigned *v_sorted = sort_pairs<unsigned long long>(k_in, v_in, k_out, v_out, n, 35, tmp1, st, n_images, wsb) ? v_out : v_in;
    rpn_decode_topk<<<dim3((unsigned)((c + 255) / 256), ni), 256, 0, st>>>(P.L, v_sorted, cbox, cscore, ck_in, cv_in, counters,
                                                                          (float)image_w, (float)image_h, min_size, kBoxClip, wsb);
    // per-level NMS on the level-major, score-descending candidates, then ONE sort of the survivors by score
    rpn_nms_mask<<<dim3(P.nb, P.nb, n_levels * ni), 64, 0, st>>>(P.L, cbox, ck_in, nms_thresh, mask, P.nb, wsb);
    rpn_nms_scan<<<dim3(n_levels, ni), 64, 0, st>>>(P.L, mask, P.nb, ck_in, post_nms_top_n, fkeys, counters + 1, wsb);
    // (the survivors' keys are only needed in order of their values: fkeys / cv_in are spent here)
    const unsigned *cv_sorted = sort_pairs<unsigned>(fkeys, cv_in, ck_out, cv_out, c, 32, tmp2, st, n_images, wsb) ? cv_out : cv_in;
    rpn_emit_sorted<<<dim3((post_nms_top_n + 255) / 256, ni), 256, 0, st>>>(cv_sorted, counters + 1, post_nms_top_n, cbox, cscore,
                                                                           (float4 *)proposals, scores, count, wsb);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opdet_rpn_proposals_f32(const float *const *head_out, int n_levels, const int *gh, const int *gw,
                                       const int *anchor_sizes, int image_h, int image_w, int padded_h, int padded_w,
                                       int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float min_size,
                                       float *proposals, float *scores, int *count, void *workspace,
                                       size_t workspace_bytes, void *stream)
{
    return opdet_rpn_proposals_batch_f32(head_out, 1, n_levels, gh, gw, anchor_sizes, image_h, image_w, padded_h, padded_w, pre_nms_top_n,
                                         post_nms_top_n, nms_thresh, min_size, proposals, scores, count, workspace, workspace_bytes,
                                         stream);
}

// ---- MultiScaleRoIAlign ----------------------------------------------------------------------------
extern "C" int opdet_roi_align_batch_f32(const float *const *feats, int n_images, const int *fh, const int *fw, int C, int image_h,
                                         const float *rois, const int *count, int max_rois, float *out, void *stream)
{
    if (!feats || !fh || !fw || !rois || !count || !out) return fail(OPNET_EINVAL, "null pointer");
    if (n_images < 1 || n_images > DET_MAX_IMAGES) return fail(OPNET_ESHAPE, "1..%d images per call", DET_MAX_IMAGES);
    if ((((uintptr_t)rois) & 15u) || (((uintptr_t)out) & 15u)) return fail(OPNET_EINVAL, "rois / out must be 16-byte aligned");
    if (C <= 0 || (C & 3) || C > 1024 || 256 % (C / 4)) return fail(OPNET_ESHAPE, "C=%d: C/4 must divide 256", C);
    if (max_rois <= 0 || image_h <= 0) return fail(OPNET_ESHAPE, "bad roi_align sizes");
    RoiLevels L;
    L.C = C;
    for (int l = 0; l < 4; ++l) {
        if (!feats[l] || fh[l] <= 0 || fw[l] <= 0) return fail(OPNET_EINVAL, "bad feature level %d", l);
        L.feat[l] = feats[l]; L.fh[l] = fh[l]; L.fw[l] = fw[l];
        // MultiScaleRoIAlign.infer_scale: 2 ** round(log2(feature / image)) from the heights
        L.scale[l] = exp2f(rintf(log2f((float)fh[l] / (float)image_h)));
    }
    roi_align_levels<<<dim3(max_rois, n_images), 256, 0, (hipStream_t)stream>>>(L, (const float4 *)rois, count, (float4 *)out);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}
extern "C" int opdet_roi_align_f32(const float *const *feats, const int *fh, const int *fw, int C, int image_h,
                                   const float *rois, const int *count, int max_rois, float *out, void *stream)
{
    return opdet_roi_align_batch_f32(feats, 1, fh, fw, C, image_h, rois, count, max_rois, out, stream);
}

// ---- detections ------------------------------------------------------------------------------------
struct DetPlan { size_t ncand; int cap; size_t sort; };

static int det_plan(DetPlan *P, int max_rois, int num_classes)
{
    if (max_rois <= 0 || num_classes < 2) return fail(OPNET_ESHAPE, "bad detection sizes");
    P->ncand = (size_t)max_rois * (num_classes - 1);
    // scores of one roi sum to 1, so at most 19 classes per roi can clear any threshold >= 0.05
    size_t cap = (size_t)max_rois * 19 < P->ncand ? (size_t)max_rois * 19 : P->ncand;
    if (cap > 32768) cap = 32768;
    P->cap = (int)cap;
    P->sort = sort_scratch_bytes(P->ncand);
    return OPNET_OK;
}

static size_t det_bytes(const DetPlan &P)
{
    return up256(P.ncand * 16) + up256(P.ncand * 4) * 6 + up256(P.sort) + 256 + nms_bytes(P.cap);
}

extern "C" size_t opdet_detections_workspace_bytes_batch(int n_images, int max_rois, int num_classes)
{
    DetPlan P;
    if (n_images < 1 || n_images > DET_MAX_IMAGES) { fail(OPNET_ESHAPE, "1..%d images per call", DET_MAX_IMAGES); return 0; }
    if (det_plan(&P, max_rois, num_classes)) return 0;
    return (size_t)n_images * up256(det_bytes(P));
}
extern "C" size_t opdet_detections_workspace_bytes(int max_rois, int num_classes)
{
    return opdet_detections_workspace_bytes_batch(1, max_rois, num_classes);
}

extern "C" int opdet_detections_batch_f32(const float *class_logits, const float *box_regression, const float *proposals,
                                          const int *count, int n_images, int max_rois, int num_classes, int logits_stride,
                                          int reg_stride, int image_h, int image_w, int orig_h, int orig_w, float score_thresh,
                                          float nms_thresh, int max_det, float *boxes, float *scores, long long *labels, int *n_det,
                                          void *workspace, size_t workspace_bytes, void *stream)
{
    if (logits_stride == 0) logits_stride = num_classes;
    if (reg_stride == 0) reg_stride = 4 * num_classes;
    if (logits_stride < num_classes || reg_stride < 4 * num_classes) return fail(OPNET_ESHAPE, "row strides shorter than the rows");
    if (n_images < 1 || n_images > DET_MAX_IMAGES) return fail(OPNET_ESHAPE, "1..%d images per call", DET_MAX_IMAGES);
    if (!class_logits || !box_regression || !proposals || !count || !boxes || !scores || !labels || !n_det || !workspace)
        return fail(OPNET_EINVAL, "null pointer");
    if ((((uintptr_t)proposals) & 15u) || (((uintptr_t)boxes) & 15u) || (((uintptr_t)workspace) & 255u))
        return fail(OPNET_EINVAL, "proposals / boxes must be 16-byte and workspace 256-byte aligned");
    if (image_h <= 0 || image_w <= 0 || orig_h <= 0 || orig_w <= 0 || max_det <= 0) return fail(OPNET_ESHAPE, "bad sizes");
    if (score_thresh < 0.05f) return fail(OPNET_ESHAPE, "score_thresh below 0.05 is not supported (candidate bound)");
    DetPlan P;
    if (int rc = det_plan(&P, max_rois, num_classes)) return rc;
    const size_t wsb = up256(det_bytes(P));          // one workspace block per image
    if (workspace_bytes < n_images * wsb) return fail(OPNET_EWORKSPACE, "workspace %zu B < %zu B", workspace_bytes, n_images * wsb);
    hipStream_t st = (hipStream_t)stream;
    const size_t c = P.ncand;
    char *p = (char *)workspace;
    float4 *cbox = (float4 *)p; p += up256(c * 16);
    int *cgroup = (int *)p;     p += up256(c * 4);
    float *cscore = (float *)p; p += up256(c * 4);
    unsigned *ck_in = (unsigned *)p;  p += up256(c * 4);
    unsigned *ck_out = (unsigned *)p; p += up256(c * 4);
    unsigned *cv_in = (unsigned *)p;  p += up256(c * 4);
    unsigned *cv_out = (unsigned *)p; p += up256(c * 4);
    void *tmp = p; p += up256(P.sort);
    int *n_valid = (int *)p; p += 256;
    NmsBuffers nb;
    carve_nms(p, P.cap, &nb);

    const unsigned ni = (unsigned)n_images;
    det_stage_init<<<dim3(1, ni), 256, 0, st>>>(n_valid, 1, nullptr, 0, wsb);
    det_score_boxes<<<dim3(max_rois, ni), 256, 0, st>>>(class_logits, box_regression, (const float4 *)proposals, count, num_classes,
                                                       (float)image_w, (float)image_h, score_thresh, 1e-2f, kBoxClip, cbox, cgroup,
                                                       cscore, ck_in, cv_in, n_valid, wsb, logits_stride, reg_stride);
    const unsigned *cv_sorted = sort_pairs<unsigned>(ck_in, cv_in, ck_out, cv_out, c, 32, tmp, st, n_images, wsb) ? cv_out : cv_in;
    det_gather_sorted<<<dim3((unsigned)((P.cap + 255) / 256), ni), 256, 0, st>>>(cv_sorted, cbox, cgroup, cscore, nb.sbox, nb.sgroup,
                                                                                nb.sscore, n_valid, P.cap, wsb);
    if (int rc = run_nms(nb, n_valid, P.cap, nms_thresh, max_det, n_images, wsb, st)) return rc;
    // GeneralizedRCNNTransform.postprocess: boxes back to the original frame, ratio = float(orig) / float(resized)
    const float rw = (float)((double)orig_w / (double)image_w), rh = (float)((double)orig_h / (double)image_h);
    det_emit<<<ni, 128, 0, st>>>(nb.kept, nb.n_kept, max_det, nb.sbox, nb.sgroup, nb.sscore, rw, rh, (float4 *)boxes, scores,
                                 labels, n_det, wsb);
    HIP_TRY(hipGetLastError());
    return OPNET_OK;
}

extern "C" int opdet_detections_f32(const float *class_logits, const float *box_regression, const float *proposals,
                                    const int *count, int max_rois, int num_classes, int image_h, int image_w,
                                    int orig_h, int orig_w, float score_thresh, float nms_thresh, int max_det,
                                    float *boxes, float *scores, long long *labels, int *n_det, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    return opdet_detections_batch_f32(class_logits, box_regression, proposals, count, 1, max_rois, num_classes, 0, 0, image_h, image_w, orig_h,
                                      orig_w, score_thresh, nms_thresh, max_det, boxes, scores, labels, n_det, workspace, workspace_bytes,
                                      stream);
}
