// det_head_kernels.hip - the discrete back half of the detector (reference object_detection/models.py:9 builds
// torchvision's fasterrcnn_resnet50_fpn; detector.py:84 calls it): RPN proposal selection, multi-scale
// RoIAlign, detection post-processing.  The dense parts (RPN head, fc6/fc7, predictors) run on the MFMA
// conv / GEMM kernels of conv_kernels.hip.  Everything here is HBM/latency-bound integer and fp32 work.
//
// Arithmetic that feeds a discrete decision (sort keys, thresholds, IoU tests) is written with explicit
// round-to-nearest intrinsics (no FMA contraction) so that identical inputs give identical decisions to the
// unfused fp32 restatement in oracle/detector_oracle.py.  PARITY UNPINNED vs torchvision.
//
// IMAGES OF A PASS: every kernel takes an image index from its grid (blockIdx.y, or folded into blockIdx.z / .x where stated) and
// works on that image's slice of every buffer: workspace buffers of consecutive images lie `wsb` BYTES apart (one workspace block
// per image, same layout in each), the caller's arrays are image-major.  One launch per stage serves the whole pass; a one-image
// call is the same launch with one image.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DET_MAX_LEVELS 5
#define DET_ANCHORS 3           // aspect ratios per position
#define DET_HEAD_C 16           // packed RPN head output channels: 3 objectness + 12 deltas + 1 pad

// image `i` of a buffer whose per-image slices lie `bytes` apart
template <typename T>
__device__ __forceinline__ T *det_img(T *p, size_t bytes, int i) { return (T *)((char *)p + (size_t)i * bytes); }

struct RpnLevels {
    const float *head[DET_MAX_LEVELS];   // [n_images][gh, gw, 16] fp32 NHWC
    int gh[DET_MAX_LEVELS], gw[DET_MAX_LEVELS];
    int sh[DET_MAX_LEVELS], sw[DET_MAX_LEVELS];   // integer strides int(padded / grid)
    int off[DET_MAX_LEVELS + 1];         // anchor offsets of the levels in the flat key array
    int coff[DET_MAX_LEVELS + 1];        // candidate offsets (per-level top-k) in the candidate array
    float base[DET_MAX_LEVELS][DET_ANCHORS][4];
    int n_levels;
};

// float -> unsigned with the same order; ~ of it sorts descending under an ascending radix sort
__device__ __forceinline__ unsigned det_orderable(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- RPN ------------------------------------------------------------------------------------------
// keys: (level << 32) | ~orderable(objectness)  -> one ascending radix sort groups the levels and orders
// each by descending objectness; the sort is stable, so ties keep anchor order (position-major, anchor-minor)
__global__ void __launch_bounds__(256) rpn_make_keys(const RpnLevels L, unsigned long long *keys, unsigned *vals, size_t wsb)
{
    const int total = L.off[L.n_levels];
    const int img = blockIdx.y;
    keys = det_img(keys, wsb, img); vals = det_img(vals, wsb, img);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int l = 0;
        while (l + 1 < L.n_levels && i >= L.off[l + 1]) ++l;
        const int idx = i - L.off[l];
        const int pos = idx / DET_ANCHORS, a = idx - pos * DET_ANCHORS;
        const float obj = L.head[l][((long)img * L.gh[l] * L.gw[l] + pos) * DET_HEAD_C + a];
        keys[i] = ((unsigned long long)l << 32) | (unsigned)(~det_orderable(obj));
        vals[i] = (unsigned)idx;
    }
}

// BoxCoder.decode_single, unfused fp32
__device__ __forceinline__ float4 det_decode(float4 box, float dx, float dy, float dw, float dh, float clipv)
{
    const float w = __fsub_rn(box.z, box.x), h = __fsub_rn(box.w, box.y);
    const float cx = __fadd_rn(box.x, __fmul_rn(0.5f, w)), cy = __fadd_rn(box.y, __fmul_rn(0.5f, h));
    dw = fminf(dw, clipv);
    dh = fminf(dh, clipv);
    const float pcx = __fadd_rn(__fmul_rn(dx, w), cx), pcy = __fadd_rn(__fmul_rn(dy, h), cy);
    const float pw = __fmul_rn(expf(dw), w), ph = __fmul_rn(expf(dh), h);
    return make_float4(__fsub_rn(pcx, __fmul_rn(0.5f, pw)), __fsub_rn(pcy, __fmul_rn(0.5f, ph)),
                       __fadd_rn(pcx, __fmul_rn(0.5f, pw)), __fadd_rn(pcy, __fmul_rn(0.5f, ph)));
}

__device__ __forceinline__ float4 det_clip(float4 b, float width, float height)
{
    return make_float4(fminf(fmaxf(b.x, 0.f), width), fminf(fmaxf(b.y, 0.f), height),
                       fminf(fmaxf(b.z, 0.f), width), fminf(fmaxf(b.w, 0.f), height));
}

// per-level top-k candidates: decode against the anchor, clip to the resized image, drop boxes below min_size.
// ckeys = ~orderable(score) (0xffffffff for dropped candidates: they sort last), cvals = candidate index
__global__ void __launch_bounds__(256) rpn_decode_topk(const RpnLevels L, const unsigned *sorted_vals, float4 *cbox,
                                                       float *cscore, unsigned *ckeys, unsigned *cvals,
                                                       int *n_valid, float img_w, float img_h, float min_size, float clipv,
                                                       size_t wsb)
{
    const int total = L.coff[L.n_levels];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= total) return;
    const int img = blockIdx.y;
    sorted_vals = det_img(sorted_vals, wsb, img); cbox = det_img(cbox, wsb, img); cscore = det_img(cscore, wsb, img);
    ckeys = det_img(ckeys, wsb, img); cvals = det_img(cvals, wsb, img); n_valid = det_img(n_valid, wsb, img);
    int l = 0;
    while (l + 1 < L.n_levels && c >= L.coff[l + 1]) ++l;
    const int r = c - L.coff[l];
    const int idx = (int)sorted_vals[L.off[l] + r];
    const int pos = idx / DET_ANCHORS, a = idx - pos * DET_ANCHORS;
    const int y = pos / L.gw[l], x = pos - y * L.gw[l];
    const float *o = L.head[l] + ((long)img * L.gh[l] * L.gw[l] + pos) * DET_HEAD_C;
    const float fx = __fmul_rn((float)x, (float)L.sw[l]), fy = __fmul_rn((float)y, (float)L.sh[l]);
    const float4 anchor = make_float4(__fadd_rn(fx, L.base[l][a][0]), __fadd_rn(fy, L.base[l][a][1]),
                                      __fadd_rn(fx, L.base[l][a][2]), __fadd_rn(fy, L.base[l][a][3]));
    float4 b = det_decode(anchor, o[3 + 4 * a], o[4 + 4 * a], o[5 + 4 * a], o[6 + 4 * a], clipv);
    b = det_clip(b, img_w, img_h);
    const float score = o[a];
    const bool ok = __fsub_rn(b.z, b.x) >= min_size && __fsub_rn(b.w, b.y) >= min_size;
    cbox[c] = b;
    cscore[c] = score;
    ckeys[c] = ok ? ~det_orderable(score) : 0xffffffffu;
    cvals[c] = (unsigned)c;
    if (ok) atomicAdd(n_valid, 1);
}

__global__ void __launch_bounds__(256) det_gather_sorted(const unsigned *order, const float4 *cbox, const int *cgroup,
                                                         const float *cscore, float4 *sbox, int *sgroup, float *sscore,
                                                         const int *n_valid, int cap, size_t wsb)
{
    const int img = blockIdx.y;
    order = det_img(order, wsb, img); cbox = det_img(cbox, wsb, img); cgroup = det_img(cgroup, wsb, img);
    cscore = det_img(cscore, wsb, img); sbox = det_img(sbox, wsb, img); sgroup = det_img(sgroup, wsb, img);
    sscore = det_img(sscore, wsb, img); n_valid = det_img(n_valid, wsb, img);
    const int n = min(*n_valid, cap);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned c = order[i];
        sbox[i] = cbox[c];
        sgroup[i] = cgroup[c];
        sscore[i] = cscore[c];
    }
}

// ---- NMS ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long det_readlane64(unsigned long long v, int lane)
{
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

// greedy pass inside one chunk of 64 sorted boxes: lane b of the calling wave holds dg = suppression bits of box
// b against the chunk (diagonal mask block), r = the chunk's already-suppressed bits.  Everything is wave-uniform
// (readlane -> SGPRs), so the 64-step dependent chain runs on the scalar unit instead of through LDS round trips.
__device__ __forceinline__ unsigned long long det_resolve_chunk(unsigned long long dg, unsigned long long r, int lim,
                                                                int budget)
{
    if (lim < 64) r |= ~0ull << lim;          // boxes past the end are never kept
    // a box's own bit is final once its turn comes (rows only carry bits of LATER boxes), so the kept set is
    // simply the zero bits of r after the pass: per step one bit test, one select, one OR
#pragma unroll
    for (int b = 0; b < 64; ++b) {
        const unsigned long long d = det_readlane64(dg, b);
        r |= ((r >> b) & 1ull) ? 0ull : d;
    }
    unsigned long long kb = ~r;
    while (__popcll(kb) > budget) kb &= ~(1ull << (63 - __clzll((long long)kb)));   // only when the quota fills up
    return kb;
}

// suppression bits of box i against boxes j > i of the same group, 64 columns per word; only the upper
// triangle of 64x64 blocks is produced (and read).  IoU test is the strict ">" of torchvision's CUDA kernel.
// The number of boxes is only known on the device: gridDim.x workgroups per image walk the (row block, column block) pairs of the
// upper triangle that exist (a grid sized for the capacity would be ~44 000 workgroups per image, nearly all of them empty).
__global__ void __launch_bounds__(64) nms_mask(const float4 *sbox, const int *sgroup, const int *n_valid, int cap,
                                               float thresh, unsigned long long *mask, int nw, size_t wsb)
{
    const int img = blockIdx.y;
    sbox = det_img(sbox, wsb, img); sgroup = det_img(sgroup, wsb, img); n_valid = det_img(n_valid, wsb, img);
    mask = det_img(mask, wsb, img);
    const int n = min(*n_valid, cap);
    const int nbv = (n + 63) >> 6;
    const int npairs = nbv * (nbv + 1) / 2;
    __shared__ float4 cbx[64];
    __shared__ int cg[64];
    const int tid = threadIdx.x;
    for (int pidx = blockIdx.x; pidx < npairs; pidx += gridDim.x) {
        // row block rb starts at pair rb * nbv - rb (rb - 1) / 2 of the row-major upper triangle
        int rb = (int)(((float)(2 * nbv + 1) - sqrtf((float)(2 * nbv + 1) * (float)(2 * nbv + 1) - 8.0f * (float)pidx)) * 0.5f);
        rb = max(0, min(rb, nbv - 1));
        while (rb > 0 && rb * nbv - rb * (rb - 1) / 2 > pidx) --rb;
        while (rb + 1 < nbv && (rb + 1) * nbv - (rb + 1) * rb / 2 <= pidx) ++rb;
        const int cb = rb + pidx - (rb * nbv - rb * (rb - 1) / 2);
        const int j0 = cb * 64;
        __syncthreads();
        if (j0 + tid < n) {
            cbx[tid] = sbox[j0 + tid];
            cg[tid] = sgroup[j0 + tid];
        }
        __syncthreads();
        const int i = rb * 64 + tid;
        if (i >= n) continue;
        const float4 b = sbox[i];
        const int g = sgroup[i];
        const float area = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
        unsigned long long bits = 0;
        const int lim = min(64, n - j0);
        for (int t = 0; t < lim; ++t) {
            if (j0 + t <= i || cg[t] != g) continue;
            const float4 o = cbx[t];
            const float iw = fmaxf(__fsub_rn(fminf(b.z, o.z), fmaxf(b.x, o.x)), 0.f);
            const float ih = fmaxf(__fsub_rn(fminf(b.w, o.w), fmaxf(b.y, o.y)), 0.f);
            const float inter = __fmul_rn(iw, ih);
            const float oarea = __fmul_rn(__fsub_rn(o.z, o.x), __fsub_rn(o.w, o.y));
            const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area, oarea), inter));
            if (iou > thresh) bits |= 1ull << t;
        }
        mask[(size_t)i * nw + cb] = bits;
    }
}

// greedy pass over the sorted boxes, one workgroup: thread w owns word w of the "suppressed" bit vector.
// Per chunk of 64 boxes the owner of the chunk's word resolves the in-chunk dependencies from the diagonal
// mask block, then every later word ORs in the rows of the boxes that were kept.  Stops at max_keep.
__global__ void __launch_bounds__(512) nms_scan(const unsigned long long *mask, int nw, const int *n_valid, int cap,
                                                 int max_keep, int *kept, int *n_kept, size_t wsb)
{
    const int img = blockIdx.x;
    mask = det_img(mask, wsb, img); n_valid = det_img(n_valid, wsb, img); kept = det_img(kept, wsb, img);
    n_kept = det_img(n_kept, wsb, img);
    __shared__ unsigned long long chunk_rem;
    __shared__ unsigned long long keepbits;
    __shared__ int kept_total;
    const int tid = threadIdx.x;
    const int n = min(*n_valid, cap);
    const int nwords = (n + 63) >> 6;
    if (tid == 0) kept_total = 0;
    unsigned long long rem = 0;
    __syncthreads();
    for (int cw = 0; cw < nwords; ++cw) {
        // the 64 rows of this chunk, word `tid`: fetched unconditionally and up front (64 independent loads in
        // flight while the owner resolves the chunk) instead of one dependent load per kept box
        unsigned long long rows[64];
        const bool later = tid > cw && tid < nwords;
        if (later) {
#pragma unroll
            for (int b = 0; b < 64; ++b) {
                const int i = cw * 64 + b;
                rows[b] = i < n ? mask[(size_t)i * nw + tid] : 0ull;
            }
        }
        unsigned long long dg = 0;
        if (tid < 64) {
            const int i = cw * 64 + tid;
            dg = i < n ? mask[(size_t)i * nw + cw] : 0ull;
        }
        if (tid == cw) chunk_rem = rem;
        __syncthreads();
        const int base = kept_total;
        if (tid < 64) {   // wave 0 resolves the chunk on the scalar unit
            const unsigned long long kb0 = det_resolve_chunk(dg, chunk_rem, min(64, n - cw * 64), max_keep - base);
            if (tid == 0) keepbits = kb0;
        }
        __syncthreads();
        const unsigned long long kb = keepbits;
        if (tid < 64 && ((kb >> tid) & 1ull)) kept[base + __popcll(kb & ((1ull << tid) - 1ull))] = cw * 64 + tid;
        if (later) {
#pragma unroll
            for (int b = 0; b < 64; ++b) rem |= ((kb >> b) & 1ull) ? rows[b] : 0ull;
        }
        const int total = base + __popcll(kb);
        __syncthreads();
        if (tid == 0) kept_total = total;
        if (total >= max_keep) break;
    }
    __syncthreads();
    if (tid == 0) *n_kept = kept_total;
}

// ---- RPN: per-level NMS ---------------------------------------------------------------------------
// batched_nms only lets boxes of one level interact, and the candidates already sit level-major in descending
// objectness (the first sort), so the RPN runs n_levels independent small problems side by side: block-diagonal
// masks (<= pre_nms_top_n^2 / 2 pairs per level instead of one 4.7k x 4.7k triangle) and one wave per level for
// the greedy pass.  Boxes dropped for size carry key 0xffffffff and start out suppressed.
__global__ void __launch_bounds__(64) rpn_nms_mask(const RpnLevels L, const float4 *cbox, const unsigned *ckeys,
                                                   float thresh, unsigned long long *mask, int nw, size_t wsb)
{
    const int img = blockIdx.z / L.n_levels, l = blockIdx.z - img * L.n_levels;
    cbox = det_img(cbox, wsb, img); ckeys = det_img(ckeys, wsb, img); mask = det_img(mask, wsb, img);
    const int c0 = L.coff[l], n = L.coff[l + 1] - c0;
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
    __shared__ float4 cbx[64];
    __shared__ unsigned ck[64];
    const int tid = threadIdx.x;
    const int j0 = cb * 64;
    if (j0 + tid < n) {
        cbx[tid] = cbox[c0 + j0 + tid];
        ck[tid] = ckeys[c0 + j0 + tid];
    }
    __syncthreads();
    const int i = rb * 64 + tid;
    if (i >= n) return;
    const float4 b = cbox[c0 + i];
    const float area = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
    unsigned long long bits = 0;
    const int lim = min(64, n - j0);
    if (ckeys[c0 + i] != 0xffffffffu)
        for (int t = 0; t < lim; ++t) {
            if (j0 + t <= i || ck[t] == 0xffffffffu) continue;
            const float4 o = cbx[t];
            const float iw = fmaxf(__fsub_rn(fminf(b.z, o.z), fmaxf(b.x, o.x)), 0.f);
            const float ih = fmaxf(__fsub_rn(fminf(b.w, o.w), fmaxf(b.y, o.y)), 0.f);
            const float inter = __fmul_rn(iw, ih);
            const float oarea = __fmul_rn(__fsub_rn(o.z, o.x), __fsub_rn(o.w, o.y));
            const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area, oarea), inter));
            if (iou > thresh) bits |= 1ull << t;
        }
    mask[((size_t)c0 + i) * nw + cb] = bits;
}

// one wave per level (needs n_l <= 64 * 64); lane w owns word w of the level's suppressed vector.  Writes the
// final sort key of every candidate: ~orderable(score) if kept, 0xffffffff otherwise, and counts the kept.
__global__ void __launch_bounds__(64) rpn_nms_scan(const RpnLevels L, const unsigned long long *mask, int nw,
                                                   const unsigned *ckeys, int max_keep, unsigned *fkeys, int *n_kept, size_t wsb)
{
    const int l = blockIdx.x, tid = threadIdx.x, img = blockIdx.y;
    mask = det_img(mask, wsb, img); ckeys = det_img(ckeys, wsb, img); fkeys = det_img(fkeys, wsb, img);
    n_kept = det_img(n_kept, wsb, img);
    const int c0 = L.coff[l], n = L.coff[l + 1] - c0;
    const int nwords = (n + 63) >> 6;
    unsigned long long rem = 0;
    for (int wd = 0; wd < nwords; ++wd) {   // word wd = ballot over the 64 boxes of chunk wd (coalesced key reads)
        const int i = wd * 64 + tid;
        const unsigned long long bad = __ballot(i >= n || ckeys[c0 + min(i, n - 1)] == 0xffffffffu);
        if (tid == wd) rem = bad;
    }
    int total = 0;
    // rows of the chunk, word `tid`: fetched unconditionally one chunk AHEAD (64 independent loads in flight while
    // the current chunk is resolved), then selected by the keep bits - not one dependent load per kept box
    unsigned long long rows[64], nxt[64];
    auto fetch = [&](int cw, unsigned long long (&dst)[64]) {
        if (tid > cw && tid < nwords) {
#pragma unroll
            for (int b = 0; b < 64; ++b) dst[b] = cw * 64 + b < n ? mask[((size_t)c0 + cw * 64 + b) * nw + tid] : 0ull;
        }
    };
    fetch(0, rows);
    for (int cw = 0; cw < nwords && total < max_keep; ++cw) {
        const int i = cw * 64 + tid;
        const unsigned long long dg = i < n ? mask[((size_t)c0 + i) * nw + cw] : 0ull;
        if (cw + 1 < nwords) fetch(cw + 1, nxt);
        const unsigned long long kb = det_resolve_chunk(dg, det_readlane64(rem, cw), min(64, n - cw * 64), max_keep - total);
        if (i < n) fkeys[c0 + i] = ((kb >> tid) & 1ull) ? ckeys[c0 + i] : 0xffffffffu;
        if (tid > cw && tid < nwords) {
#pragma unroll
            for (int b = 0; b < 64; ++b) rem |= ((kb >> b) & 1ull) ? rows[b] : 0ull;
        }
#pragma unroll
        for (int b = 0; b < 64; ++b) rows[b] = nxt[b];
        total += __popcll(kb);
    }
    // chunks never visited (max_keep reached) keep the 0xffffffff the host pre-filled fkeys with
    if (tid == 0 && total) atomicAdd(n_kept, total);
}

__global__ void __launch_bounds__(256) rpn_emit_sorted(const unsigned *order, const int *n_kept, int max_out,
                                                       const float4 *cbox, const float *cscore, float4 *proposals,
                                                       float *scores, int *count, size_t wsb)
{
    const int img = blockIdx.y;
    order = det_img(order, wsb, img); n_kept = det_img(n_kept, wsb, img); cbox = det_img(cbox, wsb, img);
    cscore = det_img(cscore, wsb, img);
    proposals += (long)img * max_out; count += img;
    if (scores) scores += (long)img * max_out;
    const int n = min(*n_kept, max_out);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < max_out; i += gridDim.x * 256) {
        proposals[i] = i < n ? cbox[order[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (scores) scores[i] = i < n ? cscore[order[i]] : 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *count = n;
}

// per image: `nzero` counters zeroed, `nfill` words set to 0xffffffff (the keys of candidates no scan visits)
__global__ void __launch_bounds__(256) det_stage_init(int *counters, int nzero, unsigned *fill, int nfill, size_t wsb)
{
    const int img = blockIdx.y;
    counters = det_img(counters, wsb, img);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nzero) counters[i] = 0;
    if (fill) {
        fill = det_img(fill, wsb, img);
        for (int q = i; q < nfill; q += gridDim.x * 256) fill[q] = 0xffffffffu;
    }
}

// ---- MultiScaleRoIAlign ---------------------------------------------------------------------------
struct RoiLevels {
    const float *feat[4];    // [n_images][fh, fw, C] fp32 NHWC
    int fh[4], fw[4];
    float scale[4];
    int C;
};

// legacy (non-"aligned") roi_align bilinear sample, channel group c4 (float4) of one NHWC map
__device__ __forceinline__ float4 roi_sample(const float4 *f, int H, int W, int C4, float y, float x, int c4)
{
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return make_float4(0.f, 0.f, 0.f, 0.f);
    y = fmaxf(y, 0.f);
    x = fmaxf(x, 0.f);
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    const float4 v1 = f[((long)yl * W + xl) * C4 + c4], v2 = f[((long)yl * W + xh) * C4 + c4];
    const float4 v3 = f[((long)yh * W + xl) * C4 + c4], v4 = f[((long)yh * W + xh) * C4 + c4];
    return make_float4(w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x, w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y,
                       w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z, w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w);
}

// one workgroup per (roi, image); thread = (bin group, float4 channel group); out [n_images][max_rois, 7, 7, C], rows >= count zero
__global__ void __launch_bounds__(256) roi_align_levels(const RoiLevels L, const float4 *rois, const int *count,
                                                        float4 *out)
{
    const int r = blockIdx.x, img = blockIdx.y;
    const int C4 = L.C >> 2;
    const int tid = threadIdx.x;
    float4 *o = out + ((long)img * gridDim.x + r) * 49 * C4;
    rois += (long)img * gridDim.x; count += img;
    if (r >= *count) {
        for (int i = tid; i < 49 * C4; i += 256) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4 b = rois[r];
    // LevelMapper(k_min 2, k_max 5, canonical 224 @ level 4, eps 1e-6)
    const float s = sqrtf(__fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y)));
    float t = floorf(__fadd_rn(4.f, log2f(__fadd_rn(__fdiv_rn(s, 224.f), 1e-6f))));
    t = fminf(fmaxf(t, 2.f), 5.f);
    const int l = (int)t - 2;
    const float sc = L.scale[l];
    const int H = L.fh[l], W = L.fw[l];
    const float4 *f = (const float4 *)L.feat[l] + (long)img * H * W * C4;
    const float x0 = b.x * sc, y0 = b.y * sc;
    const float rw = fmaxf(b.z * sc - x0, 1.f), rh = fmaxf(b.w * sc - y0, 1.f);
    const float bw = rw / 7.f, bh = rh / 7.f;
    const int groups = 256 / C4 > 0 ? 256 / C4 : 1;
    const int c4 = tid % C4, grp = tid / C4;
    if (grp >= groups) return;
    for (int bin = grp; bin < 49; bin += groups) {
        const int ph = bin / 7, pw = bin - ph * 7;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int iy = 0; iy < 2; ++iy) {
            const float y = y0 + (float)ph * bh + ((float)iy + 0.5f) * bh / 2.f;
#pragma unroll
            for (int ix = 0; ix < 2; ++ix) {
                const float x = x0 + (float)pw * bw + ((float)ix + 0.5f) * bw / 2.f;
                const float4 v = roi_sample(f, H, W, C4, y, x, c4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        o[bin * C4 + c4] = make_float4(acc.x * 0.25f, acc.y * 0.25f, acc.z * 0.25f, acc.w * 0.25f);
    }
}

// ---- detections -----------------------------------------------------------------------------------
// one workgroup per roi: softmax over the classes, per-class decode (weights 10,10,5,5), clip, thresholds.
// Candidate id = roi * (NC-1) + (cls-1); dropped candidates get key 0xffffffff.
__global__ void __launch_bounds__(256) det_score_boxes(const float *logits, const float *reg, const float4 *rois,
                                                       const int *count, int NC, float img_w, float img_h,
                                                       float score_thresh, float min_size, float clipv, float4 *cbox,
                                                       int *cgroup, float *cscore, unsigned *ckeys, unsigned *cvals,
                                                       int *n_valid, size_t wsb, int ls, int rs)
{
    __shared__ float red[256];
    const int r = blockIdx.x, tid = threadIdx.x, img = blockIdx.y;
    const int ncand = NC - 1;
    logits += (long)img * gridDim.x * ls; reg += (long)img * gridDim.x * rs; rois += (long)img * gridDim.x; count += img;   // ls, rs: row strides
    cbox = det_img(cbox, wsb, img); cgroup = det_img(cgroup, wsb, img); cscore = det_img(cscore, wsb, img);
    ckeys = det_img(ckeys, wsb, img); cvals = det_img(cvals, wsb, img); n_valid = det_img(n_valid, wsb, img);
    const bool live = r < *count;
    const float *z = logits + (long)r * ls;
    float m = -INFINITY;
    for (int c = tid; c < NC; c += 256) m = fmaxf(m, z[c]);
    red[tid] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    m = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int c = tid; c < NC; c += 256) sum += expf(z[c] - m);
    red[tid] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    sum = red[0];
    const float4 box = rois[r];
    for (int c = 1 + tid; c < NC; c += 256) {
        const long id = (long)r * ncand + (c - 1);
        const float score = __fdiv_rn(expf(z[c] - m), sum);
        const float *d = reg + (long)r * rs + c * 4;
        float4 b = det_decode(box, __fdiv_rn(d[0], 10.f), __fdiv_rn(d[1], 10.f), __fdiv_rn(d[2], 5.f),
                              __fdiv_rn(d[3], 5.f), clipv);
        b = det_clip(b, img_w, img_h);
        const bool ok = live && score > score_thresh && __fsub_rn(b.z, b.x) >= min_size && __fsub_rn(b.w, b.y) >= min_size;
        cbox[id] = b;
        cgroup[id] = c;
        cscore[id] = score;
        ckeys[id] = ok ? ~det_orderable(score) : 0xffffffffu;
        cvals[id] = (unsigned)id;
        if (ok) atomicAdd(n_valid, 1);
    }
}

__global__ void __launch_bounds__(128) det_emit(const int *kept, const int *n_kept, int max_det, const float4 *sbox,
                                                const int *sgroup, const float *sscore, float ratio_w, float ratio_h,
                                                float4 *boxes, float *scores, long long *labels, int *n_det, size_t wsb)
{
    const int img = blockIdx.x;
    kept = det_img(kept, wsb, img); n_kept = det_img(n_kept, wsb, img); sbox = det_img(sbox, wsb, img);
    sgroup = det_img(sgroup, wsb, img); sscore = det_img(sscore, wsb, img);
    boxes += (long)img * max_det; scores += (long)img * max_det; labels += (long)img * max_det; n_det += img;
    const int n = min(*n_kept, max_det);
    for (int i = threadIdx.x; i < max_det; i += 128) {
        if (i < n) {
            const float4 b = sbox[kept[i]];
            boxes[i] = make_float4(__fmul_rn(b.x, ratio_w), __fmul_rn(b.y, ratio_h), __fmul_rn(b.z, ratio_w), __fmul_rn(b.w, ratio_h));
            scores[i] = sscore[kept[i]];
            labels[i] = sgroup[kept[i]];
        } else {
            boxes[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            scores[i] = 0.f;
            labels[i] = 0;
        }
    }
    if (threadIdx.x == 0) *n_det = n;
}
