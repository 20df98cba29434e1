// det_sort_kernels.hip - stable least-significant-digit radix sort of (key, value) pairs for the detector's selection stages
// (RegionProposalNetwork.filter_proposals: per-level top-k by objectness; RoIHeads.postprocess_detections: candidates by
// score - torchvision 0.5.0, called from baselines/detector.py:84).  It replaces the hipCUB calls of rounds 1-3: no
// vendor library is left on any path of libopnet_hip.so.  A stable sort has ONE result, so proposals and detections are what they
// were (tests/test_detector_gpu.py; tests/test_detector_sort_gpu.py holds the sort itself against torch.sort(stable=True)).
//
// Digits of RB bits (8 for 4-byte keys, 9 for the 35-bit RPN keys: four passes each), tiles of 2 048 pairs, one workgroup of four
// waves per tile (94-107 workgroups for the detector's sorts), two launches per pass:
//   rs_pass_count     a workgroup counts the digits of its tile in LDS                              -> hist[tile][digit]
//   rs_pass_scatter   thread d sums column d of hist (pairs of digit d in earlier tiles; in all tiles), wave 0 scans the 2^RB totals:
//                     the tile's first output slot per digit.  Then each WAVE walks its 512 consecutive pairs in 8 rounds of 64: a
//                     lane's rank among the lanes with the same digit comes from RB ballots, the wave's running count per digit
//                     lives in a wave-private LDS row (no workgroup barrier between rounds); the four rows are scanned once, and the
//                     pairs go out.  Waves, rounds and lanes are all in input order, so equal digits keep their order.
// (A one-launch form - the same two phases around grid-wide barriers, pairs exchanged with write-through stores - was built and
// measured at 92 us against 91 us for this one at the RPN's size: no gain to pay for its need of co-resident workgroups)
// Sorts of up to 8 192 4-byte-key pairs (the ~4 700 RPN survivors) run in ONE workgroup of 16 waves with both copies in LDS.
// SEGMENTS: every launch sorts gridDim.y (rs_sort_small: gridDim.x) independent arrays of the same length n - the images of a detector
// pass - whose buffers (and histograms) lie `seg` BYTES apart: one launch per stage for the whole pass instead of one per image.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RS_TILE 2048
#define RS_SMALL_MAX 8192

// segment `i` of a buffer whose segments lie `bytes` apart
template <typename T>
__device__ __forceinline__ T *rs_seg(T *p, size_t bytes, int i) { return (T *)((char *)p + (size_t)i * bytes); }

// lanes of this wave whose digit equals mine (me included)
template <int RB>
__device__ __forceinline__ unsigned long long rs_peers(unsigned d, bool valid)
{
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const unsigned long long s = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? s : ~s;
    }
    return m;
}

// exclusive prefix of a[0 .. NB) in place, by the calling WAVE alone (NB / 64 consecutive entries per lane)
template <int NB>
__device__ __forceinline__ void rs_wave_scan(unsigned *a)
{
    constexpr int PER = NB / 64;
    const int lane = threadIdx.x & 63;
    unsigned loc[PER], s = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { loc[j] = a[lane * PER + j]; s += loc[j]; }
    unsigned inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    unsigned run = inc - s;
#pragma unroll
    for (int j = 0; j < PER; ++j) { a[lane * PER + j] = run; run += loc[j]; }
}

// The pairs of one span in registers: wave w owns the rounds * 64 consecutive pairs from start + w * rounds * 64 (those below n), lane
// l of round r the pair wstart + r * 64 + l.  Every load is issued - the ones past the end on the last pair - so none waits on a branch.
template <typename K>
__device__ __forceinline__ void rs_load_span(const K *kin, const unsigned *vin, long start, long n, int rounds, K (&k)[8], unsigned (&v)[8])
{
    const long wstart = start + (long)(threadIdx.x >> 6) * rounds * 64 + (threadIdx.x & 63);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        long i = wstart + r * 64;
        i = i < n ? i : n - 1;
        k[r] = kin[i];
        v[r] = vin[i];
    }
}

// The span's pairs (k, v: rs_load_span) go to their slots of (kout, vout).  base[d]: the first slot of this span's pairs with digit d
// (LDS, filled by the caller, who has NOT yet synchronised after writing it); wcount: LDS [NW][NB].  Each WAVE walks its pairs in
// rounds of 64: a lane's rank among the lanes with the same digit comes from RB ballots, the wave's running count per digit lives in
// its own LDS row (no workgroup barrier between rounds); the NW rows are scanned once.  Ends with a workgroup barrier.
template <typename K, int RB, int NW>
__device__ __forceinline__ void rs_span_scatter(const K (&k)[8], const unsigned (&v)[8], K *kout, unsigned *vout, long start, long n,
                                                int rounds, int shift, unsigned mask, const unsigned *base, unsigned *wcount)
{
    constexpr int NB = 1 << RB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int q = tid; q < NW * NB; q += NW * 64) wcount[q] = 0u;
    unsigned pos[8], dg[8];
    const long wstart = start + (long)w * rounds * 64 + lane;
    __syncthreads();
    unsigned *mine = wcount + w * NB;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const bool valid = r < rounds && wstart + r * 64 < n;
        const unsigned d = (unsigned)(k[r] >> shift) & mask;
        const unsigned long long peers = rs_peers<RB>(d, valid);
        const unsigned rank = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
        const unsigned off = mine[d];         // (LDS serves a wave's reads and writes in program order)
        if (valid && rank == 0) mine[d] = off + (unsigned)__popcll(peers);
        pos[r] = off + rank;
        dg[r] = d;
    }
    __syncthreads();
    for (int d = tid; d < NB; d += NW * 64) {
        unsigned run = base[d];
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const unsigned c = wcount[q * NB + d];
            wcount[q * NB + d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (r < rounds && wstart + r * 64 < n) {
            const unsigned p = mine[dg[r]] + pos[r];
            kout[p] = k[r];
            vout[p] = v[r];
        }
    }
    __syncthreads();
}

// digits of the span's valid pairs counted into cnt[NB] (LDS, zeroed here); ends with a workgroup barrier
template <typename K, int NB>
__device__ __forceinline__ void rs_count_span(const K (&k)[8], long start, long n, int shift, unsigned mask, unsigned *cnt)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NB / 256; ++j) cnt[tid + j * 256] = 0u;
    __syncthreads();
    const long wstart = start + (long)(tid >> 6) * 512 + (tid & 63);
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (wstart + r * 64 < n) atomicAdd(&cnt[(unsigned)(k[r] >> shift) & mask], 1u);
    __syncthreads();
}

// phase 1 of a pass for one tile: load its pairs, count their digits -> hist[tile][.]
template <typename K, int RB>
__device__ __forceinline__ void rs_phase_count(const K *ka, const unsigned *va, long n, int tile, int shift, unsigned mask, unsigned *hist,
                                               unsigned *base, K (&k)[8], unsigned (&v)[8])
{
    constexpr int NB = 1 << RB, PER = NB / 256;
    const int tid = threadIdx.x;
    rs_load_span<K>(ka, va, (long)tile * RS_TILE, n, 8, k, v);
    rs_count_span<K, NB>(k, (long)tile * RS_TILE, n, shift, mask, base);
#pragma unroll
    for (int j = 0; j < PER; ++j) hist[(size_t)tile * NB + tid + j * 256] = base[tid + j * 256];
    __syncthreads();
}

// phase 2 for one tile (every tile's hist row is complete): its first output slot per digit from the column sums of hist, then the
// scatter of its pairs (k, v: as rs_load_span left them)
template <typename K, int RB>
__device__ __forceinline__ void rs_phase_scatter(const K (&k)[8], const unsigned (&v)[8], K *kb, unsigned *vb, long n, int tile, int ntiles,
                                                 int shift, unsigned mask, const unsigned *hist, unsigned *base, unsigned *wcount)
{
    constexpr int NB = 1 << RB, PER = NB / 256;
    const int tid = threadIdx.x;
    // column sums of hist for my digits (tid * PER + j): 16 rows in flight at a time
    unsigned below[PER], total[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) below[j] = total[j] = 0u;
    for (int t0 = 0; t0 < ntiles; t0 += 16) {
        unsigned h[16][PER];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int t = t0 + u < ntiles ? t0 + u : ntiles - 1;
            const unsigned *row = hist + (size_t)t * NB + tid * PER;
            if constexpr (PER == 2) {
                const unsigned long long q = *(const unsigned long long *)row;
                h[u][0] = (unsigned)q; h[u][1] = (unsigned)(q >> 32);
            } else {
                h[u][0] = *row;
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const unsigned c = t0 + u < ntiles ? h[u][j] : 0u;
                total[j] += c;
                below[j] += t0 + u < tile ? c : 0u;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) base[tid * PER + j] = total[j];
    __syncthreads();
    if (tid < 64) rs_wave_scan<NB>(base);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) base[tid * PER + j] += below[j];
    rs_span_scatter<K, RB, 4>(k, v, kb, vb, (long)tile * RS_TILE, n, 8, shift, mask, base, wcount);
}

__device__ __forceinline__ unsigned rs_pass_mask(int bits, int shift, int RB)
{
    return (bits - shift >= RB) ? ((1u << RB) - 1u) : ((1u << (bits - shift)) - 1u);
}

// a pass = these two launches of one workgroup per tile
template <typename K, int RB>
__global__ void __launch_bounds__(256) rs_pass_count(const K *ka, const unsigned *va, long n, int bits, int shift, unsigned *hist, size_t seg)
{
    __shared__ unsigned base[1 << RB];
    K k[8];
    unsigned v[8];
    ka = rs_seg(ka, seg, blockIdx.y); va = rs_seg(va, seg, blockIdx.y); hist = rs_seg(hist, seg, blockIdx.y);
    rs_phase_count<K, RB>(ka, va, n, (int)blockIdx.x, shift, rs_pass_mask(bits, shift, RB), hist, base, k, v);
}

template <typename K, int RB>
__global__ void __launch_bounds__(256) rs_pass_scatter(const K *ka, const unsigned *va, K *kb, unsigned *vb, long n, int bits, int shift,
                                                       const unsigned *hist, size_t seg)
{
    constexpr int NB = 1 << RB;
    __shared__ unsigned base[NB];
    __shared__ unsigned wcount[4 * NB];
    K k[8];
    unsigned v[8];
    ka = rs_seg(ka, seg, blockIdx.y); va = rs_seg(va, seg, blockIdx.y); hist = rs_seg(hist, seg, blockIdx.y);
    kb = rs_seg(kb, seg, blockIdx.y); vb = rs_seg(vb, seg, blockIdx.y);
    rs_load_span<K>(ka, va, (long)blockIdx.x * RS_TILE, n, 8, k, v);
    rs_phase_scatter<K, RB>(k, v, kb, vb, n, (int)blockIdx.x, (int)gridDim.x, shift, rs_pass_mask(bits, shift, RB), hist, base, wcount);
}

// n <= RS_SMALL_MAX pairs of 4-byte keys, every pass in one workgroup of 16 waves, both copies in LDS; result to kout / vout
__global__ void __launch_bounds__(1024) rs_sort_small(const unsigned *__restrict__ kin, const unsigned *__restrict__ vin,
                                                      unsigned *__restrict__ kout, unsigned *__restrict__ vout, int n, int bits, size_t seg)
{
    kin = rs_seg(kin, seg, blockIdx.x); vin = rs_seg(vin, seg, blockIdx.x);
    kout = rs_seg(kout, seg, blockIdx.x); vout = rs_seg(vout, seg, blockIdx.x);
    extern __shared__ unsigned rs_lds[];
    unsigned *ka = rs_lds, *kb = ka + RS_SMALL_MAX, *va = kb + RS_SMALL_MAX, *vb = va + RS_SMALL_MAX;
    __shared__ unsigned base[256];
    __shared__ unsigned wcount[16 * 256];
    const int tid = threadIdx.x;
    const int rounds = (n + 1023) / 1024;
    for (int i = tid; i < n; i += 1024) { ka[i] = kin[i]; va[i] = vin[i]; }
    for (int shift = 0; shift < bits; shift += 8) {
        const unsigned mask = (bits - shift >= 8) ? 255u : ((1u << (bits - shift)) - 1u);
        if (tid < 256) base[tid] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) atomicAdd(&base[(ka[i] >> shift) & mask], 1u);
        __syncthreads();
        if (tid < 64) rs_wave_scan<256>(base);
        unsigned k[8], v[8];
        rs_load_span<unsigned>(ka, va, 0, n, rounds, k, v);
        rs_span_scatter<unsigned, 8, 16>(k, v, kb, vb, 0, n, rounds, shift, mask, base, wcount);
        unsigned *t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    for (int i = tid; i < n; i += 1024) { kout[i] = ka[i]; vout[i] = va[i]; }
}
