// det_sort_kernels.hip - stable least-significant-digit radix sort of (key, value) pairs for the detector's selection stages
// (RegionProposalNetwork.filter_proposals: per-level top-k by objectness; RoIHeads.postprocess_detections: candidates by
// score - torchvision 0.5.0, called from baselines/detector.py:84).  It replaces the hipCUB calls of rounds 1-3: no
// vendor library is left on any path of libopnet_hip.so.  A stable sort has ONE result, so proposals and detections are what they
// were (tests/test_detector_gpu.py; tests/test_detector_sort_gpu.py holds the sort itself against torch.sort(stable=True)).
//
// ONE launch per sort.  Digits of RB bits (8 for 4-byte keys, 9 for the 35-bit RPN keys: four passes each), tiles of 2 048 pairs, one
// workgroup of four waves per tile (the 94-107 tiles of the detector's sorts are co-resident on the 256 CUs, which the two grid-wide
// barriers of a pass rely on: the host never launches more workgroups than RS_MAX_GRID):
//   phase 1   a workgroup counts the digits of its tile in LDS                                   -> hist[tile][digit]
//   -- grid barrier --
//   phase 2   thread d sums column d of hist (pairs of digit d in earlier tiles; in all tiles), wave 0 scans the 2^RB totals: the
//             tile's first output slot per digit.  Then each WAVE walks its 512 consecutive pairs in 8 rounds of 64: a lane's rank
//             among the lanes with the same digit comes from RB ballots, the wave's running count per digit lives in a wave-private
//             LDS row (no workgroup barrier between rounds); the four rows are scanned once, and the pairs go out.  Waves, rounds and
//             lanes are all in input order, so equal digits keep their order.
//   -- grid barrier --  (the next pass reads what this one scattered)
// Sorts of up to 8 192 4-byte-key pairs (the ~4 700 RPN survivors) run in ONE workgroup of 16 waves with both copies in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RS_TILE 2048
#define RS_SMALL_MAX 8192
#define RS_MAX_GRID 512           // 256-thread workgroups with <= 11 KB of LDS: five per CU are resident, 1 280 on the chip

// Pairs and histograms cross workgroups (any XCD) inside ONE launch: stores write through (sc1), loads are served past this CU's L1
// and this XCD's L2 state (sc1) - agent-scope relaxed atomics are exactly those instructions.  No cache-wide writeback or
// invalidate is then needed at the grid barrier (a __threadfence() there cost ~10 us per barrier: every workgroup walks its L2).
template <typename T> __device__ __forceinline__ T rs_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void rs_st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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
template <typename K, bool GLOBAL>
__device__ __forceinline__ void rs_load_span(const K *kin, const unsigned *vin, long start, long n, int rounds, K (&k)[8], unsigned (&v)[8])
{
    const long wstart = start + (long)(threadIdx.x >> 6) * rounds * 64 + (threadIdx.x & 63);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        long i = wstart + r * 64;
        i = i < n ? i : n - 1;
        if constexpr (GLOBAL) { k[r] = rs_ld(kin + i); v[r] = rs_ld(vin + i); }
        else { k[r] = kin[i]; v[r] = vin[i]; }
    }
}

// The span's pairs (k, v: rs_load_span) go to their slots of (kout, vout).  base[d]: the first slot of this span's pairs with digit d
// (LDS, filled by the caller, who has NOT yet synchronised after writing it); wcount: LDS [NW][NB].  Each WAVE walks its pairs in
// rounds of 64: a lane's rank among the lanes with the same digit comes from RB ballots, the wave's running count per digit lives in
// its own LDS row (no workgroup barrier between rounds); the NW rows are scanned once.  Ends with a workgroup barrier.
template <typename K, int RB, int NW, bool GLOBAL>
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
            if constexpr (GLOBAL) { rs_st(kout + p, k[r]); rs_st(vout + p, v[r]); }
            else { kout[p] = k[r]; vout[p] = v[r]; }
        }
    }
    __syncthreads();
}

// every workgroup of the launch has arrived `target / gridDim.x` times (bar: a zeroed word, counting arrivals).  The workgroup's
// write-through stores are complete (vmcnt 0) before it counts itself in.
__device__ __forceinline__ void rs_grid_barrier(unsigned *bar, unsigned target)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    asm volatile("" ::: "memory");
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

// grid <= RS_MAX_GRID workgroups of 256, all co-resident; hist: [passes][tiles][2^RB] words (a fresh block per pass: its rows are
// written through before the pass's first barrier and read - plainly, L2-served - only after it, and nothing in this launch touched
// them earlier, so no stale line can exist); bar: one zeroed word.  After ceil(bits / RB) passes the pairs are in (ka, va) when that
// number is even, in (kb, vb) when it is odd.
template <typename K, int RB>
__global__ void __launch_bounds__(256) rs_sort_tiled(K *ka, unsigned *va, K *kb, unsigned *vb, long n, int bits, unsigned *hist,
                                                     unsigned *bar)
{
    constexpr int NB = 1 << RB, PER = NB / 256;
    __shared__ unsigned base[NB];
    __shared__ unsigned wcount[4 * NB];
    const int tid = threadIdx.x;
    const int ntiles = (int)((n + RS_TILE - 1) / RS_TILE), nwg = gridDim.x;
    const bool one_tile = ntiles <= nwg;      // (the detector's sorts: the pairs stay in registers across the pass's first barrier)
    unsigned epoch = 0;
    for (int shift = 0; shift < bits; shift += RB, hist += (size_t)ntiles * NB) {
        const unsigned mask = (bits - shift >= RB) ? (unsigned)(NB - 1) : ((1u << (bits - shift)) - 1u);
        K k[8];
        unsigned v[8];
        for (int tile = blockIdx.x; tile < ntiles; tile += nwg) {
            rs_load_span<K, true>(ka, va, (long)tile * RS_TILE, n, 8, k, v);
            rs_count_span<K, NB>(k, (long)tile * RS_TILE, n, shift, mask, base);
#pragma unroll
            for (int j = 0; j < PER; ++j) rs_st(hist + (size_t)tile * NB + tid + j * 256, base[tid + j * 256]);
            __syncthreads();
        }
        rs_grid_barrier(bar, (unsigned)nwg * ++epoch);
        for (int tile = blockIdx.x; tile < ntiles; tile += nwg) {
            if (!one_tile) rs_load_span<K, true>(ka, va, (long)tile * RS_TILE, n, 8, k, v);
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
            rs_span_scatter<K, RB, 4, true>(k, v, kb, vb, (long)tile * RS_TILE, n, 8, shift, mask, base, wcount);
        }
        if (shift + RB < bits) rs_grid_barrier(bar, (unsigned)nwg * ++epoch);      // the next pass reads what this one scattered
        K *tk = ka; ka = kb; kb = tk;
        unsigned *tv = va; va = vb; vb = tv;
    }
}

// n <= RS_SMALL_MAX pairs of 4-byte keys, every pass in one workgroup of 16 waves, both copies in LDS; result to kout / vout
__global__ void __launch_bounds__(1024) rs_sort_small(const unsigned *__restrict__ kin, const unsigned *__restrict__ vin,
                                                      unsigned *__restrict__ kout, unsigned *__restrict__ vout, int n, int bits)
{
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
        rs_load_span<unsigned, false>(ka, va, 0, n, rounds, k, v);
        rs_span_scatter<unsigned, 8, 16, false>(k, v, kb, vb, 0, n, rounds, shift, mask, base, wcount);
        unsigned *t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    for (int i = tid; i < n; i += 1024) { kout[i] = ka[i]; vout[i] = va[i]; }
}
