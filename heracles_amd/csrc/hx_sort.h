// hx_sort.h -- stable LSD radix sort of (key, 32-bit value) pairs on the GPU, 8-bit digits.  Round 5: written for the two "next" rows that
// sort -- catalogue points by pixel (hx_mapper.hip, heracles/healpy.py:144-160) and by LDS tile (hx_nufft.hip) --, which called a library
// primitive until round 4.
//
// Per pass:  k_sort_hist     per tile of 4096 keys a 256-bin histogram (LDS atomics), written bin-major: counts[digit][tile];
//            k_scan_*        exclusive scan over the 256 x tiles counts (block sums, one block over the sums, apply);
//            k_sort_scatter  the tile again: every wave ranks ITS contiguous 1024 keys in 16 rounds of 64 -- lanes with equal digits find
//                            each other with 8 ballots, the lowest lane of a group advances the wave's own LDS counter of that digit (LDS
//                            operations of one wave execute in order: no atomics, no barrier inside the ranking loop) --, the four waves'
//                            counters give every key its place in the tile sorted by digit, the pairs go through LDS into that order
//                            and leave in runs of equal digits: consecutive threads write consecutive addresses.
// Stable (equal keys keep their input order: wave segments, rounds and lanes are all in index order), deterministic.
#pragma once
#include <hip/hip_runtime.h>

#include "hx_common.h"

namespace hx {
namespace rsort {

constexpr int NT = 256, IPT = 16, TILE = NT * IPT, NW = NT / 64, SEG = TILE / NW;  // 4096 keys per tile, 1024 per wave
constexpr int SCAN_T = 256, SCAN_IPT = 8, SCAN_TILE = SCAN_T * SCAN_IPT;

template <class K>
__device__ __forceinline__ unsigned digit_of(K key, int shift, unsigned mask)
{
    return (unsigned)((unsigned long long)key >> shift) & mask;
}

template <class K>
__global__ __launch_bounds__(NT) void k_sort_hist(const K *__restrict__ keys, unsigned long long n, int shift, unsigned mask, unsigned ntiles,
                                                  unsigned *__restrict__ counts)
{
    __shared__ unsigned hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long base = (unsigned long long)blockIdx.x * TILE;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const unsigned long long idx = base + (unsigned long long)i * NT + threadIdx.x;
        if (idx < n) atomicAdd(&hist[digit_of(keys[idx], shift, mask)], 1u);
    }
    __syncthreads();
    counts[(unsigned long long)threadIdx.x * ntiles + blockIdx.x] = hist[threadIdx.x];
}

// ---- exclusive scan of `e` unsigned values in place: block sums, their scan by one block, apply ----
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned *lds /* [NT / 64 + 1] */, unsigned &total)
{
    // inclusive scan inside the wave, then across the waves of the block
    unsigned x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned y = __shfl_up(x, o, 64);
        if ((int)(threadIdx.x & 63) >= o) x += y;
    }
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 63) lds[w] = x;
    __syncthreads();
    unsigned before = 0, all = 0;
    for (int k = 0; k < nw; ++k) {
        const unsigned s = lds[k];
        if (k < w) before += s;
        all += s;
    }
    __syncthreads();
    total = all;
    return before + x - v;
}

static __global__ __launch_bounds__(SCAN_T) void k_scan_sums(const unsigned *__restrict__ a, unsigned long long e, unsigned *__restrict__ sums)
{
    __shared__ unsigned lds[8];
    const unsigned long long base = (unsigned long long)blockIdx.x * SCAN_TILE + (unsigned long long)threadIdx.x * SCAN_IPT;
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; ++i)
        if (base + i < e) s += a[base + i];
    unsigned total;
    (void)block_exclusive_scan(s, lds, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

static __global__ __launch_bounds__(1024) void k_scan_top(unsigned *__restrict__ sums, unsigned nb)
{
    __shared__ unsigned lds[20];
    // every thread takes a contiguous share of the block sums
    const unsigned per = (nb + blockDim.x - 1) / blockDim.x;
    const unsigned lo = min(nb, threadIdx.x * per), hi = min(nb, lo + per);
    unsigned s = 0;
    for (unsigned i = lo; i < hi; ++i) s += sums[i];
    unsigned total;
    unsigned run = block_exclusive_scan(s, lds, total);
    for (unsigned i = lo; i < hi; ++i) {
        const unsigned v = sums[i];
        sums[i] = run;
        run += v;
    }
}

static __global__ __launch_bounds__(SCAN_T) void k_scan_apply(unsigned *__restrict__ a, unsigned long long e, const unsigned *__restrict__ sums)
{
    __shared__ unsigned lds[8];
    const unsigned long long base = (unsigned long long)blockIdx.x * SCAN_TILE + (unsigned long long)threadIdx.x * SCAN_IPT;
    unsigned v[SCAN_IPT], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; ++i) {
        v[i] = base + i < e ? a[base + i] : 0u;
        s += v[i];
    }
    unsigned total;
    unsigned run = sums[blockIdx.x] + block_exclusive_scan(s, lds, total);
#pragma unroll
    for (int i = 0; i < SCAN_IPT; ++i) {
        if (base + i < e) a[base + i] = run;
        run += v[i];
    }
}

// KO: the key type written (a pass may narrow 64-bit keys whose significant bits fit 32: less to move in every later pass)
template <class K, class KO = K>
__global__ __launch_bounds__(NT) void k_sort_scatter(const K *__restrict__ kin, const unsigned *__restrict__ vin, KO *__restrict__ kout,
                                                     unsigned *__restrict__ vout, unsigned long long n, int shift, unsigned mask, unsigned ntiles,
                                                     const unsigned *__restrict__ offsets)
{
    __shared__ unsigned whist[NW][256];  // per wave: keys of its segment seen so far, by digit; then their first place in the sorted tile
    __shared__ unsigned gdelta[256];     // global index of a key = its place in the sorted tile + gdelta[digit]
    __shared__ unsigned scan_lds[8];
    __shared__ KO skey[TILE];
    __shared__ unsigned sval[TILE];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
#pragma unroll
    for (int k = 0; k < NW; ++k) whist[k][t] = 0;
    __syncthreads();
    const unsigned long long base = (unsigned long long)blockIdx.x * TILE + (unsigned long long)w * SEG + lane;
    K key[IPT];
    unsigned val[IPT], place[IPT];
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const unsigned long long idx = base + (unsigned long long)r * 64;
        const bool valid = idx < n;
        key[r] = valid ? kin[idx] : (K)0;
        val[r] = valid ? vin[idx] : 0u;
        const unsigned d = digit_of(key[r], shift, mask);
        // the lanes of this wave that hold the same digit (8 ballots), invalid lanes apart
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const unsigned rk = (unsigned)__popcll(peers & lt), cnt = (unsigned)__popcll(peers);
        unsigned old = 0;
        if (valid && rk == 0) {  // the lowest lane of the group advances the wave's counter of the digit
            old = whist[w][d];
            whist[w][d] = old + cnt;
        }
        const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
        old = __shfl(old, leader, 64);
        place[r] = old + rk;  // place among the keys of this digit in this wave's segment
    }
    __syncthreads();
    {   // thread t = digit t: where the digit starts in the sorted tile, where each wave's keys of it start, and its global offset
        unsigned c[NW], tot = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { c[k] = whist[k][t]; tot += c[k]; }
        unsigned total;
        unsigned start = block_exclusive_scan(tot, scan_lds, total);
        gdelta[t] = offsets[(unsigned long long)t * ntiles + blockIdx.x] - start;
#pragma unroll
        for (int k = 0; k < NW; ++k) { whist[k][t] = start; start += c[k]; }
    }
    __syncthreads();
    const unsigned long long tile0 = (unsigned long long)blockIdx.x * TILE;
    const unsigned count = (unsigned)(n - tile0 < (unsigned long long)TILE ? n - tile0 : (unsigned long long)TILE);
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const unsigned long long idx = base + (unsigned long long)r * 64;
        if (idx < n) {
            const unsigned p = whist[w][digit_of(key[r], shift, mask)] + place[r];
            skey[p] = (KO)key[r];
            sval[p] = val[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const unsigned p = (unsigned)j * NT + t;
        if (p < count) {
            const KO k = skey[p];
            const unsigned o = gdelta[digit_of(k, shift, mask)] + p;  // (modulo 2^32: gdelta may have wrapped, the index has not)
            kout[o] = k;
            vout[o] = sval[p];
        }
    }
}

// Sorts n <= 2^32 - 16 pairs by the bits [0, end_bit) of their keys.  (k0, v0) hold the input and are overwritten; (k1, v1) are scratch of
// the same size; tmp grows to 4 x 256 x tiles + block sums.  On return *ks / *vs point at whichever pair of buffers holds the sorted pairs.
template <class K>
inline int radix_sort_pairs(K *k0, unsigned *v0, K *k1, unsigned *v1, unsigned long long n, int end_bit, DevBuf &tmp, hipStream_t st, K **ks,
                            unsigned **vs)
{
    *ks = k0;
    *vs = v0;
    if (n == 0 || end_bit <= 0) return HX_OK;
    if (n > 0xfffffff0ull) return fail(HX_ERR_UNSUPPORTED, "radix_sort_pairs: %llu pairs (at most 2^32 - 16)", n);
    const unsigned ntiles = (unsigned)((n + TILE - 1) / TILE);
    const unsigned long long e = 256ull * ntiles;
    const unsigned nsb = (unsigned)((e + SCAN_TILE - 1) / SCAN_TILE);
    HX_TRY(tmp.alloc(sizeof(unsigned) * (size_t)(e + nsb + 16)));
    unsigned *counts = tmp.as<unsigned>(), *sums = counts + e;
    for (int shift = 0; shift < end_bit; shift += 8) {
        const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        const unsigned mask = (1u << bits) - 1u;
        hipLaunchKernelGGL(k_sort_hist<K>, dim3(ntiles), dim3(NT), 0, st, *ks, n, shift, mask, ntiles, counts);
        hipLaunchKernelGGL(k_scan_sums, dim3(nsb), dim3(SCAN_T), 0, st, counts, e, sums);
        hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, sums, nsb);
        hipLaunchKernelGGL(k_scan_apply, dim3(nsb), dim3(SCAN_T), 0, st, counts, e, sums);
        K *ko = *ks == k0 ? k1 : k0;
        unsigned *vo = *vs == v0 ? v1 : v0;
        hipLaunchKernelGGL(k_sort_scatter<K>, dim3(ntiles), dim3(NT), 0, st, *ks, *vs, ko, vo, n, shift, mask, ntiles, counts);
        *ks = ko;
        *vs = vo;
    }
    HX_HIP(hipGetLastError());
    return HX_OK;
}

// The same for 64-bit keys whose bits [0, end_bit), end_bit <= 32, are all that distinguishes them (pixel indices): the first pass reads
// the 64-bit keys and writes 32-bit ones, the later passes move 8 instead of 12 bytes per pair.  k64 / v0: input (v0 is overwritten);
// ka, kb, v1: scratch of n unsigned each.  On return *ks / *vs point at the sorted 32-bit keys and their values.
inline int radix_sort_pairs_narrow(const long long *k64, unsigned *v0, unsigned *ka, unsigned *kb, unsigned *v1, unsigned long long n, int end_bit,
                                   DevBuf &tmp, hipStream_t st, unsigned **ks, unsigned **vs)
{
    *ks = ka;
    *vs = v0;
    if (n == 0) return HX_OK;
    if (end_bit <= 0 || end_bit > 32) return fail(HX_ERR_ARG, "radix_sort_pairs_narrow: %d significant bits", end_bit);
    if (n > 0xfffffff0ull) return fail(HX_ERR_UNSUPPORTED, "radix_sort_pairs: %llu pairs (at most 2^32 - 16)", n);
    const unsigned ntiles = (unsigned)((n + TILE - 1) / TILE);
    const unsigned long long e = 256ull * ntiles;
    const unsigned nsb = (unsigned)((e + SCAN_TILE - 1) / SCAN_TILE);
    HX_TRY(tmp.alloc(sizeof(unsigned) * (size_t)(e + nsb + 16)));
    unsigned *counts = tmp.as<unsigned>(), *sums = counts + e;
    unsigned *kcur = nullptr, *vcur = v0;
    for (int shift = 0; shift < end_bit; shift += 8) {
        const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        const unsigned mask = (1u << bits) - 1u;
        if (shift == 0) hipLaunchKernelGGL(k_sort_hist<long long>, dim3(ntiles), dim3(NT), 0, st, k64, n, shift, mask, ntiles, counts);
        else hipLaunchKernelGGL(k_sort_hist<unsigned>, dim3(ntiles), dim3(NT), 0, st, kcur, n, shift, mask, ntiles, counts);
        hipLaunchKernelGGL(k_scan_sums, dim3(nsb), dim3(SCAN_T), 0, st, counts, e, sums);
        hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, sums, nsb);
        hipLaunchKernelGGL(k_scan_apply, dim3(nsb), dim3(SCAN_T), 0, st, counts, e, sums);
        unsigned *ko = kcur == ka ? kb : ka, *vo = vcur == v0 ? v1 : v0;
        if (shift == 0)
            hipLaunchKernelGGL((k_sort_scatter<long long, unsigned>), dim3(ntiles), dim3(NT), 0, st, k64, vcur, ko, vo, n, shift, mask, ntiles, counts);
        else
            hipLaunchKernelGGL((k_sort_scatter<unsigned, unsigned>), dim3(ntiles), dim3(NT), 0, st, kcur, vcur, ko, vo, n, shift, mask, ntiles, counts);
        kcur = ko;
        vcur = vo;
    }
    *ks = kcur;
    *vs = vcur;
    HX_HIP(hipGetLastError());
    return HX_OK;
}

}  // namespace rsort
}  // namespace hx
