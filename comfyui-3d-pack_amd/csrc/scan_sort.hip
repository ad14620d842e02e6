// scan_sort.hip -- device-wide prefix sum and stable LSD radix sort for gfx950.
//
// These replace the cub::DeviceScan / cub::DeviceRadixSort calls of the dependency's binning
// stage.  Written for wave64: digit ranking uses 64-bit ballots
// (one match mask per lane from 8 ballots) and mbcnt prefix counts instead of 32-lane warp votes.
#include "c3d_common.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

// exclusive prefix of `v` across the 256-thread block; *total = block sum.  lds: >= 4 uints.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds, uint32_t* total) {
    const int lane = c3d_lane(), wave = threadIdx.x >> 6;
    uint32_t incl = c3d_wave_incl_scan(v);
    __syncthreads();  // protect lds reuse across calls
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        uint32_t t = lds[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + incl - v;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_block_sums(const uint32_t* __restrict__ in, uint32_t* __restrict__ sums, size_t n) {
    __shared__ uint32_t lds[4];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) s += in[base + i];
    uint32_t tot;
    block_excl_scan(s, lds, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single block: exclusive scan of m values in place
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_small_excl(uint32_t* __restrict__ a, size_t m) {
    __shared__ uint32_t lds[4];
    uint32_t carry = 0;
    for (size_t c = 0; c < m; c += SCAN_THREADS * 4) {
        size_t base = c + (size_t)threadIdx.x * 4;
        uint32_t v[4];
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { v[i] = (base + i < m) ? a[base + i] : 0u; s += v[i]; }
        uint32_t tot;
        uint32_t ex = block_excl_scan(s, lds, &tot) + carry;
#pragma unroll
        for (int i = 0; i < 4; i++) { if (base + i < m) a[base + i] = ex; ex += v[i]; }
        carry += tot;
    }
}

template <bool EXCL>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_apply(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                              const uint32_t* __restrict__ sums, size_t n) {
    __shared__ uint32_t lds[4];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = (base + i < n) ? in[base + i] : 0u; s += v[i]; }
    uint32_t tot;
    uint32_t run = block_excl_scan(s, lds, &tot) + sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (EXCL) { if (base + i < n) out[base + i] = run; run += v[i]; }
        else      { run += v[i]; if (base + i < n) out[base + i] = run; }
    }
}

size_t c3d_scan_tmp_bytes(size_t n) { return c3d_align(sizeof(uint32_t) * (size_t)(c3d_cdiv((long long)n, SCAN_TILE) + 1)); }

int c3d_scan_u32(const uint32_t* in, uint32_t* out, size_t n, bool exclusive, void* tmp, hipStream_t s) {
    if (n == 0) return 0;
    int nb = c3d_cdiv((long long)n, SCAN_TILE);
    uint32_t* sums = (uint32_t*)tmp;
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(SCAN_THREADS), 0, s, in, sums, n);
    hipLaunchKernelGGL(k_scan_small_excl, dim3(1), dim3(SCAN_THREADS), 0, s, sums, (size_t)nb);
    if (exclusive) hipLaunchKernelGGL(k_scan_apply<true>, dim3(nb), dim3(SCAN_THREADS), 0, s, in, out, sums, n);
    else           hipLaunchKernelGGL(k_scan_apply<false>, dim3(nb), dim3(SCAN_THREADS), 0, s, in, out, sums, n);
    C3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Radix sort pass: 8-bit digits, 4096 keys per 256-thread block, each wave owns a contiguous
// 1024-key chunk so that ranking is stable by construction.  Three kernels per pass:
//   k_radix_hist     per-block digit histogram -> table[digit][block], digit totals by integer atomics
//   k_radix_rowscan  one wave per digit: exclusive scan of its table row on top of the digit's base
//   k_radix_scatter  ballot ranking, block-local reorder in LDS, then run-contiguous global stores
// ------------------------------------------------------------------------------------------
#define RS_THREADS 256
#define RS_ITEMS 16
#define RS_TILE (RS_THREADS * RS_ITEMS)
#define RS_RADIX 256

__global__ void __launch_bounds__(RS_THREADS) k_radix_hist(const uint32_t* __restrict__ keys, uint32_t* __restrict__ table,
                                                            uint32_t* __restrict__ total, size_t n, const uint32_t* __restrict__ n_dev, int shift, int nblocks) {
    __shared__ uint32_t h[RS_RADIX];
    if (n_dev) n = min((size_t)*n_dev, n);      // element count resident on the device (no host round trip)
    size_t base = (size_t)blockIdx.x * RS_TILE;
    if (base >= n) { table[(size_t)threadIdx.x * nblocks + blockIdx.x] = 0; return; }   // capacity-sized launch: nothing here
    h[threadIdx.x] = 0;
    __syncthreads();
    const int lane = c3d_lane();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        size_t idx = base + (size_t)i * RS_THREADS + threadIdx.x;
        const bool ok = idx < n;
        const uint32_t d = ok ? ((keys[idx] >> shift) & (RS_RADIX - 1)) : 0u;
        // high digits of depth keys / tile ids are nearly constant: a wave whose lanes all hold one digit adds its count once
        // instead of serialising 64 LDS atomics on one counter
        const uint64_t okm = __ballot(ok);
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
        if (okm && __ballot(ok && d == d0) == okm) {
            if (lane == (int)__builtin_ctzll(okm)) atomicAdd(&h[d0], (uint32_t)__popcll(okm));
        } else if (ok) atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    const uint32_t c = h[threadIdx.x];
    table[(size_t)threadIdx.x * nblocks + blockIdx.x] = c;  // digit-major
    if (c) atomicAdd(&total[threadIdx.x], c);
}

__global__ void __launch_bounds__(RS_THREADS) k_radix_rowscan(uint32_t* __restrict__ table, const uint32_t* __restrict__ total, int nblocks) {
    const int lane = c3d_lane();
    const int d = blockIdx.x * (RS_THREADS / 64) + (threadIdx.x >> 6);   // one wave per digit
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < RS_RADIX / 64; i++) { const int dd = i * 64 + lane; if (dd < d) s += total[dd]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    uint32_t carry = s;
    uint32_t* row = table + (size_t)d * nblocks;
    for (int b0 = 0; b0 < nblocks; b0 += 64) {
        const int b = b0 + lane;
        const uint32_t v = (b < nblocks) ? row[b] : 0u;
        const uint32_t incl = c3d_wave_incl_scan(v);
        if (b < nblocks) row[b] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
}

template <bool IOTA>
__global__ void __launch_bounds__(RS_THREADS) k_radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                               uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                               const uint32_t* __restrict__ table, size_t n, const uint32_t* __restrict__ n_dev,
                                                               int shift, int nblocks) {
    __shared__ uint32_t whist[RS_THREADS / 64][RS_RADIX];
    if (n_dev) n = min((size_t)*n_dev, n);
    __shared__ uint32_t lstart[RS_RADIX];    // first local slot of each digit
    __shared__ uint32_t gbase[RS_RADIX];     // global position of that slot
    __shared__ uint32_t skey[RS_TILE];
    __shared__ uint32_t sval[RS_TILE];
    __shared__ uint32_t scan_lds[4];
    const int lane = c3d_lane(), wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (RS_THREADS / 64) * RS_RADIX; i += RS_THREADS) (&whist[0][0])[i] = 0;
    __syncthreads();

    const size_t bbase = (size_t)blockIdx.x * RS_TILE;
    const size_t wbase = bbase + (size_t)wave * (RS_TILE / 4);
    uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        size_t idx = wbase + (size_t)i * 64 + lane;
        bool ok = idx < n;
        key[i] = ok ? keys_in[idx] : 0xFFFFFFFFu;
        val[i] = IOTA ? (uint32_t)idx : (ok ? vals_in[idx] : 0u);
    }
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        size_t idx = wbase + (size_t)i * 64 + lane;
        bool ok = idx < n;
        uint32_t d = (key[i] >> shift) & (RS_RADIX - 1);
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        uint32_t prefix = whist[wave][d];
        uint32_t r = (uint32_t)__popcll(peers & lt_mask);
        if (ok && r == 0) whist[wave][d] = prefix + (uint32_t)__popcll(peers);  // lowest peer lane updates
        rank[i] = prefix + r;
    }
    __syncthreads();
    {   // thread d: digit count over the 4 waves -> block-local exclusive start; per-wave offsets
        const int d = threadIdx.x;
        uint32_t c[RS_THREADS / 64], tot = 0;
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; w++) { c[w] = whist[w][d]; tot += c[w]; }
        uint32_t blk_total;
        uint32_t ls = block_excl_scan(tot, scan_lds, &blk_total);
        lstart[d] = ls;
        gbase[d] = table[(size_t)d * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; w++) { whist[w][d] = ls; ls += c[w]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        size_t idx = wbase + (size_t)i * 64 + lane;
        if (idx < n) {
            uint32_t d = (key[i] >> shift) & (RS_RADIX - 1);
            uint32_t lp = whist[wave][d] + rank[i];
            skey[lp] = key[i];
            sval[lp] = val[i];
        }
    }
    __syncthreads();
    const int cnt = (bbase >= n) ? 0 : (int)((n - bbase) < (size_t)RS_TILE ? (n - bbase) : (size_t)RS_TILE);
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const int lp = i * RS_THREADS + threadIdx.x;     // consecutive lanes -> consecutive slots of a digit run
        if (lp < cnt) {
            const uint32_t k = skey[lp];
            const uint32_t d = (k >> shift) & (RS_RADIX - 1);
            const uint32_t pos = gbase[d] + ((uint32_t)lp - lstart[d]);
            keys_out[pos] = k;
            vals_out[pos] = sval[lp];
        }
    }
}

#define RS_MAX_PASSES 4
size_t c3d_sort_tmp_bytes(size_t n) {
    size_t nb = (size_t)c3d_cdiv((long long)(n ? n : 1), RS_TILE);
    return c3d_align(sizeof(uint32_t) * RS_RADIX * nb) + c3d_align(sizeof(uint32_t) * RS_RADIX * RS_MAX_PASSES);
}

int c3d_sort_pairs_u32(uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1, bool iota_vals,
                       size_t n, int end_bit, void* tmp, int* result_buf, hipStream_t s, const uint32_t* n_dev) {
    *result_buf = 0;
    if (n == 0) return 0;
    if (end_bit > 8 * RS_MAX_PASSES) { c3d_set_error("c3d_sort_pairs_u32: end_bit %d > %d", end_bit, 8 * RS_MAX_PASSES); return -1; }
    int nb = c3d_cdiv((long long)n, RS_TILE);
    uint32_t* table = (uint32_t*)tmp;
    uint32_t* totals = (uint32_t*)((char*)tmp + c3d_align(sizeof(uint32_t) * RS_RADIX * (size_t)nb));
    C3D_CHECK(hipMemsetAsync(totals, 0, sizeof(uint32_t) * RS_RADIX * RS_MAX_PASSES, s));
    uint32_t* k[2] = {keys0, keys1};
    uint32_t* v[2] = {vals0, vals1};
    int cur = 0, pass = 0;
    bool first = true;
    for (int shift = 0; shift < end_bit || first; shift += 8, pass++) {
        uint32_t* tot = totals + pass * RS_RADIX;
        hipLaunchKernelGGL(k_radix_hist, dim3(nb), dim3(RS_THREADS), 0, s, k[cur], table, tot, n, n_dev, shift, nb);
        hipLaunchKernelGGL(k_radix_rowscan, dim3(RS_RADIX / (RS_THREADS / 64)), dim3(RS_THREADS), 0, s, table, tot, nb);
        if (first && iota_vals)
            hipLaunchKernelGGL(k_radix_scatter<true>, dim3(nb), dim3(RS_THREADS), 0, s, k[cur], v[cur], k[cur ^ 1], v[cur ^ 1], table, n, n_dev, shift, nb);
        else
            hipLaunchKernelGGL(k_radix_scatter<false>, dim3(nb), dim3(RS_THREADS), 0, s, k[cur], v[cur], k[cur ^ 1], v[cur ^ 1], table, n, n_dev, shift, nb);
        C3D_LAUNCH_CHECK();
        cur ^= 1;
        first = false;
    }
    *result_buf = cur;
    return 0;
}
