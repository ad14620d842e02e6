// scan_sort.hip -- device-wide prefix sum and stable LSD radix sort for gfx950, single pass over the data per launch.
//
// These replace the cub::DeviceScan / cub::DeviceRadixSort calls of the dependency's binning stage.
// Written for wave64: digit ranking uses 64-bit ballots (one match mask per lane from 8 ballots) and mbcnt prefix counts
// instead of 32-lane warp votes.
//
// Round 2: both primitives are "chained scan with decoupled look-back" kernels -- ONE launch per scan and ONE launch per radix
// digit (plus one histogram launch per sort) instead of three launches per scan / per digit.  The binning chain of a view is
// latency bound (10-30 us kernels that cannot fill 256 CUs), so launches and re-reads of the keys are what it pays for:
//   depth sort  12 launches -> 5     tile sort  6 -> 3     scans  8 -> 2
// Inter-workgroup protocol (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   * a workgroup takes its tile index from a ticket counter, so every tile it will ever wait for belongs to a workgroup that is
//     already running -- no assumption about dispatch order or residency;
//   * what a tile publishes is ONE self-contained word {flag, value} written and polled with relaxed agent-scope atomics (sc1:
//     write-through / L1-bypassing on gfx950), so no fence is needed and the 8 non-coherent XCD L2s cannot serve a stale flag;
//   * every spin is bounded: on a timeout the kernel raises an error word the host turns into an exception, it never hangs the GPU.
#include "c3d_common.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)
#define LB_SPIN_LIMIT (1u << 21)     // polls of one word before a workgroup gives up (~0.1 s): bounded, never a hang

__device__ __forceinline__ void st_agent32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_agent32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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

// ------------------------------------------------------------------------------------------
// Single-pass scan.  State (zeroed before every launch): [0] ticket, [1] error, then one 64-bit word per tile:
// flag << 32 | value with flag 1 = "tile aggregate", 2 = "inclusive prefix".  Wave 0 looks back over 64 predecessors per step.
// GATHER: element i of the input is in[idx[i]] (the tile counts read in depth-rank order: the gather kernel is folded in).
// `tail` (optional): what used to be k_pair_count -- the workgroup that owns element n-1 knows the grand total and leaves
// min(total, cap) in tail_meta[0], the overflow flag / largest total in tail_status.
// ------------------------------------------------------------------------------------------
#define LB_AGG 1ull
#define LB_INCL 2ull
struct ScanTail { uint32_t* meta; uint32_t* status; uint32_t cap; const uint2* rect; uint4* einfo; };   // rect / einfo: see c3d_scan_u32_einfo

template <bool EXCL, bool GATHER>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_lb(const uint32_t* __restrict__ in, const uint32_t* __restrict__ idx, uint32_t* __restrict__ out, size_t n,
                                                           uint32_t* __restrict__ state, uint32_t* __restrict__ err, ScanTail tail) {
    __shared__ uint32_t lds[4];
    __shared__ uint32_t s_tile, s_prefix;
    unsigned long long* status = reinterpret_cast<unsigned long long*>(state + 2);
    if (threadIdx.x == 0) s_tile = atomicAdd(&state[0], 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const size_t base = (size_t)tile * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    if ((size_t)tile * SCAN_TILE >= n) return;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = (base + i < n) ? (GATHER ? in[idx[base + i]] : in[base + i]) : 0u;
        s += v[i];
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan(s, lds, &tot);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) st_agent64(&status[tile], ((tile == 0 ? LB_INCL : LB_AGG) << 32) | tot);
        uint32_t prefix = 0;
        if (tile > 0) {
            long long t0 = (long long)tile - 1;
            for (;;) {
                const long long t = t0 - lane;
                unsigned long long w = t >= 0 ? ld_agent64(&status[t]) : (LB_INCL << 32);   // below tile 0: an inclusive prefix of zero
                uint64_t incl_mask;
                int first_incl;
                uint32_t spins = 0;
                for (;;) {
                    const uint32_t flag = (uint32_t)(w >> 32);
                    incl_mask = __ballot(flag == (uint32_t)LB_INCL);
                    const uint64_t notready = __ballot(flag == 0u);
                    first_incl = incl_mask ? (int)__builtin_ctzll(incl_mask) : 63;
                    const uint64_t relevant = first_incl >= 63 ? ~0ull : ((2ull << first_incl) - 1ull);
                    if ((notready & relevant) == 0ull) break;
                    if (++spins > LB_SPIN_LIMIT) { if (lane == 0) atomicOr(err, C3D_ERR_LOOKBACK); incl_mask = 1ull; first_incl = 0; w = 0; break; }
                    __builtin_amdgcn_s_sleep(2);
                    if (flag == 0u) w = ld_agent64(&status[t]);
                }
                uint32_t contrib = (lane <= first_incl) ? (uint32_t)w : 0u;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) contrib += __shfl_xor(contrib, o, 64);
                prefix += contrib;
                if (incl_mask) break;
                t0 -= 64;
            }
            if (lane == 0) st_agent64(&status[tile], (LB_INCL << 32) | (uint32_t)(prefix + tot));
        }
        if (lane == 0) s_prefix = prefix;
    }
    __syncthreads();
    uint32_t run = ex + s_prefix;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (EXCL) { if (base + i < n) out[base + i] = run; run += v[i]; }
        else      { run += v[i]; if (base + i < n) out[base + i] = run; }
    }
    if (tail.einfo) {   // epilogue of the record-base scan: {0, tile rect, record base} per element, coalesced (what k_emit and the backward pass gather)
        uint32_t rb = ex + s_prefix;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            if (base + i < n && v[i]) { const uint2 rc = tail.rect[base + i]; tail.einfo[base + i] = make_uint4(0u, rc.x, rc.y, rb); }
            rb += v[i];
        }
    }
    if (tail.meta && base < n && base + SCAN_ITEMS >= n) {      // this thread owns element n-1: `run` is the grand total
        const uint32_t total = run;
        tail.meta[0] = total < tail.cap ? total : tail.cap;
        if (tail.status) { if (total > tail.cap) atomicOr(&tail.status[0], 1u); atomicMax(&tail.status[1], total); }
    }
}

size_t c3d_scan_tmp_bytes(size_t n) { return c3d_align(8 + sizeof(unsigned long long) * (size_t)(c3d_cdiv((long long)(n ? n : 1), SCAN_TILE) + 1)); }

// `zero_state`: false when the caller has already cleared tmp (one memset for several primitives of a view)
static int scan_launch(const uint32_t* in, const uint32_t* idx, uint32_t* out, size_t n, bool exclusive, void* tmp, hipStream_t s, bool zero_state, ScanTail tail, uint32_t* err) {
    if (n == 0) return 0;
    if (zero_state) C3D_CHECK(hipMemsetAsync(tmp, 0, c3d_scan_tmp_bytes(n), s));
    const int nb = c3d_cdiv((long long)n, SCAN_TILE);
    uint32_t* st = (uint32_t*)tmp;
    if (!err) err = st + 1;
    if (idx) {
        if (exclusive) hipLaunchKernelGGL((k_scan_lb<true, true>), dim3(nb), dim3(SCAN_THREADS), 0, s, in, idx, out, n, st, err, tail);
        else           hipLaunchKernelGGL((k_scan_lb<false, true>), dim3(nb), dim3(SCAN_THREADS), 0, s, in, idx, out, n, st, err, tail);
    } else {
        if (exclusive) hipLaunchKernelGGL((k_scan_lb<true, false>), dim3(nb), dim3(SCAN_THREADS), 0, s, in, idx, out, n, st, err, tail);
        else           hipLaunchKernelGGL((k_scan_lb<false, false>), dim3(nb), dim3(SCAN_THREADS), 0, s, in, idx, out, n, st, err, tail);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}
int c3d_scan_u32(const uint32_t* in, uint32_t* out, size_t n, bool exclusive, void* tmp, hipStream_t s, bool zero_state, uint32_t* err) {
    return scan_launch(in, nullptr, out, n, exclusive, tmp, s, zero_state, ScanTail{nullptr, nullptr, 0u, nullptr, nullptr}, err);
}
// exclusive scan of `in` (tile counts in Gaussian-id order) -> out (record bases), plus einfo[i] = {0, rect[i].x, rect[i].y, out[i]} where in[i] != 0
int c3d_scan_u32_einfo(const uint32_t* in, uint32_t* out, size_t n, void* tmp, hipStream_t s, bool zero_state, uint32_t* err, const uint2* rect, uint4* einfo,
                       uint32_t* tail_meta, uint32_t* tail_status, uint32_t tail_cap) {
    return scan_launch(in, nullptr, out, n, true, tmp, s, zero_state, ScanTail{tail_meta, tail_status, tail_cap, rect, einfo}, err);
}
int c3d_scan_gather_u32(const uint32_t* in, const uint32_t* idx, uint32_t* out, size_t n, bool exclusive, void* tmp, hipStream_t s, bool zero_state,
                        uint32_t* tail_meta, uint32_t* tail_status, uint32_t tail_cap, uint32_t* err) {
    return scan_launch(in, idx, out, n, exclusive, tmp, s, zero_state, ScanTail{tail_meta, tail_status, tail_cap, nullptr, nullptr}, err);
}
uint32_t* c3d_scan_error_word(void* tmp) { return (uint32_t*)tmp + 1; }

// ------------------------------------------------------------------------------------------
// Radix sort ("onesweep"): 8-bit digits, 4096 keys per 256-thread workgroup, each wave owns a contiguous 1024-key chunk so that
// ranking is stable by construction.
//   k_radix_hist_all  ONE read of the keys: global digit histograms of ALL passes (digit totals are permutation invariant)
//   k_onesweep        per digit: ballot ranking, block-local reorder in LDS, per-digit chained scan over the tiles, run-contiguous
//                     global stores
// The chained scan is TWO-LEVEL.  At these sizes (245 - 1000 tiles on 256 CUs) every workgroup of a pass is resident at once and
// reaches the scan at the same moment, the regime in which a flat look-back walks ~100 predecessors per tile (1 KB of status words each)
// before it meets an inclusive prefix: 130 MB of polling per pass at 4 M keys.  Tiles are therefore grouped by 16:
//   a tile adds up the aggregates of the <= 15 tiles before it IN ITS GROUP (one batch of independent loads) and takes the rest from the
//   inclusive prefix of the previous group; the last tile of a group publishes the group aggregate, looks back over the GROUP records
//   (32 in flight per lane: 1000 tiles = 62 groups = two batches) and publishes the group's inclusive prefix.
// Three dependent hand-offs whatever the tile count, ~17 KB of status reads per tile.  Every wait is on a tile with a LOWER ticket.
// Digits that do not occur in this pass (ghist == 0: most of the high bytes) skip the scan altogether.
// State (zeroed before every sort): ghist[16 copies][passes][256] | ticket[passes], error | per pass: tile words [tiles][256] then group words
// [tiles/16 + 1][256]; a word = flag << 30 | count, flag 1 = aggregate, 2 = inclusive.
// ------------------------------------------------------------------------------------------
#define RS_THREADS 256
#define RS_ITEMS 16
#define RS_TILE (RS_THREADS * RS_ITEMS)
#define RS_RADIX 256
#define RS_MAX_PASSES 4
#define RS_FLAG_AGG (1u << 30)
#define RS_FLAG_INCL (2u << 30)
#define RS_VALUE_MASK ((1u << 30) - 1u)
#define RS_GROUP 16
#define RS_LOOKBACK 32
#define RS_HIST_SPLIT 16      // copies of the global histogram (workgroup b adds to copy b % 16): 245 - 1000 workgroups adding to ONE set of 256 counters serialise at the
                              // memory-side atomic unit (measured: 23 us for 1 M keys, profiles/r02c); consumers add the 16 copies up
static inline size_t sort_pass_words(size_t nb) { return RS_RADIX * (nb + nb / RS_GROUP + 1); }

__global__ void __launch_bounds__(RS_THREADS) k_radix_hist_all(const uint32_t* __restrict__ keys, uint32_t* __restrict__ ghist, size_t n,
                                                                const uint32_t* __restrict__ n_dev, int passes) {
    __shared__ uint32_t h[RS_MAX_PASSES][RS_RADIX];
    if (n_dev) n = min((size_t)*n_dev, n);      // element count resident on the device (no host round trip)
    const size_t base = (size_t)blockIdx.x * RS_TILE;
    if (base >= n) return;                      // capacity-sized launch: nothing here
    for (int p = 0; p < passes; p++) h[p][threadIdx.x] = 0;
    __syncthreads();
    const int lane = c3d_lane();
#pragma unroll 4
    for (int i = 0; i < RS_ITEMS; i++) {
        const size_t idx = base + (size_t)i * RS_THREADS + threadIdx.x;
        const bool ok = idx < n;
        const uint32_t k = ok ? keys[idx] : 0u;
        const uint64_t okm = __ballot(ok);
        for (int p = 0; p < passes; p++) {
            const uint32_t d = (k >> (8 * p)) & (RS_RADIX - 1);
            // high digits of depth keys / tile ids are nearly constant: a wave whose lanes all hold one digit adds its count once
            // instead of serialising 64 LDS atomics on one counter
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
            if (okm && __ballot(ok && d == d0) == okm) {
                if (lane == (int)__builtin_ctzll(okm)) atomicAdd(&h[p][d0], (uint32_t)__popcll(okm));
            } else if (ok) atomicAdd(&h[p][d], 1u);
        }
    }
    __syncthreads();
    uint32_t* mine = ghist + (size_t)(blockIdx.x % RS_HIST_SPLIT) * RS_MAX_PASSES * RS_RADIX;
    for (int p = 0; p < passes; p++) {
        const uint32_t c = h[p][threadIdx.x];
        if (c) atomicAdd(&mine[p * RS_RADIX + threadIdx.x], c);
    }
}

// poll one status word until its flag is at least `need` (1 = any, 2 = inclusive); bounded
__device__ __forceinline__ uint32_t rs_wait(const uint32_t* p, uint32_t x, uint32_t need, uint32_t* err) {
    uint32_t spins = 0;
    while ((x >> 30) < need) {
        if (++spins > LB_SPIN_LIMIT) { atomicOr(err, C3D_ERR_LOOKBACK); return RS_FLAG_INCL; }
        __builtin_amdgcn_s_sleep(2);
        x = ld_agent32(p);
    }
    return x;
}

// ITEMS keys per thread: 16 (4096 keys and 39 KB of LDS per workgroup: four workgroups per CU) or 8 (2048 keys, 22 KB: seven per CU, twice the tiles) -- C3D_SORT_ITEMS
template <bool IOTA, int ITEMS>
__global__ void __launch_bounds__(RS_THREADS, ITEMS > 16 ? 2 : 4) k_onesweep(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                             const uint32_t* __restrict__ ghist, uint32_t* __restrict__ ticket, uint32_t* __restrict__ err,
                                                             uint32_t* __restrict__ tile_words, uint32_t* __restrict__ group_words, size_t n,
                                                             const uint32_t* __restrict__ n_dev, int shift, unsigned long long* __restrict__ dbg) {
    __shared__ uint32_t whist[RS_THREADS / 64][RS_RADIX];
    __shared__ uint32_t lstart[RS_RADIX];    // first local slot of each digit
    __shared__ uint32_t gbase[RS_RADIX];     // global position of that slot
    __shared__ uint32_t skey[RS_THREADS * ITEMS];
    __shared__ uint32_t sval[RS_THREADS * ITEMS];
    __shared__ uint32_t scan_lds[4];
    __shared__ uint32_t s_tile;
    if (n_dev) n = min((size_t)*n_dev, n);
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    const int lane = c3d_lane(), wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (RS_THREADS / 64) * RS_RADIX; i += RS_THREADS) (&whist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const size_t bbase = (size_t)tile * (RS_THREADS * ITEMS);
    if (bbase >= n) return;                  // capacity-sized launch: tickets beyond the data leave at once (nobody waits for them)
#define RS_STAMP(k) do { if (dbg && threadIdx.x == 0) dbg[(size_t)tile * 8 + (k)] = (unsigned long long)wall_clock64(); } while (0)   // profiling hook (profiles/microbench/sort_phases.py)
    RS_STAMP(0);
    const size_t wbase = bbase + (size_t)wave * (RS_THREADS * ITEMS / 4);
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        size_t idx = wbase + (size_t)i * 64 + lane;
        bool ok = idx < n;
        key[i] = ok ? keys_in[idx] : 0xFFFFFFFFu;
        val[i] = IOTA ? (uint32_t)idx : (ok ? vals_in[idx] : 0u);
    }
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (dbg) { uint32_t x = 0; for (int i = 0; i < ITEMS; i++) x ^= key[i] ^ val[i]; if (x == 0x12345u) dbg[7] = 1; }   // wait for the loads before stamping
    RS_STAMP(1);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
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
    RS_STAMP(2);
    // thread d owns digit d: count over the 4 waves, published at once (successors can already add it up), then the block-local layout
    const int d = threadIdx.x;
    const uint32_t grp = tile / RS_GROUP, gr = tile % RS_GROUP;
    const bool leader = gr == RS_GROUP - 1;
    uint32_t gtotal = 0;                                 // occurrences of this digit in the whole input
#pragma unroll
    for (int c = 0; c < RS_HIST_SPLIT; c++) gtotal += ghist[(size_t)c * RS_MAX_PASSES * RS_RADIX + d];
    uint32_t tot = 0;
    {
        uint32_t c[RS_THREADS / 64];
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; w++) { c[w] = whist[w][d]; tot += c[w]; }
        if (gtotal) st_agent32(tile_words + (size_t)tile * RS_RADIX + d, RS_FLAG_AGG | tot);
        uint32_t blk_total, dummy;
        uint32_t ls = block_excl_scan(tot, scan_lds, &blk_total);
        const uint32_t digit_base = block_excl_scan(gtotal, scan_lds, &dummy);   // where digit d starts in the output
        lstart[d] = ls;
        gbase[d] = digit_base;
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; w++) { whist[w][d] = ls; ls += c[w]; }
    }
    __syncthreads();
    // block-local reorder first: it needs no global prefix and frees the key / val / rank registers for the scan's loads
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        size_t idx = wbase + (size_t)i * 64 + lane;
        if (idx < n) {
            uint32_t dd = (key[i] >> shift) & (RS_RADIX - 1);
            uint32_t lp = whist[wave][dd] + rank[i];
            skey[lp] = key[i];
            sval[lp] = val[i];
        }
    }
    RS_STAMP(3);
    // chained scan of digit d over the tiles (two-level, see the header of this section)
    if (gtotal) {
        uint32_t sum_in = 0;
        {   // aggregates of the tiles before this one in its group: all loads first, then resolve
            const uint32_t* base = tile_words + (size_t)grp * RS_GROUP * RS_RADIX + d;
            uint32_t w[RS_GROUP - 1];
#pragma unroll
            for (int i = 0; i < RS_GROUP - 1; i++) w[i] = ((uint32_t)i < gr) ? ld_agent32(base + (size_t)i * RS_RADIX) : RS_FLAG_AGG;
#pragma unroll
            for (int i = 0; i < RS_GROUP - 1; i++)
                if ((uint32_t)i < gr) sum_in += rs_wait(base + (size_t)i * RS_RADIX, w[i], 1u, err) & RS_VALUE_MASK;
        }
        uint32_t gexcl = 0;          // everything in earlier groups
        uint32_t* gw = group_words + d;
        if (leader) {
            const uint32_t gtot = sum_in + tot;
            st_agent32(gw + (size_t)grp * RS_RADIX, (grp == 0 ? RS_FLAG_INCL : RS_FLAG_AGG) | gtot);
            if (grp > 0) {
                long long t = (long long)grp - 1;
                bool done = false;
                while (!done) {
                    uint32_t w[RS_LOOKBACK];
#pragma unroll
                    for (int i = 0; i < RS_LOOKBACK; i++) w[i] = (t - i >= 0) ? ld_agent32(gw + (size_t)(t - i) * RS_RADIX) : RS_FLAG_INCL;
#pragma unroll
                    for (int i = 0; i < RS_LOOKBACK; i++) {
                        if (!done) {
                            const uint32_t x = (t - i >= 0) ? rs_wait(gw + (size_t)(t - i) * RS_RADIX, w[i], 1u, err) : RS_FLAG_INCL;
                            gexcl += x & RS_VALUE_MASK;
                            done = (x >> 30) == 2u;
                        }
                    }
                    t -= RS_LOOKBACK;
                }
                st_agent32(gw + (size_t)grp * RS_RADIX, RS_FLAG_INCL | (gexcl + gtot));
            }
        } else if (grp > 0) {
            const uint32_t* p = gw + (size_t)(grp - 1) * RS_RADIX;
            gexcl = rs_wait(p, ld_agent32(p), 2u, err) & RS_VALUE_MASK;
        }
        gbase[d] += gexcl + sum_in;
    }
    __syncthreads();
    RS_STAMP(4);
    const int cnt = (int)((n - bbase) < (size_t)(RS_THREADS * ITEMS) ? (n - bbase) : (size_t)(RS_THREADS * ITEMS));
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int lp = i * RS_THREADS + threadIdx.x;     // consecutive lanes -> consecutive slots of a digit run
        if (lp < cnt) {
            const uint32_t k = skey[lp];
            const uint32_t dd = (k >> shift) & (RS_RADIX - 1);
            const uint32_t pos = gbase[dd] + ((uint32_t)lp - lstart[dd]);
            keys_out[pos] = k;
            vals_out[pos] = sval[lp];
        }
    }
    RS_STAMP(5);
#undef RS_STAMP
}

static unsigned long long* g_sort_dbg = nullptr;     // profiling hook: [pass][tile][8] wall_clock64 stamps (100 MHz), see c3d_test_sort_phases
static inline size_t sort_head_bytes() { return c3d_align(sizeof(uint32_t) * (RS_HIST_SPLIT * RS_RADIX * RS_MAX_PASSES + RS_MAX_PASSES + 4)); }
static int sort_items() {      // keys per thread of k_onesweep (see there)
    static int v = -1;
    if (v < 0) { const char* e = getenv("C3D_SORT_ITEMS"); v = e ? atoi(e) : 16; if (v != 8 && v != 16 && v != 32) v = 16; }
    return v;
}
#define RS_MIN_TILE (RS_THREADS * 8)      // the state is sized for the smaller tile
size_t c3d_sort_tmp_bytes(size_t n) {
    const size_t nb = (size_t)c3d_cdiv((long long)(n ? n : 1), RS_MIN_TILE);
    return sort_head_bytes() + c3d_align(sizeof(uint32_t) * sort_pass_words(nb) * RS_MAX_PASSES);
}
uint32_t* c3d_sort_error_word(void* tmp) { return (uint32_t*)tmp + RS_HIST_SPLIT * RS_RADIX * RS_MAX_PASSES + RS_MAX_PASSES; }
size_t c3d_sort_state_bytes(size_t n, int end_bit) {
    int passes = (end_bit + 7) / 8;
    if (passes < 1) passes = 1;
    if (passes > RS_MAX_PASSES) passes = RS_MAX_PASSES;
    return sort_head_bytes() + sizeof(uint32_t) * sort_pass_words((size_t)c3d_cdiv((long long)(n ? n : 1), RS_MIN_TILE)) * passes;
}

int c3d_sort_pairs_u32(uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1, bool iota_vals,
                       size_t n, int end_bit, void* tmp, int* result_buf, hipStream_t s, const uint32_t* n_dev, bool zero_state, uint32_t* err_out) {
    *result_buf = 0;
    if (n == 0) return 0;
    if (end_bit > 8 * RS_MAX_PASSES) { c3d_set_error("c3d_sort_pairs_u32: end_bit %d > %d", end_bit, 8 * RS_MAX_PASSES); return -1; }
    if (n > (size_t)RS_VALUE_MASK) { c3d_set_error("c3d_sort_pairs_u32: %zu elements exceed the 2^30 - 1 the chained scan's status words hold", n); return -1; }
    int passes = (end_bit + 7) / 8;
    if (passes < 1) passes = 1;
    const int items = sort_items();
    const int nb = c3d_cdiv((long long)n, RS_THREADS * items), nb_hist = c3d_cdiv((long long)n, RS_TILE);
    uint32_t* ghist = (uint32_t*)tmp;
    uint32_t* tickets = ghist + RS_HIST_SPLIT * RS_RADIX * RS_MAX_PASSES;
    uint32_t* err = err_out ? err_out : c3d_sort_error_word(tmp);
    uint32_t* status = (uint32_t*)((char*)tmp + sort_head_bytes());
    if (zero_state) C3D_CHECK(hipMemsetAsync(tmp, 0, c3d_sort_state_bytes(n, end_bit), s));
    uint32_t* k[2] = {keys0, keys1};
    uint32_t* v[2] = {vals0, vals1};
    hipLaunchKernelGGL(k_radix_hist_all, dim3(nb_hist), dim3(RS_THREADS), 0, s, keys0, ghist, n, n_dev, passes);
    int cur = 0;
    for (int pass = 0; pass < passes; pass++) {
        uint32_t* tw = status + (size_t)pass * sort_pass_words((size_t)nb);
        uint32_t* gw = tw + (size_t)RS_RADIX * nb;
#define RS_SWEEP(IOTA_, ITEMS_) hipLaunchKernelGGL((k_onesweep<IOTA_, ITEMS_>), dim3(nb), dim3(RS_THREADS), 0, s, k[cur], v[cur], k[cur ^ 1], v[cur ^ 1], ghist + pass * RS_RADIX, \
                                                   tickets + pass, err, tw, gw, n, n_dev, 8 * pass, g_sort_dbg ? g_sort_dbg + (size_t)pass * nb * 8 : nullptr)
        if (pass == 0 && iota_vals) { if (items == 8) RS_SWEEP(true, 8); else if (items == 32) RS_SWEEP(true, 32); else RS_SWEEP(true, 16); }
        else { if (items == 8) RS_SWEEP(false, 8); else if (items == 32) RS_SWEEP(false, 32); else RS_SWEEP(false, 16); }
#undef RS_SWEEP
        C3D_LAUNCH_CHECK();
        cur ^= 1;
    }
    *result_buf = cur;
    return 0;
}

// ------------------------------------------------------------------------------------------
// Segmented sort (round 3): every segment [ranges[t].x, ranges[t].y) of `vals` is reordered by key_table[val], ascending and STABLE (equal keys keep their
// input order).  The 3DGS binning uses it for the depth order INSIDE each tile's list -- the lists leave the tile sort in Gaussian-id order -- in place of a global
// depth sort of all Gaussians before emission (five latency-bound launches + a second chained scan per view).  One 256-thread workgroup per segment.
//   n <= SEG_CAP: LSD radix sort on 8-bit digits entirely in LDS: ballot ranking as in k_onesweep (each wave owns a contiguous chunk -> stable), keys and values
//                 live in registers between passes, digits on which all keys of the segment agree (the high bytes of a tile's depths, usually) are skipped;
//   n >  SEG_CAP: the same ranking chunk by chunk through a global ping-pong of the values (vals <-> vbuf1; keys are re-gathered from the table; four passes, the
//                 result ends in vals); rare (a tile with more than 4096 splats) and correct rather than fast.
// ------------------------------------------------------------------------------------------
#define SEG_CAP 4096
#define SEG_ITEMS (SEG_CAP / 256)
// rank of each of the wave's items among the items of this workgroup chunk that precede it with the same digit, within the wave (+ the wave's running count of the
// digit in whist[wave][]); item i of lane l is element i * 64 + l of the wave's contiguous run.  whist must be zero on entry.
// Element p = first + i * 64 + lane exists iff p < n.
__device__ __forceinline__ void seg_rank(const uint32_t key[SEG_ITEMS], uint32_t first, uint32_t n, int items, int shift, uint32_t (*whist)[RS_RADIX], int wave, int lane,
                                         uint32_t rank[SEG_ITEMS]) {
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int i = 0; i < SEG_ITEMS; i++) {
        if (i < items) {
            const bool ok = first + (uint32_t)i * 64u + (uint32_t)lane < n;
            const uint32_t d = (key[i] >> shift) & (RS_RADIX - 1);
            uint64_t peers = __ballot(ok);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const uint64_t m = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            const uint32_t prefix = whist[wave][d];
            const uint32_t r = (uint32_t)__popcll(peers & lt_mask);
            if (ok && r == 0) whist[wave][d] = prefix + (uint32_t)__popcll(peers);
            rank[i] = prefix + r;
        }
    }
}
// Segments of up to SEGW_CAP elements: ONE WAVE per segment (64-thread workgroups), nothing to synchronise with -- a tile's list is ~500 entries, and the
// 256-thread kernel below spends its time in ~7 barriers per pass and in per-pass costs that do not shrink with the segment (256-digit tables for 4 waves).
#define SEGW_CAP 1024
__global__ void __launch_bounds__(64, 4) k_segment_sort_w(const uint2* __restrict__ ranges, int nseg, const uint32_t* __restrict__ key_table, uint32_t* __restrict__ vals) {
    __shared__ uint32_t cnt[1][RS_RADIX];
    __shared__ uint32_t skey[SEGW_CAP];
    __shared__ uint32_t sval[SEGW_CAP];
    const int t = blockIdx.x;
    if (t >= nseg) return;
    const uint2 rg = ranges[t];
    const uint32_t n = rg.y > rg.x ? rg.y - rg.x : 0u;
    if (n <= 1u || n > SEGW_CAP) return;
    const int lane = threadIdx.x;
    const int items = (int)((n + 63u) / 64u);
    uint32_t key[SEG_ITEMS], val[SEG_ITEMS], rank[SEG_ITEMS];
    uint32_t vor = 0u, vand = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < SEG_ITEMS; i++) {
        key[i] = 0xFFFFFFFFu; val[i] = 0u;
        if (i < items) {
            const uint32_t p = (uint32_t)i * 64u + (uint32_t)lane;
            if (p < n) { val[i] = vals[rg.x + p]; key[i] = key_table[val[i]]; vor |= key[i]; vand &= key[i]; }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { vor |= (uint32_t)__shfl_xor((int)vor, o, 64); vand &= (uint32_t)__shfl_xor((int)vand, o, 64); }
    const uint32_t diff = vor & ~vand;
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 8 * pass;
        if (!((diff >> shift) & 0xFFu)) continue;
        __syncthreads();                                   // one wave: these cost nothing and keep the LDS accesses of the phases in program order
#pragma unroll
        for (int k = 0; k < 4; k++) cnt[0][k * 64 + lane] = 0u;
        __syncthreads();
        seg_rank(key, 0u, n, items, shift, cnt, 0, lane, rank);
        __syncthreads();
        {   // lane l owns digits 4l .. 4l+3: counts -> starts
            const uint32_t c0 = cnt[0][4 * lane], c1 = cnt[0][4 * lane + 1], c2 = cnt[0][4 * lane + 2], c3 = cnt[0][4 * lane + 3];
            const uint32_t tot = c0 + c1 + c2 + c3;
            const uint32_t ex = c3d_wave_incl_scan(tot) - tot;
            cnt[0][4 * lane] = ex; cnt[0][4 * lane + 1] = ex + c0; cnt[0][4 * lane + 2] = ex + c0 + c1; cnt[0][4 * lane + 3] = ex + c0 + c1 + c2;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SEG_ITEMS; i++)
            if (i < items && (uint32_t)i * 64u + (uint32_t)lane < n) {
                const uint32_t lp = cnt[0][(key[i] >> shift) & (RS_RADIX - 1)] + rank[i];
                skey[lp] = key[i]; sval[lp] = val[i];
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SEG_ITEMS; i++) {
            const uint32_t p = (uint32_t)i * 64u + (uint32_t)lane;
            if (i < items && p < n) { key[i] = skey[p]; val[i] = sval[p]; }
        }
    }
#pragma unroll
    for (int i = 0; i < SEG_ITEMS; i++) {
        const uint32_t p = (uint32_t)i * 64u + (uint32_t)lane;
        if (i < items && p < n) vals[rg.x + p] = val[i];
    }
}
__global__ void __launch_bounds__(256, 4) k_segment_sort(const uint2* __restrict__ ranges, int nseg, const uint32_t* __restrict__ key_table, uint32_t* __restrict__ vals,
                                                       uint32_t* __restrict__ vbuf1) {
    __shared__ uint32_t whist[4][RS_RADIX];
    __shared__ uint32_t dstart[RS_RADIX];     // big path: where the next element of each digit goes
    __shared__ uint32_t skey[SEG_CAP];
    __shared__ uint32_t sval[SEG_CAP];
    __shared__ uint32_t scan_lds[4];
    __shared__ uint32_t s_or[4], s_and[4];
    const int t = blockIdx.x;
    if (t >= nseg) return;
    const uint2 rg = ranges[t];
    const uint32_t n = rg.y > rg.x ? rg.y - rg.x : 0u;
    if (n <= SEGW_CAP) return;                                  // k_segment_sort_w's
    const int lane = c3d_lane(), wave = threadIdx.x >> 6;
    uint32_t key[SEG_ITEMS], val[SEG_ITEMS], rank[SEG_ITEMS];
    if (n <= SEG_CAP) {
        const int items = (int)((n + 255u) / 256u);               // per lane; a wave's run is items * 64 consecutive elements
        const uint32_t wbase = (uint32_t)wave * (uint32_t)items * 64u;
        uint32_t vor = 0u, vand = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < SEG_ITEMS; i++) {
            key[i] = 0xFFFFFFFFu; val[i] = 0u;
            if (i < items) {
                const uint32_t p = wbase + (uint32_t)i * 64u + (uint32_t)lane;
                if (p < n) { val[i] = vals[rg.x + p]; key[i] = key_table[val[i]]; vor |= key[i]; vand &= key[i]; }
            }
        }
        // bits on which the keys of the segment differ: a digit with none is a pass that would not move anything
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { vor |= (uint32_t)__shfl_xor((int)vor, o, 64); vand &= (uint32_t)__shfl_xor((int)vand, o, 64); }
        if (lane == 0) { s_or[wave] = vor; s_and[wave] = vand; }
        __syncthreads();
        const uint32_t diff = (s_or[0] | s_or[1] | s_or[2] | s_or[3]) & ~(s_and[0] & s_and[1] & s_and[2] & s_and[3]);
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 8 * pass;
            if (!((diff >> shift) & 0xFFu)) continue;            // uniform over the workgroup
            __syncthreads();
            for (int i = threadIdx.x; i < 4 * RS_RADIX; i += 256) (&whist[0][0])[i] = 0;
            __syncthreads();
            seg_rank(key, wbase, n, items, shift, whist, wave, lane, rank);
            __syncthreads();
            {   // thread d owns digit d: block-local start of every (wave, digit) run
                const int d = threadIdx.x;
                uint32_t c[4], tot = 0, dummy;
#pragma unroll
                for (int w = 0; w < 4; w++) { c[w] = whist[w][d]; tot += c[w]; }
                uint32_t ls = block_excl_scan(tot, scan_lds, &dummy);
#pragma unroll
                for (int w = 0; w < 4; w++) { whist[w][d] = ls; ls += c[w]; }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < SEG_ITEMS; i++)
                if (i < items && wbase + (uint32_t)i * 64u + (uint32_t)lane < n) {
                    const uint32_t lp = whist[wave][(key[i] >> shift) & (RS_RADIX - 1)] + rank[i];
                    skey[lp] = key[i]; sval[lp] = val[i];
                }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < SEG_ITEMS; i++) {
                const uint32_t p = wbase + (uint32_t)i * 64u + (uint32_t)lane;
                if (i < items && p < n) { key[i] = skey[p]; val[i] = sval[p]; }
            }
        }
#pragma unroll
        for (int i = 0; i < SEG_ITEMS; i++) {
            const uint32_t p = wbase + (uint32_t)i * 64u + (uint32_t)lane;
            if (i < items && p < n) vals[rg.x + p] = val[i];
        }
        return;
    }
    // ---- more than SEG_CAP elements: four passes through a global ping-pong of the values, chunk by chunk in order (agent-scope accesses: the workgroup re-reads
    //      what its other waves wrote in the previous pass)
    uint32_t *vsrc = vals + rg.x, *vdst = vbuf1 + rg.x;
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 8 * pass;
        dstart[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < n; j += 256) atomicAdd(&dstart[(key_table[ld_agent32(vsrc + j)] >> shift) & (RS_RADIX - 1)], 1u);
        __syncthreads();
        {
            uint32_t dummy;
            const uint32_t cnt = dstart[threadIdx.x];
            const uint32_t ex = block_excl_scan(cnt, scan_lds, &dummy);
            __syncthreads();
            dstart[threadIdx.x] = ex;
        }
        __syncthreads();
        for (uint32_t cbase = 0; cbase < n; cbase += SEG_CAP) {
            for (int i = threadIdx.x; i < 4 * RS_RADIX; i += 256) (&whist[0][0])[i] = 0;
            __syncthreads();
            const uint32_t wbase = cbase + (uint32_t)wave * (SEG_CAP / 4);
#pragma unroll
            for (int i = 0; i < SEG_ITEMS; i++) {
                const uint32_t p = wbase + (uint32_t)i * 64u + (uint32_t)lane;
                val[i] = p < n ? ld_agent32(vsrc + p) : 0u;
                key[i] = p < n ? key_table[val[i]] : 0xFFFFFFFFu;
            }
            seg_rank(key, wbase, n, SEG_ITEMS, shift, whist, wave, lane, rank);
            __syncthreads();
            {   // thread d: this chunk's elements of digit d go to dstart[d] .. in wave order
                const int d = threadIdx.x;
                uint32_t ls = dstart[d], tot = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) { const uint32_t c = whist[w][d]; whist[w][d] = ls + tot; tot += c; }
                dstart[d] = ls + tot;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < SEG_ITEMS; i++)
                if (wbase + (uint32_t)i * 64u + (uint32_t)lane < n) {
                    const uint32_t pos = whist[wave][(key[i] >> shift) & (RS_RADIX - 1)] + rank[i];
                    st_agent32(vdst + pos, val[i]);
                }
            __syncthreads();
        }
        __threadfence();
        __syncthreads();
        { uint32_t* x = vsrc; vsrc = vdst; vdst = x; }
    }
    // four passes: the result is back in vals
}
// vals: in / out.  vbuf1: scratch of at least the same length as vals (touched only by segments longer than 4096).
int c3d_segment_sort_u32(const uint2* ranges, int nseg, const uint32_t* key_table, uint32_t* vals, uint32_t* vbuf1, hipStream_t s) {
    if (nseg <= 0) return 0;
    hipLaunchKernelGGL(k_segment_sort_w, dim3(nseg), dim3(64), 0, s, ranges, nseg, key_table, vals);
    hipLaunchKernelGGL(k_segment_sort, dim3(nseg), dim3(256), 0, s, ranges, nseg, key_table, vals, vbuf1);
    C3D_LAUNCH_CHECK();
    return 0;
}

// test / profiling hook: one sort with per-tile phase stamps -> stamps[passes][tiles][8] (device), wall_clock64 ticks
int c3d_sort_set_debug(unsigned long long* stamps) { g_sort_dbg = stamps; return 0; }
