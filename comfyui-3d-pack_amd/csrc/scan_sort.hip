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
//
// Round 4: every kernel takes a VIEW dimension (blockIdx.y = view, `vs` = bytes between the workspace slices of consecutive views; all per-view
// pointers -- keys, values, state, device-resident counts -- are given for view 0 and advanced by blockIdx.y * vs).  The V views of a step are
// sorted / scanned by ONE launch per stage: a 1 M-key pass is 16 MB of traffic and 17 us of fixed latencies, eight of them in one launch are
// bandwidth-sized work.  Tickets, status words and look-backs are per view (a workgroup only ever waits for lower tickets of ITS view); the error word is shared.
#include <type_traits>
#include "c3d_common.h"


#define SCAN_THREADS 256
#ifndef SCAN_VEC
#define SCAN_VEC 4                       // 16-byte chunks per lane: a tile is SCAN_VEC sub-tiles of 1024 elements
#endif
#define SCAN_ITEMS (4 * SCAN_VEC)
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
// GATHER (the emit-offset scan): element i of the input is the AREA of the tile rect rect[idx[i]] (in = the rects as uint2 {x0 | y0 << 16, x1 | y1 << 16},
// idx = the depth order), and the gathered rect is left behind in rank order (tail.rsort[i], coalesced) so that the emission needs no gather of its own: ONE random
// 8-byte access per Gaussian and view instead of two (a 4-byte count here, a 16-byte rect in k_emit), each of which costs a whole 64-byte line.
// `tail` (optional): what used to be k_pair_count -- the workgroup that owns element n-1 knows the grand total and leaves
// min(total, cap) in tail_meta[0], the overflow flag / largest total in tail_status.
// ------------------------------------------------------------------------------------------
#define LB_AGG 1ull
#define LB_INCL 2ull
// hint (0 = none): the pair count the LAUNCHES of the chain were sized for (c3d_gs_forward_nosync: first_capacity < cap); a larger count is served by workgroups that loop --
// status bit C3D_ST_BEYOND_HINT only says so.  early (optional): DEVICE-VISIBLE address of two words of pinned host memory; the tail stores {overflow / hint bits, total} there
// itself, as ONE 64-bit system-scope store -- the host sees the count of a call while the rest of its chain is still running, without a copy in the stream.
struct ScanTail { uint32_t* meta; uint32_t* status; uint32_t cap; uint2* rsort; uint32_t hint; unsigned long long* early; int rect4; };   // rsort, rect4 (the input holds packed 4-byte rects): GATHER
__device__ __forceinline__ uint32_t rect_area(uint2 r) { return ((r.y & 0xFFFFu) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.x >> 16)); }

// Data movement (round 4: with V views per launch the scans are bandwidth-sized work, and eight 4-byte accesses per lane at a 32-byte lane stride cost eight
// partially used cache lines per wave instruction): a tile is SCAN_VEC sub-tiles of 1024 elements and lane t owns elements [4t, 4t + 4) of EACH sub-tile -- every
// load / store is 16 bytes per lane on consecutive addresses.  The block scan runs over the SCAN_VEC sub-tile sums of every lane at once.
template <bool EXCL, bool GATHER>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_lb(const uint32_t* __restrict__ in, const uint32_t* __restrict__ idx, uint32_t* __restrict__ out, size_t n,
                                                           uint32_t* __restrict__ state, uint32_t* __restrict__ err, ScanTail tail, size_t vs) {
    __shared__ uint32_t lds[SCAN_VEC][4];
    __shared__ uint32_t s_tile, s_prefix;
    // view = (linear block id) % V: one XCD per view at V = 8 (see k_onesweep) -- the gather of the emit-offset scan then finds half of a view's 8 MB of rects in its
    // XCD's L2 instead of an eighth of eight views': 0.153 -> 0.133 ms for the 8-view step (profiles/r05r_*).  All views hold N elements: no stealing needed.
    const size_t view_off = (size_t)((blockIdx.x + gridDim.x * blockIdx.y) % gridDim.y) * vs;
#define SC_VP(p_) ((p_) ? (decltype(p_))((char*)(p_) + view_off) : (p_))
    in = SC_VP(in); idx = SC_VP(idx); out = SC_VP(out); state = SC_VP(state);
    tail.meta = SC_VP(tail.meta); tail.rsort = SC_VP(tail.rsort);   // err / tail.status: shared by the views
#undef SC_VP
    unsigned long long* status = reinterpret_cast<unsigned long long*>(state + 2);
    if (threadIdx.x == 0) s_tile = atomicAdd(&state[0], 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    if ((size_t)tile * SCAN_TILE >= n) return;
    const int lane = c3d_lane(), wave = threadIdx.x >> 6;
    uint32_t v[SCAN_VEC][4];
    uint32_t hs[SCAN_VEC];
#pragma unroll
    for (int h = 0; h < SCAN_VEC; h++) {
        const size_t b = (size_t)tile * SCAN_TILE + (size_t)h * (SCAN_THREADS * 4) + (size_t)threadIdx.x * 4;
        if (GATHER) {
            const uint2* rects = reinterpret_cast<const uint2*>(in);
            uint2 rc[4];
            if (tail.rect4) {      // (uniform) 4-byte packed rects: the gather table of a view is half the size
                if (b + 3 < n) {
                    const uint4 j = *reinterpret_cast<const uint4*>(idx + b);
                    const uint32_t q0 = in[j.x], q1 = in[j.y], q2 = in[j.z], q3 = in[j.w];
                    rc[0] = c3d_rect_unpack(q0); rc[1] = c3d_rect_unpack(q1); rc[2] = c3d_rect_unpack(q2); rc[3] = c3d_rect_unpack(q3);
                    *reinterpret_cast<uint4*>(tail.rsort + b) = make_uint4(rc[0].x, rc[0].y, rc[1].x, rc[1].y);
                    *reinterpret_cast<uint4*>(tail.rsort + b + 2) = make_uint4(rc[2].x, rc[2].y, rc[3].x, rc[3].y);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) { rc[i] = (b + i < n) ? c3d_rect_unpack(in[idx[b + i]]) : make_uint2(0u, 0u); if (b + i < n) tail.rsort[b + i] = rc[i]; }
                }
            } else if (b + 3 < n) {
                const uint4 j = *reinterpret_cast<const uint4*>(idx + b);
                rc[0] = rects[j.x]; rc[1] = rects[j.y]; rc[2] = rects[j.z]; rc[3] = rects[j.w];
                *reinterpret_cast<uint4*>(tail.rsort + b) = make_uint4(rc[0].x, rc[0].y, rc[1].x, rc[1].y);
                *reinterpret_cast<uint4*>(tail.rsort + b + 2) = make_uint4(rc[2].x, rc[2].y, rc[3].x, rc[3].y);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) { rc[i] = (b + i < n) ? rects[idx[b + i]] : make_uint2(0u, 0u); if (b + i < n) tail.rsort[b + i] = rc[i]; }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) v[h][i] = rect_area(rc[i]);
        } else if (b + 3 < n) {
            const uint4 q = *reinterpret_cast<const uint4*>(in + b); v[h][0] = q.x; v[h][1] = q.y; v[h][2] = q.z; v[h][3] = q.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) v[h][i] = (b + i < n) ? in[b + i] : 0u;
        }
        hs[h] = v[h][0] + v[h][1] + v[h][2] + v[h][3];
    }
    // block-wide exclusive scan of all sub-tile sums at once
    uint32_t ex[SCAN_VEC], tot = 0;
#pragma unroll
    for (int h = 0; h < SCAN_VEC; h++) {
        const uint32_t inc = c3d_wave_incl_scan(hs[h]);
        if (lane == 63) lds[h][wave] = inc;
        ex[h] = inc - hs[h];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < SCAN_VEC; h++) {
        ex[h] += tot;                                // a sub-tile follows all of the sub-tiles before it
#pragma unroll
        for (int w = 0; w < SCAN_THREADS / 64; w++) {
            const uint32_t t = lds[h][w];
            if (w < wave) ex[h] += t;
            tot += t;
        }
    }
    if (threadIdx.x < 64) {
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
    const uint32_t pre = s_prefix;
#pragma unroll
    for (int h = 0; h < SCAN_VEC; h++) {
        const size_t b = (size_t)tile * SCAN_TILE + (size_t)h * (SCAN_THREADS * 4) + (size_t)threadIdx.x * 4;
        uint32_t run = pre + ex[h], o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t e = run; run += v[h][i]; o[i] = EXCL ? e : run; }
        if (b + 3 < n) *reinterpret_cast<uint4*>(out + b) = make_uint4(o[0], o[1], o[2], o[3]);
        else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (b + i < n) out[b + i] = o[i];
        }
        if (tail.meta && b < n && b + 4 >= n) {      // this lane owns element n-1: `run` is the grand total (elements past n are zeros)
            const uint32_t total = run;
            tail.meta[0] = total < tail.cap ? total : tail.cap;
            const uint32_t bits = (total > tail.cap ? C3D_ST_OVERFLOW : 0u) | ((tail.hint && total > tail.hint) ? C3D_ST_BEYOND_HINT : 0u);
            if (tail.status) { if (bits) atomicOr(&tail.status[0], bits); atomicMax(&tail.status[1], total); }
            if (tail.early) __hip_atomic_store(tail.early, ((unsigned long long)total << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

size_t c3d_scan_tmp_bytes(size_t n) { return c3d_align(8 + sizeof(unsigned long long) * (size_t)(c3d_cdiv((long long)(n ? n : 1), SCAN_TILE) + 1)); }

// `zero_state`: false when the caller has already cleared tmp (one memset for several primitives of a view)
static int scan_launch(const uint32_t* in, const uint32_t* idx, uint32_t* out, size_t n, bool exclusive, void* tmp, hipStream_t s, bool zero_state, ScanTail tail, uint32_t* err,
                       int V = 1, size_t vs = 0) {
    if (n == 0 || V <= 0) return 0;
    if (zero_state) {
        if (V != 1) { c3d_set_error("scan: a multi-view launch clears its state through c3d_zero_views"); return -1; }
        C3D_CHECK(hipMemsetAsync(tmp, 0, c3d_scan_tmp_bytes(n), s));
    }
    const dim3 grid(c3d_cdiv((long long)n, SCAN_TILE), V);
    uint32_t* st = (uint32_t*)tmp;
    if (!err) err = st + 1;
    if (idx) {
        hipLaunchKernelGGL((k_scan_lb<false, true>), grid, dim3(SCAN_THREADS), 0, s, in, idx, out, n, st, err, tail, vs);
    } else {
        if (exclusive) hipLaunchKernelGGL((k_scan_lb<true, false>), grid, dim3(SCAN_THREADS), 0, s, in, idx, out, n, st, err, tail, vs);
        else           hipLaunchKernelGGL((k_scan_lb<false, false>), grid, dim3(SCAN_THREADS), 0, s, in, idx, out, n, st, err, tail, vs);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}
int c3d_scan_u32(const uint32_t* in, uint32_t* out, size_t n, bool exclusive, void* tmp, hipStream_t s, bool zero_state, uint32_t* err) {
    return scan_launch(in, nullptr, out, n, exclusive, tmp, s, zero_state, ScanTail{nullptr, nullptr, 0u, nullptr, 0u, nullptr, 0}, err);
}

// ------------------------------------------------------------------------------------------
// ONE launch clears up to three byte regions in each of V workspace slices (the states of a view's single-pass primitives, its tile ranges, its
// "record written" bytes): region r of view v = base0 + v * vs + off[r], bytes[r] long (multiples of 4; 16-byte aligned starts).
// Replaces 3 V hipMemsetAsync calls (each of them a kernel launch of its own).
// ------------------------------------------------------------------------------------------
struct ZeroRegions { size_t off[3], bytes[3]; };
__global__ void __launch_bounds__(256) k_zero_views(char* __restrict__ base0, size_t vs, ZeroRegions zr) {
    const int r = blockIdx.z;
    char* p = base0 + (size_t)blockIdx.y * vs + zr.off[r];
    const size_t n16 = zr.bytes[r] / 16, n4 = (zr.bytes[r] % 16) / 4;
    uint4* p16 = reinterpret_cast<uint4*>(p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p16[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < n4) reinterpret_cast<uint32_t*>(p + n16 * 16)[threadIdx.x] = 0u;
}
int c3d_zero_views(void* base0, size_t vs, int V, const size_t* off, const size_t* bytes, int regions, hipStream_t s) {
    if (V <= 0 || regions <= 0) return 0;
    if (regions > 3) { c3d_set_error("c3d_zero_views: at most three regions"); return -1; }
    ZeroRegions zr{};
    size_t most = 0;
    for (int r = 0; r < regions; r++) {
        if ((off[r] & 15) || (bytes[r] & 3)) { c3d_set_error("c3d_zero_views: region %d is not 16-byte aligned / a multiple of 4 bytes", r); return -1; }
        zr.off[r] = off[r]; zr.bytes[r] = bytes[r];
        if (bytes[r] > most) most = bytes[r];
    }
    int nb = c3d_cdiv((long long)(most / 16 + 1), 256 * 4);      // four 16-byte stores per thread
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_zero_views, dim3(nb, V, regions), dim3(256), 0, s, (char*)base0, vs, zr);
    C3D_LAUNCH_CHECK();
    return 0;
}
// bytes [0, min(*count, cap)) of p cleared, the count resident on the device: the "record written" bytes of a backward pass whose buffers were sized for a capacity
__global__ void __launch_bounds__(256) k_zero_count(uint4* __restrict__ p, const uint32_t* __restrict__ count, uint32_t cap) {
    const uint32_t c = *count < cap ? *count : cap;
    const size_t n16 = ((size_t)c + 15) / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
int c3d_zero_count(void* p, const uint32_t* count, uint32_t cap, hipStream_t s) {      // p: 16-byte aligned, room for cap rounded up to 16 bytes
    if (cap == 0) return 0;
    int nb = c3d_cdiv((long long)cap / 16 + 1, 256 * 4);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_zero_count, dim3(nb), dim3(256), 0, s, (uint4*)p, count, cap);
    C3D_LAUNCH_CHECK();
    return 0;
}
int c3d_scan_rect_gather(const uint2* rect, const uint32_t* idx, uint32_t* out, uint2* rsort, size_t n, void* tmp, hipStream_t s, bool zero_state,
                         uint32_t* tail_meta, uint32_t* tail_status, uint32_t tail_cap, uint32_t* err, int V, size_t vs, uint32_t tail_hint, unsigned long long* tail_early, bool rect4) {
    return scan_launch(reinterpret_cast<const uint32_t*>(rect), idx, out, n, false, tmp, s, zero_state, ScanTail{tail_meta, tail_status, tail_cap, rsort, tail_hint, tail_early, rect4 ? 1 : 0}, err, V, vs);
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
#ifndef RS_ITEMS
#define RS_ITEMS 16
#endif
#define RS_TILE (RS_THREADS * RS_ITEMS)
#define RS_RADIX 256
#define RS_MAX_PASSES C3D_SORT_MAX_PASSES
#define RS_FLAG_AGG (1u << 30)
#define RS_FLAG_INCL (2u << 30)
#define RS_VALUE_MASK ((1u << 30) - 1u)
#define RS_GROUP 16
#define RS_LOOKBACK 32
#define RS_SLOTS 1024             // workgroups of k_onesweep resident at once: 256 CUs x 4
#define RS_HIST_SPLIT C3D_SORT_HIST_SPLIT      // copies of the global histogram (workgroup b adds to copy b % 16): 245 - 1000 workgroups adding to ONE set of 256 counters serialise at the
                              // memory-side atomic unit (measured: 23 us for 1 M keys, profiles/r02c); consumers add the 16 copies up
__host__ __device__ static inline size_t sort_pass_words(size_t nb) { return RS_RADIX * (nb + nb / RS_GROUP + 1); }

// (Round 6, measured and dropped, profiles/r06/r06s_*: a workgroup counting 2-8 consecutive tiles before it adds its counters to the global copies -- a quarter of the memory-side
//  atomics -- made the depth sort of the 8-view step 0.198 -> 0.209 ms; what is kept of it is that a tile's 16 keys per lane are all requested before the first is counted:
//  one-view sort 0.0907 -> 0.0876 ms.  Also dropped, r06v_*: counting the upper digits -- two or three values per view -- by ballot instead of 64 lanes adding to two LDS counters:
//  0.202 -> 0.210 ms per 8 views; the LDS atomics are not what the kernel waits for either.)
__global__ void __launch_bounds__(RS_THREADS) k_radix_hist_all(const uint32_t* __restrict__ keys, uint32_t* __restrict__ ghist, size_t n,
                                                                const uint32_t* __restrict__ n_dev, int passes, size_t vs) {
    __shared__ uint32_t h[RS_MAX_PASSES][RS_RADIX];
    keys = c3d_view_ptr(keys, vs); ghist = c3d_view_ptr(ghist, vs); n_dev = c3d_view_ptr(n_dev, vs);      // (view = XCD as in k_onesweep: measured, no difference -- it only reads)
    const uint32_t htile = blockIdx.x;
    if (n_dev) n = min((size_t)*n_dev, n);      // element count resident on the device (no host round trip)
    const size_t base = (size_t)htile * RS_TILE;
    if (base >= n) return;                      // capacity-sized launch: nothing here
    for (int p = 0; p < passes; p++) h[p][threadIdx.x] = 0;
    const int lane = c3d_lane();
    uint32_t key[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const size_t idx = base + (size_t)i * RS_THREADS + threadIdx.x;
        key[i] = idx < n ? keys[idx] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const bool ok = base + (size_t)i * RS_THREADS + threadIdx.x < n;
        const uint32_t k = key[i];
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
    uint32_t* mine = ghist + (size_t)(htile % RS_HIST_SPLIT) * RS_MAX_PASSES * RS_RADIX;
    for (int p = 0; p < passes; p++) {
        const uint32_t c = h[p][threadIdx.x];
        if (c) atomicAdd(&mine[p * RS_RADIX + threadIdx.x], c);
    }
}

// poll one status word until its flag is at least `need` (1 = any, 2 = inclusive); bounded.  The spin itself is ONE out-of-line function: inlined at its 48 call sites (which the
// unrolled look-back batches multiply further) it was 432 copies of the loop and most of the kernel's 58 KB -- against a 64 KB instruction cache shared by two CUs whose
// workgroups sit in different phases of the tile.
__device__ __attribute__((noinline)) uint32_t rs_wait_slow(const uint32_t* p, uint32_t need, uint32_t* err) {
    uint32_t x, spins = 0;
    do {
        if (++spins > LB_SPIN_LIMIT) { atomicOr(err, C3D_ERR_LOOKBACK); return RS_FLAG_INCL; }
        __builtin_amdgcn_s_sleep(2);
        x = ld_agent32(p);
    } while ((x >> 30) < need);
    return x;
}
__device__ __forceinline__ uint32_t rs_wait(const uint32_t* p, uint32_t x, uint32_t need, uint32_t* err) {
    return ((x >> 30) < need) ? rs_wait_slow(p, need, err) : x;
}

// 16 keys per thread: 4096 keys and 39 KB of LDS per workgroup, four workgroups per CU.  Measured and dropped: 8 keys per thread (twice the tiles: the tile sort of an
// 8-view step 0.48 -> 0.82 ms, profiles/r04 -- a pass is bound by tiles in flight x tile lifetime, and the lifetime does not shrink with the tile); ONE 16 KB reorder
// buffer used for the keys and then for the values (23 KB: seven workgroups per CU -- but the kernel wants ~126 VGPRs, and capped at 72-96 it spills: 0.48 -> 0.8-1.06 ms).
// Round 4, second half, measured and dropped (profiles/r04u_*): a workgroup that KEEPS GOING -- draws the ticket of its next tile and requests that tile's keys before it ranks
// the current one, the recipe that took a third off the MS-SSIM kernels.  Tile sort of an 8-view step 0.48 -> 2.0 ms (4.7 ms where the extra registers spill).  A ticket drawn
// early is a tile whose aggregate is published LATE (after the workgroup's current tile, look-back wait included), and every higher ticket waits for it: the decoupled look-back
// lives on aggregates appearing at once, in parallel; holding tickets serialises it.
// Round 5 (profiles/r05q_*): ranking through v_bitop3_b32 and tiles without bounds tests (see `front` below): 126 -> 81 VGPRs, rank phase of a lone wave 3.3 -> 2.3 us,
// depth sort of an 8-view step 0.256 -> 0.237 ms, tile sort 0.488 -> 0.470 ms on one box.  With 81 VGPRs the ONE-reorder-buffer form (keys, then values, through one 16 KB buffer
// and back into registers before the chained scan: 22.5 KB) fits five and six workgroups per CU without spilling -- measured on the same box: 0.236 / 0.478 ms at five, 0.241 / 0.477
// at six.  Tiles in flight are not what bounds a pass (same finding as the lean kernel of profiles/r05ij); not kept.
// RANGES (the LAST pass of the tile sort): the pass also leaves the per-tile ranges of the sorted list -- what a separate kernel used to find by reading the sorted keys back (k_ranges:
// one more launch on the chain and 16 MB per view).  Equal keys sit next to each other in the workgroup's reorder buffer and go to consecutive output positions, so an element whose
// LDS neighbour holds another key (or none) is the first / last of ITS workgroup's stretch of that key: it grows the tile's {~start, end} words by atomicMax from the cleared state
// {0, 0} -- a few dozen atomics per 4096 keys (the input of the last pass is ordered by the low digit: a workgroup's keys are a few image tiles).
template <bool IOTA, int ITEMS, int STAY, bool RANGES = false>
__global__ void __launch_bounds__(RS_THREADS, 4) k_onesweep(const uint32_t* __restrict__ keys_in0, const uint32_t* __restrict__ vals_in0,
                                                             uint32_t* __restrict__ keys_out0, uint32_t* __restrict__ vals_out0,
                                                             const uint32_t* __restrict__ ghist0, uint32_t* __restrict__ ticket, uint32_t* __restrict__ err,
                                                             uint32_t* __restrict__ tile_words0, uint32_t* __restrict__ group_words0, size_t n_cap,
                                                             const uint32_t* __restrict__ n_dev, int shift, unsigned long long* __restrict__ dbg, size_t vs,
                                                             uint2* __restrict__ ranges0) {
    __shared__ uint32_t whist[RS_THREADS / 64][RS_RADIX];
    __shared__ uint32_t lstart[RS_RADIX];    // first local slot of each digit
    __shared__ uint32_t gbase[RS_RADIX];     // global position of that slot
    __shared__ uint32_t skey[RS_THREADS * ITEMS];
    __shared__ uint32_t sval[RS_THREADS * ITEMS];
    __shared__ uint32_t scan_lds[4];
    __shared__ uint32_t s_tile, s_view;
    // Which view a workgroup of a V-view launch serves (round 5, profiles/r05r_* ... r05y_*): its HOME view is (linear block id) % V, not blockIdx.y.  Block b runs on XCD b % 8, so
    // with V = 8 each view has ONE XCD (V = 16: two views per XCD; V = 4: two XCDs per view), and the half-line digit runs that neighbouring tiles of a view write next to each
    // other meet in one L2 instead of leaving eight L2s as partial lines: tile sort of the 8-view step 0.460 -> 0.371 ms, depth sort 0.232 -> 0.218 ms on one box.
    // A workgroup STAYS (round 5): it draws a ticket of its view, sorts that tile, draws the next -- the launch is one workgroup per residency slot, not one per tile of the
    // pair CAPACITY.  (With a grid sized for the capacity, half of the workgroups of the 8-view step found no tile: 977 same-address ticket atomics per view and pass just to
    // learn that, 12 ns each.)  When its view has no tile left it looks at the other views' counters ONCE (lane v of wave 0 reads view v's) and carries on with the first that
    // has: work stealing, without which a pass of unequal views would last as long as its largest view on one XCD.  Only after the PREVIOUS tile is finished is the next ticket
    // drawn -- a ticket held early is an aggregate published late (see above) -- and whoever holds a ticket is running: every wait of the chained scan is on a resident workgroup.
    const uint32_t V = gridDim.y;
    uint32_t view = (blockIdx.x + gridDim.x * blockIdx.y) % V;
    const int lane = c3d_lane(), wave = threadIdx.x >> 6;
    uint32_t gview = 0xFFFFFFFFu, gtotal = 0, digit_base = 0;      // thread d: occurrences of digit d in the view's input and where the digit starts in the output -- kept while the view stays
#define RS_AT(p_, v_) ((decltype(p_))((char*)(p_) + (size_t)(v_) * vs))
#define RS_COUNT(v_) (n_dev ? min((size_t)*RS_AT(n_dev, v_), n_cap) : n_cap)      /* element count resident on the device (no host round trip) */
    // STAY: the grid does not cover the capacity, tiles are drawn until none is left.  A launch of at most two rounds (2 x RS_SLOTS tiles: the depth sort of <= 8 views) runs the
    // instance without the loop, one workgroup per tile -- there a staying workgroup's last, failed draw is a measurable tail (0.205 -> 0.224 ms for the four depth passes
    // of the 8-view step), and nobody steals: a workgroup without a tile leaves.
  for (;;) {
    size_t n = RS_COUNT(view);               // requested BEFORE the ticket (behind the barrier it is a dependent scalar load on every tile's critical path)
    if (STAY == 1) __syncthreads();          // the previous tile's scatter has read skey / sval / gbase / lstart
    if (threadIdx.x == 0) s_tile = atomicAdd(RS_AT(ticket, view), 1u);
    for (int i = threadIdx.x; i < (RS_THREADS / 64) * RS_RADIX; i += RS_THREADS) (&whist[0][0])[i] = 0;
    __syncthreads();
    uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tile);      // uniform by construction; in a scalar register the tile's base pointers are scalar too
    while ((size_t)tile * (RS_THREADS * ITEMS) >= n) {      // this view has no tile left
        if (!STAY || V == 1) return;
        // one look at the other views' counters (lane v of wave 0 reads view v's), then a ticket of the first view that has tiles.  Only now: the counters are the words the
        // ticket atomics serialise on, and a look per tile (tried: next to every draw, so that a failed draw would know at once where to go) queues behind them -- tile sort
        // 0.377 -> 0.57 ms (profiles/r05zz_*)
        __syncthreads();                     // everybody has read s_tile
        if (threadIdx.x < 64) {
            bool has = false;
            if ((uint32_t)lane < V && (uint32_t)lane != view) has = (size_t)ld_agent32(RS_AT(ticket, lane)) * (RS_THREADS * ITEMS) < RS_COUNT(lane);
            const uint64_t m = __ballot(has);
            if (lane == 0) {
                const uint64_t above = (view + 1 < 64) ? (m >> (view + 1)) : 0ull;      // the next view upwards first, so that thieves spread over the views
                const uint32_t pick = !m ? 0xFFFFFFFFu : (above ? view + 1 + (uint32_t)__builtin_ctzll(above) : (uint32_t)__builtin_ctzll(m));
                s_view = pick;
                if (m) s_tile = atomicAdd(RS_AT(ticket, pick), 1u);
            }
        }
        __syncthreads();
        if (s_view == 0xFFFFFFFFu) return;   // every view is done
        view = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_view);
        tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tile);
        n = RS_COUNT(view);
    }
    const uint32_t* keys_in = RS_AT(keys_in0, view); const uint32_t* vals_in = IOTA ? vals_in0 : RS_AT(vals_in0, view);
    uint32_t* keys_out = RS_AT(keys_out0, view); uint32_t* vals_out = RS_AT(vals_out0, view);
    const uint32_t* ghist = RS_AT(ghist0, view); uint32_t* tile_words = RS_AT(tile_words0, view); uint32_t* group_words = RS_AT(group_words0, view);
    const size_t bbase = (size_t)tile * (RS_THREADS * ITEMS);
#define RS_STAMP(k) do { if (dbg && threadIdx.x == 0) dbg[(size_t)tile * 8 + (k)] = (unsigned long long)wall_clock64(); } while (0)   // profiling hook (profiles/microbench/sort_phases.py)
    RS_STAMP(0);
    // Everything below addresses the tile with 32-bit offsets from scalar base pointers.  `full` (all tiles of a pass but the last) is a uniform branch around
    // two instances of the same code: the full one carries no bounds test at all (round 5: 16 exec-masked load blocks with 64-bit compares, a ballot and a
    // select per item gone).
    const uint32_t cnt = (uint32_t)((n - bbase) < (size_t)(RS_THREADS * ITEMS) ? (n - bbase) : (size_t)(RS_THREADS * ITEMS));      // keys in this tile
    const uint32_t* kin = keys_in + bbase;
    const uint32_t* vin = vals_in + bbase;
    const uint32_t woff = (uint32_t)wave * (RS_THREADS * ITEMS / 4) + (uint32_t)lane;      // the lane's first key: wave w owns keys [1024 w, 1024 w + 1024) of the tile
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
    auto front = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const uint32_t off = woff + (uint32_t)i * 64u;
            const bool ok = FULL || off < cnt;
            const uint32_t offc = FULL ? off : (ok ? off : 0u);      // a clamped address and a select instead of a branch per load
            const uint32_t k = kin[offc];
            key[i] = ok ? k : 0xFFFFFFFFu;
            if (IOTA) val[i] = (uint32_t)bbase + off;
            else { const uint32_t v = vin[offc]; val[i] = ok ? v : 0u; }
        }
        if (dbg) { uint32_t x = 0; for (int i = 0; i < ITEMS; i++) x ^= key[i] ^ val[i]; if (x == 0x12345u) dbg[7] = 1; }   // wait for the loads before stamping
        RS_STAMP(1);
        // ranking: the lanes holding the same digit, found bit by bit.  Per bit one sign-extending field extract (t = 0 / ~0), one ballot, and per mask half ONE
        // v_bitop3_b32 (gfx950): peers & ~(ballot ^ t) -- keep the lanes whose bit equals mine (0x90 = a & ~(b ^ c)).  The generic spelling
        // (peers &= bit ? m : ~m) compiled to 10 VALU instructions per bit, this is 5; a lone wave issues one VALU instruction per ~4 cycles, so the
        // 16 x 8 bits were most of the phase.  The rank among the peers is mbcnt (bits of the mask below my lane), no lane mask needed.
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const bool ok = FULL || (woff + (uint32_t)i * 64u) < cnt;
            const uint32_t ds = key[i] >> shift;
            uint32_t plo = ~0u, phi = ~0u;
            if (!FULL) { const uint64_t okm = __ballot(ok); plo = (uint32_t)okm; phi = (uint32_t)(okm >> 32); }
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const int t = __builtin_amdgcn_sbfe((int)ds, b, 1);
                const uint64_t m = __ballot(t != 0);
                plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)m, (uint32_t)t, 0x90);
                phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(m >> 32), (uint32_t)t, 0x90);
            }
            const uint32_t d = ds & (RS_RADIX - 1);
            const uint32_t prefix = whist[wave][d];
            const uint32_t r = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            if (ok && r == 0) whist[wave][d] = prefix + (uint32_t)__popc(plo) + (uint32_t)__popc(phi);  // lowest peer lane updates
            rank[i] = prefix + r;
        }
    };
    const bool full = cnt == (uint32_t)(RS_THREADS * ITEMS);
    if (full) front(std::true_type{}); else front(std::false_type{});
    __syncthreads();
    RS_STAMP(2);
    // thread d owns digit d: count over the 4 waves, published at once (successors can already add it up), then the block-local layout
    const int d = threadIdx.x;
    const uint32_t grp = tile / RS_GROUP, gr = tile % RS_GROUP;
    const bool leader = gr == RS_GROUP - 1;
    const bool new_view = gview != view;                 // (uniform) the digit totals of the view's input and the digit starts: once per view a workgroup serves
    if (new_view) {
        gtotal = 0;
#pragma unroll
        for (int c = 0; c < RS_HIST_SPLIT; c++) gtotal += ghist[(size_t)c * RS_MAX_PASSES * RS_RADIX + d];
    }
    uint32_t tot = 0;
    {
        uint32_t c[RS_THREADS / 64];
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; w++) { c[w] = whist[w][d]; tot += c[w]; }
        if (gtotal) st_agent32(tile_words + (size_t)tile * RS_RADIX + d, RS_FLAG_AGG | tot);
        uint32_t blk_total, dummy;
        uint32_t ls = block_excl_scan(tot, scan_lds, &blk_total);
        if (new_view) { digit_base = block_excl_scan(gtotal, scan_lds, &dummy); gview = view; }   // where digit d starts in the output
        lstart[d] = ls;
        gbase[d] = digit_base;
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; w++) { whist[w][d] = ls; ls += c[w]; }
    }
    __syncthreads();
    // block-local reorder first: it needs no global prefix and frees the key / val / rank registers for the scan's loads
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        if (full || (woff + (uint32_t)i * 64u) < cnt) {
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
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const uint32_t lp = (uint32_t)i * RS_THREADS + threadIdx.x;     // consecutive lanes -> consecutive slots of a digit run
        if (full || lp < cnt) {
            const uint32_t k = skey[lp], v = sval[lp];
            const uint32_t dd = (k >> shift) & (RS_RADIX - 1);
            const uint32_t pos = gbase[dd] + ((uint32_t)lp - lstart[dd]);
            if (!RANGES) keys_out[pos] = k;      // the ranges are what the sorted keys were read for: the last pass of a tile sort leaves no keys (4 of its 16 bytes per pair)
            vals_out[pos] = v;
            if (RANGES) {      // a slot whose predecessor holds another key starts its key's stretch here -- and ends the predecessor's: both words from the one compare
                const uint32_t kp = lp ? skey[lp - 1u] : ~k;
                if (kp != k) {
                    atomicMax(&(RS_AT(ranges0, view) + k)->x, ~pos);
                    if (lp) {
                        const uint32_t dp = (kp >> shift) & (RS_RADIX - 1);
                        atomicMax(&(RS_AT(ranges0, view) + kp)->y, gbase[dp] + ((uint32_t)lp - 1u - lstart[dp]) + 1u);
                    }
                }
                if (lp + 1u == cnt) atomicMax(&(RS_AT(ranges0, view) + k)->y, pos + 1u);
            }
        }
    }
    RS_STAMP(5);
    if (STAY == 0) return;
    // STAY == 2 (one view; c3d_gs_forward_nosync): the grid was sized for a HINT of the element count.  A count the grid covers -- the usual case -- is one tile per workgroup and
    // no failed draw, exactly the STAY == 0 launch; a larger count makes the workgroups stay and draw on, so the pass is correct for every count the buffers hold.
    if (STAY == 2) { if ((size_t)gridDim.x * (RS_THREADS * ITEMS) >= n) return; __syncthreads(); }      // (the barrier STAY == 1 has at the head of the loop)
  }
#undef RS_STAMP
#undef RS_AT
#undef RS_COUNT
}

// ------------------------------------------------------------------------------------------
// Small sorts (round 6): up to 16384 keys per view, values = the element indices.  The node's own default scene (10 000 Gaussians, /root/reference/nodes.py:1175-1198) made the
// depth sort a histogram launch and four onesweep launches of THREE tiles each -- 55 us of a 410 us training iteration for 80 KB of keys, every launch the minimum a launch costs.
// Here ONE workgroup per view (16 waves) does all passes: a lane keeps its 16 (key, id) pairs in registers, ranks them by ballot as k_onesweep does, and the whole array changes
// places through LDS (128 KB of the CU's 160) between passes -- no state in memory, no clear, no chained scan, no histogram kernel; keys and ids reach HBM once, sorted.
// ------------------------------------------------------------------------------------------
#define SS_THREADS 1024
#define SS_ITEMS 16
#define SS_MAX (SS_THREADS * SS_ITEMS)
__global__ void __launch_bounds__(SS_THREADS) k_sort_small(const uint32_t* __restrict__ keys_in0, uint32_t* __restrict__ keys_out0, uint32_t* __restrict__ vals_out0, uint32_t n,
                                                            int passes, size_t vs) {
    __shared__ uint32_t skey[SS_MAX];
    __shared__ uint32_t sval[SS_MAX];
    __shared__ uint32_t whist[SS_THREADS / 64][RS_RADIX];      // per wave and digit: count, then the position of the wave's first key of that digit
    __shared__ uint32_t scan_lds[4];
    const uint32_t* keys_in = c3d_view_ptr(keys_in0, vs);
    uint32_t* keys_out = c3d_view_ptr(keys_out0, vs);
    uint32_t* vals_out = c3d_view_ptr(vals_out0, vs);
    const int lane = c3d_lane(), wave = threadIdx.x >> 6;
    // every wave owns the same number of 64-element rows, R = ceil(n / 1024) <= 16, of a contiguous stretch of the array (ranking is stable by construction): 10 000 keys are
    // 10 rows in each of the 16 waves, not 16 rows in ten of them
    const int R = (int)((n + SS_THREADS - 1) / SS_THREADS);
    const uint32_t woff = (uint32_t)wave * (uint32_t)(64 * R) + (uint32_t)lane;
    uint32_t key[SS_ITEMS], val[SS_ITEMS], rank[SS_ITEMS];
#pragma unroll
    for (int i = 0; i < SS_ITEMS; i++) {
        const uint32_t idx = woff + (uint32_t)i * 64u;
        const bool ok = i < R && idx < n;
        key[i] = ok ? keys_in[idx] : 0xFFFFFFFFu;
        val[i] = idx;
    }
    // The digits are taken from key - (smallest key) -- the same order, ties included -- with the keys 0xFFFFFFFF (culled Gaussians; elements past the end) mapped to the value behind
    // the largest other key: depth keys of one view span a factor < 4 in depth, 24 bits of the float pattern, so the fourth pass has nothing to move and is not run (a uniform
    // trip count: everything lives in this workgroup).  The keys that are written are the original ones.
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int i = 0; i < SS_ITEMS; i++)
        if (key[i] != 0xFFFFFFFFu) { kmin = min(kmin, key[i]); kmax = max(kmax, key[i]); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64)); kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64)); }
    if (lane == 0) { whist[0][wave] = kmin; whist[1][wave] = kmax; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < SS_THREADS / 64; w++) { kmin = min(kmin, whist[0][w]); kmax = max(kmax, whist[1][w]); }
    __syncthreads();                                     // whist is cleared by the first pass
    const uint32_t top = kmin <= kmax ? (kmax - kmin) + 1u : 0u;      // what a culled key maps to (no other key at all: everything maps to 0)
    if (kmin > kmax) kmin = 0xFFFFFFFFu;
    {
        const int bits = 32 - __builtin_clz(top | 1u);
        passes = min(passes, max(1, (bits + 7) >> 3));
    }
    for (int pass = 0; pass < passes; pass++) {
        const int shift = 8 * pass;
        for (int i = threadIdx.x; i < (SS_THREADS / 64) * RS_RADIX; i += SS_THREADS) (&whist[0][0])[i] = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SS_ITEMS; i++) {
            if (i < R) {      // (uniform)
                const bool ok = woff + (uint32_t)i * 64u < n;      // elements past the end take no part: n elements are ranked, n positions are filled
                const uint64_t okm = __ballot(ok);
                const uint32_t ds = (key[i] == 0xFFFFFFFFu ? top : key[i] - kmin) >> shift;
                uint32_t plo = (uint32_t)okm, phi = (uint32_t)(okm >> 32);
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    const int t = __builtin_amdgcn_sbfe((int)ds, b, 1);
                    const uint64_t m = __ballot(t != 0);
                    plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)m, (uint32_t)t, 0x90);
                    phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(m >> 32), (uint32_t)t, 0x90);
                }
                const uint32_t d = ds & (RS_RADIX - 1);
                const uint32_t prefix = whist[wave][d];
                const uint32_t r = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
                if (ok && r == 0) whist[wave][d] = prefix + (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
                rank[i] = prefix + r;
            }
        }
        __syncthreads();
        // thread d < 256: where digit d starts (exclusive scan of the digit totals over waves 0-3), then where each wave's keys of that digit start
        uint32_t c[SS_THREADS / 64], tot = 0, incl = 0;
        if (threadIdx.x < RS_RADIX) {
#pragma unroll
            for (int w = 0; w < SS_THREADS / 64; w++) { c[w] = whist[w][threadIdx.x]; tot += c[w]; }
            incl = c3d_wave_incl_scan(tot);
            if (lane == 63) scan_lds[wave] = incl;
        }
        __syncthreads();
        if (threadIdx.x < RS_RADIX) {
            uint32_t base = incl - tot;
            for (int w = 0; w < wave; w++) base += scan_lds[w];
#pragma unroll
            for (int w = 0; w < SS_THREADS / 64; w++) { whist[w][threadIdx.x] = base; base += c[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SS_ITEMS; i++) {
            if (i < R && woff + (uint32_t)i * 64u < n) {
                const uint32_t pos = whist[wave][((key[i] == 0xFFFFFFFFu ? top : key[i] - kmin) >> shift) & (RS_RADIX - 1)] + rank[i];
                skey[pos] = key[i];
                sval[pos] = val[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SS_ITEMS; i++) {
            const uint32_t idx = woff + (uint32_t)i * 64u;
            if (i < R && idx < n) { key[i] = skey[idx]; val[i] = sval[idx]; }
        }
    }
#pragma unroll
    for (int i = 0; i < SS_ITEMS; i++) {
        const uint32_t idx = woff + (uint32_t)i * 64u;
        if (i < R && idx < n) { keys_out[idx] = key[i]; vals_out[idx] = val[i]; }
    }
}

static unsigned long long* g_sort_dbg = nullptr;     // profiling hook: [pass][tile][8] wall_clock64 stamps (100 MHz), see c3d_test_sort_phases
static inline size_t sort_head_bytes() { return c3d_align(sizeof(uint32_t) * (RS_HIST_SPLIT * RS_RADIX * RS_MAX_PASSES + RS_MAX_PASSES + 4)); }
#define RS_MIN_TILE RS_TILE
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

// The state of a sort whose layout holds n elements, cleared for the min(*n_dev, n) elements that are really there: the fixed head (histograms, tickets) and, per pass, the
// tile words of the tiles in use and the group words of their groups -- what a launch sized for a hint of the count needs, whatever the buffers hold (a memset of the whole
// state is 1 KB per 4096 elements of CAPACITY and pass).  `pre`: bytes in front of tmp cleared by the same launch (the tile ranges + meta words of the binning state; 16-byte
// aligned, a multiple of 16).
__global__ void __launch_bounds__(256) k_sort_zero_state(uint4* __restrict__ pre, size_t pre16, uint32_t* __restrict__ status, size_t nb, int passes, size_t n_cap,
                                                          const uint32_t* __restrict__ n_dev) {
    const size_t n = min((size_t)*n_dev, n_cap);
    const size_t tiles = (n + RS_TILE - 1) / RS_TILE, groups = tiles / RS_GROUP + 1;
    const int seg = blockIdx.y;
    uint4* p; size_t cnt;
    if (seg == 0) { p = pre; cnt = pre16; }
    else {
        uint32_t* tw = status + (size_t)((seg - 1) >> 1) * sort_pass_words(nb);
        if ((seg - 1) & 1) { p = reinterpret_cast<uint4*>(tw + (size_t)RS_RADIX * nb); cnt = groups * (RS_RADIX / 4); }
        else { p = reinterpret_cast<uint4*>(tw); cnt = tiles * (RS_RADIX / 4); }
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
int c3d_sort_zero_state_counted(void* tmp, size_t n, int end_bit, const uint32_t* n_dev, size_t n_hint, void* pre, size_t pre_bytes, hipStream_t s) {
    int passes = (end_bit + 7) / 8;
    if (passes < 1) passes = 1;
    if (passes > RS_MAX_PASSES || !n_dev || ((uintptr_t)pre & 15) || (pre_bytes & 15) || (char*)pre + pre_bytes != (char*)tmp) {
        c3d_set_error("c3d_sort_zero_state_counted: bad argument (pre must end where tmp starts, 16-byte granules)"); return -1; }
    const size_t nb = (size_t)c3d_cdiv((long long)(n ? n : 1), RS_TILE);
    const size_t nb_hint = (size_t)c3d_cdiv((long long)(n_hint ? n_hint : n), RS_TILE);
    int gx = (int)((nb_hint * (RS_RADIX / 4) + 1023) / 1024);      // four 16-byte stores per thread at the hinted count; a larger count loops
    if (gx < 16) gx = 16;
    if (gx > 1024) gx = 1024;
    // segment 0 = pre + the sort's head (contiguous), then tile words / group words of every pass
    hipLaunchKernelGGL(k_sort_zero_state, dim3(gx, 1 + 2 * passes), dim3(256), 0, s, (uint4*)pre, (pre_bytes + sort_head_bytes()) / 16, (uint32_t*)((char*)tmp + sort_head_bytes()), nb, passes, n, n_dev);
    C3D_LAUNCH_CHECK();
    return 0;
}

int c3d_sort_pairs_u32(uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1, bool iota_vals,
                       size_t n, int end_bit, void* tmp, int* result_buf, hipStream_t s, const uint32_t* n_dev, bool zero_state, uint32_t* err_out, int V, size_t vs, bool hist_done,
                       size_t n_hint, uint2* ranges) {
    *result_buf = 0;
    if (n == 0 || V <= 0) return 0;
    if (zero_state && V != 1) { c3d_set_error("c3d_sort_pairs_u32: a multi-view launch clears its state through c3d_zero_views"); return -1; }
    if (end_bit > 8 * RS_MAX_PASSES) { c3d_set_error("c3d_sort_pairs_u32: end_bit %d > %d", end_bit, 8 * RS_MAX_PASSES); return -1; }
    if (ranges && iota_vals && end_bit <= 8) { c3d_set_error("c3d_sort_pairs_u32: ranges with a one-pass iota sort"); return -1; }
    if (n > (size_t)RS_VALUE_MASK) { c3d_set_error("c3d_sort_pairs_u32: %zu elements exceed the 2^30 - 1 the chained scan's status words hold", n); return -1; }
    int passes = (end_bit + 7) / 8;
    if (passes < 1) passes = 1;
    if (iota_vals && n <= (size_t)SS_MAX && !n_dev && !hist_done && !ranges && !g_sort_dbg) {      // a small sort: one workgroup per view, all passes in one launch, no state (k_sort_small)
        hipLaunchKernelGGL(k_sort_small, dim3(1, V), dim3(SS_THREADS), 0, s, (const uint32_t*)keys0, (passes & 1) ? keys1 : keys0, (passes & 1) ? vals1 : vals0, (uint32_t)n, passes, vs);
        C3D_LAUNCH_CHECK();
        *result_buf = passes & 1;
        return 0;
    }
    const int nb = c3d_cdiv((long long)n, RS_TILE), nb_hist = nb;
    // k_onesweep's workgroups stay and draw tile after tile: one per residency slot of the chip (MI355X: 256 CUs x 4 workgroups of 39 KB LDS), spread evenly over the views
    // n_hint (one view, with n_dev): the launch is sized for n_hint elements although the buffers (and the state layout) hold n -- workgroups loop if the count exceeds the hint (STAY == 2)
    const bool hinted = n_hint > 0 && n_hint < n && V == 1 && n_dev != nullptr;
    const int nb_launch = hinted ? c3d_cdiv((long long)n_hint, RS_TILE) : nb;
    const bool stay = (long long)nb_launch * V > 2 * RS_SLOTS;       // (at most two rounds: one workgroup per tile, see STAY)
    const int nbx = !stay ? nb_launch : (RS_SLOTS / V > 0 ? RS_SLOTS / V : 1);
    const int mode = stay ? 1 : (hinted ? 2 : 0);
    uint32_t* ghist = (uint32_t*)tmp;
    uint32_t* tickets = ghist + RS_HIST_SPLIT * RS_RADIX * RS_MAX_PASSES;
    uint32_t* err = err_out ? err_out : c3d_sort_error_word(tmp);
    uint32_t* status = (uint32_t*)((char*)tmp + sort_head_bytes());
    if (hist_done && zero_state) { c3d_set_error("c3d_sort_pairs_u32: hist_done with zero_state would clear the producer's histograms"); return -1; }
    if (zero_state) C3D_CHECK(hipMemsetAsync(tmp, 0, c3d_sort_state_bytes(n, end_bit), s));
    uint32_t* k[2] = {keys0, keys1};
    uint32_t* v[2] = {vals0, vals1};
    if (!hist_done) hipLaunchKernelGGL(k_radix_hist_all, dim3(nb_hist, V), dim3(RS_THREADS), 0, s, keys0, ghist, n, n_dev, passes, vs);
    int cur = 0;
    for (int pass = 0; pass < passes; pass++) {
        uint32_t* tw = status + (size_t)pass * sort_pass_words((size_t)nb);
        uint32_t* gw = tw + (size_t)RS_RADIX * nb;
#define RS_SWEEP(IOTA_, STAY_, RNG_) hipLaunchKernelGGL((k_onesweep<IOTA_, RS_ITEMS, STAY_, RNG_>), dim3(nbx, V), dim3(RS_THREADS), 0, s, k[cur], v[cur], k[cur ^ 1], v[cur ^ 1], ghist + pass * RS_RADIX, \
                                           tickets + pass, err, tw, gw, n, n_dev, 8 * pass, (g_sort_dbg && V == 1) ? g_sort_dbg + (size_t)pass * nb * 8 : nullptr, vs, ranges)
        if (pass == 0 && iota_vals) { if (mode == 1) RS_SWEEP(true, 1, false); else if (mode == 2) RS_SWEEP(true, 2, false); else RS_SWEEP(true, 0, false); }
        else if (ranges && pass == passes - 1) { if (mode == 1) RS_SWEEP(false, 1, true); else if (mode == 2) RS_SWEEP(false, 2, true); else RS_SWEEP(false, 0, true); }
        else { if (mode == 1) RS_SWEEP(false, 1, false); else if (mode == 2) RS_SWEEP(false, 2, false); else RS_SWEEP(false, 0, false); }
#undef RS_SWEEP
        C3D_LAUNCH_CHECK();
        cur ^= 1;
    }
    *result_buf = cur;
    return 0;
}

// test / profiling hook: one sort with per-tile phase stamps -> stamps[passes][tiles][8] (device), wall_clock64 ticks
int c3d_sort_set_debug(unsigned long long* stamps) { g_sort_dbg = stamps; return 0; }
