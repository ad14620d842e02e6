// scan_wave.h -- the record-base scan (A2) as a ONE-WAVE-per-tile chained scan that rides in another kernel's launch (round 5).
//
// The exclusive scan of `tiles` in Gaussian-id order (-> rbase, einfo) feeds the BACKWARD pass only; as a launch of its own (k_scan_lb<true, false>, 0.125 ms per
// 8-view step) it sat on the forward chain's critical path for nothing.  Its tiles now occupy the first blocks of the recording forward compositing launch: 64-thread
// workgroups like the compositing waves, one wave = one 1024-element tile, dispatched ahead of each view's quadrant waves.  Same inter-workgroup rules as scan_sort.hip
// (ticket counter, one {flag, value} word per published quantity, relaxed agent-scope atomics, bounded spins).  What it buys, measured on one box in alternation
// (profiles/r05f_scan_in_composite_ab.txt): the compositing launch 0.925 -> 1.012 ms, the scan launch gone: 8-view step 5.134 -> 5.075 ms (1.2 %).
#pragma once
#include "c3d_common.h"

#define SCANW_TILE 1024            // elements per wave: 4 sub-tiles of 256, lane t owns elements [4t, 4t + 4) of each (16-byte loads / stores on consecutive addresses)
#define SCANW_GROUP 64             // tiles per group of the two-level hand-over
#define SCANW_SPIN_LIMIT (1u << 21)
// Hand-over between tiles: NOT the windowed decoupled look-back of scan_sort.hip.  With ~1000 one-wave tiles per view all resident at once, walking 64 predecessors per
// dependent step cost a late tile ~15 round trips through the memory-side cache (measured: the fused scan added its whole stand-alone time, 0.1 ms per 8-view step, to
// the compositing launch, profiles/r05e_*).  Here every load a tile needs is independent of the others' VALUES: tile t adds up the aggregates of the tiles before it in
// its group of 64 (one load per lane) and the totals of all earlier groups (one load per lane per 64 groups; a group's total is published by its last tile from the
// same in-group sum -- no chain).  Two hand-overs deep whatever the tile count.  Every wait is on a tile with a lower ticket (already running) and bounded.
// state block: [0] ticket, [1] unused, then one 64-bit word per tile (flag << 32 | aggregate) and one per group (flag << 32 | total); zero before the launch
static inline size_t scan_wave_tiles(size_t n) { return ((n ? n : 1) + SCANW_TILE - 1) / SCANW_TILE; }
static inline size_t scan_wave_tmp_bytes(size_t n) { return c3d_align(8 + sizeof(unsigned long long) * (scan_wave_tiles(n) + scan_wave_tiles(n) / SCANW_GROUP + 2)); }
static inline int scan_wave_blocks(size_t n) { return (int)((scan_wave_tiles(n) + 31) / 32 * 32); }      // a multiple of 32: the compositing blocks behind keep their XCD / quadrant mapping

#ifdef __HIPCC__
struct ScanWaveJob {      // all pointers: view 0's (view v lies v * vs bytes behind); blocks == 0: no job
    const uint32_t* in; uint32_t* out; const uint2* rect; uint4* einfo; uint32_t* state; uint32_t* err; uint32_t n; int blocks;
    int rect4;      // `rect` holds 4-byte packed rects (c3d_rect_pack): the element count is the rect's area (`in` is not read) and einfo is EIGHT bytes per element,
                    // {packed rect, out[i]} -- half the scan's traffic (16 instead of 32 bytes per element), and half the table the backward compositing gathers from
};
// value of a published word (polls until its flag is set; bounded)
__device__ __forceinline__ uint32_t scan_wave_wait(const unsigned long long* p, uint32_t* err) {
    unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    while ((uint32_t)(w >> 32) == 0u) {
        if (++spins > SCANW_SPIN_LIMIT) { atomicOr(err, C3D_ERR_LOOKBACK); return 0u; }
        __builtin_amdgcn_s_sleep(2);
        w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return (uint32_t)w;
}
// out[i] = exclusive prefix of in[0..i); einfo[i] = {0, rect[i].x, rect[i].y, out[i]} where in[i] != 0 (rect4: see ScanWaveJob).  Called by a whole 64-lane workgroup.
__device__ __forceinline__ void scan_wave_tile(const ScanWaveJob& j, size_t vs) {
    const uint32_t* __restrict__ in = c3d_view_ptr(j.in, vs);
    uint32_t* __restrict__ out = c3d_view_ptr(j.out, vs);
    const uint2* __restrict__ rect = c3d_view_ptr(j.rect, vs);
    uint4* __restrict__ einfo = c3d_view_ptr(j.einfo, vs);
    uint32_t* state = c3d_view_ptr(j.state, vs);
    const int lane = (int)threadIdx.x;
    const size_t n = j.n;
    const uint32_t tiles = (uint32_t)((n + SCANW_TILE - 1) / SCANW_TILE);
    unsigned long long* tile_words = reinterpret_cast<unsigned long long*>(state + 2);
    unsigned long long* group_words = tile_words + tiles;
    uint32_t tk = 0;
    if (lane == 0) tk = atomicAdd(&state[0], 1u);
    const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
    if (tile >= tiles) return;
    uint32_t v[4][4], hs[4];
    const uint32_t* __restrict__ rect4 = reinterpret_cast<const uint32_t*>(rect);
    uint32_t rq[4][4];      // rect4: the packed rects of the lane's elements (the counts come out of them, and they go into einfo)
    auto area4 = [](uint32_t r) { return (((r >> 16) & 0xFFu) - (r & 0xFFu)) * ((r >> 24) - ((r >> 8) & 0xFFu)); };
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const size_t b = (size_t)tile * SCANW_TILE + (size_t)h * 256 + (size_t)lane * 4;
        if (j.rect4) {
            if (b + 3 < n) { const uint4 q = *reinterpret_cast<const uint4*>(rect4 + b); rq[h][0] = q.x; rq[h][1] = q.y; rq[h][2] = q.z; rq[h][3] = q.w; }
            else {
#pragma unroll
                for (int i = 0; i < 4; i++) rq[h][i] = (b + i < n) ? rect4[b + i] : 0u;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) v[h][i] = area4(rq[h][i]);
        } else if (b + 3 < n) { const uint4 q = *reinterpret_cast<const uint4*>(in + b); v[h][0] = q.x; v[h][1] = q.y; v[h][2] = q.z; v[h][3] = q.w; }
        else {
#pragma unroll
            for (int i = 0; i < 4; i++) v[h][i] = (b + i < n) ? in[b + i] : 0u;
        }
        hs[h] = v[h][0] + v[h][1] + v[h][2] + v[h][3];
    }
    uint32_t ex[4], tot = 0;
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const uint32_t inc = c3d_wave_incl_scan(hs[h]);
        ex[h] = tot + inc - hs[h];                                                   // a sub-tile follows all of the sub-tiles before it
        tot += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    if (lane == 0) __hip_atomic_store(&tile_words[tile], (1ull << 32) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t grp = tile / SCANW_GROUP, r = tile % SCANW_GROUP;
    // (1) the tiles before this one in its group: one word per lane, all loads in flight together
    uint32_t in_sum = ((uint32_t)lane < r) ? scan_wave_wait(&tile_words[(size_t)grp * SCANW_GROUP + lane], j.err) : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) in_sum += __shfl_xor(in_sum, o, 64);
    // the last tile of a group (with successors) publishes the group's total straight away: it depends on the group's own tiles only, not on earlier groups -- no chain
    if (r == SCANW_GROUP - 1 && tile + 1 < tiles && lane == 0)
        __hip_atomic_store(&group_words[grp], (1ull << 32) | (uint32_t)(in_sum + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (2) the totals of the earlier groups, 64 per step (independent loads as well)
    uint32_t g_sum = 0;
    for (uint32_t g0 = 0; g0 < grp; g0 += 64)
        if (g0 + (uint32_t)lane < grp) g_sum += scan_wave_wait(&group_words[g0 + lane], j.err);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) g_sum += __shfl_xor(g_sum, o, 64);
    const uint32_t prefix = in_sum + g_sum;
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const size_t b = (size_t)tile * SCANW_TILE + (size_t)h * 256 + (size_t)lane * 4;
        uint32_t run = prefix + ex[h], e[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { e[i] = run; run += v[h][i]; }
        if (j.rect4) {
            uint2* __restrict__ e8 = reinterpret_cast<uint2*>(einfo);
            if (b + 3 < n) {
                *reinterpret_cast<uint4*>(out + b) = make_uint4(e[0], e[1], e[2], e[3]);
                // (entries of culled elements are written too: two 16-byte stores per lane instead of four predicated 8-byte ones; nobody reads them)
                *reinterpret_cast<uint4*>(e8 + b) = make_uint4(rq[h][0], e[0], rq[h][1], e[1]);
                *reinterpret_cast<uint4*>(e8 + b + 2) = make_uint4(rq[h][2], e[2], rq[h][3], e[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (b + i < n) { out[b + i] = e[i]; e8[b + i] = make_uint2(rq[h][i], e[i]); }
            }
        } else if (b + 3 < n) {
            *reinterpret_cast<uint4*>(out + b) = make_uint4(e[0], e[1], e[2], e[3]);
            const uint4 r01 = *reinterpret_cast<const uint4*>(rect + b), r23 = *reinterpret_cast<const uint4*>(rect + b + 2);
            if (v[h][0]) einfo[b] = make_uint4(0u, r01.x, r01.y, e[0]);
            if (v[h][1]) einfo[b + 1] = make_uint4(0u, r01.z, r01.w, e[1]);
            if (v[h][2]) einfo[b + 2] = make_uint4(0u, r23.x, r23.y, e[2]);
            if (v[h][3]) einfo[b + 3] = make_uint4(0u, r23.z, r23.w, e[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (b + i < n) { out[b + i] = e[i]; if (v[h][i]) { const uint2 rc = rect[b + i]; einfo[b + i] = make_uint4(0u, rc.x, rc.y, e[i]); } }
        }
    }
}
#endif
