// scan_wave.h -- the record-base scan (A2) as a ONE-WAVE-per-tile chained scan that rides in another kernel's launch (round 5).
//
// The exclusive scan of `tiles` in Gaussian-id order (-> rbase, einfo) feeds the BACKWARD pass only; as a launch of its own (k_scan_lb<true, false>, 0.124 ms per
// 8-view step) it sat on the forward chain's critical path for nothing, and like every single-pass chained scan it is bound by look-back latency, not by the 32 bytes
// per Gaussian it moves.  Its tiles now occupy the first blocks of the recording forward compositing launch: 64-thread workgroups like the compositing waves, one wave
// = one 1024-element tile, dispatched ahead of the quadrant waves, which fill every other wave slot and keep the vector pipe busy while the scan's waves wait for their
// predecessors.  Same protocol as scan_sort.hip (ticket counter, one {flag, value} word per tile, relaxed agent-scope atomics, bounded spins).
#pragma once
#include "c3d_common.h"

#define SCANW_TILE 1024            // elements per wave: 4 sub-tiles of 256, lane t owns elements [4t, 4t + 4) of each (16-byte loads / stores on consecutive addresses)
#define SCANW_SPIN_LIMIT (1u << 21)
// state block: [0] ticket, [1] error word (unused here: the caller passes its own), then one 64-bit word per tile; zero before the launch
static inline size_t scan_wave_tmp_bytes(size_t n) { return c3d_align(8 + sizeof(unsigned long long) * ((n ? n : 1) / SCANW_TILE + 2)); }
static inline int scan_wave_blocks(size_t n) { return (int)(((n + SCANW_TILE - 1) / SCANW_TILE + 31) / 32 * 32); }      // a multiple of 32: the compositing blocks behind keep their XCD / quadrant mapping

#ifdef __HIPCC__
struct ScanWaveJob {      // all pointers: view 0's (view v lies v * vs bytes behind); blocks == 0: no job
    const uint32_t* in; uint32_t* out; const uint2* rect; uint4* einfo; uint32_t* state; uint32_t* err; uint32_t n; int blocks;
};
// out[i] = exclusive prefix of in[0..i); einfo[i] = {0, rect[i].x, rect[i].y, out[i]} where in[i] != 0.  Called by a whole 64-lane workgroup.
__device__ __forceinline__ void scan_wave_tile(const ScanWaveJob& j, size_t vs) {
    const uint32_t* __restrict__ in = c3d_view_ptr(j.in, vs);
    uint32_t* __restrict__ out = c3d_view_ptr(j.out, vs);
    const uint2* __restrict__ rect = c3d_view_ptr(j.rect, vs);
    uint4* __restrict__ einfo = c3d_view_ptr(j.einfo, vs);
    uint32_t* state = c3d_view_ptr(j.state, vs);
    unsigned long long* status = reinterpret_cast<unsigned long long*>(state + 2);
    const int lane = (int)threadIdx.x;
    const size_t n = j.n;
    uint32_t tk = 0;
    if (lane == 0) tk = atomicAdd(&state[0], 1u);
    const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
    if ((size_t)tile * SCANW_TILE >= n) return;
    uint32_t v[4][4], hs[4];
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const size_t b = (size_t)tile * SCANW_TILE + (size_t)h * 256 + (size_t)lane * 4;
        if (b + 3 < n) { const uint4 q = *reinterpret_cast<const uint4*>(in + b); v[h][0] = q.x; v[h][1] = q.y; v[h][2] = q.z; v[h][3] = q.w; }
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
    if (lane == 0) __hip_atomic_store(&status[tile], ((tile == 0 ? 2ull : 1ull) << 32) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t prefix = 0;
    if (tile > 0) {      // decoupled look-back, 64 predecessors per step: flag 1 = tile aggregate, 2 = inclusive prefix
        long long t0 = (long long)tile - 1;
        for (;;) {
            const long long t = t0 - lane;
            unsigned long long w = t >= 0 ? __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 32);      // below tile 0: an inclusive prefix of zero
            uint64_t incl_mask;
            int first_incl;
            uint32_t spins = 0;
            for (;;) {
                const uint32_t flag = (uint32_t)(w >> 32);
                incl_mask = __ballot(flag == 2u);
                const uint64_t notready = __ballot(flag == 0u);
                first_incl = incl_mask ? (int)__builtin_ctzll(incl_mask) : 63;
                const uint64_t relevant = first_incl >= 63 ? ~0ull : ((2ull << first_incl) - 1ull);
                if ((notready & relevant) == 0ull) break;
                if (++spins > SCANW_SPIN_LIMIT) { if (lane == 0) atomicOr(j.err, C3D_ERR_LOOKBACK); incl_mask = 1ull; first_incl = 0; w = 0; break; }
                __builtin_amdgcn_s_sleep(2);
                if (flag == 0u) w = __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            uint32_t contrib = (lane <= first_incl) ? (uint32_t)w : 0u;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) contrib += __shfl_xor(contrib, o, 64);
            prefix += contrib;
            if (incl_mask) break;
            t0 -= 64;
        }
        if (lane == 0) __hip_atomic_store(&status[tile], (2ull << 32) | (uint32_t)(prefix + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const size_t b = (size_t)tile * SCANW_TILE + (size_t)h * 256 + (size_t)lane * 4;
        uint32_t run = prefix + ex[h], e[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { e[i] = run; run += v[h][i]; }
        if (b + 3 < n) {
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
