// c3d_common.h -- shared device/host helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wave = 64 lanes everywhere in this tree; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define C3D_WAVE 64
#define C3D_TILE_X 16
#define C3D_TILE_Y 16
#define C3D_TILE_PIX (C3D_TILE_X * C3D_TILE_Y)

// error plumbing: every extern "C" entry returns 0 on success or a hipError_t / negative code;
// c3d_last_error() returns a thread-local message.
void c3d_set_error(const char* fmt, ...);
#define C3D_CHECK(expr)                                                                   \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            c3d_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return (int)_e;                                                               \
        }                                                                                 \
    } while (0)
#define C3D_LAUNCH_CHECK() C3D_CHECK(hipGetLastError())

static inline size_t c3d_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int c3d_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// `lanes` arguments of the multi-view entry points: the number of view GROUPS a call splits its views into (every stage of the chain is one launch per group, the
// groups follow each other on the caller's stream).  Rounds 1-3 ran one chain per view on a pool of library-owned streams; that pool is gone (round 4): no call
// creates streams or events of its own any more.
#define C3D_MAX_LANES 8

// ---- optional event timing (prof.hip) ----
#define C3D_PROF_SLOTS 22
enum { C3D_P_PREPROCESS = 0, C3D_P_DEPTH_SORT, C3D_P_SCAN, C3D_P_EMIT, C3D_P_TILE_SORT, C3D_P_RANGES, C3D_P_COMPOSITE_FWD,
       C3D_P_COMPOSITE_BWD, C3D_P_PREPROCESS_BWD, C3D_P_ADAM, C3D_P_MESH_RASTERIZE, C3D_P_MESH_INTERPOLATE, C3D_P_MESH_TEXTURE,
       C3D_P_MESH_ANTIALIAS, C3D_P_MESH_BWD, C3D_P_OTHER, C3D_P_MESH_RASTERIZE_BWD, C3D_P_MESH_INTERPOLATE_BWD,
       C3D_P_MESH_TEXTURE_BWD, C3D_P_MESH_ANTIALIAS_BWD, C3D_P_MSSSIM,
       C3D_P_MESH_RAS_TRI /* k_ras_tri alone, nested inside the rasterize group: the mesh line's per-kernel roofline */ };
// (C3D_P_RANGES has had no kernel since round 6 -- the last tile-sort pass leaves the per-tile ranges -- and keeps its slot so that the slot numbers of the others stay put)
void* c3d_prof_begin(int slot, hipStream_t s);
void c3d_prof_end(void* h, hipStream_t s);
struct C3dProfScope {
    void* h; hipStream_t s;
    C3dProfScope(int slot, hipStream_t st) : h(c3d_prof_begin(slot, st)), s(st) {}
    ~C3dProfScope() { c3d_prof_end(h, s); }
};

// ---- scan / sort primitives (scan_sort.hip) ----
// Single-pass (decoupled look-back) kernels.  Each primitive keeps a small state block at the head of `tmp` that must be zero when
// its kernels start: with zero_state = true (default, one view only) the call issues the hipMemsetAsync itself; a caller that runs several
// primitives of one view -- or of V views -- clears all their state blocks with ONE launch (c3d_zero_views) and passes false.
// A timed-out inter-workgroup wait (every spin is bounded) ORs C3D_ERR_LOOKBACK into `err` (device; nullptr: the primitive's own error word).
// V / vs: the launch covers V views (blockIdx.y); every pointer is that of view 0 and view v's lies v * vs bytes behind it (workspace
// slices at a uniform stride).  `err`, `tail_status` are shared by the views.
#define C3D_ERR_LOOKBACK 2u
// Inclusive or exclusive prefix sum of n uint32 values. `tmp` needs c3d_scan_tmp_bytes(n).
size_t c3d_scan_tmp_bytes(size_t n);
int c3d_scan_u32(const uint32_t* in, uint32_t* out, size_t n, bool exclusive, void* tmp, hipStream_t s, bool zero_state = true, uint32_t* err = nullptr);
// inclusive scan of the areas of the tile rects rect[idx[i]] ({x0 | y0 << 16, x1 | y1 << 16}; the gather is folded into the load) -> out; the gathered rects are left
// in rsort[i].  tail_meta (optional, device): receives min(total, tail_cap) in [0]; tail_status (optional): [0] |= 1 when total > tail_cap, [1] = max(total).
int c3d_scan_rect_gather(const uint2* rect, const uint32_t* idx, uint32_t* out, uint2* rsort, size_t n, void* tmp, hipStream_t s, bool zero_state,
                         uint32_t* tail_meta, uint32_t* tail_status, uint32_t tail_cap, uint32_t* err = nullptr, int V = 1, size_t vs = 0, uint32_t tail_hint = 0,
                         unsigned long long* tail_early = nullptr, bool rect4 = false);      // rect4: `rect` holds 4-byte packed rects (c3d_rect_pack); rsort receives them unpacked
// tail_hint (0 = none): the count the launches of the chain were sized for; tail_status[0] |= C3D_ST_BEYOND_HINT when the total exceeds it (information: workgroups looped),
// |= C3D_ST_OVERFLOW when it exceeds tail_cap (the buffers).  tail_early (optional): device-visible address of 8 bytes of pinned host memory that receives total << 32 | bits.
#define C3D_ST_OVERFLOW 1u
#define C3D_ST_BEYOND_HINT 4u
int c3d_zero_count(void* p, const uint32_t* count, uint32_t cap, hipStream_t s);            // bytes [0, min(*count, cap)) cleared, the count resident on the device
// the state of a sort laid out for n elements, cleared for the min(*n_dev, n) that are there (+ pre_bytes in front of tmp); launch sized for n_hint elements
int c3d_sort_zero_state_counted(void* tmp, size_t n, int end_bit, const uint32_t* n_dev, size_t n_hint, void* pre, size_t pre_bytes, hipStream_t s);
uint32_t* c3d_scan_error_word(void* tmp);
// up to three byte regions (off[r], bytes[r]: 16-byte aligned, multiples of 4) cleared in each of V slices, one launch
int c3d_zero_views(void* base0, size_t vs, int V, const size_t* off, const size_t* bytes, int regions, hipStream_t s);

// Stable LSD radix sort of (key,val) uint32 pairs over key bits [0, end_bit), n < 2^30.
// keys/vals are ping-pong buffers [2][n]; result index (0 or 1) is returned through *result_buf.
// iota_vals => the values are the element indices (vals0 is not read).  tmp: c3d_sort_tmp_bytes(n); the part that must be zero is
// the first c3d_sort_state_bytes(n, end_bit) bytes.
// n_dev (optional, device, per view): the real element count is min(*n_dev, n) -- n is then the capacity the launch is sized for, so a
// data-dependent count never has to come back to the host.
size_t c3d_sort_tmp_bytes(size_t n);
size_t c3d_sort_state_bytes(size_t n, int end_bit);
uint32_t* c3d_sort_error_word(void* tmp);
int c3d_sort_set_debug(unsigned long long* stamps);   // profiling hook (nullptr = off): [pass][tile][8] wall_clock64 stamps (single-view sorts)
// hist_done: the digit histograms of the keys are already in `tmp` -- the kernel that PRODUCED the keys counted them (k_emit for the tile sort: the keys are in its registers when it
// stores them) -- so the sort does not read the keys a first time.  Layout at the head of tmp (zeroed with the rest of the state, before the producer runs): uint32
// [C3D_SORT_HIST_SPLIT copies][C3D_SORT_MAX_PASSES][256]; a producer workgroup adds the count of digit d of pass p (key bits [8p, 8p + 8)) to ONE copy (any; spread them), the sort
// adds the copies up.
#define C3D_SORT_HIST_SPLIT 16
#define C3D_SORT_MAX_PASSES 4
int c3d_sort_pairs_u32(uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1, bool iota_vals,
                       size_t n, int end_bit, void* tmp, int* result_buf, hipStream_t s, const uint32_t* n_dev = nullptr, bool zero_state = true, uint32_t* err = nullptr,
                       int V = 1, size_t vs = 0, bool hist_done = false, size_t n_hint = 0,      // n_hint (one view, n_dev given): size the launches for n_hint < n elements; a larger count loops
                       uint2* ranges = nullptr);      // ranges (per view, cleared): the last pass leaves {~first position, last position + 1} of every key value there (atomicMax; {0, 0} = no such key) -- and writes NO sorted keys, only the values

// ---- device helpers ----
#ifdef __HIPCC__
// tile rect of a Gaussian, {x0 | y0 << 16, x1 | y1 << 16}, in FOUR bytes x0 | y0 << 8 | x1 << 16 | y1 << 24 when the tile grid is at most 255 x 255 (images up to 4080 px a side;
// GsParams::rect4): the one random access of the binning chain -- the emit-offset scan gathers rect[order[i]] -- then runs over a 4 MB table per 1 M Gaussians instead of 8 MB,
// which one XCD's L2 holds (round 5's counters: 838 MB fetched per 16-view launch for 190 MB of useful input, one 64-byte line per 8-byte gather)
__device__ __forceinline__ uint32_t c3d_rect_pack(uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) { return x0 | (y0 << 8) | (x1 << 16) | (y1 << 24); }
__device__ __forceinline__ uint2 c3d_rect_unpack(uint32_t r) { return make_uint2((r & 0xFFu) | ((r & 0xFF00u) << 8), ((r >> 16) & 0xFFu) | ((r >> 24) << 16)); }
// multi-view launches: blockIdx.y = view; per-view pointers are given for view 0 and view v's lies v * vs bytes behind it (NULL stays NULL)
template <class T> __device__ __forceinline__ T* c3d_view_ptr(T* p, size_t vs) { return p ? (T*)((char*)p + (size_t)blockIdx.y * vs) : p; }
__device__ __forceinline__ int c3d_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// wave-wide inclusive scan (64 lanes) of a uint32
__device__ __forceinline__ uint32_t c3d_wave_incl_scan(uint32_t v) {
    const int lane = c3d_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
// wave-wide float sum; every lane gets the total (xor butterfly)
__device__ __forceinline__ float c3d_wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
#endif
