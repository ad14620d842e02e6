// gs_backward.hip -- 3DGS backward: per-tile compositing gradients (A7) and per-Gaussian chain rule (A8).
// Contract: SURVEY.md section 2.3-A (A7, A8) and Appendix A; the outputs are the tensors the autograd
// Function behind main_3DGS_renderer.py:927-936 (reference call site) must return.
#include "gs_internal.h"
#include "gs_math.h"

// --- wave64 reduction of the backward pass's per-pair sums ------------------------------------------------------------------------------------------
// Rounds 1-3 folded the values across lanes with v_permlane32/16_swap + DPP butterflies (69 VALU instructions per walked pair, 144 cycles of them the reduction);
// round 4 first transposed all nine values through a wave-private LDS tile (9 ds_write_addtid_b32 + 4 ds_read_b128 per lane: 49 VALU / 17 LDS instructions per pair)
// -- which the SQ counters showed LDS-bound (LDS pipe 81 % busy, profiles/r04y_sq_*) -- and then put one DPP fold in front of the transposition (below).  The two
// earlier forms are gone from the source; their measurements are profiles/r03z_*, r04l_*, r04q_* (same-box A/B of the last two).
#define BWD_RED_ROW 68      // dwords per tile row: 64 lanes + 4 (16 lanes of a ds_read_b128 group then hit 16 different four-bank groups)
__device__ __forceinline__ uint32_t lds_scalar_address(const void* p) {   // LDS byte address (low half of the generic address) as a scalar, for M0
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p);
}
// base = lds_scalar_address(tile), taken once per kernel.  M0 is in the clobber list (the backend merges / hoists identical M0 initialisations of its own and must
// not assume an earlier value survives the block).  The s_nop 0 inside the block is the wait state the ISA demands between an SALU write of M0 and an LDS "add-TID" instruction
// (the compiler's hazard recogniser does not look inside inline asm; without it the first store of a wave goes to whatever M0 held before).  ds_write_addtid_b32:
// address = M0 + offset + 4 * lane -- no address VGPR, 2 LDS cycles per store against 4 for ds_write_b32.  LDS operations of one wave execute in order, so the reads
// see this pair's stores and the next pair's stores come after them.
// --- ONE DPP fold, then the transposition ---------------------------------------------------------------------------------------------------------------
// SQ counters of the transposing kernel (profiles/r04y_sq_*): LDS pipe 81 % busy (48 LDS-array cycles per walked pair: 10 for the splat record, 18 for the nine
// stores, 16 for the four 16-byte reads, 4 for the result) while its 49 VALU instructions per pair leave the vector pipe room -- the reduction had moved the kernel
// from VALU-bound to LDS-bound.  Here neighbouring values are folded pairwise BEFORE they go through LDS: within every 16-lane row, lanes 0-7 take
// a[l] + a[l + 8] of the even value and lanes 8-15 take b[l] + b[l - 8] of the odd one (two v_add_f32_dpp row_ror:8, the second writing banks 2-3 only), so five
// registers carry nine (ten) values: 5 stores + 2 reads of 16 bytes per lane (8 columns) instead of 9 + 4: 32 LDS-array cycles per pair, +9 DPP adds - 4 packed adds.
// Lane L owns value k = L >> 2 and lane-row r = L & 3: eight columns from bwd_fold_slot(k) + 16 r.  Fixed order of additions: bit-reproducible.
#define BWD_FOLD_ROWS 5
// Which (row, half) a value travels in is chosen so that the 16-byte reads are free of bank conflicts: ds_read_b128 serves lanes {0-3, 12-15, 20-27} -- values 0, 3, 5, 6 --
// in one LDS cycle and {4-11, 16-19, 28-31} -- values 1, 2, 4, 7 -- in the next; a lane's four banks start at (row * 68 + 8 half + 16 r) mod 64, so the four values of
// a group need four different (row * 4 + 8 half) mod 16.  With value k -> (row, half):  0 (0,0)  3 (1,0)  5 (0,1)  6 (1,1)  |  1 (2,1)  2 (3,1)  4 (2,0)  7 (3,0)  |  8 (4,0)  9 (4,1)
// they are 0, 4, 8, 12 in both groups.  (The obvious k -> (k / 2, k % 2) costs a second cycle on 14 % of the LDS time: SQ_LDS_BANK_CONFLICT, profiles/r04s_sq_*.)
__device__ __forceinline__ int bwd_fold_slot(int k) {      // dword offset of value k's first column inside the tile (lane-row 0)
    const int row = (int)((0x4431021320ull >> (4 * k)) & 15ull), half = (0x266 >> k) & 1;     // rows 0,2,3,1,2,0,1,3,4,4   halves 0,1,1,0,0,1,1,0,0,1  for k = 0..9
    return row * BWD_RED_ROW + 8 * half;
}
// (clang warns that M0 is a reserved register it will not save around the block -- which is what is wanted: the clobber only tells it the value is gone)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int NV>
__device__ __forceinline__ float wave_reduce_fold(const float* tile /* [BWD_FOLD_ROWS][BWD_RED_ROW], wave-private */, uint32_t base, float (&v)[NV], int lane) {
    static_assert(NV == 9 || NV == 10, "nine values, ten with the depth channel");
    // rows: (v0 | v5) (v3 | v6) (v4 | v1) (v7 | v2) (v8 | v9).  s_nop 1: the wait states between the VALU writes of the products and the first DPP read; every later
    // DPP source was written before the block as well
    if (NV == 10)
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %0, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" "v_add_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                     "v_add_f32_dpp %2, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" "v_add_f32_dpp %3, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                     "v_add_f32_dpp %4, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                     "s_mov_b32 m0, %10\n\t"
                     "s_nop 0\n\t"
                     "ds_write_addtid_b32 %0 offset:0\n\t" "ds_write_addtid_b32 %1 offset:272\n\t" "ds_write_addtid_b32 %2 offset:544\n\t"
                     "ds_write_addtid_b32 %3 offset:816\n\t" "ds_write_addtid_b32 %4 offset:1088"
                     : "+v"(v[0]), "+v"(v[3]), "+v"(v[4]), "+v"(v[7]), "+v"(v[8])
                     : "v"(v[5]), "v"(v[6]), "v"(v[1]), "v"(v[2]), "v"(v[NV - 1]), "s"(base) : "memory", "m0");
    else
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %0, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" "v_add_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                     "v_add_f32_dpp %2, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t" "v_add_f32_dpp %3, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                     "s_mov_b32 m0, %9\n\t"
                     "s_nop 0\n\t"
                     "ds_write_addtid_b32 %0 offset:0\n\t" "ds_write_addtid_b32 %1 offset:272\n\t" "ds_write_addtid_b32 %2 offset:544\n\t"
                     "ds_write_addtid_b32 %3 offset:816\n\t" "ds_write_addtid_b32 %4 offset:1088"
                     : "+v"(v[0]), "+v"(v[3]), "+v"(v[4]), "+v"(v[7]), "+v"(v[8])
                     : "v"(v[5]), "v"(v[6]), "v"(v[1]), "v"(v[2]), "s"(base) : "memory", "m0");
    const int k = min(lane >> 2, NV - 1), r = lane & 3;
    const float4* rp = reinterpret_cast<const float4*>(tile + bwd_fold_slot(k) + 16 * r);
    const float4 a = rp[0], b = rp[1];
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f p0 = v2f{a.x, a.y} + v2f{a.z, a.w}, p1 = v2f{b.x, b.y} + v2f{b.z, b.w};
    const v2f q0 = p0 + p1;
    float t = q0.x + q0.y;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "+v"(t));
    return t;                           // lanes 4k .. 4k+3 hold the wave's sum of value k (k < NV); lanes past 4 NV - 1 repeat value NV - 1
}
#pragma clang diagnostic pop

// ------------------------------------------------------------------------------------------
// A7 composite backward: a workgroup per 16x16 tile, wave w = its 8x8 quadrant w, one pixel per lane; the tile's list is visited back to front in rounds of
// BWD_ROUND (64) positions staged through LDS, and a wave walks exactly the positions its quadrant blended (the forward pass's activity bytes).  Every lane
// re-derives alpha / T for its pixel; the nine (ten with a depth gradient) per-pair partial sums are summed over the wave (wave_reduce_fold), parked in LDS per
// (wave, position), added over the four waves in a fixed order and written ONCE as a 48-byte record per (tile, splat) pair at the pair's record index.  The records
// of one Gaussian are contiguous there, so the per-Gaussian kernel (A8) sums them without a single atomic: the whole backward pass is deterministic.
// record layout (GS_PAIR_FLOATS = 12): [c0, c2, c1, depth | m0, m1y, m1x, m2xx | m2xy, 0, m2yy, 0]  (m* = moments of dL/dG*G about the pixel)
// ------------------------------------------------------------------------------------------
// Positions per staging round.  64 keeps a workgroup at 18.4 KB of LDS (eight workgroups = 32 waves per CU); 128 and 32 measured slower in rounds 1-2.
#define BWD_ROUND 64

// Round 3, measured and dropped (commit a5d0b08, profiles/r03/r03b_*): this kernel as ONE WAVE PER QUADRANT (64-lane workgroups, no barriers, a wave-private
// compacted list as in k_composite_fwd_w, one record per (quadrant, pair) in a four-slot record group that A8 adds up).  k_composite_bwd_w<true,false> 0.371 ms
// against 0.393 ms for this kernel on the same box (-5.5 %): the 25 % of wave-round slots that wait at this kernel's barriers are not lost time -- eight
// waves per SIMD from several workgroups keep the VALU pipe 92 % busy either way -- while A8 went 0.58 -> 0.99 ms per 8-view step over the 2.2 x as many
// records and their four valid bytes per pair.  Net -4 % on the step: reverted.  What bounds this kernel is the VALU cost of a walked pair on gfx950
// (profiles/r01f_valu_rate_microbench.txt): ~127 cycles of per-pixel arithmetic + ~144 of the ten-value wave reduction (8 v_permlane swaps at 8.3,
// 12 DPP adds at 4.1, 9 adds at 2.9) = the 271 the SQ counters show.
// Round 4, measured and dropped (profiles/r04*, same-box A/B): reading the list in windows of 256 positions, compacting each to the pairs some quadrant blended (30 % of
// the positions a pixel reached) and staging only those -- rounds of 64 BLENDED pairs instead of 64 list positions: 2.45-2.47 ms against 2.41-2.42 ms per 8-view launch.
// The per-round work (record gather, record index, barriers, epilogue) is not where the time goes; the walk is.
// DEPTH: some caller-supplied dL/ddepth exists.  The fused training step has none (the reference's loss reads image and alpha only, main_3DGS.py:184-192):
// its instance drops the depth channel from the per-splat dot product, from the products and from the ten-value reduction (nine values).
// blockIdx.y = view of a multi-view launch: the state pointers are view 0's (view v lies v * vs bytes behind), pixel-space inputs come from the per-view table `px`.
// RECT4: einfo holds eight bytes per Gaussian, {packed rect, record base} (GsParams::rect4) -- a compile-time switch: as a uniform branch at the gather it cost the kernel 1.5 %
template <bool LOSS, bool DEPTH, bool RECT4>
__global__ void __launch_bounds__(256) k_composite_bwd(GsParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                        const uint4* __restrict__ einfo,
                                                        const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                                                        const float4* __restrict__ rec2, const float* __restrict__ final_T,
                                                        const uint32_t* __restrict__ n_contrib, GsBwdPix px,
                                                        const uint8_t* __restrict__ pact, size_t pstride, float4* __restrict__ pairgrad, uint8_t* __restrict__ pvalid, uint32_t cap,
                                                        GsPixelLossW plw, size_t vs) {
    ranges = c3d_view_ptr(ranges, vs); point_list = c3d_view_ptr(point_list, vs); einfo = c3d_view_ptr(einfo, vs);
    rec0 = c3d_view_ptr(rec0, vs); rec1 = c3d_view_ptr(rec1, vs); rec2 = c3d_view_ptr(rec2, vs); final_T = c3d_view_ptr(final_T, vs);
    n_contrib = c3d_view_ptr(n_contrib, vs); pact = c3d_view_ptr(pact, vs); pairgrad = c3d_view_ptr(pairgrad, vs); pvalid = c3d_view_ptr(pvalid, vs);
    const int vw = blockIdx.y;
    const float* __restrict__ bg = px.bg[vw];
    const float* __restrict__ dL_dcolor = px.dcolor[vw];
    const float* __restrict__ dL_ddepth = px.ddepth[vw];
    const float* __restrict__ dL_dalpha_px = px.dalpha[vw];
    __shared__ float4 s0[BWD_ROUND];
    __shared__ float4 s1[BWD_ROUND];
    __shared__ float4 s2[BWD_ROUND];
    __shared__ uint32_t se[BWD_ROUND];
    __shared__ uint32_t smask[BWD_ROUND];
    constexpr int NV = DEPTH ? 10 : 9;                          // values summed per walked pair: colour 3 [, depth], m0, m1 x 2, m2 x 3
    __shared__ float acc[4][NV][BWD_ROUND + 1];                 // per wave and value: the sums of the round's splats (+1: the writers of a wave -- lanes 0, 4, 8, ... -- land in different banks)
    __shared__ __attribute__((aligned(16))) float red[4][BWD_FOLD_ROWS][BWD_RED_ROW];   // wave_reduce_fold's transposition tile, one per wave (18.4 KB per workgroup in all: eight per CU)
    __shared__ int s_uptow[4];   // per quadrant: the deepest list position (+1) one of its pixels blended = how far its plane of the activity record is valid
    int tx, ty;
    if (!gs_block_tile(blockIdx.x, p.gx, p.gy, tx, ty)) return;
    const int tile = ty * p.gx + tx;
    const int lane = c3d_lane(), wave = threadIdx.x >> 6;
    const int X0 = tx * C3D_TILE_X, Y0 = ty * C3D_TILE_Y;
    const int pxi = X0 + ((wave & 1) << 3) + (lane & 7), pyi = Y0 + ((wave >> 1) << 3) + (lane >> 3);
    const bool inside = pxi < p.W && pyi < p.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const uint2 rg = gs_tile_range(ranges[tile]);
    const size_t P = (size_t)p.W * p.H, pid = (size_t)pyi * p.W + pxi;

    const float T_final = inside ? final_T[pid] : 0.f;
    float T = T_final;
    const int last = inside ? (int)n_contrib[pid] : 0;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f, dLd = 0.f, dLa = 0.f;
    if (inside) {
        if (dL_dcolor) { dLp0 = dL_dcolor[pid]; dLp1 = dL_dcolor[P + pid]; dLp2 = dL_dcolor[2 * P + pid]; }
        dLd = (DEPTH && dL_ddepth) ? dL_ddepth[pid] : 0.f;
        dLa = dL_dalpha_px ? dL_dalpha_px[pid] : 0.f;
    }
    if (LOSS) {   // the step's pixel loss: clamp, optional mask, L1 / L2 / alpha-MSE value and gradient
        __shared__ float s_loss[4];
        struct { const float *color, *alpha, *tcolor, *talpha, *cmask; float w_l1, w_l2, w_a, scale; float* tile_loss; } pl =
            {px.color[vw], px.alpha[vw], px.tcolor[vw], px.talpha[vw], px.cmask[vw], plw.w_l1, plw.w_l2, plw.w_a, plw.scale, c3d_view_ptr(plw.tile_loss, vs)};
        float l = 0.f;
        if (inside) {
            const float inv3p = 1.f / (3.f * (float)P), invp = 1.f / (float)P;
            const float mk = pl.cmask ? pl.cmask[pid] : 1.f;
            float gch[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float c = pl.color[ch * P + pid];
                const float cc = fminf(fmaxf(c, 0.f), 1.f);
                const float d = (cc - pl.tcolor[ch * P + pid]) * mk;
                l += (pl.w_l1 * fabsf(d) + pl.w_l2 * d * d) * inv3p;
                const float pass = (c >= 0.f && c <= 1.f) ? 1.f : 0.f;
                const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                gch[ch] = pl.scale * pass * mk * (pl.w_l1 * sg + 2.f * pl.w_l2 * d) * inv3p;
            }
            dLp0 += gch[0]; dLp1 += gch[1]; dLp2 += gch[2];
            if (pl.talpha) { const float d = pl.alpha[pid] - pl.talpha[pid]; l += pl.w_a * d * d * invp; dLa += pl.scale * 2.f * pl.w_a * d * invp; }
        }
        l = c3d_wave_sum(l * pl.scale);
        if (lane == 0) s_loss[wave] = l;
        __syncthreads();
        if (threadIdx.x == 0 && pl.tile_loss) pl.tile_loss[tile] = s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3];
    }
    const uint32_t red_base = lds_scalar_address(&red[wave][0][0]);
    const float bg_dot = bg[0] * dLp0 + bg[1] * dLp1 + bg[2] * dLp2;
    float Rdot = T_final * bg_dot;

    // list positions no pixel of the tile reached need no work and get no record
    if (threadIdx.x < 4) s_uptow[threadIdx.x] = 0;
    __syncthreads();
    atomicMax(&s_uptow[wave], last);
    __syncthreads();
    const int up0 = s_uptow[0], up1 = s_uptow[1], up2 = s_uptow[2], up3 = s_uptow[3];
    const int upto = max(max(up0, up1), max(up2, up3));   // positions [0, upto) matter
    // record index of the pair (this tile, Gaussian gid): the Gaussian's record base + row-major position of the tile inside its rect
    auto emit_index = [&](uint32_t gid) -> uint32_t {
        if (RECT4) {      // {packed rect, record base}: eight bytes per Gaussian -- the table this gather runs over is half the size
            const uint2 e8 = reinterpret_cast<const uint2*>(einfo)[gid];
            const int ex0 = (int)(e8.x & 0xFFu), ey0 = (int)((e8.x >> 8) & 0xFFu), ex1 = (int)((e8.x >> 16) & 0xFFu);
            return e8.y + (uint32_t)((ty - ey0) * (ex1 - ex0) + (tx - ex0));
        }
        const uint4 ei = einfo[gid];
        const int ex0 = (int)(ei.y & 0xFFFFu), ey0 = (int)(ei.y >> 16), ex1 = (int)(ei.z & 0xFFFFu);
        return ei.w + (uint32_t)((ty - ey0) * (ex1 - ex0) + (tx - ex0));
    };
    // Which (quadrant, splat) pairs to walk comes from the forward pass (`pact`, k_composite_fwd<true>): exactly those that blended the splat
    // into at least one pixel.  Everything else -- the geometric quadrant test, "behind the deepest pixel of the quadrant", occluded, below
    // 1/255 on the pixel grid -- is implied, so a walked pair always has an active lane.  A pair no quadrant blended gets NO record: `pvalid`
    // (one byte per pair, cleared by the launcher) marks the pairs that do; the per-Gaussian pass skips the others.
    for (int base = 0; base < upto; base += BWD_ROUND) {
        __syncthreads();
        const int n = min(BWD_ROUND, upto - base);
        gs_stage_round(point_list + rg.x + (upto - 1 - base), n, rec0, s0, s1, s2, -1);   // slot t <- list position upto-1-base-t
        if ((int)threadIdx.x < n) {
            const int kk = upto - 1 - base - (int)threadIdx.x;          // position in the tile's list
            const uint32_t kpos = rg.x + (uint32_t)kk;
            // plane w is read only where quadrant w's forward wave certainly wrote it (gs_pair_activity)
            smask[threadIdx.x] = (kk < up0 ? (uint32_t)(pact[kpos] & 1u) : 0u) | (kk < up1 ? (uint32_t)(pact[pstride + kpos] & 1u) << 1 : 0u) |
                                 (kk < up2 ? (uint32_t)(pact[2 * pstride + kpos] & 1u) << 2 : 0u) | (kk < up3 ? (uint32_t)(pact[3 * pstride + kpos] & 1u) << 3 : 0u);
            se[threadIdx.x] = emit_index(point_list[kpos]);   // unconditionally: the two dependent loads overlap the activity byte's
        }
        __syncthreads();
        if ((int)threadIdx.x < n) {
            const float4 a0 = s0[threadIdx.x];
            s0[threadIdx.x] = make_float4(a0.x, a0.y, GS_CONIC_HALF * a0.z, GS_CONIC_FULL * a0.w);   // conic pre-scaled as in the forward pass
            s1[threadIdx.x].x = GS_CONIC_HALF * s1[threadIdx.x].x;
        }
        __syncthreads();
        for (int c = 0; c < n; c += 64) {
            const int jj = c + lane;
            uint64_t m = __ballot(jj < n && ((smask[jj] >> wave) & 1u));
            while (m) {
                const int bitpos = (int)__builtin_ctzll(m);
                const int j = c + bitpos;
                m = gs_clear_bit64(m, bitpos);
                const int k = upto - 1 - base - j;   // list position of this splat
                const float4 a0 = s0[j], a1 = s1[j], a2 = s2[j];
                const float dx = a0.x - pxf, dy = a0.y - pyf;
                const float power = gs_power(a0, a1.x, dx, dy);      // log2(e) * (-q/2)
                const float G = __builtin_amdgcn_exp2f(power);
                const float oG = a1.y * G;                               // alpha before the 0.99 cap
                // act = (k < last) && (power <= 0) && (alpha >= 1/255) as a wave mask in an SGPR pair: the compares are written as asm so that the
                // mask can feed v_cndmask_b32_e64 directly (a ballot of the C++ bool costs a select + compare round trip through a VGPR).
                // min(0.99, oG) >= 1/255  <=>  oG >= 1/255: the test reads oG, the same bits the forward pass's decision was taken on.
                uint64_t am;
                {
                    uint64_t c0, c1, c2;
                    asm("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(c0) : "s"(k), "v"(last));     // k is wave-uniform: scalar operand, no v_mov
                    asm("v_cmp_ge_f32_e64 %0, 0, %1" : "=s"(c1) : "v"(power));
                    asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(c2) : "v"(1.f / 255.f), "v"(oG));
                    am = c0 & c1 & c2;
                }
                {   // no branch on am: it is non-zero for every recorded pair (and were it not, the sums below would come out as zeros)
                    // dL/dalpha_k = T_k (c_k . dL) - [sum_{j behind k} (c_j . dL) alpha_j T_j + T_final bg . dL] / (1 - alpha_k)
                    // with (c . dL) taken over colour, depth and alpha channels; Rdot carries the bracket.
                    // ONE select per pair (round 4; three before): an inactive lane continues with oG = 0, hence alpha = 0, 1 / (1 - alpha) = 1 exactly, T unchanged,
                    // weight 0 and moments 0 -- the same instructions for every lane, no exec-masked branch, no zero-initialised temporaries.
                    const float oGe = sel64z(am, oG);
                    float alpha;   // fminf(0.99f, oGe) without the v_max_f32 x, x the compiler puts in front of it (oGe comes out of inline asm: it cannot know the value is canonical)
                    asm("v_min_f32_e32 %0, 0x3f7d70a4, %1" : "=v"(alpha) : "v"(oGe));
                    const float inv = __builtin_amdgcn_rcpf(1.f - alpha);
                    const float Tn = T * inv;
                    T = Tn;
                    const float w = alpha * Tn;
                    const float sdot = DEPTH ? __builtin_fmaf(a1.z, dLp0, __builtin_fmaf(a1.w, dLp1, __builtin_fmaf(a2.x, dLp2, __builtin_fmaf(a2.y, dLd, dLa))))
                                             : __builtin_fmaf(a1.z, dLp0, __builtin_fmaf(a1.w, dLp1, __builtin_fmaf(a2.x, dLp2, dLa)));
                    const float dL_dalpha = Tn * sdot - Rdot * inv;
                    Rdot += w * sdot;
                    // screen-space part as raw moments of w2 = dL/dG * G; turned into mean/conic/opacity gradients per Gaussian in A8
                    const float m0 = oGe * dL_dalpha;
                    const float m1x = m0 * dx, m1y = m0 * dy;
                    float vals[NV];
                    vals[0] = w * dLp0; vals[1] = w * dLp1; vals[2] = w * dLp2; vals[3] = m0; vals[4] = m1x; vals[5] = m1y;
                    vals[6] = m1x * dx; vals[7] = m1x * dy; vals[8] = m1y * dy;
                    if (DEPTH) vals[NV - 1] = w * dLd;
                    const float tsum = wave_reduce_fold<NV>(&red[wave][0][0], red_base, vals, lane);
                    // every lane stores: the four lanes of a quad hold the same sum and lanes past value NV - 1 hold value NV - 1's (wave_reduce_fold), so all writers of
                    // an address carry the same bits -- no exec save / branch / restore around one store per walked pair
                    acc[wave][min(lane >> 2, NV - 1)][j] = tsum;
                }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < n) {   // one record per (tile, splat): fixed-order sum over the waves that handled it
            const int j = threadIdx.x;
            const uint32_t mk = smask[j];
            if (mk && se[j] < cap) {
                float r[NV];
#pragma unroll
                for (int q = 0; q < NV; q++) r[q] = 0.f;
#pragma unroll
                for (int w = 0; w < 4; w++)
                    if ((mk >> w) & 1u) {   // the other waves left nothing in acc
#pragma unroll
                        for (int q = 0; q < NV; q++) r[q] += acc[w][q][j];
                    }
                // record layout (read by the per-Gaussian pass): [c0, c2, c1, depth | m0, m1y, m1x, m2xx | m2xy, 0, m2yy, 0]
                float4* out = pairgrad + (size_t)se[j] * 3;
                out[0] = make_float4(r[0], r[2], r[1], DEPTH ? r[NV - 1] : 0.f);
                out[1] = make_float4(r[3], r[5], r[4], r[6]);
                out[2] = make_float4(r[7], 0.f, r[8], 0.f);
                pvalid[se[j]] = 1;
            }
        }
    }
}

// Two stages, both with a fixed order (the same loss bits every run): every view sums ITS terms (per-tile partials + the MS-SSIM term's slot) in one
// workgroup of ONE launch for all views of a group (blockIdx.x = view), and one lane adds the V view sums, in view order, after the last group.
__global__ void __launch_bounds__(1024) k_sum_view_loss(const float* __restrict__ t, int n, float* __restrict__ out, size_t vs) {
    __shared__ float red[16];
    t = (const float*)((const char*)t + (size_t)blockIdx.x * vs); out = (float*)((char*)out + (size_t)blockIdx.x * vs);
    float l = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) l += t[i];      // terms strided over the lanes, then a fixed tree
    l = c3d_wave_sum(l);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) { float q = 0.f; for (int w = 0; w < 16; w++) q += red[w]; out[0] = q; }
}
// status / status_host (optional): the step's two status words are final when this launch runs (it is ordered behind every kernel that can raise a bit) -- lane 0 stores them,
// {flags, largest pair count}, as ONE 64-bit system-scope store into pinned host memory the device can address: the host of a launch-bound training loop learns how the step
// went without a copy launch in the stream, and two kernels (per-Gaussian pass, optimizer) before the step's last kernel has finished
__device__ __forceinline__ void gs_status_to_host(const uint32_t* status, unsigned long long* status_host) {
    if (status_host) __hip_atomic_store(status_host, ((unsigned long long)status[1] << 32) | status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(64) k_sum_views_loss(const float* __restrict__ first_view_sum, size_t stride, int V, float* __restrict__ loss_out,
                                                        const uint32_t* __restrict__ status, unsigned long long* __restrict__ status_host) {
    if (threadIdx.x != 0) return;
    float acc = 0.f;
    for (int v = 0; v < V; v++) acc += *(const float*)((const char*)first_view_sum + (size_t)v * stride);      // views in order
    loss_out[0] += acc;             // one writer: the launch is ordered behind everything else that touches *loss_out on this stream
    gs_status_to_host(status, status_host);
}
// both stages in one launch when all views of the step went through ONE group (the usual case): the same per-view pattern, the views one after the other, the same bits
__global__ void __launch_bounds__(1024) k_sum_group_loss(const float* __restrict__ t0, int n, size_t vs, int V, float* __restrict__ loss_out,
                                                         const uint32_t* __restrict__ status, unsigned long long* __restrict__ status_host) {
    // The launch is ONE workgroup on the step's critical path: the views' terms are requested together (four strides of every view in flight per lane) -- view after view
    // with a barrier pair each, the dependent round trips were 34 us of the 8-view step (profiles/r06z_fwdbwd_kernel_stats.csv).  The additions are those of the loop
    // `for v: for i: l += t_v[i]` in the same order (a term past the end adds +0 to a sum that is never -0).
    __shared__ float red[GS_MAX_BWD_VIEWS][16];
    float l[GS_MAX_BWD_VIEWS];
#pragma unroll
    for (int v = 0; v < GS_MAX_BWD_VIEWS; v++) l[v] = 0.f;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * 1024) {
        float x[4][GS_MAX_BWD_VIEWS];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * 1024;
#pragma unroll
            for (int v = 0; v < GS_MAX_BWD_VIEWS; v++)
                x[u][v] = (v < V && i < n) ? ((const float*)((const char*)t0 + (size_t)v * vs))[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int v = 0; v < GS_MAX_BWD_VIEWS; v++) l[v] += x[u][v];
        }
    }
#pragma unroll
    for (int v = 0; v < GS_MAX_BWD_VIEWS; v++) {
        if (v < V) {      // (uniform)
            const float s = c3d_wave_sum(l[v]);
            if ((threadIdx.x & 63) == 0) red[v][threadIdx.x >> 6] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = 0.f;
        for (int v = 0; v < V; v++) { float q = 0.f; for (int w = 0; w < 16; w++) q += red[v][w]; acc += q; }
        loss_out[0] += acc;
        gs_status_to_host(status, status_host);
    }
}
int gs_launch_sum_group_loss(const float* terms, int n, int V, size_t vs, float* loss_out, hipStream_t s, const uint32_t* status, unsigned long long* status_host) {
    if (n <= 0 || V <= 0 || !loss_out) return 0;
    if (V > GS_MAX_BWD_VIEWS) { c3d_set_error("gs_launch_sum_group_loss: %d views in one group (at most %d)", V, GS_MAX_BWD_VIEWS); return -1; }
    hipLaunchKernelGGL(k_sum_group_loss, dim3(1), dim3(1024), 0, s, terms, n, vs, V, loss_out, status, status_host);
    C3D_LAUNCH_CHECK();
    return 0;
}
int gs_launch_sum_view_loss(const float* terms, int n, float* view_sum, int V, size_t vs, hipStream_t s) {
    if (n <= 0 || V <= 0) return 0;
    hipLaunchKernelGGL(k_sum_view_loss, dim3(V), dim3(1024), 0, s, terms, n, view_sum, vs);
    C3D_LAUNCH_CHECK();
    return 0;
}
int gs_launch_sum_tile_loss(const float* first_view_sum, size_t view_stride_bytes, int V, float* loss_out, hipStream_t s, const uint32_t* status, unsigned long long* status_host) {
    if (V == 0 || !loss_out) return 0;
    hipLaunchKernelGGL(k_sum_views_loss, dim3(1), dim3(64), 0, s, first_view_sum, view_stride_bytes, V, loss_out, status, status_host);
    C3D_LAUNCH_CHECK();
    return 0;
}

// V views per launch (grid.y).  pvalid must be clear on entry (the fused paths clear it with the binning state: c3d_zero_views; the drop-in path below).
int gs_launch_composite_bwd(const GsParams& p, const GsGeom& g, const GsBinning& b, int res, const GsImage& im, const GsBwdPix& px, bool depth,
                            float* pairgrad, uint8_t* pvalid, hipStream_t s, uint32_t cap, const GsPixelLossW* plw, int V, size_t vs) {
    const int tiles = p.gx * p.gy;
    if (tiles == 0 || V <= 0) return 0;
    const dim3 grid(gs_block_count(p.gx, p.gy), V);
#define GS_BWD_LAUNCH(LOSS_, DEPTH_, PL_)                                                                                                                        \
    do { if (p.rect4) hipLaunchKernelGGL((k_composite_bwd<LOSS_, DEPTH_, true>), grid, dim3(256), 0, s, p, b.ranges, b.tval[res], g.einfo, g.rec0, g.rec1, g.rec2, im.final_T, \
                       im.n_contrib, px, gs_pair_activity(b, res), b.pair_stride, (float4*)pairgrad, pvalid, cap, PL_, vs);                                         \
         else hipLaunchKernelGGL((k_composite_bwd<LOSS_, DEPTH_, false>), grid, dim3(256), 0, s, p, b.ranges, b.tval[res], g.einfo, g.rec0, g.rec1, g.rec2, im.final_T, \
                       im.n_contrib, px, gs_pair_activity(b, res), b.pair_stride, (float4*)pairgrad, pvalid, cap, PL_, vs); } while (0)
    if (plw) { if (depth) GS_BWD_LAUNCH(true, true, *plw); else GS_BWD_LAUNCH(true, false, *plw); }
    else     { if (depth) GS_BWD_LAUNCH(false, true, GsPixelLossW{}); else GS_BWD_LAUNCH(false, false, GsPixelLossW{}); }
#undef GS_BWD_LAUNCH
    C3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Per-view geometric chain rule shared by the A8 kernels: summed moments of one Gaussian in one view -> gradient of the projected mean
// (NDC-scaled, g2x/g2y), contribution to the 3D mean (EWA Jacobian, projection, depth output) and to the 3D covariance.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void bwd_geom_chain(const float3 m, const float c3[6], const Mat16& V, const Mat16& PJ, float tanfovx, float tanfovy,
                                               float focal_x, float focal_y, int W, int H, float cA, float cB, float cC, float m1x, float m1y,
                                               float m2xx, float m2xy, float m2yy, float g_depth, float& g2x, float& g2y, float dmean[3], float dcov[6]) {
    g2x = -(cA * m1x + cB * m1y) * (0.5f * W);
    g2y = -(cC * m1y + cB * m1x) * (0.5f * H);
    const float dcx = -0.5f * m2xx, dcy = -0.5f * m2xy, dcz = -0.5f * m2yy;
    float T2[2][3], ST0[3], ST1[3];
    float3 t; bool xin, yin;
    ewa_T2(m, V, tanfovx, tanfovy, focal_x, focal_y, T2, t, xin, yin);
    sigma_T(c3, T2, ST0, ST1);
    const float a = T2[0][0] * ST0[0] + T2[0][1] * ST0[1] + T2[0][2] * ST0[2] + 0.3f;
    const float b = T2[0][0] * ST1[0] + T2[0][1] * ST1[1] + T2[0][2] * ST1[2];
    const float c = T2[1][0] * ST1[0] + T2[1][1] * ST1[1] + T2[1][2] * ST1[2] + 0.3f;
    const float denom = a * c - b * b;
    const float d2i = 1.f / (denom * denom + 0.0000001f);
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) dcov[i] = 0.f;
    if (d2i != 0.f) {
        dL_da = d2i * (-c * c * dcx + 2.f * b * c * dcy + (denom - a * c) * dcz);
        dL_dc = d2i * (-a * a * dcz + 2.f * a * b * dcy + (denom - a * c) * dcx);
        dL_db = d2i * 2.f * (b * c * dcx - (denom + 2.f * b * b) * dcy + a * b * dcz);
        dcov[0] = T2[0][0] * T2[0][0] * dL_da + T2[0][0] * T2[1][0] * dL_db + T2[1][0] * T2[1][0] * dL_dc;
        dcov[3] = T2[0][1] * T2[0][1] * dL_da + T2[0][1] * T2[1][1] * dL_db + T2[1][1] * T2[1][1] * dL_dc;
        dcov[5] = T2[0][2] * T2[0][2] * dL_da + T2[0][2] * T2[1][2] * dL_db + T2[1][2] * T2[1][2] * dL_dc;
        dcov[1] = 2.f * T2[0][0] * T2[0][1] * dL_da + (T2[0][0] * T2[1][1] + T2[0][1] * T2[1][0]) * dL_db + 2.f * T2[1][0] * T2[1][1] * dL_dc;
        dcov[2] = 2.f * T2[0][0] * T2[0][2] * dL_da + (T2[0][0] * T2[1][2] + T2[0][2] * T2[1][0]) * dL_db + 2.f * T2[1][0] * T2[1][2] * dL_dc;
        dcov[4] = 2.f * T2[0][2] * T2[0][1] * dL_da + (T2[0][1] * T2[1][2] + T2[0][2] * T2[1][1]) * dL_db + 2.f * T2[1][1] * T2[1][2] * dL_dc;
    }
    float dT[2][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dT[0][k] = 2.f * ST0[k] * dL_da + ST1[k] * dL_db;
        dT[1][k] = 2.f * ST1[k] * dL_dc + ST0[k] * dL_db;
    }
    const float dJ00 = V.m[0] * dT[0][0] + V.m[4] * dT[0][1] + V.m[8] * dT[0][2];
    const float dJ02 = V.m[2] * dT[0][0] + V.m[6] * dT[0][1] + V.m[10] * dT[0][2];
    const float dJ11 = V.m[1] * dT[1][0] + V.m[5] * dT[1][1] + V.m[9] * dT[1][2];
    const float dJ12 = V.m[2] * dT[1][0] + V.m[6] * dT[1][1] + V.m[10] * dT[1][2];
    const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = (xin ? 1.f : 0.f) * (-focal_x * tz2 * dJ02);
    const float dty = (yin ? 1.f : 0.f) * (-focal_y * tz2 * dJ12);
    const float dtz = -focal_x * tz2 * dJ00 - focal_y * tz2 * dJ11 + (2.f * focal_x * t.x) * tz3 * dJ02 + (2.f * focal_y * t.y) * tz3 * dJ12;
#pragma unroll
    for (int j = 0; j < 3; j++) dmean[j] = V.m[4 * j] * dtx + V.m[4 * j + 1] * dty + V.m[4 * j + 2] * dtz;
    // screen-space mean (NDC-scaled gradient) -> 3D mean
    const float4 mh = xform4x4(m, PJ);
    const float mw = 1.f / (mh.w + 0.0000001f);
    const float mul1 = mh.x * mw * mw, mul2 = mh.y * mw * mw;
#pragma unroll
    for (int j = 0; j < 3; j++)
        dmean[j] += (PJ.m[4 * j] * mw - PJ.m[4 * j + 3] * mul1) * g2x + (PJ.m[4 * j + 1] * mw - PJ.m[4 * j + 3] * mul2) * g2y;
    // depth output -> 3D mean
    const float mul3 = V.m[2] * m.x + V.m[6] * m.y + V.m[10] * m.z + V.m[14];
#pragma unroll
    for (int j = 0; j < 3; j++) dmean[j] += (V.m[4 * j + 2] - V.m[4 * j + 3] * mul3) * g_depth;
}

// cov3D gradient -> scale and (unnormalised-quaternion) rotation gradients.  dscale_mod: the dependency's backward differentiates
// Sigma = R diag(mod s)^2 R^T w.r.t. (mod s) and returns that as dL/dscale -- no `mod` factor (dscale_mod = 1, the default, identical
// to the wheel); the exact derivative multiplies by mod (dscale_mod = scale_modifier, C3D_GS_FLAG_EXACT_DSCALE in the settings).
__device__ __forceinline__ void bwd_cov_to_scale_rot(const float dcov[6], const float3 sc, const float4 q, float scale_modifier, float dscale_mod, float gs3[3], float4& dq) {
    float R[3][3];
    quat_to_R(q, R);
    const float s[3] = {scale_modifier * sc.x, scale_modifier * sc.y, scale_modifier * sc.z};
    const float Gm[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]}, {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]}, {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
    float dM[3][3], dR[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) dM[i][k] = 2.f * (Gm[i][0] * R[0][k] + Gm[i][1] * R[1][k] + Gm[i][2] * R[2][k]) * s[k];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        gs3[k] = dscale_mod * (dM[0][k] * R[0][k] + dM[1][k] * R[1][k] + dM[2][k] * R[2][k]);
#pragma unroll
        for (int i = 0; i < 3; i++) dR[i][k] = dM[i][k] * s[k];
    }
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    dq.x = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
    dq.y = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2.f * x * dR[2][2]);
    dq.z = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
    dq.w = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
}

// ------------------------------------------------------------------------------------------
// A8 preprocess backward: one lane per Gaussian, pure streaming.
// ------------------------------------------------------------------------------------------
// RAW: inputs are the raw parameters (see k_preprocess) and the outputs are gradients w.r.t. them: the exp / sigmoid / normalize
// backward passes of the GaussianModel accessors are applied here; dL/dSH goes to the split dL_dsh (= d f_dc) / dL_df_rest pair.
// ACC: add into the existing contents of the parameter-gradient outputs (view loops) instead of overwriting.
template <bool STAGED, bool RAW, bool ACC, int REST3 = SH_REST>
__global__ void __launch_bounds__(256) k_preprocess_bwd(GsParams p, GsGeom g, const int* __restrict__ radii, const float* __restrict__ means3D,
                                                         const float* __restrict__ shs, const float* __restrict__ f_rest, const float* __restrict__ colors_precomp,
                                                         const float* __restrict__ scales, const float* __restrict__ rotations,
                                                         const float* __restrict__ cov3D_precomp, const float4* __restrict__ pairgrad, const uint8_t* __restrict__ pvalid,
                                                         float* __restrict__ dL_dmean2D, float* __restrict__ dL_dcolors,
                                                         float* __restrict__ dL_dopacity, float* __restrict__ dL_dmeans3D,
                                                         float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh, float* __restrict__ dL_df_rest,
                                                         float* __restrict__ dL_dscales, float* __restrict__ dL_drots, uint32_t cap) {
    extern __shared__ float sh_lds[];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * blockDim.x;
    const int gcount = min((int)blockDim.x, p.N - (int)g0);
    float* shl = sh_lds + threadIdx.x * SH_ROW;
    if (STAGED) {
        if (RAW) sh_stage_in_split<REST3>(shs, f_rest, g0, gcount, sh_lds);
        else     sh_stage_in(shs, g0, gcount, sh_lds);
        __syncthreads();
    }
    const bool culled = idx < p.N && radii[idx] <= 0;
    if (STAGED && (idx >= p.N || culled)) {
        if (idx < p.N) {
#pragma unroll
            for (int k = 0; k < SH_M3; k++) shl[k] = 0.f;
        }
    }
    if (idx < p.N && culled) {
        // culled: this kernel owns its outputs (callers allocate them uninitialised); with ACC the parameter gradients just stay as they are
        dL_dmean2D[3 * idx] = 0.f; dL_dmean2D[3 * idx + 1] = 0.f; dL_dmean2D[3 * idx + 2] = 0.f;
        if (dL_dcolors) { dL_dcolors[3 * idx] = 0.f; dL_dcolors[3 * idx + 1] = 0.f; dL_dcolors[3 * idx + 2] = 0.f; }
        if (!ACC) {
            dL_dopacity[idx] = 0.f;
            dL_dmeans3D[3 * idx] = 0.f; dL_dmeans3D[3 * idx + 1] = 0.f; dL_dmeans3D[3 * idx + 2] = 0.f;
            if (dL_dcov3D) {
#pragma unroll
                for (int i = 0; i < 6; i++) dL_dcov3D[6 * idx + i] = 0.f;
            }
            if (!colors_precomp && !STAGED) {
                float* dsh = dL_dsh + (size_t)idx * p.M * 3;
                for (int k = 0; k < 3 * p.M; k++) dsh[k] = 0.f;
            }
            if (!cov3D_precomp) {
                dL_dscales[3 * idx] = 0.f; dL_dscales[3 * idx + 1] = 0.f; dL_dscales[3 * idx + 2] = 0.f;
                *reinterpret_cast<float4*>(dL_drots + 4 * idx) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    if (idx < p.N && !culled) {
    // sum this Gaussian's (tile, splat) records: contiguous in emit order, fixed order -> deterministic
    float pr[GS_PAIR_FLOATS];
#pragma unroll
    for (int q = 0; q < GS_PAIR_FLOATS; q++) pr[q] = 0.f;
    {
        const uint32_t cnt = g.tiles[idx];
        const uint32_t e0 = g.rbase[idx], e1 = min(e0 + cnt, cap);   // cap: capacity of the pair buffers (overflow is reported, never read)
        for (uint32_t e = e0; e < e1; e++) {
            if (!pvalid[e]) continue;
            const float4 v0 = pairgrad[(size_t)e * 3], v1 = pairgrad[(size_t)e * 3 + 1], v2 = pairgrad[(size_t)e * 3 + 2];
            pr[0] += v0.x; pr[1] += v0.y; pr[2] += v0.z; pr[3] += v0.w;
            pr[4] += v1.x; pr[5] += v1.y; pr[6] += v1.z; pr[7] += v1.w;
            pr[8] += v2.x; pr[10] += v2.z;
        }
    }
    const float gcol[3] = {pr[0], pr[2], pr[1]};          // record rows: c0, c2, c1
    const float g_depth = pr[3];
    // moments -> gradients of the projected mean (NDC-scaled), conic and opacity
    const float4 q0 = g.rec0[GS_REC(idx)], q1 = g.rec1[GS_REC(idx)];
    const float cA = q0.z, cB = q0.w, cC = q1.x, opac = q1.y;
    const float m0 = pr[4], m1y = pr[5], m1x = pr[6], m2xx = pr[7], m2xy = pr[8], m2yy = pr[10];
    if (dL_dcolors) { dL_dcolors[3 * idx] = gcol[0]; dL_dcolors[3 * idx + 1] = gcol[1]; dL_dcolors[3 * idx + 2] = gcol[2]; }
    {
        float go = (opac > 0.f) ? m0 / opac : 0.f;
        if (RAW) go *= opac * (1.f - opac);                      // sigmoid'
        dL_dopacity[idx] = ACC ? dL_dopacity[idx] + go : go;
    }

    const Mat16 V = load_mat16(p.view), PJ = load_mat16(p.proj);
    const float3 m = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    float c3[6];
    float3 sc = make_float3(0, 0, 0);
    float4 q = make_float4(1, 0, 0, 0);
    float qnorm = 1.f;
    if (cov3D_precomp) {
#pragma unroll
        for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * idx + i];
    } else {
        sc = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
        q = *reinterpret_cast<const float4*>(rotations + 4 * idx);
        if (RAW) {
            sc = make_float3(expf(sc.x), expf(sc.y), expf(sc.z));
            qnorm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
            const float inv = 1.f / qnorm;
            q = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
        }
        cov3d_from_scale_rot(sc, p.scale_modifier, q, c3);
    }
    float g2x, g2y, dmean[3], dcov[6];
    bwd_geom_chain(m, c3, V, PJ, p.tanfovx, p.tanfovy, p.focal_x, p.focal_y, p.W, p.H, cA, cB, cC, m1x, m1y, m2xx, m2xy, m2yy, g_depth, g2x, g2y, dmean, dcov);
    dL_dmean2D[3 * idx] = g2x; dL_dmean2D[3 * idx + 1] = g2y; dL_dmean2D[3 * idx + 2] = 0.f;
    if (dL_dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dcov3D[6 * idx + i] = dcov[i];
    }
    // colour -> SH coefficients and view direction
    if (!colors_precomp) {
        const float3 cp_ = load_vec3_const(p.campos);
        const float vx = m.x - cp_.x, vy = m.y - cp_.y, vz = m.z - cp_.z;
        const float s2 = vx * vx + vy * vy + vz * vz;
        const float len = sqrtf(s2);
        const float dxn = vx / len, dyn = vy / len, dzn = vz / len;
        const uint8_t cl = g.clamped[idx];
        const float dR0 = (cl & 1) ? 0.f : gcol[0], dR1 = (cl & 2) ? 0.f : gcol[1], dR2 = (cl & 4) ? 0.f : gcol[2];
        const float* shg = shs + (size_t)idx * p.M * 3;
        float* dshg = dL_dsh + (size_t)idx * p.M * 3;
        float dd0 = 0.f, dd1 = 0.f, dd2 = 0.f;
        // in the staged variant each lane overwrites its own LDS row in place: coefficient in, gradient out
#define GS_BWD_TERM(k, Bk, dBx, dBy, dBz)                                                                        \
    {                                                                                                            \
        const float b_ = (Bk);                                                                                   \
        float w_;                                                                                                \
        if (STAGED) {                                                                                            \
            w_ = shl[3 * (k)] * dR0 + shl[3 * (k) + 1] * dR1 + shl[3 * (k) + 2] * dR2;                           \
            shl[3 * (k)] = b_ * dR0; shl[3 * (k) + 1] = b_ * dR1; shl[3 * (k) + 2] = b_ * dR2;                   \
        } else {                                                                                                 \
            w_ = shg[3 * (k)] * dR0 + shg[3 * (k) + 1] * dR1 + shg[3 * (k) + 2] * dR2;                           \
            dshg[3 * (k)] = b_ * dR0; dshg[3 * (k) + 1] = b_ * dR1; dshg[3 * (k) + 2] = b_ * dR2;                \
        }                                                                                                        \
        dd0 += (dBx) * w_; dd1 += (dBy) * w_; dd2 += (dBz) * w_;                                                 \
    }
        SH_FOREACH(p.deg, dxn, dyn, dzn, GS_BWD_TERM);
#undef GS_BWD_TERM
        const int nc = sh_ncoef(p.deg);
        if (STAGED) { for (int k = 3 * nc; k < SH_M3; k++) shl[k] = 0.f; }
        else        { for (int k = 3 * nc; k < 3 * p.M; k++) dshg[k] = 0.f; }   // coefficients above the active degree
        const float inv32 = 1.f / sqrtf(s2 * s2 * s2);
        dmean[0] += ((s2 - vx * vx) * dd0 - vy * vx * dd1 - vz * vx * dd2) * inv32;
        dmean[1] += (-vx * vy * dd0 + (s2 - vy * vy) * dd1 - vz * vy * dd2) * inv32;
        dmean[2] += (-vx * vz * dd0 - vy * vz * dd1 + (s2 - vz * vz) * dd2) * inv32;
    }
    if (ACC) { dmean[0] += dL_dmeans3D[3 * idx]; dmean[1] += dL_dmeans3D[3 * idx + 1]; dmean[2] += dL_dmeans3D[3 * idx + 2]; }
    dL_dmeans3D[3 * idx] = dmean[0]; dL_dmeans3D[3 * idx + 1] = dmean[1]; dL_dmeans3D[3 * idx + 2] = dmean[2];

    // cov3D -> scale, rotation (d/dscale as the dependency returns it unless the settings carry C3D_GS_FLAG_EXACT_DSCALE)
    if (!cov3D_precomp) {
        float gs3[3];
        float4 dq;
        bwd_cov_to_scale_rot(dcov, sc, q, p.scale_modifier, p.dscale_mod, gs3, dq);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float gs_ = gs3[k];
            if (RAW) gs_ *= (k == 0 ? sc.x : (k == 1 ? sc.y : sc.z));          // exp'
            dL_dscales[3 * idx + k] = ACC ? dL_dscales[3 * idx + k] + gs_ : gs_;
        }
        if (RAW) {   // through q = raw / |raw|: (g - q (q.g)) / |raw|
            const float dot = dq.x * q.x + dq.y * q.y + dq.z * q.z + dq.w * q.w, inv = 1.f / qnorm;
            dq = make_float4((dq.x - q.x * dot) * inv, (dq.y - q.y * dot) * inv, (dq.z - q.z * dot) * inv, (dq.w - q.w * dot) * inv);
        }
        if (ACC) { const float4 o4 = *reinterpret_cast<const float4*>(dL_drots + 4 * idx); dq.x += o4.x; dq.y += o4.y; dq.z += o4.z; dq.w += o4.w; }
        *reinterpret_cast<float4*>(dL_drots + 4 * idx) = dq;
    }
    }   // visible Gaussian
    if (STAGED) {
        __syncthreads();
        if (RAW) sh_stage_out_split<ACC, REST3>(dL_dsh, dL_df_rest, g0, gcount, sh_lds);
        else     sh_stage_out(dL_dsh, g0, gcount, sh_lds);
    }
}

// ------------------------------------------------------------------------------------------
// A8 over all the views of a training step in ONE pass (fused step, raw parameters, SH degree storage 16): a lane owns a Gaussian, walks the
// views, sums that view's pair records and applies the view's chain rule, and writes every parameter gradient exactly once.  Against V
// launches of k_preprocess_bwd<.,.,ACC> this removes (V-1) read-modify-write sweeps over the 236 B/Gaussian gradient set and (V-1) reads of
// the 192 B/Gaussian SH coefficients.  View-independent work (exp / normalize, cov3D, cov3D -> scale / quaternion chain, activation
// derivatives) is done once, on the summed covariance gradient (the chain is linear in it).
// Two kernels (one kernel holding 48 SH gradient sums per lane next to the whole geometric chain needed 256 VGPRs: two waves per SIMD, ~2.3 TB/s; round 2):
// (G) the geometric chain, which needs no LDS and ~100 VGPRs, hands the masked colour gradient of every view over in a 12 B/Gaussian/view array;
// (S) the SH part keeps the coefficients in LDS and only the 48 sums + a direction in registers.
// ------------------------------------------------------------------------------------------
template <bool ACC, int CHUNK, int MINB>
__global__ void __launch_bounds__(256, MINB) k_bwd_views_geom(GsParams p, GsBwdViews vs, const float* __restrict__ means3D, const float* __restrict__ scales,
                                                          const float* __restrict__ rotations, float* __restrict__ dL_dopacity, float* __restrict__ dL_dmeans3D,
                                                          float* __restrict__ dL_dscales, float* __restrict__ dL_drots, uint32_t cap, int first, int last) {
    const int idx = first + blockIdx.x * blockDim.x + threadIdx.x;   // Gaussians [first, last): the whole cloud, or one range of a chunked pass
    if (idx >= last) return;
    const float3 m = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    float3 sc = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
    float4 q = *reinterpret_cast<const float4*>(rotations + 4 * idx);
    sc = make_float3(expf(sc.x), expf(sc.y), expf(sc.z));
    const float qnorm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    {
        const float inv = 1.f / qnorm;
        q = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    }
    float c3[6];
    cov3d_from_scale_rot(sc, p.scale_modifier, q, c3);
    float dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dopac = 0.f;
    // The walk over views is a chain of dependent gathers per lane (radius / record span -> valid bytes -> records -> splat record): the span of
    // the NEXT view is fetched while this one is worked on, the valid bytes and the records of four pairs are in flight together, and a Gaussian
    // that no pixel of the view blended (no valid record: roughly every other one at the BASELINE workload) skips the splat record, the two
    // matrices and the whole geometric chain -- all its sums are zero.
    // (a culled Gaussian has tiles == 0, a visible one whose alpha box misses every tile too: the tile count alone says whether there is anything to walk -- the radius
    //  is not read, one load less at the head of every view's dependent chain)
    uint32_t n_cnt = vs.v[0].tiles[idx], n_e0 = vs.v[0].rbase[idx];
    for (int v = 0; v < vs.V; v++) {
        const GsBwdView& vw = vs.v[v];
        float* d2 = vw.dmean2D + 3 * (size_t)idx;
        float* gc = vw.gcol + 3 * (size_t)idx;
        const uint32_t cnt = n_cnt, e0 = n_e0;
        if (v + 1 < vs.V) { const GsBwdView& nx = vs.v[v + 1]; n_cnt = nx.tiles[idx]; n_e0 = nx.rbase[idx]; }
        float pr[GS_PAIR_FLOATS];
#pragma unroll
        for (int k = 0; k < GS_PAIR_FLOATS; k++) pr[k] = 0.f;
        bool any = false;
        if (cnt > 0) {
            const uint32_t e1 = min(e0 + cnt, cap);
            for (uint32_t e = e0; e < e1; e += CHUNK) {
                uint8_t pv[CHUNK];
#pragma unroll
                for (int i = 0; i < CHUNK; i++) pv[i] = (e + i < e1) ? vw.pvalid[e + i] : (uint8_t)0;
                float4 r0[CHUNK], r1[CHUNK], r2[CHUNK];
#pragma unroll
                for (int i = 0; i < CHUNK; i++) {
                    r0[i] = r1[i] = r2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (pv[i]) { const float4* rp = vw.pairgrad + (size_t)(e + i) * 3; r0[i] = rp[0]; r1[i] = rp[1]; r2[i] = rp[2]; }
                }
#pragma unroll
                for (int i = 0; i < CHUNK; i++) {   // adding the zeros of an invalid slot changes nothing (x + 0 = x; the sums start at +0)
                    any = any || pv[i];
                    pr[0] += r0[i].x; pr[1] += r0[i].y; pr[2] += r0[i].z; pr[3] += r0[i].w;
                    pr[4] += r1[i].x; pr[5] += r1[i].y; pr[6] += r1[i].z; pr[7] += r1[i].w;
                    pr[8] += r2[i].x; pr[10] += r2[i].z;
                }
            }
        }
        if (!any) { d2[0] = 0.f; d2[1] = 0.f; d2[2] = 0.f; gc[0] = 0.f; gc[1] = 0.f; gc[2] = 0.f; continue; }
        {   // colour gradient, zeroed where the forward clamped the channel at 0 (record rows: c0, c2, c1)
            const uint8_t cl = vw.clamped[idx];
            gc[0] = (cl & 1) ? 0.f : pr[0]; gc[1] = (cl & 2) ? 0.f : pr[2]; gc[2] = (cl & 4) ? 0.f : pr[1];
        }
        const float4 q0 = vw.rec0[GS_REC(idx)], q1 = vw.rec1[GS_REC(idx)];
        const float opac = q1.y;
        {
            const float go = (opac > 0.f) ? pr[4] / opac : 0.f;
            dopac += go * opac * (1.f - opac);
        }
        const Mat16 V = load_mat16(vw.view), PJ = load_mat16(vw.proj);
        float g2x, g2y, dm[3], dc[6];
        bwd_geom_chain(m, c3, V, PJ, vw.tanfovx, vw.tanfovy, vw.focal_x, vw.focal_y, p.W, p.H, q0.z, q0.w, q1.x, pr[6], pr[5], pr[7], pr[8], pr[10], pr[3], g2x, g2y, dm, dc);
        d2[0] = g2x; d2[1] = g2y; d2[2] = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) dcov[i] += dc[i];
#pragma unroll
        for (int j = 0; j < 3; j++) dmean[j] += dm[j];
    }
    float gs3[3];
    float4 dq;
    bwd_cov_to_scale_rot(dcov, sc, q, p.scale_modifier, p.dscale_mod, gs3, dq);
    gs3[0] *= sc.x; gs3[1] *= sc.y; gs3[2] *= sc.z;
    {
        const float dot = dq.x * q.x + dq.y * q.y + dq.z * q.z + dq.w * q.w, inv = 1.f / qnorm;
        dq = make_float4((dq.x - q.x * dot) * inv, (dq.y - q.y * dot) * inv, (dq.z - q.z * dot) * inv, (dq.w - q.w * dot) * inv);
    }
    if (ACC) {
        dopac += dL_dopacity[idx];
#pragma unroll
        for (int j = 0; j < 3; j++) { dmean[j] += dL_dmeans3D[3 * idx + j]; gs3[j] += dL_dscales[3 * idx + j]; }
        const float4 o4 = *reinterpret_cast<const float4*>(dL_drots + 4 * idx);
        dq.x += o4.x; dq.y += o4.y; dq.z += o4.z; dq.w += o4.w;
    }
    dL_dopacity[idx] = dopac;
#pragma unroll
    for (int j = 0; j < 3; j++) { dL_dmeans3D[3 * idx + j] = dmean[j]; dL_dscales[3 * idx + j] = gs3[j]; }
    *reinterpret_cast<float4*>(dL_drots + 4 * idx) = dq;
}

struct GsShViews { int V; const float* campos[GS_MAX_BWD_VIEWS]; const float* gcol[GS_MAX_BWD_VIEWS]; };
// dL/dSH over all views + the view-direction term of dL/dmean (added to what k_bwd_views_geom wrote)
template <bool ACC, int REST3>
__global__ void __launch_bounds__(256) k_bwd_views_sh(int first, int last, int deg_in, GsShViews vs, const float* __restrict__ means3D, const float* __restrict__ f_dc,
                                                        const float* __restrict__ f_rest, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_df_dc,
                                                        float* __restrict__ dL_df_rest) {
    extern __shared__ float sh_lds[];
    const int N = last;                                              // Gaussians [first, last); `first` is a multiple of 4 (16-byte rows of f_rest)
    const int idx = first + blockIdx.x * blockDim.x + threadIdx.x;
    const size_t g0 = (size_t)first + (size_t)blockIdx.x * blockDim.x;
    const int gcount = min((int)blockDim.x, N - (int)g0);
    float* shl = sh_lds + threadIdx.x * SH_ROW;
    sh_stage_in_split<REST3>(f_dc, f_rest, g0, gcount, sh_lds);
    __syncthreads();
    const int deg = min(deg_in, REST3 >= 45 ? 3 : (REST3 >= 24 ? 2 : (REST3 >= 9 ? 1 : 0)));      // never above the storage's degree: the unused bands compile out
    float gsh[SH_M3];
#pragma unroll
    for (int k = 0; k < SH_M3; k++) gsh[k] = 0.f;
    if (idx < N) {
        const float3 m = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        float dmx = 0.f, dmy = 0.f, dmz = 0.f;
        for (int v = 0; v < vs.V; v++) {
            const float* gc = vs.gcol[v] + 3 * (size_t)idx;
            const float dR0 = gc[0], dR1 = gc[1], dR2 = gc[2];
            if (dR0 == 0.f && dR1 == 0.f && dR2 == 0.f) continue;
            const float3 cp = load_vec3_const(vs.campos[v]);
            const float vx = m.x - cp.x, vy = m.y - cp.y, vz = m.z - cp.z;
            const float s2 = vx * vx + vy * vy + vz * vz;
            const float len = sqrtf(s2);
            const float dxn = vx / len, dyn = vy / len, dzn = vz / len;
            float dd0 = 0.f, dd1 = 0.f, dd2 = 0.f;
#define GS_BWDS_TERM(k, Bk, dBx, dBy, dBz)                                                              \
    {                                                                                                   \
        const float b_ = (Bk);                                                                          \
        const float w_ = shl[3 * (k)] * dR0 + shl[3 * (k) + 1] * dR1 + shl[3 * (k) + 2] * dR2;          \
        gsh[3 * (k)] += b_ * dR0; gsh[3 * (k) + 1] += b_ * dR1; gsh[3 * (k) + 2] += b_ * dR2;           \
        dd0 += (dBx) * w_; dd1 += (dBy) * w_; dd2 += (dBz) * w_;                                        \
    }
            SH_FOREACH(deg, dxn, dyn, dzn, GS_BWDS_TERM);
#undef GS_BWDS_TERM
            const float inv32 = 1.f / sqrtf(s2 * s2 * s2);
            dmx += ((s2 - vx * vx) * dd0 - vy * vx * dd1 - vz * vx * dd2) * inv32;
            dmy += (-vx * vy * dd0 + (s2 - vy * vy) * dd1 - vz * vy * dd2) * inv32;
            dmz += (-vx * vz * dd0 - vy * vz * dd1 + (s2 - vz * vz) * dd2) * inv32;
        }
        dL_dmeans3D[3 * idx] += dmx; dL_dmeans3D[3 * idx + 1] += dmy; dL_dmeans3D[3 * idx + 2] += dmz;
    }
#pragma unroll
    for (int k = 0; k < 3 + REST3; k++) shl[k] = gsh[k];
    __syncthreads();
    sh_stage_out_split<ACC, REST3>(dL_df_dc, dL_df_rest, g0, gcount, sh_lds);
}

int gs_launch_preprocess_bwd_views(const GsParams& p0, const GsBwdViews& views, const float* means3D, const float* f_dc, const float* f_rest,
                                   const float* scaling_raw, const float* rotation_raw, float* dL_dopacity_raw, float* dL_dmeans3D, float* dL_df_dc,
                                   float* dL_df_rest, float* dL_dscaling_raw, float* dL_drotation_raw, bool accumulate, hipStream_t s, uint32_t cap, int first, int count) {
    if (p0.N == 0 || views.V == 0) return 0;
    if (count < 0) count = p0.N - first;
    if (first < 0 || count < 0 || first + count > p0.N || (first & 3)) { c3d_set_error("gs_launch_preprocess_bwd_views: bad Gaussian range [%d, %d + %d)", first, first, count); return -1; }
    if (count == 0) return 0;
    const int last = first + count;
    const dim3 grid(c3d_cdiv(count, 256)), block(256);
    const size_t lds = 256 * SH_ROW * sizeof(float);
    GsShViews sv;
    sv.V = views.V;
    for (int i = 0; i < views.V; i++) { sv.campos[i] = views.v[i].campos; sv.gcol[i] = views.v[i].gcol; }
    // four pairs in flight, four workgroups per CU (128 VGPRs): eight in flight (160 VGPRs) and two or three (96, spilling) measure the same or worse.
    // Round 5, measured and dropped (profiles/r05b_a8_variants.txt, same box, 8-view step): three workgroups per CU (136 VGPRs, no scratch) 0.589 -> 0.633 ms; a
    // two-view software pipeline (the next view's valid bytes and the span of the one after requested under this view's records: one dependent round trip per
    // view instead of two) 0.599 ms at four workgroups (36 B of scratch), 0.656 ms at three -- the walk is not bound by the depth of its dependent-load chain
#define GS_A8_GEOM(ACC_) hipLaunchKernelGGL((k_bwd_views_geom<ACC_, 4, 4>), grid, block, 0, s, p0, views, means3D, scaling_raw, rotation_raw, dL_dopacity_raw, dL_dmeans3D, dL_dscaling_raw, dL_drotation_raw, cap, first, last)
#define GS_A8_SH_ACC(R3_) hipLaunchKernelGGL((k_bwd_views_sh<true, R3_>), grid, block, lds, s, first, last, p0.deg, sv, means3D, f_dc, f_rest, dL_dmeans3D, dL_df_dc, dL_df_rest)
#define GS_A8_SH_SET(R3_) hipLaunchKernelGGL((k_bwd_views_sh<false, R3_>), grid, block, lds, s, first, last, p0.deg, sv, means3D, f_dc, f_rest, dL_dmeans3D, dL_df_dc, dL_df_rest)
    if (accumulate) {
        GS_A8_GEOM(true);
        GS_BY_SH_COEFFS(p0.M, GS_A8_SH_ACC);
    } else {
        GS_A8_GEOM(false);
        GS_BY_SH_COEFFS(p0.M, GS_A8_SH_SET);
    }
#undef GS_A8_SH_ACC
#undef GS_A8_SH_SET
#undef GS_A8_GEOM
    C3D_LAUNCH_CHECK();
    return 0;
}

int gs_launch_preprocess_bwd(const GsParams& p, const GsGeom& g, const int* radii, const float* means3D, const float* shs,
                             const float* colors_precomp, const float* scales, const float* rotations, const float* cov3D_precomp,
                             const float* pairgrad, const uint8_t* pvalid, float* dL_dmean2D, float* dL_dcolors, float* dL_dopacity,
                             float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drots, hipStream_t s, uint32_t cap) {
    if (p.N == 0) return 0;
    const bool staged = shs && !colors_precomp && p.M == 16 && ((uintptr_t)shs % 16 == 0) && ((uintptr_t)dL_dsh % 16 == 0);
    if (staged)
        hipLaunchKernelGGL((k_preprocess_bwd<true, false, false>), dim3(c3d_cdiv(p.N, 256)), dim3(256), 256 * SH_ROW * sizeof(float), s, p, g, radii, means3D, shs,
                           (const float*)nullptr, colors_precomp, scales, rotations, cov3D_precomp, (const float4*)pairgrad, pvalid, dL_dmean2D, dL_dcolors, dL_dopacity,
                           dL_dmeans3D, dL_dcov3D, dL_dsh, (float*)nullptr, dL_dscales, dL_drots, cap);
    else
        hipLaunchKernelGGL((k_preprocess_bwd<false, false, false>), dim3(c3d_cdiv(p.N, 256)), dim3(256), 0, s, p, g, radii, means3D, shs,
                           (const float*)nullptr, colors_precomp, scales, rotations, cov3D_precomp, (const float4*)pairgrad, pvalid, dL_dmean2D, dL_dcolors, dL_dopacity,
                           dL_dmeans3D, dL_dcov3D, dL_dsh, (float*)nullptr, dL_dscales, dL_drots, cap);
    C3D_LAUNCH_CHECK();
    return 0;
}
int gs_launch_preprocess_bwd_raw(const GsParams& p, const GsGeom& g, const int* radii, const float* means3D, const float* f_dc, const float* f_rest,
                                 const float* scaling_raw, const float* rotation_raw, const float* pairgrad, const uint8_t* pvalid, float* dL_dmean2D,
                                 float* dL_dopacity_raw, float* dL_dmeans3D, float* dL_df_dc, float* dL_df_rest, float* dL_dscaling_raw,
                                 float* dL_drotation_raw, bool accumulate, hipStream_t s, uint32_t cap) {
    if (p.N == 0) return 0;
#define GS_A8_RAW(ACC_, R3_) hipLaunchKernelGGL((k_preprocess_bwd<true, true, ACC_, R3_>), dim3(c3d_cdiv(p.N, 256)), dim3(256), 256 * SH_ROW * sizeof(float), s, p, g, radii, means3D, f_dc, f_rest, \
                           (const float*)nullptr, scaling_raw, rotation_raw, (const float*)nullptr, (const float4*)pairgrad, pvalid, dL_dmean2D, (float*)nullptr,                          \
                           dL_dopacity_raw, dL_dmeans3D, (float*)nullptr, dL_df_dc, dL_df_rest, dL_dscaling_raw, dL_drotation_raw, cap)
#define GS_A8_RAW_ACC(R3_) GS_A8_RAW(true, R3_)
#define GS_A8_RAW_SET(R3_) GS_A8_RAW(false, R3_)
    if (accumulate) { GS_BY_SH_COEFFS(p.M, GS_A8_RAW_ACC); }
    else            { GS_BY_SH_COEFFS(p.M, GS_A8_RAW_SET); }
#undef GS_A8_RAW_ACC
#undef GS_A8_RAW_SET
#undef GS_A8_RAW
    C3D_LAUNCH_CHECK();
    return 0;
}
