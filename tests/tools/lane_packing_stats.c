/* lane_packing_stats.c -- how full are the lanes of the compositing kernels, and what would packing half-quadrants / 16-lane strips buy?  
 * Test-side analysis over the ORACLE's forward state (tests/tools/lane_packing_stats.py drives it): for every 16x16 tile, every list position k and each of its four
 * 8x8 quadrants (= one wave of k_composite_bwd) the set of pixels that BLEND the splat -- k < n_contrib[pixel], power <= 0, opacity * exp(power) >= 1/255: the lanes
 * the backward kernel's `am` mask enables -- is formed as a 64-bit lane mask (lane = 8 * row + column, as in the kernel), and the walk is priced four ways:
 *   passes_now      one pass per (quadrant, pair) with a non-empty mask                               (what k_composite_bwd walks today)
 *   passes_adjacent today's order, but two list-consecutive walked pairs whose masks live in opposite 32-lane halves share a pass   (the verdict's proposal)
 *   passes_2stream  rows 0-3 and rows 4-7 as two independent streams, each walking only the pairs that touch it: max(n_top, n_bottom) per quadrant
 *   passes_4stream  the four 16-lane strips (two pixel rows each) as independent streams: max over the strips
 * out[0..7] = passes_now, lanes_on (sum of popcounts), passes_adjacent, passes_2stream, passes_4stream, pairs_one_half_only, quadrants, pairs_reached */
#include <math.h>
#include <stdint.h>
#include <string.h>
void lane_packing_stats(int W, int H, int gx, int gy, const float *xy, const float *conic_opacity, const uint32_t *point_list, const uint32_t *ranges,
                        const uint32_t *n_contrib, int nthreads, double *out) {
    double s_now = 0, s_lanes = 0, s_adj = 0, s_2 = 0, s_4 = 0, s_half = 0, s_quads = 0, s_reached = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : s_now, s_lanes, s_adj, s_2, s_4, s_half, s_quads, s_reached)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int q = 0; q < 4; q++) {
            const int QX = tx * 16 + (q & 1) * 8, QY = ty * 16 + (q >> 1) * 8;
            uint32_t last[64], upto = 0;
            for (int l = 0; l < 64; l++) {
                const int px = QX + (l & 7), py = QY + (l >> 3);
                last[l] = (px < W && py < H) ? n_contrib[(size_t)py * W + px] : 0;
                if (last[l] > upto) upto = last[l];
            }
            if (r0 + upto > r1) upto = r1 - r0;
            s_quads += 1;
            s_reached += upto;
            long n_now = 0, n_adj = 0, n_top = 0, n_bot = 0, n_strip[4] = {0, 0, 0, 0};
            int pending = 0;      /* adjacent packing, walking back to front as the kernel does: 1 = an unpaired top-only pass is open, 2 = a bottom-only one */
            for (long k = (long)upto - 1; k >= 0; k--) {
                const uint32_t g = point_list[r0 + k];
                const float X = xy[2 * g], Y = xy[2 * g + 1];
                const float *co = conic_opacity + 4 * g;
                uint64_t m = 0;
                for (int l = 0; l < 64; l++) {
                    if ((uint32_t)k >= last[l]) continue;
                    const float dx = X - (float)(QX + (l & 7)), dy = Y - (float)(QY + (l >> 3));
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.f) continue;
                    if (co[3] * expf(power) < 1.f / 255.f) continue;
                    m |= 1ull << l;
                }
                if (!m) continue;
                n_now++;
                s_lanes += __builtin_popcountll(m);
                const int top = (m & 0xFFFFFFFFull) != 0, bot = (m >> 32) != 0;
                n_top += top; n_bot += bot;
                for (int s = 0; s < 4; s++) n_strip[s] += ((m >> (16 * s)) & 0xFFFFull) != 0;
                if (top && bot) { n_adj++; pending = 0; }
                else {
                    s_half += 1;
                    const int mine = top ? 1 : 2;
                    if (pending && pending != mine) pending = 0;          /* rides in the open pass of the opposite half */
                    else { n_adj++; pending = mine; }
                }
            }
            s_now += n_now; s_adj += n_adj; s_2 += n_top > n_bot ? n_top : n_bot;
            long mx = 0;
            for (int s = 0; s < 4; s++) if (n_strip[s] > mx) mx = n_strip[s];
            s_4 += mx;
        }
    }
    out[0] = s_now; out[1] = s_lanes; out[2] = s_adj; out[3] = s_2; out[4] = s_4; out[5] = s_half; out[6] = s_quads; out[7] = s_reached;
}

/* ---- the FORWARD kernel's walk (round 5, second question): k_composite_fwd_w walks, per quadrant, every list entry whose alpha >= 1/255 ellipse reaches the quadrant's
 * 8x8 RECTANGLE (gs_rect_hit: the exact minimum of the quadratic form over the continuous rectangle), until all of its pixels have saturated.  How many of those passes blend
 * into no pixel at all, and which cheaper-to-decide classes do they fall in?
 *   fwd_passes       rectangle hits walked while some pixel of the quadrant is alive
 *   blended          ... that blend into at least one pixel (= what the backward kernel walks)
 *   hit_alive_box    rectangle hits if the rectangle were the bounding box of the ALIVE pixels at the start of the 64-entry chunk (a pixel dies when it saturates)
 *   hit_lattice      rectangle hits that reach alpha >= 1/255 on at least one pixel CENTRE of the quadrant, dead or alive (the continuous test admits slivers between centres)
 * out[0..4] = fwd_passes, blended, hit_alive_box, hit_lattice, hit_lattice_alive (centre of an alive pixel, alive as of the chunk start) */
static float qmin_rect(float A, float B, float C, float X, float Y, float xl, float xh, float yl, float yh) {
    float cx = X < xl ? xl : (X > xh ? xh : X), cy = Y < yl ? yl : (Y > yh ? yh : Y);
    if (cx == X && cy == Y) return 0.f;
    float best = 3.0e38f;
    for (int e = 0; e < 2; e++) { float dx = (e ? xh : xl) - X, dy = -B * dx / C; if (dy < yl - Y) dy = yl - Y; if (dy > yh - Y) dy = yh - Y; float q = A * dx * dx + 2.f * B * dx * dy + C * dy * dy; if (q < best) best = q; }
    for (int e = 0; e < 2; e++) { float dy = (e ? yh : yl) - Y, dx = -B * dy / A; if (dx < xl - X) dx = xl - X; if (dx > xh - X) dx = xh - X; float q = A * dx * dx + 2.f * B * dx * dy + C * dy * dy; if (q < best) best = q; }
    return best;
}
static int rect_hit(const float *co, float X, float Y, float rx0, float ry0, float w, float h) {
    const float A = co[0], B = co[1], C = co[2], o = co[3];
    const float t = 255.f * o;
    if (!(t > 1.f)) return 0;
    const float det = A * C - B * B;
    const float sxx = C / det, syy = A / det;                       /* the blurred 2D covariance back from the conic */
    const float ex = sqrtf(2.f * logf(t) * sxx) * 1.0005f + 0.02f, ey = sqrtf(2.f * logf(t) * syy) * 1.0005f + 0.02f;
    if (!(X + ex >= rx0 && X - ex <= rx0 + w && Y + ey >= ry0 && Y - ey <= ry0 + h)) return 0;
    return qmin_rect(A, B, C, X, Y, rx0, rx0 + w, ry0, ry0 + h) <= 2.f * logf(t) * 1.0005f + 1e-3f;
}
void fwd_walk_stats(int W, int H, int gx, int gy, const float *xy, const float *conic_opacity, const uint32_t *point_list, const uint32_t *ranges, int nthreads, double *out) {
    double s_walk = 0, s_blend = 0, s_box = 0, s_lat = 0, s_lat_alive = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : s_walk, s_blend, s_box, s_lat, s_lat_alive)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int q = 0; q < 4; q++) {
            const int QX = tx * 16 + (q & 1) * 8, QY = ty * 16 + (q >> 1) * 8;
            float T[64];
            uint64_t alive = 0;
            for (int l = 0; l < 64; l++) { T[l] = 1.f; if (QX + (l & 7) < W && QY + (l >> 3) < H) alive |= 1ull << l; }
            for (uint32_t base = r0; base < r1 && alive; base += 64) {
                /* bounding box of the alive pixels as of the chunk start */
                int xmin = 8, xmax = -1, ymin = 8, ymax = -1;
                const uint64_t alive0 = alive;
                for (int l = 0; l < 64; l++) if ((alive0 >> l) & 1) { int x = l & 7, y = l >> 3; if (x < xmin) xmin = x; if (x > xmax) xmax = x; if (y < ymin) ymin = y; if (y > ymax) ymax = y; }
                for (uint32_t j = base; j < r1 && j < base + 64 && alive; j++) {
                    const uint32_t g = point_list[j];
                    const float X = xy[2 * g], Y = xy[2 * g + 1];
                    const float *co = conic_opacity + 4 * g;
                    if (!rect_hit(co, X, Y, (float)QX, (float)QY, 7.f, 7.f)) continue;
                    s_walk += 1;
                    s_box += rect_hit(co, X, Y, (float)(QX + xmin), (float)(QY + ymin), (float)(xmax - xmin), (float)(ymax - ymin));
                    int any = 0, lat = 0, lat_alive = 0;
                    for (int l = 0; l < 64; l++) {
                        const float dx = X - (float)(QX + (l & 7)), dy = Y - (float)(QY + (l >> 3));
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.f) continue;
                        float a = co[3] * expf(power);
                        if (a < 1.f / 255.f) continue;
                        lat = 1;
                        if ((alive0 >> l) & 1) lat_alive = 1;
                        if (!((alive >> l) & 1)) continue;
                        if (a > 0.99f) a = 0.99f;
                        const float tt = T[l] * (1.f - a);
                        if (tt < 0.0001f) { alive &= ~(1ull << l); continue; }
                        T[l] = tt; any = 1;
                    }
                    s_blend += any; s_lat += lat; s_lat_alive += lat_alive;
                }
            }
        }
    }
    out[0] = s_walk; out[1] = s_blend; out[2] = s_box; out[3] = s_lat; out[4] = s_lat_alive;
}
