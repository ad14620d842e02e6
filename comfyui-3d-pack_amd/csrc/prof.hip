// prof.hip -- optional per-kernel timing with HIP events recorded on the launch stream.
// Disabled by default (zero overhead: one predictable branch per launch).  bench.py turns it on to obtain
// the dominant kernel's average duration inside the timed region without an external profiler.
#include "c3d_common.h"
#include <mutex>
#include <vector>

namespace {
struct Rec { int slot; hipEvent_t e0, e1; };
const char* kNames[C3D_PROF_SLOTS] = {"gs_preprocess", "gs_depth_sort", "gs_offsets_scan", "gs_emit", "gs_tile_sort", "gs_ranges",
                                      "gs_composite_fwd", "gs_composite_bwd", "gs_preprocess_bwd", "adam", "mesh_rasterize",
                                      "mesh_interpolate", "mesh_texture", "mesh_antialias", "mesh_bwd", "other", "mesh_rasterize_bwd",
                                      "mesh_interpolate_bwd", "mesh_texture_bwd", "mesh_antialias_bwd", "msssim", "mesh_ras_tri"};
bool g_on = false;
unsigned long long g_mask = ~0ull;   // slots that are timed while profiling is on (c3d_prof_select)
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
double g_ms[C3D_PROF_SLOTS];
long long g_cnt[C3D_PROF_SLOTS];

hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void drain() {   // caller holds the lock
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.e1) == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { g_ms[r.slot] += ms; g_cnt[r.slot]++; }
        }
        g_pool.push_back(r.e0); g_pool.push_back(r.e1);
    }
    g_recs.clear();
}
}  // namespace

bool c3d_prof_on() { return g_on; }
void* c3d_prof_begin(int slot, hipStream_t s) {
    if (!g_on || !((g_mask >> slot) & 1ull)) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_recs.size() > 60000) drain();
    Rec r{slot, get_event(), get_event()};
    if (!r.e0 || !r.e1) return nullptr;
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
    return (void*)(uintptr_t)g_recs.size();
}
void c3d_prof_end(void* h, hipStream_t s) {
    if (!h) return;
    std::lock_guard<std::mutex> lk(g_mu);
    size_t i = (size_t)(uintptr_t)h - 1;
    if (i < g_recs.size()) (void)hipEventRecord(g_recs[i].e1, s);
}

extern "C" {
int c3d_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain();
    for (int i = 0; i < C3D_PROF_SLOTS; i++) { g_ms[i] = 0; g_cnt[i] = 0; }
    g_on = on != 0;
    return 0;
}
// restrict the timing to the slots whose bit is set (two event records per timed launch perturb a multi-stream schedule: bench.py times only the
// dominant kernel inside its timed region and everything in a separate pass); stays in force until changed, ~0 = all
int c3d_prof_select(unsigned long long mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = mask;
    return 0;
}
int c3d_prof_slots(void) { return C3D_PROF_SLOTS; }
const char* c3d_prof_name(int slot) { return (slot >= 0 && slot < C3D_PROF_SLOTS) ? kNames[slot] : ""; }
int c3d_prof_read(int slot, double* total_ms, long long* launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain();
    if (slot < 0 || slot >= C3D_PROF_SLOTS) return -1;
    if (total_ms) *total_ms = g_ms[slot];
    if (launches) *launches = g_cnt[slot];
    return 0;
}
}
