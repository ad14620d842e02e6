// lanes.hip -- the per-device pool of library-owned HIP streams behind C3dLanes (c3d_common.h).
#include "c3d_common.h"
#include <mutex>

namespace {
struct LanePool {
    bool init = false;
    hipStream_t st[C3D_MAX_LANES - 1];
    hipEvent_t fork, join[C3D_MAX_LANES - 1];
    std::mutex busy;              // held from a call's fork to its join: two host threads (or two step objects) on one device take turns instead of re-recording each other's events
};
LanePool g_pools[16];
std::mutex g_pool_mu;
int lane_pool(LanePool** out) {
    int dev = 0;
    C3D_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) { c3d_set_error("c3d: device ordinal %d out of range", dev); return -1; }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    LanePool& lp = g_pools[dev];
    if (!lp.init) {
        C3D_CHECK(hipEventCreateWithFlags(&lp.fork, hipEventDisableTiming));
        for (int i = 0; i < C3D_MAX_LANES - 1; i++) {
            C3D_CHECK(hipStreamCreateWithFlags(&lp.st[i], hipStreamNonBlocking));
            C3D_CHECK(hipEventCreateWithFlags(&lp.join[i], hipEventDisableTiming));
        }
        lp.init = true;
    }
    *out = &lp;
    return 0;
}
}  // namespace

int C3dLanes::fork(hipStream_t caller, int lanes, int n_chains) {
    if (pool_) { c3d_set_error("C3dLanes::fork: already forked"); return -1; }
    s[0] = caller;
    L = lanes < n_chains ? lanes : n_chains;
    if (L > C3D_MAX_LANES) L = C3D_MAX_LANES;
    if (L <= 1) { L = 1; return 0; }
    LanePool* lp = nullptr;
    if (lane_pool(&lp)) { L = 1; return -1; }
    lp->busy.lock();
    pool_ = lp;
    bool ok = hipEventRecord(lp->fork, caller) == hipSuccess;
    for (int l = 1; l < L && ok; l++) {
        s[l] = lp->st[l - 1];
        ok = hipStreamWaitEvent(s[l], lp->fork, 0) == hipSuccess;
    }
    if (!ok) {      // nothing has been queued on the lanes yet: give the pool back
        lp->busy.unlock(); pool_ = nullptr; L = 1;
        c3d_set_error("lane fork failed");
        return -1;
    }
    return 0;
}

// always executed (destructor), so the caller's stream never runs ahead of work queued on the lanes -- also after an error
int C3dLanes::join(const char* who) {
    if (!pool_) return 0;
    LanePool* lp = (LanePool*)pool_;
    int rc = 0;
    for (int l = 1; l < L; l++)
        if (hipEventRecord(lp->join[l - 1], s[l]) != hipSuccess || hipStreamWaitEvent(s[0], lp->join[l - 1], 0) != hipSuccess) {
            (void)hipDeviceSynchronize();
            c3d_set_error("%s: lane join failed", who); rc = -1;
        }
    pool_ = nullptr;
    lp->busy.unlock();
    return rc;
}
