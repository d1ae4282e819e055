// rbpf_api.hip -- C ABI of librbpf_hip.so (see include/rbpf_hip.h).  Host orchestration only:
// configuration checks, LUT construction, device memory, stream ordering.  All per-particle work
// runs in the gfx950 kernels of kernels_*.hip; there is no CPU compute path.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <limits>

#include "rbpf_internal.h"

using namespace rbpf;

static thread_local std::string g_create_err;

#define HIP_TRY(h, call)                                                                           \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return RBPF_EDEVICE;                                                                   \
        }                                                                                          \
    } while (0)

// Every entry point runs on the handle's device and leaves the caller's current device as it found it (a process may
// hold engines on several GPUs, and torch keeps its own notion of the current device).
struct DevGuard {
    int prev = -1; bool switched = false;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != dev) { if (hipSetDevice(dev) == hipSuccess) switched = prev >= 0; else (void)hipGetLastError(); }
    }
    ~DevGuard() { if (switched) (void)hipSetDevice(prev); }
    DevGuard(const DevGuard&) = delete; DevGuard& operator=(const DevGuard&) = delete;
};
#define ON_DEVICE(h) DevGuard dev_guard_((h)->cfg.device)

static int fail(rbpf_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}

template <typename T>
static int dev_alloc(rbpf_handle* h, T** out, size_t n) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) { h->err = std::string("hipMalloc: ") + hipGetErrorString(e); return RBPF_ENOMEM; }
    h->allocs.push_back(p);
    *out = static_cast<T*>(p);
    return RBPF_OK;
}
#define ALLOC(h, ptr, n) do { int rc_ = dev_alloc(h, &(ptr), (n)); if (rc_) return rc_; } while (0)

static int check_device_error(rbpf_handle* h) {
    int32_t e = 0;
    HIP_TRY(h, hipMemcpyAsync(&e, h->v.err, sizeof(e), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (e != 0) {
        int32_t z = 0;
        hipMemcpyAsync(h->v.err, &z, sizeof(z), hipMemcpyHostToDevice, h->stream);
        hipStreamSynchronize(h->stream);
        const char* what = e == RBPF_ENOMEM ? "tile pool or work-item queue exhausted"
                         : e == RBPF_ERANGE ? "a pose or beam left the addressable tile lattice (raise lattice_radius)"
                         : "device-side error";
        return fail(h, e, what);
    }
    return RBPF_OK;
}

template <typename T>
static int scratch(rbpf_handle* h, T** ptr, size_t* cap, size_t n) {
    if (*cap >= n) return RBPF_OK;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr; *cap = 0;
    size_t want = std::max<size_t>(n, 1024);
    HIP_TRY(h, hipMalloc((void**)ptr, want * sizeof(T)));
    *cap = want;
    return RBPF_OK;
}

extern "C" {

int rbpf_default_config(rbpf_config* c) {
    if (!c) return RBPF_EINVAL;
    memset(c, 0, sizeof(*c));
    c->n_particles = 1;          // main.py:44
    c->n_samples = 30;           // robot.py:17
    c->max_beams = 1081;
    c->tile_len_m = 40;          // hybridmap.py:68
    c->cell_size = 0.05;         // hybridmap.py:67
    c->lattice_radius = 3;
    c->pool_tiles = 0;
    c->log_odds_occ = 0.80;      // gridmap.py:20-24
    c->log_odds_nearby = 0.20;
    c->max_odds_occ = 3.0;
    c->log_odds_emp = -0.30;
    c->min_odds_emp = -3.0;
    c->quantum = 0.1;
    c->occupied_threshold = 1.0; // hybridmap.py:18
    c->max_ray_m = 15.0;         // hybridmap.py:107
    c->weight_min_range = 0.01;  // robot.py:130
    c->weight_max_range = 25.0;
    c->match_min_range = 1e-3;   // hybridmap.py:218
    c->match_max_range = 11.0;   // hybridmap.py:20
    c->resample_spread = 200.0;  // main.py:50
    c->vel_noise[0] = 0.02; c->vel_noise[1] = 0.01; c->vel_noise[2] = 0.2; c->vel_noise[3] = 0.02;  // Freid101IMUData.py:51-55
    c->device = 0;
    c->ndt_refine = 1;           // matchScanCustom.m:32-50 runs its second stage on every valid grid match
    c->seed = 42;
    return RBPF_OK;
}

static bool to_quanta(double val, double q, int& out) {
    double r = val / q;
    long n = lround(r);
    if (fabs(r - (double)n) > 1e-9 || n < -127 || n > 127) return false;
    out = (int)n;
    return true;
}

int rbpf_create(const rbpf_config* cfg, rbpf_handle** out) {
    if (!cfg || !out) return fail(nullptr, RBPF_EINVAL, "null argument");
    *out = nullptr;
    const rbpf_config& c = *cfg;
    if (c.n_particles < 1) return fail(nullptr, RBPF_EINVAL, "n_particles must be >= 1");
    if (c.n_samples < 1 || c.n_samples > 32) return fail(nullptr, RBPF_EINVAL, "n_samples must be in 1..32");
    if (c.max_beams < 1 || c.max_beams > 4095) return fail(nullptr, RBPF_EINVAL, "max_beams must be in 1..4095");
    if (c.lattice_radius < 0 || c.lattice_radius > 3) return fail(nullptr, RBPF_EINVAL, "lattice_radius must be in 0..3");
    if (c.ndt_refine < 0 || c.ndt_refine > 2) return fail(nullptr, RBPF_EINVAL, "ndt_refine must be 0, 1 or 2");
    if (!(c.cell_size > 0) || c.tile_len_m < 1) return fail(nullptr, RBPF_EINVAL, "cell_size/tile_len_m");   // gridmap.py:29
    const int dim = (int)llround((double)c.tile_len_m / c.cell_size);                                        // gridmap.py:31
    if (dim < WIN || dim > 4096 || dim % 16 != 0) return fail(nullptr, RBPF_EINVAL, "tile dimension must be a multiple of 16 in 128..4096 cells");
    if (!(c.max_ray_m > 0) || c.max_ray_m / c.cell_size + 4 >= dim)
        return fail(nullptr, RBPF_EINVAL, "max_ray_m must be shorter than one tile");
    CellConsts cc;
    if (!(c.quantum > 0) || !to_quanta(c.log_odds_occ, c.quantum, cc.occ) || !to_quanta(c.log_odds_nearby, c.quantum, cc.nearby) ||
        !to_quanta(c.log_odds_emp, c.quantum, cc.emp) || !to_quanta(c.max_odds_occ, c.quantum, cc.vmax) ||
        !to_quanta(c.min_odds_emp, c.quantum, cc.vmin))
        return fail(nullptr, RBPF_EINVAL, "log-odds constants must be integer multiples of quantum within int8");
    if (cc.emp >= 0 || cc.occ <= 0 || cc.nearby < 0 || cc.vmin > 0 || cc.vmax < 0)
        return fail(nullptr, RBPF_EINVAL, "log-odds constants have the wrong sign");
    cc.thr = (int)floor(c.occupied_threshold / c.quantum + 1e-9);

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e == hipSuccess && (c.device < 0 || c.device >= n_dev)) e = hipErrorInvalidDevice;
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, RBPF_EDEVICE, std::string("device ") + std::to_string(c.device) + ": " + hipGetErrorString(e) +
                                     " (librbpf_hip needs an MI355X; there is no CPU path)"); }
    DevGuard dev_guard_(c.device);         // the caller's current device is restored on return

    rbpf_handle* h = new rbpf_handle();
    h->cfg = c;
    memset(&h->counters, 0, sizeof(h->counters));
    DevView& v = h->v;
    memset(&v, 0, sizeof(v));
    v.P = c.n_particles; v.K = c.n_samples; v.B = 0; v.dim = dim; v.R = c.lattice_radius; v.L = 2 * v.R + 1;
    v.pool_tiles = c.pool_tiles > 0 ? c.pool_tiles : 2 * c.n_particles;
    if (v.pool_tiles < v.P) { delete h; return fail(nullptr, RBPF_EINVAL, "pool_tiles must be >= n_particles"); }
    v.cs = c.cell_size; v.tile_len = (double)c.tile_len_m;
    v.quantum = c.quantum;
    double inv = 1.0 / c.quantum;
    v.inv_quantum = fabs(inv - round(inv)) < 1e-9 ? round(inv) : 0.0;
    v.cc = cc;
    v.w_min_range = c.weight_min_range; v.w_max_range = c.weight_max_range;
    v.reach = (int)(c.max_ray_m / c.cell_size) + 3;

    // ---- global-index LUT (hybridmap.py:123,136 + gridmap.py:93) -------------------------------
    {
        const int half = v.R * dim + dim / 2 + 2;
        std::vector<uint32_t>& lut = h->h_lut;
        int g_first = INT32_MAX;
        for (int g = -half; g <= half; ++g) {
            double pos = (double)g * v.cs;                       // hybridmap.py:123
            int lat;
            if (!tile_of_coord(pos, v.tile_len, v.R, lat)) { if (g_first != INT32_MAX) break; else continue; }
            double rel = pos - (double)lat * v.tile_len;         // hybridmap.py:136
            int cidx = trunc_to_int(rel / v.cs + (double)dim / 2.0);   // gridmap.py:93
            if (cidx < 0 || cidx >= dim) { delete h; return fail(nullptr, RBPF_EINVAL, "cell index formula leaves the tile for this cell_size (the reference would raise IndexError)"); }
            if (g_first == INT32_MAX) g_first = g;
            uint32_t ent = ((uint32_t)(lat + v.R) << 16) | (uint32_t)cidx;
            if (!lut.empty() && ent < lut.back()) { delete h; return fail(nullptr, RBPF_EINVAL, "cell index formula is not monotone for this cell_size"); }
            lut.push_back(ent);
        }
        v.g_min = g_first; v.n_lut = (int)lut.size();
    }

    int rc = RBPF_OK;
    auto build = [&]() -> int {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
        for (int k = 0; k < rbpf_handle::N_KERN; ++k)
            for (int e = 0; e < 2; ++e) {
                h->ring[k][e].resize(rbpf_handle::RING);
                // timing only: no system-scope fence when the event completes (it would flush the caches between the
                // kernels it brackets and slow the very thing it measures)
                for (auto& ev : h->ring[k][e]) HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableSystemFence));
                h->begin_used[k].assign(rbpf_handle::RING, nullptr);
            }
        const size_t P = v.P, LL = (size_t)v.L * v.L, cells = (size_t)dim * dim;
        uint32_t* d_lut; ALLOC(h, d_lut, h->h_lut.size()); v.lut = d_lut;
        HIP_TRY(h, hipMemcpy(d_lut, h->h_lut.data(), h->h_lut.size() * 4, hipMemcpyHostToDevice));
        {   // inverse LUT: first global index of every 128-cell window of every lattice row
            const int KW = (dim + WIN - 1) / WIN;
            std::vector<int32_t> gw((size_t)v.L * (KW + 1));
            for (int a = 0; a < v.L; ++a)
                for (int k = 0; k <= KW; ++k) {
                    uint32_t key = k < KW ? (((uint32_t)a << 16) | (uint32_t)(k * WIN)) : ((uint32_t)(a + 1) << 16);
                    auto it = std::lower_bound(h->h_lut.begin(), h->h_lut.end(), key);
                    gw[(size_t)a * (KW + 1) + k] = v.g_min + (int32_t)(it - h->h_lut.begin());
                }
            int32_t* d_gw; ALLOC(h, d_gw, gw.size()); v.gwin = d_gw;
            HIP_TRY(h, hipMemcpy(d_gw, gw.data(), gw.size() * 4, hipMemcpyHostToDevice));
        }
        ALLOC(h, v.px, P); ALLOC(h, v.py, P); ALLOC(h, v.pth, P); ALLOC(h, v.cov, 9 * P); ALLOC(h, v.weight, P);
        ALLOC(h, v.slot, P); ALLOC(h, v.global_id, P);
        v.ow = (dim + 31) / 32;
        ALLOC(h, v.tile_tab, P * LL); ALLOC(h, v.pool, (size_t)v.pool_tiles * cells);
        ALLOC(h, v.occ, (size_t)v.pool_tiles * dim * v.ow + 4);       // (+4: the matcher's staging loads two / three words at a time)
        HIP_TRY(h, hipMemset(v.occ, 0, (size_t)v.pool_tiles * dim * v.ow * 4));
        ALLOC(h, v.tile_bbox, (size_t)v.pool_tiles * 4); ALLOC(h, v.free_stack, v.pool_tiles); ALLOC(h, v.free_top, 1);
        {   // the scan block: one device buffer, uploaded with one copy per scan (rbpf_set_scan)
            const size_t MB = (size_t)c.max_beams, MBP = (MB + 15) & ~(size_t)15;
            const size_t MBW = (MB + 63) & ~(size_t)63;           // the weighting's list is padded to whole passes of 64 beams
            h->scan_bytes = 3 * MBP * 8 + 4 * MBP * 4 + MBP + 2 * MBW * 4 + MBW * 2;
            ALLOC(h, h->d_scan, h->scan_bytes);
            unsigned char* d = h->d_scan;
            v.bx = reinterpret_cast<double*>(d); v.by = v.bx + MBP; v.bscale = v.by + MBP;
            v.msel_x = reinterpret_cast<float*>(d + 3 * MBP * 8); v.msel_y = v.msel_x + MBP; v.asel_x = v.msel_y + MBP; v.asel_y = v.asel_x + MBP;
            v.bflags = d + 3 * MBP * 8 + 4 * MBP * 4;
            v.wsel_x = reinterpret_cast<const float*>(d + 3 * MBP * 8 + 4 * MBP * 4 + MBP); v.wsel_y = v.wsel_x + MBW;
            v.wsel_idx = reinterpret_cast<const uint16_t*>(v.wsel_y + MBW);
            rbpf_handle::PinnedRing* rings[3] = {&h->ring_scan, &h->ring_last, &h->ring_idx};
            const size_t bytes[3] = {h->scan_bytes, MB * 16, (size_t)P * 8};
            for (int r = 0; r < 3; ++r) {
                rings[r]->slot_bytes = (bytes[r] + 255) & ~(size_t)255;
                HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&rings[r]->base), rings[r]->slot_bytes * rbpf_handle::PinnedRing::N, hipHostMallocDefault));
                for (int i = 0; i < rbpf_handle::PinnedRing::N; ++i) HIP_TRY(h, hipEventCreateWithFlags(&rings[r]->ev[i], hipEventDisableTiming | hipEventDisableSystemFence));   // guards host memory the copy only reads
            }
        }
        ALLOC(h, v.upd_pose, 3 * P);
        ALLOC(h, v.prop_prep, 24 * P);
        ALLOC(h, v.prop_samp, 256 * P);
        ALLOC(h, v.stats, ST_COUNT); ALLOC(h, v.err, 1);
        ALLOC(h, v.mu_fallback, P); HIP_TRY(h, hipMemset(v.mu_fallback, 0, P * 4));
        ALLOC(h, h->d_did_early, 1);
        HIP_TRY(h, hipMemset(h->d_did_early, 0, 4));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_weights, hipEventDisableTiming | hipEventDisableSystemFence));   // device-side ordering only
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_early, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_jobs, hipEventDisableTiming | hipEventDisableSystemFence));
        {   // RBPF_MAP_KERNEL=window keeps the 128x128-window map update for every particle, =ray / =ev run that first kernel (and
            // windows behind it); default: the event-walk kernel, windows for what it gives back (tests, comparisons)
            const char* mk = getenv("RBPF_MAP_KERNEL");
            v.mu_mode = (mk && std::string(mk) == "window") ? 1 : (mk && std::string(mk) == "ray") ? 3 : (mk && std::string(mk) == "ev") ? 5 : 0;
            const char* ms = getenv("RBPF_MATCH_STAGE");    // "slow": the matcher's field is staged bit by bit (tests)
            v.match_stage_slow = (ms && std::string(ms) == "slow") ? 1 : 0;
            const char* ws = getenv("RBPF_WSAFE");          // test knob: the weighting's guard band in cells (0 shows what the band is for)
            v.wsafe_override = ws ? (float)atof(ws) : -1.0f;
            const char* we = getenv("RBPF_WEIGHT_ENTRY");   // "f64": rbpf_weight_samples runs the float64 kernel (comparison)
            v.weight_entry_f64 = (we && std::string(we) == "f64") ? 1 : 0;
            v.ndt_refine = h->cfg.ndt_refine;
            const char* dd = getenv("RBPF_MATCH_DEDUP");     // "0": every particle runs the matcher, duplicates included (tests)
            h->dedup_enabled = !(dd && std::string(dd) == "0");
        }
        ALLOC(h, v.dup_of, P); v.dups_valid = 0;
        ALLOC(h, h->d_last_xy, 2 * (size_t)c.max_beams); ALLOC(h, h->d_tmp_sel, 2 * (size_t)c.max_beams);
        if (raycast_lds_bytes(c.max_beams, v.reach) > 160 * 1024) return fail(h, RBPF_EINVAL, "max_beams too large for the LDS window layout");
        match_geometry(c, c.cell_size, h->mN, h->mds, h->mmcs, h->md0, h->mncr);
        h->mlds = match_lds_bytes(h->mN, c.max_beams, match_max_coarse(h->mncr, 0.7, h->mmcs), match_per_rot(0.7, h->mmcs));
        if (h->mlds > 160 * 1024) return fail(h, RBPF_EINVAL, "matcher region does not fit in LDS for this cell_size");
        if (c.ndt_refine && ndt_cells(h->mmcs) >= 2) {         // the matcher hands its staged field to the NDT kernel
            if (ndt_lds_bytes(h->mN, c.max_beams) > 160 * 1024) return fail(h, RBPF_EINVAL, "NDT stage: matcher region does not fit in LDS");
            ALLOC(h, v.ndt_occ, P * (size_t)h->mN * (h->mN / 32)); ALLOC(h, v.ndt_aux, 5 * P);
        }
        ALLOC(h, h->d_match, 13 * P); ALLOC(h, h->d_bad, P); ALLOC(h, h->d_guess_full, P * (size_t)c.n_samples * 3);
        {
            ResampleBuffers& r = h->rs;
            ALLOC(h, r.T, P); ALLOC(h, r.idx, P); ALLOC(h, r.did, 1); ALLOC(h, r.slot2, P); ALLOC(h, r.dead_list, P);
            ALLOC(h, r.jobs, 2 * P); ALLOC(h, r.n_jobs, 2); ALLOC(h, r.pending_free, (size_t)v.pool_tiles); ALLOC(h, r.n_pending, 1);
            ALLOC(h, r.px2, P); ALLOC(h, r.py2, P); ALLOC(h, r.pth2, P); ALLOC(h, r.cov2, 9 * P); ALLOC(h, r.w2, P);
            HIP_TRY(h, hipMemset(r.n_pending, 0, 4)); HIP_TRY(h, hipMemset(r.n_jobs, 0, 8)); HIP_TRY(h, hipMemset(r.did, 0, 4));
        }
        HIP_TRY(h, hipMemset(v.pool, 0, (size_t)v.pool_tiles * cells));
        HIP_TRY(h, hipMemset(v.px, 0, P * 8)); HIP_TRY(h, hipMemset(v.py, 0, P * 8)); HIP_TRY(h, hipMemset(v.pth, 0, P * 8));
        HIP_TRY(h, hipMemset(v.cov, 0, 9 * P * 8));
        HIP_TRY(h, hipMemset(v.stats, 0, ST_COUNT * 8)); HIP_TRY(h, hipMemset(v.err, 0, 4));
        // robot.py:20-28 / hybridmap.py:70: weight 1.0, one empty tile centred (0,0) per particle
        std::vector<double> w(P, 1.0);
        HIP_TRY(h, hipMemcpy(v.weight, w.data(), P * 8, hipMemcpyHostToDevice));
        std::vector<int32_t> ids(P), tab(P * LL, -1), fs(v.pool_tiles), bb((size_t)v.pool_tiles * 4);
        for (size_t p = 0; p < P; ++p) { ids[p] = (int32_t)p; tab[p * LL + (size_t)v.R * v.L + v.R] = (int32_t)p; }
        for (int t = 0; t < v.pool_tiles; ++t) { bb[4 * t] = INT32_MAX; bb[4 * t + 1] = -1; bb[4 * t + 2] = INT32_MAX; bb[4 * t + 3] = -1; }
        int32_t top = v.pool_tiles - (int32_t)P;
        for (int i = 0; i < top; ++i) fs[i] = v.pool_tiles - 1 - i;   // pop order: P, P+1, ...
        HIP_TRY(h, hipMemcpy(v.slot, ids.data(), P * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(v.global_id, ids.data(), P * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(v.tile_tab, tab.data(), P * LL * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(v.free_stack, fs.data(), (size_t)v.pool_tiles * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(v.free_top, &top, 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(v.tile_bbox, bb.data(), bb.size() * 4, hipMemcpyHostToDevice));
        // the ray-cast kernel needs more than the default 64 KiB of dynamic LDS
        return RBPF_OK;
    };
    rc = build();
    if (rc != RBPF_OK) { g_create_err = h->err; rbpf_destroy(h); return rc; }
    *out = h;
    return RBPF_OK;
}

// Teardown order: (1) everything queued on the handle's stream has finished; (2) every event that guards host memory
// a kernel or a copy may still touch has completed - the early read-back and the pinned rings can be in use by work on
// ANOTHER stream (rbpf_resample_indices_global_early takes one); (3) device memory, pinned memory, events; (4) the
// stream, only if the handle created it.  A borrowed stream (rbpf_set_stream) is synchronised once and otherwise left
// alone: it belongs to the caller and may already be gone when a late destructor runs.
int rbpf_destroy(rbpf_handle* h) {
    if (!h) return RBPF_OK;
    ON_DEVICE(h);
    if (h->stream || h->own_stream) { if (hipStreamSynchronize(h->stream) != hipSuccess) (void)hipGetLastError(); }
    else if (hipStreamSynchronize(nullptr) != hipSuccess) (void)hipGetLastError();       // borrowed null stream
    auto wait_ev = [](hipEvent_t ev) { if (ev && hipEventSynchronize(ev) != hipSuccess) (void)hipGetLastError(); };
    if (h->early_n > 0 || h->h_early) wait_ev(h->ev_early);
    if (h->h_jobs_used) wait_ev(h->ev_jobs);
    if (h->ev_weights_valid) wait_ev(h->ev_weights);
    for (rbpf_handle::PinnedRing* r : {&h->ring_scan, &h->ring_last, &h->ring_idx})
        for (int i = 0; i < rbpf_handle::PinnedRing::N; ++i) if (r->used[i]) wait_ev(r->ev[i]);
    for (void* p : h->allocs) (void)hipFree(p);
    for (void* p : {(void*)h->d_guess, (void*)h->d_prs, (void*)h->d_w, (void*)h->d_gT, (void*)h->d_gidx, (void*)h->d_i32, (void*)h->d_jobs})
        if (p) (void)hipFree(p);
    if (h->h_jobs) (void)hipHostFree(h->h_jobs);
    if (h->h_early) (void)hipHostFree(h->h_early);
    for (rbpf_handle::PinnedRing* r : {&h->ring_scan, &h->ring_last, &h->ring_idx}) {
        if (r->base) (void)hipHostFree(r->base);
        for (int i = 0; i < rbpf_handle::PinnedRing::N; ++i) if (r->ev[i]) (void)hipEventDestroy(r->ev[i]);
    }
    for (hipEvent_t ev : {h->ev_weights, h->ev_jobs, h->ev_early}) if (ev) (void)hipEventDestroy(ev);
    for (int k = 0; k < rbpf_handle::N_KERN; ++k) for (int e = 0; e < 2; ++e) for (auto& ev : h->ring[k][e]) if (ev) (void)hipEventDestroy(ev);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    (void)hipGetLastError();
    delete h;
    return RBPF_OK;
}

const char* rbpf_last_error(const rbpf_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int rbpf_set_stream(rbpf_handle* h, void* s) {
    if (!h) return RBPF_EINVAL;
    ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));        // nullptr = the default stream
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->own_stream = false;
    h->stream = static_cast<hipStream_t>(s);
    h->last_end = nullptr;
    return RBPF_OK;
}

// gives a borrowed stream back: everything queued on it by this handle has finished when the call returns, and the
// handle works on a stream of its own again (as after rbpf_create)
int rbpf_release_stream(rbpf_handle* h) {
    if (!h) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (h->own_stream) return RBPF_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->early_n > 0) HIP_TRY(h, hipEventSynchronize(h->ev_early));
    h->stream = nullptr;
    HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    h->last_end = nullptr;
    return RBPF_OK;
}

int rbpf_abi_struct_bytes(int32_t* config_bytes, int32_t* counters_bytes) {
    if (config_bytes) *config_bytes = (int32_t)sizeof(rbpf_config);
    if (counters_bytes) *counters_bytes = (int32_t)sizeof(rbpf_counters);
    return RBPF_OK;
}

int rbpf_synchronize(rbpf_handle* h) {
    if (!h) return RBPF_EINVAL;
    ON_DEVICE(h);
    return check_device_error(h);
}

int rbpf_set_profiling_families(rbpf_handle* h, uint32_t mask) {
    if (!h) return RBPF_EINVAL;
    ON_DEVICE(h);
    return rbpf_set_profiling(h, mask ? (int)(0x100u | (mask & 0x1Fu)) : 0);
}

int rbpf_set_profiling(rbpf_handle* h, int on) {
    if (!h) return RBPF_EINVAL;
    ON_DEVICE(h);
    h->profiling = on != 0;
    h->prof_mask = !on ? 0u : (on & 0x100) ? ((unsigned)on & 0x1Fu) : 0x1Fu;
    for (int k = 0; k < rbpf_handle::N_KERN; ++k) h->ring_n[k] = 0;
    h->last_end = nullptr;
    HIP_TRY(h, hipMemsetAsync(h->v.stats, 0, ST_COUNT * sizeof(unsigned long long), h->stream));   // counters restart
    return RBPF_OK;
}

int rbpf_get_kernel_ms(rbpf_handle* h, int32_t which, double* out_ms, int32_t cap, int32_t* n_out) {
    if (!h || !n_out || which < 0 || which >= rbpf_handle::N_KERN) return RBPF_EINVAL;
    ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    int n = std::min(h->ring_n[which], rbpf_handle::RING);
    int first = h->ring_n[which] - n;
    int m = 0;
    for (int i = 0; i < n && m < cap; ++i) {
        int slot = (first + i) % rbpf_handle::RING;
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->begin_used[which][slot], h->ring[which][1][slot]) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (out_ms) out_ms[m] = ms;
        ++m;
    }
    *n_out = m;
    return RBPF_OK;
}

int rbpf_get_counters(rbpf_handle* h, rbpf_counters* out) {
    if (!h || !out) return RBPF_EINVAL;
    ON_DEVICE(h);
    unsigned long long st[ST_COUNT];
    int32_t top = 0;
    HIP_TRY(h, hipMemcpyAsync(st, h->v.stats, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&top, h->v.free_top, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    rbpf_counters& c = h->counters;
    c.scan_updates = h->scan_updates;
    c.ray_cells_visited = st[ST_RAY_CELLS]; c.cells_written = st[ST_CELLS_WRITTEN];
    c.cells_gathered = st[ST_GATHERS]; c.slow_cells = st[ST_SLOW_CELLS];
    c.resample_copies = st[ST_COPIES]; c.bytes_copied = st[ST_COPY_BYTES];
    c.tiles_in_use = (uint64_t)(h->v.pool_tiles - top);
    c.window_fallbacks = st[ST_WINDOW_FALLBACKS];
    c.ndt_runs = st[ST_NDT_RUNS]; c.ndt_evaluations = st[ST_NDT_EVALS]; c.ndt_accepted = st[ST_NDT_ACCEPTED];
    c.match_shared = st[ST_MATCH_SHARED];
    for (int k = 0; k < 7; ++k) c.reserved[k] = st[8 + k];
    c.stamp7 = st[15];
    {   // the packed form saturates per field; the exact tallies follow in their own fields
        auto sat16 = [](unsigned long long x) { return x > 65535ull ? 65535ull : x; };
        c.fallback_reasons = sat16(st[ST_FALLBACK_REASONS]) | (sat16(st[ST_FB_BOUND]) << 16) | (sat16(st[ST_FB_TABLES]) << 32);
        c.fallback_geometry = st[ST_FALLBACK_REASONS]; c.fallback_bound = st[ST_FB_BOUND]; c.fallback_tables = st[ST_FB_TABLES];
        c.map_events = st[ST_MAP_EVENTS]; c.map_event_overflows = st[ST_EV_OVERFLOWS];
    }
    c.map_windows = st[ST_MAP_WINDOWS];
    if (h->profiling) {
        double* dst[5] = {&c.ms_raycast, &c.ms_weight, &c.ms_resample, &c.ms_match, &c.ms_ndt};
        for (int k = 0; k < 5; ++k) {
            if (h->ring_n[k] == 0) continue;
            int slot = (h->ring_n[k] - 1) % rbpf_handle::RING;
            float ms = 0;
            if (hipEventElapsedTime(&ms, h->begin_used[k][slot], h->ring[k][1][slot]) == hipSuccess) *dst[k] = ms;
        }
        (void)hipGetLastError();
    }
    *out = c;
    return RBPF_OK;
}

// ---- a1 ---------------------------------------------------------------------------------------------
// the scan block from sensor-frame end points: range classes, compacted matcher lists, one upload
static int upload_scan_points(rbpf_handle* h, const double* px, const double* py, int32_t B) {
    const rbpf_config& c = h->cfg;
    const size_t MBP = ((size_t)c.max_beams + 15) & ~(size_t)15;
    unsigned char* slot = static_cast<unsigned char*>(h->ring_scan.acquire());
    double* sx = reinterpret_cast<double*>(slot);
    double* sy = sx + MBP;
    double* sc = sy + MBP;
    float* mx = reinterpret_cast<float*>(slot + 3 * MBP * 8);               // compacted beam lists for the matcher
    float* my = mx + MBP; float* ax = my + MBP; float* ay = ax + MBP;       // (float32, sensor frame)
    uint8_t* fl = slot + 3 * MBP * 8 + 4 * MBP * 4;
    const size_t MBW = ((size_t)c.max_beams + 63) & ~(size_t)63;
    float* wx = reinterpret_cast<float*>(fl + MBP); float* wy = wx + MBW;   // the weighting's beams (kernels_propose.hip, weight_beams)
    uint16_t* wi = reinterpret_cast<uint16_t*>(wy + MBW);
    const float w_inv_cs = (float)((double)h->v.dim / h->v.tile_len), w_lim = 1.5f * (float)h->v.dim;
    int nm = 0, na = 0, nw = 0;
    for (int i = 0; i < B; ++i) {
        const double x = px[i], y = py[i];
        double dist = sqrt(x * x + y * y);               // robot.py:129, hybridmap.py:105,217
        uint8_t f = 0;
        if (dist < c.weight_max_range && dist > c.weight_min_range) f |= BF_WEIGHT;   // robot.py:130
        if (dist < c.match_max_range && dist > c.match_min_range) f |= BF_MATCH;      // hybridmap.py:218
        if (dist < c.match_max_range) f |= BF_MATCH_ADJ;                              // hybridmap.py:172
        double s = 1.0;
        if (dist > c.max_ray_m) { f |= BF_LONG; s = c.max_ray_m / dist; }             // hybridmap.py:107-108
        sx[i] = x; sy[i] = y; sc[i] = s; fl[i] = f;
        if (f & BF_MATCH) { mx[nm] = (float)x; my[nm] = (float)y; ++nm; }
        if (f & BF_MATCH_ADJ) { ax[na] = (float)x; ay[na] = (float)y; ++na; }
        if (f & BF_WEIGHT) {
            const float x32 = (float)x, y32 = (float)y;
            const bool in_budget = h->v.dim <= 2048 && (fabsf(x32) + fabsf(y32)) * w_inv_cs <= w_lim;   // the error budget's premise
            wx[nw] = in_budget ? x32 : NAN; wy[nw] = y32; wi[nw] = (uint16_t)i; ++nw;
        }
    }
    for (int i = nw; i < ((nw + 63) & ~63); ++i) { wx[i] = NAN; wy[i] = 0.0f; wi[i] = 0; }
    DevView& v = h->v;
    v.n_msel = nm; v.n_asel = na; v.n_wsel = nw;
    {   // pinned and device-mapped: a kernel pulls the block over (no copy-engine latency in the stream); DMA otherwise
        void* mapped = nullptr;
        if (hipHostGetDevicePointer(&mapped, slot, 0) == hipSuccess && mapped) launch_ingest(mapped, h->d_scan, h->scan_bytes, h->stream);
        else { (void)hipGetLastError(); HIP_TRY(h, hipMemcpyAsync(h->d_scan, slot, h->scan_bytes, hipMemcpyHostToDevice, h->stream)); }
    }
    h->ring_scan.submitted(h->stream);
    v.B = B;
    h->have_scan = true;
    return RBPF_OK;
}

int rbpf_set_scan(rbpf_handle* h, const double* ranges, const double* angles, int32_t B) {
    if (!h || !ranges || !angles) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (B < 1 || B > h->cfg.max_beams) return fail(h, RBPF_EINVAL, "n_beams out of range");
    std::vector<double> x((size_t)B), y((size_t)B);
    for (int i = 0; i < B; ++i) {
        x[i] = ranges[i] * cos(angles[i]);               // lidar.py:78
        y[i] = ranges[i] * sin(angles[i]);               // lidar.py:79
    }
    return upload_scan_points(h, x.data(), y.data(), B);
}

// the same from a Scan object's own state: its sensor-frame end points x(), y() (lidar.py:76-87)
int rbpf_set_scan_xy(rbpf_handle* h, const double* x, const double* y, int32_t B) {
    if (!h || !x || !y) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (B < 1 || B > h->cfg.max_beams) return fail(h, RBPF_EINVAL, "n_beams out of range");
    return upload_scan_points(h, x, y, B);
}

// ---- a2 ---------------------------------------------------------------------------------------------
int rbpf_imu_update(rbpf_handle* h, int32_t model, const double* d, double dt_ticks) {
    if (!h || !d) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (model < 0 || model > 2) return fail(h, RBPF_EINVAL, "unknown motion model");
    launch_imu_update(h->v, model, d[0], d[1], d[2], dt_ticks, h->cfg.vel_noise, h->stream);
    HIP_TRY(h, hipGetLastError());
    return RBPF_OK;
}

// ---- a4 test entry --------------------------------------------------------------------------------------
static int ensure_sample_buffers(rbpf_handle* h, size_t n) {
    if (h->d_guess_n >= n) return RBPF_OK;
    if (h->d_guess) { hipFree(h->d_guess); hipFree(h->d_prs); hipFree(h->d_w); h->d_guess = nullptr; }
    HIP_TRY(h, hipMalloc((void**)&h->d_guess, n * 3 * 8));
    HIP_TRY(h, hipMalloc((void**)&h->d_prs, n * 8));
    HIP_TRY(h, hipMalloc((void**)&h->d_w, n * 8));
    h->d_guess_n = n;
    return RBPF_OK;
}

int rbpf_weight_samples(rbpf_handle* h, const double* guesses, const double* prs, int32_t K, double* out_w) {
    if (!h || !guesses || !prs || !out_w) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (!h->have_scan) return fail(h, RBPF_ESTATE, "rbpf_set_scan has not been called");
    if (K < 1 || K > 32) return fail(h, RBPF_EINVAL, "n_samples must be in 1..32");
    const size_t n = (size_t)h->v.P * K;
    int rc = ensure_sample_buffers(h, n);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_guess, guesses, n * 3 * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_prs, prs, n * 8, hipMemcpyHostToDevice, h->stream));
    h->prof_begin(1);
    if (h->v.weight_entry_f64 || K > 32) launch_weight_samples(h->v, h->d_guess, h->d_prs, K, h->d_w, h->stream);
    else launch_weight_samples_product(h->v, h->d_guess, h->d_prs, K, h->d_w, h->stream);   // the look-ups of every scan step (kernels_propose.hip)
    h->prof_end(1);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out_w, h->d_w, n * 8, hipMemcpyDeviceToHost, h->stream));
    return check_device_error(h);
}

// ---- a5 test entry --------------------------------------------------------------------------------------
static int run_map_update(rbpf_handle* h, const uint8_t* d_bad = nullptr) {
    DevView& v = h->v;
    // The map update's timing events ride on the first kernel's dispatch (its own start and end: the dominant kernel, without
    // the follow-up launch that usually finds nothing to do) - two event records and their barriers less in the stream per step.
    if (((h->prof_mask >> 0) & 1u) && map_update_first_kernel(v) != 0) {
        const int slot = h->ring_n[0] % rbpf_handle::RING;
        hipEvent_t t0 = h->ring[0][0][slot], t1 = h->ring[0][1][slot];
        launch_map_update_fused(v, d_bad, h->stream, t0, t1);
        h->begin_used[0][slot] = t0; h->last_end = nullptr; h->ring_n[0]++;
    } else {
        h->prof_begin(0);
        launch_map_update_fused(v, d_bad, h->stream);
        h->prof_end(0);
    }
    HIP_TRY(h, hipGetLastError());
    h->scan_updates++;
    return RBPF_OK;
}

int rbpf_map_update(rbpf_handle* h, const double* poses) {
    if (!h) return RBPF_EINVAL;
    ON_DEVICE(h);
    h->v.dups_valid = 0;
    if (!h->have_scan) return fail(h, RBPF_ESTATE, "rbpf_set_scan has not been called");
    DevView& v = h->v;
    const size_t P = v.P;
    if (poses) {
        std::vector<double> soa(3 * P);
        for (size_t p = 0; p < P; ++p) { soa[p] = poses[3 * p]; soa[P + p] = poses[3 * p + 1]; soa[2 * P + p] = poses[3 * p + 2]; }
        HIP_TRY(h, hipMemcpyAsync(v.upd_pose, soa.data(), 3 * P * 8, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    } else {
        HIP_TRY(h, hipMemcpyAsync(v.upd_pose, v.px, P * 8, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(v.upd_pose + P, v.py, P * 8, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(v.upd_pose + 2 * P, v.pth, P * 8, hipMemcpyDeviceToDevice, h->stream));
    }
    int rc = run_map_update(h);
    if (rc) return rc;
    return check_device_error(h);
}

static int run_matcher(rbpf_handle* h, int32_t adj, const double* last_scan_xy, int32_t n_last) {
    DevView& v = h->v;
    if (adj && !last_scan_xy) {                 // the device-resident previous scan (rbpf_refresh_last_scan / rbpf_import_last_scan)
        if (h->n_last_dev < 0) return fail(h, RBPF_ESTATE, "adj = 1 without last_scan_xy needs rbpf_refresh_last_scan first");
        n_last = h->n_last_dev;
    } else if (adj && (n_last < 0 || n_last > h->cfg.max_beams))
        return fail(h, RBPF_EINVAL, "adj = 1 needs last_scan_xy with at most max_beams points");
    if (adj && last_scan_xy) {
        void* slot = h->ring_last.acquire();
        memcpy(slot, last_scan_xy, (size_t)n_last * 16);
        HIP_TRY(h, hipMemcpyAsync(h->d_last_xy, slot, (size_t)n_last * 16, hipMemcpyHostToDevice, h->stream));
        h->ring_last.submitted(h->stream);
    }
    // both stages are one kernel each: their timing events ride on the dispatch and take the kernel's own start and end
    hipEvent_t t0 = nullptr, t1 = nullptr;
    auto take_events = [&](int k) { t0 = t1 = nullptr; if (!((h->prof_mask >> k) & 1u)) return; const int slot = h->ring_n[k] % rbpf_handle::RING;
                                    t0 = h->ring[k][0][slot]; t1 = h->ring[k][1][slot]; h->begin_used[k][slot] = t0; h->last_end = nullptr; h->ring_n[k]++; };
    take_events(3);
    const bool ndt = launch_match_particles(v, adj ? 1 : 0, h->d_last_xy, adj ? n_last : 0, h->d_match, h->mN, h->mds, h->mmcs, h->md0,
                                            h->mncr, h->cfg.match_max_range, h->cfg.max_beams, h->mlds, 1, h->stream, t0, t1);
    if (ndt) {
        take_events(4);
        launch_match_particles(v, adj ? 1 : 0, h->d_last_xy, adj ? n_last : 0, h->d_match, h->mN, h->mds, h->mmcs, h->md0,
                               h->mncr, h->cfg.match_max_range, h->cfg.max_beams, h->mlds, 2, h->stream, t0, t1);
    }
    HIP_TRY(h, hipGetLastError());
    return RBPF_OK;
}

// ---- Robot.map_update for every particle (robot.py:59-115) -------------------------------------------------
int rbpf_scan_update(rbpf_handle* h, int32_t adj, const double* last_scan_xy, int32_t n_last,
                     const double* match_override, const double* guesses) {
    int rc = rbpf_scan_update_begin(h, adj, last_scan_xy, n_last, match_override, guesses);
    return rc ? rc : rbpf_scan_update_end(h);
}

// first half: scan matcher, proposal, weighting, moments (robot.py:62-114); the weights are final here unless a
// particle took the NaN branch
int rbpf_scan_update_begin(rbpf_handle* h, int32_t adj, const double* last_scan_xy, int32_t n_last,
                           const double* match_override, const double* guesses) {
    if (!h) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (!h->have_scan) return fail(h, RBPF_ESTATE, "rbpf_set_scan has not been called");
    DevView& v = h->v;
    const size_t P = v.P;
    if (match_override) {
        HIP_TRY(h, hipMemcpyAsync(h->d_match, match_override, 13 * P * 8, hipMemcpyHostToDevice, h->stream));
    } else {
        int rc = run_matcher(h, adj, last_scan_xy, n_last);
        if (rc) return rc;
    }
    const double* d_g = nullptr;
    if (guesses) {
        HIP_TRY(h, hipMemcpyAsync(h->d_guess_full, guesses, P * (size_t)v.K * 3 * 8, hipMemcpyHostToDevice, h->stream));
        d_g = h->d_guess_full;
    }
    if (!match_override && !guesses) h->prof_begin_chained(1); else h->prof_begin(1);     // right after the matcher's last kernel
    // the matcher ran once per group of exact duplicates (copies made by the last resample, untouched since): every
    // member reads its representative's row; the proposal below is what makes the copies differ, so the groups end here
    const int32_t* match_of = (!match_override && v.dups_valid) ? v.dup_of : nullptr;
    launch_propose_weight(v, h->d_match, match_of, d_g, h->d_bad, h->cfg.seed, (uint32_t)h->scan_updates, nullptr, h->stream);
    v.dups_valid = 0;
    h->prof_end(1);
    HIP_TRY(h, hipGetLastError());
    // the event orders an early weight export on ANOTHER stream behind the weighting; recorded only once such a caller exists
    h->ev_weights_valid = false;
    if (h->record_ev_weights) { HIP_TRY(h, hipEventRecord(h->ev_weights, h->stream)); h->ev_weights_valid = true; }
    h->scan_begun = true; h->begin_seen = true;
    return RBPF_OK;
}

// second half: the map update at the new mean pose and the NaN-covariance branch (robot.py:115, 73-78)
int rbpf_scan_update_end(rbpf_handle* h) {
    if (!h) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (!h->scan_begun) return fail(h, RBPF_ESTATE, "rbpf_scan_update_begin has not been called");
    h->scan_begun = false;
    // HybridMap.update at the new mean pose (robot.py:115), then - in the same launch - the robot.py:73-78 weight
    // increment of the particles on the NaN-covariance branch, on their updated maps
    return run_map_update(h, h->d_bad);
}

// matchScanCustom(curr, ref, guess, cells_per_m, pose_range) -> pose, cov, score  (hybridmap.py:244-251)
int rbpf_match_scan(rbpf_handle* h, const double* curr_xy, int32_t n_curr, const double* ref_xy, int32_t n_ref,
                    const double* guess3, int32_t cells_per_m, const double* pose_range3, double* pose_out3,
                    double* cov_out9, double* score_out) {
    if (!h || !curr_xy || !guess3 || !pose_range3 || !pose_out3 || !cov_out9 || !score_out) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (n_curr < 0 || n_curr > h->cfg.max_beams || n_ref < 0 || (n_ref > 0 && !ref_xy)) return fail(h, RBPF_EINVAL, "point counts out of range");
    if (cells_per_m < 1) return fail(h, RBPF_EINVAL, "cells_per_m must be >= 1");
    rbpf_config c = h->cfg;
    c.match_max_range = 15.0;                       // matchScanCustom.m:11 'MaxRange', 15
    int N, ds, ncr; double mcs, d0;
    match_geometry(c, 1.0 / (double)cells_per_m, N, ds, mcs, d0, ncr);
    ncr = (int)floor(fabs(pose_range3[2]) / (4 * d0));             // coarse rotation step = 4 * d0
    if (ncr * 4 * d0 >= fabs(pose_range3[2])) --ncr;
    if (ncr < 0) ncr = 0;
    size_t lds = match_lds_bytes(N, h->cfg.max_beams, match_max_coarse(ncr, std::max(pose_range3[0], pose_range3[1]), mcs),
                                 match_per_rot(std::max(pose_range3[0], pose_range3[1]), mcs));
    if (lds > 160 * 1024) return fail(h, RBPF_EINVAL, "matcher region does not fit in LDS for this resolution");
    std::vector<float> sel(2 * (size_t)h->cfg.max_beams, 0.f);
    for (int i = 0; i < n_curr; ++i) { sel[i] = (float)curr_xy[2 * i]; sel[h->cfg.max_beams + i] = (float)curr_xy[2 * i + 1]; }
    double *d_ref = nullptr, *d_out = nullptr, *d_aux = nullptr; uint32_t* d_occ = nullptr;
    HIP_TRY(h, hipMalloc((void**)&d_ref, std::max<size_t>((size_t)n_ref, 1) * 16));
    HIP_TRY(h, hipMalloc((void**)&d_out, 13 * 8));
    if (c.ndt_refine && ndt_cells(mcs) >= 2 && ndt_lds_bytes(N, h->cfg.max_beams) <= 160 * 1024) {
        HIP_TRY(h, hipMalloc((void**)&d_occ, (size_t)N * (N / 32) * 4));
        HIP_TRY(h, hipMalloc((void**)&d_aux, 5 * 8));
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_tmp_sel, sel.data(), sel.size() * 4, hipMemcpyHostToDevice, h->stream));
    if (n_ref) HIP_TRY(h, hipMemcpyAsync(d_ref, ref_xy, (size_t)n_ref * 16, hipMemcpyHostToDevice, h->stream));
    launch_match_single(h->v, d_ref, n_ref, guess3, pose_range3, h->d_tmp_sel, h->d_tmp_sel + h->cfg.max_beams, n_curr, d_out,
                        N, ds, mcs, d0, ncr, h->cfg.max_beams, lds, d_occ, d_aux, h->stream);
    double out[13];
    HIP_TRY(h, hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, h->stream));
    int rc = check_device_error(h);
    (void)hipFree(d_ref); (void)hipFree(d_out); (void)hipFree(d_occ); (void)hipFree(d_aux);
    if (rc) return rc;
    // matchScanCustom.m:19,52-57 validity gate
    const double PI = 3.141592653589793;
    double dth = fmod(out[2] - guess3[2] + PI, 2 * PI); if (dth < 0) dth += 2 * PI; dth -= PI;
    bool valid = fabs(out[0] - guess3[0]) < fabs(pose_range3[0]) && fabs(out[1] - guess3[1]) < fabs(pose_range3[1]) &&
                 fabs(dth) < fabs(pose_range3[2]) && !(out[3] != out[3]);
    for (int i = 0; i < 3; ++i) pose_out3[i] = out[i];
    for (int i = 0; i < 9; ++i) cov_out9[i] = valid ? out[3 + i] : std::numeric_limits<double>::quiet_NaN();
    *score_out = valid ? out[12] : 0.0;
    return RBPF_OK;
}
// HybridMap.get_scan_match up to the engine call (hybridmap.py:210-242) for one particle: the curr / ref point lists
int rbpf_match_inputs(rbpf_handle* h, int32_t particle, const double* guess3, double* curr_xy, int32_t* n_curr,
                      double* ref_xy, int32_t* n_ref, int32_t cap_ref) {
    if (!h || !guess3 || !curr_xy || !n_curr || !ref_xy || !n_ref || cap_ref < 0) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (!h->have_scan) return fail(h, RBPF_ESTATE, "rbpf_set_scan has not been called");
    DevView& v = h->v;
    if (particle < 0 || particle >= v.P) return fail(h, RBPF_EINVAL, "particle index out of range");
    const size_t LL = (size_t)v.L * v.L, mask_words = LL * v.dim * v.ow, n_rows = (size_t)v.L * v.dim;
    double *d_all = nullptr, *d_ref = nullptr, *d_curr = nullptr; int* d_counts = nullptr; uint32_t* d_mask = nullptr; int* d_rows = nullptr;
    HIP_TRY(h, hipMalloc((void**)&d_all, (size_t)v.B * 16)); HIP_TRY(h, hipMalloc((void**)&d_curr, (size_t)v.B * 16));
    HIP_TRY(h, hipMalloc((void**)&d_ref, std::max<size_t>(cap_ref, 1) * 16)); HIP_TRY(h, hipMalloc((void**)&d_counts, 16));
    HIP_TRY(h, hipMalloc((void**)&d_mask, mask_words * 4)); HIP_TRY(h, hipMalloc((void**)&d_rows, n_rows * 4));
    HIP_TRY(h, hipMemsetAsync(d_mask, 0, mask_words * 4, h->stream));
    HIP_TRY(h, hipMemsetAsync(d_counts, 0, 16, h->stream));
    const int win = (int)(1.8 / h->cfg.cell_size);                 // gridmap.py:143
    launch_match_inputs(v, particle, guess3, d_all, d_counts, d_mask, d_rows, d_ref, cap_ref, d_curr, win,
                        h->cfg.match_max_range, h->stream);
    int counts[3] = {0, 0, 0};
    HIP_TRY(h, hipMemcpyAsync(counts, d_counts, 12, hipMemcpyDeviceToHost, h->stream));
    int rc = check_device_error(h);
    if (rc == RBPF_OK) {
        *n_curr = counts[1]; *n_ref = counts[2];
        if (counts[1] > 0) HIP_TRY(h, hipMemcpy(curr_xy, d_curr, (size_t)counts[1] * 16, hipMemcpyDeviceToHost));
        int nr = std::min(counts[2], cap_ref);
        if (nr > 0) HIP_TRY(h, hipMemcpy(ref_xy, d_ref, (size_t)nr * 16, hipMemcpyDeviceToHost));
    }
    (void)hipFree(d_all); (void)hipFree(d_curr); (void)hipFree(d_ref); (void)hipFree(d_counts); (void)hipFree(d_mask); (void)hipFree(d_rows);
    return rc;
}

// ---- resample (main.py:46-79) -------------------------------------------------------------------------------
static double internal_uniform(rbpf_handle* h) {
    uint64_t z = h->cfg.seed + 0x9E3779B97F4A7C15ull * (++h->resample_draws);   // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

static void swap_state_buffers(rbpf_handle* h) {
    DevView& v = h->v; ResampleBuffers& r = h->rs;
    std::swap(v.px, r.px2); std::swap(v.py, r.py2); std::swap(v.pth, r.pth2);
    std::swap(v.cov, r.cov2); std::swap(v.weight, r.w2); std::swap(v.slot, r.slot2);
}

int rbpf_resample(rbpf_handle* h, double u, int32_t* idx_out, int32_t* did_resample) {
    if (!h) return RBPF_EINVAL;
    ON_DEVICE(h);
    DevView& v = h->v;
    if (u != u) u = internal_uniform(h);
    if (!(u >= 0.0 && u < 1.0)) return fail(h, RBPF_EINVAL, "u must lie in [0, 1)");
    h->prof_begin(2);
    launch_resample_local(v, h->rs, v.weight, u, h->cfg.resample_spread, h->stream);
    h->prof_end(2);
    HIP_TRY(h, hipGetLastError());
    swap_state_buffers(h);
    v.dups_valid = h->dedup_enabled ? 1 : 0;            // the kernel wrote the groups of exact duplicates (identity if it did not resample)
    if (idx_out) HIP_TRY(h, hipMemcpyAsync(idx_out, h->rs.idx, (size_t)v.P * 4, hipMemcpyDeviceToHost, h->stream));
    if (did_resample) HIP_TRY(h, hipMemcpyAsync(did_resample, h->rs.did, 4, hipMemcpyDeviceToHost, h->stream));
    if (idx_out || did_resample) return check_device_error(h);
    return RBPF_OK;
}

// ---- multi-GPU pieces: one handle per rank, the collectives are the caller's (RCCL) ---------------------------------
int rbpf_set_global_ids(rbpf_handle* h, const int32_t* ids) {
    if (!h || !ids) return RBPF_EINVAL;
    ON_DEVICE(h);
    HIP_TRY(h, hipMemcpyAsync(h->v.global_id, ids, (size_t)h->v.P * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return RBPF_OK;
}

int rbpf_export_weights(rbpf_handle* h, void* d_global, int32_t n_global) {
    if (!h || !d_global || n_global < h->v.P) return RBPF_EINVAL;
    ON_DEVICE(h);
    launch_export_weights(h->v, static_cast<double*>(d_global), n_global, nullptr, h->stream);
    HIP_TRY(h, hipGetLastError());
    return RBPF_OK;                                     // stream-ordered: see the header about collectives on other streams
}

int rbpf_export_weights_early(rbpf_handle* h, void* d_global, int32_t n_global, void* aux_stream) {
    if (!h || !d_global || n_global < h->v.P) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (!h->begin_seen) return fail(h, RBPF_ESTATE, "rbpf_scan_update_begin has not been called");
    hipStream_t s = static_cast<hipStream_t>(aux_stream);
    if (s != h->stream) {                                 // same stream: stream order is the ordering
        if (!h->ev_weights_valid) {                       // first such call: order behind everything queued so far, record in time from now on
            HIP_TRY(h, hipEventRecord(h->ev_weights, h->stream));
            h->ev_weights_valid = true; h->record_ev_weights = true;
        }
        HIP_TRY(h, hipStreamWaitEvent(s, h->ev_weights, 0));
    }
    launch_export_weights(h->v, static_cast<double*>(d_global), n_global, h->d_bad, s);
    HIP_TRY(h, hipGetLastError());
    return RBPF_OK;
}

int rbpf_resample_indices_global_early(rbpf_handle* h, const void* d_global, int32_t n_global, double u, void* aux_stream) {
    if (!h || !d_global || n_global < 1) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (!(u >= 0.0 && u < 1.0)) return fail(h, RBPF_EINVAL, "u must lie in [0, 1)");
    hipStream_t s = static_cast<hipStream_t>(aux_stream);
    int rc = scratch(h, &h->d_gT, &h->d_gT_cap, (size_t)n_global);
    if (rc) return rc;
    rc = scratch(h, &h->d_gidx, &h->d_gidx_cap, (size_t)n_global);
    if (rc) return rc;
    const size_t need = (size_t)n_global * 4 + 16;
    if (need > h->h_early_bytes) {                       // pinned landing zone of the read-back
        if (h->h_early) HIP_TRY(h, hipHostFree(h->h_early));
        // a kernel writes it through its device address, the host reads it after ev_early, which is created WITHOUT
        // hipEventDisableSystemFence and so releases at system scope: plain pinned memory is enough
        HIP_TRY(h, hipHostMalloc(&h->h_early, need, hipHostMallocDefault));
        h->h_early_bytes = need;
    }
    launch_resample_indices(n_global, static_cast<const double*>(d_global), u, h->cfg.resample_spread, h->d_gT, h->d_gidx,
                            h->d_did_early, h->v.err, s);
    HIP_TRY(h, hipGetLastError());
    unsigned char* dst = static_cast<unsigned char*>(h->h_early);
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, dst, 0) == hipSuccess && mapped) {      // one kernel writes the landing zone directly
        launch_readback(mapped, static_cast<const double*>(d_global) + n_global, h->d_did_early, h->d_gidx, n_global, s);
    } else {
        (void)hipGetLastError();
        HIP_TRY(h, hipMemcpyAsync(dst, static_cast<const double*>(d_global) + n_global, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpyAsync(dst + 8, h->d_did_early, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpyAsync(dst + 16, h->d_gidx, (size_t)n_global * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(h, hipEventRecord(h->ev_early, s));
    h->early_n = n_global;
    return RBPF_OK;                                      // nothing waited for: what is queued behind it keeps the GPU busy
}

int rbpf_resample_indices_global_wait(rbpf_handle* h, int32_t* idx_out, int32_t* did_resample, double* nan_branch_ranks) {
    if (!h || !idx_out || !did_resample || !nan_branch_ranks) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (h->early_n <= 0) return fail(h, RBPF_ESTATE, "rbpf_resample_indices_global_early has not been called");
    HIP_TRY(h, hipEventSynchronize(h->ev_early));        // only up to the read-back, not the work queued after it
    const unsigned char* src = static_cast<const unsigned char*>(h->h_early);
    memcpy(nan_branch_ranks, src, 8);
    memcpy(did_resample, src + 8, 4);
    memcpy(idx_out, src + 16, (size_t)h->early_n * 4);
    h->early_n = 0;
    return RBPF_OK;
}

int rbpf_resample_indices_global(rbpf_handle* h, const void* d_global, int32_t n_global, double u, int32_t* idx_out,
                                 int32_t* did_resample) {
    if (!h || !d_global || !idx_out || !did_resample || n_global < 1) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (!(u >= 0.0 && u < 1.0)) return fail(h, RBPF_EINVAL, "u must lie in [0, 1)");
    int rc = scratch(h, &h->d_gT, &h->d_gT_cap, (size_t)n_global);
    if (rc) return rc;
    rc = scratch(h, &h->d_gidx, &h->d_gidx_cap, (size_t)n_global);
    if (rc) return rc;
    launch_resample_indices(n_global, static_cast<const double*>(d_global), u, h->cfg.resample_spread, h->d_gT, h->d_gidx,
                            h->rs.did, h->v.err, h->stream);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(idx_out, h->d_gidx, (size_t)n_global * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(did_resample, h->rs.did, 4, hipMemcpyDeviceToHost, h->stream));
    return check_device_error(h);
}

int rbpf_apply_resample_local(rbpf_handle* h, const int32_t* new_src, const int32_t* new_global_id) {
    if (!h || !new_src || !new_global_id) return RBPF_EINVAL;
    ON_DEVICE(h);
    DevView& v = h->v;
    // sources must be sorted ascending with the arrivals (-1) last: duplicates of one ancestor are then adjacent
    for (int j = 1; j < v.P; ++j) {
        bool ok = new_src[j] < 0 ? true : (new_src[j - 1] >= 0 && new_src[j - 1] <= new_src[j]);
        if (!ok) return fail(h, RBPF_EINVAL, "new_src must be sorted ascending with -1 entries last");
    }
    for (int j = 0; j < v.P; ++j) if (new_src[j] >= v.P) return fail(h, RBPF_EINVAL, "new_src out of range");
    int32_t* slot = static_cast<int32_t*>(h->ring_idx.acquire());   // pinned: the caller's arrays are free on return
    memcpy(slot, new_src, (size_t)v.P * 4);
    memcpy(slot + v.P, new_global_id, (size_t)v.P * 4);
    // both index vectors go over in one kernel from the pinned, device-mapped slot (nothing below reads global_id:
    // only the weight export does, after this call)
    void* mapped = nullptr;
    const bool by_kernel = hipHostGetDevicePointer(&mapped, slot, 0) == hipSuccess && mapped;
    if (by_kernel) launch_ingest2(static_cast<const int32_t*>(mapped), h->rs.idx, static_cast<const int32_t*>(mapped) + v.P, v.global_id, v.P, h->stream);
    else { (void)hipGetLastError(); HIP_TRY(h, hipMemcpyAsync(h->rs.idx, slot, (size_t)v.P * 4, hipMemcpyHostToDevice, h->stream)); }
    h->prof_begin(2);
    launch_resample_apply_sources(v, h->rs, h->stream);
    h->prof_end(2);
    HIP_TRY(h, hipGetLastError());
    swap_state_buffers(h);
    v.dups_valid = h->dedup_enabled ? 1 : 0;            // arrivals are their own representatives
    if (!by_kernel) HIP_TRY(h, hipMemcpyAsync(v.global_id, slot + v.P, (size_t)v.P * 4, hipMemcpyHostToDevice, h->stream));
    h->ring_idx.submitted(h->stream);
    return RBPF_OK;                                     // no host synchronisation; device errors surface at the next check
}

int32_t rbpf_pack_meta_width(rbpf_handle* h) { return h ? 2 + 6 * h->v.L * h->v.L : -1; }

int64_t rbpf_packed_particle_bytes(rbpf_handle* h) {
    if (!h) return -1;
    const DevView& v = h->v;
    return 128 + (int64_t)v.L * v.L * ((int64_t)v.dim * v.dim + (int64_t)v.dim * v.ow * 4);
}

// meta record of one particle: [0] tiles, [1] payload bytes / 16, then per lattice position (has, x0, x1, ya, yb, offset / 16)
// job lists of the pack / unpack kernels go through one pinned buffer: the copy is asynchronous and the std::vector may
// die on return; the event tells when the buffer may be overwritten
static int stage_jobs(rbpf_handle* h, const void* src, size_t bytes) {
    if (h->h_jobs_used) HIP_TRY(h, hipEventSynchronize(h->ev_jobs));
    if (bytes > h->h_jobs_bytes) {
        if (h->h_jobs) HIP_TRY(h, hipHostFree(h->h_jobs));
        h->h_jobs_bytes = std::max<size_t>(bytes * 2, 1 << 16);
        HIP_TRY(h, hipHostMalloc(&h->h_jobs, h->h_jobs_bytes, hipHostMallocDefault));
    }
    memcpy(h->h_jobs, src, bytes);
    int rc = scratch(h, &h->d_jobs, &h->d_jobs_cap, bytes);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_jobs, h->h_jobs, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipEventRecord(h->ev_jobs, h->stream));
    h->h_jobs_used = true;
    return RBPF_OK;
}

// Layout of n packed particles from their gathered tile boxes (g: per particle and lattice position 5 ints: tile index or
// -1, x0, x1, y0, y1 - the output of gather_meta_kernel).  Fills the receiver's meta rows (optional) and, for the sender
// (local_idx given), the pack jobs.  Returns the total payload bytes.  Host only.
static int64_t layout_packed(const DevView& v, const int32_t* g, int n, const int32_t* local_idx, int32_t* meta_out,
                             std::vector<PackJobHost>* jobs) {
    const int LL = v.L * v.L, W = 2 + 6 * LL;
    int64_t off = 0;
    for (int i = 0; i < n; ++i) {
        int32_t* m = meta_out ? meta_out + (size_t)i * W : nullptr;
        const int64_t start = off;
        if (jobs) jobs->push_back({local_idx[i], -1, 0, 0, 0, 0, (long long)off});
        off += 128;
        int nt = 0;
        for (int pos = 0; pos < LL; ++pos) {
            const int32_t* e = &g[((size_t)i * LL + pos) * 5];
            int32_t z[6] = {0, 0, 0, 0, 0, 0};
            int32_t* mm = m ? m + 2 + 6 * pos : z;
            mm[0] = mm[1] = mm[2] = mm[3] = mm[4] = mm[5] = 0;
            if (e[0] < 0) continue;
            ++nt;
            mm[0] = 1;
            int x0 = e[1], x1 = e[2], y0 = e[3], y1 = e[4];
            if (x0 > x1 || y0 > y1) { mm[1] = 0; mm[2] = -1; mm[3] = 0; mm[4] = 0; mm[5] = (int32_t)((off - start) / 16); continue; }   // empty tile
            int ya = y0 & ~15, yb = std::min((y1 | 15) + 1, v.dim);
            mm[1] = x0; mm[2] = x1; mm[3] = ya; mm[4] = yb; mm[5] = (int32_t)((off - start) / 16);
            if (jobs) jobs->push_back({local_idx[i], e[0], x0, x1, ya, yb, (long long)off});
            int64_t bytes = (int64_t)(x1 - x0 + 1) * (yb - ya) + (int64_t)(x1 - x0 + 1) * v.ow * 4;
            off += (bytes + 15) & ~(int64_t)15;
        }
        if (m) { m[0] = nt; m[1] = (int32_t)((off - start) / 16); }
    }
    return off;
}

int rbpf_pack_particles(rbpf_handle* h, const int32_t* local_idx, int32_t n, void* d_buf, int64_t cap_bytes,
                        int32_t* meta_out, int64_t* bytes_out) {
    if (!h || n < 0 || !bytes_out || (n > 0 && (!local_idx || !d_buf || !meta_out))) return RBPF_EINVAL;
    ON_DEVICE(h);
    *bytes_out = 0;
    if (n == 0) return RBPF_OK;
    DevView& v = h->v;
    const int LL = v.L * v.L;
    for (int i = 0; i < n; ++i) if (local_idx[i] < 0 || local_idx[i] >= v.P) return fail(h, RBPF_EINVAL, "local index out of range");
    int rc = scratch(h, &h->d_i32, &h->d_i32_cap, (size_t)n * (1 + 5 * LL));
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_i32, local_idx, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    launch_gather_meta(v, h->d_i32, n, h->d_i32 + n, h->stream);
    std::vector<int32_t> g((size_t)n * LL * 5);
    HIP_TRY(h, hipMemcpyAsync(g.data(), h->d_i32 + n, g.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::vector<PackJobHost> jobs;
    const int64_t off = layout_packed(v, g.data(), n, local_idx, meta_out, &jobs);
    if (off > cap_bytes) return fail(h, RBPF_ENOMEM, "pack buffer too small");
    rc = stage_jobs(h, jobs.data(), jobs.size() * sizeof(PackJobHost));
    if (rc) return rc;
    launch_pack(v, h->d_jobs, (int)jobs.size(), d_buf, h->stream);
    HIP_TRY(h, hipGetLastError());
    *bytes_out = off;
    return RBPF_OK;                                     // the buffer is filled in stream order: send it on the handle's stream
}

// ---- the same in three steps, with one host wait for a whole migration (thesis_amd/sharding.py) --------------------
// 1. rbpf_gather_pack_meta: the tile boxes of the departing particles, gathered into a device buffer (nothing waited
//    for) - the ranks exchange these records while they are still on the device and read their own and the incoming
//    ones back together;  2. rbpf_meta_from_raw: records -> the layout rows rbpf_unpack_particles takes (host only);
// 3. rbpf_pack_particles_raw: packs with the records already on the host (nothing waited for).
int32_t rbpf_pack_raw_width(rbpf_handle* h) { return h ? 5 * h->v.L * h->v.L : -1; }

int rbpf_gather_pack_meta(rbpf_handle* h, const int32_t* local_idx, int32_t n, void* d_raw) {
    if (!h || n < 0 || (n > 0 && (!local_idx || !d_raw))) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (n == 0) return RBPF_OK;
    DevView& v = h->v;
    if (n > v.P) return fail(h, RBPF_EINVAL, "more departing particles than particles");
    for (int i = 0; i < n; ++i) if (local_idx[i] < 0 || local_idx[i] >= v.P) return fail(h, RBPF_EINVAL, "local index out of range");
    int rc = scratch(h, &h->d_i32, &h->d_i32_cap, (size_t)v.P + 4);
    if (rc) return rc;
    int32_t* slot = static_cast<int32_t*>(h->ring_idx.acquire());          // pinned: the caller's array is free on return
    memcpy(slot, local_idx, (size_t)n * 4);
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, slot, 0) == hipSuccess && mapped) launch_ingest(mapped, h->d_i32, (size_t)n * 4, h->stream);
    else { (void)hipGetLastError(); HIP_TRY(h, hipMemcpyAsync(h->d_i32, slot, (size_t)n * 4, hipMemcpyHostToDevice, h->stream)); }
    h->ring_idx.submitted(h->stream);
    launch_gather_meta(v, h->d_i32, n, static_cast<int32_t*>(d_raw), h->stream);
    HIP_TRY(h, hipGetLastError());
    return RBPF_OK;
}

int rbpf_meta_from_raw(rbpf_handle* h, const int32_t* raw, int32_t n, int32_t* meta_out, int64_t* bytes_out) {
    if (!h || n < 0 || !bytes_out || (n > 0 && (!raw || !meta_out))) return RBPF_EINVAL;
    *bytes_out = n ? layout_packed(h->v, raw, n, nullptr, meta_out, nullptr) : 0;
    return RBPF_OK;
}

int rbpf_pack_particles_raw(rbpf_handle* h, const int32_t* local_idx, int32_t n, const int32_t* raw, void* d_buf,
                            int64_t cap_bytes, int64_t* bytes_out) {
    if (!h || n < 0 || !bytes_out || (n > 0 && (!local_idx || !raw || !d_buf))) return RBPF_EINVAL;
    ON_DEVICE(h);
    *bytes_out = 0;
    if (n == 0) return RBPF_OK;
    DevView& v = h->v;
    for (int i = 0; i < n; ++i) if (local_idx[i] < 0 || local_idx[i] >= v.P) return fail(h, RBPF_EINVAL, "local index out of range");
    std::vector<PackJobHost> jobs;
    const int64_t off = layout_packed(v, raw, n, local_idx, nullptr, &jobs);
    if (off > cap_bytes) return fail(h, RBPF_ENOMEM, "pack buffer too small");
    int rc = stage_jobs(h, jobs.data(), jobs.size() * sizeof(PackJobHost));
    if (rc) return rc;
    launch_pack(v, h->d_jobs, (int)jobs.size(), d_buf, h->stream);
    HIP_TRY(h, hipGetLastError());
    *bytes_out = off;
    return RBPF_OK;
}

// installs n received particles at the given local indices (after rbpf_apply_resample_local); weight <- 1.0 (main.py:77-78)
int rbpf_unpack_particles(rbpf_handle* h, const int32_t* local_idx, int32_t n, const void* d_buf, const int32_t* meta_in) {
    if (!h || n < 0 || (n > 0 && (!local_idx || !d_buf || !meta_in))) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (n == 0) return RBPF_OK;
    DevView& v = h->v;
    const int LL = v.L * v.L, W = 2 + 6 * LL;
    std::vector<UnpackJobHost> jobs;
    int64_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (local_idx[i] < 0 || local_idx[i] >= v.P) return fail(h, RBPF_EINVAL, "local index out of range");
        const int32_t* m = meta_in + (size_t)i * W;
        jobs.push_back({local_idx[i], -1, 1, 0, 0, 0, 0, 0, (long long)off});
        for (int pos = 0; pos < LL; ++pos) {
            const int32_t* mm = m + 2 + 6 * pos;
            jobs.push_back({local_idx[i], pos, mm[0], mm[1], mm[2], mm[3], mm[4], 0, (long long)(off + (int64_t)mm[5] * 16)});
        }
        off += (int64_t)m[1] * 16;
    }
    int rc = stage_jobs(h, jobs.data(), jobs.size() * sizeof(UnpackJobHost));
    if (rc) return rc;
    launch_unpack(v, h->rs, h->d_jobs, (int)jobs.size(), d_buf, h->stream);
    HIP_TRY(h, hipGetLastError());
    return RBPF_OK;                                     // no host synchronisation; device errors surface at the next check
}

// ---- state access -----------------------------------------------------------------------------------------
int rbpf_get_poses(rbpf_handle* h, double* out) {
    if (!h || !out) return RBPF_EINVAL;
    ON_DEVICE(h);
    const size_t P = h->v.P;
    std::vector<double> t(3 * P);
    HIP_TRY(h, hipMemcpyAsync(t.data(), h->v.px, P * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(t.data() + P, h->v.py, P * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(t.data() + 2 * P, h->v.pth, P * 8, hipMemcpyDeviceToHost, h->stream));
    int rc = check_device_error(h);
    for (size_t p = 0; p < P; ++p) { out[3 * p] = t[p]; out[3 * p + 1] = t[P + p]; out[3 * p + 2] = t[2 * P + p]; }
    return rc;
}

int rbpf_get_covs(rbpf_handle* h, double* out) {
    if (!h || !out) return RBPF_EINVAL;
    ON_DEVICE(h);
    const size_t P = h->v.P;
    std::vector<double> t(9 * P);
    HIP_TRY(h, hipMemcpyAsync(t.data(), h->v.cov, 9 * P * 8, hipMemcpyDeviceToHost, h->stream));
    int rc = check_device_error(h);
    for (size_t p = 0; p < P; ++p) for (int k = 0; k < 9; ++k) out[9 * p + k] = t[(size_t)k * P + p];
    return rc;
}

int rbpf_get_weights(rbpf_handle* h, double* out) {
    if (!h || !out) return RBPF_EINVAL;
    ON_DEVICE(h);
    HIP_TRY(h, hipMemcpyAsync(out, h->v.weight, (size_t)h->v.P * 8, hipMemcpyDeviceToHost, h->stream));
    return check_device_error(h);
}

int rbpf_set_state(rbpf_handle* h, const double* poses, const double* covs, const double* weights) {
    if (!h) return RBPF_EINVAL;
    ON_DEVICE(h);
    h->v.dups_valid = 0;
    const size_t P = h->v.P;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (poses) {
        std::vector<double> t(3 * P);
        for (size_t p = 0; p < P; ++p) { t[p] = poses[3 * p]; t[P + p] = poses[3 * p + 1]; t[2 * P + p] = poses[3 * p + 2]; }
        HIP_TRY(h, hipMemcpy(h->v.px, t.data(), P * 8, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->v.py, t.data() + P, P * 8, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->v.pth, t.data() + 2 * P, P * 8, hipMemcpyHostToDevice));
    }
    if (covs) {
        std::vector<double> t(9 * P);
        for (size_t p = 0; p < P; ++p) for (int k = 0; k < 9; ++k) t[(size_t)k * P + p] = covs[9 * p + k];
        HIP_TRY(h, hipMemcpy(h->v.cov, t.data(), 9 * P * 8, hipMemcpyHostToDevice));
    }
    if (weights) HIP_TRY(h, hipMemcpy(h->v.weight, weights, P * 8, hipMemcpyHostToDevice));
    return RBPF_OK;
}

int rbpf_get_dim(rbpf_handle* h, int32_t* out) {
    if (!h || !out) return RBPF_EINVAL;
    *out = h->v.dim;
    return RBPF_OK;
}

static int fetch_tab(rbpf_handle* h, int32_t particle, std::vector<int32_t>& tab) {
    if (particle < 0 || particle >= h->v.P) return fail(h, RBPF_EINVAL, "particle index out of range");
    const size_t LL = (size_t)h->v.L * h->v.L;
    int32_t slot = 0;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(&slot, h->v.slot + particle, 4, hipMemcpyDeviceToHost));
    tab.resize(LL);
    HIP_TRY(h, hipMemcpy(tab.data(), h->v.tile_tab + (size_t)slot * LL, LL * 4, hipMemcpyDeviceToHost));
    return RBPF_OK;
}

int rbpf_refresh_last_scan(rbpf_handle* h, int32_t particle) {
    if (!h) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (!h->have_scan) return fail(h, RBPF_ESTATE, "rbpf_set_scan has not been called");
    if (particle < 0 || particle >= h->v.P) return fail(h, RBPF_EINVAL, "particle out of range");
    launch_last_scan(h->v, particle, h->d_last_xy, h->stream);
    HIP_TRY(h, hipGetLastError());
    h->n_last_dev = h->v.B;
    return RBPF_OK;
}

int rbpf_export_last_scan(rbpf_handle* h, void* d_out_xy, int32_t* n_points) {
    if (!h || !d_out_xy || !n_points) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (h->n_last_dev < 0) return fail(h, RBPF_ESTATE, "no device-resident previous scan");
    HIP_TRY(h, hipMemcpyAsync(d_out_xy, h->d_last_xy, (size_t)h->n_last_dev * 16, hipMemcpyDeviceToDevice, h->stream));
    *n_points = h->n_last_dev;
    return RBPF_OK;
}

int rbpf_import_last_scan(rbpf_handle* h, const void* d_xy, int32_t n_points) {
    if (!h || !d_xy || n_points < 0 || n_points > h->cfg.max_beams) return RBPF_EINVAL;
    ON_DEVICE(h);
    HIP_TRY(h, hipMemcpyAsync(h->d_last_xy, d_xy, (size_t)n_points * 16, hipMemcpyDeviceToDevice, h->stream));
    h->n_last_dev = n_points;
    return RBPF_OK;
}

int rbpf_get_rng_state(rbpf_handle* h, uint64_t* scan_updates, uint64_t* resample_draws) {
    if (!h || !scan_updates || !resample_draws) return RBPF_EINVAL;
    *scan_updates = h->scan_updates; *resample_draws = h->resample_draws;
    return RBPF_OK;
}

int rbpf_set_rng_state(rbpf_handle* h, uint64_t scan_updates, uint64_t resample_draws) {
    if (!h) return RBPF_EINVAL;
    h->scan_updates = scan_updates; h->resample_draws = resample_draws;
    return RBPF_OK;
}

int rbpf_get_tile_count(rbpf_handle* h, int32_t particle, int32_t* out_n) {
    if (!h || !out_n) return RBPF_EINVAL;
    ON_DEVICE(h);
    std::vector<int32_t> tab;
    int rc = fetch_tab(h, particle, tab);
    if (rc) return rc;
    int n = 0;
    for (int32_t t : tab) n += t >= 0;
    *out_n = n;
    return RBPF_OK;
}

int rbpf_get_tile(rbpf_handle* h, int32_t particle, int32_t k, double* centre2, int8_t* cells) {
    if (!h || !centre2 || !cells) return RBPF_EINVAL;
    ON_DEVICE(h);
    std::vector<int32_t> tab;
    int rc = fetch_tab(h, particle, tab);
    if (rc) return rc;
    const DevView& v = h->v;
    int n = 0;
    for (int a = 0; a < v.L; ++a)
        for (int b = 0; b < v.L; ++b) {
            int32_t t = tab[(size_t)a * v.L + b];
            if (t < 0) continue;
            if (n++ == k) {
                centre2[0] = (double)(a - v.R) * v.tile_len;
                centre2[1] = (double)(b - v.R) * v.tile_len;
                HIP_TRY(h, hipMemcpy(cells, v.pool + (size_t)t * v.dim * v.dim, (size_t)v.dim * v.dim, hipMemcpyDeviceToHost));
                return RBPF_OK;
            }
        }
    return fail(h, RBPF_EINVAL, "tile index out of range");
}

int rbpf_set_tile(rbpf_handle* h, int32_t particle, double cx, double cy, const int8_t* cells) {
    if (!h || !cells) return RBPF_EINVAL;
    ON_DEVICE(h);
    h->v.dups_valid = 0;
    std::vector<int32_t> tab;
    int rc = fetch_tab(h, particle, tab);
    if (rc) return rc;
    DevView& v = h->v;
    double fa = cx / v.tile_len, fb = cy / v.tile_len;
    int a = (int)lround(fa), b = (int)lround(fb);
    if (fabs(fa - a) > 1e-9 || fabs(fb - b) > 1e-9 || abs(a) > v.R || abs(b) > v.R)
        return fail(h, RBPF_EINVAL, "tile centre must lie on the tile lattice inside lattice_radius");
    // the reference's cells never leave [min_odds_emp, max_odds_occ] (gridmap.py:86-117); the map kernels rely on it
    for (size_t i = 0, n = (size_t)v.dim * v.dim; i < n; ++i)
        if (cells[i] < v.cc.vmin || cells[i] > v.cc.vmax) return fail(h, RBPF_EINVAL, "cell value outside [min_odds_emp, max_odds_occ]");
    const size_t LL = (size_t)v.L * v.L, idx = (size_t)(a + v.R) * v.L + (b + v.R);
    int32_t t = tab[idx];
    if (t < 0) {   // pop a tile from the free stack on the host side
        int32_t top = 0;
        HIP_TRY(h, hipMemcpy(&top, v.free_top, 4, hipMemcpyDeviceToHost));
        if (top <= 0) return fail(h, RBPF_ENOMEM, "tile pool exhausted");
        --top;
        HIP_TRY(h, hipMemcpy(&t, v.free_stack + top, 4, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(v.free_top, &top, 4, hipMemcpyHostToDevice));
        int32_t slot = 0;
        HIP_TRY(h, hipMemcpy(&slot, v.slot + particle, 4, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(v.tile_tab + (size_t)slot * LL + idx, &t, 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(h, hipMemcpy(v.pool + (size_t)t * v.dim * v.dim, cells, (size_t)v.dim * v.dim, hipMemcpyHostToDevice));
    {   // occupancy bitmask of the uploaded cells (cell > threshold, gridmap.py:153)
        std::vector<uint32_t> occ((size_t)v.dim * v.ow, 0u);
        for (int x = 0; x < v.dim; ++x)
            for (int y = 0; y < v.dim; ++y)
                if ((int)cells[(size_t)x * v.dim + y] > v.cc.thr) occ[(size_t)x * v.ow + (y >> 5)] |= 1u << (y & 31);
        HIP_TRY(h, hipMemcpy(v.occ + (size_t)t * v.dim * v.ow, occ.data(), occ.size() * 4, hipMemcpyHostToDevice));
    }
    int32_t bb[4] = {0, v.dim - 1, 0, v.dim - 1};       // unknown content: the whole tile counts as written
    HIP_TRY(h, hipMemcpy(v.tile_bbox + 4 * (size_t)t, bb, sizeof(bb), hipMemcpyHostToDevice));
    return RBPF_OK;
}

int rbpf_get_odds_at(rbpf_handle* h, int32_t particle, const double* xy, int32_t n, double* out_vals, uint8_t* out_none) {
    if (!h || !xy || !out_vals || !out_none || n < 0) return RBPF_EINVAL;
    ON_DEVICE(h);
    if (particle < 0 || particle >= h->v.P) return fail(h, RBPF_EINVAL, "particle index out of range");
    if (n == 0) return RBPF_OK;
    double *d_xy = nullptr, *d_v = nullptr; uint8_t* d_n = nullptr;
    HIP_TRY(h, hipMalloc((void**)&d_xy, (size_t)n * 16));
    HIP_TRY(h, hipMalloc((void**)&d_v, (size_t)n * 8));
    HIP_TRY(h, hipMalloc((void**)&d_n, (size_t)n));
    hipMemcpyAsync(d_xy, xy, (size_t)n * 16, hipMemcpyHostToDevice, h->stream);
    launch_get_odds(h->v, particle, d_xy, n, d_v, d_n, h->stream);
    hipMemcpyAsync(out_vals, d_v, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(out_none, d_n, (size_t)n, hipMemcpyDeviceToHost, h->stream);
    int rc = check_device_error(h);
    hipFree(d_xy); hipFree(d_v); hipFree(d_n);
    return rc;
}

}  // extern "C"
