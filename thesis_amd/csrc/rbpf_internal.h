// rbpf_internal.h -- handle layout and kernel-launch prototypes of librbpf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rbpf_hip.h"
#include "rbpf_math.h"

namespace rbpf {

// The dynamic-LDS limit of a kernel is a per-DEVICE attribute; a process may hold handles on several GPUs.  `set` is the
// launcher's own table of what it has set so far (one entry per device).
static const int MAX_DEVICES = 16;
inline void ensure_dynamic_lds(const void* fn, size_t bytes, size_t* set) {
    static std::mutex mu;                              // handles on several host threads share the launchers' tables
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool tracked = dev >= 0 && dev < MAX_DEVICES;
    if (tracked && bytes <= set[dev]) return;
    const hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (rc != hipSuccess) { (void)hipGetLastError(); return; }   // not recorded as set: the launch that follows reports the error
    if (tracked) set[dev] = bytes;
}


static const int BLOCK = 256;          // 4 waves of 64
static const int WIN = 128;            // LDS window edge (storage cells)
static const int MAX_ITEMS_PER_PARTICLE = 64;

// beam range classes, computed on the host from the float64 distance (rbpf_set_scan)
enum { BF_WEIGHT = 1, BF_MATCH = 2, BF_LONG = 4, BF_MATCH_ADJ = 8 };

// Everything a kernel needs, passed by value.
struct DevView {
    // configuration
    int P, K, B, dim, R, L;            // L = 2R+1 lattice edge
    int pool_tiles;
    int reach;                         // longest ray in cells (+ margin)
    double cs, tile_len;
    double quantum, inv_quantum;       // inv_quantum = round(1/quantum) when exact, else 0
    CellConsts cc;
    double w_min_range, w_max_range;
    // LUT over global cell indices g in [g_min, g_min + n_lut)
    const uint32_t* lut; int g_min, n_lut;
    const int32_t* gwin;               // [L][KW+1] first global index of each 128-cell window of a lattice row (KW = ceil(dim/WIN)); [KW] = next tile
    // particle state, SoA, logical particle order
    double *px, *py, *pth;             // [P]
    double *cov;                       // [9][P]
    double *weight;                    // [P]
    int32_t* slot;                     // [P] logical particle -> map slot
    int32_t* global_id;                // [P] id of the particle in the multi-GPU job
    // maps
    int32_t* tile_tab;                 // [P slots][L*L] pool tile id or -1
    int8_t*  pool;                     // [pool_tiles][dim*dim], cell[x*dim + y]
    uint32_t* occ;                     // [pool_tiles][dim][ow] occupancy bits (cell > threshold), bit y&31 of word y>>5
    int ow;                            // words per occupancy row = ceil(dim / 32)
    int32_t* tile_bbox;                // [pool_tiles][4] x_min, x_max, y_min, y_max (inclusive)
    int32_t* free_stack;               // [pool_tiles]
    int32_t* free_top;                 // [1] number of free tiles on the stack
    // scan (sensor frame)
    const double *bx, *by, *bscale;    // [B]
    const uint8_t* bflags;             // [B]
    float *msel_x, *msel_y; int n_msel;  // beams with BF_MATCH, compacted (metres, sensor frame)
    float *asel_x, *asel_y; int n_asel;  // beams with BF_MATCH_ADJ, compacted
    const float *wsel_x, *wsel_y; const uint16_t* wsel_idx; int n_wsel;   // beams with BF_WEIGHT, compacted, single precision (x = NaN: outside
                                         // the fast look-up's error budget), padded with NaN to a multiple of 64; their beam numbers
    // per-update scratch
    double*  upd_pose;                 // [3][P] poses used by the current map update
    double*  prop_prep;                // [P][24] proposal frame of the current scan update (kernels_propose.hip: U, A, mean, log c)
    double*  prop_samp;                // [P][256] its K samples: cos, sin, pose, motion probability, single-precision frame
    int32_t* mu_fallback;              // [P] != 0: the map-update kernel that ran first gave the particle back to the next one
    int mu_mode;                       // 0 = event-walk kernel (kernels_mapev.hip; the global-index kernel of round 2, kernels_mapray.hip, where
                                       // that one is not available: more than 1536 beams), then 128x128 windows for what it gave back;
                                       // 1 = 128x128 windows only; 3 = global-index kernel, then windows; 5 = event-walk kernel, then windows
    uint32_t* ndt_occ; double* ndt_aux; // NDT stage: the matcher's staged field per particle, its grid optimum (kernels_match.hip)
    int ndt_refine;                    // rbpf_config.ndt_refine: NDT stage of matchScanCustom.m:32-50 (0 off, 1 reference rule, 2 always)
    int32_t* dup_of; int dups_valid;    // representative of each particle's group of exact duplicates since the last resample (kernels_resample.hip); valid until the next proposal
    float wsafe_override;              // >= 0: the weighting's guard band in cells (RBPF_WSAFE, a test knob); < 0: the built-in value
    int weight_entry_f64;              // 1: rbpf_weight_samples runs the float64 kernel of round 1 (RBPF_WEIGHT_ENTRY=f64) instead of the product's look-ups
    int match_stage_slow;              // 1 = the matcher stages its field bit by bit (RBPF_MATCH_STAGE=slow; the check of the fast path)
    unsigned long long* stats;         // [8] device counters
    int32_t* err;                      // [1] sticky device error code
};

// double buffers and scratch of the resample step
struct ResampleBuffers {
    int32_t *T, *idx, *did;            // [P], [P], [1]
    int32_t *slot2, *dead_list, *jobs; // [P], [P], [P][2]
    int32_t *n_jobs;                   // [2] job count, queue head
    int32_t *pending_free, *n_pending; // [pool_tiles], [1]
    double *px2, *py2, *pth2, *cov2, *w2;
};

enum { ST_RAY_CELLS = 0, ST_CELLS_WRITTEN = 1, ST_GATHERS = 2, ST_SLOW_CELLS = 3,
       ST_COPIES = 4, ST_COPY_BYTES = 5, ST_WINDOW_FALLBACKS = 6, ST_FALLBACK_REASONS = 7,
       /* 8..15: phase stamps of a -DRBPF_STAMPS build */
       ST_NDT_RUNS = 16, ST_NDT_EVALS = 17, ST_NDT_ACCEPTED = 18, ST_MATCH_SHARED = 19, ST_MAP_WINDOWS = 20,
       ST_FB_BOUND = 21, ST_FB_TABLES = 22, ST_MAP_EVENTS = 23, ST_EV_OVERFLOWS = 24, ST_COUNT = 28 };   // ST_FALLBACK_REASONS (7) counts reason 1 (geometry / index map)

}  // namespace rbpf

struct rbpf_handle {
    rbpf_config cfg;
    rbpf::DevView v;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool profiling = false;
    unsigned prof_mask = 0;                     // kernel families whose launches are bracketed by timing events (bit k = family k)
    bool have_scan = false;
    bool dedup_enabled = true;                  // exact duplicates share one matcher run (RBPF_MATCH_DEDUP=0 turns it off)
    std::string err;
    std::vector<uint32_t> h_lut;
    std::vector<void*> allocs;
    // host staging
    // pinned staging rings for the per-step uploads (scan block, previous scan): a slot is reused only after the copy
    // that read it has completed (its event), so uploading never drains the stream
    struct PinnedRing {
        static const int N = 4;
        unsigned char* base = nullptr; size_t slot_bytes = 0; int next = 0; hipEvent_t ev[N] = {}; bool used[N] = {};
        void* acquire() { if (used[next]) (void)hipEventSynchronize(ev[next]); return base + (size_t)next * slot_bytes; }
        void submitted(hipStream_t s) { (void)hipEventRecord(ev[next], s); used[next] = true; next = (next + 1) % N; }
    };
    PinnedRing ring_scan, ring_last, ring_idx;
    hipEvent_t ev_weights = nullptr; bool ev_weights_valid = false, record_ev_weights = false, begin_seen = false;   // recorded after the weighting kernel of rbpf_scan_update_begin
    int32_t* d_did_early = nullptr; bool scan_begun = false;
    void* h_jobs = nullptr; size_t h_jobs_bytes = 0; hipEvent_t ev_jobs = nullptr; bool h_jobs_used = false;   // pinned job-list staging
    hipEvent_t ev_early = nullptr; void* h_early = nullptr; size_t h_early_bytes = 0; int early_n = 0;   // early resample read-back
    unsigned char* d_scan = nullptr; size_t scan_bytes = 0;   // device scan block, same layout as a ring_scan slot
    // scratch device buffers for test entries
    double* d_guess = nullptr; double* d_prs = nullptr; double* d_w = nullptr; size_t d_guess_n = 0;
    int mN = 0, mds = 1, mncr = 0; double mmcs = 0, md0 = 0; size_t mlds = 0;
    double* d_last_xy = nullptr; float* d_tmp_sel = nullptr;
    int n_last_dev = -1;                       // points of the device-resident previous scan (rbpf_refresh_last_scan), -1 = none
    double* d_match = nullptr; uint8_t* d_bad = nullptr; double* d_guess_full = nullptr;
    unsigned long long resample_draws = 0;
    int32_t* d_gT = nullptr; size_t d_gT_cap = 0; int32_t* d_gidx = nullptr; size_t d_gidx_cap = 0;
    int32_t* d_i32 = nullptr; size_t d_i32_cap = 0; unsigned char* d_jobs = nullptr; size_t d_jobs_cap = 0;
    // profiling: a ring of HIP-event pairs per kernel family, recorded on the handle's stream
    static const int N_KERN = 5, RING = 512;        // 0 map update, 1 propose/weight, 2 resample, 3 match (grid stage), 4 match (NDT stage)
    std::vector<hipEvent_t> ring[N_KERN][2];
    int ring_n[N_KERN] = {0, 0, 0, 0, 0};
    std::vector<hipEvent_t> begin_used[N_KERN];     // the event that marks a launch's start: its own, or the previous family's end
    hipEvent_t last_end = nullptr;
    void prof_begin(int k) { if (!((prof_mask >> k) & 1u)) return; hipEvent_t e = ring[k][0][ring_n[k] % RING]; (void)hipEventRecord(e, stream); begin_used[k][ring_n[k] % RING] = e; }
    // the previous timed family ended right before this one starts (nothing enqueued in between): one record serves both
    void prof_begin_chained(int k) { if (!((prof_mask >> k) & 1u)) return; if (!last_end) { prof_begin(k); return; } begin_used[k][ring_n[k] % RING] = last_end; }
    void prof_end(int k) { if (!((prof_mask >> k) & 1u)) return; last_end = ring[k][1][ring_n[k] % RING]; (void)hipEventRecord(last_end, stream); ring_n[k]++; }
    rbpf::ResampleBuffers rs;
    rbpf_counters counters;
    unsigned long long scan_updates = 0;
};

namespace rbpf {
// kernel launchers (one translation unit per kernel family)
void launch_weight_samples(const DevView& v, const double* d_guesses, const double* d_prs, int K,
                           double* d_out_w, hipStream_t s);
void launch_weight_samples_product(const DevView& v, const double* d_guesses, const double* d_prs, int K, double* d_out_w, hipStream_t s);
int map_update_first_kernel(const DevView& v);   // 0 none, 1 event walk, 2 global-index kernel
void launch_map_update_fused(const DevView& v, const uint8_t* d_bad, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);   // picks the kernel(s) below; d_bad: NaN-branch weight increments after the update (or nullptr)
void launch_ingest(const void* mapped_src, void* d_dst, size_t bytes, hipStream_t s);   // bytes rounded up to 16
void launch_ingest2(const int32_t* mapped_a, int32_t* d_a, const int32_t* mapped_b, int32_t* d_b, int n, hipStream_t s);
void launch_readback(void* mapped_dst, const double* d_nan_elem, const int32_t* d_did, const int32_t* d_idx, int n, hipStream_t s);
bool map_update_ray_available(const DevView& v);
void launch_map_update_ray(const DevView& v, const int32_t* only, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
bool map_update_ev_available(const DevView& v);
void launch_map_update_ev(const DevView& v, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void launch_get_odds(const DevView& v, int particle, const double* d_xy, int n, double* d_vals,
                     uint8_t* d_none, hipStream_t s);
void launch_last_scan(const DevView& v, int particle, double* d_out_xy, hipStream_t s);
void launch_imu_update(const DevView& v, int model, double d0, double d1, double d2, double dt_ticks,
                       const double* vel_noise, hipStream_t s);
void launch_resample_indices(int P, const double* d_w, double u, double spread, int32_t* d_T, int32_t* d_idx,
                             int32_t* d_did, int32_t* d_err, hipStream_t s);
void launch_resample_apply(const DevView& v, const ResampleBuffers& b, hipStream_t s);
void launch_resample_local(const DevView& v, const ResampleBuffers& b, const double* d_w, double u, double spread, hipStream_t s);
void launch_resample_apply_sources(const DevView& v, const ResampleBuffers& b, hipStream_t s);
void launch_export_weights(const DevView& v, double* d_out, int n_global, const uint8_t* d_bad, hipStream_t s);
void launch_sources_to_T(int P, const int32_t* d_idx, int32_t* d_T, int32_t* d_did, hipStream_t s);
void launch_gather_meta(const DevView& v, const int32_t* d_local, int n, int32_t* d_out, hipStream_t s);
void launch_pack(const DevView& v, const void* d_jobs, int n_jobs, void* d_buf, hipStream_t s);
void launch_unpack(const DevView& v, const ResampleBuffers& b, const void* d_jobs, int n_jobs, const void* d_buf, hipStream_t s);
struct PackJobHost { int32_t particle, tile, x0, x1, ya, yb; long long off; };
struct UnpackJobHost { int32_t particle, pos, has, x0, x1, ya, yb, pad; long long off; };
void launch_propose_weight(const DevView& v, const double* d_match, const int32_t* d_match_of, const double* d_guesses, uint8_t* d_bad,
                           uint64_t seed, uint32_t stream, double* d_dbg_w, hipStream_t s);
void launch_bad_weight(const DevView& v, const uint8_t* d_bad, hipStream_t s);
size_t match_lds_bytes(int N, int B, int n_coarse, int per_rot);
int match_sc_capacity(int N, int B, size_t lds);
int match_per_rot(double max_range_m, double mcs);
size_t ndt_lds_bytes(int N, int B);
int ndt_cells(double mcs);
void match_geometry(const rbpf_config& c, double cell_size, int& N, int& ds, double& mcs, double& d0, int& n_coarse_rot);
int match_max_coarse(int n_coarse_rot, double max_range_m, double mcs);
bool launch_match_particles(const DevView& v, int mode, const double* d_ref, int n_ref, double* d_out, int N, int ds,
                            double mcs, double d0, int ncr, double max_range, int cap_sel, size_t lds, int stage, hipStream_t s,
                            hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);   // t0 / t1: timing events carried by the stage's kernel dispatch
void launch_match_single(const DevView& v, const double* d_ref, int n_ref, const double* guess3, const double* range3,
                         const float* d_sel_x, const float* d_sel_y, int n_sel, double* d_out, int N, int ds, double mcs,
                         double d0, int ncr, int cap_sel, size_t lds, uint32_t* d_ndt_occ, double* d_ndt_aux, hipStream_t s);
void launch_match_inputs(const DevView& v, int particle, const double* guess3, double* d_all_curr, int* d_counts,
                         uint32_t* d_mask, int* d_row_cnt, double* d_ref, int cap_ref, double* d_curr, int win,
                         double match_max, hipStream_t s);
size_t raycast_lds_bytes(int B, int reach);
}  // namespace rbpf
