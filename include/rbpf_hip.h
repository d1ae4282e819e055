/*
 * rbpf_hip.h -- C ABI of librbpf_hip.so, the MI355X (gfx950) RBPF-SLAM particle-update engine.
 *
 * This is the drop-in boundary for the reference's per-particle hot path (reference =
 * amansanghvi/Thesis, cited file:line).  The reference has no FFI; its seams are Python
 * methods called once per particle from list comprehensions (main.py:144,157-160).  Each
 * entry point below replaces one of those per-particle methods by ONE batched call over all
 * particles of a handle.  The Python mirror of the reference's classes (thesis_amd/) binds
 * these symbols with ctypes; INTEGRATION.md shows the stub a maintainer of the reference
 * would add.
 *
 * Conventions
 *   - every function returns 0 (RBPF_OK) or a negative RBPF_E* code; the message is
 *     available from rbpf_last_error();
 *   - the caller owns every host buffer passed in or out (C-contiguous, float64 / int32 /
 *     int8); the library owns all device memory behind the opaque handle; no pointer is
 *     retained after a call returns; device pointers handed over explicitly (the rbpf_export_* /
 *     rbpf_resample_indices_global* / rbpf_pack_* calls) are used in stream order by the work that call queues;
 *   - calls on one handle are not re-entrant; one HIP stream per handle (rbpf_set_stream);
 *   - every call runs on the handle's device (rbpf_config.device) and leaves the caller's current
 *     HIP device as it found it;
 *   - a soft scan-matcher failure is not an error: it is reported per particle as NaN
 *     covariance and the engine takes the reference's fallback branch (robot.py:73-78).
 *
 * Map cells are stored as int8 multiples of `quantum` (0.1 for the reference's constants
 * +0.8/+0.2/-0.3 clamped to [-3,3], gridmap.py:20-24): in exact arithmetic every reachable
 * log-odds value is such a multiple, the reference's float64 cells deviate from it only by
 * accumulated rounding noise (< 1e-12).  rbpf_create() fails with RBPF_EINVAL if a constant
 * is not an integer multiple of `quantum` or does not fit in int8.
 */
#ifndef RBPF_HIP_H
#define RBPF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBPF_OK 0
#define RBPF_EINVAL (-1)    /* bad argument / configuration */
#define RBPF_ENOMEM (-2)    /* device or tile-pool exhaustion */
#define RBPF_EDEVICE (-3)   /* HIP runtime error (message has the HIP string) */
#define RBPF_ESTATE (-4)    /* call order violated (e.g. no scan set) */
#define RBPF_ERANGE (-5)    /* a pose left the addressable tile lattice */

/* motion-model families (IMUData.py:9-40 callbacks; a2 of SURVEY.md section 8) */
#define RBPF_IMU_UNICYCLE 0          /* DefaultIMUData.py:26-54: data = (speed, omega)          */
#define RBPF_IMU_ABSOLUTE 1          /* IntelIMUData.py:23-36:   data = (x, y, theta) passthrough */
#define RBPF_IMU_VELOCITY 2          /* Freid101IMUData.py:34-55: data = (vx, vy, omega)         */

typedef struct rbpf_handle rbpf_handle;

typedef struct rbpf_config {
    int32_t n_particles;      /* P: particles held by this handle (main.py:44 NUM_PARTICLES)      */
    int32_t n_samples;        /* K: proposal samples per particle (robot.py:17, 30)               */
    int32_t max_beams;        /* upper bound on beams per scan (<= 4095)                          */
    int32_t tile_len_m;       /* tile edge in metres (hybridmap.py:68, 40)                        */
    double  cell_size;        /* metres per cell (hybridmap.py:67, 0.05)                          */
    int32_t lattice_radius;   /* tiles addressable per axis: -R..R (default 3 => +-140 m)         */
    int32_t pool_tiles;       /* tile-pool capacity, 0 => 2*P                                     */
    double  log_odds_occ;     /* gridmap.py:20  +0.80                                             */
    double  log_odds_nearby;  /* gridmap.py:21  +0.20                                             */
    double  max_odds_occ;     /* gridmap.py:22  +3.0                                              */
    double  log_odds_emp;     /* gridmap.py:23  -0.30                                             */
    double  min_odds_emp;     /* gridmap.py:24  -3.0                                              */
    double  quantum;          /* cell unit; the five constants above must be multiples (0.1)      */
    double  occupied_threshold; /* gridmap.py:17 / hybridmap.py:18  1.0 (strict >)                */
    double  max_ray_m;        /* hybridmap.py:107 rays longer than this are shortened (15.0)      */
    double  weight_min_range; /* robot.py:130  0.01                                               */
    double  weight_max_range; /* robot.py:130  25.0                                               */
    double  match_min_range;  /* hybridmap.py:218  1e-3                                           */
    double  match_max_range;  /* hybridmap.py:20,218  11.0                                        */
    double  resample_spread;  /* main.py:50  200.0                                                */
    double  vel_noise[4];     /* RBPF_IMU_VELOCITY Q: (a0 + a1|v|dt)^2, (b0 deg + b1|w|dt)^2;
                                 Freid101IMUData.py:51-55 => {0.02, 0.01, 0.2, 0.02}             */
    int32_t device;           /* HIP device ordinal                                               */
    int32_t ndt_refine;       /* second matcher stage, matchScanCustom.m:32-50 (NDT, CellSize 0.1 m, 500 iterations):
                                 0 off; 1 (default) on, accepted by the reference's rule (valid pose and
                                 2*ndtScore > gridScore);
                                 2 on, every valid NDT pose is taken (diagnostic).  Inactive when the matcher cell
                                 is 0.1 m or coarser (no NDT cell can hold the 3 points a Gaussian needs).          */
    uint64_t seed;            /* Philox seed for on-device proposal sampling                      */
} rbpf_config;

typedef struct rbpf_counters {
    /* cumulative since rbpf_create or the last rbpf_set_profiling call: */
    uint64_t scan_updates;        /* rbpf_scan_update/rbpf_map_update calls                       */
    uint64_t ray_cells_visited;   /* sum of Bresenham points over all rays                        */
    uint64_t cells_written;       /* unique cells written per update, summed (|W| over particles) */
    uint64_t cells_gathered;      /* map cells read by the weighting kernels (K*B gathers per particle-update) */
    uint64_t tiles_in_use;        /* tiles allocated from the pool (current)                      */
    uint64_t resample_copies;     /* tile copies made by resampling                               */
    uint64_t bytes_copied;        /* bytes moved by those copies (read + write)                   */
    double   ms_raycast;          /* HIP-event time of the last ray-cast kernel                   */
    double   ms_weight;           /* ... of the last weighting kernel                             */
    double   ms_match;            /* ... of the last scan-match kernels                           */
    double   ms_resample;         /* ... of the last resample (plan + copies)                     */
    uint64_t slow_cells;          /* flagged cells replayed by the exact membership scan          */
    uint64_t reserved[7];         /* phase cycle sums of a -DRBPF_STAMPS diagnostic build, else 0 */
    uint64_t window_fallbacks;    /* particles the first map-update kernel handed to the 128x128-window kernel */
    uint64_t ndt_runs;            /* matches that entered the NDT stage                           */
    uint64_t ndt_evaluations;     /* NDT score/gradient/Hessian evaluations, summed over runs     */
    uint64_t ndt_accepted;        /* runs whose pose replaced the grid pose (matchScanCustom.m:39-41) */
    uint64_t match_shared;        /* particles that took the match result of an exact duplicate (a copy made by the
                                     last resample: same pose, covariance and map) instead of repeating the search */
    uint64_t fallback_reasons;    /* why particles left the first map-update kernel, packed: three 16-bit tallies
                                     (geometry / index map, counter bound, event tables), each SATURATING at 65535;
                                     the exact tallies are fallback_geometry / _bound / _tables below             */
    double   ms_ndt;              /* HIP-event time of the last NDT-stage kernel                  */
    uint64_t stamp7;              /* eighth phase stamp of a -DRBPF_STAMPS diagnostic build, else 0 */
    uint64_t map_windows;         /* LDS windows the map update processed, summed over particles (1 per particle when
                                     the whole ray fan fits one window)                                            */
    uint64_t fallback_geometry;   /* particles handed to the 128x128-window kernel: fan / index-map form / LDS rows  */
    uint64_t fallback_bound;      /* ... the 8-bit hit fields could overflow (slope-bucket bound)                   */
    uint64_t fallback_tables;     /* ... event tables full (global-index kernel)                                    */
    uint64_t map_events;          /* event-walk kernel: passes over cells that also got an occupied / nearby hit in
                                     the same scan, found by the walk's returning adds, summed over particles       */
    uint64_t map_event_overflows; /* ... particles whose list of such passes was full (every flagged cell of theirs
                                     was then replayed by the exact membership test)                               */
} rbpf_counters;

/* ---- lifecycle ------------------------------------------------------------------------- */
int  rbpf_default_config(rbpf_config* cfg);
/* Robot.__init__ x P (robot.py:20-28): pose 0, cov 0, weight 1, one empty tile centred (0,0). */
int  rbpf_create(const rbpf_config* cfg, rbpf_handle** out);
int  rbpf_destroy(rbpf_handle* h);
const char* rbpf_last_error(const rbpf_handle* h);   /* h may be NULL (create failures)      */
int  rbpf_set_stream(rbpf_handle* h, void* hip_stream); /* e.g. torch's current stream (borrowed: the handle
                                                           never destroys it)                       */
/* gives a borrowed stream back (waits for what this handle queued on it); the handle then works on a stream of its
 * own again.  Call it before the owner of the stream goes away, or simply before rbpf_destroy. */
int  rbpf_release_stream(rbpf_handle* h);
/* sizeof(rbpf_config), sizeof(rbpf_counters) as this library was built: a binding checks its struct layouts */
int  rbpf_abi_struct_bytes(int32_t* config_bytes, int32_t* counters_bytes);
int  rbpf_synchronize(rbpf_handle* h);
int  rbpf_get_counters(rbpf_handle* h, rbpf_counters* out);
int  rbpf_set_profiling(rbpf_handle* h, int on);     /* per-kernel HIP events; resets the rings */
/* the same for a subset of the kernel families (bit k = family k of rbpf_get_kernel_ms): every record costs a few
 * microseconds of stream time, a benchmark brackets only what it needs inside its timed region */
int  rbpf_set_profiling_families(rbpf_handle* h, uint32_t mask);
/* durations (ms) of the launches recorded since rbpf_set_profiling, HIP events on the handle's stream;
 * which: 0 ray-cast map-update kernel, 1 proposal/weighting kernel, 2 resample kernels, 3 scan-match grid stage,
 * 4 scan-match NDT stage.  Synchronises the stream. */
int  rbpf_get_kernel_ms(rbpf_handle* h, int32_t which, double* out_ms, int32_t cap, int32_t* n_out);

/* ---- a1: scan geometry (Scan.__init__, lidar.py:76-80) ------------------------------------ */
/* ranges[B], angles[B] -> sensor-frame endpoints (host libm cos/sin, as the reference), the
 * per-beam range classes (robot.py:130, hybridmap.py:107,218) and the upload. */
int  rbpf_set_scan(rbpf_handle* h, const double* ranges, const double* angles, int32_t n_beams);
/* the same from the end points a reference Scan object holds (Scan.x(), Scan.y(), lidar.py:82-87): what the per-object
 * facade (thesis_amd/dropin.py) has in hand when main.py:157 passes it a Scan */
int  rbpf_set_scan_xy(rbpf_handle* h, const double* x, const double* y, int32_t n_beams);

/* ---- a2: Robot.imu_update (robot.py:45-57) for every particle ------------------------------ */
int  rbpf_imu_update(rbpf_handle* h, int32_t model, const double* data3, double dt_ticks);

/* ---- a4: Robot._generate_sample_weight (robot.py:118-139), test entry ------------------------ */
/* guesses[P*K*3], motion_prs[P*K] -> out_w[P*K] against each particle's own map. */
int  rbpf_weight_samples(rbpf_handle* h, const double* guesses, const double* motion_prs,
                         int32_t n_samples, double* out_w);

/* ---- a5: HybridMap.update (hybridmap.py:95-145), test entry ----------------------------------- */
/* poses[P*3] or NULL (= each particle's latest pose). */
int  rbpf_map_update(rbpf_handle* h, const double* poses);

/* ---- a3+a4+a5(+a6/a7): Robot.map_update (robot.py:59-115) for every particle ------------------ */
/* adj            : main.py:156-159 (0 = match against own map, 1 = against last_scan_xy)
 * last_scan_xy   : [n_last*2] global endpoints of the reference scan (adj=1), else NULL
 * match_override : NULL => built-in correlative matcher; else [P*13] = pose(3), cov(9), score(1)
 *                  per particle, as returned by Map.get_scan_match (engine-seam double)
 * guesses        : NULL => on-device Philox proposal; else [P*K*3] explicit samples
 *                  (replaces np.random.multivariate_normal, robot.py:81)                       */
int  rbpf_scan_update(rbpf_handle* h, int32_t adj, const double* last_scan_xy, int32_t n_last,
                      const double* match_override, const double* guesses);
/* main.py:167-168: last_scan = scan.from_global_reference(particles[0].get_latest_pose()).  Computed on the device
 * from the current scan and the pose of `particle`, and kept there: a later rbpf_scan_update with adj = 1 and
 * last_scan_xy = NULL uses it, so the driver loop never has to read a pose back.  rbpf_export_last_scan /
 * rbpf_import_last_scan copy it to / from a device buffer of max_beams * 2 doubles (stream-ordered), for the rank
 * that owns particle 0 to broadcast it in a multi-GPU job. */
int  rbpf_refresh_last_scan(rbpf_handle* h, int32_t particle);
int  rbpf_export_last_scan(rbpf_handle* h, void* d_out_xy, int32_t* n_points);
int  rbpf_import_last_scan(rbpf_handle* h, const void* d_xy, int32_t n_points);
/* The same in two halves, for a driver that wants the weights as early as possible (multi-GPU resampling):
 * _begin = scan matcher, proposal, weighting, moments (robot.py:62-114): the weights are final here unless a particle
 * took the NaN-covariance branch; _end = the map update at the new mean pose and that branch (robot.py:115, 73-78). */
int  rbpf_scan_update_begin(rbpf_handle* h, int32_t adj, const double* last_scan_xy, int32_t n_last,
                            const double* match_override, const double* guesses);
int  rbpf_scan_update_end(rbpf_handle* h);

/* ---- a6/a7: scan matcher, stateless twin of the engine seam (hybridmap.py:244-251) ------------ */
int  rbpf_match_scan(rbpf_handle* h, const double* curr_xy, int32_t n_curr, const double* ref_xy,
                     int32_t n_ref, const double* guess3, int32_t cells_per_m,
                     const double* pose_range3, double* pose_out3, double* cov_out9,
                     double* score_out);
/* matcher inputs built from particle p's map (hybridmap.py:210-242), test/inspection entry */
int  rbpf_match_inputs(rbpf_handle* h, int32_t particle, const double* guess3, double* curr_xy,
                       int32_t* n_curr, double* ref_xy, int32_t* n_ref, int32_t cap_ref);

/* ---- a8+a9: resample (main.py:46-79) ---------------------------------------------------------- */
/* u in [0,1) replaces np.random.random() (main.py:59); NaN => internal Philox draw.
 * idx_out[P] (may be NULL) receives the ancestor index of every new particle. */
int  rbpf_resample(rbpf_handle* h, double u, int32_t* idx_out, int32_t* did_resample);

/* multi-GPU pieces (one handle per rank; the collectives themselves are the caller's, over RCCL).
 * Every particle carries a global id (0 .. n_global-1, its index in the reference's particle list); the
 * Philox proposal streams are keyed by it, so results do not depend on which rank holds a particle. */
int  rbpf_set_global_ids(rbpf_handle* h, const int32_t* ids_p);
/* zero a device vector of n_global doubles and scatter the local weights into it at the global ids, so that
 * an all-reduce(sum) yields the full weight vector on every rank.  Stream-ordered on the handle's stream, no host
 * synchronisation: run the collective on the same stream (rbpf_set_stream with the communicator's stream, e.g.
 * torch's current stream) or call rbpf_synchronize() first ...                                                 */
int  rbpf_export_weights(rbpf_handle* h, void* d_global_weights, int32_t n_global);
/* ... then compute the global systematic-resampling ancestors from it (main.py:46-67), identically on every
 * rank; idx_out[n_global] on the host. */
int  rbpf_resample_indices_global(rbpf_handle* h, const void* d_global_weights, int32_t n_global,
                                  double u, int32_t* idx_out, int32_t* did_resample);
/* The early variants, for overlapping the global resample with rbpf_scan_update_end.  Queue, between
 * rbpf_scan_update_begin and rbpf_scan_update_end and on `stream` (the handle's own stream, or another one: the calls
 * wait for the weighting kernel through an event):
 *   rbpf_export_weights_early            the vector has n_global + 1 elements; the last one is 1 on a rank with a
 *                                        particle on the NaN-covariance branch (robot.py:73-78)
 *   (the caller's all-reduce(sum) on that stream)
 *   rbpf_resample_indices_global_early   ancestors + read-back into pinned memory; returns at once
 * then rbpf_scan_update_end, and finally
 *   rbpf_resample_indices_global_wait    waits for the read-back only (an event), not for the map update queued
 *                                        behind it.  *nan_branch_ranks != 0: weights of NaN-branch particles change in
 *                                        rbpf_scan_update_end - discard the result and use the late calls above. */
int  rbpf_export_weights_early(rbpf_handle* h, void* d_global_weights_n_plus_1, int32_t n_global, void* stream);
int  rbpf_resample_indices_global_early(rbpf_handle* h, const void* d_global_weights_n_plus_1, int32_t n_global, double u,
                                        void* stream);
int  rbpf_resample_indices_global_wait(rbpf_handle* h, int32_t* idx_out, int32_t* did_resample, double* nan_branch_ranks);
/* serialise n local particles (state + the written boxes of their tiles + occupancy masks) into d_buf;
 * meta_out[n * rbpf_pack_meta_width()] describes the layout for the receiver (host ints) */
int32_t rbpf_pack_meta_width(rbpf_handle* h);
int64_t rbpf_packed_particle_bytes(rbpf_handle* h);     /* upper bound of one particle's payload */
int  rbpf_pack_particles(rbpf_handle* h, const int32_t* local_idx, int32_t n, void* d_buf, int64_t cap_bytes,
                         int32_t* meta_out, int64_t* bytes_out);
/* local part of a global resample: new local particle j continues local particle new_src[j] (sorted ascending)
 * or, for new_src[j] = -1 (last), arrives from another rank and is installed by rbpf_unpack_particles; weights
 * restart at 1.0 (main.py:77-78) */
int  rbpf_apply_resample_local(rbpf_handle* h, const int32_t* new_src, const int32_t* new_global_id);
int  rbpf_unpack_particles(rbpf_handle* h, const int32_t* local_idx, int32_t n, const void* d_buf,
                           const int32_t* meta_in);
/* The migration with ONE host wait (thesis_amd/sharding.py): (1) the tile boxes of the departing particles are gathered
 * into d_raw ([n][rbpf_pack_raw_width()] int32, device) without waiting; the ranks exchange these records while they
 * are on the device and read their own and the incoming ones back in one copy; (2) records -> the layout rows of
 * rbpf_unpack_particles and the payload size (host only, no device work); (3) the pack with the records already on
 * the host (nothing waited for).  Same payload format as rbpf_pack_particles. */
int32_t rbpf_pack_raw_width(rbpf_handle* h);
int  rbpf_gather_pack_meta(rbpf_handle* h, const int32_t* local_idx, int32_t n, void* d_raw);
int  rbpf_meta_from_raw(rbpf_handle* h, const int32_t* raw, int32_t n, int32_t* meta_out, int64_t* bytes_out);
int  rbpf_pack_particles_raw(rbpf_handle* h, const int32_t* local_idx, int32_t n, const int32_t* raw, void* d_buf,
                             int64_t cap_bytes, int64_t* bytes_out);

/* ---- state access (Robot.get_latest_pose/weight, HybridMap readback; main.py:152,170-176) ----- */
int  rbpf_get_poses(rbpf_handle* h, double* out_p3);
int  rbpf_get_covs(rbpf_handle* h, double* out_p9);
int  rbpf_get_weights(rbpf_handle* h, double* out_p);
int  rbpf_set_state(rbpf_handle* h, const double* poses_p3, const double* covs_p9,
                    const double* weights_p);             /* any pointer may be NULL           */
/* position of the two internal random streams (sample draws: one per scan update; resample uniforms), for checkpoints
   that continue a run bit-identically; replaces the `np.random` state the reference pickles (main.py:183-210) */
int  rbpf_get_rng_state(rbpf_handle* h, uint64_t* scan_updates, uint64_t* resample_draws);
int  rbpf_set_rng_state(rbpf_handle* h, uint64_t scan_updates, uint64_t resample_draws);
int  rbpf_get_tile_count(rbpf_handle* h, int32_t particle, int32_t* out_n);
/* k-th tile of a particle in lattice order: centre (metres) and dim*dim cells, cell[x*dim+y] */
int  rbpf_get_tile(rbpf_handle* h, int32_t particle, int32_t k, double* centre2, int8_t* cells);
/* cells must lie in [min_odds_emp, max_odds_occ] / quantum, as every map of the reference does (RBPF_EINVAL) */
int  rbpf_set_tile(rbpf_handle* h, int32_t particle, double cx, double cy, const int8_t* cells);
int  rbpf_get_dim(rbpf_handle* h, int32_t* out_dim);
/* HybridMap.get_odds_at (hybridmap.py:85-93) for n points against particle p's map;
 * out_none[i] = 1 where the reference returns None */
int  rbpf_get_odds_at(rbpf_handle* h, int32_t particle, const double* xy, int32_t n,
                      double* out_vals, uint8_t* out_none);

#ifdef __cplusplus
}
#endif
#endif /* RBPF_HIP_H */
