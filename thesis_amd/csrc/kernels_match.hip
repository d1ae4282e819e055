// kernels_match.hip -- a6 + a7: the scan-match stage of Robot.map_update (robot.py:62-69), i.e.
// HybridMap.get_scan_match / get_scan_adj (hybridmap.py:147-261) + matchScanCustom.m.
//
// PARITY UNPINNED.  The reference hands its point lists to MATLAB Navigation Toolbox R2021a
// (matchScansGrid, then matchScans/NDT); that code is closed source, absent from the reference tree,
// and no recorded outputs exist.  What is kept is the interface and the decision logic:
//   inputs   the scan, the particle's own map (adj = 0) or the previous accepted scan (adj = 1),
//            the odometry pose as initial guess, the search window of robot.py:62-65 with the
//            rotation range hard-wired to pi/6 (hybridmap.py:249);
//   outputs  pose = guess + offset (hybridmap.py:253-255), a 3x3 covariance, a score; a failed match
//            is reported as NaN covariance / score 0 (matchScanCustom.m:25-28) and sends the particle
//            down robot.py:73-78.
// The numerics are this library's own two-level correlative matcher (Olson 2009 style):
//   field    occupancy bits of the region around the guess (cells with log-odds > threshold,
//            gridmap.py:153) plus their 3x3 dilation, staged in LDS;  hit = 2 on an occupied cell, 1 on a
//            dilated one;
//   coarse   4x4 max-pooled field, rotation step 4*d0, translation step 4 cells, every 8th beam;
//   fine     around the coarse optimum: rotation step d0 = cell/max_range, translation step 1 cell,
//            every 4th beam;
//   cov      second moments of exp((s - s_best)/tau) over the fine candidates + a floor.
//   ndt      (rbpf_config.ndt_refine) the second stage of matchScanCustom.m:32-50: Normal Distributions Transform
//            (Biber & Strasser 2003, the algorithm MATLAB's matchScans documents) with the call site's CellSize 0.1 m
//            and MaxIterations 500, started from the correlative optimum and accepted by the reference's rule
//            (valid pose and 2 * ndtScore > gridScore; the covariance stays the grid one).  Its own kernel
//            (ndt_kernel) on the occupancy field match_kernel staged.  Restated on the CPU in
//            oracle/matcher_oracle.py; MATLAB's own numerics stay unpinned.
#include <limits.h>
#include <string.h>

#include <hip/hip_ext.h>

#include "rbpf_internal.h"
#include "rbpf_device.h"

namespace rbpf {

#ifdef RBPF_STAMPS
#define MSTAMP(k) do { if (tid == 0) { long long t_ = clock64(); st_acc[k] += t_ - st_prev; st_prev = t_; } } while (0)
#else
#define MSTAMP(k) do { } while (0)
#endif

static const int MBLOCK = 512;          // threads per particle (8 waves)
static const int M_COARSE = 4;          // coarse cell = 4 fine cells
static const int M_FINE_T = 4;          // fine translations: -4..4 cells around the coarse optimum
static const int M_FINE_R = 4;          // fine rotations:   -4..4 steps of d0

struct MatchArgs {
    int mode;                   // 0: field from the particle's map; 1: field from ref points (last scan)
    int single;                 // 1: stateless twin (one problem: explicit guess/range, curr points as beams)
    const double* ref_xy; int n_ref;       // mode 1 points (global frame)
    double guess[3]; double range[3];      // single = 1
    double* out;                // [P][13] pose, cov, score
    int N;                      // region edge in matcher cells (multiple of 32)
    int ds;                     // matcher cell = ds map cells
    double mcs;                 // matcher cell size in metres
    double d0;                  // fine rotation step
    double rot_range;           // pi/6 (hybridmap.py:249)
    double max_range;           // beams/points farther than this are ignored (hybridmap.py:20)
    const float* sel_x; const float* sel_y; int n_sel;   // selected beams / curr points, sensor frame, metres
    int n_coarse_rot;           // rotations on each side at the coarse level
    int cap_sel;                // LDS capacity for selected beams
    int sc_cap;                 // entries of the score table in LDS (the coarse level runs in groups of rotations that fit)
    double cell_off;            // 0.5 in the stateless twin: its points are snapped to cell corners (hybridmap.py:226-227)
    int ndt;                    // 0 off; 1 matchScanCustom.m:38-44 acceptance; 2 take every valid NDT pose (diagnostic)
    int ndt_nc;                 // NDT cell edge in matcher cells (0.1 m, matchScanCustom.m:37); < 2: no cell can hold 3 points
    int ndt_max_iter;           // matchScanCustom.m:36
    const int32_t* dup_of;      // particles whose entry is not their own index are exact duplicates of that particle: skipped
    uint32_t* ndt_occ;          // [particles][N][N/32] the staged occupancy field, handed to the NDT kernel
    double* ndt_aux;            // [particles][5] grid optimum (cells, cells, rad), its full score, ok flag
};

// two consecutive words of an occupancy mask row (4-byte aligned: one global_load_dwordx2)
struct __attribute__((packed, aligned(4))) MaskPair { uint32_t lo, hi; };
struct __attribute__((packed, aligned(4))) MaskTriple { uint32_t w0, w1, w2; };

struct MatchLds {
    uint32_t* occ;      // [N][N/32]
    uint32_t* dil;      // [N][N/32]
    uint32_t* crs;      // [N/4][words]
    float* fx4; float* fy4; // [ceil(nb/4) padded to 4] every 4th beam (fine level), contiguous
    float* cx8; float* cy8; // [ceil(nb/8) padded to 4] every 8th beam (coarse level), contiguous
    int* sc;            // [n coarse candidates] then reused for fine
};

// the coarse map is stored with overlapping words: word h of a row holds coarse columns [16h, 16h + 32), so that any
// 8 neighbouring columns lie inside ONE word
__host__ __device__ inline int match_crs_words(int N) { return ((N / M_COARSE) + 15) / 16; }
// ... and stored column-major with CRS_PAD zero rows on either side: crs[h * match_crs_stride(N) + CRS_PAD + cu], so that
// the rows of consecutive x translations are consecutive words and a row outside the region reads 0 without a test
static const int CRS_PAD = 7;
__host__ __device__ inline int match_crs_stride(int N) { return N / M_COARSE + 2 * CRS_PAD; }
__host__ __device__ inline int match_crs_total(int N) { return (match_crs_words(N) * match_crs_stride(N) + 3) & ~3; }

// LDS of a match problem without its score table: occupancy + dilation bitmasks, the coarse map, the decimated beams
static size_t match_lds_base(int N, int B) {
    size_t words = (size_t)N * (N / 32);
    size_t dec = (size_t)(((B + 3) / 4 + 7) & ~3) + (size_t)(((B + 7) / 8 + 7) & ~3);
    return 2 * words * 4 + (size_t)match_crs_total(N) * 4 + dec * 8 + 256;
}
// The score table holds the coarse candidates of as many rotations as fit when TWO workgroups share a CU's 160 KB (the
// kernel's instruction stream keeps one workgroup's eight waves busy half of the time), at least one rotation's
// (`per_rot`) and the fine level's; the coarse level runs in groups of rotations.
__global__ __launch_bounds__(MBLOCK) void match_kernel(DevView v, MatchArgs a);
size_t match_lds_bytes(int N, int B, int n_coarse, int per_rot) {
    const size_t fine = (size_t)(2 * M_FINE_R + 1) * (2 * M_FINE_T + 1) * (2 * M_FINE_T + 1);
    // half a CU's 160 KB less the kernel's static LDS (asked from the runtime; the LDS is handed out in 512-byte units)
    size_t fixed = 2048;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(match_kernel)) == hipSuccess) fixed = (fa.sharedSizeBytes + 511) & ~(size_t)511;
    const size_t base = match_lds_base(N, B), two_wg = (160 * 1024) / 2 - fixed;
    size_t nsc = (size_t)n_coarse;
    if (base + nsc * 4 > two_wg) nsc = base < two_wg ? (two_wg - base) / 4 : 0;
    if (nsc < (size_t)per_rot) nsc = (size_t)per_rot;
    if (nsc < fine) nsc = fine;
    if (nsc > (size_t)n_coarse && (size_t)n_coarse >= fine) nsc = (size_t)n_coarse;
    return base + nsc * 4;
}
int match_sc_capacity(int N, int B, size_t lds) { return (int)((lds - match_lds_base(N, B)) / 4); }

// lane i takes lane i + N of its row of 16 lanes (0 where there is none): the last four steps of a wave sum that ends in lane 0
// with the additions paired as __shfl_down pairs them, through the data-parallel-primitive path
template <int N>
__device__ __forceinline__ double dpp_row_shl_f64(double x) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x100 + N, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x100 + N, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ double wave_sum_lane0_f64(double x) {
    x += __shfl_down(x, 32, 64); x += __shfl_down(x, 16, 64);
    x += dpp_row_shl_f64<8>(x); x += dpp_row_shl_f64<4>(x); x += dpp_row_shl_f64<2>(x); x += dpp_row_shl_f64<1>(x);
    return x;
}

__device__ __forceinline__ int field_hit(const MatchLds& s, int N, int u, int w) {
    if ((unsigned)u >= (unsigned)N || (unsigned)w >= (unsigned)N) return 0;
    int idx = u * (N >> 5) + (w >> 5);
    uint32_t m = 1u << (w & 31);
    return ((s.occ[idx] & m) ? 1 : 0) + ((s.dil[idx] & m) ? 1 : 0);
}
// guess, search window (robot.py:62-65) and region origin of one match problem
// pose5 = the particle's x, y, theta, cov[0][0], cov[1][1] (not looked at by the stateless twin)
__device__ inline void match_frame_from(const MatchArgs& a, const double* pose5, double* g, double* rng, int* org) {
    double gx, gy, gth, rx, ry;
    if (a.single) {
        gx = a.guess[0]; gy = a.guess[1]; gth = a.guess[2]; rx = a.range[0]; ry = a.range[1];
    } else {
        gx = pose5[0]; gy = pose5[1]; gth = pose5[2];
        double c00 = pose5[3], c11 = pose5[4];
        double p0 = sqrt(c00) * 30.0, p1 = sqrt(c11) * 30.0;                 // robot.py:62
        ry = fmax(fmin(4 * p1, 0.7), 0.1);                                   // robot.py:64
        rx = fmax(fmin(4 * p0, 0.7), 0.1);                                   // robot.py:65
    }
    g[0] = gx; g[1] = gy; g[2] = gth; rng[0] = rx; rng[1] = ry;
    org[0] = (int)floor(gx / a.mcs) - a.N / 2;
    org[1] = (int)floor(gy / a.mcs) - a.N / 2;
}

__device__ inline void match_frame(const DevView& v, const MatchArgs& a, int p, double* g, double* rng, int* org) {
    double pose5[5] = {0, 0, 0, 0, 0};
    if (!a.single) { pose5[0] = v.px[p]; pose5[1] = v.py[p]; pose5[2] = v.pth[p]; pose5[3] = v.cov[p]; pose5[4] = v.cov[(size_t)4 * v.P + p]; }
    match_frame_from(a, pose5, g, rng, org);
}

__global__ __launch_bounds__(MBLOCK) void match_kernel(DevView v, MatchArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N, W = N >> 5, tid = threadIdx.x, p = blockIdx.x;
    // An exact duplicate (a copy made by the last resample: same pose, covariance and map) would repeat its
    // representative's search bit for bit; the proposal kernel reads the representative's row instead.
    // (the loads the set-up needs are issued together with the duplicate test's: one round trip to memory instead of three)
    const int dup = a.dup_of ? a.dup_of[p] : p;
    double pre[5] = {0, 0, 0, 0, 0}; int pre_slot = 0;
    if (!a.single && tid == 0) { pre[0] = v.px[p]; pre[1] = v.py[p]; pre[2] = v.pth[p]; pre[3] = v.cov[p]; pre[4] = v.cov[(size_t)4 * v.P + p]; }
    if (!a.single) pre_slot = v.slot[p];
    if (dup != p) { if (tid == 0) atomicAdd(&v.stats[ST_MATCH_SHARED], 1ull); return; }
    MatchLds s;
    s.occ = reinterpret_cast<uint32_t*>(smem);
    s.dil = s.occ + (size_t)N * W;
    s.crs = s.dil + (size_t)N * W;
    const int cap4 = ((a.cap_sel + 3) / 4 + 7) & ~3, cap8 = ((a.cap_sel + 7) / 8 + 7) & ~3;
    s.fx4 = reinterpret_cast<float*>(s.crs + match_crs_total(N)); s.fy4 = s.fx4 + cap4;
    s.cx8 = s.fy4 + cap4; s.cy8 = s.cx8 + cap8;
    s.sc = reinterpret_cast<int*>(s.cy8 + cap8);
    __shared__ double s_g[3], s_rng[2];
    __shared__ int s_org[2], s_nb, s_best, s_bestc;
    __shared__ unsigned s_def[32], s_slowg;  // per region word: defect columns of the index map; words not inside one mapped tile
    __shared__ double s_mom[10], s_wmom[MBLOCK / 64][10];
    __shared__ int s_tab[49];

#ifdef RBPF_STAMPS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = clock64();
#endif
    // ---- guess, search window (robot.py:62-65), region origin ----------------------------------------------
    if (tid == 0) {
        match_frame_from(a, pre, s_g, s_rng, s_org);
        s_nb = 0; s_best = INT_MIN; s_bestc = 0; s_slowg = 0;
        for (int i = 0; i < 10; ++i) s_mom[i] = 0.0;
    }
    if (!a.single) {
        const int32_t* tab = v.tile_tab + (size_t)pre_slot * v.L * v.L;
        for (int i = tid; i < v.L * v.L; i += MBLOCK) s_tab[i] = tab[i];
    }
    // the occupancy words are OR-ed into when rasterised from points (mode 1); every other word of the three fields is
    // written before it is read
    if (a.mode != 0) for (int i = tid; i < N * W; i += MBLOCK) s.occ[i] = 0;
    if (tid < 32) s_def[tid] = 0;
    __syncthreads();
    const int ox = s_org[0], oy = s_org[1];

    MSTAMP(0);
    // ---- field ------------------------------------------------------------------------------------------------
    if (a.mode == 0) {
        // occupancy of the particle's own map over the region: bits of the per-tile occupancy masks kept by the
        // map-update kernel (cell > threshold, gridmap.py:153), re-addressed through the global-index LUT
        int16_t* colmap = reinterpret_cast<int16_t*>(s.sc);     // [N*ds] (lat << 12 | cidx) of every region column
        uint32_t* rowbase = s.dil;                              // [N*ds] word offset of every region row in the mask pool
        for (int i = tid; i < N * a.ds; i += MBLOCK) {
            const int gyi = oy * a.ds + i, gxi = ox * a.ds + i;
            int16_t e = -1;
            if (lut_valid_g(v, gyi)) {
                const uint32_t ey = lut_at(v, gyi);
                e = (int16_t)((lut_lat(ey) << 12) | lut_cidx(ey));
                // the reference's float index formula is the plain cell arithmetic except at isolated
                // columns: those are this word's defect bits
                const int dev_ = lut_cidx(ey) - (gyi - (lut_lat(ey) - v.R) * v.dim + v.dim / 2);
                if (dev_ == -1) atomicOr(&s_def[(i >> 5) & 31], 1u << (i & 31));        // the only deviation the formula produces
                else if (dev_ != 0) atomicOr(&s_slowg, 1u << ((i >> 5) & 31));
                const int g0 = gyi - (i & 31);                  // the word's first column: must be the same tile
                if ((i & 31) && !(lut_valid_g(v, g0) && lut_lat(lut_at(v, g0)) == lut_lat(ey))) atomicOr(&s_slowg, 1u << ((i >> 5) & 31));
            } else atomicOr(&s_slowg, 1u << ((i >> 5) & 31));
            colmap[i] = e;
            uint32_t rb = 0xFFFFFFFFu;                          // high byte = lattice row of the tile table, low 24 bits = storage row
            if (lut_valid_g(v, gxi)) { const uint32_t ex = lut_at(v, gxi); rb = ((uint32_t)lut_lat(ex) << 24) | (uint32_t)lut_cidx(ex); }
            rowbase[i] = rb;
        }
        __syncthreads();
        MSTAMP(6);
        // One region word (32 columns of one row) per lane.  A wave takes 16 rows x 4 adjacent words, so that the words
        // with many defect columns meet in few waves and the loads of a row stay contiguous.
        if (a.ds == 2 && 2 * W <= 32 && !v.match_stage_slow) {
            // Matcher cell = 2 x 2 map cells (0.025 m maps): a region word is 64 map columns of two map rows.  Same
            // scheme as below on 64-bit windows, then neighbouring bit pairs are OR-ed and the even bits compressed.
            for (int q = tid; q < N * W; q += MBLOCK) {
                const int u = q / W, wv = q % W;
                const int e0 = colmap[wv * 64], e1 = colmap[wv * 64 + 32];
                const bool fast2 = !((s_slowg >> (2 * wv)) & 3u) && e0 >= 0 && e1 >= 0 && (e0 >> 12) == (e1 >> 12);
                uint32_t bits = 0;
                if (fast2) {
                    const int lat = e0 >> 12;
                    const int c0 = (oy * 2 + wv * 64) - (lat - v.R) * v.dim + v.dim / 2;      // storage column of the first map column
                    const int cy = max(c0 - 1, 0), wi = cy >> 5, sft = c0 - (wi << 5);         // 0..32
                    const unsigned long long d64 = ((unsigned long long)s_def[2 * wv + 1] << 32) | s_def[2 * wv];
                    unsigned long long acc = 0;
                    for (int du = 0; du < 2; ++du) {
                        const uint32_t rb = rowbase[u * 2 + du];
                        if (rb == 0xFFFFFFFFu) continue;
                        const int t = s_tab[(rb >> 24) * v.L + lat];
                        if (t < 0) continue;
                        const uint32_t* row = v.occ + ((size_t)t * v.dim + (rb & 0xFFFFFFu)) * v.ow;
                        const MaskTriple m3 = *reinterpret_cast<const MaskTriple*>(row + wi);   // 64 columns from bit sft <= 32: three words
                        const unsigned long long A = ((unsigned long long)m3.w1 << 32) | m3.w0, B = (unsigned long long)m3.w2;
                        unsigned long long b64 = sft == 0 ? A : (A >> sft) | (B << (64 - sft));
                        if (d64) {                                                              // columns stored one cell lower
                            const int s1 = sft - 1;
                            const unsigned long long below = s1 < 0 ? A << 1 : s1 == 0 ? A : (A >> s1) | (B << (64 - s1));
                            b64 = (b64 & ~d64) | (below & d64);
                        }
                        acc |= b64;
                    }
                    unsigned long long x = (acc | (acc >> 1)) & 0x5555555555555555ull;          // bit 2j = map columns 2j, 2j + 1
                    x = (x | (x >> 1)) & 0x3333333333333333ull;
                    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
                    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
                    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
                    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
                    bits = (uint32_t)x;
                } else {
                    for (int du = 0; du < 2; ++du) {
                        const uint32_t rb = rowbase[u * 2 + du];
                        if (rb == 0xFFFFFFFFu) continue;
                        for (int b = 0; b < 64; ++b) {
                            const int e = colmap[wv * 64 + b];
                            if (e < 0) continue;
                            const int t = s_tab[(rb >> 24) * v.L + (e >> 12)];
                            if (t < 0) continue;
                            const int cyb = e & 0xFFF;
                            const uint32_t wd = v.occ[((size_t)t * v.dim + (rb & 0xFFFFFFu)) * v.ow + (cyb >> 5)];
                            if ((wd >> (cyb & 31)) & 1u) bits |= 1u << (b >> 1);
                        }
                    }
                }
                s.occ[u * W + wv] = bits;
            }
        } else {
        // A wave takes 16 rows x 4 adjacent words at a time; a lane keeps its column word over a run of row blocks, so the
        // column part of the address (tile column, window shift, defect bits) is worked out once per run.  The 32 columns
        // of a fast word lie in two consecutive mask words: one 8-byte load.
        const int SW = 4;                                       // rows per lane in flight
        const int GT = (W + 3) / 4, NRB = N / 16;
        const unsigned slowg = (a.ds != 1 || W > 32 || v.match_stage_slow) ? 0xFFFFFFFFu : s_slowg;
        const int wave = tid >> 6, ln = tid & 63;
        for (int g = 0; g < GT; ++g) {
            const int wv = g * 4 + (ln >> 4);
            const bool wv_ok = wv < W, fastw = wv_ok && !((slowg >> wv) & 1u);
            int lat = 0, wi = 0, sft = 0; uint32_t d = 0;
            if (fastw) {
                lat = colmap[wv * 32] >> 12;
                const int c0 = (oy + wv * 32) - (lat - v.R) * v.dim + v.dim / 2;       // storage column of the word's first column
                wi = max(c0 - 1, 0) >> 5;                                               // the window starts one column lower (defects)
                sft = c0 - (wi << 5);                                                   // 0..32
                d = s_def[wv];                                                          // columns stored one cell lower
            }
            for (int rb0 = wave; rb0 < NRB; rb0 += SW * (MBLOCK / 64)) {
                MaskPair mp[SW];
#pragma unroll
                for (int k = 0; k < SW; ++k) {                  // issue the mask loads of SW rows first
                    const int rbk = rb0 + k * (MBLOCK / 64);
                    mp[k].lo = 0; mp[k].hi = 0;
                    if (rbk >= NRB || !fastw) continue;
                    const uint32_t rb = rowbase[rbk * 16 + (ln & 15)];
                    if (rb == 0xFFFFFFFFu) continue;
                    const int t = s_tab[(rb >> 24) * v.L + lat];
                    if (t < 0) continue;
                    mp[k] = *reinterpret_cast<const MaskPair*>(v.occ + ((size_t)t * v.dim + (rb & 0xFFFFFFu)) * v.ow + wi);
                }
#pragma unroll
                for (int k = 0; k < SW; ++k) {
                    const int rbk = rb0 + k * (MBLOCK / 64), u = rbk * 16 + (ln & 15);
                    if (rbk >= NRB || !wv_ok) continue;
                    uint32_t bits;
                    if (fastw) {
                        // 32 consecutive storage columns: a funnel shift of the loaded window, then the defect columns
                        const unsigned long long w01 = ((unsigned long long)mp[k].hi << 32) | mp[k].lo;
                        bits = sft < 32 ? (uint32_t)(w01 >> sft) : mp[k].hi;
                        if (d) {
                            const int s1 = sft - 1;
                            const uint32_t below = s1 < 0 ? bits << 1 : (uint32_t)(w01 >> s1);
                            bits = (bits & ~d) | (below & d);
                        }
                    } else {                                      // tile edge, unmapped column or coarser matcher cell: bit by bit
                        bits = 0;
                        for (int du = 0; du < a.ds; ++du) {
                            const uint32_t rb = rowbase[u * a.ds + du];
                            if (rb == 0xFFFFFFFFu) continue;
                            for (int b = 0; b < 32 * a.ds; ++b) {
                                const int e = colmap[wv * 32 * a.ds + b];
                                if (e < 0) continue;
                                const int t = s_tab[(rb >> 24) * v.L + (e >> 12)];
                                if (t < 0) continue;
                                const int cy = e & 0xFFF;
                                const uint32_t wd = v.occ[((size_t)t * v.dim + (rb & 0xFFFFFFu)) * v.ow + (cy >> 5)];
                                if ((wd >> (cy & 31)) & 1u) bits |= 1u << (b / a.ds);
                            }
                        }
                    }
                    s.occ[u * W + wv] = bits;
                }
            }
        }
        }
        __syncthreads();                                        // colmap memory is the score table again below
    } else {
        // occupancy rasterised from reference points (previous accepted scan, hybridmap.py:167-171)
        const int RP = 4;                                       // points per thread in flight
        for (int i0 = tid; i0 < a.n_ref; i0 += RP * MBLOCK) {
          double2 pt[RP];
#pragma unroll
          for (int k = 0; k < RP; ++k) { const int i = i0 + k * MBLOCK; pt[k] = i < a.n_ref ? reinterpret_cast<const double2*>(a.ref_xy)[i] : make_double2(0.0, 0.0); }
#pragma unroll
          for (int k = 0; k < RP; ++k) {
            if (i0 + k * MBLOCK >= a.n_ref) continue;
            double rx = pt[k].x, ry = pt[k].y;
            double dx = rx - s_g[0], dy = ry - s_g[1];
            if (!(sqrt(dx * dx + dy * dy) < a.max_range)) continue;   // hybridmap.py:171 (11 m); matchScanCustom.m:11 (15 m)
            int u = (int)floor(rx / a.mcs + a.cell_off) - ox, w = (int)floor(ry / a.mcs + a.cell_off) - oy;
            if ((unsigned)u < (unsigned)N && (unsigned)w < (unsigned)N) atomicOr(&s.occ[u * W + (w >> 5)], 1u << (w & 31));
          }
        }
    }
    // selected beams (in matcher-cell units, sensor frame)
    {
        const float inv = (float)(1.0 / a.mcs);
        const float FAR = -1.0e6f;                          // padding beams fall outside the region: no hit
        for (int i = tid; i < cap4; i += MBLOCK) { const int b = 4 * i; const bool ok = b < a.n_sel; s.fx4[i] = ok ? a.sel_x[b] * inv : FAR; s.fy4[i] = ok ? a.sel_y[b] * inv : FAR; }
        for (int i = tid; i < cap8; i += MBLOCK) { const int b = 8 * i; const bool ok = b < a.n_sel; s.cx8[i] = ok ? a.sel_x[b] * inv : FAR; s.cy8[i] = ok ? a.sel_y[b] * inv : FAR; }
        if (tid == 0) s_nb = a.n_sel;
    }
    __syncthreads();
    if (a.ndt_occ) {                                          // the NDT kernel reads the same field (32 KB per particle)
        uint32_t* dst = a.ndt_occ + (size_t)p * N * W;
        for (int i = tid; i < N * W; i += MBLOCK) dst[i] = s.occ[i];
    }
    MSTAMP(1);
    // 3x3 dilation and 4x4 max-pool of the dilated field.  A thread takes one column word and a run of rows and slides
    // a three-row window of horizontally dilated words down it: three LDS reads per row instead of nine.
    {
        const int nchunk = max(1, MBLOCK / W), rows_per = (N + nchunk - 1) / nchunk;
        const int wv = tid % W, chunk = tid / W;
        // (sixteen words per row: the lanes of a DPP row hold one field row, and a word's neighbours come from the lanes beside
        // it - 0 at the row's ends - instead of two more LDS reads)
        const bool row16 = W == 16;
        auto hdil = [&](int u) -> uint32_t {
            const bool inr = u >= 0 && u < N;
            const uint32_t c = inr ? s.occ[u * W + wv] : 0u;
            uint32_t lb, rb;
            if (row16) {
                lb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(c >> 31), 0x111, 0xF, 0xF, true);     // row_shr:1: bit 31 of the word before
                rb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(c << 31), 0x101, 0xF, 0xF, true);     // row_shl:1: bit 0 of the word after
            } else {
                lb = (inr && wv > 0) ? s.occ[u * W + wv - 1] >> 31 : 0u;
                rb = (inr && wv + 1 < W) ? s.occ[u * W + wv + 1] << 31 : 0u;
            }
            return c | (c << 1) | (c >> 1) | lb | rb;
        };
        if (chunk < nchunk) {
            const int u0 = chunk * rows_per, u1 = min(N, u0 + rows_per);
            uint32_t up = hdil(u0 - 1), mid = hdil(u0);
            for (int u = u0; u < u1; ++u) {
                const uint32_t dn = hdil(u + 1);
                s.dil[u * W + wv] = up | mid | dn;
                up = mid; mid = dn;
            }
        }
    }
    __syncthreads();
    // one coarse word per thread: two 16-column halves, each the OR of four field rows of two field words whose 4-bit
    // groups collapse into one bit
    for (int q = tid; q < (N / M_COARSE) * match_crs_words(N); q += MBLOCK) {
        const int cu = q / match_crs_words(N), h = q % match_crs_words(N);
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int wv = 2 * h + k;                                   // field words 2h .. 2h + 3 = coarse columns 16h .. 16h + 31
            uint32_t d = 0;
            if (wv < W) {
#pragma unroll
                for (int r = 0; r < M_COARSE; ++r) d |= s.dil[(cu * M_COARSE + r) * W + wv];
            }
            d |= d >> 1; d |= d >> 2; d &= 0x11111111u;                  // bit 4j = any of bits 4j .. 4j+3
            d = (d | (d >> 3)) & 0x03030303u;                            // two bits per byte
            d = (d | (d >> 6)) & 0x000F000Fu;                            // four bits per half
            d = (d | (d >> 12)) & 0xFFu;                                 // eight bits
            out |= d << (8 * k);
        }
        s.crs[h * match_crs_stride(N) + CRS_PAD + cu] = out;
    }
    for (int q = tid; q < match_crs_words(N) * 2 * CRS_PAD; q += MBLOCK) {        // the zero rows before and after every column
        const int h = q / (2 * CRS_PAD), r = q % (2 * CRS_PAD);
        s.crs[h * match_crs_stride(N) + (r < CRS_PAD ? r : N / M_COARSE + r)] = 0;
    }
    __syncthreads();

    const int nb = s_nb;
    const float fx = (float)(s_g[0] / a.mcs - (double)ox + a.cell_off), fy = (float)(s_g[1] / a.mcs - (double)oy + a.cell_off);
    const double gth = s_g[2];
    // translation half-widths in matcher cells; candidates must satisfy |d| < range (matchScanCustom.m:53)
    const double rxc = s_rng[0] / a.mcs, ryc = s_rng[1] / a.mcs;
    const int ktx = (int)ceil(rxc / M_COARSE) - 1;    // largest k with k*M_COARSE < range
    const int kty = (int)ceil(ryc / M_COARSE) - 1;
    const int ntx = 2 * max(ktx, 0) + 1, nty = 2 * max(kty, 0) + 1;
    const int nr = 2 * a.n_coarse_rot + 1;
    // The score table holds 16-bit sums, two per word, laid out as the byte lanes of the accumulators fall: a pass of 8 y
    // translations is four words (y 0|2, 1|3, 4|6, 5|7), so an item adds four candidates' sums with two atomics and no
    // extraction.  It holds the candidates of RG rotations; the coarse level runs in equal groups of rotations when the
    // table of all of them would keep a second workgroup off the CU.
    const int NP = (nty + 7) >> 3;                                       // passes of 8 y translations
    const int rot_words = ntx * NP * 4;
    const int RGmax = max(1, min(nr, a.sc_cap / rot_words));
    const int n_groups = (nr + RGmax - 1) / RGmax;
    const int RG = (nr + n_groups - 1) / n_groups;

    MSTAMP(2);
    // Neighbouring translation candidates along y are consecutive bits of one mask row, so one LDS read scores a whole
    // row of them: a work item is (rotation, beam slice); its per-candidate sums are byte lanes of two registers.
    const int n4 = ((nb + 3) / 4 + 3) & ~3;                                      // padded with far-away beams
    const float gthf = (float)remainder(gth, 6.283185307179586);
    const int NC4 = N / M_COARSE;
    uint32_t* const sc2 = reinterpret_cast<uint32_t*>(s.sc);
    // ---- coarse level -------------------------------------------------------------------------------------------
    // work item = (rotation, beam slice): a beam is rotated once and looked up for every x translation (a shift by whole
    // coarse cells) and, through the byte lanes, for 8 y translations per LDS read; slices add their sums with atomics
    int g_best = INT_MIN, g_key = INT_MAX;                               // best score so far and its tie-break key (uniform)
    for (int r0 = 0; r0 < nr; r0 += RG) {
    const int nrg = min(RG, nr - r0), n_words = nrg * rot_words;
    for (int i = tid; i < n_words; i += MBLOCK) sc2[i] = 0;             // candidate sums are accumulated with atomics
    if (tid == 0) { s_best = INT_MIN; s_bestc = INT_MAX; }
    __syncthreads();
    {
        const int MAXTX = 7;
        const int NSC = max(1, MBLOCK / nrg), nb8 = (nb + 7) / 8, per = (nb8 + NSC - 1) / NSC;   // beams per slice
        const int RS = match_crs_stride(N);
        static_assert(MAXTX <= CRS_PAD, "the zero rows must cover a pass of x translations");
        for (int item = tid; item < nrg * NSC; item += MBLOCK) {
            const int irl = item / NSC, ir = r0 + irl, sl = item % NSC;
            float sn, cs;
            __sincosf(gthf + (float)((double)(ir - a.n_coarse_rot) * M_COARSE * a.d0), &sn, &cs);
            const int g_lo = sl * per, g_hi = min(nb8, g_lo + per);
            for (int pass = 0; pass < NP; ++pass)                       // up to 8 y translations per pass
            for (int t0 = 0; t0 < ntx; t0 += MAXTX)                     // up to MAXTX x translations per pass
            for (int gg = g_lo; gg < g_hi; gg += 240) {                 // byte-lane sums stay below 256
                const float ty0 = fy + (float)((pass * 8 - max(kty, 0)) * M_COARSE);
                uint32_t accA[MAXTX], accB[MAXTX];
#pragma unroll
                for (int t = 0; t < MAXTX; ++t) { accA[t] = 0; accB[t] = 0; }
                const int ge = min(g_hi, gg + 240);
#pragma unroll 2
                for (int g = gg; g < ge; ++g) {
                    const float bxs = s.cx8[g], bys = s.cy8[g];
                    const int cu0 = ((int)floorf(cs * bxs - sn * bys + fx) + (t0 - max(ktx, 0)) * M_COARSE) >> 2;   // x translation t looks at coarse row cu0 + t
                    const int cw0 = (int)floorf(sn * bxs + cs * bys + ty0) >> 2;             // candidate j looks at coarse column cw0 + j
                    // the 7 rows are 7 consecutive words of one column of overlapping words (which holds all 8 candidates);
                    // rows outside the region are zero rows, a beam outside altogether reads the zero rows of column 0
                    const bool ok = cw0 >= 0 && cw0 + 7 < NC4 && (unsigned)(cu0 + CRS_PAD) < (unsigned)(NC4 + CRS_PAD);
                    const uint32_t sh = (uint32_t)(cw0 & 15);
                    int ci = (cw0 >> 4) * RS + (CRS_PAD + cu0);
                    ci = ok ? ci : 0;
                    const uint32_t* col = s.crs + ci;
#pragma unroll
                    for (int t = 0; t < MAXTX; ++t) {
                        const uint32_t word = col[t];
                        accA[t] += (__builtin_amdgcn_ubfe(word, sh, 4u) * 0x00204081u) & 0x01010101u;
                        accB[t] += (__builtin_amdgcn_ubfe(word, sh + 4u, 4u) * 0x00204081u) & 0x01010101u;
                    }
                }
#pragma unroll
                for (int t = 0; t < MAXTX; ++t) {
                    if (t0 + t >= ntx) continue;
                    uint32_t* dst = sc2 + ((irl * ntx + t0 + t) * NP + pass) * 4;
                    const uint32_t A = accA[t], B = accB[t];
                    if (A) { atomicAdd(dst + 0, A & 0x00FF00FFu); atomicAdd(dst + 1, (A >> 8) & 0x00FF00FFu); }
                    if (B) { atomicAdd(dst + 2, B & 0x00FF00FFu); atomicAdd(dst + 3, (B >> 8) & 0x00FF00FFu); }
                }
            }
        }
    }
    __syncthreads();
    // word k of a pass holds y translations j0 = 4 (k >> 1) + (k & 1) (low half) and j0 + 2 (high half)
    {
        int mx = INT_MIN;
        for (int w = tid; w < n_words; w += MBLOCK) {
            const int k = w & 3, iy0 = ((w >> 2) % NP) * 8 + 4 * (k >> 1) + (k & 1);
            const uint32_t val = sc2[w];
            if (iy0 < nty) mx = max(mx, (int)(val & 0xFFFFu));
            if (iy0 + 2 < nty) mx = max(mx, (int)(val >> 16));
        }
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
        if ((tid & 63) == 0) atomicMax(&s_best, mx);
    }
    __syncthreads();
    const int best_group = s_best;
    for (int w = tid; w < n_words; w += MBLOCK) {
        const int k = w & 3, iy0 = ((w >> 2) % NP) * 8 + 4 * (k >> 1) + (k & 1);
        const uint32_t val = sc2[w];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int iy = iy0 + 2 * hf, sco = hf ? (int)(val >> 16) : (int)(val & 0xFFFFu);
            if (iy < nty && sco == best_group) {
                // ties: the candidate closest to the guess, then the lowest index (deterministic)
                const int rt = (w >> 2) / NP;                                    // (rotation in the group) * ntx + x translation
                const int ir = r0 + rt / ntx, itx = rt % ntx;
                const int cnd = (ir * ntx + itx) * nty + iy;                     // index over all rotations
                int dr = ir - a.n_coarse_rot, dx = itx - max(ktx, 0), dy = iy - max(kty, 0);
                int key = ((dr * dr + dx * dx + dy * dy) << 16) | cnd;
                atomicMin(&s_bestc, key);
            }
        }
    }
    __syncthreads();
    if (best_group > g_best || (best_group == g_best && s_bestc < g_key)) { g_best = best_group; g_key = s_bestc; }
    __syncthreads();                                                     // (the next group resets the table and the two words)
    }
    if (tid == 0) s_bestc = g_key;
    __syncthreads();
    const int cbest = s_bestc & 0xFFFF;
    const int cir = cbest / (ntx * nty) - a.n_coarse_rot, cit = cbest % (ntx * nty);
    const int ctx = (cit / nty - max(ktx, 0)) * M_COARSE, cty = (cit % nty - max(kty, 0)) * M_COARSE;
    __syncthreads();
    const int FR = 2 * M_FINE_R + 1, FT = 2 * M_FINE_T + 1;
    const int n_fine = FR * FT * FT;
    for (int i = tid; i < n_fine; i += MBLOCK) s.sc[i] = 0;
    if (tid == 0) s_best = INT_MIN;
    __syncthreads();

    MSTAMP(3);
    // ---- fine level -----------------------------------------------------------------------------------------------
    // work item = (rotation, x translation, beam slice); the 9 y translations are 9 consecutive bits of the rows
    {
        const int NS = 6;                                           // 9 * 9 * 6 = 486 items for 512 threads
        const int per = (((n4 + NS - 1) / NS) + 3) & ~3;
        for (int item = tid; item < FR * FT * NS; item += MBLOCK) {
            // (x translation slowest: the nine x translations of a beam read rows u .. u + 8 of one column word - two LDS banks - and
            // must not sit in the lanes of one wave)
            const int sl = item % NS, ir = (item / NS) % FR - M_FINE_R, ix = item / (NS * FR) - M_FINE_T;
            const double dth = (double)(cir * M_COARSE + ir) * a.d0;
            float sn, cs;
            __sincosf(gthf + (float)dth, &sn, &cs);
            const float tx = fx + (float)(ctx + ix), ty0 = fy + (float)(cty - M_FINE_T);
            int sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            const int b_lo = sl * per, b_hi = min(n4, b_lo + per);
            for (int bb = b_lo; bb < b_hi; bb += 120) {             // two fields per beam: byte lanes stay below 256
                uint32_t accA = 0, accB = 0; int accC = 0;
                const int be = min(b_hi, bb + 120);
                for (int b = bb; b < be; b += 4) {
                    const float4 bx4 = *reinterpret_cast<const float4*>(s.fx4 + b), by4 = *reinterpret_cast<const float4*>(s.fy4 + b);
                    const float ex[4] = {cs * bx4.x - sn * by4.x + tx, cs * bx4.y - sn * by4.y + tx, cs * bx4.z - sn * by4.z + tx, cs * bx4.w - sn * by4.w + tx};
                    const float ey[4] = {sn * bx4.x + cs * by4.x + ty0, sn * bx4.y + cs * by4.y + ty0, sn * bx4.z + cs * by4.z + ty0, sn * bx4.w + cs * by4.w + ty0};
                    uint32_t ol[4], oh[4], dl[4], dh[4]; int sh[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int u = (int)floorf(ex[k]), w0 = (int)floorf(ey[k]);
                        const bool in = (unsigned)u < (unsigned)N && w0 >= 0 && w0 + 8 < N;
                        const int i0 = in ? u * W + (w0 >> 5) : 0;
                        sh[k] = w0 & 31;
                        const bool two = in && sh[k] > 23 && (w0 >> 5) + 1 < W;
#ifdef FINE_ABLATE_LDS     // diagnostic: half of the fine level's LDS reads (wrong scores)
                        ol[k] = in ? s.occ[i0] : 0u; dl[k] = ol[k];
                        oh[k] = 0u; dh[k] = 0u; (void)two;
#else
                        ol[k] = in ? s.occ[i0] : 0u; dl[k] = in ? s.dil[i0] : 0u;
                        oh[k] = two ? s.occ[i0 + 1] : 0u; dh[k] = two ? s.dil[i0 + 1] : 0u;
#endif
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t bo = __builtin_amdgcn_alignbit(oh[k], ol[k], (uint32_t)sh[k]) & 0x1FFu;   // funnel shift of hi:lo
                        const uint32_t bd = __builtin_amdgcn_alignbit(dh[k], dl[k], (uint32_t)sh[k]) & 0x1FFu;
                        accA += (((bo & 0xFu) * 0x00204081u) & 0x01010101u) + (((bd & 0xFu) * 0x00204081u) & 0x01010101u);
                        accB += ((((bo >> 4) & 0xFu) * 0x00204081u) & 0x01010101u) + ((((bd >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
                        accC += (int)(bo >> 8) + (int)(bd >> 8);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { sum[j] += (accA >> (8 * j)) & 0xFFu; sum[4 + j] += (accB >> (8 * j)) & 0xFFu; }
                sum[8] += accC;
            }
            const int base = ((ir + M_FINE_R) * FT + (ix + M_FINE_T)) * FT;
#pragma unroll
            for (int j = 0; j < 9; ++j) if (sum[j]) atomicAdd(&s.sc[base + j], sum[j]);
        }
    }
    __syncthreads();
    for (int cnd = tid; cnd < n_fine; cnd += MBLOCK) {
        const int ir = cnd / (FT * FT) - M_FINE_R, ix = (cnd / FT) % FT - M_FINE_T, iy = cnd % FT - M_FINE_T;
        const double dth = (double)(cir * M_COARSE + ir) * a.d0;
        const int dx = ctx + ix, dy = cty + iy;
        // candidates outside the search window (matchScanCustom.m:52-57) do not take part
        if (!(fabs(dth) < a.rot_range && fabs((double)dx) < rxc && fabs((double)dy) < ryc)) s.sc[cnd] = -1;
        atomicMax(&s_best, s.sc[cnd]);
    }
    __syncthreads();
    if (tid == 0) s_bestc = INT_MAX;
    __syncthreads();
    const int best = s_best;
    for (int cnd = tid; cnd < n_fine; cnd += MBLOCK) {
        if (s.sc[cnd] == best) {
            const int ir = cnd / (FT * FT) - M_FINE_R, ix = (cnd / FT) % FT - M_FINE_T, iy = cnd % FT - M_FINE_T;
            int dr = cir * M_COARSE + ir, dx = ctx + ix, dy = cty + iy;
            int key = (min(dr * dr + dx * dx + dy * dy, 32767) << 16) | cnd;
            atomicMin(&s_bestc, key);
        }
    }
    __syncthreads();
    const int fbest = s_bestc & 0xFFFF;
    const double bth = (double)(cir * M_COARSE + (fbest / (FT * FT) - M_FINE_R)) * a.d0;
    const double bdx = (double)(ctx + (fbest / FT) % FT - M_FINE_T) * a.mcs, bdy = (double)(cty + fbest % FT - M_FINE_T) * a.mcs;

    MSTAMP(4);
    // score of the selected pose over ALL beams (the search levels subsample them)
    __syncthreads();
    if (tid == 0) s_best = 0;
    __syncthreads();
    {
        double snd, csd;
        sincos(gth + bth, &snd, &csd);
        const float sn = (float)snd, cs = (float)csd;
        const float tx = fx + (float)(bdx / a.mcs), ty = fy + (float)(bdy / a.mcs);
        const float inv1 = (float)(1.0 / a.mcs);
        int sc = 0;
        for (int b = tid; b < nb; b += MBLOCK) {
            const float bxs = a.sel_x[b] * inv1, bys = a.sel_y[b] * inv1;   // (all beams: from global memory, the LDS holds the decimated sets)
            float ex = cs * bxs - sn * bys + tx, ey = sn * bxs + cs * bys + ty;
            sc += field_hit(s, N, (int)floorf(ex), (int)floorf(ey));
        }
        for (int off = 32; off > 0; off >>= 1) sc += __shfl_down(sc, off, 64);
        if ((tid & 63) == 0) atomicAdd(&s_best, sc);
    }
    __syncthreads();
    const int full_score = s_best;

    // ---- covariance: second moments of exp((s - s_best)/tau) over the fine candidates ----------------------------
    {
        const double tau = fmax(1.0, 0.02 * (double)((nb + 3) / 4) * 2.0);
        double m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int cnd = tid; cnd < n_fine; cnd += MBLOCK) {
            int sc = s.sc[cnd];
            if (sc < 0) continue;
            const int ir = cnd / (FT * FT) - M_FINE_R, ix = (cnd / FT) % FT - M_FINE_T, iy = cnd % FT - M_FINE_T;
            double w = exp((double)(sc - best) / tau);
            double ex = (double)(ctx + ix) * a.mcs - bdx, ey = (double)(cty + iy) * a.mcs - bdy;
            double et = (double)(cir * M_COARSE + ir) * a.d0 - bth;
            m[0] += w; m[1] += w * ex; m[2] += w * ey; m[3] += w * et;
            m[4] += w * ex * ex; m[5] += w * ex * ey; m[6] += w * ex * et; m[7] += w * ey * ey; m[8] += w * ey * et; m[9] += w * et * et;
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {                    // fixed-order reduction: results must not depend on wave timing
            const double x = wave_sum_lane0_f64(m[k]);
            if ((tid & 63) == 0) s_wmom[tid >> 6][k] = x;
        }
    }
    __syncthreads();
    if (tid < 10) { double x = 0.0; for (int w = 0; w < MBLOCK / 64; ++w) x += s_wmom[w][tid]; s_mom[tid] = x; }
    __syncthreads();
    MSTAMP(5);
#ifdef RBPF_STAMPS
    if (tid == 0 && !a.ndt_occ) for (int k = 0; k < 8; ++k) atomicAdd(&v.stats[8 + k], (unsigned long long)st_acc[k]);   // with the NDT stage on, its kernel reports
#endif
    if (tid == 0) {
        double* o = a.out + (size_t)p * 13;
        const bool ok = best > 0 && nb > 0;                       // no overlap at all: matchScanCustom.m:25-28
        o[0] = s_g[0] + bdx; o[1] = s_g[1] + bdy; o[2] = s_g[2] + bth;   // hybridmap.py:253-255
        if (!ok) {
            for (int k = 0; k < 9; ++k) o[3 + k] = NAN;
            o[12] = 0.0;
        } else {
            const double sw = s_mom[0];
            double mu[3] = {s_mom[1] / sw, s_mom[2] / sw, s_mom[3] / sw};
            double c[3][3];
            c[0][0] = s_mom[4] / sw - mu[0] * mu[0]; c[0][1] = s_mom[5] / sw - mu[0] * mu[1]; c[0][2] = s_mom[6] / sw - mu[0] * mu[2];
            c[1][1] = s_mom[7] / sw - mu[1] * mu[1]; c[1][2] = s_mom[8] / sw - mu[1] * mu[2]; c[2][2] = s_mom[9] / sw - mu[2] * mu[2];
            c[1][0] = c[0][1]; c[2][0] = c[0][2]; c[2][1] = c[1][2];
            // floor: a quarter cell / a quarter rotation step of quantisation noise
            const double fl_t = (a.mcs * a.mcs) / 16.0, fl_r = (a.d0 * a.d0) / 16.0;
            c[0][0] = fmax(c[0][0], 0.0) + fl_t; c[1][1] = fmax(c[1][1], 0.0) + fl_t; c[2][2] = fmax(c[2][2], 0.0) + fl_r;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[3 + 3 * i + j] = c[i][j];
            o[12] = 0.5 * (double)full_score;
        }
        if (a.ndt_aux) {
            double* x = a.ndt_aux + (size_t)p * 5;
            x[0] = bdx / a.mcs; x[1] = bdy / a.mcs; x[2] = bth; x[3] = (double)full_score; x[4] = ok ? 1.0 : 0.0;
        }
    }
}

// ---- NDT refinement (oracle/matcher_oracle.py states the algorithm) ---------------------------------------------
// The reference cloud is the occupancy bitmask itself: the points of an NDT cell are the centres of its occupied
// matcher cells, so a cell's mean and covariance follow from integer sums over its nc x nc bits.
struct NdtCell { double mx, my, b00, b01, b11; };          // mean (offset from the cell's first matcher cell), inverse covariance
__device__ inline bool ndt_cell_stats(int n, int sx, int sy, int sxx, int sxy, int syy, NdtCell& c) {
    if (n < 3) return false;
    const double nn = (double)n, mx = (double)sx / nn, my = (double)sy / nn;
    const double ca = ((double)sxx - (double)sx * mx) / (nn - 1.0);
    const double cb = ((double)sxy - (double)sx * my) / (nn - 1.0);
    const double cc = ((double)syy - (double)sy * my) / (nn - 1.0);
    const double half_tr = 0.5 * (ca + cc), disc = sqrt(0.25 * (ca - cc) * (ca - cc) + cb * cb);
    const double l1 = half_tr + disc, l2 = half_tr - disc;
    double k = 0.0;
    if (l2 < 1e-3 * l1) { k = (1e-3 * l1 - l2) / (l1 - l2); if (!(fabs(k) <= 1.79e308)) k = 0.0; }   // smaller eigenvalue >= 0.001 * larger
    const double a2 = ca + k * (l1 - ca), b2 = cb + k * (-cb), c2 = cc + k * (l1 - cc);
    const double det = a2 * c2 - b2 * b2;
    c.mx = mx; c.my = my; c.b00 = c2 / det; c.b01 = -b2 / det; c.b11 = a2 / det;
    return true;
}
// one (beam, NDT cell) term of f = -score, its gradient and Hessian: m[0..9] = f, g x y t, H xx xy xt yy yt tt
__device__ __forceinline__ void ndt_term(const NdtCell& c, double qx0, double qy0, double ex, double ey, double rx, double ry, double* m) {
    const double dx = ex - (qx0 + 0.5 + c.mx), dy = ey - (qy0 + 0.5 + c.my);
    const double e0 = c.b00 * dx + c.b01 * dy, e1 = c.b01 * dx + c.b11 * dy;
    const double sg = exp(-0.5 * (dx * e0 + dy * e1));
    const double c0 = e0, c1 = e1, c2 = e0 * (-ry) + e1 * rx;
    const double bj0 = c.b00 * (-ry) + c.b01 * rx, bj1 = c.b01 * (-ry) + c.b11 * rx;
    m[0] -= sg;
    m[1] += sg * c0; m[2] += sg * c1; m[3] += sg * c2;
    m[4] += sg * (-c0 * c0 + c.b00);
    m[5] += sg * (-c0 * c1 + c.b01);
    m[6] += sg * (-c0 * c2 + bj0);
    m[7] += sg * (-c1 * c1 + c.b11);
    m[8] += sg * (-c1 * c2 + bj1);
    m[9] += sg * (-c2 * c2 + (-ry) * bj0 + rx * bj1 + e0 * (-rx) + e1 * (-ry));
}
// The same term in single precision, two NDT cells at a time (the hot form, nc == 2): the offset from the cell mean comes
// from the beam's fractional position (formed in double: coordinates run to hundreds of cells) and is small; everything after
// it is packed float arithmetic with fused multiply-adds, summed per thread in float and across threads in double.
// oracle/matcher_oracle.py states the same formulas (float32 terms, float64 sums); the two agree to ~1e-7, not bit for bit.
// A cell without a Gaussian has a far-away mean in the table (NDT_FAR): its weight underflows to exactly 0.
typedef float ndt_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ndt_f2 ndt_fma2(ndt_f2 a, ndt_f2 b, ndt_f2 c) { return __builtin_elementwise_fma(a, b, c); }
static const float NDT_FAR = 1.0e6f;
__device__ __forceinline__ void ndt_term_pair(ndt_f2 b00, ndt_f2 b01, ndt_f2 b11, ndt_f2 dx, ndt_f2 dy, float rxs, float rys, ndt_f2* m) {
    const ndt_f2 rx = {rxs, rxs}, ry = {rys, rys};
    const ndt_f2 e0 = ndt_fma2(b01, dy, b00 * dx), e1 = ndt_fma2(b11, dy, b01 * dx);
    const ndt_f2 q = ndt_fma2(dy, e1, dx * e0) * -0.5f;
    const ndt_f2 sg = {__expf(q.x), __expf(q.y)};
    const ndt_f2 c2 = ndt_fma2(e1, rx, -(e0 * ry));
    const ndt_f2 bj0 = ndt_fma2(b01, rx, -(b00 * ry)), bj1 = ndt_fma2(b11, rx, -(b01 * ry));
    m[0] -= sg;
    m[1] = ndt_fma2(sg, e0, m[1]); m[2] = ndt_fma2(sg, e1, m[2]); m[3] = ndt_fma2(sg, c2, m[3]);
    m[4] = ndt_fma2(sg, ndt_fma2(-e0, e0, b00), m[4]);
    m[5] = ndt_fma2(sg, ndt_fma2(-e0, e1, b01), m[5]);
    m[6] = ndt_fma2(sg, ndt_fma2(-e0, c2, bj0), m[6]);
    m[7] = ndt_fma2(sg, ndt_fma2(-e1, e1, b11), m[7]);
    m[8] = ndt_fma2(sg, ndt_fma2(-e1, c2, bj1), m[8]);
    const ndt_f2 h = ndt_fma2(rx, bj1, -(ry * bj0)) - ndt_fma2(e1, ry, e0 * rx);
    m[9] = ndt_fma2(sg, ndt_fma2(-c2, c2, h), m[9]);
}
// the NDT kernel's copy of the field has a zero row above and below and a zero word left and right of every row
__device__ __forceinline__ uint32_t ndt_bit(const uint32_t* occ, int N, int u, int w) {
    if ((unsigned)u >= (unsigned)N || (unsigned)w >= (unsigned)N) return 0u;
    return (occ[(u + 1) * ((N >> 5) + 2) + (w >> 5) + 1] >> (w & 31)) & 1u;
}
// sum over the wave's 64 lanes, complete in lane 63: four steps inside the rows of 16 lanes and two broadcasts of a row's
// last lane, all through the data-parallel-primitive path (no LDS round trips)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double ndt_dpp_f64(double x) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    int lo, hi;
    if (ROW_MASK == 0xF) {                   // every lane reads a lane: no value needed for the rows that stay out
        lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, 0xF, 0xF, false);
        hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, false);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    }
    return __builtin_bit_cast(double, ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ double ndt_wave_sum63(double x) {
    x += ndt_dpp_f64<0xB1, 0xF>(x);          // quad_perm [1,0,3,2]
    x += ndt_dpp_f64<0x4E, 0xF>(x);          // quad_perm [2,3,0,1]
    x += ndt_dpp_f64<0x141, 0xF>(x);         // row_half_mirror
    x += ndt_dpp_f64<0x140, 0xF>(x);         // row_mirror: every lane of a row holds the row's sum
    x += ndt_dpp_f64<0x142, 0xA>(x);         // row_bcast:15 into rows 1 and 3
    x += ndt_dpp_f64<0x143, 0xC>(x);         // row_bcast:31 into rows 2 and 3
    return x;
}
// One beam at the pose (tx, ty, theta), region cell units; four overlapping grids shifted by half an NDT cell and
// anchored to the global cell index: the general form.  (nc == 2, i.e. 0.05 m matcher cells -- every shipped
// configuration -- goes through a 16-entry table by 2 x 2 occupancy pattern in the kernel instead.)
__device__ __forceinline__ void ndt_point(const uint32_t* occ, int N, int nc, int ox, int oy,
                                          double bx, double by, double tx, double ty, double sn, double cs, double* m) {
    const double rx = cs * bx - sn * by, ry = sn * bx + cs * by;
    const double ex = rx + tx, ey = ry + ty;
    const int u = (int)floor(ex), w = (int)floor(ey);
    if ((unsigned)u >= (unsigned)N || (unsigned)w >= (unsigned)N) return;
    const int h = nc >> 1;
    for (int g = 0; g < 4; ++g) {
        const int gx = (g & 1) ? h : 0, gy = (g & 2) ? h : 0;
        int mu = (u + ox - gx) % nc; if (mu < 0) mu += nc;
        int mw = (w + oy - gy) % nc; if (mw < 0) mw += nc;
        const int u0 = u - mu, w0 = w - mw;
        int n = 0, sx = 0, sy = 0, sxx = 0, sxy = 0, syy = 0;
        for (int i = 0; i < nc; ++i)
            for (int j = 0; j < nc; ++j)
                if (ndt_bit(occ, N, u0 + i, w0 + j)) { ++n; sx += i; sy += j; sxx += i * i; sxy += i * j; syy += j * j; }
        NdtCell c;
        if (!ndt_cell_stats(n, sx, sy, sxx, sxy, syy, c)) continue;
        ndt_term(c, (double)u0, (double)w0, ex, ey, rx, ry, m);
    }
}

// (H + lam * diag(|H_ii| + 1e-12)) d = -g by Cholesky; false when the damped matrix is not positive definite
__device__ inline bool ndt_lm_step(const double* m, double lam, double* d) {
    const double A00 = m[4] + lam * (fabs(m[4]) + 1e-12), A11 = m[7] + lam * (fabs(m[7]) + 1e-12), A22 = m[9] + lam * (fabs(m[9]) + 1e-12);
    const double A10 = m[5], A20 = m[6], A21 = m[8];
    if (!(A00 > 0)) return false;
    const double l00 = sqrt(A00), l10 = A10 / l00, l20 = A20 / l00;
    const double d1 = A11 - l10 * l10;
    if (!(d1 > 0)) return false;
    const double l11 = sqrt(d1), l21 = (A21 - l20 * l10) / l11;
    const double d2 = A22 - l20 * l20 - l21 * l21;
    if (!(d2 > 0)) return false;
    const double l22 = sqrt(d2);
    const double y0 = -m[1] / l00, y1 = (-m[2] - l10 * y0) / l11, y2 = (-m[3] - l20 * y0 - l21 * y1) / l22;
    d[2] = y2 / l22;
    d[1] = (y1 - l21 * d[2]) / l11;
    d[0] = (y0 - l10 * d[1] - l20 * d[2]) / l00;
    return true;
}

static const int NBLOCK = 256;            // NDT kernel: threads per particle
static const int NDT_STRIDE = 1;          // beams used by the ascent (1: all; the final score always uses all)

size_t ndt_lds_bytes(int N, int B) { (void)B; return (size_t)(N + 2) * (N / 32 + 2) * 4 + 64; }    // the field with its zero border only: four workgroups per CU at N = 512

// The second matcher stage (matchScanCustom.m:32-50) for the particles whose grid stage succeeded: damped Newton ascent
// of the NDT score from the grid optimum.  Thread 0 holds the optimiser state; every evaluation is one pass over the
// (beam, grid) pairs and a fixed-order reduction, so results do not depend on wave timing.
__global__ __launch_bounds__(NBLOCK, 4) void ndt_kernel(DevView v, MatchArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N, W = N >> 5, tid = threadIdx.x, p = blockIdx.x;
    uint32_t* occ = reinterpret_cast<uint32_t*>(smem);
    const float* __restrict__ bx = a.sel_x;     // beams stay in global memory (the same 8.6 KB for every workgroup: L1/L2 hits)
    const float* __restrict__ by = a.sel_y;
    const float inv = (float)(1.0 / a.mcs);     // matcher cells per metre, float as the grid stage stages its beams
    __shared__ double s_g[3], s_rng[2], s_trial[6], s_w[NBLOCK / 64][10];     // s_trial: pose, -, sin, cos
    __shared__ int s_org[2], s_go;
    __shared__ float s_lut[5][16];              // a 2 x 2 NDT cell by occupancy pattern: 0.5 + mean x, 0.5 + mean y, inverse covariance (3)
    __shared__ unsigned s_ok;
#ifdef RBPF_STAMPS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = clock64();
#endif
    if (a.dup_of && a.dup_of[p] != p) return;                  // exact duplicate of another particle (see match_kernel)
    const double* aux = a.ndt_aux + (size_t)p * 5;
    if (aux[4] == 0.0 || a.n_sel <= 0) return;                  // the grid stage failed: matchScanCustom.m:25-28, uniform
    if (tid == 0) { match_frame(v, a, p, s_g, s_rng, s_org); s_ok = 0; }
    const int WPD = W + 2;                                      // words of a row of the bordered field
    {
        const uint32_t* src = a.ndt_occ + (size_t)p * N * W;
        int u = tid / W, wv = tid % W;                          // (one division; the position then advances by NBLOCK words)
        const int du = NBLOCK / W, dw = NBLOCK % W;
        const int SL = 16;                                      // loads in flight per thread
        for (int i0 = tid; i0 < N * W; i0 += SL * NBLOCK) {
            uint32_t val[SL]; int dst[SL];
#pragma unroll
            for (int k = 0; k < SL; ++k) {
                const int i = i0 + k * NBLOCK;
                val[k] = i < N * W ? src[i] : 0u;
                dst[k] = (u + 1) * WPD + wv + 1;
                u += du; wv += dw;
                if (wv >= W) { wv -= W; ++u; }
            }
#pragma unroll
            for (int k = 0; k < SL; ++k)
                if (i0 + k * NBLOCK < N * W) occ[dst[k]] = val[k];
        }
        for (int i = tid; i < WPD; i += NBLOCK) { occ[i] = 0; occ[(N + 1) * WPD + i] = 0; }
        for (int i = tid; i < N; i += NBLOCK) { occ[(i + 1) * WPD] = 0; occ[(i + 1) * WPD + W + 1] = 0; }
    }
    __syncthreads();
    if (tid < 16) {             // pattern bit 0: cell (0,0), bit 1: (0,1), bit 2: (1,0), bit 3: (1,1)
        const int b0 = tid & 1, b1 = (tid >> 1) & 1, b2 = (tid >> 2) & 1, b3 = (tid >> 3) & 1;
        NdtCell c = {0, 0, 0, 0, 0};
        const bool okc = ndt_cell_stats(b0 + b1 + b2 + b3, b2 + b3, b1 + b3, b2 + b3, b3, b1 + b3, c);
        if (okc) atomicOr(&s_ok, 1u << tid);
        s_lut[0][tid] = okc ? (float)(0.5 + c.mx) : NDT_FAR; s_lut[1][tid] = okc ? (float)(0.5 + c.my) : NDT_FAR;
        s_lut[2][tid] = okc ? (float)c.b00 : 1.0f; s_lut[3][tid] = okc ? (float)c.b01 : 0.0f; s_lut[4][tid] = okc ? (float)c.b11 : 1.0f;
    }
    const int ox = s_org[0], oy = s_org[1], nb = a.n_sel, nc = a.ndt_nc;
    const double gth = s_g[2];
    const double X0 = s_g[0] / a.mcs - (double)ox + a.cell_off, Y0 = s_g[1] / a.mcs - (double)oy + a.cell_off;
    // the optimiser state lives in LDS: one thread uses it, and registers would be reserved for it in every lane
    __shared__ double s_cur[10], s_pose[3], s_step[3], s_lam;
    __shared__ int s_evals;
    double* const cur = s_cur; double* const ndt_p = s_pose; double* const step = s_step;
    double& lam = s_lam; int& evals = s_evals;
    if (tid == 0) {
        ndt_p[0] = X0 + aux[0]; ndt_p[1] = Y0 + aux[1]; ndt_p[2] = gth + aux[2];
        lam = 1e-3; step[0] = step[1] = step[2] = 0.0; evals = 0;
    }
    if (tid == 0) { s_trial[0] = ndt_p[0]; s_trial[1] = ndt_p[1]; s_trial[2] = ndt_p[2]; sincos(ndt_p[2], &s_trial[4], &s_trial[5]); s_go = 1; }
    __syncthreads();
    MSTAMP(0);                                   // staging
    // one evaluation: every `stride`-th beam at the pose in s_trial; per-wave sums land in s_w (then a barrier)
    auto evaluate = [&](int stride) {
        const double tx = s_trial[0], ty = s_trial[1], snd = s_trial[4], csd = s_trial[5];
        double m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int nbs = (nb + stride - 1) / stride;
        if (nc == 2) {
            // One beam per lane and pass.  The four NDT cells that hold a point (one per shifted grid) are the four 2 x 2
            // windows of the 3 x 3 cells round it, whatever the grids' phase: two packed pairs of terms, no branches.
            ndt_f2 mf[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) mf[k] = ndt_f2{0.0f, 0.0f};
            for (int i = tid; i < nbs; i += NBLOCK) {
                const int b = i * stride;
                const double bxd = (double)(bx[b] * inv), byd = (double)(by[b] * inv);
                const double rx = csd * bxd - snd * byd, ry = snd * bxd + csd * byd;
                const double ex = rx + tx, ey = ry + ty;
                const double fu = floor(ex), fw = floor(ey);
                const int u = (int)fu, w = (int)fw;
                const bool inside = (unsigned)u < (unsigned)N && (unsigned)w < (unsigned)N;
                const float fxr = (float)(ex - fu), fyr = (float)(ey - fw);          // position inside the cell, [0, 1)
                // the 3 x 3 occupancy bits round the beam's cell: bit 3 r + c = cell (u - 1 + r, w - 1 + c); the border of
                // zeros makes every address valid once (u, w) is clamped into the region
                uint32_t nb9 = 0;
                {
                    const int uc = min(max(u, 0), N - 1), c0 = min(max(w, 0), N - 1) - 1;     // first column, -1 .. N - 2
                    const uint32_t* row = occ + uc * WPD + ((c0 >> 5) + 1);
                    const uint32_t sh = (uint32_t)(c0 & 31);
#pragma unroll
                    for (int r = 0; r < 3; ++r)
                        nb9 |= (__builtin_amdgcn_alignbit(row[r * WPD + 1], row[r * WPD], sh) & 7u) << (3 * r);
                    nb9 = inside ? nb9 : 0u;
                }
                const float rxf = (float)rx, ryf = (float)ry;
                const float x1 = fxr + 1.0f, y1 = fyr + 1.0f;
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2) {                // windows (a2, 0) and (a2, 1): rows u - 1 + a2 .., columns w - 1 .. / w ..
                    const uint32_t pA = ((nb9 >> (3 * a2)) & 3u) | ((nb9 >> (3 * a2 + 1)) & 12u);
                    const uint32_t pB = ((nb9 >> (3 * a2 + 1)) & 3u) | ((nb9 >> (3 * a2 + 2)) & 12u);
                    const ndt_f2 cx = {s_lut[0][pA], s_lut[0][pB]}, cy = {s_lut[1][pA], s_lut[1][pB]};
                    const ndt_f2 b00 = {s_lut[2][pA], s_lut[2][pB]}, b01 = {s_lut[3][pA], s_lut[3][pB]}, b11 = {s_lut[4][pA], s_lut[4][pB]};
                    const float xs = a2 ? fxr : x1;
                    const ndt_f2 dx = ndt_f2{xs, xs} - cx, dy = ndt_f2{y1, fyr} - cy;
                    ndt_term_pair(b00, b01, b11, dx, dy, rxf, ryf, mf);
                }
            }
#pragma unroll
            for (int k = 0; k < 10; ++k) m[k] = (double)mf[k].x + (double)mf[k].y;
        } else {
            for (int i = tid; i < nbs; i += NBLOCK) ndt_point(occ, N, nc, ox, oy, (double)(bx[i * stride] * inv), (double)(by[i * stride] * inv), tx, ty, snd, csd, m);
        }
        MSTAMP(1);                               // beams
#pragma unroll
        for (int k = 0; k < 10; ++k) m[k] = ndt_wave_sum63(m[k]);
        if ((tid & 63) == 63) {
#pragma unroll
            for (int k = 0; k < 10; ++k) s_w[tid >> 6][k] = m[k];
        }
        __syncthreads();
        MSTAMP(2);                               // reduction + barrier
    };
    // (NDT_STRIDE > 1 would run the ascent on a subsample; the score that decides acceptance always covers all beams.)
    for (;;) {
        evaluate(NDT_STRIDE);
        if (tid == 0) {
            const double tx = s_trial[0], ty = s_trial[1], tt = s_trial[2];
            double tr[10];
            for (int k = 0; k < 10; ++k) { double x = 0.0; for (int w = 0; w < NBLOCK / 64; ++w) x += s_w[w][k]; tr[k] = x; }
            bool go = true;
            if (evals == 0) { for (int k = 0; k < 10; ++k) cur[k] = tr[k]; }
            else if (tr[0] < cur[0]) {
                for (int k = 0; k < 10; ++k) cur[k] = tr[k];
                ndt_p[0] = tx; ndt_p[1] = ty; ndt_p[2] = tt;
                lam = fmax(lam * 0.1, 1e-7);
            } else {
                lam *= 100.0;                  // a rejected trial: the quadratic model is off (a cell boundary), damp hard
                if (lam > 1e7) go = false;
            }
            ++evals;
            if (go && evals > a.ndt_max_iter) go = false;
            if (go) {
                bool have = false;
                while (lam <= 1e7 && !(have = ndt_lm_step(cur, lam, step))) lam *= 10.0;
                if (!have) go = false;
                else if (fmax(fabs(step[0]), fabs(step[1])) < 1e-2 && fabs(step[2]) < 2e-4) go = false;   // converged: step below 0.01 cells, 2e-4 rad
                else { s_trial[0] = ndt_p[0] + step[0]; s_trial[1] = ndt_p[1] + step[1]; s_trial[2] = ndt_p[2] + step[2]; }
            }
            if (!go) { s_trial[0] = ndt_p[0]; s_trial[1] = ndt_p[1]; s_trial[2] = ndt_p[2]; }
            sincos(s_trial[2], &s_trial[4], &s_trial[5]);
            s_go = go ? 1 : 0;
        }
        __syncthreads();
        MSTAMP(3);                               // optimiser step (one thread) + barrier
        if (!s_go) break;
    }
    if (NDT_STRIDE != 1) evaluate(1);
    if (tid == 0) {
        atomicAdd(&v.stats[ST_NDT_RUNS], 1ull); atomicAdd(&v.stats[ST_NDT_EVALS], (unsigned long long)evals);
        // matchScanCustom.m:38-44: the NDT pose replaces the grid pose when it is inside the search window (:52-57)
        // and twice its score exceeds the grid score; the covariance stays the grid one
        double score = -cur[0];
        if (NDT_STRIDE != 1) { score = 0.0; for (int w = 0; w < NBLOCK / 64; ++w) score -= s_w[w][0]; }
        const double ddx = (ndt_p[0] - X0) * a.mcs, ddy = (ndt_p[1] - Y0) * a.mcs, ddt = ndt_p[2] - gth;
        const bool valid = fabs(ddx) < s_rng[0] && fabs(ddy) < s_rng[1] && fabs(remainder(ddt, 6.283185307179586)) < a.rot_range &&
                           (s_g[0] + ddx != 0.0 || s_g[1] + ddy != 0.0 || s_g[2] + ddt != 0.0);
        if (valid && (a.ndt == 2 || 2.0 * score > 0.5 * aux[3])) {
            double* o = a.out + (size_t)p * 13;
            o[0] = s_g[0] + ddx; o[1] = s_g[1] + ddy; o[2] = s_g[2] + ddt; o[12] = score;
            atomicAdd(&v.stats[ST_NDT_ACCEPTED], 1ull);
        }
#ifdef RBPF_STAMPS
        for (int k = 0; k < 8; ++k) atomicAdd(&v.stats[8 + k], (unsigned long long)st_acc[k]);
#endif
    }
}


// region edge in matcher cells and the matcher-cell scale for a configuration
void match_geometry(const rbpf_config& c, double cell_size, int& N, int& ds, double& mcs, double& d0, int& n_coarse_rot) {
    ds = 1;
    for (;;) {
        mcs = cell_size * ds;
        int half = (int)ceil((c.match_max_range + 0.5 + 0.7) / mcs) + 2;
        N = ((2 * half + 31) / 32) * 32;
        // occupancy + dilation bitmasks must fit in LDS next to the beam list and the score table
        if (2 * (size_t)N * (N / 32) * 4 <= 120 * 1024 || ds >= 8) break;
        ds *= 2;
    }
    d0 = mcs / c.match_max_range;
    const double rot = 3.141592653589793 / 6;
    n_coarse_rot = (int)floor(rot / (M_COARSE * d0));
    if (n_coarse_rot * M_COARSE * d0 >= rot) --n_coarse_rot;
    if (n_coarse_rot < 0) n_coarse_rot = 0;
}

int match_max_coarse(int n_coarse_rot, double max_range_m, double mcs) {
    int k = (int)ceil(max_range_m / mcs / M_COARSE);
    return (2 * n_coarse_rot + 1) * (2 * k + 1) * (2 * k + 1);
}
int match_per_rot(double max_range_m, double mcs) {          // coarse candidates of one rotation, at most
    int k = (int)ceil(max_range_m / mcs / M_COARSE);
    return (2 * k + 1) * (2 * k + 1);
}

// t0 / t1 (or nullptr): timing events that take the kernel's own start and end (carried by its dispatch: hipExtLaunchKernelGGL)
static void launch_match(const DevView& v, const MatchArgs& a_in, int grid, size_t lds, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
    MatchArgs a = a_in;
    a.sc_cap = match_sc_capacity(a.N, a.cap_sel, lds);
    static size_t lds_set[MAX_DEVICES] = {};
    ensure_dynamic_lds(reinterpret_cast<const void*>(match_kernel), lds, lds_set);
    if (t0 && t1) hipExtLaunchKernelGGL(match_kernel, dim3(grid), dim3(MBLOCK), lds, s, t0, t1, 0, v, a);
    else hipLaunchKernelGGL(match_kernel, dim3(grid), dim3(MBLOCK), lds, s, v, a);
}

// NDT cell edge in matcher cells for matchScanCustom.m:37 ('CellSize', 0.1)
int ndt_cells(double mcs) { return (int)floor(0.1 / mcs + 0.5); }

static void launch_ndt(const DevView& v, const MatchArgs& a, int grid, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
    size_t lds = ndt_lds_bytes(a.N, a.cap_sel);
    static size_t lds_set[MAX_DEVICES] = {};
    ensure_dynamic_lds(reinterpret_cast<const void*>(ndt_kernel), lds, lds_set);
    if (t0 && t1) hipExtLaunchKernelGGL(ndt_kernel, dim3(grid), dim3(NBLOCK), lds, s, t0, t1, 0, v, a);
    else hipLaunchKernelGGL(ndt_kernel, dim3(grid), dim3(NBLOCK), lds, s, v, a);
}

// stage 1: the grid search (match_kernel); stage 2: the NDT refinement (ndt_kernel) -- two calls so that the host can
// time them apart.  Returns whether the NDT stage is active for this configuration.
bool launch_match_particles(const DevView& v, int mode, const double* d_ref, int n_ref, double* d_out, int N, int ds,
                            double mcs, double d0, int ncr, double max_range, int cap_sel, size_t lds, int stage, hipStream_t s,
                            hipEvent_t t0, hipEvent_t t1) {
    MatchArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = mode; a.single = 0; a.ref_xy = d_ref; a.n_ref = n_ref; a.out = d_out;
    a.N = N; a.ds = ds; a.mcs = mcs; a.d0 = d0; a.rot_range = 3.141592653589793 / 6; a.max_range = max_range;
    a.sel_x = mode ? v.asel_x : v.msel_x; a.sel_y = mode ? v.asel_y : v.msel_y; a.n_sel = mode ? v.n_asel : v.n_msel;
    a.n_coarse_rot = ncr; a.cap_sel = cap_sel;
    a.ndt = v.ndt_refine; a.ndt_nc = ndt_cells(mcs); a.ndt_max_iter = 500;
    a.dup_of = v.dups_valid ? v.dup_of : nullptr;
    const bool ndt = a.ndt && a.ndt_nc >= 2 && v.ndt_occ;
    if (ndt) { a.ndt_occ = v.ndt_occ; a.ndt_aux = v.ndt_aux; }
    if (stage == 1) launch_match(v, a, v.P, lds, s, t0, t1);
    if (stage == 2 && ndt) launch_ndt(v, a, v.P, s, t0, t1);
    return ndt;
}

void launch_match_single(const DevView& v, const double* d_ref, int n_ref, const double* guess3, const double* range3,
                         const float* d_sel_x, const float* d_sel_y, int n_sel, double* d_out, int N, int ds, double mcs,
                         double d0, int ncr, int cap_sel, size_t lds, uint32_t* d_ndt_occ, double* d_ndt_aux, hipStream_t s) {
    MatchArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = 1; a.single = 1; a.ref_xy = d_ref; a.n_ref = n_ref; a.out = d_out;
    for (int i = 0; i < 3; ++i) { a.guess[i] = guess3[i]; a.range[i] = range3[i]; }
    a.N = N; a.ds = ds; a.mcs = mcs; a.d0 = d0; a.rot_range = range3[2]; a.max_range = 15.0;
    a.sel_x = d_sel_x; a.sel_y = d_sel_y; a.n_sel = n_sel; a.n_coarse_rot = ncr; a.cap_sel = cap_sel;
    a.cell_off = 0.5;
    a.ndt = v.ndt_refine; a.ndt_nc = ndt_cells(mcs); a.ndt_max_iter = 500;
    const bool ndt = a.ndt && a.ndt_nc >= 2 && d_ndt_occ;
    if (ndt) { a.ndt_occ = d_ndt_occ; a.ndt_aux = d_ndt_aux; }
    launch_match(v, a, 1, lds, s);
    if (ndt) launch_ndt(v, a, 1, s);
}

}  // namespace rbpf
