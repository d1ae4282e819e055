// kernels_mapfan.hip -- a5: HybridMap.update (hybridmap.py:95-145), the whole ray fan of a particle in ONE LDS window.
//
// One 1024-thread workgroup per particle (16 waves, one workgroup per CU, ~155 KB of LDS).  Same exact semantics as
// kernels_mapupdate.hip (see there for the ordered-replay argument); what differs is the staging:
//
//   * the window is the fan's bounding box in "unrolled storage" coordinates U = lattice * dim + storage index
//     (the reference's float index formula, SURVEY quirk 3, taken from the global-index LUT), so rays are never
//     clipped and every per-window phase of the 128x128 kernel runs once per particle instead of ~9 times;
//   * hit counters are 8-bit fields (bit 7 = "receives an occupied / nearby hit", bits 0-6 = hits).  Cells closer
//     than CHUNK steps to the sensor, which most rays cross, are counted in a separate 31x31 block of 16-bit
//     fields.  Every add returns the old field: if any 8-bit field is ever seen at >= 96 the workgroup gives the
//     particle back untouched (fallback flag) and the 128x128-window kernel processes it - so no field can
//     silently overflow;
//   * ordered events of flagged cells are collected AFTER the walk, when every cell's event count is known:
//     exact-size buckets by prefix sum, filled by re-walking only the 16-step chunks that met a flagged cell.
//
// Anything the layout cannot hold (fan wider than the window, more flagged cells / events / marked chunks than the
// tables) also takes the fallback; nothing has been written to the map at that point.
#include "rbpf_mapupdate.h"

namespace rbpf {

// Diagnostic build only (-DRBPF_STAMPS): thread 0 of every workgroup sums the cycles between phase boundaries.
#ifdef RBPF_STAMPS
#define STAMP(k) do { if (tid == 0) { long long t_ = clock64(); st_acc[k] += t_ - st_prev; st_prev = t_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// hand the particle to the window kernel (uniform over the workgroup; nothing has been written to the map yet);
// the reason codes are tallied in a diagnostic counter: 1 geometry, 2 walk (8-bit guard, flagged-cell table),
// 3 events, 4 replay lists
#define GIVE_BACK(reason) do { if (tid == 0) { v.mu_fallback[p] = (reason); atomicAdd(&v.stats[(reason) == 1 ? ST_FALLBACK_REASONS : (reason) == 2 ? ST_FB_BOUND : ST_FB_TABLES], 1ull); } return; } while (0)

static const int FB = 1024;                    // threads per particle
static const int MINI_R = CHUNK - 1;           // cells with Chebyshev distance <= MINI_R from the start cell ...
static const int MINI_W = 2 * MINI_R + 1;      // ... live in the 16-bit block
static const int FNB_MAX = 4 * FB;             // flagged cells (two per beam at most)
static const int FEV = 5120;                   // ordered events
static const int FMARK = 256;                  // chunks that met a flagged cell more than 32 steps before the ray's end
static const int FLIST = 256;                  // deep buckets (17..64 events) / membership-scan cells per particle
static const int MAXLEV = 20;                  // chunks per ray
static const int FIX_SHIFT = 22;               // fixed-point bits of the walk's DDA (fix_slope)
static const int BLK = 128;                    // cells per block of the flagged-cell directory
// an 8-bit field seen with bits 5 and 6 set (>= 96 hits) -> the particle goes to the window kernel

struct FanGeom {
    int ncell, nblk, fanw, nb, bpad;             // ncell = window capacity in cells (rows * stride of a particle's fan must fit)
    int o_cnt, o_fpre, o_mini, o_rend, o_rinfo, o_perm, o_ux, o_uy, o_gpx, o_gpy, o_bcell, o_floff, o_oldv, o_off,
        o_lists, o_mark, o_rmask, o_dummy, o_bev;
    int bytes;
    bool ok;
};

__host__ __device__ inline int fan_al16(int x) { return (x + 15) & ~15; }

// LDS layout: everything that scales with the beam count and the LUT width first, then as many window cells as fit
// (1 byte of counters + 2 bytes of directory per 128 cells).
__host__ __device__ inline FanGeom fan_geom(int B, int reach) {
    FanGeom g;
    g.fanw = (2 * reach + 8 + 7) & ~7;
    g.nb = 2 * B < FNB_MAX ? 2 * B : FNB_MAX;
    g.bpad = (B + 3) & ~3;
    int fixed = 0;
    fixed += fan_al16(((MINI_W * MINI_W + 1) / 2) * 4) + fan_al16(g.bpad * 4) + fan_al16(B) + fan_al16(g.bpad * 2);
    fixed += 2 * fan_al16(g.fanw * 2) + 2 * fan_al16(g.fanw) + fan_al16(g.nb * 4) + 2 * fan_al16(g.nb) + fan_al16(((g.nb + 1) / 2) * 4);
    fixed += fan_al16(2 * FLIST * 2) + FMARK * 4 + g.bpad * 4 + 256 + FEV * 2;
    const int avail = 160 * 1024 - 1024 - fixed - 64;      // 1 KB for the kernel's static LDS
    int ncell = avail > 0 ? (int)(((long long)avail * 64) / 65) & ~(BLK - 1) : 0;
    if (ncell > BLK * FB) ncell = BLK * FB;                // one directory block per thread in the scan
    g.ncell = ncell;
    g.nblk = ncell / BLK;
    int o = 0;
    g.o_cnt = o;   o += ncell;
    g.o_fpre = o;  o += fan_al16((g.nblk + 2) * 2);
    g.o_mini = o;  o += fan_al16(((MINI_W * MINI_W + 1) / 2) * 4);
    g.o_rend = o;  o += fan_al16(g.bpad * 4);
    g.o_rinfo = o; o += fan_al16(B);
    g.o_perm = o;  o += fan_al16(g.bpad * 2);
    g.o_ux = o;    o += fan_al16(g.fanw * 2);
    g.o_uy = o;    o += fan_al16(g.fanw * 2);
    g.o_gpx = o;   o += fan_al16(g.fanw);
    g.o_gpy = o;   o += fan_al16(g.fanw);
    g.o_bcell = o; o += fan_al16(g.nb * 4);
    g.o_floff = o; o += fan_al16(g.nb);
    g.o_oldv = o;  o += fan_al16(g.nb);
    g.o_off = o;   o += fan_al16(((g.nb + 1) / 2) * 4);
    g.o_lists = o; o += fan_al16(2 * FLIST * 2);
    g.o_mark = o;  o += FMARK * 4;
    g.o_rmask = o; o += fan_al16(g.bpad * 4);
    g.o_dummy = o; o += 256;
    g.o_bev = o;   o += FEV * 2;
    g.bytes = o;
    g.ok = ncell >= 32768 && g.bytes + 1024 <= 160 * 1024 && reach + 1 <= CHUNK * MAXLEV && B <= 4095 && reach >= 3 && 2LL * reach * reach < (1LL << FIX_SHIFT) &&
           (B + 64) * MAXLEV < 65536;
    return g;
}

bool map_update_fan_available(const DevView& v) {
    const int sat = (v.cc.vmax - v.cc.vmin + (-v.cc.emp) - 1) / (-v.cc.emp);
    return fan_geom(v.B, v.reach).ok && v.dim % 32 == 0 && v.L * v.L <= 49 && sat <= 64 && v.cc.emp < 0 && v.cc.vmax - v.cc.vmin <= 127 && v.cc.vmin <= 0 && v.cc.vmax >= 0 &&
           v.cc.vmin >= -127 && (int)sat * -v.cc.emp <= 127 && v.cc.thr >= v.cc.vmin && v.cc.thr < v.cc.vmax;
}

// atomicAdd(&arr[key], 1) for every lane with valid = true, one LDS atomic per distinct key in the wave (a handful of
// keys shared by many lanes would otherwise serialise); returns the value the lane's own add would have returned.
__device__ __forceinline__ int wave_keyed_inc(int* arr, int key, bool valid, int lane) {
    int res = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader, 64);
        const unsigned long long same = __ballot(valid && key == k);
        int base = 0;
        if (lane == leader) base = atomicAdd(&arr[k], __popcll(same));
        base = __shfl(base, leader, 64);
        if (valid && key == k) res = base + __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return res;
}

// 8-bit / 16-bit hit-counter fields packed into 32-bit LDS words
template <bool MINI> struct Fld;
template <> struct Fld<false> {
    static __device__ __forceinline__ int word(int c) { return c >> 2; }
    static __device__ __forceinline__ int sh(int c) { return (c & 3) * 8; }
    static const uint32_t FLAG = 0x80u, MASK = 0xFFu, CNT = 0x7Fu;
};
template <> struct Fld<true> {
    static __device__ __forceinline__ int word(int c) { return c >> 1; }
    static __device__ __forceinline__ int sh(int c) { return (c & 1) * 16; }
    static const uint32_t FLAG = 0x8000u, MASK = 0xFFFFu, CNT = 0x7FFFu;
};

// Fixed-point DDA for the reference's Bresenham variant: minor(j) = floor((2*dmin*j + dmaj) / (2*dmaj)) (rbpf_math.h) equals
// (fix_slope * j + 2^(FIX_SHIFT-1)) >> FIX_SHIFT with fix_slope = ceil(dmin * 2^FIX_SHIFT / dmaj): the accumulated
// excess is below j / 2^FIX_SHIFT, while the exact value is either an integer or at least 1 / (2*dmaj) below the next
// one - so the two floors agree as long as 2 * dmaj * j < 2^FIX_SHIFT (rays up to 1447 cells).
__device__ __forceinline__ uint32_t fix_slope(int dmin, int dmaj) {
    return dmaj ? (((uint32_t)dmin << FIX_SHIFT) + (uint32_t)dmaj - 1u) / (uint32_t)dmaj : 0u;
}

// One chunk of a ray whose index map is the identity from the start cell to the chunk's end: the field index advances
// by constants.  cw = counter words (main window or the 16-bit block), c = field index of step jlo.  SAT: read the
// field first and skip the add on an unflagged cell that already has `sat` hits (max(v + n*emp, vmin) is vmin for every
// n >= sat; spares the serialised same-address atomics next to the sensor); skipped adds go to `sink` (a zero word).
// guard accumulates f & (f << 1) of every 8-bit field seen: bit 6 set = some field was at >= 96.
template <bool MINI, bool SAT>
__device__ __forceinline__ void walk_ident(uint32_t* __restrict__ cw, uint32_t* __restrict__ sink, int cm, int dmajc, int dminc,
                                           uint32_t facc, uint32_t fstep, int jlo, int jhi, int n, uint32_t sat,
                                           bool near_ok, uint32_t& hmask, uint32_t& guard) {
    // cm = field index of (major step jlo, minor offset 0); the minor offset of step j is facc >> FIX_SHIFT (fixed-point
    // DDA, exact: see fix_slope)
    typedef Fld<MINI> F;
    int j = jlo;
    for (; j + 3 <= jhi; j += 4) {
        int cc[4]; uint32_t h[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            cc[u] = cm + __mul24((int)(facc >> FIX_SHIFT), dminc);
            cm += dmajc; facc += fstep;
        }
        uint32_t* ap[4]; uint32_t av[4];
        if (SAT) {
#pragma unroll
            for (int u = 0; u < 4; ++u) h[u] = cw[F::word(cc[u])];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t f = (h[u] >> F::sh(cc[u])) & F::MASK;
                const bool skip = f - sat < F::FLAG - sat;                 // sat <= f < FLAG: unflagged and saturated
                ap[u] = skip ? sink : cw + F::word(cc[u]);
                av[u] = skip ? 0u : 1u << F::sh(cc[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { ap[u] = cw + F::word(cc[u]); av[u] = 1u << F::sh(cc[u]); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) h[u] = atomicAdd(ap[u], av[u]);
        uint32_t any = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            h[u] = (h[u] >> F::sh(cc[u])) & F::MASK;                       // a skipped add returned the sink's zero
            any |= h[u];
            if (!MINI) guard |= h[u] & (h[u] << 1);
        }
        if (any & F::FLAG) {                                               // rare: a cell with ordered events
#pragma unroll
            for (int u = 0; u < 4; ++u) if (h[u] & F::FLAG) hmask |= 1u << (j + u - jlo);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((h[u] & F::FLAG) && near_ok && n - 1 - (j + u) == 1) {   // hybridmap.py:139-142: the NEARBY event
                    const uint32_t f2 = (atomicAdd(cw + F::word(cc[u]), 1u << F::sh(cc[u])) >> F::sh(cc[u])) & F::MASK;
                    if (!MINI) guard |= f2 & (f2 << 1);
                }
        }
    }
    if (j <= jhi) {                                                        // one to three steps left: one masked group
        int cc[3]; uint32_t h[3]; uint32_t* ap[3]; uint32_t av[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            cc[u] = cm + __mul24((int)(facc >> FIX_SHIFT), dminc);
            cm += dmajc; facc += fstep;
            if (j + u > jhi) cc[u] = cc[0];                                // dead step: a valid address, nothing added
        }
        if (SAT) {
#pragma unroll
            for (int u = 0; u < 3; ++u) h[u] = cw[F::word(cc[u])];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            bool skip = j + u > jhi;
            if (SAT) { const uint32_t f = (h[u] >> F::sh(cc[u])) & F::MASK; skip = skip || (f - sat < F::FLAG - sat); }
            ap[u] = skip ? sink : cw + F::word(cc[u]);
            av[u] = skip ? 0u : 1u << F::sh(cc[u]);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) h[u] = atomicAdd(ap[u], av[u]);
        uint32_t any = 0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            h[u] = av[u] ? (h[u] >> F::sh(cc[u])) & F::MASK : 0u;
            any |= h[u];
            if (!MINI) guard |= h[u] & (h[u] << 1);
        }
        if (any & F::FLAG) {
#pragma unroll
            for (int u = 0; u < 3; ++u) if (h[u] & F::FLAG) hmask |= 1u << (j + u - jlo);
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if ((h[u] & F::FLAG) && near_ok && n - 1 - (j + u) == 1) {
                    const uint32_t f2 = (atomicAdd(cw + F::word(cc[u]), 1u << F::sh(cc[u])) >> F::sh(cc[u])) & F::MASK;
                    if (!MINI) guard |= f2 & (f2 << 1);
                }
        }
    }
}

__global__ __launch_bounds__(FB) void map_update_fan_kernel(DevView v) {
    extern __shared__ __align__(16) unsigned char smem[];
    const FanGeom G = fan_geom(v.B, v.reach);
    uint32_t* const cnt = reinterpret_cast<uint32_t*>(smem + G.o_cnt);     // [ncell] 8-bit fields
    uint8_t*  const cnt8 = smem + G.o_cnt;
    uint16_t* const fpre = reinterpret_cast<uint16_t*>(smem + G.o_fpre);   // [nblk] directory: ids of block k = [fpre[k-1], fpre[k])
    uint32_t* const fpre32 = reinterpret_cast<uint32_t*>(smem + G.o_fpre);
    uint32_t* const mini = reinterpret_cast<uint32_t*>(smem + G.o_mini);   // [MINI_W^2] 16-bit fields
    int32_t*  const r_end = reinterpret_cast<int32_t*>(smem + G.o_rend);   // [B] packed end cell relative to the start
    uint8_t*  const r_info = smem + G.o_rinfo;                             // [B]
    uint16_t* const perm = reinterpret_cast<uint16_t*>(smem + G.o_perm);   // rays ordered by falling chunk count
    uint16_t* const ux = reinterpret_cast<uint16_t*>(smem + G.o_ux);       // U of global column fxl + i
    uint16_t* const uy = reinterpret_cast<uint16_t*>(smem + G.o_uy);
    uint8_t*  const gpx = smem + G.o_gpx;                                  // index-map irregularities before column i
    uint8_t*  const gpy = smem + G.o_gpy;
    uint32_t* const bcell = reinterpret_cast<uint32_t*>(smem + G.o_bcell); // [nb] wx | wy << 16 of flagged cell id
    uint8_t*  const msz = smem + G.o_floff;                                // [nb] events of flagged cell id (saturated at 255)
    uint8_t*  const oldv = smem + G.o_oldv;                                // [nb] value before the scan, then the replayed value
    uint32_t* const off32 = reinterpret_cast<uint32_t*>(smem + G.o_off);   // [nb] 16-bit bucket offsets (fill pointers)
    uint16_t* const off16 = reinterpret_cast<uint16_t*>(smem + G.o_off);
    uint16_t* const bigc = reinterpret_cast<uint16_t*>(smem + G.o_lists);  // [FLIST] ids with 17..64 events: one wave each
    uint16_t* const slowc = bigc + FLIST;
    uint32_t* const mark = reinterpret_cast<uint32_t*>(smem + G.o_mark);   // [FMARK] walk item | steps that met a flagged cell << 16 (early steps only)
    uint32_t* const rmask = reinterpret_cast<uint32_t*>(smem + G.o_rmask); // [B] bit i: step n - 32 + i of the ray met a flagged cell                                  // [FLIST] more than 64: exact membership scan
    uint32_t* const dummy = reinterpret_cast<uint32_t*>(smem + G.o_dummy); // per-lane sink for skipped adds
    uint16_t* const bev = reinterpret_cast<uint16_t*>(smem + G.o_bev);     // [FEV] (beam << 3) | rank

    __shared__ int s_fb;
    __shared__ int s_need[49], s_tab[49];
    __shared__ int s_fan[4];
    __shared__ int s_cntc[MAXLEV + 1], s_fill[MAXLEV + 1], s_lp[MAXLEV + 1], s_nk[MAXLEV + 1];
    __shared__ int s_wsum[FB / 64];
    __shared__ int s_nflag, s_nmark, s_nbig, s_nslow, s_ev, s_written;
    __shared__ int s_wq, s_pq[8];           // work queues: waves fetch 64 items at a time (walk; write-back per tile)
    __shared__ unsigned long long s_cells;

    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LL = v.L * v.L;
    const int KW = (v.dim + WIN - 1) / WIN;
    int32_t* tab = v.tile_tab + (size_t)v.slot[p] * LL;

#ifdef RBPF_STAMPS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = clock64();
#endif
    // =============================================== setup ===============================================
    // Every thread derives the pose quantities itself (same inputs, same arithmetic): no broadcast, no barrier.
    // this thread's first two beams: loaded before anything else so that the latency overlaps the pose arithmetic
    const int pb0 = tid, pb1 = tid + FB;
    const double pre_x0 = pb0 < v.B ? v.bx[pb0] : 0.0, pre_y0 = pb0 < v.B ? v.by[pb0] : 0.0;
    const double pre_x1 = pb1 < v.B ? v.bx[pb1] : 0.0, pre_y1 = pb1 < v.B ? v.by[pb1] : 0.0;
    const int pre_f0 = pb0 < v.B ? v.bflags[pb0] : 0, pre_f1 = pb1 < v.B ? v.bflags[pb1] : 0;
    const double s_px = v.upd_pose[p], s_py = v.upd_pose[v.P + p];
    double s_s, s_c;
    sincos(v.upd_pose[2 * v.P + p], &s_s, &s_c);
    const int x0 = UNI(trunc_to_int(s_px / v.cs)), y0 = UNI(trunc_to_int(s_py / v.cs));   // hybridmap.py:102
    {
        int lx, ly;                                                          // hybridmap.py:98-100
        bool ok = tile_of_coord(s_px, v.tile_len, v.R, lx) && tile_of_coord(s_py, v.tile_len, v.R, ly);
        if (ok) ok = tab[(lx + v.R) * v.L + (ly + v.R)] >= 0;
        const bool in_lut = lut_valid_g(v, x0 - v.reach) && lut_valid_g(v, x0 + v.reach) &&
                            lut_valid_g(v, y0 - v.reach) && lut_valid_g(v, y0 + v.reach);
        if (ok && !in_lut) { if (tid == 0) atomicCAS(v.err, 0, RBPF_ERANGE); ok = false; }
        if (tid == 0) v.mu_fallback[p] = 0;
        if (!UNI(ok)) return;
    }
    // Fans that do not fit the window are found only after the whole setup.  When most particles of the previous launch
    // ended that way (small cells, long rays), this launch leaves its particles to the next kernel at once; every 16th
    // particle still tries, so that the decision follows the data.
    {
        int32_t* const cur = v.mu_hint + 2 * (v.mu_step % 3);
        const int32_t* const prev = v.mu_hint + 2 * ((v.mu_step + 2) % 3);
        if (p == 0 && tid < 2) v.mu_hint[2 * ((v.mu_step + 1) % 3) + tid] = 0;
        const bool skip = prev[0] >= 8 && 2 * prev[1] > prev[0] && (p & 15) != 0;
        if (UNI(skip)) { GIVE_BACK(1); }
        if (tid == 0) atomicAdd(&cur[0], 1);
    }
    // the index map over everything a ray can reach: U of global column fxl + i (fyl + i), i <= 2 * reach
    const int fxl = x0 - v.reach, fyl = y0 - v.reach, nfx = 2 * v.reach + 1, nfy = nfx;
    for (int i = tid; i < G.fanw; i += FB) {
        const int gxq = fxl + i, gyq = fyl + i;
        uint32_t ex = lut_valid_g(v, gxq) ? lut_at(v, gxq) : LUT_INVALID, ey = lut_valid_g(v, gyq) ? lut_at(v, gyq) : LUT_INVALID;
        ux[i] = ex != LUT_INVALID ? (uint16_t)(lut_lat(ex) * v.dim + lut_cidx(ex)) : 0xFFFFu;
        uy[i] = ey != LUT_INVALID ? (uint16_t)(lut_lat(ey) * v.dim + lut_cidx(ey)) : 0xFFFFu;
    }
    if (tid == 0) {
        s_fan[0] = x0; s_fan[1] = x0; s_fan[2] = y0; s_fan[3] = y0;
        s_cells = 0; s_fb = 0; s_written = 0;
        s_nflag = 0; s_nmark = 0; s_nbig = 0; s_nslow = 0; s_ev = 0; s_wq = 0;
        for (int i = 0; i < 8; ++i) s_pq[i] = 0;
    }
    if (tid <= MAXLEV) { s_cntc[tid] = 0; s_fill[tid] = 0; }
    for (int i = tid; i < LL; i += FB) { s_need[i] = 0; s_tab[i] = tab[i]; }
    {   // clear the counters (independent of everything above: overlaps the LUT reads)
        uint4* c4 = reinterpret_cast<uint4*>(smem + G.o_cnt);
        const int n16 = (G.o_mini - G.o_cnt) >> 4;                           // counters and directory are adjacent
        for (int i = tid; i < n16; i += FB) c4[i] = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < (MINI_W * MINI_W + 1) / 2; i += FB) mini[i] = 0;
        if (tid < 64) dummy[tid] = 0;
        for (int i = tid; i < G.bpad; i += FB) rmask[i] = 0;
    }
    __syncthreads();
    STAMP(7);

    // lattice coordinate (biased) of a column of the LUT: rays are shorter than a tile, so it is the start tile's or a neighbour's
    const int Uxs = UNI(ux[x0 - fxl]), Uys = UNI(uy[y0 - fyl]);
    const int a0 = Uxs / v.dim, b0 = Uys / v.dim;
    auto lat_x = [&](int g) { const int U = ux[g - fxl]; return a0 + (U >= (a0 + 1) * v.dim ? 1 : 0) - (U < a0 * v.dim ? 1 : 0); };
    auto lat_y = [&](int g) { const int U = uy[g - fyl]; return b0 + (U >= (b0 + 1) * v.dim ? 1 : 0) - (U < b0 * v.dim ? 1 : 0); };
    {
        unsigned long long my_cells = 0;
        int fx0 = x0, fx1 = x0, fy0 = y0, fy1 = y0;
        for (int b0_ = 0; b0_ < v.B; b0_ += FB) {                                   // wave-uniform trip count (wave_keyed_inc)
            const int b = b0_ + tid;
            int nch_b = 0;
            if (b < v.B) {
            const double x = b == pb0 ? pre_x0 : b == pb1 ? pre_x1 : v.bx[b], y = b == pb0 ? pre_y0 : b == pb1 ? pre_y1 : v.by[b];
            const int bf = b == pb0 ? pre_f0 : b == pb1 ? pre_f1 : (int)v.bflags[b];
            double gx = (s_c * x + (-s_s) * y) + s_px;                             // lidar.py:123
            double gy = (s_s * x + s_c * y) + s_py;
            int x1 = trunc_to_int(gx / v.cs), y1 = trunc_to_int(gy / v.cs);        // hybridmap.py:106
            if (bf & BF_LONG) {                                                    // hybridmap.py:107-113
                double sc = v.bscale[b];
                x1 = trunc_to_int((double)x0 + sc * (double)(x1 - x0));
                y1 = trunc_to_int((double)y0 + sc * (double)(y1 - y0));
            }
            int ddx = x1 - x0, ddy = y1 - y0;
            if (ddx < -v.reach || ddx > v.reach || ddy < -v.reach || ddy > v.reach) {
                atomicCAS(v.err, 0, RBPF_ERANGE);
                ddx = 0; ddy = -1; x1 = x0; y1 = y0 - 1;                           // degenerate: no points
            }
            r_end[b] = (int32_t)(((uint32_t)ddx & 0xFFFFu) | ((uint32_t)ddy << 16));
            Ray r = ray_make(x0, y0, x1, y1);
            int info = 0;
            if (r.n > 0) {
                info = RI_VALID | ((bf & BF_LONG) ? 0 : RI_OCC);
                my_cells += (unsigned long long)r.n;
                fx0 = min(fx0, x1); fx1 = max(fx1, x1); fy0 = min(fy0, y1); fy1 = max(fy1, y1);
                const int a1 = lat_x(x1), b1 = lat_y(y1);
                if (r.n >= 2 && (info & RI_OCC)) {                                 // hybridmap.py:139-142
                    int nx, ny;
                    ray_point(r, r.n - 2, nx, ny);
                    if (lat_x(nx) == a1 && lat_y(ny) == b1) info |= RI_NEAR;      // hybridmap.py:141 same tile as the end cell
                    info |= ((nx - x1 + 1) & 3) << 3;
                    info |= ((ny - y1 + 1) & 3) << 5;
                }
                // tiles entered by this ray (staircase start -> [corner] -> end)
                s_need[a0 * v.L + b0] = 1;
                if (a1 != a0 || b1 != b0) {
                    s_need[a1 * v.L + b1] = 1;
                    if (a1 != a0 && b1 != b0) {
                        int gxb = r.sx > 0 ? v.gwin[a1 * (KW + 1)] : v.gwin[a0 * (KW + 1)] - 1;
                        int gyb = r.sy > 0 ? v.gwin[b1 * (KW + 1)] : v.gwin[b0 * (KW + 1)] - 1;
                        int ox = gxb - x0; ox = ox < 0 ? -ox : ox;
                        int oy = gyb - y0; oy = oy < 0 ? -oy : oy;
                        int jx = r.steep ? first_j_minor_ge(r, ox) : ox;
                        int jy = r.steep ? oy : first_j_minor_ge(r, oy);
                        if (jx < jy) s_need[a1 * v.L + b0] = 1;
                        else if (jy < jx) s_need[a0 * v.L + b1] = 1;
                    }
                }
            }
            r_info[b] = (uint8_t)info;
            nch_b = (r.n + CHUNK - 1) / CHUNK;
            }
            wave_keyed_inc(s_cntc, nch_b, nch_b > 0, lane);
        }
        {
            const int ws = wave_sum((int)my_cells);
            fx0 = wave_min(fx0); fx1 = wave_max(fx1); fy0 = wave_min(fy0); fy1 = wave_max(fy1);
            if (lane == 0) {
                atomicAdd(&s_cells, (unsigned long long)ws);
                atomicMin(&s_fan[0], fx0); atomicMax(&s_fan[1], fx1);
                atomicMin(&s_fan[2], fy0); atomicMax(&s_fan[3], fy1);
            }
        }
    }
    // irregular steps of the index map (U(g+1) - U(g) != 1) before each column: the last two waves, x and y
    if (wave >= FB / 64 - 2) {
        const bool isy = wave == FB / 64 - 1;
        const uint16_t* uu = isy ? uy : ux;
        uint8_t* gp = isy ? gpy : gpx;
        const int per = (G.fanw + 63) / 64, i0 = lane * per;
        int loc = 0;
        for (int i = i0; i < i0 + per; ++i) if (i + 1 < nfx && (int)uu[i + 1] - (int)uu[i] != 1) ++loc;
        int run = wave_excl_scan(loc, lane);
        for (int i = i0; i < i0 + per && i < G.fanw; ++i) {
            gp[i] = (uint8_t)run;
            if (run > 255) s_fb = 1;                                         // cannot be told apart in 8 bits
            if (i + 1 < nfx && (int)uu[i + 1] - (int)uu[i] != 1) ++run;
        }
    }
    __syncthreads();
    // ---- the window: the fan's bounding box in U coordinates ----
    const int bxl = UNI(s_fan[0]), bxh = UNI(s_fan[1]), byl = UNI(s_fan[2]), byh = UNI(s_fan[3]);
    const int Ux0 = UNI(ux[bxl - fxl]), Uy0al = UNI(uy[byl - fyl] & ~3);
    const int rows_u = UNI(ux[bxh - fxl]) - Ux0 + 1, cols_u = UNI(uy[byh - fyl]) - Uy0al + 1;
    const int stride = (cols_u + 3) & ~3;                                      // this particle's window: rows_u x stride cells
    const int wxc = Uxs - Ux0, wyc = Uys - Uy0al;                              // the start cell in window coordinates
    const int total_irreg = UNI((int)gpx[bxh - fxl] - (int)gpx[bxl - fxl] + (int)gpy[byh - fyl] - (int)gpy[byl - fyl]);
    if (rows_u * stride > G.ncell || rows_u < 1 || cols_u < 1 || UNI(s_fb)) { if (tid == 0) atomicAdd(&v.mu_hint[2 * (v.mu_step % 3) + 1], 1); GIVE_BACK(1); }
    if (tid == 0) {   // level table: a level's items are padded to whole waves; lane l of a wave takes ray
                      // l * (waves of the level) + wave, so that the lanes of one LDS instruction touch cells far apart
        int cntc[MAXLEV + 1];
        for (int c = 0; c <= MAXLEV; ++c) cntc[c] = s_cntc[c];
        int nk = 0;
        for (int c = 1; c <= MAXLEV; ++c) nk += cntc[c];                     // nk = rays with more than k chunks
        int lp = 0;
        for (int k = 0; k < MAXLEV; ++k) { s_lp[k] = lp; s_nk[k] = nk; lp += (nk + 63) & ~63; nk -= cntc[k + 1]; }
        s_lp[MAXLEV] = lp;
    }
    if (tid < LL && s_need[tid] && s_tab[tid] < 0) {                         // allocate missing tiles (kept zero-filled)
        int idx = atomicSub(v.free_top, 1) - 1;
        if (idx < 0) {
            atomicAdd(v.free_top, 1);
            atomicCAS(v.err, 0, RBPF_ENOMEM);
            s_need[tid] = 0;
        } else {
            int t = v.free_stack[idx];
            s_tab[tid] = t;
            tab[tid] = t;
            v.tile_bbox[4 * t + 0] = INT_MAX; v.tile_bbox[4 * t + 1] = -1;
            v.tile_bbox[4 * t + 2] = INT_MAX; v.tile_bbox[4 * t + 3] = -1;
        }
    }
    // rays ordered by falling chunk count: the rays that own a k-th chunk are perm[0 .. N_k)
    int longer_than;                                   // lane c <= MAXLEV: rays with more than c chunks (suffix sum over the wave)
    {
        const int cl = lane <= MAXLEV ? s_cntc[lane] : 0;
        int suf = cl;
        for (int off = 1; off < 32; off <<= 1) { const int t = __shfl_down(suf, off, 64); suf += t; }   // lanes > MAXLEV hold 0
        longer_than = suf - cl;
    }
    for (int b0_ = 0; b0_ < v.B; b0_ += FB) {
        const int b = b0_ + tid;
        int nch = 0;
        if (b < v.B && (r_info[b] & RI_VALID)) {
            int x1, y1;
            unpack_end(r_end[b], x0, y0, x1, y1);
            const Ray r = ray_make(x0, y0, x1, y1);
            nch = (r.n + CHUNK - 1) / CHUNK;
        }
        const int pos = wave_keyed_inc(s_fill, nch, nch > 0, lane);
        const int start = __shfl(longer_than, nch, 64);                            // rays with more chunks come first
        if (nch > 0) perm[start + pos] = (uint16_t)b;
    }

    // window coordinates of a global cell of the fan / field helpers
    auto in_mini = [&](int wx, int wy) { return (unsigned)(wx - wxc + MINI_R) < (unsigned)MINI_W && (unsigned)(wy - wyc + MINI_R) < (unsigned)MINI_W; };
    auto mini_idx = [&](int wx, int wy) { return (wx - wxc + MINI_R) * MINI_W + (wy - wyc + MINI_R); };
    // ids of a directory block are consecutive; once the counts are consumed a flagged cell's field holds
    // 0x80 | (id - first id of its block), so the lookup is two LDS reads
    auto blk_first = [&](int blk) { return blk ? (int)fpre[blk - 1] : 0; };
    auto cell_id = [&](int c) { return blk_first(c / BLK) + (int)(cnt8[c] & 0x7Fu); };
    auto count_of = [&](int id) {                                           // events of flagged cell id
        const uint32_t e = bcell[id];
        const int wx = (int)(e & 0xFFFFu), wy = (int)(e >> 16);
        if (in_mini(wx, wy)) { const int mi = mini_idx(wx, wy); return (int)((mini[mi >> 1] >> ((mi & 1) * 16)) & 0x7FFFu); }
        return (int)(cnt8[wx * stride + wy] & 0x7Fu);
    };

    STAMP(0);
    // ---- phase 1: flag the cells that receive an "occupied" or "nearby" hit (hybridmap.py:113,137,139-142) and build
    //      the directory of flagged cells: per 128-cell block a list of cell offsets; a cell's id = its list position ----
    auto flag_cell = [&](int b, int e, int info, int& wx, int& wy) {        // e = 0: end cell, e = 1: nearby cell
        int x1, y1;
        unpack_end(r_end[b], x0, y0, x1, y1);
        if (e) { x1 += ((info >> 3) & 3) - 1; y1 += ((info >> 5) & 3) - 1; }
        wx = (int)ux[x1 - fxl] - Ux0; wy = (int)uy[y1 - fyl] - Uy0al;
    };
    uint32_t first = 0;                                                      // bit 2*i + e: this thread flagged that cell first
    {
        int it = 0;
        for (int b = tid; b < v.B; b += FB, ++it) {
            const int info = r_info[b];
            if ((info & (RI_VALID | RI_OCC)) != (RI_VALID | RI_OCC)) continue;
            for (int e = 0; e < 2; ++e) {
                if (e == 1 && !(info & RI_NEAR)) break;
                int wx, wy;
                flag_cell(b, e, info, wx, wy);
                const int c = wx * stride + wy;
                uint32_t old;
                if (in_mini(wx, wy)) { const int mi = mini_idx(wx, wy), sh = (mi & 1) * 16; old = (atomicOr(&mini[mi >> 1], 0x8000u << sh) >> sh) & 0x8000u; }
                else { const int sh = (c & 3) * 8; old = (atomicOr(&cnt[c >> 2], 0x80u << sh) >> sh) & 0x80u; }
                if (!old) {
                    first |= 1u << (2 * it + e);
                    const int blk = c / BLK;
                    atomicAdd(&fpre32[blk >> 1], 1u << ((blk & 1) * 16));
                }
            }
        }
    }
    BAR_LDS();
    {
        const int pc = tid < G.nblk ? (int)fpre[tid] : 0;                        // flagged cells of block tid
        int incl = pc;
        for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
        if (lane == 63) s_wsum[wave] = incl;
        BAR_LDS();
        int wbase = 0;
        for (int k = 0; k < wave; ++k) wbase += s_wsum[k];
        const int excl = wbase + incl - pc;
        if (tid < G.nblk) fpre[tid] = (uint16_t)excl;                            // fill pointer: ends at the block's end
        if (tid == FB - 1) { s_nflag = excl + pc; if (excl + pc > G.nb) s_fb = 1; }
    }
    BAR_LDS();
    {
        int it = 0;
        for (int b = tid; b < v.B; b += FB, ++it) {
            if (!((first >> (2 * it)) & 3u)) continue;
            const int info = r_info[b];
            for (int e = 0; e < 2; ++e) {
                if (!((first >> (2 * it + e)) & 1u)) continue;
                int wx, wy;
                flag_cell(b, e, info, wx, wy);
                const int c = wx * stride + wy, blk = c / BLK, sh = (blk & 1) * 16;
                const int id = (int)((atomicAdd(&fpre32[blk >> 1], 1u << sh) >> sh) & 0xFFFFu);
                if (id < G.nb) bcell[id] = (uint32_t)wx | ((uint32_t)wy << 16);
            }
        }
    }
    BAR_LDS();
    const int nflag = UNI(s_nflag);
    const uint32_t sat = (uint32_t)((v.cc.vmax - v.cc.vmin + (-v.cc.emp) - 1) / (-v.cc.emp));   // hits that saturate any cell: 20

    STAMP(1);
    // ---- phase 2: walk the rays, CHUNK steps per work item, four steps in flight --------------------------------
    {
        int lpk[MAXLEV];
#pragma unroll
        for (int k = 0; k < MAXLEV; ++k) lpk[k] = UNI(s_lp[k]);
        const int nitems = UNI(s_lp[MAXLEV]);
        for (;;) {                                                                 // a wave takes the next 64 items: waves that
            int qw = 0;                                                            // got cheap items take more
            if (lane == 0) qw = atomicAdd(&s_wq, 64);
            qw = UNI(qw);                                                          // levels are whole waves: every lane shares k
            if (qw >= nitems) break;
            const int q = qw + lane;
            int k = 0, base = 0, nxt = lpk[1];
#pragma unroll
            for (int kk = 1; kk < MAXLEV; ++kk) if (qw >= lpk[kk]) { k = kk; base = lpk[kk]; nxt = kk + 1 < MAXLEV ? lpk[kk + 1] : nitems; }
            const int ii = lane * ((nxt - base) >> 6) + ((qw - base) >> 6);
            if (ii >= UNI(s_nk[k])) continue;                                      // padding
            const int b = perm[ii];
            const int info = r_info[b];
            int x1, y1;
            unpack_end(r_end[b], x0, y0, x1, y1);
            const Ray r = ray_make(x0, y0, x1, y1);
            const int jlo = k * CHUNK, jhi = min(r.n - 1, jlo + CHUNK - 1);
            const int smaj = r.steep ? r.sy : r.sx, smin = r.steep ? r.sx : r.sy;
            const bool near_ok = info & RI_NEAR;
            const uint32_t fstep = fix_slope(r.dmin, r.dmaj), facc = fstep * (uint32_t)jlo + (1u << (FIX_SHIFT - 1));
            int m = (int)(facc >> FIX_SHIFT);                                      // minor offset of step jlo
            // is the index map the identity between the start cell and every cell of this chunk?
            bool ident = total_irreg == 0;
            if (!ident) {
                const int mfar = min(m + (jhi - jlo), r.dmin);
                const int ex = r.steep ? r.sx * mfar : r.sx * jhi, ey = r.steep ? r.sy * jhi : r.sy * mfar;
                const int xl = x0 + min(ex, 0) - fxl, xh = x0 + max(ex, 0) - fxl, yl = y0 + min(ey, 0) - fyl, yh = y0 + max(ey, 0) - fyl;
                ident = gpx[xl] == gpx[xh] && gpy[yl] == gpy[yh];
            }
            uint32_t hit = 0;                                                      // bit u: step jlo + u met a flagged cell
            uint32_t acc = 0;
            if (ident) {
                if (k == 0) {
                    const int dmajc = smaj * (r.steep ? 1 : MINI_W), dminc = smin * (r.steep ? MINI_W : 1);
                    walk_ident<true, true>(mini, dummy + lane, MINI_R * MINI_W + MINI_R + jlo * dmajc, dmajc, dminc, facc, fstep,
                                           jlo, jhi, r.n, sat, near_ok, hit, acc);
                } else {
                    const int dmajc = smaj * (r.steep ? 1 : stride), dminc = smin * (r.steep ? stride : 1);
                    const int cm = wxc * stride + wyc + jlo * dmajc;
                    if (k < 2)                                                       // wave-uniform: a level is whole waves
                        walk_ident<false, true>(cnt, dummy + lane, cm, dmajc, dminc, facc, fstep, jlo, jhi, r.n, sat, near_ok, hit, acc);
                    else
                        walk_ident<false, false>(cnt, dummy + lane, cm, dmajc, dminc, facc, fstep, jlo, jhi, r.n, sat, near_ok, hit, acc);
                }
            } else {
                // general path: every step through the index map; the 16-bit block is chosen per cell
                int D = 2 * r.dmin - r.dmaj + 2 * r.dmin * jlo - 2 * r.dmaj * m;   // hybridmap.py:289-300 invariant
                const int m0 = r.steep ? y0 : x0, n0 = r.steep ? x0 : y0;
                for (int j4 = jlo; j4 <= jhi; j4 += 4) {
                    uint32_t* wp[4]; int sh[4]; uint32_t fm[4]; uint32_t h[4]; bool live[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j4 + u;
                        live[u] = j <= jhi;
                        const int maj = m0 + smaj * (live[u] ? j : jhi), mnr = n0 + smin * m;
                        const int gx = r.steep ? mnr : maj, gy = r.steep ? maj : mnr;
                        if (live[u]) { if (D >= 0) { ++m; D -= 2 * r.dmaj; } D += 2 * r.dmin; }
                        const int wx = (int)ux[gx - fxl] - Ux0, wy = (int)uy[gy - fyl] - Uy0al;
                        if (in_mini(wx, wy)) { const int mi = mini_idx(wx, wy); wp[u] = mini + (mi >> 1); sh[u] = (mi & 1) * 16; fm[u] = 0x8000u; }
                        else { const int c = wx * stride + wy; wp[u] = cnt + (c >> 2); sh[u] = (c & 3) * 8; fm[u] = 0x80u; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) h[u] = *wp[u];
                    uint32_t av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t f = (h[u] >> sh[u]) & (2 * fm[u] - 1);
                        const bool need = live[u] && ((f & fm[u]) || (f & (fm[u] - 1)) < sat);
                        if (!need) { wp[u] = dummy + lane; live[u] = false; }
                        av[u] = need ? 1u << sh[u] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) h[u] = atomicAdd(wp[u], av[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (!live[u]) continue;
                        const uint32_t f = (h[u] >> sh[u]) & (2 * fm[u] - 1);
                        if (fm[u] == 0x80u) acc |= f & (f << 1);
                        if (f & fm[u]) {
                            hit |= 1u << (j4 + u - jlo);
                            if (near_ok && r.n - 1 - (j4 + u) == 1) {
                                const uint32_t f2 = (atomicAdd(wp[u], 1u << sh[u]) >> sh[u]) & (2 * fm[u] - 1);
                                if (fm[u] == 0x80u) acc |= f2 & (f2 << 1);
                            }
                        }
                    }
                }
            }
            if (hit) {   // the steps that met a flagged cell: a bit per step of the ray's last 32 steps, a list entry for earlier ones
                const int rel = jlo - (r.n - 32);                                   // bit position of step jlo
                const uint32_t late = rel >= 0 ? hit << rel : hit >> min(-rel, 31);
                const uint32_t early = rel >= 0 ? 0u : hit & ((1u << min(-rel, 16)) - 1u);
                if (late) atomicOr(&rmask[b], late);
                if (early) {
                    const int pos = atomicAdd(&s_nmark, 1);
                    if (pos < FMARK) mark[pos] = (uint32_t)q | (early << 16); else s_fb = 1;
                }
            }
            if (acc & 0x40u) s_fb = 1;                                               // a field was seen at >= GUARD
        }
    }
    BAR_LDS();
    if (UNI(s_fb)) { GIVE_BACK(2); }

    STAMP(2);
    // ---- bucket offsets from the exact event counts; the cells' values before the scan --------------------------
    {
        int m[4], pc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int id = 4 * tid + k; m[k] = id < nflag ? count_of(id) : 0; pc += m[k]; }
        int incl = pc;
        for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
        if (lane == 63) s_wsum[wave] = incl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = 4 * tid + k;
            if (id >= nflag) break;
            const uint32_t e = bcell[id];
            const int Ux = (int)(e & 0xFFFFu) + Ux0, Uy = (int)(e >> 16) + Uy0al;
            const int a = Ux / v.dim, bb = Uy / v.dim;
            const int tile = s_need[a * v.L + bb] ? s_tab[a * v.L + bb] : -1;
            oldv[id] = tile >= 0 ? (uint8_t)v.pool[(size_t)tile * v.dim * v.dim + (size_t)(Ux - a * v.dim) * v.dim + (Uy - bb * v.dim)] : 0;
            msz[id] = (uint8_t)min(m[k], 255);
            const int c = (int)(e & 0xFFFFu) * stride + (int)(e >> 16);
            cnt8[c] = (uint8_t)(0x80u | (uint32_t)(id - blk_first(c / BLK)));      // the count is consumed: id lookup from now on
        }
        BAR_LDS();
        int wbase = 0;
        for (int k = 0; k < wave; ++k) wbase += s_wsum[k];
        int run = wbase + incl - pc;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = 4 * tid + k;
            if (id < nflag) off16[id] = (uint16_t)min(run, 0xFFFF);
            run += m[k];
        }
        if (tid == FB - 1) s_ev = run;
    }
    BAR_LDS();
    if (UNI(s_ev) > FEV) { GIVE_BACK(3); }
    // the 16-bit block's counts go into the main fields (saturated; flagged cells already hold their id)
    for (int mi = tid; mi < MINI_W * MINI_W; mi += FB) {
        const int wx = wxc + mi / MINI_W - MINI_R, wy = wyc + mi % MINI_W - MINI_R;
        if (wx < 0 || wx >= rows_u || wy < 0 || wy >= stride) continue;
        const uint32_t f = (mini[mi >> 1] >> ((mi & 1) * 16)) & 0xFFFFu;
        if (!(f & 0x8000u)) cnt8[wx * stride + wy] = (uint8_t)min(f, sat);
    }
    BAR_LDS();
    STAMP(3);
    // ---- re-walk the steps that met a flagged cell: every such hit becomes an event in its cell's bucket ----
    // steps: bit i = step jbase + i of beam b
    auto emit_events = [&](int b, int jbase, uint32_t steps) {
        const int info = r_info[b];
        int x1, y1;
        unpack_end(r_end[b], x0, y0, x1, y1);
        const Ray r = ray_make(x0, y0, x1, y1);
        const int jlo = jbase + __ffs(steps) - 1, jhi = jbase + 31 - __clz(steps);
        const int smaj = r.steep ? r.sy : r.sx, smin = r.steep ? r.sx : r.sy;
        const int m0 = r.steep ? y0 : x0, n0 = r.steep ? x0 : y0;
        const bool occ = info & RI_OCC, near_ok = info & RI_NEAR;
        int m = jlo ? ray_minor_at(r, jlo) : 0;
        int D = 2 * r.dmin - r.dmaj + 2 * r.dmin * jlo - 2 * r.dmaj * m;
        for (int j = jlo; j <= jhi; ++j) {
            const int maj = m0 + smaj * j, mnr = n0 + smin * m;
            const int gx = r.steep ? mnr : maj, gy = r.steep ? maj : mnr;
            if (D >= 0) { ++m; D -= 2 * r.dmaj; }
            D += 2 * r.dmin;
            if (!((steps >> (j - jbase)) & 1u)) continue;
            const int c = ((int)ux[gx - fxl] - Ux0) * stride + ((int)uy[gy - fyl] - Uy0al);
            if (!(cnt8[c] & 0x80u)) continue;
            const int id = cell_id(c);
            const int rem = r.n - 1 - j;
            const int rank = (rem == 0) ? (occ ? EV_OCC : EV_E_LAST) : rem == 1 ? EV_E_2 : rem == 2 ? EV_E_3 : EV_E_FAR;
            const int shf = (id & 1) * 16;
            int slot = (int)((atomicAdd(&off32[id >> 1], 1u << shf) >> shf) & 0xFFFFu);
            if (slot < FEV) bev[slot] = (uint16_t)((b << 3) | rank);
            if (near_ok && rem == 1) {
                slot = (int)((atomicAdd(&off32[id >> 1], 1u << shf) >> shf) & 0xFFFFu);
                if (slot < FEV) bev[slot] = (uint16_t)((b << 3) | EV_NEAR);
            }
        }
    };
    for (int b = tid; b < v.B; b += FB) {                                            // the last 32 steps of every ray
        const uint32_t steps = rmask[b];
        if (!steps) continue;
        int x1, y1;
        unpack_end(r_end[b], x0, y0, x1, y1);
        emit_events(b, ray_make(x0, y0, x1, y1).n - 32, steps);
    }
    const int nmark = UNI(s_nmark);
    for (int qm = tid; qm < nmark; qm += FB) {                                       // earlier steps (rays that graze a flagged cell)
        const uint32_t me = mark[qm];
        const int q = (int)(me & 0xFFFFu);                                           // item index of the walk: level, ray
        int k = 0, base = 0, nxt = s_lp[1];
        for (int kk = 1; kk < MAXLEV; ++kk) if (q >= s_lp[kk]) { k = kk; base = s_lp[kk]; nxt = s_lp[kk + 1]; }
        emit_events(perm[((q - base) & 63) * ((nxt - base) >> 6) + ((q - base) >> 6)], k * CHUNK, me >> 16);
    }
    BAR_LDS();

    STAMP(4);
    // ---- phase 3: flagged cells, ordered replay (bucket of cell id = bev[off[id] - m, off[id])) ------------------
    for (int id = tid; id < nflag; id += FB) {
        const int m = msz[id];
        if (m > 64) { const int pos = atomicAdd(&s_nslow, 1); if (pos < FLIST) slowc[pos] = (uint16_t)id; else s_fb = 1; continue; }
        if (m > 16) { const int pos = atomicAdd(&s_nbig, 1); if (pos < FLIST) bigc[pos] = (uint16_t)id; else s_fb = 1; continue; }
        int val = (int)(int8_t)oldv[id];
        const uint16_t* evp = bev + ((int)off16[id] - m);
        if (m <= 8) val = replay_sorted<8>(evp, m, val, v.cc);
        else val = replay_sorted<16>(evp, m, val, v.cc);
        const uint32_t e = bcell[id];
        cnt8[(int)(e & 0xFFFFu) * stride + (int)(e >> 16)] = (uint8_t)(0x80u | (uint32_t)(val - v.cc.vmin));   // the new value, for phase 4
    }
    BAR_LDS();
    if (UNI(s_fb)) { GIVE_BACK(4); }        // still nothing written to the map
    {
        const int nbig = UNI(s_nbig);
        for (int k = wave; k < nbig; k += FB / 64) {
            const int id = bigc[k];
            const int m = msz[id];                                                 // 17..64 events, one per lane
            const int start = (int)off16[id] - m;
            const uint32_t key = lane < m ? (uint32_t)bev[start + lane] : 0xFFFFFFFFu;
            int rank = 0;
            for (int e = 0; e < m; ++e) {
                const uint32_t ke = __shfl(key, e, 64);
                rank += (ke < key) || (ke == key && e < lane);
            }
            const uint32_t sorted = (uint32_t)__builtin_amdgcn_ds_permute((lane < m ? rank : lane) << 2, (int)key);
            const int BIG = 1000000;
            Caf f = {0, -BIG, BIG};
            if (lane < m) {
                const int rk = (int)(sorted & 7u);
                f = rk == EV_OCC ? Caf{v.cc.occ, -BIG, v.cc.vmax} : rk == EV_NEAR ? Caf{v.cc.nearby, -BIG, v.cc.vmax}
                                                                                : Caf{v.cc.emp, v.cc.vmin, BIG};
            }
            for (int off = 1; off < 64; off <<= 1) {
                Caf g;
                g.a = __shfl_down(f.a, off, 64); g.lo = __shfl_down(f.lo, off, 64); g.hi = __shfl_down(f.hi, off, 64);
                if ((lane & (2 * off - 1)) == 0) f = caf_then(f, g);
            }
            if (lane == 0) {
                const uint32_t e = bcell[id];
                cnt8[(int)(e & 0xFFFFu) * stride + (int)(e >> 16)] = (uint8_t)(0x80u | (uint32_t)(caf_apply(f, (int)(int8_t)oldv[id]) - v.cc.vmin));
            }
        }
        const int nslow = UNI(s_nslow);
        for (int k = wave; k < nslow; k += FB / 64) {
            const int id = slowc[k];
            const uint32_t e = bcell[id];
            const int Uxc = (int)(e & 0xFFFFu) + Ux0, Uyc = (int)(e >> 16) + Uy0al;
            int gxc[4], gyc[4], ngx = 0, ngy = 0;
            for (int i = 0; i < nfx && ngx < 4; ++i) if ((int)ux[i] == Uxc) gxc[ngx++] = fxl + i;
            for (int i = 0; i < nfy && ngy < 4; ++i) if ((int)uy[i] == Uyc) gyc[ngy++] = fyl + i;
            const int val = replay_cell_wave(v, r_info, r_end, x0, y0, gxc, ngx, gyc, ngy, (int)(int8_t)oldv[id], lane);
            if (lane == 0) cnt8[(int)(e & 0xFFFFu) * stride + (int)(e >> 16)] = (uint8_t)(0x80u | (uint32_t)(val - v.cc.vmin));
        }
        if (tid == 0 && nslow) atomicAdd(&v.stats[ST_SLOW_CELLS], (unsigned long long)nslow);
    }
    BAR_LDS();

    STAMP(5);
    // ---- phase 4: one read-modify-write per touched 32-cell group, tile by tile --------------------------------
    //      unflagged cell: v = max(v + n*emp, min) (gridmap.py:97-101, n times); flagged cell: the replayed value.
    //      The group's word of the tile's occupancy bitmask (cell > threshold, gridmap.py:153) is rebuilt.
    {
        int my_written = 0, combo = 0;
        const int eabs = -v.cc.emp;
        const uint32_t kb1 = (uint32_t)(128 + v.cc.vmin) * 0x01010101u;             // byte-wise: (cell ^ 0x80) - kb1 = cell - vmin
        const uint32_t satb = sat * 0x01010101u, sadd = (128u - sat) * 0x01010101u;
        const uint32_t oadd = (uint32_t)(127 - (v.cc.thr - v.cc.vmin)) * 0x01010101u; // bit 7 of (R + oadd) = cell > thr
        const int Ux1 = Ux0 + rows_u - 1, Uy1 = Uy0al + cols_u - 1;
        const int gpt = v.dim >> 5;                                                // 32-cell groups per tile row
        for (int a = Ux0 / v.dim; a <= Ux1 / v.dim; ++a)
        for (int bt = Uy0al / v.dim; bt <= Uy1 / v.dim; ++bt) {
            if (a >= v.L || bt >= v.L || !s_need[a * v.L + bt]) continue;          // uniform
            const int tile = UNI(s_tab[a * v.L + bt]);
            if (tile < 0) continue;
            const int wx_lo = max(0, a * v.dim - Ux0), wx_hi = min(rows_u - 1, (a + 1) * v.dim - 1 - Ux0);
            const int g_lo = max(Uy0al >> 5, bt * gpt), g_hi = min(Uy1 >> 5, (bt + 1) * gpt - 1);
            const int ngr = g_hi - g_lo + 1, items = (wx_hi - wx_lo + 1) * ngr;
            int8_t* __restrict__ tile_base = v.pool + (size_t)tile * v.dim * v.dim;
            int bx0 = INT_MAX, bx1 = -1, by0 = INT_MAX, by1 = -1;
            // an item in two halves, so that a lane can have the loads of two groups in flight before it computes either
            struct Item { uint32_t n[8]; uint32_t pre[8]; uint32_t* g_ptr; int row, col; bool live; };
            auto fetch = [&](int it, Item& I) {
                I.live = false;
                if (it >= items) return;
                const int rr = it / ngr, gg = it - rr * ngr;
                const int wx = wx_lo + rr, Gy = g_lo + gg;
                I.row = wx + Ux0 - a * v.dim; I.col = 32 * Gy - bt * v.dim;
                const int wy_first = 32 * Gy - Uy0al;                              // multiple of 4, may be negative
                uint32_t any = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const int wyw = wy_first + 4 * w;
                    I.n[w] = (wyw >= 0 && wyw < stride) ? cnt[(wx * stride + wyw) >> 2] : 0u;
                    any |= I.n[w];
                }
                if (!any) return;
                I.live = true;
                I.g_ptr = reinterpret_cast<uint32_t*>(tile_base + (size_t)I.row * v.dim + I.col);
                const uint4 q0 = reinterpret_cast<const uint4*>(I.g_ptr)[0], q1 = reinterpret_cast<const uint4*>(I.g_ptr)[1];
                I.pre[0] = q0.x; I.pre[1] = q0.y; I.pre[2] = q0.z; I.pre[3] = q0.w; I.pre[4] = q1.x; I.pre[5] = q1.y; I.pre[6] = q1.z; I.pre[7] = q1.w;
            };
            auto finish = [&](const Item& I) {
                if (!I.live) return;
                // Four cells per 32-bit word at a time (bytes never carry into each other: cells lie in [vmin, vmax],
                // hit counts below 96, sat * |emp| below 128): O = cell - vmin, R = max(O - |emp| * min(n, sat), 0).
                uint32_t occ = 0, touched = 0, out[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const uint32_t Ob = (I.pre[w] ^ 0x80808080u) - kb1;                     // cells biased to [0, vmax - vmin]
                    uint32_t R = Ob;
                    out[w] = I.pre[w];
                    if (I.n[w]) {
                        const uint32_t nw = I.n[w], n7 = nw & 0x7F7F7F7Fu;
                        const uint32_t ge = (n7 + sadd) & 0x80808080u;                      // fields >= sat
                        const uint32_t gem = ge | (ge - (ge >> 7));
                        const uint32_t m = (satb & gem) | (n7 & ~gem);                      // min(n, sat)
                        const uint32_t dec = eabs == 3 ? m + (m << 1) : m * (uint32_t)eabs;
                        const uint32_t T1 = (Ob | 0x80808080u) - dec;
                        const uint32_t pos = T1 & 0x80808080u;                              // O - dec >= 0
                        R = T1 & 0x7F7F7F7Fu & (pos | (pos - (pos >> 7)));
                        const uint32_t fl = nw & 0x80808080u;
                        if (fl) {                                                           // replayed cells: the field holds value - vmin
                            const uint32_t flm = fl | (fl - (fl >> 7));
                            R = (n7 & flm) | (R & ~flm);
                        }
                        out[w] = (R + kb1) ^ 0x80808080u;
                        const uint32_t nz = ((n7 + 0x7F7F7F7Fu) | nw) & 0x80808080u;        // fields that are not zero
                        touched |= __builtin_amdgcn_udot4(nz >> 7, 0x08040201u, 0u, false) << (4 * w);
                    }
                    occ |= __builtin_amdgcn_udot4(((R + oadd) & 0x80808080u) >> 7, 0x08040201u, 0u, false) << (4 * w);   // cell > thr
                }
                // the whole group goes back in two 16-byte stores (untouched words keep their value; this workgroup is
                // the tile's only writer)
                reinterpret_cast<uint4*>(I.g_ptr)[0] = make_uint4(out[0], out[1], out[2], out[3]);
                reinterpret_cast<uint4*>(I.g_ptr)[1] = make_uint4(out[4], out[5], out[6], out[7]);
                my_written += __popc(touched);
                by0 = min(by0, I.col + __ffs(touched) - 1); by1 = max(by1, I.col + 31 - __clz(touched));
                v.occ[((size_t)tile * v.dim + I.row) * v.ow + (I.col >> 5)] = occ;
                bx0 = min(bx0, I.row); bx1 = max(bx1, I.row);
            };
            // waves fetch 128 items at a time, two per lane: groups without a touched cell cost next to nothing, so a
            // static split would leave some waves with most of the work
            const int ci = combo++;                                                // uniform over the workgroup
            if (ci < 8) {
                while (true) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_pq[ci], 64);
                    base = UNI(base);
                    if (base >= items) break;
                    Item A;
                    fetch(base + lane, A);
                    finish(A);
                }
            } else {
                for (int it = tid; it < items; it += FB) { Item A; fetch(it, A); finish(A); }
            }

            bx0 = wave_min(bx0); bx1 = wave_max(bx1); by0 = wave_min(by0); by1 = wave_max(by1);
            if (lane == 0 && bx1 >= 0) {                                           // this workgroup is the tile's only writer
                atomicMin(&v.tile_bbox[4 * tile + 0], bx0); atomicMax(&v.tile_bbox[4 * tile + 1], bx1);
                atomicMin(&v.tile_bbox[4 * tile + 2], by0); atomicMax(&v.tile_bbox[4 * tile + 3], by1);
            }
        }
        const int ww = wave_sum(my_written);
        if (lane == 0 && ww) atomicAdd(&s_written, ww);
    }
    BAR_LDS();
    STAMP(6);
    if (tid == 0) {
        if (s_cells) atomicAdd(&v.stats[ST_RAY_CELLS], s_cells);
        if (s_written) atomicAdd(&v.stats[ST_CELLS_WRITTEN], (unsigned long long)s_written);
#ifdef RBPF_STAMPS
        for (int k = 0; k < 8; ++k) atomicAdd(&v.stats[8 + k], (unsigned long long)st_acc[k]);
#endif
    }
}

void launch_map_update_fan(const DevView& v, hipStream_t s) {
    const FanGeom g = fan_geom(v.B, v.reach);
    static size_t lds_set[MAX_DEVICES] = {};
    ensure_dynamic_lds(reinterpret_cast<const void*>(map_update_fan_kernel), (size_t)g.bytes, lds_set);
    hipLaunchKernelGGL(map_update_fan_kernel, dim3(v.P), dim3(FB), (size_t)g.bytes, s, v);
}

}  // namespace rbpf
