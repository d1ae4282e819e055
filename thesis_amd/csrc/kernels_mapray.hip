// kernels_mapray.hip -- a5: HybridMap.update (hybridmap.py:95-145), counters in GLOBAL cell-index space.
//
// One 1024-thread workgroup per particle (16 waves, one workgroup per CU).  Same exact semantics as the other two map
// kernels (kernels_mapupdate.hip has the ordered-replay argument); what differs:
//
//   * The LDS hit counters are indexed by the reference's global integer cell index (hybridmap.py:102,106,123), the
//     space its Bresenham walks in, not by storage index.  A ray step is then pure arithmetic - no index-map lookups,
//     no "irregular" path on the negative side of a tile where the reference's float index formula (gridmap.py:93,
//     SURVEY quirk 3) mis-rounds.  The index map is verified per particle to have the form U(g) = g + C - G(g),
//     G(g) in {0,1} (it can only round DOWN); the write-back folds it in: storage cell s receives global cell s - C
//     when that one is not glitched, plus global cell s - C + 1 when that one is.
//   * Walk: lanes are rays, a work item is a 16-step chunk of 64 rays, each step from the closed form
//     minor(j) = (fstep * j + 2^21) >> 22 (rbpf_math.h / fix_slope); the adds are fire-and-forget LDS atomics: nothing
//     waits for a returned value, nothing is detected.
//   * Cells that receive an "occupied" / "nearby" hit (the only ones whose clamped adds do not commute) are not found
//     by the walk.  The rays through a cell at major distance j, minor offset c are exactly those of the matching
//     direction class whose fixed-point slope lies in [ceil((c*2^22 - 2^21)/j), ceil(((c+1)*2^22 - 2^21)/j)) and that
//     are longer than j; rays are bucketed by (class, slope >> 14) and sorted once per particle, so each such cell
//     GATHERS its few rays from a contiguous run, sorts the (beam, rank) events in registers and replays them.  Cells
//     next to the sensor, which most rays cross, take their events from a second walk of the 16-bit block instead and
//     need only the number of unoccupied passes between consecutive occupied / nearby events.  This happens before
//     anything is written, so every table overflow still hands the particle back untouched.
//   * A fan larger than the LDS window is processed in strips of storage rows (one launch, any fan size): cell sizes
//     of 0.025 m and 15 m rays included.  8-bit counters cannot overflow: a bound on the hits of any cell with j >= 16
//     is checked from the slope buckets first (cells nearer than NEAR_R steps live in a 16-bit block).
#include <hip/hip_ext.h>

#include "rbpf_mapupdate.h"

namespace rbpf {

#ifdef RBPF_STAMPS
#define STAMP(k) do { if (tid == 0) { long long t_ = clock64(); st_acc[k] += t_ - st_prev; st_prev = t_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// hand the particle to the window kernel (uniform over the workgroup; nothing has been written to the map yet);
// reason codes: 1 geometry / index map, 2 counter bound, 3 event tables
#define GIVE_BACK(reason) do { if (tid == 0) { v.mu_fallback[p] = (reason); atomicAdd(&v.stats[(reason) == 1 ? ST_FALLBACK_REASONS : (reason) == 2 ? ST_FB_BOUND : ST_FB_TABLES], 1ull); } return; } while (0)

static const int RB = 1024;                    // threads per particle
static const int NEAR_R = 16;                  // ray steps j < NEAR_R are counted in the 16-bit block
static const int LCH = 16;                     // steps per chunk of the walk beyond it
static const int NEAR_W = 2 * NEAR_R + 1;      // cells with Chebyshev distance <= NEAR_R from the start cell (one spare ring)
static const int NBIN = 256;                   // slope buckets per direction class
static const int RFIX = 22;                    // fixed-point bits of the slope (fix_slope)
static const int BIN_SHIFT = RFIX - 8;
static const int RSLOW = 256;                  // flagged cells replayed by the wave-wide exact scan, per particle
static const int ECAP = 16;                    // events per flagged cell kept in its list (more: the wave-wide scan)
static const int NNEAR = 128;                  // flagged cells inside the 16-bit block that collect their events from the block's walk
static const int NSPC = 32;                    // occupied / nearby events of such a cell
static const int NPOOL = 48;                   // flagged cells with more than ECAP events: lists of PCAP events, sorted and folded by a wave
static const int PCAP = 64;
static const int RSPEC = 512;                  // flagged cells on an axis or a diagonal through the sensor, per particle
static const int MAXLEV = 63;                  // whole 16-step chunks per ray (reach < 1000 cells)
static const int NB_WIN = NBIN / NEAR_R + 2;   // slope buckets that can hold the rays through one cell beyond the 16-bit block
static const int HIT_BOUND = 63;               // per direction class; two classes can meet in one cell: 126 < 128

struct RayGeom {
    int fanw, bpad, ncell;
    int o_cnt, o_mini, o_rend, o_rinfo, o_fstep, o_ux, o_uy, o_gxb, o_gyb, o_gym, o_bins, o_brays, o_oval, o_slow, o_oldv, o_rcc, o_rdmaj, o_perm, o_rpos, o_pflag, o_nid, o_nearl;
    int bytes;
    bool ok;
};

__host__ __device__ inline int ray_al16(int x) { return (x + 15) & ~15; }

__host__ __device__ inline RayGeom ray_geom(int B, int reach) {
    RayGeom g;
    g.fanw = (2 * reach + 8 + 7) & ~7;
    g.bpad = (B + 3) & ~3;
    int o = 0;
    g.o_mini = o;  o += ray_al16(((NEAR_W * NEAR_W + 1) / 2) * 4);
    g.o_rend = o;  o += ray_al16(g.bpad * 4);
    g.o_fstep = o; o += ray_al16(g.bpad * 4);
    g.o_rinfo = o; o += ray_al16(g.bpad);
    g.o_ux = o;    o += ray_al16(g.fanw * 2);
    g.o_uy = o;    o += ray_al16(g.fanw * 2);
    g.o_gxb = o;   o += ray_al16(g.fanw);
    g.o_gyb = o;   o += ray_al16(g.fanw);
    g.o_gym = o;   o += ray_al16(g.fanw + 16);
    g.o_bins = o;  o += 8 * NBIN * 2;                  // 2048 packed 16-bit fill pointers = one 32-bit word per thread
    g.o_brays = o; o += ray_al16(g.bpad * 2);
    g.o_oval = o;  o += ray_al16(g.bpad * 2);
    g.o_slow = o;  o += RSLOW * 2;
    g.o_oldv = o;  o += ray_al16(g.bpad * 2);
    g.o_rcc = o;   o += ray_al16(g.bpad * 4);
    g.o_rdmaj = o; o += ray_al16(g.bpad * 2);
    g.o_perm = o;  o += ray_al16(g.bpad * 2);
    g.o_rpos = o;  o += ray_al16(g.bpad * 2);
    g.o_pflag = o; o += ray_al16(g.bpad * 2);
    g.o_nid = o;   o += ray_al16(NEAR_W * NEAR_W);
    g.o_nearl = o; o += NNEAR * 2;
    g.o_cnt = o;
    const int avail = 160 * 1024 - 2048 - o - 64;      // 2 KB for the kernel's static LDS
    g.ncell = avail > 0 ? avail & ~127 : 0;
    g.bytes = o + g.ncell;
    g.ok = g.ncell >= 24576 && B <= 4095 && reach >= NEAR_R + 4 && reach < 1000 && 2LL * reach * reach < (1LL << RFIX);
    return g;
}

// the flagged-cell pass borrows the counter window: 16-bit event counts, ECAP events per pair, the special list
__host__ __device__ inline bool ray_lists_fit(const RayGeom& g) {
    const int npair = 2 * g.bpad;
    return npair * 2 + npair * ECAP * 2 + RSPEC * 2 + g.bpad * 8 + npair * 2 + NPOOL * (PCAP * 2 + 4) + 64 <= g.ncell &&   // + the sorted ray records, the list of pairs in play, the long lists
           npair * 2 + npair * ECAP * 2 + RSPEC * 2 + (RB / 64) * 256 * 4 <= g.ncell;                                      // the bucket sort's key scratch in the records' place
}

bool map_update_ray_available(const DevView& v) {
    const int sat = (v.cc.vmax - v.cc.vmin + (-v.cc.emp) - 1) / (-v.cc.emp);
    const RayGeom g = ray_geom(v.B, v.reach);
    return g.ok && ray_lists_fit(g) && v.dim % 32 == 0 && v.L * v.L <= 49 && v.cc.emp < 0 && sat <= 31 && v.cc.vmax - v.cc.vmin <= 127 &&
           v.cc.vmin <= 0 && v.cc.vmax >= 0 && v.cc.vmin >= -127 && sat * -v.cc.emp <= 127 && v.cc.thr >= v.cc.vmin && v.cc.thr < v.cc.vmax;
}

__device__ __forceinline__ uint32_t ray_fix_slope(int dmin, int dmaj) {
    return dmaj ? (((uint32_t)dmin << RFIX) + (uint32_t)dmaj - 1u) / (uint32_t)dmaj : 0u;
}

// Fire-and-forget adds on hit fields, addressed by field index + the array's LDS address folded into the index (the
// addresses are multiples of 4, so the field's position inside its word is unchanged): the LDS instruction takes the
// masked index as its address, without a separate add of the array's base.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ int lds_addr(const void* p) { return (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p; }
__device__ __forceinline__ void fld8_add(int cabs, uint32_t one) {      // 8-bit fields; cabs = index + lds_addr(array); one = 1 or 0
    __hip_atomic_fetch_add((lds_u32*)(uintptr_t)(uint32_t)(cabs & ~3), one << ((cabs & 3) * 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void fld16_add(int cabs, uint32_t one) {     // 16-bit fields; cabs = index + lds_addr(array) / 2
    __hip_atomic_fetch_add((lds_u32*)(uintptr_t)(uint32_t)((cabs << 1) & ~3), one << ((cabs & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// byte-wise min(field, sat) of four unflagged 7-bit hit counts; flagged bytes (bit 7: the field holds a replayed value)
// pass through
__device__ __forceinline__ uint32_t premin4(uint32_t x, uint32_t satb, uint32_t sadd) {
    const uint32_t n7 = x & 0x7F7F7F7Fu;
    const uint32_t ge = (n7 + sadd) & 0x80808080u;
    const uint32_t gem = ge | (ge - (ge >> 7));
    const uint32_t m = (satb & gem) | (n7 & ~gem);
    const uint32_t fl = x & 0x80808080u;
    const uint32_t flm = fl | (fl - (fl >> 7));
    return (x & flm) | (m & ~flm);
}

__global__ __launch_bounds__(RB) void map_update_ray_kernel(DevView v, const int32_t* __restrict__ only) {
    if (only && !only[blockIdx.x]) return;                                 // the kernel that ran first did this particle
    extern __shared__ __align__(16) unsigned char smem[];
    const RayGeom G = ray_geom(v.B, v.reach);
    uint32_t* const cnt = reinterpret_cast<uint32_t*>(smem + G.o_cnt);     // 8-bit hit fields, [row = global x][col = global y]
    uint8_t*  const cnt8 = smem + G.o_cnt;
    uint32_t* const mini = reinterpret_cast<uint32_t*>(smem + G.o_mini);   // [NEAR_W^2] 16-bit fields around the start cell
    int32_t*  const r_end = reinterpret_cast<int32_t*>(smem + G.o_rend);   // [B] packed end cell relative to the start
    uint32_t* const r_fstep = reinterpret_cast<uint32_t*>(smem + G.o_fstep); // [B] fixed-point slope
    uint8_t*  const r_info = smem + G.o_rinfo;                             // [B]
    uint16_t* const ux = reinterpret_cast<uint16_t*>(smem + G.o_ux);       // U of global column fxl + i
    uint16_t* const uy = reinterpret_cast<uint16_t*>(smem + G.o_uy);
    uint8_t*  const gxb = smem + G.o_gxb;                                  // G of global column fxl + i (0 / 1)
    uint8_t*  const gyb = smem + G.o_gyb;
    uint8_t*  const gym = smem + G.o_gym;                                  // G of window column lc as a byte mask (0 / 0xFF)
    uint32_t* const bins32 = reinterpret_cast<uint32_t*>(smem + G.o_bins); // [8 * NBIN] 16-bit counts, then fill pointers
    uint16_t* const bins16 = reinterpret_cast<uint16_t*>(smem + G.o_bins);
    uint16_t* const brays = reinterpret_cast<uint16_t*>(smem + G.o_brays); // ray ids ordered by (class, slope bucket)
    uint8_t*  const oval = smem + G.o_oval;                                // [2 * B] replayed value - vmin of the cell flagged by (beam, e); 0xFF = none / not the owner
    uint16_t* const slowl = reinterpret_cast<uint16_t*>(smem + G.o_slow);  // [RSLOW] (beam << 1) | e
    uint8_t*  const oldv8 = smem + G.o_oldv;                               // [2 * B] its value before the scan
    uint32_t* const r_cc = reinterpret_cast<uint32_t*>(smem + G.o_rcc);    // [B] window-address steps of the ray: per major step | per minor step << 16
    uint16_t* const r_dmaj = reinterpret_cast<uint16_t*>(smem + G.o_rdmaj); // [B] last step of the ray
    uint16_t* const perm = reinterpret_cast<uint16_t*>(smem + G.o_perm);   // rays ordered by falling count of whole 16-step chunks
    uint16_t* const rpos = reinterpret_cast<uint16_t*>(smem + G.o_rpos);   // [B] position of the ray in brays
    uint8_t*  const nid = smem + G.o_nid;                                  // [NEAR_W^2] list position of the flagged cell a block cell is a source of, 0xFF = none
    uint16_t* const nearl = reinterpret_cast<uint16_t*>(smem + G.o_nearl); // [NNEAR] owner pairs of those cells
    uint8_t*  const pflag = smem + G.o_pflag;                              // [2 * B] 0 = pair not in play, 1 = gathered, 2 = other classes too

    __shared__ int s_fb;
    __shared__ int s_need[49], s_tab[49];
    __shared__ int s_fan[4];
    __shared__ int s_wsum[RB / 64];
    __shared__ int s_nslow, s_written, s_nspec, s_nact, s_exact, s_nnear, s_npool, s_wbq;
    __shared__ int s_lcnt[MAXLEV + 1], s_lfill[MAXLEV + 1], s_nk[MAXLEV + 2], s_lp[MAXLEV + 3], s_nlev;
    __shared__ unsigned long long s_cells;
    __shared__ double s_sincos[2];
    __shared__ uint8_t s_ggf[96];                     // 32-column storage group g (from T_lo >> 5): a glitched column among its 33

    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LL = v.L * v.L;
    const int KW = (v.dim + WIN - 1) / WIN;
    int32_t* tab = v.tile_tab + (size_t)v.slot[p] * LL;

#ifdef RBPF_STAMPS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = clock64();
#endif
    // =============================================== setup ===============================================
    const int pb0 = tid, pb1 = tid + RB;
    const double pre_x0 = pb0 < v.B ? v.bx[pb0] : 0.0, pre_y0 = pb0 < v.B ? v.by[pb0] : 0.0;
    const double pre_x1 = pb1 < v.B ? v.bx[pb1] : 0.0, pre_y1 = pb1 < v.B ? v.by[pb1] : 0.0;
    const int pre_f0 = pb0 < v.B ? v.bflags[pb0] : 0, pre_f1 = pb1 < v.B ? v.bflags[pb1] : 0;
    const double s_px = v.upd_pose[p], s_py = v.upd_pose[v.P + p];
    if (wave == 0) {   // one wave takes the sine and cosine (a few hundred instructions); the others read them after the first barrier
        double sn, cs_;
        sincos(v.upd_pose[2 * v.P + p], &sn, &cs_);
        if (lane == 0) { s_sincos[0] = sn; s_sincos[1] = cs_; }
    }
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
    // the index map over everything a ray can reach, with a margin of two columns (the sources of a storage cell are its
    // own global index and the next one)
    const int fxl = x0 - v.reach - 2, fyl = y0 - v.reach - 2, nfx = 2 * v.reach + 5;
    for (int i = tid; i < G.fanw; i += RB) {
        const int gxq = fxl + i, gyq = fyl + i;
        uint32_t ex = lut_valid_g(v, gxq) ? lut_at(v, gxq) : LUT_INVALID, ey = lut_valid_g(v, gyq) ? lut_at(v, gyq) : LUT_INVALID;
        ux[i] = ex != LUT_INVALID ? (uint16_t)(lut_lat(ex) * v.dim + lut_cidx(ex)) : 0xFFFFu;
        uy[i] = ey != LUT_INVALID ? (uint16_t)(lut_lat(ey) * v.dim + lut_cidx(ey)) : 0xFFFFu;
    }
    if (tid == 0) {
        s_fan[0] = x0; s_fan[1] = x0; s_fan[2] = y0; s_fan[3] = y0;
        s_cells = 0; s_fb = 0; s_written = 0; s_nslow = 0; s_nspec = 0; s_nact = 0; s_exact = 0; s_nnear = 0; s_npool = 0;
    }
    for (int i = tid; i < LL; i += RB) { s_need[i] = 0; s_tab[i] = tab[i]; }
    bins32[tid] = 0;
    if (tid < 96) s_ggf[tid] = 0;
    if (tid <= MAXLEV) s_lcnt[tid] = 0;
    for (int i = tid; i < (NEAR_W * NEAR_W + 1) / 2; i += RB) mini[i] = 0;
    for (int i = tid; i < (NEAR_W * NEAR_W + 3) / 4; i += RB) reinterpret_cast<uint32_t*>(nid)[i] = 0xFFFFFFFFu;
    for (int i = tid; i < (2 * G.bpad + 3) / 4; i += RB) reinterpret_cast<uint32_t*>(oval)[i] = 0xFFFFFFFFu;
    __syncthreads();
    STAMP(0);
    const double s_s = s_sincos[0], s_c = s_sincos[1];

    const int C = v.R * v.dim + v.dim / 2;
    const int Uxs = UNI(ux[x0 - fxl]), Uys = UNI(uy[y0 - fyl]);
    const int a0 = Uxs / v.dim, b0 = Uys / v.dim;
    auto lat_x = [&](int g) { const int U = ux[g - fxl]; return a0 + (U >= (a0 + 1) * v.dim ? 1 : 0) - (U < a0 * v.dim ? 1 : 0); };
    auto lat_y = [&](int g) { const int U = uy[g - fyl]; return b0 + (U >= (b0 + 1) * v.dim ? 1 : 0) - (U < b0 * v.dim ? 1 : 0); };
    // the storage cell (U_x << 16 | U_y) flagged by pair (beam, e): the beam's end cell (e = 0) or the cell before it (e = 1,
    // only when it lies in the end cell's tile); ~0 = none.  And the ray's window-address steps (per major / minor step).
    auto pair_cell = [&](int pr) -> uint32_t {
        const int b = pr >> 1, info = r_info[b];
        if ((info & (RI_VALID | RI_OCC)) != (RI_VALID | RI_OCC) || ((pr & 1) && !(info & RI_NEAR))) return 0xFFFFFFFFu;
        const int32_t re = r_end[b];
        int x1 = x0 + (int)(int16_t)(re & 0xFFFF), y1 = y0 + (int)(int16_t)((uint32_t)re >> 16);
        if (pr & 1) { x1 += ((info >> 3) & 3) - 1; y1 += ((info >> 5) & 3) - 1; }
        return ((uint32_t)ux[x1 - fxl] << 16) | (uint32_t)uy[y1 - fyl];
    };
    auto pair_gcell = [&](int pr) -> uint32_t {          // the same as a global cell relative to the start, biased: never ~0
        const int b = pr >> 1, info = r_info[b];
        if ((info & (RI_VALID | RI_OCC)) != (RI_VALID | RI_OCC) || ((pr & 1) && !(info & RI_NEAR))) return 0xFFFFFFFFu;
        const int32_t re = r_end[b];
        int dx = (int)(int16_t)(re & 0xFFFF), dy = (int)(int16_t)((uint32_t)re >> 16);
        if (pr & 1) { dx += ((info >> 3) & 3) - 1; dy += ((info >> 5) & 3) - 1; }
        return (uint32_t)(dx + 0x4000) | ((uint32_t)(dy + 0x4000) << 16);
    };
    // direction class and slope bucket of a ray
    auto ray_key = [&](int ddx, int ddy, uint32_t fstep) {
        const int adx = ddx < 0 ? -ddx : ddx, ady = ddy < 0 ? -ddy : ddy;
        const int cls = (ady > adx ? 4 : 0) | (ddx > 0 ? 2 : 0) | (ddy > 0 ? 1 : 0);
        const int bin = min((int)(fstep >> BIN_SHIFT), NBIN - 1);
        return cls * NBIN + bin;
    };
    {
        unsigned long long my_cells = 0;
        int fx0 = x0, fx1 = x0, fy0 = y0, fy1 = y0;
        for (int b = tid; b < v.B; b += RB) {
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
            uint32_t fstep = 0;
            if (r.n > 0) {
                info = RI_VALID | ((bf & BF_LONG) ? 0 : RI_OCC);
                my_cells += (unsigned long long)r.n;
                fx0 = min(fx0, x1); fx1 = max(fx1, x1); fy0 = min(fy0, y1); fy1 = max(fy1, y1);
                const int a1 = lat_x(x1), b1 = lat_y(y1);
                if (r.n >= 2 && (info & RI_OCC)) {                                 // hybridmap.py:139-142
                    int nx, ny;
                    ray_point(r, r.n - 2, nx, ny);
                    if (lat_x(nx) == a1 && lat_y(ny) == b1) info |= RI_NEAR;          // hybridmap.py:141 same tile as the end cell
                    info |= ((nx - x1 + 1) & 3) << 3;
                    info |= ((ny - y1 + 1) & 3) << 5;
                }
                // tiles entered by this ray (staircase start -> [corner] -> end)
                s_need[a0 * v.L + b0] = 1;
                if (a1 != a0 || b1 != b0) {
                    s_need[a1 * v.L + b1] = 1;
                    if (a1 != a0 && b1 != b0) {
                        int gxb_ = r.sx > 0 ? v.gwin[a1 * (KW + 1)] : v.gwin[a0 * (KW + 1)] - 1;
                        int gyb_ = r.sy > 0 ? v.gwin[b1 * (KW + 1)] : v.gwin[b0 * (KW + 1)] - 1;
                        int ox = gxb_ - x0; ox = ox < 0 ? -ox : ox;
                        int oy = gyb_ - y0; oy = oy < 0 ? -oy : oy;
                        int jx = r.steep ? first_j_minor_ge(r, ox) : ox;
                        int jy = r.steep ? oy : first_j_minor_ge(r, oy);
                        if (jx < jy) s_need[a1 * v.L + b0] = 1;
                        else if (jy < jx) s_need[a0 * v.L + b1] = 1;
                    }
                }
                fstep = ray_fix_slope(r.dmin, r.dmaj);
                const int key = ray_key(ddx, ddy, fstep);
                atomicAdd(&bins32[key >> 1], 1u << ((key & 1) * 16));
                const int nfull = (r.dmaj + 1 - NEAR_R) / LCH;                       // whole chunks beyond the 16-bit block
                if (nfull >= 1) atomicAdd(&s_lcnt[min(nfull, MAXLEV)], 1);
            }
            r_dmaj[b] = (uint16_t)(r.n > 0 ? r.dmaj : 0);
            r_info[b] = (uint8_t)info;
            r_fstep[b] = fstep;
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
    __syncthreads();
    STAMP(0);
    // ---- the window: the fan's bounding box in global cell indices; strips of storage rows if it does not fit ----
    const int bxl = UNI(s_fan[0]), bxh = UNI(s_fan[1]), byl = UNI(s_fan[2]), byh = UNI(s_fan[3]);
    // the reference's index formula over the fan (one column more on either side): U(g) = g + C - G(g) with G in {0, 1}
    for (int i = tid; i < G.fanw; i += RB) {
        const int dxg = (fxl + i + C) - (int)ux[i], dyg = (fyl + i + C) - (int)uy[i];
        if (fxl + i >= bxl - 1 && fxl + i <= bxh + 1 && (unsigned)dxg > 1u) s_fb = 1;   // also: the LUT ends inside the fan
        if (fyl + i >= byl - 1 && fyl + i <= byh + 1 && (unsigned)dyg > 1u) s_fb = 1;
        gxb[i] = (uint8_t)(dxg & 1); gyb[i] = (uint8_t)(dyg & 1);
    }
    const int S_lo = UNI(ux[bxl - fxl]), S_hi = UNI(ux[bxh - fxl]);           // storage rows / columns the fan can write
    const int T_lo = UNI(uy[byl - fyl]), T_hi = UNI(uy[byh - fyl]);
    const int gy_base = (T_lo - C) & ~3;                                      // window column 0 (C is a multiple of 4)
    int stride = (T_hi - C + 2 - gy_base + 3) & ~3;                           // columns gy_base .. T_hi - C + 1
    if (((stride >> 2) & 1) == 0) stride += 4;                                // rows an odd number of banks apart
    const int rows_cap = G.ncell / stride;                                    // global rows a window can hold
    BAR_LDS();
    if (UNI(s_fb) || rows_cap < 8) { GIVE_BACK(1); }
    if (tid < LL && s_need[tid] && s_tab[tid] < 0) {                          // allocate missing tiles (kept zero-filled)
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
    if (wave == 0) {   // levels: N_k = rays with at least k whole chunks (suffix sums over the wave: MAXLEV = 63)
        const int k = lane;                                                    // lane 0 is unused (level 0 = the 16-bit block)
        const int ck = k >= 1 ? s_lcnt[k] : 0;
        int suf = ck;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(suf, o, 64); if (lane + o < 64) suf += t; }
        // suf = N_k; rays with more chunks come first in perm
        const int nwk = k >= 1 ? (suf + 63) >> 6 : 0;
        int pre = nwk;                                                         // inclusive prefix of the levels' wave counts
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pre, o, 64); if (lane >= o) pre += t; }
        const unsigned long long live = __ballot(k >= 1 && suf > 0);
        const int nlev = live ? 63 - __clzll((long long)live) : 0;
        if (k >= 1) { s_lfill[k] = suf - ck; s_nk[k] = suf; s_lp[k] = pre - nwk; }
        if (k == 63) s_lp[64] = pre;
        if (k == 0) { s_nlev = nlev; s_nk[MAXLEV + 1] = 0; }
    }
    for (int lc = tid; lc < stride + 16 && lc < G.fanw + 16; lc += RB) {       // column glitch mask in window coordinates
        const int i = lc + gy_base - fyl;
        const bool gl = i >= 0 && i < nfx && gyb[i];
        gym[lc] = gl ? 0xFFu : 0u;
        if (gl) {   // the write-back's 32-column groups that see this column: its own and, for a group's first four columns, the one before
            const int sc = lc + gy_base + C, g = (sc >> 5) - (T_lo >> 5);
            if ((unsigned)g < 96u) s_ggf[g] = 1;
            if ((sc & 31) < 4 && (unsigned)(g - 1) < 96u) s_ggf[g - 1] = 1;
        }
    }
    uint32_t* const farh = cnt;                                                // [8 * NBIN] 16-bit counts of the rays longer than the 16-bit block (the window is not in use yet)
    farh[tid] = 0;
    {   // bucket fill pointers: exclusive prefix sum over the 2048 (class, bucket) counts, two per thread
        const uint32_t w2 = bins32[tid];
        const int c0 = (int)(w2 & 0xFFFFu), c1 = (int)(w2 >> 16), pc = c0 + c1;
        int incl = pc;
        for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
        if (lane == 63) s_wsum[wave] = incl;
        BAR_LDS();
        int wbase = 0;
        for (int k = 0; k < wave; ++k) wbase += s_wsum[k];
        const int excl = wbase + incl - pc;
        bins32[tid] = (uint32_t)excl | ((uint32_t)(excl + c0) << 16);
    }
    BAR_LDS();
    // window-address steps of a ray: per major step (cj) and per minor step (cm)
    auto ray_steps = [&](int b, int& cj, int& cm) {
        const int32_t re = r_end[b];
        const int ex = (int)(int16_t)(re & 0xFFFF), ey = (int)(int16_t)((uint32_t)re >> 16);
        const int aex = ex < 0 ? -ex : ex, aey = ey < 0 ? -ey : ey;
        const int sxs = ex > 0 ? stride : -stride, sy1 = ey > 0 ? 1 : -1;
        cj = aey > aex ? sy1 : sxs; cm = aey > aex ? sxs : sy1;
    };
    for (int b = tid; b < v.B; b += RB) {
        if (!(r_info[b] & RI_VALID)) continue;
        const int32_t e = r_end[b];
        const int key = ray_key((int)(int16_t)(e & 0xFFFF), (int)(int16_t)((uint32_t)e >> 16), r_fstep[b]);
        const int sh = (key & 1) * 16;
        const int pos = (int)((atomicAdd(&bins32[key >> 1], 1u << sh) >> sh) & 0xFFFFu);
        brays[pos] = (uint16_t)b;
        { int cj, cm; ray_steps(b, cj, cm); r_cc[b] = ((uint32_t)cj & 0xFFFFu) | ((uint32_t)cm << 16); }
        const int nfull = ((int)r_dmaj[b] + 1 - NEAR_R) / LCH;
        if (nfull >= 1) perm[atomicAdd(&s_lfill[min(nfull, MAXLEV)], 1)] = (uint16_t)b;
        if ((int)r_dmaj[b] >= NEAR_R) atomicAdd(&farh[key >> 1], 1u << sh);            // only these reach the 8-bit fields
    }
    BAR_LDS();
    // bucket key -> [start, end) in brays (the fill pointers have advanced to the bucket ends)
    auto bkt_start = [&](int key) { return key ? (int)bins16[key - 1] : 0; };
    auto bkt_end = [&](int key) { return (int)bins16[key]; };
    {   // no 8-bit field can overflow: a cell at major distance j >= NEAR_R is hit, per direction class, only by rays
        // that are at least that long and whose slope lies in a window of width 2^RFIX / j + 1 <= 2^RFIX / NEAR_R + 1, i.e. in
        // at most NB_WIN consecutive buckets.  First with all rays of the buckets (two reads of the fill pointers); only when that
        // bound fails, with the rays that are long enough to reach an 8-bit field.
        int mx = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int key = 2 * tid + k, cls = key / NBIN, bin = key % NBIN;
            mx = max(mx, bkt_end(cls * NBIN + min(bin + NB_WIN - 1, NBIN - 1)) - bkt_start(key));
        }
        mx = wave_max(mx);
        if (lane == 0 && mx > HIT_BOUND) s_exact = 1;
        BAR_LDS();
        if (UNI(s_exact)) {
            const uint16_t* fh = reinterpret_cast<const uint16_t*>(farh);
            mx = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int key = 2 * tid + k, cls = key / NBIN, bin = key % NBIN;
                int sum = 0;
                for (int d = 0; d < NB_WIN && bin + d < NBIN; ++d) sum += fh[cls * NBIN + bin + d];
                mx = max(mx, sum);
            }
            mx = wave_max(mx);
            if (lane == 0 && mx > HIT_BOUND) s_fb = 1;
        }
    }
    BAR_LDS();                                                                 // the window's first words become event counts next
    STAMP(1);

    // ---- flagged cells: those that receive an "occupied" or "nearby" hit (hybridmap.py:113,137,139-142) --------------
    // Pair (beam b, e) flags the storage cell of b's end cell (e = 0) or of the cell before it (e = 1); the smallest pair
    // that flags a storage cell owns it.  The rays of a direction class are sorted by slope, and the rays through a cell at
    // major distance j are exactly those with minor(j) = c: a contiguous run in that order around b itself.  Every pair
    // scans its neighbours in its own class (a handful), tests them exactly against the cell's global source cells and
    // keeps the (beam, rank) events; cells that rays of another class can reach too (on an axis or a diagonal through the
    // sensor) are finished by a second, dense pass with one lane per (cell, class).  The counter window is not in use yet:
    // the event lists live there.
    const int sat = (v.cc.vmax - v.cc.vmin + (-v.cc.emp) - 1) / (-v.cc.emp);        // hits that saturate any cell: 20
    const int npair = 2 * G.bpad;
    uint32_t* const evn32 = cnt;                                                   // [npair] 16-bit event counts
    uint16_t* const evl = reinterpret_cast<uint16_t*>(cnt + npair / 2);            // [npair][ECAP] (beam << 3) | rank
    uint16_t* const spl = evl + npair * ECAP;                                      // [RSPEC] pairs that need the other classes
    struct FCell { int sx, sy; int gx0, gx1, gy0, gy1; int ngx, ngy; };            // storage cell and its source global cells (scalars: no indexed arrays)
    auto cell_sources = [&](uint32_t sc, FCell& f) {
        f.sx = (int)(sc >> 16); f.sy = (int)(sc & 0xFFFFu);
        const int ax = f.sx - C, ay = f.sy - C;                                     // sources: a (if not glitched), a + 1 (if glitched)
        const bool xa = !gxb[ax - fxl], xb = gxb[ax + 1 - fxl], ya = !gyb[ay - fyl], yb = gyb[ay + 1 - fyl];
        f.ngx = (xa ? 1 : 0) + (xb ? 1 : 0); f.gx0 = xa ? ax : ax + 1; f.gx1 = ax + 1;
        f.ngy = (ya ? 1 : 0) + (yb ? 1 : 0); f.gy0 = ya ? ay : ay + 1; f.gy1 = ay + 1;
    };
    auto old_value = [&](const FCell& f) {
        const int a = a0 + (f.sx >= (a0 + 1) * v.dim ? 1 : 0) - (f.sx < a0 * v.dim ? 1 : 0);   // rays are shorter than a tile
        const int bb = b0 + (f.sy >= (b0 + 1) * v.dim ? 1 : 0) - (f.sy < b0 * v.dim ? 1 : 0);
        const int tile = s_need[a * v.L + bb] ? s_tab[a * v.L + bb] : -1;
        return tile >= 0 ? (int)v.pool[(size_t)tile * v.dim * v.dim + (size_t)(f.sx - a * v.dim) * v.dim + (f.sy - bb * v.dim)] : 0;
    };
    // does ray rb pass through the global cell at offset (ddx, ddy) from the start cell?  If so: the event's rank and
    // whether the NEARBY event follows (hybridmap.py:139-142).  Exact: the closed form of the reference's Bresenham.
    struct RayP { int ex, ey, dmaj, steep, smaj, smin, info; uint32_t fs; };
    auto load_ray = [&](int rb, RayP& r) {
        const int32_t re = r_end[rb];
        r.ex = (int)(int16_t)(re & 0xFFFF); r.ey = (int)(int16_t)((uint32_t)re >> 16);
        const int aex = r.ex < 0 ? -r.ex : r.ex, aey = r.ey < 0 ? -r.ey : r.ey;
        r.steep = aey > aex; r.dmaj = r.steep ? aey : aex;
        const int sx = r.ex > 0 ? 1 : -1, sy = r.ey > 0 ? 1 : -1;                    // hybridmap.py:282-283
        r.smaj = r.steep ? sy : sx; r.smin = r.steep ? sx : sy;
        r.fs = r_fstep[rb]; r.info = r_info[rb];
    };
    auto ray_hits = [&](const RayP& r, int ddx, int ddy, int& rank, bool& nearev) {
        const int j = (r.steep ? ddy : ddx) * r.smaj, c = (r.steep ? ddx : ddy) * r.smin;   // both must be >= 0
        if (j < 0 || c < 0 || j > r.dmaj) return false;
        if ((int)((r.fs * (uint32_t)j + (1u << (RFIX - 1))) >> RFIX) != c) return false;
        const int rem = r.dmaj - j;                                                  // steps left after this one
        rank = rem == 0 ? ((r.info & RI_OCC) ? EV_OCC : EV_E_LAST) : rem == 1 ? EV_E_2 : rem == 2 ? EV_E_3 : EV_E_FAR;
        nearev = rem == 1 && (r.info & RI_NEAR);
        return true;
    };
    // ---- the 16-bit block: steps 0 .. NEAR_R - 1 of every ray (it outlives the windows) ----
    auto walk_block = [&](bool collect, int* ncur, const int* nend, uint16_t* nev) {
        const int nw0 = (v.B + 63) >> 6;
        for (int q = wave; q < nw0; q += RB / 64) {
            const int b = lane * nw0 + q;                                          // the 64 rays of an instruction point in different directions
            if (b >= v.B || !(r_info[b] & RI_VALID)) continue;
            const uint32_t fs = r_fstep[b];
            const uint32_t cc = r_cc[b];
            const int cj = (int)(int16_t)(cc & 0xFFFFu), cm = (int)cc >> 16;
            const int dmaj = (int)r_dmaj[b];
            const int mj = cj == 1 || cj == -1 ? cj : (cj > 0 ? NEAR_W : -NEAR_W);   // the same steps in the 16-bit block
            const int mm = cm == 1 || cm == -1 ? cm : (cm > 0 ? NEAR_W : -NEAR_W);
            uint32_t facc = 1u << (RFIX - 1);
            int aj = NEAR_R * NEAR_W + NEAR_R;
            if (!collect) {
                aj += lds_addr(mini) >> 1;
#pragma unroll
                for (int u = 0; u < NEAR_R; ++u) {
                    fld16_add(aj + __mul24((int)(facc >> RFIX), mm), u <= dmaj ? 1u : 0u);   // (a step past the end stays inside the block: adds nothing)
                    facc += fs; aj += mj;
                }
            } else {
                const int info = r_info[b];
#pragma unroll
                for (int u = 0; u < NEAR_R; ++u) {
                    const int c = aj + __mul24((int)(facc >> RFIX), mm);
                    const int id = nid[c];
                    if (id != 0xFF && u <= dmaj) {                                     // rare: a source of a flagged cell
                        const int rem = dmaj - u;
                        const bool nearev = rem == 1 && (info & RI_NEAR);
                        const int rank = rem == 0 ? ((info & RI_OCC) ? EV_OCC : EV_E_LAST) : rem == 1 ? EV_E_2 : rem == 2 ? EV_E_3 : EV_E_FAR;
                        const int pos = atomicAdd(&ncur[id], nearev ? 2 : 1);
                        if (pos < nend[id]) nev[pos] = (uint16_t)(b << 3 | rank);
                        if (nearev && pos + 1 < nend[id]) nev[pos + 1] = (uint16_t)(b << 3 | EV_NEAR);
                    }
                    facc += fs; aj += mj;
                }
            }
        }
    };
    walk_block(false, nullptr, nullptr, nullptr);                                  // (no barrier: the waves go on to the bucket sort, whose threads mostly idle)
    {   // order the rays of every bucket by slope (ties by beam): the class is then sorted as a whole; rpos = inverse.
        // A regular scan has one or two rays per bucket (a thread sorts them in place); rays that end on a near surface
        // share end cells, hence slopes: buckets of a dozen or more are ranked by a wave.
        uint16_t* const bigb = evl + npair * ECAP;                                   // (the special list's place: not in use yet)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int key = 2 * tid + k, st = bkt_start(key), en = bkt_end(key);
            if (en - st > 256) s_fb = 1;
            else if (en - st > 6) { const int pos = atomicAdd(&s_nspec, 1); if (pos < RSPEC) bigb[pos] = (uint16_t)key; else s_fb = 1; }
            else for (int i = st + 1; i < en; ++i) {
                const int rb = brays[i];
                const uint32_t fbase = (uint32_t)(key % NBIN) << BIN_SHIFT;          // slopes of a bucket differ in their low 14 (15 in the last) bits
                const uint32_t kf = ((r_fstep[rb] - fbase) << 12) | (uint32_t)rb;
                int q = i - 1;
                while (q >= st) {
                    const int ro = brays[q];
                    if ((((r_fstep[ro] - fbase) << 12) | (uint32_t)ro) <= kf) break;
                    brays[q + 1] = (uint16_t)ro; --q;
                }
                brays[q + 1] = (uint16_t)rb;
            }
        }
        for (int i = tid; i < npair / 2; i += RB) evn32[i] = 0;
        for (int i = tid; i < npair / 4; i += RB) reinterpret_cast<uint32_t*>(pflag)[i] = 0;
        BAR_LDS();
        const int nbig = UNI(min(s_nspec, RSPEC));
        uint32_t* const kscr = reinterpret_cast<uint32_t*>(bigb + RSPEC) + wave * 256;   // the wave's keys (behind the list of big buckets: the ray records are not built yet)
        for (int k = wave; k < nbig; k += RB / 64) {
            const int key = bigb[k], st = bkt_start(key), n = bkt_end(key) - st;   // 7 .. 256 rays (more than 64: a wall right in front of the sensor)
            const uint32_t fbase = (uint32_t)(key % NBIN) << BIN_SHIFT;
            if (n <= 64) {
                const int rb = lane < n ? (int)brays[st + lane] : 0;
                const uint32_t kf = lane < n ? ((r_fstep[rb] - fbase) << 12) | (uint32_t)rb : 0xFFFFFFFFu;
                int rank = 0;
                for (int e = 0; e < n; ++e) rank += (uint32_t)__shfl((int)kf, e, 64) < kf;      // keys are distinct (the beam is part of them)
                if (lane < n) brays[st + rank] = (uint16_t)rb;                       // (every lane has read its own entry)
            } else {
                int rb[4]; uint32_t kf[4]; int rank[4] = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int at = lane + 64 * i;
                    rb[i] = at < n ? (int)brays[st + at] : 0;
                    kf[i] = at < n ? ((r_fstep[rb[i]] - fbase) << 12) | (uint32_t)rb[i] : 0xFFFFFFFFu;
                    if (at < n) kscr[at] = kf[i];
                }
                for (int e = 0; e < n; ++e) {
                    const uint32_t ke = kscr[e];
#pragma unroll
                    for (int i = 0; i < 4; ++i) rank[i] += ke < kf[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) if (lane + 64 * i < n) brays[st + rank[i]] = (uint16_t)rb[i];
            }
        }
        BAR_LDS();
        if (tid == 0) s_nspec = 0;
    }
    // the sorted order as packed records: one 8-byte read per candidate ray
    uint2* const srec = reinterpret_cast<uint2*>(spl + RSPEC);                     // [B] {slope | OCC << 24 | NEAR << 25, dmaj | beam << 16}
    uint16_t* const alist = reinterpret_cast<uint16_t*>(srec + G.bpad);            // [2 * B] pairs in play
    uint32_t* const poolkey = reinterpret_cast<uint32_t*>(alist + ((npair + 7) & ~7));   // [NPOOL] pair | events << 16 of a long list, ~0 = abandoned
    uint16_t* const pool = reinterpret_cast<uint16_t*>(poolkey + NPOOL);           // [NPOOL][PCAP]
    for (int q = tid; q < UNI(bkt_end(8 * NBIN - 1)); q += RB) {
        const int rb = brays[q];
        rpos[rb] = (uint16_t)q;
        const int info = r_info[rb];
        srec[q] = make_uint2(r_fstep[rb] | ((info & RI_OCC) ? 1u << 24 : 0u) | ((info & RI_NEAR) ? 1u << 25 : 0u),
                             (uint32_t)r_dmaj[rb] | ((uint32_t)rb << 16));
    }
    BAR_LDS();
    STAMP(2);
    // ---- pass 1: one lane per pair scans its beam's neighbours in slope order ----
    // In the frame of the beam's class a storage cell's global source cells are the major steps {j1, j2} x the minor
    // offsets {c1, c2} (the second of each only where the index map repeats a cell); a ray of the class crosses the cell
    // iff minor(j) is c1 or c2 for one of those j it reaches.
    // Pairs in play, listed in beam order: neighbouring beams end on the same surface at similar distances, so the lanes
    // of a wave scan windows of similar width and read neighbouring records.
    {
        for (int i0 = 0; i0 < v.B; i0 += RB) {                                     // wave-uniform trip count (ballot)
            const int i = i0 + tid;
            const int b = i < v.B ? i : -1;
            bool play[2] = {false, false};
            if (b >= 0 && (r_info[b] & (RI_VALID | RI_OCC)) == (RI_VALID | RI_OCC)) {
                // a smaller pair of the neighbouring beams on the same global cell: this one cannot be the owner (the other
                // duplicates show up in the scan)
                const uint32_t g0 = pair_gcell(2 * b), g1 = pair_gcell(2 * b + 1);
                const uint32_t p0 = b >= 1 ? pair_gcell(2 * b - 2) : 0xFFFFFFFFu, p1 = b >= 1 ? pair_gcell(2 * b - 1) : 0xFFFFFFFFu,
                               q0 = b >= 2 ? pair_gcell(2 * b - 4) : 0xFFFFFFFFu;
                if (g0 != p0 && g0 != p1 && g0 != q0) { play[0] = true; pflag[2 * b] = 3; }            // 3 = in play, not classified yet
                if (g1 != 0xFFFFFFFFu && g1 != g0 && g1 != p0 && g1 != p1 && g1 != q0) { play[1] = true; pflag[2 * b + 1] = 3; }
            }
            const unsigned long long m0 = __ballot(play[0]), m1 = __ballot(play[1]);
            int base = 0;
            if (lane == 0 && (m0 | m1)) base = atomicAdd(&s_nact, __popcll(m0) + __popcll(m1));
            base = __shfl(base, 0, 64);
            const unsigned long long lt = (1ull << lane) - 1ull;
            const int pos = base + __popcll(m0 & lt) + __popcll(m1 & lt);
            if (play[0]) alist[pos] = (uint16_t)(2 * b);
            if (play[1]) alist[pos + (play[0] ? 1 : 0)] = (uint16_t)(2 * b + 1);
        }
    }
    BAR_LDS();
    for (int it0 = 0; it0 < UNI(s_nact); it0 += RB) {                              // dense: (almost) every lane has a pair
        const int it = it0 + tid;
        const bool have = it < UNI(s_nact);
        const int mykey = have ? (int)alist[it] : 0;
        const int b = mykey >> 1;
        int nev = 0, oldv = 0, cap = ECAP, slot = -1;
        bool act = false;
        if (have) {
            const int32_t re_b = r_end[b];
            const int ex = (int)(int16_t)(re_b & 0xFFFF), ey = (int)(int16_t)((uint32_t)re_b >> 16);
            const int aex = ex < 0 ? -ex : ex, aey = ey < 0 ? -ey : ey;
            const int steep = aey > aex, dmaj_b = steep ? aey : aex;
            const int smaj = steep ? (ey > 0 ? 1 : -1) : (ex > 0 ? 1 : -1), smin = steep ? (ex > 0 ? 1 : -1) : (ey > 0 ? 1 : -1);
            const int cls = steep * 4 + (ex > 0 ? 2 : 0) + (ey > 0 ? 1 : 0);
            const int cst = bkt_start(cls * NBIN), cen = bkt_end(cls * NBIN + NBIN - 1);
            FCell f;
            cell_sources(pair_cell(mykey), f);
            oldv = old_value(f);                                                   // in flight during the scan
            // sources along the major and the minor axis of the class frame
            const int gmaj0 = steep ? f.gy0 : f.gx0, gmaj1 = steep ? f.gy1 : f.gx1, nmaj = steep ? f.ngy : f.ngx;
            const int gmin0 = steep ? f.gx0 : f.gy0, gmin1 = steep ? f.gx1 : f.gy1, nmin = steep ? f.ngx : f.ngy;
            const int omaj = steep ? y0 : x0, omin = steep ? x0 : y0;
            const int cj1 = (gmaj0 - omaj) * smaj, cj2 = nmaj > 1 ? (gmaj1 - omaj) * smaj : -1;
            const int cc1 = (gmin0 - omin) * smin, cc2 = nmin > 1 ? (gmin1 - omin) * smin : -1;
            const int jmin = nmaj > 1 ? min(cj1, cj2) : cj1, jmax = nmaj > 1 ? max(cj1, cj2) : cj1;
            const int cmin = nmin > 1 ? min(cc1, cc2) : cc1, cmax = nmin > 1 ? max(cc1, cc2) : cc1;
            bool near_cell = false;
#pragma unroll
            for (int ix = 0; ix < 2; ++ix)
#pragma unroll
            for (int iy = 0; iy < 2; ++iy) {
                if (ix >= f.ngx || iy >= f.ngy) continue;
                const int sgx = ix ? f.gx1 : f.gx0, sgy = iy ? f.gy1 : f.gy0;
                const int ddx = sgx - x0, ddy = sgy - y0;
                if (max(ddx < 0 ? -ddx : ddx, ddy < 0 ? -ddy : ddy) < NEAR_R) near_cell = true;
            }
            const bool special = cmin <= 0 || cmax >= jmin;                         // an axis or a diagonal through the sensor: other classes reach it
            if (near_cell) {   // most rays cross it; claimed through bit 15 of the first source's 16-bit field
                const int mi = (f.gx0 - x0 + NEAR_R) * NEAR_W + (f.gy0 - y0 + NEAR_R), sh = (mi & 1) * 16;
                if (!((atomicOr(&mini[mi >> 1], 0x8000u << sh) >> sh) & 0x8000u)) {
                    // every source within the block's walk (steps 0 .. NEAR_R - 1): its events come from that walk;
                    // otherwise (a source on the block's rim) the scan over all beams
                    bool inside = true;
#pragma unroll
                    for (int ix = 0; ix < 2; ++ix)
#pragma unroll
                    for (int iy = 0; iy < 2; ++iy) {
                        if (ix >= f.ngx || iy >= f.ngy) continue;
                        const int ddx = (ix ? f.gx1 : f.gx0) - x0, ddy = (iy ? f.gy1 : f.gy0) - y0;
                        if (max(ddx < 0 ? -ddx : ddx, ddy < 0 ? -ddy : ddy) >= NEAR_R) inside = false;
                    }
                    int pos = inside ? atomicAdd(&s_nnear, 1) : NNEAR;
                    if (pos < NNEAR) {
                        nearl[pos] = (uint16_t)mykey;
#pragma unroll
                        for (int ix = 0; ix < 2; ++ix)
#pragma unroll
                        for (int iy = 0; iy < 2; ++iy) {
                            if (ix >= f.ngx || iy >= f.ngy) continue;
                            nid[((ix ? f.gx1 : f.gx0) - x0 + NEAR_R) * NEAR_W + ((iy ? f.gy1 : f.gy0) - y0 + NEAR_R)] = (uint8_t)pos;
                        }
                    } else {
                        pos = atomicAdd(&s_nslow, 1);
                        if (pos < RSLOW) slowl[pos] = (uint16_t)mykey; else s_fb = 1;
                    }
                }
                pflag[mykey] = 0;
            } else {
                act = true;
                // window on minor(dmaj_b) of a ray that can reach a source: minor() never decreases and moves by at most one
                // per major step, so minor(js) = cs needs minor(dmaj_b) in [cs, cs + (dmaj_b - js)] (js <= dmaj_b) or
                // [cs - (js - dmaj_b), cs] (js > dmaj_b)
                const int lo = cmin - max(0, jmax - dmaj_b), hi = cmax + max(0, dmaj_b - jmin);
                pflag[mykey] = special ? 2 : 1;
                if (special) {
                    const int pos = atomicAdd(&s_nspec, 1);
                    if (pos < RSPEC) spl[pos] = (uint16_t)mykey; else s_fb = 1;
                }
                uint16_t* myev = evl + mykey * ECAP;
                const int q0 = rpos[b];
                // one candidate record: past the window (-> that direction is finished), or tested against the cell's sources
                auto candidate = [&](const uint2 rr, const bool down, bool& finished) {
                    const uint32_t fs = rr.x & 0xFFFFFFu;
                    const int dmaj_r = (int)(rr.y & 0xFFFFu), rb = (int)(rr.y >> 16);
                    const int mg = (int)((fs * (uint32_t)dmaj_b + (1u << (RFIX - 1))) >> RFIX);
                    if (down ? mg < lo : mg > hi) { finished = true; return; }      // sorted by slope: nothing further can reach the cell
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const int jj = w ? cj2 : cj1;
                        if (jj < 0 || jj > dmaj_r) continue;
                        const int m = (int)((fs * (uint32_t)jj + (1u << (RFIX - 1))) >> RFIX);
                        if (m != cc1 && m != cc2) continue;
                        const int rem = dmaj_r - jj;
                        const bool occ_r = (rr.x >> 24) & 1u, nearev = rem == 1 && ((rr.x >> 25) & 1u);
                        const int rank = rem == 0 ? (occ_r ? EV_OCC : EV_E_LAST) : rem == 1 ? EV_E_2 : rem == 2 ? EV_E_3 : EV_E_FAR;
                        if ((rank == EV_OCC && (rb << 1) < mykey) || (nearev && (rb << 1 | 1) < mykey)) { act = false; return; }
                        if (nev + 2 > cap && cap == ECAP && !special) {                    // the list is full: move to a long one
                            slot = atomicAdd(&s_npool, 1);
                            if (slot < NPOOL) {
                                uint16_t* const pl = pool + slot * PCAP;
                                for (int i = 0; i < nev; ++i) pl[i] = myev[i];
                                myev = pl; cap = PCAP;
                            } else slot = -1;
                        }
                        if (nev < cap) myev[nev] = (uint16_t)(rb << 3 | rank);
                        ++nev;
                        if (nearev) { if (nev < cap) myev[nev] = (uint16_t)(rb << 3 | EV_NEAR); ++nev; }
                    }
                };
                // both directions at once - b and the steeper slopes upwards, the flatter ones downwards: two independent
                // record reads in flight per step, half the steps (the order of the events does not matter: they are sorted)
                int qu = q0, qd = q0 - 1;
                bool fu = qu >= cen, fd = qd < cst;
                while (act && !(fu && fd)) {
                    const uint2 ru = srec[fu ? q0 : qu], rd = srec[fd ? q0 : qd];
                    if (!fu) candidate(ru, false, fu);
                    if (act && !fd) candidate(rd, true, fd);
                    ++qu; --qd;
                    fu = fu || qu >= cen; fd = fd || qd < cst;
                }
                if (!act) pflag[mykey] = 0;
            }
        }
        // finalize: the network that fits the longest list of the wave
        bool fin = false;
        if (act) {
            oldv8[mykey] = (uint8_t)oldv;
            if (pflag[mykey] == 2) {                                               // the other classes add their events in pass 2
                atomicAdd(&evn32[mykey >> 1], (uint32_t)nev << ((mykey & 1) * 16));
            } else if (nev > cap) {
                const int pos = atomicAdd(&s_nslow, 1);
                if (pos < RSLOW) slowl[pos] = (uint16_t)mykey; else s_fb = 1;
            } else if (slot < 0) fin = true;
        }
        if (slot >= 0 && slot < NPOOL) poolkey[slot] = act && nev <= cap ? (uint32_t)mykey | ((uint32_t)nev << 16) : 0xFFFFFFFFu;
        const int nmax = wave_max(fin ? nev : 0);
        if (fin) {
            const uint16_t* myev = evl + mykey * ECAP;
            const int val = nmax <= 4 ? replay_sorted<4>(myev, nev, oldv, v.cc) : nmax <= 8 ? replay_sorted<8>(myev, nev, oldv, v.cc)
                                                                                            : replay_sorted<ECAP>(myev, nev, oldv, v.cc);
            oval[mykey] = (uint8_t)(val - v.cc.vmin);
        }
    }
    BAR_LDS();
    STAMP(3);
    {   // long lists (17 .. PCAP events): a wave sorts by counting and folds the clamped adds with a shuffle tree
        const int npool = UNI(min(s_npool, NPOOL));
        for (int k = wave; k < npool; k += RB / 64) {
            const uint32_t pk = poolkey[k];
            if (pk == 0xFFFFFFFFu) continue;
            const int key = (int)(pk & 0xFFFFu), m = (int)(pk >> 16);
            const uint32_t ekey = lane < m ? (uint32_t)pool[k * PCAP + lane] : 0xFFFFFFFFu;
            int rank = 0;
            for (int e = 0; e < m; ++e) {
                const uint32_t ke = (uint32_t)__shfl((int)ekey, e, 64);
                rank += (ke < ekey) || (ke == ekey && e < lane);
            }
            const uint32_t sorted = (uint32_t)__builtin_amdgcn_ds_permute((lane < m ? rank : lane) << 2, (int)ekey);
            const int BIG = 1000000;
            Caf fc = {0, -BIG, BIG};
            if (lane < m) {
                const int rk = (int)(sorted & 7u);
                fc = rk == EV_OCC ? Caf{v.cc.occ, -BIG, v.cc.vmax} : rk == EV_NEAR ? Caf{v.cc.nearby, -BIG, v.cc.vmax} : Caf{v.cc.emp, v.cc.vmin, BIG};
            }
            for (int off = 1; off < 64; off <<= 1) {
                Caf g;
                g.a = __shfl_down(fc.a, off, 64); g.lo = __shfl_down(fc.lo, off, 64); g.hi = __shfl_down(fc.hi, off, 64);
                if ((lane & (2 * off - 1)) == 0) fc = caf_then(fc, g);
            }
            if (lane == 0) oval[key] = (uint8_t)(caf_apply(fc, (int)(int8_t)oldv8[key]) - v.cc.vmin);
        }
    }
    // ---- pass 2b: cells other classes reach too; one lane per (cell, other class) ----
    {
        const int nspec = UNI(min(s_nspec, RSPEC));
        for (int it = tid; it < nspec * 8; it += RB) {
            const int pair = spl[it >> 3], cls = it & 7, b = pair >> 1;
            RayP me;
            load_ray(b, me);
            if (cls == me.steep * 4 + (me.ex > 0 ? 2 : 0) + (me.ey > 0 ? 1 : 0)) continue;      // done in pass 1
            FCell f;
            cell_sources(pair_cell(pair), f);
            // slope buckets of this class that can hold a ray through one of the sources
            const int steep = cls >> 2, smaj = steep ? ((cls & 1) ? 1 : -1) : ((cls & 2) ? 1 : -1), smin = steep ? ((cls & 2) ? 1 : -1) : ((cls & 1) ? 1 : -1);
            int blo = NBIN, bhi = -1;
#pragma unroll
            for (int ix = 0; ix < 2; ++ix)
#pragma unroll
            for (int iy = 0; iy < 2; ++iy) {
                if (ix >= f.ngx || iy >= f.ngy) continue;
                const int sgx = ix ? f.gx1 : f.gx0, sgy = iy ? f.gy1 : f.gy0;
                const int ddx = sgx - x0, ddy = sgy - y0;
                const int j = (steep ? ddy : ddx) * smaj, c = (steep ? ddx : ddy) * smin;
                if (j <= 0 || c < 0 || c > j) continue;
                const float inv = (float)NBIN / (float)j;
                blo = min(blo, max(0, (int)(((float)c - 0.5f) * inv) - 1));
                bhi = max(bhi, min(NBIN - 1, (int)(((float)c + 0.5f) * inv) + 1));
            }
            if (bhi < blo) continue;
            const int psh = (pair & 1) * 16;
            for (int q = bkt_start(cls * NBIN + blo); q < bkt_end(cls * NBIN + bhi); ++q) {
                const int rb = brays[q];
                RayP r;
                load_ray(rb, r);
#pragma unroll
                for (int ix = 0; ix < 2; ++ix)
#pragma unroll
                for (int iy = 0; iy < 2; ++iy) {
                    if (ix >= f.ngx || iy >= f.ngy) continue;
                    const int sgx = ix ? f.gx1 : f.gx0, sgy = iy ? f.gy1 : f.gy0;
                    int rank; bool nearev;
                    if (!ray_hits(r, sgx - x0, sgy - y0, rank, nearev)) continue;
                    int pos = (int)((atomicAdd(&evn32[pair >> 1], 1u << psh) >> psh) & 0xFFFFu);
                    if (pos < ECAP) evl[pair * ECAP + pos] = (uint16_t)(rb << 3 | rank);
                    if (nearev) {
                        pos = (int)((atomicAdd(&evn32[pair >> 1], 1u << psh) >> psh) & 0xFFFFu);
                        if (pos < ECAP) evl[pair * ECAP + pos] = (uint16_t)(rb << 3 | EV_NEAR);
                    }
                }
            }
        }
    }
    BAR_LDS();
    // ---- pass 3: the cells of pass 2: owner check, sort, replay ----
    for (int it = tid; it < UNI(min(s_nspec, RSPEC)); it += RB) {
        const int pair = spl[it];
        if (pflag[pair] != 2) continue;                                          // abandoned in pass 1: a smaller pair owns the cell
        const int m = (int)((evn32[pair >> 1] >> ((pair & 1) * 16)) & 0xFFFFu);
        if (m > ECAP) {                                                          // (duplicates may both land here: same value twice)
            const int pos = atomicAdd(&s_nslow, 1);
            if (pos < RSLOW) slowl[pos] = (uint16_t)pair; else s_fb = 1;
            continue;
        }
        const uint16_t* evp = evl + pair * ECAP;
        bool owner = true;
        for (int k = 0; k < m; ++k) {
            const int key = evp[k], rank = key & 7, rp = (key >> 3) << 1;
            if ((rank == EV_OCC && rp < pair) || (rank == EV_NEAR && (rp | 1) < pair)) owner = false;
        }
        if (!owner) continue;
        const int val = m <= 8 ? replay_sorted<8>(evp, m, (int)(int8_t)oldv8[pair], v.cc) : replay_sorted<ECAP>(evp, m, (int)(int8_t)oldv8[pair], v.cc);
        oval[pair] = (uint8_t)(val - v.cc.vmin);
    }
    BAR_LDS();
    if (UNI(s_fb)) { GIVE_BACK(UNI(s_nslow) > RSLOW || UNI(s_nspec) > RSPEC ? 3 : 2); }
    // ---- flagged cells inside the block: their events come from a second walk of the block.  All unoccupied passes
    //      are the same clamped add, so only their NUMBER between consecutive occupied / nearby events (in beam order)
    //      matters: a wave per cell sorts those few events and counts the passes into the intervals they bound. ----
    {
        const int nnear = UNI(min(s_nnear, NNEAR));
        if (nnear > 0) {
            int* const ncur = reinterpret_cast<int*>(cnt);                         // [NNEAR] fill pointers (the event lists of the passes above are done with)
            int* const nend = ncur + NNEAR;                                        // [NNEAR] end of the cell's list
            int* const nbeg = nend + NNEAR;                                        // [NNEAR]
            uint16_t* const wsp = reinterpret_cast<uint16_t*>(nbeg + NNEAR);       // [waves][NSPC] sorted occupied / nearby events
            int* const wcnt = reinterpret_cast<int*>(wsp + (RB / 64) * NSPC);      // [waves][NSPC + 1] passes per interval
            uint16_t* const nev = reinterpret_cast<uint16_t*>(wcnt + (RB / 64) * (NSPC + 1));
            const int nev_cap = min((G.ncell - (int)((unsigned char*)nev - (unsigned char*)cnt)) / 2, 32767);
            if (wave == 0) {                                                       // list sizes: the passes counted by the first walk + room for nearby events
                int run = 0;
                for (int i0 = 0; i0 < nnear; i0 += 64) {
                    const int id = i0 + lane;
                    int need = 0;
                    if (id < nnear) {
                        FCell f;
                        cell_sources(pair_cell(nearl[id]), f);
#pragma unroll
                        for (int ix = 0; ix < 2; ++ix)
#pragma unroll
                        for (int iy = 0; iy < 2; ++iy) {
                            if (ix >= f.ngx || iy >= f.ngy) continue;
                            const int mi = ((ix ? f.gx1 : f.gx0) - x0 + NEAR_R) * NEAR_W + ((iy ? f.gy1 : f.gy0) - y0 + NEAR_R);
                            need += (int)((mini[mi >> 1] >> ((mi & 1) * 16)) & 0x7FFFu);
                        }
                        need += NSPC;
                    }
                    const int ex = run + wave_excl_scan(need, lane);
                    if (id < nnear) {
                        const int beg = min(ex, nev_cap), end = min(ex + need, nev_cap);   // a list that does not fit: the scan over all beams
                        nbeg[id] = beg; ncur[id] = beg; nend[id] = end < ex + need ? beg : end;
                    }
                    run += wave_sum(need);
                }
            }
            BAR_LDS();
            walk_block(true, ncur, nend, nev);
            BAR_LDS();
            for (int id = wave; id < nnear; id += RB / 64) {
                const int key = nearl[id];
                const int beg = nbeg[id], m = ncur[id] - beg;
                uint16_t* const sp = wsp + wave * NSPC;
                int* const ic = wcnt + wave * (NSPC + 1);
                FCell f;
                cell_sources(pair_cell(key), f);
                const int val0 = lane == 0 ? old_value(f) : 0;
                bool ok = beg + m <= nend[id] && nend[id] > beg;
                // the occupied / nearby events, sorted by (beam, rank)
                int k = 0;
                for (int i0 = 0; i0 < m && ok; i0 += 64) {
                    const int i = i0 + lane;
                    const uint32_t ev = i < m ? nev[beg + i] : 0u;
                    const bool spc = i < m && (ev & 7u) >= (uint32_t)EV_OCC;
                    const unsigned long long mk = __ballot(spc);
                    const int at = k + __popcll(mk & ((1ull << lane) - 1ull));
                    if (spc && at < NSPC) sp[at] = (uint16_t)ev;
                    k += __popcll(mk);
                }
                if (k > NSPC) ok = false;
                if (!ok) {                                                         // (uniform) too many events: the scan over all beams
                    if (lane == 0) { const int pos = atomicAdd(&s_nslow, 1); if (pos < RSLOW) slowl[pos] = (uint16_t)key; else s_fb = 1; }
                    continue;
                }
                {
                    const uint32_t mine = lane < k ? (uint32_t)sp[lane] : 0xFFFFFFFFu;
                    int rk = 0;
                    for (int e = 0; e < k; ++e) { const uint32_t o = (uint32_t)__shfl((int)mine, e, 64); rk += (o < mine) || (o == mine && e < lane); }
                    if (lane <= NSPC) ic[lane] = 0;
                    if (lane < k) sp[rk] = (uint16_t)mine;                         // (every lane has read its own entry)
                }
                // passes: interval = number of occupied / nearby events of smaller beams (a beam's own passes come first)
                for (int i0 = 0; i0 < m; i0 += 64) {
                    const int i = i0 + lane;
                    const uint32_t ev = i < m ? nev[beg + i] : 0xFFFFu;
                    if (i < m && (ev & 7u) < (uint32_t)EV_OCC) {
                        const uint32_t bm = ev >> 3;
                        int iv = 0;
                        for (int e = 0; e < k; ++e) iv += ((uint32_t)sp[e] >> 3) < bm;
                        atomicAdd(&ic[iv], 1);
                    }
                }
                {   // lane e <= k: the passes of interval e, then event e; folded in lane order by a shuffle tree
                    const int BIG = 1000000;
                    Caf fc = {0, -BIG, BIG};
                    if (lane <= k) {
                        const int n = min(ic[lane], sat);
                        fc = Caf{n * v.cc.emp, v.cc.vmin, BIG};                   // n clamped adds of emp (gridmap.py:97-101) = one with n * emp
                        if (lane < k) {
                            const int rk = (int)(sp[lane] & 7u);
                            fc = caf_then(fc, rk == EV_OCC ? Caf{v.cc.occ, -BIG, v.cc.vmax} : Caf{v.cc.nearby, -BIG, v.cc.vmax});
                        }
                    }
                    for (int off = 1; off < 64; off <<= 1) {
                        Caf g;
                        g.a = __shfl_down(fc.a, off, 64); g.lo = __shfl_down(fc.lo, off, 64); g.hi = __shfl_down(fc.hi, off, 64);
                        if ((lane & (2 * off - 1)) == 0) fc = caf_then(fc, g);
                    }
                    if (lane == 0) oval[key] = (uint8_t)(caf_apply(fc, val0) - v.cc.vmin);
                }
            }
            BAR_LDS();
            if (UNI(s_fb)) { GIVE_BACK(3); }                                       // (the list of cells for the scan over all beams is full)
        }
    }
    {   // cells near the sensor and cells with more than ECAP events: exact membership test over all beams, one cell at a
        // time by the whole workgroup.  A lane folds the events of its beam (the clamped adds compose associatively: Caf),
        // a wave folds its 64 consecutive beams with a shuffle tree, thread 0 folds the waves' results in beam order.
        const int nslow = UNI(s_nslow);
        int* const s_caf = reinterpret_cast<int*>(cnt);                            // [chunks of 64 beams][3] (the event lists are done with)
        const int BIG = 1000000;
        const Caf fE = {v.cc.emp, v.cc.vmin, BIG}, fO = {v.cc.occ, -BIG, v.cc.vmax}, fN = {v.cc.nearby, -BIG, v.cc.vmax};
        for (int k = 0; k < nslow; ++k) {
            const int key = UNI(slowl[k]);
            FCell f;
            cell_sources(pair_cell(key), f);
            const int val0 = tid == 0 ? old_value(f) : 0;
            for (int base = 0; base < v.B; base += RB) {
                const int b = base + tid;
                Caf fb = {0, -BIG, BIG};
                if (b < v.B && (r_info[b] & RI_VALID)) {
                    RayP r;
                    load_ray(b, r);
                    bool occ_ev = false, near_ev = false;
#pragma unroll
                    for (int ix = 0; ix < 2; ++ix)
#pragma unroll
                    for (int iy = 0; iy < 2; ++iy) {
                        if (ix >= f.ngx || iy >= f.ngy) continue;
                        const int sgx = ix ? f.gx1 : f.gx0, sgy = iy ? f.gy1 : f.gy0;
                        int rank; bool ne;
                        if (!ray_hits(r, sgx - x0, sgy - y0, rank, ne)) continue;
                        if (rank == EV_OCC) occ_ev = true; else fb = caf_then(fb, fE);   // the beam's passes come before its last step
                        near_ev = near_ev || ne;
                    }
                    if (occ_ev) fb = caf_then(fb, fO);
                    if (near_ev) fb = caf_then(fb, fN);
                }
                if (base + 64 * wave < v.B) {                                      // (wave-uniform)
                    for (int off = 1; off < 64; off <<= 1) {
                        Caf g;
                        g.a = __shfl_down(fb.a, off, 64); g.lo = __shfl_down(fb.lo, off, 64); g.hi = __shfl_down(fb.hi, off, 64);
                        if ((lane & (2 * off - 1)) == 0) fb = caf_then(fb, g);
                    }
                    if (lane == 0) { int* o = s_caf + 3 * ((base >> 6) + wave); o[0] = fb.a; o[1] = fb.lo; o[2] = fb.hi; }
                }
            }
            BAR_LDS();
            if (tid == 0) {
                int val = val0;
                for (int c = 0; c < (v.B + 63) >> 6; ++c) { const Caf g = {s_caf[3 * c], s_caf[3 * c + 1], s_caf[3 * c + 2]}; val = caf_apply(g, val); }
                oval[key] = (uint8_t)(val - v.cc.vmin);
            }
            BAR_LDS();
        }
        if (tid == 0 && nslow) atomicAdd(&v.stats[ST_SLOW_CELLS], (unsigned long long)nslow);
    }
    STAMP(3);

    // =============================================== windows ==============================================
    const uint32_t satb = (uint32_t)sat * 0x01010101u, sadd = (128u - (uint32_t)sat) * 0x01010101u;
    int n_win = 0;
    for (int S0 = S_lo; S0 <= S_hi; S0 += rows_cap - 1, ++n_win) {
        const int S1 = min(S_hi, S0 + rows_cap - 2);                             // storage rows S0..S1
        const int gx_base = S0 - C, rows_w = S1 - S0 + 2;                        // global rows gx_base .. gx_base + rows_w - 1
        BAR_LDS();                                                               // the previous window is done with the counters
        {
            uint4* c4 = reinterpret_cast<uint4*>(cnt);
            const int n16 = (rows_w * stride + 15) >> 4;
            for (int i = tid; i < n16; i += RB) c4[i] = make_uint4(0, 0, 0, 0);
        }
        BAR_LDS();
        STAMP(4);
        // ---- walk: lanes are rays, a work item is one 16-step chunk of 64 rays ----
        // Level k >= 1 = steps NEAR_R + 16 (k - 1) .. + 15.  The rays that own a whole k-th chunk are perm[0 .. N_k) (rays ordered by falling
        // chunk count), so every lane of an item runs all 16 steps: no predicates.  Lane l of the w-th wave of a level takes
        // ray l * (waves of the level) + w: the 64 rays of one instruction point in different directions and touch
        // different cells.  Step j of a ray is field base0 + j * cj + minor(j) * cm, minor(j) = (fstep * j + 2^21) >> 22:
        // five instructions and a fire-and-forget LDS add.  The first NEAR_R steps went into the 16-bit block (before the windows: it
        // outlives the windows); the last, partial chunk of every ray is a predicated item of its own.
        {
            const int rx0 = x0 - gx_base, ry0 = y0 - gy_base;
            const int base0 = rx0 * stride + ry0;
            const bool whole = S0 == S_lo && S1 == S_hi;                           // one window holds the fan: no row test
            const int cnt_lds = lds_addr(cnt);
            const int nlev = UNI(s_nlev);                                          // levels 1 .. nlev have whole chunks
            const int nitems = UNI(s_lp[nlev + 1]);                               // (levels above nlev have no waves: s_lp stays flat)
            // one whole chunk of a ray in a strip: every step tests its row
            auto strip_chunk = [&](int b, int k) {
                const uint32_t fs = r_fstep[b];
                const uint32_t cc = r_cc[b];
                const int cj = (int)(int16_t)(cc & 0xFFFFu), cm = (int)cc >> 16;
                const int j0 = NEAR_R + (k - 1) * LCH;
                uint32_t facc = (uint32_t)__umul24(fs, (uint32_t)j0) + (1u << (RFIX - 1));
                int aj = base0 + __mul24(j0, cj);
                // rows: row = rx0 + j * rj + m * rm, where (rj, rm) = (+-1, 0) for a ray along x and (0, +-1) along y
                const int rj = (cj == 1 || cj == -1) ? 0 : (cj > 0 ? 1 : -1), rm = (cm == 1 || cm == -1) ? 0 : (cm > 0 ? 1 : -1);
                int rowj = rx0 + j0 * rj;
#pragma unroll
                for (int u = 0; u < LCH; ++u) {
                    const int m = (int)(facc >> RFIX);
                    const bool in = (unsigned)(rowj + m * rm) < (unsigned)rows_w;
                    const int c = in ? aj + __mul24(m, cm) : 4 * lane;                     // outside the strip: nothing added, a word of the lane's own
                    atomicAdd(&cnt[c >> 2], in ? 1u << ((c & 3) * 8) : 0u);
                    facc += fs; aj += cj; rowj += rj;
                }
            };
            bool by_level = true;
            int nch_all = 0;                                                       // whole chunks of all rays (uniform)
            for (int kk = 1; kk <= nlev; ++kk) nch_all += UNI(s_nk[kk]);
            if (!whole && nch_all <= 65535) {                                      // (16-bit prefixes; decided per particle: the strips share perm's place)
                // A strip holds a part of every ray: the chunks of a ray that can reach the strip's rows are a run of levels
                // (exactly for a ray along x, from a single-precision bound with two steps of slack for a ray along y).  Their
                // counts are prefix-summed over the rays and a lane takes one (ray, chunk) item found by bisection, so that a
                // strip costs its own share of the walk and not the whole walk again.
                uint16_t* const cpre = perm;                                       // [B] first item of the ray (the level order is not needed in strips)
                uint8_t* const cka = pflag;                                        // [B] first level of the ray in this strip
                int nloc[4], tot = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int b = 4 * tid + i;
                    int n = 0;
                    if (b < v.B && (r_info[b] & RI_VALID)) {
                        const int nfull = ((int)r_dmaj[b] + 1 - NEAR_R) / LCH;
                        if (nfull >= 1) {
                            const uint32_t fs = r_fstep[b];
                            const uint32_t cc = r_cc[b];
                            const int cj = (int)(int16_t)(cc & 0xFFFFu), cm = (int)cc >> 16;
                            const int rj = (cj == 1 || cj == -1) ? 0 : (cj > 0 ? 1 : -1), rm = (cm == 1 || cm == -1) ? 0 : (cm > 0 ? 1 : -1);
                            int jlo = NEAR_R, jhi = NEAR_R + nfull * LCH - 1;
                            if (rj > 0) { jlo = max(jlo, -rx0); jhi = min(jhi, rows_w - 1 - rx0); }
                            else if (rj < 0) { jlo = max(jlo, rx0 - rows_w + 1); jhi = min(jhi, rx0); }
                            else {
                                const int mlo = max(rm > 0 ? -rx0 : rx0 - rows_w + 1, 0), mhi = rm > 0 ? rows_w - 1 - rx0 : rx0;
                                if (mhi < mlo) jhi = -1;
                                else if (fs == 0) { if (mlo > 0) jhi = -1; }                  // the minor offset stays 0
                                else {   // minor(j) = floor(j * fs / 2^22 + 1/2) in [mlo, mhi]
                                    const float inv = 4194304.0f / (float)fs;
                                    jlo = max(jlo, (int)(((float)mlo - 0.5f) * inv) - 2);
                                    const float ju = ((float)mhi + 0.5f) * inv;
                                    if (ju < 1.0e9f) jhi = min(jhi, (int)ju + 2);
                                }
                            }
                            if (jhi >= jlo) {
                                const int ka = 1 + (jlo - NEAR_R) / LCH, kb = 1 + (jhi - NEAR_R) / LCH;
                                n = kb - ka + 1;
                                cka[b] = (uint8_t)ka;
                            }
                        }
                    }
                    nloc[i] = n; tot += n;
                }
                int incl = tot;
                for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                if (lane == 63) s_wsum[wave] = incl;
                BAR_LDS();
                int run = incl - tot;
                for (int k = 0; k < wave; ++k) run += s_wsum[k];
                int total = 0;
                for (int k = 0; k < RB / 64; ++k) total += s_wsum[k];
                total = UNI(total);
                by_level = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int b = 4 * tid + i; if (b < v.B) cpre[b] = (uint16_t)run; run += nloc[i]; }
                BAR_LDS();
                for (int q = wave * 64; q < total; q += RB) {
                    const int it = q + lane;
                    if (it >= total) continue;
                    int b = 0;
                    for (int st = 2048; st; st >>= 1) { const int cand = b + st; if (cand < v.B && (int)cpre[cand] <= it) b = cand; }
                    strip_chunk(b, (int)cka[b] + (it - (int)cpre[b]));
                }
            }
            if (by_level)
            for (int q = wave; q < nitems; q += RB / 64) {
                int k = 1;
                for (int kk = 2; kk <= nlev; ++kk) if (q >= UNI(s_lp[kk])) k = kk;
                const int nk = UNI(s_nk[k]), nwk = (nk + 63) >> 6, wslot = q - UNI(s_lp[k]);
                const int ii = lane * nwk + wslot;
                if (ii >= nk) continue;
                const int b = perm[ii];
                if (whole) {
                    const uint32_t fs = r_fstep[b];
                    const uint32_t cc = r_cc[b];
                    const int cj = (int)(int16_t)(cc & 0xFFFFu), cm = (int)cc >> 16;
                    const int j0 = NEAR_R + (k - 1) * LCH;
                    uint32_t facc = (uint32_t)__umul24(fs, (uint32_t)j0) + (1u << (RFIX - 1));
                    int aj = base0 + __mul24(j0, cj) + cnt_lds;
#pragma unroll
                    for (int u = 0; u < LCH; ++u) {
                        fld8_add(aj + __mul24((int)(facc >> RFIX), cm), 1u);
                        facc += fs; aj += cj;
                    }
                } else strip_chunk(b, k);
            }
            // the partial chunk at the end of every ray with at least NEAR_R + 1 steps
            for (int b = tid; b < v.B; b += RB) {
                const int dmaj = (int)r_dmaj[b];
                if (!(r_info[b] & RI_VALID) || dmaj < NEAR_R) continue;
                const int j0 = NEAR_R + ((dmaj + 1 - NEAR_R) / LCH) * LCH;         // first step after the whole chunks
                if (j0 > dmaj) continue;
                const uint32_t fs = r_fstep[b];
                const uint32_t cc = r_cc[b];
                const int cj = (int)(int16_t)(cc & 0xFFFFu), cm = (int)cc >> 16;
                const int rj = (cj == 1 || cj == -1) ? 0 : (cj > 0 ? 1 : -1), rm = (cm == 1 || cm == -1) ? 0 : (cm > 0 ? 1 : -1);
                uint32_t facc = (uint32_t)__umul24(fs, (uint32_t)j0) + (1u << (RFIX - 1));
                int aj = base0 + __mul24(j0, cj);
                int rowj = rx0 + j0 * rj;
                const int left = dmaj - j0;
                if (whole) {
                    aj += cnt_lds;
#pragma unroll
                    for (int u = 0; u < LCH - 1; ++u) {                            // branch-free: a dead step adds nothing to a word of the lane's own
                        const bool in = u <= left;
                        fld8_add(in ? aj + __mul24((int)(facc >> RFIX), cm) : cnt_lds + 4 * lane, in ? 1u : 0u);
                        facc += fs; aj += cj;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < LCH - 1; ++u) {
                        const int m = (int)(facc >> RFIX);
                        const bool in = u <= left && (unsigned)(rowj + m * rm) < (unsigned)rows_w;
                        const int c = in ? aj + __mul24(m, cm) : 4 * lane;
                        atomicAdd(&cnt[c >> 2], in ? 1u << ((c & 3) * 8) : 0u);
                        facc += fs; aj += cj; rowj += rj;
                    }
                }
            }
        }
        BAR_LDS();
        STAMP(5);
        // ---- the 16-bit block's counts go into the window (saturated: only min(n, sat) matters for an unflagged cell) ----
        for (int mi = tid; mi < NEAR_W * NEAR_W; mi += RB) {
            const int row = x0 + mi / NEAR_W - NEAR_R - gx_base, col = y0 + mi % NEAR_W - NEAR_R - gy_base;
            if ((unsigned)row >= (unsigned)rows_w || (unsigned)col >= (unsigned)stride) continue;
            const uint32_t f = (mini[mi >> 1] >> ((mi & 1) * 16)) & 0x7FFFu;       // bit 15: claim flag of a flagged cell
            if (f) cnt8[row * stride + col] = (uint8_t)min(f, (uint32_t)sat);
        }
        BAR_LDS();
        // ---- the owners put their cell's replayed value into the field of its first source, the other sources vanish ----
        for (int pr = tid; pr < 2 * v.B; pr += RB) {
            const uint32_t ov = oval[pr];
            if (ov == 0xFFu) continue;
            FCell f;
            cell_sources(pair_cell(pr), f);
            if (f.sx < S0 || f.sx > S1) continue;
#pragma unroll
            for (int ix = 0; ix < 2; ++ix)
#pragma unroll
            for (int iy = 0; iy < 2; ++iy) {
                if (ix >= f.ngx || iy >= f.ngy) continue;
                const int sgx = ix ? f.gx1 : f.gx0, sgy = iy ? f.gy1 : f.gy0;
                cnt8[(sgx - gx_base) * stride + (sgy - gy_base)] = (ix | iy) ? (uint8_t)0 : (uint8_t)(0x80u | ov);
            }
        }
        if (tid == 0) s_wbq = 0;
        BAR_LDS();
        STAMP(6);
        // ---- write-back: one read-modify-write per touched 32-cell group of storage cells, tile by tile ----
        {
            int my_written = 0;
            const int eabs = -v.cc.emp;
            const uint32_t kb1 = (uint32_t)(128 + v.cc.vmin) * 0x01010101u;             // byte-wise: (cell ^ 0x80) - kb1 = cell - vmin
            const uint32_t oadd = (uint32_t)(127 - (v.cc.thr - v.cc.vmin)) * 0x01010101u; // bit 7 of (R + oadd) = cell > thr
            const int gpt = v.dim >> 5;                                                // 32-cell groups per tile row
            // waves draw batches of 64 items (32-cell groups) from a queue (as kernels_mapev.hip: the rows at a fan's rim hold few
            // touched groups, a fixed share per wave leaves the workgroup waiting for its slowest wave)
            auto next_batch = [&]() -> int { int g = 0; if (lane == 0) g = atomicAdd(&s_wbq, 1); return UNI(g); };
            int batch = next_batch(), batch0 = 0;
            for (int a = S0 / v.dim; a <= S1 / v.dim; ++a)
            for (int bt = T_lo / v.dim; bt <= T_hi / v.dim; ++bt) {
                if (a >= v.L || bt >= v.L || !s_need[a * v.L + bt]) continue;          // uniform
                const int tile = UNI(s_tab[a * v.L + bt]);
                if (tile < 0) continue;
                const int sr_lo = max(S0, a * v.dim), sr_hi = min(S1, (a + 1) * v.dim - 1);      // storage rows
                const int g_lo = max(T_lo >> 5, bt * gpt), g_hi = min(T_hi >> 5, (bt + 1) * gpt - 1);
                const int ngr = g_hi - g_lo + 1, items = (sr_hi - sr_lo + 1) * ngr;
                int8_t* __restrict__ tile_base = v.pool + (size_t)tile * v.dim * v.dim;
                int bx0 = INT_MAX, bx1 = -1, by0 = INT_MAX, by1 = -1;
                const int nbatch = (items + 63) >> 6;
                const float inv_ngr = 1.0f / (float)ngr;
                for (; batch < batch0 + nbatch; batch = next_batch()) {
                    const int it = ((batch - batch0) << 6) + lane;
                    if (it >= items) continue;
                    const int rr = (int)(((float)it + 0.5f) * inv_ngr), gg = it - rr * ngr;   // it / ngr (kernels_mapev.hip has the error bound)
                    const int srow = sr_lo + rr, Gy = g_lo + gg;
                    const int ia = srow - C - fxl;                                     // source rows a (if not glitched), a + 1 (if glitched)
                    const bool va = !gxb[ia], vb = gxb[ia + 1];
                    if (!va && !vb) continue;                                          // no global row maps here
                    const int lr = srow - C - gx_base;                                 // window row of source a
                    const int lc0 = 32 * Gy - C - gy_base;                             // window column of the group's first cell, multiple of 4
                    uint32_t n[8];
                    uint32_t any = 0;
                    const bool both = va && vb;
                    if (!both && !s_ggf[Gy - (T_lo >> 5)]) {   // one source row, no glitched column: the fields are the group's counts
                        const int rowo = (lr + (va ? 0 : 1)) * stride + lc0;
#pragma unroll
                        for (int w = 0; w < 8; ++w) {
                            const int lc = lc0 + 4 * w;
                            n[w] = (lc >= 0 && lc < stride) ? cnt[(rowo + 4 * w) >> 2] : 0u;
                            any |= n[w];
                        }
                    } else {
                        // glitched columns in the group (its 32 cells and the one after)
                        uint32_t gm[9];
#pragma unroll
                        for (int w = 0; w < 9; ++w) {
                            const int lc = lc0 + 4 * w;
                            gm[w] = (lc >= 0 && lc < stride + 12) ? *reinterpret_cast<const uint32_t*>(gym + lc) : 0u;
                        }
#pragma unroll
                        for (int w = 0; w < 8; ++w) n[w] = 0;
                        for (int src = 0; src < 2; ++src) {
                            if (src == 0 ? !va : !vb) continue;
                            const int row = lr + src;
                            uint32_t x[9];
#pragma unroll
                            for (int w = 0; w < 9; ++w) {
                                const int lc = lc0 + 4 * w;
                                x[w] = (lc >= 0 && lc < stride) ? cnt[(row * stride + lc) >> 2] : 0u;
                            }
#pragma unroll
                            for (int w = 0; w < 9; ++w) x[w] = premin4(x[w], satb, sadd);
#pragma unroll
                            for (int w = 0; w < 8; ++w) {
                                const uint32_t keep = x[w] & ~gm[w];
                                const uint32_t mv = ((x[w] & gm[w]) >> 8) | ((x[w + 1] & gm[w + 1]) << 24);
                                n[w] += keep + mv;
                            }
                        }
#pragma unroll
                        for (int w = 0; w < 8; ++w) any |= n[w];
                    }
                    if (!any) continue;
                    const int row_t = srow - a * v.dim, col_t = 32 * Gy - bt * v.dim;
                    uint32_t* g_ptr = reinterpret_cast<uint32_t*>(tile_base + (size_t)row_t * v.dim + col_t);
                    const uint4 q0 = reinterpret_cast<const uint4*>(g_ptr)[0], q1 = reinterpret_cast<const uint4*>(g_ptr)[1];
                    const uint32_t pre[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                    uint32_t occ = 0, touched = 0, out[8];
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        // branch-free (a word without hits passes through unchanged: dec = 0, no flag, nz = 0)
                        const uint32_t Ob = (pre[w] ^ 0x80808080u) - kb1;                   // cells biased to [0, vmax - vmin]
                        const uint32_t nw = n[w], n7 = nw & 0x7F7F7F7Fu;
                        const uint32_t ge = (n7 + sadd) & 0x80808080u;                      // fields >= sat
                        const uint32_t gem = ge | (ge - (ge >> 7));
                        const uint32_t m = (satb & gem) | (n7 & ~gem);                      // min(n, sat)
                        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                        const uint32_t dec = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, m) * (us2)(unsigned short)eabs);   // byte-wise: sat * |emp| < 128, no carries
                        const uint32_t T1 = (Ob | 0x80808080u) - dec;
                        const uint32_t pos = T1 & 0x80808080u;                              // O - dec >= 0
                        uint32_t R = T1 & 0x7F7F7F7Fu & (pos | (pos - (pos >> 7)));
                        const uint32_t fl = nw & 0x80808080u;                               // replayed cells: the field holds value - vmin
                        const uint32_t flm = fl | (fl - (fl >> 7));
                        R = (n7 & flm) | (R & ~flm);
                        out[w] = (R + kb1) ^ 0x80808080u;
                        const uint32_t nz = ((n7 + 0x7F7F7F7Fu) | nw) & 0x80808080u;        // fields that are not zero
                        touched |= __builtin_amdgcn_udot4(nz >> 7, 0x08040201u, 0u, false) << (4 * w);
                        occ |= __builtin_amdgcn_udot4(((R + oadd) & 0x80808080u) >> 7, 0x08040201u, 0u, false) << (4 * w);   // cell > thr
                    }
                    reinterpret_cast<uint4*>(g_ptr)[0] = make_uint4(out[0], out[1], out[2], out[3]);
                    reinterpret_cast<uint4*>(g_ptr)[1] = make_uint4(out[4], out[5], out[6], out[7]);
                    my_written += __popc(touched);
                    by0 = min(by0, col_t + __ffs(touched) - 1); by1 = max(by1, col_t + 31 - __clz(touched));
                    v.occ[((size_t)tile * v.dim + row_t) * v.ow + (col_t >> 5)] = occ;
                    bx0 = min(bx0, row_t); bx1 = max(bx1, row_t);
                }
                batch0 += nbatch;
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
        STAMP(7);
    }
    BAR_LDS();
    if (tid == 0) {
        if (s_cells) atomicAdd(&v.stats[ST_RAY_CELLS], s_cells);
        if (s_written) atomicAdd(&v.stats[ST_CELLS_WRITTEN], (unsigned long long)s_written);
        atomicAdd(&v.stats[ST_MAP_WINDOWS], (unsigned long long)n_win);
#ifdef RBPF_STAMPS
        for (int k = 0; k < 8; ++k) atomicAdd(&v.stats[8 + k], (unsigned long long)st_acc[k]);
#endif
    }
}

void launch_map_update_ray(const DevView& v, const int32_t* only, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {   // t0 / t1: as launch_map_update_ev
    const RayGeom g = ray_geom(v.B, v.reach);
    static size_t lds_set[MAX_DEVICES] = {};
    ensure_dynamic_lds(reinterpret_cast<const void*>(map_update_ray_kernel), (size_t)g.bytes, lds_set);
    if (t0 && t1) hipExtLaunchKernelGGL(map_update_ray_kernel, dim3(v.P), dim3(RB), (size_t)g.bytes, s, t0, t1, 0, v, only);
    else hipLaunchKernelGGL(map_update_ray_kernel, dim3(v.P), dim3(RB), (size_t)g.bytes, s, v, only);
}

}  // namespace rbpf
