// kernels_mapupdate.hip -- a5: HybridMap.update (hybridmap.py:95-145) for all particles, one kernel.
//
// One 512-thread workgroup per particle.
//
//  setup     transforms the B beam endpoints (lidar.py:111-128), forms the integer start/end cells
//            (hybridmap.py:102-113) and keeps them in LDS (4 B + 1 B per ray); decides exactly which
//            lattice tiles the rays enter (the reference creates a tile when the first ray cell falls
//            into it, hybridmap.py:124-133) and allocates them from the pool.
//
//  windows   the workgroup then walks the 128x128-cell windows of the touched tiles that the ray fan's
//            bounding box overlaps.  A window is staged in LDS as 16-bit hit counters.  Rays are clipped to
//            the window with the closed form of the reference's Bresenham (rbpf_math.h), so every ray
//            cell is visited exactly once over all windows.  The reference applies clamped adds in beam
//            order (hybridmap.py:103, gridmap.py:86-117); that order matters only for cells that receive an
//            "occupied" or "nearby" hit in this scan.  Those cells are flagged first; hits on them are kept
//            as (beam, rank) events in per-cell LDS buckets (the window's event slots are shared out evenly
//            among its flagged cells) and replayed in order.  All other cells only receive "empty" hits,
//            which commute: max(v + n*emp, min).  A cell whose bucket overflows is replayed by one wave
//            with an exact closed-form membership test over all beams, folded in beam order with the
//            associative composition of clamped adds.  The result is bit-identical to the sequential
//            reference on the int8 lattice.
//
// HBM traffic: per window only the 4-byte words that contain a touched cell are read and written;
// roofline = HBM (read-modify-write of the touched cells), no MFMA.
#include "rbpf_mapupdate.h"

namespace rbpf {

// Diagnostic build only (-DRBPF_STAMPS): thread 0 of every workgroup sums the cycles between phase boundaries;
// the sums go to the reserved counters and are never read by the kernel.
#ifdef RBPF_STAMPS
#define STAMP(k) do { if (tid == 0) { long long t_ = clock64(); st_acc[k] += t_ - st_prev; st_prev = t_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif


static const int MU_BLOCK = 512;       // 8 waves per particle
static const int NB_MAX = 1536;        // bucket ids per window (flagged cells beyond it take the membership-scan path)
static const int EV_TOT = 4096;        // event slots per window
__host__ __device__ inline int mu_fan_width(int reach) { return (2 * reach + 8 + 7) & ~7; }
__host__ __device__ inline int mu_nb(int B) { int nb = 2 * B; return nb < NB_MAX ? nb : NB_MAX; }
// chunk table entries: every ray has at most ceil((WIN + 2) / CHUNK) = 9 chunks in a window; the same memory later
// holds the two replay work lists (NB deep buckets + the overflow list)
__host__ __device__ inline int mu_chunk_cap(int B) { int a = 9 * B, b = 2 * mu_nb(B) + 64; int c = a > b ? a : b; return (c + 7) & ~7; }

size_t raycast_lds_bytes(int B, int reach) {
    size_t nb = mu_nb(B);
    size_t bpad = (size_t)((B + 3) & ~3);
    size_t bytes = (size_t)WIN * WIN / 2 * 4 + (size_t)WIN * WIN / 32 * 4 + (size_t)WIN * WIN / 32 * 2 + (size_t)EV_TOT * 2 + nb * 2 +
                   (size_t)mu_chunk_cap(B) * 2 + 2 * (size_t)mu_fan_width(reach) * 2 + bpad * 4 + bpad * 2 * 2 + 256 + (size_t)((B + 15) & ~15) + ((nb + 15) & ~(size_t)15);
    return (bytes + 15) & ~(size_t)15;
}


struct MuLds {
    uint32_t* cnt;    // [WIN*WIN/2] two 16-bit hit counters per word; flagged cells hold their bucket id
    uint32_t* flag;   // [WIN*WIN/32]
    uint16_t* fpre;   // [WIN*WIN/32] flagged cells before each flag word (bucket id = rank in cell order)
    uint16_t* bev;    // [EV_TOT] (beam << 3) | rank; bucket id at [id*cap, (id+1)*cap)
    uint16_t* bcell;  // [NB] local cell index of bucket id
    uint8_t*  oldv;   // [NB] value of the flagged cell before this scan (from the prefetched words)
    uint16_t* chunk;  // [CH_CAP] walk work items (beam << 3) | chunk; later the replay work lists
    uint16_t* fanx;   // [FANW] storage cell index of every global column the ray fan can reach (x axis)
    uint16_t* fany;   // [FANW] same for y; both filled once per particle from the global-index LUT
    int32_t*  r_end;  // [B] packed (dx & 0xFFFF) | (dy << 16) relative to the start cell
    int16_t*  seg_lo; // [B] first / last step of the ray inside the current window
    int16_t*  seg_hi;
    uint8_t*  r_info; // [B]
    uint32_t* dummy;  // [64] per-lane sink for the atomics of skipped steps
};

__device__ __forceinline__ uint32_t cnt16_get(const uint32_t* cnt, int c) { return (cnt[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu; }
__device__ __forceinline__ void cnt16_set(uint32_t* cnt, int c, uint32_t val) { reinterpret_cast<uint16_t*>(cnt)[c] = (uint16_t)val; }
__device__ __forceinline__ bool flag_get(const uint32_t* flag, int c) { return (flag[c >> 5] >> (c & 31)) & 1u; }


// first global index g in [lo, hi] whose storage index fan[g - f0] is >= target (hi + 1 if none), from a close guess
__device__ __forceinline__ int fan_first_ge(const uint16_t* fan, int f0, int lo, int hi, int target, int guess) {
    int g = min(max(guess, lo), hi + 1);
    while (g > lo && (int)fan[g - 1 - f0] >= target) --g;
    while (g <= hi && (int)fan[g - f0] < target) ++g;
    return g;
}

// a hit on a flagged cell: the counter value returned by the atomic is the event's slot in the cell's bucket
__device__ __forceinline__ void walk_flagged(const MuLds& s, int cc, uint32_t hv, int b, int rem, bool occ, bool near_ok,
                                             int nbk, int cap) {
    const int id = s.fpre[cc >> 5] + __popc(s.flag[cc >> 5] & ((1u << (cc & 31)) - 1u));
    const int rank = (rem == 0) ? (occ ? EV_OCC : EV_E_LAST) : rem == 1 ? EV_E_2 : rem == 2 ? EV_E_3 : EV_E_FAR;
    int pos = (int)(hv & 0x7FFFu);
    if (id < nbk && pos < cap) s.bev[id * cap + pos] = (uint16_t)((b << 3) | rank);
    if (near_ok && rem == 1) {
        const int sh = (cc & 1) * 16;
        pos = (int)((atomicAdd(&s.cnt[cc >> 1], 1u << sh) >> sh) & 0x7FFFu);
        if (id < nbk && pos < cap) s.bev[id * cap + pos] = (uint16_t)((b << 3) | EV_NEAR);
    }
}


// one particle's map update by one workgroup; `only`: this launch follows the whole-fan kernel and takes what it gave back
__device__ __forceinline__ void map_update_particle(const DevView& v, bool only) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int NB = mu_nb(v.B);
    MuLds s;
    s.cnt = reinterpret_cast<uint32_t*>(smem);
    s.flag = s.cnt + WIN * WIN / 2;
    const int CH_CAP = mu_chunk_cap(v.B);
    const int BPAD = (v.B + 3) & ~3;
    s.fpre = reinterpret_cast<uint16_t*>(s.flag + WIN * WIN / 32);
    s.bev = s.fpre + WIN * WIN / 32;
    s.bcell = s.bev + EV_TOT;
    s.chunk = s.bcell + NB;
    const int FANW = mu_fan_width(v.reach);
    s.fanx = s.chunk + CH_CAP;
    s.fany = s.fanx + FANW;
    s.r_end = reinterpret_cast<int32_t*>(s.fany + FANW);
    s.seg_lo = reinterpret_cast<int16_t*>(s.r_end + BPAD);
    s.seg_hi = s.seg_lo + BPAD;
    s.dummy = reinterpret_cast<uint32_t*>(s.seg_hi + BPAD);
    s.r_info = reinterpret_cast<uint8_t*>(s.dummy + 64);
    s.oldv = s.r_info + ((v.B + 15) & ~15);

    __shared__ double s_c, s_s, s_px, s_py;
    __shared__ int s_x0, s_y0, s_skip;
    __shared__ int s_need[49], s_tab[49];
    __shared__ int s_fan[4];                       // ray fan bounding box: gx min, gx max, gy min, gy max
    __shared__ int s_nflag, s_bb[4], s_written, s_nslow, s_nbig, s_nchunk, s_irreg, s_wsum[MU_BLOCK / 64], s_tot_written, s_tot_slow;
    static_assert(WIN * WIN / 32 == MU_BLOCK, "one flag word per thread");
    __shared__ unsigned long long s_cells;

    const int p = blockIdx.x, tid = threadIdx.x;
    const int LL = v.L * v.L;
    const int KW = (v.dim + WIN - 1) / WIN;        // windows per tile axis
    int32_t* tab = v.tile_tab + (size_t)v.slot[p] * LL;

#ifdef RBPF_STAMPS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = clock64();
#endif
    // =============================================== setup ===============================================
    if (tid == 0) {
        double px = v.upd_pose[p], py = v.upd_pose[v.P + p], th = v.upd_pose[2 * v.P + p];
        double sn, cs;
        sincos(th, &sn, &cs);
        s_c = cs; s_s = sn; s_px = px; s_py = py;
        int x0 = trunc_to_int(px / v.cs), y0 = trunc_to_int(py / v.cs);      // hybridmap.py:102
        s_x0 = x0; s_y0 = y0;
        // hybridmap.py:98-100: no tile holds the robot position -> the update is a no-op
        int lx, ly;
        bool ok = tile_of_coord(px, v.tile_len, v.R, lx) && tile_of_coord(py, v.tile_len, v.R, ly);
        if (ok) ok = tab[(lx + v.R) * v.L + (ly + v.R)] >= 0;
        const int reach = v.reach;                 // the whole fan must stay inside the LUT (lattice radius)
        bool in_lut = lut_valid_g(v, x0 - reach) && lut_valid_g(v, x0 + reach) &&
                      lut_valid_g(v, y0 - reach) && lut_valid_g(v, y0 + reach);
        if (ok && !in_lut) { atomicCAS(v.err, 0, RBPF_ERANGE); ok = false; }
        s_skip = ok ? 0 : 1;
        s_fan[0] = x0; s_fan[1] = x0; s_fan[2] = y0; s_fan[3] = y0;
        s_cells = 0; s_tot_written = 0; s_tot_slow = 0;
    }
    for (int i = tid; i < LL; i += MU_BLOCK) { s_need[i] = 0; s_tab[i] = tab[i]; }
    __syncthreads();
    if (s_skip) return;

    const int x0 = UNI(s_x0), y0 = UNI(s_y0);
    const int a0 = lut_lat(lut_at(v, x0)), b0 = lut_lat(lut_at(v, y0));
    {
        unsigned long long my_cells = 0;
        int fx0 = x0, fx1 = x0, fy0 = y0, fy1 = y0;
        for (int b = tid; b < v.B; b += MU_BLOCK) {
            const double x = v.bx[b], y = v.by[b];
            const int bf = v.bflags[b];
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
            s.r_end[b] = (int32_t)(((uint32_t)ddx & 0xFFFFu) | ((uint32_t)ddy << 16));
            Ray r = ray_make(x0, y0, x1, y1);
            int info = 0;
            if (r.n > 0) {
                info = RI_VALID | ((bf & BF_LONG) ? 0 : RI_OCC);
                my_cells += (unsigned long long)r.n;
                fx0 = min(fx0, x1); fx1 = max(fx1, x1); fy0 = min(fy0, y1); fy1 = max(fy1, y1);
                if (r.n >= 2 && (info & RI_OCC)) {                                 // hybridmap.py:139-142
                    int nx, ny;
                    ray_point(r, r.n - 2, nx, ny);
                    if (same_tile(v, nx, ny, x1, y1)) info |= RI_NEAR;
                    info |= ((nx - x1 + 1) & 3) << 3;
                    info |= ((ny - y1 + 1) & 3) << 5;
                }
                // tiles entered by this ray (staircase start -> [corner] -> end)
                const int a1 = lut_lat(lut_at(v, x1)), b1 = lut_lat(lut_at(v, y1));
                s_need[a0 * v.L + b0] = 1;
                if (a1 != a0 || b1 != b0) {
                    s_need[a1 * v.L + b1] = 1;
                    if (a1 != a0 && b1 != b0) {
                        // first global index on the far side of each boundary, in the ray's direction
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
            s.r_info[b] = (uint8_t)info;
        }
        {
            const int ws = wave_sum((int)my_cells);                      // < 64 * 16 rays * 2^16 steps
            fx0 = wave_min(fx0); fx1 = wave_max(fx1); fy0 = wave_min(fy0); fy1 = wave_max(fy1);
            if ((tid & 63) == 0) {
                atomicAdd(&s_cells, (unsigned long long)ws);
                atomicMin(&s_fan[0], fx0); atomicMax(&s_fan[1], fx1);
                atomicMin(&s_fan[2], fy0); atomicMax(&s_fan[3], fy1);
            }
        }
    }
    __syncthreads();
    // the storage index of every global column of the fan, once per particle (the window loop never reads the LUT again)
    const int fx0 = UNI(s_fan[0] - 1), fy0g = UNI(s_fan[2] - 1);
    for (int i = tid; i < FANW; i += MU_BLOCK) {
        const int gxq = fx0 + i, gyq = fy0g + i;
        s.fanx[i] = lut_valid_g(v, gxq) ? (uint16_t)lut_cidx(lut_at(v, gxq)) : 0xFFFFu;
        s.fany[i] = lut_valid_g(v, gyq) ? (uint16_t)lut_cidx(lut_at(v, gyq)) : 0xFFFFu;
    }
    // allocate missing tiles (free tiles are kept zero-filled)
    if (tid < LL && s_need[tid] && s_tab[tid] < 0) {
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
    __syncthreads();

    STAMP(0);
    // ============================================ window loop ==============================================
    const size_t tile_cells = (size_t)v.dim * v.dim;
    const int lane = tid & 63, wave = tid >> 6;
    const int sat = (v.cc.vmax - v.cc.vmin + (-v.cc.emp) - 1) / (-v.cc.emp);     // hits that saturate any cell: 20
    for (int t = 0; t < LL; ++t) {
        if (!s_need[t]) continue;                          // uniform over the workgroup
        const int la = t / v.L, lb = t % v.L;
        const int tile = UNI(s_tab[t]);
        const int* gwx = v.gwin + la * (KW + 1);
        const int* gwy = v.gwin + lb * (KW + 1);
        const int lox = UNI(max(gwx[0], s_fan[0])), hix = UNI(min(gwx[KW] - 1, s_fan[1]));
        const int loy = UNI(max(gwy[0], s_fan[2])), hiy = UNI(min(gwy[KW] - 1, s_fan[3]));
        if (lox > hix || loy > hiy) continue;
        // windows are placed relative to the fan: x from the first touched storage column, y from the first touched
        // 32-cell group (the write-back and the occupancy words are 32 cells wide)
        const int cx_lo = UNI(s.fanx[lox - fx0]), cx_hi = UNI(s.fanx[hix - fx0]);
        const int cy_lo = UNI(s.fany[loy - fy0g] & ~31), cy_hi = UNI(s.fany[hiy - fy0g]);
        int8_t* __restrict__ tile_base = v.pool + (size_t)tile * tile_cells;
        int tile_bb[4] = {INT_MAX, -1, INT_MAX, -1};       // thread 0 accumulates the tile's written box

        for (int wx0 = cx_lo; wx0 <= cx_hi; wx0 += WIN)
        for (int wy0 = cy_lo; wy0 <= cy_hi; wy0 += WIN) {
            // global index range [gxa,gxb) x [gya,gyb) of the window: the storage index is non-decreasing in the global one
            // (the index map is the identity plus an offset up to isolated off-by-one glitches: start from that guess)
            const int gxa = UNI(fan_first_ge(s.fanx, fx0, lox, hix, wx0, lox + (wx0 - cx_lo)));
            const int gxb = UNI(fan_first_ge(s.fanx, fx0, lox, hix, wx0 + WIN, lox + (wx0 + WIN - cx_lo)));
            const int gya = UNI(fan_first_ge(s.fany, fy0g, loy, hiy, wy0, loy + (wy0 - (int)s.fany[loy - fy0g])));
            const int gyb = UNI(fan_first_ge(s.fany, fy0g, loy, hiy, wy0 + WIN, loy + (wy0 + WIN - (int)s.fany[loy - fy0g])));
            if (gxa >= gxb || gya >= gyb) continue;        // uniform
            const int nx_ = gxb - gxa, ny_ = gyb - gya;
            const uint16_t* lutx = s.fanx + (gxa - fx0);   // storage column of global column gxa + i: lutx[i] - wx0 is window-local
            const uint16_t* luty = s.fany + (gya - fy0g);
            const int offx = UNI((int)lutx[0] - wx0), offy = UNI((int)luty[0] - wy0);
            if (tid == 0) s_irreg = 0;
            BAR_LDS();                                     // previous window fully done with LDS
            // ---- phase 0: clear the counters; is the window's index map a pure offset?  (No off-by-one glitch of the
            //      reference's index formula inside: always the case on the positive side of a tile, SURVEY quirk 3.
            //      A repeat and a skip can cancel, so every step is checked.)
            for (int i = tid; i < nx_ - 1; i += MU_BLOCK) if ((int)lutx[i + 1] - (int)lutx[i] != 1) s_irreg = 1;
            for (int i = tid; i < ny_ - 1; i += MU_BLOCK) if ((int)luty[i + 1] - (int)luty[i] != 1) s_irreg = 1;
            // ---- clear the counters ----------------------------------------------------
            {
                uint4* c4 = reinterpret_cast<uint4*>(s.cnt);
                for (int i = tid; i < WIN * WIN / 8; i += MU_BLOCK) c4[i] = make_uint4(0, 0, 0, 0);
                for (int i = tid; i < WIN * WIN / 32; i += MU_BLOCK) s.flag[i] = 0;
                if (tid == 0) {
                    s_nflag = 0; s_written = 0; s_nslow = 0; s_nbig = 0; s_nchunk = 0;
                    s_bb[0] = INT_MAX; s_bb[1] = -1; s_bb[2] = INT_MAX; s_bb[3] = -1;
                }
            }
            BAR_LDS();
            STAMP(1);

            // ---- phase 1: flag the cells that receive an "occupied" or "nearby" hit; clip every ray to the window ----
            for (int b0 = 0; b0 < v.B; b0 += MU_BLOCK) {          // wave-uniform trip count (the loop body uses shuffles)
                const int b = b0 + tid;
                const int info = b < v.B ? s.r_info[b] : 0;
                int jlo = 1, jhi = 0;
                if (info & RI_VALID) {
                    int x1, y1;
                    unpack_end(s.r_end[b], x0, y0, x1, y1);
                    if ((info & RI_OCC)) {                                              // hybridmap.py:113,137
                        if (x1 >= gxa && x1 < gxb && y1 >= gya && y1 < gyb) {
                            int c = ((int)lutx[x1 - gxa] - wx0) * WIN + ((int)luty[y1 - gya] - wy0);
                            atomicOr(&s.flag[c >> 5], 1u << (c & 31));
                            atomicOr(&s.cnt[c >> 1], 0x8000u << ((c & 1) * 16));
                        }
                        if (info & RI_NEAR) {                                         // hybridmap.py:139-142
                            int nx = x1 + ((info >> 3) & 3) - 1, ny = y1 + ((info >> 5) & 3) - 1;
                            if (nx >= gxa && nx < gxb && ny >= gya && ny < gyb) {
                                int c = ((int)lutx[nx - gxa] - wx0) * WIN + ((int)luty[ny - gya] - wy0);
                                atomicOr(&s.flag[c >> 5], 1u << (c & 31));
                                atomicOr(&s.cnt[c >> 1], 0x8000u << ((c & 1) * 16));
                            }
                        }
                    }
                    // clip j to the window: major axis by interval arithmetic, minor axis by the closed form
                    if (!(max(x0, x1) < gxa || min(x0, x1) >= gxb || max(y0, y1) < gya || min(y0, y1) >= gyb)) {
                        Ray r = ray_make(x0, y0, x1, y1);
                        int ma = r.steep ? gya : gxa, mb = r.steep ? gyb : gxb;      // major bounds [ma, mb)
                        int na = r.steep ? gxa : gya, nb = r.steep ? gxb : gyb;      // minor bounds [na, nb)
                        int m0 = r.steep ? y0 : x0, n0 = r.steep ? x0 : y0;
                        int smaj = r.steep ? r.sy : r.sx, smin = r.steep ? r.sx : r.sy;
                        jlo = smaj > 0 ? ma - m0 : m0 - (mb - 1);
                        jhi = smaj > 0 ? (mb - 1) - m0 : m0 - ma;
                        int olo = smin > 0 ? na - n0 : n0 - (nb - 1);                // minor offset range [olo, ohi]
                        int ohi = smin > 0 ? (nb - 1) - n0 : n0 - na;
                        jlo = max(jlo, 0); jhi = min(jhi, r.n - 1);
                        if (ohi < 0) jhi = -1;
                        else if (r.dmin == 0) { if (olo > 0) jhi = -1; }
                        else {
                            if (olo > 0) jlo = max(jlo, first_j_minor_ge(r, olo));
                            jhi = min(jhi, last_j_minor_le(r, ohi));
                        }
                    }
                }
                // chunks of up to CHUNK steps, one table entry each: (beam << 4) | chunk index; the table space of a
                // wave is claimed with one atomic
                const int nch = jlo <= jhi ? (jhi - jlo) / CHUNK + 1 : 0;
                const int wpre = wave_excl_scan(nch, lane);
                int wbase = 0;
                if (lane == 63) wbase = atomicAdd(&s_nchunk, wpre + nch);
                wbase = __shfl(wbase, 63, 64);
                if (nch) {
                    s.seg_lo[b] = (int16_t)jlo; s.seg_hi[b] = (int16_t)jhi;
                    const int base = wbase + wpre;
                    for (int k = 0; k < nch; ++k) if (base + k < CH_CAP) s.chunk[base + k] = (uint16_t)((b << 4) | k);
                }
            }
            BAR_LDS();
            STAMP(7);
            // rank of every flagged cell among the window's flagged cells = its bucket id (cell order)
            {
                const int w = tid;                          // one flag word per thread (WIN*WIN/32 == MU_BLOCK)
                const uint32_t bits = s.flag[w];
                int pc = __popc(bits), incl = pc;
                for (int off = 1; off < 64; off <<= 1) { int n = __shfl_up(incl, off, 64); if (lane >= off) incl += n; }
                if (lane == 63) s_wsum[wave] = incl;
                BAR_LDS();
                int wbase = 0;
                for (int k = 0; k < wave; ++k) wbase += s_wsum[k];
                const int excl = wbase + incl - pc;
                s.fpre[w] = (uint16_t)excl;
                if (tid == MU_BLOCK - 1) s_nflag = excl + pc;
                uint32_t bb = bits; int id = excl;
                while (bb) {                                  // flag word w covers exactly this thread's prefetched group
                    int bit = __ffs(bb) - 1; bb &= bb - 1;
                    if (id < NB) s.bcell[id] = (uint16_t)(w * 32 + bit);
                    ++id;
                }
            }
            BAR_LDS();
            // event slots are shared out evenly: few flagged cells (a near wall under dense beams) get deep buckets
            const int nflag = UNI(s_nflag);
            const int cap = min(64, max(4, EV_TOT / max(nflag, 1)));
            const int nbk = min(min(nflag, NB), EV_TOT / cap);
            STAMP(2);

            // ---- phase 2: walk the clipped rays, CHUNK steps per work item, four steps in flight ------------------------
            {
                const int nchunk = UNI(min(s_nchunk, CH_CAP));
                const bool ident = UNI(s_irreg) == 0;
                if (s_nchunk > CH_CAP && tid == 0) atomicCAS(v.err, 0, RBPF_ENOMEM);   // cannot happen: B*5 entries
                for (int q = tid; q < nchunk; q += MU_BLOCK) {
                    const int desc = s.chunk[q];
                    const int b = desc >> 4;
                    const int info = s.r_info[b];
                    int x1, y1;
                    unpack_end(s.r_end[b], x0, y0, x1, y1);
                    const Ray r = ray_make(x0, y0, x1, y1);
                    const int jlo = s.seg_lo[b] + (desc & 15) * CHUNK;
                    const int jhi = min((int)s.seg_hi[b], jlo + CHUNK - 1);
                    const int m0 = r.steep ? y0 : x0, n0 = r.steep ? x0 : y0;
                    const int smaj = r.steep ? r.sy : r.sx, smin = r.steep ? r.sx : r.sy;
                    const bool occ = info & RI_OCC;
                    const bool near_ok = info & RI_NEAR;
                    int m = ray_minor_at(r, jlo);
                    int D = 2 * r.dmin - r.dmaj + 2 * r.dmin * jlo - 2 * r.dmaj * m;   // hybridmap.py:289-300 invariant
                    if (ident) {
                        // identity window: the local cell index advances by constants, no index-map reads
                        const int lx0 = (r.steep ? n0 + smin * m : m0 + smaj * jlo) - gxa + offx;
                        const int ly0 = (r.steep ? m0 + smaj * jlo : n0 + smin * m) - gya + offy;
                        int c = lx0 * WIN + ly0;
                        const int dmajc = smaj * (r.steep ? 1 : WIN), dminc = smin * (r.steep ? WIN : 1);
                        const bool check_sat = (desc & 15) < 2 && s.seg_lo[b] < 2 * CHUNK;    // only near the sensor
                        int j = jlo;
                        for (; j + 3 <= jhi; j += 4) {
                            int cc[4]; uint32_t h[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                cc[u] = c;
                                if (D >= 0) { c += dminc; D -= 2 * r.dmaj; }
                                D += 2 * r.dmin; c += dmajc;
                            }
                            if (check_sat) {
#pragma unroll
                                for (int u = 0; u < 4; ++u) h[u] = reinterpret_cast<const uint16_t*>(s.cnt)[cc[u]];
                            }
                            uint32_t* ap[4]; uint32_t av[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const bool skip = check_sat && (h[u] - (uint32_t)sat < 0x8000u - (uint32_t)sat);
                                ap[u] = skip ? &s.dummy[lane] : &s.cnt[cc[u] >> 1];
                                av[u] = skip ? 0u : 1u << ((cc[u] & 1) * 16);
                                if (skip) cc[u] = -1;
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) h[u] = atomicAdd(ap[u], av[u]);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (cc[u] < 0) continue;
                                const uint32_t hv = (h[u] >> ((cc[u] & 1) * 16)) & 0xFFFFu;
                                if (hv & 0x8000u) walk_flagged(s, cc[u], hv, b, r.n - 1 - (j + u), occ, near_ok, nbk, cap);
                            }
                        }
                        for (; j <= jhi; ++j) {
                            const int c1 = c;
                            if (D >= 0) { c += dminc; D -= 2 * r.dmaj; }
                            D += 2 * r.dmin; c += dmajc;
                            const int sh = (c1 & 1) * 16;
                            const uint32_t hv = (atomicAdd(&s.cnt[c1 >> 1], 1u << sh) >> sh) & 0xFFFFu;
                            if (hv & 0x8000u) walk_flagged(s, c1, hv, b, r.n - 1 - j, occ, near_ok, nbk, cap);
                        }
                        continue;
                    }
                    for (int j4 = jlo; j4 <= jhi; j4 += 4) {
                        // Branch-free on purpose: four cells' index maps, counters and atomics are issued back to back
                        // (behind a branch the compiler drains the LDS queue after every access).  Steps past the
                        // chunk end repeat the last cell and add 0 to a per-lane dummy word.
                        int c[4]; uint32_t h[4]; int lxv[4], lyv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int j = j4 + u;
                            const bool live = j <= jhi;
                            const int maj = m0 + smaj * (live ? j : jhi), mnr = n0 + smin * m;
                            int ix = (r.steep ? mnr : maj) - gxa, iy = (r.steep ? maj : mnr) - gya;
                            if (live) { if (D >= 0) { ++m; D -= 2 * r.dmaj; } D += 2 * r.dmin; }
                            ix = min(max(ix, 0), nx_ - 1); iy = min(max(iy, 0), ny_ - 1);
                            lxv[u] = (int)lutx[ix] - wx0; lyv[u] = (int)luty[iy] - wy0;
                            c[u] = live ? 0 : -1;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int cc = lxv[u] * WIN + lyv[u];
                            h[u] = (s.cnt[cc >> 1] >> ((cc & 1) * 16)) & 0xFFFFu;
                            c[u] = c[u] < 0 ? -1 : cc;
                        }
                        uint32_t* ap[4]; uint32_t av[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            // an unflagged cell only needs min(n, sat) hits: max(v + n*emp, vmin) is vmin for every
                            // n >= sat; skipping spares the serialised same-address atomics of the cells next to the
                            // sensor, which every ray crosses
                            const bool need = c[u] >= 0 && !(h[u] - (uint32_t)sat < 0x8000u - (uint32_t)sat);
                            ap[u] = need ? &s.cnt[c[u] >> 1] : &s.dummy[lane];
                            av[u] = need ? 1u << ((c[u] & 1) * 16) : 0u;
                            if (!need) c[u] = -1;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) h[u] = atomicAdd(ap[u], av[u]);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (c[u] < 0) continue;
                            const uint32_t hv = (h[u] >> ((c[u] & 1) * 16)) & 0xFFFFu;
                            if (!(hv & 0x8000u)) continue;    // flagged: the counter value is the slot in the cell's bucket
                            const int cc = c[u], j = j4 + u;
                            const int id = s.fpre[cc >> 5] + __popc(s.flag[cc >> 5] & ((1u << (cc & 31)) - 1u));
                            const int rem = r.n - 1 - j;
                            const int rank = (rem == 0) ? (occ ? EV_OCC : EV_E_LAST) : rem == 1 ? EV_E_2 : rem == 2 ? EV_E_3 : EV_E_FAR;
                            int pos = (int)(hv & 0x7FFFu);
                            if (id < nbk && pos < cap) s.bev[id * cap + pos] = (uint16_t)((b << 3) | rank);
                            if (near_ok && rem == 1) {
                                const int sh = (cc & 1) * 16;
                                pos = (int)((atomicAdd(&s.cnt[cc >> 1], 1u << sh) >> sh) & 0x7FFFu);
                                if (id < nbk && pos < cap) s.bev[id * cap + pos] = (uint16_t)((b << 3) | EV_NEAR);
                            }
                        }
                    }
                }
            }
            BAR_LDS();
            STAMP(3);
            // prefetch this thread's 32-cell group of the window (128 rows x 4 groups = 512 threads): issued
            // after the walk (holding eight more registers across it makes the compiler spill them to scratch)
            const int g_lx = tid / (WIN / 32), g_ly = (tid % (WIN / 32)) * 32;
            const int g_row = wx0 + g_lx, g_col = wy0 + g_ly;
            const bool g_in = g_row < v.dim && g_col < v.dim;
            const int g_nw = g_in ? min(8, (v.dim - g_col) >> 2) : 0;    // words inside the row (dim is a multiple of 16)
            uint32_t* const g_ptr = reinterpret_cast<uint32_t*>(tile_base + (size_t)(g_in ? g_row : 0) * v.dim + (g_in ? g_col : 0));
            uint32_t pre[8];
            if (g_nw == 8) {
                const uint4 a0 = reinterpret_cast<const uint4*>(g_ptr)[0], a1 = reinterpret_cast<const uint4*>(g_ptr)[1];
                pre[0] = a0.x; pre[1] = a0.y; pre[2] = a0.z; pre[3] = a0.w; pre[4] = a1.x; pre[5] = a1.y; pre[6] = a1.z; pre[7] = a1.w;
            } else {
#pragma unroll
                for (int w = 0; w < 8; ++w) pre[w] = w < g_nw ? g_ptr[w] : 0u;
            }
            {   // values of this group's flagged cells before the scan, for the replay (flag word `tid` = this group)
                uint32_t bb = s.flag[tid]; int id = s.fpre[tid];
                while (bb) {
                    const int bit = __ffs(bb) - 1; bb &= bb - 1;
                    if (id < NB) {
                        uint32_t wsel = pre[0];
#pragma unroll
                        for (int q = 1; q < 8; ++q) wsel = (bit >> 2) == q ? pre[q] : wsel;
                        s.oldv[id] = (uint8_t)((wsel >> (8 * (bit & 3))) & 0xFFu);
                    }
                    ++id;
                }
            }
            BAR_LDS();

            // ---- phase 3: flagged cells, ordered replay; the new value goes back into the cell's LDS slot ----------------
            // (the chunk table is dead now: its memory holds the two work lists of this phase)
            uint16_t* bigc = s.chunk;                       // buckets of 17..cap events: folded by one wave each
            uint16_t* slowc = s.chunk + NB;                 // bucket overflow / no bucket: exact membership scan
            for (int id = tid; id < min(nflag, NB); id += MU_BLOCK) {
                const int c = s.bcell[id];
                const int m = id < nbk ? (int)(cnt16_get(s.cnt, c) & 0x7FFFu) : INT_MAX;
                if (m > cap) { slowc[atomicAdd(&s_nslow, 1)] = (uint16_t)c; continue; }
                if (m > 16) { bigc[atomicAdd(&s_nbig, 1)] = (uint16_t)id; continue; }
                int val = (int)(int8_t)s.oldv[id];
                // replay in ascending (beam, rank): bitonic sorting network over registers, then a sequential fold
                const uint16_t* evp = s.bev + id * cap;
                if (m <= 8) val = replay_sorted<8>(evp, m, val, v.cc);
                else val = replay_sorted<16>(evp, m, val, v.cc);
                cnt16_set(s.cnt, c, 0x8000u | ((uint32_t)val & 0xFFu));
            }
            if (nflag > NB) {   // more flagged cells than bucket ids: the rest is replayed by membership scan
                for (int id = NB + tid; id < nflag; id += MU_BLOCK) {
                    // the id-th flagged cell in cell order: find its word by the prefix counts
                    int lo = 0, hi = WIN * WIN / 32 - 1;
                    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if ((int)s.fpre[mid] <= id) lo = mid; else hi = mid - 1; }
                    uint32_t bb = s.flag[lo]; int k = id - s.fpre[lo];
                    while (k--) bb &= bb - 1;
                    const int c = lo * 32 + __ffs(bb) - 1;
                    int pos = atomicAdd(&s_nslow, 1);
                    if (pos < CH_CAP - NB) slowc[pos] = (uint16_t)c; else atomicCAS(v.err, 0, RBPF_ENOMEM);
                }
            }
            BAR_LDS();
            {
                const int nbig = s_nbig;
                for (int k = wave; k < nbig; k += MU_BLOCK / 64) {
                    const int id = bigc[k];
                    const int c = s.bcell[id];
                    const int m = (int)(cnt16_get(s.cnt, c) & 0x7FFFu);            // 17..64 events, one per lane
                    const uint32_t key = lane < m ? (uint32_t)s.bev[id * cap + lane] : 0xFFFFFFFFu;
                    int rank = 0;
                    for (int e = 0; e < m; ++e) {
                        const uint32_t ke = __shfl(key, e, 64);
                        rank += (ke < key) || (ke == key && e < lane);
                    }
                    // move every event to the lane of its rank, then fold the clamped adds in lane order
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
                        int val = caf_apply(f, (int)(int8_t)s.oldv[id]);
                        cnt16_set(s.cnt, c, 0x8000u | ((uint32_t)val & 0xFFu));
                    }
                }
                const int nslow = min(s_nslow, CH_CAP - NB);
                for (int k = wave; k < nslow; k += MU_BLOCK / 64) {
                    const int c = slowc[k];
                    const int lx = c / WIN, ly = c % WIN;
                    int gxc[4], gyc[4], ngx = 0, ngy = 0;
                    for (int i = 0; i < nx_ && ngx < 4; ++i) if ((int)lutx[i] - wx0 == lx) gxc[ngx++] = gxa + i;
                    for (int i = 0; i < ny_ && ngy < 4; ++i) if ((int)luty[i] - wy0 == ly) gyc[ngy++] = gya + i;
                    int val = replay_cell_wave(v, s.r_info, s.r_end, x0, y0, gxc, ngx, gyc, ngy,
                                               (int)tile_base[(size_t)(wx0 + lx) * v.dim + (wy0 + ly)], lane);
                    if (lane == 0) cnt16_set(s.cnt, c, 0x8000u | ((uint32_t)val & 0xFFu));
                }
                if (tid == 0 && nslow) s_tot_slow += nslow;
            }
            BAR_LDS();
            STAMP(4);

            // ---- phase 4: one read-modify-write per touched 32-cell group (one group per thread) ----------------------
            //      unflagged cell: v = max(v + n*emp, min) (gridmap.py:97-101, n times); flagged cell: the replayed value.
            //      The group's word of the tile's occupancy bitmask (cell > threshold, gridmap.py:153; read by the scan
            //      matcher) is rebuilt from the 32 new values.
            int my_written = 0;
            int bx0 = INT_MAX, bx1 = -1, by0 = INT_MAX, by1 = -1;
            {
                const int lx = g_lx, ly = g_ly;
                const uint4* c4 = reinterpret_cast<const uint4*>(s.cnt + ((lx * WIN + ly) >> 1));
                uint32_t n[16];
                {
                    const uint4 q0 = c4[0], q1 = c4[1], q2 = c4[2], q3 = c4[3];
                    n[0] = q0.x; n[1] = q0.y; n[2] = q0.z; n[3] = q0.w; n[4] = q1.x; n[5] = q1.y; n[6] = q1.z; n[7] = q1.w;
                    n[8] = q2.x; n[9] = q2.y; n[10] = q2.z; n[11] = q2.w; n[12] = q3.x; n[13] = q3.y; n[14] = q3.z; n[15] = q3.w;
                }
                uint32_t any = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) any |= n[k];
                const int row = g_row, col = g_col;
                if (any && g_in) {
                    uint32_t occ = 0;
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        uint32_t word = pre[w];
                        const uint32_t cw0 = n[2 * w], cw1 = n[2 * w + 1];
                        if (cw0 | cw1) {
                            const uint32_t nn[4] = {cw0 & 0xFFFFu, cw0 >> 16, cw1 & 0xFFFFu, cw1 >> 16};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                if (nn[k]) {
                                    int val;
                                    if (nn[k] & 0x8000u) val = (int)(int8_t)(nn[k] & 0xFFu);
                                    else val = cell_emp_n((int)(int8_t)((word >> (8 * k)) & 0xFFu), (int)nn[k], v.cc);
                                    word = (word & ~(0xFFu << (8 * k))) | (((uint32_t)val & 0xFFu) << (8 * k));
                                    ++my_written;
                                    by0 = min(by0, col + 4 * w + k); by1 = max(by1, col + 4 * w + k);
                                }
                            }
                            if (w < g_nw) g_ptr[w] = word;
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            occ |= ((int)(int8_t)((word >> (8 * k)) & 0xFFu) > v.cc.thr ? 1u : 0u) << (4 * w + k);
                    }
                    v.occ[((size_t)tile * v.dim + row) * v.ow + (col >> 5)] = occ;
                    bx0 = row; bx1 = row;
                }
            }
            STAMP(5);
            {
                const int ww = wave_sum(my_written);
                bx0 = wave_min(bx0); bx1 = wave_max(bx1); by0 = wave_min(by0); by1 = wave_max(by1);
                if (lane == 0 && ww) {
                    atomicAdd(&s_written, ww);
                    atomicMin(&s_bb[0], bx0); atomicMax(&s_bb[1], bx1);
                    atomicMin(&s_bb[2], by0); atomicMax(&s_bb[3], by1);
                }
            }
            BAR_LDS();
            STAMP(6);
            if (tid == 0 && s_written) {
                s_tot_written += s_written;
                tile_bb[0] = min(tile_bb[0], s_bb[0]); tile_bb[1] = max(tile_bb[1], s_bb[1]);
                tile_bb[2] = min(tile_bb[2], s_bb[2]); tile_bb[3] = max(tile_bb[3], s_bb[3]);
            }
        }
        if (tid == 0 && tile_bb[1] >= 0) {          // this workgroup is the tile's only writer
            v.tile_bbox[4 * tile + 0] = min(v.tile_bbox[4 * tile + 0], tile_bb[0]);
            v.tile_bbox[4 * tile + 1] = max(v.tile_bbox[4 * tile + 1], tile_bb[1]);
            v.tile_bbox[4 * tile + 2] = min(v.tile_bbox[4 * tile + 2], tile_bb[2]);
            v.tile_bbox[4 * tile + 3] = max(v.tile_bbox[4 * tile + 3], tile_bb[3]);
        }
    }
    if (tid == 0) {
        if (only) atomicAdd(&v.stats[ST_WINDOW_FALLBACKS], 1ull);
        if (s_cells) atomicAdd(&v.stats[ST_RAY_CELLS], s_cells);
        if (s_tot_written) atomicAdd(&v.stats[ST_CELLS_WRITTEN], (unsigned long long)s_tot_written);
        if (s_tot_slow) atomicAdd(&v.stats[ST_SLOW_CELLS], (unsigned long long)s_tot_slow);
#ifdef RBPF_STAMPS
        for (int k = 0; k < 8; ++k) atomicAdd(&v.stats[8 + k], (unsigned long long)st_acc[k]);
#endif
    }
}

// only != nullptr: process just the particles the whole-fan kernel gave back (only[p] != 0).  bad != nullptr: particles
// on the NaN-covariance branch (robot.py:73-78) get their weight increment here, after their map is updated (by this
// workgroup or, earlier, by the whole-fan kernel): one launch less per step.
__global__ __launch_bounds__(MU_BLOCK, 4) void map_update_kernel(DevView v, const int32_t* __restrict__ only,
                                                                 const uint8_t* __restrict__ bad) {   // 2 workgroups per CU
    if (!only || only[blockIdx.x]) map_update_particle(v, only != nullptr);
    if (bad && bad[blockIdx.x]) {
        __syncthreads();
        nan_branch_weight(v, blockIdx.x, threadIdx.x, MU_BLOCK);
    }
}

// the first kernel of the chain for this engine: 0 none (128x128 windows for every particle), 1 the event walk, 2 the global-index kernel
int map_update_first_kernel(const DevView& v) {
    // Default first kernel: the event walk (kernels_mapev.hip) on grids of 0.04 m and coarser, where a fan usually fits its one LDS
    // window; on finer grids every fan takes three or four strips, and there the global-index kernel of round 2 (kernels_mapray.hip:
    // no returning adds, a leaner strip set-up) is a quarter faster (8192 x 181 beams x 0.025 m: 0.82 against 1.02 ms per 2048).
    const bool ev_ok = map_update_ev_available(v), ray_ok = map_update_ray_available(v);
    const bool ev_first = v.mu_mode == 5 || (v.mu_mode == 0 && (v.dim <= 1024 || !ray_ok));
    if (ev_first && ev_ok) return 1;
    if ((v.mu_mode == 0 || v.mu_mode == 3) && ray_ok) return 2;
    if (v.mu_mode == 0 && ev_ok) return 1;
    return 0;
}

// t0 / t1 (or nullptr): timing events for the FIRST kernel's own start and end (ignored when there is none)
void launch_map_update_fused(const DevView& v, const uint8_t* d_bad, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    size_t lds = raycast_lds_bytes(v.B, v.reach);
    static size_t lds_set[MAX_DEVICES] = {};   // more than the default 64 KiB of dynamic LDS
    ensure_dynamic_lds(reinterpret_cast<const void*>(map_update_kernel), lds, lds_set);
    // the chain: the first kernel leaves mu_fallback[p] != 0 for the particles it could not hold, the window kernel takes those
    const int first = map_update_first_kernel(v);
    if (first == 1) launch_map_update_ev(v, s, t0, t1);
    else if (first == 2) launch_map_update_ray(v, nullptr, s, t0, t1);
    hipLaunchKernelGGL(map_update_kernel, dim3(v.P), dim3(MU_BLOCK), lds, s, v, first ? (const int32_t*)v.mu_fallback : (const int32_t*)nullptr, d_bad);
}

}  // namespace rbpf
