// kernels_raycast.hip -- a5: HybridMap.update (hybridmap.py:95-145) for all particles.
//
// Two kernels per update:
//
//  ray_setup_kernel   one workgroup per particle: transforms the B beam endpoints
//                     (lidar.py:111-128), forms the integer start/end cells (hybridmap.py:102-113),
//                     decides exactly which lattice tiles the rays enter (the reference creates a
//                     tile when the first ray cell falls into it, hybridmap.py:124-133), allocates
//                     them from the pool, and emits one work item per 128x128-cell window of each
//                     touched tile that the ray fan's bounding box overlaps.
//
//  raycast_window_kernel   persistent workgroups pull window items from a queue.  A window is
//                     staged in LDS as 16-bit hit counters.  Rays are clipped to the window with the
//                     closed form of the reference's Bresenham, so every ray cell is walked exactly
//                     once over all windows.  The reference applies clamped adds in beam order
//                     (hybridmap.py:103, gridmap.py:86-117), which matters only for cells that
//                     receive an "occupied" or "nearby" hit in this scan.  Those cells are flagged
//                     first; hits on them are kept as (beam, rank) events in small per-cell LDS
//                     buckets and replayed in order.  All other cells only ever receive "empty" hits,
//                     which commute: max(v + n*emp, min).  A bucket that overflows falls back to an
//                     exact closed-form membership scan over all beams.  Results are bit-identical to
//                     the sequential reference on the int8 lattice.
//
// HBM traffic per window: only the 4-byte words that contain a touched cell are read and written.
#include <limits.h>

#include "rbpf_internal.h"
#include "rbpf_device.h"

namespace rbpf {

// =================================================================================================
// ray_setup_kernel
// =================================================================================================
__global__ __launch_bounds__(BLOCK) void ray_setup_kernel(DevView v, int items_cap_per_particle) {
    __shared__ double s_c, s_s, s_px, s_py;
    __shared__ int s_x0, s_y0, s_skip;
    __shared__ int s_need[49];
    __shared__ int s_tab[49];
    __shared__ int s_bb[4];                        // gx min, gx max, gy min, gy max
    __shared__ unsigned long long s_cells;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int LL = v.L * v.L;
    int32_t* tab = v.tile_tab + (size_t)v.slot[p] * LL;

    if (tid == 0) {
        double px = v.upd_pose[p], py = v.upd_pose[v.P + p], th = v.upd_pose[2 * v.P + p];
        double s, c;
        sincos(th, &s, &c);
        s_c = c; s_s = s; s_px = px; s_py = py;
        int x0 = trunc_to_int(px / v.cs), y0 = trunc_to_int(py / v.cs);      // hybridmap.py:102
        s_x0 = x0; s_y0 = y0;
        // hybridmap.py:98-100: no tile holds the robot position -> the update is a no-op
        int lx, ly;
        bool ok = tile_of_coord(px, v.tile_len, v.R, lx) && tile_of_coord(py, v.tile_len, v.R, ly);
        if (ok) ok = tab[(lx + v.R) * v.L + (ly + v.R)] >= 0;
        // the whole fan must stay inside the LUT (lattice radius)
        const int reach = v.reach;                 // rays are shorter than one tile (checked at create)
        bool in_lut = lut_valid_g(v, x0 - reach) && lut_valid_g(v, x0 + reach) &&
                      lut_valid_g(v, y0 - reach) && lut_valid_g(v, y0 + reach);
        if (ok && !in_lut) { atomicCAS(v.err, 0, RBPF_ERANGE); ok = false; }
        s_skip = ok ? 0 : 1;
        s_bb[0] = x0; s_bb[1] = x0; s_bb[2] = y0; s_bb[3] = y0;
        s_cells = 0;
        v.ray_start[2 * p] = ok ? x0 : INT_MIN;
        v.ray_start[2 * p + 1] = y0;
    }
    for (int i = tid; i < LL; i += BLOCK) { s_need[i] = 0; s_tab[i] = tab[i]; }
    __syncthreads();
    if (s_skip) return;

    const int x0 = s_x0, y0 = s_y0;
    const uint32_t e0x = lut_at(v, x0), e0y = lut_at(v, y0);
    const int a0 = lut_lat(e0x), b0 = lut_lat(e0y);
    unsigned long long my_cells = 0;
    for (int b = tid; b < v.B; b += BLOCK) {
        const double x = v.bx[b], y = v.by[b];
        double gx = (s_c * x + (-s_s) * y) + s_px;                             // lidar.py:123
        double gy = (s_s * x + s_c * y) + s_py;
        int x1 = trunc_to_int(gx / v.cs), y1 = trunc_to_int(gy / v.cs);        // hybridmap.py:106
        if (v.bflags[b] & BF_LONG) {                                           // hybridmap.py:107-113
            double sc = v.bscale[b];
            x1 = trunc_to_int((double)x0 + sc * (double)(x1 - x0));
            y1 = trunc_to_int((double)y0 + sc * (double)(y1 - y0));
        }
        int ddx = x1 - x0, ddy = y1 - y0;
        if (ddx < -32768 || ddx > 32767 || ddy < -32768 || ddy > 32767 ||
            !lut_valid_g(v, x1) || !lut_valid_g(v, y1)) {
            atomicCAS(v.err, 0, RBPF_ERANGE);
            ddx = 0; ddy = -1; x1 = x0; y1 = y0 - 1;                           // degenerate: no points
        }
        v.ray_end[(size_t)p * v.B + b] = (int32_t)(((uint32_t)ddx & 0xFFFFu) | ((uint32_t)ddy << 16));
        Ray r = ray_make(x0, y0, x1, y1);
        if (r.n == 0) continue;
        my_cells += (unsigned long long)r.n;
        atomicMin(&s_bb[0], x1); atomicMax(&s_bb[1], x1);
        atomicMin(&s_bb[2], y1); atomicMax(&s_bb[3], y1);
        // tiles entered by this ray (staircase start -> [corner] -> end)
        const int a1 = lut_lat(lut_at(v, x1)), b1 = lut_lat(lut_at(v, y1));
        s_need[a0 * v.L + b0] = 1;
        if (a1 != a0 || b1 != b0) {
            s_need[a1 * v.L + b1] = 1;
            if (a1 != a0 && b1 != b0) {
                // first global index on the far side of each boundary, in the ray's direction
                int gxb = r.sx > 0 ? lut_lower_bound(v, (uint32_t)a1 << 16) : lut_lower_bound(v, (uint32_t)a0 << 16) - 1;
                int gyb = r.sy > 0 ? lut_lower_bound(v, (uint32_t)b1 << 16) : lut_lower_bound(v, (uint32_t)b0 << 16) - 1;
                int ox = gxb - x0; ox = ox < 0 ? -ox : ox;
                int oy = gyb - y0; oy = oy < 0 ? -oy : oy;
                int jx = r.steep ? ray_first_j_minor_ge(r, ox) : ox;
                int jy = r.steep ? oy : ray_first_j_minor_ge(r, oy);
                if (jx < jy) s_need[a1 * v.L + b0] = 1;
                else if (jy < jx) s_need[a0 * v.L + b1] = 1;
            }
        }
    }
    if (my_cells) atomicAdd(&s_cells, my_cells);
    __syncthreads();

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
    if (tid == 0 && s_cells) atomicAdd(&v.stats[ST_RAY_CELLS], s_cells);

    // one work item per window of each touched tile overlapped by the fan's bounding box
    if (tid < LL && s_need[tid]) {
        const int a = tid / v.L, b = tid % v.L;
        int gxa = lut_lower_bound(v, (uint32_t)a << 16), gxb = lut_lower_bound(v, (uint32_t)(a + 1) << 16) - 1;
        int gya = lut_lower_bound(v, (uint32_t)b << 16), gyb = lut_lower_bound(v, (uint32_t)(b + 1) << 16) - 1;
        int lox = max(gxa, s_bb[0]), hix = min(gxb, s_bb[1]);
        int loy = max(gya, s_bb[2]), hiy = min(gyb, s_bb[3]);
        if (lox <= hix && loy <= hiy) {
            int wx_lo = lut_cidx(lut_at(v, lox)) / WIN, wx_hi = lut_cidx(lut_at(v, hix)) / WIN;
            int wy_lo = lut_cidx(lut_at(v, loy)) / WIN, wy_hi = lut_cidx(lut_at(v, hiy)) / WIN;
            int cnt = (wx_hi - wx_lo + 1) * (wy_hi - wy_lo + 1);
            int base = atomicAdd(v.n_items, cnt);
            if (base + cnt > v.P * items_cap_per_particle) {
                atomicCAS(v.err, 0, RBPF_ENOMEM);
            } else {
                int k = base;
                for (int wx = wx_lo; wx <= wx_hi; ++wx)
                    for (int wy = wy_lo; wy <= wy_hi; ++wy, ++k) {
                        v.items[4 * k + 0] = p;
                        v.items[4 * k + 1] = s_tab[tid];
                        v.items[4 * k + 2] = (wx * WIN) | ((wy * WIN) << 16);
                        v.items[4 * k + 3] = a | (b << 16);
                    }
            }
        }
    }
}

// =================================================================================================
// raycast_window_kernel
// =================================================================================================
struct WinShared {
    uint32_t* cnt;    // [WIN*WIN/2] two 16-bit hit counters per word; flagged cells hold their bucket id
    uint32_t* flag;   // [WIN*WIN/32]
    uint32_t* bcnt;   // [NB] events appended per bucket
    uint16_t* bev;    // [EV_TOT] (beam << 3) | rank, bucket id at [id * cap, (id + 1) * cap), cap chosen per window
    uint16_t* bcell;  // [NB] local cell index of the bucket
    uint16_t* slowc;  // [NB] local cell indices left to the wave-cooperative replay
    int16_t*  lutx;   // [WIN+8] global x offset -> local storage x
    int16_t*  luty;
};

static const int NB_MAX = 2304;
static const int EV_TOT = 10240;        // event slots per window, shared out evenly among its flagged cells
__host__ __device__ inline int raycast_nb(int B) { int nb = 2 * B; return nb < NB_MAX ? nb : NB_MAX; }

size_t raycast_lds_bytes(int B) {
    size_t nb = raycast_nb(B);
    return (size_t)WIN * WIN / 2 * 4 + (size_t)WIN * WIN / 32 * 4 + nb * 4 + (size_t)EV_TOT * 2 + nb * 2 + nb * 2 +
           2 * (WIN + 8) * 2 + 64;
}

__device__ __forceinline__ uint32_t cnt16_get(const uint32_t* cnt, int c) {
    return (cnt[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu;
}
__device__ __forceinline__ void cnt16_set(uint32_t* cnt, int c, uint32_t val) {
    // only one thread touches a given cell here; the neighbour half-word is written by atomics
    // only in other phases, so a 16-bit store is safe
    reinterpret_cast<uint16_t*>(cnt)[c] = (uint16_t)val;
}
__device__ __forceinline__ bool flag_get(const uint32_t* flag, int c) { return (flag[c >> 5] >> (c & 31)) & 1u; }

// same-tile test of hybridmap.py:141 (m.is_in_map(nearby_pos) with m = tile of the end cell)
__device__ __forceinline__ bool same_tile(const DevView& v, int xa, int ya, int xb, int yb) {
    return lut_lat(lut_at(v, xa)) == lut_lat(lut_at(v, xb)) && lut_lat(lut_at(v, ya)) == lut_lat(lut_at(v, yb));
}

// exact ordered replay for one storage cell by membership tests over all beams (slow path)
__device__ int replay_cell_slow(const DevView& v, const int32_t* __restrict__ rays, int x0, int y0,
                                const int* gxc, int ngx, const int* gyc, int ngy, int val) {
    for (int b = 0; b < v.B; ++b) {
        int x1, y1;
        unpack_end(rays[b], x0, y0, x1, y1);
        Ray r = ray_make(x0, y0, x1, y1);
        if (r.n == 0) continue;
        const bool occ = !(v.bflags[b] & BF_LONG);
        int js[4], nj = 0;
        for (int ix = 0; ix < ngx; ++ix)
            for (int iy = 0; iy < ngy; ++iy) {
                int gx = gxc[ix], gy = gyc[iy];
                int j = r.steep ? (gy - y0) * r.sy : (gx - x0) * r.sx;
                if (j < 0 || j >= r.n) continue;
                int qx, qy;
                ray_point(r, j, qx, qy);
                if (qx == gx && qy == gy) js[nj++] = j;
            }
        if (nj == 0) continue;
        for (int a = 1; a < nj; ++a) {                      // tiny insertion sort
            int key = js[a], c = a - 1;
            while (c >= 0 && js[c] > key) { js[c + 1] = js[c]; --c; }
            js[c + 1] = key;
        }
        bool near_here = false;
        for (int a = 0; a < nj; ++a) {
            int j = js[a];
            if (j == r.n - 1 && occ) val = cell_occ(val, v.cc);
            else val = cell_emp(val, v.cc);
            if (occ && r.n >= 2 && j == r.n - 2) {
                int nx, ny;
                ray_point(r, j, nx, ny);
                near_here = same_tile(v, nx, ny, x1, y1);
            }
        }
        if (near_here) val = cell_near(val, v.cc);
    }
    return val;
}

// Clamped-add functions v -> min(max(v + a, lo), hi) are closed under composition, so the ordered
// sequence of a cell's events can be folded associatively: each lane folds the events of one beam, the
// wave folds 64 beams in beam order with a shuffle tree.
struct Caf { int a, lo, hi; };
__device__ __forceinline__ Caf caf_then(Caf f, Caf g) {          // g after f
    Caf r;
    r.a = f.a + g.a;
    int lo = f.lo + g.a; lo = lo < g.lo ? g.lo : lo; r.lo = lo > g.hi ? g.hi : lo;
    int hi = f.hi + g.a; hi = hi < g.lo ? g.lo : hi; r.hi = hi > g.hi ? g.hi : hi;
    return r;
}
__device__ __forceinline__ int caf_apply(Caf f, int x) { int t = x + f.a; t = t < f.lo ? f.lo : t; return t > f.hi ? f.hi : t; }

__device__ int replay_cell_wave(const DevView& v, const int32_t* __restrict__ rays, int x0, int y0, const int* gxc,
                                int ngx, const int* gyc, int ngy, int val, int lane) {
    const int BIG = 1000000;
    const Caf fE = {v.cc.emp, v.cc.vmin, BIG}, fO = {v.cc.occ, -BIG, v.cc.vmax}, fN = {v.cc.nearby, -BIG, v.cc.vmax};
    for (int base = 0; base < v.B; base += 64) {
        const int b = base + lane;
        Caf f = {0, -BIG, BIG};
        bool has = false;
        if (b < v.B) {
            int x1, y1;
            unpack_end(rays[b], x0, y0, x1, y1);
            Ray r = ray_make(x0, y0, x1, y1);
            if (r.n > 0) {
                const bool occ = !(v.bflags[b] & BF_LONG);
                int js[4], nj = 0;
                for (int ix = 0; ix < ngx; ++ix)
                    for (int iy = 0; iy < ngy; ++iy) {
                        int gx = gxc[ix], gy = gyc[iy];
                        int j = r.steep ? (gy - y0) * r.sy : (gx - x0) * r.sx;
                        if (j < 0 || j >= r.n) continue;
                        int qx, qy;
                        ray_point(r, j, qx, qy);
                        if (qx == gx && qy == gy) js[nj++] = j;
                    }
                for (int a = 1; a < nj; ++a) {
                    int key = js[a], c = a - 1;
                    while (c >= 0 && js[c] > key) { js[c + 1] = js[c]; --c; }
                    js[c + 1] = key;
                }
                bool near_here = false;
                for (int a = 0; a < nj; ++a) {
                    int j = js[a];
                    f = caf_then(f, (j == r.n - 1 && occ) ? fO : fE);
                    if (occ && r.n >= 2 && j == r.n - 2) {
                        int nx, ny;
                        ray_point(r, j, nx, ny);
                        near_here = same_tile(v, nx, ny, x1, y1);
                    }
                }
                if (near_here) f = caf_then(f, fN);
                has = nj > 0;
            }
        }
        if (__ballot(has) == 0ull) continue;
        for (int off = 1; off < 64; off <<= 1) {
            Caf g;
            g.a = __shfl_down(f.a, off, 64); g.lo = __shfl_down(f.lo, off, 64); g.hi = __shfl_down(f.hi, off, 64);
            if ((lane & (2 * off - 1)) == 0) f = caf_then(f, g);
        }
        val = caf_apply(f, val);        // lane 0 holds the fold of the whole chunk
        val = __shfl(val, 0, 64);
    }
    return val;
}

__global__ __launch_bounds__(BLOCK) void raycast_window_kernel(DevView v, int* __restrict__ queue_head) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int NB = raycast_nb(v.B);
    WinShared s;
    s.cnt = reinterpret_cast<uint32_t*>(smem);
    s.flag = s.cnt + WIN * WIN / 2;
    s.bcnt = s.flag + WIN * WIN / 32;
    s.bev = reinterpret_cast<uint16_t*>(s.bcnt + NB);
    s.bcell = s.bev + EV_TOT;
    s.slowc = s.bcell + NB;
    s.lutx = reinterpret_cast<int16_t*>(s.slowc + NB);
    s.luty = s.lutx + (WIN + 8);
    __shared__ int s_item, s_nflag, s_g[4], s_bb[4], s_written, s_slow, s_nslow;
    const int tid = threadIdx.x;
    const size_t tile_cells = (size_t)v.dim * v.dim;

    for (;;) {
        __syncthreads();                                   // protects s_item and LDS reuse
        if (tid == 0) s_item = atomicAdd(queue_head, 1);
        __syncthreads();
        const int item = s_item;
        if (item >= *v.n_items) return;                    // uniform exit: the queue is drained

        const int p = v.items[4 * item + 0];
        const int tile = v.items[4 * item + 1];
        const int wx0 = v.items[4 * item + 2] & 0xFFFF, wy0 = (uint32_t)v.items[4 * item + 2] >> 16;
        const int la = v.items[4 * item + 3] & 0xFFFF, lb = (uint32_t)v.items[4 * item + 3] >> 16;
        const int x0 = v.ray_start[2 * p], y0 = v.ray_start[2 * p + 1];
        const int32_t* __restrict__ rays = v.ray_end + (size_t)p * v.B;

        // ---- phase 0: clear LDS, window extent in global index space -------------------------------
        for (int i = tid; i < WIN * WIN / 2; i += BLOCK) s.cnt[i] = 0;
        for (int i = tid; i < WIN * WIN / 32; i += BLOCK) s.flag[i] = 0;
        for (int i = tid; i < NB; i += BLOCK) s.bcnt[i] = 0;
        if (tid < 4) {
            int lat = tid < 2 ? la : lb;
            int w0 = tid < 2 ? wx0 : wy0;
            int c = (tid & 1) ? min(w0 + WIN, v.dim) : w0;
            // (tid&1): first global index whose storage index is >= the window's end (exclusive bound)
            uint32_t key = ((uint32_t)lat << 16) | (uint32_t)c;
            if ((tid & 1) && c >= v.dim) key = (uint32_t)(lat + 1) << 16;
            s_g[tid] = lut_lower_bound(v, key);
        }
        if (tid == 0) {
            s_nflag = 0; s_written = 0; s_slow = 0; s_nslow = 0;
            s_bb[0] = INT_MAX; s_bb[1] = -1; s_bb[2] = INT_MAX; s_bb[3] = -1;
        }
        __syncthreads();
        const int gxa = s_g[0], gxb = s_g[1], gya = s_g[2], gyb = s_g[3];   // [gxa,gxb) x [gya,gyb)
        for (int i = tid; i < gxb - gxa && i < WIN + 8; i += BLOCK) s.lutx[i] = (int16_t)(lut_cidx(lut_at(v, gxa + i)) - wx0);
        for (int i = tid; i < gyb - gya && i < WIN + 8; i += BLOCK) s.luty[i] = (int16_t)(lut_cidx(lut_at(v, gya + i)) - wy0);
        __syncthreads();

        // ---- phase 1: flag the cells that receive an "occupied" or "nearby" hit -------------------------
        for (int b = tid; b < v.B; b += BLOCK) {
            if (v.bflags[b] & BF_LONG) continue;            // end_is_occ = False (hybridmap.py:113)
            int x1, y1;
            unpack_end(rays[b], x0, y0, x1, y1);
            Ray r = ray_make(x0, y0, x1, y1);
            if (r.n == 0) continue;
            if (x1 >= gxa && x1 < gxb && y1 >= gya && y1 < gyb) {
                int c = s.lutx[x1 - gxa] * WIN + s.luty[y1 - gya];
                atomicOr(&s.flag[c >> 5], 1u << (c & 31));
            }
            if (r.n >= 2) {                                 // hybridmap.py:139-142
                int nx, ny;
                ray_point(r, r.n - 2, nx, ny);
                if (nx >= gxa && nx < gxb && ny >= gya && ny < gyb && same_tile(v, nx, ny, x1, y1)) {
                    int c = s.lutx[nx - gxa] * WIN + s.luty[ny - gya];
                    atomicOr(&s.flag[c >> 5], 1u << (c & 31));
                }
            }
        }
        __syncthreads();
        // bucket ids for flagged cells (0xFFFF = no bucket, replayed by the slow path)
        for (int w = tid; w < WIN * WIN / 32; w += BLOCK) {
            uint32_t bits = s.flag[w];
            while (bits) {
                int bit = __ffs(bits) - 1;
                bits &= bits - 1;
                int c = w * 32 + bit;
                int id = atomicAdd(&s_nflag, 1);
                if (id < NB) { cnt16_set(s.cnt, c, id); s.bcell[id] = (uint16_t)c; }
                else cnt16_set(s.cnt, c, 0xFFFFu);
            }
        }
        __syncthreads();

        // event slots are shared out evenly: few flagged cells (a near wall under dense beams) get deep buckets
        const int nflag = s_nflag;
        const int cap = min(64, max(4, EV_TOT / max(nflag, 1)));
        const int nbk = min(min(nflag, NB), EV_TOT / cap);

        // ---- phase 2: walk the clipped rays ------------------------------------------------------------
        for (int b = tid; b < v.B; b += BLOCK) {
            int x1, y1;
            unpack_end(rays[b], x0, y0, x1, y1);
            Ray r = ray_make(x0, y0, x1, y1);
            if (r.n == 0) continue;
            // clip j to the window: major axis by interval arithmetic, minor axis by the closed form
            int ma = r.steep ? gya : gxa, mb = r.steep ? gyb : gxb;      // major bounds [ma, mb)
            int na = r.steep ? gxa : gya, nb = r.steep ? gxb : gyb;      // minor bounds [na, nb)
            int m0 = r.steep ? y0 : x0, n0 = r.steep ? x0 : y0;
            int smaj = r.steep ? r.sy : r.sx, smin = r.steep ? r.sx : r.sy;
            int jlo = smaj > 0 ? ma - m0 : m0 - (mb - 1);
            int jhi = smaj > 0 ? (mb - 1) - m0 : m0 - ma;
            int olo = smin > 0 ? na - n0 : n0 - (nb - 1);                // minor offset range [olo, ohi]
            int ohi = smin > 0 ? (nb - 1) - n0 : n0 - na;
            if (ohi < 0) continue;
            jlo = max(jlo, 0); jhi = min(jhi, r.n - 1);
            if (r.dmin == 0) { if (olo > 0) continue; }
            else {
                if (olo > 0) jlo = max(jlo, ray_first_j_minor_ge(r, olo));
                jhi = min(jhi, ray_last_j_minor_le(r, ohi));
            }
            if (jlo > jhi) continue;
            const bool occ = !(v.bflags[b] & BF_LONG);
            bool near_ok = false;
            if (occ && r.n >= 2 && jlo <= r.n - 2 && jhi >= r.n - 2) {
                int nx, ny;
                ray_point(r, r.n - 2, nx, ny);
                near_ok = same_tile(v, nx, ny, x1, y1);
            }
            int m = ray_minor_at(r, jlo);
            int D = 2 * r.dmin - r.dmaj + 2 * r.dmin * jlo - 2 * r.dmaj * m;   // hybridmap.py:289-300 invariant
            for (int j = jlo; j <= jhi; ++j) {
                int maj = m0 + smaj * j, mnr = n0 + smin * m;
                int gx = r.steep ? mnr : maj, gy = r.steep ? maj : mnr;
                int ix = gx - gxa, iy = gy - gya;
                if (ix >= 0 && ix < gxb - gxa && iy >= 0 && iy < gyb - gya) {
                    int c = s.lutx[ix] * WIN + s.luty[iy];
                    if (flag_get(s.flag, c)) {
                        uint32_t id = cnt16_get(s.cnt, c);
                        if ((int)id < nbk) {
                            int rem = r.n - 1 - j;
                            int rank = (rem == 0) ? (occ ? EV_OCC : EV_E_LAST) : rem == 1 ? EV_E_2 : rem == 2 ? EV_E_3 : EV_E_FAR;
                            uint32_t pos = atomicAdd(&s.bcnt[id], 1u);
                            if ((int)pos < cap) s.bev[id * cap + pos] = (uint16_t)((b << 3) | rank);
                            if (near_ok && rem == 1) {
                                pos = atomicAdd(&s.bcnt[id], 1u);
                                if ((int)pos < cap) s.bev[id * cap + pos] = (uint16_t)((b << 3) | EV_NEAR);
                            }
                        }
                    } else {
                        atomicAdd(&s.cnt[c >> 1], 1u << ((c & 1) * 16));
                    }
                }
                if (D >= 0) { ++m; D -= 2 * r.dmaj; }
                D += 2 * r.dmin;
            }
        }
        __syncthreads();

        // ---- phase 3a: cells that only received "empty" hits: v = max(v + n*emp, min) -----------------
        int8_t* __restrict__ tile_base = v.pool + (size_t)tile * tile_cells;
        int my_written = 0;
        int bx0 = INT_MAX, bx1 = -1, by0 = INT_MAX, by1 = -1;
        for (int q = tid; q < WIN * WIN / 4; q += BLOCK) {     // 4 consecutive y cells = one 32-bit word
            int lx = q / (WIN / 4), ly = (q % (WIN / 4)) * 4;
            uint32_t w0 = s.cnt[(lx * WIN + ly) >> 1], w1 = s.cnt[((lx * WIN + ly) >> 1) + 1];
            uint32_t fl = (s.flag[(lx * WIN + ly) >> 5] >> ((lx * WIN + ly) & 31)) & 0xFu;
            uint32_t n[4] = {w0 & 0xFFFFu, w0 >> 16, w1 & 0xFFFFu, w1 >> 16};
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) { if ((fl >> k) & 1u) n[k] = 0; any |= n[k] != 0; }
            if (!any) continue;
            uint32_t* gp = reinterpret_cast<uint32_t*>(tile_base + (size_t)(wx0 + lx) * v.dim + (wy0 + ly));
            uint32_t word = *gp;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (n[k]) {
                    int val = (int)(int8_t)((word >> (8 * k)) & 0xFFu);
                    val = cell_emp_n(val, (int)n[k], v.cc);
                    word = (word & ~(0xFFu << (8 * k))) | (((uint32_t)val & 0xFFu) << (8 * k));
                    ++my_written;
                    by0 = min(by0, wy0 + ly + k); by1 = max(by1, wy0 + ly + k);
                }
            }
            *gp = word;
            bx0 = min(bx0, wx0 + lx); bx1 = max(bx1, wx0 + lx);
        }
        __syncthreads();   // flagged cells share 32-bit words with phase-3a cells: finish 3a first

        // ---- phase 3b: flagged cells, ordered replay from their buckets ----------------------------------
        for (int id = tid; id < min(nflag, NB); id += BLOCK) {
            const int c = s.bcell[id];
            const int m = id < nbk ? (int)s.bcnt[id] : INT_MAX;
            if (m > cap) { s.slowc[atomicAdd(&s_nslow, 1)] = (uint16_t)c; continue; }   // bucket overflow / no bucket
            const int lx = c / WIN, ly = c % WIN;
            int8_t* gp = tile_base + (size_t)(wx0 + lx) * v.dim + (wy0 + ly);
            int val = *gp;
            // replay in ascending (beam, rank): selection by repeated minimum, m is small
            uint32_t last = 0; bool first = true;
            for (int k = 0; k < m; ++k) {
                uint32_t best = 0xFFFFFFFFu;
                int dup = 0;
                for (int e = 0; e < m; ++e) {
                    uint32_t key = s.bev[id * cap + e];
                    if (!first && key <= last) continue;
                    if (key < best) { best = key; dup = 1; } else if (key == best) ++dup;
                }
                if (best == 0xFFFFFFFFu) break;
                for (int d = 0; d < dup; ++d) val = cell_apply_rank(val, (int)(best & 7u), v.cc);
                k += dup - 1;
                last = best; first = false;
            }
            *gp = (int8_t)val;
            ++my_written;
            bx0 = min(bx0, wx0 + lx); bx1 = max(bx1, wx0 + lx);
            by0 = min(by0, wy0 + ly); by1 = max(by1, wy0 + ly);
        }
        if (nflag > NB) {   // more flagged cells than bucket ids
            for (int w = tid; w < WIN * WIN / 32; w += BLOCK) {
                uint32_t bits = s.flag[w];
                while (bits) {
                    int bit = __ffs(bits) - 1;
                    bits &= bits - 1;
                    int c = w * 32 + bit;
                    if (cnt16_get(s.cnt, c) != 0xFFFFu) continue;
                    int k = atomicAdd(&s_nslow, 1);
                    if (k < NB) s.slowc[k] = (uint16_t)c;
                    else {      // beyond every list: one lane replays it alone (never seen in practice)
                        const int lx = c / WIN, ly = c % WIN;
                        int8_t* gp = tile_base + (size_t)(wx0 + lx) * v.dim + (wy0 + ly);
                        int gxc[4], gyc[4], ngx = 0, ngy = 0;
                        for (int i = 0; i < gxb - gxa && ngx < 4; ++i) if (s.lutx[i] == lx) gxc[ngx++] = gxa + i;
                        for (int i = 0; i < gyb - gya && ngy < 4; ++i) if (s.luty[i] == ly) gyc[ngy++] = gya + i;
                        *gp = (int8_t)replay_cell_slow(v, rays, x0, y0, gxc, ngx, gyc, ngy, (int)*gp);
                        ++my_written;
                        bx0 = min(bx0, wx0 + lx); bx1 = max(bx1, wx0 + lx);
                        by0 = min(by0, wy0 + ly); by1 = max(by1, wy0 + ly);
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 3c: cells without a usable bucket: one wave each, exact membership scan over all beams ----
        {
            const int nslow = min(s_nslow, NB);
            const int lane = tid & 63, wave = tid >> 6;
            for (int k = wave; k < nslow; k += BLOCK / 64) {
                const int c = s.slowc[k];
                const int lx = c / WIN, ly = c % WIN;
                int8_t* gp = tile_base + (size_t)(wx0 + lx) * v.dim + (wy0 + ly);
                int gxc[4], gyc[4], ngx = 0, ngy = 0;
                for (int i = 0; i < gxb - gxa && ngx < 4; ++i) if (s.lutx[i] == lx) gxc[ngx++] = gxa + i;
                for (int i = 0; i < gyb - gya && ngy < 4; ++i) if (s.luty[i] == ly) gyc[ngy++] = gya + i;
                int val = replay_cell_wave(v, rays, x0, y0, gxc, ngx, gyc, ngy, (int)*gp, lane);
                if (lane == 0) {
                    *gp = (int8_t)val;
                    ++my_written;
                    bx0 = min(bx0, wx0 + lx); bx1 = max(bx1, wx0 + lx);
                    by0 = min(by0, wy0 + ly); by1 = max(by1, wy0 + ly);
                }
            }
            if (tid == 0) s_slow = nslow;
        }
        // ---- phase 4: bounding box of written cells (bounds resample copies), counters -----------------
        if (my_written) {
            atomicAdd(&s_written, my_written);
            atomicMin(&s_bb[0], bx0); atomicMax(&s_bb[1], bx1);
            atomicMin(&s_bb[2], by0); atomicMax(&s_bb[3], by1);
        }
        __syncthreads();
        if (tid == 0 && s_written) {
            atomicAdd(&v.stats[ST_CELLS_WRITTEN], (unsigned long long)s_written);
            if (s_slow) atomicAdd(&v.stats[ST_SLOW_CELLS], (unsigned long long)s_slow);
            atomicMin(&v.tile_bbox[4 * tile + 0], s_bb[0]); atomicMax(&v.tile_bbox[4 * tile + 1], s_bb[1]);
            atomicMin(&v.tile_bbox[4 * tile + 2], s_bb[2]); atomicMax(&v.tile_bbox[4 * tile + 3], s_bb[3]);
        }
    }
}

int raycast_items_cap(const rbpf_config& cfg) {
    int reach = (int)(cfg.max_ray_m / cfg.cell_size) + 3;
    int nwin = (2 * reach + 1) / WIN + 3;    // alignment + one tile split per axis
    return nwin * nwin;
}

void launch_ray_setup(const DevView& v, hipStream_t s) {
    (void)hipMemsetAsync(v.n_items, 0, 2 * sizeof(int32_t), s);      // item count + queue head
    hipLaunchKernelGGL(ray_setup_kernel, dim3(v.P), dim3(BLOCK), 0, s, v, v.items_cap);
}

void launch_raycast_windows(const DevView& v, hipStream_t s) {
    size_t lds = raycast_lds_bytes(v.B);
    int blocks_per_cu = (int)(160 * 1024 / (lds + 256));
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu > 4) blocks_per_cu = 4;
    int grid = 256 * blocks_per_cu;
    static size_t lds_attr = 0;
    if (lds > lds_attr) {   // more than the default 64 KiB of dynamic LDS
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(raycast_window_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_attr = lds;
    }
    hipLaunchKernelGGL(raycast_window_kernel, dim3(grid), dim3(BLOCK), lds, s, v, v.n_items + 1);
}

}  // namespace rbpf
