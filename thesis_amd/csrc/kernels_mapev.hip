// kernels_mapev.hip -- a5: HybridMap.update (hybridmap.py:95-145): the event-detecting walk (round 3).
//
// One 1024-thread workgroup per particle, LDS window of 8-bit hit fields in GLOBAL cell-index space (as
// kernels_mapray.hip: a ray step is pure arithmetic, the index map is folded in by the write-back).  Same exact
// semantics as the other map kernels (kernels_mapupdate.hip has the ordered-replay argument).  What is different:
//
//   * The cells that receive an "occupied" / "nearby" hit in this scan (the only ones whose clamped adds do not commute)
//     carry a FLAG BIT in their field before the walk starts; every add of the walk RETURNS the old field, and a step that
//     sees the flag appends (beam, step) to an event list.  No slope buckets, no sort, no gather: a scan of 1081 beams in a
//     room leaves ~130 such passes per particle.
//   * Unflagged cells are written back from their counts: max(v + n * emp, min).  The flagged cells get a value from their
//     counts too; their real value follows.
//   * After the write-back the window's LDS is free: the flagged cells are grouped by storage cell through a hash table,
//     each listed pass is counted into the INTERVAL between two occupied / nearby events that its beam falls into - all
//     unoccupied passes are the same clamped add, so only their number between consecutive occupied / nearby events (in
//     beam order) matters - and one lane per flagged cell folds emp^n0 . ev0 . emp^n1 . ev1 ... from the cell's old value
//     and stores the byte (and its occupancy bit).
//   * A beam's own last step (occupied) and the step before it (the pass that precedes its own "nearby" hit) are not
//     walked at all: they are known without looking (the fold adds that pass in front of the beam's first event).
//   * Any partition of the ray steps may be written back on its own (clamped adds of one sign compose), so a fan larger than
//     the window is processed in strips of rows, each with its own read-modify-write.
//   * dim need not be a multiple of 32 (0.1 m cells: dim 400): the last 32-cell group of a tile row is partial.
//
//   Reference: hybridmap.py:95-145 (update), :274-301 (Bresenham), gridmap.py:86-117 (clamped adds).
#include <hip/hip_ext.h>

#include "rbpf_mapupdate.h"

namespace rbpf {

#ifdef RBPF_STAMPS
#define STAMP(k) do { if (tid == 0) { long long t_ = clock64(); st_acc[k] += t_ - st_prev; st_prev = t_; } } while (0)
#ifdef EV_BARRIER_WAITS      // diagnostic: cycles every wave spends at the workgroup's barriers (printed by workgroup 0)
#undef BAR_LDS
#define BAR_LDS() do { const long long t0_ = clock64(); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); { const long long d_ = clock64() - t0_; bar_wait += d_; if (bar_n < 12) bar_w[bar_n] = (int)d_; } ++bar_n; } while (0)
#endif
#else
#define STAMP(k) do { } while (0)
#endif

static const int EB = 1024;                    // threads per particle
static const int NEAR_R = 16;                  // ray steps j < NEAR_R are counted in the 16-bit block round the start cell
static const int LCH = 16;                     // steps per chunk of the walk beyond it
static const int NEAR_W = 2 * NEAR_R + 1;
static const int NBIN = 256;                   // slope buckets per direction class (counter bound)
static const int NB_WIN = NBIN / NEAR_R + 2;   // buckets that can hold the rays through one cell beyond the 16-bit block
static const int HIT_BOUND = 62;               // per direction class; two classes can meet in a cell, +1 for the flag's count bit: 125 < 128
static const int MAXLEV = 63;                  // whole 16-step chunks per ray (reach < 1000 cells)
static const int EVCAP = 3072;                 // passes over flagged cells kept per particle (more: exact replay of every flagged cell)
static const int NPAIR = 3;                    // pairs (beam, e) per thread kept in registers: 3 * EB pairs = 1536 beams

__host__ __device__ inline int ev_al16(int x) { return (x + 15) & ~15; }

struct EvGeom {
    int fanw, bpad, ncell, T, logT, E;
    int o_mini, o_fs, o_end, o_nE, o_perm, o_kl, o_info, o_ux, o_uy, o_gxb, o_gyb, o_gym, o_evl, o_cnt;
    int p_keys, p_cnta, p_offs, p_oldv, p_rlist, p_evl, p_ic, p_bytes;              // the window's LDS after the write-back
    int bytes;
    bool ok;
};
__host__ __device__ inline EvGeom ev_geom(int B, int reach) {
    EvGeom g;
    g.fanw = (2 * reach + 8 + 7) & ~7;
    g.bpad = (B + 3) & ~3;
    int o = 0;
    g.o_mini = o;  o += ev_al16(((NEAR_W * NEAR_W + 1) / 2) * 4);
    g.o_fs = o;    o += ev_al16(g.bpad * 4);
    g.o_end = o;   o += ev_al16(g.bpad * 4);
    g.o_nE = o;    o += ev_al16(g.bpad * 2);
    g.o_perm = o;  o += ev_al16(g.bpad * 2);
    g.o_kl = o;    o += ev_al16(g.bpad * 2);
    g.o_info = o;  o += ev_al16(g.bpad);
    g.o_ux = o;    o += ev_al16(g.fanw * 2);
    g.o_uy = o;    o += ev_al16(g.fanw * 2);
    g.o_gxb = o;   o += ev_al16(g.fanw);
    g.o_gyb = o;   o += ev_al16(g.fanw);
    g.o_gym = o;   o += ev_al16(g.fanw + 16);
    g.o_evl = o;   o += EVCAP * 4;
    g.o_cnt = o;
    const int avail = 160 * 1024 - 2560 - o - 64;      // 2.5 KB for the kernel's static LDS
    g.ncell = avail > 0 ? avail & ~127 : 0;
    g.bytes = o + g.ncell;
    // after the write-back: hash table over the flagged storage cells (at most 2 B of them)
    int T = 1024, lt = 10;
    while (T < 3 * g.bpad) { T <<= 1; ++lt; }
    g.T = T; g.logT = lt;
    g.E = NPAIR * EB;                                  // pairs (beam, e)
    int q = 0;
    g.p_keys = q;  q += T * 4;                         // storage cell of the slot
    g.p_cnta = q;  q += T * 4;                         // head of the slot's list of pairs
    g.p_offs = q;  q += T * 2;                         // passes after the cell's last event
    g.p_oldv = q;  q += T;
    g.p_rlist = q; q += T * 2;
    g.p_evl = q;   q += ev_al16(g.E * 2) * 2;          // next pair of the list; the lists laid out for the fold
    g.p_ic = q;    q += ev_al16(g.E * 2);              // passes right before the pair's event
    g.p_bytes = q;
    g.ok = g.ncell >= 24576 && g.p_bytes <= g.ncell && 2 * 8 * NBIN * 2 <= g.ncell && 2 * B <= NPAIR * EB && reach >= NEAR_R + 4 && reach < 1000;
    return g;
}

bool map_update_ev_available(const DevView& v) {
    const int sat = (v.cc.vmax - v.cc.vmin + (-v.cc.emp) - 1) / (-v.cc.emp);
    const EvGeom g = ev_geom(v.B, v.reach);
    const int gpt = (v.dim + 31) >> 5;
    return g.ok && v.dim % 8 == 0 && 3 * gpt <= 192 && v.L * v.L <= 49 && v.cc.emp < 0 && sat <= 31 &&
           v.cc.vmax - v.cc.vmin <= 127 && v.cc.vmin <= 0 && v.cc.vmax >= 0 && v.cc.vmin >= -127 && sat * -v.cc.emp <= 127 &&
           v.cc.thr >= v.cc.vmin && v.cc.thr < v.cc.vmax;
}

// int(x / cell_size) (hybridmap.py:102,106) without the division when the product with the reciprocal is safely inside
// a cell: x / c and x * (1 / c) differ by a few units in the last place, so they truncate alike unless an integer lies
// within 1e-9 of the product; the division decides the rest.
__device__ __forceinline__ int ev_cell_of(double x, double cs, double inv_cs) {
    const double q = x * inv_cs;
    const double t = __builtin_trunc(q);
    const double f = q - t, af = f < 0 ? -f : f;
    if (af > 1e-9 && af < 1.0 - 1e-9) return (int)t;
    return trunc_to_int(x / cs);
}

// 32-bit fixed-point slope: ceil(dmin * 2^32 / dmaj), the diagonal clamped to 2^32 - 1.  With it
//   minor(j) = (slope * j + 2^31) >> 32  ==  (2 * dmin * j + dmaj) / (2 * dmaj)     (hybridmap.py:286-300 in closed form)
// for every j <= dmaj as long as 2 * j * dmaj < 2^32: the slope errs upwards by less than 2^-32 per step and a value
// (2 dmin j + dmaj) / (2 dmaj) that is not an integer lies at least 1 / (2 dmaj) below the next one.
__device__ __forceinline__ uint32_t ev_fix_slope(int dmin, int dmaj) {
    if (dmaj <= 0 || dmin <= 0) return 0u;
    if (dmin >= dmaj) return 0xFFFFFFFFu;
    const unsigned long long num = (unsigned long long)(unsigned)dmin << 32;
    unsigned long long q = (unsigned long long)((double)num / (double)dmaj);      // within one of the quotient
    long long r = (long long)num - (long long)(q * (unsigned)dmaj);
    if (r < 0) { --q; r += dmaj; }
    if (r >= dmaj) { ++q; r -= dmaj; }
    return (uint32_t)(q + (r > 0 ? 1u : 0u));
}
__device__ __forceinline__ int ev_minor(uint32_t fs, int j) {
    return (int)(((unsigned long long)fs * (unsigned)j + 0x80000000ull) >> 32);
}

typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ int ev_lds_addr(const void* p) { return (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p; }
__device__ __forceinline__ uint32_t ev_lds_add_rtn(int byte_addr, uint32_t val) {
    return __hip_atomic_fetch_add((lds_u32*)(uintptr_t)(uint32_t)byte_addr, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// byte-wise min(x, sat) of four 7-bit counts
__device__ __forceinline__ uint32_t ev_min4(uint32_t n7, uint32_t satb, uint32_t sadd) {
    const uint32_t ge = (n7 + sadd) & 0x80808080u;
    const uint32_t gem = ge | (ge - (ge >> 7));
    return (satb & gem) | (n7 & ~gem);
}

// One word of four cells through the write-back's arithmetic (gridmap.py:97-101, n times, byte-wise): returns the new word;
// `touched` / `occ` get the word's four bits (field not zero / cell > threshold).
struct EvWb { uint32_t kb1, oadd, satb, sadd; int eabs; };
__device__ __forceinline__ uint32_t ev_wb_word(const EvWb& k, uint32_t pre, uint32_t n7, uint32_t& touched4, uint32_t& occ4) {
    const uint32_t Ob = (pre ^ 0x80808080u) - k.kb1;                          // cells biased to [0, vmax - vmin]
    const uint32_t m = ev_min4(n7, k.satb, k.sadd);                           // min(n, sat)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const uint32_t dec = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, m) * (us2)(unsigned short)k.eabs);   // byte-wise: sat * |emp| < 128, no carries
    const uint32_t T1 = (Ob | 0x80808080u) - dec;
    const uint32_t pos = T1 & 0x80808080u;                                    // O - dec >= 0
    const uint32_t R = T1 & 0x7F7F7F7Fu & (pos | (pos - (pos >> 7)));
    const uint32_t nz = (n7 + 0x7F7F7F7Fu) & 0x80808080u;                     // fields that are not zero
    touched4 = __builtin_amdgcn_udot4(nz >> 7, 0x08040201u, 0u, false);
    occ4 = __builtin_amdgcn_udot4(((R + k.oadd) & 0x80808080u) >> 7, 0x08040201u, 0u, false);   // cell > thr
    return (R + k.kb1) ^ 0x80808080u;
}

// open addressing, linear probing, keys never ~0
__device__ __forceinline__ int ev_hash_insert(uint32_t* keys, int T, int logT, uint32_t sc) {
    uint32_t h = (sc * 2654435761u) >> (32 - logT);
    for (;;) {
        const uint32_t old = atomicCAS(&keys[h], 0xFFFFFFFFu, sc);
        if (old == 0xFFFFFFFFu || old == sc) return (int)h;
        h = (h + 1) & (uint32_t)(T - 1);                                         // (the table has more slots than there can be keys)
    }
}
__device__ __forceinline__ int ev_hash_insert2(uint32_t* keys, int T, int logT, uint32_t sc, bool& created) {
    uint32_t h = (sc * 2654435761u) >> (32 - logT);
    for (;;) {
        const uint32_t old = atomicCAS(&keys[h], 0xFFFFFFFFu, sc);
        created = old == 0xFFFFFFFFu;
        if (created || old == sc) return (int)h;
        h = (h + 1) & (uint32_t)(T - 1);
    }
}
__device__ __forceinline__ int ev_hash_find(const uint32_t* keys, int T, int logT, uint32_t sc) {
    uint32_t h = (sc * 2654435761u) >> (32 - logT);
    for (int it = 0; it < T; ++it) {
        const uint32_t k = keys[h];
        if (k == sc) return (int)h;
        if (k == 0xFFFFFFFFu) return -1;
        h = (h + 1) & (uint32_t)(T - 1);
    }
    return -1;
}

// hand the particle to the window kernel (uniform over the workgroup; nothing has been written to the map yet);
// reason codes: 1 geometry / index map, 2 counter bound
#define EV_GIVE_BACK(reason) do { if (tid == 0) { v.mu_fallback[p] = (reason); atomicAdd(&v.stats[(reason) == 1 ? ST_FALLBACK_REASONS : ST_FB_BOUND], 1ull); } return; } while (0)

__global__ __launch_bounds__(EB) void map_update_ev_kernel(DevView v) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const EvGeom G = ev_geom(v.B, v.reach);
    uint32_t* const cnt = reinterpret_cast<uint32_t*>(smem + G.o_cnt);     // 8-bit fields: bit 0 = flagged, bits 1-7 = passes; [row = global x][col = global y]
    uint8_t*  const cnt8 = smem + G.o_cnt;
    uint32_t* const mini = reinterpret_cast<uint32_t*>(smem + G.o_mini);   // [NEAR_W^2] 16-bit fields round the start cell, same format
    uint32_t* const r_fs = reinterpret_cast<uint32_t*>(smem + G.o_fs);     // [B] 32-bit fixed-point slope
    int32_t*  const r_end = reinterpret_cast<int32_t*>(smem + G.o_end);    // [B] packed end cell relative to the start
    uint16_t* const r_nE = reinterpret_cast<uint16_t*>(smem + G.o_nE);     // [B] steps the walk takes
    uint16_t* const perm = reinterpret_cast<uint16_t*>(smem + G.o_perm);   // rays ordered by falling count of whole chunks
    uint8_t*  const r_info = smem + G.o_info;                              // [B]
    uint16_t* const r_kl = reinterpret_cast<uint16_t*>(smem + G.o_kl);     // [B] first whole chunk of the ray's share of the level walk | chunks << 8 (a strip: those inside it)
    uint16_t* const ux = reinterpret_cast<uint16_t*>(smem + G.o_ux);       // U of global column fxl + i
    uint16_t* const uy = reinterpret_cast<uint16_t*>(smem + G.o_uy);
    uint8_t*  const gxb = smem + G.o_gxb;                                  // G of global column fxl + i (0 / 1)
    uint8_t*  const gyb = smem + G.o_gyb;
    uint8_t*  const gym = smem + G.o_gym;                                  // G of window column lc as a byte mask (0 / 0xFF)
    uint32_t* const evlist = reinterpret_cast<uint32_t*>(smem + G.o_evl);  // [EVCAP] beam << 10 | step: passes over flagged cells

    __shared__ int s_fb, s_exact;
    __shared__ int s_need[49], s_tab[49];
    __shared__ int s_fan[4];
    __shared__ int s_wsum[EB / 64], s_wsum2[EB / 64];
    __shared__ int s_lcnt[MAXLEV + 1], s_lfill[MAXLEV + 1], s_nk[MAXLEV + 2], s_lp[MAXLEV + 3], s_nlev;
    __shared__ int s_nev, s_written, s_wbq;
    __shared__ unsigned long long s_cells;
    __shared__ double s_sincos[2];
    __shared__ uint8_t s_ggf[192];                    // per (tile column, 32-column group): a glitched column among its 33

    const int LL = v.L * v.L;
    const int KW = (v.dim + WIN - 1) / WIN;
    int32_t* tab = v.tile_tab + (size_t)v.slot[p] * LL;
#ifdef RBPF_STAMPS
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = clock64();
    long long bar_wait = 0; int bar_n = 0; const long long t_begin = clock64(); int bar_w[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // =============================================== setup ===============================================
    // this thread's beams: one per thread (two for the first few)
    const int pb0 = tid, pb1 = tid + EB;
    const double pre_x0 = pb0 < v.B ? v.bx[pb0] : 0.0, pre_y0 = pb0 < v.B ? v.by[pb0] : 0.0, pre_s0 = pb0 < v.B ? v.bscale[pb0] : 0.0;
    const double pre_x1 = pb1 < v.B ? v.bx[pb1] : 0.0, pre_y1 = pb1 < v.B ? v.by[pb1] : 0.0, pre_s1 = pb1 < v.B ? v.bscale[pb1] : 0.0;
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
        const bool in_lut = lut_valid_g(v, x0 - v.reach - 2) && lut_valid_g(v, x0 + v.reach + 2) &&
                            lut_valid_g(v, y0 - v.reach - 2) && lut_valid_g(v, y0 + v.reach + 2);
        if (ok && !in_lut) { if (tid == 0) atomicCAS(v.err, 0, RBPF_ERANGE); ok = false; }
        if (tid == 0) v.mu_fallback[p] = 0;
        if (!UNI(ok)) return;
    }
    // the index map over everything a ray can reach, with a margin of two columns (the sources of a storage cell are its
    // own global index and the next one)
    const int fxl = x0 - v.reach - 2, fyl = y0 - v.reach - 2, nfx = 2 * v.reach + 5;
    for (int i = tid; i < G.fanw; i += EB) {
        const int gxq = fxl + i, gyq = fyl + i;
        const uint32_t ex = lut_valid_g(v, gxq) ? lut_at(v, gxq) : LUT_INVALID, ey = lut_valid_g(v, gyq) ? lut_at(v, gyq) : LUT_INVALID;
        ux[i] = ex != LUT_INVALID ? (uint16_t)(lut_lat(ex) * v.dim + lut_cidx(ex)) : 0xFFFFu;
        uy[i] = ey != LUT_INVALID ? (uint16_t)(lut_lat(ey) * v.dim + lut_cidx(ey)) : 0xFFFFu;
    }
    uint16_t* const s_bins = reinterpret_cast<uint16_t*>(cnt);               // [8 NBIN] rays per (class, slope bucket) (the window is not in use yet)
    uint16_t* const s_far = s_bins + 8 * NBIN;                               // ... of the rays that reach the 8-bit fields
    if (tid == 0) {
        s_fan[0] = x0; s_fan[1] = x0; s_fan[2] = y0; s_fan[3] = y0;
        s_cells = 0; s_fb = 0; s_exact = 0; s_nev = 0; s_written = 0;
    }
    for (int i = tid; i < LL; i += EB) { s_need[i] = 0; s_tab[i] = tab[i]; }
    for (int i = tid; i < 8 * NBIN; i += EB) reinterpret_cast<uint32_t*>(s_bins)[i] = 0;     // both bucket arrays
    if (tid < 192) s_ggf[tid] = 0;
    if (tid <= MAXLEV) s_lcnt[tid] = 0;
    for (int i = tid; i < (NEAR_W * NEAR_W + 1) / 2; i += EB) mini[i] = 0;
    __syncthreads();
    STAMP(0);
    const double s_s = s_sincos[0], s_c = s_sincos[1];

    const int C = v.R * v.dim + v.dim / 2;
    const int Uxs = UNI(ux[x0 - fxl]), Uys = UNI(uy[y0 - fyl]);
    const int a0 = Uxs / v.dim, b0 = Uys / v.dim;
    auto lat_x = [&](int g) { const int U = ux[g - fxl]; return a0 + (U >= (a0 + 1) * v.dim ? 1 : 0) - (U < a0 * v.dim ? 1 : 0); };   // (rays are shorter than a tile)
    auto lat_y = [&](int g) { const int U = uy[g - fyl]; return b0 + (U >= (b0 + 1) * v.dim ? 1 : 0) - (U < b0 * v.dim ? 1 : 0); };
    {
        unsigned long long my_cells = 0;
        int fx0 = x0, fx1 = x0, fy0 = y0, fy1 = y0;
        const double inv_cs = 1.0 / v.cs;
        for (int b = tid; b < v.B; b += EB) {
            const double x = b == pb0 ? pre_x0 : b == pb1 ? pre_x1 : v.bx[b], y = b == pb0 ? pre_y0 : b == pb1 ? pre_y1 : v.by[b];
            const int bf = b == pb0 ? pre_f0 : b == pb1 ? pre_f1 : (int)v.bflags[b];
            double gx = (s_c * x + (-s_s) * y) + s_px;                             // lidar.py:123
            double gy = (s_s * x + s_c * y) + s_py;
            int x1 = ev_cell_of(gx, v.cs, inv_cs), y1 = ev_cell_of(gy, v.cs, inv_cs);   // hybridmap.py:106
            if (bf & BF_LONG) {                                                    // hybridmap.py:107-113
                const double sc = b == pb0 ? pre_s0 : b == pb1 ? pre_s1 : v.bscale[b];
                x1 = trunc_to_int((double)x0 + sc * (double)(x1 - x0));
                y1 = trunc_to_int((double)y0 + sc * (double)(y1 - y0));
            }
            int ddx = x1 - x0, ddy = y1 - y0;
            if (ddx < -v.reach || ddx > v.reach || ddy < -v.reach || ddy > v.reach) {
                atomicCAS(v.err, 0, RBPF_ERANGE);
                ddx = 0; ddy = -1; x1 = x0; y1 = y0 - 1;                           // degenerate: no points
            }
            Ray r = ray_make(x0, y0, x1, y1);
            int info = 0, nE = 0;
            uint32_t fs = 0;
            if (r.n > 0) {
                info = RI_VALID | ((bf & BF_LONG) ? 0 : RI_OCC);
                my_cells += (unsigned long long)r.n;
                fx0 = min(fx0, x1); fx1 = max(fx1, x1); fy0 = min(fy0, y1); fy1 = max(fy1, y1);
                fs = ev_fix_slope(r.dmin, r.dmaj);
                const int a1 = lat_x(x1), b1 = lat_y(y1);
                if (r.n >= 2 && (info & RI_OCC)) {                                 // hybridmap.py:139-142
                    const int jn = r.n - 2, mn = ev_minor(fs, jn);
                    const int nx = r.steep ? x0 + r.sx * mn : x0 + r.sx * jn, ny = r.steep ? y0 + r.sy * jn : y0 + r.sy * mn;
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
                // steps the walk takes: all of them but the beam's own occupied step and the pass before its own nearby hit
                nE = r.n - ((info & RI_OCC) ? 1 : 0) - ((info & RI_NEAR) ? 1 : 0);
                const int cls = (r.steep ? 4 : 0) | (ddx > 0 ? 2 : 0) | (ddy > 0 ? 1 : 0);
                const int key = cls * NBIN + (int)(fs >> 24);
                atomicAdd(reinterpret_cast<unsigned int*>(s_bins) + (key >> 1), 1u << ((key & 1) * 16));
                if (nE > NEAR_R) atomicAdd(reinterpret_cast<unsigned int*>(s_far) + (key >> 1), 1u << ((key & 1) * 16));   // only these reach the 8-bit fields
                const int nfull = (nE - NEAR_R) / LCH;                               // whole chunks beyond the 16-bit block
                if (nE > NEAR_R && nfull >= 1) atomicAdd(&s_lcnt[min(nfull, MAXLEV)], 1);
            }
            r_fs[b] = fs;
            r_end[b] = (int32_t)(((uint32_t)ddx & 0xFFFFu) | ((uint32_t)ddy << 16));
            r_nE[b] = (uint16_t)nE;
            r_info[b] = (uint8_t)info;
            r_kl[b] = (uint16_t)(1 | ((nE > NEAR_R ? (nE - NEAR_R) / LCH : 0) << 8));
        }
        const int ws = wave_sum((int)my_cells);
        fx0 = wave_min(fx0); fx1 = wave_max(fx1); fy0 = wave_min(fy0); fy1 = wave_max(fy1);
        if (lane == 0) {
            atomicAdd(&s_cells, (unsigned long long)ws);
            atomicMin(&s_fan[0], fx0); atomicMax(&s_fan[1], fx1);
            atomicMin(&s_fan[2], fy0); atomicMax(&s_fan[3], fy1);
        }
    }
    __syncthreads();
    // ---- the window: the fan's bounding box in global cell indices; strips of storage rows if it does not fit ----
    const int bxl = UNI(s_fan[0]), bxh = UNI(s_fan[1]), byl = UNI(s_fan[2]), byh = UNI(s_fan[3]);
    // the reference's index formula over the fan (one column more on either side): U(g) = g + C - G(g) with G in {0, 1}
    for (int i = tid; i < G.fanw; i += EB) {
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
    const int gpt = (v.dim + 31) >> 5;                                        // 32-cell groups per tile row (the last one may be partial)
    const int bt_lo = T_lo / v.dim;
    if (wave == 0) {   // levels: N_k = rays with at least k whole chunks (suffix sums over the wave: MAXLEV = 63)
        const int k = lane;                                                    // lane 0 is unused (level 0 = the 16-bit block)
        const int ck = k >= 1 ? s_lcnt[k] : 0;
        int suf = ck;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(suf, o, 64); if (lane + o < 64) suf += t; }
        const int nwk = k >= 1 ? (suf + 63) >> 6 : 0;
        int pre = nwk;                                                         // inclusive prefix of the levels' wave counts
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pre, o, 64); if (lane >= o) pre += t; }
        const unsigned long long live = __ballot(k >= 1 && suf > 0);
        const int nlev = live ? 63 - __clzll((long long)live) : 0;
        if (k >= 1) { s_lfill[k] = suf - ck; s_nk[k] = suf; s_lp[k] = pre - nwk; }
        if (k == 63) s_lp[64] = pre;
        if (k == 0) { s_nlev = nlev; s_nk[MAXLEV + 1] = 0; }
    }
    {   // no 8-bit field can overflow: a cell at major distance j >= NEAR_R is hit, per direction class, only by rays
        // whose slope lies in a window of width 2^32 / j + 1, i.e. in at most NB_WIN consecutive buckets: bounded with all
        // rays of those buckets, or (the smaller of the two) with the rays long enough to reach an 8-bit field.
        // inclusive prefix sums over the 2048 (class, bucket) counts, two per thread
        const int c0 = s_bins[2 * tid], c1 = s_bins[2 * tid + 1], f0 = s_far[2 * tid], f1 = s_far[2 * tid + 1];
        int incl = c0 + c1, incl2 = f0 + f1;
        for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(incl, o, 64), n2 = __shfl_up(incl2, o, 64); if (lane >= o) { incl += n; incl2 += n2; } }
        if (lane == 63) { s_wsum[wave] = incl; s_wsum2[wave] = incl2; }
        __syncthreads();
        int base = incl - (c0 + c1), base2 = incl2 - (f0 + f1);
        for (int k = 0; k < wave; ++k) { base += s_wsum[k]; base2 += s_wsum2[k]; }
        s_bins[2 * tid] = (uint16_t)(base + c0); s_bins[2 * tid + 1] = (uint16_t)(base + c0 + c1);
        s_far[2 * tid] = (uint16_t)(base2 + f0); s_far[2 * tid + 1] = (uint16_t)(base2 + f0 + f1);
        __syncthreads();
        int mx2 = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = 2 * tid + i, cls = key / NBIN, bin = key % NBIN;
            const int hi = cls * NBIN + min(bin + NB_WIN - 1, NBIN - 1);
            mx2 = max(mx2, min((int)s_bins[hi] - (key ? (int)s_bins[key - 1] : 0), (int)s_far[hi] - (key ? (int)s_far[key - 1] : 0)));
        }
        mx2 = wave_max(mx2);
        if (lane == 0 && mx2 > HIT_BOUND) s_exact = 1;
    }
    // rays ordered by falling count of whole chunks
    for (int b = tid; b < v.B; b += EB) {
        const int nE = r_nE[b];
        const int nfull = (nE - NEAR_R) / LCH;
        if (nE > NEAR_R && nfull >= 1) perm[atomicAdd(&s_lfill[min(nfull, MAXLEV)], 1)] = (uint16_t)b;
    }
    for (int lc = tid; lc < stride + 16 && lc < G.fanw + 16; lc += EB) {       // column glitch mask in window coordinates
        const int i = lc + gy_base - fyl;
        const bool gl = i >= 0 && i < nfx && gyb[i];
        gym[lc] = gl ? 0xFFu : 0u;
        if (gl) {   // the write-back's groups that see this column: its own and, for a group's first four columns, the one before
            const int sc = lc + gy_base + C, bt = sc / v.dim, t = sc - bt * v.dim, gt = t >> 5;
            const int idx = (bt - bt_lo) * gpt + gt;
            if ((unsigned)idx < 192u) s_ggf[idx] = 1;
            if ((t & 31) < 4 && (unsigned)(idx - 1) < 192u) s_ggf[idx - 1] = 1;   // (gt = 0: the last group of the tile before)
        }
    }
    __syncthreads();
    if (UNI(s_fb) || rows_cap < 8) { EV_GIVE_BACK(1); }
    if (UNI(s_exact)) { EV_GIVE_BACK(2); }
    if (tid < LL && s_need[tid] && s_tab[tid] < 0) {                          // allocate missing tiles (kept zero-filled)
        int idx = atomicSub(v.free_top, 1) - 1;
        if (idx < 0) {
            atomicAdd(v.free_top, 1);
            atomicCAS(v.err, 0, RBPF_ENOMEM);
        } else {
            int t = v.free_stack[idx];
            s_tab[tid] = t;                                                    // (a new tile's cells are zero: an old value read through either state of the table is 0)
            tab[tid] = t;
            v.tile_bbox[4 * t + 0] = INT_MAX; v.tile_bbox[4 * t + 1] = -1;
            v.tile_bbox[4 * t + 2] = INT_MAX; v.tile_bbox[4 * t + 3] = -1;
        }
    }
    STAMP(1);

    // the storage cell (U_x << 16 | U_y) flagged by pair (beam, e): the beam's end cell (e = 0) or the cell before it (e = 1,
    // only when it lies in the end cell's tile); ~0 = none
    auto pair_cell = [&](int pr) -> uint32_t {
        const int b = pr >> 1, info = r_info[b];
        if ((info & (RI_VALID | RI_OCC)) != (RI_VALID | RI_OCC) || ((pr & 1) && !(info & RI_NEAR))) return 0xFFFFFFFFu;
        const int32_t re = r_end[b];
        int x1 = x0 + (int)(int16_t)(re & 0xFFFF), y1 = y0 + (int)(int16_t)((uint32_t)re >> 16);
        if (pr & 1) { x1 += ((info >> 3) & 3) - 1; y1 += ((info >> 5) & 3) - 1; }
        return ((uint32_t)ux[x1 - fxl] << 16) | (uint32_t)uy[y1 - fyl];
    };
    struct FCell { int sx, sy; int gx0, gx1, gy0, gy1; int ngx, ngy; };            // storage cell and its source global cells
    auto cell_sources = [&](uint32_t sc, FCell& f) {
        f.sx = (int)(sc >> 16); f.sy = (int)(sc & 0xFFFFu);
        const int ax = f.sx - C, ay = f.sy - C;                                     // sources: a (if not glitched), a + 1 (if glitched)
        const bool xa = !gxb[ax - fxl], xb = gxb[ax + 1 - fxl], ya = !gyb[ay - fyl], yb = gyb[ay + 1 - fyl];
        f.ngx = (xa ? 1 : 0) + (xb ? 1 : 0); f.gx0 = xa ? ax : ax + 1; f.gx1 = ax + 1;
        f.ngy = (ya ? 1 : 0) + (yb ? 1 : 0); f.gy0 = ya ? ay : ay + 1; f.gy1 = ay + 1;
    };
    // tile and offset of a storage cell (rays are shorter than a tile: the lattice coordinate moves by at most one)
    auto cell_addr = [&](int sx, int sy, int& tile, int& row_t, int& col_t) {
        const int a = a0 + (sx >= (a0 + 1) * v.dim ? 1 : 0) - (sx < a0 * v.dim ? 1 : 0);
        const int bb = b0 + (sy >= (b0 + 1) * v.dim ? 1 : 0) - (sy < b0 * v.dim ? 1 : 0);
        tile = ((unsigned)a < (unsigned)v.L && (unsigned)bb < (unsigned)v.L) ? s_tab[a * v.L + bb] : -1;
        row_t = sx - a * v.dim; col_t = sy - bb * v.dim;
    };
    // direction of a ray: steps of the major / minor axis in global cells
    struct RayDir { int steep, sx, sy; };
    auto ray_dir = [&](int b) {
        const int32_t re = r_end[b];
        const int ex = (int)(int16_t)(re & 0xFFFF), ey = (int)(int16_t)((uint32_t)re >> 16);
        const int aex = ex < 0 ? -ex : ex, aey = ey < 0 ? -ey : ey;
        RayDir d; d.steep = aey > aex; d.sx = ex > 0 ? 1 : -1; d.sy = ey > 0 ? 1 : -1;          // hybridmap.py:282-283
        return d;
    };
    const int sat = (v.cc.vmax - v.cc.vmin + (-v.cc.emp) - 1) / (-v.cc.emp);        // passes that saturate any cell: 20
    const uint32_t satb = (uint32_t)sat * 0x01010101u, sadd = (128u - (uint32_t)sat) * 0x01010101u;
    const int cnt_lds = ev_lds_addr(cnt), mini_lds = ev_lds_addr(mini);
    // a thin fan: fewer ray cells than two fifths of the fan's box (181 beams on a 0.025 m grid: a seventh)
    const bool sparse = (long long)UNI((int)s_cells) * 5 < 2LL * (long long)(S_hi - S_lo + 1) * (long long)(T_hi - T_lo + 1);
    // a lane's steps that met a flagged cell (bit 16 + u of m = step j0 + u): into the event list
    auto log_events = [&](uint32_t m, int b, int j0, int ev_lo, int ev_hi, int gx_base, bool filter) {
        while (m) {
            const int lb = __ffs((int)m) - 1;
            m &= m - 1;
            const int j = j0 + lb - 16;
            if (filter) {   // strips share a global row with their neighbours: the strip that owns its storage row reports
                const RayDir d = ray_dir(b);
                const int row = x0 + d.sx * (d.steep ? ev_minor(r_fs[b], j) : j) - gx_base;
                if (row < ev_lo || row > ev_hi) continue;
            }
            const int pos = atomicAdd(&s_nev, 1);
            if (pos < EVCAP) evlist[pos] = ((uint32_t)b << 10) | (uint32_t)j;
        }
    };
    // this thread's pairs (beam, e): pair = tid, tid + EB, tid + 2 EB; cell and old value stay in registers until the
    // flagged cells are folded (the old value must be read before the write-back)
    uint32_t my_sc[NPAIR]; int my_old[NPAIR];
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
        const int pr = tid + i * EB;
        my_sc[i] = pr < 2 * v.B ? pair_cell(pr) : 0xFFFFFFFFu;
        my_old[i] = 0;
        if (my_sc[i] != 0xFFFFFFFFu) {
            int tile, row_t, col_t;
            cell_addr((int)(my_sc[i] >> 16), (int)(my_sc[i] & 0xFFFFu), tile, row_t, col_t);
            if (tile >= 0) my_old[i] = (int)v.pool[(size_t)tile * v.dim * v.dim + (size_t)row_t * v.dim + col_t];   // (in flight until the fold)
        }
    }

    // =============================================== windows ==============================================
    int n_win = 0;
    for (int S0 = S_lo; S0 <= S_hi; S0 += rows_cap - 1, ++n_win) {
        const int S1 = min(S_hi, S0 + rows_cap - 2);                             // storage rows S0..S1
        const int gx_base = S0 - C, rows_w = S1 - S0 + 2;                        // global rows gx_base .. gx_base + rows_w - 1
        const bool whole = S0 == S_lo && S1 == S_hi;                             // one window holds the fan
        BAR_LDS();                                                               // the previous window (or the slope buckets) is done with the counters
        {
            uint4* c4 = reinterpret_cast<uint4*>(cnt);
            const int n16 = (rows_w * stride + 15) >> 4;
            for (int i = tid; i < n16; i += EB) c4[i] = make_uint4(0, 0, 0, 0);
        }
        BAR_LDS();
        // ---- flags: every global cell that maps to a storage cell with an occupied / nearby hit; the flag comes with a
        //      count of one, so a flagged field is never zero (the write-back takes "touched" from the field) ----
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) {
            uint32_t sc = my_sc[i];
            asm volatile("" : "+v"(sc));                                         // (the cell's sources are worked out here, strip by strip: hoisted out of the strips' loop they were 24 registers spilled to scratch)
            if (sc == 0xFFFFFFFFu) continue;
            FCell f;
            cell_sources(sc, f);
#pragma unroll
            for (int ix = 0; ix < 2; ++ix)
#pragma unroll
            for (int iy = 0; iy < 2; ++iy) {
                if (ix >= f.ngx || iy >= f.ngy) continue;
                const int sgx = ix ? f.gx1 : f.gx0, sgy = iy ? f.gy1 : f.gy0;
                const int ddx = sgx - x0, ddy = sgy - y0;
                if (max(ddx < 0 ? -ddx : ddx, ddy < 0 ? -ddy : ddy) < NEAR_R) {
                    if (n_win == 0) { const int mi = (ddx + NEAR_R) * NEAR_W + (ddy + NEAR_R); atomicOr(&mini[mi >> 1], 3u << ((mi & 1) * 16)); }
                } else {
                    const int row = sgx - gx_base, col = sgy - gy_base;
                    if ((unsigned)row < (unsigned)rows_w && (unsigned)col < (unsigned)stride) { const int c = row * stride + col; atomicOr(&cnt[c >> 2], 3u << ((c & 3) * 8)); }
                }
            }
        }
        BAR_LDS();
        STAMP(2);
        // rows of this window whose passes are reported from here (the first and the last global row belong to two strips)
        const int ev_lo = (!whole && gxb[gx_base - fxl]) ? 1 : 0, ev_hi = (!whole && !gxb[gx_base + rows_w - 1 - fxl]) ? rows_w - 2 : rows_w - 1;
        // ---- the 16-bit block: steps 0 .. NEAR_R - 1 of every ray, once ----
        if (n_win == 0) {
            const int nw0 = (v.B + 63) >> 6;
            for (int q = wave; q < nw0; q += EB / 64) {
                const int b = lane * nw0 + q;                                          // the 64 rays of an instruction point in different directions
                const bool valid = b < v.B;
                const int nE = valid ? (int)r_nE[b] : 0;
                if (__ballot(nE > 0) == 0ull) continue;
                const uint32_t fs = valid ? r_fs[b] : 0u;
                const RayDir d = ray_dir(valid ? b : 0);
                const int mj = d.steep ? d.sy : d.sx * NEAR_W, mm = d.steep ? d.sx * NEAR_W : d.sy;
                const int d0 = mj, d1 = mj + mm;
                uint32_t acc = 0x80000000u, m = 0;
                int c = NEAR_R * NEAR_W + NEAR_R + (mini_lds >> 1);                    // field index, the array's LDS address folded in
                uint32_t ret[NEAR_R]; int sh[NEAR_R];
#pragma unroll
                for (int u = 0; u < NEAR_R; ++u) {
                    sh[u] = c << 4;
                    ret[u] = ev_lds_add_rtn((c << 1) & ~3, (u < nE ? 2u : 0u) << (sh[u] & 31));
                    const uint32_t nacc = acc + fs;
                    c += nacc < acc ? d1 : d0;
                    acc = nacc;
                }
                __builtin_amdgcn_sched_barrier(0);                                     // sixteen adds in flight, then their answers
#pragma unroll
                for (int u = 0; u < NEAR_R; ++u) m = __builtin_amdgcn_alignbit(u < nE ? ret[u] >> (sh[u] & 31) : 0u, m, 1);
                if (m) log_events(m, b, 0, 0, 0, 0, false);
            }
        }
        // ---- walk: lanes are rays, a work item is one 16-step chunk of 64 rays.  Level k >= 1 = steps NEAR_R + 16 (k - 1) .. + 15;
        //      the rays that own a whole k-th chunk are perm[0 .. N_k); lane l of the w-th wave of a level takes ray l * waves + w.
        //      A step: field += 2 with the old word returned, bit 0 of the old field = flagged; the address moves by one of two
        //      constants, chosen by the carry of the slope accumulator. ----
        {
            const int rx0 = x0 - gx_base, ry0 = y0 - gy_base;
            const int base0 = rx0 * stride + ry0 + cnt_lds;
            const int win_lo = cnt_lds, win_n = rows_w * stride;
            // one predicated chunk of ray b (steps j0 .. j0 + 15, those in [jlo, jhi] and inside the window count): the tail of a
            // ray, or - in a strip - a chunk the strip's edge cuts.  Branch-free: a dead step adds nothing to a word of the lane's own.
            auto walk_pred = [&](int b, int j0, int jlo, int jhi) {
                const uint32_t fs = r_fs[b];
                const RayDir d = ray_dir(b);
                const int sxs = d.sx * stride;
                const int cj = d.steep ? d.sy : sxs, cm = d.steep ? sxs : d.sy;
                const unsigned long long pr64 = (unsigned long long)fs * (unsigned)j0 + 0x80000000ull;
                uint32_t acc = (uint32_t)pr64, m = 0, inm = 0;
                int c = base0 + __mul24(j0, cj) + __mul24((int)(pr64 >> 32), cm);
                const int d0 = cj, d1 = cj + cm;
                uint32_t ret[LCH]; int sh[LCH];
#pragma unroll
                for (int u = 0; u < LCH; ++u) {
                    const bool in = j0 + u >= jlo && j0 + u <= jhi && (unsigned)(c - win_lo) < (unsigned)win_n;
                    sh[u] = c << 3;
                    ret[u] = ev_lds_add_rtn(in ? c & ~3 : win_lo + 4 * lane, (in ? 2u : 0u) << (sh[u] & 31));
                    inm |= in ? 1u << u : 0u;
                    const uint32_t nacc = acc + fs;
                    c += nacc < acc ? d1 : d0;
                    acc = nacc;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < LCH; ++u) m = __builtin_amdgcn_alignbit((inm >> u) & (ret[u] >> (sh[u] & 31)), m, 1);
                if (m) log_events(m, b, j0, ev_lo, ev_hi, gx_base, !whole);
            };
            if (whole) {
                // the tail of every ray with more than NEAR_R steps (its whole chunks follow, level by level)
                for (int b = tid; b < v.B; b += EB) {
                    const int nE = (int)r_nE[b];
                    if (nE <= NEAR_R) continue;
                    const int j0 = NEAR_R + ((nE - NEAR_R) / LCH) * LCH;            // first step after the whole chunks
                    if (j0 < nE) walk_pred(b, j0, j0, nE - 1);
                }
            } else {
                // A strip holds a part of every ray: the steps NEAR_R .. nE - 1 whose row lies in the strip are a run
                // [lo, hi] (rows never turn back along a ray).  The whole chunks inside the run are the ray's share of the
                // strip's level walk (relative levels: rays ordered by how many such chunks they have); the one or two chunks
                // that the strip's edge or the ray's end cuts are walked here, predicated.
                if (tid <= MAXLEV) s_lcnt[tid] = 0;
                BAR_LDS();
                auto first_j = [&](uint32_t fs, int mval) -> int {               // first step with minor(j) >= mval (mval >= 1)
                    if (fs == 0u) return 1 << 20;
                    const double jd = ((double)mval * 4294967296.0 - 2147483648.0) / (double)fs;
                    if (jd > 4000.0) return 1 << 20;
                    int j = (int)jd;
                    while (ev_minor(fs, j) < mval) ++j;
                    while (j > 0 && ev_minor(fs, j - 1) >= mval) --j;
                    return j;
                };
                for (int b = tid; b < v.B; b += EB) {
                    const int nE = (int)r_nE[b];
                    int ka = 1, len = 0;
                    if (nE > NEAR_R) {
                        const uint32_t fs = r_fs[b];
                        const RayDir d = ray_dir(b);
                        int lo = NEAR_R, hi = nE - 1;
                        if (!d.steep) {                                            // row = rx0 + sx * j
                            if (d.sx > 0) { lo = max(lo, -rx0); hi = min(hi, rows_w - 1 - rx0); }
                            else { lo = max(lo, rx0 - (rows_w - 1)); hi = min(hi, rx0); }
                        } else {                                                   // row = rx0 + sx * minor(j), minor never decreases
                            const int mlo = d.sx > 0 ? -rx0 : rx0 - (rows_w - 1), mhi = d.sx > 0 ? rows_w - 1 - rx0 : rx0;
                            if (mhi < 0) hi = -1;
                            else {
                                if (mlo > 0) lo = max(lo, first_j(fs, mlo));
                                hi = min(hi, first_j(fs, mhi + 1) - 1);
                            }
                        }
                        if (lo <= hi) {
                            const int nfull = (nE - NEAR_R) / LCH;
                            const int kf = (lo - NEAR_R) / LCH + 1, kl = (hi - NEAR_R) / LCH + 1;   // the chunks that hold the first / the last step
                            const int a = lo == NEAR_R + (kf - 1) * LCH ? kf : kf + 1;             // first chunk that lies inside as a whole
                            const int e = min(hi == NEAR_R + kl * LCH - 1 ? kl : kl - 1, nfull);   // last one
                            if (e >= a) { ka = a; len = e - a + 1; }
                            if (kf < a || kf > e) walk_pred(b, NEAR_R + (kf - 1) * LCH, lo, hi);
                            if (kl != kf && (kl < a || kl > e)) walk_pred(b, NEAR_R + (kl - 1) * LCH, lo, hi);
                        }
                    }
                    r_kl[b] = (uint16_t)(ka | (len << 8));
                    if (len) atomicAdd(&s_lcnt[min(len, MAXLEV)], 1);
                }
                BAR_LDS();
                if (wave == 0) {   // relative levels: N_k = rays with at least k whole chunks inside the strip
                    const int k = lane;
                    const int ck = k >= 1 ? s_lcnt[k] : 0;
                    int suf = ck;
                    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(suf, o, 64); if (lane + o < 64) suf += t; }
                    const int nwk = k >= 1 ? (suf + 63) >> 6 : 0;
                    int pre = nwk;
                    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pre, o, 64); if (lane >= o) pre += t; }
                    const unsigned long long live = __ballot(k >= 1 && suf > 0);
                    if (k >= 1) { s_lfill[k] = suf - ck; s_nk[k] = suf; s_lp[k] = pre - nwk; }
                    if (k == 63) s_lp[64] = pre;
                    if (k == 0) { s_nlev = live ? 63 - __clzll((long long)live) : 0; s_nk[MAXLEV + 1] = 0; }
                }
                BAR_LDS();
                for (int b = tid; b < v.B; b += EB) {
                    const int len = (int)r_kl[b] >> 8;
                    if (len) perm[atomicAdd(&s_lfill[min(len, MAXLEV)], 1)] = (uint16_t)b;
                }
                BAR_LDS();
            }
            // ---- the level walk: every lane runs all 16 steps of a whole chunk (inside the window as a whole) ----
            const int nlev_w = UNI(s_nlev);
            const int nitems = UNI(s_lp[nlev_w + 1]);                             // (levels above nlev have no waves: s_lp stays flat)
            // the next item's ray is fetched while this one's adds are in flight
            int k = 1, kn = 1;
            int lp_c = UNI(s_lp[1]), lp_n = UNI(s_lp[2]), nk_c = UNI(s_nk[1]);    // first item of the level, of the next level; rays of the level
            int nb = -1, nka = 1; uint32_t nfs = 0; int32_t nend = 0;
            auto fetch_item = [&](int q) {
                nb = -1;
                if (q >= nitems) return;
                while (kn < nlev_w && q >= lp_n) { ++kn; lp_c = lp_n; lp_n = UNI(s_lp[kn + 1]); nk_c = UNI(s_nk[kn]); }
                const int nwk = (nk_c + 63) >> 6, wslot = q - lp_c;
                const int ii = lane * nwk + wslot;
                if (ii < nk_c) { nb = perm[ii]; nfs = r_fs[nb]; nend = r_end[nb]; nka = (int)r_kl[nb] & 0xFF; }
            };
            // (Drawing the items from a queue balances the waves - the oldest wave of every SIMD is served first and waits a third
            // of the walk for the youngest - and was 0.7 % faster, at the price of 20 spilled registers: 330 MB of scratch traffic per
            // 4096-particle launch.  A fixed share per wave it is.)
            fetch_item(wave);
            for (int q = wave; q < nitems; q += EB / 64) {
                const int b = nb; const uint32_t fs = nfs; const int32_t re = nend;
                k = kn + nka - 1;                                                 // the ray's chunk at this relative level
                if (b < 0) { fetch_item(q + EB / 64); continue; }
                const int ex = (int)(int16_t)(re & 0xFFFF), ey = (int)(int16_t)((uint32_t)re >> 16);
                const int aex = ex < 0 ? -ex : ex, aey = ey < 0 ? -ey : ey;
                const int sxs = ex > 0 ? stride : -stride, sy1 = ey > 0 ? 1 : -1;
                const int cj = aey > aex ? sy1 : sxs, cm = aey > aex ? sxs : sy1;
                const int j0 = NEAR_R + (k - 1) * LCH;
                const unsigned long long pr64 = (unsigned long long)fs * (unsigned)j0 + 0x80000000ull;
                uint32_t acc = (uint32_t)pr64, m = 0;
                int c = base0 + __mul24(j0, cj) + __mul24((int)(pr64 >> 32), cm);
                const int d0 = cj, d1 = cj + cm;
                uint32_t ret[LCH]; int sh[LCH];
#pragma unroll
                for (int u = 0; u < LCH; ++u) {
                    sh[u] = c << 3;
                    ret[u] = ev_lds_add_rtn(c & ~3, 2u << (sh[u] & 31));
                    const uint32_t nacc = acc + fs;
                    c += nacc < acc ? d1 : d0;
                    acc = nacc;
                }
                fetch_item(q + EB / 64);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);                                // lgkmcnt(0): one wait, then the sixteen answers
#pragma unroll
                for (int u = 0; u < LCH; ++u) m = __builtin_amdgcn_alignbit(ret[u] >> (sh[u] & 31), m, 1);
                if (m) log_events(m, b, j0, ev_lo, ev_hi, gx_base, !whole);
            }
        }
        BAR_LDS();
        STAMP(3);
        // ---- the 16-bit block's counts go into the window (saturated: only min(n, sat) matters for an unflagged cell;
        //      a flagged one arrives with its count of one and stays "touched") ----
        for (int mi = tid; mi < NEAR_W * NEAR_W; mi += EB) {
            const int row = x0 + mi / NEAR_W - NEAR_R - gx_base, col = y0 + mi % NEAR_W - NEAR_R - gy_base;
            if ((unsigned)row >= (unsigned)rows_w || (unsigned)col >= (unsigned)stride) continue;
            const uint32_t f = ((mini[mi >> 1] >> ((mi & 1) * 16)) & 0xFFFFu) >> 1;
            if (f) cnt8[row * stride + col] = (uint8_t)(min(f, (uint32_t)sat) << 1);
        }
        if (tid == 0) s_wbq = 0;
        BAR_LDS();
        // ---- write-back: one read-modify-write per touched 32-cell group of storage cells, tile by tile.  Storage cell s
        //      receives global cell s - C where that one is not glitched plus global cell s - C + 1 where that one is.  The
        //      flagged cells get a value from their counts like all others; their real value follows after the windows. ----
        {
            int my_written = 0;
            const int eabs = -v.cc.emp;
            const EvWb wbk = {(uint32_t)(128 + v.cc.vmin) * 0x01010101u, (uint32_t)(127 - (v.cc.thr - v.cc.vmin)) * 0x01010101u, satb, sadd, -v.cc.emp};
            const uint32_t kb1 = (uint32_t)(128 + v.cc.vmin) * 0x01010101u;             // byte-wise: (cell ^ 0x80) - kb1 = cell - vmin
            const uint32_t oadd = (uint32_t)(127 - (v.cc.thr - v.cc.vmin)) * 0x01010101u; // bit 7 of (R + oadd) = cell > thr
            // Waves draw batches of 64 items (32-cell groups) from a queue: the rows at the fan's rim hold few touched groups, and
            // with a fixed share per wave the workgroup waited a quarter of the write-back's time for its slowest wave.
            auto next_batch = [&]() -> int { int g = 0; if (lane == 0) g = atomicAdd(&s_wbq, 1); return UNI(g); };
            int batch = next_batch(), batch0 = 0;                        // batch0: the first batch of the tile at hand
            for (int a = S0 / v.dim; a <= S1 / v.dim; ++a)
            for (int bt = T_lo / v.dim; bt <= T_hi / v.dim; ++bt) {
                if (a >= v.L || bt >= v.L) continue;                                   // uniform
                const int tile = UNI(s_tab[a * v.L + bt]);
                if (tile < 0) continue;
                const int sr_lo = max(S0, a * v.dim), sr_hi = min(S1, (a + 1) * v.dim - 1);      // storage rows
                const int g_lo = max(T_lo - bt * v.dim, 0) >> 5, g_hi = min(T_hi - bt * v.dim, v.dim - 1) >> 5;   // groups of this tile's rows
                const int ngr = g_hi - g_lo + 1, items = (sr_hi - sr_lo + 1) * ngr;
                int8_t* __restrict__ tile_base = v.pool + (size_t)tile * v.dim * v.dim;
                int bx0 = INT_MAX, bx1 = -1, by0 = INT_MAX, by1 = -1;
                const int nbatch = (items + 63) >> 6;
                const float inv_ngr = 1.0f / (float)ngr;
                for (; batch < batch0 + nbatch; batch = next_batch()) {
                    const int it = ((batch - batch0) << 6) + lane;
                    if (it >= items) continue;
                    const int rr = (int)(((float)it + 0.5f) * inv_ngr), gg = it - rr * ngr;   // it / ngr: (it + 0.5) / ngr is at least 0.5 / 192 from a whole number, the float product's error 1e-4 of that
                    const int srow = sr_lo + rr, gt = g_lo + gg;
                    const int ia = srow - C - fxl;                                     // source rows a (if not glitched), a + 1 (if glitched)
                    const bool va = !gxb[ia], vb = gxb[ia + 1];
                    if (!va && !vb) continue;                                          // no global row maps here
                    const int lr = srow - C - gx_base;                                 // window row of source a
                    const int lc0 = bt * v.dim + 32 * gt - C - gy_base;                // window column of the group's first cell, multiple of 4
                    const int nw = min(32, v.dim - 32 * gt) >> 2;                      // words of this group (8; fewer in a tile's last group)
                    uint32_t n[8];
                    uint32_t any = 0;
                    const bool both = va && vb;
                    if (!both && !s_ggf[(bt - bt_lo) * gpt + gt]) {   // one source row, no glitched column: the fields are the group's counts
                        const int rowo = (lr + (va ? 0 : 1)) * stride + lc0;
#pragma unroll
                        for (int w = 0; w < 8; ++w) {
                            const int lc = lc0 + 4 * w;
                            n[w] = (lc >= 0 && lc < stride && w < nw) ? (cnt[(rowo + 4 * w) >> 2] >> 1) & 0x7F7F7F7Fu : 0u;
                            any |= n[w];
                        }
                    } else {
                        uint32_t gm[9];                                                // glitched columns in the group (its 32 cells and the one after)
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
                                x[w] = (lc >= 0 && lc < stride) ? (cnt[(row * stride + lc) >> 2] >> 1) & 0x7F7F7F7Fu : 0u;
                            }
#pragma unroll
                            for (int w = 0; w < 9; ++w) x[w] = ev_min4(x[w], satb, sadd);
#pragma unroll
                            for (int w = 0; w < 8; ++w) {
                                const uint32_t keep = x[w] & ~gm[w];
                                const uint32_t mv = ((x[w] & gm[w]) >> 8) | ((x[w + 1] & gm[w + 1]) << 24);
                                n[w] += keep + mv;
                            }
                        }
#pragma unroll
                        for (int w = 0; w < 8; ++w) { if (w >= nw) n[w] = 0; any |= n[w]; }
                    }
                    if (!any) continue;
                    const int row_t = srow - a * v.dim, col_t = 32 * gt;
                    uint32_t* g_ptr = reinterpret_cast<uint32_t*>(tile_base + (size_t)row_t * v.dim + col_t);
                    uint32_t pre[8];
                    if (nw == 8) {
                        const uint4 q0 = reinterpret_cast<const uint4*>(g_ptr)[0], q1 = reinterpret_cast<const uint4*>(g_ptr)[1];
                        pre[0] = q0.x; pre[1] = q0.y; pre[2] = q0.z; pre[3] = q0.w; pre[4] = q1.x; pre[5] = q1.y; pre[6] = q1.z; pre[7] = q1.w;
                    } else {
#pragma unroll
                        for (int w = 0; w < 8; ++w) pre[w] = w < nw ? g_ptr[w] : 0u;
                    }
                    if (sparse) {   // a thin fan (few beams on a fine grid): a group holds one or two touched words - the arithmetic and
                                    // the stores are theirs alone; the other words only give their occupancy bits (the group's 32 bytes are
                                    // one memory sector: read whole, written by the word)
                        uint32_t nzm = 0;
#pragma unroll
                        for (int w = 0; w < 8; ++w) nzm |= n[w] ? 1u << w : 0u;
                        if (__popc(nzm) <= 3) {
                            uint32_t occ = 0, touched = 0;
#pragma unroll
                            for (int w = 0; w < 8; ++w)                                       // cell > thr of the cells as they are
                                occ |= __builtin_amdgcn_udot4(((((pre[w] ^ 0x80808080u) - wbk.kb1) + wbk.oadd) & 0x80808080u) >> 7, 0x08040201u, 0u, false) << (4 * w);
                            if (nw < 8) occ &= (1u << (4 * nw)) - 1u;
                            uint32_t mm = nzm;
#pragma unroll
                            for (int q = 0; q < 3; ++q) {
                                const int wq = mm ? __ffs((int)mm) - 1 : -1;
                                mm &= mm - 1;
                                if (wq < 0) continue;
                                uint32_t pw = 0, nv = 0;
#pragma unroll
                                for (int w = 0; w < 8; ++w) { pw = w == wq ? pre[w] : pw; nv = w == wq ? n[w] : nv; }
                                uint32_t t4, o4;
                                g_ptr[wq] = ev_wb_word(wbk, pw, nv, t4, o4);
                                touched |= t4 << (4 * wq);
                                occ = (occ & ~(0xFu << (4 * wq))) | (o4 << (4 * wq));
                            }
                            v.occ[((size_t)tile * v.dim + row_t) * v.ow + gt] = occ;
                            my_written += __popc(touched);
                            by0 = min(by0, col_t + __ffs(touched) - 1); by1 = max(by1, col_t + 31 - __clz(touched));
                            bx0 = min(bx0, row_t); bx1 = max(bx1, row_t);
                            continue;
                        }
                    }
                    uint32_t occ = 0, touched = 0, out[8];
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        // branch-free (a word without hits passes through unchanged: dec = 0, nz = 0)
                        const uint32_t Ob = (pre[w] ^ 0x80808080u) - kb1;                   // cells biased to [0, vmax - vmin]
                        const uint32_t n7 = n[w];
                        const uint32_t m = ev_min4(n7, satb, sadd);                         // min(n, sat)
                        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                        const uint32_t dec = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, m) * (us2)(unsigned short)eabs);   // byte-wise: sat * |emp| < 128, no carries
                        const uint32_t T1 = (Ob | 0x80808080u) - dec;
                        const uint32_t pos = T1 & 0x80808080u;                              // O - dec >= 0
                        const uint32_t R = T1 & 0x7F7F7F7Fu & (pos | (pos - (pos >> 7)));
                        out[w] = (R + kb1) ^ 0x80808080u;
                        const uint32_t nz = (n7 + 0x7F7F7F7Fu) & 0x80808080u;               // fields that are not zero
                        touched |= __builtin_amdgcn_udot4(nz >> 7, 0x08040201u, 0u, false) << (4 * w);
                        occ |= __builtin_amdgcn_udot4(((R + oadd) & 0x80808080u) >> 7, 0x08040201u, 0u, false) << (4 * w);   // cell > thr
                    }
                    if (nw == 8) {
                        reinterpret_cast<uint4*>(g_ptr)[0] = make_uint4(out[0], out[1], out[2], out[3]);
                        reinterpret_cast<uint4*>(g_ptr)[1] = make_uint4(out[4], out[5], out[6], out[7]);
                    } else {
#pragma unroll
                        for (int w = 0; w < 8; ++w) if (w < nw) g_ptr[w] = out[w];
                        occ &= (1u << (4 * nw)) - 1u;                                      // (cells past the tile's last column are not cells)
                    }
                    my_written += __popc(touched);
                    by0 = min(by0, col_t + __ffs(touched) - 1); by1 = max(by1, col_t + 31 - __clz(touched));
                    v.occ[((size_t)tile * v.dim + row_t) * v.ow + gt] = occ;
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
        STAMP(4);
    }
    BAR_LDS();                                                                // the window's LDS is free (the write-back's stores are still on their way)
#ifdef EV_STAMP_SPLIT
    STAMP(5);
#endif

    // ======================================== flagged cells ========================================
    uint32_t* const keys = reinterpret_cast<uint32_t*>(smem + G.o_cnt + G.p_keys);    // [T] storage cell, ~0 = empty
    uint32_t* const head = reinterpret_cast<uint32_t*>(smem + G.o_cnt + G.p_cnta);    // [T] first pair of the cell's list, 0xFFFF = none
    uint16_t* const iclast = reinterpret_cast<uint16_t*>(smem + G.o_cnt + G.p_offs);  // [T] passes after the cell's last event
    int8_t*   const oldc = reinterpret_cast<int8_t*>(smem + G.o_cnt + G.p_oldv);      // [T] the cell's value before the scan
    uint16_t* const rlist = reinterpret_cast<uint16_t*>(smem + G.o_cnt + G.p_rlist);  // [records] slots in use
    uint16_t* const nextp = reinterpret_cast<uint16_t*>(smem + G.o_cnt + G.p_evl);    // [pairs] next pair (beam << 1 | nearby) on the same cell
    uint16_t* const evl = nextp + ev_al16(G.E * 2) / 2;                               // [pairs] the lists, one after the other (laid out by the fold)
    uint16_t* const ic16 = reinterpret_cast<uint16_t*>(smem + G.o_cnt + G.p_ic);      // [pairs] passes right before the pair's event
    const int T = G.T;
    const int nev_all = UNI(s_nev);
    const bool overflow = nev_all > EVCAP;
    for (int i = tid; i < T / 4; i += EB) { reinterpret_cast<uint4*>(keys)[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); reinterpret_cast<uint4*>(head)[i] = make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu); }
    for (int i = tid; i < T / 8; i += EB) reinterpret_cast<uint4*>(iclast)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < ev_al16(G.E * 2) / 16; i += EB) reinterpret_cast<uint4*>(ic16)[i] = make_uint4(0, 0, 0, 0);
    if (tid == 0) { s_wsum[0] = 0; s_wsum[1] = 0; }
    BAR_LDS();
    // every pair joins the list of its cell (the cell's first pair lists the cell and leaves its old value)
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
        if (my_sc[i] == 0xFFFFFFFFu) continue;
        bool created;
        const int h = ev_hash_insert2(keys, T, G.logT, my_sc[i], created);
        {   // a record per new cell: the wave's new cells take their places in the record list with ONE add (1300 returning adds
            // on one LDS word stand in line otherwise)
            const unsigned long long mk = __ballot(created);
            if (created) {
                const int first = __ffsll((long long)mk) - 1;
                int base = 0;
                if (lane == first) base = atomicAdd(&s_wsum[1], __popcll(mk));
                base = __shfl(base, first, 64);
                rlist[base + __popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)h; oldc[h] = (int8_t)my_old[i];
            }
        }
        nextp[tid + i * EB] = (uint16_t)atomicExch(&head[h], (uint32_t)(tid + i * EB));
    }
    BAR_LDS();
#ifndef EV_STAMP_SPLIT
    STAMP(5);
#endif
    if (!overflow) {
        // every listed pass is counted in front of the first event of a beam that is not smaller than its own (a beam's own
        // passes come before its own occupied / nearby hit), or behind the cell's last event
        for (int e = tid; e < nev_all; e += EB) {
            const uint32_t ev = evlist[e];
            const int b = (int)(ev >> 10), j = (int)(ev & 1023u);
            const RayDir d = ray_dir(b);
            const int mn = ev_minor(r_fs[b], j);
            const int gx = x0 + d.sx * (d.steep ? mn : j), gy = y0 + d.sy * (d.steep ? j : mn);
            const int h = ev_hash_find(keys, T, G.logT, ((uint32_t)ux[gx - fxl] << 16) | (uint32_t)uy[gy - fyl]);
            if (h < 0) continue;                                                 // (cannot happen: only flagged fields report)
            int best = 0xFFFF;
            for (int m = (int)head[h]; m != 0xFFFF; m = nextp[m]) if (m >= 2 * b && m < best) best = m;
            uint16_t* const slot = best != 0xFFFF ? &ic16[best] : &iclast[h];
            atomicAdd(reinterpret_cast<unsigned int*>(reinterpret_cast<uintptr_t>(slot) & ~(uintptr_t)3), (reinterpret_cast<uintptr_t>(slot) & 2) ? 0x10000u : 1u);
        }
    }
    __syncthreads();                                                          // every store of the write-back has landed: the flagged cells' bytes follow
    STAMP(6);
    // the byte and the occupancy bit of a flagged cell
    auto store_cell = [&](uint32_t key, int val) {
        int tile, row_t, col_t;
        cell_addr((int)(key >> 16), (int)(key & 0xFFFFu), tile, row_t, col_t);
        if (tile < 0) return;
        v.pool[(size_t)tile * v.dim * v.dim + (size_t)row_t * v.dim + col_t] = (int8_t)val;
        uint32_t* ow = &v.occ[((size_t)tile * v.dim + row_t) * v.ow + (col_t >> 5)];
        if (val > v.cc.thr) atomicOr(ow, 1u << (col_t & 31)); else atomicAnd(ow, ~(1u << (col_t & 31)));
    };
    const int NR = UNI(s_wsum[1]);
    if (!overflow) {
        for (int r = tid; r < NR; r += EB) {
            const int h = rlist[r];
            int val = (int)oldc[h];
            // the cell's pairs, laid out one after the other
            int n = 0;
            for (int m = (int)head[h]; m != 0xFFFF; m = nextp[m]) ++n;
            const int o2 = atomicAdd(&s_wsum[0], n);
            { int i = 0; for (int m = (int)head[h]; m != 0xFFFF; m = nextp[m]) evl[o2 + i++] = (uint16_t)m; }
            // the events in ascending (beam, nearby) order: the smallest one above the last, n times (a handful per cell)
            int cur = -2;
            for (int i = 0; i < n; ++i) {
                int ek = 0x7FFFFFFF, nx = 0x7FFFFFFF;                                 // the next event and the one after it
                for (int q = 0; q < n; ++q) { const int e = evl[o2 + q]; if (e > cur) { if (e < ek) { nx = ek; ek = e; } else if (e < nx) nx = e; } }
                const int beam = ek >> 1;
                // the beam's own pass over the cell before its nearby hit is not in the list: it precedes the beam's first event here
                const bool has_near = (ek & 1) || nx == ek + 1;
                const int np = (int)ic16[ek] + ((beam != (cur >> 1) && has_near) ? 1 : 0);
                val = max(val + min(np, sat) * v.cc.emp, v.cc.vmin);              // gridmap.py:97-101, np times
                val = min(val + ((ek & 1) ? v.cc.nearby : v.cc.occ), v.cc.vmax);  // gridmap.py:86-90 / 108-112
                cur = ek;
            }
            val = max(val + min((int)iclast[h], sat) * v.cc.emp, v.cc.vmin);
            store_cell(keys[h], val);
        }
    } else {
        // more passes over flagged cells than the list holds: every flagged cell is replayed by a wave with the exact
        // closed-form membership test over all beams (rbpf_mapupdate.h), whatever the list says
        for (int r = wave; r < NR; r += EB / 64) {
            const int h = rlist[r];
            const uint32_t key = keys[h];
            FCell f;
            cell_sources(key, f);
            const int gxc[2] = {f.gx0, f.gx1}, gyc[2] = {f.gy0, f.gy1};
            const int val = replay_cell_wave(v, r_info, r_end, x0, y0, gxc, f.ngx, gyc, f.ngy, (int)oldc[h], lane);
            if (lane == 0) store_cell(key, val);
        }
        if (tid == 0) atomicAdd(&v.stats[ST_SLOW_CELLS], (unsigned long long)NR);
    }
    STAMP(7);
#if defined(RBPF_STAMPS) && defined(EV_BARRIER_WAITS)
    if (p == 0 && lane == 0) printf("wave %2d: %6lld cycles at %d barriers of %lld: %d %d %d %d %d %d %d %d\n", wave, bar_wait, bar_n, clock64() - t_begin, bar_w[0], bar_w[1], bar_w[2], bar_w[3], bar_w[4], bar_w[5], bar_w[6], bar_w[7]);
#endif
    if (tid == 0) {
        if (s_cells) atomicAdd(&v.stats[ST_RAY_CELLS], s_cells);
        if (s_written) atomicAdd(&v.stats[ST_CELLS_WRITTEN], (unsigned long long)s_written);
        atomicAdd(&v.stats[ST_MAP_WINDOWS], (unsigned long long)n_win);
        atomicAdd(&v.stats[ST_MAP_EVENTS], (unsigned long long)nev_all);
        if (overflow) atomicAdd(&v.stats[ST_EV_OVERFLOWS], 1ull);
#ifdef RBPF_STAMPS
        for (int k = 0; k < 8; ++k) atomicAdd(&v.stats[8 + k], (unsigned long long)st_acc[k]);
#endif
    }
}

// t0 / t1 (or nullptr): timing events that take the kernel's own start and end (hipExtLaunchKernelGGL: the dispatch carries
// them, no event records - and their barriers - in the stream)
void launch_map_update_ev(const DevView& v, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    const EvGeom g = ev_geom(v.B, v.reach);
    static size_t lds_set[MAX_DEVICES] = {};
    ensure_dynamic_lds(reinterpret_cast<const void*>(map_update_ev_kernel), (size_t)g.bytes, lds_set);
    if (t0 && t1) hipExtLaunchKernelGGL(map_update_ev_kernel, dim3(v.P), dim3(EB), (size_t)g.bytes, s, t0, t1, 0, v);
    else hipLaunchKernelGGL(map_update_ev_kernel, dim3(v.P), dim3(EB), (size_t)g.bytes, s, v);
}

}  // namespace rbpf
