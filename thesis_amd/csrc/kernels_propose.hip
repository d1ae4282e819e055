// kernels_propose.hip -- a3 + a4: the body of Robot.map_update (robot.py:73-114) for all particles,
// fused into one kernel per scan step:
//
//   proposal   K samples ~ N(scan_pose, scan_cov)        robot.py:81  (explicit samples or Philox)
//              motion_pr = mvn.pdf(sample) * 10          robot.py:87  (scipy: eigen pseudo-inverse)
//   weighting  w_k = (1 + sum of log-odds) * motion_pr   robot.py:118-139 (as kernels_weight.hip)
//   moments    shifted weights, weighted mean / covariance, weight increment   robot.py:89-114
//
// One 256-thread workgroup per particle; the K sample poses live in LDS, the per-sample lattice sums
// are exact integers, the moments are a K-term sequential float64 loop in the reference's order.
// A particle whose matcher covariance holds a NaN takes the reference's fallback (robot.py:73-78):
// its pose is kept, its map is updated at that pose and its weight is incremented AFTER the map
// update by bad_weight_kernel.
#include "rbpf_internal.h"
#include "rbpf_device.h"

namespace rbpf {

#ifdef RBPF_STAMPS
#define PSTAMP(k) do { if (threadIdx.x == 0) { long long t_ = clock64(); atomicAdd(&v.stats[8 + (k)], (unsigned long long)(t_ - st_prev)); st_prev = t_; } } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif


static const int KMAX = 32;

// ---- Philox4x32-10 ---------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ void philox4x32(uint32_t ctr[4], uint64_t seed) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(ctr, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
}
__device__ __forceinline__ double u01(uint32_t a, uint32_t b) {   // (0,1], 53 bits
    uint64_t m = ((uint64_t)a << 21) ^ (uint64_t)b >> 11;
    return ((double)(m & ((1ull << 53) - 1)) + 1.0) * (1.0 / 9007199254740992.0);
}
// three standard normals for (seed, stream, p, k)
__device__ void normals3(uint64_t seed, uint32_t stream, uint32_t p, uint32_t k, double z[3]) {
    uint32_t c0[4] = {p, k, stream, 0u}, c1[4] = {p, k, stream, 1u};
    philox4x32(c0, seed); philox4x32(c1, seed);
    const double TWO_PI = 6.283185307179586;
    double r0 = sqrt(-2.0 * log(u01(c0[0], c0[1]))), a0 = TWO_PI * u01(c0[2], c0[3]);
    double r1 = sqrt(-2.0 * log(u01(c1[0], c1[1]))), a1 = TWO_PI * u01(c1[2], c1[3]);
    z[0] = r0 * cos(a0); z[1] = r0 * sin(a0); z[2] = r1 * cos(a1);
}

// ---- symmetric 3x3 eigen-decomposition (cyclic Jacobi) ------------------------------------------------
__device__ void eig3_sym(const double A[9], double w[3], double V[3][3]) {
    double a[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { a[i][j] = 0.5 * (A[3 * i + j] + A[3 * j + i]); V[i][j] = i == j; }
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-20 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(a[p][q]) <= 1e-300) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {            // A <- A J
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {            // A <- J^T A
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) w[i] = a[i][i];
}

struct ProposeArgs {
    const double* match;        // [P][13] scan_pose(3), scan_cov(9), score
    const int32_t* match_of;    // the matcher ran once per group of exact duplicates: row of particle p = match_of[p] (or nullptr)
    const double* guesses;      // [P][K][3] explicit samples, or nullptr
    uint8_t* bad;               // [P] 1 = NaN covariance (robot.py:73)
    uint64_t seed; uint32_t stream;
    double* dbg_w;              // optional [P][K] raw sample weights (tests), or nullptr
};


// The proposal frame of a particle - eigen-decomposition of the matcher covariance, pseudo-inverse square root U, sampling
// matrix A, log normalisation - is a long SERIAL computation (a few thousand float64 instructions).  One thread per
// particle in a kernel of its own: 64 particles share the instruction stream that one lane of the weighting kernel's
// first wave used to run alone (a third of that kernel's instructions).
static const int PREP_W = 24;       // doubles per particle: U[9], A[9], mean[3], log c, bad, -
__global__ __launch_bounds__(64) void propose_prep_kernel(DevView v, ProposeArgs a) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= v.P) return;
    const double* m = a.match + (size_t)(a.match_of ? a.match_of[p] : p) * 13;
    double* o = v.prop_prep + (size_t)p * PREP_W;
    bool bad = false;
    for (int i = 0; i < 9; ++i) bad |= isnan(m[3 + i]);            // robot.py:73
    a.bad[p] = bad ? 1 : 0;
    o[22] = bad ? 1.0 : 0.0;
    if (bad) return;
    double w[3], V[3][3];
    eig3_sym(m + 3, w, V);
    // scipy.stats._multivariate._PSD: eps = 1e6 * eps_f64 * max|eig|; pseudo-inverse, pseudo-determinant
    double mx = fmax(fabs(w[0]), fmax(fabs(w[1]), fabs(w[2])));
    double eps = 1e6 * 2.220446049250313e-16 * mx;
    double log_pdet = 0.0; int rank = 0;
    for (int j = 0; j < 3; ++j) {
        bool keep = w[j] > eps;
        double inv_sqrt = fabs(w[j]) > eps ? sqrt(1.0 / w[j]) : 0.0;
        if (keep) { log_pdet += log(w[j]); ++rank; }
        double sq = w[j] > 0 ? sqrt(w[j]) : 0.0;
        for (int i = 0; i < 3; ++i) { o[3 * i + j] = V[i][j] * inv_sqrt; o[9 + 3 * i + j] = V[i][j] * sq; }
    }
    o[18] = m[0]; o[19] = m[1]; o[20] = m[2];
    o[21] = -0.5 * ((double)rank * 1.8378770664093453 + log_pdet);   // log(2*pi)
}

// ---- the K samples of every particle (robot.py:80-87): one thread per (particle, sample) - the normal deviates, the
//      sample pose, its motion probability, the sine and cosine of its heading and its single-precision frame in home-tile cell
//      coordinates.  (In the weighting kernel these few hundred float64 instructions were a serial prologue of 30 lanes.)
static const int SAMP_W = 256;      // doubles per particle: cos[32], sin[32], x[32], y[32], theta[32], motion pdf[32], float4 frame[32]
__global__ __launch_bounds__(256) void propose_samples_kernel(DevView v, ProposeArgs a) {
    const int gid = blockIdx.x * 256 + threadIdx.x, p = gid >> 5, k = gid & 31, K = v.K;
    if (p >= v.P || k >= K) return;
    const double* pr = v.prop_prep + (size_t)p * PREP_W;
    if (pr[22] != 0.0) return;                                       // robot.py:73-78: no proposal for this particle
    const double mean[3] = {pr[18], pr[19], pr[20]};
    double g[3];
    if (a.guesses) {
        const double* gp = a.guesses + ((size_t)p * K + k) * 3;
        g[0] = gp[0]; g[1] = gp[1]; g[2] = gp[2];
    } else {
        double z[3];
        normals3(a.seed, a.stream, (uint32_t)v.global_id[p], (uint32_t)k, z);
        for (int i = 0; i < 3; ++i) g[i] = mean[i] + ((pr[9 + 3 * i] * z[0] + pr[9 + 3 * i + 1] * z[1]) + pr[9 + 3 * i + 2] * z[2]);
    }
    // robot.py:87: pdf = exp(-0.5 * (rank*log(2pi) + log_pdet + maha)) * 10
    const double d0 = g[0] - mean[0], d1 = g[1] - mean[1], d2 = g[2] - mean[2];
    double maha = 0.0;
    for (int j = 0; j < 3; ++j) { double t = (d0 * pr[j] + d1 * pr[3 + j]) + d2 * pr[6 + j]; maha += t * t; }
    double sn, cs;
    sincos(g[2], &sn, &cs);
    // the home tile's offsets (home_tile(): the tile that holds the matcher's pose; without one the frame is not used)
    int lx, ly, off_x = 0, off_y = 0;
    if (!(v.dim & 1) && tile_of_coord(mean[0], v.tile_len, v.R, lx) && tile_of_coord(mean[1], v.tile_len, v.R, ly)) {
        off_x = v.dim / 2 - lx * v.dim; off_y = v.dim / 2 - ly * v.dim;
    }
    const double inv_cs = (double)v.dim / v.tile_len;
    double* o = v.prop_samp + (size_t)p * SAMP_W;
    o[k] = cs; o[32 + k] = sn; o[64 + k] = g[0]; o[96 + k] = g[1]; o[128 + k] = g[2];
    o[160 + k] = exp(pr[21] - 0.5 * maha) * 10;
    reinterpret_cast<float4*>(o + 192)[k] = make_float4((float)(cs * inv_cs), (float)(sn * inv_cs), (float)(g[0] * inv_cs + (double)off_x), (float)(g[1] * inv_cs + (double)off_y));
}

// ---- weighting: K gathers per beam (robot.py:118-139).  A lane is a sample, 32 lanes one beam: the samples of a beam end
//      on neighbouring cells, so the 64 byte loads of an instruction fall into a dozen cache lines; a lane does eight beams
//      at a time - eight cell addresses in the particle's home tile formed without a branch, the eight byte loads issued
//      back to back, then added.  The address comes from SINGLE-precision arithmetic in home-tile cell coordinates (the fused
//      multiply-adds are exact in the sense of the budget below whether they are packed two beams to an instruction or not).
//      Error budget, in cells, for |rotated beam| <= 1.5 dim and |result| < dim: the conversions of x, y, cos / cell,
//      sin / cell cost 4 * 1.5 dim * 2^-24, the offset's rounding dim * 2^-24, the two fused multiply-adds 2.5 dim * 2^-24 and
//      dim * 2^-24: 9.5 dim * 2^-24 = 4.5e-4 at dim = 800, 1.2e-3 at 2048.  The address is taken only when the point lies
//      more than WSAFE (1e-3 for dim <= 1024, else 2e-3) inside its cell on both axes: the cell index is then the
//      reference's (gridmap.py:119-128 on the float64 point).  The other 0.4 - 0.8 % of the look-ups - and every look-up of a
//      beam outside the budget's premise, NaN in the host's list - are redone in float64 the reference's way
//      (lookup_cell_home) from a queue, after the fast ones.
//      Shared by propose_weight_kernel (every scan step) and weight_samples_product_kernel (rbpf_weight_samples: the
//      per-sample test entry), so that the tests of the entry are tests of the product's look-ups.
struct WeightFrame {
    const double* s_c; const double* s_s; const double (*s_g)[3]; const float4* s_q; int* s_sum;
    const int* s_tab; const unsigned long long* s_base;
    uint32_t* s_redo; int* s_nredo;                    // queue of the look-ups that go the float64 way: beam << 5 | sample
    float* s_bx; float* s_by;                          // the beams that count (robot.py:130), single precision: a chunk of DevView::wsel_x / wsel_y
};
static const int REDO_CAP = 2048;
static const int WB_CAP = 1536;                        // beams staged at a time
__device__ __forceinline__ void weight_beams(const DevView& v, const HomeTile& home, const WeightFrame& f, int K, int tid, long long& st_prev) {
    const double* const s_c = f.s_c; const double* const s_s = f.s_s; const double (*s_g)[3] = f.s_g; const float4* const s_q = f.s_q;
    int* const s_sum = f.s_sum; const int* const s_tab = f.s_tab; const unsigned long long* const s_base = f.s_base;
    // (the home tile's address is the same for the whole workgroup: kept in scalar registers, so that a look-up is a
    // 32-bit offset from a scalar base - no 64-bit address arithmetic per load)
    typedef __attribute__((address_space(1))) const int8_t global_i8;
    const global_i8* hbase;
    {
        const unsigned long long hb = (unsigned long long)(home.ok ? home.base : v.pool);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)hb), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hb >> 32));
        hbase = (const global_i8*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    }
    const float WSAFE = v.wsafe_override >= 0.0f ? v.wsafe_override : (v.dim <= 1024 ? 1e-3f : 2e-3f);   // (the override is a test knob: RBPF_WSAFE)
    const bool f32_ok = home.ok && v.dim <= 2048;
    // Work split: a lane is a SAMPLE (32 lanes = the up to 32 samples of one beam, two beams per wave, eight per pass of the
    // workgroup).  The samples of a beam end within a few cells of each other, so the 64 byte loads of one instruction fall into
    // a dozen cache lines instead of one per lane - the texture unit takes a line per cycle - and a lane keeps its sample's frame
    // in registers and its sum to itself.  The beams that count (robot.py:130) are packed into LDS first, single precision; a
    // beam outside the error budget's premise is stored as NaN: every look-up of it fails the fast test and goes the exact way.
    const int k = tid & 31, grp = tid >> 5;
    const bool live = k < K;
    typedef float wf2 __attribute__((ext_vector_type(2)));
    const float4 q = s_q[min(k, K - 1)];
    const wf2 qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z}, qw = {q.w, q.w};
    int acc = 0;
    const int n_all = v.n_wsel, npad_all = (n_all + 63) & ~63;        // (the list is padded with NaN to whole passes: fails the fast test, dropped on the slow way)
    for (int c0 = 0; c0 < npad_all; c0 += WB_CAP) {
        const int n = min(n_all - c0, WB_CAP), npad = min(npad_all - c0, WB_CAP);
        __syncthreads();
        for (int j = 4 * tid; j < npad; j += 4 * BLOCK) {
            float4 x4 = *reinterpret_cast<const float4*>(v.wsel_x + c0 + j);
            if (!f32_ok) x4 = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            *reinterpret_cast<float4*>(f.s_bx + j) = x4;
            *reinterpret_cast<float4*>(f.s_by + j) = *reinterpret_cast<const float4*>(v.wsel_y + c0 + j);
        }
        __syncthreads();
        PSTAMP(1);
        // a group of 32 lanes takes eight neighbouring beams per pass: two 16-byte LDS reads per coordinate, packed arithmetic on pairs
        if (live)
        for (int j0 = 8 * grp; j0 < npad; j0 += 8 * (BLOCK / 32)) {
            uint32_t addr[8]; bool fast[8];
            const float4 xa = *reinterpret_cast<const float4*>(f.s_bx + j0), xb = *reinterpret_cast<const float4*>(f.s_bx + j0 + 4);
            const float4 ya = *reinterpret_cast<const float4*>(f.s_by + j0), yb = *reinterpret_cast<const float4*>(f.s_by + j0 + 4);
            const wf2 bx2[4] = {{xa.x, xa.y}, {xa.z, xa.w}, {xb.x, xb.y}, {xb.z, xb.w}}, by2[4] = {{ya.x, ya.y}, {ya.z, ya.w}, {yb.x, yb.y}, {yb.z, yb.w}};
            uint32_t redo = 0;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const wf2 cx2 = __builtin_elementwise_fma(qx, bx2[h], __builtin_elementwise_fma(-qy, by2[h], qz));   // lidar.py:123, in cells
                const wf2 cy2 = __builtin_elementwise_fma(qy, bx2[h], __builtin_elementwise_fma(qx, by2[h], qw));
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int u = 2 * h + e;
                    const float cx = e ? cx2.y : cx2.x, cy = e ? cy2.y : cy2.x;
                    int ix, iy;                                       // floor to int32 in one instruction
                    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ix) : "v"(cx));
                    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iy) : "v"(cy));
                    const float rx = __builtin_amdgcn_fractf(cx), ry = __builtin_amdgcn_fractf(cy);   // x - floor(x)
                    // (bit-wise &: the conditions are cheap, a short-circuit would be a branch per look-up)
                    fast[u] = ((int)(fminf(rx, ry) > WSAFE) & (int)(fmaxf(rx, ry) < 1.0f - WSAFE) & (int)(max((unsigned)ix, (unsigned)iy) < (unsigned)v.dim)) != 0;
                    addr[u] = fast[u] ? __umul24((uint32_t)ix, (uint32_t)v.dim) + (uint32_t)iy : 0u;
#ifdef WEIGHT_ABLATE_ADDR
                    addr[u] &= 63u;                 // diagnostic build: every look-up falls into one cache line
#endif
                    redo |= fast[u] ? 0u : 1u << u;
                }
            }
            int val[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) val[u] = hbase[addr[u]];
            // (the empty statement keeps the eight loads together: the compiler would otherwise sink each into the
            // select that uses it and wait for it there, one load at a time)
            asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]), "+v"(val[6]), "+v"(val[7]));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += fast[u] ? val[u] : 0;
            // The look-ups that need the reference's float64 operations (under 1 % of them, but some lane of a wave has one in
            // most passes) are queued and done after the loop with all lanes busy; a full queue: done here.
            if (redo) {
                for (uint32_t r = redo; r; r &= r - 1) if (j0 + __ffs(r) - 1 >= n) redo &= ~(r & -r);     // (padding)
                const int nr = __popc(redo);
                int slot = nr ? atomicAdd(f.s_nredo, nr) : 0;
                const bool queued = slot + nr <= REDO_CAP;
                if (nr && !queued && slot <= REDO_CAP) f.s_nredo[1] = slot; // the first push that does not fit: the queue ends here
                for (uint32_t r = redo; r; r &= r - 1) {
                    const int b = (int)v.wsel_idx[c0 + j0 + __ffs(r) - 1];
                    if (queued) { f.s_redo[slot++] = (uint32_t)b << 5 | (uint32_t)k; continue; }
                    const double x = v.bx[b], y = v.by[b];
                    const double gx = (s_c[k] * x + (-s_s[k]) * y) + s_g[k][0];   // lidar.py:123
                    const double gy = (s_s[k] * x + s_c[k] * y) + s_g[k][1];
                    int vv;
                    if (lookup_cell_home(v, home, s_tab, s_base, gx, gy, vv)) acc += vv;
                }
            }
        }
    }
    if (live && acc) atomicAdd(&s_sum[k], acc);
    __syncthreads();
    PSTAMP(2);
    const int n_redo = f.s_nredo[0] <= REDO_CAP ? f.s_nredo[0] : f.s_nredo[1];
    for (int i = tid; i < n_redo; i += BLOCK) {                       // float64, the reference's operations
        const uint32_t e = f.s_redo[i];
        const int b = (int)(e >> 5), kk = (int)(e & 31u);
        const double x = v.bx[b], y = v.by[b];
        const double gx = (s_c[kk] * x + (-s_s[kk]) * y) + s_g[kk][0];   // lidar.py:123
        const double gy = (s_s[kk] * x + s_c[kk] * y) + s_g[kk][1];
        int vv;
        if (lookup_cell_home(v, home, s_tab, s_base, gx, gy, vv)) atomicAdd(&s_sum[kk], vv);
    }
    PSTAMP(3);
}

__global__ __launch_bounds__(BLOCK) void propose_weight_kernel(DevView v, ProposeArgs a) {
    __shared__ double s_c[KMAX], s_s[KMAX], s_g[KMAX][3], s_pr[KMAX], s_w[KMAX];
    __shared__ int s_sum[KMAX];
    __shared__ float4 s_q[KMAX];                       // single-precision sample frame in home-tile cells: (cos, sin) / cell, offset x, y
    __shared__ int s_tab[49];
    __shared__ unsigned long long s_base[49];          // byte offset of each lattice tile in the pool, ~0 = none
    __shared__ double s_mean[3];
    __shared__ double s_mom[16];                       // moments: mean[3], norm, sig[9], min_w
    __shared__ int s_bad, s_nredo[2];
    __shared__ uint32_t s_redo[REDO_CAP];
    __shared__ __align__(16) float s_bx[WB_CAP], s_by[WB_CAP];
    const int p = blockIdx.x, tid = threadIdx.x, K = v.K;
    long long st_prev = clock64();
    if (tid == 0) { s_nredo[0] = 0; s_nredo[1] = 0; }
    const int LL = v.L * v.L;
    {   // the frame propose_prep_kernel left
        const double* pr = v.prop_prep + (size_t)p * PREP_W;
        if (tid < 3) s_mean[tid] = pr[18 + tid];
        if (tid == 0) s_bad = pr[22] != 0.0;
    }
    const int32_t* tab = v.tile_tab + (size_t)v.slot[p] * LL;
    for (int i = tid; i < LL; i += BLOCK) {
        const int t = tab[i];
        s_tab[i] = t;
        s_base[i] = t >= 0 ? (unsigned long long)t * (unsigned long long)v.dim * (unsigned long long)v.dim : ~0ull;
    }
    __syncthreads();
    if (s_bad) {
        if (tid == 0) { v.upd_pose[p] = v.px[p]; v.upd_pose[v.P + p] = v.py[p]; v.upd_pose[2 * v.P + p] = v.pth[p]; }
        return;
    }
    const HomeTile home = home_tile(v, s_tab, s_mean[0], s_mean[1]);
    if (tid < K) {                                                   // the samples propose_samples_kernel left
        const double* sp = v.prop_samp + (size_t)p * SAMP_W;
        s_c[tid] = sp[tid]; s_s[tid] = sp[32 + tid]; s_g[tid][0] = sp[64 + tid]; s_g[tid][1] = sp[96 + tid]; s_g[tid][2] = sp[128 + tid];
        s_pr[tid] = sp[160 + tid];
        s_q[tid] = reinterpret_cast<const float4*>(sp + 192)[tid];
        s_sum[tid] = 0;
    }
    __syncthreads();

    {
        const WeightFrame wf{s_c, s_s, s_g, s_q, s_sum, s_tab, s_base, s_redo, s_nredo, s_bx, s_by};
        PSTAMP(0);
        weight_beams(v, home, wf, K, tid, st_prev);
    }
    __syncthreads();
    if (tid < K) {
        double obs = v.inv_quantum > 0 ? (v.inv_quantum + (double)s_sum[tid]) / v.inv_quantum
                                       : 1.0 + (double)s_sum[tid] * v.quantum;
        s_w[tid] = obs * s_pr[tid];                                   // robot.py:138
        if (a.dbg_w) a.dbg_w[(size_t)p * K + tid] = s_w[tid];
    }
    __syncthreads();

    // ---- moments (robot.py:89-114): every accumulator is a sequential float64 sum over the K samples in the reference's
    //      order; the 13 accumulators are independent of each other, so each gets a lane ------------------------------
    if (tid < 64) {
        double mw = tid < K ? s_w[tid] : 1.7976931348623157e308;                   // min over the samples (any order)
        for (int off = 32; off > 0; off >>= 1) mw = fmin(mw, __shfl_xor(mw, off, 64));
        const double min_w = mw;
        if (tid < K) s_w[tid] = (s_w[tid] - min_w) + 1e-2;                          // robot.py:96
        __builtin_amdgcn_wave_barrier();
        if (tid < 4) {                                                             // lanes 0-2: mean[i] * norm, lane 3: norm
            double acc = 0.0;
            for (int k = 0; k < K; ++k) acc = acc + (tid < 3 ? s_g[k][tid] * s_w[k] : s_w[k]);
            s_mom[tid] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        const double norm = s_mom[3];
        const double mean0 = s_mom[0] / norm, mean1 = s_mom[1] / norm, mean2 = s_mom[2] / norm;
        if (tid < 9) {                                                             // lane 3 i + j: sig[i][j]
            const int i = tid / 3, j = tid % 3;
            const double mi = i == 0 ? mean0 : i == 1 ? mean1 : mean2, mj = j == 0 ? mean0 : j == 1 ? mean1 : mean2;
            double acc = 0.0;
            for (int k = 0; k < K; ++k) acc = acc + ((s_g[k][i] + (-mi)) * (s_g[k][j] + (-mj))) * s_w[k];
            v.cov[(size_t)tid * v.P + p] = acc / norm;                             // robot.py:107,110
        }
        if (tid == 0) {
            const double total = norm + min_w * (double)K;                         // robot.py:108
            v.px[p] = mean0; v.py[p] = mean1; v.pth[p] = mean2;                    // robot.py:111-113
            v.weight[p] = total + v.weight[p];                                     // robot.py:114
            v.upd_pose[p] = mean0; v.upd_pose[v.P + p] = mean1; v.upd_pose[2 * v.P + p] = mean2;   // robot.py:115
        }
    }
    PSTAMP(4);
}

// rbpf_weight_samples (test entry for a4): explicit sample poses and motion probabilities, the PRODUCT's look-ups.  The
// home tile is the tile of the first sample pose (the product takes the tile of the matcher's pose).
__global__ __launch_bounds__(BLOCK) void weight_samples_product_kernel(DevView v, const double* __restrict__ guesses,
                                                                       const double* __restrict__ prs, int K, double* __restrict__ out_w) {
    __shared__ double s_c[KMAX], s_s[KMAX], s_g[KMAX][3];
    __shared__ int s_sum[KMAX];
    __shared__ float4 s_q[KMAX];
    __shared__ int s_tab[49];
    __shared__ unsigned long long s_base[49];
    __shared__ int s_nredo[2];
    __shared__ uint32_t s_redo[REDO_CAP];
    __shared__ __align__(16) float s_bx[WB_CAP], s_by[WB_CAP];
    const int p = blockIdx.x, tid = threadIdx.x;
    long long st_prev = clock64();
    if (tid == 0) { s_nredo[0] = 0; s_nredo[1] = 0; }
    const int LL = v.L * v.L;
    const int32_t* tab = v.tile_tab + (size_t)v.slot[p] * LL;
    for (int i = tid; i < LL; i += BLOCK) {
        const int t = tab[i];
        s_tab[i] = t;
        s_base[i] = t >= 0 ? (unsigned long long)t * (unsigned long long)v.dim * (unsigned long long)v.dim : ~0ull;
    }
    __syncthreads();
    const double* g0 = guesses + (size_t)p * K * 3;
    const HomeTile home = home_tile(v, s_tab, g0[0], g0[1]);
    const double inv_cs = (double)v.dim / v.tile_len;
    if (tid < K) {
        const double* gp = guesses + ((size_t)p * K + tid) * 3;
        double sn, cs;
        sincos(gp[2], &sn, &cs);
        s_c[tid] = cs; s_s[tid] = sn; s_g[tid][0] = gp[0]; s_g[tid][1] = gp[1]; s_g[tid][2] = gp[2];
        s_q[tid] = make_float4((float)(cs * inv_cs), (float)(sn * inv_cs), (float)(gp[0] * inv_cs + (double)home.off_x), (float)(gp[1] * inv_cs + (double)home.off_y));
        s_sum[tid] = 0;
    }
    __syncthreads();
    {
        const WeightFrame wf{s_c, s_s, s_g, s_q, s_sum, s_tab, s_base, s_redo, s_nredo, s_bx, s_by};
        weight_beams(v, home, wf, K, tid, st_prev);
    }
    __syncthreads();
    if (tid < K) {
        double obs = v.inv_quantum > 0 ? (v.inv_quantum + (double)s_sum[tid]) / v.inv_quantum : 1.0 + (double)s_sum[tid] * v.quantum;
        out_w[(size_t)p * K + tid] = obs * prs[(size_t)p * K + tid];
    }
}
void launch_weight_samples_product(const DevView& v, const double* d_guesses, const double* d_prs, int K, double* d_out_w, hipStream_t s) {
    hipLaunchKernelGGL(weight_samples_product_kernel, dim3(v.P), dim3(BLOCK), 0, s, v, d_guesses, d_prs, K, d_out_w);
}

// robot.py:75-77 for particles on the NaN branch: weight += (1 + sum log-odds at the latest pose) * 1,
// evaluated on the map AFTER its update.
__global__ __launch_bounds__(BLOCK) void bad_weight_kernel(DevView v, const uint8_t* __restrict__ bad) {
    if (!bad[blockIdx.x]) return;
    nan_branch_weight(v, blockIdx.x, threadIdx.x, BLOCK);
}

void launch_propose_weight(const DevView& v, const double* d_match, const int32_t* d_match_of, const double* d_guesses, uint8_t* d_bad,
                           uint64_t seed, uint32_t stream, double* d_dbg_w, hipStream_t s) {
    ProposeArgs a{d_match, d_match_of, d_guesses, d_bad, seed, stream, d_dbg_w};
    hipLaunchKernelGGL(propose_prep_kernel, dim3((v.P + 63) / 64), dim3(64), 0, s, v, a);
    hipLaunchKernelGGL(propose_samples_kernel, dim3((v.P * 32 + 255) / 256), dim3(256), 0, s, v, a);
    hipLaunchKernelGGL(propose_weight_kernel, dim3(v.P), dim3(BLOCK), 0, s, v, a);
}
void launch_bad_weight(const DevView& v, const uint8_t* d_bad, hipStream_t s) {
    hipLaunchKernelGGL(bad_weight_kernel, dim3(v.P), dim3(BLOCK), 0, s, v, d_bad);
}

}  // namespace rbpf
