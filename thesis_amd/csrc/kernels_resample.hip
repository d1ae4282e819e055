// kernels_resample.hip -- a8 + a9: resample (main.py:46-79) and Robot.copy (robot.py:141-149).
//
// The reference draws ONE uniform and walks the particles once with a sequential longdouble
// running sum (main.py:61-64).  Here the running sums are a parallel prefix scan in double-double
// arithmetic (about 106 significant bits), i.e. closer to the exact sums than the reference's 64-bit
// mantissa, so the integer ancestor indices agree unless a quotient lies within ~1e-14 of an integer.
//   T_i   = floor((cum_i - start) / slice) + 1          survivors up to and including particle i
//   idx_j = first i with T_i > j                        ancestor of new particle j
// Particles keep their map "slot"; only duplicated ancestors cost a tile copy (as in the reference,
// main.py:70-74), into the slot of a particle that died, restricted to the bounding boxes of the
// cells ever written.
#include <limits.h>

#include "rbpf_internal.h"

namespace rbpf {

static const int PLAN_THREADS = 1024;

__device__ __forceinline__ dd dd_two_prod(double a, double b) {
    double p = a * b;
    double e = __fma_rn(a, b, -p);          // exact error of the product (explicit FMA)
    return {p, e};
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b) {
    dd p = dd_two_prod(a.hi, b);
    double lo = p.lo + a.lo * b;
    double hi = p.hi + lo;
    return {hi, lo - (hi - p.hi)};
}
__device__ __forceinline__ dd dd_sub(dd a, dd b) { return dd_add(a, dd{-b.hi, -b.lo}); }
__device__ __forceinline__ dd dd_div_d(dd a, double b) {
    double q1 = a.hi / b;
    dd r = dd_sub(a, dd_two_prod(q1, b));
    double q2 = (r.hi + r.lo) / b;
    double hi = q1 + q2;
    return {hi, q2 - (hi - q1)};
}
// floor(a / b) for a, b double-double, b > 0
__device__ long long dd_floor_div(dd a, dd b) {
    double q0 = floor(a.hi / b.hi);
    if (!(fabs(q0) < 9.0e15)) return q0 > 0 ? LLONG_MAX / 2 : LLONG_MIN / 2;
    for (int it = 0; it < 4; ++it) {
        dd r = dd_sub(a, dd_mul_d(b, q0));              // a - q0*b
        if (r.hi < 0 || (r.hi == 0 && r.lo < 0)) { q0 -= 1.0; continue; }
        dd r2 = dd_sub(r, b);
        if (r2.hi > 0 || (r2.hi == 0 && r2.lo >= 0)) { q0 += 1.0; continue; }
        break;
    }
    return (long long)q0;
}

struct ResampleArgs {
    int P;
    const double* w;          // [P] weights (global vector in the multi-GPU case)
    double u;                 // uniform in [0,1)
    double spread;            // main.py:50
    int32_t* T;               // [P] survivors up to and including i
    int32_t* did;             // [1]
    int32_t* err;
};

// ---- kernel 1: trigger test, weight shift, double-double prefix sums, T array --------------------------
__device__ void resample_plan_stage(const ResampleArgs& a) {
    __shared__ double s_red[PLAN_THREADS];
    __shared__ double s_hi[PLAN_THREADS], s_lo[PLAN_THREADS];
    __shared__ double s_max, s_min, s_min2;
    const int tid = threadIdx.x, nt = PLAN_THREADS;
    const int chunk = (a.P + nt - 1) / nt;
    const int i0 = min(tid * chunk, a.P), i1 = min(i0 + chunk, a.P);

    // main.py:50  max(weights) - min(weights) > 200
    double mx = -INFINITY, mn = INFINITY;
    for (int i = i0; i < i1; ++i) { double w = a.w[i]; mx = fmax(mx, w); mn = fmin(mn, w); }
    const int lane = tid & 63, wave = tid >> 6, nw = nt / 64;
    for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o, 64)); mn = fmin(mn, __shfl_xor(mn, o, 64)); }
    if (lane == 0) { s_red[wave] = mx; s_red[nw + wave] = mn; }
    __syncthreads();
    if (tid == 0) {
        double a = s_red[0], b = s_red[nw];
        for (int w = 1; w < nw; ++w) { a = fmax(a, s_red[w]); b = fmin(b, s_red[nw + w]); }
        s_max = a; s_min = b;
    }
    __syncthreads();
    const bool go = (s_max - s_min) > a.spread;
    if (!go) {
        for (int i = i0; i < i1; ++i) a.T[i] = i + 1;           // identity
        if (tid == 0) *a.did = 0;
        return;
    }
    // main.py:53-55: -inf -> 0, then every non-zero weight += |min| when the minimum is negative
    double mn2 = INFINITY;
    for (int i = i0; i < i1; ++i) { double w = a.w[i]; if (w == -INFINITY) w = 0.0; mn2 = fmin(mn2, w); }
    for (int o = 32; o > 0; o >>= 1) mn2 = fmin(mn2, __shfl_xor(mn2, o, 64));
    if (lane == 0) s_red[wave] = mn2;
    __syncthreads();
    if (tid == 0) { double b = s_red[0]; for (int w = 1; w < nw; ++w) b = fmin(b, s_red[w]); s_min2 = b; }
    __syncthreads();
    const double shift = s_min2 < 0 ? fabs(s_min2) : 0.0;
    auto adj = [&](double w) -> dd {
        if (w == -INFINITY) return dd{0.0, 0.0};
        if (shift != 0.0 && w != 0.0) return dd_two_sum(w, shift);
        return dd{w, 0.0};
    };

    // chunk totals, then an inclusive scan over the 1024 totals: shuffles inside a wave, the 16 wave totals through LDS
    dd tot = {0.0, 0.0};
    for (int i = i0; i < i1; ++i) tot = dd_add(tot, adj(a.w[i]));
    dd incl = tot;
    for (int off = 1; off < 64; off <<= 1) {
        const dd n = {__shfl_up(incl.hi, off, 64), __shfl_up(incl.lo, off, 64)};
        if (lane >= off) incl = dd_add(n, incl);
    }
    if (lane == 63) { s_hi[wave] = incl.hi; s_lo[wave] = incl.lo; }
    __syncthreads();
    dd wbase = {0.0, 0.0}, total = {0.0, 0.0};
    for (int w = 0; w < nw; ++w) {
        const dd x = {s_hi[w], s_lo[w]};
        if (w < wave) wbase = dd_add(wbase, x);
        total = dd_add(total, x);
    }
    dd prev = {__shfl_up(incl.hi, 1, 64), __shfl_up(incl.lo, 1, 64)};
    if (lane == 0) prev = dd{0.0, 0.0};
    dd run = dd_add(wbase, prev);                            // sum of everything before this thread's chunk
    const dd slice = dd_div_d(total, (double)a.P);            // main.py:57
    const dd start = dd_mul_d(slice, a.u);                    // main.py:59
    bool bad = !(slice.hi > 0) || !isfinite(slice.hi);
    for (int i = i0; i < i1; ++i) {
        run = dd_add(run, adj(a.w[i]));                       // main.py:62
        long long t = bad ? 0 : dd_floor_div(dd_sub(run, start), slice) + 1;   // main.py:63
        if (t < 0) t = 0;
        if (t > a.P) t = (long long)a.P + 1;
        a.T[i] = (int32_t)t;
    }
    __syncthreads();
    if (tid == 0) {
        *a.did = 1;
        if (a.T[a.P - 1] != a.P) atomicCAS(a.err, 0, RBPF_ESTATE);   // main.py:66-67 AssertionError
    }
}

// ---- kernel 2: ancestors by binary search over T -----------------------------------------------------
__device__ __forceinline__ void resample_expand_one(int P, const int32_t* __restrict__ T, int32_t* __restrict__ idx, int j) {
    int lo = 0, hi = P - 1;                                  // first i with T[i] > j
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (T[mid] > j) hi = mid; else lo = mid + 1;
    }
    idx[j] = lo;
}

// ---- kernel 3: pair duplicated ancestors with dead particles' map slots --------------------------------
struct PairArgs {
    int P;
    const int32_t* T; const int32_t* idx; const int32_t* did;
    const int32_t* slot_old; int32_t* slot_new;
    int32_t* dead_list;       // [P] scratch
    int32_t* jobs;            // [P][2] src slot, dst slot
    int32_t* n_jobs;          // [1] (+ queue head behind it)
    int32_t* err;
};

// exclusive prefix sum over the 1024 threads: shuffles inside a wave, 16 wave totals through LDS (two barriers)
__device__ int block_exclusive_scan_1024(int val, int* s_buf, int tid, int& total) {
    const int lane = tid & 63, wave = tid >> 6;
    int incl = val;
    for (int off = 1; off < 64; off <<= 1) { int n = __shfl_up(incl, off, 64); if (lane >= off) incl += n; }
    __syncthreads();                                     // s_buf may still be read from a previous call
    if (lane == 63) s_buf[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < PLAN_THREADS / 64; ++w) { const int x = s_buf[w]; if (w < wave) base += x; tot += x; }
    total = tot;
    return base + incl - val;
}

__device__ void resample_pair_stage(const PairArgs& a) {
    __shared__ int s_buf[PLAN_THREADS];
    const int tid = threadIdx.x, nt = PLAN_THREADS;
    const int chunk = (a.P + nt - 1) / nt;
    const int i0 = min(tid * chunk, a.P), i1 = min(i0 + chunk, a.P);
    if (!*a.did) {
        for (int i = i0; i < i1; ++i) a.slot_new[i] = a.slot_old[i];
        if (tid == 0) { a.n_jobs[0] = 0; a.n_jobs[1] = 0; }
        return;
    }
    // dead particles (no survivor): cnt_i = T_i - T_{i-1} == 0
    int n_dead = 0;
    for (int i = i0; i < i1; ++i) n_dead += (a.T[i] - (i > 0 ? a.T[i - 1] : 0)) == 0;
    int total_dead;
    int base = block_exclusive_scan_1024(n_dead, s_buf, tid, total_dead);
    for (int i = i0; i < i1; ++i)
        if ((a.T[i] - (i > 0 ? a.T[i - 1] : 0)) == 0) a.dead_list[base++] = i;
    __syncthreads();
    // duplicates: new particle j whose ancestor equals that of j-1 (main.py:70-74); idx < 0 marks a particle that
    // arrives from another rank: it takes a dead particle's map slot too, but nothing is copied for it here
    int n_dup = 0, n_job = 0;
    for (int j = i0; j < i1; ++j) {
        const bool in = a.idx[j] < 0, dup = !in && j > 0 && a.idx[j] == a.idx[j - 1];
        n_dup += in || dup; n_job += dup;
    }
    int total_dup, total_job;
    int dbase = block_exclusive_scan_1024(n_dup, s_buf, tid, total_dup);
    __syncthreads();
    int jbase = block_exclusive_scan_1024(n_job, s_buf, tid, total_job);
    for (int j = i0; j < i1; ++j) {
        const bool in = a.idx[j] < 0, dup = !in && j > 0 && a.idx[j] == a.idx[j - 1];
        if (in || dup) {
            int dst = a.slot_old[a.dead_list[dbase]];
            if (dup) { a.jobs[2 * jbase] = a.slot_old[a.idx[j]]; a.jobs[2 * jbase + 1] = dst; ++jbase; }
            a.slot_new[j] = dst;
            ++dbase;
        } else {
            a.slot_new[j] = a.slot_old[a.idx[j]];
        }
    }
    if (tid == 0 && total_dup != total_dead) atomicCAS(a.err, 0, RBPF_ESTATE);
    if (tid == 0) { a.n_jobs[0] = total_job; a.n_jobs[1] = 0; }
}

// ---- kernel 4: permute the small per-particle state, weights <- 1.0 (main.py:77-78) ---------------------
struct GatherArgs {
    int P;
    const int32_t* idx; const int32_t* did;
    const double *px, *py, *pth, *cov, *w;
    double *px2, *py2, *pth2, *cov2, *w2;
    const int32_t* T; int32_t* dup_of;  // dup_of[j] = first new particle with j's ancestor: the copies of one ancestor are
                                        // exact duplicates (state and map) until the next proposal draws their samples
};
__device__ __forceinline__ void resample_gather_one(const GatherArgs& a, int j) {
    const int i = a.idx[j];
    if (a.dup_of) a.dup_of[j] = (i >= 0 && *a.did) ? (i > 0 ? a.T[i - 1] : 0) : j;
    if (i < 0) return;                 // arrives from another rank: filled by the unpack kernel
    a.px2[j] = a.px[i]; a.py2[j] = a.py[i]; a.pth2[j] = a.pth[i];
#pragma unroll
    for (int k = 0; k < 9; ++k) a.cov2[(size_t)k * a.P + j] = a.cov[(size_t)k * a.P + i];
    a.w2[j] = *a.did ? 1.0 : a.w[i];
}

// ---- kernel 5: tile copies -----------------------------------------------------------------------------
struct CopyArgs {
    DevView v;
    const int32_t* jobs; int32_t* n_jobs;     // n_jobs[0] = jobs, n_jobs[1] = queue head
    int32_t* pending_free; int32_t* n_pending;
    int gather; GatherArgs ga;                // the state permutation rides along (a thread per particle before the tile jobs)
};

template <int NT>
__device__ __forceinline__ void copy_rows(int8_t* dst, const int8_t* src, int dim, int x0, int x1, int y0, int y1,
                                          int tid) {
    // rows x0..x1, columns rounded out to 16-byte groups; src == nullptr zero-fills; four loads in flight per thread
    const int ya = y0 & ~15, yb = min((y1 | 15) + 1, dim);
    const int per_row = (yb - ya) / 16;
    const int n = (x1 - x0 + 1) * per_row;
    for (int q0 = tid; q0 < n; q0 += 4 * NT) {
        uint4 val[4]; size_t off[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = q0 + k * NT;
            const int qq = q < n ? q : q0;
            off[k] = (size_t)(x0 + qq / per_row) * dim + ya + (qq % per_row) * 16;
            val[k] = src ? *reinterpret_cast<const uint4*>(src + off[k]) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (q0 + k * NT < n) *reinterpret_cast<uint4*>(dst + off[k]) = val[k];
    }
}

// occupancy-bitmask words of rows x0..x1, columns y0..y1 of a tile; src == nullptr zero-fills; eight loads in flight per thread
template <int NT>
__device__ __forceinline__ void copy_occ_rows(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, int ow, int x0, int x1,
                                              int y0, int y1, int tid) {
    const int wa = y0 >> 5, per_row = (y1 >> 5) - wa + 1;
    const int n = (x1 - x0 + 1) * per_row;
    for (int q0 = tid; q0 < n; q0 += 8 * NT) {
        uint32_t val[8]; size_t off[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = q0 + k * NT;
            const int qq = q < n ? q : q0;
            off[k] = (size_t)(x0 + qq / per_row) * ow + wa + qq % per_row;
            val[k] = src ? src[off[k]] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (q0 + k * NT < n) dst[off[k]] = val[k];
    }
}

// 1024 threads per job: a job is one tile box (~100 KB), its duration is set by the copies one workgroup keeps in flight
static const int COPY_BLOCK = 1024;
__global__ __launch_bounds__(COPY_BLOCK) void resample_copy_kernel(CopyArgs a) {
    __shared__ int s_item, s_tile, s_ts[49], s_td[49];
    const DevView& v = a.v;
    const int tid = threadIdx.x;
    const int LL = v.L * v.L;
    const size_t cells = (size_t)v.dim * v.dim;
    if (a.gather)
        for (int j = blockIdx.x * COPY_BLOCK + tid; j < a.ga.P; j += gridDim.x * COPY_BLOCK) resample_gather_one(a.ga, j);
    // a workgroup's first job is its own index; only the jobs beyond the grid go through the queue head (one returning atomic
    // on one address per workgroup and job is a chain of L2 round trips that 512 workgroups would stand in line for)
    for (int round = 0;; ++round) {
        int job = blockIdx.x;
        if (round > 0) {
            __syncthreads();
            if (tid == 0) s_item = (int)gridDim.x + atomicAdd(&a.n_jobs[1], 1);
            __syncthreads();
            job = s_item;
        }
        if (job >= a.n_jobs[0]) return;
        const int src_slot = a.jobs[2 * job], dst_slot = a.jobs[2 * job + 1];
        if (tid < LL) { s_ts[tid] = v.tile_tab[(size_t)src_slot * LL + tid]; s_td[tid] = v.tile_tab[(size_t)dst_slot * LL + tid]; }
        __syncthreads();
      for (int pos = 0; pos < LL; ++pos) {
        const int ts = s_ts[pos];
        int td = s_td[pos];
        if (ts < 0 && td < 0) continue;
        if (ts >= 0) {
            int sb[4], db[4] = {INT_MAX, -1, INT_MAX, -1};
            for (int k = 0; k < 4; ++k) sb[k] = v.tile_bbox[4 * ts + k];
            if (td < 0) {                                     // destination has no tile here: allocate
                if (tid == 0) {
                    int idx = atomicSub(v.free_top, 1) - 1;
                    if (idx < 0) { atomicAdd(v.free_top, 1); atomicCAS(v.err, 0, RBPF_ENOMEM); s_tile = -1; }
                    else { s_tile = v.free_stack[idx]; v.tile_tab[(size_t)dst_slot * LL + pos] = s_tile; }
                }
                __syncthreads();
                td = s_tile;
                if (td < 0) continue;
            } else {
                for (int k = 0; k < 4; ++k) db[k] = v.tile_bbox[4 * td + k];
            }
            // copy the union of both written regions (outside its box a tile is zero)
            int x0 = min(sb[0], db[0]), x1 = max(sb[1], db[1]), y0 = min(sb[2], db[2]), y1 = max(sb[3], db[3]);
            if (x0 <= x1 && y0 <= y1) {
                copy_rows<COPY_BLOCK>(v.pool + (size_t)td * cells, v.pool + (size_t)ts * cells, v.dim, x0, x1, y0, y1, tid);
                copy_occ_rows<COPY_BLOCK>(v.occ + (size_t)td * v.dim * v.ow, v.occ + (size_t)ts * v.dim * v.ow, v.ow, x0, x1, y0, y1, tid);
                if (tid == 0) {
                    const int ya = y0 & ~15, yb = min((y1 | 15) + 1, v.dim);
                    atomicAdd(&v.stats[ST_COPY_BYTES], 2ull * (unsigned long long)(x1 - x0 + 1) * (yb - ya));
                }
            }
            if (tid == 0) {
                for (int k = 0; k < 4; ++k) v.tile_bbox[4 * td + k] = sb[k];
                atomicAdd(&v.stats[ST_COPIES], 1ull);
            }
        } else {                                              // source has no tile here: release the destination's
            int db[4];
            for (int k = 0; k < 4; ++k) db[k] = v.tile_bbox[4 * td + k];
            if (db[0] <= db[1] && db[2] <= db[3])
            {
                copy_rows<COPY_BLOCK>(v.pool + (size_t)td * cells, nullptr, v.dim, db[0], db[1], db[2], db[3], tid);
                copy_occ_rows<COPY_BLOCK>(v.occ + (size_t)td * v.dim * v.ow, nullptr, v.ow, db[0], db[1], db[2], db[3], tid);
            }
            if (tid == 0) {
                v.tile_bbox[4 * td + 0] = INT_MAX; v.tile_bbox[4 * td + 1] = -1;
                v.tile_bbox[4 * td + 2] = INT_MAX; v.tile_bbox[4 * td + 3] = -1;
                v.tile_tab[(size_t)dst_slot * LL + pos] = -1;
                a.pending_free[atomicAdd(a.n_pending, 1)] = td;   // pushed back after the kernel
            }
        }
      }
    }
}

__global__ void resample_release_kernel(DevView v, const int32_t* pending, int32_t* n_pending) {
    // single thread block: return released tiles to the free stack (they are zero-filled)
    const int n = *n_pending;
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = atomicAdd(v.free_top, n);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) v.free_stack[s_base + i] = pending[i];
    __syncthreads();
    if (threadIdx.x == 0) *n_pending = 0;
}

// ---- multi-GPU pieces -------------------------------------------------------------------------------------------
// scatter the local weights to their global particle ids (the caller all-reduces the vector over RCCL)
__global__ void export_weights_kernel(int P, const double* __restrict__ w, const int32_t* __restrict__ gid,
                                      double* __restrict__ out, int n_global, const uint8_t* __restrict__ bad) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P && gid[p] >= 0 && gid[p] < n_global) out[gid[p]] = w[p];
    // early export: out[n_global] counts (after the all-reduce) the ranks with a particle on the NaN-covariance branch,
    // whose weight still changes after the map update (robot.py:73-78)
    if (p < P && bad && bad[p]) out[n_global] = 1.0;
}

// T[i] = number of new local particles whose source is an old local particle <= i (sources sorted ascending, -1 last)
__device__ void sources_to_T_stage(int P, const int32_t* __restrict__ idx, int32_t* __restrict__ T, int32_t* __restrict__ did) {
    __shared__ int s_buf[PLAN_THREADS];
    const int tid = threadIdx.x, nt = PLAN_THREADS;
    const int chunk = (P + nt - 1) / nt;
    const int i0 = min(tid * chunk, P), i1 = min(i0 + chunk, P);
    for (int i = i0; i < i1; ++i) T[i] = 0;
    __syncthreads();
    for (int j = i0; j < i1; ++j) if (idx[j] >= 0) atomicAdd(&T[idx[j]], 1);
    __syncthreads();
    int sum = 0;
    for (int i = i0; i < i1; ++i) sum += T[i];
    int total;
    int run = block_exclusive_scan_1024(sum, s_buf, tid, total);
    for (int i = i0; i < i1; ++i) { run += T[i]; T[i] = run; }
    if (tid == 0) *did = 1;
}

// tile ids and written boxes of n particles -> out[n][LL*5] (tile id, x0, x1, y0, y1)
__global__ void gather_meta_kernel(DevView v, const int32_t* __restrict__ local_idx, int n, int32_t* __restrict__ out) {
    const int LL = v.L * v.L;
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n * LL) return;
    const int i = q / LL, pos = q % LL;
    const int t = v.tile_tab[(size_t)v.slot[local_idx[i]] * LL + pos];
    int32_t* o = out + (size_t)q * 5;
    o[0] = t;
    for (int k = 0; k < 4; ++k) o[1 + k] = t >= 0 ? v.tile_bbox[4 * t + k] : 0;
}

struct PackJob { int32_t particle, tile, x0, x1, ya, yb; long long off; };   // particle = local index; tile < 0: state block

__global__ __launch_bounds__(BLOCK) void pack_kernel(DevView v, const PackJob* __restrict__ jobs, unsigned char* __restrict__ buf) {
    const PackJob j = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    unsigned char* dst = buf + j.off;
    if (j.tile < 0) {                                   // 13 doubles: pose, covariance, weight
        if (tid < 13) {
            const int p = j.particle;
            double val = tid == 0 ? v.px[p] : tid == 1 ? v.py[p] : tid == 2 ? v.pth[p]
                       : tid < 12 ? v.cov[(size_t)(tid - 3) * v.P + p] : v.weight[p];
            reinterpret_cast<double*>(dst)[tid] = val;
        }
        return;
    }
    const size_t cells = (size_t)v.dim * v.dim;
    const int8_t* src = v.pool + (size_t)j.tile * cells;
    const int per_row = (j.yb - j.ya) / 16, rows = j.x1 - j.x0 + 1;
    for (int q = tid; q < rows * per_row; q += BLOCK) {
        const int x = j.x0 + q / per_row, y = j.ya + (q % per_row) * 16;
        reinterpret_cast<uint4*>(dst)[q] = *reinterpret_cast<const uint4*>(src + (size_t)x * v.dim + y);
    }
    uint32_t* docc = reinterpret_cast<uint32_t*>(dst + (size_t)rows * per_row * 16);
    const uint32_t* socc = v.occ + ((size_t)j.tile * v.dim + j.x0) * v.ow;
    for (int q = tid; q < rows * v.ow; q += BLOCK) docc[q] = socc[q];
}

// one workgroup per (incoming particle, lattice position): install the received tile (or release the slot's old one)
struct UnpackJob { int32_t particle, pos, has, x0, x1, ya, yb, pad; long long off; };

__global__ __launch_bounds__(BLOCK) void unpack_kernel(DevView v, const UnpackJob* __restrict__ jobs, const unsigned char* __restrict__ buf,
                                                       int32_t* pending_free, int32_t* n_pending) {
    __shared__ int s_tile;
    const UnpackJob j = jobs[blockIdx.x];
    const int tid = threadIdx.x, LL = v.L * v.L;
    if (j.pos < 0) {                                    // state block
        if (tid < 13) {
            const double val = reinterpret_cast<const double*>(buf + j.off)[tid];
            const int p = j.particle;
            if (tid == 0) v.px[p] = val; else if (tid == 1) v.py[p] = val; else if (tid == 2) v.pth[p] = val;
            else if (tid < 12) v.cov[(size_t)(tid - 3) * v.P + p] = val; else v.weight[p] = 1.0;   // main.py:77-78
        }
        return;
    }
    const int slot = v.slot[j.particle];
    int td = v.tile_tab[(size_t)slot * LL + j.pos];
    const size_t cells = (size_t)v.dim * v.dim;
    if (!j.has) {
        if (td < 0) return;
        int db[4];
        for (int k = 0; k < 4; ++k) db[k] = v.tile_bbox[4 * td + k];
        if (db[0] <= db[1] && db[2] <= db[3]) {
            copy_rows<BLOCK>(v.pool + (size_t)td * cells, nullptr, v.dim, db[0], db[1], db[2], db[3], tid);
            copy_occ_rows<BLOCK>(v.occ + (size_t)td * v.dim * v.ow, nullptr, v.ow, db[0], db[1], db[2], db[3], tid);
        }
        if (tid == 0) {
            v.tile_bbox[4 * td + 0] = INT_MAX; v.tile_bbox[4 * td + 1] = -1; v.tile_bbox[4 * td + 2] = INT_MAX; v.tile_bbox[4 * td + 3] = -1;
            v.tile_tab[(size_t)slot * LL + j.pos] = -1;
            pending_free[atomicAdd(n_pending, 1)] = td;
        }
        return;
    }
    if (td < 0) {
        if (tid == 0) {
            int idx = atomicSub(v.free_top, 1) - 1;
            if (idx < 0) { atomicAdd(v.free_top, 1); atomicCAS(v.err, 0, RBPF_ENOMEM); s_tile = -1; }
            else { s_tile = v.free_stack[idx]; v.tile_tab[(size_t)slot * LL + j.pos] = s_tile; }
        }
        __syncthreads();
        td = s_tile;
        if (td < 0) return;
    } else {                                            // wipe what the slot's previous owner had written
        int db[4];
        for (int k = 0; k < 4; ++k) db[k] = v.tile_bbox[4 * td + k];
        if (db[0] <= db[1] && db[2] <= db[3]) {
            copy_rows<BLOCK>(v.pool + (size_t)td * cells, nullptr, v.dim, db[0], db[1], db[2], db[3], tid);
            copy_occ_rows<BLOCK>(v.occ + (size_t)td * v.dim * v.ow, nullptr, v.ow, db[0], db[1], db[2], db[3], tid);
        }
        __syncthreads();
    }
    if (j.x0 <= j.x1) {
        const unsigned char* src = buf + j.off;
        int8_t* dst = v.pool + (size_t)td * cells;
        const int per_row = (j.yb - j.ya) / 16, rows = j.x1 - j.x0 + 1;
        for (int q = tid; q < rows * per_row; q += BLOCK) {
            const int x = j.x0 + q / per_row, y = j.ya + (q % per_row) * 16;
            *reinterpret_cast<uint4*>(dst + (size_t)x * v.dim + y) = reinterpret_cast<const uint4*>(src)[q];
        }
        const uint32_t* socc = reinterpret_cast<const uint32_t*>(src + (size_t)rows * per_row * 16);
        uint32_t* docc = v.occ + ((size_t)td * v.dim + j.x0) * v.ow;
        for (int q = tid; q < rows * v.ow; q += BLOCK) docc[q] = socc[q];
    }
    if (tid == 0) { v.tile_bbox[4 * td + 0] = j.x0; v.tile_bbox[4 * td + 1] = j.x1; v.tile_bbox[4 * td + 2] = j.ya; v.tile_bbox[4 * td + 3] = j.yb - 1; }
}

// ---- kernels: the stages one by one (any P), or the small ones fused into one single-workgroup launch ------------
__global__ __launch_bounds__(PLAN_THREADS) void resample_plan_kernel(ResampleArgs a) { resample_plan_stage(a); }
__global__ void resample_expand_kernel(int P, const int32_t* __restrict__ T, int32_t* __restrict__ idx) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < P) resample_expand_one(P, T, idx, j);
}
__global__ __launch_bounds__(PLAN_THREADS) void resample_pair_kernel(PairArgs a) { resample_pair_stage(a); }
__global__ void resample_gather_kernel(GatherArgs a) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < a.P) resample_gather_one(a, j);
}
__global__ __launch_bounds__(PLAN_THREADS) void sources_to_T_kernel(int P, const int32_t* __restrict__ idx, int32_t* __restrict__ T,
                                                                     int32_t* __restrict__ did) { sources_to_T_stage(P, idx, T, did); }

// The planning stages are single-workgroup or trivially parallel; for moderate P one workgroup runs them back to back
// (a stage reads what the previous one wrote to global memory: same CU, __syncthreads() orders it), which saves
// three kernel launches and their dispatch gaps per resample.
enum { RS_PLAN = 1, RS_SRC2T = 2, RS_EXPAND = 4, RS_PAIR = 8, RS_GATHER = 16 };
static const int RS_FUSE_MAX = 4096;        // beyond this the single workgroup costs more than the launches it saves
static const int RS_GATHER_FUSE_MAX = 1024; // ... and the state permutation alone beyond this
struct FusedArgs { int stages; ResampleArgs ra; int32_t* idx; PairArgs pa; GatherArgs ga; };
__global__ __launch_bounds__(PLAN_THREADS) void resample_fused_kernel(FusedArgs f) {
    // T and the ancestor indices also live in LDS from the stage that makes them: the later stages search and chase them
    // (a binary search per new particle, neighbour comparisons), which through global memory is a chain of L2 round trips
    // in a kernel that runs on one CU while the others wait
    extern __shared__ __align__(16) int32_t s_dyn[];                       // 2 * RS_FUSE_MAX words (dynamic: the stages' own arrays fill the static 64 KB)
    int32_t* const s_T = s_dyn; int32_t* const s_idx = s_dyn + RS_FUSE_MAX;
    const int tid = threadIdx.x, P = f.ra.P;
#ifdef RBPF_STAMPS
    long long st[6]; st[0] = clock64();
#define RSTAMP(k) st[k] = clock64()
#else
#define RSTAMP(k) do { } while (0)
#endif
    if (f.stages & RS_PLAN) { resample_plan_stage(f.ra); __syncthreads(); }
    RSTAMP(1);
    if (f.stages & RS_SRC2T) { sources_to_T_stage(P, f.idx, f.ra.T, f.ra.did); __syncthreads(); }
    for (int j = tid; j < P; j += PLAN_THREADS) s_T[j] = f.ra.T[j];
    __syncthreads();
    if (f.stages & RS_EXPAND) { for (int j = tid; j < P; j += PLAN_THREADS) { resample_expand_one(P, s_T, s_idx, j); f.idx[j] = s_idx[j]; } }
    else for (int j = tid; j < P; j += PLAN_THREADS) s_idx[j] = f.idx[j];
    __syncthreads();
    RSTAMP(2);
    if (f.stages & RS_PAIR) { PairArgs pa = f.pa; pa.T = s_T; pa.idx = s_idx; resample_pair_stage(pa); __syncthreads(); }
    RSTAMP(3);
    if (f.stages & RS_GATHER) { GatherArgs ga = f.ga; ga.T = s_T; ga.idx = s_idx; for (int j = tid; j < ga.P; j += PLAN_THREADS) resample_gather_one(ga, j); }
    RSTAMP(4);
#ifdef RBPF_STAMPS
    if (tid == 0) printf("resample_fused: plan %lld expand %lld pair %lld gather %lld cycles\n", st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3]);
#endif
}

static void launch_resample_fused(const FusedArgs& f, hipStream_t s) {
    const size_t lds = 2 * RS_FUSE_MAX * sizeof(int32_t);
    static size_t lds_set[MAX_DEVICES] = {};
    ensure_dynamic_lds(reinterpret_cast<const void*>(resample_fused_kernel), lds, lds_set);
    hipLaunchKernelGGL(resample_fused_kernel, dim3(1), dim3(PLAN_THREADS), lds, s, f);
}

void launch_export_weights(const DevView& v, double* d_out, int n_global, const uint8_t* d_bad, hipStream_t s) {
    (void)hipMemsetAsync(d_out, 0, (size_t)(n_global + (d_bad ? 1 : 0)) * 8, s);
    hipLaunchKernelGGL(export_weights_kernel, dim3((v.P + 255) / 256), dim3(256), 0, s, v.P, v.weight, v.global_id, d_out, n_global, d_bad);
}
void launch_sources_to_T(int P, const int32_t* d_idx, int32_t* d_T, int32_t* d_did, hipStream_t s) {
    hipLaunchKernelGGL(sources_to_T_kernel, dim3(1), dim3(PLAN_THREADS), 0, s, P, d_idx, d_T, d_did);
}
void launch_gather_meta(const DevView& v, const int32_t* d_local, int n, int32_t* d_out, hipStream_t s) {
    int tot = n * v.L * v.L;
    hipLaunchKernelGGL(gather_meta_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, v, d_local, n, d_out);
}
void launch_pack(const DevView& v, const void* d_jobs, int n_jobs, void* d_buf, hipStream_t s) {
    if (n_jobs) hipLaunchKernelGGL(pack_kernel, dim3(n_jobs), dim3(BLOCK), 0, s, v, static_cast<const PackJob*>(d_jobs), static_cast<unsigned char*>(d_buf));
}
void launch_unpack(const DevView& v, const ResampleBuffers& b, const void* d_jobs, int n_jobs, const void* d_buf, hipStream_t s) {
    if (n_jobs) hipLaunchKernelGGL(unpack_kernel, dim3(n_jobs), dim3(BLOCK), 0, s, v, static_cast<const UnpackJob*>(d_jobs),
                                   static_cast<const unsigned char*>(d_buf), b.pending_free, b.n_pending);
    hipLaunchKernelGGL(resample_release_kernel, dim3(1), dim3(256), 0, s, v, b.pending_free, b.n_pending);
}

// ---- host-side launch sequence --------------------------------------------------------------------------
void launch_resample_indices(int P, const double* d_w, double u, double spread, int32_t* d_T, int32_t* d_idx,
                             int32_t* d_did, int32_t* d_err, hipStream_t s) {
    ResampleArgs ra{P, d_w, u, spread, d_T, d_did, d_err};
    if (P <= RS_FUSE_MAX) {
        FusedArgs f{RS_PLAN | RS_EXPAND, ra, d_idx, PairArgs{}, GatherArgs{}};
        launch_resample_fused(f, s);
        return;
    }
    hipLaunchKernelGGL(resample_plan_kernel, dim3(1), dim3(PLAN_THREADS), 0, s, ra);
    hipLaunchKernelGGL(resample_expand_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, d_T, d_idx);
}

// the whole local resample (main.py:46-79): plan, ancestors, slot pairing, state permutation, tile copies
void launch_resample_local(const DevView& v, const ResampleBuffers& b, const double* d_w, double u, double spread, hipStream_t s) {
    if (v.P > RS_FUSE_MAX) {
        launch_resample_indices(v.P, d_w, u, spread, b.T, b.idx, b.did, v.err, s);
        launch_resample_apply(v, b, s);
        return;
    }
    // the state permutation moves 200 bytes per particle: through one CU it took as long as the three planning stages together
    // (25 us of the fused kernel's 54 at 4096 particles), so beyond RS_GATHER_FUSE_MAX particles it is a launch of its own
    const bool fuse_gather = v.P <= RS_GATHER_FUSE_MAX;
    const GatherArgs ga{v.P, b.idx, b.did, v.px, v.py, v.pth, v.cov, v.weight, b.px2, b.py2, b.pth2, b.cov2, b.w2, b.T, v.dup_of};
    FusedArgs f{RS_PLAN | RS_EXPAND | RS_PAIR | (fuse_gather ? RS_GATHER : 0), ResampleArgs{v.P, d_w, u, spread, b.T, b.did, v.err}, b.idx,
                PairArgs{v.P, b.T, b.idx, b.did, v.slot, b.slot2, b.dead_list, b.jobs, b.n_jobs, v.err}, ga};
    launch_resample_fused(f, s);
    CopyArgs ca{v, b.jobs, b.n_jobs, b.pending_free, b.n_pending, fuse_gather ? 0 : 1, ga};
    hipLaunchKernelGGL(resample_copy_kernel, dim3(512), dim3(COPY_BLOCK), 0, s, ca);
    hipLaunchKernelGGL(resample_release_kernel, dim3(1), dim3(256), 0, s, v, b.pending_free, b.n_pending);
}

// the local part of a global resample: new_src (sorted sources, -1 = arrival) in b.idx -> T, slots, state, tile copies
void launch_resample_apply_sources(const DevView& v, const ResampleBuffers& b, hipStream_t s) {
    if (v.P > RS_FUSE_MAX) {
        launch_sources_to_T(v.P, b.idx, b.T, b.did, s);
        launch_resample_apply(v, b, s);
        return;
    }
    const bool fuse_gather = v.P <= RS_GATHER_FUSE_MAX;
    const GatherArgs ga{v.P, b.idx, b.did, v.px, v.py, v.pth, v.cov, v.weight, b.px2, b.py2, b.pth2, b.cov2, b.w2, b.T, v.dup_of};
    FusedArgs f{RS_SRC2T | RS_PAIR | (fuse_gather ? RS_GATHER : 0), ResampleArgs{v.P, nullptr, 0.0, 0.0, b.T, b.did, v.err}, b.idx,
                PairArgs{v.P, b.T, b.idx, b.did, v.slot, b.slot2, b.dead_list, b.jobs, b.n_jobs, v.err}, ga};
    launch_resample_fused(f, s);
    CopyArgs ca{v, b.jobs, b.n_jobs, b.pending_free, b.n_pending, fuse_gather ? 0 : 1, ga};
    hipLaunchKernelGGL(resample_copy_kernel, dim3(512), dim3(COPY_BLOCK), 0, s, ca);
    hipLaunchKernelGGL(resample_release_kernel, dim3(1), dim3(256), 0, s, v, b.pending_free, b.n_pending);
}

void launch_resample_apply(const DevView& v, const ResampleBuffers& b, hipStream_t s) {
    PairArgs pa{v.P, b.T, b.idx, b.did, v.slot, b.slot2, b.dead_list, b.jobs, b.n_jobs, v.err};
    hipLaunchKernelGGL(resample_pair_kernel, dim3(1), dim3(PLAN_THREADS), 0, s, pa);
    GatherArgs ga{v.P, b.idx, b.did, v.px, v.py, v.pth, v.cov, v.weight,
                  b.px2, b.py2, b.pth2, b.cov2, b.w2, b.T, v.dup_of};
    CopyArgs ca{v, b.jobs, b.n_jobs, b.pending_free, b.n_pending, 1, ga};          // (the state permutation rides with the tile copies)
    hipLaunchKernelGGL(resample_copy_kernel, dim3(512), dim3(COPY_BLOCK), 0, s, ca);
    hipLaunchKernelGGL(resample_release_kernel, dim3(1), dim3(256), 0, s, v, b.pending_free, b.n_pending);
}

}  // namespace rbpf
