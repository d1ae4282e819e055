/*
 * rbpf_oracle.c -- CPU oracle, C restatement of the reference's per-particle hot path.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ as the checker and by bench.py's cpu_baseline leg as the
 * "reference-equivalent CPU path"; never linked or loaded by thesis_amd/.
 *
 * Loop-faithful restatement (reference = amansanghvi/Thesis, cited file:line) of
 *   HybridMap.update            hybridmap.py:95-145   (+ get_affected_points :274-301)
 *   HybridMap.get_odds_at       hybridmap.py:85-93    (+ GridMap.get_cell gridmap.py:120-128)
 *   Robot._generate_sample_weight  robot.py:118-139
 *   Robot.map_update (moments)  robot.py:89-115
 * float64 cells indexed [x][y], the reference's two float index formulas, x87 long double where the
 * reference holds np.longdouble (robot.py:25,92-94,119,124).  A pose is np.longdouble after the first
 * map_update and Python float before; `ld` selects the arithmetic of the pose-dependent expressions.
 *
 * Pinned by tests/test_oracle_c.py against tests/golden/G3, G5, G6 (captured from the reference).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; x86-64 only because of the 80-bit long double).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_TILES 64

typedef struct {
    long cx, cy;          /* tile centre (integer metres on the 40 m lattice) */
    double* map;          /* dim*dim, map[x*dim + y] */
} tile_t;

typedef struct {
    double cs;            /* cell size */
    long len;             /* tile length in metres */
    int dim;
    int n_tiles;
    tile_t tiles[MAX_TILES];
    double occ, nearby, emp, max_occ, min_emp;
    unsigned long long cells_visited;
} omap_t;

omap_t* orc_map_new(double cs, long len_m) {
    omap_t* m = (omap_t*)calloc(1, sizeof(omap_t));
    m->cs = cs; m->len = len_m;
    m->dim = (int)lround((double)len_m / cs);                     /* gridmap.py:31 */
    m->occ = 0.80; m->nearby = 0.20; m->emp = -0.30; m->max_occ = 3.0; m->min_emp = -3.0;   /* gridmap.py:20-24 */
    m->n_tiles = 1;                                               /* hybridmap.py:70 */
    m->tiles[0].cx = 0; m->tiles[0].cy = 0;
    m->tiles[0].map = (double*)calloc((size_t)m->dim * m->dim, sizeof(double));
    return m;
}

void orc_map_free(omap_t* m) {
    if (!m) return;
    for (int i = 0; i < m->n_tiles; ++i) free(m->tiles[i].map);
    free(m);
}

omap_t* orc_map_copy(const omap_t* s) {                           /* hybridmap.py:315-320 */
    omap_t* m = (omap_t*)malloc(sizeof(omap_t));
    memcpy(m, s, sizeof(omap_t));
    size_t n = (size_t)s->dim * s->dim;
    for (int i = 0; i < s->n_tiles; ++i) {
        m->tiles[i].map = (double*)malloc(n * sizeof(double));
        memcpy(m->tiles[i].map, s->tiles[i].map, n * sizeof(double));
    }
    return m;
}

int orc_map_n_tiles(const omap_t* m) { return m->n_tiles; }
int orc_map_dim(const omap_t* m) { return m->dim; }
unsigned long long orc_map_cells_visited(const omap_t* m) { return m->cells_visited; }
double* orc_map_tile(omap_t* m, int k, double* centre2) {
    centre2[0] = (double)m->tiles[k].cx; centre2[1] = (double)m->tiles[k].cy;
    return m->tiles[k].map;
}
void orc_map_set_tile(omap_t* m, long cx, long cy, const double* cells) {
    int k;
    for (k = 0; k < m->n_tiles; ++k) if (m->tiles[k].cx == cx && m->tiles[k].cy == cy) break;
    if (k == m->n_tiles) {
        m->tiles[k].cx = cx; m->tiles[k].cy = cy;
        m->tiles[k].map = (double*)calloc((size_t)m->dim * m->dim, sizeof(double));
        m->n_tiles++;
    }
    memcpy(m->tiles[k].map, cells, (size_t)m->dim * m->dim * sizeof(double));
}

/* hybridmap.py:44-45 on a double position */
static int in_map_d(const omap_t* m, const tile_t* t, double x, double y) {
    double r = (double)m->len / 2;
    return x >= t->cx - r && x < t->cx + r && y >= t->cy - r && y < t->cy + r;
}
static int in_map_l(const omap_t* m, const tile_t* t, long double x, long double y) {
    double r = (double)m->len / 2;
    return x >= t->cx - r && x < t->cx + r && y >= t->cy - r && y < t->cy + r;
}

/* hybridmap.py:193-208, one axis */
static long map_centre_1d(const omap_t* m, double v) {
    long approx = lrint(v / (double)m->len);                      /* int(round()) : half to even */
    for (long a = approx - 1; a < approx + 2; ++a) {
        long mc = a * m->len;
        if (v < mc + (double)m->len / 2 && v >= mc - (double)m->len / 2) return mc;
    }
    return 0;
}

static tile_t* tile_with_pos_d(omap_t* m, double x, double y) {  /* hybridmap.py:263-272 */
    for (int i = 0; i < m->n_tiles; ++i) if (in_map_d(m, &m->tiles[i], x, y)) return &m->tiles[i];
    return NULL;
}

/* gridmap.py:93-94 "set" formula */
static int set_index(const omap_t* m, double rel) { return (int)(rel / m->cs + m->dim / 2.0); }

static void clamp_occ(const omap_t* m, double* c) { double v = *c + m->occ; *c = v < m->max_occ ? v : m->max_occ; }
static void clamp_near(const omap_t* m, double* c) { double v = *c + m->nearby; *c = v < m->max_occ ? v : m->max_occ; }
static void clamp_emp(const omap_t* m, double* c) { double v = *c + m->emp; *c = v > m->min_emp ? v : m->min_emp; }

/* hybridmap.py:274-301 into a caller buffer; returns the number of points */
static int affected_points(long x0, long y0, long x1, long y1, long* px, long* py) {
    long dx = labs(x1 - x0), dy = labs(y1 - y0);
    int n = 0;
    if (dx == 0) { for (long y = y0; y < y1 + 1; ++y) { px[n] = x0; py[n] = y; ++n; } return n; }
    if (dy == 0) { for (long x = x0; x < x1 + 1; ++x) { px[n] = x; py[n] = y0; ++n; } return n; }
    long xs = x1 - x0 > 0 ? 1 : -1, ys = y1 - y0 > 0 ? 1 : -1;
    int steep = dy > dx;
    if (steep) { long t = dx; dx = dy; dy = t; }
    long D = 2 * dy - dx, y = 0;
    for (long x = 0; x < dx + 1; ++x) {
        if (steep) { px[n] = x0 + xs * y; py[n] = y0 + ys * x; } else { px[n] = x0 + xs * x; py[n] = y0 + ys * y; }
        ++n;
        if (D >= 0) { y += 1; D -= 2 * dx; }
        D += 2 * dy;
    }
    return n;
}

/* HybridMap.update, hybridmap.py:95-145.  sx, sy: sensor-frame endpoints (lidar.py:78-79). */
void orc_map_update(omap_t* m, const long double* pose, int ld, const double* sx, const double* sy, int nb) {
    const double cs = m->cs;
    const double c = cos((double)pose[2]), s = sin((double)pose[2]);   /* math.cos takes a float, lidar.py:115 */
    long sx0, sy0;
    if (ld) {
        int ok = 0;                                               /* hybridmap.py:98-100 */
        for (int i = 0; i < m->n_tiles; ++i) ok |= in_map_l(m, &m->tiles[i], pose[0], pose[1]);
        if (!ok) return;
        sx0 = (long)(pose[0] / cs); sy0 = (long)(pose[1] / cs);   /* hybridmap.py:102 in longdouble */
    } else {
        if (!tile_with_pos_d(m, (double)pose[0], (double)pose[1])) return;
        sx0 = (long)((double)pose[0] / cs); sy0 = (long)((double)pose[1] / cs);
    }
    long* px = (long*)malloc(sizeof(long) * 8192);
    long* py = (long*)malloc(sizeof(long) * 8192);
    for (int i = 0; i < nb; ++i) {
        int end_is_occ = 1;
        double dist = sqrt(sx[i] * sx[i] + sy[i] * sy[i]);        /* hybridmap.py:105 */
        long ex, ey;
        if (ld) {                                                 /* lidar.py:123 matmul in longdouble */
            long double gx = ((long double)c * sx[i] + (long double)(-s) * sy[i]) + pose[0] * 1;
            long double gy = ((long double)s * sx[i] + (long double)c * sy[i]) + pose[1] * 1;
            ex = (long)(gx / cs); ey = (long)(gy / cs);           /* hybridmap.py:106 */
        } else {
            double gx = (c * sx[i] + (-s) * sy[i]) + (double)pose[0];
            double gy = (s * sx[i] + c * sy[i]) + (double)pose[1];
            ex = (long)(gx / cs); ey = (long)(gy / cs);
        }
        if (dist > 15) {                                          /* hybridmap.py:107-113 */
            double scale = 15.0 / dist;
            ex = (long)(sx0 + scale * (ex - sx0));
            ey = (long)(sy0 + scale * (ey - sy0));
            end_is_occ = 0;
        }
        int n = affected_points(sx0, sy0, ex, ey, px, py);
        m->cells_visited += (unsigned long long)n;
        for (int j = 0; j < n; ++j) {
            double posx = px[j] * cs, posy = py[j] * cs;          /* hybridmap.py:123 */
            tile_t* t = tile_with_pos_d(m, posx, posy);
            if (!t) {                                             /* hybridmap.py:125-133 */
                long ncx = map_centre_1d(m, posx), ncy = map_centre_1d(m, posy);
                t = tile_with_pos_d(m, (double)ncx, (double)ncy);
                if (!t) {
                    t = &m->tiles[m->n_tiles++];
                    t->cx = ncx; t->cy = ncy;
                    t->map = (double*)calloc((size_t)m->dim * m->dim, sizeof(double));
                }
            }
            double rx = posx - t->cx, ry = posy - t->cy;          /* hybridmap.py:136 */
            double* cell = &t->map[(size_t)set_index(m, rx) * m->dim + set_index(m, ry)];
            if (end_is_occ && px[j] == ex && py[j] == ey) {       /* hybridmap.py:137 */
                clamp_occ(m, cell);
                if (j > 0) {                                      /* hybridmap.py:139-142 */
                    double nx = px[j - 1] * cs, ny = py[j - 1] * cs;
                    if (in_map_d(m, t, nx, ny))
                        clamp_near(m, &t->map[(size_t)set_index(m, nx - t->cx) * m->dim + set_index(m, ny - t->cy)]);
                }
            } else {
                clamp_emp(m, cell);                               /* hybridmap.py:144 */
            }
        }
    }
    free(px); free(py);
}

/* HybridMap.get_odds_at (hybridmap.py:85-93) on a longdouble position; returns 0 for None */
static int odds_at_l(const omap_t* m, long double x, long double y, double* out) {
    for (int i = 0; i < m->n_tiles; ++i) {
        const tile_t* t = &m->tiles[i];
        if (!in_map_l(m, t, x, y)) continue;
        long double rx = x - t->cx, ry = y - t->cy;
        double half = (double)m->len / 2;
        if (ry < -half || ry >= half) return 0;                   /* gridmap.py:121-124 */
        if (rx < -half || rx >= half) return 0;
        int ix = (int)(rx / m->len * m->dim + m->dim / 2);        /* gridmap.py:126 (dim/2 exact: dim even) */
        int iy = (int)(ry / m->len * m->dim + m->dim / 2);
        *out = t->map[(size_t)ix * m->dim + iy];
        return 1;
    }
    return 0;
}
static int odds_at_d(const omap_t* m, double x, double y, double* out) {
    for (int i = 0; i < m->n_tiles; ++i) {
        const tile_t* t = &m->tiles[i];
        if (!in_map_d(m, t, x, y)) continue;
        double rx = x - t->cx, ry = y - t->cy;
        double half = (double)m->len / 2;
        if (ry < -half || ry >= half) return 0;
        if (rx < -half || rx >= half) return 0;
        int ix = (int)(rx / m->len * m->dim + m->dim / 2.0);
        int iy = (int)(ry / m->len * m->dim + m->dim / 2.0);
        *out = t->map[(size_t)ix * m->dim + iy];
        return 1;
    }
    return 0;
}

int orc_get_odds_at(const omap_t* m, double x, double y, double* out) { return odds_at_d(m, x, y, out); }

/* Robot._generate_sample_weight, robot.py:118-139.  guesses[K][3], prs[K] -> w[K] (long double) */
void orc_sample_weight(const omap_t* m, const long double* guesses, int ld, int K, const double* sx, const double* sy,
                       int nb, const double* prs, long double* w) {
    for (int k = 0; k < K; ++k) {
        const long double* g = guesses + 3 * k;
        const double c = cos((double)g[2]), s = sin((double)g[2]);
        long double obs = 1.0L;                                   /* robot.py:124 */
        for (int j = 0; j < nb; ++j) {
            double dist = sqrt(sx[j] * sx[j] + sy[j] * sy[j]);    /* robot.py:129 */
            if (dist < 25 && dist > 0.01) {
                double o;
                int ok;
                if (ld) {
                    long double gx = ((long double)c * sx[j] + (long double)(-s) * sy[j]) + g[0] * 1;
                    long double gy = ((long double)s * sx[j] + (long double)c * sy[j]) + g[1] * 1;
                    ok = odds_at_l(m, gx, gy, &o);
                } else {
                    double gx = (c * sx[j] + (-s) * sy[j]) + (double)g[0];
                    double gy = (s * sx[j] + c * sy[j]) + (double)g[1];
                    ok = odds_at_d(m, gx, gy, &o);
                }
                if (ok) obs += o;                                 /* robot.py:135 */
            }
        }
        w[k] = obs * prs[k];                                      /* robot.py:138 */
    }
}

/* Robot.map_update after the matcher, robot.py:80-115, with explicit samples and motion_prs.
 * state: pose[3], cov[9], weight (all long double, in/out). */
void orc_robot_map_update(omap_t* m, long double* pose, long double* cov, long double* weight, const double* guesses,
                          const double* prs, int K, const double* sx, const double* sy, int nb) {
    long double* g = (long double*)malloc(sizeof(long double) * 3 * K);
    long double* w = (long double*)malloc(sizeof(long double) * K);
    for (int i = 0; i < 3 * K; ++i) g[i] = guesses[i];            /* np.random.multivariate_normal returns float64 */
    orc_sample_weight(m, g, 0, K, sx, sy, nb, prs, w);
    long double min_w = w[0];
    for (int k = 1; k < K; ++k) if (w[k] < min_w) min_w = w[k];   /* robot.py:89 */
    for (int k = 0; k < K; ++k) w[k] = w[k] - min_w + 1e-2;       /* robot.py:90 */
    long double mean[3] = {0, 0, 0}, norm = 0;
    for (int k = 0; k < K; ++k) {                                 /* robot.py:95-97 */
        for (int i = 0; i < 3; ++i) mean[i] = mean[i] + g[3 * k + i] * w[k];
        norm = norm + w[k];
    }
    for (int i = 0; i < 3; ++i) mean[i] = mean[i] / norm;         /* robot.py:103 */
    long double sig[9] = {0};
    for (int k = 0; k < K; ++k) {                                 /* robot.py:104-106 */
        long double d[3];
        for (int i = 0; i < 3; ++i) d[i] = g[3 * k + i] + (-mean[i]);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) sig[3 * i + j] = sig[3 * i + j] + d[i] * d[j] * w[k];
    }
    for (int i = 0; i < 9; ++i) cov[i] = sig[i] / norm;           /* robot.py:107,110 */
    norm = norm + min_w * K;                                      /* robot.py:108 */
    for (int i = 0; i < 3; ++i) pose[i] = mean[i];                /* robot.py:111-113 */
    *weight = norm + *weight;                                     /* robot.py:114 */
    orc_map_update(m, pose, 1, sx, sy, nb);                       /* robot.py:115 */
    free(g); free(w);
}

/* ctypes helpers: long double values cross the boundary as (hi, lo) double pairs */
void orc_ld_to_pair(const long double* v, int n, double* hi, double* lo) {
    for (int i = 0; i < n; ++i) { hi[i] = (double)v[i]; lo[i] = (double)(v[i] - (long double)hi[i]); }
}
void orc_pair_to_ld(const double* hi, const double* lo, int n, long double* v) {
    for (int i = 0; i < n; ++i) v[i] = (long double)hi[i] + (long double)lo[i];
}
int orc_sizeof_long_double(void) { return (int)sizeof(long double); }

/* ---- timing entry for bench.py's cpu_baseline: n particle-updates of the reference-equivalent path ---------
 * Each particle-update = 30-sample weighting + moments + ray-cast map update (robot.py:80-115), i.e. the
 * reference's Robot.map_update with the scan matcher excluded. */
double orc_bench_particle_updates(int n_particles, int n_steps, const double* sx_all, const double* sy_all, int nb,
                                  const double* guesses_all, const double* prs_all, int K, double cs) {
    double acc = 0;
    for (int p = 0; p < n_particles; ++p) {
        omap_t* m = orc_map_new(cs, 40);
        long double pose[3] = {0, 0, 0}, cov[9] = {0}, weight = 1.0L;
        for (int s = 0; s < n_steps; ++s) {
            const double* sx = sx_all + (size_t)s * nb; const double* sy = sy_all + (size_t)s * nb;
            orc_robot_map_update(m, pose, cov, &weight, guesses_all + ((size_t)s * n_particles + p) * K * 3,
                                 prs_all + ((size_t)s * n_particles + p) * K, K, sx, sy, nb);
        }
        acc += (double)weight + (double)pose[0];
        orc_map_free(m);
    }
    return acc;
}
