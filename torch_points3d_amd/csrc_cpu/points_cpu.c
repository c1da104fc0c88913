/* Host-side radius / k-NN searches on a uniform grid (C-ABI: include/tp3d_cpu.h).
 *
 * Specification: the `torch_points_kernels.points_cpu` calls of the reference's data transforms and registration
 * dataset builders (file:line list in the header).  The reference binds torch-points-kernels 0.7.0 (nanoflann KD-tree);
 * this is an independent implementation with the same result sets: strict d2 < r^2 membership, squared distances,
 * `sorted` = closest first.  Unsorted results come in ascending support index (nanoflann's traversal order is not a
 * contract).  No OpenMP: worker threads are plain pthreads created and joined per call, so a process that forks
 * (DataLoader workers) never inherits a dormant thread pool. */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "tp3d_cpu.h"

typedef struct {
    const float *pts;
    int64_t n;
    float lo[3], cell, inv;
    int dim[3];
    int64_t *start; /* ncell + 1 */
    int64_t *order; /* n point ids, cell by cell, ascending id inside a cell */
} Grid;

int tp3d_cpu_abi_version(void) { return 2; }

static inline float sqdist3(const float *a, const float *b)
{
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

static inline int cell_of(const Grid *g, float v, int axis)
{
    int c = (int)floorf((v - g->lo[axis]) * g->inv);
    if (c < 0) c = 0;
    if (c >= g->dim[axis]) c = g->dim[axis] - 1;
    return c;
}

void *tp3d_cpu_grid_build(const float *points, int64_t n, float cell)
{
    if (n < 0 || (n > 0 && !points) || !(cell > 0.0f)) return NULL;
    Grid *g = (Grid *)calloc(1, sizeof(Grid));
    if (!g) return NULL;
    g->pts = points;
    g->n = n;
    float hi[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) g->lo[a] = 0.0f;
    if (n > 0) {
        for (int a = 0; a < 3; ++a) g->lo[a] = hi[a] = points[a];
        for (int64_t i = 1; i < n; ++i)
            for (int a = 0; a < 3; ++a) {
                const float v = points[i * 3 + a];
                if (v < g->lo[a]) g->lo[a] = v;
                if (v > hi[a]) hi[a] = v;
            }
    }
    /* a NaN / Inf coordinate has no cell (and would keep the coarsening loop below from ever ending): refuse */
    for (int64_t i = 0; i < 3 * n; ++i)
        if (!isfinite(points[i])) {
            free(g);
            return NULL;
        }
    for (int a = 0; a < 3; ++a)
        if (!isfinite(hi[a] - g->lo[a])) { /* finite ends whose difference overflows */
            free(g);
            return NULL;
        }
    /* at most ~4 cells per point and 512 per axis: coarsen the cells of a sparse / huge box */
    float c = cell;
    for (int round = 0;; ++round) {
        if (round > 512 || !isfinite(c)) { /* 1.5^512 overflows float long before: cannot happen with finite bounds */
            free(g);
            return NULL;
        }
        double total = 1.0;
        int ok = 1;
        for (int a = 0; a < 3; ++a) {
            const double d = floor((double)(hi[a] - g->lo[a]) / c) + 1.0;
            if (d > 512.0) ok = 0;
            total *= d;
        }
        if (ok && total <= 4.0 * (double)(n > 64 ? n : 64)) break;
        c *= 1.5f;
    }
    g->cell = c;
    g->inv = 1.0f / c;
    for (int a = 0; a < 3; ++a) g->dim[a] = (int)floor((double)(hi[a] - g->lo[a]) / c) + 1;
    const int64_t ncell = (int64_t)g->dim[0] * g->dim[1] * g->dim[2];
    g->start = (int64_t *)calloc((size_t)ncell + 1, sizeof(int64_t));
    g->order = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    int64_t *cid = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    if (!g->start || !g->order || !cid) {
        free(cid);
        tp3d_cpu_grid_free(g);
        return NULL;
    }
    for (int64_t i = 0; i < n; ++i) {
        const float *p = points + i * 3;
        cid[i] = ((int64_t)cell_of(g, p[2], 2) * g->dim[1] + cell_of(g, p[1], 1)) * g->dim[0] + cell_of(g, p[0], 0);
        g->start[cid[i] + 1]++;
    }
    for (int64_t k = 0; k < ncell; ++k) g->start[k + 1] += g->start[k];
    int64_t *cur = (int64_t *)malloc((size_t)ncell * sizeof(int64_t));
    if (!cur) {
        free(cid);
        tp3d_cpu_grid_free(g);
        return NULL;
    }
    memcpy(cur, g->start, (size_t)ncell * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) g->order[cur[cid[i]]++] = i; /* ascending id inside every cell */
    free(cur);
    free(cid);
    return g;
}

void tp3d_cpu_grid_free(void *grid)
{
    Grid *g = (Grid *)grid;
    if (!g) return;
    free(g->start);
    free(g->order);
    free(g);
}

/* ---------------------------------------------------------------------------------------------- threading */
typedef void (*range_fn)(void *ctx, int64_t lo, int64_t hi);
typedef struct {
    range_fn fn;
    void *ctx;
    int64_t lo, hi;
} Job;
static void *job_main(void *p)
{
    Job *j = (Job *)p;
    j->fn(j->ctx, j->lo, j->hi);
    return NULL;
}
static void parallel_ranges(range_fn fn, void *ctx, int64_t n, int threads)
{
    if (threads > 64) threads = 64;
    if (threads <= 1 || n < 2 * (int64_t)threads) {
        fn(ctx, 0, n);
        return;
    }
    pthread_t th[64];
    Job jobs[64];
    int running[64];
    for (int t = 0; t < threads; ++t) {
        jobs[t].fn = fn;
        jobs[t].ctx = ctx;
        jobs[t].lo = n * t / threads;
        jobs[t].hi = n * (t + 1) / threads;
        /* the last range -- and any range whose thread cannot be created -- runs on the calling thread */
        running[t] = t < threads - 1 && pthread_create(&th[t], NULL, job_main, &jobs[t]) == 0;
        if (!running[t]) fn(ctx, jobs[t].lo, jobs[t].hi);
    }
    for (int t = 0; t < threads; ++t)
        if (running[t]) pthread_join(th[t], NULL);
}

/* ---------------------------------------------------------------------------------------------- radius search */
typedef struct {
    int64_t id;
    float d2;
} Hit;

static int hit_by_dist(const void *a, const void *b)
{
    const Hit *x = (const Hit *)a, *y = (const Hit *)b;
    if (x->d2 < y->d2) return -1;
    if (x->d2 > y->d2) return 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}
static int hit_by_id(const void *a, const void *b)
{
    const Hit *x = (const Hit *)a, *y = (const Hit *)b;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

/* visits every support point with d2 < r2 around q; returns the count, optionally collecting (id, d2) */
static int64_t ball_visit(const Grid *g, const float *q, float radius, float r2, Hit **buf, int64_t *cap)
{
    int c0[3], c1[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = q[a] - radius, hi = q[a] + radius;
        if (hi < g->lo[a] || lo > g->lo[a] + g->cell * (float)g->dim[a]) return 0; /* ball misses the box */
        c0[a] = cell_of(g, lo, a);
        c1[a] = cell_of(g, hi, a);
    }
    int64_t cnt = 0;
    for (int z = c0[2]; z <= c1[2]; ++z)
        for (int y = c0[1]; y <= c1[1]; ++y) {
            const int64_t row = ((int64_t)z * g->dim[1] + y) * g->dim[0];
            const int64_t s = g->start[row + c0[0]], e = g->start[row + c1[0] + 1]; /* x-run of cells is contiguous */
            for (int64_t j = s; j < e; ++j) {
                const int64_t id = g->order[j];
                const float d2 = sqdist3(g->pts + id * 3, q);
                if (d2 < r2) {
                    if (buf) {
                        if (cnt == *cap) {
                            const int64_t ncap = *cap ? *cap * 2 : 64;
                            Hit *nb = (Hit *)realloc(*buf, (size_t)ncap * sizeof(Hit));
                            if (!nb) return -1;
                            *buf = nb;
                            *cap = ncap;
                        }
                        (*buf)[cnt].id = id;
                        (*buf)[cnt].d2 = d2;
                    }
                    ++cnt;
                }
            }
        }
    return cnt;
}

typedef struct {
    const Grid *g;
    const float *query;
    float radius;
    int64_t *counts;
    int limit, sorted;
    const int64_t *offsets;
    int64_t *idx;
    float *dist2;
    int failed;
} BallCtx;

static void count_range(void *p, int64_t lo, int64_t hi)
{
    BallCtx *c = (BallCtx *)p;
    const float r2 = c->radius * c->radius;
    for (int64_t q = lo; q < hi; ++q) c->counts[q] = ball_visit(c->g, c->query + q * 3, c->radius, r2, NULL, NULL);
}

static void fill_range(void *p, int64_t lo, int64_t hi)
{
    BallCtx *c = (BallCtx *)p;
    const float r2 = c->radius * c->radius;
    Hit *buf = NULL;
    int64_t cap = 0;
    for (int64_t q = lo; q < hi; ++q) {
        const int64_t n = ball_visit(c->g, c->query + q * 3, c->radius, r2, &buf, &cap);
        if (n < 0) {
            c->failed = 1;
            break;
        }
        if (n > 1) qsort(buf, (size_t)n, sizeof(Hit), c->sorted ? hit_by_dist : hit_by_id);
        const int64_t room = c->offsets[q + 1] - c->offsets[q];
        int64_t keep = n;
        if (c->limit > 0 && keep > c->limit) keep = c->limit;
        if (keep > room) keep = room;
        int64_t *io = c->idx + c->offsets[q];
        float *dd = c->dist2 + c->offsets[q];
        for (int64_t j = 0; j < keep; ++j) {
            io[j] = buf[j].id;
            dd[j] = buf[j].d2;
        }
        for (int64_t j = keep; j < room; ++j) {
            io[j] = -1;
            dd[j] = -1.0f;
        }
    }
    free(buf);
}

int tp3d_cpu_ball_count(const void *grid, const float *query, int64_t nq, float radius, int64_t *counts, int threads)
{
    if (!grid || nq < 0 || !(radius >= 0.0f) || (nq > 0 && (!query || !counts))) return TP3D_CPU_E_BADARG;
    BallCtx c = {(const Grid *)grid, query, radius, counts, 0, 0, NULL, NULL, NULL, 0};
    parallel_ranges(count_range, &c, nq, threads);
    return TP3D_CPU_OK;
}

int tp3d_cpu_ball_fill(const void *grid, const float *query, int64_t nq, float radius, int limit, int sorted,
                       const int64_t *offsets, int64_t *idx, float *dist2, int threads)
{
    if (!grid || nq < 0 || !(radius >= 0.0f) || !offsets) return TP3D_CPU_E_BADARG;
    if (nq > 0 && (!query || (offsets[nq] > 0 && (!idx || !dist2)))) return TP3D_CPU_E_BADARG;
    BallCtx c = {(const Grid *)grid, query, radius, NULL, limit, sorted, offsets, idx, dist2, 0};
    parallel_ranges(fill_range, &c, nq, threads);
    return c.failed ? TP3D_CPU_E_NOMEM : TP3D_CPU_OK;
}

/* ---------------------------------------------------------------------------------------------- k nearest */
typedef struct {
    const Grid *g;
    const float *query;
    int k;
    int64_t *idx;
    float *dist2;
} KnnCtx;

static inline int better(float d2, int64_t id, float bd2, int64_t bid) { return d2 < bd2 || (d2 == bd2 && id < bid); }

static void knn_range(void *p, int64_t lo, int64_t hi)
{
    KnnCtx *c = (KnnCtx *)p;
    const Grid *g = c->g;
    const int k = c->k;
    for (int64_t q = lo; q < hi; ++q) {
        const float *qp = c->query + q * 3;
        int64_t *bi = c->idx + q * k;
        float *bd = c->dist2 + q * k;
        int have = 0;
        int cc[3];
        for (int a = 0; a < 3; ++a) cc[a] = cell_of(g, qp[a], a);
        const int maxr = (g->dim[0] > g->dim[1] ? (g->dim[0] > g->dim[2] ? g->dim[0] : g->dim[2])
                                                : (g->dim[1] > g->dim[2] ? g->dim[1] : g->dim[2]));
        for (int R = 0; R <= maxr; ++R) {
            /* cells on the shell of Chebyshev radius R around the query's cell */
            for (int z = cc[2] - R; z <= cc[2] + R; ++z) {
                if (z < 0 || z >= g->dim[2]) continue;
                for (int y = cc[1] - R; y <= cc[1] + R; ++y) {
                    if (y < 0 || y >= g->dim[1]) continue;
                    const int face = (z == cc[2] - R || z == cc[2] + R || y == cc[1] - R || y == cc[1] + R);
                    const int step = face ? 1 : (2 * R > 0 ? 2 * R : 1);
                    for (int x = cc[0] - R; x <= cc[0] + R; x += step) {
                        if (x < 0 || x >= g->dim[0]) continue;
                        const int64_t cell = ((int64_t)z * g->dim[1] + y) * g->dim[0] + x;
                        for (int64_t j = g->start[cell]; j < g->start[cell + 1]; ++j) {
                            const int64_t id = g->order[j];
                            const float d2 = sqdist3(g->pts + id * 3, qp);
                            if (have < k || better(d2, id, bd[have - 1], bi[have - 1])) {
                                int pos = have < k ? have : k - 1;
                                while (pos > 0 && better(d2, id, bd[pos - 1], bi[pos - 1])) {
                                    bd[pos] = bd[pos - 1];
                                    bi[pos] = bi[pos - 1];
                                    --pos;
                                }
                                bd[pos] = d2;
                                bi[pos] = id;
                                if (have < k) ++have;
                            }
                        }
                    }
                }
            }
            if (have == k || have == g->n) {
                /* every point outside the visited block is at least `gap` away: the distance from the query to the
                 * nearest face of the block that still has cells behind it */
                float gap = INFINITY;
                for (int a = 0; a < 3; ++a) {
                    if (cc[a] - R > 0) gap = fminf(gap, qp[a] - (g->lo[a] + g->cell * (float)(cc[a] - R)));
                    if (cc[a] + R < g->dim[a] - 1) gap = fminf(gap, (g->lo[a] + g->cell * (float)(cc[a] + R + 1)) - qp[a]);
                }
                if (have == g->n || gap == INFINITY) break;
                if (gap > 0.0f && bd[have - 1] < gap * gap * 0.999f) break;
            }
        }
        for (int j = have; j < k; ++j) {
            bi[j] = -1;
            bd[j] = -1.0f;
        }
    }
}

int tp3d_cpu_knn(const void *grid, const float *query, int64_t nq, int k, int64_t *idx, float *dist2, int threads)
{
    if (!grid || nq < 0 || k <= 0 || (nq > 0 && (!query || !idx || !dist2))) return TP3D_CPU_E_BADARG;
    KnnCtx c = {(const Grid *)grid, query, k, idx, dist2};
    parallel_ranges(knn_range, &c, nq, threads);
    return TP3D_CPU_OK;
}

/* Region growing over a fixed-width neighbour table (torch_points3d/models/panoptic/pointgroup.py:101-115 clusters the
 * points of one semantic label this way): clusters = the sets reached from the lowest unvisited point through the
 * table's directed edges among unvisited points; a row ends at its first -1.  Members are listed in discovery order
 * (the seed first); clusters of fewer than min_size points are dropped.
 *   neighbours (n, width) int64;  members: n int64 slots;  cluster_start: n + 1 int64 slots
 * Returns the number of clusters kept (>= 0) or a negative error. */
int64_t tp3d_cpu_grow_clusters(const int64_t *neighbours, int64_t n, int width, int64_t min_size, int64_t *members,
                               int64_t *cluster_start)
{
    if (n < 0 || width < 0 || (n > 0 && (!members || !cluster_start || (width > 0 && !neighbours)))) return TP3D_CPU_E_BADARG;
    if (!cluster_start) return 0;
    unsigned char *visited = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int64_t *stack = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!visited || !stack) {
        free(visited);
        free(stack);
        return TP3D_CPU_E_NOMEM;
    }
    int64_t kept = 0, used = 0;
    cluster_start[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (visited[i]) continue;
        const int64_t first = used;
        int64_t top = 0;
        visited[i] = 1;
        stack[top++] = i;
        members[used++] = i;
        while (top > 0) {
            const int64_t k = stack[--top];
            const int64_t *row = neighbours + k * width;
            for (int t = 0; t < width; ++t) {
                const int64_t nb = row[t];
                if (nb < 0) break;
                if (nb >= n || visited[nb]) continue;
                visited[nb] = 1;
                stack[top++] = nb;
                members[used++] = nb;
            }
        }
        if (used - first >= min_size) cluster_start[++kept] = used;
        else used = first; /* too small: its slots are reused (the points stay visited) */
    }
    free(visited);
    free(stack);
    return kept;
}
