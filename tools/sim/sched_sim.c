/*
 * sched_sim.c -- RESEARCH TOOL (CPU), not product code and not the oracle.
 *
 * Emulates, on the host, the *schedule* of the GPU fast mode (owner-computes rounds, Poisson edge counts, stale partner
 * rows) with f64 scalars, so that the fidelity of a schedule against the sequential reference loop
 * (src/embedder.rs:1167-1309, restated in oracle/annembed_oracle.c) can be measured before a kernel is written.
 * A "round" is one launch: every node replays its own samples against a snapshot of the other rows (Jacobi), pushes
 * to the edge targets follow one of the `push` modes below.
 *
 * build: gcc -O2 -fopenmp -shared -fPIC -o libschedsim.so sched_sim.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 32

typedef struct {
    uint64_t n, dim, nnz;
    const uint64_t *indptr; const uint32_t *nbr; const float *proba; const float *emb_scale;
    const uint64_t *tptr; const uint32_t *teid; const uint32_t *tsrc; /* transposed graph: in-edges of every node */
    float *y;                  /* n x dim */
    const float *hub_odds; const uint32_t *hub_alias;
    uint64_t seed;
    /* schedule */
    double per_round;          /* expected own samples per node and round */
    int push;                  /* 0 re-evaluate at the target against the source's row after its own samples (round-1 GPU kernel)
                                  1 single gradient parked by the source, added by the target in a phase of its own after the samples
                                  2 as 1, but added at the START of the next round's launch (fused: rows become visible a round later)
                                  3 re-evaluate against round-start rows */
    int carry;                 /* 1: consecutive draws of one edge see the target moved by the source's own earlier pushes */
    int alternate;             /* 1: in round r only edges with hash parity r & 1 are drawn (at twice the rate) */
    int tile;                  /* > 0: negatives of a node-round come from `tile_chunks` random windows of `tile` consecutive rows */
    int tile_chunks;
    int group;                 /* nodes sharing one tile (a workgroup) */
    int f32math;               /* 1: f32 scalar arithmetic as the kernel */
    int gs;                    /* 1: Gauss-Seidel order (rows read are current, single thread) -- a lower bound on staleness */
} sim_t;

static inline uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t *s) { return (double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint64_t mix(uint64_t a, uint64_t b) { uint64_t s = a * 0x9E3779B97F4A7C15ULL + b; return splitmix(&s); }

static uint32_t poisson(double mu, uint64_t *s) {
    double u = u01(s), p = exp(-mu), cdf = p;
    uint32_t c = 0;
    while (u >= cdf && c < 255) { c++; p *= mu / c; cdf += p; }
    return c;
}

static inline int row_has(const sim_t *c, uint32_t i, uint32_t k) {
    for (uint64_t e = c->indptr[i]; e < c->indptr[i + 1]; e++) if (c->nbr[e] == k) return 1;
    return 0;
}

/* attraction coefficient, embedder.rs:1207-1233 (b == 1) */
static inline double attract_c(double d, double w, double s, double step) {
    double ds = d / (s * s);
    if (!(ds > 0.)) return 0.;
    double coeff = 2. / (1. + ds) / (s * s);
    double rep = 1. / fmax(ds * ds, 1.0e4);
    return fmax(step * coeff * (-w + (1. - w) * rep), -0.49);
}
static inline double repulse_c(double d, double s, double step) {
    double ds = d / (s * s);
    double coeff = 2. / (1. + ds) / (s * s);
    return fmin(step * coeff / fmax(ds * ds, 1. / 16.), 2.);
}

/* one batch.  cnt: scratch u8[nnz]; G0 / G1: scratch f32[nnz * dim]; vis: scratch f32[n * dim]; returns samples drawn */
uint64_t sim_batch(const sim_t *c, uint64_t nb_sample, double step, uint32_t iter, uint8_t *cnt, float *G, float *vis, float *mid) {
    const uint64_t n = c->n, dim = c->dim;
    const double per_node = (double)nb_sample / (double)n;
    uint32_t rounds = (uint32_t)fmax(1., ceil(per_node / c->per_round));
    const double unit = per_node / rounds;
    uint64_t drawn = 0;
    memset(G, 0, sizeof(float) * c->nnz * dim);
    for (uint32_t r = 0; r < rounds; r++) {
        const uint64_t rkey = mix(c->seed, ((uint64_t)iter << 20) | r);
        /* counts (edge-keyed: source and target agree) */
#pragma omp parallel for schedule(static) reduction(+ : drawn)
        for (int64_t e = 0; e < (int64_t)c->nnz; e++) {
            uint64_t s = mix(rkey, (uint64_t)e);
            double mu = unit * c->proba[e];
            if (c->alternate) {
                if (((mix(c->seed ^ 0x1234, (uint64_t)e) >> 7) & 1) != (r & 1)) { cnt[e] = 0; continue; }
                mu *= 2.;
            }
            cnt[e] = (uint8_t)poisson(mu, &s);
            drawn += cnt[e];
        }
        if (!c->gs) memcpy(vis, (c->push == 2 && r > 0) ? mid : c->y, sizeof(float) * n * dim);
        const float *Yv = c->gs ? c->y : vis;
        /* phase B (+ fused push of the previous round for push == 2) */
#pragma omp parallel for schedule(dynamic, 256) if (!c->gs)
        for (int64_t v = 0; v < (int64_t)n; v++) {
            double y[MAXD], g[MAXD];
            float *Y = c->y + (uint64_t)v * dim;
            for (uint64_t t = 0; t < dim; t++) { y[t] = Y[t]; g[t] = 0.; }
            if (c->push == 2) {
                /* pushes parked in the previous round (G holds them until overwritten below: a node's in-edges are other
                   nodes' out-edges, so read them before anybody rewrites -- done in a separate pass, see `mid`) */
            }
            const double s = c->emb_scale[v];
            uint64_t rs = mix(rkey ^ 0xABCDEF, (uint64_t)v);
            /* tile windows of this node's workgroup */
            uint64_t tb[16];
            if (c->tile > 0) {
                uint64_t gs = mix(rkey ^ 0x7777, (uint64_t)(v / (c->group > 0 ? c->group : 1)));
                for (int q = 0; q < c->tile_chunks; q++) tb[q] = splitmix(&gs) % (n - (uint64_t)c->tile + 1);
            }
            for (uint64_t e = c->indptr[v]; e < c->indptr[v + 1]; e++) {
                if (!cnt[e]) continue;
                const uint32_t j = c->nbr[e];
                const double w = c->proba[e];
                double yj[MAXD], gsum[MAXD];
                for (uint64_t t = 0; t < dim; t++) { yj[t] = Yv[(uint64_t)j * dim + t]; gsum[t] = 0.; }
                for (uint32_t rep = 0; rep < cnt[e]; rep++) {
                    double d = 0.;
                    for (uint64_t t = 0; t < dim; t++) { double df = y[t] - yj[t]; d += df * df; }
                    if (c->f32math) d = (double)(float)d;
                    double cij = attract_c(d, w, s, step);
                    for (uint64_t t = 0; t < dim; t++) { g[t] = (yj[t] - y[t]) * cij; y[t] -= g[t]; gsum[t] += g[t]; }
                    if (c->carry) for (uint64_t t = 0; t < dim; t++) yj[t] += g[t];
                    int got = 0, attempts = 0;
                    while (got < 5 && attempts < 64) {
                        attempts++;
                        uint32_t k;
                        if (c->tile > 0) {
                            uint64_t q = splitmix(&rs) % (uint64_t)c->tile_chunks;
                            k = (uint32_t)(tb[q] + splitmix(&rs) % (uint64_t)c->tile);
                        } else if (c->hub_odds) {
                            uint64_t x = splitmix(&rs) % n;
                            k = (u01(&rs) < c->hub_odds[x]) ? (uint32_t)x : c->hub_alias[x];
                        } else k = (uint32_t)(splitmix(&rs) % n);
                        if (k == (uint32_t)v || k == j || row_has(c, (uint32_t)v, k)) continue;
                        got++;
                        double dk = 0.;
                        for (uint64_t t = 0; t < dim; t++) { double df = y[t] - Yv[(uint64_t)k * dim + t]; dk += df * df; }
                        if (dk > 0.) {
                            double cik = repulse_c(dk, s, step);
                            for (uint64_t t = 0; t < dim; t++) g[t] = (Yv[(uint64_t)k * dim + t] - y[t]) * cik;
                        }
                        for (uint64_t t = 0; t < dim; t++) y[t] -= g[t];
                    }
                }
                if (c->push == 1 || c->push == 2) for (uint64_t t = 0; t < dim; t++) G[e * dim + t] = (float)gsum[t];
                if (c->gs && (c->push == 1)) { /* sequential order: the target moves at once */
                    for (uint64_t t = 0; t < dim; t++) c->y[(uint64_t)j * dim + t] += (float)gsum[t];
                }
            }
            for (uint64_t t = 0; t < dim; t++) Y[t] = (float)y[t];
        }
        if (c->gs) continue;
        /* phase C */
        if (c->push == 0 || c->push == 3) {
            const float *src_rows = c->push == 0 ? mid : vis;
            if (c->push == 0) memcpy(mid, c->y, sizeof(float) * n * dim);
#pragma omp parallel for schedule(dynamic, 256)
            for (int64_t v = 0; v < (int64_t)n; v++) {
                double y[MAXD];
                float *Y = c->y + (uint64_t)v * dim;
                for (uint64_t t = 0; t < dim; t++) y[t] = Y[t];
                for (uint64_t x = c->tptr[v]; x < c->tptr[v + 1]; x++) {
                    const uint32_t e = c->teid[x], u = c->tsrc[x];
                    const double su = c->emb_scale[u], w = c->proba[e];
                    for (uint32_t rep = 0; rep < cnt[e]; rep++) {
                        double d = 0.;
                        for (uint64_t t = 0; t < dim; t++) { double df = y[t] - src_rows[(uint64_t)u * dim + t]; d += df * df; }
                        double cij = attract_c(d, w, su, step);
                        for (uint64_t t = 0; t < dim; t++) y[t] += (y[t] - src_rows[(uint64_t)u * dim + t]) * cij;
                    }
                }
                for (uint64_t t = 0; t < dim; t++) Y[t] = (float)y[t];
            }
        } else {
            /* push == 1: added now, visible to the next round.  push == 2: the same addition happens at the start of the
               next launch, i.e. the rows other nodes READ in the next round do not contain it yet: emulate by keeping the
               pushes out of the snapshot taken for the next round and adding them after the snapshot. */
            if (c->push == 2) memcpy(mid, c->y, sizeof(float) * n * dim); /* what the next round's readers see */
#pragma omp parallel for schedule(dynamic, 256)
            for (int64_t v = 0; v < (int64_t)n; v++) {
                float *Y = c->y + (uint64_t)v * dim;
                for (uint64_t x = c->tptr[v]; x < c->tptr[v + 1]; x++) {
                    const uint32_t e = c->teid[x];
                    if (!cnt[e]) continue;
                    for (uint64_t t = 0; t < dim; t++) Y[t] += G[(uint64_t)e * dim + t];
                }
            }
        }
    }
    return drawn;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Matching schedule: the directed edges are partitioned into classes in which every node occurs at most once (a proper
 * edge colouring of the multigraph).  A launch = one class: each of its edges draws c ~ Poisson(mu_e / sweeps) and replays
 * its c samples EXACTLY as the reference does (one gradient, both ends, embedder.rs:1228-1239; the 5 negatives read the rows
 * as they were when the launch started).  A batch = `sweeps` passes over all classes.
 * ------------------------------------------------------------------------------------------------------------------- */
/* greedy colouring in the given edge order; mask: scratch u64[n * words]; returns the number of colours */
uint32_t sim_colour_edges(uint64_t n, uint64_t nnz, const uint32_t *src, const uint32_t *dst, const uint32_t *order,
                          uint32_t words, uint64_t *mask, uint32_t *colour) {
    memset(mask, 0, sizeof(uint64_t) * n * words);
    uint32_t ncol = 0;
    for (uint64_t x = 0; x < nnz; x++) {
        const uint32_t e = order[x];
        uint64_t *a = mask + (uint64_t)src[e] * words, *b = mask + (uint64_t)dst[e] * words;
        uint32_t c = words * 64;
        for (uint32_t w = 0; w < words; w++) {
            const uint64_t free_ = ~(a[w] | b[w]);
            if (free_) { c = w * 64 + (uint32_t)__builtin_ctzll(free_); break; }
        }
        if (c >= words * 64) return 0xFFFFFFFFu;
        a[c / 64] |= 1ull << (c % 64);
        b[c / 64] |= 1ull << (c % 64);
        colour[e] = c;
        if (c + 1 > ncol) ncol = c + 1;
    }
    return ncol;
}

/* cptr / cedge: classes as CSR over edge ids; esrc: source of every edge.  neg_fresh: 1 = negatives read current rows */
uint64_t sim_batch_colour(const sim_t *c, uint64_t nb_sample, double step, uint32_t iter, uint32_t sweeps, uint32_t ncol,
                          const uint64_t *cptr, const uint32_t *cedge, const uint32_t *esrc, float *vis, int neg_fresh,
                          int shuffle_classes) {
    const uint64_t n = c->n, dim = c->dim;
    const double unit = (double)nb_sample / (double)n / (double)sweeps;
    uint64_t drawn = 0;
    uint32_t *perm = malloc(sizeof(uint32_t) * ncol);
    for (uint32_t sw = 0; sw < sweeps; sw++) {
        for (uint32_t q = 0; q < ncol; q++) perm[q] = q;
        if (shuffle_classes) {
            uint64_t ps = mix(c->seed ^ 0x5151, ((uint64_t)iter << 20) | sw);
            for (uint32_t q = ncol - 1; q > 0; q--) { uint32_t r = (uint32_t)(splitmix(&ps) % (q + 1)); uint32_t t = perm[q]; perm[q] = perm[r]; perm[r] = t; }
        }
        for (uint32_t qq = 0; qq < ncol; qq++) {
            const uint32_t col = perm[qq];
            const uint64_t rkey = mix(c->seed, (((uint64_t)iter << 20) | sw) * 4099 + col);
            if (!neg_fresh) memcpy(vis, c->y, sizeof(float) * n * dim);
            const float *Yv = neg_fresh ? c->y : vis;
#pragma omp parallel for schedule(dynamic, 512) reduction(+ : drawn)
            for (int64_t x = (int64_t)cptr[col]; x < (int64_t)cptr[col + 1]; x++) {
                const uint32_t e = cedge[x], i = esrc[e], j = c->nbr[e];
                uint64_t rs = mix(rkey, (uint64_t)e);
                uint32_t cnt;
                if (c->alternate) { /* stratified counts: evenly spaced over the sweeps, random phase per edge and batch */
                    uint64_t us = mix(c->seed ^ 0x9999, ((uint64_t)iter << 32) | e);
                    const double u = u01(&us), lam = unit * c->proba[e];
                    cnt = (uint32_t)(floor(lam * (sw + 1) + u) - floor(lam * sw + u));
                } else cnt = poisson(unit * c->proba[e], &rs);
                if (!cnt) continue;
                drawn += cnt;
                double yi[MAXD], yj[MAXD], g[MAXD];
                float *Yi = c->y + (uint64_t)i * dim, *Yj = c->y + (uint64_t)j * dim;
                for (uint64_t t = 0; t < dim; t++) { yi[t] = Yi[t]; yj[t] = Yj[t]; g[t] = 0.; }
                const double s = c->emb_scale[i], w = c->proba[e];
                for (uint32_t rep = 0; rep < cnt; rep++) {
                    double d = 0.;
                    for (uint64_t t = 0; t < dim; t++) { double df = yi[t] - yj[t]; d += df * df; }
                    const double cij = attract_c(d, w, s, step);
                    for (uint64_t t = 0; t < dim; t++) { g[t] = (yj[t] - yi[t]) * cij; yi[t] -= g[t]; yj[t] += g[t]; }
                    int got = 0, attempts = 0;
                    while (got < 5 && attempts < 64) {
                        attempts++;
                        uint32_t k;
                        if (c->hub_odds) {
                            uint64_t z = splitmix(&rs) % n;
                            k = (u01(&rs) < c->hub_odds[z]) ? (uint32_t)z : c->hub_alias[z];
                        } else k = (uint32_t)(splitmix(&rs) % n);
                        if (k == i || k == j || row_has(c, i, k)) continue;
                        got++;
                        double dk = 0.;
                        for (uint64_t t = 0; t < dim; t++) { double df = yi[t] - Yv[(uint64_t)k * dim + t]; dk += df * df; }
                        if (dk > 0.) {
                            const double cik = repulse_c(dk, s, step);
                            for (uint64_t t = 0; t < dim; t++) g[t] = (Yv[(uint64_t)k * dim + t] - yi[t]) * cik;
                        }
                        for (uint64_t t = 0; t < dim; t++) yi[t] -= g[t];
                    }
                }
                for (uint64_t t = 0; t < dim; t++) { Yi[t] = (float)yi[t]; Yj[t] = (float)yj[t]; }
            }
        }
    }
    free(perm);
    return drawn;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Premise check for the event-ordered design: the reference's sequential loop in i.i.d. order (node uniform, edge by the row's
 * cdf -- the ROWCDF sampler of the oracle), attraction exact on current rows, but the 5 NEGATIVES read a snapshot of the rows
 * that is refreshed only `refresh` times per batch.  f32math: the attraction/repulsion scalars in f32 with one reciprocal.
 * ------------------------------------------------------------------------------------------------------------------- */
uint64_t sim_batch_seq_stale_neg(const sim_t *c, uint64_t nb_sample, double step, uint32_t iter, uint32_t refresh, float *vis) {
    const uint64_t n = c->n, dim = c->dim;
    uint64_t rs = mix(c->seed ^ 0x4242, iter);
    const uint64_t chunk = refresh ? (nb_sample + refresh - 1) / refresh : nb_sample + 1;
    for (uint64_t sidx = 0; sidx < nb_sample; sidx++) {
        if (refresh && sidx % chunk == 0) memcpy(vis, c->y, sizeof(float) * n * dim);
        const float *Yv = refresh ? vis : c->y;
        const uint32_t i = (uint32_t)(splitmix(&rs) % n);
        const double u = u01(&rs);
        uint64_t b = c->indptr[i], len = c->indptr[i + 1] - b, m = len - 1;
        float acc = 0.f;
        for (uint64_t t = 0; t < len; t++) { acc += c->proba[b + t]; if (u < acc) { m = t; break; } }
        const uint64_t e = b + m;
        const uint32_t j = c->nbr[e];
        double yi[MAXD], yj[MAXD], g[MAXD];
        float *Yi = c->y + (uint64_t)i * dim, *Yj = c->y + (uint64_t)j * dim;
        for (uint64_t t = 0; t < dim; t++) { yi[t] = Yi[t]; yj[t] = Yj[t]; g[t] = 0.; }
        const double s = c->emb_scale[i], w = c->proba[e];
        double d = 0.;
        for (uint64_t t = 0; t < dim; t++) { double df = yi[t] - yj[t]; d += df * df; }
        double cij;
        if (c->f32math) {
            const float is2 = 1.0f / ((float)s * (float)s), delta = (float)d * is2, M = fmaxf(delta * delta, 1.0e4f);
            cij = (float)d > 0.f ? fmaxf(2.0f * (float)step * is2 * ((1.f - (float)w) - (float)w * M) / ((1.f + delta) * M), -0.49f) : 0.f;
        } else cij = attract_c(d, w, s, step);
        for (uint64_t t = 0; t < dim; t++) { g[t] = (yj[t] - yi[t]) * cij; yi[t] -= g[t]; yj[t] += g[t]; }
        for (uint64_t t = 0; t < dim; t++) Yj[t] = (float)yj[t];
        int got = 0;
        while (got < 5) {
            uint32_t k;
            if (c->hub_odds) {
                uint64_t z = splitmix(&rs) % n;
                k = (u01(&rs) < c->hub_odds[z]) ? (uint32_t)z : c->hub_alias[z];
            } else k = (uint32_t)(splitmix(&rs) % n);
            if (k == i || k == j || row_has(c, i, k)) continue;
            got++;
            double dk = 0.;
            for (uint64_t t = 0; t < dim; t++) { double df = yi[t] - Yv[(uint64_t)k * dim + t]; dk += df * df; }
            if (dk > 0.) {
                double cik;
                if (c->f32math) {
                    const float is2 = 1.0f / ((float)s * (float)s), delta = (float)dk * is2;
                    cik = fminf(2.0f * (float)step * is2 / ((1.f + delta) * fmaxf(delta * delta, 1.0f / 16.0f)), 2.0f);
                } else cik = repulse_c(dk, s, step);
                for (uint64_t t = 0; t < dim; t++) g[t] = (Yv[(uint64_t)k * dim + t] - yi[t]) * cik;
            }
            for (uint64_t t = 0; t < dim; t++) yi[t] -= g[t];
        }
        for (uint64_t t = 0; t < dim; t++) Yi[t] = (float)yi[t];
    }
    return nb_sample;
}
