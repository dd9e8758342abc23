/*
 * annembed_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU restatement of the annembed embedding hot path (crate v0.1.7), written from
 * the reference's Rust sources function by function.  Every function cites the `file:line` it
 * follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; nothing under annembed_amd/ links, imports or executes it.
 *
 * PARITY PINNING (see DESIGN.md "Oracle"):
 *   - pinned by the reference's own known-answer tests: the randomized-SVD path (done in
 *     oracle/oracle.py with LAPACK gesdd/geqrf/orgqr through scipy -- the very routines the
 *     reference calls through `lax`), src/tools/svdapprox.rs tests :1034,:1046,:1160,:1191,:1231,:1270,
 *     :1310,:1459,:1497,:1530,:1575 and src/graphlaplace.rs:362 (tests/test_oracle_golden.py).
 *   - PARITY UNPINNED (the reference holds no numeric test, cannot be built here -- no cargo/rustc --
 *     and is not reproducible run to run): to_proba_edges, the dmap kernel/density/laplacian, the CE
 *     SGD sample and the CE value.  For these the check is two independent restatements of the same
 *     cited lines (this file and the numpy one in tests/golden/make_golden.py) agreeing with each
 *     other.
 *   - the RNG streams (rand::rng() thread RNG at src/embedder.rs:1121,1182; Xoshiro256++ + Ziggurat
 *     at src/tools/svdapprox.rs:70-73; WeightedAliasIndex at src/embedder.rs:987) live in crates that
 *     are not under /root/reference (rand 0.9, rand_distr 0.5, rand_xoshiro 0.7; no Cargo.lock):
 *     PARITY UNPINNED.  The build defines its own counter-based stream (Philox4x32-10, below) and the
 *     device code reproduces exactly that.
 *   - the summation order of ndarray's Array1::sum() (q.sum(), src/diffmaps.rs:469,:546,:889,:932) lives in the
 *     ndarray crate (0.15 / 0.16, not vendored): restated from its published numeric_util::unrolled_fold --
 *     PARITY UNPINNED (ndarray_sum_f32 below); it moves q_mean by ~1 ulp.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PROBA_MIN 1.0e-4f /* src/embedder.rs:50 */

enum { ORC_OK = 0, ORC_ERR_ARG = 1, ORC_ERR_ISOLATED = 3, ORC_ERR_PROBA_RANGE = 4, ORC_ERR_BETA = 9 };

/* ============================================================================================ */
/* Philox4x32-10 counter RNG (Salmon et al., SC'11) and the build's stream convention           */
/* ============================================================================================ */
typedef struct {
    uint32_t key[2];
    uint32_t ctr[3]; /* c0, c1, c2 ; c3 is the running block index */
    uint32_t blk;
    uint32_t buf[4];
    int pos;
} orc_stream;

static void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }

static void stream_init(orc_stream *s, uint64_t seed, uint64_t c01, uint32_t c2) {
    s->key[0] = (uint32_t)seed; s->key[1] = (uint32_t)(seed >> 32);
    s->ctr[0] = (uint32_t)c01; s->ctr[1] = (uint32_t)(c01 >> 32); s->ctr[2] = c2;
    s->blk = 0; s->pos = 4;
}
static inline uint32_t stream_u32(orc_stream *s) {
    if (s->pos == 4) {
        uint32_t c[4] = {s->ctr[0], s->ctr[1], s->ctr[2], s->blk++};
        philox4x32_10(c, s->key, s->buf);
        s->pos = 0;
    }
    return s->buf[s->pos++];
}
static inline uint64_t stream_u64(orc_stream *s) {
    uint64_t hi = stream_u32(s);
    uint64_t lo = stream_u32(s);
    return (hi << 32) | lo;
}
/* uniform integer in [0, n): high 64 bits of a 64x64 product (bias < n * 2^-64) */
static inline uint64_t stream_index(orc_stream *s, uint64_t n) {
    return (uint64_t)(((unsigned __int128)stream_u64(s) * n) >> 64);
}
/* uniform f32 in [0,1) with 24 random bits */
static inline float stream_f32(orc_stream *s) { return (float)(stream_u32(s) >> 8) * (1.0f / 16777216.0f); }

#define ORC_TAG_OMEGA 0xFFFF0001u
#define ORC_TAG_PROJ 0xFFFF0002u
#define ORC_TAG_RANDINIT 0xFFFF0003u

/* N(0,1) fill, row-major, element e -> block e/4, lane e%4 (Box-Muller on word pairs).
 * Stands in for RandomGaussianMatrix::new, src/tools/svdapprox.rs:69-76 (seed 4664397). */
static inline void box_muller(uint32_t w0, uint32_t w1, float *z0, float *z1) {
    float u1 = (float)((w0 >> 8) + 1u) * (1.0f / 16777216.0f); /* (0,1] */
    float u2 = (float)(w1 >> 8) * (1.0f / 16777216.0f);        /* [0,1) */
    float r = sqrtf(-2.0f * logf(u1));
    float a = 6.28318530717958647692f * u2;
    *z0 = r * cosf(a);
    *z1 = r * sinf(a);
}
void orc_gaussian_fill(float *out, uint64_t count, uint64_t seed, uint32_t tag) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint64_t nblk = (count + 3) / 4;
    for (uint64_t b = 0; b < nblk; b++) {
        uint32_t c[4] = {(uint32_t)b, (uint32_t)(b >> 32), tag, 0}, w[4];
        float z[4];
        philox4x32_10(c, key, w);
        box_muller(w[0], w[1], &z[0], &z[1]);
        box_muller(w[2], w[3], &z[2], &z[3]);
        for (int t = 0; t < 4; t++)
            if (4 * b + t < count) out[4 * b + t] = z[t];
    }
}

/* ============================================================================================ */
/* a1. KGraph flattening: tail of kgraph_from_hnsw_all, src/fromhnsw/kgraph.rs:486-546           */
/* ============================================================================================ */
typedef struct { uint64_t key; uint32_t val; int used; } orc_hslot;
typedef struct { orc_hslot *slots; uint64_t cap; uint32_t count; } orc_indexset;
static uint64_t hash64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
static uint32_t indexset_insert_full(orc_indexset *s, uint64_t key, uint64_t *order) {
    uint64_t h = hash64(key) & (s->cap - 1);
    while (s->slots[h].used) {
        if (s->slots[h].key == key) return s->slots[h].val;
        h = (h + 1) & (s->cap - 1);
    }
    s->slots[h].used = 1; s->slots[h].key = key; s->slots[h].val = s->count;
    if (order) order[s->count] = key;
    return s->count++;
}
typedef struct { float d; uint32_t node; uint32_t pos; } orc_edge_tmp;
static int cmp_edge_tmp(const void *a, const void *b) {
    const orc_edge_tmp *x = a, *y = b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->pos > y->pos) - (x->pos < y->pos); /* ties: list order (the reference's
        sort_unstable_by leaves ties unspecified, kgraph.rs:508) */
}
/* out_indptr[n+1], out_nbr/out_dist sized n*nbng (upper bound), data_id_of_idx[n].
 * Returns ORC_ERR_ISOLATED if a point has no neighbour (:520-537). */
int orc_kgraph_from_ragged(const uint64_t *point_id, const uint64_t *row_ptr, const uint64_t *nbr_data_id,
                           const float *nbr_dist, uint64_t n, uint32_t nbng, uint64_t *out_indptr,
                           uint32_t *out_nbr, float *out_dist, uint64_t *data_id_of_idx) {
    orc_indexset set; set.cap = 16; while (set.cap < 4 * n + 16) set.cap <<= 1;
    set.slots = calloc(set.cap, sizeof(orc_hslot)); set.count = 0;
    uint64_t *order = malloc(sizeof(uint64_t) * (n + 1));
    /* rows are produced in point iteration order but stored at slot `index` (:544) */
    uint32_t *row_len = calloc(n, sizeof(uint32_t));
    uint32_t *tmp_nbr = malloc(sizeof(uint32_t) * n * nbng);
    float *tmp_dist = malloc(sizeof(float) * n * nbng);
    uint64_t maxlen = 0;
    for (uint64_t p = 0; p < n; p++) if (row_ptr[p + 1] - row_ptr[p] > maxlen) maxlen = row_ptr[p + 1] - row_ptr[p];
    orc_edge_tmp *vec = malloc(sizeof(orc_edge_tmp) * (maxlen + 1));
    int rc = ORC_OK;
    for (uint64_t p = 0; p < n && rc == ORC_OK; p++) {
        uint32_t index = indexset_insert_full(&set, point_id[p], order); /* :489 */
        uint64_t len = row_ptr[p + 1] - row_ptr[p];
        for (uint64_t t = 0; t < len; t++) {
            uint32_t nidx = indexset_insert_full(&set, nbr_data_id[row_ptr[p] + t], order); /* :500 */
            if (set.count > n) { rc = ORC_ERR_ARG; break; }
            vec[t].d = nbr_dist[row_ptr[p] + t]; vec[t].node = nidx; vec[t].pos = (uint32_t)t;
        }
        if (rc != ORC_OK) break;
        if (index >= n) { rc = ORC_ERR_ARG; break; }
        qsort(vec, len, sizeof(orc_edge_tmp), cmp_edge_tmp); /* :508 */
        if (len == 0) { rc = ORC_ERR_ISOLATED; break; }       /* :520-537 */
        uint32_t keep = len < nbng ? (uint32_t)len : nbng;     /* :539 */
        row_len[index] = keep;
        for (uint32_t t = 0; t < keep; t++) { tmp_nbr[(uint64_t)index * nbng + t] = vec[t].node; tmp_dist[(uint64_t)index * nbng + t] = vec[t].d; }
    }
    if (rc == ORC_OK) {
        out_indptr[0] = 0;
        for (uint64_t i = 0; i < n; i++) {
            if (row_len[i] == 0) { rc = ORC_ERR_ISOLATED; break; }
            out_indptr[i + 1] = out_indptr[i] + row_len[i];
            memcpy(out_nbr + out_indptr[i], tmp_nbr + i * nbng, sizeof(uint32_t) * row_len[i]);
            memcpy(out_dist + out_indptr[i], tmp_dist + i * nbng, sizeof(float) * row_len[i]);
        }
        if (data_id_of_idx) memcpy(data_id_of_idx, order, sizeof(uint64_t) * n);
    }
    free(set.slots); free(order); free(row_len); free(tmp_nbr); free(tmp_dist); free(vec);
    return rc;
}

/* Hubness::new, src/fromhnsw/hubness.rs:51-67 */
void orc_hubness(uint64_t n, const uint64_t *indptr, const uint32_t *nbr, uint32_t *counts) {
    memset(counts, 0, sizeof(uint32_t) * n);
    for (uint64_t e = 0; e < indptr[n]; e++) counts[nbr[e]]++;
}

/* ============================================================================================ */
/* a2. to_proba_edges / get_scale_from_proba_normalisation, src/tools/kdumap.rs:26-116, 132-235   */
/* ============================================================================================ */
int orc_to_proba_edges(uint64_t n, const uint64_t *indptr, const uint32_t *nbr, const float *dist,
                       float scale_rho, float beta, float *proba, float *scale_out) {
    for (uint64_t i = 0; i < n; i++) {
        uint64_t b = indptr[i], nb = indptr[i + 1] - b;
        if (nb == 0) return ORC_ERR_ISOLATED; /* :75-85 */
        const float *d = dist + b;
        float rho_x = d[0];                    /* :146 */
        float sum = 0.f;
        for (uint64_t m = 0; m < nb; m++) sum += dist[indptr[nbr[b + m]]]; /* :149-152 first dist of y_i */
        sum += rho_x;                          /* :154 */
        float mean_rho = sum / (float)(nb + 1); /* :155 */
        float scale = scale_rho * mean_rho;    /* :159 */
        scale_out[i] = scale;
        int all_equal = 0;
        float first_dist = d[0];
        int64_t last = -1;                     /* :164-166 rfind(weight > 0) */
        for (int64_t m = (int64_t)nb - 1; m >= 0; m--) if (d[m] > 0.f) { last = m; break; }
        if (last < 0) all_equal = 1;           /* :167-170 */
        if (!all_equal) {
            float last_dist = d[last];
            if (last_dist > first_dist) {      /* :178 */
                float s = 0.f;
                for (uint64_t m = 0; m < nb; m++) {
                    float w = expf(-powf(fmaxf(d[m] - first_dist, 0.f) / scale, beta)); /* :172-174 */
                    w = fmaxf(w, PROBA_MIN);   /* :185 */
                    proba[b + m] = w;
                }
                float proba_range = proba[b + nb - 1] / proba[b]; /* :190 */
                if (!(proba_range >= PROBA_MIN)) return ORC_ERR_PROBA_RANGE; /* :209 */
                for (uint64_t m = 0; m < nb; m++) s += proba[b + m]; /* :215 */
                for (uint64_t m = 0; m < nb; m++) proba[b + m] /= s; /* :216-218 */
                continue;
            } else all_equal = 1;              /* :221 */
        }
        for (uint64_t m = 0; m < nb; m++) proba[b + m] = 1.0f / (float)nb; /* :224-230 */
    }
    return ORC_OK;
}

/* NodeParam::get_perplexity, src/tools/nodeparam.rs:88-91 */
void orc_perplexity(uint64_t n, const uint64_t *indptr, const float *proba, float *perp) {
    for (uint64_t i = 0; i < n; i++) {
        float h = 0.f;
        for (uint64_t e = indptr[i]; e < indptr[i + 1]; e++) h += -proba[e] * logf(proba[e]);
        perp[i] = expf(h);
    }
}

/* ============================================================================================ */
/* a3. dmap node scales and kernel, src/diffmaps.rs:752-849, 590-675, 1020-1043                   */
/* ============================================================================================ */
/* get_dist_l2_from_node (:1020-1043) + normalisation (:801-822).  local_scales[n] (zeros replaced
 * by mean), normed[n] = local/mean, *mean_scale. */
int orc_dmap_local_scales(uint64_t n, const uint64_t *indptr, const float *dist, uint32_t nbng,
                          float *local_scales, float *normed, float *mean_scale) {
    float sum = 0.f;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t b = indptr[i], len = indptr[i + 1] - b;
        float d2 = 0.f;
        for (uint64_t m = 0; m < len && m < nbng; m++) d2 += dist[b + m] * dist[b + m]; /* :1032-1036 */
        local_scales[i] = len ? sqrtf(d2 / (float)len) : 0.f; /* :1038-1042 (divisor = len) */
    }
    for (uint64_t i = 0; i < n; i++) sum += local_scales[i]; /* :801 */
    float mean = sum / (float)n;                              /* :803 */
    if (!(mean > 0.f)) return ORC_ERR_ARG;                    /* :805 */
    for (uint64_t i = 0; i < n; i++) if (local_scales[i] <= 0.f) local_scales[i] = mean; /* :806-810 */
    for (uint64_t i = 0; i < n; i++) normed[i] = local_scales[i] / mean; /* :815-816 */
    *mean_scale = mean;                                       /* :817 */
    return ORC_OK;
}

/* scales_to_nodeparams / build_node_param (:590-675, :691-745) with remap_weight of :831-834.
 * Output rows have len+1 entries, self edge first: kcols/kvals sized nnz + n, kindptr[n+1].
 * *nb_too_low counts the PROBA_MIN clamps (:658-661). */
int orc_dmap_kernel(uint64_t n, const uint64_t *indptr, const uint32_t *nbr, const float *dist,
                    const float *scales, float epsil_param, uint64_t *kindptr, uint32_t *kcols,
                    float *kvals, uint64_t *nb_too_low) {
    float epsil = sqrtf(epsil_param); /* :824 */
    uint64_t low = 0;
    kindptr[0] = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t b = indptr[i], len = indptr[i + 1] - b, o = kindptr[i];
        if (len == 0) return ORC_ERR_ISOLATED; /* :611-615 */
        kindptr[i + 1] = o + len + 1;
        int all_equal = 0;
        int64_t last = -1;
        for (int64_t m = (int64_t)len - 1; m >= 0; m--) if (dist[b + m] > 0.f) { last = m; break; } /* :618-620 */
        if (last >= 0) { if (dist[b + last] <= dist[b]) all_equal = 1; } else all_equal = 1; /* :622-630 */
        kcols[o] = (uint32_t)i;
        if (all_equal) {
            float p = 1.0f / (float)(len + 1); /* :642 */
            kvals[o] = p;
            for (uint64_t m = 0; m < len; m++) { kcols[o + 1 + m] = nbr[b + m]; kvals[o + 1 + m] = p; }
        } else {
            float from_scale = scales[i];
            for (uint64_t m = 0; m < len; m++) {
                float to_scale = scales[nbr[b + m]];
                float local_scale = sqrtf(to_scale * from_scale);                 /* :656 */
                float arg = powf((dist[b + m] - 0.f) / (epsil * local_scale), 2.0f); /* :832 */
                float w = expf(-arg);                                             /* :833 */
                if (w < PROBA_MIN) { w = PROBA_MIN; low++; }                      /* :658-661 */
                kcols[o + 1 + m] = nbr[b + m]; kvals[o + 1 + m] = w;
            }
            kvals[o] = 1.0f; /* :667-668 (nb_edges > 1 always here) */
        }
    }
    if (nb_too_low) *nb_too_low = low;
    return ORC_OK;
}

/* -------------------------------------------------------------------------------------------- */
/* CSR-branch symmetrisation shared by kernel0_to_density (:898-929) and compute_laplacian         */
/* (:513-544).  The reference iterates a HashMap<(i,j),w>; the iteration order only changes f32    */
/* summation order.  Here: keys sorted by (i,j); a duplicate (i,j) inside a row keeps the LAST      */
/* weight (HashMap::insert overwrites).                                                            */
/* -------------------------------------------------------------------------------------------- */
typedef struct { uint32_t i, j; float w; uint32_t seq; } orc_trip;
static int cmp_trip(const void *a, const void *b) {
    const orc_trip *x = a, *y = b;
    if (x->i != y->i) return x->i < y->i ? -1 : 1;
    if (x->j != y->j) return x->j < y->j ? -1 : 1;
    return (x->seq > y->seq) - (x->seq < y->seq);
}
/* unique sorted key list from kernel rows; returns count; row_start[n+1] */
static uint64_t build_edge_map(uint64_t n, const uint64_t *kindptr, const uint32_t *kcols, const float *kvals,
                               orc_trip **out, uint64_t **row_start_out) {
    uint64_t nnz = kindptr[n];
    orc_trip *t = malloc(sizeof(orc_trip) * (nnz ? nnz : 1));
    uint64_t c = 0;
    for (uint64_t i = 0; i < n; i++)
        for (uint64_t e = kindptr[i]; e < kindptr[i + 1]; e++) { t[c].i = (uint32_t)i; t[c].j = kcols[e]; t[c].w = kvals[e]; t[c].seq = (uint32_t)(e - kindptr[i]); c++; }
    qsort(t, c, sizeof(orc_trip), cmp_trip);
    uint64_t u = 0;
    for (uint64_t x = 0; x < c; x++) {
        if (x + 1 < c && t[x + 1].i == t[x].i && t[x + 1].j == t[x].j) continue; /* later insert wins */
        t[u++] = t[x];
    }
    uint64_t *rs = calloc(n + 1, sizeof(uint64_t));
    for (uint64_t x = 0; x < u; x++) rs[t[x].i + 1]++;
    for (uint64_t i = 0; i < n; i++) rs[i + 1] += rs[i];
    *out = t; *row_start_out = rs;
    return u;
}
static int lookup(const orc_trip *t, const uint64_t *rs, uint32_t i, uint32_t j, float *w) {
    uint64_t lo = rs[i], hi = rs[i + 1];
    while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (t[mid].j < j) lo = mid + 1; else hi = mid; }
    if (lo < rs[i + 1] && t[lo].j == j) { *w = t[lo].w; return 1; }
    return 0;
}

/* ndarray's Array1::sum() on a contiguous array (ndarray 0.15/0.16 numeric_util::unrolled_fold, a dependency that is
 * not vendored under /root/reference: order restated from its published source -- parity unpinned): eight interleaved
 * accumulators p_k += x[8 t + k], combined as acc = 0 + (p0 + p4) + (p1 + p5) + (p2 + p6) + (p3 + p7), then the
 * tail elements in order. */
static float ndarray_sum_f32(const float *x, uint64_t n) {
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint64_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int k = 0; k < 8; k++) p[k] = p[k] + x[i + k];
    float acc = 0.f;
    acc = acc + (p[0] + p[4]);
    acc = acc + (p[1] + p[5]);
    acc = acc + (p[2] + p[6]);
    acc = acc + (p[3] + p[7]);
    for (; i < n; i++) acc = acc + x[i];
    return acc;
}
float orc_ndarray_sum_f32(const float *x, uint64_t n) { return ndarray_sum_f32(x, n); }

/* a4. kernel0_to_density, CSR branch, src/diffmaps.rs:898-942: q[n] (normalised density),
 * beta_scales[n] = q^beta * mean_scale. */
void orc_dmap_density_csr(uint64_t n, const uint64_t *kindptr, const uint32_t *kcols, const float *kvals,
                          uint32_t max_nbng, float beta, float mean_scale, float *q, float *beta_scales) {
    orc_trip *t; uint64_t *rs;
    uint64_t u = build_edge_map(n, kindptr, kcols, kvals, &t, &rs);
    for (uint64_t i = 0; i < n; i++) q[i] = 0.f;
    for (uint64_t x = 0; x < u; x++) { /* :912-929 */
        float tv, sym = t[x].w;
        if (lookup(t, rs, t[x].j, t[x].i, &tv)) sym = fmaxf(t[x].w, tv);
        q[t[x].i] += sym;
        q[t[x].j] += sym;
    }
    for (uint64_t i = 0; i < n; i++) q[i] /= (float)max_nbng; /* :931 */
    float s = ndarray_sum_f32(q, n);                         /* q.sum(): ndarray order */
    float q_mean = s / (float)n;                             /* :932 */
    for (uint64_t i = 0; i < n; i++) q[i] /= q_mean;         /* :933 */
    for (uint64_t i = 0; i < n; i++) beta_scales[i] = powf(q[i], beta) * mean_scale; /* :938-942 */
    free(t); free(rs);
}

/* a5. compute_laplacian, CSR branch, src/diffmaps.rs:513-584.
 * Output CSR (columns sorted, duplicates summed = TriMat::to_csr): lindptr[n+1], lcols/lvals sized
 * 2*kindptr[n] (upper bound), normalizer[n] = sqrt(degrees). Returns nnz through *lnnz. */
void orc_dmap_laplacian_csr(uint64_t n, const uint64_t *kindptr, const uint32_t *kcols, const float *kvals,
                            uint32_t max_nbng, float alfa, uint64_t *lindptr, uint32_t *lcols, float *lvals,
                            uint64_t *lnnz, float *normalizer) {
    orc_trip *t; uint64_t *rs;
    uint64_t u = build_edge_map(n, kindptr, kcols, kvals, &t, &rs);
    uint64_t nt = 2 * u;
    uint32_t *rows = malloc(sizeof(uint32_t) * (nt ? nt : 1)), *cols = malloc(sizeof(uint32_t) * (nt ? nt : 1));
    float *vals = malloc(sizeof(float) * (nt ? nt : 1));
    float *q = calloc(n, sizeof(float)), *deg = calloc(n, sizeof(float));
    for (uint64_t x = 0; x < u; x++) { /* :527-544 */
        float tv, sym = t[x].w;
        if (lookup(t, rs, t[x].j, t[x].i, &tv)) sym = fmaxf(t[x].w, tv);
        rows[2 * x] = t[x].i; cols[2 * x] = t[x].j; vals[2 * x] = sym; q[t[x].i] += sym;
        rows[2 * x + 1] = t[x].j; cols[2 * x + 1] = t[x].i; vals[2 * x + 1] = sym; q[t[x].j] += sym;
    }
    float qs = ndarray_sum_f32(q, n);                         /* q.sum(): ndarray order */
    float q_mean = qs / (float)max_nbng;                      /* :546 (sic: / max_nbng) */
    for (uint64_t i = 0; i < n; i++) q[i] /= q_mean;          /* :548 */
    for (uint64_t x = 0; x < nt; x++) vals[x] /= powf(q[rows[x]] * q[cols[x]], alfa); /* :553-557 */
    for (uint64_t x = 0; x < nt; x++) deg[rows[x]] += vals[x]; /* :561-564 */
    for (uint64_t i = 0; i < n; i++) normalizer[i] = sqrtf(deg[i]); /* :565 */
    for (uint64_t x = 0; x < nt; x++) vals[x] /= normalizer[rows[x]] * normalizer[cols[x]]; /* :566-570 */
    /* TriMat::to_csr: sort by (row, col), sum duplicates (:572-578) */
    orc_trip *tt = malloc(sizeof(orc_trip) * (nt ? nt : 1));
    for (uint64_t x = 0; x < nt; x++) { tt[x].i = rows[x]; tt[x].j = cols[x]; tt[x].w = vals[x]; tt[x].seq = (uint32_t)x; }
    /* seq may overflow u32 only above 4G triplets: not reachable in the oracle's sizes */
    qsort(tt, nt, sizeof(orc_trip), cmp_trip);
    uint64_t c = 0;
    for (uint64_t i = 0; i <= n; i++) lindptr[i] = 0;
    for (uint64_t x = 0; x < nt;) {
        uint64_t y = x; float s = 0.f;
        while (y < nt && tt[y].i == tt[x].i && tt[y].j == tt[x].j) { s += tt[y].w; y++; }
        lcols[c] = tt[x].j; lvals[c] = s; lindptr[tt[x].i + 1]++; c++;
        x = y;
    }
    for (uint64_t i = 0; i < n; i++) lindptr[i + 1] += lindptr[i];
    *lnnz = c;
    free(t); free(rs); free(rows); free(cols); free(vals); free(q); free(deg); free(tt);
}

/* a9. embed_from_laplacian rows, src/diffmaps.rs:1213-1236.  s[r], u[n*r] row-major.
 * t: diffusion time (has_t) else min(5, ln .9 / ln(l2/l1)) (:1214-1217). y0[n*real_dim]. */
int orc_embed_from_svd(uint64_t n, uint64_t r, const float *s, const float *u, const float *normalizer,
                       const float *normed_scales, uint64_t asked_dim, float t, int has_t, float *y0,
                       uint64_t *real_dim_out) {
    if (r > 2 && s[1] > s[0]) return 6; /* :1176 spectrum not decreasing */
    uint64_t real_dim = asked_dim < r - 1 ? asked_dim : r - 1; /* :1207 */
    float *nl = malloc(sizeof(float) * r);
    for (uint64_t j = 0; j < r; j++) nl[j] = s[j] / s[0]; /* :1213 */
    float time = has_t ? t : fminf(5.0f, logf(0.9f) / logf(nl[2] / nl[1])); /* :1214-1217 */
    float sd = 0.f;
    for (uint64_t i = 0; i < n; i++) sd += normalizer[i];
    float sum_diag = sd / (float)n; /* :1223 */
    for (uint64_t i = 0; i < n; i++) {
        float weight_i = normed_scales[i] * sqrtf(normalizer[i] / sum_diag); /* :1228 */
        for (uint64_t j = 0; j < real_dim; j++) {
            float v = powf(nl[j + 1], time) * u[i * r + j + 1] / weight_i; /* :1232 */
            if (v > 10.0f) v = 10.0f; else if (v < -10.0f) v = -10.0f;    /* clip, src/tools/clip.rs */
            y0[i * real_dim + j] = v;
        }
    }
    free(nl);
    *real_dim_out = real_dim;
    return ORC_OK;
}

/* ============================================================================================ */
/* a10. set_data_box, src/embedder.rs:1376-1408                                                   */
/* ============================================================================================ */
void orc_set_data_box(float *data, uint64_t n, uint64_t dim, float box_size) {
    float max_max = 0.f;
    for (uint64_t j = 0; j < dim; j++) {
        float m = 0.f;
        for (uint64_t i = 0; i < n; i++) m += data[i * dim + j]; /* :1391-1393 */
        m /= (float)n;                                            /* :1394 */
        for (uint64_t i = 0; i < n; i++) {                        /* :1396-1401 */
            data[i * dim + j] -= m;
            max_max = fmaxf(max_max, fabsf(data[i * dim + j]));
        }
    }
    max_max /= box_size / 2.0f;                                   /* :1403 */
    for (uint64_t x = 0; x < n * dim; x++) data[x] /= max_max;    /* :1404-1407 */
}

/* a11. estimate_embedded_scales_from_initial_scales, src/embedder.rs:1356-1373 */
void orc_embedded_scales(const float *initial_scales, uint64_t n, float *out) {
    float s = 0.f;
    for (uint64_t i = 0; i < n; i++) s += initial_scales[i];
    float mean_scale = s / (float)n;                              /* :1358 */
    for (uint64_t i = 0; i < n; i++)                              /* :1363-1366 */
        out[i] = 0.2f * fmaxf(fminf(initial_scales[i] / mean_scale, 4.0f), 0.25f);
}

/* Walker/Vose alias table; stands in for rand_distr::WeightedAliasIndex::new (src/embedder.rs:919,987;
 * crate not under /root/reference -- construction order is the build's own, see header). */
void orc_alias_build(const float *w, uint64_t n, float *odds, uint32_t *alias) {
    double sum = 0.;
    for (uint64_t i = 0; i < n; i++) sum += (double)w[i];
    double *q = malloc(sizeof(double) * (n ? n : 1));
    uint32_t *small = malloc(sizeof(uint32_t) * (n ? n : 1)), *large = malloc(sizeof(uint32_t) * (n ? n : 1));
    uint64_t ns = 0, nl = 0;
    for (uint64_t i = 0; i < n; i++) {
        q[i] = (double)w[i] * (double)n / sum;
        if (q[i] < 1.0) small[ns++] = (uint32_t)i; else large[nl++] = (uint32_t)i;
    }
    while (ns > 0 && nl > 0) {
        uint32_t s = small[--ns], l = large[--nl];
        odds[s] = (float)q[s]; alias[s] = l;
        q[l] = (q[l] + q[s]) - 1.0;
        if (q[l] < 1.0) small[ns++] = l; else large[nl++] = l;
    }
    while (nl > 0) { uint32_t l = large[--nl]; odds[l] = 1.0f; alias[l] = l; }
    while (ns > 0) { uint32_t s = small[--ns]; odds[s] = 1.0f; alias[s] = s; }
    free(q); free(small); free(large);
}

/* ============================================================================================ */
/* a12-a13. EntropyOptim, src/embedder.rs:936-1345                                                */
/* ============================================================================================ */
typedef struct {
    uint64_t n, dim, nnz;
    const uint64_t *indptr; const uint32_t *nbr; const float *proba; /* NodeParams as CSR */
    const float *emb_scale;
    float *y;                 /* n x dim, updated in place */
    double b;
    uint64_t seed;
    int sampler;              /* 0 ROWCDF, 1 ALIAS */
    uint64_t node_lo, node_hi; /* positive edges are drawn from sources in [lo,hi) */
    const float *edge_odds; const uint32_t *edge_alias; const uint32_t *edge_src; /* ALIAS sampler, over edges of [lo,hi) */
    const float *hub_odds; const uint32_t *hub_alias; /* optional NodeSampler (:909-931) */
} orc_ce;

static inline int row_has(const orc_ce *c, uint32_t i, uint32_t k) { /* NodeParam::get_edge, nodeparam.rs:83-85 */
    for (uint64_t e = c->indptr[i]; e < c->indptr[i + 1]; e++) if (c->nbr[e] == k) return 1;
    return 0;
}

/* plan of one sample: the nodes it touches (deterministic given the graph and the stream) */
typedef struct { uint32_t i, j, k[5]; float w; } orc_plan;

static int sample_plan(const orc_ce *c, uint64_t s, uint32_t iter, orc_plan *p) {
    orc_stream st; stream_init(&st, c->seed, s, iter);
    uint64_t e;
    if (c->sampler == 0) {
        uint32_t i = (uint32_t)(c->node_lo + stream_index(&st, c->node_hi - c->node_lo));
        float u = stream_f32(&st);
        uint64_t b = c->indptr[i], len = c->indptr[i + 1] - b, m = len - 1;
        float acc = 0.f;
        for (uint64_t t = 0; t < len; t++) { acc += c->proba[b + t]; if (u < acc) { m = t; break; } }
        e = b + m; p->i = i;
    } else {
        uint64_t e0 = c->indptr[c->node_lo], ne = c->indptr[c->node_hi] - e0;
        uint64_t x = stream_index(&st, ne);
        float u = stream_f32(&st);
        if (!(u < c->edge_odds[x])) x = c->edge_alias[x];
        e = e0 + x; p->i = c->edge_src[x];
    }
    p->j = c->nbr[e]; p->w = c->proba[e];
    int got = 0; uint64_t attempts = 0;
    while (got < 5) { /* :1241-1299 */
        uint32_t k;
        if (c->hub_odds) { /* NodeSampler::sample :927-930 */
            uint64_t x = stream_index(&st, c->n);
            float u = stream_f32(&st);
            k = (u < c->hub_odds[x]) ? (uint32_t)x : c->hub_alias[x];
        } else k = (uint32_t)stream_index(&st, c->n); /* :1121 */
        if (++attempts > (1u << 20)) return ORC_ERR_ARG;
        if (k == p->i || k == p->j || row_has(c, p->i, k)) continue; /* :1246-1253 */
        p->k[got++] = k;
    }
    return ORC_OK;
}
int orc_ce_plan(const orc_ce *c, uint64_t s, uint32_t iter, uint32_t *nodes7, float *w) {
    orc_plan p; int rc = sample_plan(c, s, iter, &p);
    nodes7[0] = p.i; nodes7[1] = p.j; for (int t = 0; t < 5; t++) nodes7[2 + t] = p.k[t];
    *w = p.w; return rc;
}

#define ORC_MAXDIM 64
/* ce_optim_edge_shannon, src/embedder.rs:1167-1302: one SGD sample applied to c->y */
static void apply_sample_at(const orc_ce *c, const orc_plan *p, double grad_step, float *y, uint64_t stride) {
    uint64_t dim = c->dim;
    float yi[ORC_MAXDIM], yj[ORC_MAXDIM], grad[ORC_MAXDIM];
    float *Yi = y + (uint64_t)p->i * stride, *Yj = y + (uint64_t)p->j * stride;
    for (uint64_t t = 0; t < dim; t++) { yi[t] = Yi[t]; yj[t] = Yj[t]; grad[t] = 0.f; } /* :1185-1186,:1199 */
    double weight = (double)p->w;                  /* :1202 */
    double scale = (double)c->emb_scale[p->i];     /* :1204 */
    double b = c->b;
    float acc = 0.f;
    for (uint64_t t = 0; t < dim; t++) { float df = yi[t] - yj[t]; acc += df * df; } /* :1207-1211 sum::<F> */
    double d_ij = (double)acc;
    double d_ij_scaled = d_ij / (scale * scale);   /* :1214 */
    double coeff;
    if (b != 1.) {                                 /* :1216-1222 */
        double cw = 1. / (1. + pow(d_ij_scaled, b));
        coeff = 2. * b * cw * pow(d_ij_scaled, b - 1.) / (scale * scale);
    } else {
        double cw = 1. / (1. + d_ij_scaled);
        coeff = 2. * b * cw / (scale * scale);
    }
    if (d_ij_scaled > 0.) {                        /* :1223-1236 */
        double alfa = (double)(1.0f / PROBA_MIN);
        double coeff_repulsion = 1. / fmax(d_ij_scaled * d_ij_scaled, alfa);
        double coeff_ij = fmax(grad_step * coeff * (-weight + (1. - weight) * coeff_repulsion), -0.49);
        float cf = (float)coeff_ij;
        for (uint64_t t = 0; t < dim; t++) grad[t] = (yj[t] - yi[t]) * cf;
    }
    for (uint64_t t = 0; t < dim; t++) { yi[t] -= grad[t]; yj[t] += grad[t]; } /* :1237-1238 */
    for (uint64_t t = 0; t < dim; t++) Yj[t] = yj[t];                           /* :1239 */
    for (int g = 0; g < 5; g++) {                  /* :1244-1299 */
        const float *Yk = y + (uint64_t)p->k[g] * stride;
        float yk[ORC_MAXDIM];
        for (uint64_t t = 0; t < dim; t++) yk[t] = Yk[t];
        float ak = 0.f;
        for (uint64_t t = 0; t < dim; t++) { float df = yi[t] - yk[t]; ak += df * df; } /* :1267-1271 */
        double d_ik = (double)ak;
        double d_ik_scaled = d_ik / (scale * scale); /* :1274 */
        double cf2;
        if (b != 1.) {                             /* :1275-1281 */
            double cw = 1. / (1. + pow(d_ik_scaled, b));
            cf2 = 2. * b * cw * pow(d_ik_scaled, b - 1.) / (scale * scale);
        } else {
            double cw = 1. / (1. + d_ik_scaled);
            cf2 = 2. * b * cw / (scale * scale);
        }
        double alfa = 1. / 16.;                    /* :1285 */
        if (d_ik > 0.) {                           /* :1286-1295 */
            double coeff_repulsion = 1. / fmax(d_ik_scaled * d_ik_scaled, alfa);
            double coeff_ik = fmin(grad_step * cf2 * coeff_repulsion, 2.);
            float cf = (float)coeff_ik;
            for (uint64_t t = 0; t < dim; t++) grad[t] = (yk[t] - yi[t]) * cf;
        }                                          /* else: `gradient` keeps its previous value (B4) */
        for (uint64_t t = 0; t < dim; t++) yi[t] -= grad[t]; /* :1297 */
    }
    for (uint64_t t = 0; t < dim; t++) Yi[t] = yi[t]; /* :1301 */
}
static void apply_sample(const orc_ce *c, const orc_plan *p, double grad_step) { apply_sample_at(c, p, grad_step, c->y, c->dim); }

/* gradient_iteration (sequential), src/embedder.rs:1305-1309: samples s = s_begin .. s_begin+nb_sample-1 */
int orc_gradient_iteration(const orc_ce *c, uint64_t s_begin, uint64_t nb_sample, double grad_step, uint32_t iter) {
    if (c->dim > ORC_MAXDIM) return ORC_ERR_ARG;
    for (uint64_t s = 0; s < nb_sample; s++) {
        orc_plan p; int rc = sample_plan(c, s_begin + s, iter, &p);
        if (rc) return rc;
        apply_sample(c, &p, grad_step);
    }
    return ORC_OK;
}

/* gradient_iteration_threaded, src/embedder.rs:1311-1315: lock-free (Hogwild) over OpenMP threads.
 * Used as the CPU baseline.  The reference guards each row by an RwLock (:942) but releases it
 * between read and write-back (:1185-1186,:1239,:1301); the races are the same.  Like the reference, whose rows are separate
 * heap allocations (Arc<RwLock<Array1<F>>>, :994-998), every row sits on a cache line of its own during the run (rows packed
 * 8 to a line make 128 threads fight over every line: measured 3.4x one thread); chunks of 4096 samples are handed out
 * dynamically, as rayon's work stealing does. */
int orc_gradient_iteration_hogwild(const orc_ce *c, uint64_t nb_sample, double grad_step, uint32_t iter, int nthreads) {
    if (c->dim > ORC_MAXDIM) return ORC_ERR_ARG;
    int err = 0;
    const uint64_t stride = ((c->dim + 15) / 16) * 16; /* floats: multiples of 64 bytes */
    float *pad = NULL;
    if (posix_memalign((void **)&pad, 64, sizeof(float) * stride * c->n) != 0) return ORC_ERR_ARG;
    for (uint64_t i = 0; i < c->n; i++) memcpy(pad + i * stride, c->y + i * c->dim, sizeof(float) * c->dim);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 4096) reduction(| : err)
#endif
    for (int64_t s = 0; s < (int64_t)nb_sample; s++) {
        orc_plan p; int rc = sample_plan(c, (uint64_t)s, iter, &p);
        if (rc) { err |= rc; continue; }
        apply_sample_at(c, &p, grad_step, pad, stride);
    }
    for (uint64_t i = 0; i < c->n; i++) memcpy(c->y + i * c->dim, pad + i * stride, sizeof(float) * c->dim);
    free(pad);
    return err;
}
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* cauchy_edge_weight (:1322-1345) + ce_compute (:1078-1114 / :1127-1163) over edges of [lo,hi) */
double orc_ce_compute(const orc_ce *c) {
    double ce = 0.;
    for (uint64_t i = c->node_lo; i < c->node_hi; i++) {
        double scale = (double)c->emb_scale[i];
        for (uint64_t e = c->indptr[i]; e < c->indptr[i + 1]; e++) {
            const float *a = c->y + i * c->dim, *o = c->y + (uint64_t)c->nbr[e] * c->dim;
            float acc = 0.f;
            for (uint64_t t = 0; t < c->dim; t++) { float df = a[t] - o[t]; acc += df * df; } /* :1326-1330 */
            double d = (double)acc / (scale * scale);   /* :1331 */
            d = pow(d, c->b);                           /* :1333 */
            double weight = 1. / (1. + d);              /* :1336 */
            float wf = (float)weight;                   /* :1337 */
            if (!(wf < 1.0f)) wf = 1.0f - 1.1920929e-07f; /* :1338-1341 F::one() - F::epsilon() */
            double we = (double)wf;
            double wij = (double)c->proba[e];
            double term = 0.;
            if (we > 0.) term += -wij * log(we);                 /* :1150-1152 */
            if (we < 1.) term += -(1. - wij) * log(1. - we);     /* :1153-1155 */
            ce += term;
        }
    }
    return ce;
}

/* a14. h_embed projection init, src/embedder.rs:245-269.  y_small[n_small*dim] -> y0[n_large*dim].
 * Noise: N(0,1) from the build's stream, element (i*dim + j) of tag PROJ. median_dist: the 0.5
 * quantile of proj_dist over nodes >= n_small (the reference uses a CKMS sketch, kgproj.rs:403-410). */
void orc_projection_init(const float *y_small, uint64_t n_small, uint64_t n_large, uint64_t dim,
                         const uint32_t *proj_node, const float *proj_dist, float median_dist, uint64_t seed,
                         float *y0) {
    for (uint64_t i = 0; i < n_small; i++) for (uint64_t j = 0; j < dim; j++) y0[i * dim + j] = y_small[i * dim + j];
    uint64_t cnt = n_large * dim;
    float *z = malloc(sizeof(float) * (cnt ? cnt : 1));
    orc_gaussian_fill(z, cnt, seed, ORC_TAG_PROJ);
    for (uint64_t i = n_small; i < n_large; i++) {
        float ratio = proj_dist[i] / median_dist;           /* :262 */
        float correction = sqrtf(ratio / (float)dim);       /* :263 */
        for (uint64_t j = 0; j < dim; j++) {
            float cc = correction * z[i * dim + j];
            if (cc > 2.0f) cc = 2.0f; else if (cc < -2.0f) cc = -2.0f; /* :265 clip(.,2) */
            y0[i * dim + j] = y_small[(uint64_t)proj_node[i] * dim + j] + cc; /* :266-267 */
        }
    }
    free(z);
}

/* get_random_init, src/embedder.rs:456-470: U(-size/2, size/2) */
void orc_random_init(float *y, uint64_t n, uint64_t dim, float size, uint64_t seed) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint64_t cnt = n * dim, nblk = (cnt + 3) / 4;
    for (uint64_t b = 0; b < nblk; b++) {
        uint32_t c[4] = {(uint32_t)b, (uint32_t)(b >> 32), ORC_TAG_RANDINIT, 0}, w[4];
        philox4x32_10(c, key, w);
        for (int t = 0; t < 4; t++)
            if (4 * b + t < cnt) y[4 * b + t] = ((float)(w[t] >> 8) * (1.0f / 16777216.0f) - 0.5f) * size;
    }
}

/* ============================================================================================ */
/* SpMM helpers for oracle.py's randomized SVD (sprs::prod semantics, row-major dense)            */
/* ============================================================================================ */
/* csr_mulacc_dense_rowmaj (svdapprox.rs:366,:390): out[m x l] += A[m x n] * rhs[n x l] */
void orc_csr_mul_dense(uint64_t m, const uint64_t *indptr, const uint32_t *ind, const float *val,
                       const float *rhs, uint64_t l, float *out) {
    for (uint64_t i = 0; i < m; i++)
        for (uint64_t e = indptr[i]; e < indptr[i + 1]; e++) {
            const float *r = rhs + (uint64_t)ind[e] * l; float a = val[e];
            for (uint64_t c = 0; c < l; c++) out[i * l + c] += a * r[c];
        }
}
/* csc_mulacc_dense_rowmaj on the transpose view (svdapprox.rs:379): out[n x l] += A^T * rhs[m x l] */
void orc_csr_t_mul_dense(uint64_t m, const uint64_t *indptr, const uint32_t *ind, const float *val,
                         const float *rhs, uint64_t l, float *out) {
    for (uint64_t i = 0; i < m; i++)
        for (uint64_t e = indptr[i]; e < indptr[i + 1]; e++) {
            float *o = out + (uint64_t)ind[e] * l; float a = val[e];
            for (uint64_t c = 0; c < l; c++) o[c] += a * rhs[i * l + c];
        }
}
