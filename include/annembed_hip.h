/*
 * annembed_hip.h -- C ABI of libannembed_hip.so, the MI355X (gfx950) implementation of
 * annembed's embedding hot path:
 *
 *     kNN-graph edge weights -> diffusion-map initialisation (randomized SVD) -> cross-entropy SGD.
 *
 * This header is what a Rust `src/embedder.rs` shim (bindgen / hand-written `extern "C"` block, see
 * INTEGRATION.md) binds.  Every entry point cites the reference interface it replaces as
 * `file:line` relative to the annembed source tree (crate v0.1.7).
 *
 * Conventions
 *  - every function returns an `int32_t` status: AE_OK (0) or one of the AE_ERR_* codes; the
 *    reference's `exit(1)` / `panic!` sites are mapped to codes, nothing aborts across the ABI.
 *    `ae_last_error_message()` returns a thread-local, human readable string for the last failure.
 *  - all arrays are caller-allocated host memory unless the parameter name starts with `d_`
 *    (device pointer).  The library never frees caller memory.  Opaque handles own device memory
 *    and are released with the matching `ae_*_destroy`.
 *  - node indices are u32 (reference: `NodeIdx = usize`, src/tools/nodeparam.rs:18), row pointers
 *    u64, distances / probabilities / coordinates f32 (the reference examples instantiate F = f32).
 *  - one process drives one HIP device through one library stream; entry points are thread-safe (a
 *    process-wide recursive lock serialises them: the reference's objects are Send + Sync and driven
 *    from one thread that fans out on rayon -- here the fan-out is the GPU).
 *  - the library is GPU-only: there is no CPU fallback.  Without a HIP device every compute entry
 *    point fails with AE_ERR_NO_DEVICE.
 */
#ifndef ANNEMBED_HIP_H
#define ANNEMBED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------- */
/* status codes                                                                                 */
/* ------------------------------------------------------------------------------------------- */
enum {
    AE_OK = 0,
    AE_ERR_INVALID_ARG = 1,
    AE_ERR_NO_DEVICE = 2,    /* no HIP device / HIP runtime error                                */
    AE_ERR_ISOLATED_NODE = 3, /* src/tools/kdumap.rs:75-85, src/diffmaps.rs:611-615,
                                 src/fromhnsw/kgraph.rs:520-537                                  */
    AE_ERR_PROBA_RANGE = 4,  /* assert at src/tools/kdumap.rs:209                                */
    AE_ERR_SVD = 5,          /* src/graphlaplace.rs:118-122                                      */
    AE_ERR_SPECTRUM = 6,     /* "svd spectrum not decreasing" src/diffmaps.rs:1176               */
    AE_ERR_EMBED = 7,        /* Embedder::embed -> Err(1), src/embedder.rs:183-191               */
    AE_ERR_STATE = 8,        /* call order violated (e.g. get_embedded before embed)             */
    AE_ERR_BETA = 9,         /* "beta cannot be > 0." src/diffmaps.rs:827-830                    */
    AE_ERR_OOM = 10
};

const char *ae_last_error_message(void);
/* A call that returned AE_OK may still have something to say (thread-local, cleared by the next outermost call): e.g.
   ae_entropy_optim_set_comm / ae_embedder_embed with exchanges_per_batch = 1 on >= 4 ranks and >= 2^20 nodes -- outside the validated
   envelope (one exchange per batch left the edges 10-21 % short at 11 M nodes in 8 shards).  Empty string: nothing. */
const char *ae_last_warning_message(void);
/* library / build identification: "annembed_hip <version> gfx950" */
const char *ae_version(void);
int32_t ae_device_count(int32_t *count);
int32_t ae_set_device(int32_t device);
/* synchronise the library's stream on the current device */
int32_t ae_synchronize(void);
/* Summation order of the global f32 sums of the STAGE-LEVEL entry points (ae_dmap_*, ae_set_data_box ...: the mean of the scales, the
   column means of set_data_box, the laplacian's normalisers).  AE_SUM_TREE (the default since round 6): two-level f64 tree reductions --
   deterministic, microseconds; the results differ from the reference's order in the last bits.  AE_SUM_REFERENCE_ORDER: the reference's
   sequential f32 order as single-lane chains -- what bit parity with the oracle needs (the parity tests ask for it), 40-100 ms each at
   11 M nodes (src/diffmaps.rs:801-822).  ae_embedder_embed and ae_entropy_optim_create do not look at this setting: they take the
   reference order exactly when the CE mode that follows is the bit-exact AE_CE_SEQUENTIAL, trees otherwise. */
enum { AE_SUM_REFERENCE_ORDER = 0, AE_SUM_TREE = 1 };
int32_t ae_set_summation_order(uint32_t order);
/* raw hipStream_t the library launches on (for hipEvent timing by the caller) */
int32_t ae_get_stream(void **stream);

/* ------------------------------------------------------------------------------------------- */
/* parameter PODs                                                                               */
/* ------------------------------------------------------------------------------------------- */

/* EmbedderParams, src/embedparams.rs:77-103 (field for field) + build extras */
typedef struct ae_embedder_params {
    uint64_t asked_dim;           /* :79  default 2  */
    uint8_t dmap_init;            /* :81  default 1.  The diffusion-map initialisation comes from a rank-20 SVD
                                     (graphlaplace.rs:97-125), i.e. at most 19 coordinates: with dmap_init and
                                     asked_dim > 19, embed() fails with AE_ERR_EMBED (the reference hard-wires a 2-D
                                     initialisation, embedder.rs:319; here it has asked_dim columns) */
    double beta;                  /* :83  default 1. */
    double b;                     /* :85  default 1. */
    double scale_rho;             /* :87  default 1. */
    double grad_step;             /* :89  default 2. */
    uint64_t nb_sampling_by_edge; /* :91  default 10 */
    uint64_t nb_grad_batch;       /* :93  default 20 */
    uint64_t grad_factor;         /* :96  default 4  */
    uint64_t hierarchy_layer;     /* :98  default 0  */
    uint8_t hubness_weighting;    /* :102 default 0  */
    /* ---- build extras (no reference counterpart: the reference draws from an unseeded
     * thread RNG, src/embedder.rs:1121,1182) ---- */
    uint64_t seed;      /* Philox key for every random draw of the embedding. default 4664397  */
    uint32_t ce_mode;   /* AE_CE_* below. default AE_CE_AUTO                                    */
    uint32_t ce_sampler; /* AE_SAMPLER_* below. default AE_SAMPLER_ROWCDF                       */
    uint32_t ce_precision; /* AE_PRECISION_* below. default AE_PRECISION_F64: the reference's arithmetic -- f32 coordinates,
                              f64 scalars (src/embedder.rs:1207-1229) -- in every faithful mode.  AE_PRECISION_F32 is an explicit
                              opt-in to f32 scalars with hardware reciprocals, honoured by AE_CE_SLICED only (about 8 % faster at the
                              configs[3] shape; AE_CE_SEQUENTIAL / AE_CE_ORDERED / AE_CE_EVENT always compute the reference's f64
                              scalars; AE_CE_HOGWILD is f32 by definition) */
} ae_embedder_params;

enum {
    /* Owner-computes ROUNDS (ce_node.hip): thread v owns y_v and replays, round by round, the samples whose source or
       target is v against the other rows as they were when the round started; per-edge sample counts are Poisson with
       the means of the reference's i.i.d. edge draw.  f32 arithmetic.  A throughput mode: it is NOT inside the
       reference's own run-to-run envelope (stale partner rows change what the stiff attraction step converges to: final
       cross entropy 0.6-1.07x the sequential loop's).  The five negatives of a sample are read from an LDS
       tile of rows the wave loads once per round: T consecutive rows from a uniform random start (uniform sampler; every
       node equally likely, as embedder.rs:1121), T alias-table draws (hubness-weighted sampler, from 2^20 nodes on).
       Shards over devices (so does the faithful AE_CE_SLICED).  asked_dim <= 32 and rows of <= 32 neighbours
       (longer rows: asked_dim in {2,3,4,8,16}); anything else fails with AE_ERR_INVALID_ARG. */
    AE_CE_HOGWILD = 0,
    /* Deterministic: the result of executing samples 0,1,2,... of the reference's `gradient_iteration`
       (src/embedder.rs:1305-1309) in order, obtained by a device-side dataflow over row versions.  Bit-exact against the
       CPU oracle; the parity mode.  For batches of up to 2^24 samples the plan of batch (nb_sample, iter + 1) is prepared on two
       internal CU-masked streams while batch (nb_sample, iter) runs (any other next call is prepared afresh: the result never
       depends on it); ae_entropy_optim_destroy waits for a prepared set nobody asked for. */
    AE_CE_SEQUENTIAL = 1,
    /* One thread per sample with racy read-modify-write of both end points, the literal transcription
       of the rayon loop.  Kept for comparison only: on a GPU with more lanes than nodes most updates are
       lost, it is NOT statistically equivalent to the reference at small N. */
    AE_CE_SAMPLE_RACY = 2,
    /* Event-ordered (ce_event.hip), the counterpart of the reference's rayon loop (src/embedder.rs:1311-1315): every
       sample is applied to the current rows of both its end points with ONE gradient (embedder.rs:1228-1239), in an
       i.i.d. random order, the reference's f64 scalar arithmetic -- a sequentially consistent execution; only the five
       negatives' rows are read unsynchronised.  Not reproducible sample by sample (neither is the reference); its
       statistics are those of the sequential loop.  asked_dim in {2,3,4,8,16}, rows of <= 32 neighbours, one device,
       at most as many nodes as the device holds resident lanes (~80 k on MI355X); otherwise AE_ERR_INVALID_ARG. */
    AE_CE_EVENT = 3,
    /* Default: the fastest mode whose output is the reference's.  AE_CE_ORDERED for batches of up to 2^25 (33.6 M) samples (2^27 before round 5's merged slices), AE_CE_SLICED
       beyond (measured cross-over: ~29 M samples per batch on an exact kNN graph with hubs -- its under-filled time slices run as one
       launch each --, ~30 M on a lattice of uniform in-degree); both statistically faithful (as the reference's own threaded loop is
       not reproducible sample by sample either) -- the bit-exact replay of the reference's sequential loop is AE_CE_SEQUENTIAL,
       by name.  How faithful, measured (32-48 seeds a side on the stiffest graph of the suite -- 60 k points in 64 blobs, k = 6, 2-D,
       40 batches; ratios to AE_CE_SEQUENTIAL's means, 2 SE = 0.7 % on the final cross entropy, 1.2 % on the median edge length;
       profiles/r06/r6_blobs_forms.txt): AE_CE_ORDERED 1.002 / 0.994, AE_CE_EVENT 1.003 / 0.995, AE_CE_SLICED with one launch per class
       0.997-1.002 / 0.992-1.006 whatever its palette -- all inside the standard error.  AE_CE_SLICED where its slices run MERGED (what
       AE_CE_AUTO runs from 2^25 samples per batch up to a few 10^8, and on the ranks of a sharded run from 4 ranks on): with the CLASS
       WINDOW of round 6 (a workgroup reads its negatives only once the class half a palette before its own is through; rows written
       through the caches) cross entropy +0.12 +- 0.22 %, median edge -0.18 +- 0.43 % against one launch per class at 256 seeds a side --
       inside the standard error too (profiles/r06/r6_blobs_window256.txt).  Without the window (rounds 4-5, and the optimistic passes
       still: +0.38 +- 0.22 % / -0.69 +- 0.41 % against the exact mode at 256 seeds, profiles/r06/r6_blobs_optwindow256.txt; the 32-seed runs
       of earlier rounds read +1 % / -2 %): cross entropy +0.34 +- 0.23 %, median edge -0.7 +- 0.4 % -- a RESOLVED bias on stiff 2-D graphs (not
       visible at 8 columns).  Its cause, isolated in round 6: the AGE of the negatives' rows.  A merged launch without the window reads
       a slice's negatives as the slice found them (half an event per node stale on average); with one launch per class they are a step
       old.  One launch per class with its negatives read from a copy of the coordinates refreshed every 1 / 4 / 16 slices reproduces
       sign and size: cross entropy 1.004 / 1.028 / 1.110, median edge 0.991 / 0.946 / 0.820.  Every asked_dim in [1, 64]: coordinate rows are stored zero-padded to 2, 3, 4, 8, 16, 32 or 64 columns (a zero
       column adds +0 to every distance and never moves).  A sharded node range (several GPUs) runs AE_CE_SLICED whatever the
       batch size (see there; refused with AE_ERR_INVALID_ARG when more than 10 % of the range's edge mass crosses shards: the
       approximate AE_CE_HOGWILD still shards, by name).  ae_entropy_optim_get_ce_mode reports the choice. */
    AE_CE_AUTO = 4,
    /* Time-sliced execution on conflict-free classes (ce_slice.hip): the batch's events (the same edge-keyed Poisson process
       as AE_CE_EVENT) are cut into thin time slices; the graph's edges are coloured once so that every class is a forest of in-stars
       (no node is the source of two edges of a class, none source of one and target of another; max row length + 5 ... + 9 classes
       whatever the in-degrees), and a step = the events of one class in one slice is one launch: every lane applies its sample exactly as
       src/embedder.rs:1207-1301 (both rows, one gradient); the events of a step that share their target run as a chain through the
       target's row (the reference: the row's lock); class order drawn afresh per slice.  The few edges without a colour and, on
       graphs of a few million edges, all of them run optimistically instead: an event that holds both its rows exclusively runs,
       the others are deferred to the next pass.  Scalar arithmetic as `ce_precision` says (default:
       the reference's f64 scalars).  Negatives: uniform / NodeSampler draws; in crowded steps the samples of a workgroup take them
       from a shared tile of rows, never two from the same run of consecutive rows.  On one device the nodes are relabelled at
       random internally: the result does not depend on the caller's numbering (a sharded range: inside every rank's range, once a
       communicator is attached).
       Two events of an edge inside a slice stay together with the probability an i.i.d. sequence gives them.  Statistical
       parity like AE_CE_EVENT (over ten seeds the means of CE and of the edge-length quantiles are the exact mode's within a
       standard error on graphs of 8 columns; on stiff 2-column graphs one launch per class and the merged slices with their class
       window are inside the standard error, the optimistic form sits at CE +0.4 %, median edge -0.7 % against the exact mode at 256
       seeds -- its negatives are a slice old: see AE_CE_AUTO), throughput-bound, any asked_dim in [1, 64], rows of <= 32 neighbours, <= 2^27 nodes; one device, or a
       sharded node range with a communicator (ae_entropy_optim_set_comm / ae_embedder_set_comm): a shard generates the events of
       the edges whose source it owns, cross-shard edges fire as two half events, other shards' rows are read as of the last
       exchange.  Where the slices run merged (AE_SLICE_MERGED_WINDOW) the launches rely on the device starting a grid's workgroups in
       index order -- a workgroup only ever waits for one started before it -- with every wait under a poll budget: a violation (seen only
       under CU masks that strand workgroups) ends the batch with AE_ERR_STATE instead of a hang.  Do not run this mode on a CU-masked
       stream. */
    AE_CE_SLICED = 5,
    /* The samples of AE_CE_SEQUENTIAL (same Philox plan, same order, same f64 arithmetic) with only their two END POINTS as
       dependencies: every attraction is applied to the rows the previous writers of i and j produced, exactly as the sequential
       loop does; the five negatives are read as the memory system has them (every sample also stores its rows in place) -- what the
       reference's threaded loop guarantees (rows under a lock for the update, negatives through try_read, embedder.rs:1257-1265).
       Not reproducible in the last bits of the negatives' contributions; statistical parity; about half the latency of
       AE_CE_SEQUENTIAL on small graphs (a C2 batch is 1 565 dependency levels deep instead of 4 305).  Any asked_dim, one device,
       < 2^31 samples per batch.  If the dataflow kernel's poll budget is ever exceeded (AE_ERR_STATE: another process holding part
       of the GPU), the coordinates hold a PARTIAL batch -- rows are stored in place -- where AE_CE_SEQUENTIAL leaves the batch's start. */
    AE_CE_ORDERED = 6
};
enum {
    AE_PRECISION_F64 = 0, /* coordinates f32, scalar coefficients f64: what the reference computes */
    AE_PRECISION_F32 = 1  /* scalar coefficients in f32 (AE_CE_SLICED only): a throughput option, narrower than the reference */
};
enum {
    /* edge ~ uniform source node x per-row inverse CDF.  Same law as the alias table because every
       row of probabilities sums to 1 (src/tools/kdumap.rs:215-218). */
    AE_SAMPLER_ROWCDF = 0,
    /* edge ~ Walker/Vose alias table over all edges, as WeightedAliasIndex at src/embedder.rs:987 */
    AE_SAMPLER_ALIAS = 1
};

/* fills *p with EmbedderParams::default(), src/embedparams.rs:107-132 */
int32_t ae_embedder_params_default(ae_embedder_params *p);

/* DiffusionParams, src/diffmaps.rs:72-87 */
typedef struct ae_diffusion_params {
    uint64_t asked_dim; /* :74 */
    float alfa;         /* :76 */
    float beta;         /* :78 */
    float epsil;        /* :80 */
    float t;            /* :82  valid when has_t        */
    uint8_t has_t;
    uint64_t gnbn;      /* :84  valid when has_gnbn     */
    uint8_t has_gnbn;
} ae_diffusion_params;

/* DiffusionParams::new(asked_dim, t_opt, g_opt): alfa .5, beta -.1, epsil 2 -- src/diffmaps.rs:95-105.
   Pass has_t = 0 / has_gnbn = 0 for None. */
int32_t ae_diffusion_params_new(ae_diffusion_params *p, uint64_t asked_dim, float t, uint8_t has_t,
                                uint64_t gnbn, uint8_t has_gnbn);
/* setters with the reference's clamps: set_alfa :122-136, set_beta :140-148, set_epsil :151-160 */
int32_t ae_diffusion_params_set_alfa(ae_diffusion_params *p, float alfa);
int32_t ae_diffusion_params_set_beta(ae_diffusion_params *p, float beta);
int32_t ae_diffusion_params_set_epsil(ae_diffusion_params *p, float epsil);

/* ------------------------------------------------------------------------------------------- */
/* a1. KGraph -- src/fromhnsw/kgraph.rs:109-120                                                 */
/* ------------------------------------------------------------------------------------------- */
typedef struct ae_kgraph ae_kgraph;

/* Build from an already flattened graph: rows sorted by increasing distance (the invariant of
   kgraph.rs:508-509), indices already dense NodeIdx.  `indptr` has n+1 entries.
   Errors: AE_ERR_ISOLATED_NODE if a row is empty (kgraph.rs:520-537), AE_ERR_INVALID_ARG if a row is
   not sorted, longer than max_nbng, holds an index >= n or the row's own node (kgraph.rs:501). */
int32_t ae_kgraph_create(const uint64_t *indptr, const uint32_t *nbr, const float *dist, uint64_t n,
                         uint32_t max_nbng, ae_kgraph **out);

/* The tail of kgraph_from_hnsw_all (kgraph.rs:496-546) on device: for each point (in iteration
   order) concatenate its neighbour lists, remap DataId -> NodeIdx in first-seen order (IndexSet
   semantics of :489,:500: the point itself first, then its neighbours in list order), sort
   ascending by distance, truncate to nbng.
   point_id[n_points]; row_ptr[n_points+1] into nbr_data_id / nbr_dist (ragged, all layers
   concatenated).  data_id_of_idx_out (optional, n_points) receives the IndexSet (NodeIdx -> DataId). */
int32_t ae_kgraph_from_ragged(const uint64_t *point_id, const uint64_t *row_ptr,
                              const uint64_t *nbr_data_id, const float *nbr_dist, uint64_t n_points,
                              uint32_t nbng, ae_kgraph **out, uint64_t *data_id_of_idx_out);

int32_t ae_kgraph_destroy(ae_kgraph *g);
int32_t ae_kgraph_get_nb_nodes(const ae_kgraph *g, uint64_t *n);   /* kgraph.rs:147 */
int32_t ae_kgraph_get_max_nbng(const ae_kgraph *g, uint32_t *k);   /* kgraph.rs:152 */
int32_t ae_kgraph_get_nb_edges(const ae_kgraph *g, uint64_t *nnz);
/* get_neighbours (kgraph.rs:157): copies the CSR back. Any pointer may be NULL. */
int32_t ae_kgraph_get_neighbours(const ae_kgraph *g, uint64_t *indptr, uint32_t *nbr, float *dist);

/* "KGraph distance batching" (north star; no reference counterpart -- the reference copies
   hnsw_rs's distances, kgraph.rs:504): recompute dist[e] = || x[i] - x[nbr[e]] ||_2 for every edge
   from a row-major n x dim f32 coordinate matrix, then re-sort each row by the new distances. */
int32_t ae_kgraph_fill_l2_distances(ae_kgraph *g, const float *x, uint64_t dim);

/* Exact brute-force kNN graph (L2) of a row-major n x dim f32 matrix: the build's stand-in for
   the hnsw_rs producer in benchmarks.  Self matches are excluded (kgraph.rs:502
   asserts index != neighbour).  Definition: F(i, j) = f32 sum, coordinates in order, of
   (x_i[t] - x_j[t])^2; row i = the nbng points with the smallest (F, j); dist = sqrtf(F).
   For nbng <= 56 the candidates come from an MFMA pass (|p|^2 - 2 <x, p>), are re-evaluated with
   the definition and certified by an error bound; uncertified rows take the plain kernel, so the
   rows are exact for any input (knn.hip).  AE_KNN_LEGACY=1 forces the plain kernel. */
int32_t ae_kgraph_bruteforce_l2(const float *x, uint64_t n, uint64_t dim, uint32_t nbng,
                                ae_kgraph **out);
/* The same graph -- the exact GLOBAL kNN rows, bit for bit -- of points that come SORTED INTO GROUPS (clusters; group g = rows
   bounds[g] .. bounds[g + 1]): the k nearest inside the own group first, then, group by group, only the points a triangle-inequality
   bound cannot exclude (|x - m_g| - max_y |y - m_g| against the point's current k-th distance) are run against the group.  What
   brute force costs n^2 costs sum of squares of the groups plus the pairs the bound lets through (configs[3]'s 11 M Higgs-shaped
   points in 64 overlapping clusters: DESIGN 7).  stats3 (may be NULL): rows recomputed by the brute-force fallback, query-point
   pairs of the second phase, of the first.  nbng <= 56. */
int32_t ae_kgraph_bruteforce_l2_grouped(const float *x, uint64_t n, uint64_t dim, uint32_t nbng, const uint64_t *bounds,
                                        uint32_t groups, ae_kgraph **out, uint64_t *stats3);

/* Hubness::new (src/fromhnsw/hubness.rs:39-76): in-degree count of every node. counts[n] */
int32_t ae_kgraph_hubness(const ae_kgraph *g, uint32_t *counts);

/* KGraphProjection accessors, src/fromhnsw/kgproj.rs:376-410.  small: graph of the n_small first
   nodes; large: graph on all nodes; proj_node/proj_dist[n_large]: for every node >= n_small the
   nearest node of the small graph and its distance (entries < n_small are ignored). */
typedef struct ae_kgraph_projection ae_kgraph_projection;
int32_t ae_kgraph_projection_create(const ae_kgraph *small, const ae_kgraph *large,
                                    const uint32_t *proj_node, const float *proj_dist,
                                    ae_kgraph_projection **out);
int32_t ae_kgraph_projection_destroy(ae_kgraph_projection *p);
/* h_embed's projection initialisation alone (src/embedder.rs:245-269): y_small [n_small x dim] -> y0 [n_large x dim], host arrays.
   n_small states the rows of y_small and must be the small graph's node count (checked: the library reads n_small x dim floats);
   dim in [1, 64]. */
int32_t ae_projection_init(const ae_kgraph_projection *proj, const float *y_small, uint64_t n_small, uint64_t dim, uint64_t seed, float *y0);

/* ------------------------------------------------------------------------------------------- */
/* a2. to_proba_edges -- src/tools/kdumap.rs:26-116, 132-235                                    */
/* ------------------------------------------------------------------------------------------- */
/* NodeParams (src/tools/nodeparam.rs:111-114) as device CSR: proba[nnz] aligned with the graph's
   nbr[], scale[n]. */
typedef struct ae_node_params ae_node_params;
int32_t ae_to_proba_edges(const ae_kgraph *g, float scale_rho, float beta, ae_node_params **out);
/* NodeParams::new(params, max_nbng) (src/tools/nodeparam.rs:117-119) from caller-provided edge
   probabilities (aligned with the graph's nbr[]) and per-node scales. */
int32_t ae_node_params_from_host(const ae_kgraph *g, const float *proba, const float *scale,
                                 ae_node_params **out);
int32_t ae_node_params_destroy(ae_node_params *np);
/* copies back; any pointer may be NULL.  proba has nnz entries, scale n entries. */
int32_t ae_node_params_get(const ae_node_params *np, float *proba, float *scale);
/* NodeParam::get_perplexity (nodeparam.rs:88-91) for every node: exp(-sum p ln p) */
int32_t ae_node_params_perplexity(const ae_node_params *np, float *perplexity);

/* ------------------------------------------------------------------------------------------- */
/* a3-a9. diffusion-map initialisation -- src/diffmaps.rs                                       */
/* ------------------------------------------------------------------------------------------- */
/* GraphLaplacian (src/graphlaplace.rs:21-35) produced by DiffusionMaps::laplacian_from_kgraph
   (diffmaps.rs:397-422 = compute_dmap_nodeparams :752-849 + kernel0_to_density :855-952 +
   compute_laplacian :427-587).  The dense / CSR switch at FULL_MAT_REPR = 5000 nodes
   (graphlaplace.rs:13) is reproduced; force_repr overrides it (0 = reference rule, 1 = dense,
   2 = CSR) so tests can exercise both branches on one graph. */
typedef struct ae_laplacian ae_laplacian;
int32_t ae_dmap_laplacian_from_kgraph(const ae_kgraph *g, const ae_diffusion_params *dp,
                                      int32_t force_repr, ae_laplacian **out);
int32_t ae_laplacian_destroy(ae_laplacian *l);
/* is_csr, nnz (CSR) or n*n (dense) */
int32_t ae_laplacian_info(const ae_laplacian *l, int32_t *is_csr, uint64_t *n, uint64_t *nnz);
/* copies back the symmetric kernel.  CSR: indptr[n+1], indices[nnz], values[nnz] (columns sorted,
   duplicates summed -- TriMat::to_csr semantics, diffmaps.rs:572-578).  Dense: values[n*n]
   row-major, indptr/indices ignored. */
int32_t ae_laplacian_get_kernel(const ae_laplacian *l, uint64_t *indptr, uint32_t *indices,
                                float *values);
/* normalizer (sqrt degrees, :565,:581), normed_scales (:815-822), q density (:949) and
   beta_scales (:842).  Any pointer may be NULL; each has n entries; mean_scale is a scalar. */
int32_t ae_laplacian_get_vectors(const ae_laplacian *l, float *normalizer, float *normed_scales,
                                 float *q_density, float *beta_scales, float *mean_scale);

/* GraphLaplacian::do_svd (graphlaplace.rs:127-134): dense & n <= 5000 -> full SVD (here: the
   leading `rank` singular triplets, converged to f32 roundoff, instead of all n); otherwise
   do_approx_svd = direct_svd(RANK(rank = 20, nbiter = 5)) (graphlaplace.rs:97-125).
   s[rank_out], u[n * rank_out] row-major.  *rank_out <= rank_cap = 20. */
int32_t ae_laplacian_do_svd(ae_laplacian *l, float *s, float *u, uint64_t *rank_out);

/* DiffusionMaps::embed_from_kgraph (diffmaps.rs:1047-1075): laplacian_from_kgraph + do_svd +
   embed_from_laplacian (:1145-1243).  y0 receives n x real_dim row-major, real_dim =
   min(asked_dim, rank-1) written to *real_dim. */
int32_t ae_dmap_embed_from_kgraph(const ae_kgraph *g, const ae_diffusion_params *dp, float *y0,
                                  uint64_t *real_dim);

/* ------------------------------------------------------------------------------------------- */
/* a7-a8. tools::svdapprox -- src/tools/svdapprox.rs                                            */
/* ------------------------------------------------------------------------------------------- */
/* MatRepr (src/tools/matrepr.rs:23-32) */
typedef struct ae_matrepr ae_matrepr;
int32_t ae_matrepr_from_csr(const uint64_t *indptr, const uint32_t *indices, const float *values,
                            uint64_t nrows, uint64_t ncols, ae_matrepr **out);
int32_t ae_matrepr_from_dense(const float *values /* row-major */, uint64_t nrows, uint64_t ncols,
                              ae_matrepr **out);
int32_t ae_matrepr_destroy(ae_matrepr *m);

/* RangeApproxMode::RANK(RangeRank{rank, nbiter}) -> subspace_iteration_{csr,full}
   (svdapprox.rs:285-333, 343-408).  q receives nrows x l row-major, l = min(nrows, ncols, rank)
   written to *l_out. */
int32_t ae_subspace_iteration(const ae_matrepr *m, uint64_t rank, uint64_t nbiter, float *q,
                              uint64_t *l_out);

/* SvdApprox::direct_svd(RANK) (svdapprox.rs:721-799): s[l], u[nrows*l] row-major,
   vt[l*ncols] row-major (u / vt may be NULL). */
int32_t ae_svd_approx_rank(const ae_matrepr *m, uint64_t rank, uint64_t nbiter, float *s, float *u,
                           float *vt, uint64_t *l_out);

/* adaptative_range_finder_matrep (svdapprox.rs:444-597, Halko-Martinsson-Tropp algorithm 4.2;
   RangeApproxMode::EPSIL of RangeApprox::get_approximator :240-247): orthonormal q[nrows x l]
   row-major, l <= min(max_rank, 4096) written to *l_out; q must hold nrows * min(max_rank, 4096) floats.
   r probe vectors; stops when the largest probe norm falls below epsil / (10 sqrt(2 pi)) times its
   initial value (:465, :515), at max_rank, or on a vanishing vector (:524-532). */
int32_t ae_adaptative_range_finder(const ae_matrepr *m, double epsil, uint64_t r, uint64_t max_rank,
                                   float *q, uint64_t *l_out);

/* SvdApprox::direct_svd(EPSIL(RangePrecision{epsil, step, max_rank})) (svdapprox.rs:721-799 with the
   range finder above; step <= 1 is reset to 2 as RangePrecision::new :167-179): s[l], u[nrows*l],
   vt[l*ncols] row-major (u / vt may be NULL, sized for l = min(max_rank, 64): the SVD panels are at
   most 64 wide, a larger max_rank is clamped). */
int32_t ae_svd_approx_epsil(const ae_matrepr *m, double epsil, uint64_t step, uint64_t max_rank,
                            float *s, float *u, float *vt, uint64_t *l_out);

/* transpose_dense_mult_csr (svdapprox.rs:116-139): b[l x ncols] = q^T * m, q is nrows x l. */
int32_t ae_transpose_dense_mult(const ae_matrepr *m, const float *q, uint64_t l, float *b);

/* ------------------------------------------------------------------------------------------- */
/* a10-a13. EntropyOptim -- src/embedder.rs:936-1315                                            */
/* ------------------------------------------------------------------------------------------- */
typedef struct ae_entropy_optim ae_entropy_optim;

/* set_data_box (embedder.rs:1376-1408) on an n x dim row-major host array, in place on device. */
int32_t ae_set_data_box(float *y, uint64_t n, uint64_t dim, float box_size);

/* EntropyOptim::new (embedder.rs:964-1025).  y0: n x asked_dim row-major initial embedding.
   hub_counts: NULL, or in-degree counts (n) when params->hubness_weighting (embedder.rs:826-833).
   Sharding (multi-GPU, no reference counterpart): this handle draws its positive edges only from
   source nodes [node_lo, node_hi) and runs `samples_share` = that fraction of every batch's
   samples; pass node_lo = 0, node_hi = n for the single-GPU path. */
int32_t ae_entropy_optim_create(const ae_kgraph *g, const ae_node_params *np,
                                const ae_embedder_params *params, const float *y0,
                                const uint32_t *hub_counts, uint64_t node_lo, uint64_t node_hi,
                                ae_entropy_optim **out);
int32_t ae_entropy_optim_destroy(ae_entropy_optim *o);
int32_t ae_entropy_optim_get_nb_edges(const ae_entropy_optim *o, uint64_t *nnz); /* :1027 */
/* ---- multi-GPU: one process per GPU, RCCL over xGMI (no reference counterpart: the reference is one shared-memory
 * process, src/embedder.rs:1311-1315; SURVEY 8b "8-GPU entry point", north star "RCCL all-gather of the low-dim coordinate
 * array ... once per CE batch").  Rank 0 obtains a 128-byte id (ae_comm_unique_id) and hands it to the other ranks by any
 * host channel; every rank calls ae_comm_init on its own device (ae_set_device first).  A communicator attached to an
 * EntropyOptim created on this rank's node range [node_lo, node_hi) -- the ranges of the ranks must tile [0, n) in rank
 * order -- makes ae_entropy_optim_gradient_iteration exchange the owned coordinate rows itself: in place, on the library's
 * stream, `exchanges_per_batch` times per batch at equal runs of rounds / time slices (1 = once per batch, at its end: the north star's
 * figure, and OUTSIDE the validated envelope from 4 ranks and ~10^6 nodes on -- the call then succeeds with a warning,
 * ae_last_warning_message; 0 = the library's choice: 4).  Two modes
 * shard: the time-sliced mode (AE_CE_SLICED; what AE_CE_AUTO resolves to on a sharded range) runs a shard's own events on current rows
 * and reads the other shards' rows -- negatives, the far ends of cross-shard edges, which fire as two half events -- as of the last
 * exchange: faithful for node orders with few cross-shard edges (connected components / locality; more than 10 % of a shard's edge mass
 * on cross-shard edges is refused with AE_ERR_INVALID_ARG).  Exchanges per batch: on graphs of up to ~10^5 nodes one is enough
 * (1 ... 240 measured: no trend); at 10^6 ... 10^7 nodes in 8 shards, from a random start, one exchange per batch leaves the edges
 * 6 ... 21 % short (the other shards' rows are a whole batch old while the layout still moves fast), 4 match the one-device run:
 * ask for 4 or more.  Attaching the communicator prepares the time-sliced mode again (edge colouring included: a
 * fraction of a second at 10^7 nodes): every rank then knows every rank's range and relabels the nodes at random inside each; the rounds mode
 * (AE_CE_HOGWILD, by name) is approximate whatever the partition.  The final cross entropy is the sum of the
 * ranks' ae_entropy_optim_ce values (ae_comm_all_reduce_sum).  RCCL is loaded on the first ae_comm_* call. */
typedef struct ae_comm ae_comm;
int32_t ae_comm_unique_id(uint8_t *id128);
int32_t ae_comm_init(int32_t rank, int32_t world, const uint8_t *id128, ae_comm **out);
/* The same communicator over a POSIX shared-memory segment instead of RCCL (ranks of ONE machine; device -> segment -> device):
 * for validation -- several ranks may share one GPU, which RCCL refuses -- and for hosts where RCCL cannot be loaded.  `name`
 * (<= 80 characters, no slash) is the same on every rank and unique per job; max_bytes >= the largest exchange, i.e. the
 * coordinate array n x row stride x 4 (row stride: ae_entropy_optim_device_coords).  Not a performance path. */
int32_t ae_comm_init_hostmem(int32_t rank, int32_t world, const char *name, uint64_t max_bytes, ae_comm **out);
int32_t ae_comm_destroy(ae_comm *c);
int32_t ae_comm_all_reduce_sum(ae_comm *c, double *value);
int32_t ae_entropy_optim_set_comm(ae_entropy_optim *o, ae_comm *c, uint32_t exchanges_per_batch);
/* bytes of coordinate rows this rank has RECEIVED through the exchanges of its batches since the handle was created (the other
   ranks' rows x row stride x 4 per exchange, in both sharding modes) -- the volume a scaling estimate needs */
int32_t ae_entropy_optim_comm_bytes(const ae_entropy_optim *o, uint64_t *bytes);
/* The sharded protocol on ONE device (validation; no reference counterpart): one batch of `world` rounds-mode handles of
 * the same graph whose node ranges tile [0, n) in order, run in lockstep -- round r of every shard, then, at the exchange
 * points, every shard's owned rows copied into the other shards' coordinate arrays.  Kernel for kernel and exchange for
 * exchange what `world` processes with a communicator attached do, so the fidelity of the sharded CE loop (shards x
 * exchanges_per_batch) can be measured against the un-sharded run where only one GPU is at hand.  nb_sample[q] = samples of
 * shard q (nb_sampling_by_edge x its own edges). */
int32_t ae_entropy_optim_gradient_iteration_lockstep(ae_entropy_optim *const *shards, uint32_t world, const uint64_t *nb_sample,
                                                     double grad_step, uint64_t iter, uint32_t exchanges_per_batch);
/* the AE_CE_* mode the handle runs (AE_CE_AUTO resolved at create) */
int32_t ae_entropy_optim_get_ce_mode(const ae_entropy_optim *o, uint32_t *ce_mode);
/* AE_CE_SLICED (no reference counterpart): how the graph's edges were scheduled -- the number of colour classes that run as
   conflict-free matchings (0 = every event runs optimistically), the share of the edge probability mass in the overflow class, the
   rounds the colouring took, and the time slices of the last batch.  AE_ERR_STATE for a handle in another mode. */
int32_t ae_entropy_optim_slice_info(const ae_entropy_optim *o, uint32_t *classes, double *overflow_fraction,
                                    uint32_t *colouring_rounds, uint32_t *slices_last_batch);
/* AE_CE_SLICED: hubs.  The classes of the colouring are forests of in-stars: the events of a step that share their TARGET run as a
   chain through the target's row, as the row's lock serialises them in the reference (embedder.rs:942,1185-1186,1239,1301).
   max_in_degree: the largest in-degree of the graph (the reference's hubness count, src/fromhnsw/hubness.rs:39-76);
   busiest_row_events_per_step: the expected length of the longest chain of a step (0 when everything runs optimistically). */
int32_t ae_entropy_optim_slice_hub_info(const ae_entropy_optim *o, uint32_t *max_in_degree, double *busiest_row_events_per_step);
/* AE_CE_SLICED: the launch form of the handle's LAST batch (0 before the first) -- it decides how old the negatives' rows are, i.e. which
   of the fidelity figures under AE_CE_AUTO applies: AE_SLICE_PER_CLASS (one launch per class and slice: negatives a step old; inside the
   exact mode's standard error), AE_SLICE_PER_CLASS_LINES (the same on node lines: a source's row, scale and neighbour ids as one request),
   AE_SLICE_MERGED_WINDOW (every class of a slice in one launch, a workgroup's negatives read once the class half a palette before its
   own is through: negatives half a slice old at most; inside the standard error at 256 seeds -- what merged slices run as),
   AE_SLICE_MERGED (the same without the window, the form of rounds 4-5, debug builds only: negatives a slice old -- the resolved
   bias on stiff 2-D graphs), AE_SLICE_OPTIMISTIC (no classes: every event through the optimistic passes; the same bias).
   AE_ERR_STATE for another mode. */
enum { AE_SLICE_NONE = 0, AE_SLICE_PER_CLASS = 1, AE_SLICE_PER_CLASS_LINES = 2, AE_SLICE_MERGED = 3, AE_SLICE_OPTIMISTIC = 4, AE_SLICE_MERGED_WINDOW = 5 };
int32_t ae_entropy_optim_slice_form(const ae_entropy_optim *o, uint32_t *form);
/* ce_compute_threaded (embedder.rs:1127-1163) over this handle's edges */
int32_t ae_entropy_optim_ce(ae_entropy_optim *o, double *ce);
/* gradient_iteration_threaded(nb_sample, grad_step) (embedder.rs:1311-1315).  `iter` keys the RNG
   stream (the reference passes nothing: its RNG is unseeded).  Asynchronous on the handle's
   stream in Hogwild mode. */
int32_t ae_entropy_optim_gradient_iteration(ae_entropy_optim *o, uint64_t nb_sample,
                                            double grad_step, uint64_t iter);
/* The nodes sample `s` of batch `iter` touches: nodes7 = {i, j, k1..k5} (positive edge :1182-1184,
   accepted negatives :1241-1253) and the edge probability w.  Deterministic given graph + seed. */
int32_t ae_entropy_optim_plan(ae_entropy_optim *o, uint64_t s_begin, uint64_t count, uint64_t iter,
                              uint32_t *nodes7, float *w);
/* Hogwild mode draws a Poisson(nb_sample) number of samples per batch: total drawn so far, and the
   number of kernel launches (rounds) the last batch was split into. */
int32_t ae_entropy_optim_samples_drawn(ae_entropy_optim *o, uint64_t *samples, uint32_t *rounds);
/* embedded scales (embedder.rs:1356-1373), n entries */
int32_t ae_entropy_optim_get_scales(const ae_entropy_optim *o, float *emb_scale);
/* current coordinates, n x asked_dim row-major */
int32_t ae_entropy_optim_get_embedded(const ae_entropy_optim *o, float *y);
/* device pointer to the coordinate array (for an all-gather by the caller): n rows of `dim` floats, where `dim` is the ROW STRIDE --
   asked_dim when that is 2, 3, 4, 8 or 16, else asked_dim zero-padded to 2 / 8 / 16 / 32 / 64 */
int32_t ae_entropy_optim_device_coords(ae_entropy_optim *o, void **d_y, uint64_t *n, uint64_t *dim);
/* average duration in ms of the SGD kernel launches since the last call (hipEvent on the handle's
   stream) and their count; resets the accumulators. */
int32_t ae_entropy_optim_kernel_time(ae_entropy_optim *o, double *avg_ms, uint64_t *launches);
/* AE_CE_SEQUENTIAL: average duration of the dataflow kernel alone (hipEvents around its launch) since the last call */
int32_t ae_entropy_optim_dataflow_time(ae_entropy_optim *o, double *avg_ms, uint64_t *launches);

/* entropy_optimize (embedder.rs:794-904) in one call: CE before, nb_grad_batch batches with
   step = grad_step * (1 - iter/nb_batch) (:875), CE after.  y: n x asked_dim out. */
int32_t ae_entropy_optimize(const ae_kgraph *g, const ae_node_params *np,
                            const ae_embedder_params *params, const float *y0, float *y,
                            double *ce_before, double *ce_after);

/* ------------------------------------------------------------------------------------------- */
/* Embedder -- src/embedder.rs:84-453                                                           */
/* ------------------------------------------------------------------------------------------- */
typedef struct ae_embedder ae_embedder;
/* Embedder::new (embedder.rs:107) -- the graph must outlive the embedder */
int32_t ae_embedder_new(const ae_kgraph *g, const ae_embedder_params *params, ae_embedder **out);
/* Embedder::from_hkgraph (embedder.rs:120) */
int32_t ae_embedder_from_hkgraph(const ae_kgraph_projection *p, const ae_embedder_params *params,
                                 ae_embedder **out);
int32_t ae_embedder_destroy(ae_embedder *e);
/* Multi-GPU embedding at the boundary the reference's callers use (no reference counterpart: the reference is one process;
   SURVEY 8b "8-GPU entry point").  Every rank builds the same graph, creates the same Embedder, attaches its communicator
   (ae_comm_init / ae_comm_init_hostmem) and calls ae_embedder_embed: the initialisation runs replicated and rank 0's initial
   embedding is broadcast (the replicas start bit-identical), rank r optimises the r-th contiguous share of the source nodes
   (both stages of a hierarchical embedding), the coordinate rows are all-gathered `exchanges_per_batch` times per CE batch
   inside the library, the reported cross entropies are sums over the ranks, and after embed() every rank holds the whole
   embedding IN THE CALLER'S NODE ORDER.  The graph may come in any node order (the reference's is file order): embed() partitions
   it by locality first (ae_kgraph_partition below, on rank 0, broadcast), runs on the relabelled graph and hands the rows back in
   the caller's order; a partition that still leaves more than 10 % of a rank's edge mass on cross-rank edges is refused on every
   rank alike (AE_ERR_INVALID_ARG).  exchanges_per_batch = 0 leaves the number to the library: 4 where (next to) nothing crosses the
   ranks, 8 below 3 % of a rank's edge mass, 16 above (cross-rank edges fire against replicas as old as the last exchange; measured
   in tools/run_part_fidelity.py).  ce_mode: AE_CE_AUTO / AE_CE_SLICED (the faithful time-sliced mode) or AE_CE_HOGWILD (the
   approximate rounds mode, by name).  A NULL or one-rank communicator changes nothing. */
int32_t ae_embedder_set_comm(ae_embedder *e, ae_comm *comm, uint32_t exchanges_per_batch);
/* The locality partition a multi-GPU embed() applies internally (partition.hip; SURVEY 8e "contiguous node ranges ... after
   locality reordering").  The reference numbers its nodes in IndexSet insertion order of the HNSW points -- file order,
   src/fromhnsw/kgraph.rs:489,500 -- so contiguous id ranges of the caller's graph would cut (world - 1) / world of the edges.
   ae_kgraph_partition: connected components (packed whole into the ranks: a graph of separated clusters is cut nowhere), a
   component that must be split is split by recursive coordinate bisection of `y` (n x dim, host; e.g. the diffusion-map
   initialisation; NULL: id order -- a partition, not a good one).  np (may be NULL) weighs the report's edges with their
   probabilities.  order[pos] = the caller's id of the node at position pos; ranges[2 r], ranges[2 r + 1] = rank r's positions.
   ae_kgraph_permuted: the same graph with the node at position p = node order[p] (rows keep their distance order): what the
   ranks create their sharded ae_entropy_optim on; coordinates come back through order (y_caller[order[p]] = y[p]). */
typedef struct ae_partition_report {
    uint64_t components;          /* connected components of the undirected graph */
    uint64_t splits;              /* components (or pieces) cut by coordinate bisection */
    double cross_mass;            /* share of the edge (probability) mass on edges whose ends lie in different ranges */
    double cross_mass_worst_rank; /* the largest share any one rank sees among the edges with an end in its range */
    double imbalance;             /* largest range / (n / world) - 1 */
} ae_partition_report;
int32_t ae_kgraph_partition(const ae_kgraph *g, const ae_node_params *np, const float *y, uint64_t dim, uint32_t world,
                            uint32_t *order, uint64_t *ranges, ae_partition_report *report);
int32_t ae_kgraph_permuted(const ae_kgraph *g, const uint32_t *order, ae_kgraph **out);
/* the report of the partition the last multi-GPU embed() applied (AE_ERR_STATE before it, or after a one-rank embed) */
int32_t ae_embedder_get_partition_report(const ae_embedder *e, ae_partition_report *report);
/* Embedder::embed (embedder.rs:183): one_step_embed (:298) or h_embed (:194). Ok(1) -> AE_OK */
int32_t ae_embedder_embed(ae_embedder *e);
int32_t ae_embedder_get_nb_nodes(const ae_embedder *e, uint64_t *n);          /* :785 */
/* get_embedded (:378) / get_embedded_reindexed (:384; data_id_of_idx may be NULL = identity) */
int32_t ae_embedder_get_embedded(const ae_embedder *e, float *y);
int32_t ae_embedder_get_embedded_reindexed(const ae_embedder *e, const uint64_t *data_id_of_idx,
                                           float *y);
int32_t ae_embedder_get_initial_embedding(const ae_embedder *e, float *y0);   /* :426 */
int32_t ae_embedder_get_hubness(const ae_embedder *e, uint32_t *counts);      /* :156 */
/* CE before / after the gradient iterations (logged by the reference at :846-886) */
int32_t ae_embedder_get_cross_entropy(const ae_embedder *e, double *before, double *after);

/* ------------------------------------------------------------------------------------------------
 * Quality estimate: Embedder::get_quality_estimate_from_edge_length,
 * src/embedder.rs:620-753 (+ get_transformed_kgraph :478-522, get_max_edge_length_embedded_kgraph
 * :527-554, KGraph::compute_max_edge src/fromhnsw/kgraph.rs:167-183).
 * The numbers the reference logs / prints at :690-731.  The radius of a node is the distance of its
 * nbng-th neighbour in the EMBEDDED space: exact here (device brute force), an hnsw_rs approximation
 * in the reference; quantiles are exact order statistics at rank floor(q*count) (CKMS eps = 0.01 in
 * the reference). `quality` is 0 as in the reference (:630, :751).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ae_quality_report {
    uint64_t nb_nodes, nb_edges;
    uint32_t kgraph_nbng;        /* "neighbourhood size used in embedding"                       */
    uint32_t nbng;               /* "neighbourhood size used in target space"                    */
    uint64_t nb_without_match;   /* neighbourhoods without a match                               */
    double mean_nbmatch;         /* mean number of neighbours conserved when match               */
    double radii_quantiles[6];   /* embedded radii at 0.05 0.25 0.5 0.75 0.85 0.95               */
    double ratio_quantiles[6];   /* embedded edge length / radius, same probabilities            */
    double median_ratio, mean_ratio;
    double quality;
} ae_quality_report;
/* y: n x dim row-major embedding in the node order of g.  ratio_by_node[n] ("continuity_ratio.csv",
   :743) and first_dist[n] ("first_dist.csv", :735) may be NULL. */
int32_t ae_quality_estimate_from_edge_length(const ae_kgraph *g, const float *y, uint32_t dim,
                                             uint32_t nbng, ae_quality_report *rep,
                                             double *ratio_by_node, double *first_dist);
/* same on an Embedder after embed(): its kgraph (or the large graph of its projection, :481-487) */
int32_t ae_embedder_get_quality_estimate_from_edge_length(const ae_embedder *e, uint32_t nbng,
                                                          ae_quality_report *rep,
                                                          double *ratio_by_node, double *first_dist);

#ifdef __cplusplus
}
#endif
#endif /* ANNEMBED_HIP_H */
