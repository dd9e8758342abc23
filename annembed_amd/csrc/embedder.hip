// embedder.hip -- Embedder (src/embedder.rs:84-453): the driver above the stage functions.
//
//   Embedder::embed           :183   dispatch
//   Embedder::one_step_embed  :298   dmap init (or random) -> to_proba_edges -> entropy_optimize
//   Embedder::h_embed         :194   small graph first, projection init, then the large graph
//   entropy_optimize          :794   hubness sampler, CE before, batches, CE after
//
// Build decisions on reference defects: B1 the dmap init uses asked_dim (the
// reference hard-wires 2, :319); B2 random init does not depend on initial_space (:348 vs :459);
// B3 the last batch runs with step 0 like the reference (:875).
#include "internal.h"
#include "linalg.h"
#include "philox.h"

using namespace ae;

extern "C" {
int32_t ae_entropy_optim_destroy(ae_entropy_optim* o);
int32_t ae_entropy_optim_ce(ae_entropy_optim* o, double* ce);
int32_t ae_entropy_optim_gradient_iteration(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint64_t iter);
int32_t ae_entropy_optim_get_nb_edges(const ae_entropy_optim* o, uint64_t* nnz);
int32_t ae_entropy_optim_get_embedded(const ae_entropy_optim* o, float* y);
int32_t ae_entropy_optim_kernel_time(ae_entropy_optim* o, double* avg_ms, uint64_t* launches);
int32_t ae_kgraph_hubness(const ae_kgraph* g, uint32_t* counts);
int32_t ae_entropy_optim_device_coords(ae_entropy_optim* o, void** d_y, uint64_t* n, uint64_t* dim);
}

namespace {

// get_random_init, embedder.rs:456-470: U(-size/2, size/2)
__global__ void random_init_kernel(float* __restrict__ y, uint64_t count, float size, uint64_t seed) {
    uint64_t nblk = (count + 3) / 4;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < nblk; b += stride) {
        uint32_t w[4];
        philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), kTagRandInit, 0, (uint32_t)seed, (uint32_t)(seed >> 32), w);
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (4 * b + t < count) y[4 * b + t] = ((float)(w[t] >> 8) * (1.0f / 16777216.0f) - 0.5f) * size;
    }
}

// h_embed projection init, embedder.rs:245-269
__global__ void projection_init_kernel(const float* __restrict__ y_small, uint64_t n_small, uint64_t n_large, uint32_t dim,
                                       const uint32_t* __restrict__ proj_node, const float* __restrict__ proj_dist, float median_dist,
                                       uint64_t seed, float* __restrict__ y0) {
    uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (idx >= n_large * dim) return;
    const uint64_t i = idx / dim;
    const uint32_t j = (uint32_t)(idx % dim);
    if (i < n_small) { y0[idx] = y_small[idx]; return; }  // :250-254
    const uint64_t b = idx >> 2;
    uint32_t w[4];
    philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), kTagProj, 0, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    float z[4];
    box_muller(w[0], w[1], z[0], z[1]);
    box_muller(w[2], w[3], z[2], z[3]);
    const int lane = (int)(idx & 3);
    const float zn = lane == 0 ? z[0] : lane == 1 ? z[1] : lane == 2 ? z[2] : z[3];
    const float ratio = proj_dist[i] / median_dist;     // :262
    const float correction = sqrtf(ratio / (float)dim);  // :263
    float cc = correction * zn;
    cc = cc > 2.0f ? 2.0f : (cc < -2.0f ? -2.0f : cc);  // :265 clip(., 2)
    y0[idx] = y_small[(uint64_t)proj_node[i] * dim + j] + cc;  // :266-267
}

}  // namespace

struct ae_embedder {
    const ae_kgraph* g = nullptr;
    const ae_kgraph_projection* proj = nullptr;
    ae_embedder_params params;
    bool done = false;
    uint64_t n = 0;
    std::vector<float> embedding, initial_embedding;
    std::vector<uint32_t> hubness;
    double ce_before = 0., ce_after = 0.;
    ae_comm* comm = nullptr;       // multi-GPU: this process is one rank of the embedding (ae_embedder_set_comm)
    uint32_t comm_exchanges = 1;
};

namespace {

// the multi-GPU context of one embed() call
struct Dist {
    ae_comm* comm = nullptr;
    uint32_t exchanges = 1;
    bool active() const { return comm_world(comm) > 1; }
};

struct StageResult {
    DevBuf<float> y;  // n x dim final embedding (device)
    double ce_before = 0., ce_after = 0.;
    std::vector<uint32_t> hubness;
};

void rc_check(int32_t rc) {
    if (rc != AE_OK) throw Error(rc, ae_last_error_message());
}

// entropy_optimize, embedder.rs:794-904, from a device-resident initial embedding.  Multi-GPU (no reference counterpart): the
// initial embedding is rank 0's on every rank (broadcast: the replicas start bit-identical), this rank optimises its contiguous
// share of the source nodes, the coordinate rows are all-gathered inside ae_entropy_optim_gradient_iteration, the cross entropies
// are sums over the ranks; after the last batch every rank holds the whole embedding.
void entropy_optimize_device(const ae_kgraph* g, const ae_node_params* np, const ae_embedder_params& params, float* d_y0,
                             StageResult& out, const Dist& dist) {
    std::vector<uint32_t> hub;
    if (params.hubness_weighting) {  // :810-834
        hub.resize(g->n);
        rc_check(ae_kgraph_hubness(g, hub.data()));
    }
    uint64_t lo = 0, hi = g->n;
    if (dist.active()) {
        if (params.ce_mode != AE_CE_HOGWILD && params.ce_mode != AE_CE_AUTO && params.ce_mode != AE_CE_SLICED)
            fail(AE_ERR_INVALID_ARG, "a multi-GPU embedding runs the time-sliced mode (ce_mode = AE_CE_AUTO / AE_CE_SLICED: faithful, for node orders with few cross-shard "
                                     "edges) or the approximate rounds mode (AE_CE_HOGWILD, by name); the other modes need the whole graph on one device");
        const uint64_t world = (uint64_t)comm_world(dist.comm), rank = (uint64_t)comm_rank(dist.comm);
        if (g->n < world * ((uint64_t)g->max_nbng + 8)) fail(AE_ERR_INVALID_ARG, "graph too small for %llu ranks", (unsigned long long)world);
        const uint64_t base = g->n / world, rem = g->n % world;  // contiguous ranges, the remainder spread over the first ranks
        lo = rank * base + std::min(rank, rem);
        hi = lo + base + (rank < rem ? 1 : 0);
        comm_broadcast_f32(dist.comm, d_y0, g->n * params.asked_dim, 0);
    }
    ae_entropy_optim* o = entropy_optim_create_impl(g, np, &params, d_y0, true, hub.empty() ? nullptr : hub.data(), lo, hi);
    try {
        if (dist.active()) entropy_optim_attach_comm(o, dist.comm, dist.exchanges);
        double ce = 0.;
        rc_check(ae_entropy_optim_ce(o, &ce));  // :846
        out.ce_before = comm_all_reduce_sum(dist.comm, ce);
        uint64_t nnz = 0;
        rc_check(ae_entropy_optim_get_nb_edges(o, &nnz));
        const uint64_t nb_sample = params.nb_sampling_by_edge * nnz;  // :858
        for (uint64_t iter = 1; iter <= params.nb_grad_batch; iter++) {  // :873
            const double step = params.grad_step * (1. - (double)iter / (double)params.nb_grad_batch);  // :875
            rc_check(ae_entropy_optim_gradient_iteration(o, nb_sample, step, iter));
        }
        double ms;
        uint64_t cnt;
        rc_check(ae_entropy_optim_kernel_time(o, &ms, &cnt));
        rc_check(ae_entropy_optim_ce(o, &ce));  // :885
        out.ce_after = comm_all_reduce_sum(dist.comm, ce);
        void* dy = nullptr;
        uint64_t stride = 0;  // rows are stored zero-padded to a stride of 2, 3, 4, 8, 16, 32 or 64 floats
        rc_check(ae_entropy_optim_device_coords(o, &dy, nullptr, &stride));
        out.y.alloc(g->n * params.asked_dim);
        AE_HIP(hipMemcpy2DAsync(out.y.p, sizeof(float) * params.asked_dim, dy, sizeof(float) * stride, sizeof(float) * params.asked_dim, g->n,
                                hipMemcpyDeviceToDevice, stream()));
        sync();
        out.hubness = std::move(hub);
    } catch (...) {
        ae_entropy_optim_destroy(o);
        throw;
    }
    ae_entropy_optim_destroy(o);
}

// one_step_embed, embedder.rs:298-371.  Returns the device embedding; initial embedding optionally copied out.
void one_step_embed_device(const ae_kgraph* g, const ae_embedder_params& params, StageResult& out, std::vector<float>* initial_out, const Dist& dist) {
    const uint64_t n = g->n, dim = params.asked_dim;
    DevBuf<float> y0;
    // the single-lane reference-order sums of the initialisation only where the CE loop after them is the bit-exact mode
    TreeSums sums(dist.active() || resolve_ce_mode(params.ce_mode, dim, false, params.nb_sampling_by_edge * g->nnz, g->max_nbng, g->nnz) != AE_CE_SEQUENTIAL);
    if (params.dmap_init) {  // :308-345
        ae_diffusion_params dp;
        memset(&dp, 0, sizeof(dp));
        dp.asked_dim = dim;  // B1 (reference: 2, :319)
        dp.alfa = 0.5f;      // :320
        dp.beta = -0.1f;     // :321
        dp.epsil = 2.0f;     // DiffusionParams::new default, diffmaps.rs:100
        dp.t = 5.0f;         // :317
        dp.has_t = 1;
        dp.gnbn = 12;        // :318
        dp.has_gnbn = 1;
        ae_laplacian lap;
        dmap_laplacian_device(g, &dp, 0, &lap);
        const uint32_t rd = embed_from_laplacian_device(&lap, dim, dp.t, true, y0, nullptr);
        if (rd != dim) fail(AE_ERR_EMBED, "dmap initialisation provides %u dimensions, asked %llu (rank 20 svd)", rd, (unsigned long long)dim);
        set_data_box_device(y0.p, n, dim, 10.f);  // :345
    } else {
        y0.alloc(n * dim);  // :348, B2
        hipLaunchKernelGGL(random_init_kernel, dim3(grid_cap((n * dim + 3) / 4, 256)), dim3(256), 0, stream(), y0.p, n * dim, 1.0f, params.seed);
        check_launch("random_init");
    }
    ae_node_params np;
    to_proba_edges_device(g, (float)params.scale_rho, (float)params.beta, &np);  // :351-355
    if (dist.active()) comm_broadcast_f32(dist.comm, y0.p, n * dim, 0);  // (before it is reported: get_initial_embedding is rank 0's everywhere)
    if (initial_out) { initial_out->resize(n * dim); y0.download(initial_out->data(), n * dim); }
    entropy_optimize_device(g, &np, params, y0.p, out, dist);  // :356
}

// h_embed, embedder.rs:194-295
void h_embed_device(const ae_kgraph_projection* proj, const ae_embedder_params& params, StageResult& out, std::vector<float>* initial_out, const Dist& dist) {
    ae_embedder_params first = params;
    first.nb_grad_batch = params.grad_factor * params.nb_grad_batch;  // :204-205
    first.grad_step = 1.;                                             // :207
    first.hierarchy_layer = 0;                                        // :208
    StageResult res1;
    one_step_embed_device(proj->small_graph, first, res1, nullptr, dist);  // :213
    const ae_kgraph* large = proj->large_graph;
    ae_node_params np;
    to_proba_edges_device(large, (float)params.scale_rho, (float)params.beta, &np);  // :226-230
    const uint64_t n_small = proj->small_graph->n, n_large = large->n, dim = params.asked_dim;
    DevBuf<float> y0(n_large * dim);
    hipLaunchKernelGGL(projection_init_kernel, dim3(blocks_for(n_large * dim, 256)), dim3(256), 0, stream(), res1.y.p, n_small, n_large,
                       (uint32_t)dim, proj->proj_node.p, proj->proj_dist.p, proj->median_dist, params.seed, y0.p);  // :245-269
    check_launch("projection_init");
    if (initial_out) { initial_out->resize(n_large * dim); y0.download(initial_out->data(), n_large * dim); }
    entropy_optimize_device(large, &np, params, y0.p, out, dist);  // :275
}

}  // namespace

extern "C" {

// stage-level entry of the projection initialisation (embedder.rs:245-269): the first n_small rows are y_small, every other
// node starts at its projection's row plus clip(N(0,1) sqrt(proj_dist / median / dim), 2).  Host arrays in, host array out.
int32_t ae_projection_init(const ae_kgraph_projection* proj, const float* y_small, uint64_t n_small_rows, uint64_t dim, uint64_t seed, float* y0) {
    return guard([&] {
        require_device();
        if (!proj || !y_small || !y0 || dim == 0 || dim > 64) fail(AE_ERR_INVALID_ARG, "bad argument (dim must be in [1, 64])");
        const uint64_t n_small = proj->small_graph->n, n_large = proj->large_graph->n;
        if (n_small_rows != n_small) fail(AE_ERR_INVALID_ARG, "y_small has %llu rows, the small graph %llu nodes", (unsigned long long)n_small_rows, (unsigned long long)n_small);
        DevBuf<float> ys(n_small * dim), out(n_large * dim);
        ys.upload(y_small, n_small * dim);
        hipLaunchKernelGGL(projection_init_kernel, dim3(blocks_for(n_large * dim, 256)), dim3(256), 0, stream(), (const float*)ys.p, n_small,
                           n_large, (uint32_t)dim, proj->proj_node.p, proj->proj_dist.p, proj->median_dist, seed, out.p);
        check_launch("projection_init");
        out.download(y0, n_large * dim);
    });
}

int32_t ae_embedder_new(const ae_kgraph* g, const ae_embedder_params* params, ae_embedder** out) {
    return guard([&] {
        if (!g || !params || !out) fail(AE_ERR_INVALID_ARG, "null argument");
        std::unique_ptr<ae_embedder> e(new ae_embedder);
        e->g = g;
        e->params = *params;
        e->n = g->n;
        *out = e.release();
    });
}
int32_t ae_embedder_from_hkgraph(const ae_kgraph_projection* p, const ae_embedder_params* params, ae_embedder** out) {
    return guard([&] {
        if (!p || !params || !out) fail(AE_ERR_INVALID_ARG, "null argument");
        std::unique_ptr<ae_embedder> e(new ae_embedder);
        e->proj = p;
        e->params = *params;
        e->n = p->large_graph->n;
        *out = e.release();
    });
}
int32_t ae_embedder_destroy(ae_embedder* e) {
    return guard([&] { delete e; });
}

// Multi-GPU embedding (no reference counterpart; SURVEY 8b "8-GPU entry point"): every rank of the communicator builds the same
// graph, creates the same Embedder and calls embed(); see include/annembed_hip.h
int32_t ae_embedder_set_comm(ae_embedder* e, ae_comm* comm, uint32_t exchanges_per_batch) {
    return guard([&] {
        if (!e) fail(AE_ERR_INVALID_ARG, "null argument");
        e->comm = comm;
        e->comm_exchanges = exchanges_per_batch ? exchanges_per_batch : 4u;   // (0: the library's choice)
    });
}

int32_t ae_embedder_embed(ae_embedder* e) {
    int32_t rc = guard([&] {
        require_device();
        if (!e) fail(AE_ERR_INVALID_ARG, "null argument");
        if (e->params.asked_dim == 0 || e->params.asked_dim > 64) fail(AE_ERR_INVALID_ARG, "asked_dim must be in [1,64]");
        StageResult res;
        Dist dist;
        dist.comm = e->comm;
        dist.exchanges = e->comm_exchanges;
        if (e->g) one_step_embed_device(e->g, e->params, res, &e->initial_embedding, dist);  // :184-186
        else h_embed_device(e->proj, e->params, res, &e->initial_embedding, dist);           // :187-190
        e->embedding.resize(e->n * e->params.asked_dim);
        res.y.download(e->embedding.data(), e->embedding.size());
        e->hubness = std::move(res.hubness);
        e->ce_before = res.ce_before;
        e->ce_after = res.ce_after;
        e->done = true;
    });
    return rc;
}

int32_t ae_embedder_get_nb_nodes(const ae_embedder* e, uint64_t* n) {
    return guard([&] {
        if (!e || !n) fail(AE_ERR_INVALID_ARG, "null argument");
        *n = e->n;
    });
}
int32_t ae_embedder_get_embedded(const ae_embedder* e, float* y) {
    return guard([&] {
        if (!e || !y) fail(AE_ERR_INVALID_ARG, "null argument");
        if (!e->done) fail(AE_ERR_STATE, "get_embedded called before embed()");
        memcpy(y, e->embedding.data(), sizeof(float) * e->embedding.size());
    });
}
// get_embedded_reindexed, embedder.rs:384-405: row i goes to row DataId(i)
int32_t ae_embedder_get_embedded_reindexed(const ae_embedder* e, const uint64_t* data_id_of_idx, float* y) {
    return guard([&] {
        if (!e || !y) fail(AE_ERR_INVALID_ARG, "null argument");
        if (!e->done) fail(AE_ERR_STATE, "get_embedded_reindexed called before embed()");
        const uint64_t dim = e->params.asked_dim;
        for (uint64_t i = 0; i < e->n; i++) {
            const uint64_t origin = data_id_of_idx ? data_id_of_idx[i] : i;
            if (origin >= e->n) fail(AE_ERR_INVALID_ARG, "DataIds must be contiguous in 0..n for reindexing");
            memcpy(y + origin * dim, e->embedding.data() + i * dim, sizeof(float) * dim);
        }
    });
}
int32_t ae_embedder_get_initial_embedding(const ae_embedder* e, float* y0) {
    return guard([&] {
        if (!e || !y0) fail(AE_ERR_INVALID_ARG, "null argument");
        if (!e->done) fail(AE_ERR_STATE, "get_initial_embedding called before embed()");
        memcpy(y0, e->initial_embedding.data(), sizeof(float) * e->initial_embedding.size());
    });
}
int32_t ae_embedder_get_hubness(const ae_embedder* e, uint32_t* counts) {
    return guard([&] {
        if (!e || !counts) fail(AE_ERR_INVALID_ARG, "null argument");
        if (!e->done || e->hubness.empty()) fail(AE_ERR_STATE, "hubness is only available after embed() with hubness_weighting");
        memcpy(counts, e->hubness.data(), sizeof(uint32_t) * e->hubness.size());
    });
}
int32_t ae_embedder_get_quality_estimate_from_edge_length(const ae_embedder* e, uint32_t nbng, ae_quality_report* rep,
                                                          double* ratio_by_node, double* first_dist) {
    if (!e || !e->done) {  // "cannot ask for embedded quality before embedding", embedder.rs:633-636
        set_last_error("get_quality_estimate_from_edge_length called before embed()");
        return e ? AE_ERR_STATE : AE_ERR_INVALID_ARG;
    }
    const ae_kgraph* g = e->g ? e->g : e->proj->large_graph;  // :481-487
    return ae_quality_estimate_from_edge_length(g, e->embedding.data(), (uint32_t)(e->embedding.size() / e->n), nbng, rep, ratio_by_node,
                                                first_dist);
}
int32_t ae_embedder_get_cross_entropy(const ae_embedder* e, double* before, double* after) {
    return guard([&] {
        if (!e) fail(AE_ERR_INVALID_ARG, "null argument");
        if (!e->done) fail(AE_ERR_STATE, "get_cross_entropy called before embed()");
        if (before) *before = e->ce_before;
        if (after) *after = e->ce_after;
    });
}

}  // extern "C"
