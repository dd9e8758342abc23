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
    bool has_partition = false;    // the last embed() ran sharded: the report of its (last stage's) locality partition
    ae_partition_report partition;
};

namespace {

// the multi-GPU context of one embed() call
struct Dist {
    ae_comm* comm = nullptr;
    uint32_t exchanges = 1;
    ae_partition_report* report = nullptr;   // where the stage's partition report goes
    const float* part_y = nullptr;           // coordinates the partition bisects along, if not the stage's initial embedding
    uint32_t part_dim = 0;
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
    // Multi-GPU: the graph arrives in the caller's node order -- the reference's is IndexSet insertion order of the HNSW points, file
    // order (kgraph.rs:489,500): no locality -- and the sharded loop wants contiguous ranges with few cross-range edges.  Rank 0
    // partitions (connected components packed whole, a component that must be cut is cut by coordinate bisection of the initial
    // embedding: partition.hip), the order is broadcast, every rank relabels its copy of the graph, the node parameters, the initial
    // embedding and the hubness counts, runs on the relabelled problem and hands the rows back in the caller's order.
    Partition part;
    std::unique_ptr<ae_kgraph> g_part;
    ae_node_params np_part;
    DevBuf<float> y0_part;
    const ae_kgraph* gr = g;
    const ae_node_params* npr = np;
    float* y0r = d_y0;
    const uint64_t dim = params.asked_dim;
    if (dist.active()) {
        if (params.ce_mode != AE_CE_HOGWILD && params.ce_mode != AE_CE_AUTO && params.ce_mode != AE_CE_SLICED)
            fail(AE_ERR_INVALID_ARG, "a multi-GPU embedding runs the time-sliced mode (ce_mode = AE_CE_AUTO / AE_CE_SLICED: faithful) or the approximate rounds mode "
                                     "(AE_CE_HOGWILD, by name); the other modes need the whole graph on one device");
        const uint32_t world = (uint32_t)comm_world(dist.comm), rank = (uint32_t)comm_rank(dist.comm);
        if (g->n < (uint64_t)world * ((uint64_t)g->max_nbng + 8)) fail(AE_ERR_INVALID_ARG, "graph too small for %u ranks", world);
        comm_broadcast_f32(dist.comm, d_y0, g->n * dim, 0);
        // rank 0's partition on every rank: order + ranges + report
        int32_t code = AE_OK;
        std::string msg;
        DevBuf<uint32_t> head(2 * (size_t)world * 2 + 16);   // ranges (as u32 pairs lo/hi words), status, report
        std::vector<uint32_t> hhead(head.n, 0u);
        if (rank == 0) {
            try {
                if (dist.part_y) partition_nodes_device(g, np->proba.p, dist.part_y, dist.part_dim, dist.part_dim, world, part);
                else partition_nodes_device(g, np->proba.p, d_y0, (uint32_t)dim, (uint32_t)dim, world, part);
            } catch (const Error& e) {
                code = e.code;
                msg = e.msg;
            }
            if (code == AE_OK) {
                for (uint32_t x = 0; x < 2 * world; x++) { hhead[2 * x] = (uint32_t)part.ranges[x]; hhead[2 * x + 1] = (uint32_t)(part.ranges[x] >> 32); }
                double rep[3] = {part.cross_mass, part.cross_mass_worst_rank, part.imbalance};
                memcpy(&hhead[4 * world + 2], rep, sizeof(rep));
                hhead[4 * world + 8] = (uint32_t)part.components; hhead[4 * world + 9] = (uint32_t)(part.components >> 32);
                hhead[4 * world + 10] = (uint32_t)part.splits;
            }
            hhead[4 * world] = (uint32_t)code;
        }
        head.upload(hhead.data(), hhead.size());
        sync();
        comm_broadcast_u32(dist.comm, head.p, head.n, 0);
        hhead = head.to_host();
        if (hhead[4 * world] != AE_OK) {
            if (rank == 0) fail(code, "%s", msg.c_str());
            fail((int32_t)hhead[4 * world], "rank 0 could not partition the graph (its own message says why)");
        }
        if (rank != 0) {
            part.ranges.resize(2 * (size_t)world);
            for (uint32_t x = 0; x < 2 * world; x++) part.ranges[x] = (uint64_t)hhead[2 * x] | ((uint64_t)hhead[2 * x + 1] << 32);
            double rep[3];
            memcpy(rep, &hhead[4 * world + 2], sizeof(rep));
            part.cross_mass = rep[0]; part.cross_mass_worst_rank = rep[1]; part.imbalance = rep[2];
            part.components = (uint64_t)hhead[4 * world + 8] | ((uint64_t)hhead[4 * world + 9] << 32);
            part.splits = hhead[4 * world + 10];
            part.order.alloc(g->n);
            part.perm.alloc(g->n);
        }
        comm_broadcast_u32(dist.comm, part.order.p, g->n, 0);
        comm_broadcast_u32(dist.comm, part.perm.p, g->n, 0);
        if (dist.report) {
            dist.report->components = part.components;
            dist.report->splits = part.splits;
            dist.report->cross_mass = part.cross_mass;
            dist.report->cross_mass_worst_rank = part.cross_mass_worst_rank;
            dist.report->imbalance = part.imbalance;
        }
        // every rank takes the same decision from the same numbers (the time-sliced mode; the rounds mode is approximate anyway)
        if (params.ce_mode != AE_CE_HOGWILD && part.cross_mass_worst_rank > ce_slice_max_cross_mass() && !debug_knob("AE_SL_ANY_PARTITION"))
            fail(AE_ERR_INVALID_ARG, "multi-GPU embedding over %u ranks: after the locality partition %.1f %% of a rank's edge probability mass still lies on cross-rank edges "
                                     "(limit %.0f %%; %.1f %% over all edges): this graph does not shard in the faithful mode -- run it on one device, or ask for the approximate "
                                     "rounds mode (AE_CE_HOGWILD)", world, 100. * part.cross_mass_worst_rank, 100. * ce_slice_max_cross_mass(), 100. * part.cross_mass);
        // the relabelled problem
        np_part.proba.release();
        g_part.reset(kgraph_permuted_device(g, part.order.p, part.perm.p, np->proba.p, &np_part.proba));
        np_part.g = g_part.get();
        np_part.scale.alloc(g->n);
        permute_rows_device(np->scale.p, np_part.scale.p, g->n, 1, part.order.p, false);
        y0_part.alloc(g->n * dim);
        permute_rows_device(d_y0, y0_part.p, g->n, (uint32_t)dim, part.order.p, false);
        if (!hub.empty()) {
            const std::vector<uint32_t> ho = part.order.to_host();
            std::vector<uint32_t> h2(g->n);
            for (uint64_t p = 0; p < g->n; p++) h2[p] = hub[ho[p]];
            out.hubness = hub;   // (reported in the caller's order)
            hub.swap(h2);
        }
        sync();
        gr = g_part.get();
        npr = &np_part;
        y0r = y0_part.p;
        lo = part.ranges[2 * rank];
        hi = part.ranges[2 * rank + 1];
    }
    ae_entropy_optim* o = entropy_optim_create_impl(gr, npr, &params, y0r, true, hub.empty() ? nullptr : hub.data(), lo, hi);
    try {
        // Exchanges per batch when the caller leaves the choice to the library: a cross-rank edge fires as two half events, each against a
        // replica of the far end that is as old as the last exchange, so the more mass crosses the fresher the replicas have to be.
        // Measured (tools/run_part_fidelity.py: 20 000 points uniform in a square, two ranks, 4 % of the mass crossing, three runs each,
        // edge-length quartiles against the exact mode's four-seed mean): 1 exchange per batch +3 ... +5 %, 4: +2 ... +2.6 %, 16: -1 ... +1.4 %
        // (the exact mode's own scatter); graphs cut nowhere (component partitions) hold at 4 up to 11 M nodes.
        uint32_t exchanges = dist.exchanges;
        if (dist.active() && !exchanges) exchanges = part.cross_mass_worst_rank < 0.005 ? 4u : (part.cross_mass_worst_rank < 0.03 ? 8u : 16u);
        if (dist.active()) entropy_optim_attach_comm(o, dist.comm, exchanges);
        double ce = 0.;
        rc_check(ae_entropy_optim_ce(o, &ce));  // :846
        out.ce_before = comm_all_reduce_sum(dist.comm, ce);
        uint64_t nnz = 0;
        rc_check(ae_entropy_optim_get_nb_edges(o, &nnz));
        const uint64_t nb_sample = params.nb_sampling_by_edge * nnz;  // :858
        for (uint64_t iter = 1; iter <= params.nb_grad_batch; iter++) {  // :873
            const double step = params.grad_step * (1. - (double)iter / (double)params.nb_grad_batch);  // :875
            const int32_t rc = ae_entropy_optim_gradient_iteration(o, nb_sample, step, iter);
            // a batch that failed on one rank (it still joined the batch's exchanges: ce_slice.hip) ends the loop on every rank
            if (dist.active() && comm_all_reduce_sum(dist.comm, rc == AE_OK ? 0. : 1.) > 0. && rc == AE_OK)
                fail(AE_ERR_STATE, "a CE batch failed on another rank (its own message says why)");
            rc_check(rc);
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
        if (dist.active()) {   // back to the caller's node order
            DevBuf<float> back(g->n * dim);
            permute_rows_device(out.y.p, back.p, g->n, (uint32_t)dim, part.order.p, true);
            sync();
            out.y = std::move(back);
        }
        sync();
        if (!dist.active()) out.hubness = std::move(hub);
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
    // A random start says nothing about the graph: a multi-GPU run then bisects along diffusion-map coordinates computed for the
    // partition alone (rank 0 is the one that partitions; a graph the diffusion map refuses -- e.g. disconnected: a degenerate
    // spectrum -- is packed by its components and, where one must be cut, cut along the random start: refused if that crosses too much)
    DevBuf<float> part_y;
    Dist stage = dist;
    if (dist.active() && !params.dmap_init && comm_rank(dist.comm) == 0) {
        try {
            ae_diffusion_params dp;
            memset(&dp, 0, sizeof(dp));
            const uint64_t pd = std::min<uint64_t>(std::max<uint64_t>(dim, 2), 8);
            dp.asked_dim = pd; dp.alfa = 0.5f; dp.beta = -0.1f; dp.epsil = 2.0f; dp.t = 5.0f; dp.has_t = 1; dp.gnbn = 12; dp.has_gnbn = 1;
            ae_laplacian lap;
            dmap_laplacian_device(g, &dp, 0, &lap);
            if (embed_from_laplacian_device(&lap, pd, dp.t, true, part_y, nullptr) == pd) { stage.part_y = part_y.p; stage.part_dim = (uint32_t)pd; }
        } catch (const Error&) {
            stage.part_y = nullptr;
        }
    }
    entropy_optimize_device(g, &np, params, y0.p, out, stage);  // :356
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
        e->comm_exchanges = exchanges_per_batch;   // (0: the library's choice, by the partition's cross-rank mass: entropy_optimize_device)
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
        dist.report = &e->partition;
        e->has_partition = dist.active();
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
int32_t ae_embedder_get_partition_report(const ae_embedder* e, ae_partition_report* report) {
    return guard([&] {
        if (!e || !report) fail(AE_ERR_INVALID_ARG, "null argument");
        if (!e->done || !e->has_partition) fail(AE_ERR_STATE, "no partition: get_partition_report is for a multi-GPU embed()");
        *report = e->partition;
    });
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
