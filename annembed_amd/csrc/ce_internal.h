// ce_internal.h -- shared between ce.hip (EntropyOptim handle, sequential + per-sample kernels) and
// ce_node.hip (node-centric owner-computes Hogwild kernel).
#pragma once
#include "internal.h"

namespace ae {

struct CeDev {
    uint64_t n, nnz;
    uint32_t dim, uniform_k;
    const uint64_t* indptr;
    const uint32_t* nbr;
    const float* proba;
    const float* emb_scale;
    float* y;
    double b;
    uint64_t seed;
    uint32_t sampler;
    uint64_t node_lo, node_hi, edge_lo, shard_edges;
    const float* edge_odds;
    const uint32_t* edge_alias;
    const uint32_t* edge_src;
    const float* hub_odds;       // non-null = hubness-weighted negative sampling (NodeSampler, embedder.rs:915-930)
    const uint32_t* hub_alias;
    const uint2* hub_tab;        // the same alias table, one 8-byte entry per node {odds bits, alias}: one random access per draw
    const float* yneg;           // (experiment, AE_SL_NEG_SNAPSHOT: tools/run_blobs_forms.py) where the time-sliced mode reads its NEGATIVES' rows if not in y: a snapshot of y some slices old
    uint32_t ystride;            // floats from one node's row of `y` to the next: dim, or more where the time-sliced mode keeps a node's dependency words behind its row (ce_slice_kernels.h)
};

// one in-edge (u -> v) of the transposed graph: everything thread v needs to replay the sample's
// effect on y_v without touching the forward rows of u
struct InEdge {
    uint32_t src;   // u
    uint32_t eid;   // index of the edge in the forward CSR (keys the per-edge sample count)
    float w;        // proba[eid]
    float s_src;    // embedded scale of u
};

}  // namespace ae

using namespace ae;  // internal header: the handle below is declared at global scope for the C ABI

// Coordinate rows are stored with a row stride of pad_dim(asked_dim) floats, the columns beyond asked_dim exactly zero: a zero
// column adds +0 to every squared distance and never moves (its gradient component is (0 - 0) * c), so every kernel is
// instantiated for the strides below only and still computes the reference's arithmetic for ANY asked_dim in [1, 64]
// (the reference is generic in the dimension and publishes 15-D runs, src/embedder.rs:604-618).
inline uint32_t ae_pad_dim(uint64_t d) {
    if (d <= 2) return 2;
    if (d <= 4) return (uint32_t)d;
    if (d <= 8) return 8;
    if (d <= 16) return 16;
    if (d <= 32) return 32;
    return 64;
}
#define AE_DISPATCH_DIM(dim, FN, ...)          \
    switch (dim) {                             \
        case 2: FN<2>(__VA_ARGS__); break;     \
        case 3: FN<3>(__VA_ARGS__); break;     \
        case 4: FN<4>(__VA_ARGS__); break;     \
        case 8: FN<8>(__VA_ARGS__); break;     \
        case 16: FN<16>(__VA_ARGS__); break;   \
        case 32: FN<32>(__VA_ARGS__); break;   \
        case 64: FN<64>(__VA_ARGS__); break;   \
        default: ::ae::fail(AE_ERR_INVALID_ARG, "internal: row stride %u is not instantiated", (unsigned)(dim)); \
    }

struct ae_comm;

struct ae_entropy_optim {
    const ae_kgraph* g = nullptr;
    const ae_node_params* np = nullptr;
    ae_embedder_params params;
    CeDev dev;
    DevBuf<float> y, emb_scale;
    DevBuf<float> edge_odds, hub_odds;
    DevBuf<uint32_t> edge_alias, edge_src, hub_alias;
    DevBuf<uint2> hub_tab;
    DevBuf<double> partial;
    DevBuf<unsigned int> err;
    // sequential-mode scratch
    DevBuf<uint32_t> plan_nodes, order;
    DevBuf<float> plan_w;
    // device-scheduled sequential mode: row versions, and what depends only on (graph, stream, batch): plan, sorted write
    // events, predecessors
    struct DfSet {
        DevBuf<uint32_t> plan_nodes, pred;
        DevBuf<float> plan_w;
        DevBuf<uint64_t> keys0, keys1, rowptr;
    };
    DfSet df_sets[2];
    // The set of batch b + 1 is prepared while the dataflow kernel of batch b runs, each on its own share of the CUs (two CU-masked
    // streams: a latency-bound kernel slows down when throughput kernels share its CUs' memory queues, not when they run elsewhere).
    struct DfAhead {
        hipStream_t run = nullptr, prep = nullptr;   // created on first use; null if the runtime refuses CU masks (then: no overlap)
        hipEvent_t start = nullptr, ran = nullptr, prepared = nullptr;
        int run_cus = 0, tried = 0;
        bool valid = false, relaxed = false;         // a prepared set is waiting: for batch (S, iter) in df_sets[set]
        uint64_t S = 0;
        uint32_t iter = 0, set = 0;
    } df_ahead;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> df_events;  // around the dataflow kernel alone
    DevBuf<float> df_ver;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double events_folded_ms = 0.;
    uint64_t events_folded = 0;
    uint64_t sample_offset = 0;
    // node-centric (owner-computes) Hogwild: transposed graph (in-edges), built at create
    DevBuf<uint64_t> tptr;
    DevBuf<InEdge> tin;
    DevBuf<unsigned long long> sample_counter;  // samples actually drawn (Poisson total), for verification
    uint32_t rounds = 1;
    float in_weight_max = 1.f;                  // largest sum of in-edge probabilities over the nodes (sizes the rounds)
    DevBuf<uint8_t> cnt;     // per-edge sample counts of the current round
    DevBuf<uint32_t> tot;    // per-node planned out-samples
    DevBuf<uint32_t> plan;   // per-node sample plans (cap slots x 6 words)
    // event-ordered mode (ce_event.hip): graph statistics that size the windows, rendezvous slots
    float ev_wave_rate_max[3] = {0.f, 0.f, 0.f}, ev_node_rate_max = 0.f, ev_pmax = 0.f;  // waves of 64 / 32 / 16 nodes
    uint32_t ev_indeg_max = 0;
    int ev_npw = 0;  // nodes per wave
    uint64_t ev_resident_blocks = 0;
    DevBuf<float> ev_slots;
    // time-sliced optimistic mode (ce_slice.hip)
    DevBuf<uint32_t> sl_erec, sl_owner, sl_counts, sl_cnt, sl_offs, sl_keys0, sl_keys1, sl_vals0, sl_vals1, sl_sptr, sl_lists;
    DevBuf<unsigned long long> sl_done;
    DevBuf<uint32_t> sl_chain_head, sl_chain_next;  // chain rounds: per node the head of its pending in-events' list (kept all-NIL between rounds), per list position the next link
    DevBuf<char> sl_sort_tmp;                   // rocPRIM's temporary storage of the event sort (histograms), kept with the handle
    float sl_pmax = 0.f;
    bool sl_prepared = false;                   // ce_slice_prepare has run (a sharded handle defers it to the communicator's attach or its first batch)
    DevBuf<uint8_t> sl_color, sl_class_pos;     // per edge: its colour class (a matching) or the overflow mark; per batch: the class order of every slice
    DevBuf<uint32_t> sl_erec_gen;               // coloured graphs: the edge records in event-generation order (the edges of a class sorted by target) ...
    DevBuf<uint8_t> sl_color_gen;               // ... and their classes (then sl_erec / sl_color are released)
    DevBuf<unsigned long long> sl_dep;          // merged slices: per node, classes through with the node << 32 | classes with an event on it; all zero between slices
    DevBuf<uint32_t> sl_chunk_flag;             // hand-over flags of the hub chains, one per 64-event chunk of the sorted events
    DevBuf<uint32_t> sl_hub_pool;               // (hubness weighting) the batch's pool of NodeSampler draws for the tiles of negatives
    // internal node numbering of the time-sliced mode (one device): node v lives in row sl_perm[v] of sl_y / the static records during a
    // batch (a uniform random relabelling: what runs of consecutive rows hold has nothing to do with the caller's labels)
    DevBuf<uint32_t> sl_perm;
    DevBuf<float> sl_y;
    DevBuf<uint32_t> sl_class_done;             // merged slices, the class window: per slice and class position, the workgroups that are through (zeroed per segment)
    DevBuf<float> sl_neg_snap;                  // (experiment AE_SL_NEG_SNAPSHOT) the copy of the coordinates the negatives are read from
    uint32_t sl_last_form = 0;                  // AE_SLICE_*: the launch form of the last batch (ae_entropy_optim_slice_form)
    uint32_t sl_y_lines = 0;                    // floats per node line whose static part (embedded scale, neighbour ids) sl_y currently holds; 0: none
    DevBuf<uint2> sl_hub_tab;                   // the NodeSampler's alias table in internal numbering
    uint32_t sl_max_in_degree = 0;              // largest in-degree of the graph (the longest chains)
    uint64_t sl_gen_edges = 0;                  // edges this handle generates events for (a shard: those with an end in its node range)
    double sl_gen_mass = 0.;                    // their probability mass (the whole graph: n)
    double sl_cross_frac = 0.;                  // share of it on cross-shard edges (one end here, one elsewhere)
    DevBuf<float> sl_node_ov;                   // per node: probability mass of its overflow edges
    float sl_node_ov_max = 0.f;                 // its maximum (the busiest row of the overflow class)
    DevBuf<float> sl_srec;                      // per node: static record {embedded scale, neighbour ids, edge probabilities}
    uint32_t sl_srec_floats = 16;
    uint32_t sl_classes = 0, sl_color_rounds = 0;
    double sl_ov_frac = 1.;                     // share of the edge probability mass in the overflow class
    double sl_ov_frac_sched = -1.;              // (a sharded run) the largest of the ranks' shares: what every rank cuts its slices by
    // multi-GPU (comm.hip): the communicator, every rank's node range, exchanges of the owned rows per batch
    ae_comm* comm = nullptr;
    std::vector<uint64_t> comm_ranges;
    float* comm_y = nullptr;                    // the array the in-batch exchanges act on, if not y (the time-sliced mode's internal copy)
    bool comm_equal = false;
    uint32_t comm_exchanges = 1;
    uint64_t comm_bytes = 0;                    // bytes of coordinate rows received through exchanges since the handle was created (the other ranks' rows x stride x 4 per exchange; every mode)
    ~ae_entropy_optim() {
        if (df_ahead.prep) { (void)hipStreamSynchronize(df_ahead.prep); (void)hipStreamDestroy(df_ahead.prep); }
        if (df_ahead.run) { (void)hipStreamSynchronize(df_ahead.run); (void)hipStreamDestroy(df_ahead.run); }
        for (hipEvent_t e : {df_ahead.start, df_ahead.ran, df_ahead.prepared}) if (e) (void)hipEventDestroy(e);
        for (auto& e : df_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (auto& e : events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    }
};

namespace ae {
// node-centric Hogwild batch (ce_node.hip): `rounds` launches, expected nb_sample samples in total
bool ce_node_supports(const ae_entropy_optim* o);
void ce_node_build_transpose(ae_entropy_optim* o);
void ce_node_gradient_iteration(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint32_t iter);
void ce_node_gradient_iteration_lockstep(ae_entropy_optim* const* shards, uint32_t world, const uint64_t* nb_sample, double grad_step,
                                         uint32_t iter, uint32_t exchanges);
// event-ordered batch (ce_event.hip): sequentially consistent attraction steps in an i.i.d. order, `rounds` = windows
// time-sliced optimistic batch (ce_slice.hip): exact samples on current rows, i.i.d. order, any graph size; `rounds` = slices
void ce_slice_prepare(ae_entropy_optim* o);
const char* ce_slice_unsupported(const ae_entropy_optim* o);
void ce_slice_gradient_iteration(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint32_t iter);
bool comm_shares_devices(const ae_comm* c);  // comm.hip: the shared-memory transport (its ranks may share one GPU)
void ce_comm_exchange(ae_entropy_optim* o);  // comm.hip: all-gather of the owned coordinate rows on the library's stream
void ce_event_prepare(ae_entropy_optim* o);
const char* ce_event_unsupported(const ae_entropy_optim* o);
void ce_event_gradient_iteration(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint32_t iter);
}  // namespace ae
