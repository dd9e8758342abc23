// internal.h -- cross-file entry points of the library (C++ side, below the C ABI).
#pragma once
#include "objects.h"

struct ae_entropy_optim;
struct ae_comm;

namespace ae {
void to_proba_edges_device(const ae_kgraph* g, float scale_rho, float beta, ae_node_params* np);
void dmap_laplacian_device(const ae_kgraph* g, const ae_diffusion_params* dp, int force_repr, ae_laplacian* lap);
uint32_t embed_from_laplacian_device(ae_laplacian* lap, uint64_t asked_dim, float t, bool has_t, DevBuf<float>& y0,
                                     std::vector<float>* s_out);
void set_data_box_device(float* d_y, uint64_t n, uint64_t dim, float box_size);
// EntropyOptim::new with the initial embedding given on the host or already on the device
ae_entropy_optim* entropy_optim_create_impl(const ae_kgraph* g, const ae_node_params* np, const ae_embedder_params* params,
                                            const float* y0, bool y0_on_device, const uint32_t* hub_counts, uint64_t node_lo,
                                            uint64_t node_hi);
// the mode ae_embedder_params.ce_mode stands for on a given problem (AE_CE_AUTO resolved; see include/annembed_hip.h)
uint32_t resolve_ce_mode(uint32_t mode, uint64_t dim, bool sharded, uint64_t samples_per_batch, uint32_t max_nbng, uint64_t nnz);
// communicator (comm.hip): RCCL or the shared-memory transport; a null or one-rank communicator makes every call a no-op
int comm_rank(const ae_comm* c);
int comm_world(const ae_comm* c);
void comm_broadcast_f32(ae_comm* c, float* d_ptr, uint64_t count, int root);
double comm_all_reduce_sum(ae_comm* c, double value);
void entropy_optim_attach_comm(ae_entropy_optim* o, ae_comm* c, uint32_t exchanges_per_batch);
double ce_slice_max_cross_mass();   // ce_slice.hip: a sharded range with more of its edge mass on cross-shard edges is refused
void comm_broadcast_u32(ae_comm* c, uint32_t* d_ptr, uint64_t count, int root);
// locality partition of a graph in the caller's node order into `world` contiguous position ranges (partition.hip)
struct Partition {
    DevBuf<uint32_t> order;          // order[pos] = caller's id of the node at position pos
    DevBuf<uint32_t> perm;           // perm[id] = pos
    std::vector<uint64_t> ranges;    // lo_0, hi_0, lo_1, hi_1, ...: the ranks' position ranges (they tile [0, n) in rank order)
    uint64_t components = 0, splits = 0;
    double cross_mass = 0., cross_mass_worst_rank = 0., imbalance = 0.;
};
void partition_nodes_device(const ae_kgraph* g, const float* d_proba, const float* d_y, uint32_t ydim, uint32_t ystride, uint32_t world, Partition& out);
void partition_cross_mass_device(const ae_kgraph* g, const float* d_proba, Partition& p);
ae_kgraph* kgraph_permuted_device(const ae_kgraph* g, const uint32_t* d_order, const uint32_t* d_perm, const float* d_extra, DevBuf<float>* extra2);
void permute_rows_device(const float* d_src, float* d_dst, uint64_t n, uint32_t dim, const uint32_t* d_order, bool back);
}  // namespace ae
