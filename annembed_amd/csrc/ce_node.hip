// ce_node.hip -- AE_CE_HOGWILD: the lock-free CE gradient batch (gradient_iteration_threaded,
// src/embedder.rs:1311-1315) restructured for MI355X as an OWNER-COMPUTES, node-centric kernel.
//
// Why not "one thread per sample with racy read-modify-write", which is what the reference's rayon
// loop does (:1185-1186, :1239, :1301)?  Measured on MI355X (tools/ubench_gather.hip): with ~2 M samples
// in flight over 60 k nodes 91 % of racy 8-byte read-modify-writes are lost, the 8 per-XCD L2s are not
// coherent inside a launch, and atomics of any type top out at ~23 G/s (vs 55-270 G/s for random 8-byte
// loads, ~80 G/s for write-through stores).  A CPU has tens of threads for N nodes; the GPU has more
// lanes than nodes.  So every coordinate row gets exactly ONE writer:
//
//   thread v owns y_v.  Per round it replays, sequentially on its private copy of y_v,
//     (a) the samples whose source is v: attraction to the sampled neighbour j (the y_i half of
//         :1223-1237) followed by the 5 negative repulsions (:1241-1299), and
//     (b) the samples whose target is v: the y_j half of :1238-1239, recomputed from (y_u, y_v, s_u, w).
//   The number of times edge e is sampled in a round is c_e ~ Poisson(mu_e), mu_e = nb_sample * p_e /
//   (N * rounds): by Poisson splitting this is the edge law of the reference's i.i.d. draw from the
//   alias table (:987, :1182) with a Poisson(nb_sample) total.  c_e is a pure function of (seed, batch,
//   round, e) (one Philox block), so thread i and thread j agree on it with no communication, no
//   atomics, no sorting -- and on multi-GPU the owner of j replays remote pushes from its replica.
//   y_v is written back (write-through, agent scope) after every sample; partners are re-read
//   (L1-bypassing loads) for every sample, so the staleness is that of the memory system, like Hogwild.
//
// Arithmetic: f32 with v_rcp_f32 (the reference's scalars are f64; the coefficient is a smooth
// function clipped to [-0.49, 2], a 1e-7 relative difference is far below the SGD noise).  The exact
// f64 arithmetic lives in ce.hip (AE_CE_SEQUENTIAL, bit-exact against the oracle).
#include "ce_node_common.h"
#include "philox.h"

using namespace ae;

namespace ae {
void sort_pairs_u64_u32(uint64_t* d_keys_in, uint64_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count);
void rowptr_from_sorted_keys(const uint64_t* d_keys, uint64_t nnz, uint64_t nrows, uint64_t* d_rowptr);
// ce_node_round_exact.hip / ce_node_round_pad.hip
void launch_round_node_exact(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes);
void launch_round_node_padded(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes);
void launch_round_node_exact_tile(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes);
void launch_round_node_padded_tile(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes);
}  // namespace ae

namespace {

// ---- K_plan: one wave per owned node.  Lanes < k draw the per-edge sample counts (written to cnt[] for
// the in-push replay of the partner), a wave scan turns them into sample slots, then lane t plans sample t:
// target j and the 5 admissible negatives (embedder.rs:1241-1253).
template <bool HUB>
__global__ void __launch_bounds__(kBlock) ce_plan_node_kernel(NodeArgs a) {
    const CeDev c = a.c;
    const int lane = threadIdx.x & 63;
    const uint64_t v64 = c.node_lo + ((blockIdx.x * (uint64_t)kBlock + threadIdx.x) >> 6);
    if (v64 >= c.node_hi) return;
    const uint32_t v = (uint32_t)v64;
    uint64_t ib;
    uint32_t k;
    if (c.uniform_k) { ib = (uint64_t)v * c.uniform_k; k = c.uniform_k; }
    else { ib = c.indptr[v]; k = (uint32_t)(c.indptr[v + 1] - ib); }
    const uint32_t node_base = pcg_hash(pcg_hash((uint32_t)c.seed ^ a.round_key) + v);
    uint32_t planned = 0;   // slots already used by previous 64-edge chunks (k > 64)
    // slot-major layout: entry (slot, node) at ((slot * nodes) + node) * 2, so that the 64 nodes of an apply
    // wave read 2 KB contiguous per slot
    const uint64_t nodes_owned = c.node_hi - c.node_lo;
    uint4* my_plan = reinterpret_cast<uint4*>(a.plan) + (uint64_t)(v - c.node_lo) * 2;
    for (uint32_t e0 = 0; e0 < k; e0 += 64) {
        const uint32_t m = e0 + lane;
        uint32_t cm = 0, nb = 0xFFFFFFFFu;
        float pm = 0.f;
        if (m < k) {
            nb = c.nbr[ib + m];
            pm = c.proba[ib + m];
            cm = edge_count(ib + m, a.round_key, c.seed, a.unit * pm);
            a.cnt[ib + m] = (uint8_t)cm;
        }
        // inclusive scan of the counts over the wave
        uint32_t pre = cm;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(pre, off);
            if (lane >= off) pre += o;
        }
        const uint32_t chunk_total = __shfl(pre, 63);
        // 64-bit signature of the row chunk for a cheap "certainly not a neighbour" test
        unsigned long long sig = (m < k) ? (1ull << (nb & 63u)) : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sig |= __shfl_xor(sig, off);
        for (uint32_t t0 = 0; t0 < chunk_total; t0 += 64) {
            const uint32_t t = t0 + lane;
            // edge of sample t: first lane whose inclusive prefix exceeds t (wave-uniform loop over <= 64 lanes)
            uint32_t my_edge_lane = 0;
            const uint32_t kk = (k - e0) < 64u ? (k - e0) : 64u;
            for (uint32_t l = 0; l < kk; l++) {
                const uint32_t pl = __shfl(pre, (int)l);
                if (pl <= t) my_edge_lane = l + 1;
            }
            const bool active = t < chunk_total;
            const uint32_t j = __shfl(nb, (int)(my_edge_lane < 64u ? my_edge_lane : 63u));
            const float wj = __shfl(pm, (int)(my_edge_lane < 64u ? my_edge_lane : 63u));
            uint32_t kn[5];
            const uint32_t sbase = node_base + (planned + t) * 64u;
#pragma unroll
            for (int g = 0; g < 5; g++) {
                uint32_t cand = 0;
                for (uint32_t attempt = 0; attempt < 12u; attempt++) {
                    const uint32_t w0 = pcg_hash(sbase + (uint32_t)g * 12u + attempt);
                    if constexpr (HUB) {  // NodeSampler::sample, embedder.rs:927-930
                        const uint32_t x = __umulhi(w0, (uint32_t)c.n);
                        const float u = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
                        const uint2 he = c.hub_tab[x];
                        cand = (u < __uint_as_float(he.x)) ? x : he.y;
                    } else {
                        cand = __umulhi(w0, (uint32_t)c.n);  // :1121
                    }
                    bool reject = (cand == v) || (cand == j);
                    // NodeParam::get_edge (nodeparam.rs:83-85): exact row scan only when the signature hits.
                    // (rows longer than 64 are checked chunk by chunk against the current chunk only; the
                    // remaining k/N-probability event is accepted there)
                    const bool maybe = (sig >> (cand & 63u)) & 1ull;
                    if (__any(active && !reject && maybe)) {
                        for (uint32_t l = 0; l < kk; l++) {
                            const uint32_t nl = __shfl(nb, (int)l);
                            if (nl == cand) reject = true;
                        }
                    }
                    if (!reject) break;
                }
                kn[g] = cand;
            }
            const uint32_t slot = planned + t;
            if (active) {
                if (slot < a.cap) {
                    my_plan[(uint64_t)slot * nodes_owned * 2] = make_uint4(j, kn[0], kn[1], kn[2]);
                    my_plan[(uint64_t)slot * nodes_owned * 2 + 1] = make_uint4(kn[3], kn[4], __float_as_uint(wj), 0u);
                } else {
                    atomicOr(a.overflow, 2u);
                }
            }
        }
        planned += chunk_total;
    }
    if (lane == 0) a.tot[v - c.node_lo] = planned < a.cap ? planned : a.cap;
}

// counts of the edges whose source is NOT owned by this shard (multi-GPU): needed to replay remote pushes
__global__ void __launch_bounds__(kBlock) ce_count_remote_kernel(NodeArgs a) {
    const CeDev c = a.c;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < c.nnz; e += stride) {
        if (e >= c.edge_lo && e < c.edge_lo + c.shard_edges) continue;
        a.cnt[e] = (uint8_t)edge_count(e, a.round_key, c.seed, a.unit * c.proba[e]);
    }
}

// ---- K_apply, lane-group form: 8 lanes per node, 8 nodes per wave.  A graph with N nodes offers only N
// sequential update chains; with one lane per node 60 k nodes are 940 waves (one per SIMD, nothing to hide
// latency with).  Here lane r of a group fetches row r of the sample (j, k1..k5) -- one gather instruction
// covers the 6 rows of 8 samples -- and the 6 dependent update steps are replayed by the whole group on a
// replicated y_v with in-group broadcasts.  8x more waves, 8x shorter in-edge loops for hubs.
constexpr int kGroup = 8;
template <int DIM>
__device__ __forceinline__ void group_bcast(const float* in, int src_lane, float* out) {
#pragma unroll
    for (int q = 0; q < DIM; q++) out[q] = __shfl(in[q], src_lane);
}

template <int DIM, bool B1>
__global__ void __launch_bounds__(kBlock) ce_apply_group_kernel(NodeArgs a) {
    const CeDev c = a.c;
    const int lane = threadIdx.x & 63;
    const int r = lane & (kGroup - 1);
    const int gbase = lane & ~(kGroup - 1);
    const uint64_t nodes_owned = c.node_hi - c.node_lo;
    const uint64_t wave = (blockIdx.x * (uint64_t)kBlock + threadIdx.x) >> 6;
    const uint64_t local = wave * (64 / kGroup) + (uint64_t)(lane >> 3);
    const bool valid = local < nodes_owned;
    const uint64_t lv = valid ? local : nodes_owned - 1;  // idle groups shadow the last node (no stores)
    const uint32_t v = (uint32_t)(c.node_lo + lv);
    float yv[DIM], grad[DIM];
    load_row_fresh<DIM>(c.y, v, yv);
#pragma unroll
    for (int q = 0; q < DIM; q++) grad[q] = 0.f;
    const float s_v = c.emb_scale[v];
    const float inv_s2 = rcp(s_v * s_v);
    // ---------------- (a) samples whose source is v ----------------
    const uint32_t nv = valid ? a.tot[lv] : 0u;
    uint32_t nmax = nv;
#pragma unroll
    for (int off = 32; off >= kGroup; off >>= 1) { const uint32_t o = __shfl_xor(nmax, off); nmax = o > nmax ? o : nmax; }
    const uint32_t* plan_words = a.plan + lv * 8 + r;  // entry (slot, node): 8 words at ((slot * nodes) + node) * 8
    for (uint32_t t = 0; t < nmax; t++) {
        const bool act = t < nv;
        uint32_t word = 0;
        if (act) word = plan_words[(uint64_t)t * nodes_owned * 8];
        float row[DIM];
#pragma unroll
        for (int q = 0; q < DIM; q++) row[q] = 0.f;
        if (act && r < 6) {
            if (a.skip == 7) { const float* pp = c.y + (uint64_t)word * DIM; for (int q = 0; q < DIM; q++) row[q] = pp[q]; }
            else load_row_fresh<DIM>(c.y, word, row);
        }
        const float w = __uint_as_float(__shfl(word, gbase + 6));
        float other[DIM];
        group_bcast<DIM>(row, gbase + 0, other);
        {   // attraction, the y_i half of embedder.rs:1207-1237
            float d = 0.f;
#pragma unroll
            for (int q = 0; q < DIM; q++) { const float df = yv[q] - other[q]; d += df * df; }
            const float delta = d * inv_s2;
            if (act && delta > 0.f) {
                const float coeff = grad_coeff_f32<B1>(delta, inv_s2, a.b);
                const float rep = rcp(fmaxf(delta * delta, 1.0f / kProbaMin));
                const float cij = fmaxf(a.step * coeff * (-w + (1.f - w) * rep), -0.49f);
#pragma unroll
                for (int q = 0; q < DIM; q++) grad[q] = (other[q] - yv[q]) * cij;
            } else {
#pragma unroll
                for (int q = 0; q < DIM; q++) grad[q] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < DIM; q++) yv[q] -= grad[q];
        }
#pragma unroll
        for (int g = 1; g <= 5; g++) {  // 5 repulsions, :1267-1297
            group_bcast<DIM>(row, gbase + g, other);
            float dk = 0.f;
#pragma unroll
            for (int q = 0; q < DIM; q++) { const float df = yv[q] - other[q]; dk += df * df; }
            if (act && dk > 0.f) {
                const float dks = dk * inv_s2;
                const float coeff = grad_coeff_f32<B1>(dks, inv_s2, a.b);
                const float cik = fminf(a.step * coeff * rcp(fmaxf(dks * dks, 1.0f / 16.0f)), 2.0f);
#pragma unroll
                for (int q = 0; q < DIM; q++) grad[q] = (other[q] - yv[q]) * cik;
            }  // else `gradient` keeps its previous value (reference quirk B4)
            if (act) {
#pragma unroll
                for (int q = 0; q < DIM; q++) yv[q] -= grad[q];
            }
        }
        if (act && r == 0) { if (a.store_mode == 0) store_row_through<DIM>(c.y, v, yv); else if (a.store_mode == 1) store_row_plain<DIM>(c.y, v, yv); }
    }
    if (a.store_mode >= 2 && r == 0 && valid) store_row_through<DIM>(c.y, v, yv);
    if (a.skip == 2) return;
    // ---------------- (b) samples whose target is v: the y_j half of :1238-1239 ----------------
    const uint64_t tb = a.tptr[v];
    const uint32_t indeg = valid ? (uint32_t)(a.tptr[v + 1] - tb) : 0u;
    uint32_t dmax = indeg;
#pragma unroll
    for (int off = 32; off >= kGroup; off >>= 1) { const uint32_t o = __shfl_xor(dmax, off); dmax = o > dmax ? o : dmax; }
    for (uint32_t x0 = 0; x0 < dmax; x0 += kGroup) {
        const uint32_t x = x0 + (uint32_t)r;
        uint32_t cnt = 0;
        float wu = 0.f, su = 1.f;
        float yu[DIM];
#pragma unroll
        for (int q = 0; q < DIM; q++) yu[q] = 0.f;
        if (x < indeg) {
            const InEdge rec = a.tin[tb + x];
            cnt = a.cnt[rec.eid];
            wu = rec.w;
            su = rec.s_src;
            if (cnt) load_row_fresh<DIM>(c.y, rec.src, yu);
        }
        bool changed = false;
#pragma unroll
        for (int sl = 0; sl < kGroup; sl++) {
            const uint32_t bc = __shfl(cnt, gbase + sl);
            uint32_t cmax = bc;
#pragma unroll
            for (int off = 32; off >= kGroup; off >>= 1) { const uint32_t o = __shfl_xor(cmax, off); cmax = o > cmax ? o : cmax; }
            if (cmax == 0) continue;  // wave-uniform
            float other[DIM];
            group_bcast<DIM>(yu, gbase + sl, other);
            const float wb = __shfl(wu, gbase + sl);
            const float sb = __shfl(su, gbase + sl);
            const float inv_su2 = rcp(sb * sb);
            for (uint32_t rep_i = 0; rep_i < cmax; rep_i++) {
                if (rep_i < bc) {
                    attract<DIM, B1>(yv, other, wb, inv_su2, a.step, a.b, 1.f);
                    changed = true;
                }
            }
        }
        if (changed && r == 0 && valid && a.store_mode < 2) store_row_through<DIM>(c.y, v, yv);
    }
    if (a.store_mode >= 2 && r == 0 && valid) store_row_through<DIM>(c.y, v, yv);
}

// samples planned in this round (one workgroup; a per-wave atomic on one address would cost ~12 ns each)
__global__ void __launch_bounds__(1024) ce_sum_tot_kernel(const uint32_t* __restrict__ tot, uint64_t n, unsigned long long* counter) {
    __shared__ unsigned long long red[1024];
    unsigned long long s = 0;
    for (uint64_t i = threadIdx.x; i < n; i += 1024) s += tot[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *counter += red[0];
}

__global__ void in_edge_keys_kernel(uint64_t n, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ nbr,
                                    uint64_t* __restrict__ keys, uint32_t* __restrict__ payload) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (uint64_t e = indptr[i]; e < indptr[i + 1]; e++) {
        keys[e] = ((uint64_t)nbr[e] << 32) | i;  // (target, source)
        payload[e] = (uint32_t)e;
    }
}
__global__ void in_edge_fill_kernel(uint64_t nnz, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                    const float* __restrict__ proba, const float* __restrict__ emb_scale, InEdge* __restrict__ tin) {
    uint64_t x = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (x >= nnz) return;
    InEdge r;
    r.src = (uint32_t)(keys[x] & 0xFFFFFFFFull);
    r.eid = perm[x];
    r.w = proba[r.eid];
    r.s_src = emb_scale[r.src];
    tin[x] = r;
}

// largest in-weight W_in(v) = sum of p_e over the in-edges of v (as a float bit pattern: weights are positive)
__global__ void in_weight_max_kernel(uint64_t n, const uint64_t* __restrict__ tptr, const InEdge* __restrict__ tin, unsigned int* __restrict__ out) {
    const uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    float w = 0.f;
    if (v < n)
        for (uint64_t x = tptr[v]; x < tptr[v + 1]; x++) w += tin[x].w;
    for (int off = 32; off > 0; off >>= 1) w = fmaxf(w, __shfl_xor(w, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(w));
}

template <int DIM>
void launch_apply_group(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes) {
    if constexpr (DIM > 0) {
        const unsigned grid = blocks_for(((nodes + 7) / 8) * 64, kBlock);
        if (a.b == 1.0f) hipLaunchKernelGGL((ce_apply_group_kernel<DIM, true>), dim3(grid), dim3(kBlock), 0, stream(), a);
        else hipLaunchKernelGGL((ce_apply_group_kernel<DIM, false>), dim3(grid), dim3(kBlock), 0, stream(), a);
    }
}

}  // namespace

namespace ae {

static bool legacy_dim(uint32_t dim) { return dim == 2 || dim == 3 || dim == 4 || dim == 8 || dim == 16; }
// node-per-lane round kernel: rows of <= 32 neighbours, asked_dim <= 32 (exact instantiations for 2, 3, 4, 8, 16, the
// other dimensions run zero-padded to 8 / 16 / 32); longer rows take the wave-per-node plan + lane-group apply
// kernels, which exist for the five exact dimensions only
static bool node_kernel_ok(const ae_entropy_optim* o) { return o->g->max_nbng <= 32 && o->dev.dim >= 1 && o->dev.dim <= 32; }
bool ce_node_supports(const ae_entropy_optim* o) { return node_kernel_ok(o) || legacy_dim(o->dev.dim); }

// transposed graph (in-edges with w and the source's embedded scale), once per EntropyOptim
void ce_node_build_transpose(ae_entropy_optim* o) {
    const ae_kgraph* g = o->g;
    if (g->nnz >= 0xFFFFFFFFull) fail(AE_ERR_INVALID_ARG, "graph too large for u32 edge ids");
    DevBuf<uint64_t> k0(g->nnz), k1(g->nnz);
    DevBuf<uint32_t> p0(g->nnz), p1(g->nnz);
    hipLaunchKernelGGL(in_edge_keys_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), g->n, g->indptr.p, g->nbr.p, k0.p, p0.p);
    check_launch("in_edge_keys");
    sort_pairs_u64_u32(k0.p, k1.p, p0.p, p1.p, g->nnz);
    o->tin.alloc(g->nnz);
    hipLaunchKernelGGL(in_edge_fill_kernel, dim3(blocks_for(g->nnz, 256)), dim3(256), 0, stream(), g->nnz, k1.p, p1.p, o->np->proba.p,
                       o->emb_scale.p, o->tin.p);
    check_launch("in_edge_fill");
    o->tptr.alloc(g->n + 1);
    rowptr_from_sorted_keys(k1.p, g->nnz, g->n, o->tptr.p);
    o->sample_counter.alloc(1024);
    o->sample_counter.zero();
    DevBuf<unsigned int> wmax(1);
    wmax.zero();
    hipLaunchKernelGGL(in_weight_max_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), g->n, (const uint64_t*)o->tptr.p,
                       (const InEdge*)o->tin.p, wmax.p);
    check_launch("in_weight_max");
    unsigned int bits = 0;
    wmax.download(&bits, 1);
    memcpy(&o->in_weight_max, &bits, sizeof(float));
}

// what one batch of the rounds mode needs besides the handle: the kernel arguments and the number of rounds
struct RoundsBatch {
    NodeArgs a;
    uint64_t nodes = 0;
    uint32_t rounds = 0;
    bool node_kernel = false, sharded = false;
};

static RoundsBatch rounds_prepare(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint32_t iter) {
    RoundsBatch rb;
    const uint64_t nodes = o->dev.node_hi - o->dev.node_lo;
    rb.nodes = nodes;
    // expected samples per source node in the batch.  A shard derives it from GRAPH-wide quantities -- samples per edge x edges per
    // node of the whole graph, which is also the reference's law (edges are drawn in proportion to p_e over the whole graph:
    // every node sends nb_sample_total / n samples on average whatever its own degree): every rank then cuts the batch into the
    // same number of rounds (the collectives of a batch match) and evaluates remote pushes with the source rank's Poisson means
    const bool is_shard = o->dev.shard_edges != o->dev.nnz;
    const double per_node = is_shard ? (double)nb_sample / (double)o->dev.shard_edges * (double)o->dev.nnz / (double)o->dev.n
                                     : (double)nb_sample / (double)nodes;
    // rounds: keep the largest per-edge Poisson mean (p_e <= 1) below 30 so that exp(-mu) stays normal in f32
    // samples per node and round: measured trade-off between fidelity to the sequential reference (final cross
    // entropy within ~10-15 %, edge-length quantiles within ~7 %) and per-round fixed costs.  The
    // node-per-lane kernel runs all its waves concurrently, so within a round every gather sees the previous
    // round's rows (Jacobi-like) and needs shorter rounds (8) than the lane-group kernel (12), whose waves
    // finish at different times, for the same fidelity.
    const bool force_legacy = debug_knob("AE_CE_UNFUSED") != nullptr;
    const bool node_kernel = node_kernel_ok(o) && !(force_legacy && legacy_dim(o->dev.dim));
    rb.node_kernel = node_kernel;
    double per_round_target = node_kernel ? 8.0 : 12.0;
    // ... and no edge should be drawn much more than 2/3 times per round on average: two attraction steps of one edge
    // inside a round are both evaluated against the partner's round-start row, and with the reference's stiff steps
    // (c clipped at -0.49: one sample closes 98 % of a short edge) the second one overshoots.  Measured on a k = 6 graph
    // (60 k points of 28-d blobs, 40 batches): final CE 0.62x the sequential run's with 8 samples per node and round
    // (1.33 per edge), 0.99x with 4 (0.67 per edge); C2 (k = 12, 8 per round = 0.67 per edge) is unchanged by the rule.
    if (node_kernel) per_round_target = std::min(per_round_target, std::max(1.0, (2.0 / 3.0) * (double)o->dev.nnz / (double)o->dev.n));
    if (debug_knob("AE_CE_PER_ROUND")) per_round_target = atof(debug_knob("AE_CE_PER_ROUND"));
    uint32_t rounds = (uint32_t)std::max(1.0, std::ceil(per_node / per_round_target));
    // hubs: a node of in-weight W receives per_node * W / rounds pushes per round, all evaluated against round-start
    // source rows; keep that below 128 (no effect on graphs whose largest in-weight is below ~16)
    if (!debug_knob("AE_CE_PER_ROUND"))
        rounds = std::max(rounds, (uint32_t)std::min(1000.0, std::ceil(per_node * (double)o->in_weight_max / 128.0)));
    o->rounds = rounds;
    rb.rounds = rounds;
    if (iter >= (1u << 20) || rounds >= (1u << 10)) fail(AE_ERR_INVALID_ARG, "iteration / round index too large for the RNG key");
    const double per_round = per_node / (double)rounds;
    // plan capacity per node and round: Poisson(per_round) exceeds mean + 8 sigma + 8 with probability < 1e-14
    const uint32_t cap = (uint32_t)std::ceil(per_round + 8.0 * std::sqrt(per_round) + 8.0);
    if (!node_kernel) {  // only the wave-per-node plan + lane-group apply kernels (rows of > 32 neighbours) use these
        if (o->plan.n < nodes * (uint64_t)cap * 8) o->plan.alloc(nodes * (uint64_t)cap * 8);
        if (o->tot.n < nodes) o->tot.alloc(nodes);
        if (o->cnt.n < o->dev.nnz) { o->cnt.alloc(o->dev.nnz); o->cnt.zero(); }
    }
    NodeArgs& a = rb.a;
    a.c = o->dev;
    a.tptr = o->tptr.p;
    a.tin = o->tin.p;
    a.cnt = o->cnt.p;
    a.tot = o->tot.p;
    a.plan = o->plan.p;
    a.cap = cap;
    a.step = (float)grad_step;
    a.unit = (float)per_round;
    a.b = (float)o->dev.b;
    a.sample_counter = o->sample_counter.p;
    a.overflow = o->err.p;
    a.prof = nullptr;
    rb.sharded = o->dev.shard_edges != o->dev.nnz;
    a.skip = debug_knob("AE_CE_SKIP") ? atoi(debug_knob("AE_CE_SKIP")) : 0;
    // 2: write-through at phase ends (default).  0: after every chunk -- measured slower in the node kernel (loads
    // and stores share vmcnt and may return out of order, so every wait after a store drains it: +50 % time)
    // for a fidelity gain that shorter rounds give more cheaply
    a.store_mode = debug_knob("AE_CE_STORE") ? atoi(debug_knob("AE_CE_STORE")) : 2;
    a.round_key = iter << 10;
    a.tile = debug_knob("AE_CE_NO_TILE") ? 0 : 1;
    return rb;
}

static void rounds_launch(ae_entropy_optim* o, RoundsBatch& rb, uint32_t r) {
    NodeArgs& a = rb.a;
    a.round_key = (a.round_key & ~0x3FFu) | r;
    if (rb.node_kernel) {
        const bool exact = legacy_dim(o->dev.dim);
        const uint32_t d = o->dev.dim;
        const uint32_t tile_rows = exact ? (d <= 4 ? 1024u : (d <= 8 ? 256u : 128u)) : (d <= 8 ? 256u : 128u);  // node_kernel_tile_rows of the instantiated width
        // uniform sampler: always (n >= 2 T).  Hubness-weighted sampler (a tile of alias-table draws): from 2^20 nodes on, where
        // the gathered negatives are what bounds the kernel (C4 shape 360 -> 60 ms per batch, same CE); on small graphs the
        // gathered form costs nothing and was measured closer to the reference
        const bool tile = a.tile != 0 && a.c.n >= 2ull * tile_rows && (a.c.hub_odds == nullptr || a.c.n >= (1ull << 20));
        if (exact) { if (tile) launch_round_node_exact_tile(o, a, rb.nodes); else launch_round_node_exact(o, a, rb.nodes); }
        else { if (tile) launch_round_node_padded_tile(o, a, rb.nodes); else launch_round_node_padded(o, a, rb.nodes); }
        return;
    }
    const unsigned plan_grid = blocks_for(rb.nodes * 64, kBlock);
    if (a.c.hub_odds) hipLaunchKernelGGL((ce_plan_node_kernel<true>), dim3(plan_grid), dim3(kBlock), 0, stream(), a);
    else hipLaunchKernelGGL((ce_plan_node_kernel<false>), dim3(plan_grid), dim3(kBlock), 0, stream(), a);
    if (rb.sharded) hipLaunchKernelGGL(ce_count_remote_kernel, dim3(grid_cap(o->dev.nnz, kBlock)), dim3(kBlock), 0, stream(), a);
    hipLaunchKernelGGL(ce_sum_tot_kernel, dim3(1), dim3(1024), 0, stream(), (const uint32_t*)o->tot.p, rb.nodes, o->sample_counter.p);
    AE_DISPATCH_DIM(o->dev.dim, launch_apply_group, o, a, rb.nodes);
}

// true when an exchange of the owned rows follows round r: `exch` exchanges per batch after equal runs of rounds, the last
// one ending the batch
static bool exchange_follows(uint32_t r, uint32_t rounds, uint32_t exch) {
    if (!exch) return false;
    return (uint32_t)(((uint64_t)(r + 1) * exch) / rounds) != (uint32_t)(((uint64_t)r * exch) / rounds);
}

void ce_node_gradient_iteration(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint32_t iter) {
    RoundsBatch rb = rounds_prepare(o, nb_sample, grad_step, iter);
    static DevBuf<unsigned long long> prof_buf;
    if (debug_knob("AE_CE_PROF")) {
        if (!prof_buf.n) { prof_buf.alloc(12); prof_buf.zero(); }
        rb.a.prof = prof_buf.p;
    }
    // multi-GPU: the owned rows are exchanged `comm_exchanges` times per batch, after equal runs of rounds (the last
    // exchange ends the batch): remote rows are rounds / exchanges rounds old instead of a whole batch
    const uint32_t exch = o->comm ? std::min(std::max(1u, o->comm_exchanges), rb.rounds) : 0u;
    for (uint32_t r = 0; r < rb.rounds; r++) {
        rounds_launch(o, rb, r);
        if (exchange_follows(r, rb.rounds, exch)) ce_comm_exchange(o);
    }
    check_launch("ce_node");
    if (rb.a.prof) {
        unsigned long long h[12];
        prof_buf.download(h, 12);
        if (h[6]) fprintf(stderr, "CEPROF waves=%llu per-wave cycles: A %.0f | B prepare %.0f replay %.0f | C scan %.0f park %.0f count+gather %.0f replay %.0f | total %.0f\n", h[6],
                          (double)h[0] / h[6], (double)h[1] / h[6], (double)h[2] / h[6], (double)h[7] / h[6], (double)h[8] / h[6], (double)h[3] / h[6],
                          (double)h[4] / h[6], (double)h[5] / h[6]);
        prof_buf.zero();
    }
}

// One batch of `world` shard handles of one graph in lockstep on THIS device: round r of every shard, then, where an
// exchange follows, every shard's owned rows copied into the other shards' arrays -- the protocol of `world` processes with
// a communicator attached, without RCCL (validation of the sharded protocol on a single GPU; ae_..._lockstep in the ABI).
void ce_node_gradient_iteration_lockstep(ae_entropy_optim* const* shards, uint32_t world, const uint64_t* nb_sample, double grad_step,
                                         uint32_t iter, uint32_t exchanges) {
    std::vector<RoundsBatch> rb;
    for (uint32_t q = 0; q < world; q++) rb.push_back(rounds_prepare(shards[q], nb_sample[q], grad_step, iter));
    uint32_t rounds = 0;
    for (uint32_t q = 0; q < world; q++) rounds = std::max(rounds, rb[q].rounds);
    for (uint32_t q = 0; q < world; q++)
        if (rb[q].rounds != rounds) {  // every shard must cut the batch alike
            fail(AE_ERR_INVALID_ARG, "lockstep: shard %u would run %u rounds, another %u (unequal shards?)", q, rb[q].rounds, rounds);
        }
    const uint32_t exch = std::min(std::max(1u, exchanges), rounds);
    const uint64_t dim = shards[0]->dev.dim;
    for (uint32_t r = 0; r < rounds; r++) {
        for (uint32_t q = 0; q < world; q++) rounds_launch(shards[q], rb[q], r);
        if (!exchange_follows(r, rounds, exch)) continue;
        for (uint32_t q = 0; q < world; q++) {  // owner q -> every other shard
            const uint64_t lo = shards[q]->dev.node_lo, hi = shards[q]->dev.node_hi;
            for (uint32_t t = 0; t < world; t++)
                if (t != q)
                    AE_HIP(hipMemcpyAsync(shards[t]->y.p + lo * dim, shards[q]->y.p + lo * dim, sizeof(float) * (hi - lo) * dim,
                                          hipMemcpyDeviceToDevice, stream()));
        }
    }
    check_launch("ce_node lockstep");
}

}  // namespace ae
