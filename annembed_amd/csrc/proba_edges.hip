// proba_edges.hip -- a2: to_proba_edges / get_scale_from_proba_normalisation
// (src/tools/kdumap.rs:26-116, 132-235) on device.
//
// One thread per node: rows hold <= max_nbng (tens of) edges, the only irregular access is the
// gather of each neighbour's first-neighbour distance dist[indptr[y]] (:149-152).  Algorithmic
// traffic per node: k*(4 nbr + 4 dist + 8 indptr gather + 4 dist gather) read, 4k + 4 written.
#include "objects.h"

using namespace ae;

#pragma clang fp contract(off)

__global__ void __launch_bounds__(256) to_proba_edges_kernel(uint64_t n, const uint64_t* __restrict__ indptr,
                                                             const uint32_t* __restrict__ nbr, const float* __restrict__ dist,
                                                             float scale_rho, float beta, float* __restrict__ proba,
                                                             float* __restrict__ scale_out, unsigned long long* err) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = indptr[i];
    const uint64_t nb = indptr[i + 1] - b;
    if (nb == 0) {  // kdumap.rs:75-85
        atomicMin(err, ((unsigned long long)AE_ERR_ISOLATED_NODE << 48) | i);
        return;
    }
    const float* d = dist + b;
    const float rho_x = d[0];  // :146
    float sum = 0.f;
    for (uint64_t m = 0; m < nb; m++) sum += dist[indptr[nbr[b + m]]];  // :149-152
    sum += rho_x;                                                       // :154
    const float mean_rho = sum / (float)(nb + 1);                       // :155
    const float scale = scale_rho * mean_rho;                           // :159
    scale_out[i] = scale;
    const float first_dist = rho_x;
    bool all_equal = false;
    long last = -1;  // :164-166 rfind(weight > 0)
    for (long m = (long)nb - 1; m >= 0; m--)
        if (d[m] > 0.f) { last = m; break; }
    if (last < 0) all_equal = true;  // :167-170
    if (!all_equal) {
        if (d[last] > first_dist) {  // :178
            float s = 0.f;
            float w_first = 0.f, w_last = 0.f;
            for (uint64_t m = 0; m < nb; m++) {
                float w = expf(-powf(fmaxf(d[m] - first_dist, 0.f) / scale, beta));  // :172-174
                w = fmaxf(w, kProbaMin);                                              // :185
                proba[b + m] = w;
                if (m == 0) w_first = w;
                w_last = w;
                s += w;  // :215 (same left-to-right order)
            }
            const float proba_range = w_last / w_first;  // :190
            if (!(proba_range >= kProbaMin)) {           // :209
                atomicMin(err, ((unsigned long long)AE_ERR_PROBA_RANGE << 48) | i);
                return;
            }
            for (uint64_t m = 0; m < nb; m++) proba[b + m] /= s;  // :216-218
            return;
        }
        all_equal = true;  // :221
    }
    const float p = 1.0f / (float)nb;  // :224-230
    for (uint64_t m = 0; m < nb; m++) proba[b + m] = p;
}

// NodeParam::get_perplexity, src/tools/nodeparam.rs:88-91
__global__ void perplexity_kernel(uint64_t n, const uint64_t* __restrict__ indptr, const float* __restrict__ proba,
                                  float* __restrict__ perp) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float h = 0.f;
    for (uint64_t e = indptr[i]; e < indptr[i + 1]; e++) h += -proba[e] * logf(proba[e]);
    perp[i] = expf(h);
}

namespace ae {
// shared with the embedder driver
void to_proba_edges_device(const ae_kgraph* g, float scale_rho, float beta, ae_node_params* np) {
    np->g = g;
    np->proba.alloc(g->nnz);
    np->scale.alloc(g->n);
    DevBuf<unsigned long long> err(1);
    unsigned long long init = ~0ull;
    err.upload(&init, 1);
    hipLaunchKernelGGL(to_proba_edges_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), g->n, g->indptr.p, g->nbr.p,
                       g->dist.p, scale_rho, beta, np->proba.p, np->scale.p, err.p);
    check_launch("to_proba_edges");
    unsigned long long herr;
    err.download(&herr, 1);
    if (herr != ~0ull) {
        int32_t code = (int32_t)(herr >> 48);
        unsigned long long node = herr & ((1ull << 48) - 1);
        if (code == AE_ERR_ISOLATED_NODE)
            fail(code, "to_proba_edges , node rank %llu, has no neighbour, use hnsw.set_keeping_pruned(true)", node);
        fail(code, "proba range too low edge proba at node %llu, increase scale_rho or reduce beta", node);
    }
}
}  // namespace ae

extern "C" {

int32_t ae_to_proba_edges(const ae_kgraph* g, float scale_rho, float beta, ae_node_params** out) {
    return guard([&] {
        require_device();
        if (!g || !out) fail(AE_ERR_INVALID_ARG, "null argument");
        std::unique_ptr<ae_node_params> np(new ae_node_params);
        to_proba_edges_device(g, scale_rho, beta, np.get());
        *out = np.release();
    });
}
int32_t ae_node_params_from_host(const ae_kgraph* g, const float* proba, const float* scale, ae_node_params** out) {
    return guard([&] {
        require_device();
        if (!g || !proba || !scale || !out) fail(AE_ERR_INVALID_ARG, "null argument");
        std::unique_ptr<ae_node_params> np(new ae_node_params);
        np->g = g;
        np->proba.alloc(g->nnz);
        np->proba.upload(proba, g->nnz);
        np->scale.alloc(g->n);
        np->scale.upload(scale, g->n);
        sync();
        *out = np.release();
    });
}
int32_t ae_node_params_destroy(ae_node_params* np) {
    return guard([&] { delete np; });
}
int32_t ae_node_params_get(const ae_node_params* np, float* proba, float* scale) {
    return guard([&] {
        if (!np) fail(AE_ERR_INVALID_ARG, "null argument");
        if (proba) np->proba.download(proba, np->g->nnz);
        if (scale) np->scale.download(scale, np->g->n);
    });
}
int32_t ae_node_params_perplexity(const ae_node_params* np, float* perplexity) {
    return guard([&] {
        require_device();
        if (!np || !perplexity) fail(AE_ERR_INVALID_ARG, "null argument");
        DevBuf<float> p(np->g->n);
        hipLaunchKernelGGL(perplexity_kernel, dim3(blocks_for(np->g->n, 256)), dim3(256), 0, stream(), np->g->n, np->g->indptr.p,
                           np->proba.p, p.p);
        check_launch("perplexity");
        p.download(perplexity, np->g->n);
    });
}
}
