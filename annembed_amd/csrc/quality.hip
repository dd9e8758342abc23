// quality.hip -- Embedder::get_quality_estimate_from_edge_length, src/embedder.rs:620-753, with
// get_transformed_kgraph (:478-522) and get_max_edge_length_embedded_kgraph (:527-554), on the device.
//
// What the reference computes, per node i of the original graph:
//   * transformed neighbourhood (:495-516): for the neighbours in their stored order a RUNNING MINIMUM of the
//     embedded L2 distances (`node_edge_length = distl2(..).min(node_edge_length)`, pushed for every edge), then
//     sorted ascending -- kept as is (the multiset is what the statistics below read);
//   * radius_i (:527-554, kgraph.rs:167-183): largest edge of node i in a kNN graph of size nbng built on the
//     EMBEDDED points.  The reference builds that graph with hnsw_rs (approximate, un-vendored: parity unpinned);
//     here it is the exact nbng-th neighbour distance (brute force on the device);
//   * matches_i = #{e : w_e <= radius_i}, ratio_e = w_e / radius_i (f64), node_ratio_i = mean_e ratio_e.
// Summary: nodes without a match, mean matches of the others, quantiles of the radii and of the ratios at
// 0.05 .25 .5 .75 .85 .95 (reference: CKMS sketches with eps = 0.01, i.e. any element within 1 % of the rank;
// here the exact order statistic at rank floor(q * count)), median and mean ratio.
#include "internal.h"
#include "linalg.h"

#include <rocprim/rocprim.hpp>

using namespace ae;

namespace {

// thread per node: running minimum of the embedded edge lengths (embedder.rs:499-512), stored ascending
__global__ void transformed_edges_kernel(uint64_t n, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ nbr,
                                         const float* __restrict__ y, uint32_t dim, float* __restrict__ tw) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = indptr[i], e = indptr[i + 1];
    float running = 3.402823466e+38f;  // F::max_value()
    for (uint64_t x = b; x < e; x++) {
        const uint32_t j = nbr[x];
        float s = 0.f;  // distl2, embedder.rs:54-65: sum of squares in coordinate order, then sqrt
        for (uint32_t t = 0; t < dim; t++) {
            const float df = y[i * dim + t] - y[(uint64_t)j * dim + t];
            s += df * df;
        }
        running = fminf(sqrtf(s), running);
        tw[e - 1 - (x - b)] = running;  // non-increasing sequence written backwards = sorted ascending
    }
}

// thread per node: matches, ratios (f64), per-node mean ratio, first (smallest) transformed length
__global__ void quality_node_kernel(uint64_t n, const uint64_t* __restrict__ indptr, const float* __restrict__ tw,
                                    const uint64_t* __restrict__ e_indptr, const float* __restrict__ e_dist, double* __restrict__ radius,
                                    double* __restrict__ ratio, double* __restrict__ node_ratio, double* __restrict__ first_dist,
                                    unsigned long long* __restrict__ acc_u, double* __restrict__ acc_ratio) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    unsigned long long nomatch = 0, matches = 0;
    double sum = 0.;
    if (i < n) {
        double r = 0.;  // compute_max_edge, kgraph.rs:167-183
        for (uint64_t x = e_indptr[i]; x < e_indptr[i + 1]; x++) r = fmax(r, (double)e_dist[x]);
        radius[i] = r;
        const uint64_t b = indptr[i], e = indptr[i + 1];
        unsigned m = 0;
        double nr = 0.;
        for (uint64_t x = b; x < e; x++) {
            const double w = (double)tw[x];
            if (w <= r) m++;                 // :671-673
            const double q = w / r;          // :674-676
            ratio[x] = q;
            nr += q;
        }
        sum = nr;
        node_ratio[i] = nr / fmax(1.0, (double)(e - b));  // :679
        first_dist[i] = (double)tw[b];                     // :680
        nomatch = m == 0 ? 1ull : 0ull;
        matches = m;
    }
    // block reduction -> one atomic per block and quantity
    __shared__ unsigned long long s_a[256], s_b[256];
    __shared__ double s_c[256];
    s_a[threadIdx.x] = nomatch; s_b[threadIdx.x] = matches; s_c[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_a[threadIdx.x] += s_a[threadIdx.x + off];
            s_b[threadIdx.x] += s_b[threadIdx.x + off];
            s_c[threadIdx.x] += s_c[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(&acc_u[0], s_a[0]);
        atomicAdd(&acc_u[1], s_b[0]);
        atomicAdd(acc_ratio, s_c[0]);
    }
}

const double kQuantiles[6] = {0.05, 0.25, 0.5, 0.75, 0.85, 0.95};

// exact order statistics of a device array (sorted copy by rocPRIM radix sort)
void quantiles_of(const double* d_vals, uint64_t count, double* out6) {
    DevBuf<double> sorted(count);
    size_t tmp_bytes = 0;
    AE_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_vals, sorted.p, count, 0, 64, stream()));
    DevBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    AE_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, d_vals, sorted.p, count, 0, 64, stream()));
    for (int q = 0; q < 6; q++) {
        uint64_t rank = (uint64_t)(kQuantiles[q] * (double)count);
        if (rank >= count) rank = count - 1;
        AE_HIP(hipMemcpyAsync(&out6[q], sorted.p + rank, sizeof(double), hipMemcpyDeviceToHost, stream()));
    }
    sync();
}

}  // namespace

extern "C" {

int32_t ae_quality_estimate_from_edge_length(const ae_kgraph* g, const float* y, uint32_t dim, uint32_t nbng, ae_quality_report* rep,
                                             double* ratio_by_node, double* first_dist) {
    return guard([&] {
        require_device();
        if (!g || !y || !rep || dim == 0) fail(AE_ERR_INVALID_ARG, "null argument");
        if (nbng == 0 || nbng >= g->n) fail(AE_ERR_INVALID_ARG, "nbng must be in 1 .. nb_nodes - 1");
        const uint64_t n = g->n;
        // kNN graph of the embedded points (exact; the reference's is an hnsw_rs approximation)
        ae_kgraph* eg_raw = nullptr;
        int32_t rc = ae_kgraph_bruteforce_l2(y, n, dim, nbng, &eg_raw);
        if (rc != AE_OK) throw Error(rc, ae_last_error_message());
        std::unique_ptr<ae_kgraph> eg(eg_raw);
        DevBuf<float> dy;
        dy.alloc(n * dim);
        dy.upload(y, n * dim);
        DevBuf<float> tw(g->nnz);
        hipLaunchKernelGGL(transformed_edges_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, g->indptr.p, g->nbr.p, dy.p, dim, tw.p);
        check_launch("transformed_edges");
        DevBuf<double> radius(n), ratio(g->nnz), node_ratio(n), first(n), acc_ratio(1);
        DevBuf<unsigned long long> acc_u(2);
        acc_u.zero();
        acc_ratio.zero();
        hipLaunchKernelGGL(quality_node_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, g->indptr.p, tw.p, eg->indptr.p,
                           eg->dist.p, radius.p, ratio.p, node_ratio.p, first.p, acc_u.p, acc_ratio.p);
        check_launch("quality_node");
        unsigned long long hu[2];
        double hr = 0.;
        acc_u.download(hu, 2);
        acc_ratio.download(&hr, 1);
        memset(rep, 0, sizeof(*rep));
        rep->nb_nodes = n;
        rep->nb_edges = g->nnz;
        rep->kgraph_nbng = g->max_nbng;
        rep->nbng = nbng;
        rep->nb_without_match = hu[0];
        rep->mean_nbmatch = (n > hu[0]) ? (double)hu[1] / (double)(n - hu[0]) : 0.;  // :688-689
        quantiles_of(radius.p, n, rep->radii_quantiles);
        quantiles_of(ratio.p, g->nnz, rep->ratio_quantiles);
        rep->median_ratio = rep->ratio_quantiles[2];
        rep->mean_ratio = hr / (double)g->nnz;  // :677-678, :730
        rep->quality = 0.;                      // the reference returns Some(0.) (:630, :751)
        if (ratio_by_node) node_ratio.download(ratio_by_node, n);
        if (first_dist) first.download(first_dist, n);
    });
}

}  // extern "C"
