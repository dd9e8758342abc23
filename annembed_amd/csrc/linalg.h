// linalg.h -- device linear-algebra building blocks shared by svd.hip and dmap.hip.
// All panels are tall-skinny row-major f32 (rows x l, l <= kMaxL); small l x l matrices are f64.
#pragma once
#include "objects.h"

namespace ae {

constexpr int kMaxL = 64;  // widest panel supported (reference uses rank 20; tests up to 28/32)

// out[count] ~ N(0,1): element e of the (seed, tag) Philox stream (Box-Muller), row-major fill order
void gaussian_fill_device(float* d_out, uint64_t count, uint64_t seed, uint32_t tag);

// Y[m x l] = A * X[n x l]   (A = CSR or dense MatRepr)
void mat_mul_panel(const ae_matrepr& a, const float* d_x, float* d_y, uint32_t l);
// Y[n x l] = A^T * X[m x l] (builds and caches the transpose CSR on first use unless a.symmetric)
void mat_t_mul_panel(ae_matrepr& a, const float* d_x, float* d_y, uint32_t l);
// builds a.transpose (CSR of A^T) on device
void build_transpose(ae_matrepr& a);

// G[l x l] (f64, device) = Y^T Y
void gram_panel(const float* d_y, uint64_t rows, uint32_t l, double* d_g);
// Y <- Y * M  (M l x lout f64 device, row-major); in place allowed when lout <= l
void apply_panel(const float* d_y, uint64_t rows, uint32_t l, const double* d_m, uint32_t lout, float* d_out);

// Householder-QR stand-in (do_qr, svdapprox.rs:998-1013): orthonormalise the columns of Y in place.
// Two passes of Gram + symmetric eigendecomposition + scaling (SVQB); rank-deficient directions
// become zero columns.  d_work: >= 3*l*l doubles.
void orthonormalize_panel(float* d_y, uint64_t rows, uint32_t l, double* d_work);

// Optimistic single-pass CholeskyQR on an f64 (MFMA) Gram: no host round trip; a failed pivot leaves Y untouched and
// raises a sticky flag that orthonormalize_fast_failed() reports (and clears) -- the caller then redoes its
// computation with orthonormalize_panel (eigen route for rank-deficient panels).
void orthonormalize_panel_fast(float* d_y, uint64_t rows, uint32_t l);
bool orthonormalize_fast_failed();

// eigendecomposition of a symmetric l x l f64 matrix on device (cyclic Jacobi, one workgroup):
// evals[l] descending, evecs[l x l] row-major with eigenvectors in columns.
void jacobi_eigh_device(const double* d_g, uint32_t l, double* d_evals, double* d_evecs);

// Summation order of the three sums below: two-level f64 tree reductions (deterministic, microseconds) or the reference's f32 orders as
// single-lane chains (bit parity of the dmap initialisation with the oracle: 40-100 ms each at 11 M nodes).  A TreeSums scope alive on
// this thread decides (true: trees, false: the reference order -- the embedder and EntropyOptim::new open one: the reference order exactly
// when the CE mode that follows is the bit-exact one); without a scope the process-wide default does (ae_set_summation_order: trees unless
// the caller asked for the reference order).
struct TreeSums {
    explicit TreeSums(bool on);
    ~TreeSums();
    int prev;
};
bool tree_sums();
int tree_sums_scope();   // -1: no scope on this thread, 0 / 1: the innermost scope's choice
void set_tree_sums_default(bool on);

// exact sequential f32 sum (the reference's iter().sum::<f32>() order) of d_x[0..n) with stride
float seq_sum_f32(const float* d_x, uint64_t n, uint64_t stride = 1);
// the dim column sums of a row-major n x dim array, each in row order (dim sequential chains in one pass)
void seq_sum_cols_f32(const float* d_x, uint64_t n, uint32_t dim, float* host_out);
// f32 sum in the order of ndarray's Array1::sum() (eight interleaved accumulators, see svd.hip)
float ndarray_sum_f32(const float* d_x, uint64_t n);

// exact L2 kNN rows (n x k) on the matrix cores, knn.hip; returns the number of rows recomputed by the brute-force fallback
uint64_t knn_mfma(const float* d_x, uint64_t n, uint64_t dim, uint32_t k, uint32_t* d_nbr, float* d_dist);
void bruteforce_knn_rows(const float* d_x, uint64_t n, uint64_t dim, uint32_t k, const uint32_t* d_rows, uint64_t nrows,
                         uint32_t* d_nbr, float* d_dist);
void bruteforce_knn_rect(const float* d_x, uint64_t dim, uint32_t k, const uint32_t* d_qrows, uint64_t q_begin, const uint32_t* d_places, uint64_t nrows,
                         uint64_t p_begin, uint64_t p_end, uint32_t* d_nbr, float* d_dist, bool out_by_list, bool raw, const uint32_t* d_orig);
// exact GLOBAL kNN of points sorted into groups (clusters): own group first, then only the groups a triangle-inequality bound cannot exclude
void knn_grouped(const float* d_x, uint64_t n, uint64_t dim, uint32_t k, const uint64_t* bounds, uint32_t groups, uint32_t* d_nbr, float* d_dist, uint64_t* stats);
}  // namespace ae
