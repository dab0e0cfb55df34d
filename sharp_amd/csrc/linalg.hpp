// linalg.hpp -- batched fp64 building blocks shared by the clustering stages:
// row centring/normalisation and a TN GEMM on the f64 MFMA (v_mfma_f64_16x16x4_f64).
#pragma once
#include "common.hpp"

namespace sharp {

// One GEMM of a batch:  C[M x N] (row-major, ldc) = sum_k At[k][0..M) * Bt[k][0..N)
// At is K x M row-major (lda), Bt is K x N row-major (ldb): both operands are "k-major", so tile
// staging reads contiguous rows.  epilogue: 0 = store v ; 1 = store 1 - clamp(v, -1, 1), zero diagonal ; 2 = clamp(v), unit diagonal
// symmetric: At == Bt and M == N -> only tiles on/above the diagonal are computed and mirrored.
struct GemmTask {
    const double *At;
    const double *Bt;
    double *C;
    int M, N, K;
    long long lda, ldb, ldc;
    int epilogue;
    int symmetric;
    int fast;   // 1: operands are zero-padded to tile multiples (K rows to a multiple of 16, M/N to 128, lda/ldb % 4 == 0): unchecked 16-byte loads,
                //    register-prefetched K tiles
    double *nn = nullptr;   // fast symmetric distance tasks: nn[slot * ldc + row] = the smallest off-diagonal entry of the row over column tile `slot`
                            // (128 columns; every slot of every row is written exactly once: by the tile's row reduction or, below the diagonal,
                            // by the mirrored tile's column reduction)
    int nn_square = 0;      // the minima are of v * v (ward.D2 agglomerates squared distances)
};

// Launch over a device-resident array of `count` GemmTask (max_M/max_N size the grid).
// fast: 128x128 MFMA tiles (needs 16-byte aligned panels padded to 4 columns); symmetric: every task of the batch has
// symmetric = 1, so only upper-triangle tiles are launched
void gemm_tn_f64_batched(const GemmTask *d_tasks, int count, int max_M, int max_N, const char *timer_name, bool fast = false,
                         bool symmetric = false);

// Row preparation for one matrix of a batch (R/get_opt_hclust.R:66-74 + the centring inside cor()):
//   mode 0 (feature rows): z = (x - mean)/sd(p-1)  [t(scale(t(mat)))], re-centre as cor() does,
//                          u = c / ||c||  -> Cr (n x p row-major), Ct (p x nld), nrm = 1
//   mode 1 (symmetric similarity S): c = x - mean (rows of S as features for get_CH),
//                          Cr, Ct = centred rows, nrm = ||c||;  D = 1 - S is written as well.
struct RowPrepTask {
    const double *src;   // n x p row-major, leading dimension lds
    long long lds;
    int n, p, nld;
    int p_pad;           // rows of Ct (>= p, multiple of 16; the extra rows are zero-filled)
    int mode;
    double *Cr;          // n x p (ld p)
    double *Ct;          // p x nld
    double *nrm;         // n
    double *D;           // nld x nld (mode 1 only)
};
void row_prep_batched(const RowPrepTask *d_tasks, int count, int max_n, int max_p, bool all_feature_rows = false);


#ifdef SHARP_LAB
// (lab builds only) The correlation-distance matrix of one task on the integer matrix cores (tools/lab/gemm_i8.hip): D = 1 - clamp(U U^T), zero diagonal, from the
// unit rows Cr (n x p row-major, ld p) cut into kDistI8Slices 7-bit digits per entry.
constexpr int kDistI8Slices = 7;
struct DistI8Task {
    const double *Cr;    // n x p, ld p
    double *D;           // nld x nld
    signed char *sl;     // the digits: dist_i8_slice_bytes(nld, p) bytes
    double *scale;       // nld: the power of two of each row (0: padding row)
    int n, p, nld, ksteps;   // ksteps = ceil(p / 32)
};
size_t dist_i8_slice_bytes(int nld, int p);
void dist_i8_batched(const DistI8Task *d_tasks, int count, int max_n);
void dist_i8_slices(const DistI8Task *d_tasks, int count, int max_n);     // rows -> digits
void dist_i8_products(const DistI8Task *d_tasks, int count, int max_n);   // digits -> D
#endif

}  // namespace sharp
