// upload.hpp -- host matrices (what R hands over: fp64, genes x cells, column-major; or a dgCMatrix's three slots) -> expression
// blocks in HBM.  A block is stored as fp32 when every value survives the round trip through float (counts, UMI data: half the bytes
// of the one pass over X) and as fp64 otherwise (TPM / CPM-like doubles), so that log2(X + 1) and the projection see exactly the
// numbers the reference computes with (R/SHARP.R:110-117,343-345).  SHARP_X_STORAGE = fp32 | fp64 forces the choice.
#pragma once
#include "common.hpp"

namespace sharp {

struct HostBlock {               // the resident copy of a host matrix; kept between calls like every other workspace
    DevBuf<float> f;
    DevBuf<double> d;
    bool f64 = false;
    long long ld = 0;            // column stride in elements (m rounded up to 16 bytes)
    double max_abs = 0;          // largest |value| of an fp64 block (decides the fixed-point scale of the log-mode accumulation)
    XRef ref() const {
        XRef r = f64 ? XRef(d.p) : XRef(f.p);
        if (f64 && max_abs > 3.4028234663852886e38) r.log_fix_bits = 41;
        return r;
    }
    void release() { f.release(); d.release(); }
};

// the resident copies an upload slot rotates through (sharp_SHARP_unlimited_multi: later blocks are uploaded while earlier ones are clustered):
// three (two for blocks beyond 8 GB); four with SHARP_HOST_GROUP >= 2, when the compute thread takes several arrived blocks together
constexpr int kHostRing = 4;
struct HostBlockPair { HostBlock hb[kHostRing]; };

// X: m x n column-major doubles, column stride ld >= m (pageable memory).  Threaded narrowing / copying into pinned slabs, DMA'd
// while the next slab is prepared.  The wire carries the narrowest type that holds every value exactly: unsigned 16-bit integers (counts),
// else floats (both stored as fp32), else doubles.
void upload_block(const double *X, int m, long long n, long long ld, HostBlock &hb);
// canonical CSC (colptr n + 1 entries, 0-based row indices): only the non-zeros cross PCIe -- a u16 / int32 row index and a u16 / float /
// double value each: 4 bytes per non-zero for counts over at most 65 536 genes, against the 12 R holds -- the dense block is built on the device
void upload_block_csc(const int *colptr, const int *rowidx, const double *val, int m, long long n, HostBlock &hb);
// the same into a caller-owned fp32 device block (sharp_csc_to_dense_dev: the *_dev entry points take fp32): values are narrowed
void upload_csc_into_f32(const int *colptr, const int *rowidx, const double *val, int m, long long n, float *dX, long long ld);
// a packed CSC block already in device memory (colptr: n + 1 int64; row indices of 16 or 32 bits; values u16 / float / double) -> the dense
// block dX (fp32, or fp64 when dx_f64): what a block file of the compact format holds, expanded where it lands
void expand_packed_csc_dev(const long long *d_colptr, const void *d_idx, int idx_bits, const void *d_val, int val_bits, int m, long long n,
                           void *dX, long long ld, bool dx_f64);
void upload_release_staging();   // the pinned staging buffers (sharp_trim)
int upload_last_storage();       // 32 or 64: what the most recent upload_block / upload_block_csc chose (0: none yet)
int upload_last_wire();          // 16, 32 or 64: the width of a value of that block on PCIe (u16 counts / float / double)

}  // namespace sharp
