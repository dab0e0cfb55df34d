// projector.hpp -- the K sparse ternary random-projection matrices of ranM()
// (R/ranM.R:11-33), drawn with R's RNG stream (on the GPU; a host build exists for cross-checks) and kept on the device as
// gene-major packed row lists: for gene g the entries (k*p + c, sign) of every
// projector in the group, 16 bits each (bit 15 = negative).
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "common.hpp"

namespace sharp {

struct ProjectorGroup {
    int k0 = 0, kcount = 0;      // projectors [k0, k0+kcount)
    int ncomp = 0;               // kcount * p output components
    long long nnz = 0;
    double mean_len = 0;         // mean entries per gene row
    int max_len = 0;
    int gw = 16;                 // lanes per gene in the scatter pass; a segment holds 4*gw entries
    long long nseg = 0;          // total segments (m main + overflow)
    // Gene g's row list lives in segment g (fixed stride, so its address needs no lookup); the few
    // genes with more than 4*gw entries continue in overflow segments listed in (ovf_gene, ovf_info).
    // A segment holds logical entries in lane-major order: position 4*lane + q  <->  entry q*gw + lane,
    // so one 8-byte load per lane fetches that lane's four entries; unused positions hold 0xFFFF.
    DevBuf<uint16_t> ent;        // nseg * 4 * gw
    int novf = 0;
    DevBuf<uint32_t> ovf_gene;   // sorted genes that overflow
    DevBuf<uint2> ovf_info;      // x = first overflow segment, y = number of overflow segments
};

struct Projector {
    int m = 0, p = 0, K = 0;
    double val = 0;              // |R[g,c]| = sqrt(s), s = sqrt(m)
    // host copies, one CSR per projector: col >= 0 -> +, ~col (negative) -> -
    std::vector<std::vector<uint32_t>> h_rowptr;
    std::vector<std::vector<int32_t>> h_ent;
    std::vector<ProjectorGroup> groups;
    // device build: per projector the list of non-zero elements (index r*p + c of the byrow fill, bit 31 = negative)
    bool device_built = false;
    unsigned int hit_cap = 0;
    DevBuf<uint32_t> d_hits;
    DevBuf<unsigned int> d_nhits;
    std::vector<unsigned int> h_nhits;
    long long nnz_total() const { long long s = 0; for (auto &g : groups) s += g.nnz; return s; }
};

// max output components one scatter launch can hold in LDS / address with 15 bits
constexpr int kMaxCompPerGroup = 12288;

std::shared_ptr<Projector> build_projector(int m, int p, int K, const double *seeds);
// E[row_map ? row_map[cell] : cell][k*p + c]; d_row_map: optional device array of n output rows (rp.hip)
void project_dev(const Projector &pr, const float *dX, int m, int n, long long ld, int log_flag, double *dE, long long ldE,
                 const int *d_row_map);
int register_projector(std::shared_ptr<Projector> pr);
std::shared_ptr<Projector> get_projector(int handle);
void drop_projector(int handle);

}  // namespace sharp
