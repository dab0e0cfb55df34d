// Micro-benchmark: ONE rebuild round of hclust_rnn_kernel's plain rows, as the kernel does it (na = 2000 old clusters, np = 200 pairs: 400 of the
// old columns belong to pair members and are parked in the wave's LDS stage, the other 1600 go to their new column; then one Lance-Williams value
// per merged column), with the pieces switchable -- which of them costs the streaming rate that the bare sweep reaches (tools/micro/sweep_mlp.hip)?
//   F bit 0: pair members' entries parked in the LDS stage (else: skipped)      bit 1: the Lance-Williams loop over the merged columns
//     bit 2: rows handed out by an LDS counter (else: a static stride)          bit 3: the DPP wave reductions + result writes per row
//     bit 4: the surviving entries keep their OLD column (aligned stores with holes, no compaction)
//     bit 5: the surviving entries go through a per-wave LDS window and leave as aligned 64-entry stores (a write combiner)
//     bit 6: the row's tail as one more full batch of 8 with clamped loads (else: batches of 4, 2, 1)
// Build: hipcc --offload-arch=gfx950 -O3 sweep_real.hip -o sweep_real
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
constexpr int NA = 2000, NP = 200, NS = NA - 2 * NP, NB = NS + NP, LD = 2048;
__device__ __forceinline__ double lw(int method, double d1, double d2, double d12, double mi, double mj, double mk) {
  switch (method) {
    case 1: case 8: { double dn = (mi + mk) * d1 + (mj + mk) * d2 - mk * d12; return dn / (mi + mj + mk); }
    case 2: return d1 < d2 ? d1 : d2;
    case 3: return d1 > d2 ? d1 : d2;
    case 4: return (mi * d1 + mj * d2) / (mi + mj);
    default: return (d1 + d2) / 2;
  }
}
struct Best { double v; int i; };
__device__ __forceinline__ Best wave_best(Best x) {
  for (int o = 32; o > 0; o >>= 1) {
    const double v = __shfl_xor(x.v, o); const int i = __shfl_xor(x.i, o);
    if (v < x.v || (v == x.v && i < x.i)) { x.v = v; x.i = i; }
  }
  return x;
}
template <int F>
__global__ __launch_bounds__(1024) void round_k(const double *Dsrc0, double *Ddst0, int passes, int method, double *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  double *dnn = reinterpret_cast<double *>(sm);                  // [NA]
  unsigned short *csz = reinterpret_cast<unsigned short *>(dnn + NA);   // [NA]
  unsigned short *colmap = csz + NA, *oldidx = colmap + NA, *partner = oldidx + NA, *nn = partner + NA;
  int *ctl = reinterpret_cast<int *>(nn + NA);
  double *stage = reinterpret_cast<double *>(ctl + 16);         // 16 waves x 2 rows x 2 NP
  double *window = stage + 16 * 2 * NP;                         // 16 waves x 2 rows x 128 entries (the two rows of a wave SHARE a stage here: LDS room; timing only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // every fifth column a pair member: columns 5k+1 and 5k+3 of k < NP/... (400 members)
  if (tid == 0) {
    int nsn = 0, rank = 0;
    for (int j = 0; j < NA; ++j) {
      const bool mem = (j % 5 == 1 || j % 5 == 3) && rank < 2 * NP;
      if (mem) { colmap[j] = (unsigned short)(0x8000 | rank); if ((rank & 1) == 0) oldidx[NS + rank / 2] = (unsigned short)j; else partner[oldidx[NS + rank / 2]] = (unsigned short)j; ++rank; }
      else { colmap[j] = (unsigned short)nsn; oldidx[nsn++] = (unsigned short)j; }
      csz[j] = 1; dnn[j] = 0.5;
    }
    ctl[2] = 0;
  }
  __syncthreads();
  const double *Dsrc = Dsrc0 + (size_t)blockIdx.x * NA * LD;
  double *Ddst = Ddst0 + (size_t)blockIdx.x * NA * LD;
  double acc = 0;
  for (int p = 0; p < passes; ++p) {
    int qs = wave * 2;
    for (;;) {
      int q;
      if (F & 4) { q = 0; if (lane == 0) q = atomicAdd(&ctl[2], 2); q = __builtin_amdgcn_readfirstlane(q); }
      else { q = qs; qs += 32; }
      if (q >= NS) break;
      const double *r[2]; double *w[2], *stg[2]; double mn[2], sc[2]; int ix[2], ao[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ao[t] = __builtin_amdgcn_readfirstlane(oldidx[q + t]);
        r[t] = Dsrc + (size_t)ao[t] * LD; w[t] = Ddst + (size_t)(q + t) * LD;
        stg[t] = stage + (size_t)wave * 2 * NP; mn[t] = 1e300; sc[t] = 1e300; ix[t] = 0x7fffffff;
      }
      int j0 = lane;
      int filled = 0;
      double *win[2] = {window + ((size_t)wave * 2) * 128, window + ((size_t)wave * 2 + 1) * 128};
      auto pass = [&](auto U_) {
        constexpr int U = decltype(U_)::value;
        for (; ((F & 64) && U == 8) ? (j0 - lane < NA) : (j0 + 64 * (U - 1) < NA); j0 += 64 * U) {
          unsigned cm[U]; double x[2][U];
#pragma unroll
          for (int u = 0; u < U; ++u) { const int j = j0 + 64 * u; cm[u] = j < NA ? colmap[j] : 0x8000u; }   // (beyond the row: treated like a pair member, parked nowhere)
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) { const int j = j0 + 64 * u; x[t][u] = r[t][((F & 64) && j >= NA) ? NA - 1 : j]; }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (cm[u] & 0x8000u) {
              if ((F & 1) && (!(F & 64) || j0 + 64 * u < NA)) {
#pragma unroll
                for (int t = 0; t < 2; ++t) stg[t][cm[u] & 0x7fffu] = x[t][u];
              }
            } else {
              const int B = (F & 16) ? j0 + 64 * u : (int)cm[u];
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                const double v = x[t][u];
                if (F & 32) win[t][B & 127] = v; else w[t][B] = v;
                sc[t] = fmin(sc[t], fmax(mn[t], v)); if (v < mn[t]) { mn[t] = v; ix[t] = B; }
              }
            }
            if (F & 32) {
              // survivors of this step occupy new columns [lo, hi): every aligned block of 64 that is now complete leaves as one full store
              const unsigned long long live = __ballot(!(cm[u] & 0x8000u));
              const int cnt = __popcll(live);
              const int hi = filled + cnt;                       // new columns below hi are in the window
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              if ((hi >> 6) > (filled >> 6)) {
                const int base = (filled >> 6) << 6;
#pragma unroll
                for (int t = 0; t < 2; ++t) w[t][base + lane] = win[t][(base + lane) & 127];
              }
              filled = hi;
            }
          }
          if (U == 1) break;
        }
      };
      pass(std::integral_constant<int, 8>()); pass(std::integral_constant<int, 4>()); pass(std::integral_constant<int, 2>()); pass(std::integral_constant<int, 1>());
      if ((F & 32) && (filled & 63)) {                           // the last, partial block
        const int base = (filled >> 6) << 6;
        if (base + lane < filled) {
#pragma unroll
          for (int t = 0; t < 2; ++t) w[t][base + lane] = win[t][(base + lane) & 127];
        }
      }
      if (F & 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double nr_[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) nr_[t] = csz[ao[t]];
        for (int B = NS + lane; B < NB; B += 64) {
          const int rk = B - NS; const int k1 = oldidx[B] & 0x7fff, l1 = partner[k1];
          const double nk_ = csz[k1], nl_ = csz[l1], hQ = dnn[k1];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const double v = lw(method, stg[t][2 * rk], stg[t][2 * rk + 1], hQ, nk_, nl_, nr_[t]);
            w[t][B] = v; sc[t] = fmin(sc[t], fmax(mn[t], v));
            if (v < mn[t] || (v == mn[t] && B < ix[t])) { mn[t] = v; ix[t] = B; }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (F & 8) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          Best g; g.v = mn[t]; g.i = ix[t]; g = wave_best(g);
          const int tie = __ballot(sc[t] == g.v) != 0ull ? 1 : 0;
          if (lane == 0) dnn[NA - 1 - (q + t) % 64] = g.v + tie;
          if (lane == 1) nn[q + t] = (unsigned short)g.i;
        }
      } else acc += mn[0] + mn[1] + sc[0] + sc[1] + ix[0] + ix[1];
    }
    __syncthreads();
    if (tid == 0) ctl[2] = 0;
    __syncthreads();
  }
  if (acc == 123.456) sink[blockIdx.x] = acc;
}
template <int F> void run(const char *name, int wgs, int method, double *a, double *b, double *sink) {
  const int passes = 4;
  const size_t lds = NA * 8 + 6 * NA * 2 + 64 + 16 * 2 * NP * 8 + 16 * 2 * 128 * 8;
  auto k = round_k<F>;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), lds, 0, a, b, 1, method, sink); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), lds, 0, a, b, passes, method, sink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)wgs * passes * 8.0 * ((double)NS * NA + (double)NS * NB);     // plain rows: read na, write nb entries each
  printf("%-46s wgs=%-4d %8.3f ms  %6.2f TB/s (read+write)  %6.1f GB/s per workgroup\n", name, wgs, ms, bytes / ms / 1e9, bytes / ms / 1e6 / wgs);
}
int main() {
  const size_t n = (size_t)256 * NA * LD;
  double *a, *b, *sink; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&sink, 8 * 256));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
  for (int wgs : {125, 188}) {
    run<0>("sweep only", wgs, 1, a, b, sink);
    run<1>("+ stage", wgs, 1, a, b, sink);
    run<3>("+ stage + Lance-Williams (ward)", wgs, 1, a, b, sink);
    run<3>("+ stage + Lance-Williams (average)", wgs, 4, a, b, sink);
    run<7>("+ stage + LW (ward) + dynamic rows", wgs, 1, a, b, sink);
    run<15>("+ stage + LW (ward) + dynamic + reductions", wgs, 1, a, b, sink);
    run<12>("sweep + dynamic + reductions", wgs, 1, a, b, sink);
    run<16>("sweep only, survivors keep their old column", wgs, 1, a, b, sink);
    run<32>("sweep only, survivors through an LDS window", wgs, 1, a, b, sink);
    run<47>("everything, survivors through an LDS window", wgs, 1, a, b, sink);
    run<64>("sweep only, tail as a clamped batch of 8", wgs, 1, a, b, sink);
    run<79>("everything, tail as a clamped batch of 8", wgs, 1, a, b, sink);
    run<111>("everything, LDS window + clamped tail", wgs, 1, a, b, sink);
  }
  return 0;
}
