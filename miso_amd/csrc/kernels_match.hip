// kernels_match.hip -- alignments -> compatibility on the device.
//
// What splicing_matchIso / splicing_matchIso_paired + splicing_genomic_to_iso compute per read and
// isoform (solve.c:8-108, 141-218; gff.c:1041-1084), emitted directly in the packed form the
// sampler consumes: single-end (K + 31) / 32 u32 compatibility mask words per read, paired-end K u16 fragment-length
// indices per pair (0xFFFF = incompatible or outside the fragment distribution).  One thread per
// read (pair); the gene's exon tables and the host-parsed CIGAR blocks are read-only inputs.
// Integer work throughout: bit-exact against host.cpp's match_iso[_paired] and the CPU checker.
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "miso_amd.h"

namespace miso {

namespace {

// solve.c:36-100 for one read and one isoform: walk the CIGAR blocks along the isoform's exons
__device__ __forceinline__ bool read_fits(const int *exstart, const int *exend, int ex, int last, int p,
                                          const int *ops, int nops) {
  while (ex < last && (p < exstart[ex] || exend[ex] < p)) ex++;
  if (ex >= last) return false;
  for (int c = 0; c < nops; c++) {
    const int o = ops[c];
    if (o > 0) {
      if (p + o - 1 > exend[ex]) return false;
      p += o;
    } else {
      if (p != exend[ex] + 1) return false;
      p -= o;
      ex++;
      if (ex >= last || p != exstart[ex]) return false;
    }
  }
  return true;
}

// gff.c:1041-1084: 1-based position of genomic coordinate p inside the isoform, -1 outside its exons
__device__ __forceinline__ int genomic_to_iso(const int *exstart, const int *exend, int ex, int last, int p) {
  int before = 0;
  for (; ex < last; ex++) {
    if (exend[ex] < p) { before += exend[ex] - exstart[ex] + 1; continue; }
    if (exstart[ex] <= p) return before + (p - exstart[ex]) + 1;
    return -1;
  }
  return -1;
}

}  // namespace

__global__ __launch_bounds__(256) void match_kernel(const MatchEvent *events, const int2 *blocks,
                                                    const int *exidx, const int *exstart, const int *exend,
                                                    const int *pos, const int *opidx, const int *ops,
                                                    const int *rlen, int readLength, int overHang,
                                                    int paired, int frag_start, int il, uint32_t *masks,
                                                    uint16_t *frags) {
  const int2 bt = blocks[blockIdx.x];
  const MatchEvent E = events[bt.x];
  const int r = bt.y + static_cast<int>(threadIdx.x);   // read (pair) of the event
  if (r >= E.n_reads) return;
  const int *xi = exidx + E.exidx_off;                  // K + 1 offsets into the event's exon arrays
  const int *xs = exstart + E.ex_off, *xe = exend + E.ex_off;
  const int mates = paired ? 2 : 1;
  constexpr int MW = (MISO_MAX_ISOFORMS + 63) / 64;     // mask words of a gene with the most isoforms
  uint64_t m[2][MW];
  for (int t = 0; t < 2; t++) for (int w = 0; w < MW; w++) m[t][w] = 0u;
  int p0[2] = {0, 0};
  for (int t = 0; t < mates; t++) {
    const int g = E.read_off + mates * r + t;           // global read index
    const int *o = ops + opidx[g];
    const int nops = opidx[g + 1] - opidx[g];
    const int p = pos[g];
    p0[t] = p;
    const bool usable = rlen[g] >= readLength && nops > 0 && o[0] >= overHang && o[nops - 1] >= overHang;
    if (usable)
      for (int k = 0; k < E.K; k++)
        if (read_fits(xs, xe, xi[k], xi[k + 1], p, o, nops)) {
#pragma unroll
          for (int w = 0; w < MW; w++) if (w == (k >> 6)) m[t][w] |= 1ull << (k & 63);   // (static indices: the words stay in registers)
        }
  }
  if (!paired) {   // (K + 31) / 32 words per read, read-major (runtime.hip resolve_pending)
    const int W32 = (E.K + 31) >> 5;
#pragma unroll
    for (int w = 0; w < 2 * MW; w++)
      if (w < W32) masks[E.out_off + W32 * r + w] = static_cast<uint32_t>(m[0][w >> 1] >> (32 * (w & 1)));
    return;
  }
  uint16_t *row = frags + (static_cast<size_t>(E.out_off) + r) * E.K;
  for (int k = 0; k < E.K; k++) {
    uint16_t f = FRAG_NONE;
    uint64_t both = 0;
#pragma unroll
    for (int w = 0; w < MW; w++) if (w == (k >> 6)) both = m[0][w] & m[1][w];
    if ((both >> (k & 63)) & 1ull) {
      const int frag = genomic_to_iso(xs, xe, xi[k], xi[k + 1], p0[1]) -
                       genomic_to_iso(xs, xe, xi[k], xi[k + 1], p0[0]) + readLength;
      if (frag >= frag_start && frag < frag_start + il) f = static_cast<uint16_t>(frag - frag_start);
    }
    row[k] = f;
  }
}

}  // namespace miso
