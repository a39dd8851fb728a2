// kernels_match.hip -- alignments -> compatibility on the device.
//
// What splicing_matchIso / splicing_matchIso_paired + splicing_genomic_to_iso compute per read and
// isoform (solve.c:8-108, 141-218; gff.c:1041-1084), emitted directly in the packed form the
// sampler consumes: single-end one u32 compatibility mask per read (two words, low then high, from 33 isoforms on), paired-end K u16 fragment-length
// indices per pair (0xFFFF = incompatible or outside the fragment distribution).  One thread per
// read (pair); the gene's exon tables and the host-parsed CIGAR blocks are read-only inputs.
// Integer work throughout: bit-exact against host.cpp's match_iso[_paired] and the CPU checker.
#include <hip/hip_runtime.h>

#include "device.hpp"

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
  uint64_t m[2] = {0u, 0u};
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
        if (read_fits(xs, xe, xi[k], xi[k + 1], p, o, nops)) m[t] |= 1ull << k;
  }
  if (!paired) {
    if (E.K <= 32) masks[E.out_off + r] = static_cast<uint32_t>(m[0]);
    else { masks[E.out_off + 2 * r] = static_cast<uint32_t>(m[0]); masks[E.out_off + 2 * r + 1] = static_cast<uint32_t>(m[0] >> 32); }
    return;
  }
  const uint64_t both = m[0] & m[1];
  uint16_t *row = frags + (static_cast<size_t>(E.out_off) + r) * E.K;
  for (int k = 0; k < E.K; k++) {
    uint16_t f = FRAG_NONE;
    if ((both >> k) & 1ull) {
      const int frag = genomic_to_iso(xs, xe, xi[k], xi[k + 1], p0[1]) -
                       genomic_to_iso(xs, xe, xi[k], xi[k + 1], p0[0]) + readLength;
      if (frag >= frag_start && frag < frag_start + il) f = static_cast<uint16_t>(frag - frag_start);
    }
    row[k] = f;
  }
}

}  // namespace miso
