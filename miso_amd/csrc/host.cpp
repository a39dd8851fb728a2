// host.cpp -- gene model, alignment matching and event packing (host side of the path).
//
// Reference behaviour followed (paths under /root/reference/pysplicing/):
//   make_gene          src/simulator.c:9-66, src/gff.c:583-657, 728-777
//   parse_cigars       src/solve.c:220-306
//   match_iso          src/solve.c:8-108
//   normal_fragment    src/simulator.c:198-219 + src/util.c:17-32, normalised miso_paired.c:303-307
//   match_iso_paired   src/solve.c:141-218, src/gff.c:855-898, 1041-1084
//   pack_event         src/miso.c:762-786, src/miso_paired.c:386-419 (set-up quantities)
#include "host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>

#include "device.hpp"
#include "miso_philox.h"

namespace miso {

const char *strerror_code(int code) {
  switch (code) {
  case MISO_SUCCESS: return "No error";
  case MISO_FAILURE: return "Failed";
  case MISO_ENOMEM: return "Out of memory";
  case MISO_EINVAL: return "Invalid value";
  case MISO_UNIMPLEMENTED: return "Unimplemented function call";
  case MISO_EINTERNAL: return "Internal error, likely a bug";
  case MISO_ENODEVICE: return "No usable HIP device";
  default: return "Unknown error";
  }
}

Error::Error(int c, const std::string &reason, const char *file, int line) : code(c) {
  const char *base = std::strrchr(file, '/');
  text = std::string("Error at ") + (base ? base + 1 : file) + ":" + std::to_string(line) + ": " +
         reason + ", " + strerror_code(c);
}

Gene make_gene(const int *exons, int n_exons, const int *isoforms, int n_flat, const char *id,
               const char *seqid, const char *source, int strand) {
  if (!exons || !isoforms || n_exons <= 0 || n_flat <= 0)
    MISO_FAIL(MISO_EINVAL, "Gene needs at least one exon and one isoform");
  if (isoforms[n_flat - 1] >= 0)
    MISO_FAIL(MISO_EINVAL, "Isoform list must be terminated by -1");
  Gene g;
  g.id = id ? id : "insilicogene";
  g.seqid = seqid ? seqid : "seq1";
  g.source = source ? source : "protein_coding";
  g.strand = strand;
  g.exidx.push_back(0);
  int len = 0, nex = 0;
  for (int i = 0; i < n_flat; i++) {
    const int e = isoforms[i];
    if (e < 0) {
      g.exidx.push_back(static_cast<int>(g.exstart.size()));
      g.isolen.push_back(len);
      g.noexons.push_back(nex);
      len = nex = 0;
      continue;
    }
    if (e >= n_exons) MISO_FAIL(MISO_EINVAL, "Isoform refers to an exon that does not exist");
    g.exstart.push_back(exons[2 * e]);
    g.exend.push_back(exons[2 * e + 1]);
    len += exons[2 * e + 1] - exons[2 * e] + 1;
    nex++;
  }
  g.K = static_cast<int>(g.isolen.size());
  return g;
}

CigarTable parse_cigars(const char *const *cigar, int n, int maxReadLength) {
  CigarTable t;
  t.idx.reserve(n + 1);
  t.len.reserve(n);
  for (int i = 0; i < n; i++) {
    const char *s = cigar[i];
    int phase = 0;  // 0: leading clips allowed, 1: body, 2: trailing clips only
    int covered = 0;
    t.idx.push_back(static_cast<int>(t.ops.size()));
    while (s && *s) {
      char *end = nullptr;
      long l = std::strtol(s, &end, 10);
      const char op = *end;
      const bool clip = (op == 'S' || op == 'H');
      if (phase == 0 && !clip) phase = 1;
      else if (phase == 1 && clip) phase = 2;
      else if (phase == 2 && !clip)
        MISO_FAIL(MISO_EINVAL,
                  "Bad CIGAR string: `S' and 'H' may appear only at the beginning and the end");
      switch (op) {
      case 'M': case '=': case 'X': case 'S': case 'H': case 'D':
        if (maxReadLength > 0 && covered + l > maxReadLength) l = maxReadLength - covered;
        covered += static_cast<int>(l);
        t.ops.push_back(static_cast<int>(l));
        break;
      case 'N': t.ops.push_back(static_cast<int>(-l)); break;
      case 'I': break;
      default:
        MISO_FAIL(MISO_EINVAL, "Unsupported CIGAR string (`MNSHDI=X' are supported)");
      }
      s = end + 1;
    }
    t.len.push_back(covered);
  }
  t.idx.push_back(static_cast<int>(t.ops.size()));
  return t;
}

void match_iso(const Gene &g, const int *pos, const char *const *cigar, int n, int overHang,
               int readLength, double *match) {
  if (overHang == 0) overHang = 1;
  if (overHang < 1) MISO_FAIL(MISO_EINVAL, "Overhang length invalid. Must be positive");
  if (readLength < 0) MISO_FAIL(MISO_EINVAL, "Read length cannot be negative");
  const CigarTable ct = parse_cigars(cigar, n, readLength);
  const int K = g.K;
  for (int r = 0; r < n; r++) {
    const int *ops = ct.ops.data() + ct.idx[r];
    const int nops = ct.idx[r + 1] - ct.idx[r];
    double *col = match + static_cast<size_t>(r) * K;
    const bool usable = ct.len[r] >= readLength && nops > 0 && ops[0] >= overHang &&
                        ops[nops - 1] >= overHang;
    for (int k = 0; k < K; k++) {
      col[k] = 0.0;
      if (!usable) continue;
      int p = pos[r], ex = g.exidx[k];
      const int last = g.exidx[k + 1];
      while (ex < last && (p < g.exstart[ex] || g.exend[ex] < p)) ex++;
      if (ex >= last) continue;
      bool ok = true;
      for (int c = 0; c < nops && ok; c++) {
        if (ops[c] > 0) {
          if (p + ops[c] - 1 > g.exend[ex]) ok = false; else p += ops[c];
        } else if (p != g.exend[ex] + 1) {
          ok = false;
        } else {
          p -= ops[c];
          ex++;
          if (ex >= last || p != g.exstart[ex]) ok = false;
        }
      }
      if (ok) col[k] = 1.0;
    }
  }
}

FragmentDist normal_fragment(double mean, double var, double numDevs, int minLength) {
  const double sd = std::sqrt(var);
  if (!(sd > 0)) MISO_FAIL(MISO_EINVAL, "Invalid `sigma' for normal");
  FragmentDist fd;
  fd.start = static_cast<int>(mean - sd * numDevs);
  int end = static_cast<int>(mean + sd * numDevs);
  if (fd.start < minLength) fd.start = minLength;
  if (end < fd.start) end = fd.start;
  fd.prob.resize(end - fd.start + 1);
  const double inv_sqrt_2pi = 0.398942280401432677939946059934;
  double sum = 0.0;
  for (int i = fd.start, j = 0; i <= end; i++, j++) {
    const double x = (i - mean) / sd;
    fd.prob[j] = inv_sqrt_2pi * std::exp(-0.5 * x * x) / sd;
  }
  for (double v : fd.prob) sum += v;
  const double scale = 1.0 / sum;
  for (double &v : fd.prob) v *= scale;
  return fd;
}

namespace {
// position of genomic coordinate p inside isoform k (1-based), -1 when p is not in an exon
int genomic_to_iso(const Gene &g, int k, int p) {
  int before = 0;  // isoform bases in the exons left of the one containing p
  for (int ex = g.exidx[k]; ex < g.exidx[k + 1]; ex++) {
    if (g.exend[ex] < p) { before += g.exend[ex] - g.exstart[ex] + 1; continue; }
    if (g.exstart[ex] <= p) return before + (p - g.exstart[ex]) + 1;
    return -1;
  }
  return -1;
}
}  // namespace

void match_iso_paired(const Gene &g, const int *pos, const char *const *cigar, int npos,
                      int readLength, int overHang, const FragmentDist &fd, double *match,
                      int *fraglen) {
  const int K = g.K, n = npos / 2, il = static_cast<int>(fd.prob.size());
  std::vector<double> mate(static_cast<size_t>(K) * (npos > 0 ? npos : 1));
  match_iso(g, pos, cigar, npos, overHang, readLength, mate.data());
  for (int r = 0; r < n; r++) {
    for (int k = 0; k < K; k++) {
      double v = 0.0;
      int fl = -1;
      if (mate[static_cast<size_t>(2 * r) * K + k] != 0 &&
          mate[static_cast<size_t>(2 * r + 1) * K + k] != 0) {
        const int frag =
            genomic_to_iso(g, k, pos[2 * r + 1]) - genomic_to_iso(g, k, pos[2 * r]) + readLength;
        if (frag >= fd.start && frag < fd.start + il) { v = fd.prob[frag - fd.start]; fl = frag; }
      }
      match[static_cast<size_t>(r) * K + k] = v;
      if (fraglen) fraglen[static_cast<size_t>(r) * K + k] = fl;
    }
  }
}

namespace {
struct SplitMix64 {  // Steele, Lea, Flood: "Fast splittable pseudorandom number generators", 2014
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double u01() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// isoform coordinate (1-based) -> genomic coordinate
int iso_to_genomic(const Gene &g, int k, int p) {
  for (int ex = g.exidx[k]; ex < g.exidx[k + 1]; ex++) {
    const int len = g.exend[ex] - g.exstart[ex] + 1;
    if (p <= len) return g.exstart[ex] + p - 1;
    p -= len;
  }
  return -1;
}

std::string cigar_for(const Gene &g, int k, int start, int readLength) {
  std::string out;
  int ex = g.exidx[k], left = readLength, rs = start;
  while (g.exend[ex] < rs) ex++;
  while (g.exend[ex] < rs + left - 1) {
    const int m = g.exend[ex] - rs + 1;
    out += std::to_string(m) + "M" + std::to_string(g.exstart[ex + 1] - g.exend[ex] - 1) + "N";
    left -= m;
    rs = g.exstart[ex + 1];
    ex++;
  }
  out += std::to_string(left) + "M";
  return out;
}

int pick(const std::vector<double> &cum, double u) {
  const double r = u * cum.back();
  int w = 0;
  while (w + 1 < static_cast<int>(cum.size()) && r > cum[w]) w++;
  return w;
}
}  // namespace

SimReads simulate_reads(const Gene &g, const double *expr, int n, int readLength, uint64_t seed) {
  SplitMix64 rng{seed};
  std::vector<int> eff(g.K);
  std::vector<double> cum(g.K);
  double acc = 0.0;
  for (int k = 0; k < g.K; k++) {
    const int l = g.isolen[k] - readLength + 1;
    eff[k] = l > 0 ? l : 0;
    acc += expr[k] * eff[k];
    cum[k] = acc;
  }
  if (!(acc > 0)) MISO_FAIL(MISO_FAILURE, "No isoform is possible");
  SimReads out;
  out.isoform.resize(n); out.position.resize(n); out.cigar.resize(n);
  for (int i = 0; i < n; i++) {
    const int k = pick(cum, rng.u01());
    const int p = 1 + static_cast<int>(rng.u01() * eff[k]);
    out.isoform[i] = k;
    out.position[i] = iso_to_genomic(g, k, p);
    out.cigar[i] = cigar_for(g, k, out.position[i], readLength);
  }
  return out;
}

SimReads simulate_paired_reads(const Gene &g, const double *expr, int npairs, int readLength,
                               const FragmentDist &fd, uint64_t seed) {
  SplitMix64 rng{seed};
  const int il = static_cast<int>(fd.prob.size());
  // P(isoform) ~ expression x sum_f P(f) * #positions a fragment of length f has on it
  std::vector<double> cum(g.K), fcum(il);
  double acc = 0.0;
  for (int k = 0; k < g.K; k++) {
    double w = 0.0;
    for (int j = 0; j < il; j++) {
      const int slots = g.isolen[k] - (fd.start + j) + 1;
      if (slots > 0) w += fd.prob[j] * slots;
    }
    acc += expr[k] * w;
    cum[k] = acc;
  }
  if (!(acc > 0)) MISO_FAIL(MISO_FAILURE, "No isoform is possible");
  SimReads out;
  out.isoform.resize(2 * npairs); out.position.resize(2 * npairs); out.cigar.resize(2 * npairs);
  for (int i = 0; i < npairs; i++) {
    const int k = pick(cum, rng.u01());
    double facc = 0.0;  // fragment length given the isoform: P(f) x #positions
    for (int j = 0; j < il; j++) {
      const int slots = g.isolen[k] - (fd.start + j) + 1;
      facc += slots > 0 ? fd.prob[j] * slots : 0.0;
      fcum[j] = facc;
    }
    const int frag = fd.start + pick(fcum, rng.u01());
    const int slots = g.isolen[k] - frag + 1;
    const int p = 1 + static_cast<int>(rng.u01() * slots);
    const int p2 = p + frag - readLength;
    for (int m = 0; m < 2; m++) {
      out.isoform[2 * i + m] = k;
      out.position[2 * i + m] = iso_to_genomic(g, k, m ? p2 : p);
      out.cigar[2 * i + m] = cigar_for(g, k, out.position[2 * i + m], readLength);
    }
  }
  return out;
}

void validate_params(const miso_params_t &p) {
  // miso.c:674-717 / miso_paired.c:285-339, plus what this build does not restate
  if (!p.paired) {
    if (p.algorithm != MISO_ALGO_REASSIGN && p.algorithm != MISO_ALGO_MARGINAL && p.algorithm != MISO_ALGO_CLASSES)
      MISO_FAIL(MISO_EINVAL, "`algorithm` is invalid");
    if (p.algorithm == MISO_ALGO_CLASSES && p.overHang > 1)   // assignment.c:103-106
      MISO_FAIL(MISO_UNIMPLEMENTED, "Overhang is not implemented in assignment matrix yet.");
  }
  if (p.start == MISO_START_GIVEN)
    MISO_FAIL(MISO_EINVAL, "`start_psi' must be given when starting from a given PSI");
  if (p.start == MISO_START_RANDOM || p.start == MISO_START_LINEAR)
    MISO_FAIL(MISO_UNIMPLEMENTED, "Only START_AUTO and START_UNIFORM run on the GPU");
  if (p.start < 0 || p.start > MISO_START_LINEAR) MISO_FAIL(MISO_EINVAL, "`start` is invalid");
  const int ov = p.overHang == 0 ? 1 : p.overHang;
  if (ov < 1 || ov >= p.readLength / 2)
    MISO_FAIL(MISO_EINVAL, "Overhang length invalid. Must be between 0 and readLength/2");
  if (p.noChains < 1) MISO_FAIL(MISO_EINVAL, "Number of chains must be at least one.");
  if (p.stop == MISO_STOP_CONVERGENT_MEAN && p.noChains == 1)
    MISO_FAIL(MISO_EINVAL, "Cannot access convergence with one chain only");
  if (p.stop != MISO_STOP_FIXEDNO && p.stop != MISO_STOP_CONVERGENT_MEAN) MISO_FAIL(MISO_EINVAL, "`stop` is invalid");
  if (p.noLag < 1 || p.noBurnIn < 0 || p.noIterations < p.noBurnIn)
    MISO_FAIL(MISO_EINVAL, "Invalid iteration / burn-in / lag combination");
}

// miso.c:556-636, as it stands: per chain a running mean whose divisor is 1 again for the chain's second sample and a
// running SUM of squares (never divided), W = the mean over chains of those sums squared, B from the chain means with
// noSamples (all chains' samples) as n; converged when sqrt(((n - 1) / n W + B / n) / W) <= 1.1 for every isoform (a NaN
// -- W = 0 -- is "not converged").  Same operations in the same order as the CPU checker's restatement: the decision
// is part of the counter contract.
bool convergent_mean(const double *samples, int K, int C, int noSamples) {
  std::vector<double> mean(samples, samples + static_cast<size_t>(K) * C), ssq(static_cast<size_t>(K) * C, 0.0);
  for (int i = C, j = 0, l = 1; i < noSamples; i++, j = (j + 1) % C) {
    for (int k = 0; k < K; k++) {
      const double x = samples[static_cast<size_t>(i) * K + k], m0 = mean[j * K + k];
      const double mk = m0 + (x - m0) / l;
      ssq[j * K + k] = ssq[j * K + k] + (x - m0) * (x - mk);
      mean[j * K + k] = mk;
    }
    if (j == C - 1) l++;
  }
  bool stop = true;
  for (int k = 0; k < K; k++) {
    double all = 0.0, B = 0.0, W = 0.0;
    for (int j = 0; j < C; j++) all += mean[j * K + k];
    all /= C;
    for (int j = 0; j < C; j++) { const double t = mean[j * K + k] - all; B += t * t; }
    B *= noSamples / (C - 1.0);
    for (int j = 0; j < C; j++) { const double t = ssq[j * K + k]; W += t * t; }
    W /= C;
    const double rhat = std::sqrt(((noSamples - 1.0) / noSamples * W + B / noSamples) / W);
    stop = stop && rhat <= 1.1;
  }
  return stop;
}

PackedEvent pack_event(const miso_params_t &p, const FragmentDist *fd, int K, int N,
                       const double *match, const int *fraglen, const int *isolen,
                       const int *noexons, const double *hyper) {
  if (K < 2) MISO_FAIL(MISO_EINVAL, "At least two isoforms are needed");
  if (K > MISO_MAX_ISOFORMS) MISO_FAIL(MISO_UNIMPLEMENTED, "More than 256 isoforms");
  if (p.paired && (!fd || !fraglen)) MISO_FAIL(MISO_EINTERNAL, "Paired event without fragments");
  const int W = (K + 63) / 64;   // mask words per read
  // the packed form only needs which isoforms a read is compatible with (and, paired-end, the
  // fragment length in each); a caller-made single-end matrix with values other than 0/1 keeps
  // its values for the header's read classes
  std::vector<uint64_t> masks(static_cast<size_t>(N > 0 ? N : 1) * W, 0u);
  std::vector<uint16_t> frags;
  bool binary = true;
  if (p.paired) frags.assign(static_cast<size_t>(N) * K, FRAG_NONE);
  for (int i = 0; i < N; i++) {
    uint64_t *m = masks.data() + static_cast<size_t>(i) * W;
    for (int k = 0; k < K; k++) {
      const size_t j = static_cast<size_t>(i) * K + k;
      if (match[j] != 0) {
        m[k >> 6] |= 1ull << (k & 63);
        if (p.paired) {
          // the fragment length indexes the fragment-probability and score tables: a caller-made problem
          // outside [start, start + il) would read out of bounds on the host and on the device
          const long fl = static_cast<long>(fraglen[j]) - fd->start;
          if (fl < 0 || fl >= static_cast<long>(fd->prob.size()))
            MISO_FAIL(MISO_EINVAL, "Fragment length outside the fragment-length distribution");
          frags[j] = static_cast<uint16_t>(fl);
        }
      }
      if (!p.paired && match[j] != 0.0 && match[j] != 1.0) binary = false;
    }
  }
  return pack_event_masks(p, fd, K, N, masks.data(), p.paired ? frags.data() : nullptr,
                          (p.paired || binary) ? nullptr : match, isolen, noexons, hyper, W);
}

PackedEvent pack_event_masks(const miso_params_t &p, const FragmentDist *fd, int K, int N,
                             const uint64_t *masks, const uint16_t *frags, const double *se_values,
                             const int *isolen, const int *noexons, const double *hyper, int W) {
  if (K < 2) MISO_FAIL(MISO_EINVAL, "At least two isoforms are needed");
  if (K > MISO_MAX_ISOFORMS) MISO_FAIL(MISO_UNIMPLEMENTED, "More than 256 isoforms");
  if (W != (K + 63) / 64) MISO_FAIL(MISO_EINTERNAL, "Mask words do not match the isoform count");
  if (K > 64 && !p.paired && p.algorithm != MISO_ALGO_REASSIGN)
    MISO_FAIL(MISO_UNIMPLEMENTED, "The MARGINAL and CLASSES algorithms take at most 64 isoforms");
  const bool small = K <= 32;   // the class tables, work units and dense records of the kernels for up to 32 isoforms
  if (p.paired && (!fd || (N > 0 && !frags))) MISO_FAIL(MISO_EINTERNAL, "Paired event without fragments");
  const int ov = p.overHang == 0 ? 1 : p.overHang;
  PackedEvent e;
  e.K = K; e.N = N; e.paired = p.paired != 0;
  e.hyper.assign(K, 1.0);
  if (hyper) e.hyper.assign(hyper, hyper + K);
  e.base_count.assign(K, 0);
  e.fixed_ass.assign(N, -1);

  // --- constants ---
  e.consts.assign(3 * K + CONST_EXTRA, 0.0);
  const int il = p.paired ? static_cast<int>(fd->prob.size()) : 0;
  std::vector<double> isoscore_tab;  // paired: il x K
  const bool marginal = !p.paired && p.algorithm == MISO_ALGO_MARGINAL;
  if (marginal && se_values) MISO_FAIL(MISO_UNIMPLEMENTED, "The MARGINAL algorithm needs a 0/1 match matrix");
  std::vector<double> invlen(K, 1.0);
  if (!p.paired) {
    for (int k = 0; k < K; k++) {
      const int l = isolen[k] - p.readLength + 1 - 2 * (noexons[k] - 1) * (ov - 1);
      const int eff = l > 0 ? l : 0;
      e.consts[k] = std::log(static_cast<double>(eff));
      e.consts[K + k] = -std::log(static_cast<double>(l));
      // miso.c:800-808: the marginal algorithm's match matrix is divided by the effective length where that is not 0
      if (marginal && eff != 0) invlen[k] = 1.0 / eff;
    }
  } else {
    isoscore_tab.resize(static_cast<size_t>(il) * K);
    std::vector<double> ass(K, 0.0);
    for (int j = 0; j < il; j++) {
      for (int k = 0; k < K; k++) {
        const double lp = isolen[k] - fd->start - j + 1 - 2 * (noexons[k] - 1) * (ov - 1);
        isoscore_tab[static_cast<size_t>(k) * il + j] = -std::log(lp) + fd->prob[j];
        if (lp > 0) ass[k] += lp;
      }
    }
    for (int k = 0; k < K; k++) e.consts[k] = std::log(ass[k]);
    e.sfix_table.resize(static_cast<size_t>(K) * il);
    for (size_t i = 0; i < e.sfix_table.size(); i++) {
      const double s = isoscore_tab[i];
      e.sfix_table[i] = (std::isfinite(s) && std::fabs(s) < 31.0)
                            ? static_cast<int32_t>(std::llrint(s * MISO_SFIX_SCALE))
                            : SFIX_BAD;
    }
  }
  double asum = 0.0, lgeach = 0.0;
  for (int k = 0; k < K; k++) {
    e.consts[2 * K + k] = e.hyper[k] - 1.0;
    asum += e.hyper[k];
    lgeach += std::lgamma(e.hyper[k]);
  }
  const double sigma = 0.2 / K / K;
  e.consts[3 * K + 0] = std::lgamma(asum);
  e.consts[3 * K + 1] = lgeach;
  e.consts[3 * K + 2] = sigma;
  e.consts[3 * K + 3] = (K - 1 == 1) ? sigma : std::sqrt(sigma);
  e.consts[3 * K + 4] = std::pow(2 * M_PI * sigma, -0.5 * (K - 1));

  // --- reads: classes for the header, fixed vs drawing reads for the device ---
  // Read classes for the header (miso.c:762 / miso_paired.c:386-391), in the lexicographic column
  // order of matrix.pmt:546-562.  Columns of 0/1 (always, for matches built from alignments; the
  // paired-end classes are binarised by definition) are keyed by their bit pattern, isoform 0 most
  // significant; anything else (a caller-made single-end matrix) takes the general map.
  std::map<std::vector<double>, double> cls;
  std::vector<std::pair<uint64_t, double>> bcls;   // (bit-reversed mask, count), small
  const bool binary = se_values == nullptr;
  auto reversed = [K](uint64_t m) { uint64_t r = 0; for (int k = 0; k < K; k++) r = (r << 1) | ((m >> k) & 1u); return r; };
  uint64_t last_mask = ~0ull; size_t last_cls = 0;
  // more than 64 isoforms: the classes keyed by the column itself, isoform 0 in the top bit of word 0 -- the keys' order IS
  // the lexicographic column order
  std::map<std::vector<uint64_t>, double> big_cls;
  auto big_key = [K, W](const uint64_t *row) {
    std::vector<uint64_t> key(W, 0);
    for (int k = 0; k < K; k++) if ((row[k >> 6] >> (k & 63)) & 1ull) key[k >> 6] |= 1ull << (63 - (k & 63));
    return key;
  };
  for (int i = 0; i < N; i++) {
    const uint64_t *mrow = masks + static_cast<size_t>(i) * W;
    const uint64_t mask = mrow[0];
    int nv = 0;
    for (int w = 0; w < W; w++) nv += __builtin_popcountll(mrow[w]);
    if (binary && W > 1) {
      big_cls[big_key(mrow)] += 1.0;
    } else if (binary) {
      if (mask != last_mask) {   // reads of one class tend to come in runs
        const uint64_t rev = reversed(mask);
        size_t c = 0;
        while (c < bcls.size() && bcls[c].first != rev) c++;
        if (c == bcls.size()) bcls.emplace_back(rev, 0.0);
        last_mask = mask; last_cls = c;
      }
      bcls[last_cls].second += 1.0;
    } else {
      cls[std::vector<double>(se_values + static_cast<size_t>(i) * K, se_values + static_cast<size_t>(i + 1) * K)] += 1.0;
    }
    if (nv == 0) continue;
    if (nv == 1) {
      int first = 0;
      for (int w = 0; w < W; w++) if (mrow[w]) { first = 64 * w + __builtin_ctzll(mrow[w]); break; }
      e.fixed_ass[i] = first;
      e.base_count[first]++;
      if (p.paired) {
        const int32_t v = e.sfix_table[static_cast<size_t>(first) * il + frags[static_cast<size_t>(i) * K + first]];
        if (v == SFIX_BAD) e.base_bad = 1; else e.base_sfix += v;
      }
      continue;
    }
    e.fixed_ass[i] = -2;
    e.draw_index.push_back(i);
    if (!p.paired) { e.draw_mask.push_back(mask); for (int w = 1; w < W; w++) e.draw_mask_x.push_back(mrow[w]); }
    else e.draw_frag.insert(e.draw_frag.end(), frags + static_cast<size_t>(i) * K, frags + static_cast<size_t>(i + 1) * K);
    e.n_draw++;
  }
  if (p.paired && e.n_draw > 1) {
    // Paired-end draw order (the contract's, like the single-end one below): the
    // drawing reads by their fragment-length rows -- isoform 0 most significant, an incompatible isoform below every
    // length -- ties by read index.  The reference walks its reads in the order of their match columns too
    // (miso_paired.c:24-86 through splicing_order_matches); what the order buys on the device: neighbouring lanes look up
    // neighbouring entries of the fragment tables (LDS banks) and of the event's score table (one cache line instead of
    // sixteen per gather).
    auto key = [&](int r, int k) { const uint16_t f = e.draw_frag[static_cast<size_t>(r) * K + k]; return f == FRAG_NONE ? -1 : static_cast<int>(f); };
    std::vector<int32_t> by(e.n_draw);
    for (int r = 0; r < e.n_draw; r++) by[r] = r;
    std::sort(by.begin(), by.end(), [&](int32_t x, int32_t y) {
      for (int k = 0; k < K; k++) { const int a = key(x, k), b = key(y, k); if (a != b) return a < b; }
      return e.draw_index[x] < e.draw_index[y];
    });
    std::vector<int32_t> idx(e.n_draw); std::vector<uint16_t> fr(static_cast<size_t>(e.n_draw) * K);
    for (int r = 0; r < e.n_draw; r++) {
      idx[r] = e.draw_index[by[r]];
      std::memcpy(fr.data() + static_cast<size_t>(r) * K, e.draw_frag.data() + static_cast<size_t>(by[r]) * K, sizeof(uint16_t) * K);
    }
    e.draw_index.swap(idx); e.draw_frag.swap(fr);
  }
  if (p.paired && K == 2) {   // sampler_k2 MODE 2: may the read loop skip the "bad score" bookkeeping?
    e.pe_delta = true;
    for (int r = 0; r < e.n_draw && e.pe_delta; r++)
      if (e.sfix_table[e.draw_frag[static_cast<size_t>(r) * 2]] == SFIX_BAD ||
          e.sfix_table[static_cast<size_t>(il) + e.draw_frag[static_cast<size_t>(r) * 2 + 1]] == SFIX_BAD)
        e.pe_delta = false;
  }
  if (p.paired)   // whole quads of reads on the device: pad with incompatible reads
    while ((e.draw_frag.size() / K) % 4) e.draw_frag.insert(e.draw_frag.end(), K, FRAG_NONE);
  if (p.paired && K == 2 && e.pe_delta && pe_k2_entries(il) <= 0xFFFF) {   // sampler_k2 MODE 2 (device.hpp)
    const int nq = (e.n_draw + 3) / 4 + 1;
    e.draw_dense.resize(static_cast<size_t>(nq) * 8);   // u16 pairs: one u32 per read, little-endian
    for (int r = 0; r < 4 * nq; r++) {
      const bool real = r < e.n_draw;
      e.draw_dense[2 * r] = static_cast<uint16_t>(real ? e.draw_frag[static_cast<size_t>(r) * 2] : 2 * il);
      e.draw_dense[2 * r + 1] = static_cast<uint16_t>(real ? il + e.draw_frag[static_cast<size_t>(r) * 2 + 1] : 2 * il + 1);
    }
  }
  if (p.paired && K >= 3 && K <= PE_DENSE_KMAX && pe_dense_il2(il) <= 256) {
    // Quad records of pe_dense (device.hpp): every (read, isoform) as one byte -- no validity tests left in
    // the read loop -- and one flag per read: which of the reference's two stopping rules applies
    // (miso_paired.c:64-75: exactly two compatible isoforms or more).
    const int il2 = pe_dense_il2(il), qd = pe_dense_quad_dwords(K);
    const int nq = (e.n_draw + 3) / 4 + 1;   // + one quad of padding reads: what the lanes beyond the last quad process
    e.draw_dense.assign(static_cast<size_t>(nq) * qd * 2, 0);
    uint8_t *bytes = reinterpret_cast<uint8_t *>(e.draw_dense.data());
    e.dense_nobad = true;
    for (int r = 0; r < 4 * nq; r++) {
      uint8_t *quad = bytes + static_cast<size_t>(r / 4) * qd * 4;
      int nv = 0;
      for (int k = 0; k < K; k++) {
        const uint16_t f = r < e.n_draw ? e.draw_frag[static_cast<size_t>(r) * K + k] : FRAG_NONE;
        int idx = il;                                        // PE_ZERO
        if (f != FRAG_NONE) {
          idx = f; nv++;
          if (e.sfix_table[static_cast<size_t>(k) * il + f] == SFIX_BAD) e.dense_nobad = false;
        } else if (r >= e.n_draw && k == 0) idx = il + 1;    // PE_ONE: a padding read picks isoform 0
        quad[(r % 4) * K + k] = static_cast<uint8_t>(idx);
      }
      if (nv > 2) quad[4 * K] |= static_cast<uint8_t>(1u << (r % 4));
    }
    e.sfix_dense.assign(static_cast<size_t>(K) * il2, 0);
    for (int k = 0; k < K; k++)
      std::memcpy(e.sfix_dense.data() + static_cast<size_t>(k) * il2, e.sfix_table.data() + static_cast<size_t>(k) * il, il * sizeof(int32_t));
  }
  if (!p.paired && W > 1) {
    // draw order, more than 64 isoforms: by column (isoform 0 most significant, 0 < 1), ties by read index -- a stable sort
    // on the columns' keys (no class tables: those serve the kernels for up to 32 isoforms)
    std::vector<std::vector<uint64_t>> keys(e.n_draw);
    for (int r = 0; r < e.n_draw; r++) keys[r] = big_key(masks + static_cast<size_t>(e.draw_index[r]) * W);
    std::vector<int32_t> by(e.n_draw);
    for (int r = 0; r < e.n_draw; r++) by[r] = r;
    std::stable_sort(by.begin(), by.end(), [&](int32_t x, int32_t y) { return keys[x] < keys[y]; });
    std::vector<int32_t> idx(e.n_draw); std::vector<uint64_t> m0(e.n_draw), mx(e.draw_mask_x.size());
    for (int r = 0; r < e.n_draw; r++) {
      idx[r] = e.draw_index[by[r]]; m0[r] = e.draw_mask[by[r]];
      for (int w = 1; w < W; w++) mx[static_cast<size_t>(r) * (W - 1) + w - 1] = e.draw_mask_x[static_cast<size_t>(by[r]) * (W - 1) + w - 1];
    }
    e.draw_index.swap(idx); e.draw_mask.swap(m0); e.draw_mask_x.swap(mx);
  }
  if (!p.paired && W == 1) {
    // draw order: by column (isoform 0 most significant, 0 < 1), ties by read index
    // (a stable counting sort over the distinct masks: usually a handful, at most n_draw)
    std::vector<std::pair<uint64_t, uint64_t>> dm;   // (reversed mask, mask) of the distinct drawing masks
    std::vector<int32_t> cid(e.n_draw);
    {
      uint64_t lm = ~0ull; int32_t lc = 0;
      for (int r = 0; r < e.n_draw; r++) {
        const uint64_t m = e.draw_mask[r];
        if (m != lm) {
          size_t c = 0;
          while (c < dm.size() && dm[c].second != m) c++;
          if (c == dm.size()) dm.emplace_back(reversed(m), m);
          lm = m; lc = static_cast<int32_t>(c);
        }
        cid[r] = lc;
      }
    }
    std::vector<int32_t> rank(dm.size()), start(dm.size() + 1, 0);
    {
      std::vector<int32_t> by(dm.size());
      for (size_t c = 0; c < dm.size(); c++) by[c] = static_cast<int32_t>(c);
      std::sort(by.begin(), by.end(), [&](int32_t x, int32_t y) { return dm[x].first < dm[y].first; });
      for (size_t j = 0; j < by.size(); j++) rank[by[j]] = static_cast<int32_t>(j);
    }
    for (int r = 0; r < e.n_draw; r++) start[rank[cid[r]] + 1]++;
    for (size_t c = 0; c < dm.size(); c++) start[c + 1] += start[c];
    std::vector<int32_t> idx(e.n_draw); std::vector<uint64_t> msk(e.n_draw);
    for (int r = 0; r < e.n_draw; r++) {
      const int32_t at = start[rank[cid[r]]]++;
      idx[at] = e.draw_index[r]; msk[at] = e.draw_mask[r];
    }
    e.draw_index.swap(idx); e.draw_mask.swap(msk);
    for (int r = 0; r < e.n_draw && small; r++)
      if (r == 0 || e.draw_mask[r] != e.draw_mask[r - 1]) { e.dcls_mask.push_back(static_cast<uint32_t>(e.draw_mask[r])); e.dcls_start.push_back(r); }
    e.dcls_start.push_back(e.n_draw);
    if (e.dcls_mask.size() > MAX_DRAW_CLASSES || !small) {
      e.dcls_mask.clear(); e.dcls_start.clear();
    } else {
      // one unit = the words of one Philox block (draws 4q .. 4q+3) that belong to one class; a
      // class's units are consecutive blocks, so the table only needs where they start
      for (size_t c = 0; c < e.dcls_mask.size(); c++) {
        const int nv = __builtin_popcount(e.dcls_mask[c]);
        const int r0 = e.dcls_start[c], r1 = e.dcls_start[c + 1];
        const int q0 = r0 >> 2, q1 = (r1 - 1) >> 2;
        const uint32_t head = (0xFu << (r0 & 3)) & 0xFu, tail = 0xFu >> (3 - ((r1 - 1) & 3));
        e.max_cls_size = std::max(e.max_cls_size, nv);
        const uint32_t row[CLS_WORDS] = {e.dcls_mask[c], static_cast<uint32_t>(e.n_units),
                                         static_cast<uint32_t>(e.n_units - q0), head | tail << 4};
        e.dcls_tab.insert(e.dcls_tab.end(), row, row + CLS_WORDS);
        e.n_units += q1 - q0 + 1;
        for (int q = q0; q <= q1; q++)   // the unit's descriptor (sampler_flat's read loop, device.hpp)
          e.unit_desc.push_back(((q == q0 ? head : 0xFu) & (q == q1 ? tail : 0xFu)) | static_cast<uint32_t>(c) << 4 |
                                static_cast<uint32_t>(q) << 12);
        for (int k = 0, j = 0; j < nv - 1; k++)   // every member but the last
          if ((e.dcls_mask[c] >> k) & 1u) { e.dcls_pairs.push_back(static_cast<uint16_t>(c << 8 | k)); j++; }
      }
      const uint32_t last[CLS_WORDS] = {0u, static_cast<uint32_t>(e.n_units), 0u, 0xFFu};
      e.dcls_tab.insert(e.dcls_tab.end(), last, last + CLS_WORDS);
      for (int k = 0; k < K; k++) {   // A_k: every read of a class whose last isoform is <= k picks <= k
        uint32_t ak = 0;
        for (size_t c = 0; c < e.dcls_mask.size(); c++)
          if (31 - __builtin_clz(e.dcls_mask[c]) <= k) ak += static_cast<uint32_t>(e.dcls_start[c + 1] - e.dcls_start[c]);
        e.dcls_tab.push_back(ak);
      }
    }
  }
  if (binary && W > 1) {
    for (const auto &kv : big_cls) {
      for (int k = 0; k < K; k++) e.class_templates.push_back((kv.first[k >> 6] >> (63 - (k & 63))) & 1ull ? 1.0 : 0.0);
      e.class_counts.push_back(kv.second);
    }
  } else if (binary) {
    std::sort(bcls.begin(), bcls.end());
    for (const auto &kv : bcls) {
      for (int k = 0; k < K; k++) e.class_templates.push_back((kv.first >> (K - 1 - k)) & 1ull ? 1.0 : 0.0);
      e.class_counts.push_back(kv.second);
      if (marginal && kv.first != 0) {   // the classes the marginal likelihood sums over, in this order (device.hpp)
        for (int k = 0; k < K; k++) e.mcls_tab.push_back(((kv.first >> (K - 1 - k)) & 1ull) ? invlen[k] : 0.0);
        e.mcls_tab.push_back(kv.second);
      }
    }
  }
  for (const auto &kv : cls) {
    e.class_templates.insert(e.class_templates.end(), kv.first.begin(), kv.first.end());
    e.class_counts.push_back(kv.second);
  }
  return e;
}


// ---- algorithm = CLASSES: the gene's possible read classes (assignment.c:90-276, stated as WHAT it computes) ----
// A read of readLength bases starting at genomic position p lies on isoform k with a definite alignment (exon pieces and
// the gaps between them) or not at all; the isoforms sharing one alignment at p are a class; one column per distinct
// class = its 0/1 pattern times the number of such positions; columns ordered patterns with a 0 in an earlier isoform
// first (splicing_i_assignmat_simplify).  Equal to the reference's matrix on random gene structures (the CPU checker's
// restatement is tested against the reference's function, this one against the checker's).
namespace {
std::vector<std::pair<uint64_t, double>> gene_classes(const Gene &g, int readLength) {
  const int K = g.K;
  // (a class is a 64-bit pattern of isoforms: ADVICE r5 -- genes of 65 ... 256 isoforms are first-class objects now)
  if (K > 64) MISO_FAIL(MISO_UNIMPLEMENTED, "the assignment matrix (algorithm = CLASSES) takes at most 64 isoforms");
  std::vector<std::pair<uint64_t, double>> cls;
  if (g.exstart.empty()) return cls;
  const int gs = *std::min_element(g.exstart.begin(), g.exstart.end()), ge = *std::max_element(g.exend.begin(), g.exend.end());
  std::vector<std::vector<int>> sig(K);
  // only positions inside some exon can start a read (the walk used to cover the introns too -- a megabase gene
  // is mostly intron): the union of the exons as sorted, merged intervals
  std::vector<std::pair<int, int>> iv;
  for (size_t i = 0; i < g.exstart.size(); i++) iv.emplace_back(g.exstart[i], g.exend[i]);
  std::sort(iv.begin(), iv.end());
  std::vector<std::pair<int, int>> merged;
  for (const auto &x : iv) {
    if (!merged.empty() && x.first <= merged.back().second + 1) merged.back().second = std::max(merged.back().second, x.second);
    else merged.push_back(x);
  }
  (void) gs;
  for (const auto &span : merged)
  for (int p = span.first; p <= std::min(span.second, ge - readLength + 1); p++) {
    bool any = false;
    for (int k = 0; k < K; k++) {
      sig[k].clear();
      for (int i = g.exidx[k]; i < g.exidx[k + 1]; i++) {
        if (g.exstart[i] <= p && p <= g.exend[i]) {
          int rem = readLength, cur = p, j = i;
          for (;;) {
            const int avail = g.exend[j] - cur + 1;
            if (rem <= avail) { sig[k].push_back(rem); rem = 0; break; }
            sig[k].push_back(avail); rem -= avail;
            if (j + 1 >= g.exidx[k + 1]) break;
            sig[k].push_back(-(g.exstart[j + 1] - g.exend[j] - 1));
            j++; cur = g.exstart[j];
          }
          if (rem != 0) sig[k].clear(); else any = true;
          break;
        }
      }
    }
    if (!any) continue;
    uint64_t done = 0;
    for (int k = 0; k < K; k++) {
      if (sig[k].empty() || ((done >> k) & 1ull)) continue;
      uint64_t m = 1ull << k;
      for (int k2 = k + 1; k2 < K; k2++) if (sig[k2] == sig[k]) m |= 1ull << k2;
      done |= m;
      size_t c = 0;
      while (c < cls.size() && cls[c].first != m) c++;
      if (c == cls.size()) cls.emplace_back(m, 0.0);
      cls[c].second += 1.0;
    }
  }
  std::sort(cls.begin(), cls.end(), [](const std::pair<uint64_t, double> &x, const std::pair<uint64_t, double> &y) {
    const uint64_t d = x.first ^ y.first;
    return d != 0 && !((x.first >> __builtin_ctzll(d)) & 1ull);   // 0 at the first isoform where they differ: first
  });
  return cls;
}
}  // namespace

std::vector<double> assignment_matrix(const Gene &g, int readLength, int overHang) {
  if (overHang > 1) MISO_FAIL(MISO_UNIMPLEMENTED, "Overhang is not implemented in assignment matrix yet.");
  const auto cls = gene_classes(g, readLength);
  std::vector<double> m(cls.size() * static_cast<size_t>(g.K), 0.0);
  for (size_t c = 0; c < cls.size(); c++)
    for (int k = 0; k < g.K; k++) if ((cls[c].first >> k) & 1ull) m[c * g.K + k] = cls[c].second;
  return m;
}

void attach_gene_classes(PackedEvent &e, const miso_params_t &p, const Gene &g) {
  if (p.paired || p.algorithm != MISO_ALGO_CLASSES) return;
  const int K = g.K;
  const int ov = p.overHang == 0 ? 1 : p.overHang;
  std::vector<double> a = assignment_matrix(g, p.readLength, ov);
  const size_t nc = a.size() / K;
  for (int k = 0; k < K; k++) {   // matrix.pmt:1525-1541: every isoform's row sums to 1
    double rowsum = 0.0;
    for (size_t c = 0; c < nc; c++) rowsum += a[c * K + k];
    for (size_t c = 0; c < nc; c++) a[c * K + k] /= rowsum;
  }
  // solve.c:122-134 read by read = the event's own read classes (pattern, reads), each to the first column with its pattern
  std::vector<double> reads(nc, 0.0);
  const size_t nrc = e.class_counts.size();
  for (size_t r = 0; r < nrc; r++) {
    for (size_t c = 0; c < nc; c++) {
      bool same = true;
      for (int k = 0; k < K && same; k++) {
        const double m1 = e.class_templates[r * K + k], m2 = a[c * K + k];
        same = (m1 > 0 && m2 > 0) || (m1 == 0 && m2 == 0);
      }
      if (same) { reads[c] += e.class_counts[r]; break; }
    }
  }
  e.mcls_tab.clear();
  for (size_t c = 0; c < nc; c++) {   // a class without reads adds log(score) * 0 (miso.c:293): nothing
    if (reads[c] == 0.0) continue;
    e.mcls_tab.insert(e.mcls_tab.end(), a.begin() + c * K, a.begin() + (c + 1) * K);
    e.mcls_tab.push_back(reads[c]);
  }
}

}  // namespace miso
