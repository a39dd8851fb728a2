// host.hpp -- host-side model of the sampler path: gene, alignment matching, packed problems.
//
// Product code (libmiso_amd.so).  Independent of the CPU checker, which has its own C
// restatement; the two meet only in tests.
#pragma once

#include <cstdint>
#include <exception>
#include <string>
#include <vector>

#include "miso_amd.h"

namespace miso {

// ---- errors: the reference's convention (splicing_error.h:548 SPLICING_CHECK / pyerror.c) ----
struct Error : std::exception {
  int code;
  std::string text;  // "Error at file:line: reason, strerror"
  Error(int code, const std::string &reason, const char *file, int line);
  struct Formatted {};
  Error(int c, const std::string &formatted_text, Formatted) : code(c), text(formatted_text) {}  // re-raise
  const char *what() const noexcept override { return text.c_str(); }
};
#define MISO_FAIL(code, reason) throw ::miso::Error((code), (reason), __FILE__, __LINE__)

const char *strerror_code(int code);

// ---- gene model: one gene of the reference's splicing_gff_t (simulator.c:9-66) ----
struct Gene {
  int K = 0;                  // isoforms (mRNAs)
  std::vector<int> exidx;     // K+1 offsets: isoform k owns exons [exidx[k], exidx[k+1])
  std::vector<int> exstart;   // exon coordinates per isoform, in the order given
  std::vector<int> exend;
  std::vector<int> isolen;    // gff.c:583-619
  std::vector<int> noexons;   // gff.c:624-657
  std::string id, seqid, source;
  int strand = 2;
};

Gene make_gene(const int *exons, int n_exons, const int *isoforms, int n_flat, const char *id,
               const char *seqid, const char *source, int strand);

// ---- alignments -> compatibility (solve.c:8-108, 141-306) ----
struct CigarTable {
  std::vector<int> ops;  // +len aligned block, -len skipped region
  std::vector<int> idx;  // n+1 offsets
  std::vector<int> len;  // reference-consuming length per read, clipped to readLength
};
CigarTable parse_cigars(const char *const *cigar, int n, int maxReadLength);

void match_iso(const Gene &g, const int *pos, const char *const *cigar, int n, int overHang,
               int readLength, double *match /* K x n */);

struct FragmentDist {  // simulator.c:198-219, normalised as miso_paired.c:303-307
  int start = 0;
  std::vector<double> prob;
};
FragmentDist normal_fragment(double mean, double var, double numDevs, int minLength);

void match_iso_paired(const Gene &g, const int *pos, const char *const *cigar, int npos,
                      int readLength, int overHang, const FragmentDist &fd, double *match,
                      int *fraglen /* may be null */);

// ---- synthetic reads (the module's simulateReads / simulatePairedReads: simulator.c:68-196,
// 221-442).  Same sampling scheme -- isoform ~ expression x effective length, start uniform on the
// isoform (paired: fragment length from the discretised normal), CIGAR from the exon structure --
// driven by a private splitmix64 stream instead of the reference's global generator.
struct SimReads {
  std::vector<int> isoform, position;
  std::vector<std::string> cigar;
};
SimReads simulate_reads(const Gene &g, const double *expr, int n, int readLength, uint64_t seed);
SimReads simulate_paired_reads(const Gene &g, const double *expr, int npairs, int readLength,
                               const FragmentDist &fd, uint64_t seed);

// ---- a packed event: everything the kernels need, nothing they do not ----
//
// The reference keeps match[K x N] doubles and walks all N reads every iteration.  Only reads
// with >= 2 compatible isoforms ever consume a random number (miso.c:65-80); the others are
// constants of the chain.  Packing therefore splits the reads into
//   * "fixed" reads (0 or 1 compatible isoform): folded into base_count[k] (and, paired-end,
//     into base_sfix), never touched again by the device;
//   * "drawing" reads, kept in read order (their rank is their RNG address, miso_philox.h):
//     single-end: one bit mask per read (bit k = compatible with isoform k; two u32 words from 33 isoforms on);
//     paired-end: K u16 fragment-length indices per read (0xFFFF = incompatible).
struct PackedEvent {
  int K = 0, N = 0, n_draw = 0;
  bool paired = false;
  std::vector<double> hyper;            // K
  std::vector<double> consts;           // 3K+5 doubles, layout in device.hpp
  std::vector<int32_t> base_count;      // K
  int64_t base_sfix = 0;                // paired: sum of fixed reads' scores, 2^-26 fixed point
  int32_t base_bad = 0;                 // paired: a fixed read has a non-finite score
  // The drawing reads (>= 2 compatible isoforms) in DRAW ORDER: draw r uses Gibbs word r of the
  // iteration.  Paired-end: input order.  Single-end: stable order by compatibility column, columns
  // compared lexicographically like the reference's own read order (matrix.pmt:546-562), so that
  // the reads of one class are consecutive.
  std::vector<int32_t> draw_index;      // n_draw: read index of draw r
  std::vector<uint64_t> draw_mask;      // single-end: n_draw (the device gets the low words, then -- K > 32 -- the high words)
  std::vector<uint64_t> draw_mask_x;    // ... more than 64 isoforms: mask words 1 .. W - 1 of every draw, n_draw x (W - 1)
  std::vector<uint32_t> dcls_mask;      // single-end: distinct masks among the drawing reads, in draw order
  std::vector<int32_t> dcls_start;      // ... first draw of each class (+ n_draw at the end)
  std::vector<uint32_t> dcls_tab;       // ... device class table, CLS_WORDS per class + sentinel (device.hpp)
  std::vector<uint16_t> dcls_pairs;     // ... (class << 8 | isoform) for every class member but its last
                                        // (all four empty with more than MAX_DRAW_CLASSES classes)
  int n_units = 0;                      // ... work units: (Philox block, class) incidences
  std::vector<uint32_t> unit_desc;      // ... one word per unit, in unit order: word mask | class << 4 | Philox block << 12 (device.hpp)
  int max_cls_size = 0;                 // ... most isoforms any drawing class is compatible with
  std::vector<uint16_t> draw_frag;      // paired-end: n_draw x K
  std::vector<int32_t> sfix_table;      // paired-end: K x il fixed-point isoscores
  bool pe_delta = false;                // paired-end, two isoforms: every drawing read's two scores are finite
  // paired-end, 3 <= K <= PE_DENSE_KMAX (device.hpp): the drawing reads as quad records for pe_dense
  // (kernels_grp.inl) -- one byte per (read, isoform), flags; two isoforms: sampler_k2 MODE 2's u16 pairs --
  // and the scores in that index space (layouts: device.hpp)
  std::vector<uint16_t> draw_dense;     // raw little-endian storage; empty = not available
  std::vector<int32_t> sfix_dense;      // K x (il + 2)
  bool dense_nobad = false;             // no compatible (read, isoform) of a drawing read has a non-finite score
  std::vector<double> mcls_tab;         // algorithm = MARGINAL / CLASSES: K + 1 doubles per class (device.hpp DevEvent::off_mcls)
  std::vector<int32_t> fixed_ass;       // N: -1 / isoform for fixed reads, -2 for drawing reads
  // header material for the caller (miso.c:762, miso_paired.c:386-391)
  std::vector<double> class_templates;  // K x ncls
  std::vector<double> class_counts;     // ncls
};

struct SamplerParams {
  miso_params_t p{};
  int n_samples() const { return p.noChains * (p.noIterations - p.noBurnIn) / p.noLag; }
};

void validate_params(const miso_params_t &p);
// stop = CONVERGENT_MEAN: the reference's test on the kept samples of one event (miso.c:556-636); samples: noSamples
// columns of K values, column i from chain i % C.  true = stop.
bool convergent_mean(const double *samples, int K, int C, int noSamples);

PackedEvent pack_event(const miso_params_t &p, const FragmentDist *fd, int K, int N,
                       const double *match, const int *fraglen, const int *isolen,
                       const int *noexons, const double *hyper);
// the same from what the packing really needs: per read the u32 compatibility mask, paired-end the
// K u16 fragment-length indices (FRAG_NONE = incompatible); se_values (single-end, optional): the
// match matrix when it holds values other than 0/1, for the header's read classes only
// (W: 64-bit mask words per read, (K + 63) / 64 -- 1 up to 64 isoforms)
PackedEvent pack_event_masks(const miso_params_t &p, const FragmentDist *fd, int K, int N,
                             const uint64_t *masks, const uint16_t *frags, const double *se_values,
                             const int *isolen, const int *noexons, const double *hyper, int W = 1);
// algorithm = CLASSES (miso.c:788-803): the event's table of the gene's possible read classes (splicing_assignment_matrix,
// assignment.c:90-276; rows normalised; the reads of every class, solve.c:110-137) from the gene's structure and the
// event's own read classes.  After pack_event*; MISO_UNIMPLEMENTED with an overhang above 1, as the reference.
void attach_gene_classes(PackedEvent &e, const miso_params_t &p, const Gene &g);
// the matrix itself (tests): K x (returned) columns, column-major
std::vector<double> assignment_matrix(const Gene &g, int readLength, int overHang);
// (33 ... MISO_MAX_ISOFORMS isoforms: no read classes, work units or dense records are made -- those serve the kernels
// for up to 32 isoforms; such an event is sampled by sampler_wave, lane k = isoform k -- beyond 64 isoforms by sampler_big,
// the chain's vectors in LDS -- runtime.hip)

}  // namespace miso
