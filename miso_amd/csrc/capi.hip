// capi.hip -- the extern "C" surface declared in include/miso_amd.h.
//
// Error convention of the reference (splicing_error.h:548 SPLICING_CHECK, pyerror.c:27-44): every
// entry point returns an int code; the text "Error at file:line: reason, strerror" of the last
// failure is kept per thread for the binding to raise.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "miso_philox.h"

#include "batch.hpp"
#include "coop.hpp"
#include "miso_alnio.h"

extern "C" int miso_usable_threads(void);   // alnio.cpp: affinity mask capped by the cgroup quota

// A batch of whole genes is up to fifteen kernels side by side (isoform-count classes x size buckets), each on its
// own stream; the HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and a
// kernel waits for the one before it in its queue.  Measured (16 384 genes of 3-20 isoforms, heavy-tailed read
// counts, 1 500 iterations): 1 queue 811 ms, 2: 506, 4: 290, 8: 198, 16: 195.  The runtime reads the variable when it
// initialises, at the first HIP call: loading this library early enough sets 8 unless the host chose a value.
// If the host process has ALREADY initialised the GPU runtime (torch imported and used first), the variable would be read
// by nobody and only mislead this library (and any other HIP user of the process) about what is in effect: then it is
// left alone and miso_hw_queues_in_effect() says "unknown" (0).  MISO_HW_QUEUES_IN_EFFECT=n: a host that knows tells.
#include <cstdlib>
#include <dlfcn.h>
namespace miso { int g_hw_queues_in_effect = 0; }
__attribute__((constructor(101))) static void miso_amd_runtime_defaults() {
  bool runtime_up = false;
  // hsa_system_get_info answers HSA_STATUS_ERROR_NOT_INITIALIZED before hsa_init and initialises nothing itself
  if (void *h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD)) {
    using info_fn = int (*)(int, void *);
    if (auto fn = reinterpret_cast<info_fn>(dlsym(h, "hsa_system_get_info"))) {
      uint16_t major = 0;
      runtime_up = fn(0 /* HSA_SYSTEM_INFO_VERSION_MAJOR */, &major) == 0;
    }
    dlclose(h);
  }
  if (const char *told = std::getenv("MISO_HW_QUEUES_IN_EFFECT")) { miso::g_hw_queues_in_effect = std::atoi(told); return; }
  if (runtime_up) return;   // too late to choose, and what was chosen cannot be asked: unknown
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  miso::g_hw_queues_in_effect = std::atoi(std::getenv("GPU_MAX_HW_QUEUES"));
}
// alnio.cpp: one event's reads into growing buffers (the C entry point wraps it)
int miso_aln_collect_reads(const miso_alnfile_t *f, int ref, int64_t start, int64_t end, int paired,
                           int strand_rule, int target_strand, int given_read_len,
                           std::vector<int32_t> &pos_out, std::string &cig_out, int64_t *n_reads,
                           int64_t *n_strand_discarded);

using namespace miso;

struct miso_gene { Gene g; };

namespace {
thread_local std::string g_last_error;
struct Rethrow { int code; };  // an inner entry point already recorded the error text

template <class F> int guarded(F &&f) {
  try {
    f();
    return MISO_SUCCESS;
  } catch (const Rethrow &r) {
    return r.code;
  } catch (const Error &e) {
    g_last_error = e.text;
    return e.code;
  } catch (const std::bad_alloc &) {
    g_last_error = "Error at capi.hip:0: allocation failed, Out of memory";
    return MISO_ENOMEM;
  } catch (const std::exception &e) {
    g_last_error = std::string("Error at capi.hip:0: ") + e.what() + ", Internal error, likely a bug";
    return MISO_EINTERNAL;
  }
}

void need(const void *p, const char *what) {
  if (!p) MISO_FAIL(MISO_EINVAL, std::string(what) + " must not be NULL");
}

void fill_rundata(const miso_batch &b, int K, int accepted, int64_t iterations, miso_rundata_t *rd) {
  rd->noIso = K; rd->noIters = b.p.noIterations; rd->maxIters = 0; rd->noBurnIn = b.p.noBurnIn;
  rd->noLag = b.p.noLag; rd->noAccepted = accepted;
  rd->noRejected = static_cast<int>(b.p.noChains * iterations - accepted);
  rd->noChains = b.p.noChains; rd->noSamples = b.S();
}

const PackedEvent &event_at(const miso_batch *b, int i) {
  if (i < 0 || i >= static_cast<int>(b->events.size())) MISO_FAIL(MISO_EINVAL, "event index out of range");
  return b->events[i];
}
}  // namespace

// ---- the sample rows of `.miso` files, written natively (misopy/miso_sampler.py:456-464) ----
namespace {

// "%.*f" of a finite x with |x| < 2^31 / 10^D, exactly as printf rounds it (to nearest, ties to even
// on the EXACT binary value): scale, round, and fall back to snprintf when the scaled value is so
// close to a tie that the multiplication's rounding could have moved it across.
template <int D> inline char *fmt_fixed(char *o, double x) {
  constexpr double P = D == 4 ? 10000.0 : 100.0;
  if (!(x == x)) { std::memcpy(o, "nan", 3); return o + 3; }          // Python prints nan, never -nan
  const double ax = x < 0 ? -x : x;
  if (ax < 200000.0) {
    const double t = ax * P, r = std::floor(t + 0.5), d = t + 0.5 - r;
    if (d > 1e-6 && d < 1.0 - 1e-6) {
      uint64_t v = static_cast<uint64_t>(r);
      const bool neg = (x < 0 || (x == 0 && std::signbit(x)));        // "-0.00" like printf
      char tmp[24]; int n = 0;
      for (int i = 0; i < D; i++) { tmp[n++] = static_cast<char>('0' + v % 10); v /= 10; }
      tmp[n++] = '.';
      do { tmp[n++] = static_cast<char>('0' + v % 10); v /= 10; } while (v);
      if (neg) *o++ = '-';
      while (n) *o++ = tmp[--n];
      return o;
    }
  }
  return o + std::snprintf(o, 336, D == 4 ? "%.4f" : "%.2f", x);   // 1e308 has 309 digits
}

void write_miso_file(const miso_batch *b, int i, const char *path, const char *header) {
  const PackedEvent &e = event_at(b, i);
  const DevEvent &d = b->h_events[i];
  const unsigned char *out = b->h_out.data();
  const int S = b->S(), K = e.K;
  const double *samples = reinterpret_cast<const double *>(out + d.off_samples);   // K x S, column-major
  const double *loglik = reinterpret_cast<const double *>(out + d.off_loglik);
  std::string buf;
  buf.reserve(std::strlen(header) + 32 + static_cast<size_t>(S) * (8 * K + 16));
  buf.append(header);
  buf.append("sampled_psi\tlog_score\n");
  std::vector<char> rowbuf(static_cast<size_t>(K + 1) * 336 + 8);
  char *row = rowbuf.data();
  for (int s = 0; s < S; s++) {
    char *o = row;
    for (int k = 0; k < K; k++) {
      if (k) *o++ = ',';
      o = fmt_fixed<4>(o, samples[static_cast<size_t>(s) * K + k]);
    }
    *o++ = '\t';
    o = fmt_fixed<2>(o, loglik[s]);
    *o++ = '\n';
    buf.append(row, o - row);
  }
  FILE *f = std::fopen(path, "w");
  if (!f) MISO_FAIL(MISO_FAILURE, std::string("cannot open ") + path + " for writing");
  const size_t w = std::fwrite(buf.data(), 1, buf.size(), f);
  if (std::fclose(f) != 0 || w != buf.size()) MISO_FAIL(MISO_FAILURE, std::string("short write to ") + path);
}

}  // namespace


extern "C" {

const char *miso_strerror(int code) { return strerror_code(code); }
const char *miso_last_error(void) { return g_last_error.c_str(); }

int miso_device_count(int *count) {
  return guarded([&] { need(count, "count"); *count = device_count(); });
}
int miso_set_device(int device) { return guarded([&] { set_device(device); }); }

int miso_create_gene(const int *exons, int n_exons, const int *isoforms, int n_flat, const char *id,
                     const char *seqid, const char *source, int strand, miso_gene_t **gene) {
  return guarded([&] {
    need(gene, "gene");
    auto g = std::make_unique<miso_gene>();
    g->g = make_gene(exons, n_exons, isoforms, n_flat, id, seqid, source, strand);
    *gene = g.release();
  });
}
void miso_gene_destroy(miso_gene_t *gene) { delete gene; }
int miso_gene_noiso(const miso_gene_t *gene, int *noiso) {
  return guarded([&] { need(gene, "gene"); need(noiso, "noiso"); *noiso = gene->g.K; });
}
int miso_gene_isolength(const miso_gene_t *gene, int *isolength) {
  return guarded([&] {
    need(gene, "gene"); need(isolength, "isolength");
    std::memcpy(isolength, gene->g.isolen.data(), sizeof(int) * gene->g.K);
  });
}

int miso_match_iso(const miso_gene_t *gene, const int *position, const char *const *cigarstr,
                   int n_reads, int overHang, int readLength, double *match) {
  return guarded([&] {
    need(gene, "gene"); need(match, "match");
    match_iso(gene->g, position, cigarstr, n_reads, overHang, readLength, match);
  });
}

int miso_match_iso_paired(const miso_gene_t *gene, const int *position, const char *const *cigarstr,
                          int n_positions, int readLength, int overHang, double normalMean,
                          double normalVar, double numDevs, double *match, int *fragmentLength) {
  return guarded([&] {
    need(gene, "gene"); need(match, "match");
    const FragmentDist fd = normal_fragment(normalMean, normalVar, numDevs, readLength);
    match_iso_paired(gene->g, position, cigarstr, n_positions, readLength, overHang, fd, match,
                     fragmentLength);
  });
}

int miso_batch_create(const miso_params_t *params, miso_batch_t **batch) {
  return guarded([&] { need(params, "params"); need(batch, "batch"); *batch = batch_new(*params); });
}
void miso_batch_destroy(miso_batch_t *batch) { delete batch; }

int miso_batch_add_event(miso_batch_t *b, const miso_gene_t *gene, const int *position,
                         const char *const *cigarstr, int n_positions, const double *hyperp,
                         int n_hyperp, int *event_index) {
  return guarded([&] {
    need(b, "batch"); need(gene, "gene");
    const Gene &g = gene->g;
    if (hyperp && n_hyperp != g.K) MISO_FAIL(MISO_EINVAL, "Invalid hyperparameter vector length");
    if (b->uploaded) MISO_FAIL(MISO_EINVAL, "batch already uploaded");
    const int N = b->p.paired ? n_positions / 2 : n_positions;
    if (b->p.device_match) {   // row f1: parse now (errors surface here), match on the GPU at upload
      if (g.K < 2) MISO_FAIL(MISO_EINVAL, "At least two isoforms are needed");
      if (g.K > MISO_MAX_ISOFORMS) MISO_FAIL(MISO_UNIMPLEMENTED, "More than 256 isoforms");
      if (b->p.overHang < 0) MISO_FAIL(MISO_EINVAL, "Overhang length invalid. Must be positive");
      if (b->p.readLength < 0) MISO_FAIL(MISO_EINVAL, "Read length cannot be negative");
      miso_batch::Pending pe;
      pe.event = static_cast<int>(b->events.size());
      pe.gene = g;
      pe.pos.assign(position, position + (b->p.paired ? 2 * N : N));
      pe.ct = parse_cigars(cigarstr, b->p.paired ? 2 * N : N, b->p.readLength);
      if (hyperp) pe.hyper.assign(hyperp, hyperp + g.K);
      b->pending.push_back(std::move(pe));
      PackedEvent ph;
      ph.K = g.K; ph.N = N; ph.paired = b->p.paired != 0;
      b->events.push_back(std::move(ph));
      if (event_index) *event_index = static_cast<int>(b->events.size()) - 1;
      return;
    }
    std::vector<double> match(static_cast<size_t>(g.K) * (N > 0 ? N : 1));
    std::vector<int> fraglen;
    if (b->p.paired) {
      fraglen.resize(match.size());
      match_iso_paired(g, position, cigarstr, n_positions, b->p.readLength, b->p.overHang, b->fd,
                       match.data(), fraglen.data());
    } else {
      match_iso(g, position, cigarstr, N, b->p.overHang, b->p.readLength, match.data());
    }
    b->events.push_back(pack_event(b->p, b->p.paired ? &b->fd : nullptr, g.K, N, match.data(),
                                   b->p.paired ? fraglen.data() : nullptr, g.isolen.data(),
                                   g.noexons.data(), hyperp));
    attach_gene_classes(b->events.back(), b->p, g);   // algorithm = CLASSES only
    if (event_index) *event_index = static_cast<int>(b->events.size()) - 1;
  });
}

int miso_batch_add_problem(miso_batch_t *b, int noiso, int n_reads, const double *match,
                           const int *fragmentLength, const int *isolength, const int *noexons,
                           const double *hyperp, int *event_index) {
  return guarded([&] {
    need(b, "batch"); need(isolength, "isolength"); need(noexons, "noexons");
    if (noiso < 0 || n_reads < 0) MISO_FAIL(MISO_EINVAL, "Negative isoform or read count");
    if (n_reads > 0) need(match, "match");
    if (b->p.paired && n_reads > 0) need(fragmentLength, "fragmentLength");
    if (b->uploaded) MISO_FAIL(MISO_EINVAL, "batch already uploaded");
    if (!b->p.paired && b->p.algorithm == MISO_ALGO_CLASSES)
      MISO_FAIL(MISO_UNIMPLEMENTED, "The CLASSES algorithm needs the gene's structure (miso_batch_add_event)");
    b->events.push_back(pack_event(b->p, b->p.paired ? &b->fd : nullptr, noiso, n_reads, match,
                                   fragmentLength, isolength, noexons, hyperp));
    if (event_index) *event_index = static_cast<int>(b->events.size()) - 1;
  });
}

int miso_simulate_reads(const miso_gene_t *gene, const double *expression, int n_reads,
                        int readLength, double normalMean, double normalVar, double numDevs,
                        uint64_t sim_seed, int *isoform, int *position, char *cigar,
                        int cigar_stride) {
  return guarded([&] {
    need(gene, "gene"); need(expression, "expression"); need(position, "position"); need(cigar, "cigar");
    const bool paired = normalVar > 0;
    const SimReads r = paired
        ? simulate_paired_reads(gene->g, expression, n_reads, readLength,
                                normal_fragment(normalMean, normalVar, numDevs, readLength), sim_seed)
        : simulate_reads(gene->g, expression, n_reads, readLength, sim_seed);
    for (size_t i = 0; i < r.position.size(); i++) {
      position[i] = r.position[i];
      if (isoform) isoform[i] = r.isoform[i];
      if (static_cast<int>(r.cigar[i].size()) >= cigar_stride) MISO_FAIL(MISO_EINVAL, "CIGAR string too long");
      std::strcpy(cigar + i * static_cast<size_t>(cigar_stride), r.cigar[i].c_str());
    }
  });
}

int miso_batch_add_simulated(miso_batch_t *b, const miso_gene_t *gene, const double *expression,
                             int n_reads, uint64_t sim_seed, const double *hyperp, int n_hyperp,
                             int *event_index) {
  return guarded([&] {
    need(b, "batch"); need(gene, "gene"); need(expression, "expression");
    const SimReads r = b->p.paired
        ? simulate_paired_reads(gene->g, expression, n_reads, b->p.readLength, b->fd, sim_seed)
        : simulate_reads(gene->g, expression, n_reads, b->p.readLength, sim_seed);
    std::vector<const char *> cig(r.cigar.size());
    for (size_t i = 0; i < cig.size(); i++) cig[i] = r.cigar[i].c_str();
    const int rc = miso_batch_add_event(b, gene, r.position.data(), cig.data(),
                                        static_cast<int>(r.position.size()), hyperp, n_hyperp, event_index);
    if (rc) throw Rethrow{rc};
  });
}

int miso_batch_size(const miso_batch_t *b, int *n) {
  return guarded([&] { need(b, "batch"); need(n, "n_events"); *n = static_cast<int>(b->events.size()); });
}
int miso_batch_set_event_id(miso_batch_t *b, int event_index, uint32_t event_id) {
  return guarded([&] {
    need(b, "batch");
    (void) event_at(b, event_index);
    if (b->uploaded) MISO_FAIL(MISO_EINVAL, "batch already uploaded");
    if (b->event_ids.size() < b->events.size()) b->event_ids.resize(b->events.size(), -1);
    b->event_ids[event_index] = static_cast<int64_t>(event_id);
  });
}
int miso_batch_upload(miso_batch_t *b, int device) {
  return guarded([&] { need(b, "batch"); b->upload(device); });
}
int miso_batch_launch(miso_batch_t *b, uint64_t seed, uint32_t first_event_id) {
  return guarded([&] { need(b, "batch"); b->launch(seed, first_event_id); });
}
int miso_batch_sync(miso_batch_t *b, float *ms) {
  return guarded([&] { need(b, "batch"); b->sync(ms); });
}
int miso_batch_download(miso_batch_t *b) {
  return guarded([&] { need(b, "batch"); b->download(); });
}
int miso_batch_run(miso_batch_t *b, int device, uint64_t seed, uint32_t first_event_id) {
  return guarded([&] {
    need(b, "batch");
    b->upload(device); b->launch(seed, first_event_id); b->sync(nullptr); b->download();
  });
}

int miso_batch_event_info(const miso_batch_t *b, int i, int *noiso, int *n_reads, int *n_samples,
                          int *n_classes) {
  return guarded([&] {
    need(b, "batch");
    const PackedEvent &e = event_at(b, i);
    if (noiso) *noiso = e.K;
    if (n_reads) *n_reads = e.N;
    if (n_samples) *n_samples = b->S();
    if (n_classes) *n_classes = static_cast<int>(e.class_counts.size());
  });
}

int miso_batch_get_result(const miso_batch_t *b, int i, double *samples, double *logLik,
                          double *class_templates, double *class_counts, int *assignment,
                          miso_rundata_t *rundata) {
  return guarded([&] {
    need(b, "batch");
    const PackedEvent &e = event_at(b, i);
    // read classes are host-side set-up quantities (miso.c:762); everything else needs the run
    if (class_templates)
      std::memcpy(class_templates, e.class_templates.data(), sizeof(double) * e.class_templates.size());
    if (class_counts)
      std::memcpy(class_counts, e.class_counts.data(), sizeof(double) * e.class_counts.size());
    if (!samples && !logLik && !assignment && !rundata) return;
    if (!b->downloaded) MISO_FAIL(MISO_EINVAL, "results not downloaded yet");
    const DevEvent &d = b->h_events[i];
    const unsigned char *out = b->h_out.data();
    const int S = b->S();
    if (samples) std::memcpy(samples, out + d.off_samples, sizeof(double) * S * e.K);
    if (logLik) std::memcpy(logLik, out + d.off_loglik, sizeof(double) * S);
    if (assignment) {  // chain 0, final state (miso.c:943-946)
      const uint8_t *da = out + d.off_drawass;
      for (int k = 0; k < e.N; k++) assignment[k] = e.fixed_ass[k];
      for (int r = 0; r < e.n_draw; r++) assignment[e.draw_index[r]] = da[r];
    }
    if (rundata) {
      const ChainStats *st = reinterpret_cast<const ChainStats *>(out + d.off_stats);
      int acc = 0;
      for (int c = 0; c < b->p.noChains; c++) acc += st[c].accepted;
      // (stop = CONVERGENT_MEAN: the counts of the last round, paired-end of all rounds -- batch.hpp iters_counted)
      fill_rundata(*b, e.K, acc, b->iters_counted.empty() ? b->p.noIterations : b->iters_counted[i], rundata);
    }
  });
}

int miso_batch_add_event_aln(miso_batch_t *b, const miso_gene_t *gene, const miso_alnfile_t *f, int ref,
                             int64_t start, int64_t end, int strand_rule, int target_strand,
                             int given_read_len, int64_t min_reads, const double *hyperp, int n_hyperp,
                             int64_t *n_reads, int *event_index) {
  if (event_index) *event_index = -1;
  int64_t n = 0;
  thread_local std::vector<int32_t> pos;
  thread_local std::string cig;
  thread_local std::vector<const char *> cptr;
  int rc = guarded([&] {
    need(b, "batch"); need(gene, "gene"); need(f, "alignment file");
    if (miso_aln_collect_reads(f, ref, start, end, b->p.paired ? 1 : 0, strand_rule, target_strand,
                               given_read_len, pos, cig, &n, nullptr))
      MISO_FAIL(MISO_EINVAL, std::string("alignment reader: ") + miso_aln_last_error());
  });
  if (rc) return rc;
  if (n_reads) *n_reads = n;
  if (n == 0 || n < min_reads) return MISO_SUCCESS;   // the caller's skip rules (run_miso.py:139-147)
  for (auto &p : pos) p += 1;                          // 0-based -> 1-based (miso_sampler.py:284)
  cptr.clear();
  const char *c = cig.data();
  for (size_t i = 0; i < pos.size(); i++) { cptr.push_back(c); c += std::strlen(c) + 1; }
  return miso_batch_add_event(b, gene, pos.data(), cptr.data(), static_cast<int>(pos.size()), hyperp,
                              n_hyperp, event_index);
}

// Many events at once: the reads of every region are collected and their CIGARs parsed on n_threads
// host threads, then appended in order (40 000 events x 1000 reads: the serial loop of
// miso_batch_add_event_aln calls was the longest stage of a whole-genome run).
int miso_batch_add_events_aln(miso_batch_t *b, int n, const miso_gene_t *const *genes,
                              const miso_alnfile_t *f, const int *ref, const int64_t *start,
                              const int64_t *end, int strand_rule, const int *target_strand,
                              int given_read_len, int64_t min_reads, int n_threads, int64_t *n_reads,
                              int *event_index) {
  return guarded([&] {
    need(b, "batch"); need(f, "alignment file");
    if (n < 0) MISO_FAIL(MISO_EINVAL, "negative event count");
    if (n == 0) return;
    need(genes, "genes"); need(ref, "ref"); need(start, "start"); need(end, "end");
    need(target_strand, "target_strand"); need(n_reads, "n_reads"); need(event_index, "event_index");
    if (b->uploaded) MISO_FAIL(MISO_EINVAL, "batch already uploaded");
    if (!b->p.device_match) {   // host matching packs at add time: keep the serial path
      for (int i = 0; i < n; i++) {
        const int rc = miso_batch_add_event_aln(b, genes[i], f, ref[i], start[i], end[i], strand_rule,
                                                target_strand[i], given_read_len, min_reads, nullptr, 0,
                                                &n_reads[i], &event_index[i]);
        if (rc) throw Rethrow{rc};
      }
      return;
    }
    if (b->p.overHang < 0) MISO_FAIL(MISO_EINVAL, "Overhang length invalid. Must be positive");
    if (b->p.readLength < 0) MISO_FAIL(MISO_EINVAL, "Read length cannot be negative");
    const int paired = b->p.paired ? 1 : 0;
    std::vector<miso_batch::Pending> made(n);
    std::vector<int> err_code(n, 0);
    std::vector<std::string> err_text(n);
    std::atomic<int> next{0};
    auto work = [&] {
      std::vector<int32_t> pos; std::string cig; std::vector<const char *> cptr;
      for (;;) {
        const int i = next.fetch_add(1);
        if (i >= n) return;
        n_reads[i] = 0; event_index[i] = -1;
        try {
          if (!genes[i]) MISO_FAIL(MISO_EINVAL, "gene must not be NULL");
          const Gene &g = genes[i]->g;
          if (g.K < 2) MISO_FAIL(MISO_EINVAL, "At least two isoforms are needed");
          if (g.K > MISO_MAX_ISOFORMS) MISO_FAIL(MISO_UNIMPLEMENTED, "More than 256 isoforms");
          int64_t k = 0;
          if (miso_aln_collect_reads(f, ref[i], start[i], end[i], paired, strand_rule, target_strand[i],
                                     given_read_len, pos, cig, &k, nullptr))
            MISO_FAIL(MISO_EINVAL, std::string("alignment reader: ") + miso_aln_last_error());
          n_reads[i] = k;
          if (k == 0 || k < min_reads) continue;
          cptr.clear();
          const char *c = cig.data();
          for (size_t r = 0; r < pos.size(); r++) { cptr.push_back(c); c += std::strlen(c) + 1; }
          miso_batch::Pending &pe = made[i];
          pe.gene = g;
          pe.pos.resize(pos.size());
          for (size_t r = 0; r < pos.size(); r++) pe.pos[r] = pos[r] + 1;   // 1-based (miso_sampler.py:284)
          pe.ct = parse_cigars(cptr.data(), static_cast<int>(pos.size()), b->p.readLength);
        } catch (const Error &e) {
          err_code[i] = e.code; err_text[i] = e.text;
        } catch (const std::bad_alloc &) {
          err_code[i] = MISO_ENOMEM; err_text[i] = "Error at capi.hip:0: allocation failed, Out of memory";
        } catch (const std::exception &e) {   // nothing may leave a worker thread (std::terminate)
          err_code[i] = MISO_EINTERNAL; err_text[i] = std::string("Error at capi.hip:0: ") + e.what() + ", Internal error";
        } catch (...) {
          err_code[i] = MISO_EINTERNAL; err_text[i] = "Error at capi.hip:0: unknown exception, Internal error";
        }
      }
    };
    int T = n_threads > 0 ? n_threads : miso_usable_threads();
    T = std::max(1, std::min(T, n));
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    for (int i = 0; i < n; i++)
      if (err_code[i]) { g_last_error = err_text[i]; throw Rethrow{err_code[i]}; }   // first in order; nothing added
    for (int i = 0; i < n; i++) {
      if (n_reads[i] == 0 || n_reads[i] < min_reads) continue;
      miso_batch::Pending &pe = made[i];
      const int N = static_cast<int>(pe.pos.size()) / (paired ? 2 : 1);
      pe.event = static_cast<int>(b->events.size());
      PackedEvent ph;
      ph.K = pe.gene.K; ph.N = N; ph.paired = paired != 0;
      b->pending.push_back(std::move(pe));
      b->events.push_back(std::move(ph));
      event_index[i] = static_cast<int>(b->events.size()) - 1;
    }
  });
}

int miso_gene_assignment_matrix(const miso_gene_t *gene, int readLength, int overHang, double *matrix, int max_cols,
                                int *n_cols) {
  return guarded([&] {
    need(gene, "gene"); need(matrix, "matrix"); need(n_cols, "n_cols");
    const std::vector<double> m = assignment_matrix(gene->g, readLength, overHang == 0 ? 1 : overHang);
    const int nc = static_cast<int>(m.size() / std::max(1, gene->g.K));
    if (nc > max_cols) MISO_FAIL(MISO_EINVAL, "more read classes than the caller's matrix holds");
    std::memcpy(matrix, m.data(), m.size() * sizeof(double));
    *n_cols = nc;
  });
}

int miso_selftest_convergent_mean(const double *samples, int noiso, int noChains, int noSamples, int *stop) {
  return guarded([&] {
    need(samples, "samples"); need(stop, "stop");
    if (noiso < 1 || noChains < 2 || noSamples < noChains) MISO_FAIL(MISO_EINVAL, "needs two chains and a sample of each");
    *stop = convergent_mean(samples, noiso, noChains, noSamples) ? 1 : 0;
  });
}

int miso_batch_rounds(const miso_batch_t *b, int *rounds) {
  return guarded([&] { need(b, "batch"); need(rounds, "rounds"); *rounds = b->rounds; });
}

int miso_selftest_format(const double *x, int n, int decimals, char *out, int stride) {
  return guarded([&] {
    need(x, "x"); need(out, "out");
    if ((decimals != 2 && decimals != 4) || stride < 344) MISO_FAIL(MISO_EINVAL, "decimals must be 2 or 4, stride >= 344");
    for (int i = 0; i < n; i++) {
      char *o = out + static_cast<size_t>(i) * stride;
      char *e = decimals == 4 ? fmt_fixed<4>(o, x[i]) : fmt_fixed<2>(o, x[i]);
      *e = '\0';
    }
  });
}

int miso_batch_write_miso_files(const miso_batch_t *b, int n, const int *event_index,
                                const char *const *paths, const char *const *headers, int n_threads) {
  return guarded([&] {
    need(b, "batch");
    if (n < 0 || (n > 0 && (!event_index || !paths || !headers))) MISO_FAIL(MISO_EINVAL, "null argument");
    if (!b->downloaded) MISO_FAIL(MISO_EINVAL, "results not downloaded yet");
    for (int j = 0; j < n; j++) { (void) event_at(b, event_index[j]); need(paths[j], "path"); need(headers[j], "header"); }
    int T = n_threads > 0 ? n_threads : miso_usable_threads();
    T = std::max(1, std::min(T, std::min(n, 64)));
    std::atomic<int> next{0};
    std::mutex mu; std::string first_error; int first_code = 0;
    auto work = [&] {
      for (;;) {
        const int j = next.fetch_add(1);
        if (j >= n) return;
        try {
          write_miso_file(b, event_index[j], paths[j], headers[j]);
        } catch (const Error &err) {
          std::lock_guard<std::mutex> g(mu);
          if (!first_code) { first_code = err.code; first_error = err.text; }
        } catch (const std::bad_alloc &) {
          std::lock_guard<std::mutex> g(mu);
          if (!first_code) { first_code = MISO_ENOMEM; first_error = "Error at capi.hip:0: allocation failed, Out of memory"; }
        } catch (const std::exception &e) {   // nothing may leave a worker thread (std::terminate)
          std::lock_guard<std::mutex> g(mu);
          if (!first_code) { first_code = MISO_EINTERNAL; first_error = std::string("Error at capi.hip:0: ") + e.what() + ", Internal error"; }
        } catch (...) {
          std::lock_guard<std::mutex> g(mu);
          if (!first_code) { first_code = MISO_EINTERNAL; first_error = "Error at capi.hip:0: unknown exception, Internal error"; }
        }
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (first_code) { g_last_error = first_error; throw Rethrow{first_code}; }
  });
}

// The run-dependent fields of the `.miso` header line (miso_sampler.py:376-454), for many events in one call: per event one
// line "U<TAB>percent_accept<TAB>counts<TAB>assigned_counts\n", U = 1 when every read is unassigned (the caller skips such
// an event, miso_sampler.py:352-354).  The same text the Python formatting of miso_amd/miso_sampler.py miso_header()
// produces -- "%.2f", "(1,0):12,(0,1):34", "0:5,1:41" (reads_utils.py:37-46) -- without 30 microseconds of interpreter
// per event: a whole-genome run's header stage was as long as its sampling.
int miso_batch_header_fields(const miso_batch_t *b, int n, const int *event_index, char *buf, int64_t cap, int64_t *needed) {
  return guarded([&] {
    need(b, "batch"); need(needed, "needed");
    if (n < 0 || (n > 0 && !event_index)) MISO_FAIL(MISO_EINVAL, "null argument");
    if (!b->downloaded) MISO_FAIL(MISO_EINVAL, "results not downloaded yet");
    for (int j = 0; j < n; j++) (void) event_at(b, event_index[j]);
    std::vector<std::string> lines(static_cast<size_t>(n));
    int T = std::max(1, std::min(miso_usable_threads(), std::min(n / 64 + 1, 64)));
    std::atomic<int> next{0};
    std::atomic<bool> failed{false};
    auto work = [&] {
      try {
      std::vector<int64_t> cnt;
      char num[64];
      for (;;) {
        const int j0 = next.fetch_add(64);
        if (j0 >= n) return;
        for (int j = j0; j < std::min(n, j0 + 64); j++) {
          const int i = event_index[j];
          const PackedEvent &e = b->events[i];
          const DevEvent &d = b->h_events[i];
          const unsigned char *out = b->h_out.data();
          // assigned counts of chain 0's final state (miso.c:943-946): isoform -> reads, up to the largest isoform seen
          cnt.assign(static_cast<size_t>(e.K), 0);
          int top = -1;
          for (int k = 0; k < e.N; k++) { const int a = e.fixed_ass[k]; if (a >= 0) { cnt[a]++; top = std::max(top, a); } }
          const uint8_t *da = out + d.off_drawass;
          // (a drawing read's slot in fixed_ass is overwritten by its pick, as in miso_batch_get_result)
          for (int r = 0; r < e.n_draw; r++) {
            const int was = e.fixed_ass[e.draw_index[r]];
            if (was >= 0) cnt[was]--;
            const int a = da[r]; cnt[a]++; top = std::max(top, a);
          }
          if (top >= 0) { top = -1; for (int k = 0; k < e.K; k++) if (cnt[k] > 0) top = k; }
          std::string &ln = lines[j];
          ln += top < 0 ? "1\t" : "0\t";
          const ChainStats *st = reinterpret_cast<const ChainStats *>(out + d.off_stats);
          int64_t acc = 0;
          for (int c = 0; c < b->p.noChains; c++) acc += st[c].accepted;
          miso_rundata_t rd;
          fill_rundata(*b, e.K, static_cast<int>(acc), b->iters_counted.empty() ? b->p.noIterations : b->iters_counted[i], &rd);
          const double pa = static_cast<double>(rd.noAccepted) / static_cast<double>(rd.noAccepted + rd.noRejected) * 100.0;
          std::snprintf(num, sizeof num, "%.2f", pa);
          ln += num; ln += '\t';
          const size_t ncls = e.class_counts.size();
          for (size_t c = 0; c < ncls; c++) {
            if (c) ln += ',';
            ln += '(';
            for (int k = 0; k < e.K; k++) {
              if (k) ln += ',';
              std::snprintf(num, sizeof num, "%d", static_cast<int>(e.class_templates[c * e.K + k]));
              ln += num;
            }
            if (e.K == 1) ln += ',';                       // Python's one-element tuple
            std::snprintf(num, sizeof num, "):%d", static_cast<int>(e.class_counts[c]));
            ln += num;
          }
          ln += '\t';
          for (int k = 0; k <= top; k++) {
            std::snprintf(num, sizeof num, "%s%d:%lld", k ? "," : "", k, static_cast<long long>(cnt[k]));
            ln += num;
          }
          ln += '\n';
        }
      }
      } catch (...) { failed = true; }   // nothing may leave a worker thread (std::terminate)
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (failed) MISO_FAIL(MISO_ENOMEM, "allocation failed while formatting header fields");
    int64_t total = 1;
    for (const std::string &ln : lines) total += static_cast<int64_t>(ln.size());
    *needed = total;
    if (!buf || cap < total) return;
    char *o = buf;
    for (const std::string &ln : lines) { std::memcpy(o, ln.data(), ln.size()); o += ln.size(); }
    *o = 0;
  });
}

int miso_batch_get_trace(const miso_batch_t *b, int i, uint64_t *counts_hash, int32_t *counts_trace) {
  return guarded([&] {
    need(b, "batch");
    const PackedEvent &e = event_at(b, i);
    if (!b->downloaded) MISO_FAIL(MISO_EINVAL, "results not downloaded yet");
    const DevEvent &d = b->h_events[i];
    const unsigned char *out = b->h_out.data();
    if (counts_hash) {
      const ChainStats *st = reinterpret_cast<const ChainStats *>(out + d.off_stats);
      for (int c = 0; c < b->p.noChains; c++) counts_hash[c] = st[c].counts_hash;
    }
    if (counts_trace) {
      if (d.off_trace == NO_TRACE) MISO_FAIL(MISO_EINVAL, "batch was created without want_counts_trace");
      std::memcpy(counts_trace, out + d.off_trace,
                  sizeof(int32_t) * (static_cast<size_t>(b->p.noIterations) + 1) * b->p.noChains * e.K);
    }
  });
}

int miso_batch_from_samples(int n_events, const int *noiso, int n_samples, const double *const *samples, int device,
                            miso_batch_t **batch) {
  return guarded([&] {
    need(batch, "batch");
    if (n_events > 0) { need(noiso, "noiso"); need(samples, "samples"); }
    for (int i = 0; i < n_events; i++) need(samples[i], "samples[i]");
    miso_params_t p{};
    p.readLength = 36; p.overHang = 1; p.noChains = 1; p.noIterations = n_samples; p.maxIterations = n_samples;
    p.noBurnIn = 0; p.noLag = 1; p.algorithm = MISO_ALGO_REASSIGN; p.start = MISO_START_AUTO; p.stop = MISO_STOP_FIXEDNO;
    std::unique_ptr<miso_batch> b(batch_new(p));
    b->adopt_samples(n_events, noiso, n_samples, samples, device);
    *batch = b.release();
  });
}

int miso_batch_summarize(miso_batch_t *b, double confidence_level) {
  return guarded([&] { need(b, "batch"); b->summarize(confidence_level); });
}

int miso_batch_summarize_as_text(miso_batch_t *b, double confidence_level) {
  return guarded([&] { need(b, "batch"); b->summarize(confidence_level, true); });
}

int miso_batch_get_summary(const miso_batch_t *b, int i, double *mean, double *ci_low, double *ci_high) {
  return guarded([&] {
    need(b, "batch");
    const PackedEvent &e = event_at(b, i);
    if (!b->summarized) MISO_FAIL(MISO_EINVAL, "miso_batch_summarize has not run");
    const double *s = b->h_summary.data() + b->h_sum_off[i];
    for (int k = 0; k < e.K; k++) {
      if (mean) mean[k] = s[3 * k];
      if (ci_low) ci_low[k] = s[3 * k + 1];
      if (ci_high) ci_high[k] = s[3 * k + 2];
    }
  });
}

int miso_batch_compare(miso_batch_t *a, miso_batch_t *b, double smoothing) {
  return guarded([&] { need(a, "batch"); need(b, "batch"); a->compare(*b, smoothing); });
}

int miso_batch_get_comparison(const miso_batch_t *b, int i, double *mean1, double *mean2, double *bayes_factor,
                              double *density0) {
  return guarded([&] {
    need(b, "batch");
    const PackedEvent &e = event_at(b, i);
    if (!b->compared) MISO_FAIL(MISO_EINVAL, "miso_batch_compare has not run");
    size_t off = 0;
    for (int j = 0; j < i; j++) off += 4 * static_cast<size_t>(b->events[j].K);
    const double *s = b->h_compare.data() + off;
    for (int k = 0; k < e.K; k++) {
      if (mean1) mean1[k] = s[4 * k];
      if (mean2) mean2[k] = s[4 * k + 1];
      if (bayes_factor) bayes_factor[k] = s[4 * k + 2];
      if (density0) density0[k] = s[4 * k + 3];
    }
  });
}

int miso_batch_last_match_ms(const miso_batch_t *b, float *ms) {
  return guarded([&] { need(b, "batch"); need(ms, "ms"); *ms = b->match_ms; });
}

int miso_batch_get_match(const miso_batch_t *b, int i, double *match, int *fragmentLength) {
  return guarded([&] {
    need(b, "batch"); need(match, "match");
    const PackedEvent &e = event_at(b, i);
    if (!b->p.device_match || !b->p.want_counts_trace || !b->uploaded)
      MISO_FAIL(MISO_EINVAL, "device match outputs are kept only for uploaded device_match batches with want_counts_trace");
    const size_t n = static_cast<size_t>(e.N) * e.K;
    if (e.paired) {
      const std::vector<uint16_t> &f = b->kept_frags.at(i);
      for (size_t j = 0; j < n; j++) {
        const bool ok = f[j] != FRAG_NONE;
        match[j] = ok ? b->fd.prob[f[j]] : 0.0;
        if (fragmentLength) fragmentLength[j] = ok ? b->fd.start + f[j] : -1;
      }
    } else {
      const std::vector<uint64_t> &m = b->kept_masks.at(i);
      const int W = (e.K + 63) / 64;   // mask words per read
      for (int r = 0; r < e.N; r++)
        for (int k = 0; k < e.K; k++) match[static_cast<size_t>(r) * e.K + k] = (m[static_cast<size_t>(r) * W + (k >> 6)] >> (k & 63)) & 1ull ? 1.0 : 0.0;
    }
  });
}

int miso_contract_version(void) { return MISO_CONTRACT_VERSION; }

int miso_batch_set_clock_probe(miso_batch_t *b, int on) {
  return guarded([&] { need(b, "batch"); b->clock_probe = on != 0; if (on) b->probe_failed = false; });
}

int miso_batch_last_clock(const miso_batch_t *b, double *shader_ghz, double *window_ms) {
  return guarded([&] {
    need(b, "batch");
    if (shader_ghz) *shader_ghz = b->last_clock_ghz;
    if (window_ms) *window_ms = b->last_probe_ms;
  });
}

int miso_batch_coop_retries(const miso_batch_t *b, int *n) {
  return guarded([&] { need(b, "batch"); need(n, "n"); *n = b->coop_retries; });
}

int miso_batch_last_kernels(const miso_batch_t *b, char *buf, int buflen) {
  return guarded([&] {
    need(b, "batch"); need(buf, "buf");
    if (buflen <= static_cast<int>(b->last_kernels.size())) MISO_FAIL(MISO_EINVAL, "buffer too small");
    std::strcpy(buf, b->last_kernels.c_str());
  });
}

int miso_batch_launch_stats(const miso_batch_t *b, miso_kernel_stat_t *stats, int max_kernels, int *n_kernels) {
  return guarded([&] {
    need(b, "batch"); need(n_kernels, "n_kernels");
    if (!b->launched) MISO_FAIL(MISO_EINVAL, "batch not launched");
    if (b->stats_builder) {   // the walk over events and wavefronts happens here, once, not inside every launch
      miso_batch_t *mb = const_cast<miso_batch_t *>(b);
      mb->stats_builder();
      mb->stats_builder = nullptr;
    }
    *n_kernels = static_cast<int>(b->kernel_stats.size());
    for (int i = 0; stats && i < max_kernels && i < *n_kernels; i++) stats[i] = b->kernel_stats[i];
  });
}

int miso_plan_lanes(const int *n_draw, int n_events, int chains, int paired, int resident_workgroups, int max_chains_per_wave,
                    const double *cost5, double forced_target, int *n_runs, int *run_first_event, int *run_first_workgroup,
                    int *run_lanes, double *estimate3) {
  return guarded([&] {
    need(n_draw, "n_draw"); need(n_runs, "n_runs"); need(run_first_event, "run_first_event");
    need(run_first_workgroup, "run_first_workgroup"); need(run_lanes, "run_lanes");
    for (int i = 1; i < n_events; i++) if (n_draw[i] > n_draw[i - 1]) MISO_FAIL(MISO_EINVAL, "n_draw must not increase along the list");
    static const int se_widths[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 32, 64};
    static const int pe_widths[] = {4, 8, 16, 32, 64};
    LaneCost cost = paired ? k2_cost_paired() : k2_cost_single();
    if (cost5) { cost.block = cost5[0]; for (int i = 1; i <= 4; i++) cost.step[i] = cost5[i]; }
    const LanePlan pl = paired ? plan_lanes(n_draw, n_events, chains, pe_widths, 5, 4, 4, resident_workgroups, max_chains_per_wave, cost, forced_target, COOP_MAX_N, COOP_MAX_WGS)
                               : plan_lanes(n_draw, n_events, chains, se_widths, 12, 8, 8, resident_workgroups, max_chains_per_wave, cost, forced_target, COOP_MAX_N, COOP_MAX_WGS);
    *n_runs = pl.n_segs;
    for (int i = 0; i <= pl.n_segs; i++) { run_first_event[i] = pl.seg_slot[i]; run_first_workgroup[i] = pl.seg_block[i]; }
    for (int i = 0; i < pl.n_segs; i++) run_lanes[i] = pl.seg_lanes[i];
    if (estimate3) { estimate3[0] = pl.est_total; estimate3[1] = pl.est_max; estimate3[2] = pl.rounds; }
  });
}

int miso_batch_set_collapsed(miso_batch_t *b, int on) {
  return guarded([&] {
    need(b, "batch");
    if (on && b->p.paired) MISO_FAIL(MISO_EINVAL, "the collapsed Gibbs step needs exchangeable reads: single-end only");
    b->collapsed = on != 0;
    b->collapsed_level = on;
    b->k2_plan_key = -1;
  });
}

int miso_batch_get_placement(const miso_batch_t *b, int i, uint32_t *hw_id) {
  return guarded([&] {
    need(b, "batch"); need(hw_id, "hw_id");
    (void) event_at(b, i);
    if (!b->downloaded) MISO_FAIL(MISO_EINVAL, "results not downloaded yet");
    const ChainStats *st = reinterpret_cast<const ChainStats *>(b->h_out.data() + b->h_events[i].off_stats);
    for (int c = 0; c < b->p.noChains; c++) hw_id[c] = st[c].hw_id;
  });
}

int miso_batch_algorithmic_bytes(const miso_batch_t *b, double *bytes) {
  return guarded([&] {
    need(b, "batch"); need(bytes, "bytes");
    // per chain-iteration SE (8K+20)N, PE (8K+28)N; per event
    // + 8KN (match load) + 8(K+1)S (samples + logLik store)
    double total = 0.0;
    const double CM = static_cast<double>(b->p.noChains) * b->p.noIterations;
    for (const PackedEvent &e : b->events) {
      const double per_it = (8.0 * e.K + (b->p.paired ? 28.0 : 20.0)) * e.N;
      total += CM * per_it + 8.0 * e.K * e.N + 8.0 * (e.K + 1) * b->S();
    }
    *bytes = total;
  });
}

// one event per call = a batch of one
static int run_one(const miso_params_t &p, const miso_gene_t *gene, const int *position,
                   const char *const *cigarstr, int n_positions, const double *hyperp, int n_hyperp,
                   uint64_t seed, double *samples, double *logLik, double *templates, double *counts,
                   int *n_classes, int *assignment, miso_rundata_t *rundata) {
  return guarded([&] {
    need(gene, "gene");
    if (!hyperp || n_hyperp != gene->g.K) MISO_FAIL(MISO_EINVAL, "Invalid hyperparameter vector length");
    std::unique_ptr<miso_batch> b(batch_new(p));
    int rc = miso_batch_add_event(b.get(), gene, position, cigarstr, n_positions, hyperp, n_hyperp, nullptr);
    if (rc) throw Rethrow{rc};
    int dev = 0;
    (void) hipGetDevice(&dev);
    b->upload(dev); b->launch(seed, 0); b->sync(nullptr); b->download();
    rc = miso_batch_get_result(b.get(), 0, samples, logLik, templates, counts, assignment, rundata);
    if (rc) throw Rethrow{rc};
    if (n_classes) *n_classes = static_cast<int>(b->events[0].class_counts.size());
  });
}

int miso_run(const miso_gene_t *gene, const int *position, const char *const *cigarstr, int n_reads,
             int readLength, int overHang, int noChains, int noIterations, int maxIterations,
             int noBurnIn, int noLag, const double *hyperp, int n_hyperp, int algorithm, int start,
             int stop, uint64_t seed, double *samples, double *logLik, double *class_templates,
             double *class_counts, int *n_classes, int *assignment, miso_rundata_t *rundata) {
  miso_params_t p{};
  p.paired = 0; p.readLength = readLength; p.overHang = overHang; p.noChains = noChains;
  p.noIterations = noIterations; p.maxIterations = maxIterations; p.noBurnIn = noBurnIn;
  p.noLag = noLag; p.algorithm = algorithm; p.start = start; p.stop = stop;
  return run_one(p, gene, position, cigarstr, n_reads, hyperp, n_hyperp, seed, samples, logLik,
                 class_templates, class_counts, n_classes, assignment, rundata);
}

int miso_run_paired(const miso_gene_t *gene, const int *position, const char *const *cigarstr,
                    int n_positions, int readLength, int overHang, int noChains, int noIterations,
                    int maxIterations, int noBurnIn, int noLag, const double *hyperp, int n_hyperp,
                    int start, int stop, double normalMean, double normalVar, double numDevs,
                    uint64_t seed, double *samples, double *logLik, double *bin_class_templates,
                    double *bin_class_counts, int *n_classes, int *assignment,
                    miso_rundata_t *rundata) {
  miso_params_t p{};
  p.paired = 1; p.readLength = readLength; p.overHang = overHang; p.noChains = noChains;
  p.noIterations = noIterations; p.maxIterations = maxIterations; p.noBurnIn = noBurnIn;
  p.noLag = noLag; p.algorithm = MISO_ALGO_REASSIGN; p.start = start; p.stop = stop;
  p.normalMean = normalMean; p.normalVar = normalVar; p.numDevs = numDevs;
  return run_one(p, gene, position, cigarstr, n_positions, hyperp, n_hyperp, seed, samples, logLik,
                 bin_class_templates, bin_class_counts, n_classes, assignment, rundata);
}

int miso_selftest_detmath(const double *x, int n, double *out_exp, double *out_log, double *out_sqrt,
                          double *out_qnorm) {
  return guarded([&] { selftest_detmath(x, n, out_exp, out_log, out_sqrt, out_qnorm); });
}
int miso_selftest_philox(const uint32_t *ctr_key6, int n, uint32_t *out4) {
  return guarded([&] { selftest_philox(ctr_key6, n, out4); });
}

}  // extern "C"
