/*
 * miso_amd.h -- C ABI of libmiso_amd.so: the MI355X-native MISO posterior sampler.
 *
 * This is the drop-in boundary for the one hot path of yarden/MISO: what the reference's
 * CPython glue (pysplicing/src/pysplicing.c) binds in the C core for `createGene`, `MISO` and
 * `MISOPaired`.  Plain pointers and sizes only; no Python, no torch, no HIP types.
 *
 *   reference symbol (file:line under /root/reference/pysplicing)   replaced by
 *   ------------------------------------------------------------   ------------------------
 *   splicing_create_gene        src/simulator.c:9                   miso_create_gene
 *   splicing_gff_destroy2       src/gff.c:80                        miso_gene_destroy
 *   splicing_gff_noiso_one      src/gff.c:684                       miso_gene_noiso
 *   splicing_gff_isolength_one  src/gff.c:689                       miso_gene_isolength
 *   splicing_miso               src/miso.c:638   (include/splicing.h:203)  miso_run
 *   splicing_miso_paired        src/miso_paired.c:241 (splicing.h:216)     miso_run_paired
 *   splicing_matchIso           src/solve.c:8                       miso_match_iso
 *   splicing_matchIso_paired    src/solve.c:141                     miso_match_iso_paired
 *   splicing_strerror           src/error.c:71                      miso_strerror
 *   splicing_error handler hook src/pyerror.c:27-44                 miso_last_error
 *
 * The reference runs ONE event per call on one CPU core.  A GPU needs thousands of events in
 * flight, so besides the per-event calls (a batch of one) the library exports a batch object:
 * add events, upload once, launch, read results per event.  Events are independent
 * (SURVEY.md section 8e): sharding a run over GPUs is a static split of the event list, each
 * shard a batch on its own device; results do not depend on the split because every random
 * draw is addressed by (seed, global event id, chain, iteration) -- include/miso_philox.h.
 *
 * Matrices are column-major as in the reference (include/splicing_matrix.h:67):
 * samples[K x S] stores sample s at samples[s*K .. s*K+K-1]; class_templates[K x ncls] likewise.
 * Error codes are the reference's (include/splicing_error.h:314-351).
 *
 * There is NO CPU fallback: every sampler entry point fails with MISO_ENODEVICE when no HIP
 * device is usable.
 */
#ifndef MISO_AMD_H
#define MISO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (splicing_error.h:314-351) ---- */
#define MISO_SUCCESS 0
#define MISO_FAILURE 1
#define MISO_ENOMEM 2
#define MISO_EINVAL 4
#define MISO_UNIMPLEMENTED 12
#define MISO_EINTERNAL 38
#define MISO_ENODEVICE 60 /* new: no usable HIP device / HIP runtime error */

/* ---- enums (splicing.h:59-62, 148-158; pysplicing/__init__.py:2-13) ---- */
#define MISO_ALGO_REASSIGN 0
#define MISO_ALGO_MARGINAL 1
#define MISO_ALGO_CLASSES 2
#define MISO_START_AUTO 0
#define MISO_START_UNIFORM 1
#define MISO_START_RANDOM 2
#define MISO_START_GIVEN 3
#define MISO_START_LINEAR 4
#define MISO_STOP_FIXEDNO 0
#define MISO_STOP_CONVERGENT_MEAN 1

/* Most isoforms of one gene.  The reference has no limit (miso.c:696, gff.c:684); here an isoform is a bit of a read's
   compatibility mask ((K + 31) / 32 words per read) and an entry of the chain's vectors; from 65 isoforms on the chain's
   vectors live in LDS (sampler_big) and a read's pick is returned as one byte: 256.  A gene with more is reported
   MISO_UNIMPLEMENTED by the one-event calls and skipped with a message by the batch callers. */
#define MISO_MAX_ISOFORMS 256

/* splicing_miso_rundata_t (splicing.h:143-146), same field order */
typedef struct miso_rundata {
  int noIso, noIters, maxIters, noBurnIn, noLag, noAccepted, noRejected, noChains, noSamples;
} miso_rundata_t;

typedef struct miso_gene miso_gene_t;   /* one gene = the reference's splicing_gff_t handle */
typedef struct miso_batch miso_batch_t;

/* ---- errors ---- */
const char *miso_strerror(int code);
/* "Error at <file>:<line>: <reason>, <strerror>" of the calling thread's last failure
   (the text pyerror.c:27-44 turns into the Python exception). */
const char *miso_last_error(void);

/* ---- device ---- */
int miso_device_count(int *count);
int miso_set_device(int device);

/* ---- gene model ---- */
/* exons: 2*n_exons ints (start,end; 1-based inclusive). isoforms: exon indices, each isoform
   terminated by -1 (the flattening pyconvert.c:55-87 produces). strand: 0 +, 1 -, 2 unknown. */
int miso_create_gene(const int *exons, int n_exons, const int *isoforms, int n_isoforms_flat,
                     const char *id, const char *seqid, const char *source, int strand,
                     miso_gene_t **gene);
void miso_gene_destroy(miso_gene_t *gene);
int miso_gene_noiso(const miso_gene_t *gene, int *noiso);
int miso_gene_isolength(const miso_gene_t *gene, int *isolength /* noiso */);

/* The gene's possible read classes (splicing_assignment_matrix, assignment.c:90-276; the module's assignmentMatrix,
   pysplicing.c:324-349): one column per distinct set of isoforms that share an alignment of a readLength-base read at some
   start position, its entries that number of positions for the set's isoforms and 0 for the others; columns ordered as the
   reference orders them.  matrix: noiso x max_cols doubles, column-major; *n_cols = columns (MISO_EINVAL if more than
   max_cols).  overHang > 1: MISO_UNIMPLEMENTED, as in the reference.  What algorithm = MISO_ALGO_CLASSES sums over. */
int miso_gene_assignment_matrix(const miso_gene_t *gene, int readLength, int overHang, double *matrix, int max_cols,
                                int *n_cols);

/* ---- problem construction on the host (input builder of the path) ---- */
/* match: noiso x n_reads, 1.0 / 0.0 */
int miso_match_iso(const miso_gene_t *gene, const int *position, const char *const *cigarstr,
                   int n_reads, int overHang, int readLength, double *match);
/* position/cigarstr hold 2*n_pairs mates (consecutive). match: noiso x n_pairs fragment
   probabilities (0 = incompatible); fragmentLength: noiso x n_pairs (-1 = none), may be NULL */
int miso_match_iso_paired(const miso_gene_t *gene, const int *position,
                          const char *const *cigarstr, int n_positions, int readLength,
                          int overHang, double normalMean, double normalVar, double numDevs,
                          double *match, int *fragmentLength);

/* ---- one event per call: splicing_miso / splicing_miso_paired ---- */
/* Output sizes: samples noiso*S, logLik S with S = noChains*(noIterations-noBurnIn)/noLag;
   class_templates noiso*n_reads (worst case), class_counts n_reads, assignment n_reads.
   Any output pointer may be NULL.  seed: the one addition to the reference signature. */
int miso_run(const miso_gene_t *gene, const int *position, const char *const *cigarstr,
             int n_reads, int readLength, int overHang, int noChains, int noIterations,
             int maxIterations, int noBurnIn, int noLag, const double *hyperp, int n_hyperp,
             int algorithm, int start, int stop, uint64_t seed, double *samples, double *logLik,
             double *class_templates, double *class_counts, int *n_classes, int *assignment,
             miso_rundata_t *rundata);

int miso_run_paired(const miso_gene_t *gene, const int *position, const char *const *cigarstr,
                    int n_positions, int readLength, int overHang, int noChains,
                    int noIterations, int maxIterations, int noBurnIn, int noLag,
                    const double *hyperp, int n_hyperp, int start, int stop, double normalMean,
                    double normalVar, double numDevs, uint64_t seed, double *samples,
                    double *logLik, double *bin_class_templates, double *bin_class_counts,
                    int *n_classes, int *assignment, miso_rundata_t *rundata);

/* ---- many events per launch ---- */
typedef struct miso_params {
  int paired;                /* 0 single-end (splicing_miso), 1 paired-end (splicing_miso_paired) */
  int readLength, overHang;
  int noChains, noIterations, maxIterations, noBurnIn, noLag;
  int algorithm, start, stop;
  double normalMean, normalVar, numDevs; /* paired only */
  int want_counts_trace;     /* tests: keep the per-iteration assignment counts of every chain */
  int device_match;          /* 1: miso_batch_add_event only parses the CIGAR strings; the
                                compatibility of every read with every isoform (solve.c:8-108,
                                141-218) is computed by one GPU kernel for the whole batch at
                                miso_batch_upload.  Read classes (miso_batch_event_info's
                                n_classes, class templates) are then available after the upload.
                                0: on the host, at miso_batch_add_event */
} miso_params_t;

int miso_batch_create(const miso_params_t *params, miso_batch_t **batch);
void miso_batch_destroy(miso_batch_t *batch);

/* Adds one event built from alignments; hyperp NULL = all ones. *event_index = position in
   the batch (also its offset from first_event_id in the RNG address). */
int miso_batch_add_event(miso_batch_t *batch, const miso_gene_t *gene, const int *position,
                         const char *const *cigarstr, int n_positions, const double *hyperp,
                         int n_hyperp, int *event_index);

/* Adds one event from an already built problem: match noiso x n_reads as miso_match_iso[_paired]
   returns it (+ fragmentLength for paired), isoform lengths and exon counts (gff.c:583-657). */
int miso_batch_add_problem(miso_batch_t *batch, int noiso, int n_reads, const double *match,
                           const int *fragmentLength, const int *isolength, const int *noexons,
                           const double *hyperp, int *event_index);

/* Synthetic reads for one gene (the module's simulateReads / simulatePairedReads,
   pysplicing.c:280-330 -> simulator.c:68, 221): n_reads single-end reads, or n_reads PAIRS when
   normalVar > 0.  position: n (2n paired) ints; cigar: n (2n) slots of cigar_stride bytes;
   isoform (may be NULL): true isoform of every read.  Deterministic in sim_seed. */
int miso_simulate_reads(const miso_gene_t *gene, const double *expression, int n_reads,
                        int readLength, double normalMean, double normalVar, double numDevs,
                        uint64_t sim_seed, int *isoform, int *position, char *cigar,
                        int cigar_stride);

/* miso_simulate_reads + miso_batch_add_event in one call (the batch's readLength / paired /
   fragment parameters apply): bulk synthetic workloads without string traffic over the ABI. */
int miso_batch_add_simulated(miso_batch_t *batch, const miso_gene_t *gene,
                             const double *expression, int n_reads, uint64_t sim_seed,
                             const double *hyperp, int n_hyperp, int *event_index);

int miso_batch_size(const miso_batch_t *batch, int *n_events);

/* pack + copy to HBM (idempotent) */
/* The random stream of an event is addressed by (seed, event id): by default first_event_id (a launch
 * argument) + the event's index in the batch.  A caller that drops events between numbering and
 * batching (skip rules) pins the id here instead, so results do not depend on batch composition,
 * chunk size or the number of GPUs.  Before miso_batch_upload. */
int miso_batch_set_event_id(miso_batch_t *batch, int event_index, uint32_t event_id);
/* Single-end batches: on != 0 selects the COLLAPSED Gibbs step for the two-isoform events (csrc/kernels_lane.hip).
   The reference reassigns every read by itself (miso.c:30-91 inside miso.c:493-552) and then uses the per-isoform
   counts only (miso.c:243-307); reads compatible with the same isoforms are exchangeable, so their counts are drawn
   directly -- Binomial(n, psi_0 / (psi_0 + psi_1)) from the counter RNG, include/miso_binomial.h -- which is the same
   Markov chain on (psi, counts) at O(1) instead of O(reads) per iteration.  The run's last reassignment is made per
   read, so the returned assignment is a per-read draw.  Different draws than the default mode (same distribution):
   checked bit for bit against the checker's collapsed mode and statistically against the reference.  Events with
   more than two isoforms of such a batch run as always -- unless on == 2: then they draw, per compatibility class, a
   chain of binomials (sampler_lane_k; same contract, same checker).  That pays from ~10^4 reads per event only: with
   ~1000 reads spread over the classes of a five- or ten-isoform event it is slower than the per-read sweep
   (profiles/r03_collapsed.txt).  Before miso_batch_launch; MISO_EINVAL for paired-end. */
int miso_batch_set_collapsed(miso_batch_t *batch, int on);
int miso_batch_upload(miso_batch_t *batch, int device);
/* enqueue the sampler kernels for every event on the batch's stream; returns immediately */
int miso_batch_launch(miso_batch_t *batch, uint64_t seed, uint32_t first_event_id);
/* wait for the stream; *kernel_ms (may be NULL) = HIP-event time of the launches */
int miso_batch_sync(miso_batch_t *batch, float *kernel_ms);
/* copy results back to the host (after sync) */
int miso_batch_download(miso_batch_t *batch);
/* upload + launch + sync + download */
int miso_batch_run(miso_batch_t *batch, int device, uint64_t seed, uint32_t first_event_id);

int miso_batch_event_info(const miso_batch_t *batch, int event_index, int *noiso, int *n_reads,
                          int *n_samples, int *n_classes);
int miso_batch_get_result(const miso_batch_t *batch, int event_index, double *samples,
                          double *logLik, double *class_templates, double *class_counts,
                          int *assignment, miso_rundata_t *rundata);

/* The `.miso` files of many events at once (misopy/miso_sampler.py:444-464, output_miso_results):
 * for each j < n, paths[j] receives headers[j] (the caller's "#isoforms=...\n" line), the column
 * line "sampled_psi\tlog_score\n" and one row per kept sample, "%.4f,%.4f,...\t%.2f\n", with the
 * digits Python's % operator prints (nan for NaN).  The rows are formatted and written by n_threads
 * host threads (<= 0: all cores): at 5000 rows per event the reference's Python loop is the slowest
 * step of a whole-genome run once sampling takes milliseconds.  Needs downloaded results. */
int miso_batch_write_miso_files(const miso_batch_t *batch, int n, const int *event_index,
                                const char *const *paths, const char *const *headers, int n_threads);
/* tests: the writer's number formatter on its own ("%.2f" / "%.4f"), NUL-terminated strings `stride`
 * bytes apart */
int miso_selftest_format(const double *x, int n, int decimals, char *out, int stride);
/* tests: the stopping rule of stop = MISO_STOP_CONVERGENT_MEAN on its own (splicing_i_check_convergent_mean,
 * miso.c:556-636): samples = noSamples columns of noiso values, column i from chain i % noChains; *stop = 1 converged.
 * Host arithmetic, no device needed. */
int miso_selftest_convergent_mean(const double *samples, int noiso, int noChains, int noSamples, int *stop);
/* parity instrumentation: FNV-1a over every chain's per-iteration assignment counts
   (counts_hash: noChains words) and, if want_counts_trace, the counts themselves
   ((noIterations+1) x noChains x noiso int32, row m = counts the MH step of iteration m saw,
   last row = final state). */
int miso_batch_get_trace(const miso_batch_t *batch, int event_index, uint64_t *counts_hash,
                         int32_t *counts_trace);

/* Posterior summaries on the device, from the samples of the last launch (after miso_batch_sync;
   no miso_batch_download needed): what summarize_miso computes per event
   (misopy/credible_intervals.py:4-72): per isoform the mean of psi and the Chen-Shao credible
   interval = order statistics int(round(alpha/2 n)) - 1 and int(round((1-alpha/2) n)) - 1 of the
   sorted samples, alpha = 1 - confidence_level.  mean / ci_low / ci_high: noiso doubles each. */
int miso_batch_summarize(miso_batch_t *batch, double confidence_level);
/* The same summaries of the samples AS THE `.miso` FILE HANDS THEM ON: summarize_miso never sees the sampler's
   doubles, it parses the file's "%.4f" text (misopy/samples_utils.py:130-262).  Every sample is first rounded to
   four decimals exactly as a correctly rounded "%.4f" prints it and read back as the nearest double; the credible
   interval bounds are order statistics of those values, the mean their exact sum / n: what the reference computes
   from the file this run writes, bit for bit for the bounds, to the last bit or two of a float sum for the mean. */
int miso_batch_summarize_as_text(miso_batch_t *batch, double confidence_level);
int miso_batch_get_summary(const miso_batch_t *batch, int event_index, double *mean, double *ci_low,
                           double *ci_high);

/* Samples that were produced elsewhere -- parsed `.miso` files: summarize_miso and compare_miso work on directories of
   them (misopy/samples_utils.py:263-329, hypothesis_test.py:186-345).  Event i has noiso[i] isoforms and n_samples
   samples in the file's layout (samples[i]: n_samples rows of noiso[i] values).  The batch lives on `device` and
   supports miso_batch_summarize[_as_text], miso_batch_compare and their getters only. */
int miso_batch_from_samples(int n_events, const int *noiso, int n_samples, const double *const *samples, int device,
                            miso_batch_t **batch);

/* Two-sample comparison on the device (compare_miso, misopy/hypothesis_test.py:89-179, 348-380):
   `sample1` and `sample2` hold the SAME events in the same order (one batch per RNA-seq sample),
   both launched on the same device.  Per event and isoform: index-paired delta = psi1 - psi2;
   mean|delta| <= 0.009 or constant delta -> Bayes factor 0 (posterior peaked on the null);
   otherwise Gaussian KDE of delta with covariance factor `smoothing` (reference: 0.3) evaluated at
   0, BF = 1 / density, 1e12 if the density is 0, capped at 1e12.  Results are stored in sample1. */
int miso_batch_compare(miso_batch_t *sample1, miso_batch_t *sample2, double smoothing);
int miso_batch_get_comparison(const miso_batch_t *sample1, int event_index, double *mean1, double *mean2,
                              double *bayes_factor, double *density_at_0);   /* noiso doubles each */

/* device_match batches: kernel time of the matching launch done by miso_batch_upload, and (tests:
   want_counts_trace batches only) the kernel's output for event i in the layout of
   miso_match_iso[_paired]: match noiso x n_reads, fragmentLength likewise or NULL. */
int miso_batch_last_match_ms(const miso_batch_t *batch, float *ms);
int miso_batch_get_match(const miso_batch_t *batch, int event_index, double *match, int *fragmentLength);

/* The run-dependent fields of the `.miso` header line (misopy/miso_sampler.py:376-454) of n events in one call.  Writes,
   per event, the line "U<TAB>percent_accept<TAB>counts<TAB>assigned_counts\n" into buf (NUL-terminated): U = 1 when every
   read of the event is unassigned (the caller skips it, miso_sampler.py:352-354); percent_accept as "%.2f"; counts =
   "(1,0):12,(0,1):34" (read class : reads); assigned_counts = "0:5,1:41" (reads_utils.py:37-46).  *needed = bytes the
   text takes incl. the NUL; nothing is written when cap is smaller (call again).  Needs downloaded results. */
int miso_batch_header_fields(const miso_batch_t *batch, int n, const int *event_index, char *buf, int64_t cap,
                             int64_t *needed);

/* names of the kernels the last launch used (for profiles): e.g. "sampler_k2<3, false>" */
int miso_batch_last_kernels(const miso_batch_t *batch, char *buf, int buflen);

/* How many launches of this batch miso_batch_sync() had to repeat because a chain spread over several workgroups
   (the events with 10^4 ... 10^5 reads) did not get all of them resident in time on a busy device.  The repeat runs
   in the same process with one workgroup per chain and returns the same results bit for bit; the batch never fails
   for it -- the reference's workers share nothing either (misopy/miso.py:165-187).  0 on an idle device. */
int miso_batch_coop_retries(const miso_batch_t *batch, int *n);

/* stop = MISO_STOP_CONVERGENT_MEAN (miso.c:903-925, miso_paired.c:501-523): after a launch whose chains have not
   converged for some events -- splicing_i_check_convergent_mean on their kept samples --, miso_batch_sync() runs those
   events again with noIterations' = 3 noIterations - 2 noBurnIn, noBurnIn' = noIterations (while noIterations <
   maxIterations), and the LAST noSamples of that round replace the event's samples (miso.c:976-983): every getter,
   the summaries and the file writer see a batch of noSamples samples per event, as the reference returns them.
   rundata.noAccepted / noRejected count the last round (paired-end: all rounds), as the reference's do.
   *rounds = rounds the slowest event of the last launch took (1: every event converged on its own schedule, and
   always with MISO_STOP_FIXEDNO). */
int miso_batch_rounds(const miso_batch_t *batch, int *rounds);

/* Measurement: what the last launch put on the device, kernel by kernel (bench.py's VALU roofline
   prices it with the kernels' instruction counts, tools/isa_count.py): wavefronts launched, the sum
   over wavefronts of the read loop's trips per Gibbs step, Gibbs steps run (noIterations + 1),
   chains, and the Philox words (uniforms) the read loops generate per Gibbs step. */
typedef struct {
  char name[64];
  double waves, trips, iterations, chains, words;
} miso_kernel_stat_t;
int miso_batch_launch_stats(const miso_batch_t *batch, miso_kernel_stat_t *stats, int max_kernels,
                            int *n_kernels);
/* The version of the counter-mode contract this library draws by (include/miso_philox.h MISO_CONTRACT_VERSION): seeded
   results are comparable between builds -- and with the CPU checker -- only at equal versions.  No device needed. */
int miso_contract_version(void);

/* Measurement: the shader clock the last launch ran at.  With the probe on, every launch starts ONE extra wavefront on a
   stream of its own that sleeps beside the sampler kernels and reads the shader-cycle counter (s_memtime) and the
   constant reference clock (hipDeviceAttributeWallClockRate) at both ends of the launch; miso_batch_sync() turns the pair
   into cycles per nanosecond.  *shader_ghz = 0 when the probe is off or its window did not cover the launch (its stream
   shared the batch's hardware queue, a profiler serialised the dispatches); *window_ms = what it covered.  The reference
   has no counterpart (a CPU's clock is not part of its results either): bench.py prices its VALU roofline with it. */
int miso_batch_set_clock_probe(miso_batch_t *batch, int on);
int miso_batch_last_clock(const miso_batch_t *batch, double *shader_ghz, double *window_ms);
/* Host arithmetic, no device needed: the lanes-per-chain plan of the two-isoform sampler's one-launch-many-widths
   kernel (sampler_k2_multi) for a list of events ordered by drawing reads, most first -- the answer to "events are
   independent and cost O(reads)" (miso.c:845-900) on a machine whose unit of work is a 64-lane wavefront.
   n_draw[n_events]: reads with two compatible isoforms per event; chains per event; paired: 0 / 1;
   resident_workgroups: what the device holds at once (256 single-end, 512 paired-end on MI355X);
   max_chains_per_wave: LDS limit (64 = none); cost5: {per block, step with 1, 2, 3, >= 4 cooperating lanes} in VALU
   instructions or NULL for the measured defaults; forced_target > 0 fixes the bound on a wavefront's step.
   Out: *n_runs runs (<= 16); run r holds events [run_first_event[r], run_first_event[r + 1]) on workgroups
   [run_first_workgroup[r], run_first_workgroup[r + 1]) with run_lanes[r] lanes per chain (512 = one chain per
   workgroup); both first_* arrays have *n_runs + 1 entries (caller provides 17); estimate3 (may be NULL):
   {sum of wavefront steps, longest wavefront step, 1 = one round of resident workgroups / 2 = several}. */
int miso_plan_lanes(const int *n_draw, int n_events, int chains, int paired, int resident_workgroups,
                    int max_chains_per_wave, const double *cost5, double forced_target, int *n_runs,
                    int *run_first_event, int *run_first_workgroup, int *run_lanes, double *estimate3);

/* placement diagnostics: HW_REG_HW_ID of the wavefront that ran each chain of event i (noChains
   words; 0 for kernels that do not record it).  Needs downloaded results. */
int miso_batch_get_placement(const miso_batch_t *batch, int event_index, uint32_t *hw_id);

/* bytes the kernels of the last launch moved by the reference algorithm's accounting
   (SE (8K+20)N, PE (8K+28)N per chain-iteration + load/store) */
int miso_batch_algorithmic_bytes(const miso_batch_t *batch, double *bytes);

/* device-side self test of the arithmetic contract: evaluates miso_detmath / Philox on the GPU
   for n inputs; out_* are host arrays of n doubles (used by tests/test_gpu_contract.py) */
int miso_selftest_detmath(const double *x, int n, double *out_exp, double *out_log,
                          double *out_sqrt, double *out_qnorm);
int miso_selftest_philox(const uint32_t *ctr_key6, int n, uint32_t *out4);

#ifdef __cplusplus
}
#endif
#endif /* MISO_AMD_H */
