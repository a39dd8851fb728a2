/*
 * miso_oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's posterior-sampler path
 * (/root/reference/pysplicing/src/{miso,miso_paired,solve,gff,simulator,util,random}.c).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (miso_amd/, include/) never does.
 *
 * Parity status: PINNED.  In ORC_MODE_STREAM the oracle reproduces the real reference
 * (oracle/_ref/libmiso_ref.so, built from the reference's own sources) bit for bit on psi
 * samples, class templates/counts, match matrices and run data for identical reads and
 * MT19937 seed (tests/test_oracle_vs_ref.py, tests/golden/).  ORC_MODE_COUNTER is the same
 * code with three documented switches (RNG addressing, deterministic log/exp, count-based
 * score sums) and is what the HIP kernels must match bit for bit.
 */
#ifndef MISO_ORACLE_H
#define MISO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes = the reference's (include/splicing_error.h:314-351) */
/* most isoforms a gene may have here (fixed-size scratch in the restatement; the reference has no limit, miso.c:696) */
#define ORC_MAXK 256
#define ORC_SUCCESS 0
#define ORC_FAILURE 1
#define ORC_ENOMEM 2
#define ORC_EINVAL 4
#define ORC_UNIMPLEMENTED 12

/* enums = the reference's (include/splicing.h:59-62, 148-158) */
#define ORC_ALGO_REASSIGN 0
#define ORC_START_AUTO 0
#define ORC_START_UNIFORM 1
#define ORC_STOP_FIXEDNO 0

#define ORC_MODE_STREAM 0  /* MT19937 sequential stream, libm, per-read sums: == reference */
#define ORC_MODE_COUNTER 1 /* Philox addressed draws, miso_detmath, count sums: == device  */
#define ORC_MODE_COLLAPSED 2 /* counter mode with the COLLAPSED Gibbs step (single-end): per compatibility class the
                                counts are drawn as a chain of binomials (include/miso_binomial.h) instead of one
                                uniform per read; the run's LAST reassignment is per read, so that the returned
                                assignment is a per-read draw as in the reference: == device, collapsed batches */

typedef struct orc_gene orc_gene_t;

/* run-time options that have no counterpart in the reference's signatures */
typedef struct {
  int mode;           /* ORC_MODE_* */
  uint64_t seed;      /* counter mode */
  uint32_t event_id;  /* counter mode: global index of this event in the run */
  int per_read_sums;  /* counter mode only: 1 = keep the reference's per-read score sums
                         (used to show count sums agree to rounding) */
} orc_opts_t;

/* optional extra outputs for parity tests (any pointer may be NULL) */
typedef struct {
  int32_t *counts_trace;   /* [(M+1) x C x K]: per-isoform assignment counts at the start of
                              iteration m (row m < M) and after the last re-assignment (row M) */
  uint64_t *counts_hash;   /* [C]: FNV-1a over the rows above, per chain */
  double *final_psi;       /* [K x C] */
  int32_t *accepted;       /* [C] */
} orc_trace_t;

/* --- stream RNG (random.c:301-448, 727-785, 1543-1551) --- */
void orc_rng_seed(unsigned long seed);
double orc_unif01(void);
double orc_normal01(void);
long orc_integer(long l, long h);

/* --- gene model (simulator.c:9-66, gff.c:583-777) --- */
orc_gene_t *orc_gene_create(const int *exons, int nex, const int *isoforms, int nisoflat);
void orc_gene_destroy(orc_gene_t *g);
int orc_gene_noiso(const orc_gene_t *g);
int orc_gene_isolength(const orc_gene_t *g, int *out);

/* --- problem construction (solve.c:8-108, 141-306; gff.c:855-1084; simulator.c:198-219) --- */
int orc_parse_cigar(const char **cigar, int noreads, int maxReadLength, int **numcigar,
                    int **cigaridx, int **cigarlength);
int orc_match_iso(const orc_gene_t *g, const int *pos, const char **cigar, int nreads,
                  int overHang, int readLength, double *match);
int orc_normal_fragment(double mean, double var, double numDevs, int minLength,
                        double **fragmentProb, int *fragmentStart, int *il);
int orc_match_iso_paired(const orc_gene_t *g, const int *pos, const char **cigar, int npos,
                         int readLength, int overHang, double mean, double var,
                         double numDevs, double *match, int *fraglen);

/* --- simulators (simulator.c:68-196, 221-442), stream RNG --- */
int orc_simulate_reads(const orc_gene_t *g, const double *expr, int nreads, int readLength,
                       int *isoform, int *pos, char *cigar_out, int cigar_stride);
int orc_simulate_paired_reads(const orc_gene_t *g, const double *expr, int npairs,
                              int readLength, double mean, double var, double numDevs,
                              int *isoform, int *pos, char *cigar_out, int cigar_stride);

/* --- the samplers (miso.c:638-986, miso_paired.c:241-574) ---
   Array shapes as in oracle/ref_shim.c. rundata = 9 ints in splicing_miso_rundata_t order. */
int orc_miso(const orc_gene_t *g, const int *pos, const char **cigar, int nreads,
             int readLength, int overHang, int noChains, int noIterations, int maxIterations,
             int noBurnIn, int noLag, const double *hyper, int nhyper, int algorithm,
             int start, int stop, const orc_opts_t *opts, double *samples, double *logLik,
             double *match, double *class_templates, double *class_counts, int *ncls,
             int *assignment, int *rundata, orc_trace_t *trace);

int orc_miso_paired(const orc_gene_t *g, const int *pos, const char **cigar, int npos,
                    int readLength, int overHang, int noChains, int noIterations,
                    int maxIterations, int noBurnIn, int noLag, const double *hyper,
                    int nhyper, int start, int stop, double mean, double var, double numDevs,
                    const orc_opts_t *opts, double *samples, double *logLik, double *match,
                    double *bin_templates, double *bin_counts, int *ncls, int *assignment,
                    int *rundata, orc_trace_t *trace);

/* --- exposed pieces for unit tests --- */
double orc_qnorm_libm(double p);
double orc_qnorm_det(double p);
double orc_det_exp(double x);
double orc_det_log(double x);
double orc_det_sqrt(double x);
void orc_binomial(uint64_t seed, uint32_t event_id, int32_t n, double p, int count, int32_t *out);
void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                uint32_t *out4);
void orc_philox_r(int rounds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                  uint32_t *out4);
int orc_philox_rounds(void);
int orc_contract_version(void);   /* include/miso_philox.h MISO_CONTRACT_VERSION as this checker was built */
int orc_convergent_mean(const double *samples, int K, int C, int noSamples);
int orc_assignment_matrix(const orc_gene_t *g, int readLength, int overHang, double *out, int max_cols);
double orc_score_classes(int K, const double *psi, const double *hyper, const double *amat, int ncls, const double *matches);
uint32_t orc_split_word(uint64_t seed, uint32_t event_id, uint32_t chain, uint32_t iteration, uint32_t r);

#ifdef __cplusplus
}
#endif
#endif
