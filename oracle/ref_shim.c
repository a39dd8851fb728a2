/*
 * ref_shim.c -- TEST INFRASTRUCTURE, never shipped, never on the product path.
 *
 * Plain-pointer entry points around the *real* reference C core
 * (the .c files of /root/reference/pysplicing/src), compiled in place by oracle/Makefile into
 * oracle/_ref/libmiso_ref.so.  No reference source is copied: this file only
 * #includes the reference's public headers at build time and calls
 *   splicing_create_gene      (simulator.c:9)
 *   splicing_matchIso         (solve.c:8)      splicing_matchIso_paired (solve.c:141)
 *   splicing_miso             (miso.c:638)     splicing_miso_paired     (miso_paired.c:241)
 *   splicing_simulate_reads   (simulator.c:68) splicing_simulate_paired_reads (simulator.c:221)
 *   splicing_rng_seed / RNG_UNIF01 / RNG_NORMAL (random.c:301-448, 727-785, 1543)
 * so that tests can (a) pin oracle/miso_oracle.c against the reference bit for bit and
 * (b) time the reference as bench.py's cpu_baseline (kind "reference").
 *
 * All arrays are caller-allocated; matrices use the reference's column-major layout
 * (splicing_matrix.h:67  MATRIX(m,i,j) = data[j*nrow+i]).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "splicing.h"
#include "splicing_error.h"
#include "splicing_random.h"

static int g_handler_set = 0;

static void ref_quiet_handler(const char *reason, const char *file, int line,
                              int err) {
  (void) reason; (void) file; (void) line; (void) err;
  SPLICING_FINALLY_FREE();
}

static void ref_quiet_warning(const char *reason, const char *file, int line,
                              int err) {
  (void) reason; (void) file; (void) line; (void) err;
}

static void ref_init(void) {
  if (!g_handler_set) {
    splicing_set_error_handler(ref_quiet_handler);
    splicing_set_warning_handler(ref_quiet_warning);
    g_handler_set = 1;
  }
}

/* ---- RNG ---------------------------------------------------------------- */

int ref_rng_seed(unsigned long seed) {
  ref_init();
  return splicing_rng_seed(&splicing_rng_default, seed);
}
double ref_unif01(void) { return RNG_UNIF01(); }
double ref_normal01(void) { return RNG_NORMAL(0, 1); }
long ref_integer(long l, long h) { return RNG_INTEGER(l, h); }
double ref_dnorm(double x, double mu, double sd) { return splicing_dnorm(x, mu, sd); }

/* ---- gene handle ---------------------------------------------------------- */

/* exons: 2*nex ints (start,end 1-based inclusive); isoforms: exon indices, each
   isoform terminated by -1 (pyconvert.c:55-87 builds exactly this). */
void *ref_gene_create(const int *exons, int nex, const int *isoforms, int nisoflat) {
  splicing_vector_int_t ex, iso;
  splicing_gff_t *gff;
  int i;
  ref_init();
  gff = malloc(sizeof(splicing_gff_t));
  if (!gff) return 0;
  if (splicing_gff_init(gff, 0)) { free(gff); return 0; }
  splicing_vector_int_init(&ex, 2 * nex);
  splicing_vector_int_init(&iso, nisoflat);
  for (i = 0; i < 2 * nex; i++) VECTOR(ex)[i] = exons[i];
  for (i = 0; i < nisoflat; i++) VECTOR(iso)[i] = isoforms[i];
  if (splicing_create_gene(&ex, &iso, "insilicogene", "seq1", "protein_coding",
                           SPLICING_STRAND_UNKNOWN, gff)) {
    splicing_vector_int_destroy(&ex); splicing_vector_int_destroy(&iso);
    splicing_gff_destroy(gff); free(gff); return 0;
  }
  splicing_vector_int_destroy(&ex);
  splicing_vector_int_destroy(&iso);
  return gff;
}

void ref_gene_destroy(void *g) { if (g) splicing_gff_destroy2(g); }

int ref_gene_noiso(void *g) {
  size_t n = 0;
  if (splicing_gff_noiso_one((splicing_gff_t *) g, 0, &n)) return -1;
  return (int) n;
}

int ref_gene_isolength(void *g, int *out) {
  splicing_vector_int_t v; int i, n, rc;
  splicing_vector_int_init(&v, 0);
  rc = splicing_gff_isolength_one((splicing_gff_t *) g, 0, &v);
  n = (int) splicing_vector_int_size(&v);
  for (i = 0; i < n; i++) out[i] = VECTOR(v)[i];
  splicing_vector_int_destroy(&v);
  return rc ? -1 : n;
}

/* ---- helpers -------------------------------------------------------------- */

static void fill_ivec(splicing_vector_int_t *v, const int *src, int n) {
  int i; splicing_vector_int_init(v, n);
  for (i = 0; i < n; i++) VECTOR(*v)[i] = src[i];
}
static void fill_vec(splicing_vector_t *v, const double *src, int n) {
  int i; splicing_vector_init(v, n);
  for (i = 0; i < n; i++) VECTOR(*v)[i] = src[i];
}
static void rundata_out(const splicing_miso_rundata_t *r, int *out) {
  out[0] = r->noIso; out[1] = r->noIters; out[2] = r->maxIters; out[3] = r->noBurnIn;
  out[4] = r->noLag; out[5] = r->noAccepted; out[6] = r->noRejected;
  out[7] = r->noChains; out[8] = r->noSamples;
}

/* ---- match matrices ------------------------------------------------------- */

int ref_match_iso(void *g, const int *pos, const char **cigar, int nreads,
                  int overHang, int readLength, double *match /* K x nreads */) {
  splicing_vector_int_t p; splicing_matrix_t m; int rc, n;
  fill_ivec(&p, pos, nreads);
  splicing_matrix_init(&m, 0, 0);
  rc = splicing_matchIso((splicing_gff_t *) g, 0, &p, cigar, overHang, readLength, &m);
  if (!rc) {
    n = (int) (splicing_matrix_nrow(&m) * splicing_matrix_ncol(&m));
    memcpy(match, &MATRIX(m, 0, 0), sizeof(double) * n);
  }
  splicing_matrix_destroy(&m);
  splicing_vector_int_destroy(&p);
  return rc;
}

int ref_match_iso_paired(void *g, const int *pos, const char **cigar, int npos,
                         int readLength, int overHang, double mean, double var,
                         double numDevs, double *match /* K x npos/2 */,
                         int *fraglen /* K x npos/2 */) {
  splicing_vector_int_t p; splicing_matrix_t m; splicing_matrix_int_t f; int rc;
  fill_ivec(&p, pos, npos);
  splicing_matrix_init(&m, 0, 0);
  splicing_matrix_int_init(&f, 0, 0);
  rc = splicing_matchIso_paired((splicing_gff_t *) g, 0, &p, cigar, readLength,
                                overHang, 0, 0, mean, var, numDevs, &m, &f);
  if (!rc) {
    long n = splicing_matrix_nrow(&m) * splicing_matrix_ncol(&m);
    memcpy(match, &MATRIX(m, 0, 0), sizeof(double) * n);
    memcpy(fraglen, &MATRIX(f, 0, 0), sizeof(int) * n);
  }
  splicing_matrix_int_destroy(&f);
  splicing_matrix_destroy(&m);
  splicing_vector_int_destroy(&p);
  return rc;
}

/* ---- simulators ----------------------------------------------------------- */

/* cigar_out: nreads slots of cigar_stride bytes each */
int ref_simulate_reads(void *g, const double *expr, int K, int nreads, int readLength,
                       int *isoform, int *pos, char *cigar_out, int cigar_stride) {
  splicing_vector_t e; splicing_vector_int_t iso, p; splicing_strvector_t c; int rc, i;
  fill_vec(&e, expr, K);
  splicing_vector_int_init(&iso, 0); splicing_vector_int_init(&p, 0);
  splicing_strvector_init(&c, 0);
  rc = splicing_simulate_reads((splicing_gff_t *) g, 0, &e, nreads, readLength,
                               &iso, &p, &c, 0);
  if (!rc) {
    for (i = 0; i < nreads; i++) {
      isoform[i] = VECTOR(iso)[i]; pos[i] = VECTOR(p)[i];
      strncpy(cigar_out + (size_t) i * cigar_stride, splicing_strvector_get(&c, i),
              cigar_stride - 1);
      cigar_out[(size_t) i * cigar_stride + cigar_stride - 1] = 0;
    }
  }
  splicing_strvector_destroy(&c);
  splicing_vector_int_destroy(&p); splicing_vector_int_destroy(&iso);
  splicing_vector_destroy(&e);
  return rc;
}

int ref_simulate_paired_reads(void *g, const double *expr, int K, int npairs,
                              int readLength, double mean, double var, double numDevs,
                              int *isoform, int *pos, char *cigar_out,
                              int cigar_stride) {
  splicing_vector_t e; splicing_vector_int_t iso, p; splicing_strvector_t c; int rc, i;
  fill_vec(&e, expr, K);
  splicing_vector_int_init(&iso, 0); splicing_vector_int_init(&p, 0);
  splicing_strvector_init(&c, 0);
  rc = splicing_simulate_paired_reads((splicing_gff_t *) g, 0, &e, npairs, readLength,
                                      0, 0, mean, var, numDevs, &iso, &p, &c, 0);
  if (!rc) {
    for (i = 0; i < 2 * npairs; i++) {
      isoform[i] = VECTOR(iso)[i]; pos[i] = VECTOR(p)[i];
      strncpy(cigar_out + (size_t) i * cigar_stride, splicing_strvector_get(&c, i),
              cigar_stride - 1);
      cigar_out[(size_t) i * cigar_stride + cigar_stride - 1] = 0;
    }
  }
  splicing_strvector_destroy(&c);
  splicing_vector_int_destroy(&p); splicing_vector_int_destroy(&iso);
  splicing_vector_destroy(&e);
  return rc;
}

/* the stopping rule of STOP_CONVERGENT_MEAN on its own (miso.c:556): samples K x n col-major */
int ref_convergent_mean(const double *samples, int K, int C, int n) {
  splicing_matrix_t s, means, vars; int stop = -1;
  splicing_matrix_init(&s, K, n); splicing_matrix_init(&means, K, C); splicing_matrix_init(&vars, K, C);
  memcpy(&MATRIX(s, 0, 0), samples, sizeof(double) * K * n);
  if (splicing_i_check_convergent_mean(&means, &vars, &s, &stop)) stop = -1;
  splicing_matrix_destroy(&vars); splicing_matrix_destroy(&means); splicing_matrix_destroy(&s);
  return stop;
}

/* splicing_assignment_matrix (assignment.c:90-276): noiso x ncol, column-major; returns ncol or -1 - code */
int ref_assignment_matrix(void *g, int readLength, int overHang, double *out, int max_cols) {
  splicing_matrix_t m; int rc, nc, K;
  splicing_matrix_init(&m, 0, 0);
  rc = splicing_assignment_matrix((splicing_gff_t *) g, 0, readLength, overHang, &m);
  if (rc) { splicing_matrix_destroy(&m); return -1 - rc; }
  nc = (int) splicing_matrix_ncol(&m); K = (int) splicing_matrix_nrow(&m);
  if (nc <= max_cols) memcpy(out, &MATRIX(m, 0, 0), sizeof(double) * (size_t) K * nc);
  splicing_matrix_destroy(&m);
  return nc;
}

/* splicing_score_joint for SPLICING_ALGO_CLASSES (miso.c:243-307, branch :284-295) on caller-made inputs -- the
   sampler itself reads its per-class read counts before initialising them (miso.c:790), so the formula is pinned here:
   amat K x ncls column-major (rows already normalised), matches ncls, one chain */
double ref_score_classes(int K, const double *psi, const double *hyper, const double *amat, int ncls,
                         const double *matches) {
  splicing_matrix_t mpsi, ma, mm; splicing_vector_t h, m, sc, iso; splicing_vector_int_t eff; splicing_matrix_int_t ass;
  double out; int i;
  ref_init();
  splicing_matrix_init(&mpsi, K, 1); splicing_matrix_init(&ma, K, ncls); splicing_matrix_init(&mm, K, 1);
  fill_vec(&h, hyper, K); fill_vec(&m, matches, ncls);
  splicing_vector_init(&sc, 1); splicing_vector_init(&iso, K); splicing_vector_int_init(&eff, K);
  splicing_matrix_int_init(&ass, 1, 1);
  for (i = 0; i < K; i++) MATRIX(mpsi, i, 0) = psi[i];
  memcpy(&MATRIX(ma, 0, 0), amat, sizeof(double) * (size_t) K * ncls);
  splicing_score_joint(SPLICING_ALGO_CLASSES, &ass, 0, 1, &mpsi, &h, &eff, &iso, &mm, &ma, &m, &sc);
  out = VECTOR(sc)[0];
  splicing_matrix_int_destroy(&ass); splicing_vector_int_destroy(&eff); splicing_vector_destroy(&iso);
  splicing_vector_destroy(&sc); splicing_vector_destroy(&m); splicing_vector_destroy(&h);
  splicing_matrix_destroy(&mm); splicing_matrix_destroy(&ma); splicing_matrix_destroy(&mpsi);
  return out;
}

/* ---- the samplers --------------------------------------------------------- */

/* samples: K x S col-major, S = C*(M-B)/lag; class_templates: K x ncls col-major
   (caller gives room for K x nreads); rundata: 9 ints in struct order. */
int ref_miso(void *g, const int *pos, const char **cigar, int nreads, int readLength,
             int overHang, int noChains, int noIterations, int maxIterations,
             int noBurnIn, int noLag, const double *hyper, int K, int algorithm,
             int start, int stop, double *samples, double *logLik,
             double *match /* K x nreads or NULL */, double *class_templates,
             double *class_counts, int *ncls, int *assignment, int *rundata) {
  splicing_vector_int_t p, ass; splicing_vector_t h, ll, cc;
  splicing_matrix_t s, mm, ct; splicing_miso_rundata_t rd; int rc, i;
  memset(&rd, 0, sizeof(rd));
  fill_ivec(&p, pos, nreads); fill_vec(&h, hyper, K);
  splicing_vector_int_init(&ass, 0); splicing_vector_init(&ll, 0);
  splicing_vector_init(&cc, 0); splicing_matrix_init(&s, 0, 0);
  splicing_matrix_init(&mm, 0, 0); splicing_matrix_init(&ct, 0, 0);
  rc = splicing_miso((splicing_gff_t *) g, 0, &p, cigar, readLength, overHang,
                     noChains, noIterations, maxIterations, noBurnIn, noLag, &h,
                     (splicing_algorithm_t) algorithm, (splicing_miso_start_t) start,
                     (splicing_miso_stop_t) stop, 0, &s, &ll, &mm, &ct, &cc, &ass, &rd);
  if (!rc) {
    long ns = splicing_matrix_ncol(&s), nc = splicing_matrix_ncol(&ct);
    memcpy(samples, &MATRIX(s, 0, 0), sizeof(double) * K * ns);
    memcpy(logLik, VECTOR(ll), sizeof(double) * ns);
    if (match) memcpy(match, &MATRIX(mm, 0, 0), sizeof(double) * K * nreads);
    if (class_templates) memcpy(class_templates, &MATRIX(ct, 0, 0), sizeof(double) * K * nc);
    if (class_counts) memcpy(class_counts, VECTOR(cc), sizeof(double) * nc);
    if (ncls) *ncls = (int) nc;
    for (i = 0; i < nreads; i++) assignment[i] = VECTOR(ass)[i];
    rundata_out(&rd, rundata);
  }
  splicing_matrix_destroy(&ct); splicing_matrix_destroy(&mm); splicing_matrix_destroy(&s);
  splicing_vector_destroy(&cc); splicing_vector_destroy(&ll);
  splicing_vector_int_destroy(&ass); splicing_vector_destroy(&h);
  splicing_vector_int_destroy(&p);
  return rc;
}

int ref_miso_paired(void *g, const int *pos, const char **cigar, int npos,
                    int readLength, int overHang, int noChains, int noIterations,
                    int maxIterations, int noBurnIn, int noLag, const double *hyper,
                    int K, int start, int stop, double mean, double var, double numDevs,
                    double *samples, double *logLik, double *match /* K x npos/2 */,
                    double *bin_templates, double *bin_counts, int *ncls,
                    int *assignment, int *rundata) {
  splicing_vector_int_t p, ass; splicing_vector_t h, ll, cc;
  splicing_matrix_t s, mm, ct; splicing_miso_rundata_t rd; int rc, i, n = npos / 2;
  memset(&rd, 0, sizeof(rd));
  fill_ivec(&p, pos, npos); fill_vec(&h, hyper, K);
  splicing_vector_int_init(&ass, 0); splicing_vector_init(&ll, 0);
  splicing_vector_init(&cc, 0); splicing_matrix_init(&s, 0, 0);
  splicing_matrix_init(&mm, 0, 0); splicing_matrix_init(&ct, 0, 0);
  rc = splicing_miso_paired((splicing_gff_t *) g, 0, &p, cigar, readLength, overHang,
                            noChains, noIterations, maxIterations, noBurnIn, noLag, &h,
                            (splicing_miso_start_t) start, (splicing_miso_stop_t) stop,
                            0, 0, 0, mean, var, numDevs, &s, &ll, &mm, 0, 0, &ct, &cc,
                            &ass, &rd);
  if (!rc) {
    long ns = splicing_matrix_ncol(&s), nc = splicing_matrix_ncol(&ct);
    memcpy(samples, &MATRIX(s, 0, 0), sizeof(double) * K * ns);
    memcpy(logLik, VECTOR(ll), sizeof(double) * ns);
    if (match) memcpy(match, &MATRIX(mm, 0, 0), sizeof(double) * K * n);
    if (bin_templates) memcpy(bin_templates, &MATRIX(ct, 0, 0), sizeof(double) * K * nc);
    if (bin_counts) memcpy(bin_counts, VECTOR(cc), sizeof(double) * nc);
    if (ncls) *ncls = (int) nc;
    for (i = 0; i < n; i++) assignment[i] = VECTOR(ass)[i];
    rundata_out(&rd, rundata);
  }
  splicing_matrix_destroy(&ct); splicing_matrix_destroy(&mm); splicing_matrix_destroy(&s);
  splicing_vector_destroy(&cc); splicing_vector_destroy(&ll);
  splicing_vector_int_destroy(&ass); splicing_vector_destroy(&h);
  splicing_vector_int_destroy(&p);
  return rc;
}
