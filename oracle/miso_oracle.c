/*
 * miso_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see miso_oracle.h).
 *
 * Plain-C restatement of the reference's sampler path.  Every function names the reference
 * file:line it follows (paths relative to /root/reference/pysplicing/).  Written from the
 * behaviour of those lines, not copied: flat arrays instead of splicing_vector_t, no
 * finally-stack, explicit error returns.
 *
 * Two modes (orc_opts_t.mode):
 *   ORC_MODE_STREAM   MT19937 stream in the reference's draw order, libm, per-read score sums.
 *                     Bit-for-bit equal to the real reference (pinned by tests).
 *   ORC_MODE_COUNTER  the device contract: draws addressed through include/miso_philox.h,
 *                     log/exp/qnorm from include/miso_detmath.h, joint score summed from
 *                     per-isoform counts (single-end: miso.c:265-271 / 152-156 summed per read
 *                     equal sum_k count_k*score_k up to rounding) and, paired-end, the per-read
 *                     fragment score summed in 2^-26 fixed point (order independent, miso_philox.h).
 *   ORC_MODE_COLLAPSED counter mode with the Gibbs step collapsed over exchangeable reads (single-end): per
 *                     compatibility class the counts are Multinomial(n; psi restricted to the class), drawn as a chain
 *                     of binomials (include/miso_binomial.h) -- the per-read law of miso.c:30-91 summed over the
 *                     class's reads; the run's last reassignment is per read.  Pinned statistically against
 *                     counter mode, the quadrature and the real reference (tests/test_collapsed.py).
 */
#include "miso_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "miso_detmath.h"
#include "miso_philox.h"
#include "miso_binomial.h"

/* ------------------------------------------------------------------------------------ */
/* MT19937 stream generator: Matsumoto & Nishimura 1998, 2002 seeding (the reference's   */
/* stand-alone default generator, src/random.c:301-448, itself taken from GSL).          */
/* ------------------------------------------------------------------------------------ */

#define MT_N 624
#define MT_M 397

typedef struct { uint32_t mt[MT_N]; int mti; } orc_mt_t;

static orc_mt_t g_mt; /* one global stream, as random.c:491 splicing_rng_default */
static int g_mt_seeded = 0;

static void mt_seed(orc_mt_t *s, unsigned long seed) { /* random.c:384-404 */
  int i;
  if (seed == 0) seed = 4357;
  s->mt[0] = (uint32_t) (seed & 0xffffffffUL);
  for (i = 1; i < MT_N; i++)
    s->mt[i] = (uint32_t) (1812433253UL * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (unsigned) i);
  s->mti = MT_N;
}

static uint32_t mt_get(orc_mt_t *s) { /* random.c:315-380 */
  uint32_t y;
  if (s->mti >= MT_N) {
    int kk;
    for (kk = 0; kk < MT_N; kk++) {
      uint32_t a = s->mt[kk], b = s->mt[(kk + 1) % MT_N];
      y = (a & 0x80000000u) | (b & 0x7fffffffu);
      s->mt[kk] = s->mt[(kk + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s->mti = 0;
  }
  y = s->mt[s->mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

void orc_rng_seed(unsigned long seed) { mt_seed(&g_mt, seed); g_mt_seeded = 1; }

/* random.c:382 get_real = get / 2^32; random.c:775-785 unif01 */
double orc_unif01(void) {
  if (!g_mt_seeded) orc_rng_seed(0);
  return mt_get(&g_mt) / 4294967296.0;
}

/* random.c:700-711: (long)(get_real*(h-l+1)+l) */
long orc_integer(long l, long h) { return (long) (orc_unif01() * (double) (h - l + 1) + (double) l); }

/* random.c:750-760 */
static double orc_unif(double l, double h) { return orc_unif01() * (h - l) + l; }

/* ------------------------------------------------------------------------------------ */
/* Normal quantile: Wichura's AS241 / PPND16, the algorithm behind splicing_qnorm5      */
/* (random.c:1384-1470), parameterised by the log and sqrt implementations.             */
/* ------------------------------------------------------------------------------------ */

typedef struct {
  double (*log)(double);
  double (*exp)(double);
  double (*sqrt)(double);
} orc_math_t;

static double w_det_log(double x) { return miso_det_log(x); }
static double w_det_exp(double x) { return miso_det_exp(x); }
static double w_det_sqrt(double x) { return miso_det_sqrt(x); }

static const orc_math_t MATH_LIBM = { log, exp, sqrt };
static const orc_math_t MATH_DET = { w_det_log, w_det_exp, w_det_sqrt };

static double as241(double p, const orc_math_t *M) {
  static const double a[8] = { 3.387132872796366608, 133.14166789178437745,
    1971.5909503065514427, 13731.693765509461125, 45921.953931549871457,
    67265.770927008700853, 33430.575583588128105, 2509.0809287301226727 };
  static const double b[8] = { 1.0, 42.313330701600911252, 687.1870074920579083,
    5394.1960214247511077, 21213.794301586595867, 39307.89580009271061,
    28729.085735721942674, 5226.495278852854561 };
  static const double c[8] = { 1.42343711074968357734, 4.6303378461565452959,
    5.7694972214606914055, 3.64784832476320460504, 1.27045825245236838258,
    0.24178072517745061177, 0.0227238449892691845833, 7.7454501427834140764e-4 };
  static const double d[8] = { 1.0, 2.05319162663775882187, 1.6763848301838038494,
    0.68976733498510000455, 0.14810397642748007459, 0.0151986665636164571966,
    5.475938084995344946e-4, 1.05075007164441684324e-9 };
  static const double e[8] = { 6.6579046435011037772, 5.4637849111641143699,
    1.7848265399172913358, 0.29656057182850489123, 0.026532189526576123093,
    0.0012426609473880784386, 2.71155556874348757815e-5, 2.01033439929228813265e-7 };
  static const double f[8] = { 1.0, 0.59983220655588793769, 0.13692988092273580531,
    0.0148753612908506148525, 7.868691311456132591e-4, 1.8463183175100546818e-5,
    1.4215117583164458887e-7, 2.04426310338993978564e-15 };
  double q, r, num, den;
  const double *pn, *pd;
  int i;
  if (p != p) return p;
  if (p == 0.0) return -INFINITY;
  if (p == 1.0) return INFINITY;
  if (p < 0.0 || p > 1.0) return NAN;
  q = p - 0.5;
  if (fabs(q) <= 0.425) {
    r = 0.180625 - q * q;
    num = a[7]; den = b[7];
    for (i = 6; i >= 0; i--) { num = num * r + a[i]; den = den * r + b[i]; }
    return q * num / den;
  }
  r = (q > 0) ? 1.0 - p : p;
  r = M->sqrt(-M->log(r));
  if (r <= 5.0) { r += -1.6; pn = c; pd = d; } else { r += -5.0; pn = e; pd = f; }
  num = pn[7]; den = pd[7];
  for (i = 6; i >= 0; i--) { num = num * r + pn[i]; den = den * r + pd[i]; }
  num = num / den;
  return (q < 0.0) ? -num : num;
}

double orc_qnorm_libm(double p) { return as241(p, &MATH_LIBM); }
double orc_qnorm_det(double p) { return as241(p, &MATH_DET); }
double orc_det_exp(double x) { return miso_det_exp(x); }
double orc_det_log(double x) { return miso_det_log(x); }
double orc_det_sqrt(double x) { return miso_det_sqrt(x); }
/* log factorials for miso_binomial (include/miso_binomial.h), grown on demand */
static double *g_lf = NULL; static int g_lf_n = 0;
static const double *logfact(int n) { /* at least n + 1 entries */
  if (n + 1 > g_lf_n) {
    int cap = n + 1 < 4096 ? 4096 : 2 * (n + 1);
    g_lf = realloc(g_lf, sizeof(double) * (size_t) cap);
    miso_logfact_fill(g_lf, cap);
    g_lf_n = cap;
  }
  return g_lf;
}

/* test hook: `count` draws of Binomial(n, p) from the word stream (seed, event_id, chain 0, iteration i, MISO_SITE_COUNTS) */
void orc_binomial(uint64_t seed, uint32_t event_id, int32_t n, double p, int count, int32_t *out) {
  int i;
  for (i = 0; i < count; i++) {
    miso_ustream us;
    miso_ustream_init(&us, seed, event_id, 0, (uint32_t) i, MISO_SITE_COUNTS);
    out[i] = miso_binomial(&us, n, p, logfact(n));
  }
}
void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                uint32_t *out4) {
  miso_u32x4 o = miso_philox4x32(c0, c1, c2, c3, k0, k1);
  memcpy(out4, o.v, 16);
}
/* the same round function at any number of rounds (the published known-answer vectors exist for 7 and for 10) */
void orc_philox_r(int rounds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                  uint32_t *out4) {
  miso_u32x4 o = miso_philox4x32_r(rounds, c0, c1, c2, c3, k0, k1);
  memcpy(out4, o.v, 16);
}
int orc_philox_rounds(void) { return MISO_PHILOX_ROUNDS; }
int orc_contract_version(void) { return MISO_CONTRACT_VERSION; }

/* random.c:1543-1551 splicing_norm_rand: u = (int)(2^27 u1) + u2; qnorm(u / 2^27) */
static double norm_from_unif(double u1, double u2, const orc_math_t *M) {
  const double BIG = 134217728.0;
  double u = (double) (int) (BIG * u1) + u2;
  return as241(u / BIG, M);
}

double orc_normal01(void) {
  double u1 = orc_unif01();
  double u2 = orc_unif01();
  return norm_from_unif(u1, u2, &MATH_LIBM);
}

/* ------------------------------------------------------------------------------------ */
/* Gene model: what splicing_create_gene (simulator.c:9-66) stores and what              */
/* gff.c:583-657, 728-777 later read back for ONE gene.                                  */
/* ------------------------------------------------------------------------------------ */

struct orc_gene {
  int K;        /* isoforms */
  int *exidx;   /* K+1: isoform k owns exons exidx[k]..exidx[k+1]-1 (gff.c:728-777) */
  int *exstart; /* per isoform, in the order given */
  int *exend;
  int *isolen;  /* gff.c:583-619: sum of exon lengths */
  int *noexons; /* gff.c:624-657 */
};

orc_gene_t *orc_gene_create(const int *exons, int nex, const int *isoforms, int nisoflat) {
  orc_gene_t *g;
  int i, K = 0, tot = 0, k, p;
  if (nex <= 0 || nisoflat <= 0 || isoforms[nisoflat - 1] != -1) return 0;
  for (i = 0; i < nisoflat; i++) {
    if (isoforms[i] < 0) K++;
    else { if (isoforms[i] >= nex) return 0; tot++; }
  }
  g = calloc(1, sizeof(*g));
  if (!g) return 0;
  g->K = K;
  g->exidx = calloc(K + 1, sizeof(int));
  g->exstart = calloc(tot > 0 ? tot : 1, sizeof(int));
  g->exend = calloc(tot > 0 ? tot : 1, sizeof(int));
  g->isolen = calloc(K, sizeof(int));
  g->noexons = calloc(K, sizeof(int));
  for (i = 0, k = 0, p = 0; i < nisoflat; i++) {
    if (isoforms[i] < 0) { k++; g->exidx[k] = p; continue; }
    g->exstart[p] = exons[2 * isoforms[i]];
    g->exend[p] = exons[2 * isoforms[i] + 1];
    g->isolen[k] += g->exend[p] - g->exstart[p] + 1;
    g->noexons[k] += 1;
    p++;
  }
  return g;
}

void orc_gene_destroy(orc_gene_t *g) {
  if (!g) return;
  free(g->exidx); free(g->exstart); free(g->exend); free(g->isolen); free(g->noexons);
  free(g);
}

int orc_gene_noiso(const orc_gene_t *g) { return g->K; }
int orc_gene_isolength(const orc_gene_t *g, int *out) {
  memcpy(out, g->isolen, sizeof(int) * g->K);
  return g->K;
}

/* ------------------------------------------------------------------------------------ */
/* CIGAR parsing: solve.c:220-306                                                        */
/* ------------------------------------------------------------------------------------ */

int orc_parse_cigar(const char **cigar, int noreads, int maxReadLength, int **numcigar,
                    int **cigaridx, int **cigarlength) {
  int cap = 4 * noreads + 16, pos = 0, i;
  int *num = malloc(sizeof(int) * cap);
  int *idx = malloc(sizeof(int) * (noreads + 1));
  int *clen = malloc(sizeof(int) * (noreads > 0 ? noreads : 1));
  if (!num || !idx || !clen) { free(num); free(idx); free(clen); return ORC_ENOMEM; }
  for (i = 0; i < noreads; i++) {
    char *s = (char *) cigar[i];
    int mode = 0, len = 0; /* 0 begin, 1 middle, 2 end: S/H only at the ends */
    idx[i] = pos;
    while (*s) {
      long l = strtol(s, &s, 10);
      int clip = (*s == 'S' || *s == 'H');
      int emit = 0;
      if (mode == 0 && !clip) mode = 1;
      else if (mode == 1 && clip) mode = 2;
      else if (mode == 2 && !clip) { free(num); free(idx); free(clen); return ORC_EINVAL; }
      switch (*s) {
      case 'M': case '=': case 'X': case 'S': case 'H': case 'D':
        /* all count as matching reference bases, clipped to the read length */
        if (maxReadLength > 0 && len + l > maxReadLength) l = maxReadLength - len;
        len += (int) l; emit = 1; break;
      case 'N': l = -l; emit = 1; break;
      case 'I': break; /* ignored */
      default: free(num); free(idx); free(clen); return ORC_EINVAL;
      }
      s++;
      if (emit) {
        if (pos == cap) {
          int *t; cap *= 2; t = realloc(num, sizeof(int) * cap);
          if (!t) { free(num); free(idx); free(clen); return ORC_ENOMEM; }
          num = t;
        }
        num[pos++] = (int) l;
      }
    }
    clen[i] = len;
  }
  idx[noreads] = pos;
  *numcigar = num; *cigaridx = idx; *cigarlength = clen;
  return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* Read x isoform compatibility: solve.c:8-108                                           */
/* ------------------------------------------------------------------------------------ */

int orc_match_iso(const orc_gene_t *g, const int *pos, const char **cigar, int nreads,
                  int overHang, int readLength, double *match) {
  int *num = 0, *idx = 0, *clen = 0, r, i, rc, K = g->K;
  if (overHang == 0) overHang = 1;
  if (overHang < 1) return ORC_EINVAL;
  if (readLength < 0) return ORC_EINVAL;
  rc = orc_parse_cigar(cigar, nreads, readLength, &num, &idx, &clen);
  if (rc) return rc;
  for (r = 0; r < nreads; r++) {
    const int *cig = num + idx[r];
    int nocig = idx[r + 1] - idx[r];
    double *col = match + (size_t) r * K;
    if (clen[r] < readLength || nocig == 0 || cig[0] < overHang || cig[nocig - 1] < overHang) {
      for (i = 0; i < K; i++) col[i] = 0;
      continue;
    }
    for (i = 0; i < K; i++) {
      int c, p = pos[r], ex = g->exidx[i], last = g->exidx[i + 1], ok = 1;
      while (ex < last && (p < g->exstart[ex] || g->exend[ex] < p)) ex++;
      if (ex >= last) { col[i] = 0; continue; }
      for (c = 0; c < nocig && ok; c++) {
        if (cig[c] > 0) { /* aligned block must end inside the exon */
          if (p + cig[c] - 1 > g->exend[ex]) ok = 0; else p += cig[c];
        } else {          /* skip: must leave at the exon end and land on the next start */
          if (p != g->exend[ex] + 1) ok = 0;
          else {
            p -= cig[c]; ex += 1;
            if (ex >= last || p != g->exstart[ex]) ok = 0;
          }
        }
      }
      col[i] = ok ? 1.0 : 0.0;
    }
  }
  free(num); free(idx); free(clen);
  return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* Fragment length distribution: simulator.c:198-219, util.c:17-32                        */
/* ------------------------------------------------------------------------------------ */

static double orc_dnorm(double x, double mu, double sigma) {
  const double INV_SQRT_2PI = 0.398942280401432677939946059934;
  x = (x - mu) / sigma;
  return INV_SQRT_2PI * exp(-0.5 * x * x) / sigma;
}

/* returns the NORMALISED vector the samplers use (miso_paired.c:303-307, solve.c:162-166) */
int orc_normal_fragment(double mean, double var, double numDevs, int minLength,
                        double **fragmentProb, int *fragmentStart, int *il) {
  double sd = sqrt(var), sum = 0.0, *fp;
  int fs = (int) (mean - sd * numDevs), fe = (int) (mean + sd * numDevs), i, j, n;
  if (fs < minLength) fs = minLength;
  if (fe < fs) fe = fs;
  n = fe - fs + 1;
  fp = malloc(sizeof(double) * n);
  if (!fp) return ORC_ENOMEM;
  for (i = fs, j = 0; i <= fe; i++, j++) fp[j] = orc_dnorm(i, mean, sd);
  for (j = 0; j < n; j++) sum += fp[j];
  sum = 1.0 / sum;
  for (j = 0; j < n; j++) fp[j] *= sum;
  *fragmentProb = fp; *fragmentStart = fs; *il = n;
  return ORC_SUCCESS;
}

/* genomic -> isoform coordinate: gff.c:855-898 (shift table), gff.c:1041-1084 */
static int genomic_to_iso_one(const orc_gene_t *g, int iso, int p) {
  int ex, first = g->exidx[iso], last = g->exidx[iso + 1];
  int cs = 0, ce = 0, n = 0;
  for (ex = first; ex < last && g->exend[ex] < p; ex++) { cs += g->exstart[ex]; ce += g->exend[ex]; n++; }
  if (ex < last && g->exstart[ex] <= p && p <= g->exend[ex]) {
    cs += g->exstart[ex];
    return p - (cs - ce - n - 1);
  }
  return -1;
}

/* solve.c:141-218 */
int orc_match_iso_paired(const orc_gene_t *g, const int *pos, const char **cigar, int npos,
                         int readLength, int overHang, double mean, double var,
                         double numDevs, double *match, int *fraglen) {
  int K = g->K, n = npos / 2, r, i, rc, fs, il;
  double *fp = 0, *m1;
  rc = orc_normal_fragment(mean, var, numDevs, readLength, &fp, &fs, &il);
  if (rc) return rc;
  m1 = malloc(sizeof(double) * (size_t) K * (npos > 0 ? npos : 1));
  if (!m1) { free(fp); return ORC_ENOMEM; }
  rc = orc_match_iso(g, pos, cigar, npos, overHang, readLength, m1);
  if (rc) { free(fp); free(m1); return rc; }
  for (r = 0; r < n; r++) {
    for (i = 0; i < K; i++) {
      double v = 0.0; int fl = -1;
      if (m1[(size_t) (2 * r) * K + i] != 0 && m1[(size_t) (2 * r + 1) * K + i] != 0) {
        int frag = genomic_to_iso_one(g, i, pos[2 * r + 1]) - genomic_to_iso_one(g, i, pos[2 * r]) +
                   readLength;
        if (!(frag < fs || frag >= il + fs)) { v = fp[frag - fs]; fl = frag; }
      }
      match[(size_t) r * K + i] = v;
      if (fraglen) fraglen[(size_t) r * K + i] = fl;
    }
  }
  free(fp); free(m1);
  return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* Simulators (synthetic inputs): simulator.c:68-196 and 221-442, stream RNG              */
/* ------------------------------------------------------------------------------------ */

/* isoform coordinate -> genomic: gff.c:917-953 (exlim/shift tables) */
static int iso_to_genomic_one(const orc_gene_t *g, int iso, int p) {
  int ex, first = g->exidx[iso], last = g->exidx[iso + 1];
  int cum = 0, cs = 0, ce = 0, n = 0;
  if (p == -1) return -1;
  for (ex = first; ex < last; ex++) {
    cum += g->exend[ex] - g->exstart[ex] + 1;
    cs += g->exstart[ex];
    if (!(cum + 1 <= p)) return p + (cs - ce - n - 1);
    ce += g->exend[ex]; n++;
  }
  return -1;
}

static void make_cigar(const orc_gene_t *g, int iso, int rs, int readLength, char *out, int cap) {
  int ex = g->exidx[iso], rl = readLength, w = 0;
  while (g->exend[ex] < rs) ex++;
  while (g->exend[ex] < rs + rl - 1) {
    w += snprintf(out + w, cap - w, "%iM%iN", g->exend[ex] - rs + 1,
                  g->exstart[ex + 1] - g->exend[ex] - 1);
    rl -= g->exend[ex] - rs + 1;
    rs = g->exstart[ex + 1];
    ex++;
  }
  snprintf(out + w, cap - w, "%iM", rl);
}

int orc_simulate_reads(const orc_gene_t *g, const double *expr, int nreads, int readLength,
                       int *isoform, int *pos, char *cigar_out, int cigar_stride) {
  int K = g->K, i, good = 0;
  int *eff = malloc(sizeof(int) * K);
  double *sp = malloc(sizeof(double) * K), sumpsi = 0.0;
  if (!eff || !sp) { free(eff); free(sp); return ORC_ENOMEM; }
  for (i = 0; i < K; i++) {
    int l = g->isolen[i] - readLength + 1;
    eff[i] = l > 0 ? l : 0;
    sp[i] = expr[i] * eff[i];
    if (sp[i] != 0) good++;
    sumpsi += sp[i];
  }
  if (!good) { free(eff); free(sp); return ORC_FAILURE; }
  for (i = 1; i < K; i++) sp[i] += sp[i - 1];
  for (i = 0; i < nreads; i++) {
    int w; double rnd;
    if (K == 1) w = 0;
    else if (K == 2) { rnd = orc_unif01() * sumpsi; w = (rnd < sp[0]) ? 0 : 1; }
    else { rnd = orc_unif01() * sumpsi; for (w = 0; rnd > sp[w]; w++) ; }
    isoform[i] = w;
  }
  for (i = 0; i < nreads; i++) pos[i] = (int) orc_integer(1, eff[isoform[i]]);
  for (i = 0; i < nreads; i++) pos[i] = iso_to_genomic_one(g, isoform[i], pos[i]);
  for (i = 0; i < nreads; i++)
    make_cigar(g, isoform[i], pos[i], readLength, cigar_out + (size_t) i * cigar_stride, cigar_stride);
  free(eff); free(sp);
  return ORC_SUCCESS;
}

int orc_simulate_paired_reads(const orc_gene_t *g, const double *expr, int npairs,
                              int readLength, double mean, double var, double numDevs,
                              int *isoform, int *pos, char *cigar_out, int cigar_stride) {
  int K = g->K, i, j, il, fs, fl, rc, good = 0;
  double *fp = 0, *px, *cpx, *sp, sumpsi = 0.0;
  rc = orc_normal_fragment(mean, var, numDevs, readLength, &fp, &fs, &il);
  if (rc) return rc;
  fl = fs + il - 1;
  px = malloc(sizeof(double) * il); cpx = malloc(sizeof(double) * il);
  sp = malloc(sizeof(double) * K);
  if (!px || !cpx || !sp) { free(fp); free(px); free(cpx); free(sp); return ORC_ENOMEM; }
  memcpy(px, fp, sizeof(double) * il);
  for (i = 1; i < il; i++) px[i] += px[i - 1];
  cpx[0] = px[0];
  for (i = 1; i < il; i++) cpx[i] = cpx[i - 1] + px[i];
  for (i = 0; i < K; i++) {
    int ilen = g->isolen[i];
    int r1 = ilen >= fl ? ilen - fl + 1 : 0;
    int r2 = ilen >= fs ? (ilen >= fl ? fl - fs : ilen - fs + 1) : 0;
    double s = 0.0;
    if (r1 > 0) s += r1;
    if (r2 > 0) s += cpx[r2 - 1];
    sp[i] = s * expr[i];
    if (sp[i] != 0) good++;
    sumpsi += sp[i];
  }
  if (!good) { free(fp); free(px); free(cpx); free(sp); return ORC_FAILURE; }
  for (i = 1; i < K; i++) sp[i] += sp[i - 1];
  for (i = 0; i < 2 * npairs; i += 2) {
    int w; double rnd;
    if (K == 1) w = 0;
    else if (K == 2) { rnd = orc_unif01() * sumpsi; w = (rnd < sp[0]) ? 0 : 1; }
    else { rnd = orc_unif01() * sumpsi; for (w = 0; rnd > sp[w]; w++) ; }
    isoform[i] = isoform[i + 1] = w;
  }
  for (i = 0, j = 0; i < npairs; i++) {
    int iso = isoform[2 * i], ilen = g->isolen[iso];
    int r1 = ilen >= fl ? ilen - fl + 1 : 0;
    int r2 = ilen >= fs ? (ilen >= fl ? fl - fs : ilen - fs + 1) : 0;
    int p, fragment; double s = 0.0, rnd;
    if (r1 > 0) s += r1;
    if (r2 > 0) s += cpx[r2 - 1];
    rnd = orc_unif(0, s);
    if (rnd < r1) p = (int) ceil(rnd);
    else { int w; rnd -= r1; for (w = 0; cpx[w] < rnd; w++) ; p = r1 + r2 - w; }
    if (p <= r1) rnd = orc_unif(0, 1.0); else rnd = orc_unif(0, px[r1 + r2 - p]);
    for (fragment = 0; px[fragment] < rnd; fragment++) ;
    fragment += fs;
    pos[j++] = p;
    pos[j++] = p + fragment - readLength;
  }
  for (i = 0; i < 2 * npairs; i++) pos[i] = iso_to_genomic_one(g, isoform[i], pos[i]);
  for (i = 0; i < 2 * npairs; i++)
    make_cigar(g, isoform[i], pos[i], readLength, cigar_out + (size_t) i * cigar_stride, cigar_stride);
  free(fp); free(px); free(cpx); free(sp);
  return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* Column ordering and read classes: matrix.pmt:546-601, miso_paired.c:576-702            */
/* ------------------------------------------------------------------------------------ */

static const double *g_sort_m; static int g_sort_k; static int g_sort_bin;

/* matrix.pmt:546-577: lexicographic compare of two columns (exact values, or zero/non-zero) */
static int col_cmp(int a, int b) {
  const double *ca = g_sort_m + (size_t) a * g_sort_k, *cb = g_sort_m + (size_t) b * g_sort_k;
  int i;
  for (i = 0; i < g_sort_k; i++) {
    if (g_sort_bin) {
      if (ca[i] == 0 && cb[i] != 0) return -1;
      if (ca[i] != 0 && cb[i] == 0) return 1;
    } else {
      if (ca[i] < cb[i]) return -1;
      if (ca[i] > cb[i]) return 1;
    }
  }
  return 0;
}

/*
 * The reference sorts the column indices with an unstable quicksort (matrix.pmt:579-601 ->
 * src/qsort.c, FreeBSD's libc qsort).  The order it leaves EQUAL columns in decides which read
 * consumes which uniform of the stream (miso.c:59-83), so reproducing the reference bit for bit
 * (per-read assignment, log-likelihood sums) needs the same permutation.  This is that sorting
 * algorithm -- Bentley & McIlroy, "Engineering a Sort Function", Software P&E 23(11), 1993,
 * program 7 (split-end partition, median of three / ninther pivot, insertion sort below 7
 * elements) plus FreeBSD's "no swap happened -> insertion sort" shortcut -- written for an int
 * index array.
 */
static void bm_swap(int *v, int i, int j) { int t = v[i]; v[i] = v[j]; v[j] = t; }

static int bm_med3(const int *v, int a, int b, int c) {
  return col_cmp(v[a], v[b]) < 0
             ? (col_cmp(v[b], v[c]) < 0 ? b : (col_cmp(v[a], v[c]) < 0 ? c : a))
             : (col_cmp(v[b], v[c]) > 0 ? b : (col_cmp(v[a], v[c]) < 0 ? a : c));
}

static void bm_insertion(int *v, int n) {
  int m, l;
  for (m = 1; m < n; m++)
    for (l = m; l > 0 && col_cmp(v[l - 1], v[l]) > 0; l--) bm_swap(v, l, l - 1);
}

static void bm_qsort(int *v, int n) {
  for (;;) {
    int pa, pb, pc, pd, pl, pm, pn, d, r, i, swapped = 0, nl, nr;
    if (n < 7) { bm_insertion(v, n); return; }
    pm = n / 2;
    if (n > 7) {
      pl = 0; pn = n - 1;
      if (n > 40) {
        d = n / 8;
        pl = bm_med3(v, pl, pl + d, pl + 2 * d);
        pm = bm_med3(v, pm - d, pm, pm + d);
        pn = bm_med3(v, pn - 2 * d, pn - d, pn);
      }
      pm = bm_med3(v, pl, pm, pn);
    }
    bm_swap(v, 0, pm);
    pa = pb = 1;
    pc = pd = n - 1;
    for (;;) {
      while (pb <= pc && (r = col_cmp(v[pb], v[0])) <= 0) {
        if (r == 0) { swapped = 1; bm_swap(v, pa, pb); pa++; }
        pb++;
      }
      while (pb <= pc && (r = col_cmp(v[pc], v[0])) >= 0) {
        if (r == 0) { swapped = 1; bm_swap(v, pc, pd); pd--; }
        pc--;
      }
      if (pb > pc) break;
      bm_swap(v, pb, pc);
      swapped = 1;
      pb++; pc--;
    }
    if (!swapped) { bm_insertion(v, n); return; }
    r = (pa < pb - pa) ? pa : pb - pa;               /* equal keys from the left end ... */
    for (i = 0; i < r; i++) bm_swap(v, i, pb - r + i);
    r = (pd - pc < n - pd - 1) ? pd - pc : n - pd - 1; /* ... and the right end to the middle */
    for (i = 0; i < r; i++) bm_swap(v, pb + i, n - r + i);
    nl = pb - pa; nr = pd - pc;
    if (nl > 1) bm_qsort(v, nl);
    if (nr > 1) { v += n - nr; n = nr; continue; }
    return;
  }
}

static void order_cols(const double *m, int K, int n, int bin, int *order) {
  int i;
  for (i = 0; i < n; i++) order[i] = i;
  g_sort_m = m; g_sort_k = K; g_sort_bin = bin;
  bm_qsort(order, n);
}

/* run-length over the sorted columns; templates K x ncls col-major */
static void classes(const double *m, int K, int n, const int *order, int bin,
                    double *templates, double *counts, int *ncls) {
  int i, j, nc = 0;
  for (i = 0; i < n; i++) {
    const double *cur = m + (size_t) order[i] * K;
    int same = nc > 0;
    if (same) {
      const double *prev = templates + (size_t) (nc - 1) * K;
      for (j = 0; j < K && same; j++)
        same = bin ? ((prev[j] != 0) == (cur[j] != 0)) : (memcmp(&prev[j], &cur[j], sizeof(double)) == 0);
    }
    if (!same) {
      for (j = 0; j < K; j++) templates[(size_t) nc * K + j] = bin ? (double) (cur[j] != 0) : cur[j];
      counts[nc] = 0;
      nc++;
    }
    counts[nc - 1] += 1;
  }
  *ncls = nc;
}

/* The gene's POSSIBLE read classes with the number of start positions of each (splicing_assignment_matrix,
   assignment.c:90-276, as a statement of WHAT it computes rather than of its walk over run-length encoded isoforms):
   a read of readLength bases starting at genomic position p lies on isoform k with a definite alignment (exon pieces
   and the gaps between them) or not at all; the isoforms that share one alignment at p form a class, and the matrix
   has one column per distinct class = its 0/1 pattern times the number of positions, the columns ordered patterns with
   a 0 in an earlier isoform first (splicing_i_assignmat_simplify).  Checked against the reference's own function on
   random gene structures (tests/test_oracle_vs_ref.py).  out: K x (returned columns), column-major; at most max_cols. */
static int assignment_classes(const orc_gene_t *g, int readLength, double *out, int max_cols) {
  int K = g->K, k, i, nc = 0, gs = 0, ge = 0, first = 1, p;
  int sig[64][130], siglen[64];
  uint64_t *masks; double *cnt;
  if (K < 1 || K > 64 || max_cols < 1) return -1;
  masks = malloc(sizeof(uint64_t) * (size_t) max_cols);
  cnt = calloc((size_t) max_cols, sizeof(double));
  for (k = 0; k < K; k++) for (i = g->exidx[k]; i < g->exidx[k + 1]; i++) {
    if (first || g->exstart[i] < gs) gs = g->exstart[i];
    if (first || g->exend[i] > ge) ge = g->exend[i];
    first = 0;
  }
  for (p = gs; p <= ge - readLength + 1; p++) {
    int any = 0;
    uint64_t done = 0;
    for (k = 0; k < K; k++) {
      siglen[k] = 0;
      for (i = g->exidx[k]; i < g->exidx[k + 1]; i++) {
        if (g->exstart[i] <= p && p <= g->exend[i]) {
          int rem = readLength, cur = p, j = i, n = 0;
          for (;;) {
            int avail = g->exend[j] - cur + 1;
            if (rem <= avail) { sig[k][n++] = rem; rem = 0; break; }
            sig[k][n++] = avail; rem -= avail;
            if (j + 1 >= g->exidx[k + 1] || n > 126) break;
            sig[k][n++] = -(g->exstart[j + 1] - g->exend[j] - 1);
            j++; cur = g->exstart[j];
          }
          if (rem == 0) { siglen[k] = n; any = 1; }
          break;
        }
      }
    }
    if (!any) continue;
    for (k = 0; k < K; k++) {
      uint64_t m; int k2, c;
      if (!siglen[k] || ((done >> k) & 1)) continue;
      m = 1ull << k;
      for (k2 = k + 1; k2 < K; k2++)
        if (siglen[k2] == siglen[k] && !memcmp(sig[k], sig[k2], sizeof(int) * (size_t) siglen[k])) m |= 1ull << k2;
      done |= m;
      for (c = 0; c < nc && masks[c] != m; c++) ;
      if (c == nc) { if (nc == max_cols) { free(masks); free(cnt); return -1; } masks[nc++] = m; }
      cnt[c] += 1.0;
    }
  }
  { /* column order: a pattern with 0 where the other has 1, at the first isoform where they differ, comes first */
    int a, b;
    for (a = 1; a < nc; a++) for (b = a; b > 0; b--) {
      uint64_t x = masks[b - 1], y = masks[b], d = x ^ y;
      if (!d || !((x >> __builtin_ctzll(d)) & 1)) break;
      masks[b - 1] = y; masks[b] = x;
      { double t = cnt[b - 1]; cnt[b - 1] = cnt[b]; cnt[b] = t; }
    }
  }
  for (i = 0; i < nc; i++) for (k = 0; k < K; k++) out[(size_t) i * K + k] = ((masks[i] >> k) & 1) ? cnt[i] : 0.0;
  free(masks); free(cnt);
  return nc;
}

int orc_assignment_matrix(const orc_gene_t *g, int readLength, int overHang, double *out, int max_cols) {
  if (overHang > 1) return -1 - ORC_UNIMPLEMENTED;   /* assignment.c:103-106 */
  return assignment_classes(g, readLength, out, max_cols);
}

/* ------------------------------------------------------------------------------------ */
/* Sampler pieces shared by single-end and paired-end                                    */
/* ------------------------------------------------------------------------------------ */

typedef struct {
  int K, N, C;
  const orc_opts_t *opts;
  const orc_math_t *M;
  int counter;           /* opts->mode == ORC_MODE_COUNTER or ORC_MODE_COLLAPSED */
  int collapsed;         /* opts->mode == ORC_MODE_COLLAPSED: reassign_collapsed() except for the run's last reassignment */
  int count_sums;        /* use count-based score sums */
  int marginal;          /* SPLICING_ALGO_MARGINAL: no assignments, the reads enter through the marginal likelihood */
  int ncls_a;            /* SPLICING_ALGO_CLASSES (marginal is set too): the gene's possible read classes ... */
  const double *amat;    /* ... K x ncls_a, rows normalised (miso.c:794-797), and ... */
  const double *amatches; /* ... the reads of every class (solve.c:110-137) */
  const double *match;   /* K x N */
  const int *order;      /* N (stream mode draw order) */
  int *corder;           /* N (counter mode draw order, single-end): see counter_order() */
  const double *hyper;   /* K */
  double lg_sum, lg_each; /* lgamma(sum a), sum lgamma(a_k): miso.c:165-182 */
  double sigma, sd, covarConst;
  /* single-end */
  const int *effisolen; const double *logeff; const double *isoscores;
  /* paired-end */
  int paired, fs, il;
  const int *fraglen;        /* K x N */
  const double *pisoscores;  /* il x K col-major */
  const double *assscores;   /* K */
  const int64_t *sfix;       /* K x N fixed-point per-read scores (counter mode) */
  /* state */
  int *ass;                  /* N x C */
  double *psi, *psiNew, *alpha, *alphaNew; /* K x C, (K-1) x C */
} orc_state_t;

#define SFIX_BAD ((int64_t) MISO_SFIX_BAD)

/* miso.c:219-241 + 462-468 (logit_inv and the last component) */
static void logit_inv_col(const orc_state_t *S, const double *alpha, double *psi) {
  int len = S->K - 1, i;
  double sumexp = 0.0, sumpsi = 0.0;
  for (i = 0; i < len; i++) sumexp += S->M->exp(alpha[i]);
  sumexp += 1.0;
  for (i = 0; i < len; i++) psi[i] = S->M->exp(alpha[i]) / sumexp;
  for (i = 0; i < len; i++) sumpsi += psi[i];
  psi[len] = 1 - sumpsi;
}

/* miso.c:449-471 via mvrnorm miso.c:184-200: normals chain-major, component-minor */
static void propose(orc_state_t *S, uint32_t iter, const double *alpha, double *psiOut, double *alphaOut) {
  int len = S->K - 1, i, j;
  for (j = 0; j < S->C; j++) {
    for (i = 0; i < len; i++) {
      double z;
      if (S->counter) {
        int w = 2 + 2 * i;
        miso_u32x4 b = miso_draw_block(S->opts->seed, S->opts->event_id, (uint32_t) j, iter,
                                       MISO_SITE_MH, (uint32_t) (w / 4));
        z = norm_from_unif(miso_u01(b.v[w % 4]), miso_u01(b.v[w % 4 + 1]), S->M);
      } else {
        z = orc_normal01();
      }
      alphaOut[j * len + i] = alpha[j * len + i] + S->sd * z;
    }
  }
  for (j = 0; j < S->C; j++) logit_inv_col(S, alphaOut + j * len, psiOut + j * S->K);
}

/* miso.c:11-91 (single-end, weights psi_k) and miso_paired.c:11-86 (weights psi_k*match) */
static int draw_read(const orc_state_t *S, const double *col, const double *psi, double u) {
  double cum[ORC_MAXK]; int valid[ORC_MAXK]; int nv = 0, j, w;
  double sumpsi = 0.0, rnd;
  for (j = 0; j < S->K; j++) {
    if (col[j] != 0) {
      sumpsi += S->paired ? psi[j] * col[j] : psi[j];
      valid[nv] = j; cum[nv] = sumpsi; nv++;
    }
  }
  rnd = u * sumpsi;
  if (nv == 2) return (rnd < cum[0]) ? valid[0] : valid[1];
  for (w = 0; rnd > cum[w]; w++) ;
  return valid[w];
}

static int count_valid(const double *col, int K, int *first) {
  int j, nv = 0;
  for (j = 0; j < K; j++) if (col[j] != 0) { if (!nv) *first = j; nv++; }
  return nv;
}

/* Counter-mode draw order of the single-end sampler: reads ordered by their compatibility column,
   columns compared lexicographically as splicing_order_matches does (matrix.pmt:546-562), ties by
   read index (the reference's qsort is unstable there; a contract needs a definite order).  The
   r-th read of this order with >= 2 compatible isoforms uses Gibbs word r.  Paired-end (round 5; input order before):
   by fragment-length rows, isoform 0 most significant, an incompatible isoform (-1) below every length, ties by read
   index -- the reference walks its pairs in the order of their match columns as well (miso_paired.c:24-86). */
static const double *co_match; static int co_K;
static int co_cmp(const void *a, const void *b) {
  int x = *(const int *) a, y = *(const int *) b, k;
  const double *cx = co_match + (size_t) x * co_K, *cy = co_match + (size_t) y * co_K;
  for (k = 0; k < co_K; k++) {
    int bx = cx[k] != 0, by = cy[k] != 0;
    if (bx != by) return bx - by;
  }
  return (x > y) - (x < y);
}
static const int *co_frag;
static int co_cmp_paired(const void *a, const void *b) {
  int x = *(const int *) a, y = *(const int *) b, k;
  const int *fx = co_frag + (size_t) x * co_K, *fy = co_frag + (size_t) y * co_K;
  for (k = 0; k < co_K; k++) if (fx[k] != fy[k]) return (fx[k] > fy[k]) - (fx[k] < fy[k]);
  return (x > y) - (x < y);
}
static void counter_order(orc_state_t *S) {
  int i;
  S->corder = malloc(sizeof(int) * (size_t) (S->N > 0 ? S->N : 1));
  for (i = 0; i < S->N; i++) S->corder[i] = i;
  co_K = S->K;
  if (!S->paired) { co_match = S->match; qsort(S->corder, (size_t) S->N, sizeof(int), co_cmp); }
  else { co_frag = S->fraglen; qsort(S->corder, (size_t) S->N, sizeof(int), co_cmp_paired); }
}

/* The collapsed Gibbs step (single-end).  The reads in counter order (sorted by compatibility column) fall into
   runs of equal columns = classes; a class of n reads with compatible isoforms v_0 < v_1 < ... < v_{nv-1} gets
   counts Multinomial(n; psi_v / sum psi_v) (the per-read law of miso.c:30-91 for each of its exchangeable reads),
   drawn as x_w ~ Binomial(n - x_0 - ... - x_{w-1}, psi_{v_w} / S_w), S_w = psi_{v_{nv-1}} + ... + psi_{v_w} summed in
   that order.  The assignment vector is filled in class order (first x_0 reads -> v_0, ...): only its counts are
   used until the run's last reassignment, which is per read (reassign). */
static void reassign_collapsed(orc_state_t *S, uint32_t iter) {
  int k, K = S->K, N = S->N;
  if (!S->corder) counter_order(S);
  for (k = 0; k < S->C; k++) {
    const double *psi = S->psi + k * K;
    int *ass = S->ass + (size_t) k * N;
    miso_ustream us;
    int ii = 0;
    miso_ustream_init(&us, S->opts->seed, S->opts->event_id, (uint32_t) k, iter, MISO_SITE_COUNTS);
    while (ii < N) {
      const double *col = S->match + (size_t) S->corder[ii] * K;
      int jj = ii + 1, n, nv = 0, valid[ORC_MAXK], j, w;
      double suffix[ORC_MAXK];
      while (jj < N) {
        const double *c2 = S->match + (size_t) S->corder[jj] * K;
        int same = 1;
        for (j = 0; j < K && same; j++) same = (col[j] != 0) == (c2[j] != 0);
        if (!same) break;
        jj++;
      }
      n = jj - ii;
      for (j = 0; j < K; j++) if (col[j] != 0) valid[nv++] = j;
      if (nv == 0) { for (j = ii; j < jj; j++) ass[S->corder[j]] = -1; }
      else if (nv == 1) { for (j = ii; j < jj; j++) ass[S->corder[j]] = valid[0]; }
      else {
        int rem = n, at = ii;
        double acc = 0.0;
        for (w = nv - 1; w >= 0; w--) { acc = acc + psi[valid[w]]; suffix[w] = acc; }
        for (w = 0; w < nv; w++) {
          int x = (w == nv - 1) ? rem : miso_binomial(&us, rem, psi[valid[w]] / suffix[w], logfact(rem));
          for (j = 0; j < x; j++) ass[S->corder[at + j]] = valid[w];
          at += x; rem -= x;
        }
      }
      ii = jj;
    }
  }
}

static void reassign(orc_state_t *S, uint32_t iter) {
  int k, i, K = S->K, N = S->N;
  for (k = 0; k < S->C; k++) {
    const double *psi = S->psi + k * K;
    int *ass = S->ass + (size_t) k * N;
    if (S->counter) {
      uint32_t rank = 0; miso_u32x4 blk, blk_lo; uint32_t have = 0xFFFFFFFFu; int ii;
      const int split = !S->paired && K == 2;   /* lazy low bits (miso_philox.h): two half-words per read */
      memset(&blk, 0, sizeof(blk)); memset(&blk_lo, 0, sizeof(blk_lo));
      if (!S->corder) counter_order(S);
      for (ii = 0; ii < N; ii++) {
        const double *col;
        i = S->corder[ii];
        col = S->match + (size_t) i * K;
        int first = -1, nv = count_valid(col, K, &first);
        if (nv == 0) ass[i] = -1;
        else if (nv == 1) ass[i] = first;
        else if (split) { /* = miso_split_word(seed, event, chain, iter, rank), the two blocks kept for their eight reads */
          if (rank / 8 != have) {
            have = rank / 8;
            blk = miso_draw_block(S->opts->seed, S->opts->event_id, (uint32_t) k, iter, MISO_SITE_GIBBS, have);
            blk_lo = miso_draw_block(S->opts->seed, S->opts->event_id, (uint32_t) k, iter, MISO_SITE_GIBBS_LOW, have);
          }
          ass[i] = draw_read(S, col, psi, miso_u01((miso_block_half(blk, rank & 7u) << 16) | miso_block_half(blk_lo, rank & 7u)));
          rank++;
        } else {
          if (rank / 4 != have) {
            have = rank / 4;
            blk = miso_draw_block(S->opts->seed, S->opts->event_id, (uint32_t) k, iter,
                                  MISO_SITE_GIBBS, have);
          }
          ass[i] = draw_read(S, col, psi, miso_u01(blk.v[rank % 4]));
          rank++;
        }
      }
    } else {
      for (i = 0; i < N; i++) {
        int r = S->order[i];
        const double *col = S->match + (size_t) r * K;
        int first = -1, nv = count_valid(col, K, &first);
        if (nv == 0) ass[r] = -1;
        else if (nv == 1) ass[r] = first;
        else ass[r] = draw_read(S, col, psi, orc_unif01());
      }
    }
  }
}

static void counts_of(const orc_state_t *S, int chain, int32_t *cnt) {
  const int *ass = S->ass + (size_t) chain * S->N; int i;
  for (i = 0; i < S->K; i++) cnt[i] = 0;
  for (i = 0; i < S->N; i++) if (ass[i] != -1) cnt[ass[i]]++;
}

/* miso.c:165-182 */
static double ldirichlet(const orc_state_t *S, const double *x) {
  double score = 0.0; int i;
  for (i = 0; i < S->K; i++) score += (S->hyper[i] - 1.0) * S->M->log(x[i]);
  score += S->lg_sum;
  score -= S->lg_each;
  return score;
}

/* miso.c:124-163 / miso_paired.c:88-131 and miso.c:243-307 / miso_paired.c:133-174 */
static double score_joint(const orc_state_t *S, int chain, const double *psi) {
  int K = S->K, N = S->N, i;
  const int *ass = S->ass + (size_t) chain * N;
  double logpsi[ORC_MAXK] = { 0 }, maxv, sum, readProb = 0.0, assProb = 0.0;
  int32_t cnt[ORC_MAXK];
  if (S->marginal && S->amat) { /* miso.c:284-295: the gene's classes, reads per class */
    int c, k;
    for (c = 0; c < S->ncls_a; c++) {
      double score = 0.0;
      for (k = 0; k < K; k++) score += S->amat[(size_t) c * K + k] * psi[k];
      if (score != 0) readProb += S->M->log(score) * S->amatches[c];
    }
    return readProb + assProb + ldirichlet(S, psi);
  }
  if (S->marginal) { /* miso.c:272-283; S->match holds match / effective length (miso.c:800-808) */
    int k;
    if (!S->count_sums) {
      for (i = 0; i < N; i++) {
        double isoscore = 0.0;
        for (k = 0; k < K; k++) isoscore += S->match[(size_t) i * K + k] * psi[k];
        if (isoscore != 0) readProb += S->M->log(isoscore);
      }
    } else { /* counter contract: class by class in counter order (= the order of the header's classes), count x log */
      int ii = 0;
      while (ii < N) {
        const double *col = S->match + (size_t) S->corder[ii] * K;
        int jj = ii + 1, nv = 0;
        double isoscore = 0.0;
        while (jj < N) {
          const double *c2 = S->match + (size_t) S->corder[jj] * K;
          int same = 1;
          for (k = 0; k < K && same; k++) same = (col[k] != 0) == (c2[k] != 0);
          if (!same) break;
          jj++;
        }
        for (k = 0; k < K; k++) if (col[k] != 0) { isoscore += col[k] * psi[k]; nv++; }
        if (nv > 0 && isoscore != 0) readProb = readProb + (double) (jj - ii) * S->M->log(isoscore);
        ii = jj;
      }
    }
    return readProb + assProb + ldirichlet(S, psi);
  }
  for (i = 0; i < K; i++)
    logpsi[i] = S->M->log(psi[i]) + (S->paired ? S->assscores[i] : S->logeff[i]);
  maxv = logpsi[0];
  for (i = 1; i < K; i++) if (logpsi[i] > maxv) maxv = logpsi[i];
  sum = 0.0;
  for (i = 0; i < K; i++) sum += S->M->exp(logpsi[i] - maxv);
  sum = S->M->log(sum) + maxv;
  for (i = 0; i < K; i++) logpsi[i] -= sum;

  if (!S->count_sums) {
    for (i = 0; i < N; i++) {
      if (ass[i] != -1) {
        if (S->paired) {
          int fl = S->fraglen[(size_t) i * K + ass[i]];
          readProb += S->pisoscores[(size_t) ass[i] * S->il + (fl - S->fs)];
        } else {
          readProb += S->isoscores[ass[i]];
        }
      }
    }
    for (i = 0; i < N; i++) if (ass[i] != -1) assProb += logpsi[ass[i]];
  } else {
    counts_of(S, chain, cnt);
    if (S->paired) {
      int64_t acc = 0; int bad = 0;
      for (i = 0; i < N; i++) if (ass[i] != -1) {
        int64_t v = S->sfix[(size_t) i * K + ass[i]];
        if (v == SFIX_BAD) bad = 1; else acc += v;
      }
      readProb = bad ? NAN : (double) acc * (1.0 / MISO_SFIX_SCALE);
    } else {
      for (i = 0; i < K; i++) if (cnt[i] != 0) readProb = readProb + (double) cnt[i] * S->isoscores[i];
    }
    for (i = 0; i < K; i++) if (cnt[i] != 0) assProb = assProb + (double) cnt[i] * logpsi[i];
  }
  return readProb + assProb + ldirichlet(S, psi);
}

/* miso.c:97-122 with len = K-1 (miso.c:473-491) */
static double proposal_score(const orc_state_t *S, const double *theta, const double *mu) {
  int len = S->K - 1, i;
  double ltheta = 1.0, prodTheta = 1.0, expPart = 0.0, pdfVal;
  for (i = 0; i < len; i++) { ltheta -= theta[i]; prodTheta *= theta[i]; }
  prodTheta = 1.0 / prodTheta / ltheta;
  for (i = 0; i < len; i++) {
    double tmp = S->M->log(theta[i] / ltheta) - mu[i];
    expPart += (-0.5) * tmp * tmp / S->sigma;
  }
  pdfVal = S->covarConst * prodTheta * S->M->exp(expPart);
  return S->M->log(pdfVal);
}

static uint64_t fnv_step(uint64_t h, uint32_t w) { return (h ^ (uint64_t) w) * 0x100000001B3ull; }

/* The iteration loop common to miso.c:845-900 and miso_paired.c:451-498 */
/* m_base: counter mode, stop = CONVERGENT_MEAN -- the iterations of all earlier rounds.  The loop counter m starts at 0
   in every round as the reference's does (so the round's first Metropolis-Hastings ratio again leaves the proposal
   terms out, miso.c:866 `m > 0 ? 1 : 0`), the counter-based draws are addressed by the chain's own iteration number
   m_base + m. */
static void run_chains(orc_state_t *S, uint32_t m_base, int noIterations, int noBurnIn, int noLag,
                       double *samples, double *logLik, int *rundata, orc_trace_t *trace) {
  int K = S->K, C = S->C, len = K - 1, m, j, i, lagCounter = 0, noS = 0;
  int noAccepted = 0, noRejected = 0;
  double *acceptP = malloc(sizeof(double) * C), *cJS = malloc(sizeof(double) * C),
         *pJS = malloc(sizeof(double) * C);
  uint64_t *hash = malloc(sizeof(uint64_t) * C);
  int32_t cnt[ORC_MAXK];
  for (j = 0; j < C; j++) { hash[j] = 0xCBF29CE484222325ull; cJS[j] = 0; if (trace && trace->accepted) trace->accepted[j] = 0; }

  for (m = 0; m < noIterations; m++) {
    for (j = 0; j < C; j++) { /* counts the MH step of iteration m sees */
      counts_of(S, j, cnt);
      for (i = 0; i < K; i++) {
        hash[j] = fnv_step(hash[j], (uint32_t) cnt[i]);
        if (trace && trace->counts_trace) trace->counts_trace[((size_t) m * C + j) * K + i] = cnt[i];
      }
    }
    propose(S, m_base + (uint32_t) m, S->alpha, S->psiNew, S->alphaNew);
    for (j = 0; j < C; j++) { /* miso.c:493-552 */
      double pp = score_joint(S, j, S->psiNew + j * K);
      double pc = score_joint(S, j, S->psi + j * K);
      double ptoCS = proposal_score(S, S->psi + j * K, S->alphaNew + j * len);
      double ctoPS = proposal_score(S, S->psiNew + j * K, S->alpha + j * len);
      pJS[j] = pp; cJS[j] = pc;
      acceptP[j] = (m > 0) ? S->M->exp(pp + ptoCS - (pc + ctoPS)) : S->M->exp(pp - pc);
    }
    for (j = 0; j < C; j++) { /* miso.c:869-880 */
      int acc;
      if (S->counter) {
        miso_u32x4 b = miso_draw_block(S->opts->seed, S->opts->event_id, (uint32_t) j, m_base + (uint32_t) m,
                                       MISO_SITE_MH, 0);
        acc = (acceptP[j] >= 1) || (miso_u01(b.v[0]) < acceptP[j]);
      } else {
        acc = (acceptP[j] >= 1) || (orc_unif01() < acceptP[j]);
      }
      if (acc) {
        memcpy(S->psi + j * K, S->psiNew + j * K, sizeof(double) * K);
        memcpy(S->alpha + j * len, S->alphaNew + j * len, sizeof(double) * len);
        cJS[j] = pJS[j];
        noAccepted++;
        if (trace && trace->accepted) trace->accepted[j]++;
      } else {
        noRejected++;
      }
    }
    if (m >= noBurnIn) { /* miso.c:882-893 */
      if (lagCounter == noLag - 1) {
        memcpy(samples + (size_t) noS * K, S->psi, sizeof(double) * K * C);
        memcpy(logLik + noS, cJS, sizeof(double) * C);
        noS += C;
        lagCounter = 0;
      } else {
        lagCounter++;
      }
    }
    if (S->marginal) continue;   /* miso.c:895: only the REASSIGN algorithm reassigns */
    if (S->collapsed && m != noIterations - 1) reassign_collapsed(S, m_base + (uint32_t) m); else reassign(S, m_base + (uint32_t) m);
  }
  for (j = 0; j < C; j++) {
    counts_of(S, j, cnt);
    for (i = 0; i < K; i++) {
      hash[j] = fnv_step(hash[j], (uint32_t) cnt[i]);
      if (trace && trace->counts_trace)
        trace->counts_trace[((size_t) noIterations * C + j) * K + i] = cnt[i];
    }
    if (trace && trace->counts_hash) trace->counts_hash[j] = hash[j];
  }
  if (trace && trace->final_psi) memcpy(trace->final_psi, S->psi, sizeof(double) * K * C);
  rundata[5] = noAccepted; rundata[6] = noRejected;
  free(acceptP); free(cJS); free(pJS); free(hash);
}

/* miso.c:330-447 for START_AUTO / START_UNIFORM, then miso.c:834 */
static void init_chains(orc_state_t *S, int start) {
  int K = S->K, C = S->C, i, j;
  for (j = 0; j < C; j++) {
    if (start == ORC_START_AUTO && K == 2) {
      if (!S->counter) (void) orc_unif01(); /* miso.c:365: consumed, value overwritten */
      S->alpha[j] = 0.0;
    } else if (start == ORC_START_AUTO) {
      for (i = 0; i < K - 1; i++) S->alpha[j * (K - 1) + i] = 1.0 / (K - 1);
    } else {
      for (i = 0; i < K - 1; i++) S->alpha[j * (K - 1) + i] = 0.0;
    }
  }
  propose(S, MISO_ITER_INIT, S->alpha, S->psi, S->alpha);
}

/* miso.c:556-636 (splicing_i_check_convergent_mean), line by line -- including what a textbook R-hat would not do:
   the running mean's divisor l starts at 1 for a chain's SECOND sample, chainVars holds sums of squares (not
   variances) and W averages their squares.  samples: K x noSamples, column i belongs to chain i % C. */
static int convergent_mean(const double *samples, int K, int C, int noSamples) {
  double *means = malloc(sizeof(double) * K * C), *vars = calloc((size_t) K * C, sizeof(double));
  int i, j, k, l, stop = 1;
  memcpy(means, samples, sizeof(double) * K * C);
  for (i = C, j = 0, l = 1; i < noSamples; i++, j = (j + 1) % C) {
    for (k = 0; k < K; k++) {
      double x = samples[(size_t) i * K + k], m0 = means[j * K + k];
      double mk = m0 + (x - m0) / l;
      double sk = vars[j * K + k] + (x - m0) * (x - mk);
      means[j * K + k] = mk; vars[j * K + k] = sk;
    }
    if (j == C - 1) l++;
  }
  for (k = 0; k < K; k++) {
    double mean = 0.0, B = 0.0, W = 0.0, rhat;
    for (j = 0; j < C; j++) mean += means[j * K + k];
    mean /= C;
    for (j = 0; j < C; j++) { double t = means[j * K + k] - mean; B += t * t; }
    B *= noSamples / (C - 1.0);
    for (j = 0; j < C; j++) { double t = vars[j * K + k]; W += t * t; }
    W /= C;
    rhat = sqrt(((noSamples - 1.0) / noSamples * W + B / noSamples) / W);
    stop = stop && rhat <= 1.1;
  }
  free(means); free(vars);
  return stop;
}

/* score_joint's CLASSES branch on caller-made inputs (tests: against ref_shim.c ref_score_classes), libm */
double orc_score_classes(int K, const double *psi, const double *hyper, const double *amat, int ncls,
                         const double *matches) {
  orc_state_t S; int i; double asum = 0.0, lge = 0.0;
  memset(&S, 0, sizeof(S));
  S.K = K; S.N = 0; S.C = 1; S.M = &MATH_LIBM; S.marginal = 1; S.amat = amat; S.amatches = matches; S.ncls_a = ncls;
  S.hyper = hyper;
  for (i = 0; i < K; i++) { asum += hyper[i]; lge += lgamma(hyper[i]); }
  S.lg_sum = lgamma(asum); S.lg_each = lge;
  return score_joint(&S, 0, psi);
}

uint32_t orc_split_word(uint64_t seed, uint32_t event_id, uint32_t chain, uint32_t iteration, uint32_t r) {
  return miso_split_word(seed, event_id, chain, iteration, r);
}

int orc_convergent_mean(const double *samples, int K, int C, int noSamples) {
  return convergent_mean(samples, K, C, noSamples);
}

/* The chains from their start to the stopping rule: miso.c:827-934 / miso_paired.c:431-532.
   STOP_FIXEDNO: one round.  STOP_CONVERGENT_MEAN: after a round that has not converged (and while noIterations <
   maxIterations) the schedule becomes noIterations' = 3 noIterations - 2 noBurnIn, noBurnIn' = noIterations, the
   samples are collected afresh, and of the last round the LAST noSamples (of the first round) are returned
   (miso.c:976-983).  Both modes continue the chains where they are, with the loop counter back at 0, as the reference
   does (miso.c:845-847: `for (m=0, ...`; the chains' psi, alpha and assignments are untouched by :914-925).  Counter mode
   addresses its draws by the chain's own iteration number (all rounds counted), so a device that keeps no chain state
   between launches reproduces round r by running iterations [0, G_r + N_r) from the start, G_r = the iterations of
   the earlier rounds, and keeping [G_r + B_r, G_r + N_r) -- the same chain, the same window
   (miso_amd/csrc/runtime.hip converge_rounds).  Collapsed mode: a round's last reassignment is per read (the returned
   vector); when another round follows, the collapsed step of that iteration takes its place (it depends on psi only). */
static void run_rounds(orc_state_t *S, int start, int stop, int noIterations, int maxIterations, int noBurnIn,
                       int noLag, double *samples, double *logLik, int *rundata, orc_trace_t *trace) {
  int K = S->K, C = S->C, S0 = C * (noIterations - noBurnIn) / noLag, nS = S0, round = 0;
  double *buf = samples, *lbuf = logLik;
  int acc = 0, rej = 0;
  uint32_t m_base = 0;
  for (;; round++) {
    if (round == 0) {
      init_chains(S, start);            /* miso.c:827-835 */
      if (S->marginal) {                /* miso.c:839-842: no assignment to start from */
        size_t i;
        for (i = 0; i < (size_t) (S->N > 0 ? S->N : 1) * C; i++) S->ass[i] = -1;
        if (S->counter && !S->corder) counter_order(S);
      } else if (S->N > 0) {            /* miso.c:841 */
        if (S->collapsed && noIterations > 0) reassign_collapsed(S, MISO_ITER_INIT); else reassign(S, MISO_ITER_INIT);
      }
    }
    run_chains(S, m_base, noIterations, noBurnIn, noLag, buf, lbuf, rundata, round == 0 ? trace : NULL);
    /* the single-end loop resets the two counters every round (miso.c:847), the paired-end one never does
       (miso_paired.c:345, 453) */
    if (S->paired) { acc += rundata[5]; rej += rundata[6]; rundata[5] = acc; rundata[6] = rej; }
    if (stop != 1 || maxIterations <= noIterations) break;      /* miso.c:903-912 */
    if (nS < C || convergent_mean(buf, K, C, nS)) break;         /* fewer samples than chains: nothing to assess */
    if (S->collapsed && noIterations > 0 && S->N > 0) reassign_collapsed(S, m_base + (uint32_t) noIterations - 1);
    m_base += (uint32_t) noIterations;
    { int next = 3 * noIterations - 2 * noBurnIn; noBurnIn = noIterations; noIterations = next; } /* miso.c:921-924 */
    nS = C * (noIterations - noBurnIn) / noLag;
    if (buf != samples) { free(buf); free(lbuf); }
    buf = calloc((size_t) K * (nS > 0 ? nS : 1), sizeof(double));
    lbuf = calloc((size_t) (nS > 0 ? nS : 1), sizeof(double));
  }
  if (buf != samples) {                                         /* miso.c:976-983 */
    memcpy(samples, buf + (size_t) (nS - S0) * K, sizeof(double) * (size_t) K * S0);
    memcpy(logLik, lbuf + (nS - S0), sizeof(double) * S0);
    free(buf); free(lbuf);
  }
}

static int check_common(const orc_gene_t *g, int *overHang, int readLength, int noChains,
                        int noIterations, int noBurnIn, int noLag, int nhyper, int start,
                        int stop) {
  if (start == 3) return ORC_EINVAL;            /* GIVEN without start_psi: miso.c:680-683 */
  if (start == 2 || start == 4) return ORC_UNIMPLEMENTED; /* RANDOM, LINEAR: not restated */
  if (start < 0 || start > 4) return ORC_EINVAL;
  if (*overHang == 0) *overHang = 1;
  if (*overHang < 1 || *overHang >= readLength / 2) return ORC_EINVAL; /* miso.c:690-694 */
  if (nhyper != g->K) return ORC_EINVAL;        /* miso.c:698-701 */
  if (noChains < 1) return ORC_EINVAL;          /* miso.c:703-706 */
  if (stop == 1 && noChains == 1) return ORC_EINVAL; /* miso.c:708-711 */
  if (stop != 0 && stop != 1) return ORC_EINVAL;
  if (noLag < 1 || noIterations < noBurnIn || noBurnIn < 0) return ORC_EINVAL;
  if (g->K < 2 || g->K > ORC_MAXK) return ORC_EINVAL;
  return ORC_SUCCESS;
}

static void fill_common(orc_state_t *S, const orc_gene_t *g, const orc_opts_t *opts,
                        const double *hyper, int C, int N) {
  int K = g->K, i; double asum = 0.0, lge = 0.0;
  memset(S, 0, sizeof(*S));
  S->K = K; S->N = N; S->C = C; S->opts = opts;
  S->collapsed = opts && opts->mode == ORC_MODE_COLLAPSED;
  S->counter = opts && (opts->mode == ORC_MODE_COUNTER || S->collapsed);
  S->M = S->counter ? &MATH_DET : &MATH_LIBM;
  S->count_sums = S->counter && !(opts->per_read_sums);
  S->hyper = hyper;
  for (i = 0; i < K; i++) { asum += hyper[i]; lge += lgamma(hyper[i]); }
  S->lg_sum = lgamma(asum); S->lg_each = lge;
  S->sigma = 0.2 / K / K;                          /* miso.c:328 */
  S->sd = (K - 1 == 1) ? S->sigma : sqrt(S->sigma); /* miso.c:188 */
  S->covarConst = pow(2 * M_PI * S->sigma, -0.5 * (K - 1)); /* miso.c:101 */
  S->ass = malloc(sizeof(int) * (size_t) (N > 0 ? N : 1) * C);
  S->psi = calloc((size_t) K * C, sizeof(double));
  S->psiNew = calloc((size_t) K * C, sizeof(double));
  S->alpha = calloc((size_t) K * C, sizeof(double));
  S->alphaNew = calloc((size_t) K * C, sizeof(double));
}

static void free_common(orc_state_t *S) {
  free(S->ass); free(S->psi); free(S->psiNew); free(S->alpha); free(S->alphaNew); free(S->corder);
}

/* ------------------------------------------------------------------------------------ */
/* splicing_miso: miso.c:638-986                                                         */
/* ------------------------------------------------------------------------------------ */

int orc_miso(const orc_gene_t *g, const int *pos, const char **cigar, int nreads,
             int readLength, int overHang, int noChains, int noIterations, int maxIterations,
             int noBurnIn, int noLag, const double *hyper, int nhyper, int algorithm,
             int start, int stop, const orc_opts_t *opts, double *samples, double *logLik,
             double *match_out, double *class_templates, double *class_counts, int *ncls,
             int *assignment, int *rundata, orc_trace_t *trace) {
  orc_state_t S; int K, i, rc, noSamples, ncls_a = 0;
  double *match, *amat = 0, *amatches = 0; int *order, *eff; double *logeff, *isoscores;
  static const orc_opts_t stream_opts = { ORC_MODE_STREAM, 0, 0, 0 };
  if (!opts) opts = &stream_opts;
  if (algorithm != ORC_ALGO_REASSIGN && algorithm != 1 && algorithm != 2) return ORC_EINVAL; /* miso.c:674-678 */
  if (algorithm != ORC_ALGO_REASSIGN && opts->mode == ORC_MODE_COLLAPSED) return ORC_EINVAL; /* nothing to collapse: no assignments */
  rc = check_common(g, &overHang, readLength, noChains, noIterations, noBurnIn, noLag, nhyper,
                    start, stop);
  if (rc) return rc;
  K = g->K;
  noSamples = noChains * (noIterations - noBurnIn) / noLag; /* miso.c:661 */
  rundata[0] = K; rundata[1] = noIterations; rundata[2] = 0; rundata[3] = noBurnIn;
  rundata[4] = noLag; rundata[5] = rundata[6] = 0; rundata[7] = noChains; rundata[8] = noSamples;

  match = malloc(sizeof(double) * (size_t) K * (nreads > 0 ? nreads : 1));
  order = malloc(sizeof(int) * (nreads > 0 ? nreads : 1));
  eff = malloc(sizeof(int) * K); logeff = malloc(sizeof(double) * K);
  isoscores = malloc(sizeof(double) * K);
  rc = orc_match_iso(g, pos, cigar, nreads, overHang, readLength, match); /* miso.c:758 */
  if (rc) { free(match); free(order); free(eff); free(logeff); free(isoscores); return rc; }
  order_cols(match, K, nreads, 0, order);                                  /* miso.c:760 */
  if (class_templates && class_counts)
    classes(match, K, nreads, order, 0, class_templates, class_counts, ncls); /* miso.c:762 */
  for (i = 0; i < K; i++) { /* miso.c:777-784 */
    int l = g->isolen[i] - readLength + 1 - 2 * (g->noexons[i] - 1) * (overHang - 1);
    eff[i] = l > 0 ? l : 0;
    isoscores[i] = -log((double) l);
    logeff[i] = log((double) eff[i]); /* miso.c:136-138 */
  }
  if (algorithm == 1) { /* miso.c:800-808: "probabilities divided by effective isoform length" (after the classes and
                           the read order were taken from the 0/1 matrix) */
    int j;
    for (i = 0; i < K; i++) for (j = 0; j < nreads; j++)
      if (eff[i] != 0) match[(size_t) j * K + i] /= eff[i];
  }
  if (algorithm == 2) { /* miso.c:788-803: the assignment matrix, rows normalised, and the reads of every class */
    int c, r, kk;
    if (overHang > 1) { free(match); free(order); free(eff); free(logeff); free(isoscores); return ORC_UNIMPLEMENTED; } /* assignment.c:103 */
    amat = malloc(sizeof(double) * (size_t) K * 4096);
    ncls_a = assignment_classes(g, readLength, amat, 4096);
    if (ncls_a < 0) { free(amat); free(match); free(order); free(eff); free(logeff); free(isoscores); return ORC_EINVAL; }
    for (kk = 0; kk < K; kk++) { /* matrix.pmt:1525-1541 */
      double rowsum = 0.0;
      for (c = 0; c < ncls_a; c++) rowsum += amat[(size_t) c * K + kk];
      for (c = 0; c < ncls_a; c++) amat[(size_t) c * K + kk] /= rowsum;
    }
    amatches = calloc((size_t) (ncls_a > 0 ? ncls_a : 1), sizeof(double));
    for (r = 0; r < nreads; r++) { /* solve.c:122-134: the first class with the read's pattern */
      for (c = 0; c < ncls_a; c++) {
        int same = 1;
        for (kk = 0; kk < K && same; kk++) {
          double m1 = match[(size_t) r * K + kk], m2 = amat[(size_t) c * K + kk];
          same = (m1 > 0 && m2 > 0) || (m1 == 0 && m2 == 0);
        }
        if (same) { amatches[c] += 1; break; }
      }
    }
  }
  fill_common(&S, g, opts, hyper, noChains, nreads);
  S.marginal = algorithm == 1 || algorithm == 2;
  S.amat = amat; S.amatches = amatches; S.ncls_a = ncls_a;
  S.match = match; S.order = order; S.effisolen = eff; S.logeff = logeff; S.isoscores = isoscores;
  memset(samples, 0, sizeof(double) * (size_t) K * noSamples);
  memset(logLik, 0, sizeof(double) * noSamples);

  run_rounds(&S, start, stop, noIterations, maxIterations, noBurnIn, noLag, samples, logLik, rundata, trace);

  /* miso.c:936-942: "This might not have been calculated, so we calculate it now" -- every chain, from the final psi;
     counter mode: the Gibbs words of MISO_ITER_INIT, which this algorithm has not used */
  if (S.marginal && nreads > 0) reassign(&S, MISO_ITER_INIT);
  for (i = 0; i < nreads; i++) assignment[i] = S.ass[i]; /* chain 0: miso.c:943-946 */
  if (match_out) memcpy(match_out, match, sizeof(double) * (size_t) K * nreads);
  free_common(&S);
  free(match); free(order); free(eff); free(logeff); free(isoscores); free(amat); free(amatches);
  return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* splicing_miso_paired: miso_paired.c:241-574                                           */
/* ------------------------------------------------------------------------------------ */

int orc_miso_paired(const orc_gene_t *g, const int *pos, const char **cigar, int npos,
                    int readLength, int overHang, int noChains, int noIterations,
                    int maxIterations, int noBurnIn, int noLag, const double *hyper,
                    int nhyper, int start, int stop, double mean, double var, double numDevs,
                    const orc_opts_t *opts, double *samples, double *logLik, double *match_out,
                    double *bin_templates, double *bin_counts, int *ncls, int *assignment,
                    int *rundata, orc_trace_t *trace) {
  orc_state_t S; int K, N = npos / 2, i, j, rc, noSamples, fs, il;
  double *match, *fp = 0, *pisoscores, *assscores; int *order, *fraglen; int64_t *sfix = 0;
  static const orc_opts_t stream_opts = { ORC_MODE_STREAM, 0, 0, 0 };
  if (!opts) opts = &stream_opts;
  rc = orc_normal_fragment(mean, var, numDevs, readLength, &fp, &fs, &il); /* :299-308 */
  if (rc) return rc;
  rc = check_common(g, &overHang, readLength, noChains, noIterations, noBurnIn, noLag, nhyper,
                    start, stop);
  if (!rc && opts->mode == ORC_MODE_COLLAPSED) rc = ORC_EINVAL;   /* paired-end reads are not exchangeable */
  if (rc) { free(fp); return rc; }
  K = g->K;
  noSamples = noChains * (noIterations - noBurnIn) / noLag;
  rundata[0] = K; rundata[1] = noIterations; rundata[2] = 0; rundata[3] = noBurnIn;
  rundata[4] = noLag; rundata[5] = rundata[6] = 0; rundata[7] = noChains; rundata[8] = noSamples;

  match = malloc(sizeof(double) * (size_t) K * (N > 0 ? N : 1));
  fraglen = malloc(sizeof(int) * (size_t) K * (N > 0 ? N : 1));
  order = malloc(sizeof(int) * (N > 0 ? N : 1));
  pisoscores = malloc(sizeof(double) * (size_t) il * K);
  assscores = calloc(K, sizeof(double));
  rc = orc_match_iso_paired(g, pos, cigar, npos, readLength, overHang, mean, var, numDevs,
                            match, fraglen); /* :378-383 */
  if (rc) { free(fp); free(match); free(fraglen); free(order); free(pisoscores); free(assscores); return rc; }
  order_cols(match, K, N, 0, order); /* :384 */
  if (bin_templates && bin_counts) { /* :386-391 -> classes2 (:628-681) */
    int *border = malloc(sizeof(int) * (N > 0 ? N : 1));
    order_cols(match, K, N, 1, border);
    classes(match, K, N, border, 1, bin_templates, bin_counts, ncls);
    free(border);
  }
  for (j = 0; j < il; j++) { /* :403-419 */
    double logprob = fp[j];
    for (i = 0; i < K; i++) {
      double lp = g->isolen[i] - fs - j + 1 - 2 * (g->noexons[i] - 1) * (overHang - 1);
      pisoscores[(size_t) i * il + j] = -log(lp) + logprob;
      if (lp > 0) assscores[i] += lp;
    }
  }
  for (i = 0; i < K; i++) assscores[i] = log(assscores[i]);

  fill_common(&S, g, opts, hyper, noChains, N);
  S.paired = 1; S.fs = fs; S.il = il;
  S.match = match; S.order = order; S.fraglen = fraglen; S.pisoscores = pisoscores;
  S.assscores = assscores;
  if (S.counter) { /* fixed-point per-read scores: llrint(S * 2^MISO_SFIX_BITS) */
    sfix = malloc(sizeof(int64_t) * (size_t) K * (N > 0 ? N : 1));
    for (i = 0; i < N; i++) for (j = 0; j < K; j++) {
      int fl = fraglen[(size_t) i * K + j];
      int64_t v = SFIX_BAD;
      if (fl >= 0) {
        double s = pisoscores[(size_t) j * il + (fl - fs)];
        if (isfinite(s) && fabs(s) < 31.0) v = (int64_t) llrint(s * MISO_SFIX_SCALE);
      }
      sfix[(size_t) i * K + j] = v;
    }
    S.sfix = sfix;
  }
  memset(samples, 0, sizeof(double) * (size_t) K * noSamples);
  memset(logLik, 0, sizeof(double) * noSamples);

  run_rounds(&S, start, stop, noIterations, maxIterations, noBurnIn, noLag, samples, logLik, rundata, trace);

  for (i = 0; i < N; i++) assignment[i] = S.ass[i];
  if (match_out) memcpy(match_out, match, sizeof(double) * (size_t) K * N);
  free_common(&S);
  free(fp); free(match); free(fraglen); free(order); free(pisoscores); free(assscores); free(sfix);
  return ORC_SUCCESS;
}
