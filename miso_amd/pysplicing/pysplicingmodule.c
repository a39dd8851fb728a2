/*
 * pysplicingmodule.c -- CPython-3 extension `pysplicing.pysplicing`: the drop-in for the reference's
 * CPython-2 module (pysplicing/src/pysplicing.c, pyconvert.c, pyerror.c), bound to libmiso_amd.so.
 *
 * Same names, positional order, defaults and return shapes as the reference's method table
 * (pysplicing.c:659-685) for everything on the sampler path:
 *
 *   createGene(exons, isoforms[, id, seqid, source, strand])      pysplicing.c:246-278
 *   MISO(gff, gene, readpos, readcigar, readLength[, noIterations=5000, noBurnIn=500, noLag=10,
 *        hyperp, overhang=1, no_chains=6, start, stop, algo])     pysplicing.c:41-131
 *   MISOPaired(gff, gene, readpos, readcigar, readLength, normalMean, normalVar, numDevs
 *        [, noIterations, noBurnIn, noLag, hyperp, overhang, no_chains, start, stop])
 *                                                                 pysplicing.c:152-244
 *   noIso(gff), isoLength(gff)                                    pysplicing.c:405-460
 *   simulateReads / simulatePairedReads                           pysplicing.c:280-330, 462-520
 *
 * Argument conversion follows pyconvert.c: sequences must be TUPLES (pyconvert.c:7-10, 24-27,
 * 41-44), exons are (start, end) pairs, isoforms tuples of exon indices.  Results are the same
 * 6-tuple: (samples: K tuples of S floats, logLik, class_templates: ncls tuples of K floats,
 * class_counts, assignment, rundata = (noIso, noIters, noBurnIn, noLag, noAccepted, noRejected)).
 * Errors become InternalError / MemoryError / NotImplementedError with the text
 * "Error at file:line: reason, strerror" (pyerror.c:27-44).
 *
 * Additions (keyword-only, the reference has no counterpart):
 *   seed=...        the reference draws from Python's `random` (pyrandom.c:163-184); here the
 *                   Philox seed defaults to random.getrandbits(64), so random.seed(n) makes a run
 *                   reproducible just as it does for the reference;
 *   MISOBatch / MISOPairedBatch(events, ...)  many events per GPU launch; events is a tuple of
 *                   (gff, readpos, readcigar[, hyperp]); returns a list of 6-tuples;
 *   first_event_id  global index of the first event (sharding a run over GPUs).
 * The functions of the module that are not on the sampler path raise NotImplementedError.
 * The reference's stdout side effect `printf("no chains: %d\n")` (miso.c:837) is not reproduced.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <stdlib.h>
#include <string.h>

#include "miso_amd.h"

static PyObject *InternalError;

#define GENE_CAPSULE "pysplicing.gff"

static PyObject *raise_miso(int rc) {
  const char *msg = miso_last_error();
  if (rc == MISO_ENOMEM) PyErr_SetString(PyExc_MemoryError, msg);
  else if (rc == MISO_UNIMPLEMENTED) PyErr_SetString(PyExc_NotImplementedError, msg);
  else PyErr_SetString(InternalError, msg);
  return NULL;
}

static void gene_capsule_free(PyObject *cap) {
  miso_gene_destroy((miso_gene_t *) PyCapsule_GetPointer(cap, GENE_CAPSULE));
}

static miso_gene_t *as_gene(PyObject *o) {
  if (!PyCapsule_IsValid(o, GENE_CAPSULE)) {
    PyErr_SetString(PyExc_TypeError, "expected a gene handle made by pysplicing.createGene");
    return NULL;
  }
  return (miso_gene_t *) PyCapsule_GetPointer(o, GENE_CAPSULE);
}

/* ---- pyconvert.c counterparts (tuples only) ---- */

static int to_int_vector(PyObject *o, int **out, Py_ssize_t *n) {
  Py_ssize_t i;
  if (!PyTuple_Check(o)) { PyErr_SetString(PyExc_TypeError, "Need a tuple"); return -1; }
  *n = PyTuple_Size(o);
  *out = malloc(sizeof(int) * (*n > 0 ? *n : 1));
  if (!*out) { PyErr_NoMemory(); return -1; }
  for (i = 0; i < *n; i++) {
    long v = PyLong_AsLong(PyTuple_GET_ITEM(o, i));
    if (v == -1 && PyErr_Occurred()) { free(*out); return -1; }
    (*out)[i] = (int) v;
  }
  return 0;
}

static int to_double_vector(PyObject *o, double **out, Py_ssize_t *n) {
  Py_ssize_t i;
  if (!PyTuple_Check(o)) { PyErr_SetString(PyExc_TypeError, "Need a tuple"); return -1; }
  *n = PyTuple_Size(o);
  *out = malloc(sizeof(double) * (*n > 0 ? *n : 1));
  if (!*out) { PyErr_NoMemory(); return -1; }
  for (i = 0; i < *n; i++) {
    double v = PyFloat_AsDouble(PyTuple_GET_ITEM(o, i));
    if (v == -1.0 && PyErr_Occurred()) { free(*out); return -1; }
    (*out)[i] = v;
  }
  return 0;
}

/* borrowed UTF-8 pointers; `keep` holds the bytes objects alive */
static int to_str_vector(PyObject *o, const char ***out, Py_ssize_t *n) {
  Py_ssize_t i;
  if (!PyTuple_Check(o)) { PyErr_SetString(PyExc_TypeError, "Need a tuple"); return -1; }
  *n = PyTuple_Size(o);
  *out = malloc(sizeof(char *) * (*n > 0 ? *n : 1));
  if (!*out) { PyErr_NoMemory(); return -1; }
  for (i = 0; i < *n; i++) {
    PyObject *it = PyTuple_GET_ITEM(o, i);
    const char *s = PyBytes_Check(it) ? PyBytes_AsString(it) : PyUnicode_AsUTF8(it);
    if (!s) { free(*out); return -1; }
    (*out)[i] = s;
  }
  return 0;
}

static int to_exons(PyObject *o, int **out, Py_ssize_t *nex) {
  Py_ssize_t i;
  if (!PyTuple_Check(o)) { PyErr_SetString(PyExc_TypeError, "Need a tuple"); return -1; }
  *nex = PyTuple_Size(o);
  *out = malloc(sizeof(int) * 2 * (*nex > 0 ? *nex : 1));
  if (!*out) { PyErr_NoMemory(); return -1; }
  for (i = 0; i < *nex; i++) {
    PyObject *it = PyTuple_GET_ITEM(o, i);
    if (!PyTuple_Check(it) || PyTuple_Size(it) != 2) {
      PyErr_SetString(PyExc_TypeError, "Exons must be (start, end) tuples");
      free(*out); return -1;
    }
    (*out)[2 * i] = (int) PyLong_AsLong(PyTuple_GET_ITEM(it, 0));
    (*out)[2 * i + 1] = (int) PyLong_AsLong(PyTuple_GET_ITEM(it, 1));
    if (PyErr_Occurred()) { free(*out); return -1; }
  }
  return 0;
}

static int to_isoforms(PyObject *o, int **out, Py_ssize_t *nflat) {
  Py_ssize_t i, j, n, tot = 0, p = 0;
  if (!PyTuple_Check(o)) { PyErr_SetString(PyExc_TypeError, "Need a tuple"); return -1; }
  n = PyTuple_Size(o);
  for (i = 0; i < n; i++) {
    PyObject *it = PyTuple_GET_ITEM(o, i);
    if (!PyTuple_Check(it)) { PyErr_SetString(PyExc_TypeError, "Need a tuple"); return -1; }
    tot += PyTuple_Size(it) + 1;
  }
  *out = malloc(sizeof(int) * (tot > 0 ? tot : 1));
  if (!*out) { PyErr_NoMemory(); return -1; }
  for (i = 0; i < n; i++) {
    PyObject *it = PyTuple_GET_ITEM(o, i);
    for (j = 0; j < PyTuple_Size(it); j++) {
      (*out)[p++] = (int) PyLong_AsLong(PyTuple_GET_ITEM(it, j));
      if (PyErr_Occurred()) { free(*out); return -1; }
    }
    (*out)[p++] = -1;
  }
  *nflat = tot;
  return 0;
}

static PyObject *from_doubles(const double *v, Py_ssize_t n) {
  Py_ssize_t i; PyObject *t = PyTuple_New(n);
  if (!t) return NULL;
  for (i = 0; i < n; i++) PyTuple_SET_ITEM(t, i, PyFloat_FromDouble(v[i]));
  return t;
}
static PyObject *from_ints(const int *v, Py_ssize_t n) {
  Py_ssize_t i; PyObject *t = PyTuple_New(n);
  if (!t) return NULL;
  for (i = 0; i < n; i++) PyTuple_SET_ITEM(t, i, PyLong_FromLong(v[i]));
  return t;
}
/* column-major nrow x ncol -> tuple of nrow tuples (pyconvert.c pysplicing_from_matrix) */
static PyObject *from_matrix_rows(const double *m, Py_ssize_t nrow, Py_ssize_t ncol) {
  Py_ssize_t i, j; PyObject *t = PyTuple_New(nrow);
  if (!t) return NULL;
  for (i = 0; i < nrow; i++) {
    PyObject *r = PyTuple_New(ncol);
    for (j = 0; j < ncol; j++) PyTuple_SET_ITEM(r, j, PyFloat_FromDouble(m[j * nrow + i]));
    PyTuple_SET_ITEM(t, i, r);
  }
  return t;
}

static int default_seed(unsigned long long *seed) {
  /* the reference's stream is Python's `random` (pyrandom.c:163-184): seed from it */
  PyObject *mod = PyImport_ImportModule("random"), *v;
  if (!mod) return -1;
  v = PyObject_CallMethod(mod, "getrandbits", "i", 64);
  Py_DECREF(mod);
  if (!v) return -1;
  *seed = PyLong_AsUnsignedLongLong(v);
  Py_DECREF(v);
  return PyErr_Occurred() ? -1 : 0;
}

/* ---- createGene (pysplicing.c:246-278) ---- */

static PyObject *py_create_gene(PyObject *self, PyObject *args) {
  PyObject *exons, *isoforms;
  const char *id = "insilicogene", *seqid = "seq1", *source = "protein_coding";
  int strand = 2, *ex = NULL, *iso = NULL, rc;
  Py_ssize_t nex, nflat;
  miso_gene_t *g = NULL;
  if (!PyArg_ParseTuple(args, "OO|sssi", &exons, &isoforms, &id, &seqid, &source, &strand)) return NULL;
  if (to_exons(exons, &ex, &nex)) return NULL;
  if (to_isoforms(isoforms, &iso, &nflat)) { free(ex); return NULL; }
  rc = miso_create_gene(ex, (int) nex, iso, (int) nflat, id, seqid, source, strand, &g);
  free(ex); free(iso);
  if (rc) return raise_miso(rc);
  return PyCapsule_New(g, GENE_CAPSULE, gene_capsule_free);
}

static PyObject *py_no_iso(PyObject *self, PyObject *args) {
  PyObject *gff; int n, rc; miso_gene_t *g;
  if (!PyArg_ParseTuple(args, "O", &gff)) return NULL;
  if (!(g = as_gene(gff))) return NULL;
  if ((rc = miso_gene_noiso(g, &n))) return raise_miso(rc);
  return Py_BuildValue("(i)", n); /* one entry per gene, as splicing_gff_noiso */
}

static PyObject *py_iso_length(PyObject *self, PyObject *args) {
  PyObject *gff, *inner, *outer; int n, rc, len[MISO_MAX_ISOFORMS * 4]; miso_gene_t *g;
  if (!PyArg_ParseTuple(args, "O", &gff)) return NULL;
  if (!(g = as_gene(gff))) return NULL;
  if ((rc = miso_gene_noiso(g, &n))) return raise_miso(rc);
  if (n > (int) (sizeof(len) / sizeof(len[0]))) { PyErr_SetString(InternalError, "too many isoforms"); return NULL; }
  if ((rc = miso_gene_isolength(g, len))) return raise_miso(rc);
  inner = from_ints(len, n);
  outer = PyTuple_New(1);
  PyTuple_SET_ITEM(outer, 0, inner);
  return outer;
}

/* ---- results ---- */

/* class_templates come back K x ncls column-major (one class per column); the module returns one
   tuple per class (pysplicing.c:120 transposes before converting) */
static PyObject *templates_rows(const double *ct, int K, int ncls) {
  Py_ssize_t i, j; PyObject *t = PyTuple_New(ncls);
  if (!t) return NULL;
  for (i = 0; i < ncls; i++) {
    PyObject *r = PyTuple_New(K);
    for (j = 0; j < K; j++) PyTuple_SET_ITEM(r, j, PyFloat_FromDouble(ct[i * K + j]));
    PyTuple_SET_ITEM(t, i, r);
  }
  return t;
}

static PyObject *result_tuple(miso_batch_t *b, int idx) {
  int K, N, S, ncls, rc;
  double *samples = NULL, *ll = NULL, *ct = NULL, *cc = NULL; int *ass = NULL;
  miso_rundata_t rd;
  PyObject *out = NULL;
  if ((rc = miso_batch_event_info(b, idx, &K, &N, &S, &ncls))) return raise_miso(rc);
  samples = malloc(sizeof(double) * (size_t) (K * (S > 0 ? S : 1)));
  ll = malloc(sizeof(double) * (size_t) (S > 0 ? S : 1));
  ct = malloc(sizeof(double) * (size_t) (K * (ncls > 0 ? ncls : 1)));
  cc = malloc(sizeof(double) * (size_t) (ncls > 0 ? ncls : 1));
  ass = malloc(sizeof(int) * (size_t) (N > 0 ? N : 1));
  if (!samples || !ll || !ct || !cc || !ass) { PyErr_NoMemory(); goto done; }
  if ((rc = miso_batch_get_result(b, idx, samples, ll, ct, cc, ass, &rd))) { raise_miso(rc); goto done; }
  out = Py_BuildValue("(NNNNN(iiiiii))", from_matrix_rows(samples, K, S), from_doubles(ll, S),
                      templates_rows(ct, K, ncls), from_doubles(cc, ncls), from_ints(ass, N),
                      rd.noIso, rd.noIters, rd.noBurnIn, rd.noLag, rd.noAccepted, rd.noRejected);
done:
  free(samples); free(ll); free(ct); free(cc); free(ass);
  return out;
}

/* add one event (gene, readpos, readcigar, hyperp or NULL) to a batch */
static int add_event(miso_batch_t *b, PyObject *gff, PyObject *readpos, PyObject *readcigar,
                     PyObject *hyperp) {
  int *pos = NULL; const char **cig = NULL; double *hy = NULL;
  Py_ssize_t npos = 0, ncig = 0, nhy = 0; int rc;
  miso_gene_t *g = as_gene(gff);
  if (!g) return -1;
  if (to_int_vector(readpos, &pos, &npos)) return -1;
  if (to_str_vector(readcigar, &cig, &ncig)) { free(pos); return -1; }
  if (hyperp && hyperp != Py_None && to_double_vector(hyperp, &hy, &nhy)) { free(pos); free(cig); return -1; }
  if (ncig < npos) {
    PyErr_SetString(PyExc_ValueError, "fewer CIGAR strings than read positions");
    free(pos); free(cig); free(hy); return -1;
  }
  rc = miso_batch_add_event(b, g, pos, cig, (int) npos, hy, (int) nhy, NULL);
  free(pos); free(cig); free(hy);
  if (rc) { raise_miso(rc); return -1; }
  return 0;
}

static int run_batch(miso_batch_t *b, unsigned long long seed, unsigned int first_event_id) {
  int rc, dev = 0;
  const char *env = getenv("LOCAL_RANK");  /* one process per GPU: torchrun convention */
  if (env) dev = atoi(env);
  if ((env = getenv("MISO_DEVICE"))) dev = atoi(env);
  Py_BEGIN_ALLOW_THREADS
  rc = miso_batch_run(b, dev, (uint64_t) seed, (uint32_t) first_event_id);
  Py_END_ALLOW_THREADS
  if (rc) { raise_miso(rc); return -1; }
  return 0;
}

static void fill_params(miso_params_t *p, int paired, int readLength, int overhang, int chains,
                        int iters, int burn, int lag, int algo, int start, int stop, double mean,
                        double var, double devs) {
  memset(p, 0, sizeof(*p));
  p->paired = paired; p->readLength = readLength; p->overHang = overhang; p->noChains = chains;
  p->noIterations = iters; p->maxIterations = 100000 /* pysplicing.c:43 */; p->noBurnIn = burn;
  p->noLag = lag; p->algorithm = algo; p->start = start; p->stop = stop;
  p->normalMean = mean; p->normalVar = var; p->numDevs = devs;
  p->device_match = 1;   /* read x isoform compatibility on the GPU too (solve.c:8-108, 141-218) */
}

/* ---- MISO (pysplicing.c:41-131) ---- */

static PyObject *py_miso(PyObject *self, PyObject *args, PyObject *kw) {
  static char *kwlist[] = {"gff", "gene", "readpos", "readcigar", "readLength", "noIterations",
                           "noBurnIn", "noLag", "hyperp", "overhang", "no_chains", "start", "stop",
                           "algo", "seed", NULL};
  PyObject *gff, *readpos, *readcigar, *hyperp = NULL, *seedobj = NULL, *res;
  int gene, readLength, iters = 5000, burn = 500, lag = 10, overhang = 1, chains = 6;
  int start = MISO_START_AUTO, stop = MISO_STOP_FIXEDNO, algo = MISO_ALGO_REASSIGN, rc;
  unsigned long long seed;
  miso_params_t p; miso_batch_t *b = NULL;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "OiOOi|iiiOiiiii$O", kwlist, &gff, &gene, &readpos,
                                   &readcigar, &readLength, &iters, &burn, &lag, &hyperp, &overhang,
                                   &chains, &start, &stop, &algo, &seedobj)) return NULL;
  if (gene != 0) { PyErr_SetString(InternalError, "Invalid gene id"); return NULL; }
  if (seedobj && seedobj != Py_None) { seed = PyLong_AsUnsignedLongLongMask(seedobj); if (PyErr_Occurred()) return NULL; }
  else if (default_seed(&seed)) return NULL;
  fill_params(&p, 0, readLength, overhang, chains, iters, burn, lag, algo, start, stop, 0, 0, 0);
  if ((rc = miso_batch_create(&p, &b))) return raise_miso(rc);
  if (add_event(b, gff, readpos, readcigar, hyperp) || run_batch(b, seed, 0)) { miso_batch_destroy(b); return NULL; }
  res = result_tuple(b, 0);
  miso_batch_destroy(b);
  return res;
}

/* ---- MISOPaired (pysplicing.c:152-244) ---- */

static PyObject *py_miso_paired(PyObject *self, PyObject *args, PyObject *kw) {
  static char *kwlist[] = {"gff", "gene", "readpos", "readcigar", "readLength", "normalMean",
                           "normalVar", "numDevs", "noIterations", "noBurnIn", "noLag", "hyperp",
                           "overhang", "no_chains", "start", "stop", "seed", NULL};
  PyObject *gff, *readpos, *readcigar, *hyperp = NULL, *seedobj = NULL, *res;
  int gene, readLength, iters = 5000, burn = 500, lag = 10, overhang = 1, chains = 6;
  int start = MISO_START_AUTO, stop = MISO_STOP_FIXEDNO, rc;
  double mean, var, devs; unsigned long long seed;
  miso_params_t p; miso_batch_t *b = NULL;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "OiOOiddd|iiiOiiii$O", kwlist, &gff, &gene, &readpos,
                                   &readcigar, &readLength, &mean, &var, &devs, &iters, &burn, &lag,
                                   &hyperp, &overhang, &chains, &start, &stop, &seedobj)) return NULL;
  if (gene != 0) { PyErr_SetString(InternalError, "Invalid gene id"); return NULL; }
  if (seedobj && seedobj != Py_None) { seed = PyLong_AsUnsignedLongLongMask(seedobj); if (PyErr_Occurred()) return NULL; }
  else if (default_seed(&seed)) return NULL;
  fill_params(&p, 1, readLength, overhang, chains, iters, burn, lag, MISO_ALGO_REASSIGN, start, stop,
              mean, var, devs);
  if ((rc = miso_batch_create(&p, &b))) return raise_miso(rc);
  if (add_event(b, gff, readpos, readcigar, hyperp) || run_batch(b, seed, 0)) { miso_batch_destroy(b); return NULL; }
  res = result_tuple(b, 0);
  miso_batch_destroy(b);
  return res;
}

/* ---- MISOBatch / MISOPairedBatch: many events, one launch ---- */

/* summary=<confidence level>: each result gains a 7th element (means, ci_low, ci_high), computed
   on the device from the samples (what summarize_miso derives later from the .miso file,
   credible_intervals.py:4-72) */
static PyObject *summary_tuple(miso_batch_t *b, int idx) {
  int K, rc; double m[MISO_MAX_ISOFORMS], lo[MISO_MAX_ISOFORMS], hi[MISO_MAX_ISOFORMS];
  if ((rc = miso_batch_event_info(b, idx, &K, NULL, NULL, NULL))) return raise_miso(rc);
  if ((rc = miso_batch_get_summary(b, idx, m, lo, hi))) return raise_miso(rc);
  return Py_BuildValue("(NNN)", from_doubles(m, K), from_doubles(lo, K), from_doubles(hi, K));
}

static int parse_level(PyObject *obj, double *conf) {
  *conf = 0;
  if (!obj || obj == Py_None) return 0;
  *conf = PyFloat_AsDouble(obj);
  if (PyErr_Occurred()) return -1;
  if (!(*conf > 0 && *conf < 1)) { PyErr_SetString(PyExc_ValueError, "summary must be a confidence level in (0, 1)"); return -1; }
  return 0;
}

/* a new batch holding `events` = ((gff, readpos, readcigar[, hyperp]), ...) */
static miso_batch_t *fill_batch(PyObject *events, miso_params_t *p) {
  miso_batch_t *b = NULL; Py_ssize_t i, n; int rc;
  if (!PyTuple_Check(events)) { PyErr_SetString(PyExc_TypeError, "Need a tuple"); return NULL; }
  if ((rc = miso_batch_create(p, &b))) { raise_miso(rc); return NULL; }
  n = PyTuple_Size(events);
  for (i = 0; i < n; i++) {
    PyObject *ev = PyTuple_GET_ITEM(events, i);
    if (!PyTuple_Check(ev) || PyTuple_Size(ev) < 3 || PyTuple_Size(ev) > 4) {
      PyErr_SetString(PyExc_TypeError, "each event must be (gff, readpos, readcigar[, hyperp])");
      miso_batch_destroy(b); return NULL;
    }
    if (add_event(b, PyTuple_GET_ITEM(ev, 0), PyTuple_GET_ITEM(ev, 1), PyTuple_GET_ITEM(ev, 2),
                  PyTuple_Size(ev) == 4 ? PyTuple_GET_ITEM(ev, 3) : NULL)) { miso_batch_destroy(b); return NULL; }
  }
  return b;
}

/* list of the 6-tuples of a finished batch; 7-tuples with the summary when conf > 0 */
static PyObject *results_list(miso_batch_t *b, Py_ssize_t n, double conf) {
  Py_ssize_t i; int rc;
  PyObject *out;
  if (n > 0 && conf > 0 && (rc = miso_batch_summarize(b, conf))) return raise_miso(rc);
  out = PyList_New(n);
  for (i = 0; out && i < n; i++) {
    PyObject *r = result_tuple(b, (int) i);
    if (r && conf > 0) {
      PyObject *s = summary_tuple(b, (int) i), *r7 = NULL;
      PyObject *one = s ? PyTuple_Pack(1, s) : NULL;
      if (one) r7 = PySequence_Concat(r, one);
      Py_XDECREF(one); Py_XDECREF(s); Py_DECREF(r); r = r7;
    }
    if (!r) { Py_CLEAR(out); break; }
    PyList_SET_ITEM(out, i, r);
  }
  return out;
}

static PyObject *batch_common(PyObject *events, miso_params_t *p, PyObject *seedobj,
                              unsigned int first_event_id, PyObject *summaryobj) {
  double conf; miso_batch_t *b; PyObject *out = NULL; Py_ssize_t n;
  unsigned long long seed;
  if (seedobj && seedobj != Py_None) { seed = PyLong_AsUnsignedLongLongMask(seedobj); if (PyErr_Occurred()) return NULL; }
  else if (default_seed(&seed)) return NULL;
  if (parse_level(summaryobj, &conf)) return NULL;
  if (!(b = fill_batch(events, p))) return NULL;
  n = PyTuple_Size(events);
  if (n == 0 || !run_batch(b, seed, first_event_id)) out = results_list(b, n, conf);
  miso_batch_destroy(b);
  return out;
}

/* ---- MISOCompareBatch: two RNA-seq samples over the same events, Bayes factors on the GPU ----
   (compare_miso: misopy/hypothesis_test.py:89-179, 348-380) */

static PyObject *py_miso_compare_batch(PyObject *self, PyObject *args, PyObject *kw) {
  static char *kwlist[] = {"events1", "events2", "readLength", "noIterations", "noBurnIn", "noLag",
                           "overhang", "no_chains", "start", "stop", "seed", "seed2", "first_event_id",
                           "summary", "smoothing", "paired", "event_ids", NULL};
  PyObject *ev1, *ev2, *seedobj = NULL, *seed2obj = NULL, *summaryobj = NULL, *pairedobj = NULL, *idsobj = NULL;
  PyObject *r1 = NULL, *r2 = NULL, *cmp = NULL, *out = NULL;
  int readLength, iters = 5000, burn = 500, lag = 10, overhang = 1, chains = 6;
  int start = MISO_START_AUTO, stop = MISO_STOP_FIXEDNO, rc;
  unsigned int first = 0; double smoothing = 0.3, conf = 0.95, mean = 0, var = 0, devs = 0;
  unsigned long long seed, seed2; Py_ssize_t i, n;
  miso_params_t p; miso_batch_t *b1 = NULL, *b2 = NULL;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "OOi|iiiiiii$OOIOdOO", kwlist, &ev1, &ev2, &readLength, &iters,
                                   &burn, &lag, &overhang, &chains, &start, &stop, &seedobj, &seed2obj,
                                   &first, &summaryobj, &smoothing, &pairedobj, &idsobj)) return NULL;
  if (seedobj && seedobj != Py_None) { seed = PyLong_AsUnsignedLongLongMask(seedobj); if (PyErr_Occurred()) return NULL; }
  else if (default_seed(&seed)) return NULL;
  /* the two samples must not share random numbers: identical draws would correlate the chains */
  if (seed2obj && seed2obj != Py_None) { seed2 = PyLong_AsUnsignedLongLongMask(seed2obj); if (PyErr_Occurred()) return NULL; }
  else seed2 = seed ^ 0x5851F42D4C957F2DULL;
  if (seed2 == seed) { PyErr_SetString(PyExc_ValueError, "seed2 must differ from seed"); return NULL; }
  if (summaryobj && summaryobj != Py_None && parse_level(summaryobj, &conf)) return NULL;
  if (pairedobj && pairedobj != Py_None &&
      !PyArg_ParseTuple(pairedobj, "ddd;paired must be (normalMean, normalVar, numDevs)", &mean, &var, &devs)) return NULL;
  fill_params(&p, pairedobj && pairedobj != Py_None, readLength, overhang, chains, iters, burn, lag,
              MISO_ALGO_REASSIGN, start, stop, mean, var, devs);
  if (!PyTuple_Check(ev1) || !PyTuple_Check(ev2)) { PyErr_SetString(PyExc_TypeError, "Need a tuple"); return NULL; }
  n = PyTuple_Size(ev1);
  if (PyTuple_Size(ev2) != n) { PyErr_SetString(PyExc_ValueError, "the two samples must list the same events"); return NULL; }
  if (!(b1 = fill_batch(ev1, &p))) goto done;
  if (!(b2 = fill_batch(ev2, &p))) goto done;
  /* event_ids: every event's id in the Philox counter (its number in the caller's full event list), so
     that events dropped by the caller's skip rules, the chunking and the number of GPUs change nobody's
     random stream -- as miso_batch_set_event_id does for MISOBatch's callers */
  if (idsobj && idsobj != Py_None) {
    if (!PyTuple_Check(idsobj) || PyTuple_Size(idsobj) != n) {
      PyErr_SetString(PyExc_ValueError, "event_ids must be a tuple with one id per event"); goto done;
    }
    for (i = 0; i < n; i++) {
      unsigned long id = PyLong_AsUnsignedLongMask(PyTuple_GET_ITEM(idsobj, i));
      if (PyErr_Occurred()) goto done;
      if ((rc = miso_batch_set_event_id(b1, (int) i, (uint32_t) id)) ||
          (rc = miso_batch_set_event_id(b2, (int) i, (uint32_t) id))) { raise_miso(rc); goto done; }
    }
  }
  if (n > 0 && (run_batch(b1, seed, first) || run_batch(b2, seed2, first))) goto done;
  if (!(r1 = results_list(b1, n, conf)) || !(r2 = results_list(b2, n, conf))) goto done;
  if (n > 0 && (rc = miso_batch_compare(b1, b2, smoothing))) { raise_miso(rc); goto done; }
  if (!(cmp = PyList_New(n))) goto done;
  for (i = 0; i < n; i++) {
    int K; double m1[MISO_MAX_ISOFORMS], m2[MISO_MAX_ISOFORMS], bf[MISO_MAX_ISOFORMS], d0[MISO_MAX_ISOFORMS];
    PyObject *t;
    if ((rc = miso_batch_event_info(b1, (int) i, &K, NULL, NULL, NULL)) ||
        (rc = miso_batch_get_comparison(b1, (int) i, m1, m2, bf, d0))) { raise_miso(rc); goto done; }
    t = Py_BuildValue("(NNNN)", from_doubles(m1, K), from_doubles(m2, K), from_doubles(bf, K), from_doubles(d0, K));
    if (!t) goto done;
    PyList_SET_ITEM(cmp, i, t);
  }
  out = Py_BuildValue("(OOO)", r1, r2, cmp);
done:
  Py_XDECREF(r1); Py_XDECREF(r2); Py_XDECREF(cmp);
  if (b1) miso_batch_destroy(b1);
  if (b2) miso_batch_destroy(b2);
  return out;
}

static PyObject *py_miso_batch(PyObject *self, PyObject *args, PyObject *kw) {
  static char *kwlist[] = {"events", "readLength", "noIterations", "noBurnIn", "noLag", "overhang",
                           "no_chains", "start", "stop", "algo", "seed", "first_event_id", "summary", NULL};
  PyObject *events, *seedobj = NULL, *summaryobj = NULL;
  int readLength, iters = 5000, burn = 500, lag = 10, overhang = 1, chains = 6;
  int start = MISO_START_AUTO, stop = MISO_STOP_FIXEDNO, algo = MISO_ALGO_REASSIGN;
  unsigned int first = 0; miso_params_t p;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "Oi|iiiiiiii$OIO", kwlist, &events, &readLength, &iters,
                                   &burn, &lag, &overhang, &chains, &start, &stop, &algo, &seedobj,
                                   &first, &summaryobj)) return NULL;
  fill_params(&p, 0, readLength, overhang, chains, iters, burn, lag, algo, start, stop, 0, 0, 0);
  return batch_common(events, &p, seedobj, first, summaryobj);
}

static PyObject *py_miso_paired_batch(PyObject *self, PyObject *args, PyObject *kw) {
  static char *kwlist[] = {"events", "readLength", "normalMean", "normalVar", "numDevs",
                           "noIterations", "noBurnIn", "noLag", "overhang", "no_chains", "start",
                           "stop", "seed", "first_event_id", "summary", NULL};
  PyObject *events, *seedobj = NULL, *summaryobj = NULL;
  int readLength, iters = 5000, burn = 500, lag = 10, overhang = 1, chains = 6;
  int start = MISO_START_AUTO, stop = MISO_STOP_FIXEDNO;
  double mean, var, devs; unsigned int first = 0; miso_params_t p;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "Oiddd|iiiiiii$OIO", kwlist, &events, &readLength, &mean,
                                   &var, &devs, &iters, &burn, &lag, &overhang, &chains, &start,
                                   &stop, &seedobj, &first, &summaryobj)) return NULL;
  fill_params(&p, 1, readLength, overhang, chains, iters, burn, lag, MISO_ALGO_REASSIGN, start, stop,
              mean, var, devs);
  return batch_common(events, &p, seedobj, first, summaryobj);
}

/* ---- simulateReads / simulatePairedReads (pysplicing.c:280-330, 462-520) ---- */

static PyObject *simulate_common(PyObject *gff, PyObject *expression, int n, int readLength,
                                 double mean, double var, double devs, PyObject *seedobj) {
  miso_gene_t *g = as_gene(gff); double *expr = NULL; Py_ssize_t nexpr, i, tot;
  int *iso = NULL, *pos = NULL, rc; char *cig = NULL; const int stride = 128;
  unsigned long long seed; PyObject *rc_iso, *rc_pos, *rc_cig;
  if (!g) return NULL;
  if (seedobj && seedobj != Py_None) { seed = PyLong_AsUnsignedLongLongMask(seedobj); if (PyErr_Occurred()) return NULL; }
  else if (default_seed(&seed)) return NULL;
  if (to_double_vector(expression, &expr, &nexpr)) return NULL;
  tot = (Py_ssize_t) n * (var > 0 ? 2 : 1);
  iso = malloc(sizeof(int) * (tot > 0 ? tot : 1)); pos = malloc(sizeof(int) * (tot > 0 ? tot : 1));
  cig = malloc((size_t) stride * (tot > 0 ? tot : 1));
  if (!iso || !pos || !cig) { free(expr); free(iso); free(pos); free(cig); return PyErr_NoMemory(); }
  rc = miso_simulate_reads(g, expr, n, readLength, mean, var, devs, (uint64_t) seed, iso, pos, cig, stride);
  free(expr);
  if (rc) { free(iso); free(pos); free(cig); return raise_miso(rc); }
  rc_iso = from_ints(iso, tot); rc_pos = from_ints(pos, tot);
  rc_cig = PyTuple_New(tot);
  for (i = 0; i < tot; i++) PyTuple_SET_ITEM(rc_cig, i, PyUnicode_FromString(cig + i * stride));
  free(iso); free(pos); free(cig);
  return Py_BuildValue("(NNN)", rc_iso, rc_pos, rc_cig);
}

static PyObject *py_simulate_reads(PyObject *self, PyObject *args, PyObject *kw) {
  static char *kwlist[] = {"gff", "gene", "expression", "noreads", "readLength", "seed", NULL};
  PyObject *gff, *expression, *seedobj = NULL; int gene, n, readLength;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "OiOii|$O", kwlist, &gff, &gene, &expression, &n,
                                   &readLength, &seedobj)) return NULL;
  return simulate_common(gff, expression, n, readLength, 0, 0, 0, seedobj);
}

static PyObject *py_simulate_paired_reads(PyObject *self, PyObject *args, PyObject *kw) {
  static char *kwlist[] = {"gff", "gene", "expression", "noreads", "readLength", "normalMean",
                           "normalVar", "numDevs", "seed", NULL};
  PyObject *gff, *expression, *seedobj = NULL; int gene, n, readLength; double mean, var, devs;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "OiOiiddd|$O", kwlist, &gff, &gene, &expression, &n,
                                   &readLength, &mean, &var, &devs, &seedobj)) return NULL;
  return simulate_common(gff, expression, n, readLength, mean, var, devs, seedobj);
}

/* assignmentMatrix(gff, gene, readLength[, overhang]) (pysplicing.c:324-349): K tuples (one per isoform) of ncls floats */
static PyObject *py_assignment_matrix(PyObject *self, PyObject *args) {
  PyObject *gff, *rows; int gene, readLength, overhang = 1, K, nc = 0, rc, i, j; miso_gene_t *g; double *m;
  if (!PyArg_ParseTuple(args, "Oii|i", &gff, &gene, &readLength, &overhang)) return NULL;
  if (!(g = as_gene(gff))) return NULL;
  if ((rc = miso_gene_noiso(g, &K))) return raise_miso(rc);
  m = (double *) PyMem_Malloc(sizeof(double) * (size_t) K * 8192);
  if (!m) return PyErr_NoMemory();
  rc = miso_gene_assignment_matrix(g, readLength, overhang, m, 8192, &nc);
  if (rc) { PyMem_Free(m); return raise_miso(rc); }
  rows = PyTuple_New(K);
  for (i = 0; rows && i < K; i++) {
    PyObject *r = PyTuple_New(nc);
    for (j = 0; r && j < nc; j++) PyTuple_SET_ITEM(r, j, PyFloat_FromDouble(m[(size_t) j * K + i]));
    PyTuple_SET_ITEM(rows, i, r);
  }
  PyMem_Free(m);
  return rows;
}

static PyObject *py_not_on_path(PyObject *self, PyObject *args) {
  PyErr_SetString(PyExc_NotImplementedError,
                  "this pysplicing function is not on the MISO sampler path and is not provided by "
                  "the MI355X build (only createGene, MISO, MISOPaired, noIso, isoLength, assignmentMatrix, "
                  "simulateReads, simulatePairedReads and the *Batch entry points are)");
  return NULL;
}

static PyObject *py_device_count(PyObject *self, PyObject *args) {
  int n = 0, rc = miso_device_count(&n);
  if (rc) return raise_miso(rc);
  return PyLong_FromLong(n);
}

static PyMethodDef methods[] = {
  {"createGene", py_create_gene, METH_VARARGS, "Create a gene from exons and isoforms"},
  {"MISO", (PyCFunction) py_miso, METH_VARARGS | METH_KEYWORDS, "Run MISO on a single gene (GPU)"},
  {"MISOPaired", (PyCFunction) py_miso_paired, METH_VARARGS | METH_KEYWORDS, "Run MISO on a single gene, paired-end reads (GPU)"},
  {"MISOBatch", (PyCFunction) py_miso_batch, METH_VARARGS | METH_KEYWORDS, "Run MISO on many genes in one GPU launch"},
  {"MISOPairedBatch", (PyCFunction) py_miso_paired_batch, METH_VARARGS | METH_KEYWORDS, "Paired-end MISOBatch"},
  {"MISOCompareBatch", (PyCFunction) py_miso_compare_batch, METH_VARARGS | METH_KEYWORDS, "Two samples over the same events: results, summaries and Bayes factors"},
  {"noIso", py_no_iso, METH_VARARGS, "Number of isoforms"},
  {"isoLength", py_iso_length, METH_VARARGS, "Length of the isoforms"},
  {"simulateReads", (PyCFunction) py_simulate_reads, METH_VARARGS | METH_KEYWORDS, "Simulate single-end reads"},
  {"simulatePairedReads", (PyCFunction) py_simulate_paired_reads, METH_VARARGS | METH_KEYWORDS, "Simulate paired-end reads"},
  {"deviceCount", py_device_count, METH_NOARGS, "Number of usable HIP devices"},
  {"readGFF", py_not_on_path, METH_VARARGS, "not provided"},
  {"writeGFF", py_not_on_path, METH_VARARGS, "not provided"},
  {"assignmentMatrix", py_assignment_matrix, METH_VARARGS, "The gene's possible read classes and their numbers of start positions"},
  {"solveIsoGene", py_not_on_path, METH_VARARGS, "not provided"},
  {"geneComplexity", py_not_on_path, METH_VARARGS, "not provided"},
  {"noGenes", py_not_on_path, METH_VARARGS, "not provided"},
  {"i_fromGFF", py_not_on_path, METH_VARARGS, "not provided"},
  {"toGFF", py_not_on_path, METH_VARARGS, "not provided"},
  {NULL, NULL, 0, NULL}
};

static struct PyModuleDef moduledef = {
  PyModuleDef_HEAD_INIT, "pysplicing", "MI355X-native drop-in for MISO's pysplicing sampler module", -1,
  methods, NULL, NULL, NULL, NULL
};

PyMODINIT_FUNC PyInit_pysplicing(void) {
  PyObject *m = PyModule_Create(&moduledef);
  if (!m) return NULL;
  InternalError = PyErr_NewException("pysplicing.InternalError", PyExc_Exception, NULL);
  Py_XINCREF(InternalError);
  if (PyModule_AddObject(m, "InternalError", InternalError) < 0) { Py_DECREF(m); return NULL; }
  return m;
}
