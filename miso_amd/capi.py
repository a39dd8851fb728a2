"""ctypes binding of libmiso_amd.so (include/miso_amd.h).

This is the only way Python reaches the sampler: there is no Python or CPU implementation of
the path behind it.  Importing works without a GPU (so the symbol table can be checked);
every sampler call raises ``InternalError`` when no HIP device is usable.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MISO_AMD_LIB", os.path.join(_HERE, "libmiso_amd.so"))

MISO_SUCCESS, MISO_FAILURE, MISO_ENOMEM, MISO_EINVAL = 0, 1, 2, 4
MISO_UNIMPLEMENTED, MISO_EINTERNAL, MISO_ENODEVICE = 12, 38, 60
MISO_MAX_ISOFORMS = 256     # include/miso_amd.h

# pysplicing/pysplicing/__init__.py:2-13
MISO_START_AUTO, MISO_START_UNIFORM, MISO_START_RANDOM, MISO_START_GIVEN, MISO_START_LINEAR = range(5)
MISO_STOP_FIXEDNO, MISO_STOP_CONVERGENT_MEAN = 0, 1
MISO_ALGO_REASSIGN, MISO_ALGO_MARGINAL, MISO_ALGO_CLASSES = 0, 1, 2


class InternalError(Exception):
    """pysplicing.InternalError (pysplicing.c:699-702)."""


class RunData(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("noIso", "noIters", "maxIters", "noBurnIn", "noLag",
                                       "noAccepted", "noRejected", "noChains", "noSamples")]


class Params(C.Structure):
    _fields_ = [("paired", C.c_int), ("readLength", C.c_int), ("overHang", C.c_int),
                ("noChains", C.c_int), ("noIterations", C.c_int), ("maxIterations", C.c_int),
                ("noBurnIn", C.c_int), ("noLag", C.c_int), ("algorithm", C.c_int),
                ("start", C.c_int), ("stop", C.c_int), ("normalMean", C.c_double),
                ("normalVar", C.c_double), ("numDevs", C.c_double),
                ("want_counts_trace", C.c_int), ("device_match", C.c_int)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("waves", C.c_double), ("trips", C.c_double),
                ("iterations", C.c_double), ("chains", C.c_double), ("words", C.c_double)]


_lib = None


def lib():
    """Load libmiso_amd.so; fail loudly if the HIP extension was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "miso_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as "
                "g; g.build()'` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.miso_last_error.restype = C.c_char_p
        _lib.miso_strerror.restype = C.c_char_p
        _lib.miso_batch_launch.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        _lib.miso_batch_run.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32]
        _lib.miso_batch_summarize.argtypes = [C.c_void_p, C.c_double]
        _lib.miso_batch_compare.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
    return _lib


def check(rc):
    """Raise like pyerror.c:27-44: MemoryError / NotImplementedError / InternalError."""
    if rc == MISO_SUCCESS:
        return
    msg = lib().miso_last_error().decode(errors="replace")
    if rc == MISO_ENOMEM:
        raise MemoryError(msg)
    if rc == MISO_UNIMPLEMENTED:
        raise NotImplementedError(msg)
    raise InternalError(msg)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _cigs(cigars):
    arr = (C.c_char_p * max(len(cigars), 1))()
    for i, c in enumerate(cigars):
        arr[i] = c if isinstance(c, bytes) else c.encode()
    return arr


def contract_version():
    """include/miso_philox.h MISO_CONTRACT_VERSION of the loaded library (no device needed)."""
    return int(lib().miso_contract_version())


def device_count():
    n = C.c_int(0)
    check(lib().miso_device_count(C.byref(n)))
    return n.value


def set_device(d):
    check(lib().miso_set_device(int(d)))


class Gene:
    """One gene: exons as (start, end) 1-based inclusive, isoforms as tuples of exon indices
    (py2c_gene.py:10-21 builds exactly these for the reference's createGene)."""

    def __init__(self, exons, isoforms, id="insilicogene", seqid="seq1", source="protein_coding",
                 strand=2):
        ex = np.asarray(exons, dtype=np.int32).reshape(-1)
        flat = []
        for iso in isoforms:
            flat.extend(int(e) for e in iso)
            flat.append(-1)
        flat = np.asarray(flat, dtype=np.int32)
        h = C.c_void_p()
        check(lib().miso_create_gene(_p(ex), len(ex) // 2, _p(flat), len(flat), id.encode(),
                                     seqid.encode(), source.encode(), int(strand), C.byref(h)))
        self.handle = h

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h and _lib is not None:
            _lib.miso_gene_destroy(h)

    @property
    def noiso(self):
        n = C.c_int(0)
        check(lib().miso_gene_noiso(self.handle, C.byref(n)))
        return n.value

    def isolength(self):
        out = np.zeros(self.noiso, np.int32)
        check(lib().miso_gene_isolength(self.handle, _p(out)))
        return out

    def assignment_matrix(self, read_len, overhang=1, max_cols=4096):
        """The gene's possible read classes (assignment.c:90-276): ncls x K, a class per row (include/miso_amd.h)."""
        K = self.noiso
        m = np.zeros(K * max_cols)
        n = C.c_int(0)
        check(lib().miso_gene_assignment_matrix(self.handle, int(read_len), int(overhang), _p(m), max_cols, C.byref(n)))
        return m[:K * n.value].reshape(n.value, K).copy()

    def match_iso(self, pos, cigars, read_len, overhang=1):
        pos = np.asarray(pos, dtype=np.int32)
        m = np.zeros((max(len(pos), 1), self.noiso))
        check(lib().miso_match_iso(self.handle, _p(pos), _cigs(cigars), len(pos), overhang,
                                   read_len, _p(m)))
        return m[:len(pos)]

    def match_iso_paired(self, pos, cigars, read_len, mean, var, num_devs=4.0, overhang=1):
        pos = np.asarray(pos, dtype=np.int32)
        n = len(pos) // 2
        m = np.zeros((max(n, 1), self.noiso))
        fl = np.zeros((max(n, 1), self.noiso), np.int32)
        check(lib().miso_match_iso_paired(self.handle, _p(pos), _cigs(cigars), len(pos), read_len,
                                          overhang, C.c_double(mean), C.c_double(var),
                                          C.c_double(num_devs), _p(m), _p(fl)))
        return m[:n], fl[:n]


class EventResult:
    def __init__(self, samples, loglik, templates, counts, assignment, rundata, counts_hash,
                 counts_trace):
        self.samples = samples              # [S, K]
        self.loglik = loglik                # [S]
        self.class_templates = templates    # [ncls, K]
        self.class_counts = counts          # [ncls]
        self.assignment = assignment        # [N]
        self.rundata = rundata              # RunData
        self.counts_hash = counts_hash      # [C] uint64
        self.counts_trace = counts_trace    # [(M+1), C, K] or None

    def as_tuple(self):
        """The 6-tuple pysplicing.MISO returns (pysplicing.c:112-130, pyconvert.c:172-181)."""
        rd = self.rundata
        K = self.samples.shape[1] if self.samples.ndim == 2 else 0
        return (tuple(tuple(float(v) for v in self.samples[:, k]) for k in range(K)),
                tuple(float(v) for v in self.loglik),
                tuple(tuple(float(v) for v in row) for row in self.class_templates),
                tuple(float(v) for v in self.class_counts),
                tuple(int(v) for v in self.assignment),
                (rd.noIso, rd.noIters, rd.noBurnIn, rd.noLag, rd.noAccepted, rd.noRejected))


class Batch:
    """Many events, one launch (miso_batch_* in include/miso_amd.h)."""

    def __init__(self, read_len, iters=5000, burn=500, lag=10, chains=6, overhang=1, paired=False,
                 mean=0.0, var=0.0, num_devs=4.0, start=MISO_START_AUTO, stop=MISO_STOP_FIXEDNO,
                 algo=MISO_ALGO_REASSIGN, max_iters=100000, counts_trace=False, device_match=False,
                 collapsed=False):
        """collapsed (single-end): the two-isoform events draw their assignment COUNTS directly (one exact binomial
        per iteration instead of one uniform per read; miso_batch_set_collapsed in include/miso_amd.h)."""
        self.params = Params(int(paired), read_len, overhang, chains, iters, max_iters, burn, lag,
                             algo, start, stop, mean, var, num_devs, int(counts_trace),
                             int(device_match))
        self.handle = C.c_void_p()
        check(lib().miso_batch_create(C.byref(self.params), C.byref(self.handle)))
        self.collapsed = int(collapsed)     # True / 1: two-isoform events; 2: events of any isoform count
        if collapsed:
            check(lib().miso_batch_set_collapsed(self.handle, int(collapsed)))

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h and _lib is not None:
            _lib.miso_batch_destroy(h)

    def add_event(self, gene, pos, cigars, hyper=None):
        pos = np.asarray(pos, dtype=np.int32)
        hy = None if hyper is None else np.asarray(hyper, dtype=np.float64)
        idx = C.c_int(-1)
        check(lib().miso_batch_add_event(self.handle, gene.handle, _p(pos), _cigs(cigars), len(pos),
                                         _p(hy), 0 if hy is None else len(hy), C.byref(idx)))
        return idx.value

    def set_event_id(self, idx, event_id):
        """Pin event idx's id in the Philox counter (default: first_event_id + idx)."""
        check(lib().miso_batch_set_event_id(self.handle, int(idx), C.c_uint32(int(event_id) & 0xFFFFFFFF)))

    def add_problem(self, match, isolen, noexons, fraglen=None, hyper=None):
        """match: [N, K] (C-order) as Gene.match_iso returns it."""
        match = np.ascontiguousarray(match, dtype=np.float64)
        N, K = match.shape
        isolen = np.asarray(isolen, dtype=np.int32)
        noexons = np.asarray(noexons, dtype=np.int32)
        fl = None if fraglen is None else np.ascontiguousarray(fraglen, dtype=np.int32)
        hy = None if hyper is None else np.asarray(hyper, dtype=np.float64)
        idx = C.c_int(-1)
        check(lib().miso_batch_add_problem(self.handle, K, N, _p(match), _p(fl), _p(isolen),
                                           _p(noexons), _p(hy), C.byref(idx)))
        return idx.value

    def add_simulated(self, gene, expression, n_reads, sim_seed, hyper=None):
        """Synthetic reads (miso_simulate_reads) -> add_event, all inside the library."""
        ex = np.asarray(expression, dtype=np.float64)
        hy = None if hyper is None else np.asarray(hyper, dtype=np.float64)
        idx = C.c_int(-1)
        check(lib().miso_batch_add_simulated(self.handle, gene.handle, _p(ex), int(n_reads),
                                             C.c_uint64(sim_seed), _p(hy),
                                             0 if hy is None else len(hy), C.byref(idx)))
        return idx.value

    def __len__(self):
        n = C.c_int(0)
        check(lib().miso_batch_size(self.handle, C.byref(n)))
        return n.value

    def upload(self, device=0):
        check(lib().miso_batch_upload(self.handle, int(device)))

    def launch(self, seed=0, first_event_id=0):
        check(lib().miso_batch_launch(self.handle, int(seed), int(first_event_id)))

    def sync(self):
        ms = C.c_float(0)
        check(lib().miso_batch_sync(self.handle, C.byref(ms)))
        return ms.value

    def download(self):
        check(lib().miso_batch_download(self.handle))

    def run(self, device=0, seed=0, first_event_id=0):
        check(lib().miso_batch_run(self.handle, int(device), int(seed), int(first_event_id)))

    def summarize(self, confidence_level=0.95, as_text=False):
        """Device-side posterior means and credible intervals of every event (no sample download).  as_text: of the
        samples as the `.miso` file hands them to summarize_miso, i.e. after "%.4f" (miso_batch_summarize_as_text)."""
        f = lib().miso_batch_summarize_as_text if as_text else lib().miso_batch_summarize
        f.argtypes = [C.c_void_p, C.c_double]
        check(f(self.handle, C.c_double(confidence_level)))

    def summary(self, i):
        """(mean[K], ci_low[K], ci_high[K]) of event i after summarize()."""
        K = C.c_int()
        check(lib().miso_batch_event_info(self.handle, i, C.byref(K), None, None, None))
        m, lo, hi = (np.zeros(K.value) for _ in range(3))
        check(lib().miso_batch_get_summary(self.handle, i, _p(m), _p(lo), _p(hi)))
        return m, lo, hi

    def summaries(self, indices, noiso):
        """[(mean[K], ci_low[K], ci_high[K])] of the given events after summarize() -- summary() for many events without its
        per-event allocations (noiso[j] = isoforms of event indices[j], known to the caller from the annotation)."""
        offs = np.concatenate([[0], np.cumsum(np.asarray(noiso, dtype=np.int64))])
        buf = np.zeros((3, int(offs[-1])))
        base = [buf[r].ctypes.data for r in range(3)]
        f = lib().miso_batch_get_summary
        h = self.handle
        for j, i in enumerate(indices):
            o = 8 * int(offs[j])
            check(f(h, int(i), C.c_void_p(base[0] + o), C.c_void_p(base[1] + o), C.c_void_p(base[2] + o)))
        return [(buf[0, offs[j]:offs[j + 1]], buf[1, offs[j]:offs[j + 1]], buf[2, offs[j]:offs[j + 1]]) for j in range(len(indices))]

    def compare(self, other, smoothing=0.3):
        """Two-sample comparison with `other` (same events, same order) on the device."""
        check(lib().miso_batch_compare(self.handle, other.handle, C.c_double(smoothing)))

    def comparison(self, i):
        """(mean1[K], mean2[K], bayes_factor[K], density_at_0[K]) of event i after compare()."""
        K = C.c_int()
        check(lib().miso_batch_event_info(self.handle, i, C.byref(K), None, None, None))
        m1, m2, bf, dens = (np.zeros(K.value) for _ in range(4))
        check(lib().miso_batch_get_comparison(self.handle, i, _p(m1), _p(m2), _p(bf), _p(dens)))
        return m1, m2, bf, dens

    def add_event_aln(self, gene, alnfile, chrom, start, end, strand_rule=0, target_strand=None,
                      read_len=None, min_reads=0, hyper=None):
        """One event straight from an open alignment file (sam_utils.Samfile) -- fetch, pairing and
        filters run natively (miso_batch_add_event_aln).  Returns (event index or -1, reads found)."""
        hy = None if hyper is None else np.asarray(hyper, dtype=np.float64)
        tid = alnfile.gettid(chrom)
        if tid < 0:
            return -1, 0
        idx, n = C.c_int(-1), C.c_int64(0)
        L = lib()
        L.miso_batch_add_event_aln.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                               C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64,
                                               C.c_void_p, C.c_int, C.POINTER(C.c_int64),
                                               C.POINTER(C.c_int)]
        check(L.miso_batch_add_event_aln(self.handle, gene.handle, alnfile._h, tid, int(start),
                                         int(end), int(strand_rule),
                                         ord(target_strand[0]) if target_strand else 0,
                                         int(read_len) if read_len else 0, int(min_reads), _p(hy),
                                         0 if hy is None else len(hy), C.byref(n), C.byref(idx)))
        return idx.value, n.value

    def add_events_aln(self, genes, alnfile, chroms, starts, ends, strand_rule=0, target_strands=None,
                       read_len=None, min_reads=0, threads=0):
        """Many events from one alignment file at once (miso_batch_add_events_aln: reads collected
        and CIGARs parsed on host threads).  Returns (event indices (-1 = not added), reads found)."""
        n = len(genes)
        if n == 0:
            return np.zeros(0, np.int32), np.zeros(0, np.int64)
        tids = np.asarray([alnfile.gettid(c) for c in chroms], dtype=np.int32)
        st = np.asarray(starts, dtype=np.int64)
        en = np.asarray(ends, dtype=np.int64)
        ts = np.asarray([ord(t[0]) if t else 0 for t in (target_strands or [None] * n)], dtype=np.int32)
        gh = (C.c_void_p * n)(*[g.handle for g in genes])
        idx = np.full(n, -1, np.int32)
        cnt = np.zeros(n, np.int64)
        L = lib()
        L.miso_batch_add_events_aln.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        check(L.miso_batch_add_events_aln(self.handle, n, gh, alnfile._h, _p(tids), _p(st), _p(en),
                                          int(strand_rule), _p(ts), int(read_len) if read_len else 0,
                                          int(min_reads), int(threads), _p(cnt), _p(idx)))
        return idx, cnt

    def result_lite(self, i):
        """(class_templates, class_counts, assignment, rundata) of event i -- no sample copy."""
        K, N, ncls = C.c_int(), C.c_int(), C.c_int()
        check(lib().miso_batch_event_info(self.handle, i, C.byref(K), C.byref(N), None, C.byref(ncls)))
        ct = np.zeros((max(ncls.value, 1), K.value))
        cc = np.zeros(max(ncls.value, 1))
        ass = np.zeros(max(N.value, 1), np.int32)
        rd = RunData()
        check(lib().miso_batch_get_result(self.handle, i, None, None, _p(ct), _p(cc), _p(ass),
                                          C.byref(rd)))
        return ct[:ncls.value], cc[:ncls.value], ass[:N.value], rd

    def header_fields(self, indices):
        """[(all_unassigned, percent_accept, counts, assigned_counts)] -- the run-dependent header fields of the given
        events as text, formatted natively (include/miso_amd.h miso_batch_header_fields)."""
        n = len(indices)
        if n == 0:
            return []
        idx = np.asarray(indices, dtype=np.int32)
        need = C.c_int64(0)
        cap = 96 * n + 4096
        for _ in range(2):
            buf = C.create_string_buffer(cap)
            check(lib().miso_batch_header_fields(self.handle, n, _p(idx), buf, C.c_int64(cap), C.byref(need)))
            if need.value <= cap:
                break
            cap = need.value
        out = []
        for ln in buf.raw[:need.value - 1].decode().split("\n")[:n]:
            u, pa, counts, assigned = ln.split("\t")
            out.append((u == "1", pa, counts, assigned))
        return out

    def write_miso_files(self, indices, paths, headers, threads=0):
        """The `.miso` files of the given events, rows formatted and written natively."""
        n = len(indices)
        if n == 0:
            return
        idx = np.asarray(indices, dtype=np.int32)
        pa = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
        hd = (C.c_char_p * n)(*[h.encode() for h in headers])
        check(lib().miso_batch_write_miso_files(self.handle, n, _p(idx), pa, hd, int(threads)))

    def match_ms(self):
        ms = C.c_float(0)
        check(lib().miso_batch_last_match_ms(self.handle, C.byref(ms)))
        return ms.value

    def device_match_of(self, i):
        """(match [N, K], fraglen [N, K] or None) as the GPU computed them (device_match +
        counts_trace batches, after upload)."""
        K, N = C.c_int(), C.c_int()
        check(lib().miso_batch_event_info(self.handle, i, C.byref(K), C.byref(N), None, None))
        m = np.zeros((N.value, K.value))
        fl = np.zeros((N.value, K.value), np.int32) if self.params.paired else None
        check(lib().miso_batch_get_match(self.handle, i, _p(m), _p(fl)))
        return m, fl

    def last_kernels(self):
        buf = C.create_string_buffer(2048)
        check(lib().miso_batch_last_kernels(self.handle, buf, 2048))
        return buf.value.decode()

    def rounds(self):
        """stop=CONVERGENT_MEAN: rounds the last launch took (include/miso_amd.h miso_batch_rounds)"""
        n = C.c_int(0)
        check(lib().miso_batch_rounds(self.handle, C.byref(n)))
        return n.value

    def set_clock_probe(self, on=True):
        """Measure the shader clock of every following launch (include/miso_amd.h miso_batch_set_clock_probe)."""
        check(lib().miso_batch_set_clock_probe(self.handle, int(bool(on))))

    def last_clock(self):
        """(shader clock in GHz, the probe's window in ms) of the last synced launch; (0.0, 0.0) without a valid probe."""
        ghz, ms = C.c_double(0.0), C.c_double(0.0)
        check(lib().miso_batch_last_clock(self.handle, C.byref(ghz), C.byref(ms)))
        return ghz.value, ms.value

    def coop_retries(self):
        """Launches sync() repeated with one workgroup per chain after a chain on several workgroups timed out."""
        n = C.c_int(0)
        check(lib().miso_batch_coop_retries(self.handle, C.byref(n)))
        return n.value

    def launch_stats(self):
        """{"kernels": [{name, waves, trips, iterations, chains, words}], "uniforms": Philox words the
        read loops consume per launch} of the last launch (miso_batch_launch_stats)."""
        n = C.c_int(0)
        cap = 16
        while True:   # (one record per run: a whole-gene batch with size buckets has two dozen; the call reports how many there are)
            arr = (KernelStat * cap)()
            check(lib().miso_batch_launch_stats(self.handle, arr, cap, C.byref(n)))
            if n.value <= cap:
                break
            cap = n.value
        ks = [{"name": arr[i].name.decode(), "waves": arr[i].waves, "trips": arr[i].trips,
               "iterations": arr[i].iterations, "chains": arr[i].chains, "words": arr[i].words}
              for i in range(n.value)]
        return {"kernels": ks, "uniforms": sum(k["words"] * k["iterations"] for k in ks)}

    def placement(self, i):
        """HW_REG_HW_ID of the wavefront that ran each chain of event i (after download)."""
        out = np.zeros(self.params.noChains, np.uint32)
        check(lib().miso_batch_get_placement(self.handle, i, _p(out)))
        return out

    def algorithmic_bytes(self):
        b = C.c_double(0)
        check(lib().miso_batch_algorithmic_bytes(self.handle, C.byref(b)))
        return b.value

    def classes(self, i):
        """(templates [ncls, K], counts [ncls]) of event i: available before any GPU work."""
        K, ncls = C.c_int(), C.c_int()
        check(lib().miso_batch_event_info(self.handle, i, C.byref(K), None, None, C.byref(ncls)))
        ct = np.zeros((max(ncls.value, 1), K.value))
        cc = np.zeros(max(ncls.value, 1))
        check(lib().miso_batch_get_result(self.handle, i, None, None, _p(ct), _p(cc), None, None))
        return ct[:ncls.value], cc[:ncls.value]

    def result(self, i, trace=False):
        K, N, S, ncls = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().miso_batch_event_info(self.handle, i, C.byref(K), C.byref(N), C.byref(S),
                                          C.byref(ncls)))
        K, N, S, ncls = K.value, N.value, S.value, ncls.value
        samples = np.zeros((max(S, 1), K))
        ll = np.zeros(max(S, 1))
        ct = np.zeros((max(ncls, 1), K))
        cc = np.zeros(max(ncls, 1))
        ass = np.zeros(max(N, 1), np.int32)
        rd = RunData()
        check(lib().miso_batch_get_result(self.handle, i, _p(samples), _p(ll), _p(ct), _p(cc),
                                          _p(ass), C.byref(rd)))
        Cn, M = self.params.noChains, self.params.noIterations
        h = np.zeros(Cn, np.uint64)
        tr = np.zeros((M + 1, Cn, K), np.int32) if trace else None
        check(lib().miso_batch_get_trace(self.handle, i, _p(h), _p(tr)))
        return EventResult(samples[:S], ll[:S], ct[:ncls], cc[:ncls], ass[:N], rd, h, tr)


class SamplesBatch(Batch):
    """Posterior samples produced elsewhere (parsed `.miso` files) on the device: summarize(), summary(i),
    compare(other), comparison(i) as on a sampled batch (miso_batch_from_samples)."""

    def __init__(self, samples, device=0):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in samples]     # each [S, K]
        n = len(arrs)
        S = arrs[0].shape[0] if n else 1
        if any(a.ndim != 2 or a.shape[0] != S for a in arrs):
            raise ValueError("every event needs the same number of samples")
        K = np.asarray([a.shape[1] for a in arrs], dtype=np.int32)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
        self.params = Params(0, 36, 1, 1, S, S, 0, 1, MISO_ALGO_REASSIGN, MISO_START_AUTO, MISO_STOP_FIXEDNO, 0.0, 0.0, 4.0, 0, 0)
        self.handle = C.c_void_p()
        L = lib()
        L.miso_batch_from_samples.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        check(L.miso_batch_from_samples(n, _p(K), int(S), ptrs, int(device), C.byref(self.handle)))


def simulate_reads(gene, expression, n_reads, read_len, sim_seed, mean=0.0, var=0.0, num_devs=4.0):
    """pysplicing.simulateReads / simulatePairedReads (var > 0): returns (isoform, pos, cigars)."""
    ex = np.asarray(expression, dtype=np.float64)
    n = n_reads * (2 if var > 0 else 1)
    stride = 64
    iso = np.zeros(max(n, 1), np.int32)
    pos = np.zeros(max(n, 1), np.int32)
    buf = C.create_string_buffer(max(n, 1) * stride)
    check(lib().miso_simulate_reads(gene.handle, _p(ex), int(n_reads), int(read_len),
                                    C.c_double(mean), C.c_double(var), C.c_double(num_devs),
                                    C.c_uint64(sim_seed), _p(iso), _p(pos), buf, stride))
    raw = buf.raw
    cig = [raw[i * stride:(i + 1) * stride].split(b"\0")[0] for i in range(n)]
    return iso[:n], pos[:n], cig


def selftest_detmath(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    outs = [np.zeros_like(x) for _ in range(4)]
    check(lib().miso_selftest_detmath(_p(x), len(x), *[_p(o) for o in outs]))
    return outs


def selftest_convergent_mean(samples, chains):
    """samples: S x K (row i from chain i % chains) -> True if stop=CONVERGENT_MEAN would stop (miso.c:556-636)"""
    a = np.ascontiguousarray(samples, dtype=np.float64)
    stop = C.c_int(-1)
    check(lib().miso_selftest_convergent_mean(_p(a), a.shape[1], int(chains), a.shape[0], C.byref(stop)))
    return bool(stop.value)


def selftest_philox(ctr_key6):
    a = np.ascontiguousarray(ctr_key6, dtype=np.uint32).reshape(-1, 6)
    out = np.zeros((len(a), 4), np.uint32)
    check(lib().miso_selftest_philox(_p(a), len(a), _p(out)))
    return out
