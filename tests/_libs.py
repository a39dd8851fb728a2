"""ctypes front-ends for the two CPU checkers used by the tests.

* ``OrcLib``  -- oracle/libmiso_oracle.so, our C restatement (oracle/miso_oracle.c)
* ``RefLib``  -- oracle/_ref/libmiso_ref.so, the REAL reference C core behind oracle/ref_shim.c
                 (only present where `make -C oracle ref` ran, i.e. the authoring container and
                 any box the built .so travelled to)

Both expose the same Python-level calls so a test can run one problem through either.
Test infrastructure only: nothing under miso_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_PATH = os.path.join(ROOT, "oracle", "libmiso_oracle.so")
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libmiso_ref.so")

CIG_STRIDE = 64
_vp = C.c_void_p


def _p(a):
    return a.ctypes.data_as(_vp) if a is not None else None


def _cigs(cigars):
    arr = (C.c_char_p * max(len(cigars), 1))()
    for i, c in enumerate(cigars):
        arr[i] = c if isinstance(c, bytes) else c.encode()
    return arr


def flatten_isoforms(isoforms):
    out = []
    for iso in isoforms:
        out.extend(int(e) for e in iso)
        out.append(-1)
    return np.asarray(out, dtype=np.int32)


def ensure_oracle_built():
    src = os.path.join(ROOT, "oracle", "miso_oracle.c")
    deps = [src, os.path.join(ROOT, "oracle", "miso_oracle.h"),
            os.path.join(ROOT, "include", "miso_detmath.h"),
            os.path.join(ROOT, "include", "miso_philox.h")]
    if (not os.path.exists(ORC_PATH)) or any(
            os.path.getmtime(d) > os.path.getmtime(ORC_PATH) for d in deps):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return ORC_PATH


class MisoResult:
    """Outputs of one sampler call, reference shapes (samples K x S col-major -> [S, K])."""

    def __init__(self, rc, K, S, samples, loglik, match, templates, counts, assignment, rundata,
                 trace=None):
        self.rc = rc
        self.samples = samples.reshape(S, K) if rc == 0 else None
        self.loglik = loglik
        self.match = match
        self.class_templates = templates
        self.class_counts = counts
        self.assignment = assignment
        self.rundata = rundata
        self.accepted = int(rundata[5]) if rc == 0 else None
        self.rejected = int(rundata[6]) if rc == 0 else None
        self.trace = trace or {}


class _Base:
    prefix = ""
    has_opts = False

    def __init__(self, path):
        self.lib = C.CDLL(path)
        L, p = self.lib, self.prefix
        getattr(L, p + "gene_create").restype = _vp
        getattr(L, p + "unif01").restype = C.c_double
        getattr(L, p + "normal01").restype = C.c_double
        getattr(L, p + "integer").restype = C.c_long
        getattr(L, p + "integer").argtypes = [C.c_long, C.c_long]

    def f(self, name):
        return getattr(self.lib, self.prefix + name)

    # --- rng ---
    def rng_seed(self, seed):
        self.f("rng_seed")(C.c_ulong(seed))

    def unif01(self):
        return self.f("unif01")()

    def normal01(self):
        return self.f("normal01")()

    def integer(self, lo, hi):
        return self.f("integer")(lo, hi)

    def assignment_matrix(self, g, read_len, overhang=1, max_cols=4096):
        """splicing_assignment_matrix (assignment.c:90-276): ncls x K, one class per row; None on an error code"""
        K = self.noiso(g)
        out = np.zeros(K * max_cols)
        n = self.f("assignment_matrix")(g, int(read_len), int(overhang), _p(out), max_cols)
        return None if n < 0 else out[:K * n].reshape(n, K).copy()

    def score_classes(self, psi, hyper, amat, matches):
        """the joint score of algorithm=CLASSES (miso.c:284-295 + the Dirichlet prior) for one psi; amat: ncls x K"""
        psi, hyper = np.ascontiguousarray(psi, np.float64), np.ascontiguousarray(hyper, np.float64)
        amat, matches = np.ascontiguousarray(amat, np.float64), np.ascontiguousarray(matches, np.float64)
        f = self.f("score_classes")
        f.restype = C.c_double
        return float(f(len(psi), _p(psi), _p(hyper), _p(amat), len(matches), _p(matches)))

    def convergent_mean(self, samples, chains):
        """stop=CONVERGENT_MEAN's test (miso.c:556-636): samples S x K, row i from chain i % chains -> 1 stop / 0"""
        a = np.ascontiguousarray(samples, dtype=np.float64)
        return int(self.f("convergent_mean")(_p(a), a.shape[1], int(chains), a.shape[0]))

    # --- gene ---
    def gene(self, exons, isoforms):
        ex = np.asarray(exons, dtype=np.int32).reshape(-1)
        iso = flatten_isoforms(isoforms)
        h = self.f("gene_create")(_p(ex), len(ex) // 2, _p(iso), len(iso))
        if not h:
            raise RuntimeError("gene_create failed")
        return _vp(h)

    def gene_free(self, g):
        self.f("gene_destroy")(g)

    def noiso(self, g):
        return self.f("gene_noiso")(g)

    def isolength(self, g):
        out = np.zeros(64, np.int32)
        n = self.f("gene_isolength")(g, _p(out))
        return out[:n].copy()

    # --- problem construction ---
    def match_iso(self, g, pos, cigars, read_len, overhang=1):
        K, N = self.noiso(g), len(pos)
        pos = np.asarray(pos, dtype=np.int32)
        m = np.zeros(K * max(N, 1))
        rc = self.f("match_iso")(g, _p(pos), _cigs(cigars), N, overhang, read_len, _p(m))
        return rc, m[:K * N].reshape(N, K)

    def match_iso_paired(self, g, pos, cigars, read_len, mean, var, num_devs=4.0, overhang=1):
        K, N = self.noiso(g), len(pos) // 2
        pos = np.asarray(pos, dtype=np.int32)
        m = np.zeros(K * max(N, 1))
        fl = np.zeros(K * max(N, 1), np.int32)
        rc = self.f("match_iso_paired")(g, _p(pos), _cigs(cigars), len(pos), read_len, overhang,
                                        C.c_double(mean), C.c_double(var), C.c_double(num_devs),
                                        _p(m), _p(fl))
        return rc, m[:K * N].reshape(N, K), fl[:K * N].reshape(N, K)

    # --- simulators ---
    def _unpack_cigars(self, buf, n):
        raw = buf.raw
        return [raw[i * CIG_STRIDE:(i + 1) * CIG_STRIDE].split(b"\0")[0] for i in range(n)]

    def _sim_args(self, g, expr):
        return ()

    def simulate_reads(self, g, expr, n, read_len):
        expr = np.asarray(expr, dtype=np.float64)
        iso = np.zeros(n, np.int32)
        pos = np.zeros(n, np.int32)
        buf = C.create_string_buffer(n * CIG_STRIDE)
        args = [g, _p(expr)] + ([len(expr)] if self.prefix == "ref_" else []) + \
               [n, read_len, _p(iso), _p(pos), buf, CIG_STRIDE]
        rc = self.f("simulate_reads")(*args)
        return rc, iso, pos, self._unpack_cigars(buf, n)

    def simulate_paired_reads(self, g, expr, npairs, read_len, mean, var, num_devs=4.0):
        expr = np.asarray(expr, dtype=np.float64)
        n = 2 * npairs
        iso = np.zeros(n, np.int32)
        pos = np.zeros(n, np.int32)
        buf = C.create_string_buffer(n * CIG_STRIDE)
        args = [g, _p(expr)] + ([len(expr)] if self.prefix == "ref_" else []) + \
               [npairs, read_len, C.c_double(mean), C.c_double(var), C.c_double(num_devs),
                _p(iso), _p(pos), buf, CIG_STRIDE]
        rc = self.f("simulate_paired_reads")(*args)
        return rc, iso, pos, self._unpack_cigars(buf, n)


class RefLib(_Base):
    prefix = "ref_"

    def __init__(self, path=REF_PATH):
        super().__init__(path)
        self.lib.ref_dnorm.restype = C.c_double

    @staticmethod
    def available():
        return os.path.exists(REF_PATH)

    def miso(self, g, pos, cigars, read_len, iters=5000, burn=500, lag=10, hyper=None,
             overhang=1, chains=6, start=0, stop=0, algo=0, max_iters=100000):
        K, N = self.noiso(g), len(pos)
        hyper = np.ones(K) if hyper is None else np.asarray(hyper, dtype=np.float64)
        S = max(chains * (iters - burn) // max(lag, 1), 0)
        pos = np.asarray(pos, dtype=np.int32)
        samples, ll = np.zeros(K * max(S, 1)), np.zeros(max(S, 1))
        match, ct, cc = np.zeros(K * max(N, 1)), np.zeros(K * max(N, 1)), np.zeros(max(N, 1))
        ncls, ass, rd = C.c_int(0), np.zeros(max(N, 1), np.int32), np.zeros(9, np.int32)
        rc = self.lib.ref_miso(g, _p(pos), _cigs(cigars), N, read_len, overhang, chains, iters,
                               max_iters, burn, lag, _p(hyper), len(hyper), algo, start, stop,
                               _p(samples), _p(ll), _p(match), _p(ct), _p(cc), C.byref(ncls),
                               _p(ass), _p(rd))
        n = ncls.value
        return MisoResult(rc, K, S, samples[:K * S], ll[:S], match[:K * N].reshape(N, K),
                          ct[:K * n].reshape(n, K), cc[:n], ass[:N], rd)

    def miso_paired(self, g, pos, cigars, read_len, mean, var, num_devs=4.0, iters=5000,
                    burn=500, lag=10, hyper=None, overhang=1, chains=6, start=0, stop=0,
                    max_iters=100000):
        K, N = self.noiso(g), len(pos) // 2
        hyper = np.ones(K) if hyper is None else np.asarray(hyper, dtype=np.float64)
        S = max(chains * (iters - burn) // max(lag, 1), 0)
        pos = np.asarray(pos, dtype=np.int32)
        samples, ll = np.zeros(K * max(S, 1)), np.zeros(max(S, 1))
        match, ct, cc = np.zeros(K * max(N, 1)), np.zeros(K * max(N, 1)), np.zeros(max(N, 1))
        ncls, ass, rd = C.c_int(0), np.zeros(max(N, 1), np.int32), np.zeros(9, np.int32)
        rc = self.lib.ref_miso_paired(g, _p(pos), _cigs(cigars), len(pos), read_len, overhang,
                                      chains, iters, max_iters, burn, lag, _p(hyper), len(hyper),
                                      start, stop, C.c_double(mean), C.c_double(var),
                                      C.c_double(num_devs), _p(samples), _p(ll), _p(match),
                                      _p(ct), _p(cc), C.byref(ncls), _p(ass), _p(rd))
        n = ncls.value
        return MisoResult(rc, K, S, samples[:K * S], ll[:S], match[:K * N].reshape(N, K),
                          ct[:K * n].reshape(n, K), cc[:n], ass[:N], rd)


class OrcOpts(C.Structure):
    _fields_ = [("mode", C.c_int), ("seed", C.c_uint64), ("event_id", C.c_uint32),
                ("per_read_sums", C.c_int)]


class OrcTrace(C.Structure):
    _fields_ = [("counts_trace", _vp), ("counts_hash", _vp), ("final_psi", _vp),
                ("accepted", _vp)]


class OrcLib(_Base):
    prefix = "orc_"
    STREAM, COUNTER, COLLAPSED = 0, 1, 2

    def __init__(self, path=None):
        super().__init__(path or ensure_oracle_built())
        L = self.lib
        for name in ("orc_qnorm_libm", "orc_qnorm_det", "orc_det_exp", "orc_det_log",
                     "orc_det_sqrt"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_double]

    def philox(self, ctr, key, rounds=None):
        """One Philox4x32 block at the contract's rounds (include/miso_philox.h) or at `rounds`."""
        out = np.zeros(4, np.uint32)
        a = (C.c_uint32(ctr[0]), C.c_uint32(ctr[1]), C.c_uint32(ctr[2]), C.c_uint32(ctr[3]), C.c_uint32(key[0]),
             C.c_uint32(key[1]), _p(out))
        if rounds is None:
            self.lib.orc_philox(*a)
        else:
            self.lib.orc_philox_r(C.c_int(rounds), *a)
        return out

    def split_word(self, seed, event_id, chain, iteration, r):
        """the 32-bit uniform word of the r-th drawing read of a single-end two-isoform event (miso_philox.h, lazy low bits)"""
        self.lib.orc_split_word.restype = C.c_uint32
        return int(self.lib.orc_split_word(C.c_uint64(seed), C.c_uint32(event_id), C.c_uint32(chain), C.c_uint32(iteration), C.c_uint32(r)))

    def philox_rounds(self):
        return self.lib.orc_philox_rounds()

    def contract_version(self):
        """include/miso_philox.h MISO_CONTRACT_VERSION as this checker was compiled (0: a checker older than the constant)"""
        return int(self.lib.orc_contract_version()) if hasattr(self.lib, "orc_contract_version") else 0

    def binomial(self, n, p, count, seed=1, event_id=0):
        """`count` draws of include/miso_binomial.h's Binomial(n, p) (word streams of iterations 0 .. count-1)."""
        out = np.zeros(count, np.int32)
        self.lib.orc_binomial(C.c_uint64(seed), C.c_uint32(event_id), C.c_int32(n), C.c_double(p), C.c_int(count), _p(out))
        return out

    def _opts(self, mode, seed, event_id, per_read_sums):
        return OrcOpts(mode, seed, event_id, int(per_read_sums))

    def _trace(self, want, M, Cn, K):
        if not want:
            return None, None
        bufs = {"counts_trace": np.zeros((M + 1) * Cn * K, np.int32),
                "counts_hash": np.zeros(Cn, np.uint64),
                "final_psi": np.zeros(K * Cn), "accepted": np.zeros(Cn, np.int32)}
        t = OrcTrace(_p(bufs["counts_trace"]), _p(bufs["counts_hash"]), _p(bufs["final_psi"]),
                     _p(bufs["accepted"]))
        return t, bufs

    def miso(self, g, pos, cigars, read_len, iters=5000, burn=500, lag=10, hyper=None,
             overhang=1, chains=6, start=0, stop=0, algo=0, max_iters=100000, mode=0, seed=0,
             event_id=0, per_read_sums=False, trace=False):
        K, N = self.noiso(g), len(pos)
        hyper = np.ones(K) if hyper is None else np.asarray(hyper, dtype=np.float64)
        S = max(chains * (iters - burn) // max(lag, 1), 0)
        pos = np.asarray(pos, dtype=np.int32)
        samples, ll = np.zeros(K * max(S, 1)), np.zeros(max(S, 1))
        match, ct, cc = np.zeros(K * max(N, 1)), np.zeros(K * max(N, 1)), np.zeros(max(N, 1))
        ncls, ass, rd = C.c_int(0), np.zeros(max(N, 1), np.int32), np.zeros(9, np.int32)
        opts = self._opts(mode, seed, event_id, per_read_sums)
        t, bufs = self._trace(trace, iters, chains, K)
        rc = self.lib.orc_miso(g, _p(pos), _cigs(cigars), N, read_len, overhang, chains, iters,
                               max_iters, burn, lag, _p(hyper), len(hyper), algo, start, stop,
                               C.byref(opts), _p(samples), _p(ll), _p(match), _p(ct), _p(cc),
                               C.byref(ncls), _p(ass), _p(rd), C.byref(t) if t else None)
        n = ncls.value
        if bufs:
            bufs["counts_trace"] = bufs["counts_trace"].reshape(iters + 1, chains, K)
            bufs["final_psi"] = bufs["final_psi"].reshape(chains, K)
        return MisoResult(rc, K, S, samples[:K * S], ll[:S], match[:K * N].reshape(N, K),
                          ct[:K * n].reshape(n, K), cc[:n], ass[:N], rd, bufs)

    def miso_paired(self, g, pos, cigars, read_len, mean, var, num_devs=4.0, iters=5000,
                    burn=500, lag=10, hyper=None, overhang=1, chains=6, start=0, stop=0,
                    max_iters=100000, mode=0, seed=0, event_id=0, per_read_sums=False,
                    trace=False):
        K, N = self.noiso(g), len(pos) // 2
        hyper = np.ones(K) if hyper is None else np.asarray(hyper, dtype=np.float64)
        S = max(chains * (iters - burn) // max(lag, 1), 0)
        pos = np.asarray(pos, dtype=np.int32)
        samples, ll = np.zeros(K * max(S, 1)), np.zeros(max(S, 1))
        match, ct, cc = np.zeros(K * max(N, 1)), np.zeros(K * max(N, 1)), np.zeros(max(N, 1))
        ncls, ass, rd = C.c_int(0), np.zeros(max(N, 1), np.int32), np.zeros(9, np.int32)
        opts = self._opts(mode, seed, event_id, per_read_sums)
        t, bufs = self._trace(trace, iters, chains, K)
        rc = self.lib.orc_miso_paired(g, _p(pos), _cigs(cigars), len(pos), read_len, overhang,
                                      chains, iters, max_iters, burn, lag, _p(hyper), len(hyper),
                                      start, stop, C.c_double(mean), C.c_double(var),
                                      C.c_double(num_devs), C.byref(opts), _p(samples), _p(ll),
                                      _p(match), _p(ct), _p(cc), C.byref(ncls), _p(ass), _p(rd),
                                      C.byref(t) if t else None)
        n = ncls.value
        if bufs:
            bufs["counts_trace"] = bufs["counts_trace"].reshape(iters + 1, chains, K)
            bufs["final_psi"] = bufs["final_psi"].reshape(chains, K)
        return MisoResult(rc, K, S, samples[:K * S], ll[:S], match[:K * N].reshape(N, K),
                          ct[:K * n].reshape(n, K), cc[:n], ass[:N], rd, bufs)
