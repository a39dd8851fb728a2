"""Python-3 mirror of misopy/sam_utils.py for the path in front of the sampler.

The reference goes through pysam (third-party, absent here).  `Samfile` below is backed by the
native reader of libmiso_amd.so (include/miso_alnio.h, csrc/alnio.cpp: BAM/BGZF inflated in
parallel or SAM text, indexed in memory) and offers what the reference uses of pysam:
`.references`, `.fetch(chrom, start, end)` and reads with `qname, flag, pos, cigar, rlen,
is_paired, is_read1, is_read2, is_reverse, is_unmapped, mate_is_unmapped, is_qcfail`.

Two ways to the sampler's inputs:
  * `sam_parse_reads(reads, ...)` -- the reference's function, line by line, over read objects
    (sam_utils.py:363-442 with pair_sam_reads :207-300 and read_matches_strand :320-360);
  * `Samfile.parse_reads(chrom, start, end, ...)` -- the same rules evaluated natively for one
    event in one call (miso_aln_parse_reads), which is what the batch runner uses.
`tests/test_frontend.py` checks that the two agree.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("MISO_AMD_LIB", os.path.join(_HERE, "libmiso_amd.so"))

STRAND_RULES = {None: 0, "fr-unstranded": 0, "fr-firststrand": 1}

# Global variable containing CIGAR types for conversion (sam_utils.py:303); '=' and 'X' (BAM ops
# 7, 8) are an IndexError in the reference, here they keep their SAM letters, which the sampler's
# CIGAR parser accepts (solve.c:249-299)
CIGAR_TYPES = ('M', 'I', 'D', 'N', 'S', 'H', 'P', '=', 'X')


class _Columns(C.Structure):
    _fields_ = [("n", C.c_int64), ("ref_id", C.POINTER(C.c_int32)), ("pos", C.POINTER(C.c_int32)),
                ("end", C.POINTER(C.c_int32)), ("flag", C.POINTER(C.c_int32)),
                ("l_seq", C.POINTER(C.c_int32)), ("cigar_off", C.POINTER(C.c_uint64)),
                ("cigar", C.POINTER(C.c_uint32)), ("name_off", C.POINTER(C.c_uint64)),
                ("names", C.POINTER(C.c_char))]


_lib = None
_READER_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmiso_aln.so")
_reader_only = False
_PRELOADED = {}   # absolute path -> Samfile decoded by the dispatcher before it forked this worker


def use_reader_library():
    """Load the alignment reader from libmiso_aln.so (the same code without the HIP kernels): for a
    process that must not initialise the GPU runtime -- `miso --run`'s dispatcher decodes the
    alignments ONCE and then forks one worker per GPU; the workers inherit the decoded file."""
    global _reader_only, _lib
    if _lib is None:
        _reader_only = True


def _native():
    global _lib
    if _lib is None:
        path = _READER_LIB_PATH if _reader_only else _LIB_PATH
        if not os.path.exists(path):
            raise ImportError("miso_amd: %s is missing -- build it with "
                              "`python __graft_entry__.py`" % path)
        L = C.CDLL(path)
        L.miso_aln_last_error.restype = C.c_char_p
        L.miso_aln_ref_name.restype = C.c_char_p
        L.miso_aln_ref_name.argtypes = [C.c_void_p, C.c_int]
        L.miso_aln_ref_length.restype = C.c_int64
        L.miso_aln_ref_length.argtypes = [C.c_void_p, C.c_int]
        L.miso_aln_ref_id.argtypes = [C.c_void_p, C.c_char_p]
        L.miso_aln_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.miso_aln_close.argtypes = [C.c_void_p]
        L.miso_aln_columns.argtypes = [C.c_void_p, C.POINTER(_Columns)]
        L.miso_aln_n_refs.argtypes = [C.c_void_p]
        L.miso_aln_is_bam.argtypes = [C.c_void_p]
        L.miso_aln_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                     C.c_int64, C.POINTER(C.c_int64)]
        L.miso_aln_parse_reads.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                           C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        msg = _native().miso_aln_last_error().decode(errors="replace")
        if rc == 2:
            raise MemoryError(msg)
        raise IOError(msg)


class AlignedRead(object):
    """What the reference reads of a pysam.AlignedRead."""
    __slots__ = ("qname", "flag", "pos", "cigar", "rlen", "aend", "tid")

    def __init__(self, qname, flag, pos, cigar, rlen, aend=None, tid=-1):
        self.qname, self.flag, self.pos, self.cigar, self.rlen = qname, flag, pos, cigar, rlen
        self.aend, self.tid = aend, tid

    is_paired = property(lambda self: bool(self.flag & 0x1))
    is_unmapped = property(lambda self: bool(self.flag & 0x4))
    mate_is_unmapped = property(lambda self: bool(self.flag & 0x8))
    is_reverse = property(lambda self: bool(self.flag & 0x10))
    is_read1 = property(lambda self: bool(self.flag & 0x40))
    is_read2 = property(lambda self: bool(self.flag & 0x80))
    is_qcfail = property(lambda self: bool(self.flag & 0x200))

    def __repr__(self):
        return "AlignedRead(%s, flag=%d, pos=%d, cigar=%s)" % (self.qname, self.flag, self.pos,
                                                             sam_cigar_to_str(self.cigar))


class Samfile(object):
    """pysam.Samfile as far as misopy uses it (sam_utils.py:139-186), over the native reader."""

    def __init__(self, filename, mode="rb", template=None, threads=0):
        self.filename = filename
        self._h = C.c_void_p()
        _check(_native().miso_aln_open(os.fsencode(filename), int(threads), C.byref(self._h)))
        L = _native()
        self.references = tuple(L.miso_aln_ref_name(self._h, i).decode()
                                for i in range(L.miso_aln_n_refs(self._h)))
        self.lengths = tuple(L.miso_aln_ref_length(self._h, i) for i in range(len(self.references)))
        self._ref_index = {name: i for i, name in enumerate(self.references)}
        cols = _Columns()
        _check(L.miso_aln_columns(self._h, C.byref(cols)))
        n = self.mapped_plus_unmapped = int(cols.n)

        def view(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(ptr, shape=(count,)).view(dtype)
        self.ref_id = view(cols.ref_id, n, np.int32)
        self.pos = view(cols.pos, n, np.int32)
        self.end = view(cols.end, n, np.int32)
        self.flag = view(cols.flag, n, np.int32)
        self.l_seq = view(cols.l_seq, n, np.int32)
        self.cigar_off = view(cols.cigar_off, n + 1, np.uint64)
        self.name_off = view(cols.name_off, n + 1, np.uint64)
        self.cigar = view(cols.cigar, int(self.cigar_off[-1]), np.uint32)
        self.names = C.string_at(cols.names, int(self.name_off[-1])) if n else b""
        self.is_bam = bool(L.miso_aln_is_bam(self._h))

    def close(self):
        if self._h:
            _native().miso_aln_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.mapped_plus_unmapped

    def read(self, i):
        """Record i (file order) as an AlignedRead."""
        c0, c1 = int(self.cigar_off[i]), int(self.cigar_off[i + 1])
        cigar = None if c1 == c0 else [(int(v) & 15, int(v) >> 4) for v in self.cigar[c0:c1]]
        name = self.names[int(self.name_off[i]):int(self.name_off[i + 1])].decode()
        return AlignedRead(name, int(self.flag[i]), int(self.pos[i]), cigar, int(self.l_seq[i]),
                           int(self.end[i]), int(self.ref_id[i]))

    def __iter__(self):
        return (self.read(i) for i in range(len(self)))

    def gettid(self, chrom):
        return self._ref_index.get(chrom, -1)

    def fetch_indices(self, chrom, start, end):
        tid = self.gettid(chrom)
        if tid < 0:
            raise ValueError("invalid reference `%s`" % chrom)   # what pysam raises
        if start > end:
            raise ValueError("invalid region: start (%d) > end (%d)" % (start, end))
        n = C.c_int64()
        L = _native()
        _check(L.miso_aln_fetch(self._h, tid, start, end, None, 0, C.byref(n)))
        idx = np.zeros(n.value, np.int64)
        if n.value:
            _check(L.miso_aln_fetch(self._h, tid, start, end, idx.ctypes.data, n.value, C.byref(n)))
        return idx

    def fetch(self, chrom, start, end):
        """bamfile.fetch(chrom, start, end): reads overlapping the 0-based half-open region."""
        return [self.read(int(i)) for i in self.fetch_indices(chrom, start, end)]

    def parse_reads(self, chrom, start, end, paired_end=False, strand_rule=None,
                    target_strand=None, given_read_len=None):
        """fetch + sam_parse_reads natively: ((positions, cigars), num_reads) of one event."""
        if strand_rule == "fr-secondstrand":
            raise Exception("fr-secondstrand currently unsupported.")      # sam_utils.py:331
        if strand_rule not in STRAND_RULES:
            raise Exception("Unknown strandedness rule.")                   # sam_utils.py:348, 360
        rule = STRAND_RULES[strand_rule]
        tid = self.gettid(chrom)
        if tid < 0:
            raise ValueError("invalid reference `%s`" % chrom)
        ts = ord(target_strand[0]) if target_strand else 0
        L = _native()
        n, nb, nd = C.c_int64(), C.c_int64(), C.c_int64()
        args = (self._h, tid, start, end, 1 if paired_end else 0, rule, ts,
                int(given_read_len) if given_read_len is not None else 0)
        _check(L.miso_aln_parse_reads(*args, None, 0, None, 0, C.byref(n), C.byref(nb), C.byref(nd)))
        npos = n.value * (2 if paired_end else 1)
        pos = np.zeros(npos, np.int32)
        buf = C.create_string_buffer(max(nb.value, 1))
        if npos:
            _check(L.miso_aln_parse_reads(*args, pos.ctypes.data, npos, buf, nb.value, C.byref(n),
                                          C.byref(nb), C.byref(nd)))
        cigars = tuple(buf.raw[:nb.value].decode().split("\0")[:-1]) if nb.value else ()
        self.last_strand_discarded = nd.value
        return (tuple(int(p) for p in pos), cigars), int(n.value)


def load_bam_reads(bam_filename, template=None):
    """Open (and decode) an alignment file; `template` is accepted for signature parity only."""
    path = os.path.abspath(os.path.expanduser(bam_filename))
    if path in _PRELOADED:      # decoded by the dispatcher before this worker was forked
        print("Using the alignments of %s decoded by the dispatcher" % path)
        return _PRELOADED[path]
    print("Loading BAM filename from: %s" % path)
    return Samfile(path, "rb", template=template)


def resolve_chrom(bamfile, chrom):
    """The annotation's chromosome name as the file spells it: unchanged when the file has it,
    otherwise without its first "chr" (reference: sam_utils.py:160-168)."""
    if chrom in bamfile.references:
        return chrom
    pieces = chrom.split("chr")          # "chr10" -> "10"; (odd names like "chrUn_chr1" -> "Un_": kept)
    return pieces[1] if len(pieces) > 1 else pieces[0]


def fetch_bam_reads_in_gene(bamfile, chrom, start, end, gene=None):
    """The reads overlapping a gene's region; an unknown chromosome gives no reads, not an error."""
    name = resolve_chrom(bamfile, chrom)
    try:
        return bamfile.fetch(name, start, end)
    except ValueError:
        print("Cannot fetch reads in region: %s:%d-%d" % (name, start, end))
        return []


def flag_to_strand(flag):
    return "-" if flag & 0x10 else "+"


_MATE_SUFFIXES = ("/1", "/2", "#1", "#2")


def strip_mate_id(read_name):
    """Mate suffix removed -- THREE characters, as the reference does (sam_utils.py:199-213 cuts
    `[0:-3]` for a two-character suffix); kept because it decides which reads share a name."""
    return read_name[:-3] if read_name.endswith(_MATE_SUFFIXES) else read_name


def _pairable(read):
    return read.is_paired and not (read.is_qcfail or read.is_unmapped or read.mate_is_unmapped)


def pair_sam_reads(samfile, filter_reads=True, return_unpaired=False, strand_rule=None,
                   verbose=False):
    """Mates grouped by (stripped) read name: {name: [first, second]} for the names seen exactly
    twice with the mates on opposite strands, in order of first appearance.  Rules of the reference
    (sam_utils.py:216-300): reads failing QC / unmapped / mate unmapped / not paired never enter a
    group; under fr-firststrand a just-completed pair is reversed when its first entry is a reverse
    read 1, and then again when its (new) first entry is a reverse read 2."""
    groups = OrderedDict()
    unpaired = {}
    firststrand = strand_rule == "fr-firststrand"
    for read in samfile:
        name = strip_mate_id(read.qname)
        if filter_reads and not _pairable(read):
            unpaired[name] = read
            continue
        mates = groups.setdefault(name, [])
        mates.append(read)
        if firststrand and len(mates) == 2:
            for is_mate in ("is_read1", "is_read2"):
                if getattr(mates[0], is_mate) and mates[0].is_reverse:
                    mates.reverse()
    pairs = OrderedDict()
    n_single = n_same_strand = 0
    for name, mates in groups.items():
        if len(mates) != 2:
            unpaired[name] = mates
            n_single += 1
        elif mates[0].is_reverse == mates[1].is_reverse:
            n_same_strand += 1
        else:
            pairs[name] = mates
    if verbose:
        print("Filtered out %d read pairs that were on same strand." % n_same_strand)
        print("Filtered out %d reads that had no paired mate." % n_single)
        print("  - Total read pairs: %d" % len(pairs))
    return (pairs, unpaired) if return_unpaired else pairs


def sam_cigar_to_str(sam_cigar):
    """[(op, length), ...] -> "36M", "" for no CIGAR."""
    return "".join("%d%s" % (length, CIGAR_TYPES[op]) for op, length in (sam_cigar or ()))


def read_matches_strand(read, target_strand, strand_rule, paired_end=None):
    """Does a read (or a (mate1, mate2) pair when paired_end is given) agree with the annotation's
    strand under the library's strand rule?  fr-unstranded: always.  fr-firststrand, single-end: the
    read's strand is the target's; paired: '+' target wants the first mate forward, '-' target the
    second mate reverse, any other target matches nothing (None, as in the reference).
    fr-secondstrand and unknown rules raise (sam_utils.py:325-370)."""
    if strand_rule == "fr-unstranded":
        return True
    if strand_rule == "fr-secondstrand":
        raise Exception("fr-secondstrand currently unsupported.")
    if strand_rule != "fr-firststrand":
        raise Exception("Unknown strandedness rule.")
    if paired_end is None:
        return flag_to_strand(read.flag) == target_strand
    first, second = read
    if target_strand == "+":
        return flag_to_strand(first.flag) == "+"
    if target_strand == "-":
        return flag_to_strand(second.flag) == "-"
    return None


def sam_parse_reads(samfile, paired_end=False, strand_rule=None, target_strand=None,
                    given_read_len=None, verbose=False):
    """((positions, CIGAR strings), number of reads or pairs) as the sampler takes them
    (sam_utils.py:373-452): 0-based positions, two consecutive entries per pair; reads without a
    CIGAR, of another length than given_read_len, or on the wrong strand are left out.  The strand
    is only checked when both a rule other than fr-unstranded and a target strand are given."""
    check_strand = strand_rule not in (None, "fr-unstranded") and target_strand is not None
    units = pair_sam_reads(samfile, strand_rule=strand_rule, verbose=verbose).values() if paired_end \
        else ([read] for read in samfile)
    positions, cigars = [], []
    kept = discarded = 0
    for mates in units:
        if check_strand and paired_end:
            if not read_matches_strand(mates, target_strand, strand_rule, paired_end=paired_end):
                discarded += 1
                continue
        if any(m.cigar is None for m in mates):
            continue
        if given_read_len is not None and any(m.rlen != given_read_len for m in mates):
            continue
        if check_strand and not paired_end:
            if not read_matches_strand(mates[0], target_strand, strand_rule, paired_end=paired_end):
                discarded += 1
                continue
        positions += [int(m.pos) for m in mates]
        cigars += [sam_cigar_to_str(m.cigar) for m in mates]
        kept += 1
    if check_strand and verbose:
        print("No. reads discarded due to strand violation: %d" % discarded)
    return (tuple(positions), tuple(cigars)), kept
