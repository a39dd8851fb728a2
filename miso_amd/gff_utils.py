"""GFF3 annotations for the front end: records, the gene -> mRNA -> exon tables
and the index helpers `run_miso.py` needs.

Interface names follow misopy/gff_utils.py so callers read the same (`GFF`, `Reader`,
`GFFDatabase`, `get_inclusive_txn_bounds`, `load_indexed_gff_file`, `get_gene_ids_to_gff_index`;
reference lines 55-153, 164-292, 316-489, 509-747, 955-981), the implementation is this package's
own: one pass over the file, GFF version 3 only.  Behaviour kept from the reference because it
changes results: attribute values are URL-unquoted and comma-split, start/end are swapped when
reversed, exons without an ID are named `parent@start@end@strand`, transcripts are `mRNA` or
`transcript` records, a gene's transcripts and a transcript's exons keep file order.
"""
import json
import os
import pickle
import sys
from collections import OrderedDict
from urllib.parse import unquote

INDEX_MAP_BASENAME = "genes_to_filenames.json"
BUNDLE_BASENAME = "genes_bundle.pickle"

TRANSCRIPT_TYPES = ("mRNA", "transcript")


class FormatError(Exception):
    """A line that is not a 9-column GFF3 record."""


def parse_maybe_empty(text, parse_type=str):
    """'.' is GFF's empty field."""
    return None if text == "." else parse_type(text)


def _attributes(column9):
    """`tag=v1,v2;tag2=...` -> {tag: [values]}; malformed pairs are reported and dropped."""
    out = {}
    for pair in filter(None, column9.split(";")):
        tag, sep, value = pair.partition("=")
        if not sep or "=" in value:
            sys.stderr.write("WARNING: Invalid attributes string: %s\n" % column9)
            continue
        out[unquote(tag)] = [unquote(v) for v in value.split(",")]
    return out


class GFF(object):
    """One annotation record: seqid, source, type, start, end, score, strand, phase, attributes."""
    __slots__ = ("seqid", "source", "type", "start", "end", "score", "strand", "phase", "attributes")

    def __init__(self, seqid, source, type, start, end, score=None, strand=None, phase=None,
                 attributes=None):
        if start > end:
            start, end = end, start
            if strand != "-":
                sys.stderr.write("WARNING: start > end swapped for %s %s:%d-%d\n"
                                 % (type, seqid, start, end))
        self.seqid, self.source, self.type = seqid, source, type
        self.start, self.end = start, end
        self.score, self.strand, self.phase = score, strand, phase
        self.attributes = attributes or {}
        if type == "exon" and "ID" not in self.attributes:
            self.attributes["ID"] = ["%s@%s@%s@%s" % (self.get_parent(), start, end, strand)]

    @classmethod
    def from_line(cls, line):
        cols = line.strip().split("\t")
        if len(cols) != 9:
            raise FormatError("Invalid number of fields (should be 9):\n" + line.strip())
        try:
            return cls(unquote(cols[0]), unquote(cols[1]), unquote(cols[2]), int(cols[3]), int(cols[4]),
                       parse_maybe_empty(cols[5], float), parse_maybe_empty(cols[6]),
                       parse_maybe_empty(cols[7], int), _attributes(cols[8]))
        except ValueError as err:
            raise FormatError("GFF field format error: %s" % err)

    def get_values(self, key):
        return self.attributes.get(key, [])

    def get_value(self, key):
        """First value of an attribute without trailing blanks, "" when the record has none."""
        values = self.attributes.get(key)
        return values[0].rstrip() if values else ""

    def get_id(self):
        return self.get_value("ID")

    def get_parent(self):
        return self.get_value("Parent")

    def get_name(self):
        return self.get_value("Name")

    def __getstate__(self):
        return tuple(getattr(self, f) for f in self.__slots__)

    def __setstate__(self, state):
        for f, v in zip(self.__slots__, state):
            setattr(self, f, v)

    def __repr__(self):
        return "GFF(%s %s %s:%s-%s %s %s)" % (self.type, self.get_id(), self.seqid, self.start,
                                              self.end, self.strand, self.attributes)


class Reader(object):
    """Iterates the records of a GFF3 stream; directives, comments and blank lines are skipped and a
    FASTA section ends the records."""

    def __init__(self, stream, version="3"):
        if str(version) != "3":
            raise NotImplementedError("only GFF version 3 is read")
        self._stream = stream

    def __iter__(self):
        for line in self._stream:
            if line.startswith(">"):
                break
            if line.startswith("#") or not line.strip():
                continue
            yield GFF.from_line(line)

    def read_recs(self, reverse_recs=False):
        records = list(self)
        return records[::-1] if reverse_recs else records


class GFFDatabase(object):
    """The file's genes plus the child tables `mRNAs_by_gene`, `exons_by_mRNA`, `cdss_by_exon`,
    every list in file order."""

    def __init__(self, from_filename=None, reverse_recs=False, include_introns=False,
                 suppress_warnings=False):
        self.genes, self.mRNAs, self.exons, self.cdss = [], [], [], []
        self.mRNAs_by_gene, self.exons_by_mRNA, self.cdss_by_exon = {}, {}, {}
        self.suppress_warnings = suppress_warnings
        self.from_filename = None
        if from_filename:
            self.from_file(from_filename, reverse_recs=reverse_recs, include_introns=include_introns)

    def from_file(self, filename, version="3", reverse_recs=False, include_introns=False):
        exon_like = ("exon", "intron") if include_introns else ("exon",)
        with open(filename) as stream:
            for rec in Reader(stream, version).read_recs(reverse_recs=reverse_recs):
                if rec.type == "gene":
                    self.genes.append(rec)
                elif rec.type in TRANSCRIPT_TYPES:
                    self.mRNAs.append(rec)
                    self.mRNAs_by_gene.setdefault(rec.get_parent(), []).append(rec)
                elif rec.type in exon_like:
                    self.exons.append(rec)
                    self.exons_by_mRNA.setdefault(rec.get_parent(), []).append(rec)
                elif rec.type == "CDS":
                    self.cdss.append(rec)
                    self.cdss_by_exon.setdefault(rec.get_parent(), []).append(rec)
        self.from_filename = filename

    def gene_tree(self, gene_id):
        """{'mRNAs': {transcript id: {'record', 'exons': {exon id: {'record', 'cdss'}}}}} of one gene
        and its records (transcripts, then exons, then CDSs); None when the gene has no children."""
        tree = OrderedDict()
        transcripts, exons, cdss = self.mRNAs_by_gene.get(gene_id, []), [], []
        for t in transcripts:
            node = tree[t.get_id()] = {"record": t, "exons": OrderedDict()}
            for ex in self.exons_by_mRNA.get(t.get_id(), []):
                kids = OrderedDict((c.get_id(), {"record": c}) for c in self.cdss_by_exon.get(ex.get_id(), []))
                node["exons"][ex.get_id()] = {"record": ex, "cdss": kids}
                exons.append(ex)
                cdss.extend(k["record"] for k in kids.values())
        if not (transcripts or exons or cdss):
            return None, []
        return {"mRNAs": tree}, list(transcripts) + exons + cdss

    def get_genes_records(self, genes):
        """(records, {gene id: tree}) for several genes; genes without children are left out."""
        records, hierarchy = [], {}
        for gene_id in genes:
            tree, recs = self.gene_tree(gene_id)
            if tree is None:
                if not self.suppress_warnings:
                    print("WARNING: No entries found for gene %s in GFF %s" % (gene_id, self.from_filename))
                continue
            hierarchy[gene_id] = tree
            records += recs
        return records, hierarchy


def get_inclusive_txn_bounds(gene_hierarchy):
    """(smallest transcript start, largest transcript end) of a gene's tree."""
    records = [node["record"] for node in gene_hierarchy["mRNAs"].values()]
    assert records, "gene without transcripts"
    tx_start = min(r.start for r in records)
    tx_end = max(r.end for r in records)
    assert tx_start < tx_end
    return tx_start, tx_end


def load_indexed_gff_file(indexed_gff_filename):
    """One index file written by index_gff.py: {gene id: {'gene_object', 'hierarchy'}}."""
    with open(indexed_gff_filename, "rb") as handle:
        return pickle.load(handle)


def get_gene_ids_to_gff_index(indexed_gff_dir, verbose=False):
    """gene id -> its index file.  Read from the JSON map written at indexing time (the reference
    keeps a `shelve`); an index directory without the map is scanned, chromosome by chromosome."""
    map_path = os.path.join(indexed_gff_dir, INDEX_MAP_BASENAME)
    if os.path.isfile(map_path):
        with open(map_path) as handle:
            relative = json.load(handle, object_pairs_hook=OrderedDict)
        return OrderedDict((gene, os.path.join(indexed_gff_dir, rel)) for gene, rel in relative.items())
    found = OrderedDict()
    for chrom in sorted(os.listdir(indexed_gff_dir)):
        chrom_path = os.path.abspath(os.path.join(indexed_gff_dir, chrom))
        if os.path.isdir(chrom_path):
            for name in sorted(n for n in os.listdir(chrom_path) if n.endswith(".pickle")):
                path = os.path.join(chrom_path, name)
                found.update((gene, path) for gene in load_indexed_gff_file(path))
    return found
