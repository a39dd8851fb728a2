"""Python-3 mirror of the part of misopy/gff_utils.py that stands between a GFF3 annotation and
the sampler: the record type, the v3 reader, the gene -> mRNA -> exon database
and the helpers `run_miso.py` calls.  GFF v1/v2 parsing, the writer and CDS bookkeeping beyond
what the hierarchy stores are out of scope.
"""
import json
import os
import pickle
import sys
from collections import OrderedDict, defaultdict
from urllib.parse import unquote as url_unquote

INDEX_MAP_BASENAME = "genes_to_filenames.json"
BUNDLE_BASENAME = "genes_bundle.pickle"


class FormatError(Exception):
    pass


def parse_maybe_empty(s, parse_type=str):
    """gff_utils.py:819-823."""
    return None if s == '.' else parse_type(s)


class GFF(object):
    """A record from a GFF file (gff_utils.py:316-489)."""

    def __init__(self, seqid, source, type, start, end, score=None, strand=None, phase=None,
                 attributes=None):
        self.seqid, self.source, self.type = seqid, source, type
        self.start, self.end = start, end
        self.score, self.strand, self.phase = score, strand, phase
        self.attributes = attributes if attributes else {}
        if self.start > self.end:                                   # gff_utils.py:350-355
            self.start, self.end = self.end, self.start
            if strand != '-':
                sys.stderr.write("WARNING: Swapping start and end fields, which must satisfy "
                                 "start <= end:\n%r\n" % self)
        self._set_default_exon_id()

    def _set_default_exon_id(self):
        """gff_utils.py:362-376: parent@start@end@strand for exons without an ID."""
        if self.type == "exon" and "ID" not in self.attributes:
            self.attributes['ID'] = ["%s@%s@%s@%s" % (self.get_parent(), self.start, self.end,
                                                      self.strand)]

    def get_values(self, key):
        return self.attributes.get(key, [])

    def get_value(self, key):
        """First value, trailing whitespace removed; "" when absent (gff_utils.py:463-470)."""
        if key in self.attributes:
            return self.attributes[key][0].rstrip()
        return ""

    def get_id(self):
        return self.get_value("ID")

    def get_parent(self):
        return self.get_value("Parent")

    def get_name(self):
        return self.get_value("Name")

    def __repr__(self):
        return "GFF(%s, %s, %s, %s, %s, %s)" % (self.seqid, self.type, self.start, self.end,
                                                self.strand, self.attributes)


class Reader(object):
    """GFF3 reader (gff_utils.py:509-747, version "3" only)."""

    def __init__(self, stream, version="3"):
        if str(version) != "3":
            raise NotImplementedError("only GFF version 3 is read")
        self._stream = stream

    def __iter__(self):
        for line in self._stream:
            if line.startswith("#") or line == "\n":              # directives, comments, blanks
                continue
            if line.startswith(">"):                               # FASTA section ends the records
                return
            yield self._parse_record_v3(line)

    def read_recs(self, reverse_recs=False):
        recs = list(self)
        if reverse_recs:
            recs.reverse()
        return recs

    def _parse_record_v3(self, line):
        line = line.strip()
        fields = line.split('\t')
        if len(fields) != 9:
            raise FormatError("Invalid number of fields (should be 9):\n" + line)
        try:
            return GFF(seqid=url_unquote(fields[0]), source=url_unquote(fields[1]),
                       type=url_unquote(fields[2]), start=int(fields[3]), end=int(fields[4]),
                       score=parse_maybe_empty(fields[5], float),
                       strand=parse_maybe_empty(fields[6]),
                       phase=parse_maybe_empty(fields[7], int),
                       attributes=self._parse_attributes_v3(fields[8]))
        except ValueError as e:
            raise FormatError("GFF field format error: %s" % e)

    def _parse_attributes_v3(self, s):
        attributes = {}
        for pair_string in s.split(";"):
            if len(pair_string) == 0:
                continue
            try:
                tag, value = pair_string.split("=")
                attributes[url_unquote(tag)] = [url_unquote(v) for v in value.split(",")]
            except ValueError:
                sys.stderr.write("WARNING: Invalid attributes string: %s\n" % s)
        return attributes


class GFFDatabase(object):
    """gff_utils.py:164-292: genes, mRNAs by gene, exons by mRNA, in file order."""

    def __init__(self, from_filename=None, reverse_recs=False, include_introns=False,
                 suppress_warnings=False):
        self.genes, self.mRNAs, self.exons, self.cdss = [], [], [], []
        self.mRNAs_by_gene = defaultdict(list)
        self.exons_by_mRNA = defaultdict(list)
        self.cdss_by_exon = defaultdict(list)
        self.suppress_warnings = suppress_warnings
        self.from_filename = from_filename
        if from_filename:
            self.from_file(from_filename, reverse_recs=reverse_recs,
                           include_introns=include_introns)

    def from_file(self, filename, version="3", reverse_recs=False, include_introns=False):
        with open(filename, "r") as stream:
            for record in Reader(stream, version).read_recs(reverse_recs=reverse_recs):
                if record.type == "gene":
                    self.genes.append(record)
                elif record.type == "mRNA" or record.type == "transcript":
                    self.mRNAs.append(record)
                    self.mRNAs_by_gene[record.get_parent()].append(record)
                elif record.type == "exon" or (include_introns and record.type == "intron"):
                    self.exons.append(record)
                    self.exons_by_mRNA[record.get_parent()].append(record)
                elif record.type == "CDS":
                    self.cdss.append(record)
                    self.cdss_by_exon[record.get_parent()].append(record)
        self.from_filename = filename

    def get_genes_records(self, genes):
        """gff_utils.py:226-291: (records, hierarchy) with
        hierarchy[gene]['mRNAs'][mRNA_id] = {'record', 'exons': {exon_id: {'record', 'cdss'}}}."""
        recs = []
        gene_hierarchy = {}
        for gene in genes:
            mRNAs, exons, cdss = [], [], []
            gene_hierarchy[gene] = {'mRNAs': OrderedDict()}
            genes_mRNAs = self.mRNAs_by_gene.get(gene, [])
            for mRNA_rec in genes_mRNAs:
                gene_hierarchy[gene]['mRNAs'][mRNA_rec.get_id()] = {'exons': OrderedDict(),
                                                                    'record': mRNA_rec}
                mRNAs.append(mRNA_rec)
            for mRNA_rec in genes_mRNAs:
                mRNA_rec_id = mRNA_rec.get_id()
                for exon_rec in self.exons_by_mRNA.get(mRNA_rec_id, []):
                    exon_rec_id = exon_rec.get_id()
                    gene_hierarchy[gene]['mRNAs'][mRNA_rec_id]['exons'][exon_rec_id] = \
                        {'cdss': OrderedDict(), 'record': exon_rec}
                    exons.append(exon_rec)
                    for cds_rec in self.cdss_by_exon.get(exon_rec_id, []):
                        gene_hierarchy[gene]['mRNAs'][mRNA_rec_id]['exons'][exon_rec_id]['cdss'][
                            cds_rec.get_id()] = {'record': cds_rec}
                        cdss.append(cds_rec)
            if len(mRNAs) == len(exons) == len(cdss) == 0:
                if not self.suppress_warnings:
                    print("WARNING: No entries found for gene %s in GFF %s"
                          % (gene, self.from_filename))
                del gene_hierarchy[gene]
            recs.extend(mRNAs)
            recs.extend(exons)
            recs.extend(cdss)
        return recs, gene_hierarchy


def get_inclusive_txn_bounds(gene_hierarchy):
    """gff_utils.py:955-981: the most inclusive mRNA start and end of a gene."""
    mRNA_starts, mRNA_ends = [], []
    strand = None
    for mRNA_id, mRNA_info in gene_hierarchy['mRNAs'].items():
        mRNA_rec = mRNA_info["record"]
        strand = mRNA_rec.strand
        mRNA_starts.append(mRNA_rec.start)
        mRNA_ends.append(mRNA_rec.end)
    assert strand is not None
    tx_start, tx_end = min(mRNA_starts), max(mRNA_ends)
    assert tx_start < tx_end
    return tx_start, tx_end


def load_indexed_gff_file(indexed_gff_filename):
    """gff_utils.py:55-60 (Python-3 pickles written by index_gff.py of this package)."""
    with open(indexed_gff_filename, "rb") as f:
        return pickle.load(f)


def get_gene_ids_to_gff_index(indexed_gff_dir, verbose=False):
    """gff_utils.py:89-153: gene ID -> indexed file.  The map written at indexing time is a JSON
    file (the reference shelves it); without it the directories are scanned like the reference."""
    map_fname = os.path.join(indexed_gff_dir, INDEX_MAP_BASENAME)
    if os.path.isfile(map_fname):
        with open(map_fname) as f:
            rel = json.load(f, object_pairs_hook=OrderedDict)
        return OrderedDict((k, os.path.join(indexed_gff_dir, v)) for k, v in rel.items())
    gene_ids_to_gff_index = OrderedDict()
    for chrom_dir in sorted(os.listdir(indexed_gff_dir)):
        chrom_dir_path = os.path.abspath(os.path.join(indexed_gff_dir, chrom_dir))
        if not os.path.isdir(chrom_dir_path):
            continue
        for fname in sorted(os.listdir(chrom_dir_path)):
            if not fname.endswith(".pickle"):
                continue
            path = os.path.join(chrom_dir_path, fname)
            for gene_id in load_indexed_gff_file(path):
                gene_ids_to_gff_index[gene_id] = path
    return gene_ids_to_gff_index
