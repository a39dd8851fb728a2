"""The gene model the sampler's caller reads, built from GFF3.

Plays the role of misopy/Gene.py -- imported as `gene_utils`, the name run_miso.py:19 gives it.
What the rest of the package relies on, and where the reference defines it:
  * `Gene.parts`: every exon of every transcript in transcript order, one object per GFF record, so
    an exon shared by two transcripts appears twice (make_gene_from_gff_records, Gene.py:915-1009);
  * `Exon == Exon` compares coordinates and owning gene, not labels (Gene.py:78-88) -- `py2c_gene`
    (py2c_gene.py:10-21) finds an isoform's exons with `parts.index(exon)` and therefore lands on the
    first copy of a shared exon;
  * an isoform is the list of its transcript's exons sorted by start; `genomic_start` / `genomic_end`
    are the first exon's start and the last exon's end (Gene.py:696-715);
  * isoforms keep the file order of their `mRNA` / `transcript` records and carry the transcript ID
    as label; `iso_lens` are summed exon lengths.
The pure-Python read alignment of the reference class (Gene.py:342-690) is the predecessor of
`splicing_matchIso`; that job runs on the GPU here.
"""
from .gff_utils import GFFDatabase, TRANSCRIPT_TYPES


class Interval(object):
    def __init__(self, start, end):
        assert start <= end, "interval end before start"
        self.start, self.end = start, end

    @property
    def len(self):
        return self.end - self.start + 1

    def contains(self, start, end):
        return self.start <= start and end <= self.end


class Exon(Interval):
    def __init__(self, start, end, label=None, gene=None, seq="", from_gff_record=None):
        self.gene, self.label, self.seq = gene, label, seq
        self.rec = self.parent_rec = None
        if from_gff_record is not None:
            self.rec, self.parent_rec = from_gff_record["record"], from_gff_record["parent"]
            start, end, self.label = self.rec.start, self.rec.end, self.rec.attributes["ID"][0]
        Interval.__init__(self, start, end)

    def __eq__(self, other):
        return (other is not None and (self.start, self.end) == (other.start, other.end)
                and self.gene is other.gene)

    def __ne__(self, other):
        return not self == other

    __hash__ = None

    def __repr__(self):
        return "Exon(%s:%d-%d)" % (self.label, self.start, self.end)


class Isoform(object):
    def __init__(self, gene, parts, seq=None, label=None, desc=None):
        self.gene, self.parts, self.seq, self.label, self.desc = gene, parts, seq, label, desc
        self.num_parts = len(parts)
        self.len = sum(p.len for p in parts)
        self.genomic_start, self.genomic_end = parts[0].start, parts[-1].end


class Gene(object):
    """isoform_desc: one list of part labels per isoform; parts: the exon objects those labels name
    (the first part carrying a label wins, Gene.py:246-250)."""

    def __init__(self, isoform_desc, parts, chrom=None, exons_seq=None, label="", strand="NA",
                 transcript_ids=None):
        self.label = label or "gene"
        self.chrom, self.strand = chrom, strand
        self.isoform_desc, self.transcript_ids = isoform_desc, transcript_ids
        self.parts = list(parts)
        for part in self.parts:
            part.gene = self
        self.num_parts = len(self.parts)
        if transcript_ids is not None and len(transcript_ids) != len(isoform_desc):
            raise Exception("Transcript IDs do not match number of isoforms.")
        first_with_label = {}
        for part in self.parts:
            first_with_label.setdefault(part.label, part)
        self.isoforms = []
        for n, labels in enumerate(isoform_desc):
            missing = [l for l in labels if l not in first_with_label]
            if missing:
                raise Exception("Invalid description of isoforms: refers to undefined part %s, gene: %s"
                                % (missing[0], self.label))
            self.isoforms.append(Isoform(self, [first_with_label[l] for l in labels], desc=labels,
                                         label=None if transcript_ids is None else transcript_ids[n]))
        self.iso_lens = [iso.len for iso in self.isoforms]

    def get_part_by_label(self, part_label):
        return next((p for p in self.parts if p.label == part_label), None)

    def __repr__(self):
        return "Gene(%s, %d isoforms)" % (self.label, len(self.isoforms))


def make_gene_from_gff_records(gene_label, gene_hierarchy, gene_records):
    """A Gene from one gene's tree (gff_utils.GFFDatabase.gene_tree) and records."""
    order = [r.get_id() for r in gene_records if r.type in TRANSCRIPT_TYPES]
    if not order:
        raise Exception("Error: %s has no transcripts..." % gene_label)
    chrom, strand = None, "NA"
    parts, descriptions, kept = [], [], []
    for tid in order:
        node = gene_hierarchy["mRNAs"][tid]
        chrom, strand = node["record"].seqid, node["record"].strand
        if not node["exons"]:
            print("%s has no exons" % tid)
            continue
        exons = sorted((Exon(0, 0, from_gff_record={"record": e["record"], "parent": node["record"]})
                        for e in node["exons"].values()), key=lambda ex: ex.start)
        parts += exons
        descriptions.append([ex.label for ex in exons])
        kept.append(tid)
    return Gene(descriptions, parts, label=gene_label, chrom=chrom, strand=strand, transcript_ids=kept)


def load_genes_from_gff(gff_filename, include_introns=False, reverse_recs=False,
                        suppress_warnings=False):
    """{gene id: {'gene_object': Gene, 'hierarchy': {gene id: tree}}} for every gene record that has
    transcripts, in file order (what index_gff.py serialises, Gene.py:866-912)."""
    db = GFFDatabase(gff_filename, include_introns=include_introns, reverse_recs=reverse_recs,
                     suppress_warnings=suppress_warnings)
    genes = {}
    for record in db.genes:
        gene_id = record.get_id()
        records, hierarchy = db.get_genes_records([gene_id])
        if gene_id not in hierarchy:
            if not suppress_warnings:
                print("Skipping gene %s..." % gene_id)
            continue
        hierarchy[gene_id]["gene"] = record
        genes[gene_id] = {"gene_object": make_gene_from_gff_records(gene_id, hierarchy[gene_id], records),
                          "hierarchy": hierarchy}
    if not suppress_warnings:
        print("Loaded %d genes" % len(genes))
    return genes


def gene_to_compact(gene, tx_start, tx_end):
    """What a run needs of an indexed gene, as plain tuples (index_gff.py's one-file bundle): the
    arguments `Gene` was built from plus the transcript bounds -- a whole genome loads in a fraction of
    a second instead of unpickling every GFF record."""
    return (gene.label, gene.chrom, gene.strand, tuple(gene.transcript_ids or ()),
            tuple((p.start, p.end, p.label) for p in gene.parts),
            tuple(tuple(iso.desc) for iso in gene.isoforms), int(tx_start), int(tx_end))


def gene_from_compact(entry):
    """(Gene, (tx_start, tx_end)) back from gene_to_compact's tuple."""
    label, chrom, strand, tids, parts, descs, tx_start, tx_end = entry
    exons = [Exon(s, e, label=l) for s, e, l in parts]
    gene = Gene([list(d) for d in descs], exons, label=label, chrom=chrom, strand=strand,
                transcript_ids=list(tids) if tids else None)
    return gene, (tx_start, tx_end)


def se_event_to_gene(up_len, se_len, dn_len, chrom, label=None):
    """A skipped-exon event as a two-isoform gene: exons A, B, C laid end to end from coordinate 0,
    isoforms A-B-C and A-C (Gene.py:1033-1051)."""
    bounds, start = [], 0
    for length in (up_len, se_len, dn_len):
        bounds.append((start, start + length - 1))
        start += length
    exons = [Exon(s, e, label=name) for (s, e), name in zip(bounds, "ABC")]
    return Gene([["A", "B", "C"], ["A", "C"]], exons, label=label or "", chrom=chrom)
