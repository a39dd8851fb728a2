"""`misopy/Gene.py` for Python 3, imported as `gene_utils` like the reference does (run_miso.py:19) --
the gene model the sampler's caller reads (misopy/Gene.py:11-340, 696-715, 866-1010): intervals, exons,
isoforms, genes, and the GFF -> Gene construction.  The read
alignment helpers of the reference class (align_read_*, Gene.py:342-690) are the pure-Python
predecessors of `splicing_matchIso` and are not on the path (miso_amd does them on the GPU).
"""
from . import gff_utils
from .gff_utils import GFFDatabase


class Interval(object):
    def __init__(self, start, end):
        self.start, self.end = start, end
        assert self.start <= self.end
        self.len = self.end - self.start + 1

    def contains(self, start, end):
        return self.start <= start and end <= self.end


class Exon(Interval):
    """Gene.py:45-88."""

    def __init__(self, start, end, label=None, gene=None, seq="", from_gff_record=None):
        Interval.__init__(self, start, end)
        self.gene, self.label, self.seq = gene, label, seq
        if from_gff_record is not None:
            self.rec = from_gff_record['record']
            self.parent_rec = from_gff_record['parent']
            self.start, self.end = self.rec.start, self.rec.end
            self.label = self.rec.attributes['ID'][0]              # use first ID in list

    def __eq__(self, other):
        """Gene.py:78-88: same coordinates and same gene (NOT the label)."""
        if other is None:
            return False
        return self.start == other.start and self.end == other.end and self.gene is other.gene

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        return "Exon([%d, %d], id = %s)(ParentGene = %s)" % (
            self.start, self.end, self.label, self.gene.label if self.gene else None)


class Isoform(object):
    """Gene.py:696-715."""

    def __init__(self, gene, parts, seq=None, label=None):
        self.gene, self.parts = gene, parts
        self.num_parts = len(parts)
        self.len = sum(part.len for part in parts)
        self.seq, self.label = seq, label
        self.genomic_start = self.parts[0].start
        self.genomic_end = self.parts[-1].end
        self.desc = None


class Gene(object):
    """Gene.py:114-340: parts + isoform descriptions (lists of part labels)."""

    def __init__(self, isoform_desc, parts, chrom=None, exons_seq=None, label="", strand="NA",
                 transcript_ids=None):
        self.isoform_desc = isoform_desc
        self.label = label if label != "" else "gene"
        self.parts = []
        for part in parts:                                         # create_parts, Gene.py:294-303
            part.gene = self
            self.parts.append(part)
        self.num_parts = len(self.parts)
        self.chrom, self.strand, self.transcript_ids = chrom, strand, transcript_ids
        self.isoforms, self.iso_lens = [], []
        for iso in self.isoform_desc:                              # create_isoforms, :305-323
            isoform_parts = []
            for part_label in iso:
                part = self.get_part_by_label(part_label)
                if not part:
                    raise Exception("Invalid description of isoforms: refers to undefined part "
                                    "%s, gene: %s" % (part_label, self.label))
                isoform_parts.append(part)
            isoform = Isoform(self, isoform_parts)
            isoform.desc = iso
            self.isoforms.append(isoform)
            self.iso_lens.append(isoform.len)
        if self.transcript_ids is not None:                        # assign_transcript_ids, :325-334
            if len(self.transcript_ids) != len(self.isoforms):
                raise Exception("Transcript IDs do not match number of isoforms.")
            for iso, tid in zip(self.isoforms, self.transcript_ids):
                iso.label = tid

    def get_part_by_label(self, part_label):
        """Gene.py:246-250: the first part carrying the label."""
        for part in self.parts:
            if part_label == part.label:
                return part
        return None

    def __repr__(self):
        return "gene_id: %s\nisoforms: %d" % (self.label, len(self.isoforms))


def make_gene_from_gff_records(gene_label, gene_hierarchy, gene_records):
    """Gene.py:915-1009.  One isoform per mRNA/transcript in file order, exons sorted by start;
    the gene's parts are ALL exons of all transcripts (shared exons appear once per transcript --
    py2c_gene resolves them to the first equal part)."""
    mRNAs = gene_hierarchy['mRNAs']
    transcripts, isoform_desc, used_transcript_ids = [], [], []
    chrom, strand = None, "NA"
    transcript_ids = [rec.get_id() for rec in gene_records
                      if rec.type == "mRNA" or rec.type == "transcript"]
    if len(transcript_ids) == 0:
        raise Exception("Error: %s has no transcripts..." % gene_label)
    for transcript_id in transcript_ids:
        transcript_info = mRNAs[transcript_id]
        transcript_rec = transcript_info['record']
        chrom, strand = transcript_rec.seqid, transcript_rec.strand
        transcript_exons = transcript_info['exons']
        if len(transcript_exons) == 0:
            print("%s has no exons" % transcript_id)
            continue
        exons = [Exon(info['record'].start, info['record'].end,
                      from_gff_record={'record': info['record'], 'parent': transcript_rec})
                 for exon_id, info in transcript_exons.items()]
        exons = sorted(exons, key=lambda e: e.start)
        transcripts.append(exons)
        isoform_desc.append([exon.label for exon in exons])
        used_transcript_ids.append(transcript_id)
    all_exons = []
    for transcript in transcripts:
        all_exons.extend(transcript)
    return Gene(isoform_desc, all_exons, label=gene_label, chrom=chrom, strand=strand,
                transcript_ids=used_transcript_ids)


def load_genes_from_gff(gff_filename, include_introns=False, reverse_recs=False,
                        suppress_warnings=False):
    """Gene.py:866-912: {gene_id: {'gene_object': Gene, 'hierarchy': hierarchy}} in file order."""
    gff_db = GFFDatabase(gff_filename, include_introns=include_introns, reverse_recs=reverse_recs,
                         suppress_warnings=suppress_warnings)
    gff_genes = {}
    for gene in gff_db.genes:
        gene_label = gene.get_id()
        gene_records, gene_hierarchy = gff_db.get_genes_records([gene_label])
        if gene_label not in gene_hierarchy:
            if not suppress_warnings:
                print("Skipping gene %s..." % gene_label)
            continue
        gene_hierarchy[gene_label]['gene'] = gene
        gene_obj = make_gene_from_gff_records(gene_label, gene_hierarchy[gene_label], gene_records)
        if gene_obj is None:
            continue
        gff_genes[gene_label] = {'gene_object': gene_obj, 'hierarchy': gene_hierarchy}
    if not suppress_warnings:
        print("Loaded %d genes" % len(gff_genes))
    return gff_genes


def se_event_to_gene(up_len, se_len, dn_len, chrom, label=None):
    """Gene.py:1033-1051."""
    e1 = Exon(0, up_len - 1, label='A')
    e2 = Exon(e1.end + 1, e1.end + se_len, label='B')
    e3 = Exon(e2.end + 1, e2.end + dn_len, label='C')
    return Gene([['A', 'B', 'C'], ['A', 'C']], [e1, e2, e3], label=label or "", chrom=chrom)
