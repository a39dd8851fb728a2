"""misopy/index_gff.py for Python 3: GFF3 annotation -> one indexed file per gene.

    python -m miso_amd.index_gff --index annotation.gff indexed_dir/ [--compress-id]

Same layout as the reference (index_gff.py:29-131): `indexed_dir/chrN/<gene_id>.pickle` holding
`{gene_id: {'gene_object': Gene, 'hierarchy': ...}}`, plus `genes.gff`.  Differences, by necessity:
the pickles are Python-3 pickles of miso_amd.gene_utils objects (the reference's are Python-2
pickles of misopy classes); the gene -> file map is `genes_to_filenames.json` with paths relative
to the index directory instead of a `shelve`; compressed IDs use a stable FNV-1a hash instead of
Python-2's `hash()`.
"""
import glob
import json
import os
import pickle
import sys
import time
from collections import OrderedDict, defaultdict

from . import gene_utils
from .gff_utils import BUNDLE_BASENAME, INDEX_MAP_BASENAME, get_inclusive_txn_bounds

COMPRESS_PREFIX = "misocomp"                                       # misc_utils.COMPRESS_PREFIX


def compress_event_name(event_name, prefix=COMPRESS_PREFIX):
    """index_gff.py:22-26 with a reproducible 64-bit FNV-1a instead of Python-2 hash()."""
    h = 0xCBF29CE484222325
    for b in event_name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return "%s_%d" % (prefix, h)


def serialize_genes(gff_genes, gff_filename, output_dir, compress_id=False):
    """index_gff.py:29-131."""
    genes_by_chrom = defaultdict(OrderedDict)
    for gene_id, gene_info in gff_genes.items():
        gene_obj = gene_info["gene_object"]
        entry = {'gene_object': gene_obj, 'hierarchy': gene_info["hierarchy"]}
        if compress_id:
            entry['compressed_id'] = compress_event_name(gene_id)
        genes_by_chrom[gene_obj.chrom][gene_id] = entry
    gene_id_to_filename = OrderedDict()
    compressed_id_to_gene_id = OrderedDict()
    for chrom, chrom_genes in genes_by_chrom.items():
        chrom_dir_name = chrom if chrom.startswith("chr") else "chr%s" % str(chrom)
        chrom_dir = os.path.join(output_dir, chrom_dir_name)
        os.makedirs(chrom_dir, exist_ok=True)
        for gene_id, gene_info in chrom_genes.items():
            base = gene_info['compressed_id'] if compress_id else gene_id
            gene_filename = os.path.join(chrom_dir, "%s.pickle" % base)
            with open(gene_filename, "wb") as f:
                pickle.dump({gene_id: gene_info}, f, protocol=4)
            gene_id_to_filename[gene_id] = os.path.relpath(gene_filename, output_dir)
            if compress_id:
                compressed_id_to_gene_id[base] = gene_id
    with open(os.path.join(output_dir, INDEX_MAP_BASENAME), "w") as f:
        json.dump(gene_id_to_filename, f)
    # all genes once more in ONE file: a whole-genome run then opens one pickle instead of 40 000
    # (run_miso.collect_gene_events uses it when present; the per-gene files stay the reference layout)
    bundle = {}
    for chrom_genes in genes_by_chrom.values():
        for gene_id, info in chrom_genes.items():
            bounds = get_inclusive_txn_bounds(info["hierarchy"][gene_id])
            bundle[gene_id] = gene_utils.gene_to_compact(info["gene_object"], *bounds)
    with open(os.path.join(output_dir, BUNDLE_BASENAME), "wb") as f:
        pickle.dump(bundle, f, protocol=4)
    with open(os.path.join(output_dir, "compressed_ids_to_genes.json"), "w") as f:
        json.dump(compressed_id_to_gene_id, f)
    genes_filename = os.path.join(output_dir, "genes.gff")
    with open(gff_filename) as gff_in, open(genes_filename, "w") as gff_out:
        for line in gff_in:
            if line.startswith("#"):
                continue
            fields = line.strip().split("\t")
            if len(fields) > 2 and fields[2] == "gene":
                gff_out.write(line)
    return gene_id_to_filename


def index_gff(gff_filename, output_dir, compress_id=False):
    """index_gff.py:134-166."""
    print("Indexing GFF...")
    if len(glob.glob(os.path.join(output_dir, "chr*"))) >= 1:
        print("%s appears to already be indexed. Aborting." % gff_filename)
        return
    print("  - GFF: %s" % gff_filename)
    print("  - Outputting to: %s" % output_dir)
    t1 = time.time()
    gff_genes = gene_utils.load_genes_from_gff(gff_filename)
    os.makedirs(output_dir, exist_ok=True)
    serialize_genes(gff_genes, gff_filename, output_dir, compress_id=compress_id)
    print("Indexing of GFF took %.2f seconds." % (time.time() - t1))


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Indexer of GFF files for use with MISO.")
    ap.add_argument("--index", nargs=2, metavar=("GFF", "OUTPUT_DIR"))
    ap.add_argument("--compress-id", action="store_true")
    a = ap.parse_args(argv)
    if a.index is None:
        print("Need to pass --index, for example:\n\nindex_gff --index annotation.gff indexed_annotation/")
        return 1
    gff_filename = os.path.abspath(os.path.expanduser(a.index[0]))
    output_dir = os.path.abspath(os.path.expanduser(a.index[1]))
    os.makedirs(output_dir, exist_ok=True)
    index_gff(gff_filename, output_dir, compress_id=a.compress_id)
    return 0


if __name__ == "__main__":
    sys.exit(main())
