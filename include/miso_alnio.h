/*
 * miso_alnio.h -- C ABI of the alignment reader: the step
 * immediately before the sampler.  Part of libmiso_amd.so.
 *
 * What it replaces in the reference (all through the third-party `pysam` module, which is not
 * under /root/reference and not installed here):
 *   misopy/sam_utils.py:139-150   load_bam_reads            pysam.Samfile(bam, "rb")
 *   misopy/sam_utils.py:153-186   fetch_bam_reads_in_gene   bamfile.fetch(chrom, start, end)
 *   misopy/sam_utils.py:207-300   pair_sam_reads            mate pairing by read name
 *   misopy/sam_utils.py:303-442   sam_parse_reads           strand / read-length filters,
 *                                                           (positions, CIGAR strings) tuples
 * The reference walks a Python object per read; a GPU batch needs the reads of tens of thousands
 * of events, so here the file is decoded ONCE (BGZF blocks inflated in parallel, or SAM text
 * parsed) into columns, indexed by (reference, position), and an event's reads come out of
 * miso_aln_parse_reads as flat arrays that go straight into miso_batch_add_event.
 *
 * Formats: BAM (BGZF, SAM spec v1 section 4) and SAM text; no .bai needed (the index is built in
 * memory while loading).  Coordinates follow pysam: `pos` 0-based, fetch regions 0-based
 * half-open, a record overlaps a region when pos < end && end_pos > start, end_pos = pos +
 * (reference bases consumed by M/D/N/=/X) or pos + 1 for unmapped / CIGAR-less records (htslib
 * bam_endpos).
 *
 * Errors: functions return 0 on success or a MISO_* code of miso_amd.h (MISO_EINVAL for bad
 * arguments / malformed files, MISO_ENOMEM, MISO_FAILURE for I/O); miso_aln_last_error() has the
 * text.  Not thread-safe per handle for open/close; fetch/parse are read-only and re-entrant.
 */
#ifndef MISO_ALNIO_H
#define MISO_ALNIO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct miso_alnfile miso_alnfile_t;

/* strand_rule values (misopy/sam_utils.py:320-360; settings key `strand`) */
#define MISO_STRAND_UNSTRANDED  0   /* "fr-unstranded" or no rule: nothing is discarded          */
#define MISO_STRAND_FIRSTSTRAND 1   /* "fr-firststrand"                                            */
/* "fr-secondstrand" is an exception in the reference (sam_utils.py:331): callers raise, no code */

/* Columns of the decoded file, one entry per record in file order (borrowed pointers, valid until
 * miso_aln_close).  cigar[cigar_off[i] .. cigar_off[i+1]) are BAM-encoded ops: len << 4 | op with
 * op indexing "MIDNSHP=X".  names[name_off[i] .. name_off[i+1]) is the read name (no NUL). */
typedef struct {
  int64_t n;
  const int32_t *ref_id;     /* -1 = no reference ("*")                                         */
  const int32_t *pos;        /* 0-based leftmost coordinate (pysam read.pos)                      */
  const int32_t *end;        /* htslib bam_endpos                                                 */
  const int32_t *flag;       /* SAM FLAG                                                          */
  const int32_t *l_seq;      /* query length (pysam read.rlen)                                    */
  const uint64_t *cigar_off; /* n + 1                                                             */
  const uint32_t *cigar;
  const uint64_t *name_off;  /* n + 1                                                             */
  const char *names;
} miso_aln_columns_t;

/* Open and decode a BAM or SAM file (detected by content).  n_threads <= 0: all usable cores. */
int miso_aln_open(const char *path, int n_threads, miso_alnfile_t **out);
void miso_aln_close(miso_alnfile_t *f);

int miso_aln_columns(const miso_alnfile_t *f, miso_aln_columns_t *cols);
int miso_aln_n_refs(const miso_alnfile_t *f);
const char *miso_aln_ref_name(const miso_alnfile_t *f, int ref);     /* NULL if out of range    */
int64_t miso_aln_ref_length(const miso_alnfile_t *f, int ref);
int miso_aln_ref_id(const miso_alnfile_t *f, const char *name);      /* -1 if absent            */
int miso_aln_is_bam(const miso_alnfile_t *f);

/* bamfile.fetch(chrom, start, end): indices (into the columns) of the records overlapping
 * [start, end) on reference `ref`, ordered by (pos, file order).  Two-call pattern: *n receives
 * the number of hits; at most `cap` indices are written to idx (idx may be NULL when cap == 0). */
int miso_aln_fetch(const miso_alnfile_t *f, int ref, int64_t start, int64_t end,
                   int64_t *idx, int64_t cap, int64_t *n);

/* fetch + sam_parse_reads in one call: the (positions, CIGAR strings) of one event.
 *   paired        0: single-end (sam_utils.py:417-436); 1: mates paired by name
 *                 (pair_sam_reads, sam_utils.py:207-300), two consecutive entries per pair
 *   strand_rule   MISO_STRAND_*; target_strand '+', '-', 0 = no target strand: no strand check
 *                 (sam_utils.py:385-390; the fr-firststrand mate swap of pair_sam_reads still
 *                 applies), any other character: single-end compares it with the read's strand,
 *                 paired fr-firststrand matches nothing, as the reference's function falls
 *                 through to None (sam_utils.py:337-346)
 *   given_read_len  > 0: drop reads (pairs) whose query length differs (sam_utils.py:399-404,
 *                 423-426); <= 0: no filter
 * Outputs (two-call pattern like miso_aln_fetch): *n_reads = number of reads (pairs) kept --
 * the reference's num_raw_reads; positions (0-based, 1 or 2 per read); the CIGAR strings
 * concatenated, NUL-terminated each, into cigar_buf (needed size in *cigar_bytes).
 * *n_strand_discarded (may be NULL) counts reads dropped by the strand rule. */
int miso_aln_parse_reads(const miso_alnfile_t *f, int ref, int64_t start, int64_t end, int paired,
                         int strand_rule, int target_strand, int given_read_len,
                         int32_t *positions, int64_t pos_cap, char *cigar_buf, int64_t cigar_cap,
                         int64_t *n_reads, int64_t *cigar_bytes, int64_t *n_strand_discarded);

/* One event straight from the file into a sampler batch (miso_amd.h): miso_aln_parse_reads with the
 * batch's single-end / paired-end mode, positions made 1-based (misopy/miso_sampler.py:284), then
 * miso_batch_add_event -- no per-read work in the host language.  *n_reads = reads (pairs) found;
 * the event is added only when 0 < *n_reads and *n_reads >= min_reads (run_miso.py:139-147), else
 * *event_index = -1 and the call still succeeds.  Errors as miso_batch_add_event (bad CIGAR, ...). */
struct miso_batch;
struct miso_gene;
int miso_batch_add_event_aln(struct miso_batch *batch, const struct miso_gene *gene,
                             const miso_alnfile_t *f, int ref, int64_t start, int64_t end,
                             int strand_rule, int target_strand, int given_read_len, int64_t min_reads,
                             const double *hyperp, int n_hyperp, int64_t *n_reads, int *event_index);

/* The same for n events at once, reads collected and CIGARs parsed on n_threads host threads (<= 0:
 * all usable cores), events appended in the order given; default hyperparameters.  n_reads[i] and
 * event_index[i] as above.  On an error (bad CIGAR, ...) nothing is added and the first failing event's
 * error is reported. */
int miso_batch_add_events_aln(struct miso_batch *batch, int n, const struct miso_gene *const *genes,
                              const miso_alnfile_t *f, const int *ref, const int64_t *start,
                              const int64_t *end, int strand_rule, const int *target_strand,
                              int given_read_len, int64_t min_reads, int n_threads, int64_t *n_reads,
                              int *event_index);

/* host threads the library uses by default: affinity mask capped by the cgroup CPU quota, <= 64 */
int miso_usable_threads(void);

const char *miso_aln_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
