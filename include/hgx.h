/*
 * hgx.h -- C-ABI of libhgx, the MI355X (gfx950) implementation of HISAT-genotype's
 * per-read allele-compatibility scoring + EM abundance estimation hot path.
 *
 * The reference has no FFI for this path (it is a closure-laden Python loop,
 * hisatgenotype_modules/hisatgenotype_typing_core.py:249-2171, and the free function
 * hisatgenotype_modules/hisatgenotype_typing_common.py:1282-1410).  Each entry point
 * below therefore names the reference code it replaces; the ctypes stubs a maintainer
 * would add to the reference are in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative HGX_E* code and
 *     records a message retrievable with hgx_last_error() (thread local).
 *   - "host" pointers are ordinary memory; "dev" pointers must be device memory of the
 *     current HIP device (hipMalloc / a torch tensor's data_ptr()).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - the library never falls back to a CPU implementation of a device entry point.
 */
#ifndef HGX_H
#define HGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HGX_OK            0
#define HGX_EINVAL       -1   /* bad argument                                         */
#define HGX_EHIP         -2   /* HIP runtime error (message has the hipError string)  */
#define HGX_ENOMEM       -3
#define HGX_EKEY         -4   /* reference would raise KeyError (EM quirk Q6, common:1365-1369) */
#define HGX_ECOLLISION   -5   /* class-hash collision detected by the exact verify pass */
#define HGX_EPARSE       -6   /* malformed SAM record / reference would assert          */

/* variant types, order = the reference's id order I < M < D (typing_process.py:275-295) */
#define HGX_VAR_INSERTION 0
#define HGX_VAR_SINGLE    1
#define HGX_VAR_DELETION  2

/* scoring levels (typing_core.py:1250-1291): exon = alleles grouped by exonic sequence,
 * gene = every allele.  The primary-exon level is dead code in the reference (core:1682). */
#define HGX_LEVEL_EXON 0
#define HGX_LEVEL_GENE 1

const char *hgx_last_error(void);
int hgx_version(void);

/* ---- device plumbing (so a ctypes caller needs nothing but this library) ------------- */
int hgx_device_count(int *n);
int hgx_set_device(int dev);
int hgx_dev_alloc(void **dev_ptr, size_t bytes);
int hgx_dev_free(void *dev_ptr);
int hgx_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes, void *stream);
int hgx_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes, void *stream);
int hgx_memset(void *dev_dst, int value, size_t bytes, void *stream);
int hgx_stream_sync(void *stream);

/* ---- 8a-0: packed locus index ---------------------------------------------------------
 * Replaces the per-locus dict building of typing_core.py:384-401, 476-491, 559-569.
 * Alleles are Gene_names[gene] minus the backbone, in that order (index 0..n_alleles-1);
 * variants are gene_var_list order (sorted by position).  link_bits is word-major:
 * link_bits[w * a_pad + a] bit (v & 31) of word w = v >> 5 is set iff allele a carries
 * variant v (Links[var_id] contains the allele).  a_pad = n_alleles rounded up to 64.   */
typedef struct hgx_index hgx_index;

int hgx_index_create(hgx_index **out,
                     int32_t n_alleles, int32_t n_vars,
                     const uint32_t *link_bits_host,      /* [n_words][a_pad]                 */
                     const uint64_t *exon_allele_mask,    /* host [a_pad/64]: allele_rep_set  */
                     const uint64_t *gene_allele_mask);   /* host [a_pad/64]: scored alleles  */
int hgx_index_destroy(hgx_index *ix);
int hgx_index_dims(const hgx_index *ix, int32_t *n_alleles, int32_t *a_pad, int32_t *n_vars, int32_t *n_words);
/* device address of the word-major bit matrix (for RCCL broadcast of a rank-0 index, 8e) */
int hgx_index_device_bits(const hgx_index *ix, void **dev_bits, size_t *bytes);

/* ---- 8a-5 / 8a-6: read-pair x allele compatibility -> class bitsets --------------------
 * Replaces add_count (typing_core.py:626-677) + add_stat (core:1171-1236) for a batch of
 * pairs.  A piece is one add_count call: the caller has already reduced the haplotype
 * string "left-id-..-right" to bit masks over variant words [lo_word, lo_word + n_words):
 *   masks[mask_off + 2*i]     = MP word i : known variants overlapping [left,right] (core:651-670) OR'ed with P
 *   masks[mask_off + 2*i + 1] = P  word i : the piece's own known variants (core:642-647)
 * allele a is compatible  <=>  for every i: (link_bits[lo_word+i][a] & MP_i) == P_i.
 * Per pair and level: count[a] = #compatible pieces; class = {a in level mask : count[a] == max count}
 * (max over the level's alleles, including 0: quirk Q4, core:1177-1190).
 * Outputs (device): class bitsets, one row of a_pad/64 uint64 per pair and level, and a
 * 64-bit content hash per row used by hgx_dedup_classes.                                 */
typedef struct hgx_piece {
    uint32_t mask_off;   /* index into masks[] (in uint32 units)          */
    uint16_t lo_word;    /* first 32-variant word covered                 */
    uint8_t  n_words;    /* number of words covered (>= 1)                */
    uint8_t  level;      /* HGX_LEVEL_EXON or HGX_LEVEL_GENE              */
} hgx_piece;

int hgx_score_pairs(const hgx_index *ix,
                    const hgx_piece *pieces_dev, const uint32_t *masks_dev,
                    const int32_t *pair_off_dev,          /* [n_pairs + 1] into pieces         */
                    int32_t n_pairs,
                    uint64_t *exon_bits_dev,              /* [n_pairs][a_pad/64] or NULL       */
                    uint64_t *gene_bits_dev,              /* [n_pairs][a_pad/64] or NULL       */
                    uint64_t *exon_hash_dev,              /* [n_pairs] or NULL                 */
                    uint64_t *gene_hash_dev,              /* [n_pairs] or NULL                 */
                    void *stream);

/* ---- 8a-7: class dedup -------------------------------------------------------------------
 * Replaces the Gene_cmpt / Gene_exons_cmpt dict accumulation (typing_core.py:1229-1234) and the
 * Gene_cmpt2 filtering (core:1752-1766): rows (optionally AND'ed with and_mask, empty rows
 * dropped) are grouped by content; groups come out in FIRST-SEEN order (Python dict order)
 * with the summed weight.  Exact: every row is compared with its group's first row.       */
typedef struct hgx_classes hgx_classes;   /* device-resident [n_classes][a_pad/64] + counts */

int hgx_dedup_classes(hgx_classes **out,
                      const uint64_t *rows_dev, const uint64_t *row_hash_dev_or_null,
                      const int64_t *row_weight_dev_or_null,   /* NULL = weight 1 per row   */
                      int64_t n_rows, int32_t a_pad,
                      const uint64_t *and_mask_dev_or_null,
                      void *stream);
int hgx_classes_destroy(hgx_classes *c);
int hgx_classes_dims(const hgx_classes *c, int32_t *n_classes, int32_t *a_pad);
int hgx_classes_device(const hgx_classes *c, void **bits_dev, void **count_dev /* int64 */,
                       void **first_row_dev /* int64 */);
int hgx_classes_to_host(const hgx_classes *c, uint64_t *bits_host, int64_t *count_host, int64_t *first_row_host);
int hgx_classes_from_host(hgx_classes **out, const uint64_t *bits_host, const int64_t *count_host,
                          int32_t n_classes, int32_t a_pad);

/* Gene_counts (typing_core.py:1187-1190, 1650-1651): per allele the number of pairs whose
 * class contains it, and the index of the first class (in first-seen order) containing it
 * (-1 if none) -- that is the dict insertion order used to break count ties.             */
int hgx_allele_counts(const hgx_classes *c, int64_t *count_host, int32_t *first_class_host);

/* ---- 8a-8: EM abundance ------------------------------------------------------------------
 * Replaces single_abundance (typing_common.py:1282-1410): SQUAREM-accelerated EM in FP64,
 * same step sequence, clamp, stopping rule (diff > 1e-4, < 1000 iterations) and pruning
 * (>= max/10 from iteration 10 when remove_low != 0).  allele_len_or_null != NULL selects the
 * length-normalised variant.  prob_host[a] is the abundance, or -1.0 if allele a is not in the
 * returned dict.  n_iter_host receives the number of outer iterations.                     */
int hgx_em(const hgx_classes *c, int32_t n_alleles,
           int32_t remove_low, const int32_t *allele_len_or_null /* host [n_alleles] */,
           double *prob_host, int32_t *n_iter_host, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HGX_H */
