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
#define HGX_ETYPE        -7   /* reference would raise TypeError (quirk Q3: a non-HLA locus with ONE class, typing_core.py:1787) */

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
/* host -> device without a host sync: the bytes are copied into this thread's pinned staging buffer before the call returns
 * (the caller's buffer is free at once) and travel in stream order.  Falls back to the synchronous form above 256 KB. */
int hgx_memcpy_h2d_async(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int hgx_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes, void *stream);
int hgx_memset(void *dev_dst, int value, size_t bytes, void *stream);
int hgx_stream_sync(void *stream);
/* non-blocking HIP streams so that independent stages (exon-level EM / gene-level counts) can overlap */
int hgx_stream_create(void **stream);
int hgx_stream_create_prio(void **stream, int high_priority);   /* non-blocking stream at the device's highest / lowest priority */
int hgx_stream_destroy(void *stream);
/* the library caches device scratch allocations between calls; this returns them to the driver */
int hgx_pool_trim(void);
/* HIP events on a stream, for timing individual kernels from a ctypes caller */
int hgx_event_create(void **event);
int hgx_event_destroy(void *event);
int hgx_event_record(void *event, void *stream);
int hgx_stream_wait_event(void *stream, void *event);   /* work queued on `stream` afterwards waits for `event`; no host sync */
int hgx_event_elapsed_ms(void *start_event, void *stop_event, float *ms);   /* synchronises on stop */

/* ---- 8a-0: packed locus index ---------------------------------------------------------
 * Replaces the per-locus dict building of typing_core.py:384-401, 476-491, 559-569.
 * Alleles are Gene_names[gene] minus the backbone, in that order (index 0..n_alleles-1);
 * variants are gene_var_list order (sorted by position).  link_bits is word-major:
 * link_bits[w * a_pad + a] bit (v & 31) of word w = v >> 5 is set iff allele a carries
 * variant v (Links[var_id] contains the allele).  a_pad = n_alleles rounded up to 512
 * (use hgx_a_pad()).                                                                     */
typedef struct hgx_index hgx_index;

int32_t hgx_a_pad(int32_t n_alleles);
int hgx_index_create(hgx_index **out,
                     int32_t n_alleles, int32_t n_vars,
                     const uint32_t *link_bits_host,      /* [n_words][a_pad]                 */
                     const uint64_t *exon_allele_mask,    /* host [a_pad/64]: allele_rep_set  */
                     const uint64_t *gene_allele_mask);   /* host [a_pad/64]: scored alleles  */
int hgx_index_destroy(hgx_index *ix);
int hgx_index_dims(const hgx_index *ix, int32_t *n_alleles, int32_t *a_pad, int32_t *n_vars, int32_t *n_words);
/* device address of the word-major bit matrix */
int hgx_index_device_bits(const hgx_index *ix, void **dev_bits, size_t *bytes);
/* Index broadcast (8e): the device tables of an index are ONE allocation, [link bits | exon mask | gene mask].  The rank that
 * packed the locus passes this block to the collective as the send buffer; every other rank makes an index of the same
 * shape with hgx_index_create_device (tables uninitialised) and passes ITS block as the receive buffer -- RCCL writes the
 * tables in place over xGMI, nothing bounces through the host (dist.broadcast_index: torch.distributed.broadcast on a
 * tensor that aliases the block). */
int hgx_index_device_block(const hgx_index *ix, void **dev_block, size_t *bytes);
int hgx_index_create_device(hgx_index **out, int32_t n_alleles, int32_t n_vars);

/* ---- 8e: collectives on device buffers, for callers that hold an RCCL communicator -----------------------------------
 * `rccl_comm` is an ncclComm_t (librccl is loaded on first use: the library itself does not link against it); everything
 * runs on `stream`, nothing bounces through the host but a rank-count-sized table of sizes.
 *   hgx_index_broadcast   the index' device block [link bits | exon mask | gene mask] from rank `root` to every rank, in place
 *                         (receivers: an index of the same shape from hgx_index_create_device)
 *   hgx_allreduce_sum_*   element-wise sum of a device buffer over the ranks, in place (pileup counts: u32; totals / inter-
 *                         distance histogram: i64)
 *   hgx_classes_allgather the ranks' class tables of ONE level (mine may be NULL = no classes on this rank) gathered in rank
 *                         order -- which is stream order of their pairs, so first-seen order survives -- and merged
 *                         (hgx_dedup_classes with the counts as weights): every rank gets the class set of the whole sample */
int hgx_index_broadcast(hgx_index *ix, int32_t root, void *rccl_comm, void *stream);
int hgx_allreduce_sum_u32(uint32_t *dev_buf, size_t n, void *rccl_comm, void *stream);
int hgx_allreduce_sum_i64(int64_t *dev_buf, size_t n, void *rccl_comm, void *stream);
/* (hgx_classes_allgather is declared with the class-set functions below) */

/* ---- 8a-5 / 8a-6: read-pair x allele compatibility -> class bitsets --------------------
 * Replaces add_count (typing_core.py:626-677) + add_stat (core:1171-1236) for a batch of
 * pairs.  A *piece* is the argument of one add_count call with the haplotype string
 * "left-id-..-right" already reduced to bit masks over variant words [lo_word, lo_word+n_words):
 *   masks[mask_off + 2*i]     = MP word i : known variants overlapping [left,right] (core:651-670) OR'ed with P
 *   masks[mask_off + 2*i + 1] = P  word i : the piece's own known variants (core:642-647)
 * allele a is compatible  <=>  for every i: (link_bits[lo_word+i][a] & MP_i) == P_i.
 * Identical pieces recur thousands of times in a read set, so the caller passes the table of
 * DISTINCT pieces once and every pair refers to its pieces by index:
 *   pair_ref[k] = piece index | (level << 31),  k in [pair_off[p], pair_off[p+1])
 * (one ref per add_count call, duplicates allowed: they count twice, as in the reference).
 * Per pair and level: count[a] = #refs whose piece is compatible with a; class = {a in the
 * level's allele mask : count[a] == max count} (max including 0: quirk Q4, core:1177-1190);
 * at most 65535 refs per pair and level (counters as wide as the pair needs: 2 / 4 / 8 bit planes in registers, 16 for the rare
 * pair with more than 255 refs -- long STR alleles with many alternative alignments).
 * Outputs (device): class bitsets, one row of a_pad/64 uint64 per pair and level, and a 64-bit
 * content hash per row (all-zero row <=> hash 0xFFFFFFFFFFFFFFFF) for hgx_dedup_classes.      */
typedef struct hgx_piece {
    uint32_t mask_off;   /* index into masks[] (in uint32 units)          */
    uint16_t lo_word;    /* first 32-variant word covered                 */
    uint16_t n_words;    /* number of words covered (>= 1)                */
} hgx_piece;

/* stage 1: compat_dev[piece][a_pad/64] = bitset of alleles compatible with each distinct piece.  Any piece order
 * is correct; tables sorted by lo_word (hgx_parse_sam / hgx_batch_from_haplotypes emit them so) run fastest. */
int hgx_piece_compat(const hgx_index *ix, const hgx_piece *pieces_dev, const uint32_t *masks_dev,
                     int32_t n_pieces, uint64_t *compat_dev, void *stream);

/* stage 2: one wavefront per pair combines its pieces' rows into the two class rows */
int hgx_pair_classes(const hgx_index *ix, const uint64_t *compat_dev,
                     const int32_t *pair_off_dev,         /* [n_pairs + 1] into pair_ref       */
                     const uint32_t *pair_ref_dev, int32_t n_pairs,
                     uint64_t *exon_bits_dev,             /* [n_pairs][a_pad/64] or NULL       */
                     uint64_t *gene_bits_dev,             /* [n_pairs][a_pad/64] or NULL       */
                     uint64_t *exon_hash_dev,             /* [n_pairs] or NULL                 */
                     uint64_t *gene_hash_dev,             /* [n_pairs] or NULL                 */
                     void *stream);

/* both stages; compat_scratch_dev must hold n_pieces * a_pad/64 uint64 */
int hgx_score_pairs(const hgx_index *ix,
                    const hgx_piece *pieces_dev, const uint32_t *masks_dev, int32_t n_pieces,
                    const int32_t *pair_off_dev, const uint32_t *pair_ref_dev, int32_t n_pairs,
                    uint64_t *compat_scratch_dev,
                    uint64_t *exon_bits_dev, uint64_t *gene_bits_dev,
                    uint64_t *exon_hash_dev, uint64_t *gene_hash_dev,
                    void *stream);

/* ---- 8a-7: class dedup -------------------------------------------------------------------
 * Replaces the Gene_cmpt / Gene_exons_cmpt dict accumulation (typing_core.py:1229-1234) and the
 * Gene_cmpt2 filtering (core:1752-1766): rows (optionally AND'ed with and_mask, empty rows
 * dropped) are grouped by content; groups come out in FIRST-SEEN order (Python dict order)
 * with the summed weight.  Exact: every row is compared with its group's first row; rows that share a
 * 64-bit key with a different row are re-keyed and re-checked (HGX_ECOLLISION only if that fails 8 times,
 * or from the radix-sort form).  On any failure the constructors below hand out nothing: *out = NULL. */
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
/* 8e (see the collectives next to hgx_index_broadcast): every rank's class table of one level, gathered in rank order and merged */
int hgx_classes_allgather(hgx_classes **out, const hgx_classes *mine_or_null, int32_t a_pad, void *rccl_comm, void *stream);
/* The two halves of that exchange around the collective, for callers with a transport of their own (and for tests that assemble
 * the receive buffer of a world of N ranks on one GPU).  Wire format: a rank's table as `cap` rows of (a_pad/64 + 1) 64-bit
 * words, [class bits | pair count], zero rows behind its n_classes; the receive buffer = the ranks' blocks in rank order.
 *   hgx_classes_pack_rows       mine (NULL = none) -> dev_send [cap] rows, on `stream`
 *   hgx_classes_merge_gathered  dev_recv [world][cap] rows + sizes[world] (host) -> the merged class set in rank order = the
 *                               stream order of the ranks' pairs (first-seen order of typing_core.py:1229-1234 survives)
 * hgx_rccl_stats: collectives issued by this process through the entry points above and the bytes handed to / received from
 * them since the last reset (bench.py prints the exchange bytes of a step from it). */
int hgx_classes_pack_rows(const hgx_classes *mine_or_null, int32_t a_pad, int32_t cap, void *dev_send, void *stream);
int hgx_classes_merge_gathered(hgx_classes **out, const void *dev_recv, const int32_t *sizes_host, int32_t world, int32_t cap,
                               int32_t a_pad, void *stream);
int hgx_rccl_stats(uint64_t *n_collectives, uint64_t *bytes_sent, uint64_t *bytes_received, int32_t reset);

/* 8a-5/6 stage 2 and 8a-7 in one call for ONE level (0 = exon, 1 = gene): the classes of all pairs in first-seen order with
 * their pair counts and first pairs -- the same result as hgx_pair_classes + hgx_dedup_classes on that level's rows
 * (Gene_exons_cmpt / Gene_cmpt, typing_core.py:1177-1190, 1229-1234) -- without a class row per pair: pairs are first
 * grouped by their list of piece refs (exact, list against list), one row is computed per distinct list and the row
 * dedup runs on those with the group sizes as weights.  Deep coverage repeats ref lists (3.7x at the exon level of
 * HLA-A with 1 M reads), and the 896-byte rows are the dominant HBM traffic of the per-pair form.
 * rows_scratch_dev ([n_pairs][a_pad/64]) / hash_scratch_dev ([n_pairs]) may be NULL (allocated from the pool). */
int hgx_level_classes(hgx_classes **out, const hgx_index *ix, const uint64_t *compat_dev,
                      const int32_t *pair_off_dev, const uint32_t *pair_ref_dev, int32_t n_pairs, int32_t level,
                      uint64_t *rows_scratch_dev, uint64_t *hash_scratch_dev, void *stream);
/* The same in two steps.  The grouping needs only the pair -> ref lists, not the piece bitsets: queued on its own stream it
 * runs beside hgx_piece_compat.  hgx_group_pairs only QUEUES the work; the first hgx_groups_dims /
 * hgx_level_classes_grouped call on the same host thread waits for it (one round trip on that stream).  The second step
 * must be ordered behind hgx_piece_compat; `groups` must stay alive until the returned class set has been consumed on the
 * device (destroy it after the EM / hgx_classes_to_host, or after synchronising the stream). */
typedef struct hgx_groups hgx_groups;
int hgx_group_pairs(hgx_groups **out, const int32_t *pair_off_dev, const uint32_t *pair_ref_dev, int32_t n_pairs,
                    int32_t level, void *stream);
int hgx_groups_dims(hgx_groups *g, int64_t *n_groups, int32_t *n_pairs);
int hgx_groups_destroy(hgx_groups *g);
int hgx_level_classes_grouped(hgx_classes **out, const hgx_index *ix, const uint64_t *compat_dev,
                              const int32_t *pair_off_dev, const uint32_t *pair_ref_dev, hgx_groups *groups,
                              uint64_t *rows_scratch_dev, uint64_t *hash_scratch_dev, void *stream);

/* hgx_pair_classes + hgx_dedup_classes for ONE level without a class row per pair in memory: the wavefront that computes a
 * pair's row (in registers) claims the row's slot in the class table itself -- the first row of a class is stored and published
 * as the class' representative, every later one is compared with it word for word (the exact check) and never stored.
 * rows_scratch_dev: [n_pairs][a_pad/64] (only representatives are written).  Same class set, counts, first pairs and order as
 * the two-call form; a 64-bit key shared by DIFFERENT rows (never seen on real data) returns HGX_ECOLLISION and the caller uses
 * the two-call form, which resolves collisions.  LAB LIBRARY ONLY (libhgx_lab.so: measured slower than the two-call form, kept for
 * comparison): libhgx.so exports the symbol and returns HGX_EINVAL. */
int hgx_pair_classes_dedup(hgx_classes **out, const hgx_index *ix, const uint64_t *compat_dev, const int32_t *pair_off_dev,
                           const uint32_t *pair_ref_dev, int32_t n_pairs, int32_t level, uint64_t *rows_scratch_dev, void *stream);

/* Gene_counts (typing_core.py:1187-1190, 1650-1651): per allele the number of pairs whose
 * class contains it, and the index of the first class (in first-seen order) containing it
 * (-1 if none) -- that is the dict insertion order used to break count ties.             */
int hgx_allele_counts(const hgx_classes *c, int64_t *count_host, int32_t *first_class_host);
int hgx_allele_counts_on(const hgx_classes *c, int64_t *count_host, int32_t *first_class_host, void *stream);
/* first class (dict order) containing each of a FEW given alleles, -1 if none: the tie order of the EM's result list
 * (common:1300-1305) without the full Gene_counts pass */
int hgx_first_classes(const hgx_classes *c, const int32_t *alleles_host, int32_t n, int32_t *first_class_host, void *stream);

/* ---- 8b: the dict side of the single_abundance seam ----------------------------------------------------------------------
 * single_abundance(Gene_cmpt) takes classes as strings, '-'.join(sorted(allele names)) -> count (typing_common.py:1282-1305,
 * typing_core.py:1229-1230).  hgx_keyset_create turns `n_keys` such keys (one per line: '\n'-separated, no trailing newline
 * needed) into what hgx_classes_from_host takes -- the allele names in order of first appearance (= the insertion order of the
 * reference's Gene_prob dict) and one bit row per key -- in one pass over the text instead of a Python loop per name.
 *   hgx_keyset_dims   n_names, a_pad (= hgx_a_pad(n_names)), bytes of the name pool, keys_sorted (every key's names ascending:
 *                     the form the reference itself builds, which lets the EM follow its summation order)
 *   hgx_keyset_fill   bits_host [n_keys][a_pad/64], name_pool ('\0'-terminated names, first-appearance order), name_rank [n_names]
 *                     (rank of each name in sorted order); any of them may be NULL */
typedef struct hgx_keyset hgx_keyset;
int hgx_keyset_create(hgx_keyset **out, const char *keys, size_t n_bytes, int32_t n_keys);
int hgx_keyset_dims(const hgx_keyset *ks, int32_t *n_names, int32_t *a_pad, size_t *name_pool_bytes, int32_t *keys_sorted);
int hgx_keyset_fill(const hgx_keyset *ks, uint64_t *bits_host, char *name_pool, int32_t *name_rank);
int hgx_keyset_destroy(hgx_keyset *ks);

/* ---- 8a-8: EM abundance ------------------------------------------------------------------
 * Replaces single_abundance (typing_common.py:1282-1410): SQUAREM-accelerated EM in FP64,
 * same step sequence, clamp, stopping rule (diff > 1e-4, < 1000 iterations) and pruning
 * (>= max/10 from iteration 10 when remove_low != 0).  allele_len_or_null != NULL selects the
 * length-normalised variant.  prob_host[a] is the abundance, or -1.0 if allele a is not in the
 * returned dict.  n_iter_host receives the number of outer iterations.                     */
int hgx_em(const hgx_classes *c, int32_t n_alleles,
           int32_t remove_low, const int32_t *allele_len_or_null /* host [n_alleles] */,
           double *prob_host, int32_t *n_iter_host, void *stream);
/* same, and first_class_host[a] = first class (dict order) containing allele a for the alleles in the returned dict
 * (-1 elsewhere): with name order this is the insertion order of the reference's result dict (common:1300-1305),
 * i.e. the tie order of its final stable sort -- delivered with the abundances, no second pass or sync */
int hgx_em_ordered(const hgx_classes *c, int32_t n_alleles, int32_t remove_low, const int32_t *allele_len_or_null,
                   double *prob_host, int32_t *first_class_host, int32_t *n_iter_host, void *stream);

/* Name order of the alleles: rank_host[a] = position of allele a's name among the sorted names (= its place inside a class
 * key, '-'.join(sorted(names)), typing_core.py:1229).  With it the single-wavefront EMs (hgx_em_masked; hgx_em / hgx_em_ordered
 * on <= 64 classes over <= 64 alleles) add in the reference's own order (dict order of classes, key order of alleles, terms
 * count*prob/alleles_prob, no contraction) and return bit-identical abundances; without it the result is within 1e-9 as
 * everywhere else.  The array is copied; NULL clears it. */
int hgx_classes_set_allele_rank(hgx_classes *c, const int32_t *rank_host, int32_t n);
/* 1 if the calling thread's last hgx_em / hgx_em_ordered / hgx_em_masked call took that path: its abundances are then the
 * reference's doubles, and ties between them are the reference's ties (callers sort them with a plain stable sort instead
 * of the tolerance they need for results that agree to ~1e-14 only) */
int hgx_em_last_exact(void);
/* The insertion order of the dict the calling thread's last hgx_em / hgx_em_ordered returned -- the tie order of the reference's
 * final stable sort.  It is (first WALKED class containing the allele, name order) in the LAST Gene_prob_next: classes whose
 * alleles_prob is 0 are skipped (typing_common.py:1321), so it can differ from (first class over all classes, name order), which
 * is what first_class_host gives.  Available (return 1) after an EM of more than 64 classes or alleles that ran in the reference's
 * own order (hgx_em_last_exact); order_host[a] = position, -1 for alleles outside the dict.  Returns 0 otherwise. */
int hgx_em_last_order(int32_t *order_host, int32_t n_alleles);
/* arithmetic of hgx_em / hgx_em_ordered on this thread for problems the one-workgroup kernel takes (see hgx_type_opts.em_fast):
 * 0 = the reference's order (default), 1 = table lookups, -1 = the reference's order at EVERY size (k_emx also beyond 4096
 * classes, up to 32768; a lone problem of that size runs on a cluster of workgroups: ~45 ms for 16 000 classes x 4 500 alleles, where
 * the default table-lookup path takes 1 ms).  Returns the previous setting. */
int hgx_em_set_fast(int on);

/* The exon -> gene hand-off in one call (typing_core.py:1752-1782): Gene_cmpt2 = every class of `c` filtered to the alleles of
 * mask_host (a_pad/64 words), empty ones dropped, equal ones merged with summed counts; then the EM on it (as
 * hgx_em_ordered).  *n_classes_host = number of merged classes.  With <= 64 alleles in the mask (and <= 64 merged classes)
 * filter, merge and EM run in ONE launch on one wavefront; otherwise = hgx_dedup_classes(and_mask, weights) + hgx_em_ordered. */
int hgx_em_masked(const hgx_classes *c, const uint64_t *mask_host, int32_t n_alleles, int32_t remove_low,
                  const int32_t *allele_len_or_null, double *prob_host, int32_t *first_class_host, int32_t *n_iter_host,
                  int32_t *n_classes_host, void *stream);

/* ---- host front-end: locus tables, SAM -> pieces (8a-0 .. 8a-4) ----------------------------
 * The part of typing() that precedes scoring is index-heavy string logic with no data
 * parallelism per record; it runs on the host in C++ and hands the device the distinct-piece
 * table plus per-pair piece references.                                                      */
#define HGX_BASE_HLA    0
#define HGX_BASE_CODIS  1
#define HGX_BASE_GENOME 2
#define HGX_BASE_OTHER  3

typedef struct hgx_locus_desc {
    int32_t base_kind;               /* HGX_BASE_*  (base_fname of the reference)                        */
    int32_t backbone_len;
    const char *backbone;            /* ref_seq = Genes[gene][ref_allele]                  core:386       */
    int32_t n_vars;                  /* gene_var_list order                                                */
    const int32_t *var_pos;
    const uint8_t *var_type;         /* HGX_VAR_*                                                          */
    const int32_t *var_len;          /* deletion length / insertion length / 1                             */
    const char *var_base;            /* alt base of a single, 0 otherwise                                  */
    const uint8_t *var_linked;       /* var_id in Links                                                    */
    const char *var_name_pool;       /* ids ("hv123"), '\0' separated, var order                           */
    const char *var_ins_pool;        /* inserted bases of insertions, '\0' separated, var order ("" else)  */
    int32_t n_alleles;               /* Gene_names[gene] minus the backbone, that order                    */
    const int32_t *link_off;         /* CSR [n_vars+1]: Links[var] -> allele indices (file order)          */
    const int32_t *link_allele;
    int32_t n_link_order;            /* variant indices in Links dict order (rep selection, core:86-115)  */
    const int32_t *link_order;
    int32_t n_exons;
    const int32_t *exons;            /* [n_exons][2] inclusive                                             */
    const int32_t *allele_len;       /* Gene_lengths                                                       */
    const int32_t *name_rank;        /* rank of each allele name in sorted(name) order                    */
} hgx_locus_desc;

typedef struct hgx_locus hgx_locus;
int hgx_locus_create(hgx_locus **out, const hgx_locus_desc *desc);
int hgx_locus_destroy(hgx_locus *loc);
/* derived tables: a_pad, n_words; rep_of[a] = representative allele of a's exon group or -1
 * (get_rep_alleles, core:86-115); masks are a_pad/64 words; link_bits is [n_words][a_pad]      */
int hgx_locus_dims(const hgx_locus *loc, int32_t *n_alleles, int32_t *a_pad, int32_t *n_vars, int32_t *n_words);
int hgx_locus_tables(const hgx_locus *loc, uint32_t *link_bits, uint64_t *exon_mask, uint64_t *gene_mask,
                     int32_t *rep_of);
int hgx_index_from_locus(hgx_index **out, const hgx_locus *loc);
/* number of entries / dump of the alternatives tables (get_alternatives, common:1424-1657) as text
 * "L\tkey\talt\n" / "R\tkey\talt\n" lines -- test and debugging aid                              */
int hgx_locus_alternatives_text(const hgx_locus *loc, char *buf, size_t cap, size_t *needed);

/* A batch = what one locus' read stream reduces to: distinct pieces + masks, and per pair the
 * piece refs (exon-level refs and gene-level refs, bit 31 = level).                           */
typedef struct hgx_batch hgx_batch;
int hgx_batch_destroy(hgx_batch *b);
int hgx_batch_dims(const hgx_batch *b, int32_t *n_pieces, int64_t *n_mask_u32, int32_t *n_pairs, int64_t *n_refs,
                   int32_t *n_reads);
int hgx_batch_arrays(const hgx_batch *b, const hgx_piece **pieces, const uint32_t **masks, const int32_t **pair_off,
                     const uint32_t **pair_ref);

/* integer haplotypes -> batch.  Piece q of pair p (q in [pair_off[p], pair_off[p+1])) is the
 * add_count argument "left-ids-right" with ids = variant indices (-1 for nv / unknown ids).    */
int hgx_batch_from_haplotypes(hgx_batch **out, const hgx_locus *loc, int32_t n_pairs, const int32_t *pair_off,
                              const uint8_t *piece_level, const int32_t *piece_left, const int32_t *piece_right,
                              const int32_t *piece_id_off, const int32_t *piece_ids);

typedef struct hgx_parse_opts {
    int32_t num_editdist;       /* --num-editdist, default 2                      args:294 */
    int32_t error_correction;   /* default 1                                      args:324 */
    int32_t allow_discordant;   /* default 0 (forced 1 for single-end)            args:334 */
    int32_t simulation;         /* read id = QNAME up to the first '|'            core:808 */
    int32_t base_locus;         /* subtracted from POS                            core:814 */
    int32_t keep_trace;         /* record per-read intermediates for hgx_batch_trace_text   */
    int32_t codis_choose_pairs; /* base codis && gene == "D18S51": choose_pairs at the final flush (core:1547-1552) */
    int32_t n_threads;          /* host threads for the front-end; 0 = hardware threads, capped at 2x the cgroup CPU quota */
    /* Intra-locus read sharding (8e): when ONE sample's reads of a locus are split over several ranks, error correction still
     * needs the pileup of ALL reads (get_mpileup runs over the whole alignment, common:1059-1134).  If set, the callback is
     * invoked once with this shard's counts[L][6] (A,C,G,T,N,D per backbone position) and must return with the element-wise
     * SUM over all shards in place (an all-reduce; dist.py does it with torch.distributed); non-zero return = failure.      */
    int (*pileup_exchange)(void *ctx, uint32_t *counts, int64_t n_cells);
    void *pileup_ctx;
    /* The same for CODIS D18S51 (codis_choose_pairs): choose_pairs needs the MEDIAN inner distance of the whole sample's unique
     * concordant pairs (get_pair_interdist, typing_common.py:1187-1265).  If set, the callback is invoked once with this
     * shard's histogram of the distances -- HGX_INTERDIST_BINS int64 counters: bin 0 = below -HGX_INTERDIST_HALF, bin 1 + d +
     * HGX_INTERDIST_HALF = distance d, the last bin = above -- and must return with the element-wise sum over all shards in
     * place; the median is then read from the summed histogram (exact; outside the range the parse fails).  The device front end
     * counts the distances as kernels (k_fe_interdist_*) and calls this after its pileup exchange -- the order of the host stages;
     * a route that declines after the exchange hands the summed histogram to the host stages: one exchange per parse.        */
    int (*interdist_exchange)(void *ctx, int64_t *hist, int64_t n_bins);
    void *interdist_ctx;
    /* The DEVICE form of pileup_exchange, taken by the device front end (hgx_parse_sam_dev / hgx_parse_alignment_file_dev) when
     * set: called once, between the pileup kernels and the nt_set kernel, with this shard's counters where the kernels left them
     * -- device memory, n = L*6 counters plus ONE spare element that is zero on entry (room for a caller's failure flag) -- and must
     * leave the element-wise sum over all shards there, complete or ordered on `stream` when it returns (hgx_allreduce_sum_u32 on
     * the same stream does exactly that; dist.py: RCCL, torch.distributed on an aliasing tensor, or threads of one process).
     * The host form above must be given too: it is what runs if the device route declines BEFORE its exchange (a small input, a
     * record the kernels do not take); both forms must be the same collective -- uint32, n elements -- because other ranks may
     * be on the other route.  A route that declines AFTER the exchange hands the summed table to the host stages, which then
     * do not exchange again: every rank communicates exactly once per parse.                                                   */
    int (*pileup_exchange_dev)(void *ctx, uint32_t *d_counts, int64_t n, void *stream);
    void *pileup_dev_ctx;
} hgx_parse_opts;
#define HGX_INTERDIST_HALF 65536
#define HGX_INTERDIST_BINS (2 * HGX_INTERDIST_HALF + 2)

/* SAM text (name-grouped, i.e. the stream after `sort -k1,1 -s`, core:458-468) -> batch.
 * Replaces typing_core.py:800-1406 + get_mpileup (common:1059-1134).                        */
int hgx_parse_sam(hgx_batch **out, const hgx_locus *loc, const char *sam, size_t n_bytes, const hgx_parse_opts *opts);

/* ---- 8f-3: alignment ingestion without samtools --------------------------------------------------------
 * The record stream the reference's loop consumes -- `samtools view <file> [chr:left-right] <ref_allele>` piped through
 * `sort -k1,1 -s` (typing_core.py:436-468) -- from a SAM text file or a BAM file: BGZF blocks inflated in parallel (zlib),
 * BAM records decoded to SAM text (mandatory fields + tags A c C s S i I f Z H B), header lines dropped, then grouped by a
 * STABLE bytewise sort on QNAME.
 * `regions_or_null`: NULL or "" = every record; otherwise one or more regions in samtools syntax separated by '\n' -- "name"
 * (the whole reference sequence) or "name:left-right" (1-based, inclusive; "name:left" and "name:-right" too).  As with
 * `samtools view file r1 r2`, the records that OVERLAP a region (reference span taken from the CIGAR; an unmapped or
 * zero-length record counts as one base at POS) come out region after region, in file order inside a region, before the name
 * sort; a string that names a reference as a whole wins over its "name:span" reading (HLA contig names contain ':').  The
 * reference always passes the locus backbone (`alignview_cmd += [ref_allele]`, core:443-444), and in genotype-genome mode
 * the locus span in front of it (core:438-441): typing() here does the same, so reads of other loci in a multi-locus
 * alignment never reach this locus' decode.
 * *text_out is a library-owned (pooled), NUL-terminated buffer of *n_bytes_out bytes (every record ends
 * in '\n'), ready for hgx_parse_sam; release it with hgx_free_text (not free()).  n_threads <= 0: the host's hardware threads, capped at twice the container's cgroup CPU quota if it has one (HGX_THREADS overrides). */
int hgx_read_alignments(const char *path, const char *regions_or_null, int32_t n_threads, char **text_out, size_t *n_bytes_out);
int hgx_free_text(char *text);
/* hgx_read_alignments + hgx_parse_sam in one call: the reader's buffer is tokenised in place (no copy, no trip through the
 * caller) -- the whole host side from an alignment file to the piece batch */
int hgx_parse_alignment_file(hgx_batch **out, const hgx_locus *loc, const char *path, const char *regions_or_null,
                             const hgx_parse_opts *opts);
/* SAM text (records; '@' lines skipped) -> BAM file: records encoded and BGZF blocks deflated in parallel.  ref_names = the
 * reference sequence names separated by '\n' (n_refs of them, with ref_lens).  sort_by_coordinate != 0 orders the records by
 * (reference, position) first, as `samtools sort` leaves the alignment the reference's pipeline stores
 * (typing_common.py:1041-1050).  Tags: A i f Z H B. */
int hgx_write_bam(const char *path, const char *sam, size_t n_bytes, const char *ref_names, const int32_t *ref_lens, int32_t n_refs,
                  int32_t sort_by_coordinate, int32_t n_threads);
/* per kept record: "cmp_list2 \t cmp_left \t cmp_right \t left alts \t right alts" (keep_trace) */
int hgx_batch_trace_text(const hgx_batch *b, char *buf, size_t cap, size_t *needed);
/* pileup nt_set per backbone position as a 4-bit mask A=1,C=2,G=4,T=8 and counts[L][6] (A,C,G,T,N,D) */
int hgx_batch_pileup(const hgx_batch *b, uint8_t *nt_set, uint32_t *counts);

/* ---- the per-locus body of typing() in one call --------------------------------------------------------------------
 * Replaces typing_core.py:1589-1789 for one locus and one sample: Gene_counts and their print order (core:1187-1190,
 * 1650-1651), the exon-level classes and EM #1 (core:1732-1737), the choice of exon_alleles (core:1739-1749), Gene_cmpt2 +
 * EM #2 with allele lengths (core:1752-1782), the combination of both results, and the tie orders of the reference's stable
 * sorts.  Non-HLA bases (codis, ...): gene level only, EM without pruning (core:1784-1789); ONE class there returns
 * HGX_ETYPE, none an empty result.  The call only orchestrates the device entry points above -- on the caller's stream plus a
 * recycled pair of side streams and, for >= 4096 pairs of an HLA-like locus, a host thread for the gene-level side that
 * runs beside the exon-level EM -- and returns when the result is on the host.  `loc` must be the locus `ix` was made from. */
typedef struct hgx_typing hgx_typing;     /* result of one (sample, locus) */
typedef struct hgx_dbatch hgx_dbatch;     /* a piece batch resident in HBM */
typedef struct hgx_gate hgx_gate;         /* see hgx_type_opts.gate */
typedef struct hgx_type_opts {
    int32_t remove_low;       /* remove_low_abundance_alleles for EM #1 of an HLA-like locus (args: --keep-low-abundance-alleles clears it) */
    int32_t keep_classes;     /* keep the exon / gene class sets in the result (hgx_typing_classes)                     */
    int32_t overlap;          /* -1: overlap the gene side when stream == NULL; 0 / 1: never / always (if >= 4096 pairs)  */
    int32_t per_pair_exon;    /* exon-level rows per pair + dedup instead of per distinct ref list (same result; tests)   */
    hgx_gate *gate;           /* several samples in flight on one GPU: held from entry until this sample's exon-level
                                 classes exist, so that ONE bandwidth-bound front runs at a time, beside the others' EM phases */
    void *ev_compat_begin, *ev_compat_end;   /* optional hipEvent_t recorded around hgx_piece_compat                      */
    void *ev_pairs_begin, *ev_pairs_end;     /* ... and around the gene-level hgx_pair_classes launch (bench.py)           */
    int32_t em_fast;          /* arithmetic of EM #1 for problems of up to 4096 classes x 8192 alleles (the one-workgroup kernel, hgx_emx.hip):
                                   0  the entry point's default: the ONE-TASK calls (hgx_type_dbatch / _batch / _file / _classes) run the
                                      reference's own order of floating-point operations -- abundances, pruning and stopping decisions
                                      bit-identical to typing_common.py:1282-1410; the MANY-TASK calls (hgx_type_many / _many_loci: the
                                      throughput API) run table-lookup arithmetic on the same kernel: ~3x faster per panel, abundances
                                      within 1e-8 of the reference (typically 1e-11; bar 1e-5), same stopping and pruning rules;
                                   1  table lookups in the one-task calls too;
                                   2  the reference's order in the many-task calls too (bit-identical; the hand-off EM always is);
                                  -1  the reference's order at EVERY size (k_emx up to 32768 classes, in cluster mode for a lone large
                                      problem: 33 ms per step for a 1 M-read sample with a 16 000-class EM #1 instead of 2 ms).
                                 Larger problems (EM #1 of a deep sample) take the chip-wide table-lookup path (<= 1e-9) unless -1. */
} hgx_type_opts;

int hgx_dbatch_create(hgx_dbatch **out, const hgx_batch *b, void *stream);     /* upload; returns when the copy is complete */
int hgx_dbatch_destroy(hgx_dbatch *d);
int hgx_dbatch_dims(const hgx_dbatch *d, int32_t *n_pieces, int32_t *n_pairs, int64_t *n_refs, int32_t *n_reads,
                    int64_t *sum_piece_words, int64_t *n_gene_refs);
int hgx_gate_create(hgx_gate **g);
int hgx_gate_destroy(hgx_gate *g);

int hgx_type_dbatch(hgx_typing **out, const hgx_locus *loc, const hgx_index *ix, const hgx_dbatch *db,
                    const hgx_type_opts *opts, void *stream);
int hgx_type_batch(hgx_typing **out, const hgx_locus *loc, const hgx_index *ix, const hgx_batch *batch,
                   const hgx_type_opts *opts, void *stream);                   /* upload + hgx_type_dbatch */
/* alignment file -> typing result: hgx_parse_alignment_file + hgx_type_batch (the whole of typing()'s per-locus work) */
int hgx_type_file(hgx_typing **out, const hgx_locus *loc, const hgx_index *ix, const char *path, const char *regions_or_null,
                  const hgx_parse_opts *parse_opts, const hgx_type_opts *opts, void *stream);
/* ---- the front end on the device (rows 8a-1 .. 8a-5 as kernels, csrc/hgx_front.hip) ----------------------------------------
 * RECORD route (tried first): the host only reads the file -- and, for BAM, inflates it, walks the record chain and name-sorts it;
 * the SAM text / inflated stream goes to the GPU while that happens, and the record fields (typing_core.py:800-841), the record
 * filters (typing_core.py:815-872) and the grouping of the records by decode key run as kernels, one lane per record.
 * KEY route (when the record route declines: a line with blanks, a float-typed NM tag, ...): the host tokenises, filters and groups;
 * the distinct keys (~0.28 per read at 1 M reads, ~200 bytes each) are uploaded.
 * Either way get_mpileup (typing_common.py:1059-1134), the CIGAR x MD x Zs walk + error_correct (typing_core.py:899-1124,
 * 119-243), identify_ambigious_diffs (typing_common.py:1663-1955), haplotype assembly, get_exon_haplotypes
 * (typing_core.py:1386-1406, 718-792), the piece masks of add_count's span scan (typing_core.py:641-670), the distinct-piece
 * table and the pair protocol (typing_core.py:1238-1347) run as kernels; the batch is born in HBM, byte for byte the batch
 * hgx_parse_sam / hgx_parse_alignment_file build (hgx_dbatch_to_host shows it).  Inputs the kernels do not take -- fewer than
 * 1 000 records or 300 KB of stream (the measured break-even with the host stages: csrc/hgx_front.hip FE_MIN_*), keep_trace in a many-task pass, variant ids that are not hv<n>, a record the reference would raise on,
 * a pair with more alternatives than the kernels' scratch holds -- are finished by the host stages and uploaded: the result is
 * the same batch either way, and hgx_front_last says which way the calling thread's last call went (route: 2 = record route,
 * 1 = key route, 0 = host stages; decline_code: see HGX_FE_DECLINE_* / FE_E_* in csrc/hgx_internal.hpp,
 * csrc/hgx_front_core.hpp).  hgx_type_file goes through hgx_parse_alignment_file_dev.
 * A BAM file (with at most one region) does not even get inflated on the host: the host reads it, hops through the BGZF container
 * and inflates the block(s) holding the BAM header; the deflated bytes go up, and BGZF inflate (csrc/hgx_inflate.hip), the record
 * chain walk, the region filter (samtools' overlap rule, typing_core.py:438-444) and the stable sort by read name
 * (`sort -k 1,1 -s`, typing_core.py:436-468) run as kernels in front of the record route.  A block that fails CRC-32 / ISIZE, a
 * chain that does not link up or a malformed record sends the call to the host reader, which words the error. */
int hgx_parse_sam_dev(hgx_dbatch **out, const hgx_locus *loc, const char *sam, size_t n_bytes, const hgx_parse_opts *opts, void *stream);
int hgx_parse_alignment_file_dev(hgx_dbatch **out, const hgx_locus *loc, const char *path, const char *regions_or_null,
                                 const hgx_parse_opts *opts, void *stream);
int hgx_front_last(int32_t *route, int32_t *decline_code, int64_t *bytes_to_device /* sent by that call: text / stream / key table */);
/* 2 or 3 = that call took a big SAM text in that many parts -- the line table and the record fields of the phases that had landed
 * beside the later phases' transfer (csrc/hgx_front.hip records_split) --, 0 = every kernel behind the last byte. */
int hgx_front_last_parts(int32_t *parts);
/* ONE alignment file, SEVERAL loci (typing_core.py:370 loops `locus_list` over one alignment file; the reference runs `samtools view
 * F ref_allele | sort` per locus, core:436-468 -- the file is decompressed once per locus).  hgx_alignment_open reads the file ONCE and
 * leaves its bytes in HBM: the SAM text, or the BAM stream inflated on the device; hgx_alignment_parse_dev is the per-locus rest --
 * region filter, name order and the record route of hgx_parse_alignment_file_dev as kernels over those resident bytes (read-only: the
 * loci of a panel may be parsed side by side from threads with streams of their own).  The batch is the one
 * hgx_parse_alignment_file_dev(path, regions) builds; a locus the kernels decline, a region list with several entries, a file below
 * the device front end's size gate or beyond 4 GB of stream go through exactly that call on the path (hgx_front_last tells).
 * hgx_alignment_dims: resident = the bytes are in HBM (0: every locus takes the per-path call). */
typedef struct hgx_alignment hgx_alignment;
int hgx_alignment_open(hgx_alignment **out, const char *path, int32_t n_threads, void *stream);
int hgx_alignment_dims(const hgx_alignment *al, int32_t *resident, int32_t *is_text, size_t *stream_bytes, long long *bytes_to_device);
int hgx_alignment_parse_dev(hgx_dbatch **out, hgx_alignment *al, const hgx_locus *loc, const char *regions_or_null,
                            const hgx_parse_opts *opts, void *stream);
int hgx_alignment_close(hgx_alignment *al);
/* a device batch back on the host (tests, tools): pieces, masks, refs, and the pileup tables if the kernels made them */
int hgx_dbatch_to_host(const hgx_dbatch *d, hgx_batch **out);
/* The same from class sets that already exist (intra-locus read sharding, 8e: every rank scores its share of the pairs, the
 * ranks' class tables are gathered, concatenated in rank order and merged with hgx_dedup_classes(weights = counts), which
 * keeps first-seen order): Gene_counts + ranking from `gene_classes`, EM #1 on `exon_classes` (NULL for non-HLA bases: EM on
 * the gene classes), hand-off and EM #2.  n_reads / n_pairs are reported as given. */
int hgx_type_classes(hgx_typing **out, const hgx_locus *loc, hgx_classes *exon_classes_or_null, hgx_classes *gene_classes,
                     int32_t n_reads, int32_t n_pairs, const hgx_type_opts *opts, void *stream);
int hgx_typing_destroy(hgx_typing *t);
int hgx_typing_dims(const hgx_typing *t, int32_t *n_reads, int32_t *n_pairs, int32_t *n_pieces, int64_t *n_refs,
                    int32_t *n_counted, int32_t *n_em, int32_t *n_gene_prob, double *em_seconds);
/* ranked_allele[n_counted]: alleles with a non-zero count in the reference's print order; count_per_allele[n_alleles] */
int hgx_typing_counts(const hgx_typing *t, int32_t *ranked_allele, int64_t *count_per_allele);
/* the k-th single_abundance call: its flags and its result list (allele index, abundance) in the reference's order */
int hgx_typing_em(const hgx_typing *t, int32_t k, int32_t *n_classes, int32_t *n_iter, int32_t *remove_low, int32_t *use_length,
                  int32_t *n_result, int32_t *allele, double *prob);
/* final Gene_prob (core:1732-1789): n_gene_prob entries */
int hgx_typing_gene_prob(const hgx_typing *t, int32_t *allele, double *prob);
/* the calls of MANY results in one call: per result its reads, the number of single_abundance calls and the first k entries of
 * the final Gene_prob (allele -1 / abundance 0.0 beyond the end of a shorter list); a NULL result reads as empty */
int hgx_typing_top(const hgx_typing *const *ts, int32_t n, int32_t k, int32_t *n_reads, int32_t *n_em, int32_t *allele /* [n][k] */,
                   double *prob /* [n][k] */);
/* with keep_classes: the class sets behind the result (owned by `t`; NULL if that level was not built) */
int hgx_typing_classes(const hgx_typing *t, int32_t level, const hgx_classes **out);

/* ---- many tasks of ONE locus behind one launch chain ------------------------------------------------------------------
 * The reference's unit of scale is many samples x loci: one genotyping_locus per sample through a process pool
 * (/root/reference/hisatgenotype:613-665) and the locus loop inside typing() (typing_core.py:370).  hgx_type_many types every
 * (sample, locus) task of a locus at once: the tasks' piece batches are merged (hgx_many_create: pairs task after task,
 * distinct pieces interned across tasks), scored and de-duplicated by the kernels of the one-task path with the dedup keeping
 * tasks apart, and Gene_counts, EM #1, the hand-off and EM #2 run with a task dimension -- the launch count no longer grows with
 * the number of tasks.  out[t] = the result of task t (hgx_typing_* accessors; destroy each with hgx_typing_destroy), identical
 * to hgx_type_batch on task t's batch alone.  rc_out (may be NULL): per-task status -- HGX_ETYPE / HGX_EKEY where the reference
 * would raise on THAT task (out[t] = NULL then); with rc_out NULL such a task fails the whole call.  hgx_type_opts: remove_low
 * and em_fast are honoured; class sets are not kept per task.  One call at a time per hgx_many (its staging memory is reused). */
typedef struct hgx_many hgx_many;
int hgx_many_create(hgx_many **out, const hgx_locus *loc, const hgx_batch *const *batches, int32_t n_tasks, void *stream);
/* The same batch straight from the tasks' alignment streams -- files (SAM text or BAM; regions[t] as in
 * hgx_parse_alignment_file, `regions` may be NULL) or name-grouped SAM texts in memory: the samples of a panel at one locus in
 * ONE pass of the device front end (the tasks' files are read side by side on the host's threads, their bytes land in one device
 * buffer, every record carries its task: keys, read ids and pairs never cross tasks, each task has its own pileup for the error
 * correction, the piece table is shared).  The batch is the one hgx_many_create makes of hgx_parse_alignment_file's per-task
 * batches, array for array; where the device front end declines (hgx_front_last) exactly that is done instead.
 * The reference's call site: one hisatgenotype_locus process per sample through the pool (/root/reference/hisatgenotype:613-665),
 * each reading its own alignment file (typing_core.py:826-897).                                                               */
int hgx_many_create_files(hgx_many **out, const hgx_locus *loc, const char *const *paths, const char *const *regions /* or NULL */,
                          int32_t n_tasks, const hgx_parse_opts *opts, void *stream);
int hgx_many_create_sams(hgx_many **out, const hgx_locus *loc, const char *const *sams, const size_t *n_bytes, int32_t n_tasks,
                         const hgx_parse_opts *opts, void *stream);
/* ... or ONE task from a batch that is already resident (hgx_parse_sam_dev / hgx_parse_alignment_file_dev / hgx_alignment_parse_dev):
 * the loci of one sample typed together by hgx_type_many_loci (typing_core.py:370).  The hgx_many takes `db` over on success. */
int hgx_many_from_dbatch(hgx_many **out, const hgx_locus *loc, hgx_dbatch *db, void *stream);
/* the merged device batch of `m` (owned by it) and the tasks' extents: pair_base [n_tasks + 1], the others [n_tasks]; any may be NULL */
int hgx_many_tasks(const hgx_many *m, const hgx_dbatch **db, int32_t *pair_base, int32_t *n_reads, int32_t *n_pieces, int64_t *n_refs);
int hgx_many_destroy(hgx_many *m);
int hgx_many_dims(const hgx_many *m, int32_t *n_tasks, int32_t *n_distinct_pieces, int32_t *n_pairs, int64_t *n_refs, int64_t *n_reads);
int hgx_type_many(hgx_typing **out /* [n_tasks] */, int32_t *rc_out /* [n_tasks] or NULL */, const hgx_locus *loc, const hgx_index *ix,
                  hgx_many *m, const hgx_type_opts *opts, void *stream);
/* Several loci at once (a whole panel): out[i] / rc_out[i] are locus i's arrays as in hgx_type_many.  The loci are scored side by
 * side (one host thread and stream pair per locus), and the EMs of ALL their tasks go out in one launch (one workgroup per task,
 * longest problem first), so the launch is as wide as the panel and ends with its longest problem.                              */
int hgx_type_many_loci(int32_t n_loci, hgx_typing ***out, int32_t **rc_out_or_null, const hgx_locus *const *loci,
                       const hgx_index *const *ixs, hgx_many *const *manies, const hgx_type_opts *opts, void *stream);

/* BGZF inflate on the device (SAM/BAM specification section 4.1; row 8f-3): a whole BGZF file in host memory -> its payload in host
 * memory.  One wavefront per BGZF block decodes the block's DEFLATE stream (stored / fixed / dynamic Huffman) and checks CRC-32
 * and ISIZE; *bad_blocks = blocks that did not pass (the payload is not to be used then).  *n_out = payload bytes (also when
 * out_cap is too small: HGX_EINVAL).  What hgx_parse_alignment_file_dev / hgx_type_file do with a BAM goes through the same
 * kernel, the payload staying in HBM.  Replaces the reference's `samtools view` decompression (typing_core.py:436-468).          */
int hgx_bgzf_inflate(const void *bgzf, size_t n_bytes, void *out, size_t out_cap, size_t *n_out, int32_t *bad_blocks, void *stream);
/* Test entry, host only: the block chain of a BGZF file walked by one thread and in ranges by n_threads (what the reader does in front
 * of the device inflate: each range finds a block start and the ranges must link up, csrc/hgx_inflate.hip hgx_bgzf_scan_par).
 * *n_blocks = blocks of the file (-1: not a BGZF container), *same = 1 iff both walks gave the same verdict and descriptors.   */
int hgx_bgzf_scan_compare(const void *bgzf, size_t n_bytes, int32_t n_threads, int64_t *n_blocks, int32_t *same);

/* Kernel timing for roofline reports.  hgx_em_set_timing(1) makes hgx_em time a sample of its table-lookup mat-vec launches
 * (every 4th ungated rows pass and the cols pass after it), hgx_em_set_timing(2) every plain rows / cols pass, with events
 * attached to the dispatch itself (hipExtLaunchKernelGGL: kernel begin / end, as rocprofv3 measures).  Switching timing on
 * restarts the per-thread totals; 0 switches it off and keeps them.  Slots 0/1 = rows pass (vector of <= 8192 / more elements),
 * 2/3 = cols pass (k_lutmatvec<0> / <1> with the default backend).  `executed` counts the plain passes that ran while timing was
 * on (device-side counter) and bytes_total the algorithmic bytes of the timed ones (bit matrix once + dense vectors).       */
/* Test hook.  The library reads no path-selecting environment variable: the test-suite forces an alternative path (a kernel kept
 * for comparison, a threshold moved so a small case reaches the large-problem code) with named in-process switches.  name = NULL
 * clears all, value = NULL clears one.  The names are listed in DESIGN.md ("switches"); none changes results beyond what the
 * test that uses it states.  Environment variables the library does read: HGX_THREADS, HGX_PIN, HGX_THP, HGX_NO_LIBDEFLATE,
 * HGX_MALLOC_TUNE, HGX_READ_PHASES, HGX_PINNED_IDLE_MB (host tuning) and HGX_PARSE_PROFILE, HGX_TYPE_PROFILE (timing prints on stderr).                                            */
int hgx_test_switch_set(const char *name, const char *value);
/* mat-vec backend of hgx_em: 0 = auto (table lookup), 1 = EXEC-masked FP64 VALU kernel, 3 = table-lookup kernel (256 subset
 * sums per 8 matrix columns in LDS; one lookup per 8 matrix bits); 2 = int8 MFMA kernel, in the lab build only (libhgx_lab.so,
 * -DHGX_LAB: the MFMA, persistent and resident-grid EM back-ends measured in rounds 1-2 and kept out of the product library) */
int hgx_em_set_backend(int backend);
/* test aid: one rows pass (which = 0: y[c] = count[c] / sum_a B[c][a] x[a]) or cols pass (which = 1: y[a] = sum_c
 * B[c][a] x[c]) with backend 1 or 2; x and y are host arrays of a_pad / n_classes doubles                        */
int hgx_debug_matvec(const hgx_classes *c, int which, int backend, const double *x_host, double *y_host);
int hgx_em_set_timing(int on);
/* the one-workgroup EM kernel (k_emx: hgx_em / hgx_type_* on problems of up to 4096 classes, every EM of hgx_type_many): HIP
 * events around its launches while timing is on; totals per arithmetic (fast = 0: the reference's order, 1: table lookups):
 * kernel ms, launches, jobs, applications of the EM map, algorithmic bytes (per application C * A' / 8 + 16 A' + 16 C). */
int hgx_emx_set_timing(int on);
int hgx_emx_get_timing(int fast, double *ms_total, long long *launches, long long *jobs, long long *applications, long long *bytes_total);
/* em_fast = -1 runs a problem beyond 4096 classes on a CLUSTER of workgroups (several such problems of a call side by side, sharing
 * the chip's CUs).  A cluster whose workgroups are not co-resident in time (the chip is full of other work) gives up and the problem
 * is re-run on one workgroup -- same sums in the same orders, same bits, ~7x slower.  Counts since the library was loaded: problems
 * launched on a cluster, and those that fell back (also said on stderr under HGX_TYPE_PROFILE). */
int hgx_emx_cluster_stats(long long *cluster_problems, long long *fallbacks);
/* The stream sets of the current device (hgx_type.hip: an EM stream and a gene-side stream per caller in flight, placed on hardware
 * queues by MEASURED overlap, not by creation order): sets made so far, queue classes seen, probe launches and the time they took,
 * and (em class, gene class) of the sets free at the moment, in hand-out order (`classes`: 2 ints per set, `cap` ints of room). */
/* diagnostic: n fresh streams (high_prio[i] != 0: highest priority, else lowest); us[i][j] = microseconds two 150 us one-wavefront spin
 * kernels take when launched back to back on streams i and j (the diagonal: one kernel alone) -- what the placement measures */
/* a stream for a caller whose own work is a chain of short kernels (a worker thread's main stream: the device front end of its sample
 * or locus): created on the hardware lane with the fewest chains so far; destroy with hgx_stream_destroy */
int hgx_stream_create_placed(void **stream, int high_priority);
int hgx_stream_probe_matrix(int32_t n, const int32_t *high_prio, double *us);
int hgx_stream_probe_pair(void *stream_a, void *stream_b, int32_t *same_queue);          /* 1: one behind the other, 0: side by side */
/* a chain of 16 short kernels on `light`, alone (us[0]) and beside `other` (us[1]) running mode 0: one launch of 131 072 tiny workgroups,
 * mode 1: the same chain with interleaved launches -- two chains on one LANE take ~2.3x as long (what the placement measures) */
int hgx_stream_probe_chain(void *light_stream, void *other_stream, int32_t mode, double *us);
int hgx_stream_sets_streams(void **streams /* em, gene per free set */, int32_t cap, int32_t *n_sets);
int hgx_stream_sets_info(int32_t *n_sets, int32_t *n_classes, int32_t *n_probes, double *probe_ms, int32_t *classes, int32_t cap);
/* hgx_em / hgx_em_ordered calls (default arithmetic) whose table-lookup result held two alleles of DIFFERENT class membership closer
 * than 1e-8 relative and was therefore recomputed in the reference's own order of operations (common:1282-1410: a plain stable
 * sort on the reference's own doubles decides such an order), since the library was loaded.                                    */
long long hgx_em_tie_reruns(void);
int hgx_em_get_timing(int slot, double *ms_total, int64_t *launches, int64_t *executed, int64_t *bytes_total);

#ifdef __cplusplus
}
#endif
#endif /* HGX_H */
