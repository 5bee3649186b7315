// hgx_emx.hpp -- the batched EM in the reference's own order of floating-point operations (hgx_emx.hip), shared with the
// translation units that call it (hgx_em.hip: hgx_em / hgx_em_ordered for one class set; hgx_many.hip: hgx_type_many).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include <vector>

// one allele of a returned dict (what the kernel writes; hgx_emx_run hands these out as they are when asked to)
// `order`: the allele's position in the insertion order of the RETURNED dict (the last Gene_prob_next: first WALKED class containing
// the allele, then key order -- classes whose alleles_prob is 0 are skipped, typing_common.py:1321, so this can differ from the
// first class over all classes); -1 from the table-lookup arithmetic, which does not track dict orders.  The reference's final
// stable sort breaks abundance ties in this order.
struct hgx_emx_rec { int32_t allele, first; double prob; int32_t order, pad_; };

// One EM problem: `C` classes (rows of `w64` words over the locus' allele indices, dict order), their counts, the alleles' name
// order (rank[a] = place of allele a among the sorted names = its place inside a class key) and optionally the allele
// lengths (double per allele index).  All pointers are DEVICE memory that stays valid until the call returns.
struct hgx_emx_job {
    const uint64_t *bits;       // [C][w64]
    const int64_t *count;       // [C]
    const int32_t *rank;        // [a_pad]
    const double *len;          // [a_pad] or NULL
    const uint64_t *mask;       // [w64] or NULL.  With a mask the job is the exon -> gene hand-off (typing_core.py:1752-1766): every
                                // class is filtered to the alleles of the mask (at most 64 may occur, else status 1), empty ones are
                                // dropped, equal ones merged with their counts added, in the order of their first class; the EM
                                // runs on that set and `first` / n_classes refer to it
    int32_t C, w64, a_pad, remove_low;
    int32_t any_size;           // 1 (with fast = 0): take the problem up to HGX_EMX_HARD_MAX_CLASSES classes: "the reference's order at
                                // every size".  Such a job beyond the default gate gets a cluster launch of its own (several workgroups,
                                // ~45 ms at 16 000 classes), behind the ordinary launch of the call's other jobs
    int32_t fast;               // 0: the reference's own order of operations (bit-identical abundances); 1: table-lookup mat-vecs and
                                // tree reductions on the same workgroup -- ~5x faster, abundances within 1e-8 (typically 1e-11) of the
                                // reference's, same stopping and pruning rules
    // results (HOST memory, filled by hgx_emx_run)
    double *prob;               // [n_out] abundance, or -1.0 for an allele that is not in the returned dict (NULL with `recs`, below)
    int32_t *first;             // [n_out] or NULL: first class (dict order) containing the allele, -1 elsewhere
    int32_t *order;             // [n_out] or NULL: hgx_emx_rec::order per allele, -1 elsewhere
    int32_t n_out;              // alleles reported (<= a_pad)
    size_t rec_off;             // with `recs`: the job's records are recs[rec_off .. rec_off + n_rec)
    int32_t n_rec;
    int32_t n_iter;             // outer iterations
    int32_t n_classes;          // classes the EM ran on (= C without a mask)
    int32_t status;             // 0 = done, 1 = not taken (too many classes / distinct alleles: the caller uses another path),
                                // 2 = the reference would raise KeyError (quirk Q6)
};

// limits of the kernel (a job beyond them comes back with status 1)
constexpr int HGX_EMX_MAX_CLASSES = 4096;            // what a caller gets by default (beyond it the chip-wide table-lookup EM is ~100x faster)
constexpr int HGX_EMX_HARD_MAX_CLASSES = 32768;      // what the kernel can take (hgx_emx_job::any_size)
constexpr int HGX_EMX_MAX_ALLELES = 8192;

// Runs all jobs in ONE launch (one workgroup per job) on `st` and returns when the results are on the host.
// `recs` != NULL: the returned dicts come back as records (the alleles IN the dict only; a job's records are in compact
// name order) instead of dense per-allele arrays -- many small results cost nothing to clear and scan.
int hgx_emx_run(hgx_emx_job *jobs, int n_jobs, hipStream_t st, std::vector<hgx_emx_rec> *recs = nullptr);
