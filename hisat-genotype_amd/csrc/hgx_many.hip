// hgx_many.hip -- device pieces of hgx_type_many (hgx_type.hip): many (sample, locus) tasks of ONE locus behind one launch chain.
//
// The reference's unit of scale is many samples x loci (/root/reference/hisatgenotype:613-665 Pool.apply_async(genotyping_locus),
// typing_core.py:370 locus loop).  Tasks of one locus share the index, so their piece batches are merged into one (pairs task
// after task, distinct pieces interned across tasks) and scored by the SAME kernels as one task; the dedup keeps tasks apart
// (keys salted with the task, the exact check compares tasks: hgx_dedup.hip), so the class table of the merged batch is the
// tasks' class tables one after the other, each in its own first-seen order.  What remains per task -- Gene_counts, the EMs --
// runs with a task dimension: the kernels below and k_emx (hgx_emx.hip).
#include <algorithm>
#include <vector>

#include "hgx_common.hpp"

// First class of every task.  The class table of a merged batch is in first-seen order over pairs that come task after task, so
// a task's classes are one run: start[t] = index of the first class whose first pair belongs to task t (-1 if it has none).
__global__ void k_class_tasks(const int64_t *__restrict__ first_row, int n_classes, const uint32_t *__restrict__ pair_seg,
                              int32_t *__restrict__ start) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_classes) return;
    const uint32_t t = pair_seg[first_row[c]];
    if (c == 0 || pair_seg[first_row[c - 1]] != t) start[t] = c;
}

int hgx_many_class_tasks(const hgx_classes *cl, const uint32_t *pair_seg, int32_t n_tasks, int32_t *start_dev, hipStream_t st) {
    HIPCHK(hipMemsetAsync(start_dev, 0xFF, (size_t)n_tasks * 4, st));
    if (cl->n_classes > 0)
        hipLaunchKernelGGL(k_class_tasks, dim3(nblk(cl->n_classes, 256)), dim3(256), 0, st, cl->d_first_row, cl->n_classes, pair_seg, start_dev);
    HIPCHK(hipGetLastError());
    return HGX_OK;
}

// Gene_counts of every task (typing_core.py:1187-1190): per allele the summed count of the task's classes containing it, and the
// first pair (within the task) of the first such class -- the dict insertion order that breaks count ties (core:1650-1651).
// One workgroup per (8 allele words, task): thread (r, q) walks the classes r, r + 32, ... of the task and adds the bits of word q
// of the group into 512 integer counters in LDS (integer sums: order-independent, deterministic).
__global__ __launch_bounds__(256) void k_many_counts(const uint64_t *__restrict__ bits, const int64_t *__restrict__ count,
                                                     const int64_t *__restrict__ first_row, const int32_t *__restrict__ cls_off,
                                                     const int32_t *__restrict__ pair_base, int w64, int a_pad,
                                                     int64_t *__restrict__ out_cnt, int32_t *__restrict__ out_first_pair) {
    __shared__ unsigned long long cnt[512];
    __shared__ int fcls[512];
    const int t = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < 512; i += 256) { cnt[i] = 0ull; fcls[i] = 0x7fffffff; }
    __syncthreads();
    const int c0 = cls_off[t], c1 = cls_off[t + 1];
    const int q = tid & 7, w = 8 * g + q;
    if (w < w64)
        for (int c = c0 + (tid >> 3); c < c1; c += 32) {
            uint64_t x = bits[(size_t)c * w64 + w];
            if (!x) continue;
            const unsigned long long n = (unsigned long long)count[c];
            for (; x; x &= x - 1) {
                const int b = 64 * q + __builtin_ctzll(x);
                atomicAdd(&cnt[b], n);
                atomicMin(&fcls[b], c);
            }
        }
    __syncthreads();
    for (int i = tid; i < 512; i += 256) {
        const int a = 512 * g + i;
        if (a >= a_pad) continue;
        out_cnt[(size_t)t * a_pad + a] = (int64_t)cnt[i];
        out_first_pair[(size_t)t * a_pad + a] = fcls[i] == 0x7fffffff ? -1 : (int32_t)(first_row[fcls[i]] - pair_base[t]);
    }
}

int hgx_many_counts(const hgx_classes *gcl, const int32_t *cls_off_dev, const int32_t *pair_base_dev, int32_t n_tasks,
                    int64_t *out_cnt_dev, int32_t *out_first_pair_dev, hipStream_t st) {
    if (n_tasks <= 0) return HGX_OK;
    hipLaunchKernelGGL(k_many_counts, dim3((unsigned)((gcl->w64 + 7) / 8), (unsigned)n_tasks), dim3(256), 0, st, gcl->d_bits, gcl->d_count,
                       gcl->d_first_row, cls_off_dev, pair_base_dev, gcl->w64, gcl->a_pad, out_cnt_dev, out_first_pair_dev);
    HIPCHK(hipGetLastError());
    return HGX_OK;
}

// pair -> task
__global__ void k_fill_seg(const int32_t *__restrict__ pair_base, int n_tasks, uint32_t *__restrict__ seg) {
    const int t = blockIdx.y;
    const int p0 = pair_base[t], p1 = pair_base[t + 1];
    for (int p = p0 + blockIdx.x * blockDim.x + threadIdx.x; p < p1; p += gridDim.x * blockDim.x) seg[p] = (uint32_t)t;
}
int hgx_many_fill_seg(const int32_t *pair_base_dev, int32_t n_tasks, uint32_t *seg_dev, hipStream_t st) {
    if (n_tasks <= 0) return HGX_OK;
    hipLaunchKernelGGL(k_fill_seg, dim3(64, (unsigned)n_tasks), dim3(256), 0, st, pair_base_dev, n_tasks, seg_dev);
    HIPCHK(hipGetLastError());
    return HGX_OK;
}
