// hgx_host.cpp -- host side of libhgx: locus tables (8a-0) and haplotype -> piece-mask reduction
// (the host half of add_count, typing_core.py:626-677).  Plain C++; no device code here.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <cstring>
#include <map>
#include <numeric>
#include <string_view>
#include <unordered_map>

#include <sys/mman.h>

#include "hgx_internal.hpp"

extern "C" int hgx_locus_create(hgx_locus **out, const hgx_locus_desc *d) {
    HARGCHK(out && d && d->backbone && d->backbone_len > 0 && d->n_vars >= 0 && d->n_alleles > 0);
    HARGCHK(d->n_vars == 0 || (d->var_pos && d->var_type && d->var_len && d->var_base && d->var_linked &&
                               d->var_name_pool && d->var_ins_pool && d->link_off && d->link_allele));
    HARGCHK(d->n_exons == 0 || d->exons);
    hgx_locus *L = new hgx_locus();
    L->base_kind = d->base_kind;
    L->backbone.assign(d->backbone, d->backbone + d->backbone_len);
    const int V = L->V = d->n_vars;
    L->pos.assign(d->var_pos, d->var_pos + V);
    L->type.assign(d->var_type, d->var_type + V);
    L->len.assign(d->var_len, d->var_len + V);
    L->base.assign(d->var_base, d->var_base + V);
    L->linked.assign(d->var_linked, d->var_linked + V);
    L->right.resize(V);
    L->maxright.resize(V);
    const char *np = d->var_name_pool, *ip = d->var_ins_pool;
    int cur = -1;
    for (int v = 0; v < V; ++v) {
        L->name.emplace_back(np);
        np += L->name.back().size() + 1;
        L->ins.emplace_back(ip);
        ip += L->ins.back().size() + 1;
        L->name_to_var[L->name.back()] = v;
        {
            const std::string &nm = L->name.back();
            if (nm.size() > 2 && nm.size() < 10 && nm[0] == 'h' && nm[1] == 'v') {
                long num = 0;
                bool digits = true;
                for (size_t k = 2; k < nm.size(); ++k) {
                    if (nm[k] < '0' || nm[k] > '9') { digits = false; break; }
                    num = num * 10 + (nm[k] - '0');
                }
                if (digits && num < 50000000 && !(nm.size() > 3 && nm[2] == '0')) {
                    if ((size_t)num >= L->hv_index.size()) L->hv_index.resize(num + 1, -1);
                    L->hv_index[num] = v;
                }
            }
        }
        if (v > 0 && L->pos[v] < L->pos[v - 1]) {
            hgx_set_error("variants must be sorted by position (gene_var_list order)");
            delete L;
            return HGX_EINVAL;
        }
        L->right[v] = L->type[v] == HGX_VAR_DELETION ? L->pos[v] + L->len[v] - 1 : L->pos[v];
        cur = std::max(cur, L->right[v]);                          // core:396-401
        L->maxright[v] = cur;
    }
    const int A = L->A = d->n_alleles;
    L->a_pad = hgx_a_pad(A);
    L->w64 = L->a_pad / 64;
    L->n_words = std::max(1, (V + 31) / 32);
    if (L->n_words > 65535) {
        hgx_set_error("too many variants (%d)", V);
        delete L;
        return HGX_EINVAL;
    }
    L->link_off.assign(d->link_off ? d->link_off : nullptr, d->link_off ? d->link_off + V + 1 : nullptr);
    if (V == 0) L->link_off.assign(1, 0);
    L->link_allele.assign(d->link_allele, d->link_allele + L->link_off[V]);
    for (int e = 0; e < d->n_exons; ++e) L->exons.push_back({d->exons[2 * e], d->exons[2 * e + 1]});
    if (d->allele_len) L->allele_len.assign(d->allele_len, d->allele_len + A);
    if (d->name_rank) L->name_rank.assign(d->name_rank, d->name_rank + A);

    // link bit matrix, word-major
    L->link_bits.assign((size_t)L->n_words * L->a_pad, 0u);
    std::vector<int32_t> n_av(A, 0);
    for (int v = 0; v < V; ++v) {
        for (int k = L->link_off[v]; k < L->link_off[v + 1]; ++k) {
            const int a = L->link_allele[k];
            if (a < 0 || a >= A) {
                hgx_set_error("link_allele out of range");
                delete L;
                return HGX_EINVAL;
            }
            uint32_t &w = L->link_bits[(size_t)(v >> 5) * L->a_pad + a];
            if (!((w >> (v & 31)) & 1u)) n_av[a]++;
            w |= 1u << (v & 31);
        }
    }
    // allele -> variants in gene_var_list order (core:476-487)
    L->av_off.assign(A + 1, 0);
    for (int a = 0; a < A; ++a) L->av_off[a + 1] = L->av_off[a] + n_av[a];
    L->av_var.resize(L->av_off[A]);
    {
        std::vector<int32_t> fill(L->av_off.begin(), L->av_off.end() - 1);
        // iterate variant-major so each allele's list comes out position-sorted
        for (int v = 0; v < V; ++v) {
            const uint32_t *row = &L->link_bits[(size_t)(v >> 5) * L->a_pad];
            for (int k = L->link_off[v]; k < L->link_off[v + 1]; ++k) {
                const int a = L->link_allele[k];
                if ((row[a] >> (v & 31)) & 1u) {
                    // guard against a duplicated allele inside one Links entry
                    if (fill[a] > L->av_off[a] && L->av_var[fill[a] - 1] == v) continue;
                    L->av_var[fill[a]++] = v;
                }
            }
        }
    }
    // exonic variants (core:67-78) and representative alleles (core:86-115)
    L->exonic.assign(V, 0);
    for (int v = 0; v < V; ++v)
        for (auto &e : L->exons)
            if (L->pos[v] >= e[0] && L->right[v] <= e[1]) L->exonic[v] = 1;
    L->rep_of.assign(A, -1);
    {
        // alleles in order of first appearance scanning Links (dict order) restricted to exonic variants
        std::vector<int32_t> order;
        std::vector<uint8_t> seen(A, 0);
        std::vector<std::vector<int32_t>> evars(A);
        std::vector<int32_t> lo;
        if (d->link_order && d->n_link_order > 0) lo.assign(d->link_order, d->link_order + d->n_link_order);
        else for (int v = 0; v < V; ++v) if (L->linked[v]) lo.push_back(v);
        for (int v : lo) {
            if (v < 0 || v >= V || !L->exonic[v]) continue;
            for (int k = L->link_off[v]; k < L->link_off[v + 1]; ++k) {
                const int a = L->link_allele[k];
                if (!seen[a]) { seen[a] = 1; order.push_back(a); }
                evars[a].push_back(v);
            }
        }
        std::map<std::vector<int32_t>, int32_t> group_rep;
        for (int a : order) {
            auto &vs = evars[a];
            std::sort(vs.begin(), vs.end());
            vs.erase(std::unique(vs.begin(), vs.end()), vs.end());
            auto it = group_rep.find(vs);
            if (it == group_rep.end()) { group_rep.emplace(vs, a); L->rep_of[a] = a; }
            else L->rep_of[a] = it->second;
        }
    }
    L->linked_bits.assign(L->n_words, 0u);
    for (int v = 0; v < V; ++v)
        if (L->linked[v]) L->linked_bits[v >> 5] |= 1u << (v & 31);
    L->exon_mask.assign(L->w64, 0ull);
    L->gene_mask.assign(L->w64, 0ull);
    for (int a = 0; a < A; ++a) {
        L->gene_mask[a >> 6] |= 1ull << (a & 63);
        if (L->rep_of[a] == a) L->exon_mask[a >> 6] |= 1ull << (a & 63);
    }
    // exon groups as member lists (allele_rep_groups, core:86-115): the hand-off walks the groups of the leading representatives
    L->grp_off.assign(A + 1, 0);
    for (int a = 0; a < A; ++a) if (L->rep_of[a] >= 0) L->grp_off[L->rep_of[a] + 1]++;
    for (int a = 0; a < A; ++a) L->grp_off[a + 1] += L->grp_off[a];
    L->grp_member.assign(L->grp_off[A], 0);
    {
        std::vector<int32_t> at(L->grp_off.begin(), L->grp_off.end() - 1);
        for (int a = 0; a < A; ++a) if (L->rep_of[a] >= 0) L->grp_member[at[L->rep_of[a]]++] = a;
    }
    *out = L;
    return HGX_OK;
}

extern "C" int hgx_locus_destroy(hgx_locus *loc) { delete loc; return HGX_OK; }

extern "C" int hgx_locus_dims(const hgx_locus *L, int32_t *na, int32_t *ap, int32_t *nv, int32_t *nw) {
    HARGCHK(L);
    if (na) *na = L->A;
    if (ap) *ap = L->a_pad;
    if (nv) *nv = L->V;
    if (nw) *nw = L->n_words;
    return HGX_OK;
}

extern "C" int hgx_locus_tables(const hgx_locus *L, uint32_t *bits, uint64_t *em, uint64_t *gm, int32_t *rep_of) {
    HARGCHK(L);
    if (bits) memcpy(bits, L->link_bits.data(), L->link_bits.size() * 4);
    if (em) memcpy(em, L->exon_mask.data(), L->exon_mask.size() * 8);
    if (gm) memcpy(gm, L->gene_mask.data(), L->gene_mask.size() * 8);
    if (rep_of) memcpy(rep_of, L->rep_of.data(), L->rep_of.size() * 4);
    return HGX_OK;
}

extern "C" int hgx_index_from_locus(hgx_index **out, const hgx_locus *L) {
    HARGCHK(out && L);
    return hgx_index_create(out, L->A, L->V, L->link_bits.data(), L->exon_mask.data(), L->gene_mask.data());
}

// ---------------------------------------------------------------------------------------------
// piece -> masks.  P = the piece's own known, linked variants (core:642-647).  M = known, linked
// variants outside the piece's id list whose left or right end lies in [left, right]; the
// reference finds them scanning down from lower_bound(right + 1) until the prefix-max right end
// drops below `left` (core:651-670) -- every variant above the start has pos > right and every
// variant below the stop has right end < left, so the scan bounds never change the set.
// ---------------------------------------------------------------------------------------------
uint32_t hgx_intern_masks(hgx_batch &b, uint16_t lo, uint16_t nw, const uint32_t *m) {
    PieceTable &T = b.table;
    if (T.slot.empty()) T.slot.assign(1024, -1);
    if ((T.used + 1) * 2 > T.slot.size()) {                 // grow + rehash from the piece list
        std::vector<int32_t> ns(T.slot.size() * 2, -1);
        const size_t mask = ns.size() - 1;
        for (size_t id = 0; id < b.pieces.size(); ++id) {
            const hgx_piece &pc = b.pieces[id];
            size_t h = PieceTable::hash(pc.lo_word, pc.n_words, &b.masks[pc.mask_off]) & mask;
            while (ns[h] >= 0) h = (h + 1) & mask;
            ns[h] = (int32_t)id;
        }
        T.slot.swap(ns);
    }
    const size_t mask = T.slot.size() - 1;
    size_t h = PieceTable::hash(lo, nw, m) & mask;
    for (;;) {
        const int32_t id = T.slot[h];
        if (id < 0) break;
        const hgx_piece &pc = b.pieces[id];
        if (pc.lo_word == lo && pc.n_words == nw && memcmp(&b.masks[pc.mask_off], m, 8 * (size_t)nw) == 0) return (uint32_t)id;
        h = (h + 1) & mask;
    }
    const uint32_t id = (uint32_t)b.pieces.size();
    hgx_piece pc;
    pc.mask_off = (uint32_t)b.masks.size();
    pc.lo_word = lo;
    pc.n_words = nw;
    b.pieces.push_back(pc);
    b.masks.insert(b.masks.end(), m, m + 2 * (size_t)nw);
    T.slot[h] = (int32_t)id;
    T.used++;
    return id;
}

int64_t hgx_intern_piece(hgx_batch &b, const hgx_locus &L, int32_t left, int32_t right, const int32_t *ids, int32_t n_ids) {
    // MP = M | P: M only holds linked variants and the linked members of `ids` are exactly P, so excluding the piece's
    // own ids from the span scan (core:655-657) and OR-ing P back in is the same set.  Variants with pos in
    // [left, right] are one contiguous index range [i0, i1) of the position-sorted list; the only others are variants
    // starting before `left` whose right end reaches into the span (bounded by the prefix-max right end).
    const int V = L.V;
    const int i0 = lower_bound_pos(L.pos, left), i1 = lower_bound_pos(L.pos, right + 1);
    int lo_v = 0x7fffffff, hi_v = -1;
    if (i1 > i0) { lo_v = i0; hi_v = i1 - 1; }
    int extra[64], n_extra = 0;
    for (int j = i0 - 1; j >= 0 && L.maxright[j] >= left; --j) {
        if (L.linked[j] && L.right[j] >= left && L.right[j] <= right) {
            if (n_extra < 64) extra[n_extra++] = j;
            else { hgx_set_error("too many spanning variants"); return -1; }
            lo_v = std::min(lo_v, j);
            hi_v = std::max(hi_v, j);
        }
    }
    for (int i = 0; i < n_ids; ++i) {
        const int v = ids[i];
        if (v >= 0 && v < V && L.linked[v]) { lo_v = std::min(lo_v, v); hi_v = std::max(hi_v, v); }
    }
    int lo_w = 0, hi_w = 0;
    if (hi_v >= 0) { lo_w = lo_v >> 5; hi_w = hi_v >> 5; }
    const int nw = hi_w - lo_w + 1;
    if (nw > 65535) {
        hgx_set_error("piece spans %d variant words (> 65535)", nw);
        return -1;
    }
    uint32_t small[128];
    std::vector<uint32_t> big;
    uint32_t *buf = small;
    if (nw > 64) { big.resize(2 * (size_t)nw); buf = big.data(); }
    memset(buf, 0, 8 * (size_t)nw);
    if (i1 > i0) {                                               // linked variants of the index range, word at a time
        for (int w = i0 >> 5; w <= (i1 - 1) >> 5; ++w) {
            uint32_t m = 0xffffffffu;
            if (w == (i0 >> 5)) m &= 0xffffffffu << (i0 & 31);
            if (w == ((i1 - 1) >> 5)) m &= 0xffffffffu >> (31 - ((i1 - 1) & 31));
            buf[2 * (w - lo_w)] |= m & L.linked_bits[w];
        }
    }
    for (int k = 0; k < n_extra; ++k) buf[2 * ((extra[k] >> 5) - lo_w)] |= 1u << (extra[k] & 31);
    for (int i = 0; i < n_ids; ++i) {
        const int v = ids[i];
        if (v >= 0 && v < V && L.linked[v]) {
            buf[2 * ((v >> 5) - lo_w)] |= 1u << (v & 31);
            buf[2 * ((v >> 5) - lo_w) + 1] |= 1u << (v & 31);
        }
    }
    return hgx_intern_masks(b, (uint16_t)lo_w, (uint16_t)nw, buf);
}

// Order the distinct-piece table by first covered word, then width, then CONTENT (hash of the mask words, the words themselves
// on a hash tie) and renumber the refs.  Consecutive pieces then share an index window, which is what the LDS-tiled
// compatibility kernel exploits; and the order depends on nothing but the set of pieces, so a batch is the same whatever
// the number of front-end workers that interned them, in whatever order.  Masks are re-packed in the same order.
void hgx_finalize_batch(hgx_batch &b, int n_threads) {
    std::vector<uint32_t> new_id;
    hgx_canonical_piece_order(b, n_threads, new_id);
    hgx_par_ranges(b.pair_ref.size() > 200000 ? n_threads : 1, b.pair_ref.size(), [&](int, size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; ++k) b.pair_ref[k] = (b.pair_ref[k] & 0x80000000u) | new_id[b.pair_ref[k] & 0x7fffffffu];
    });
}

// the table part of hgx_finalize_batch: pieces + masks re-ordered, new_id[old id] = new id (refs are the caller's business)
void hgx_canonical_piece_order(hgx_batch &b, int n_threads, std::vector<uint32_t> &new_id) {
    const size_t n = b.pieces.size();
    std::vector<uint32_t> order(n);
    new_id.assign(n, 0);
    std::vector<uint64_t> hash(n);
    hgx_par_ranges(n > 20000 ? n_threads : 1, n, [&](int, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            order[i] = (uint32_t)i;
            hash[i] = PieceTable::hash(b.pieces[i].lo_word, b.pieces[i].n_words, &b.masks[b.pieces[i].mask_off]);
        }
    });
    auto less = [&](uint32_t x, uint32_t y) {
        const hgx_piece &px = b.pieces[x], &py = b.pieces[y];
        if (px.lo_word != py.lo_word) return px.lo_word < py.lo_word;
        if (px.n_words != py.n_words) return px.n_words < py.n_words;
        if (hash[x] != hash[y]) return hash[x] < hash[y];
        return memcmp(&b.masks[px.mask_off], &b.masks[py.mask_off], 8 * (size_t)px.n_words) < 0;
    };
    if (n < 20000 || n_threads <= 1) std::sort(order.begin(), order.end(), less);
    else {
        // bucket by first word (a counting pass), then the buckets are sorted side by side
        uint32_t max_lo = 0;
        for (size_t i = 0; i < n; ++i) max_lo = std::max<uint32_t>(max_lo, b.pieces[i].lo_word);
        std::vector<size_t> start((size_t)max_lo + 2, 0);
        for (size_t i = 0; i < n; ++i) start[(size_t)b.pieces[i].lo_word + 1]++;
        for (size_t k = 0; k + 1 < start.size(); ++k) start[k + 1] += start[k];
        {
            std::vector<size_t> at(start.begin(), start.end() - 1);
            for (size_t i = 0; i < n; ++i) order[at[b.pieces[i].lo_word]++] = (uint32_t)i;
        }
        hgx_par_tasks(n_threads, (size_t)max_lo + 1, [&](int, size_t k) {
            std::sort(order.begin() + start[k], order.begin() + start[k + 1], less);
        });
    }
    PVec<hgx_piece> np(n);
    PVec<uint32_t> nm(b.masks.size());
    size_t at = 0;
    for (size_t k = 0; k < n; ++k) {
        const hgx_piece &src = b.pieces[order[k]];
        new_id[order[k]] = (uint32_t)k;
        np[k] = src;
        np[k].mask_off = (uint32_t)at;
        memcpy(&nm[at], &b.masks[src.mask_off], 8 * (size_t)src.n_words);
        at += 2 * (size_t)src.n_words;
    }
    nm.resize(at);
    b.pieces.swap(np);
    b.masks.swap(nm);
    b.table.clear();
}

extern "C" int hgx_batch_from_haplotypes(hgx_batch **out, const hgx_locus *L, int32_t n_pairs, const int32_t *pair_off,
                                         const uint8_t *level, const int32_t *left, const int32_t *right,
                                         const int32_t *id_off, const int32_t *ids) {
    HARGCHK(out && L && n_pairs >= 0 && pair_off);
    hgx_batch *b = new hgx_batch();
    for (int p = 0; p < n_pairs; ++p) {
        int n_lvl[2] = {0, 0};
        for (int q = pair_off[p]; q < pair_off[p + 1]; ++q) {
            if (left[q] > right[q]) {
                hgx_set_error("piece with left > right (the reference asserts, core:638)");
                delete b;
                return HGX_EPARSE;
            }
            const int64_t id = hgx_intern_piece(*b, *L, left[q], right[q], ids + id_off[q], id_off[q + 1] - id_off[q]);
            if (id < 0) { delete b; return HGX_EINVAL; }
            const uint32_t lv = level[q] ? 1u : 0u;
            if (++n_lvl[lv] > 65535) {
                hgx_set_error("more than 65535 pieces for one pair and level");
                delete b;
                return HGX_EINVAL;
            }
            b->pair_ref.push_back((uint32_t)id | (lv << 31));
        }
        b->pair_off.push_back((int32_t)b->pair_ref.size());
    }
    hgx_finalize_batch(*b);
    *out = b;
    return HGX_OK;
}

// The batches of MANY tasks of one locus as ONE batch (hgx_many_create): distinct pieces interned across the tasks (samples of
// a locus repeat each other's pieces), pairs concatenated task after task, refs renumbered; pair_base[t] = first pair of task t
// (n + 1 entries).  The piece table is re-ordered canonically, as every batch is.
int hgx_batch_merge(hgx_batch **out, const hgx_batch *const *batches, int32_t n, int32_t *pair_base) {
    HARGCHK(out && n >= 0 && (n == 0 || batches) && pair_base);
    hgx_batch *m = new hgx_batch();
    size_t n_pairs = 0, n_refs = 0, n_pieces = 0;
    for (int t = 0; t < n; ++t) {
        HARGCHK(batches[t]);
        n_pairs += batches[t]->pair_off.size() - 1;
        n_refs += batches[t]->pair_ref.size();
        n_pieces += batches[t]->pieces.size();
    }
    if (n_pairs >= (1ull << 31) || n_refs >= (1ull << 31)) { delete m; hgx_set_error("too many pairs / piece refs for one merged batch"); return HGX_EINVAL; }
    m->pair_off.clear();
    m->pair_off.reserve(n_pairs + 1);
    m->pair_off.push_back(0);
    m->pair_ref.reserve(n_refs);
    m->pieces.reserve(std::min<size_t>(n_pieces, 1u << 20));
    std::vector<uint32_t> map;
    for (int t = 0; t < n; ++t) {
        const hgx_batch &b = *batches[t];
        pair_base[t] = (int32_t)(m->pair_off.size() - 1);
        map.resize(b.pieces.size());
        for (size_t i = 0; i < b.pieces.size(); ++i) {
            const hgx_piece &pc = b.pieces[i];
            map[i] = hgx_intern_masks(*m, pc.lo_word, pc.n_words, &b.masks[pc.mask_off]);
        }
        const int32_t ref_base = (int32_t)m->pair_ref.size();
        for (uint32_t r : b.pair_ref) m->pair_ref.push_back((r & 0x80000000u) | map[r & 0x7fffffffu]);
        for (size_t p = 1; p < b.pair_off.size(); ++p) m->pair_off.push_back(ref_base + b.pair_off[p]);
        m->n_reads += b.n_reads;
    }
    pair_base[n] = (int32_t)(m->pair_off.size() - 1);
    hgx_finalize_batch(*m, hgx_default_threads());
    *out = m;
    return HGX_OK;
}

extern "C" int hgx_batch_destroy(hgx_batch *b) { delete b; return HGX_OK; }

extern "C" int hgx_batch_dims(const hgx_batch *b, int32_t *n_pieces, int64_t *n_mask_u32, int32_t *n_pairs, int64_t *n_refs,
                              int32_t *n_reads) {
    HARGCHK(b);
    if (n_pieces) *n_pieces = (int32_t)b->pieces.size();
    if (n_mask_u32) *n_mask_u32 = (int64_t)b->masks.size();
    if (n_pairs) *n_pairs = (int32_t)b->pair_off.size() - 1;
    if (n_refs) *n_refs = (int64_t)b->pair_ref.size();
    if (n_reads) *n_reads = b->n_reads;
    return HGX_OK;
}

extern "C" int hgx_batch_arrays(const hgx_batch *b, const hgx_piece **pieces, const uint32_t **masks, const int32_t **pair_off,
                                const uint32_t **pair_ref) {
    HARGCHK(b);
    if (pieces) *pieces = b->pieces.data();
    if (masks) *masks = b->masks.data();
    if (pair_off) *pair_off = b->pair_off.data();
    if (pair_ref) *pair_ref = b->pair_ref.data();
    return HGX_OK;
}


// ---- test hooks (hgx.h: hgx_test_switch_set) --------------------------------------------------------------------
// Path-forcing switches of the test-suite and the lab tools (a kernel kept for comparison, a size threshold moved so that a
// small case reaches the large-problem path).  They live in the process, not in the environment: a product run cannot pick one up
// by accident, and the query is one relaxed load while none is set.
namespace {
std::mutex g_sw_mu;
std::map<std::string, const std::string *> g_sw;      // name -> current value
// values are interned and never freed: a pointer handed out stays valid whatever another thread sets or clears afterwards
// (test hooks only: a few dozen short strings per process)
std::map<std::string, std::string *> g_sw_values;
std::atomic<int> g_sw_n{0};
}
extern "C" const char *hgx_test_switch(const char *name) {
    if (g_sw_n.load(std::memory_order_relaxed) == 0) return nullptr;
    std::lock_guard<std::mutex> g(g_sw_mu);
    auto it = g_sw.find(name);
    return it == g_sw.end() ? nullptr : it->second->c_str();
}
extern "C" int hgx_test_switch_set(const char *name, const char *value) {
    std::lock_guard<std::mutex> g(g_sw_mu);
    if (!name) g_sw.clear();
    else if (!value) g_sw.erase(name);
    else {
        auto it = g_sw_values.find(value);
        if (it == g_sw_values.end()) it = g_sw_values.emplace(value, new std::string(value)).first;
        g_sw[name] = it->second;
    }
    g_sw_n.store((int)g_sw.size(), std::memory_order_relaxed);
    return HGX_OK;
}

// ---- malloc thresholds -----------------------------------------------------------------------------------------------
// A typing call builds and drops host arrays of tens of KB to a few MB per task (Gene_counts, result lists, EM records) from
// several threads.  With glibc's defaults every block above 128 KB is its own mmap / munmap (page faults for fresh zero pages, a
// TLB shoot-down per unmap in a process with dozens of threads) and the heap top is trimmed and re-grown around every call:
// measured on the 384-task panel, 2.0 ms to destroy the results and ~1.5 ms spread over the call (17.7 -> 15.2 ms per step).
// The library therefore raises the thresholds once when it is loaded (HGX_MALLOC_TUNE=0 leaves malloc alone).
#include <malloc.h>
namespace {
struct MallocTune {
    MallocTune() {
        const char *e = getenv("HGX_MALLOC_TUNE");
        if (e && atoi(e) == 0) return;
        mallopt(M_MMAP_THRESHOLD, 32 << 20);       // (the largest value glibc accepts)
        mallopt(M_TRIM_THRESHOLD, 256 << 20);
        mallopt(M_TOP_PAD, 64 << 20);
    }
} g_malloc_tune;
}

// ---- host block pool (declared in hgx_internal.hpp) ------------------------------------------------------------
#include <map>
#include <mutex>
#include <new>
namespace {
struct HostPool {
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks;      // capacity -> block (header included)
    size_t held = 0;                                // bytes parked in free_blocks
};
constexpr size_t HOST_POOL_MAX_BYTES = 8ull << 30, HOST_POOL_MAX_BLOCKS = 1024;
HostPool &host_pool() { static HostPool p; return p; }
struct BlockHeader { size_t cap; uint64_t magic; uint64_t pad[2]; };     // 32 bytes: keeps the payload 32-byte aligned
constexpr uint64_t HOST_MAGIC = 0x6867785f686f7374ull;
constexpr size_t HOST_POOL_MIN = 1u << 20;
}   // namespace

// Blocks from another allocator (the device front end's pinned staging): while a thread holds an hgx_big_alloc_scope, its
// allocations of at least `min_bytes` come from that allocator, and hgx_host_free hands them back to it (a registry of the live
// foreign blocks: they carry no header).
namespace {
struct Foreign { std::mutex mu; std::map<void *, void (*)(void *)> live; };
Foreign &foreign() { static Foreign *f = new Foreign(); return *f; }
thread_local hgx_front_alloc g_big_alloc{nullptr, nullptr};
thread_local size_t g_big_min = 0;
std::atomic<long> g_foreign_n{0};
}
hgx_big_alloc_scope::hgx_big_alloc_scope(hgx_front_alloc a, size_t min_bytes) : old(g_big_alloc), old_min(g_big_min) { g_big_alloc = a; g_big_min = min_bytes; }
hgx_big_alloc_scope::~hgx_big_alloc_scope() { g_big_alloc = old; g_big_min = old_min; }

void *hgx_host_alloc(size_t bytes) {
    if (g_big_alloc.alloc && bytes >= g_big_min) {
        const hgx_front_alloc a = g_big_alloc;
        g_big_alloc.alloc = nullptr;                       // (the other allocator may itself be built on this one)
        void *p = a.alloc(bytes);
        g_big_alloc = a;
        if (p) {
            Foreign &F = foreign();
            std::lock_guard<std::mutex> g(F.mu);
            F.live[p] = a.release;
            g_foreign_n.fetch_add(1);
            return p;
        }
    }
    const size_t need = bytes + sizeof(BlockHeader);
    if (need >= HOST_POOL_MIN) {
        HostPool &P = host_pool();
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.free_blocks.lower_bound(need);
        if (it != P.free_blocks.end() && it->first <= 2 * need) {
            void *b = it->second;
            P.held -= it->first;
            P.free_blocks.erase(it);
            return (char *)b + sizeof(BlockHeader);
        }
    }
    size_t cap = need >= HOST_POOL_MIN ? need + need / 16 : need;      // a little slack so that similar sizes fit later
    BlockHeader *h = nullptr;
    // HGX_THP=1: big blocks on transparent huge pages where the kernel hands them out on request (THP "madvise" mode).  Measured
    // both ways: alone (tools/e2e_file.py, e2e_bam.py) the call's system time drops from 40-300 ms to 0-25 ms and slow outliers
    // get rarer; inside bench.py (a process that also holds the 400 MB text and the GPU runtime's memory) the five spaced SAM
    // calls went from 50.1-51.8 ms to 41.7-86.8 ms -- a huge page that has to be compacted for at fault time stalls the call.
    // Off by default.
    static const bool thp = [] { const char *e = getenv("HGX_THP"); return e && atoi(e) != 0; }();
    if (thp && cap >= (4u << 20)) {
        cap = (cap + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        h = (BlockHeader *)aligned_alloc(2u << 20, cap);
        if (h) (void)madvise(h, cap, MADV_HUGEPAGE);
    }
    if (!h) h = (BlockHeader *)malloc(cap);
    if (!h) throw std::bad_alloc();
    h->cap = cap;
    h->magic = HOST_MAGIC;
    return (char *)h + sizeof(BlockHeader);
}
void hgx_host_free(void *p) {
    if (!p) return;
    if (g_foreign_n.load(std::memory_order_relaxed) > 0) {
        Foreign &F = foreign();
        void (*rel)(void *) = nullptr;
        {
            std::lock_guard<std::mutex> g(F.mu);
            auto it = F.live.find(p);
            if (it != F.live.end()) { rel = it->second; F.live.erase(it); g_foreign_n.fetch_sub(1); }
        }
        if (rel) { rel(p); return; }
    }
    BlockHeader *h = (BlockHeader *)((char *)p - sizeof(BlockHeader));
    if (h->magic != HOST_MAGIC) return;               // not ours: leak rather than corrupt
    if (h->cap >= HOST_POOL_MIN) {
        HostPool &P = host_pool();
        std::lock_guard<std::mutex> g(P.mu);
        if (P.free_blocks.size() < HOST_POOL_MAX_BLOCKS && P.held + h->cap <= HOST_POOL_MAX_BYTES) {
            P.free_blocks.insert({h->cap, (void *)h});
            P.held += h->cap;
            return;
        }
    }
    free(h);
}
void hgx_host_pool_trim() {
    HostPool &P = host_pool();
    std::lock_guard<std::mutex> g(P.mu);
    for (auto &kv : P.free_blocks) free(kv.second);
    P.free_blocks.clear();
    P.held = 0;
}

// ---- persistent host worker pool (declared in hgx_internal.hpp) -------------------------------------------------
// The ingestion path runs half a dozen short parallel phases per sample (2-20 ms each); creating 100-250 threads for
// every phase costs more than some of the phases.  Workers are created once, sleep on a condition variable between
// phases and are shared by all callers; a caller that finds the pool busy (another sample being parsed at the same
// time) runs its phase on temporary threads instead of waiting.
#include <atomic>
#include <condition_variable>
#include <thread>
#include <pthread.h>
#include <sched.h>
namespace {
// One hardware thread per core of the NUMA node the calling thread runs on (restricted to the process's affinity mask), or
// an empty set.  The pool's workers are kept there: the text, the line table and the locus tables were first touched by the
// caller, and two workers on one core's SMT siblings share its pipes.  HGX_PIN=0 leaves the workers unpinned.
bool numa_node_cpus(cpu_set_t *out) {
    CPU_ZERO(out);
    if (const char *e = getenv("HGX_PIN")) if (atoi(e) == 0) return false;
    const int cpu = sched_getcpu();
    if (cpu < 0) return false;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    for (int node = 0; node < 64; ++node) {
        char path[96];
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        FILE *f = fopen(path, "r");
        if (!f) { if (node == 0) return false; break; }
        char buf[4096];
        const bool ok = fgets(buf, sizeof(buf), f) != nullptr;
        fclose(f);
        if (!ok) continue;
        cpu_set_t set;
        CPU_ZERO(&set);
        bool mine = false;
        for (char *q = buf; *q;) {                                   // "0-63,128-191"
            char *end;
            const long lo = strtol(q, &end, 10);
            if (end == q) break;
            long hi = lo;
            q = end;
            if (*q == '-') { hi = strtol(q + 1, &end, 10); q = end; }
            for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &set); if (c == cpu) mine = true; }
            while (*q == ',' || *q == '\n' || *q == ' ') ++q;
        }
        if (!mine) continue;
        // one hardware thread per core (the lowest-numbered sibling): two workers sharing a core's pipes gain little
        int n = 0;
        for (int c = 0; c < CPU_SETSIZE; ++c) {
            if (!CPU_ISSET(c, &set) || !CPU_ISSET(c, &allowed)) continue;
            snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
            long first = c;
            if (FILE *g = fopen(path, "r")) { if (fscanf(g, "%ld", &first) != 1) first = c; fclose(g); }
            if (first != c && CPU_ISSET((int)first, &allowed)) continue;
            CPU_SET(c, out);
            ++n;
        }
        return n >= 2;
    }
    return false;
}
struct WorkerPool {
    std::mutex owner;                      // one parallel phase at a time
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> workers;
    const std::function<void(int)> *body = nullptr;
    uint64_t generation = 0;
    int want = 0;                          // workers 0 .. want-1 take part in the current phase (as ids 1 .. want)
    int running = 0;
    bool stopping = false;
    bool pin_known = false, pin = false;   // workers stay on the NUMA node the first caller ran on
    cpu_set_t pin_set;

    void loop(int id) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_job.wait(lk, [&] { return stopping || generation != seen; });
            if (stopping) return;
            seen = generation;
            if (id >= want) continue;
            const std::function<void(int)> *fn = body;
            lk.unlock();
            (*fn)(id + 1);
            lk.lock();
            if (--running == 0) cv_done.notify_one();
        }
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> g(mu);
            stopping = true;
        }
        cv_job.notify_all();
        for (auto &t : workers) t.join();
    }
};
WorkerPool &worker_pool() { static WorkerPool *p = new WorkerPool(); return *p; }    // leaked on purpose: no join at exit
}   // namespace

void hgx_run_workers(int n, const std::function<void(int)> &body) {
    if (n <= 1) { body(0); return; }
    WorkerPool &P = worker_pool();
    std::unique_lock<std::mutex> own(P.owner, std::try_to_lock);
    if (!own.owns_lock()) {                 // pool busy with another caller's phase: temporary threads
        std::vector<std::thread> th;
        for (int t = 1; t < n; ++t) th.emplace_back([&body, t] { body(t); });
        body(0);
        for (auto &x : th) x.join();
        return;
    }
    {
        std::lock_guard<std::mutex> g(P.mu);
        if (!P.pin_known) { P.pin = numa_node_cpus(&P.pin_set); P.pin_known = true; }
        while ((int)P.workers.size() < n - 1) {
            const int id = (int)P.workers.size();
            P.workers.emplace_back([&P, id] { P.loop(id); });
            if (P.pin) (void)pthread_setaffinity_np(P.workers.back().native_handle(), sizeof(P.pin_set), &P.pin_set);
        }
        P.body = &body;
        P.want = n - 1;
        P.running = n - 1;
        ++P.generation;
    }
    P.cv_job.notify_all();
    body(0);
    std::unique_lock<std::mutex> lk(P.mu);
    P.cv_done.wait(lk, [&] { return P.running == 0; });
    P.body = nullptr;
}

// ---- how many host threads a parallel phase should use ---------------------------------------------------------------
// hardware_concurrency() reports the machine; a container is often confined to far less CPU TIME by a cgroup bandwidth quota
// (cpu.max / cpu.cfs_quota_us): with a quota of 16 CPUs, 256 runnable threads exhaust a 100 ms period's budget in 6 ms and
// are then all throttled until the next period -- measured here: every phase got SLOWER beyond 32 threads.  Default = twice
// the quota: a call may burst above the long-run rate, but not past one period's budget -- with 2.5 x (40 threads) the SAM call
// got faster (one box, six calls each, 0.4 s apart: 45-57 ms at 32 threads, 40-45 ms at 40) and the BAM call, whose 1.5 CPU-seconds
// became 1.8 with the extra threads, crossed the 1.6 CPU-seconds a period grants and was frozen inside every call (88 -> 119 ms);
// at most the hardware threads; HGX_THREADS overrides.
#include <cstdio>
#include <cstdlib>
int hgx_default_threads() {
    static const int cached = [] {
        if (const char *e = getenv("HGX_THREADS")) { const int v = atoi(e); if (v > 0) return std::min(v, 512); }
        int hw = (int)std::thread::hardware_concurrency();
        if (hw <= 0) hw = 1;
        double quota_cpus = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
            char q[64];
            long period = 0;
            if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) quota_cpus = atof(q) / (double)period;
            fclose(f);
        } else {
            long quota = -1, period = 0;                                           // cgroup v1
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &period) != 1) period = 0; fclose(g); }
            if (quota > 0 && period > 0) quota_cpus = (double)quota / (double)period;
        }
        int n = hw;
        if (quota_cpus > 0) n = std::min(hw, std::max(1, (int)(2.0 * quota_cpus + 0.5)));
        return std::min(n, 512);
    }();
    return cached;
}


// ------------------------------------------------------------------------------------------------
// The dict side of the single_abundance seam (typing_common.py:1282-1305): class keys '-'.join(sorted(names)) -> name table in
// first-appearance order + one bit row per key.  One pass over the text; the names are views into the caller's buffer until the
// set is built, then copied.
// ------------------------------------------------------------------------------------------------
struct hgx_keyset {
    std::vector<std::string> names;
    std::vector<int32_t> key_off{0}, key_name;       // per key: indices into names
    int32_t n_keys = 0, a_pad = 0, keys_sorted = 1;
    size_t pool_bytes = 0;
};

extern "C" int hgx_keyset_create(hgx_keyset **out, const char *keys, size_t n_bytes, int32_t n_keys) {
    HARGCHK(out && n_keys >= 0 && (keys || n_bytes == 0));
    *out = nullptr;
    hgx_keyset *ks = new hgx_keyset();
    ks->n_keys = n_keys;
    std::unordered_map<std::string_view, int32_t> index;
    index.reserve(1 << 14);
    std::vector<std::string_view> order;
    size_t at = 0;
    for (int32_t k = 0; k < n_keys; ++k) {
        if (at > n_bytes) { delete ks; hgx_set_error("hgx_keyset_create: %d keys announced, the text holds %d", n_keys, k); return HGX_EINVAL; }
        const char *line = keys + at;
        const char *nl = (const char *)memchr(line, '\n', n_bytes - at);
        const size_t len = nl ? (size_t)(nl - line) : n_bytes - at;
        std::string_view prev;
        bool first = true;
        size_t s = 0;
        while (true) {                                             // "".split("-") is [""]: every key has at least one name
            const char *dash = (const char *)memchr(line + s, '-', len - s);
            const size_t e = dash ? (size_t)(dash - line) : len;
            std::string_view nm(line + s, e - s);
            auto it = index.find(nm);
            int32_t id;
            if (it == index.end()) {
                id = (int32_t)order.size();
                index.emplace(nm, id);
                order.push_back(nm);
            } else {
                id = it->second;
            }
            ks->key_name.push_back(id);
            if (!first && nm < prev) ks->keys_sorted = 0;            // (bytewise = code-point order for UTF-8: Python's str order)
            prev = nm;
            first = false;
            if (!dash) break;
            s = e + 1;
        }
        ks->key_off.push_back((int32_t)ks->key_name.size());
        at += len + 1;
    }
    ks->names.reserve(order.size());
    for (auto v : order) { ks->names.emplace_back(v); ks->pool_bytes += v.size() + 1; }
    ks->a_pad = hgx_a_pad((int32_t)std::max<size_t>(order.size(), 1));
    *out = ks;
    return HGX_OK;
}

extern "C" int hgx_keyset_dims(const hgx_keyset *ks, int32_t *n_names, int32_t *a_pad, size_t *name_pool_bytes, int32_t *keys_sorted) {
    HARGCHK(ks);
    if (n_names) *n_names = (int32_t)ks->names.size();
    if (a_pad) *a_pad = ks->a_pad;
    if (name_pool_bytes) *name_pool_bytes = ks->pool_bytes;
    if (keys_sorted) *keys_sorted = ks->keys_sorted;
    return HGX_OK;
}

extern "C" int hgx_keyset_fill(const hgx_keyset *ks, uint64_t *bits, char *name_pool, int32_t *name_rank) {
    HARGCHK(ks);
    const int w64 = ks->a_pad / 64;
    if (bits) {
        memset(bits, 0, (size_t)ks->n_keys * w64 * 8);
        for (int32_t k = 0; k < ks->n_keys; ++k) {
            uint64_t *row = bits + (size_t)k * w64;
            for (int32_t q = ks->key_off[k]; q < ks->key_off[k + 1]; ++q) row[ks->key_name[q] >> 6] |= 1ull << (ks->key_name[q] & 63);
        }
    }
    if (name_pool) {
        char *p = name_pool;
        for (const std::string &n : ks->names) { memcpy(p, n.data(), n.size()); p[n.size()] = 0; p += n.size() + 1; }
    }
    if (name_rank) {
        std::vector<int32_t> by(ks->names.size());
        std::iota(by.begin(), by.end(), 0);
        std::sort(by.begin(), by.end(), [&](int32_t a, int32_t b) { return ks->names[a] < ks->names[b]; });
        for (size_t r = 0; r < by.size(); ++r) name_rank[by[r]] = (int32_t)r;
    }
    return HGX_OK;
}

extern "C" int hgx_keyset_destroy(hgx_keyset *ks) { delete ks; return HGX_OK; }
