// hgx_front_host.cpp -- host side of the device front end: the flat tables its kernels read (alternatives with the
// reference's spellings, variant names, exons), and -- in the LAB build only (-DHGX_LAB, libhgx_lab.so) -- an emulation that
// runs the kernels' per-key / per-pair functions (hgx_front_core.hpp) as loops on the CPU, so that the CPU test-suite can
// compare that logic with the pinned host front end where no GPU exists.  Nothing in the product library calls the emulation.
#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>

#include "hgx_internal.hpp"

int hgx_front_tables_build(hgx_locus &L, hgx_front_tables &T) {
    const int rc = hgx_build_alternatives(L);
    if (rc) return rc;
    T = hgx_front_tables();
    T.usable = true;
    for (auto &e : L.exons) { T.exons.push_back(e[0]); T.exons.push_back(e[1]); }
    T.name_off.push_back(0);
    for (int v = 0; v < L.V; ++v) {
        const std::string &nm = L.name[v];
        // the kernels resolve Zs ids through the "hv<n>" table and spell novel ids with an 'n': names must be hv<digits>
        bool ok = nm.size() > 2 && nm[0] == 'h' && nm[1] == 'v';
        for (size_t k = 2; k < nm.size() && ok; ++k) ok = nm[k] >= '0' && nm[k] <= '9';
        if (!ok && T.usable) { T.usable = false; T.why = "variant id '" + nm + "' is not hv<digits>"; }
        T.name_pool.insert(T.name_pool.end(), nm.begin(), nm.end());
        T.name_off.push_back((int32_t)T.name_pool.size());
    }
    if (L.backbone.size() > 0x1ffff) { T.usable = false; T.why = "backbone longer than the novel-id encoding's 17 position bits"; }
    for (int dir = 0; dir < 2; ++dir) {                             // the keys first: as integers and as the reference spells them
        const std::vector<AltEntry> &src = dir == 0 ? L.alts_left : L.alts_right;
        T.alt_key_off[dir].push_back((int32_t)T.alt_ints.size());
        T.alt_str_off[dir].push_back((int32_t)T.alt_chars.size());
        for (const AltEntry &e : src) {
            T.alt_anchor[dir].push_back(dir == 0 ? e.key.right : e.key.left);
            T.alt_ints.push_back(e.key.left);
            T.alt_ints.insert(T.alt_ints.end(), e.key.vars.begin(), e.key.vars.end());
            T.alt_ints.push_back(e.key.right);
            T.alt_key_off[dir].push_back((int32_t)T.alt_ints.size());
            std::string s = std::to_string(e.key.left);             // "529-hv8-hv22-606" (typing_common.py:1421)
            for (int v : e.key.vars) { s += '-'; s += L.name[v]; }
            s += '-';
            s += std::to_string(e.key.right);
            T.alt_chars.insert(T.alt_chars.end(), s.begin(), s.end());
            T.alt_str_off[dir].push_back((int32_t)T.alt_chars.size());
        }
    }
    T.alt_ht_off.push_back((int32_t)T.alt_ints.size());             // then the alternatives, back to back
    for (int dir = 0; dir < 2; ++dir) {
        const std::vector<AltEntry> &src = dir == 0 ? L.alts_left : L.alts_right;
        T.alt_list_off[dir].push_back((int32_t)T.alt_ht_off.size() - 1);
        for (const AltEntry &e : src) {
            for (const AltHt &a : e.alts) {
                T.alt_ints.push_back(a.left);
                T.alt_ints.insert(T.alt_ints.end(), a.vars.begin(), a.vars.end());
                T.alt_ints.push_back(a.right);
                T.alt_ht_off.push_back((int32_t)T.alt_ints.size());
            }
            T.alt_list_off[dir].push_back((int32_t)T.alt_ht_off.size() - 1);
        }
    }
    return HGX_OK;
}

FeLocus hgx_front_view(const hgx_locus &L, const hgx_front_tables &T) {
    FeLocus F;
    memset(&F, 0, sizeof(F));
    F.V = L.V;
    F.n_ref = (int32_t)L.backbone.size();
    F.base_kind = L.base_kind;
    F.n_exons = (int32_t)L.exons.size();
    F.n_hv = (int32_t)L.hv_index.size();
    F.pos = L.pos.data(); F.right = L.right.data(); F.len = L.len.data(); F.maxright = L.maxright.data();
    F.type = L.type.data(); F.linked = L.linked.data(); F.base = L.base.data();
    F.linked_bits = L.linked_bits.data();
    F.backbone = L.backbone.data();
    F.exons = T.exons.data();
    F.hv_index = L.hv_index.data();
    F.name_off = T.name_off.data();
    F.name_pool = T.name_pool.data();
    for (int d = 0; d < 2; ++d) {
        F.n_alt[d] = (int32_t)T.alt_anchor[d].size();
        F.alt_anchor[d] = T.alt_anchor[d].data();
        F.alt_key_off[d] = T.alt_key_off[d].data();
        F.alt_str_off[d] = T.alt_str_off[d].data();
        F.alt_list_off[d] = T.alt_list_off[d].data();
    }
    F.alt_ht_off = T.alt_ht_off.data();
    F.alt_ints = T.alt_ints.data();
    F.alt_chars = T.alt_chars.data();
    return F;
}

void hgx_front_trace_lines(const hgx_locus &L, const uint32_t *rec_info, size_t n_rec, const uint8_t *state, const uint32_t *trace_off,
                           const int32_t *pool, std::vector<std::string> &out) {
    static const char *const kType[] = {"match", "mismatch", "insertion", "deletion"};
    std::unordered_map<uint64_t, int> novel;                          // (variant type, pos, base | length) -> k of "nv<k>"
    auto nkey = [](int type, int pos, int key) { return ((uint64_t)(uint32_t)type << 56) ^ ((uint64_t)(uint32_t)pos << 24) ^ (uint64_t)(uint32_t)key; };
    auto vname = [&](int id) -> std::string {
        if (id == -1) return "unknown";
        if (id >= 0 && id < L.V) return L.name[id];
        // a novel indel: its id spells type, position and length (hgx_front_core.hpp)
        const int type = ((id >> 29) & 1) ? FE_VAR_DELETION : FE_VAR_INSERTION;
        auto it = novel.find(nkey(type, (id >> 12) & 0x1ffff, id & 0xfff));
        return it == novel.end() ? std::string("nv?") : "nv" + std::to_string(it->second);
    };
    auto join = [&](const int32_t *ids, int n) { std::string s; for (int i = 0; i < n; ++i) { if (i) s += '-'; s += vname(ids[i]); } return s; };
    std::vector<uint8_t> seen;
    for (size_t i = 0; i < n_rec; ++i) {
        const uint32_t slot = FE_REC_SLOT(rec_info[i]);
        if (state[slot] != 1 || trace_off[slot] == FE_NO_TRACE) continue;
        const int32_t *w = pool + trace_off[slot];
        const int n_nov = w[0], n_c = w[1], cleft = w[2], cright = w[3], n_l = w[4], n_r = w[5];
        w += FE_TRACE_HDR;
        for (int k = 0; k < n_nov; ++k, w += 3) novel.emplace(nkey(w[0], w[1], w[2]), (int)novel.size());
        std::string t;
        for (int k = 0; k < n_c; ++k, w += 4) {
            if (k) t += ',';
            const int type = w[0] & 3;
            t += kType[type];
            t += ':' + std::to_string(w[1]) + ':' + std::to_string(w[2]);
            if (type != FE_T_MATCH) t += ':' + vname(w[3]);
        }
        t += '\t' + std::to_string(cleft) + '\t' + std::to_string(cright) + '\t';
        std::vector<std::string> ls, rs;
        for (int a = 0; a < n_l; ++a) { ls.push_back(std::to_string(w[0]) + (w[1] ? "-" + join(w + 2, w[1]) : "")); w += 2 + w[1]; }
        for (int a = 0; a < n_r; ++a) { rs.push_back((w[1] ? join(w + 2, w[1]) + "-" : "") + std::to_string(w[0])); w += 2 + w[1]; }
        std::sort(ls.begin(), ls.end());
        std::sort(rs.begin(), rs.end());
        for (size_t k = 0; k < ls.size(); ++k) t += (k ? ";" : "") + ls[k];
        t += '\t';
        for (size_t k = 0; k < rs.size(); ++k) t += (k ? ";" : "") + rs[k];
        out.push_back(std::move(t));
    }
}

static_assert(FE_INTERDIST_HALF == HGX_INTERDIST_HALF && FE_INTERDIST_BINS == HGX_INTERDIST_BINS, "the kernels' histogram is the one the shards exchange");

#ifdef HGX_LAB
// ---- the device pipeline as loops (lab build only; mirrors hgx_front.hip stage by stage) ---------------------------------------
// *declined = 0 and *out = the batch, or *declined = the reason and no batch.
int hgx_front_emulate(hgx_batch **out, hgx_locus &L, const hgx_front_input &in, const hgx_parse_opts &opts, int *declined, int n_tasks,
                      hgx_front_totals *many) {
    *out = nullptr;
    *declined = 0;
    hgx_front_tables T;
    int rc = hgx_front_tables_build(L, T);
    if (rc) return rc;
    if (!T.usable) { *declined = HGX_FE_DECLINE_LOCUS; return HGX_OK; }
    const FeLocus F = hgx_front_view(L, T);
    const int n_ref = F.n_ref;
    n_tasks = std::max(1, n_tasks);
    if (n_tasks > 1 && opts.pileup_exchange) { *declined = HGX_FE_DECLINE_OPTS; return HGX_OK; }
    if (n_tasks > 65535) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
    hgx_batch *B = new hgx_batch();
    // k_fe_pileup (k_fe_pileup_many: a pileup per task) + k_fe_nt_set
    B->counts.assign((size_t)n_tasks * n_ref * 6, 0u);
    B->nt_set.assign((size_t)n_tasks * n_ref, 0);
    for (size_t k = 0; k < in.n_keys; ++k) {
        if (in.keys[k].n_pile == 0) continue;
        if ((int)in.keys[k].task >= n_tasks) { hgx_set_error("key of task %u in a batch of %d tasks", in.keys[k].task, n_tasks); delete B; return HGX_EINVAL; }
        uint32_t *cnt = B->counts.data() + (size_t)in.keys[k].task * n_ref * 6;
        const int r = fe_pileup_key(in.keys[k], in.text, n_ref, 0, 1, [cnt](uint32_t cell, uint32_t w) { cnt[cell] += w; });
        if (r < 0) { *declined = -r; delete B; return HGX_OK; }
    }
    if (opts.pileup_exchange && opts.pileup_exchange(opts.pileup_ctx, B->counts.data(), (int64_t)B->counts.size()) != 0) {
        hgx_set_error("pileup exchange between the ranks of a sharded locus failed");
        delete B;
        return HGX_EINVAL;
    }
    for (size_t i = 0; i < (size_t)n_tasks * n_ref; ++i) B->nt_set[i] = fe_nt_set(&B->counts[i * 6]);
    // k_fe_decode
    const size_t S = in.n_slots;
    std::vector<uint8_t> state(S, 0);
    std::vector<uint16_t> slot_task(S, 0);
    std::vector<uint32_t> key_ht_off(S, 0), key_n_ht(S, 0);
    std::vector<int32_t> ht_pool(std::max<size_t>(S * 48 + 4096, 16));
    const size_t cand_cap = S * 12 + 4096;
    std::vector<uint16_t> cand_lo(cand_cap), cand_nw(cand_cap);
    std::vector<uint64_t> cand_key(cand_cap);
    std::vector<uint32_t> cand_mask_off(cand_cap), mask_pool(cand_cap * 16);
    uint32_t cur[3] = {0, 0, 0};
    FePools pools;
    pools.ht_pool = ht_pool.data(); pools.ht_cap = (uint32_t)ht_pool.size(); pools.ht_cursor = &cur[0];
    pools.cand_lo = cand_lo.data(); pools.cand_nw = cand_nw.data(); pools.cand_key = cand_key.data();
    pools.cand_mask_off = cand_mask_off.data(); pools.cand_cap = (uint32_t)cand_cap; pools.cand_cursor = &cur[1];
    pools.mask_pool = mask_pool.data(); pools.mask_cap = (uint32_t)mask_pool.size(); pools.mask_cursor = &cur[2];
    std::vector<int32_t> trace_pool;
    std::vector<uint32_t> trace_off;
    uint32_t trace_cur = 0;
    pools.trace_pool = nullptr; pools.trace_cap = 0; pools.trace_cursor = &trace_cur; pools.key_trace_off = nullptr;
    if (opts.keep_trace && n_tasks == 1) {
        trace_pool.resize(S * 96 + 4096);
        trace_off.assign(std::max<size_t>(S, 1), FE_NO_TRACE);
        pools.trace_pool = trace_pool.data(); pools.trace_cap = (uint32_t)trace_pool.size(); pools.key_trace_off = trace_off.data();
    }
    FeParse po{opts.num_editdist, opts.error_correction};
    for (size_t k = 0; k < in.n_keys; ++k) {
        const FeKey &K = in.keys[k];
        if (K.slot == FE_NO_SLOT) continue;
        const FePile pile{B->nt_set.data() + (size_t)K.task * n_ref, B->counts.data() + (size_t)K.task * n_ref * 6};   // the key's own sample
        slot_task[K.slot] = (uint16_t)K.task;
        const int r = fe_key(F, po, pile, K, in.text, pools, state[K.slot], key_ht_off[K.slot], key_n_ht[K.slot]);
        if (r < 0) { *declined = -r; delete B; return HGX_OK; }
    }
    // candidate pieces -> distinct pieces (sort by key, run heads, word-for-word check) -> canonical order of the heads: first word,
    // width, PieceTable::hash, bytes (hgx_canonical_piece_order)
    const uint32_t n_cand = cur[1];
    std::vector<uint32_t> order(n_cand);
    for (uint32_t c = 0; c < n_cand; ++c) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cand_key[a] < cand_key[b]; });
    std::vector<uint32_t> head_of(n_cand), heads;
    for (uint32_t k = 0; k < n_cand; ++k) {
        const uint32_t c = order[k];
        if (k > 0 && cand_key[c] == cand_key[order[k - 1]]) {
            const uint32_t h = heads.back();
            if (cand_lo[c] != cand_lo[h] || cand_nw[c] != cand_nw[h] ||
                memcmp(&mask_pool[cand_mask_off[c]], &mask_pool[cand_mask_off[h]], 8 * (size_t)cand_nw[c]) != 0) {
                *declined = HGX_FE_DECLINE_COLLISION;
                delete B;
                return HGX_OK;
            }
        } else heads.push_back(c);
        head_of[c] = (uint32_t)heads.size() - 1;
    }
    std::vector<uint32_t> hord(heads.size());
    std::vector<uint64_t> hhash(heads.size());
    for (size_t k = 0; k < heads.size(); ++k) {
        hord[k] = (uint32_t)k;
        hhash[k] = fe_piece_hash(cand_lo[heads[k]], cand_nw[heads[k]], &mask_pool[cand_mask_off[heads[k]]]);
    }
    std::sort(hord.begin(), hord.end(), [&](uint32_t a, uint32_t b) {
        const uint32_t ca = heads[a], cb = heads[b];
        if (cand_lo[ca] != cand_lo[cb]) return cand_lo[ca] < cand_lo[cb];
        if (cand_nw[ca] != cand_nw[cb]) return cand_nw[ca] < cand_nw[cb];
        if (hhash[a] != hhash[b]) return hhash[a] < hhash[b];
        return memcmp(&mask_pool[cand_mask_off[ca]], &mask_pool[cand_mask_off[cb]], 8 * (size_t)cand_nw[ca]) < 0;
    });
    std::vector<uint32_t> new_id(heads.size());
    B->pieces.resize(heads.size());
    for (size_t k = 0; k < hord.size(); ++k) {
        const uint32_t c = heads[hord[k]];
        new_id[hord[k]] = (uint32_t)k;
        hgx_piece pc;
        pc.mask_off = (uint32_t)B->masks.size();
        pc.lo_word = cand_lo[c];
        pc.n_words = cand_nw[c];
        B->pieces[k] = pc;
        B->masks.insert(B->masks.end(), &mask_pool[cand_mask_off[c]], &mask_pool[cand_mask_off[c]] + 2 * (size_t)cand_nw[c]);
    }
    // k_fe_task_pieces: the distinct pieces each task's decoded keys produced (what the task's own batch would hold)
    if (many) {
        many->reads.assign((size_t)n_tasks, 0); many->pairs.assign((size_t)n_tasks, 0); many->refs.assign((size_t)n_tasks, 0);
        many->pieces.assign((size_t)n_tasks, 0);
        std::vector<std::vector<uint8_t>> seen((size_t)n_tasks, std::vector<uint8_t>(heads.size(), 0));
        for (size_t sl = 0; sl < S; ++sl) {
            if (state[sl] != 1) continue;
            uint32_t at = key_ht_off[sl];
            for (uint32_t x = 0; x < key_n_ht[sl]; ++x) {
                const int32_t *rec = &ht_pool[at];
                for (int e = 0; e <= rec[3]; ++e) {
                    uint8_t &b = seen[slot_task[sl]][new_id[head_of[(uint32_t)rec[4] + e]]];
                    if (!b) { b = 1; many->pieces[slot_task[sl]]++; }
                }
                at += FE_HT_HDR + (uint32_t)rec[2];
            }
        }
    }
    // CODIS D18S51 (front_stages of hgx_front.hip: after the pileup exchange): the sample's expected inner distance from the histogram
    // of this stream's distances -- summed over the shards of the locus first --, then choose_pairs on the stream's last pair
    long long expected = -1;
    size_t choose_at = (size_t)-1;
    if (in.want_interdist) {
        if (n_tasks > 1 || in.interdist_hist.size() != (size_t)HGX_INTERDIST_BINS) { *declined = HGX_FE_DECLINE_OPTS; delete B; return HGX_OK; }
        std::vector<int64_t> hist(in.interdist_hist);
        if (opts.interdist_exchange && opts.interdist_exchange(opts.interdist_ctx, hist.data(), (int64_t)hist.size()) != 0) {
            hgx_set_error("inter-distance exchange between the ranks of a sharded locus failed");
            delete B;
            return HGX_EINVAL;
        }
        if (hgx_interdist_median(hist.data(), &expected)) { *declined = HGX_FE_DECLINE_SIZE; delete B; return HGX_OK; }
        if (opts.codis_choose_pairs)                                   // k_fe_pair_count's atomicMax: the last run that yields a pair
            for (size_t i = 0; i < in.n_rec; ++i) {
                if (!FE_REC_HEAD(in.rec_info[i])) continue;
                uint32_t uni[FE_MAX_PAIR_HT];
                int n_uni = 0;
                if (fe_pair_union(in.rec_info, (uint32_t)i, (uint32_t)in.n_rec, state.data(), key_ht_off.data(), key_n_ht.data(), ht_pool.data(), uni, n_uni) > 0)
                    choose_at = i;
            }
    }
    // k_fe_pairs: count, scan, emit
    int64_t n_reads = 0;
    for (size_t i = 0; i < in.n_rec; ++i) {
        if (!FE_REC_HEAD(in.rec_info[i])) continue;
        uint32_t uni[FE_MAX_PAIR_HT];
        int n_uni = 0;
        const int ns = fe_pair_union(in.rec_info, (uint32_t)i, (uint32_t)in.n_rec, state.data(), key_ht_off.data(), key_n_ht.data(),
                                     ht_pool.data(), uni, n_uni, i == choose_at, expected);
        if (ns < 0) { *declined = -ns; delete B; return HGX_OK; }
        if (ns == 0) continue;
        n_reads += ns;
        size_t n_exon = 0;
        for (int x = 0; x < n_uni; ++x) n_exon += (size_t)ht_pool[uni[x] + 3];
        if (n_exon > 65535 || n_uni > 65535) { *declined = -FE_E_PAIR; delete B; return HGX_OK; }
        if (many) {
            const uint32_t t = slot_task[FE_REC_SLOT(in.rec_info[i])];
            many->reads[t] += (uint32_t)ns; many->pairs[t] += 1; many->refs[t] += n_exon + (size_t)n_uni;
        }
        for (int x = 0; x < n_uni; ++x) {
            const int32_t *rec = &ht_pool[uni[x]];
            for (int e = 0; e < rec[3]; ++e) B->pair_ref.push_back(new_id[head_of[(uint32_t)rec[4] + e]]);
        }
        for (int x = 0; x < n_uni; ++x) {
            const int32_t *rec = &ht_pool[uni[x]];
            B->pair_ref.push_back(new_id[head_of[(uint32_t)rec[4] + rec[3]]] | 0x80000000u);
        }
        B->pair_off.push_back((int32_t)B->pair_ref.size());
    }
    B->n_reads = (int32_t)n_reads;
    if (pools.trace_pool) {
        std::vector<std::string> lines;
        hgx_front_trace_lines(L, in.rec_info, in.n_rec, state.data(), trace_off.data(), trace_pool.data(), lines);
        for (auto &l : lines) B->trace.push_back(TraceRec{std::move(l)});
    }
    *out = B;
    return HGX_OK;
}
#endif

#ifdef HGX_LAB
// ---- the record stage as loops (mirrors k_fe_records .. k_fe_build of hgx_front.hip), then the key stages above -----------------
int hgx_front_emulate_records(hgx_batch **out, hgx_locus &L, const char *raw, size_t raw_bytes, const FeLine *lines, size_t n, bool binary,
                              const hgx_parse_opts &o, int *declined, int n_tasks, hgx_front_totals *many) {
    *out = nullptr;
    *declined = 0;
    if (raw_bytes >= (1ull << 32) - 64 || n >= (1ull << 30)) { *declined = HGX_FE_DECLINE_SIZE; return HGX_OK; }
    // As on the device, a record that cannot be taken apart does not stop the stage: it is made inert (FE_R_FAILED), the first
    // decline code is kept, and the call declines after the grouping -- so that the CPU suite walks the path the kernels walk.
    int first_decline = 0;
    std::vector<FeRec> recs(n);
    for (size_t i = 0; i < n; ++i) {
        const int r = binary ? fe_parse_bam_record(raw, lines[i].off, lines[i].len, o.simulation != 0, lines[i].task, recs[i])
                             : fe_parse_text_record(raw, raw_bytes, lines[i].off, lines[i].len, o.simulation != 0, lines[i].task, recs[i]);
        if (r < 0) {
            if (!first_decline) first_decline = -r;
            recs[i] = FeRec{};
            recs[i].bits = FE_R_FAILED;
            recs[i].flag = 4;
            recs[i].task = (uint16_t)lines[i].task;
        }
    }
    std::vector<uint8_t> head(n, 0), kept(n, 0), pm(n, 0);
    for (size_t i = 0; i < n; ++i) head[i] = i == 0 || !fe_same_read_id(recs[i - 1], recs[i], raw);
    const FeFilter flt{o.num_editdist, o.allow_discordant, o.base_locus};
    for (size_t i = 0; i < n; ++i) {
        int k = fe_rec_kept(recs.data(), head.data(), (uint32_t)i, flt);
        if (k < 0) { if (!first_decline) first_decline = -k; k = 0; }
        kept[i] = (uint8_t)k;
        pm[i] = fe_rec_in_pileup(recs[i], flt) ? 1 : 0;
    }
    // grouping by decode key: hash table on the 64-bit key, first record of a key = its representative, exact check against it
    std::unordered_map<uint64_t, uint32_t> slot_of_key;
    std::vector<uint32_t> rep, n_pile, slot_of(n);
    std::vector<uint8_t> any_kept;
    for (size_t i = 0; i < n; ++i) {
        auto it = slot_of_key.find(recs[i].key);
        uint32_t s;
        if (it == slot_of_key.end()) {
            s = (uint32_t)rep.size();
            slot_of_key.emplace(recs[i].key, s);
            rep.push_back((uint32_t)i); n_pile.push_back(0); any_kept.push_back(0);
        } else {
            s = it->second;
            if (!fe_rec_same_key(recs[rep[s]], recs[i], raw) && !first_decline) first_decline = HGX_FE_DECLINE_COLLISION;
        }
        slot_of[i] = s;
        n_pile[s] += pm[i];
        any_kept[s] |= kept[i];
    }
    // k_fe_interdist_*: the counted records compacted, one distance per run of exactly two that another counted record follows
    std::vector<int64_t> hist;
    const bool want_interdist = o.codis_choose_pairs || o.interdist_exchange;
    if (want_interdist && !first_decline) {
        hist.assign((size_t)HGX_INTERDIST_BINS, 0);
        std::vector<uint32_t> comp;
        for (size_t i = 0; i < n; ++i) if (fe_rec_in_interdist(recs[i])) comp.push_back((uint32_t)i);
        const size_t m = comp.size();
        for (size_t j = 0; j + 2 < m; ++j) {
            const FeRec &a = recs[comp[j]], &b = recs[comp[j + 1]], &c = recs[comp[j + 2]];
            if (j > 0 && fe_same_read_id(recs[comp[j - 1]], a, raw)) continue;
            if (!fe_same_read_id(a, b, raw) || fe_same_read_id(a, c, raw)) continue;
            long long d;
            const int r = fe_interdist_of(a, b, raw, d);
            if (r < 0) { first_decline = -r; break; }
            hist[fe_interdist_bin(d)] += 1;
        }
    }
    if (first_decline) { *declined = first_decline; return HGX_OK; }
    hgx_front_input in;
    in.want_interdist = want_interdist;
    in.interdist_hist.swap(hist);
    in.mem = hgx_front_alloc{[](size_t b) { return hgx_host_alloc(b); }, [](void *p) { hgx_host_free(p); }};
    in.text = const_cast<char *>(raw);
    in.text_borrowed = true;
    in.n_text = raw_bytes;
    std::vector<uint32_t> dslot(rep.size(), FE_NO_SLOT);
    size_t nk = 0, ns = 0, nr = 0;
    for (size_t s = 0; s < rep.size(); ++s) if (n_pile[s] > 0 || any_kept[s]) ++nk;
    for (size_t i = 0; i < n; ++i) nr += kept[i];
    in.keys = (FeKey *)in.mem.alloc(std::max<size_t>(nk, 1) * sizeof(FeKey));
    in.rec_info = (uint32_t *)in.mem.alloc(std::max<size_t>(nr, 1) * 4);
    for (size_t s = 0, k = 0; s < rep.size(); ++s) {                 // (slots are in stream order of their first records here)
        if (!(n_pile[s] > 0 || any_kept[s])) continue;
        const FeRec &f = recs[rep[s]];
        FeKey &K = in.keys[k++];
        K.pos = f.pos - (o.base_locus + 1);
        K.n_pile = n_pile[s];
        K.slot = any_kept[s] ? (uint32_t)ns++ : FE_NO_SLOT;
        dslot[s] = K.slot;
        K.cigar_off = f.cigar_off; K.seq_off = f.seq_off; K.zs_off = f.zs_off; K.md_off = f.md_off;
        K.seq_len = f.seq_len; K.cigar_len = f.cigar_len; K.zs_len = f.zs_len; K.md_len = f.md_len;
        K.flags = (uint16_t)(((f.bits & FE_R_HAS_ZS) ? FE_K_HAS_ZS : 0) | ((f.bits & FE_R_HAS_MD) ? FE_K_HAS_MD : 0) |
                             ((f.bits & FE_R_BIN) ? (FE_K_BIN_CIGAR | FE_K_PACKED_SEQ) : 0));
        K.task = f.task;
    }
    in.n_keys = nk;
    in.n_slots = ns;
    size_t prev = (size_t)-1, k = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!kept[i]) continue;
        const bool hd = prev == (size_t)-1 || !fe_same_read_id(recs[prev], recs[i], raw);
        in.rec_info[k++] = dslot[slot_of[i]] | ((recs[i].flag & 0x40) ? 1u << 30 : 0u) | (hd ? 1u << 31 : 0u);
        prev = i;
    }
    in.n_rec = nr;
    return hgx_front_emulate(out, L, in, o, declined, n_tasks, many);
}
#endif

#ifdef HGX_LAB
// fe_key_contains (the kernels' form of key.find(cur_join) != -1, typing_common.py:1744 / 1868, on variant ids) beside the text
// search it stands for, on a made-up table: names = '\n'-joined variant names ("hv12\nhv3\n..."; id = line number), key / cur = ids.
// out[0] = the id form, out[1] = the text form.
extern "C" int hgx_lab_key_contains(const char *names, const int32_t *key, int32_t n_key, const int32_t *cur, int32_t n_cur, int32_t *out) {
    HARGCHK(names && key && cur && out && n_cur > 0 && n_key >= 0);
    std::vector<int32_t> off{0};
    std::vector<char> pool;
    std::vector<std::string> nm;
    for (const char *p = names; *p;) {
        const char *e = strchr(p, '\n');
        if (!e) e = p + strlen(p);
        nm.emplace_back(p, e);
        pool.insert(pool.end(), p, e);
        off.push_back((int32_t)pool.size());
        p = *e ? e + 1 : e;
    }
    FeLocus F;
    memset(&F, 0, sizeof(F));
    F.V = (int32_t)nm.size();
    F.name_off = off.data();
    F.name_pool = pool.data();
    for (int k = 0; k < n_key; ++k) HARGCHK(key[k] >= 0 && key[k] < F.V);
    out[0] = fe_key_contains(F, key, n_key, cur, n_cur) ? 1 : 0;
    std::string hay = "529", needle;
    for (int k = 0; k < n_key; ++k) hay += "-" + nm[(size_t)key[k]];
    hay += "-606";
    for (int k = 0; k < n_cur; ++k) needle += (k ? "-" : "") + (cur[k] >= 0 && cur[k] < F.V ? nm[(size_t)cur[k]] : std::string("nv") + std::to_string(k));
    out[1] = hay.find(needle) != std::string::npos ? 1 : 0;
    return HGX_OK;
}

// SAM text -> batch through the emulated device stages (lab library only; tests/test_front_emulation.py).  *declined != 0: the
// device path would hand this input to the host stages, which then produced the batch.
extern "C" int hgx_lab_parse_sam_emulated(hgx_batch **out, const hgx_locus *loc, const char *sam, size_t n_bytes, const hgx_parse_opts *opts,
                                          int32_t *declined) {
    HARGCHK(out && loc && opts && declined);
    hgx_batch *made = nullptr;
    hgx_front_hook hook;
    hook.mem = hgx_front_alloc{[](size_t n) { return hgx_host_alloc(n); }, [](void *p) { hgx_host_free(p); }};
    hook.run = [&](hgx_locus &L, const hgx_front_input &in, const hgx_parse_opts &o, int *dec) { return hgx_front_emulate(&made, L, in, o, dec); };
    const int rc = hgx_parse_sam_hook(out, loc, sam, n_bytes, opts, &hook);
    *declined = hook.declined;
    if (rc) { delete made; return rc; }
    if (!hook.declined) *out = made;
    else delete made;
    return HGX_OK;
}

// the same with the RECORD stage emulated too (fields, filters, key grouping: the device takes the records themselves);
// path != NULL: an alignment file (SAM text or BAM) instead of text.  declined[0] = record route, declined[1] = key route.
extern "C" int hgx_lab_parse_records_emulated(hgx_batch **out, const hgx_locus *loc, const char *sam, size_t n_bytes, const char *path,
                                              const char *regions, const hgx_parse_opts *opts, int32_t *declined) {
    HARGCHK(out && loc && opts && declined && (sam || path || n_bytes == 0));
    hgx_batch *made = nullptr;
    hgx_front_hook hook;
    hook.mem = hgx_front_alloc{[](size_t n) { return hgx_host_alloc(n); }, [](void *p) { hgx_host_free(p); }};
    hook.run = [&](hgx_locus &L, const hgx_front_input &in, const hgx_parse_opts &o, int *dec) {
        delete made; made = nullptr;
        return hgx_front_emulate(&made, L, in, o, dec);
    };
    hook.records = [&](hgx_locus &L, const char *raw, size_t raw_bytes, const hgx_line *lines, size_t n, bool binary, const hgx_parse_opts &o, int *dec,
                       const hgx_bam_deferred *) {
        std::vector<FeLine> refs(n);
        for (size_t i = 0; i < n; ++i) refs[i] = FeLine{(uint32_t)((size_t)(lines[i].p - raw) - (binary ? 32u : 0u)), lines[i].len, 0u};
        return hgx_front_emulate_records(&made, L, raw, raw_bytes, refs.data(), n, binary, o, dec);
    };
    const int rc = path ? hgx_parse_alignment_file_hook(out, loc, path, regions, opts, &hook) : hgx_parse_sam_hook(out, loc, sam, n_bytes, opts, &hook);
    declined[0] = hook.declined_records;
    declined[1] = hook.declined;
    if (rc) { delete made; return rc; }
    if (!*out) *out = made;
    else delete made;
    return HGX_OK;
}
// MANY tasks of one locus through the emulated record route (hgx_front_many_dev's stages as loops): the tasks' streams are
// concatenated at their 64-byte aligned bases (the gaps filled with a byte no record holds), every line carries its task.
// out = the merged batch; pair_base [n_tasks + 1], n_reads / n_pieces / n_refs [n_tasks].
extern "C" int hgx_lab_many_emulated(hgx_batch **out, const hgx_locus *loc, const char *const *paths, const char *const *regions,
                                     const char *const *sams, const size_t *sam_bytes, int32_t n_tasks, const hgx_parse_opts *opts,
                                     int32_t *pair_base, int32_t *n_reads, int32_t *n_pieces, int64_t *n_refs, int32_t *declined) {
    HARGCHK(out && loc && opts && declined && n_tasks >= 1 && (paths || (sams && sam_bytes)));
    *out = nullptr;
    *declined = 0;
    if (opts->keep_trace || opts->codis_choose_pairs || opts->interdist_exchange || opts->pileup_exchange) { *declined = HGX_FE_DECLINE_OPTS; return HGX_OK; }   // (traces: one task per call)
    hgx_many_streams ms;
    int rc = hgx_many_read(ms, paths, regions, sams, sam_bytes, n_tasks, opts->n_threads, nullptr);
    if (rc) return rc;
    if (ms.mixed) { *declined = HGX_FE_DECLINE_OPTS; return HGX_OK; }
    const size_t total = ms.base[(size_t)n_tasks], n_lines = ms.line_base[(size_t)n_tasks];
    std::vector<char> text(total + 64, (char)0xAA);
    for (int t = 0; t < n_tasks; ++t) if (ms.raw_bytes[t]) memcpy(&text[ms.base[t]], ms.raw[t], ms.raw_bytes[t]);
    std::vector<FeLine> lines(std::max<size_t>(n_lines, 1));
    hgx_many_lines(ms, lines.data(), opts->n_threads);
    hgx_front_totals tot;
    int dec = 0;
    hgx_batch *made = nullptr;
    rc = hgx_front_emulate_records(&made, *const_cast<hgx_locus *>(loc), text.data(), total, lines.data(), n_lines, ms.binary, *opts, &dec, n_tasks, &tot);
    if (rc) { delete made; return rc; }
    if (dec) { delete made; *declined = dec; return HGX_OK; }
    int32_t base = 0;
    for (int t = 0; t < n_tasks; ++t) {
        if (pair_base) pair_base[t] = base;
        base += (int32_t)tot.pairs[t];
        if (n_reads) n_reads[t] = (int32_t)tot.reads[t];
        if (n_pieces) n_pieces[t] = (int32_t)tot.pieces[t];
        if (n_refs) n_refs[t] = (int64_t)tot.refs[t];
    }
    if (pair_base) pair_base[n_tasks] = base;
    *out = made;
    return HGX_OK;
}

// hgx_batch_merge for the CPU tests (the product reaches it through hgx_many_create, which needs a device)
extern "C" int hgx_lab_batch_merge(hgx_batch **out, const hgx_batch *const *batches, int32_t n, int32_t *pair_base) {
    return hgx_batch_merge(out, batches, n, pair_base);
}
#endif
