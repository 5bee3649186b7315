// hgx_sam.cpp -- host front-end of libhgx: SAM records -> haplotypes -> pieces (8a-1 .. 8a-5 host half).
//
// Replaces, for one locus, the streaming part of typing() that precedes scoring:
//   get_mpileup                 hisatgenotype_typing_common.py:1059-1134
//   record decode + filters     hisatgenotype_typing_core.py:800-898
//   CIGAR x MD x Zs walk        hisatgenotype_typing_core.py:899-1124
//   error_correct               hisatgenotype_typing_core.py:119-243
//   novel variants, cmp_list2   hisatgenotype_typing_core.py:404-431, 1126-1164, 1351-1368
//   get_alternatives            hisatgenotype_typing_common.py:1424-1657
//   identify_ambigious_diffs    hisatgenotype_typing_common.py:1663-1955
//   haplotype assembly          hisatgenotype_typing_core.py:1386-1406
//   get_exon_haplotypes         hisatgenotype_typing_core.py:718-792
//   pair protocol               hisatgenotype_typing_core.py:1238-1347, 1545-1587
// Variant ids are integers: [0,V) known ("hv*"), V+k the k-th novel variant ("nv<k>"), -1 "unknown".
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <stdexcept>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <string>
#include <thread>
#include <unordered_set>

#include "hgx_internal.hpp"

namespace {

struct RefError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

enum { T_MATCH = 0, T_MISMATCH = 1, T_INSERTION = 2, T_DELETION = 3 };
static const char *kTypeName[] = {"match", "mismatch", "insertion", "deletion"};

struct Cmp {
    int type, pos, len, id;   // id: -2 n/a (match), -1 unknown, [0,V) known, >= V novel
};

struct Novel {
    int type, pos, len;       // HGX_VAR_*
    char base;
    std::string ins;
};

// Novel variants of one parse (core:404-431), shared by all workers: id = V + k.  Every sequencing error that survives error
// correction becomes one (a novel single), so the table sees a lookup or an insert for a good part of the distinct reads:
// 256 shards keyed by position take the locking apart; entries live in fixed segments, so `get` needs no lock (an id is only
// ever used by a thread that obtained it from find / get_or_add, i.e. after the entry was published under a shard lock).
struct NovelTable {
    static constexpr int SHARDS = 256, SEG = 4096, MAX_SEG = 16384;
    struct Shard { std::mutex mu; std::unordered_map<uint64_t, int> map; };
    Shard shard[SHARDS];
    std::atomic<Novel *> seg[MAX_SEG];
    std::atomic<int> count{0};
    std::mutex grow;
    NovelTable() { for (auto &p : seg) p.store(nullptr, std::memory_order_relaxed); }
    NovelTable(const NovelTable &) = delete;
    ~NovelTable() { for (auto &p : seg) delete[] p.load(std::memory_order_relaxed); }
    static uint64_t key(int type, int pos, int k) { return ((uint64_t)type << 60) | ((uint64_t)(uint32_t)pos << 24) | (uint32_t)(k & 0xffffff); }
    Shard &of(uint64_t k) { return shard[(k >> 24) & (SHARDS - 1)]; }                    // by position
    const Novel &get(int k) const { return seg[k / SEG].load(std::memory_order_acquire)[k % SEG]; }
    int find(uint64_t k) {
        if (count.load(std::memory_order_acquire) == 0) return -1;
        Shard &s = of(k);
        std::lock_guard<std::mutex> g(s.mu);
        auto it = s.map.find(k);
        return it == s.map.end() ? -1 : it->second;
    }
    int get_or_add(uint64_t k, Novel &&nv) {
        Shard &s = of(k);
        std::lock_guard<std::mutex> g(s.mu);
        auto it = s.map.find(k);
        if (it != s.map.end()) return it->second;
        const int id = count.fetch_add(1, std::memory_order_acq_rel);
        if (id / SEG >= MAX_SEG) throw std::runtime_error("too many novel variants");
        if (!seg[id / SEG].load(std::memory_order_acquire)) {
            std::lock_guard<std::mutex> gg(grow);
            if (!seg[id / SEG].load(std::memory_order_acquire)) seg[id / SEG].store(new Novel[SEG], std::memory_order_release);
        }
        seg[id / SEG].load(std::memory_order_acquire)[id % SEG] = std::move(nv);
        s.map.emplace(k, id);
        return id;
    }
};

// ------------------------------------------------------------------------------------------------
// alternatives (get_alternatives)
// ------------------------------------------------------------------------------------------------
typedef std::vector<int32_t> HtVec;   // [left, id..., right]

struct AltBuilder {
    const hgx_locus &L;
    std::unordered_set<uint64_t> second;
    std::vector<std::pair<int, int>> rev;   // (right-end key, var) sorted by key, stable
    std::vector<int> rev_key;
    // key spelling -> set of alternative spellings, per direction
    std::vector<std::pair<HtVec, std::vector<HtVec>>> table[2];
    std::unordered_map<std::string, size_t> index[2];

    explicit AltBuilder(const hgx_locus &l) : L(l) {}

    static std::string spell(const hgx_locus &L, const HtVec &h) {
        std::string s = std::to_string(h[0]);
        for (size_t i = 1; i + 1 < h.size(); ++i) { s += '-'; s += L.name[h[i]]; }
        s += '-';
        s += std::to_string(h.back());
        return s;
    }

    void add(int dir, const HtVec &a, const HtVec &b) {
        const std::string ka = spell(L, a);
        auto it = index[dir].find(ka);
        size_t slot;
        if (it == index[dir].end()) {
            slot = table[dir].size();
            table[dir].push_back({a, {}});
            index[dir].emplace(ka, slot);
        } else slot = it->second;
        auto &alts = table[dir][slot].second;
        if (std::find(alts.begin(), alts.end(), b) == alts.end()) alts.push_back(b);
    }

    // candidate one-base extensions of a haplotype (nextbases, common:1447-1527)
    void next(const HtVec &ht, bool left, int exclude, std::vector<std::pair<HtVec, char>> &out) const {
        const int n = (int)L.backbone.size();
        const int pos = left ? ht[0] - 1 : ht.back() + 1;
        if (pos < 0 || pos >= n) return;
        if (left) {
            HtVec h(ht);
            h[0] = pos;
            out.push_back({h, L.backbone[pos]});
            const int prev = ht.size() > 2 ? ht[1] : -1;
            int hi = (int)(std::lower_bound(rev_key.begin(), rev_key.end(), pos + 1) - rev_key.begin());
            for (int j = hi - 1; j >= 0; --j) {
                const int v = rev[j].second;
                int p = L.pos[v];
                if (L.type[v] == HGX_VAR_DELETION) {
                    if (p == 0) continue;
                    p = p + L.len[v] - 1;
                }
                if (p > pos) continue;
                if (p < pos) break;
                if (v == exclude) continue;
                if (prev >= 0 && !second.count(((uint64_t)v << 32) | (uint32_t)prev)) continue;
                if (L.type[v] == HGX_VAR_SINGLE) {
                    HtVec h2;
                    h2.push_back(p);
                    h2.push_back(v);
                    h2.insert(h2.end(), ht.begin() + 1, ht.end());
                    out.push_back({h2, L.base[v]});
                } else if (L.type[v] == HGX_VAR_DELETION) {
                    HtVec h2;
                    h2.push_back(p - L.len[v] + 1);
                    h2.push_back(v);
                    h2.insert(h2.end(), ht.begin() + 1, ht.end());
                    next(h2, left, exclude, out);
                }
            }
        } else {
            HtVec h(ht);
            h.back() = pos;
            out.push_back({h, L.backbone[pos]});
            const int prev = ht.size() > 2 ? ht[ht.size() - 2] : -1;
            for (int j = lower_bound_pos(L.pos, pos); j < L.V; ++j) {
                const int p = L.pos[j];
                if (p < pos) continue;
                if (p > pos) break;
                if (j == exclude) continue;
                if (prev >= 0 && !second.count(((uint64_t)prev << 32) | (uint32_t)j)) continue;
                if (L.type[j] == HGX_VAR_SINGLE) {
                    HtVec h2(ht.begin(), ht.end() - 1);
                    h2.push_back(j);
                    h2.push_back(p);
                    out.push_back({h2, L.base[j]});
                } else if (L.type[j] == HGX_VAR_DELETION) {
                    HtVec h2(ht.begin(), ht.end() - 1);
                    h2.push_back(j);
                    h2.push_back(p + L.len[j] - 1);
                    next(h2, left, exclude, out);
                }
            }
        }
    }

    void recur(int orig, const HtVec &ht, const HtVec &alt, bool left, int dep) {
        std::vector<std::pair<HtVec, char>> b1, b2;
        next(ht, left, -1, b1);
        next(alt, left, orig, b2);
        bool found = false;
        for (auto &x : b1)
            for (auto &y : b2) {
                if (x.second != y.second) continue;
                if (left ? x.first[0] == y.first[0] : x.first.back() == y.first.back()) continue;
                found = true;
                recur(orig, x.first, y.first, left, dep + 1);
            }
        if (dep > 0 && !found) {
            add(left ? 0 : 1, ht, alt);
            add(left ? 0 : 1, alt, ht);
        }
    }

    void build() {
        for (int a = 0; a < L.A; ++a)
            for (int k = L.av_off[a]; k + 1 < L.av_off[a + 1]; ++k)
                second.insert(((uint64_t)L.av_var[k] << 32) | (uint32_t)L.av_var[k + 1]);
        for (int v = 0; v < L.V; ++v) {
            int p = L.pos[v];
            if (L.type[v] == HGX_VAR_DELETION) p = p + L.len[v] - 1;
            else if (L.type[v] == HGX_VAR_INSERTION) p += 1;
            rev.push_back({p, v});
        }
        std::stable_sort(rev.begin(), rev.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
        for (auto &r : rev) rev_key.push_back(r.first);
        const int n = (int)L.backbone.size();
        for (int v = 0; v < L.V; ++v) {
            const int p = L.pos[v];
            if (p == 0 || L.type[v] != HGX_VAR_DELETION) continue;
            const int dl = L.len[v];
            if (p + dl >= n) continue;
            recur(v, HtVec{p, v, p + dl - 1}, HtVec{p + dl, p + dl - 1}, true, 0);
            recur(v, HtVec{p, v, p + dl - 1}, HtVec{p, p - 1}, false, 0);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// per-locus streaming state
// ------------------------------------------------------------------------------------------------
struct Ht {
    int left, right;
    std::vector<int> ids;
    bool operator<(const Ht &o) const {
        if (left != o.left) return left < o.left;
        if (right != o.right) return right < o.right;
        return ids < o.ids;
    }
    bool operator==(const Ht &o) const { return left == o.left && right == o.right && ids == o.ids; }
};

// The alternatives of a locus with their spellings, sorted by anchor (built from hgx_locus::alts_left / alts_right once per
// parse; every chunk's Parser reads them).
struct AltTables {
    struct AltRec {
        int anchor;
        std::string key;
        HtVec key_ht;
        std::vector<HtVec> alts;
    };
    std::vector<AltRec> alt_l, alt_r;
    std::vector<int> alt_l_pos, alt_r_pos;
    explicit AltTables(const hgx_locus &L) {
        auto fill = [&](const std::vector<AltEntry> &src, std::vector<AltRec> &dst, bool left) {
            for (auto &e : src) {
                AltRec r;
                r.key_ht.push_back(e.key.left);
                r.key_ht.insert(r.key_ht.end(), e.key.vars.begin(), e.key.vars.end());
                r.key_ht.push_back(e.key.right);
                r.anchor = left ? e.key.right : e.key.left;
                r.key = AltBuilder::spell(L, r.key_ht);
                for (auto &a : e.alts) {
                    HtVec h;
                    h.push_back(a.left);
                    h.insert(h.end(), a.vars.begin(), a.vars.end());
                    h.push_back(a.right);
                    r.alts.push_back(h);
                }
                dst.push_back(std::move(r));
            }
        };
        fill(L.alts_left, alt_l, true);
        fill(L.alts_right, alt_r, false);
        for (auto &r : alt_l) alt_l_pos.push_back(r.anchor);
        for (auto &r : alt_r) alt_r_pos.push_back(r.anchor);
    }
};

struct Parser {
    const hgx_locus &L;
    hgx_parse_opts o;
    hgx_batch &B;
    NovelTable &NT;                                   // novel variants of this parse (shared by all workers)
    // alternatives with spellings, sorted by anchor: built once per parse (AltTables) and shared, read-only, by all chunks
    typedef AltTables::AltRec AltRec;
    const std::vector<AltRec> &alt_l, &alt_r;
    const std::vector<int> &alt_l_pos, &alt_r_pos;

    const hgx_batch &PILE;   // pileup tables (nt_set, counts) shared by all chunks
    struct ZsItem { int gap; char type; int id; };
    std::vector<ZsItem> zs_buf;                       // per-record scratch, reused
    std::vector<std::pair<char, int>> ops_buf;
    std::vector<Cmp> ec_buf;
    std::vector<Ht> ex_buf, union_buf;
    std::vector<int32_t> ids_buf;
    std::vector<uint32_t> eref_buf, gref_buf;
    Parser(const hgx_locus &l, const hgx_parse_opts &opts, hgx_batch &b, const hgx_batch &pile, const AltTables &at, NovelTable &nt)
        : L(l), o(opts), B(b), NT(nt), alt_l(at.alt_l), alt_r(at.alt_r), alt_l_pos(at.alt_l_pos), alt_r_pos(at.alt_r_pos), PILE(pile) {}

    // ---- variant accessors over known + novel ids --------------------------------------------
    int vtype(int id) const { return id < L.V ? L.type[id] : NT.get(id - L.V).type; }
    int vpos(int id) const { return id < L.V ? L.pos[id] : NT.get(id - L.V).pos; }
    int vlen(int id) const { return id < L.V ? L.len[id] : NT.get(id - L.V).len; }
    bool is_hv(int id) const { return id >= 0 && id < L.V; }
    std::string vname(int id) const {
        if (id == -1) return "unknown";
        if (id < L.V) return L.name[id];
        return "nv" + std::to_string(id - L.V);
    }
    int vright(int id) const { return vtype(id) == HGX_VAR_DELETION ? vpos(id) + vlen(id) - 1 : vpos(id); }

    static uint64_t nkey(int type, int pos, int k) { return NovelTable::key(type, pos, k); }

    // first variant at `pos` (known list order, then novel) of the wanted type and size/base (core:949-961, 1005-1017, 1045-1057)
    int lookup(int pos, int type, int key) const {
        for (int j = lower_bound_pos(L.pos, pos); j < L.V && L.pos[j] == pos; ++j) {
            if (L.type[j] != type) continue;
            if (type == HGX_VAR_SINGLE ? L.base[j] == (char)key : L.len[j] == key) return j;
        }
        const int k = NT.find(nkey(type, pos, key));
        return k < 0 ? -1 : L.V + k;
    }
    // core:404-431.  The reference asserts that the id is new; with several workers decoding at the same time another one may
    // have created the same variant since this record's lookup, so this is find-or-create (same id either way).
    int add_novel(int type, int pos, int key, const std::string &ins) {
        Novel nv;
        nv.type = type; nv.pos = pos; nv.base = 0;
        nv.len = 1;
        if (type == HGX_VAR_SINGLE) nv.base = (char)key;
        else nv.len = key;
        nv.ins = ins;
        return L.V + NT.get_or_add(nkey(type, pos, key), std::move(nv));
    }

    // ---- error_correct (core:119-243) over the cmp entries of one M op ------------------------------
    static int nt_bit(char c) {
        static const struct Lut { uint8_t t[256]; Lut() { memset(t, 0, 256); t['A'] = 1; t['C'] = 2; t['G'] = 4; t['T'] = 8; } } lut;
        return lut.t[(unsigned char)c];
    }
    static char single_nt(int mask) { return mask == 1 ? 'A' : mask == 2 ? 'C' : mask == 4 ? 'G' : 'T'; }

    int error_correct(std::string &read, int read_pos, std::vector<Cmp> &cl, size_t start) {
        const std::string &ref = L.backbone;
        const int n_ref = (int)ref.size();
        int ncorr = 0;
        std::vector<Cmp> &out = ec_buf;
        out.clear();
        bool stopped = false;
        for (size_t i = start; i < cl.size(); ++i) {
            Cmp c = cl[i];
            if (stopped || c.pos >= n_ref) {   // `break` of core:138-139 keeps the remaining entries untouched
                stopped = true;
                out.push_back(c);
                continue;
            }
            if (c.type == T_MATCH) {
                int last = 0;
                for (int j = 0; j < c.len; ++j) {
                    if (read_pos + j >= (int)read.size() || c.pos + j >= n_ref) continue;
                    char b = read[read_pos + j];
                    const int s = PILE.nt_set[c.pos + j];
                    if (s != 0 && !(s & nt_bit(b))) {
                        b = (s & (s - 1)) ? 'N' : single_nt(s);
                        read[read_pos + j] = b;
                        if (b == ref[c.pos + j]) throw RefError("assert read_bp != ref_bp");
                        Cmp m{T_MISMATCH, c.pos + j, 1, -1};
                        ncorr++;
                        if (b != 'N') m.id = lookup(c.pos + j, HGX_VAR_SINGLE, b);
                        if (j > last) out.push_back(Cmp{T_MATCH, c.pos + last, j - last, -2});
                        out.push_back(m);
                        last = j + 1;
                    }
                }
                if (last < c.len) out.push_back(Cmp{T_MATCH, c.pos + last, c.len - last, -2});
            } else {
                char b = read[read_pos];
                const int s = PILE.nt_set[c.pos];
                if (s != 0 && !(s & nt_bit(b))) {
                    b = (s & (s - 1)) ? 'N' : single_nt(s);
                    read[read_pos] = b;
                    if (b == 'N') c.id = -1;
                    else if (b == ref[c.pos]) { c = Cmp{T_MATCH, c.pos, 1, -2}; ncorr++; }
                    else c.id = lookup(c.pos, HGX_VAR_SINGLE, b);
                }
                out.push_back(c);
            }
            read_pos += cl[i].len;
        }
        cl.resize(start);
        for (auto &c : out) {                                     // merge adjacent matches (core:225-235)
            if (c.type == T_MATCH && cl.size() > start && cl.back().type == T_MATCH) cl.back().len += c.len;
            else cl.push_back(c);
        }
        return ncorr;
    }

    // ---- one record -> cmp_list (core:876-1164).  Returns false if the record is dropped. -----------
    bool decode(int pos, const char *cigar, std::string &read, const char *zs_str, const char *md, std::vector<Cmp> &cl) {
        std::vector<ZsItem> &zs = zs_buf;
        zs.clear();
        if (zs_str && *zs_str) {
            const char *p = zs_str;
            while (*p) {
                ZsItem z;
                z.gap = (int)strtol(p, (char **)&p, 10);
                if (*p != '|') throw RefError("malformed Zs");
                z.type = p[1];
                if (p[2] != '|') throw RefError("malformed Zs");
                p += 3;
                const char *q = p;
                while (*q && *q != ',') ++q;
                z.id = -1;
                if (q - p > 2 && p[0] == 'h' && p[1] == 'v' && !(q - p > 3 && p[2] == '0')) {   // "hv<n>": direct table
                    long num = 0;
                    bool digits = true;
                    for (const char *c = p + 2; c < q; ++c) {
                        if (*c < '0' || *c > '9') { digits = false; break; }
                        num = num * 10 + (*c - '0');
                        if (num > 100000000) { digits = false; break; }
                    }
                    if (digits && (size_t)num < L.hv_index.size()) z.id = L.hv_index[num];
                }
                if (z.id < 0) {
                    auto it = L.name_to_var.find(std::string(p, q));
                    if (it == L.name_to_var.end()) throw RefError("KeyError: Zs variant id not in the locus");
                    z.id = it->second;
                }
                zs.push_back(z);
                p = *q ? q + 1 : q;
            }
        }
        if (!md || !*md) throw RefError("assert MD != ''");
        const int md_n = (int)strlen(md);
        int md_i = 0, md_len = 0;
        size_t zs_i = 0;
        int zs_pos = zs.empty() ? 0 : zs[0].gap;
        int rp = 0, gp = pos;
        int n_ec = 0;
        bool bad = false;
        int clip0 = 0, clip1 = 0;
        cl.clear();
        std::vector<std::pair<char, int>> &ops = ops_buf;
        ops.clear();
        for (const char *p = cigar; *p;) {
            char *e;
            long n = strtol(p, &e, 10);
            if (e == p || !*e) throw RefError("malformed CIGAR");
            ops.push_back({*e, (int)n});
            p = e + 1;
        }
        auto zs_advance = [&](bool consume_base) {
            zs_i++;
            if (consume_base) zs_pos += 1;
            if (zs_i < zs.size()) zs_pos += zs[zs_i].gap;
        };
        for (size_t ci = 0; ci < ops.size(); ++ci) {
            const char op = ops[ci].first;
            const int n = ops[ci].second;
            if (op == 'M') {
                bool first = true;
                int used = 0;
                const size_t start = cl.size();
                for (;;) {
                    if (!first || md_len == 0) {
                        if (md_i >= md_n) throw RefError("IndexError: MD exhausted");
                        if (md[md_i] >= '0' && md[md_i] <= '9') {
                            int num = 0;
                            while (md_i < md_n && md[md_i] >= '0' && md[md_i] <= '9') num = num * 10 + (md[md_i++] - '0');
                            md_len += num;
                        }
                    }
                    if (md_len >= n) {
                        md_len -= n;
                        if (n > used) cl.push_back(Cmp{T_MATCH, gp + used, n - used, -2});
                        break;
                    }
                    first = false;
                    if (rp + md_len >= (int)read.size()) throw RefError("IndexError: read shorter than CIGAR");
                    const char base = read[rp + md_len];
                    if (md_i >= md_n || !strchr("ACGT", md[md_i])) throw RefError("assert MD_ref_base in ACGT");
                    md_i++;
                    if (md_len > used) cl.push_back(Cmp{T_MATCH, gp + used, md_len - used, -2});
                    int id;
                    if (rp + md_len == zs_pos && zs_i < zs.size()) {
                        if (zs[zs_i].type != 'S') throw RefError("assert Zs type S");
                        id = zs[zs_i].id;
                        zs_advance(true);
                    } else id = lookup(gp + md_len, HGX_VAR_SINGLE, base);
                    cl.push_back(Cmp{T_MISMATCH, gp + md_len, 1, id});
                    used = md_len + 1;
                    md_len += 1;
                    if (md_len == n) { md_len = 0; break; }
                }
                if (o.error_correction) n_ec += error_correct(read, rp, cl, start);
            } else if (op == 'I') {
                int id;
                if (rp == zs_pos && zs_i < zs.size()) {
                    if (zs[zs_i].type != 'I') throw RefError("assert Zs type I");
                    id = zs[zs_i].id;
                    zs_advance(false);
                } else id = lookup(gp, HGX_VAR_INSERTION, n);
                cl.push_back(Cmp{T_INSERTION, gp, n, id});
                for (int k = rp; k < rp + n && k < (int)read.size(); ++k)
                    if (read[k] == 'N') bad = true;
            } else if (op == 'D') {
                if (md_i < md_n && md[md_i] == '0') md_i++;
                if (md_i >= md_n || md[md_i] != '^') throw RefError("assert MD ^");
                md_i++;
                while (md_i < md_n && strchr("ACGT", md[md_i])) md_i++;
                int id;
                if (rp == zs_pos && zs_i < zs.size() && zs[zs_i].type == 'D') {
                    id = zs[zs_i].id;
                    zs_advance(false);
                } else id = lookup(gp, HGX_VAR_DELETION, n);
                cl.push_back(Cmp{T_DELETION, gp, n, id});
                if (gp < (int)L.backbone.size()) {                  // artificial-deletion check (core:1064-1077)
                    const uint32_t *c = &PILE.counts[(size_t)gp * 6];
                    const uint64_t dc = c[5], nc = (uint64_t)c[0] + c[1] + c[2] + c[3] + c[4];
                    if (L.base_kind == HGX_BASE_HLA && dc * 6 < nc) bad = true;
                }
            } else if (op == 'S') {
                if (ci == 0) { clip0 = n; zs_pos += n; }
                else {
                    if (ci + 1 != ops.size()) throw RefError("assert soft clip at the end");
                    clip1 = n;
                }
            } else throw RefError("assert: unsupported CIGAR op");
            if (op == 'M' || op == 'N' || op == 'D') gp += n;
            if (op == 'M' || op == 'I' || op == 'S') rp += n;
        }
        if (clip0 > 0) read.erase(0, clip0);
        if (clip1 > 0) read.erase(read.size() - std::min<size_t>(clip1, read.size()));
        if (gp > (int)L.backbone.size()) return false;
        if (n_ec > std::max(1, o.num_editdist)) return false;
        if (bad) return false;
        rp = 0;                                                     // novel variants (core:1126-1164)
        for (auto &c : cl) {
            if (c.type != T_MATCH && c.id == -1) {
                if (c.type == T_MISMATCH) {
                    const char b = read[rp];
                    if (b != 'N') c.id = add_novel(HGX_VAR_SINGLE, c.pos, b, "");
                } else if (c.type == T_DELETION) c.id = add_novel(HGX_VAR_DELETION, c.pos, c.len, "");
                else c.id = add_novel(HGX_VAR_INSERTION, c.pos, c.len, read.substr(rp, c.len));
            }
            if (c.type != T_DELETION) rp += c.len;
        }
        return true;
    }

    // ---- identify_ambigious_diffs (common:1663-1955) ------------------------------------------------
    struct AltSide {
        int coord;                // the left (or right) coordinate of the spelling
        std::vector<int> ids;     // variant ids between the coordinate and the mid part
        bool operator<(const AltSide &x) const { return coord != x.coord ? coord < x.coord : ids < x.ids; }
        bool operator==(const AltSide &x) const { return coord == x.coord && ids == x.ids; }
    };

    std::string join_ids(const std::vector<int> &ids) const {
        std::string s;
        for (size_t i = 0; i < ids.size(); ++i) { if (i) s += '-'; s += vname(ids[i]); }
        return s;
    }

    void ambiguous(const std::vector<Cmp> &c2, int &cmp_left, int &cmp_right, std::vector<AltSide> &lset, std::vector<AltSide> &rset) {
        const int n = (int)c2.size();
        const int n_ref = (int)L.backbone.size();
        cmp_left = 0;
        cmp_right = n - 1;
        const int left = c2[0].pos, right = c2[n - 1].pos + c2[n - 1].len - 1;
        lset.clear();
        rset.clear();
        auto add_unique = [](std::vector<AltSide> &s, AltSide a) { if (std::find(s.begin(), s.end(), a) == s.end()) s.push_back(std::move(a)); };
        auto seq_len_of = [&](int b, int e) {   // match + mismatch bases of c2[b..e)
            int t = 0;
            for (int k = b; k < e; ++k) {
                if (c2[k].type == T_MATCH) t += std::max(0, std::min(c2[k].pos + c2[k].len, n_ref) - c2[k].pos);
                else if (c2[k].type == T_MISMATCH) t += 1;
            }
            return t;
        };
        auto ht_of = [&](int b, int e, std::vector<int> &ids) {
            ids.clear();
            for (int k = b; k < e; ++k) if (c2[k].type != T_MATCH && c2[k].id != -1) ids.push_back(c2[k].id);
        };
        auto skip = [&](const Cmp &c) {
            if (c.type == T_MATCH) return false;
            if (c.type == T_INSERTION) return true;              // var_id = "" never starts with "hv" (common:1708-1713)
            return !is_hv(c.id);
        };
        std::vector<int> cur;
        // left direction
        bool found = false;
        if (!alt_l.empty())
        for (int i = n - 1; i >= 0; --i) {
            const Cmp &ci = c2[i];
            if (skip(ci)) continue;
            const int cur_left = ci.pos;
            const int cur_right = (ci.type == T_MATCH || ci.type == T_DELETION) ? ci.pos + ci.len - 1 : ci.pos;
            int hi = (int)(std::lower_bound(alt_l_pos.begin(), alt_l_pos.end(), cur_right + 1) - alt_l_pos.begin());
            int j = std::min(hi + 1, (int)alt_l.size()) - 1;
            if (j < 0 || alt_l_pos[j] < cur_left) {
                // no table entry anchored inside this entry: nothing can match (cheap exit)
                bool any = false;
                for (int jj = j; jj >= 0 && alt_l_pos[jj] >= cur_left; --jj) any = true;
                if (!any) continue;
            }
            ht_of(0, i + 1, cur);
            const int seqlen = seq_len_of(0, i + 1);
            const std::string cur_join = join_ids(cur);
            bool i_found = false;
            for (; j >= 0; --j) {
                const AltRec &r = alt_l[j];
                if (r.anchor < cur_left) break;
                if (r.anchor > cur_right) continue;
                if (!cur.empty() && r.key.find(cur_join) == std::string::npos) continue;
                const int flen = (int)r.key_ht.size() - 1;          // fields of key.split('-')[:-1]
                if ((int)cur.size() + 1 == flen) {
                    if (left < r.key_ht[0]) continue;
                } else {
                    int k = flen - (int)cur.size() - 1;
                    if (k < 0) k += flen;                            // Python negative index
                    if (k <= 0 || k >= flen) throw RefError("KeyError/IndexError in identify_ambigious_diffs");
                    if (left <= L.right[r.key_ht[k]]) continue;
                }
                i_found = true;
                for (const HtVec &alt : r.alts) {
                    const int a_right = alt.back();
                    if (a_right > cur_right) throw RefError("assert alt_ht_right <= cur_right");
                    int seq_pos = cur_right - a_right, cur_pos = a_right;
                    std::vector<int> part;
                    for (int k = (int)alt.size() - 2; k >= 1; --k) {
                        const int v = alt[k];
                        int vp = L.pos[v];
                        if (L.type[v] == HGX_VAR_DELETION) vp = vp + L.len[v] - 1;
                        if (vp > cur_pos) throw RefError("assert var_pos_ <= cur_pos");
                        int nsp = seq_pos + (cur_pos - vp);
                        if (nsp >= seqlen) break;
                        int ncp;
                        if (L.type[v] == HGX_VAR_SINGLE) { nsp += 1; ncp = vp - 1; }
                        else if (L.type[v] == HGX_VAR_DELETION) ncp = vp - L.len[v];
                        else throw RefError("assert: insertion in alternative");
                        part.insert(part.begin(), v);
                        if (nsp >= seqlen) break;
                        seq_pos = nsp;
                        cur_pos = ncp;
                    }
                    if (!part.empty()) {
                        const int seq_left = seqlen - seq_pos - 1;
                        AltSide s;
                        s.coord = cur_pos - seq_left;
                        s.ids = part;
                        if (found)
                            for (int jj = i + 1; jj < cmp_left; ++jj)
                                if (c2[jj].type != T_MATCH && is_hv(c2[jj].id)) s.ids.push_back(c2[jj].id);
                        add_unique(lset, s);
                    }
                }
            }
            if (i_found) {
                if (!found) {
                    cmp_left = i + 1;
                    add_unique(lset, AltSide{left, cur});
                }
                found = true;
            }
        }
        if (!found) add_unique(lset, AltSide{left, {}});
        // right direction
        found = false;
        if (!alt_r.empty())
        for (int i = 0; i < n; ++i) {
            const Cmp &ci = c2[i];
            if (skip(ci)) continue;
            const int cur_left = ci.pos;
            const int cur_right = (ci.type == T_MATCH || ci.type == T_DELETION) ? ci.pos + ci.len - 1 : ci.pos;
            int j = (int)(std::lower_bound(alt_r_pos.begin(), alt_r_pos.end(), cur_left) - alt_r_pos.begin());
            if (j >= (int)alt_r.size() || alt_r_pos[j] > cur_right) continue;
            ht_of(i, n, cur);
            const int seqlen = seq_len_of(i, n);
            const std::string cur_join = join_ids(cur);
            bool i_found = false;
            for (; j < (int)alt_r.size(); ++j) {
                const AltRec &r = alt_r[j];
                if (r.anchor > cur_right) break;
                if (r.anchor < cur_left) continue;
                if (!cur.empty() && r.key.find(cur_join) == std::string::npos) continue;
                const int flen = (int)r.key_ht.size() - 1;          // fields of key.split('-')[1:]
                const int32_t *f = r.key_ht.data() + 1;
                if ((int)cur.size() + 1 == flen) {
                    if (right > f[flen - 1]) continue;
                } else {
                    const int k = (int)cur.size();
                    if (k >= flen) throw RefError("IndexError in identify_ambigious_diffs");
                    if (k == flen - 1) throw RefError("KeyError in identify_ambigious_diffs");
                    if (right >= L.pos[f[k]]) continue;
                }
                i_found = true;
                for (const HtVec &alt : r.alts) {
                    const int a_left = alt[0];
                    if (cur_left > a_left) throw RefError("assert cur_left <= alt_ht_left");
                    int seq_pos = a_left - cur_left, cur_pos = a_left;
                    std::vector<int> part;
                    for (size_t k = 1; k + 1 < alt.size(); ++k) {
                        const int v = alt[k];
                        const int vp = L.pos[v];
                        if (vp < cur_pos) throw RefError("assert var_pos_ >= cur_pos");
                        int nsp = seq_pos + (vp - cur_pos);
                        if (nsp >= seqlen) break;
                        int ncp;
                        if (L.type[v] == HGX_VAR_SINGLE) { nsp += 1; ncp = vp + 1; }
                        else if (L.type[v] == HGX_VAR_DELETION) ncp = vp + L.len[v];
                        else throw RefError("assert: insertion in alternative");
                        part.push_back(v);
                        if (nsp >= seqlen) break;
                        seq_pos = nsp;
                        cur_pos = ncp;
                    }
                    if (!part.empty()) {
                        const int seq_left = seqlen - seq_pos - 1;
                        if (seq_left < 0) throw RefError("assert seq_left >= 0");
                        AltSide s;
                        s.coord = cur_pos + seq_left;
                        if (found)
                            for (int jj = cmp_right + 1; jj < i; ++jj)
                                if (c2[jj].type != T_MATCH && is_hv(c2[jj].id)) s.ids.push_back(c2[jj].id);
                        s.ids.insert(s.ids.end(), part.begin(), part.end());
                        add_unique(rset, s);
                    }
                }
            }
            if (i_found) {
                if (!found) {
                    cmp_right = i - 1;
                    add_unique(rset, AltSide{right, cur});
                }
                found = true;
            }
        }
        if (!found) add_unique(rset, AltSide{right, {}});
        if (cmp_right < cmp_left) {
            cmp_left = 0;
            lset.clear();
            lset.push_back(AltSide{left, {}});
        }
        // check_amb_uniqueness (validation_check.py:313-341): always on (quirk Q1)
        std::vector<std::vector<int>> seen;
        for (auto &s : lset) {
            if (s.ids.empty()) continue;
            if (std::find(seen.begin(), seen.end(), s.ids) != seen.end()) throw RefError("check_amb_uniqueness failed (reference exits)");
            seen.push_back(s.ids);
        }
        for (auto &s : rset) {
            if (s.ids.empty()) continue;
            if (std::find(seen.begin(), seen.end(), s.ids) != seen.end()) throw RefError("check_amb_uniqueness failed (reference exits)");
            seen.push_back(s.ids);
        }
    }

    // ---- get_exon_haplotypes (core:718-792) -------------------------------------------------------------
    void exon_pieces(const Ht &ht, std::vector<Ht> &out) const {
        for (auto &e : L.exons) {
            const int el = e[0], er = e[1];
            int hl = ht.left, hr = ht.right;
            if (el > hr || er < hl) continue;
            std::vector<int> ids(ht.ids);
            if (hl < el) {
                bool done = false;
                for (size_t i = 0; i < ids.size(); ++i) {
                    const int t = vtype(ids[i]), p = vpos(ids[i]);
                    if ((t != HGX_VAR_DELETION && p >= el) || (t == HGX_VAR_DELETION && p - 1 >= el)) {
                        hl = el;
                        ids.erase(ids.begin(), ids.begin() + i);
                        done = true;
                        break;
                    }
                    if (t == HGX_VAR_DELETION) {
                        const int r = p + vlen(ids[i]);
                        if (r >= el) {
                            hl = r;
                            ids.erase(ids.begin(), ids.begin() + i + 1);
                            done = true;
                            break;
                        }
                    }
                }
                if (!done) { hl = el; ids.clear(); }
            }
            if (hl < el) throw RefError("assert ht_left >= e_left");
            if (hr > er) {
                bool done = false;
                for (int i = (int)ids.size() - 1; i >= 0; --i) {
                    const int t = vtype(ids[i]);
                    int r = vpos(ids[i]);
                    if (t == HGX_VAR_DELETION) r = r + vlen(ids[i]) - 1;
                    if ((t != HGX_VAR_DELETION && r <= er) || (t == HGX_VAR_DELETION && r + 1 <= er)) {
                        hr = er;
                        ids.resize(i + 1);
                        done = true;
                        break;
                    }
                    if (t == HGX_VAR_DELETION) {
                        const int l = r - vlen(ids[i]);
                        if (l <= er) {
                            hr = l;
                            ids.resize(i);
                            done = true;
                            break;
                        }
                    }
                }
                if (!done) { hr = er; ids.clear(); }
            }
            if (hl > hr) throw RefError("assert ht_left <= ht_right");
            out.push_back(Ht{hl, hr, ids});
        }
    }

};

struct Fields {
    const char *qname; size_t qname_len;
    int flag, pos;
    const char *cigar;
    const char *seq; size_t seq_len;
    const char *zs, *md;
    bool has_nm, has_nh, yt_cp;
    uint8_t bad_tag;                 // an NM / NH tag whose value int() would reject (core:838-841): raises once the record gets that far
    const char *bad_tok;             // ... the offending text
    uint8_t kept;                    // outcome of the record filters (core:815-872), see filter_records
    long nm, nh;
    // the DECODE KEY of a record = (pos, cigar, seq, zs, md): everything its cmp_list / haplotypes depend on
    uint32_t cigar_len, zs_len, md_len;
    uint32_t rep;                    // index of the first record with an equal decode key
    uint32_t n_pile;                 // at rep: members of the group that count into the pileup (common:1076-1090)
    uint32_t slot;                   // at rep: index of the group's decode result, or NO_SLOT
    uint64_t key;                    // hash of the decode key
};
enum { KEPT_NO = 0, KEPT_YES = 1, KEPT_ERR_TAGS = 2, KEPT_ERR_FLAG = 3, KEPT_ERR_TAGVAL = 4 };
constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;

inline uint64_t hash_bytes(const char *p, size_t n, uint64_t h) {
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 32;
        p += 8;
        n -= 8;
    }
    uint64_t w = 0;
    memcpy(&w, p, n);
    h = (h ^ w ^ ((uint64_t)n << 56)) * 0xD6E8FEB86659FD93ull;
    return h ^ (h >> 29);
}
inline void set_decode_key(Fields &f) {          // (the lengths of cigar / zs / md were set by split_line)
    uint64_t h = hash_bytes(f.seq, f.seq_len, 0x243F6A8885A308D3ull ^ (uint32_t)f.pos);
    h = hash_bytes(f.cigar, f.cigar_len, h);
    h = hash_bytes(f.zs ? f.zs : "", f.zs_len, h ^ (f.zs ? 1 : 0));
    f.key = hash_bytes(f.md ? f.md : "", f.md_len, h ^ (f.md ? 2 : 0));
}
inline bool same_decode_key(const Fields &a, const Fields &b) {
    return a.key == b.key && a.pos == b.pos && a.seq_len == b.seq_len && a.cigar_len == b.cigar_len && a.zs_len == b.zs_len &&
           a.md_len == b.md_len && (a.zs == nullptr) == (b.zs == nullptr) && (a.md == nullptr) == (b.md == nullptr) &&
           memcmp(a.seq, b.seq, a.seq_len) == 0 && memcmp(a.cigar, b.cigar, a.cigar_len) == 0 &&
           (a.zs_len == 0 || memcmp(a.zs, b.zs, a.zs_len) == 0) && (a.md_len == 0 || memcmp(a.md, b.md, a.md_len) == 0);
}

// split one line on whitespace in place (the buffer is a private copy); returns false for header/empty lines
inline long parse_long(const char *p) {            // strtol(p, nullptr, 10) for the plain decimal fields of a SAM record
    bool neg = false;
    if (*p == '-') { neg = true; ++p; }
    else if (*p == '+') ++p;
    long v = 0;
    while (*p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
    return neg ? -v : v;
}

// would Python's int(text) accept it?  optional surrounding blanks cannot occur inside a token; sign, digits, single underscores
// between digits (typing_core.py:811-812, 838-841 call int() on FLAG, POS and the NM / NH values)
inline bool py_int_ok(const char *t, size_t n) {
    size_t i = 0;
    if (i < n && (t[i] == '+' || t[i] == '-')) ++i;
    if (i >= n || t[i] < '0' || t[i] > '9') return false;
    for (; i < n; ++i) {
        if (t[i] >= '0' && t[i] <= '9') continue;
        if (t[i] == '_' && i + 1 < n && t[i + 1] >= '0' && t[i + 1] <= '9' && t[i - 1] != '_') continue;
        return false;
    }
    return true;
}

inline void note_tag(Fields &f, char *tok, size_t len) {
    if (len < 5) {                                   // too short to carry a value: tok + 5 may not be dereferenced
        if (tok[0] == 'Z' && tok[1] == 's') { f.zs = tok + len; f.zs_len = 0; }
        else if (tok[0] == 'M' && tok[1] == 'D') { f.md = tok + len; f.md_len = 0; }
        else if (tok[0] == 'N' && (tok[1] == 'M' || tok[1] == 'H')) {      // int(col[5:]) of an empty string raises
            if (tok[1] == 'M') f.has_nm = true; else f.has_nh = true;
            if (!f.bad_tag) { f.bad_tag = 1; f.bad_tok = tok + len; }
        }
        else if (len >= 2 && tok[0] == 'Y' && tok[1] == 'T') f.yt_cp = false;     // YT = col[5:] = "" (typing_common.py:1229-1230)
        return;
    }
    if (tok[0] == 'Z' && tok[1] == 's') { f.zs = tok + 5; f.zs_len = (uint32_t)(len - 5); }
    else if (tok[0] == 'M' && tok[1] == 'D') { f.md = tok + 5; f.md_len = (uint32_t)(len - 5); }
    else if (tok[0] == 'N' && tok[1] == 'M') {
        f.has_nm = true; f.nm = strtol(tok + 5, nullptr, 10);
        if (!f.bad_tag && !py_int_ok(tok + 5, len - 5)) { f.bad_tag = 1; f.bad_tok = tok + 5; }
    } else if (tok[0] == 'N' && tok[1] == 'H') {
        f.has_nh = true; f.nh = strtol(tok + 5, nullptr, 10);
        if (!f.bad_tag && !py_int_ok(tok + 5, len - 5)) { f.bad_tag = 1; f.bad_tok = tok + 5; }
    }
    else if (tok[0] == 'Y' && tok[1] == 'T') f.yt_cp = len == 7 && tok[5] == 'C' && tok[6] == 'P';
}

// Split one record in place the way the reference's `line.strip().split()` does (on runs of tab / space / CR).  A record
// without spaces or CRs -- every record an aligner writes -- takes the tab-only path (memchr, 16+ bytes per step); anything
// else the byte loop.  Returns 0 for a record, else what the reference's loop dies of on this line (typing_core.py:803-814):
// 1 = fewer than six fields (the unpacking of cols[:6]: ValueError), 2 = fewer than eleven (cols[9] / cols[10]: IndexError),
// 3 / 4 = FLAG / POS that int() rejects (ValueError); *bad_n / *bad_tok say how many fields / which text.
static int split_line(char *line, char *end, Fields &f, int *bad_n, const char **bad_tok) {
    char *cols[11];
    size_t lens[11];
    int nc = 0;
    f.zs = f.md = nullptr;
    f.zs_len = f.md_len = 0;
    f.has_nm = f.has_nh = f.yt_cp = false;
    f.nm = f.nh = 0;
    f.bad_tag = 0;
    f.bad_tok = nullptr;
    const size_t n = (size_t)(end - line);
    if (!memchr(line, ' ', n) && !memchr(line, '\r', n)) {
        char *p = line;
        while (p < end) {
            char *q = (char *)memchr(p, '\t', (size_t)(end - p));
            if (!q) q = end;
            if (q > p) {                              // (empty tokens between adjacent tabs vanish, as with split())
                *q = 0;
                if (nc < 11) { cols[nc] = p; lens[nc++] = (size_t)(q - p); }
                else note_tag(f, p, (size_t)(q - p));
            }
            p = q + 1;
        }
    } else {
        char *p = line;
        while (p < end) {
            while (p < end && (*p == '\t' || *p == ' ' || *p == '\r')) ++p;
            if (p >= end) break;
            char *tok = p;
            while (p < end && *p != '\t' && *p != ' ' && *p != '\r') ++p;
            const size_t len = (size_t)(p - tok);
            if (p < end) *p++ = 0;
            if (nc < 11) { cols[nc] = tok; lens[nc++] = len; }
            else note_tag(f, tok, len);
        }
    }
    if (nc < 11) { *bad_n = nc; return nc < 6 ? 1 : 2; }
    if (!py_int_ok(cols[1], lens[1])) { *bad_tok = cols[1]; return 3; }
    if (!py_int_ok(cols[3], lens[3])) { *bad_tok = cols[3]; return 4; }
    f.qname = cols[0];
    f.qname_len = lens[0];
    f.flag = (int)strtol(cols[1], nullptr, 10);
    f.pos = (int)strtol(cols[3], nullptr, 10);
    f.cigar = cols[5];
    f.cigar_len = (uint32_t)lens[5];
    f.seq = cols[9];
    f.seq_len = lens[9];
    set_decode_key(f);
    return 0;
}

// ---- BAM records straight into Fields (no text round trip) --------------------------------------------------------------
// Character data the binary record does not hold as text (SEQ is 4-bit packed, CIGAR is binary) is spelled into a per-worker
// arena of fixed blocks (pointers stay valid); QNAME and the Z tags (Zs, MD, YT) are used in place.
struct CharArena {
    static constexpr size_t BLOCK = 4u << 20;
    std::vector<char *> blocks;
    size_t used = BLOCK;
    CharArena() = default;
    CharArena(const CharArena &) = delete;
    CharArena(CharArena &&o) noexcept : blocks(std::move(o.blocks)), used(o.used) { o.blocks.clear(); o.used = BLOCK; }
    ~CharArena() { for (char *b : blocks) hgx_host_free(b); }
    char *take(size_t n) {
        if (n > BLOCK) { blocks.insert(blocks.begin(), (char *)hgx_host_alloc(n)); return blocks.front(); }    // (never the current block)
        if (used + n > BLOCK) { blocks.push_back((char *)hgx_host_alloc(BLOCK)); used = 0; }
        char *p = blocks.back() + used;
        used += n;
        return p;
    }
};

inline uint32_t ld32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t ld16(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// The value of an integer-looking tag the way the text path reads it (strtol on the printed value): ints as they are, a float
// through its "%g" spelling, Z / A from their characters.
inline bool bam_tag_long(char t, const unsigned char *v, size_t avail, long &out) {
    switch (t) {
        case 'c': if (avail < 1) return false; out = (int8_t)v[0]; return true;
        case 'C': if (avail < 1) return false; out = v[0]; return true;
        case 's': if (avail < 2) return false; out = (int16_t)ld16(v); return true;
        case 'S': if (avail < 2) return false; out = (long)ld16(v); return true;
        case 'i': if (avail < 4) return false; out = (int32_t)ld32(v); return true;
        case 'I': if (avail < 4) return false; out = (long)ld32(v); return true;
        case 'f': { if (avail < 4) return false; float f; const uint32_t u = ld32(v); memcpy(&f, &u, 4); char b[40]; snprintf(b, sizeof b, "%g", f); out = strtol(b, nullptr, 10); return true; }
        case 'A': { if (avail < 1) return false; const char b[2] = {(char)v[0], 0}; out = strtol(b, nullptr, 10); return true; }
        case 'Z': case 'H': out = strtol((const char *)v, nullptr, 10); return true;
        default: return false;
    }
}

// one BAM record (r = start of the record after its block_size word, len = block_size) -> Fields; false = malformed
static bool split_bam(const unsigned char *r, size_t len, Fields &f, CharArena &arena) {
    static const char CIG[] = "MIDNSHP=X";
    if (len < 32) return false;
    const int32_t ref_id = (int32_t)ld32(r), pos0 = (int32_t)ld32(r + 4);
    const uint32_t l_rn = r[8], n_cig = ld16(r + 12), flag = ld16(r + 14);
    const int32_t l_seq = (int32_t)ld32(r + 16);
    if (l_rn == 0 || l_seq < 0) return false;
    size_t q = 32 + (size_t)l_rn;
    const size_t cig_at = q;
    q += 4ull * n_cig;
    const size_t seq_at = q;
    q += (size_t)(l_seq + 1) / 2 + (size_t)l_seq;
    if (q > len || r[32 + l_rn - 1] != 0) return false;
    f.zs = f.md = nullptr;
    f.zs_len = f.md_len = 0;
    f.has_nm = f.has_nh = f.yt_cp = false;
    f.bad_tag = 0;
    f.bad_tok = nullptr;
    f.nm = f.nh = 0;
    f.qname = (const char *)r + 32;
    f.qname_len = l_rn - 1;
    f.flag = (int)flag;
    f.pos = pos0 + 1;
    // tags (and the CG real-CIGAR rule: see hgx_bam.cpp find_real_cigar)
    const unsigned char *cg_items = nullptr;
    uint32_t cg_n = 0;
    while (q + 3 <= len) {
        const char t0 = (char)r[q], t1 = (char)r[q + 1], t = (char)r[q + 2];
        q += 3;
        size_t sz = 0;
        switch (t) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': {
                const void *e = memchr(r + q, 0, len - q);
                if (!e) return false;
                sz = (size_t)((const unsigned char *)e - (r + q)) + 1;
            } break;
            case 'B': {
                if (q + 5 > len) return false;
                const char st = (char)r[q];
                const uint32_t cnt = ld32(r + q + 1);
                const size_t w = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
                if (!w) return false;
                sz = 5 + w * (size_t)cnt;
                if (t0 == 'C' && t1 == 'G' && (st == 'I' || st == 'i') && cnt >= n_cig && cnt < (1u << 29) && q + sz <= len && !cg_items) {
                    cg_items = r + q + 5;
                    cg_n = cnt;
                }
            } break;
            default: return false;
        }
        if (q + sz > len) return false;
        const unsigned char *v = r + q;
        const bool text = t == 'Z' || t == 'H';
        if (t0 == 'Z' && t1 == 's') { if (text) { f.zs = (const char *)v; f.zs_len = (uint32_t)(sz - 1); } else { f.zs = ""; f.zs_len = 0; } }
        else if (t0 == 'M' && t1 == 'D') { if (text) { f.md = (const char *)v; f.md_len = (uint32_t)(sz - 1); } else { f.md = ""; f.md_len = 0; } }
        else if (t0 == 'N' && t1 == 'M') { f.has_nm = true; long x = 0; if (bam_tag_long(t, v, sz, x)) f.nm = x; else f.nm = 0; }
        else if (t0 == 'N' && t1 == 'H') { f.has_nh = true; long x = 0; if (bam_tag_long(t, v, sz, x)) f.nh = x; else f.nh = 0; }
        else if (t0 == 'Y' && t1 == 'T') f.yt_cp = text && sz == 3 && v[0] == 'C' && v[1] == 'P';
        q += sz;
    }
    // CIGAR text
    const unsigned char *cig = r + cig_at;
    uint32_t nc = n_cig;
    if (cg_items && n_cig > 0 && ref_id >= 0 && pos0 >= 0 && (ld32(cig) & 15) == 4 && (int64_t)(ld32(cig) >> 4) == (int64_t)l_seq) {
        cig = cg_items;
        nc = cg_n;
    }
    char *ct = arena.take(nc ? 11 * (size_t)nc + 1 : 2);
    char *w = ct;
    if (nc == 0) *w++ = '*';
    for (uint32_t k = 0; k < nc; ++k) {
        const uint32_t v = ld32(cig + 4 * k);
        uint32_t n = v >> 4;
        char d[12];
        int nd = 0;
        do { d[nd++] = (char)('0' + n % 10); n /= 10; } while (n);
        while (nd) *w++ = d[--nd];
        *w++ = (v & 15) < 9 ? CIG[v & 15] : '?';
    }
    *w = 0;
    f.cigar = ct;
    f.cigar_len = (uint32_t)(w - ct);
    // SEQ text
    if (l_seq == 0) {
        char *st = arena.take(2);
        st[0] = '*'; st[1] = 0;
        f.seq = st;
        f.seq_len = 1;
    } else {
        char *st = arena.take((size_t)l_seq + 2);
        const unsigned char *sp = r + seq_at;
        static const struct Pair { uint16_t t[256]; Pair() { for (int b = 0; b < 256; ++b) t[b] = (uint16_t)((unsigned char)"=ACMGRSVTWYHKDBN"[b >> 4] | ((unsigned char)"=ACMGRSVTWYHKDBN"[b & 15] << 8)); } } two;
        for (int32_t i = 0; i < l_seq; i += 2) {            // one table look-up per packed byte = two bases (little endian store)
            const uint16_t v = two.t[sp[i >> 1]];
            memcpy(st + i, &v, 2);
        }
        st[l_seq] = 0;
        f.seq = st;
        f.seq_len = (size_t)l_seq;
    }
    set_decode_key(f);
    return true;
}

}   // namespace

int hgx_interdist_median(const int64_t *hist, long long *expected) {
    int64_t total = 0;
    for (size_t b = 0; b < (size_t)HGX_INTERDIST_BINS; ++b) total += hist[b];
    *expected = -1;
    if (total == 0) return 0;
    const int64_t k = total / 2;
    int64_t seen = 0;
    for (size_t b = 0; b < (size_t)HGX_INTERDIST_BINS; ++b) {
        seen += hist[b];
        if (seen > k) {
            if (b == 0 || b + 1 == (size_t)HGX_INTERDIST_BINS) return 1;
            *expected = (long long)b - 1 - (long long)HGX_INTERDIST_HALF;
            return 0;
        }
    }
    return 0;
}

int hgx_build_alternatives(hgx_locus &L) {
    if (L.alts_built) return HGX_OK;
    AltBuilder ab(L);
    ab.build();
    for (int dir = 0; dir < 2; ++dir) {
        auto &dst = dir == 0 ? L.alts_left : L.alts_right;
        for (auto &kv : ab.table[dir]) {
            AltEntry e;
            e.key.left = kv.first[0];
            e.key.right = kv.first.back();
            e.key.vars.assign(kv.first.begin() + 1, kv.first.end() - 1);
            for (auto &a : kv.second) {
                AltHt h;
                h.left = a[0];
                h.right = a.back();
                h.vars.assign(a.begin() + 1, a.end() - 1);
                e.alts.push_back(h);
            }
            dst.push_back(e);
        }
        // Alts_left_list sorts by the right coordinate, Alts_right_list by the left one (core:585-596)
        std::stable_sort(dst.begin(), dst.end(), [dir](const AltEntry &a, const AltEntry &b) {
            return dir == 0 ? a.key.right < b.key.right : a.key.left < b.key.left;
        });
    }
    L.alts_built = true;
    return HGX_OK;
}

extern "C" int hgx_locus_alternatives_text(const hgx_locus *Lc, char *buf, size_t cap, size_t *needed) {
    HARGCHK(Lc && needed);
    hgx_locus &L = *const_cast<hgx_locus *>(Lc);
    try {
        hgx_build_alternatives(L);
    } catch (const std::exception &e) {
        hgx_set_error("%s", e.what());
        return HGX_EPARSE;
    }
    std::string s;
    auto spell = [&](const AltHt &h) {
        std::string t = std::to_string(h.left);
        for (int v : h.vars) { t += '-'; t += L.name[v]; }
        t += '-';
        t += std::to_string(h.right);
        return t;
    };
    for (int dir = 0; dir < 2; ++dir)
        for (auto &e : (dir == 0 ? L.alts_left : L.alts_right))
            for (auto &a : e.alts) {
                s += dir == 0 ? "L\t" : "R\t";
                s += spell(e.key);
                s += '\t';
                s += spell(a);
                s += '\n';
            }
    *needed = s.size();
    if (buf && cap > s.size()) memcpy(buf, s.c_str(), s.size() + 1);
    return HGX_OK;
}

namespace {

// read id of a record: QNAME, or QNAME up to the first '|' in simulation mode (core:808-809)
inline size_t read_id_len(const Fields &f, bool simulation) {
    if (simulation) {
        const char *bar = (const char *)memchr(f.qname, '|', f.qname_len);
        if (bar) return (size_t)(bar - f.qname);
    }
    return f.qname_len;
}

// ---- the streaming loop (core:800-1587), restated around the observation that deep coverage repeats records -------------
// A record's cmp_list, its ambiguity sets and its haplotypes are a function of its DECODE KEY (pos, cigar, seq, Zs, MD) -- and of
// tables that are fixed while the loop runs (locus, pileup).  At 1 M reads on a 3.5 kb locus some 70 % of the records repeat an
// earlier key, so the work is split:
//   filter_records   the record filters, in stream order inside groups of equal read ids (cheap, parallel over id-aligned chunks)
//   group_records    records with equal keys -> the first of them (hash partitions, exact key compare)
//   decode_groups    decode + error correction + cmp_list2 + identify_ambigious_diffs + haplotypes + exon clipping + piece
//                    interning, ONCE per distinct key that some kept record carries
//   emit_chunk       the pair protocol over the kept records: set union of the mates' haplotypes -> piece refs
// Results are identical to decoding every record where it stands (pinned by the golden traces and the batch comparisons).
struct HtRec {                       // one haplotype of a mate + its add_count arguments as interned pieces
    Ht ht;
    uint32_t exon_off, n_exon;       // local piece ids in the owning worker's ref pool
    uint32_t gene_local;
    uint32_t worker;
};
struct MateOut {                     // decode result of one distinct key
    uint8_t state = 0;               // 1 = haplotypes follow, 2 = decode() said no (record dropped), 3 = the reference would fail
    uint32_t worker = 0, first = 0, n = 0;          // HtRec range in the worker's arena
    int err_code = 0;
    std::string err, trace;
};
struct DecodeWorker {
    hgx_batch table;                 // pieces interned by this worker (local ids)
    std::vector<HtRec> arena;
    std::vector<uint32_t> refs;      // exon-level piece ids of the arena's haplotypes
};
struct ChunkOut {
    PVec<uint32_t> pair_ref;
    PVec<int32_t> pair_off{0};
    int32_t n_reads = 0;
    std::vector<TraceRec> trace;
    std::string error;
    int error_code = 0;
};

// record filters of core:815-872 over records [i0, i1) (a range that starts and ends at read-id boundaries)
void filter_records(const hgx_parse_opts &o, Fields *recs, const uint8_t *ok, size_t i0, size_t i1) {
    const char *grp = nullptr;
    size_t grp_len = 0;
    bool g_l = false, g_r = false, g_u = false;
    for (size_t i = i0; i < i1; ++i) {
        if (!ok[i]) continue;
        Fields &f = recs[i];
        f.kept = KEPT_NO;
        // The stream is name-grouped, so the reference's global left/right/unpaired id sets (core:857-872) reduce to
        // three flags per group of equal read ids.
        const size_t idlen = read_id_len(f, o.simulation != 0);
        if (!grp || grp_len != idlen || memcmp(grp, f.qname, idlen) != 0) {
            grp = f.qname;
            grp_len = idlen;
            g_l = g_r = g_u = false;
        }
        if (f.pos - (o.base_locus + 1) < 0) continue;
        if (f.flag & 0x4) continue;
        if (f.bad_tag) { f.kept = KEPT_ERR_TAGVAL; continue; }              // int(col[5:]) raises while the tags are read (core:838-841)
        if (!f.has_nm || !f.has_nh) { f.kept = KEPT_ERR_TAGS; continue; }
        if (f.nm > o.num_editdist) continue;
        if (f.nh > 1) continue;
        if (!o.allow_discordant && !(f.flag & 0x2)) continue;
        if (f.flag & 0x40) {
            if (g_l) continue;
            g_l = true;
        } else if (f.flag & 0x80) {
            if (g_r) continue;
            g_r = true;
        } else {
            if (!o.allow_discordant) { f.kept = KEPT_ERR_FLAG; continue; }
            if (g_u) continue;
            g_u = true;
        }
        f.kept = KEPT_YES;
    }
}

// rep / n_pile / slot of every record.  Partition p owns the keys with (hash >> 44) % P == p: its worker walks the records of
// the partition in stream order, so rep is the FIRST record of each key whatever the number of workers.
void group_records(const hgx_parse_opts &o, Fields *recs, const uint8_t *ok, size_t n, int n_threads, std::vector<uint32_t> &reps) {
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, n / 4096 + 1));
    const int P = T;
    std::vector<std::vector<std::vector<uint32_t>>> bucket(T, std::vector<std::vector<uint32_t>>(P));
    hgx_par_ranges(T, n, [&](int t, size_t b, size_t e) {
        for (auto &v : bucket[t]) v.reserve((e - b) / P + 16);
        for (size_t i = b; i < e; ++i)
            if (ok[i]) bucket[t][(recs[i].key >> 44) % P].push_back((uint32_t)i);
    });
    hgx_par_tasks(T, (size_t)P, [&](int, size_t p) {
        size_t cnt = 0;
        for (int t = 0; t < T; ++t) cnt += bucket[t][p].size();
        size_t cap = 64;
        while (cap < 2 * cnt) cap <<= 1;
        std::vector<uint32_t> slot(cap, NO_SLOT);
        const size_t mask = cap - 1;
        for (int t = 0; t < T; ++t) {
            const std::vector<uint32_t> &bk = bucket[t][p];
            for (size_t bi = 0; bi < bk.size(); ++bi) {
                const uint32_t i = bk[bi];
                if (bi + 8 < bk.size()) {                               // the walk is a chain of cache misses: look ahead
                    const Fields &nx = recs[bk[bi + 8]];
                    __builtin_prefetch(&nx);
                    __builtin_prefetch(&slot[(size_t)recs[bk[bi + 4]].key & mask]);
                }
                Fields &f = recs[i];
                size_t h = (size_t)f.key & mask;
                uint32_t r;
                for (;;) {
                    r = slot[h];
                    if (r == NO_SLOT) { slot[h] = i; r = i; f.n_pile = 0; f.slot = NO_SLOT; break; }
                    if (same_decode_key(recs[r], f)) break;
                    h = (h + 1) & mask;
                }
                f.rep = r;
                Fields &g = recs[r];
                // pileup membership (common:1076-1090): aligned, inside the locus, concordant unless discordant pairs count
                if (!(f.flag & 0x4) && f.pos - (o.base_locus + 1) >= 0 && (o.allow_discordant || (f.flag & 0x2))) g.n_pile++;
                if (f.kept == KEPT_YES) g.slot = 0;                     // some kept record carries this key: it will be decoded
            }
        }
    });
    // slots = the keys to decode, numbered in stream order of their first record (so that a single worker creates novel variants
    // in stream order): a flag per record, counted per range, prefix-summed -- no serial sort of a quarter of a million indices
    std::vector<size_t> cnt_t(T + 1, 0);
    hgx_par_ranges(T, n, [&](int t, size_t b, size_t e) {
        size_t c = 0;
        for (size_t i = b; i < e; ++i) c += ok[i] && recs[i].rep == i && recs[i].slot == 0;
        cnt_t[t + 1] = c;
    });
    for (int t = 0; t < T; ++t) cnt_t[t + 1] += cnt_t[t];
    reps.assign(cnt_t[T], 0);
    hgx_par_ranges(T, n, [&](int t, size_t b, size_t e) {
        size_t k = cnt_t[t];
        for (size_t i = b; i < e; ++i)
            if (ok[i] && recs[i].rep == i && recs[i].slot == 0) { recs[i].slot = (uint32_t)k; reps[k++] = (uint32_t)i; }
    });
}

// decode result of record `f` (the first of its key): haplotypes with their pieces, into the worker's arena
// HGX_DECODE_PROF (compile-time): cycle counters of decode_one's stages, printed by parse_lines under HGX_PARSE_PROFILE
#ifdef HGX_DECODE_PROF
#include <x86intrin.h>
static thread_local unsigned long long g_dprof[4];
#define PROF_T(v) const unsigned long long v = __rdtsc()
#define PROF_ADD(k, v) g_dprof[k] += __rdtsc() - v
#else
#define PROF_T(v)
#define PROF_ADD(k, v)
#endif
void decode_one(Parser &P, const hgx_locus &L, const hgx_parse_opts &o, const Fields &f, uint32_t widx, DecodeWorker &W, MateOut &out,
                std::string &read, std::vector<Cmp> &cl, std::vector<Cmp> &c2, std::vector<Parser::AltSide> &lset,
                std::vector<Parser::AltSide> &rset, std::vector<int> &mid, std::vector<Ht> &ex, std::vector<int32_t> &ids) {
    try {
        const int pos = f.pos - (o.base_locus + 1);
        PROF_T(t0);
        read.assign(f.seq, f.seq_len);
        const bool kept = P.decode(pos, f.cigar, read, f.zs, f.md, cl);
        PROF_ADD(0, t0);
        if (!kept) { out.state = 2; return; }
        PROF_T(t1);
        // cmp_list2 (core:1351-1368)
        c2.clear();
        for (const Cmp &c : cl) {
            if (c.type == T_MATCH) {
                if (!c2.empty() && c2.back().type == T_MATCH) c2.back().len += c.len;
                else c2.push_back(c);
            } else if (c.type == T_MISMATCH && (c.id == -1 || c.id >= L.V)) {
                if (!c2.empty() && c2.back().type == T_MATCH) c2.back().len += 1;
                else c2.push_back(Cmp{T_MATCH, c.pos, 1, -2});
            } else c2.push_back(c);
        }
        PROF_ADD(1, t1);
        PROF_T(t2);
        int cleft, cright;
        P.ambiguous(c2, cleft, cright, lset, rset);
        PROF_ADD(2, t2);
        PROF_T(t3);
        mid.clear();
        for (int k = cleft; k <= cright; ++k)
            if (c2[k].type != T_MATCH) mid.push_back(c2[k].id);
        out.worker = widx;
        out.first = (uint32_t)W.arena.size();
        auto intern = [&](const Ht &h) -> uint32_t {
            if (h.left > h.right) throw RefError("assert left <= right");
            ids.assign(h.ids.begin(), h.ids.end());
            for (auto &v : ids) if (v >= L.V) v = -1;
            const int64_t id = hgx_intern_piece(W.table, L, h.left, h.right, ids.data(), (int32_t)ids.size());
            if (id < 0) throw std::runtime_error(hgx_last_error());
            return (uint32_t)id;
        };
        for (auto &l : lset)
            for (auto &r : rset) {
                Ht h;
                h.left = l.coord;
                h.right = r.coord;
                h.ids = l.ids;
                h.ids.insert(h.ids.end(), mid.begin(), mid.end());
                h.ids.insert(h.ids.end(), r.ids.begin(), r.ids.end());
                bool dup = false;
                for (uint32_t k = out.first; k < W.arena.size() && !dup; ++k) dup = W.arena[k].ht == h;
                if (dup) continue;
                HtRec rec;
                rec.worker = widx;
                rec.exon_off = (uint32_t)W.refs.size();
                if (L.base_kind == HGX_BASE_HLA) {                   // exon pieces of the haplotype (core:1259-1276)
                    ex.clear();
                    P.exon_pieces(h, ex);
                    for (const Ht &e : ex) W.refs.push_back(intern(e));
                }
                rec.n_exon = (uint32_t)W.refs.size() - rec.exon_off;
                rec.gene_local = intern(h);
                rec.ht = std::move(h);
                W.arena.push_back(std::move(rec));
            }
        out.n = (uint32_t)W.arena.size() - out.first;
        out.state = 1;
        PROF_ADD(3, t3);
        if (o.keep_trace) {
            std::string t;
            for (size_t k = 0; k < c2.size(); ++k) {
                if (k) t += ',';
                t += kTypeName[c2[k].type];
                t += ':' + std::to_string(c2[k].pos) + ':' + std::to_string(c2[k].len);
                if (c2[k].type != T_MATCH) t += ':' + P.vname(c2[k].id);
            }
            t += '\t' + std::to_string(cleft) + '\t' + std::to_string(cright) + '\t';
            std::vector<std::string> ls, rs;
            for (auto &l : lset) ls.push_back(std::to_string(l.coord) + (l.ids.empty() ? "" : "-" + P.join_ids(l.ids)));
            for (auto &r : rset) rs.push_back((r.ids.empty() ? "" : P.join_ids(r.ids) + "-") + std::to_string(r.coord));
            std::sort(ls.begin(), ls.end());
            std::sort(rs.begin(), rs.end());
            for (size_t k = 0; k < ls.size(); ++k) t += (k ? ";" : "") + ls[k];
            t += '\t';
            for (size_t k = 0; k < rs.size(); ++k) t += (k ? ";" : "") + rs[k];
            out.trace = std::move(t);
        }
    } catch (const RefError &e) {
        out.state = 3;
        out.err = std::string("the reference would fail on this input: ") + e.what();
        out.err_code = HGX_EPARSE;
    } catch (const std::exception &e) {
        out.state = 3;
        out.err = e.what();
        out.err_code = HGX_EINVAL;
    }
}

// choose_pairs (core:680-716) over haplotype records: keep the mate pairs whose inner distance is closest to the expected one
void choose_pairs_rec(std::vector<const HtRec *> &lh, std::vector<const HtRec *> &rh, long expected) {
    if (lh.empty() || rh.empty() || std::max(lh.size(), rh.size()) < 2) return;
    long best = -1;
    std::vector<const HtRec *> nl, nr;
    auto add = [](std::vector<const HtRec *> &v, const HtRec *h) {
        for (const HtRec *x : v) if (x->ht == h->ht) return;
        v.push_back(h);
    };
    for (const HtRec *l : lh)
        for (const HtRec *r : rh) {
            const long inter = l->ht.right < r->ht.right ? (long)r->ht.left - l->ht.right - 1 : (long)l->ht.left - r->ht.right - 1;
            const long cur = std::labs(expected - inter);
            if (best < 0 || cur < best) { best = cur; nl.clear(); nr.clear(); }
            if (cur == best) { add(nl, l); add(nr, r); }
        }
    lh.swap(nl);
    rh.swap(nr);
}

// The pair protocol (core:1238-1347, 1545-1587) over records [i0, i1): kept records bring their key's haplotypes; when the
// read id changes the previous pair is flushed: set union of the two mates' haplotypes (left's first), every haplotype's exon
// pieces and then every haplotype itself as piece refs.  Chunks start at read-id boundaries, so pairs never straddle two.
void emit_chunk(const hgx_parse_opts &o, const Fields *recs, const uint8_t *ok, size_t i0, size_t i1, const std::vector<MateOut> &outs,
                const std::vector<DecodeWorker> &workers, const std::vector<std::vector<uint32_t>> &final_id, bool is_last,
                long expected_interdist, ChunkOut &out) {
    std::vector<const HtRec *> lhts, rhts, uni;
    const char *prev_id = nullptr;
    size_t prev_len = 0;
    bool have_prev = false;
    auto flush = [&]() {
        uni.assign(lhts.begin(), lhts.end());
        for (const HtRec *h : rhts) {
            bool dup = false;
            for (const HtRec *x : uni) if (x->ht == h->ht) { dup = true; break; }
            if (!dup) uni.push_back(h);
        }
        size_t n_exon = 0;
        for (const HtRec *h : uni) n_exon += h->n_exon;
        if (n_exon > 65535 || uni.size() > 65535) throw std::runtime_error("more than 65535 pieces for one pair and level");
        for (const HtRec *h : uni) {
            const uint32_t *r = workers[h->worker].refs.data() + h->exon_off;
            for (uint32_t k = 0; k < h->n_exon; ++k) out.pair_ref.push_back(final_id[h->worker][r[k]]);
        }
        for (const HtRec *h : uni) out.pair_ref.push_back(final_id[h->worker][h->gene_local] | 0x80000000u);
        out.pair_off.push_back((int32_t)out.pair_ref.size());
    };
    try {
        for (size_t i = i0; i < i1; ++i) {
            if (!ok[i]) continue;
            const Fields &f = recs[i];
            if (f.kept == KEPT_NO) continue;
            if (f.kept == KEPT_ERR_TAGS) throw RefError("TypeError: record without NM/NH tag (quirk Q8)");
            if (f.kept == KEPT_ERR_TAGVAL)
                throw RefError(std::string("ValueError: invalid literal for int() with base 10: '") + std::string(f.bad_tok ? f.bad_tok : "").substr(0, 60) + "'");
            if (f.kept == KEPT_ERR_FLAG) throw RefError("assert allow_discordant");
            const MateOut &m = outs[recs[f.rep].slot];
            if (m.state == 3) { out.error = m.err; out.error_code = m.err_code; return; }
            if (m.state != 1) continue;
            out.n_reads++;
            const size_t idlen = read_id_len(f, o.simulation != 0);
            if (!have_prev || prev_len != idlen || memcmp(prev_id, f.qname, idlen) != 0) {
                if (have_prev) flush();
                lhts.clear();
                rhts.clear();
            }
            std::vector<const HtRec *> &dst = (f.flag & 0x40) ? lhts : rhts;
            const HtRec *a = workers[m.worker].arena.data() + m.first;
            for (uint32_t k = 0; k < m.n; ++k) {
                bool dup = false;
                for (const HtRec *x : dst) if (x->ht == a[k].ht) { dup = true; break; }
                if (!dup) dst.push_back(a + k);
            }
            if (o.keep_trace) out.trace.push_back(TraceRec{m.trace});
            prev_id = f.qname;
            prev_len = idlen;
            have_prev = true;
        }
        if (have_prev) {
            if (is_last && o.codis_choose_pairs) choose_pairs_rec(lhts, rhts, expected_interdist);   // core:1547-1552
            flush();
        }
    } catch (const RefError &e) {
        out.error = std::string("the reference would fail on this input: ") + e.what();
        out.error_code = HGX_EPARSE;
    } catch (const std::exception &e) {
        out.error = e.what();
        out.error_code = HGX_EINVAL;
    }
}

// get_pair_interdist (common:1187-1265): median inner distance of unique concordant pairs (CODIS D18S51 only).  The distances of this
// stream's pairs, in stream order:
void pair_dists(const Fields *recs, const uint8_t *ok, size_t n_recs, bool simulation, std::vector<long> &dists) {
    dists.clear();
    std::string prev;
    bool hp = false;
    std::vector<std::pair<long, long>> rd;
    for (size_t i = 0; i < n_recs; ++i) {
        if (!ok[i]) continue;
        const Fields &f = recs[i];
        if (f.flag & 0x4) continue;
        if (!f.has_nh || f.nh > 1 || !f.yt_cp) continue;
        std::string id(f.qname, read_id_len(f, simulation));
        if (hp && id != prev) {
            if (rd.size() == 2)
                dists.push_back(rd[0].first <= rd[1].first ? rd[1].first - rd[0].second - 1 : rd[0].first - rd[1].second - 1);
            rd.clear();
        }
        long right = f.pos;
        for (const char *p = f.cigar; *p;) {
            char *e;
            const long n = strtol(p, &e, 10);
            if (e == p || !*e) break;
            if (*e == 'M' || *e == 'N' || *e == 'D') right += n;
            p = e + 1;
        }
        rd.push_back({(long)f.pos, right - 1});
        prev = id;
        hp = true;
    }
}
// ... as the histogram the shards of a locus exchange (hgx.h: HGX_INTERDIST_BINS counters)
void interdist_hist(const std::vector<long> &dists, std::vector<int64_t> &hist) {
    hist.assign((size_t)HGX_INTERDIST_BINS, 0);
    for (long d : dists) {
        const long b = d < -(long)HGX_INTERDIST_HALF ? 0 : (d > (long)HGX_INTERDIST_HALF - 1 ? HGX_INTERDIST_BINS - 1 : 1 + d + HGX_INTERDIST_HALF);
        hist[(size_t)b] += 1;
    }
}
long pair_interdist(const Fields *recs, const uint8_t *ok, size_t n_recs, bool simulation, const hgx_parse_opts *opts = nullptr) {
    std::vector<long> dists;
    pair_dists(recs, ok, n_recs, simulation, dists);
    if (opts && opts->interdist_exchange) {
        // reads of the sample on several ranks: the median of ALL distances from the summed histogram (element int(len / 2) of the
        // sorted list, common:1258-1262)
        std::vector<int64_t> hist;
        interdist_hist(dists, hist);
        if (opts->interdist_exchange(opts->interdist_ctx, hist.data(), (int64_t)hist.size()) != 0)
            throw std::runtime_error("inter-distance exchange between the ranks of a sharded locus failed");
        long long expected = -1;
        if (hgx_interdist_median(hist.data(), &expected)) throw std::runtime_error("median pair distance outside the exchanged histogram's range");
        return (long)expected;
    }
    std::sort(dists.begin(), dists.end());
    return dists.empty() ? -1 : dists[dists.size() / 2];
}

// Worker piece tables -> the batch's table.  Every decode worker interned its pieces into a private table; here the tables are
// united without a serial pass over them: (1) per worker, its distinct pieces are bucketed by hash partition; (2) per partition,
// one task interns the bucket entries of all workers into a partition-private table; (3) partition sizes are prefix-summed into
// global piece ids.  The ids carry no meaning yet: canonical_piece_order below orders the table by content, so the batch is the
// same whatever the number of workers.  gid[w][local id] = id in B.
void merge_tables(hgx_batch &B, std::vector<DecodeWorker> &workers, int n_threads, std::vector<std::vector<uint32_t>> &gid) {
    const size_t nw = workers.size();
    gid.assign(nw, {});
    for (size_t w = 0; w < nw; ++w) gid[w].resize(workers[w].table.pieces.size());
    if (nw == 1) {
        B.pieces.swap(workers[0].table.pieces);
        B.masks.swap(workers[0].table.masks);
        for (size_t k = 0; k < gid[0].size(); ++k) gid[0][k] = (uint32_t)k;
        return;
    }
    const int P = std::max(1, std::min(n_threads, 64));
    struct Ent { uint32_t local; };
    std::vector<std::vector<std::vector<Ent>>> bucket(nw, std::vector<std::vector<Ent>>(P));     // [worker][partition]
    hgx_par_tasks(n_threads, nw, [&](int, size_t w) {
        const hgx_batch &lb = workers[w].table;
        for (size_t k = 0; k < lb.pieces.size(); ++k) {
            const hgx_piece &pc = lb.pieces[k];
            const uint64_t h = PieceTable::hash(pc.lo_word, pc.n_words, &lb.masks[pc.mask_off]);
            bucket[w][(h >> 40) % P].push_back(Ent{(uint32_t)k});
        }
    });
    std::vector<hgx_batch> part(P);                                        // partition-private distinct pieces
    hgx_par_tasks(n_threads, (size_t)P, [&](int, size_t p) {
        for (size_t w = 0; w < nw; ++w) {
            const hgx_batch &lb = workers[w].table;
            for (const Ent &e : bucket[w][p]) {
                const hgx_piece &pc = lb.pieces[e.local];
                gid[w][e.local] = hgx_intern_masks(part[p], pc.lo_word, pc.n_words, &lb.masks[pc.mask_off]);   // partition-local for now
            }
        }
    });
    std::vector<uint32_t> pbase(P + 1, 0), mbase(P + 1, 0);
    for (int p = 0; p < P; ++p) {
        pbase[p + 1] = pbase[p] + (uint32_t)part[p].pieces.size();
        mbase[p + 1] = mbase[p] + (uint32_t)part[p].masks.size();
    }
    B.pieces.resize(pbase[P]);
    B.masks.resize(mbase[P]);
    hgx_par_tasks(n_threads, (size_t)P, [&](int, size_t p) {
        for (size_t k = 0; k < part[p].pieces.size(); ++k) {
            hgx_piece pc = part[p].pieces[k];
            pc.mask_off += mbase[p];
            B.pieces[pbase[p] + k] = pc;
        }
        if (!part[p].masks.empty()) memcpy(&B.masks[mbase[p]], part[p].masks.data(), part[p].masks.size() * 4);
    });
    hgx_par_tasks(n_threads, nw, [&](int, size_t w) {
        for (int p = 0; p < P; ++p)
            for (const Ent &e : bucket[w][p]) gid[w][e.local] += pbase[p];
    });
}

// The input of the device stages (hgx_front_input): the distinct keys that count into the pileup or are decoded, in stream order
// of their first records, their text gathered into staging memory (cigar | seq | zs | md per key), and one word per record that
// passed the filters: decode slot of its key | left mate << 30 | first kept record of its read id << 31.  Returns 0, or the
// reason (HGX_FE_DECLINE_*) why this input stays on the host.
int build_front_input(const hgx_parse_opts &o, const Fields *recs, const uint8_t *ok, size_t n, const std::vector<size_t> &cut, int n_threads,
                      size_t n_slots, hgx_front_input &in) {
    if (n >= (1ull << 30) || n_slots >= (1ull << 30)) return HGX_FE_DECLINE_SIZE;
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, n / 4096 + 1));
    std::vector<size_t> kcnt(T + 1, 0), kbytes(T + 1, 0);
    auto is_key = [&](size_t i) { return ok[i] && recs[i].rep == i && (recs[i].n_pile > 0 || recs[i].slot != NO_SLOT); };
    auto key_bytes = [&](const Fields &f) { return (size_t)f.cigar_len + f.seq_len + f.zs_len + f.md_len; };
    std::atomic<int> too_long{0};
    hgx_par_ranges(T, n, [&](int t, size_t b, size_t e) {
        size_t c = 0, by = 0;
        for (size_t i = b; i < e; ++i)
            if (is_key(i)) {
                const Fields &f = recs[i];
                if (f.cigar_len > 65535 || f.zs_len > 65535 || f.md_len > 65535 || f.seq_len > (1u << 24)) too_long.store(1);
                ++c;
                by += key_bytes(f);
            }
        kcnt[t + 1] = c;
        kbytes[t + 1] = by;
    });
    if (too_long.load()) return HGX_FE_DECLINE_SIZE;
    for (int t = 0; t < T; ++t) { kcnt[t + 1] += kcnt[t]; kbytes[t + 1] += kbytes[t]; }
    if (kbytes[T] >= (1ull << 32) - 64) return HGX_FE_DECLINE_SIZE;
    in.n_keys = kcnt[T];
    in.n_text = kbytes[T];
    in.n_slots = n_slots;
    in.keys = (FeKey *)in.mem.alloc(std::max<size_t>(in.n_keys, 1) * sizeof(FeKey));
    in.text = (char *)in.mem.alloc(in.n_text + 64);
    if (!in.keys || !in.text) throw std::bad_alloc();
    hgx_par_ranges(T, n, [&](int t, size_t b, size_t e) {
        size_t k = kcnt[t], at = kbytes[t];
        for (size_t i = b; i < e; ++i) {
            if (!is_key(i)) continue;
            const Fields &f = recs[i];
            FeKey &K = in.keys[k++];
            K.pos = f.pos - (o.base_locus + 1);
            K.n_pile = f.n_pile;
            K.slot = f.slot;
            K.cigar_off = (uint32_t)at;
            K.seq_off = K.cigar_off + f.cigar_len;
            K.zs_off = K.seq_off + (uint32_t)f.seq_len;
            K.md_off = K.zs_off + f.zs_len;
            K.seq_len = (uint32_t)f.seq_len;
            K.cigar_len = (uint16_t)f.cigar_len;
            K.zs_len = (uint16_t)f.zs_len;
            K.md_len = (uint16_t)f.md_len;
            K.flags = (uint16_t)((f.zs ? FE_K_HAS_ZS : 0) | (f.md ? FE_K_HAS_MD : 0));
            K.task = 0;
            char *w = in.text + at;
            memcpy(w, f.cigar, f.cigar_len); w += f.cigar_len;
            memcpy(w, f.seq, f.seq_len); w += f.seq_len;
            if (f.zs_len) memcpy(w, f.zs, f.zs_len);
            w += f.zs_len;
            if (f.md_len) memcpy(w, f.md, f.md_len);
            at += key_bytes(f);
        }
    });
    // the records that passed the filters, chunk by chunk (chunks start where the read id changes)
    const size_t nc = cut.size() - 1;
    std::vector<size_t> rcnt(nc + 1, 0);
    std::atomic<int> bad{0};
    hgx_par_tasks(n_threads, nc, [&](int, size_t c) {
        size_t k = 0;
        for (size_t i = cut[c]; i < cut[c + 1]; ++i) {
            if (!ok[i] || recs[i].kept == KEPT_NO) continue;
            if (recs[i].kept != KEPT_YES) { bad.store(1); break; }
            ++k;
        }
        rcnt[c + 1] = k;
    });
    if (bad.load()) return HGX_FE_DECLINE_RECORD;                 // the reference raises on such a record: the host stages say how
    for (size_t c = 0; c < nc; ++c) rcnt[c + 1] += rcnt[c];
    in.n_rec = rcnt[nc];
    in.rec_info = (uint32_t *)in.mem.alloc(std::max<size_t>(in.n_rec, 1) * 4);
    if (!in.rec_info) throw std::bad_alloc();
    hgx_par_tasks(n_threads, nc, [&](int, size_t c) {
        size_t k = rcnt[c];
        const Fields *prev = nullptr;
        size_t prev_len = 0;
        for (size_t i = cut[c]; i < cut[c + 1]; ++i) {
            if (!ok[i] || recs[i].kept != KEPT_YES) continue;
            const Fields &f = recs[i];
            const size_t idlen = read_id_len(f, o.simulation != 0);
            const bool head = !prev || prev_len != idlen || memcmp(prev->qname, f.qname, idlen) != 0;
            in.rec_info[k++] = recs[f.rep].slot | ((f.flag & 0x40) ? 1u << 30 : 0u) | (head ? 1u << 31 : 0u);
            prev = &f;
            prev_len = idlen;
        }
    });
    return 0;
}

template <class F>
void parallel_for(int n_threads, size_t n, F fn) {   // fn(thread, begin, end), on the persistent worker pool
    hgx_par_ranges(n_threads, n, fn);
}

}   // namespace

static int parse_lines(hgx_batch **out, const hgx_locus *Lc, hgx_line *lines, size_t n, const hgx_parse_opts *opts, bool binary = false,
                       hgx_front_hook *hook = nullptr, const char *raw = nullptr, size_t raw_bytes = 0, const hgx_bam_deferred *def = nullptr);

// SAM text (name-grouped) -> private writable copy + line table -> parse_lines
int hgx_parse_sam_hook(hgx_batch **out, const hgx_locus *Lc, const char *sam, size_t n_bytes, const hgx_parse_opts *opts, hgx_front_hook *hook) {
    HARGCHK(out && Lc && (sam || n_bytes == 0) && opts);
    try {
        int n_threads = opts->n_threads > 0 ? opts->n_threads : hgx_default_threads();
        n_threads = std::max(1, std::min(n_threads, 512));
        struct PoolFree { void operator()(char *p) const { hgx_host_free(p); } };
        std::unique_ptr<hgx_big_alloc_scope> pinned;                       // the record route uploads the text: staging memory for it
        if (hook && hook->records && hook->mem.alloc) pinned.reset(new hgx_big_alloc_scope(hook->mem, 1u << 20));
        std::unique_ptr<char, PoolFree> text((char *)hgx_host_alloc(n_bytes + 1));      // tokens are NUL-terminated in place
        pinned.reset();
        char *base = text.get(), *end = base + n_bytes;
        const int nt = n_bytes > (8u << 20) ? n_threads : 1;
        std::vector<std::vector<hgx_line>> part(nt);
        parallel_for(nt, n_bytes, [&](int, size_t b0, size_t e0) { memcpy(base + b0, sam + b0, e0 - b0); });
        *end = '\n';
        parallel_for(nt, n_bytes, [&](int t, size_t b0, size_t e0) {
            char *p = base + b0;                    // this worker owns the lines that START in [b0, e0)
            if (b0 > 0) {
                char *q = (char *)memchr(p - 1, '\n', end - (p - 1));
                p = q ? q + 1 : end;
            }
            part[t].reserve((e0 - b0) / 300 + 16);
            while (p < base + e0 && p < end) {
                char *e = (char *)memchr(p, '\n', end - p);
                if (!e) e = end;
                if (e > p && *p != '@') part[t].push_back(hgx_line{p, (uint32_t)(e - p), 0, 0});
                p = e + 1;
            }
        });
        std::vector<size_t> off(nt + 1, 0);
        for (int t = 0; t < nt; ++t) off[t + 1] = off[t] + part[t].size();
        std::vector<hgx_line> lines(off[nt]);
        parallel_for(nt, (size_t)nt, [&](int, size_t b, size_t e) {
            for (size_t t = b; t < e; ++t)
                if (!part[t].empty()) memcpy(&lines[off[t]], part[t].data(), part[t].size() * sizeof(hgx_line));
        });
        if (hook && hook->on_raw) hook->on_raw(base, n_bytes, 0, n_bytes);
        return parse_lines(out, Lc, lines.data(), lines.size(), opts, false, hook, base, n_bytes);
    } catch (const std::exception &e) {
        hgx_set_error("%s", e.what());
        return HGX_EINVAL;
    }
}

extern "C" int hgx_parse_sam(hgx_batch **out, const hgx_locus *Lc, const char *sam, size_t n_bytes, const hgx_parse_opts *opts) {
    return hgx_parse_sam_hook(out, Lc, sam, n_bytes, opts, nullptr);
}

extern "C" int hgx_parse_alignment_file(hgx_batch **out, const hgx_locus *Lc, const char *path, const char *regions,
                                        const hgx_parse_opts *opts) {
    return hgx_parse_alignment_file_hook(out, Lc, path, regions, opts, nullptr);
}

int hgx_parse_alignment_file_hook(hgx_batch **out, const hgx_locus *Lc, const char *path, const char *regions, const hgx_parse_opts *opts,
                                  hgx_front_hook *hook) {
    HARGCHK(out && Lc && path && opts);
    const bool prof = getenv("HGX_PARSE_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    try {
        const double t0 = now();
        hgx_align_lines al;                     // the reader's buffers are tokenised in place (no copy, no trip through the caller)
        std::unique_ptr<hgx_big_alloc_scope> pinned;   // the record route uploads the text / the inflated stream: staging memory for them
        if (hook && hook->records && hook->mem.alloc) {
            pinned.reset(new hgx_big_alloc_scope(hook->mem, 4u << 20));
            al.on_raw = hook->on_raw;
            al.defer_walk = hook->defer_walk;
            al.defer_min_bytes = hook->defer_min_bytes;
            if (al.defer_walk) { al.inflate_dev = hook->inflate_dev; al.comp_early = hook->comp_early; al.comp_sync = hook->comp_sync; }
            al.defer_text = al.defer_walk && hook->defer_text;
        }
        int rc = hgx_read_alignment_lines(path, regions, opts->n_threads, al, /*keep_binary=*/true);
        pinned.reset();
        if (rc) return rc;
        const double t1 = now();
        if (al.deferred.on) {
            // a BAM stream whose records the device walks, filters and sorts itself; should it decline, the file is read again
            // the ordinary way (a rare path: a record the kernels do not take, a chain that does not link up)
            rc = parse_lines(out, Lc, nullptr, 0, opts, !al.deferred.text, hook, al.raw, al.raw_bytes, &al.deferred);
            if (rc || !hook->declined_records) {
                if (prof) fprintf(stderr, "[hgx_parse_alignment_file] read %.1f ms, parse %.1f ms (records walked on the device)\n", (t1 - t0) * 1e3, (now() - t1) * 1e3);
                return rc;
            }
            hgx_batch_destroy(*out);
            *out = nullptr;
            hgx_align_lines al2;
            pinned.reset(new hgx_big_alloc_scope(hook->mem, 32u << 20));
            al2.on_raw = hook->on_raw;
            rc = hgx_read_alignment_lines(path, regions, opts->n_threads, al2, /*keep_binary=*/true);
            pinned.reset();
            if (rc) return rc;
            return parse_lines(out, Lc, al2.lines.data(), al2.lines.size(), opts, al2.binary, hook, al2.raw, al2.raw_bytes);
        }
        rc = parse_lines(out, Lc, al.lines.data(), al.lines.size(), opts, al.binary, hook, al.raw, al.raw_bytes);
        if (prof) fprintf(stderr, "[hgx_parse_alignment_file] read %.1f ms, parse %.1f ms\n", (t1 - t0) * 1e3, (now() - t1) * 1e3);
        return rc;
    } catch (const std::exception &e) {
        hgx_set_error("hgx_parse_alignment_file: %s", e.what());
        return HGX_ENOMEM;
    }
}

// ---- many tasks of one locus, read side by side (hgx_internal.hpp: hgx_many_streams) --------------------------------------------
int hgx_many_read(hgx_many_streams &ms, const char *const *paths, const char *const *regions, const char *const *sams, const size_t *sam_bytes,
                  int n_tasks, int n_threads, const hgx_front_alloc *mem, const std::function<int(int task)> &on_task) {
    HARGCHK(n_tasks >= 0 && (n_tasks == 0 || paths || (sams && sam_bytes)));
    ms.n_tasks = n_tasks;
    ms.al.reset(new hgx_align_lines[(size_t)std::max(n_tasks, 1)]);
    ms.raw.assign((size_t)n_tasks, nullptr);
    ms.raw_bytes.assign((size_t)n_tasks, 0);
    ms.base.assign((size_t)n_tasks + 1, 0);
    ms.line_base.assign((size_t)n_tasks + 1, 0);
    if (n_threads <= 0) n_threads = hgx_default_threads();
    n_threads = std::max(1, std::min(n_threads, 512));
    std::vector<int> rcs((size_t)n_tasks, HGX_OK);
    std::vector<std::string> errs((size_t)n_tasks);
    // few tasks: the threads go to the readers' own phases (inflate, walk, sort); many: one thread per task
    const int inner = n_tasks > 0 ? std::max(1, n_threads / n_tasks) : 1;
    hgx_par_tasks(std::min(n_threads, std::max(n_tasks, 1)), (size_t)n_tasks, [&](int, size_t t) {
        try {
            hgx_align_lines &al = ms.al[t];
            if (paths) {
                if (!paths[t]) { rcs[t] = HGX_EINVAL; errs[t] = "task without a path"; return; }
                std::unique_ptr<hgx_big_alloc_scope> pinned;
                if (mem && mem->alloc) pinned.reset(new hgx_big_alloc_scope(*mem, 1u << 20));
                rcs[t] = hgx_read_alignment_lines(paths[t], regions ? regions[t] : nullptr, inner, al, /*keep_binary=*/true);
                pinned.reset();
                if (rcs[t]) { errs[t] = hgx_last_error(); return; }
                ms.raw[t] = al.raw;
                ms.raw_bytes[t] = al.raw_bytes;
            } else {
                const char *base = sams[t], *end = base + sam_bytes[t];
                if (!base && sam_bytes[t]) { rcs[t] = HGX_EINVAL; errs[t] = "task without text"; return; }
                for (const char *p = base; p < end;) {
                    const char *e = (const char *)memchr(p, '\n', (size_t)(end - p));
                    if (!e) e = end;
                    if (e > p && *p != '@') al.lines.push_back(hgx_line{const_cast<char *>(p), (uint32_t)(e - p), 0, 0});
                    p = e + 1;
                }
                ms.raw[t] = base;
                ms.raw_bytes[t] = sam_bytes[t];
            }
            if (on_task) {
                rcs[t] = on_task((int)t);
                if (rcs[t]) errs[t] = hgx_last_error();
            }
        } catch (const std::exception &e) {
            rcs[t] = HGX_ENOMEM;
            errs[t] = e.what();
        }
    });
    for (int t = 0; t < n_tasks; ++t)
        if (rcs[t]) { hgx_set_error("task %d: %s", t, errs[t].c_str()); return rcs[t]; }
    bool any = false;
    for (int t = 0; t < n_tasks; ++t) {
        if (ms.al[t].lines.size() > 0) {
            if (any && ms.al[t].binary != ms.binary) ms.mixed = true;
            if (!any) ms.binary = ms.al[t].binary;
            any = true;
        }
        if (!on_task) ms.base[t + 1] = (ms.base[t] + ms.raw_bytes[t] + 63) & ~(size_t)63;
        ms.line_base[t + 1] = ms.line_base[t] + ms.al[t].lines.size();
    }
    return HGX_OK;
}

void hgx_many_lines(const hgx_many_streams &ms, FeLine *dst, int n_threads) {
    if (n_threads <= 0) n_threads = hgx_default_threads();
    const size_t skip = ms.binary ? 32 : 0;
    hgx_par_tasks(std::max(1, std::min(n_threads, std::max(ms.n_tasks, 1))), (size_t)ms.n_tasks, [&](int, size_t t) {
        const hgx_align_lines &al = ms.al[t];
        FeLine *d = dst + ms.line_base[t];
        const hgx_line *ln = al.lines.data();
        for (size_t i = 0, n = al.lines.size(); i < n; ++i) {
            d[i].off = (uint32_t)(ms.base[t] + (size_t)(ln[i].p - ms.raw[t]) - skip);
            d[i].len = ln[i].len;
            d[i].task = (uint32_t)t;
        }
    });
}

// lines: name-grouped records.  Text: lines[i].p[lines[i].len] is writable (it becomes the record's terminator).  Binary (BAM
// records as read): lines[i].p = the record's QNAME (32 bytes into the record), lines[i].len = its block_size.
static int parse_lines(hgx_batch **out, const hgx_locus *Lc, hgx_line *lines, size_t n, const hgx_parse_opts *opts, bool binary,
                       hgx_front_hook *hook, const char *raw, size_t raw_bytes, const hgx_bam_deferred *def) {
    HARGCHK(out && Lc && (lines || n == 0) && opts && (!def || (hook && hook->records && (raw || def->on_device))));
    *out = nullptr;
    if (hook && hook->records && (raw || (def && def->on_device))) {
        // the record route of the device front end: fields, filters and key grouping as kernels too -- nothing below runs
        hook->declined_records = 0;
        try {
            const int rc = hook->records(*const_cast<hgx_locus *>(Lc), raw, raw_bytes, lines, n, binary, *opts, &hook->declined_records, def);
            if (rc) return rc;
        } catch (const std::exception &e) {
            hgx_set_error("%s", e.what());
            return HGX_EINVAL;
        }
        if (!hook->declined_records) { hook->declined = 0; return HGX_OK; }
    }
    if (def) return HGX_OK;                           // (declined: the caller reads the file again, with its records walked)
    hgx_locus &L = *const_cast<hgx_locus *>(Lc);
    hgx_batch *B = new hgx_batch();
    try {
        const bool prof = getenv("HGX_PARSE_PROFILE") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t_prev = now();
        auto lap = [&](const char *what) {
            if (!prof) return;
            const double t = now();
            fprintf(stderr, "[hgx_parse_sam] %-22s %8.1f ms\n", what, (t - t_prev) * 1e3);
            t_prev = t;
        };
        hgx_build_alternatives(L);
        const AltTables alts(L);
        lap("alternatives");
        const int n_ref = (int)L.backbone.size();
        int n_threads = opts->n_threads > 0 ? opts->n_threads : hgx_default_threads();
        n_threads = std::max(1, std::min(n_threads, 512));
        if (opts->keep_trace) n_threads = 1;                // traces (and novel-variant numbering) follow stream order
        if (n < 20000) n_threads = 1;
        // per-record tables: written by the split below, never read before (no zero fill of ~100 bytes per record)
        struct PoolFree { void operator()(void *p) const { hgx_host_free(p); } };
        std::unique_ptr<void, PoolFree> recs_mem(hgx_host_alloc(std::max<size_t>(n, 1) * sizeof(Fields)));
        std::unique_ptr<void, PoolFree> ok_mem(hgx_host_alloc(std::max<size_t>(n, 1)));
        Fields *recs = (Fields *)recs_mem.get();
        uint8_t *ok = (uint8_t *)ok_mem.get();
        // field split + decode keys, embarrassingly parallel over lines
        std::vector<CharArena> arenas(binary ? n_threads : 0);
        std::vector<int> bad_rec(n_threads, 0);
        struct BadLine { size_t index = 0; int code = 0, n = 0; std::string tok; };
        std::vector<BadLine> bad_line(n_threads);
        parallel_for(n_threads, n, [&](int t, size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                if (binary) {
                    ok[i] = 1;
                    if (i + 6 < e) {        // name order = a random walk through the inflated stream: fetch the record six ahead
                        const char *nx = lines[i + 6].p - 32;
                        __builtin_prefetch(nx); __builtin_prefetch(nx + 64); __builtin_prefetch(nx + 128); __builtin_prefetch(nx + 192);
                        __builtin_prefetch(nx + 256);
                    }
                    if (!split_bam((const unsigned char *)lines[i].p - 32, lines[i].len, recs[i], arenas[t])) { bad_rec[t] = 1; return; }
                    continue;
                }
                char *line = lines[i].p, *lend = line + lines[i].len;
                *lend = 0;
                int bad_n = 0;
                const char *bad_tok = nullptr;
                const int bad = split_line(line, lend, recs[i], &bad_n, &bad_tok);
                ok[i] = bad ? 0 : 1;
                if (bad && (bad_line[t].code == 0 || i < bad_line[t].index)) {
                    bad_line[t].index = i; bad_line[t].code = bad; bad_line[t].n = bad_n;
                    bad_line[t].tok = bad_tok ? std::string(bad_tok).substr(0, 60) : std::string();
                }
            }
        });
        for (int v : bad_rec) if (v) { hgx_set_error("malformed BAM record"); delete B; return HGX_EPARSE; }
        {   // a line the reference's record loop cannot take apart kills it right there (typing_core.py:803-814): the first such line
            const BadLine *first = nullptr;
            for (const BadLine &b : bad_line) if (b.code && (!first || b.index < first->index)) first = &b;
            if (first) {
                char msg[200];
                if (first->code == 1) snprintf(msg, sizeof(msg), "ValueError: not enough values to unpack (expected 6, got %d)", first->n);
                else if (first->code == 2) snprintf(msg, sizeof(msg), "IndexError: list index out of range");
                else snprintf(msg, sizeof(msg), "ValueError: invalid literal for int() with base 10: '%s'", first->tok.c_str());
                hgx_set_error("the reference would fail on this input: %s (record %zu)", msg, first->index + 1);
                delete B;
                return HGX_EPARSE;
            }
        }
        lap("split");
        // chunks that start where the read id changes (the record filters and the pair protocol work inside them)
        const int n_chunks = n_threads == 1 ? 1 : n_threads * 4;
        std::vector<size_t> cut{0};
        for (int c = 1; c < n_chunks; ++c) {
            size_t i = std::max(cut.back(), n * c / n_chunks);
            while (i < n && i > 0) {
                if (ok[i] && ok[i - 1]) {
                    const size_t la = read_id_len(recs[i], opts->simulation != 0), lb = read_id_len(recs[i - 1], opts->simulation != 0);
                    if (la != lb || memcmp(recs[i].qname, recs[i - 1].qname, la) != 0) break;
                }
                ++i;
            }
            if (i > cut.back() && i < n) cut.push_back(i);
        }
        cut.push_back(n);
        const size_t nc = cut.size() - 1;
        lap("  chunk cuts");
        hgx_par_tasks(n_threads, nc, [&](int, size_t c) { filter_records(*opts, recs, ok, cut[c], cut[c + 1]); });
        lap("  record filters");
        std::vector<uint32_t> reps;                     // first records of the distinct decode keys that some kept record carries
        group_records(*opts, recs, ok, n, n_threads, reps);
        lap("  key grouping");
        if (hook) {
            // The device front end (hgx_front.hip; the lab build's emulation): everything from here on -- pileup, decode of the distinct
            // keys, piece table, pair protocol -- as kernels over the keys' text.  It may decline; the host stages below then run.
            hook->declined = 0;
            if (n < hook->min_records) hook->declined = HGX_FE_DECLINE_SMALL;
            else {
                hgx_front_input in;
                in.mem = hook->mem;
                hook->declined = build_front_input(*opts, recs, ok, n, cut, n_threads, reps.size(), in);
                if (!hook->declined && (opts->codis_choose_pairs || opts->interdist_exchange)) {
                    // CODIS D18S51: this stream's inner distances as the histogram the device stages take the median from (after
                    // their pileup exchange: the order of the exchanges is the host stages')
                    std::vector<long> dists;
                    pair_dists(recs, ok, n, opts->simulation != 0, dists);
                    interdist_hist(dists, in.interdist_hist);
                    in.want_interdist = true;
                }
                lap("  key table for the device");
                if (!hook->declined) {
                    const int rc = hook->run(L, in, *opts, &hook->declined);
                    lap("device stages");
                    if (rc) { delete B; return rc; }
                    if (!hook->declined) { delete B; return HGX_OK; }          // (the hook holds the result)
                }
            }
        }
        // pass 1: pileup over all records (common:1076-1134) = over the distinct keys, each weighted by its group's size
        std::vector<std::vector<uint32_t>> tcounts(n_threads);
        std::vector<std::string> terr(n_threads);
        parallel_for(n_threads, n, [&](int t, size_t b, size_t e) {
            std::vector<uint32_t> &cnt = tcounts[t];
            cnt.assign((size_t)n_ref * 6, 0u);
            static const struct Lut { uint8_t t[256]; Lut() { memset(t, 4, 256); t['A'] = 0; t['C'] = 1; t['G'] = 2; t['T'] = 3; } } slot_of;
            for (size_t i = b; i < e; ++i) {
                if (!ok[i] || recs[i].rep != i || recs[i].n_pile == 0) continue;
                const Fields &f = recs[i];
                const uint32_t w = f.n_pile;
                const int pos = f.pos - (opts->base_locus + 1);
                int rp = 0, gp = pos;
                for (const char *p = f.cigar; *p;) {
                    char *q;
                    const long len = strtol(p, &q, 10);
                    if (q == p || !*q) break;
                    const char op = *q;
                    if (op == 'M') {
                        const long lim = std::min<long>(len, (long)n_ref - gp);
                        if (lim > 0 && (size_t)(rp + lim) > f.seq_len) { terr[t] = "IndexError: read shorter than CIGAR"; break; }
                        uint32_t *c = &cnt[(size_t)gp * 6];
                        const unsigned char *sq = (const unsigned char *)f.seq + rp;
                        for (long j = 0; j < lim; ++j) c[j * 6 + slot_of.t[sq[j]]] += w;
                    } else if (op == 'D') {
                        const long lim = std::min<long>(len, (long)n_ref - gp);
                        for (long j = 0; j < lim; ++j) cnt[(size_t)(gp + j) * 6 + 5] += w;
                    }
                    if (op == 'M' || op == 'N' || op == 'D') gp += (int)len;
                    if (op == 'M' || op == 'I' || op == 'S') rp += (int)len;
                    p = q + 1;
                }
            }
        });
        for (auto &e : terr) if (!e.empty()) throw RefError(e);
        B->counts.assign((size_t)n_ref * 6, 0u);
        B->nt_set.assign(n_ref, 0);
        {   // partial pileups added up, columns split among the workers
            const size_t cells = B->counts.size();
            parallel_for(n_threads, cells, [&](int, size_t b, size_t e) {
                for (auto &cnt : tcounts)
                    if (!cnt.empty())
                        for (size_t k = b; k < e; ++k) B->counts[k] += cnt[k];
            });
        }
        if (opts->pileup_exchange) {             // this shard's counts -> the sum over all shards of the sample (8e)
            if (opts->pileup_exchange(opts->pileup_ctx, B->counts.data(), (int64_t)B->counts.size()) != 0)
                throw std::runtime_error("pileup exchange between the ranks of a sharded locus failed");
        }
        for (int i = 0; i < n_ref; ++i) {
            const uint32_t *c = &B->counts[(size_t)i * 6];
            const uint64_t tot = (uint64_t)c[0] + c[1] + c[2] + c[3] + c[4] + c[5];
            int m = 0;
            if (tot >= 20)
                for (int k = 0; k < 4; ++k)
                    if ((double)c[k] >= (double)tot * 0.2 || c[k] >= 7) m |= 1 << k;
            B->nt_set[i] = (uint8_t)m;
        }
        lap("pileup");
        // (a shard that does not hold the stream's last pair skips choose_pairs but still takes part in the exchange)
        const long expected = (opts->codis_choose_pairs || opts->interdist_exchange) ? pair_interdist(recs, ok, n, opts->simulation != 0, opts) : -1;
        // pass 2a: every distinct key once
        std::vector<MateOut> outs(reps.size());
        const int n_dec = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, reps.size() / 256 + 1));
        std::vector<DecodeWorker> workers(n_dec);
        NovelTable novel;
        {
            const size_t BLOCK = 128;
            const size_t n_blocks = (reps.size() + BLOCK - 1) / BLOCK;
            std::atomic<size_t> next{0};
            std::vector<double> busy(n_dec, 0.0);
            hgx_run_workers(n_dec, [&](int w) {
                const double tb = prof ? now() : 0.0;
                DecodeWorker &W = workers[w];
                Parser P(L, *opts, W.table, *B, alts, novel);
                std::string read;
                std::vector<Cmp> cl, c2;
                std::vector<Parser::AltSide> lset, rset;
                std::vector<int> mid;
                std::vector<Ht> ex;
                std::vector<int32_t> ids;
                for (size_t blk; (blk = next.fetch_add(1)) < n_blocks;)
                    for (size_t k = blk * BLOCK; k < std::min(reps.size(), (blk + 1) * BLOCK); ++k)
                        decode_one(P, L, *opts, recs[reps[k]], (uint32_t)w, W, outs[k], read, cl, c2, lset, rset, mid, ex, ids);
                if (prof) busy[w] = now() - tb;
#ifdef HGX_DECODE_PROF
                if (prof && w == 0)
                    fprintf(stderr, "[decode_one, worker 0] Mcycles: decode %.1f | cmp_list2 %.1f | ambiguous %.1f | haplotypes + intern %.1f\n",
                            g_dprof[0] / 1e6, g_dprof[1] / 1e6, g_dprof[2] / 1e6, g_dprof[3] / 1e6);
#endif
            });
            if (prof) {
                double lo = 1e30, hi = 0, sum = 0;
                for (double v : busy) { lo = std::min(lo, v); hi = std::max(hi, v); sum += v; }
                fprintf(stderr, "[hgx_parse_sam]   decode workers: %d, busy %.1f .. %.1f ms each, %.0f ms in all\n", n_dec, lo * 1e3, hi * 1e3, sum * 1e3);
            }
        }
        lap("decode distinct keys");
        // the workers' piece tables -> one table in canonical order; final_id[w][local id] = id in the batch
        std::vector<std::vector<uint32_t>> final_id;
        merge_tables(*B, workers, n_threads, final_id);
        {
            std::vector<uint32_t> new_id;
            hgx_canonical_piece_order(*B, n_threads, new_id);
            hgx_par_tasks(n_threads, final_id.size(), [&](int, size_t w) { for (auto &v : final_id[w]) v = new_id[v]; });
        }
        lap("piece table");
        // pass 2b: the pair protocol per chunk, then the chunks' refs side by side
        std::vector<ChunkOut> res(nc);
        hgx_par_tasks(n_threads, nc, [&](int, size_t c) {
            emit_chunk(*opts, recs, ok, cut[c], cut[c + 1], outs, workers, final_id, c + 1 == nc, expected, res[c]);
        });
        for (auto &r : res)
            if (r.error_code) {
                hgx_set_error("%s", r.error.c_str());
                const int code = r.error_code;
                delete B;
                return code;
            }
        std::vector<size_t> roff(nc + 1, 0), poff(nc + 1, 0);
        for (size_t c = 0; c < nc; ++c) {
            roff[c + 1] = roff[c] + res[c].pair_ref.size();
            poff[c + 1] = poff[c] + res[c].pair_off.size() - 1;
            B->n_reads += res[c].n_reads;
        }
        B->pair_ref.resize(roff[nc]);
        B->pair_off.resize(poff[nc] + 1);
        B->pair_off[0] = 0;
        hgx_par_tasks(n_threads, nc, [&](int, size_t c) {
            if (!res[c].pair_ref.empty()) memcpy(&B->pair_ref[roff[c]], res[c].pair_ref.data(), res[c].pair_ref.size() * 4);
            for (size_t k = 1; k < res[c].pair_off.size(); ++k) B->pair_off[poff[c] + k] = (int32_t)(roff[c] + res[c].pair_off[k]);
        });
        for (auto &r : res)
            for (auto &t : r.trace) B->trace.push_back(std::move(t));
        lap("pair protocol");
        // tear-down side by side: the workers' arenas hold a heap block per haplotype (hundreds of thousands of frees), the
        // chunk results MBs each -- serially, at scope exit, that was ~5 ms of the call
        hgx_par_tasks(n_threads, workers.size() + res.size(), [&](int, size_t k) {
            if (k < workers.size()) { DecodeWorker dead; std::swap(dead, workers[k]); }
            else { ChunkOut dead; std::swap(dead, res[k - workers.size()]); }
        });
        hgx_par_ranges(outs.size() > 50000 ? n_threads : 1, outs.size(), [&](int, size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k) { std::string a, b; a.swap(outs[k].err); b.swap(outs[k].trace); }
        });
        lap("tear-down");
    } catch (const RefError &e) {
        hgx_set_error("the reference would fail on this input: %s", e.what());
        delete B;
        return HGX_EPARSE;
    } catch (const std::exception &e) {
        hgx_set_error("%s", e.what());
        delete B;
        return HGX_EINVAL;
    }
    *out = B;
    return HGX_OK;
}

extern "C" int hgx_batch_trace_text(const hgx_batch *b, char *buf, size_t cap, size_t *needed) {
    HARGCHK(b && needed);
    size_t n = 0;
    for (auto &t : b->trace) n += t.text.size() + 1;
    *needed = n;
    if (buf && cap > n) {
        char *p = buf;
        for (auto &t : b->trace) {
            memcpy(p, t.text.data(), t.text.size());
            p += t.text.size();
            *p++ = '\n';
        }
        *p = 0;
    }
    return HGX_OK;
}

extern "C" int hgx_batch_pileup(const hgx_batch *b, uint8_t *nt_set, uint32_t *counts) {
    HARGCHK(b);
    if (nt_set && !b->nt_set.empty()) memcpy(nt_set, b->nt_set.data(), b->nt_set.size());
    if (counts && !b->counts.empty()) memcpy(counts, b->counts.data(), b->counts.size() * 4);
    return HGX_OK;
}
