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
    std::vector<Novel> novel;
    std::unordered_map<uint64_t, int> novel_lookup;   // (type,pos,key) -> id
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
    Parser(const hgx_locus &l, const hgx_parse_opts &opts, hgx_batch &b, const hgx_batch &pile, const AltTables &at)
        : L(l), o(opts), B(b), alt_l(at.alt_l), alt_r(at.alt_r), alt_l_pos(at.alt_l_pos), alt_r_pos(at.alt_r_pos), PILE(pile) {}

    // ---- variant accessors over known + novel ids --------------------------------------------
    int vtype(int id) const { return id < L.V ? L.type[id] : novel[id - L.V].type; }
    int vpos(int id) const { return id < L.V ? L.pos[id] : novel[id - L.V].pos; }
    int vlen(int id) const { return id < L.V ? L.len[id] : novel[id - L.V].len; }
    bool is_hv(int id) const { return id >= 0 && id < L.V; }
    std::string vname(int id) const {
        if (id == -1) return "unknown";
        if (id < L.V) return L.name[id];
        return "nv" + std::to_string(id - L.V);
    }
    int vright(int id) const { return vtype(id) == HGX_VAR_DELETION ? vpos(id) + vlen(id) - 1 : vpos(id); }

    static uint64_t nkey(int type, int pos, int k) { return ((uint64_t)type << 60) | ((uint64_t)(uint32_t)pos << 24) | (uint32_t)(k & 0xffffff); }

    // first variant at `pos` (known list order, then novel) of the wanted type and size/base (core:949-961, 1005-1017, 1045-1057)
    int lookup(int pos, int type, int key) const {
        for (int j = lower_bound_pos(L.pos, pos); j < L.V && L.pos[j] == pos; ++j) {
            if (L.type[j] != type) continue;
            if (type == HGX_VAR_SINGLE ? L.base[j] == (char)key : L.len[j] == key) return j;
        }
        auto it = novel_lookup.find(nkey(type, pos, key));
        return it == novel_lookup.end() ? -1 : it->second;
    }
    int add_novel(int type, int pos, int key, const std::string &ins) {   // core:404-431
        if (lookup(pos, type, key) >= 0) throw RefError("assert: novel variant already present");
        Novel nv;
        nv.type = type; nv.pos = pos; nv.base = 0;
        nv.len = 1;
        if (type == HGX_VAR_SINGLE) nv.base = (char)key;
        else nv.len = key;
        nv.ins = ins;
        const int id = L.V + (int)novel.size();
        novel.push_back(nv);
        novel_lookup.emplace(nkey(type, pos, key), id);
        return id;
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

    // ---- pair flush (core:1238-1291): haplotypes -> piece refs --------------------------------------------
    // choose_pairs (core:680-716): keep the mate haplotype pairs whose inner distance is closest to the expected one
    static void choose_pairs(std::vector<Ht> &lh, std::vector<Ht> &rh, long expected) {
        if (lh.empty() || rh.empty() || std::max(lh.size(), rh.size()) < 2) return;
        long best = -1;
        std::vector<Ht> nl, nr;
        auto add = [](std::vector<Ht> &v, const Ht &h) { if (std::find(v.begin(), v.end(), h) == v.end()) v.push_back(h); };
        for (const Ht &l : lh)
            for (const Ht &r : rh) {
                const long inter = l.right < r.right ? (long)r.left - l.right - 1 : (long)l.left - r.right - 1;
                const long cur = std::labs(expected - inter);
                if (best < 0 || cur < best) { best = cur; nl.clear(); nr.clear(); }
                if (cur == best) { add(nl, l); add(nr, r); }
            }
        lh.swap(nl);
        rh.swap(nr);
    }

    // pair flush (core:1238-1291): the set union of the mates' haplotypes -> piece refs
    void flush(const std::vector<Ht> &lh, const std::vector<Ht> &rh) {
        std::vector<Ht> &hts = union_buf;
        hts.clear();
        for (const Ht &h : lh) hts.push_back(h);
        for (const Ht &h : rh) if (std::find(hts.begin(), hts.end(), h) == hts.end()) hts.push_back(h);
        std::vector<Ht> &ex = ex_buf;
        std::vector<int32_t> &ids = ids_buf;
        std::vector<uint32_t> &exon_refs = eref_buf, &gene_refs = gref_buf;
        exon_refs.clear();
        gene_refs.clear();
        auto intern = [&](const Ht &h) -> uint32_t {
            if (h.left > h.right) throw RefError("assert left <= right");
            ids.assign(h.ids.begin(), h.ids.end());
            for (auto &v : ids) if (v >= L.V) v = -1;
            const int64_t id = hgx_intern_piece(B, L, h.left, h.right, ids.data(), (int32_t)ids.size());
            if (id < 0) throw std::runtime_error(hgx_last_error());
            return (uint32_t)id;
        };
        for (const Ht &h : hts) {
            if (L.base_kind == HGX_BASE_HLA) {
                ex.clear();
                exon_pieces(h, ex);
                for (const Ht &e : ex) exon_refs.push_back(intern(e));
            }
            gene_refs.push_back(intern(h) | 0x80000000u);
        }
        if (exon_refs.size() > 255 || gene_refs.size() > 255) throw std::runtime_error("more than 255 pieces for one pair and level");
        B.pair_ref.insert(B.pair_ref.end(), exon_refs.begin(), exon_refs.end());
        B.pair_ref.insert(B.pair_ref.end(), gene_refs.begin(), gene_refs.end());
        B.pair_off.push_back((int32_t)B.pair_ref.size());
    }
};

struct Fields {
    const char *qname; size_t qname_len;
    int flag, pos;
    const char *cigar;
    const char *seq; size_t seq_len;
    const char *zs, *md;
    bool has_nm, has_nh, yt_cp;
    long nm, nh;
};

// split one line on whitespace in place (the buffer is a private copy); returns false for header/empty lines
static bool split_line(char *line, char *end, Fields &f) {
    char *cols[11];
    int nc = 0;
    char *p = line;
    f.zs = f.md = nullptr;
    f.has_nm = f.has_nh = f.yt_cp = false;
    f.nm = f.nh = 0;
    while (p < end) {
        while (p < end && (*p == '\t' || *p == ' ' || *p == '\r')) ++p;
        if (p >= end) break;
        char *tok = p;
        while (p < end && *p != '\t' && *p != ' ' && *p != '\r') ++p;
        if (p < end) *p++ = 0;
        if (nc < 11) cols[nc++] = tok;
        else {
            if (tok[0] == 'Z' && tok[1] == 's') f.zs = tok + 5;
            else if (tok[0] == 'M' && tok[1] == 'D') f.md = tok + 5;
            else if (tok[0] == 'N' && tok[1] == 'M') { f.has_nm = true; f.nm = strtol(tok + 5, nullptr, 10); }
            else if (tok[0] == 'N' && tok[1] == 'H') { f.has_nh = true; f.nh = strtol(tok + 5, nullptr, 10); }
            else if (tok[0] == 'Y' && tok[1] == 'T') f.yt_cp = strcmp(tok + 5, "CP") == 0;
        }
    }
    if (nc < 11) return false;
    f.qname = cols[0];
    f.qname_len = strlen(cols[0]);
    f.flag = (int)strtol(cols[1], nullptr, 10);
    f.pos = (int)strtol(cols[3], nullptr, 10);
    f.cigar = cols[5];
    f.seq = cols[9];
    f.seq_len = strlen(cols[9]);
    return true;
}

}   // namespace

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

struct ChunkResult {
    hgx_batch local;
    std::string error;
    int error_code = 0;
};

// The streaming loop (core:800-1587) over records [i0, i1).  Chunks start at read-id boundaries of the name-grouped
// stream, so pairs, the duplicate-mate filters and the flush protocol never straddle two chunks.
void process_chunk(const hgx_locus &L, const hgx_parse_opts &o, const AltTables &alts, const Fields *recs, const uint8_t *ok,
                   size_t i0, size_t i1, const hgx_batch &pile, bool is_last, long expected_interdist, ChunkResult &out) {
    hgx_batch &B = out.local;
    try {
        Parser P(L, o, B, pile, alts);
        // The stream is name-grouped, so the reference's global left/right/unpaired id sets (core:857-872) reduce to
        // three flags per group of equal read ids.
        const char *grp = nullptr;
        size_t grp_len = 0;
        bool g_l = false, g_r = false, g_u = false;
        std::vector<Ht> lhts, rhts;   // left / right positive haplotypes of the current pair (united at the flush, core:1250-1251)
        const char *prev_id = nullptr;
        size_t prev_len = 0;
        bool have_prev = false;
        std::vector<int> mid;
        std::vector<Cmp> cl, c2;
        std::vector<Parser::AltSide> lset, rset;
        std::string read;
        for (size_t i = i0; i < i1; ++i) {
            if (!ok[i]) continue;
            const Fields &f = recs[i];
            const size_t idlen = read_id_len(f, o.simulation != 0);
            if (!grp || grp_len != idlen || memcmp(grp, f.qname, idlen) != 0) {
                grp = f.qname;
                grp_len = idlen;
                g_l = g_r = g_u = false;
            }
            const int pos = f.pos - (o.base_locus + 1);
            if (pos < 0) continue;
            if (f.flag & 0x4) continue;
            if (!f.has_nm || !f.has_nh) throw RefError("TypeError: record without NM/NH tag (quirk Q8)");
            if (f.nm > o.num_editdist) continue;
            if (f.nh > 1) continue;
            if (!o.allow_discordant && !(f.flag & 0x2)) continue;
            const bool is_left = (f.flag & 0x40) != 0;
            if (is_left) {
                if (g_l) continue;
                g_l = true;
            } else if (f.flag & 0x80) {
                if (g_r) continue;
                g_r = true;
            } else {
                if (!o.allow_discordant) throw RefError("assert allow_discordant");
                if (g_u) continue;
                g_u = true;
            }
            read.assign(f.seq, f.seq_len);
            if (!P.decode(pos, f.cigar, read, f.zs, f.md, cl)) continue;
            B.n_reads++;
            if (!have_prev || prev_len != idlen || memcmp(prev_id, f.qname, idlen) != 0) {
                if (have_prev) P.flush(lhts, rhts);
                lhts.clear();
                rhts.clear();
            }
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
            int cleft, cright;
            P.ambiguous(c2, cleft, cright, lset, rset);
            mid.clear();
            for (int k = cleft; k <= cright; ++k)
                if (c2[k].type != T_MATCH) mid.push_back(c2[k].id);
            for (auto &l : lset)
                for (auto &r : rset) {
                    Ht h;
                    h.left = l.coord;
                    h.right = r.coord;
                    h.ids = l.ids;
                    h.ids.insert(h.ids.end(), mid.begin(), mid.end());
                    h.ids.insert(h.ids.end(), r.ids.begin(), r.ids.end());
                    std::vector<Ht> &dst = is_left ? lhts : rhts;
                    if (std::find(dst.begin(), dst.end(), h) == dst.end()) dst.push_back(std::move(h));
                }
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
                B.trace.push_back(TraceRec{t});
            }
            prev_id = f.qname;
            prev_len = idlen;
            have_prev = true;
        }
        if (have_prev) {
            if (is_last && o.codis_choose_pairs) Parser::choose_pairs(lhts, rhts, expected_interdist);   // core:1547-1552
            P.flush(lhts, rhts);
        }
    } catch (const RefError &e) {
        out.error = std::string("the reference would fail on this input: ") + e.what();
        out.error_code = HGX_EPARSE;
    } catch (const std::exception &e) {
        out.error = e.what();
        out.error_code = HGX_EINVAL;
    }
}

// get_pair_interdist (common:1187-1265): median inner distance of unique concordant pairs (CODIS D18S51 only)
long pair_interdist(const Fields *recs, const uint8_t *ok, size_t n_recs, bool simulation) {
    std::vector<long> dists;
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
    std::sort(dists.begin(), dists.end());
    return dists.empty() ? -1 : dists[dists.size() / 2];
}

// Chunk results -> one batch.  Every chunk interned its pieces into a private table; here the tables are united without a
// serial pass over them: (1) per chunk, its distinct pieces are bucketed by hash partition; (2) per partition, one worker
// interns the bucket entries of all chunks (in chunk order) into a partition-private table; (3) partition sizes are
// prefix-summed into global piece ids; (4) per chunk, refs are renumbered and copied to their place in the united arrays.
// The ids given here carry no meaning: hgx_finalize_batch orders the table by content, so the batch is the same whatever the
// number of workers and chunks.
void merge_chunks(hgx_batch &B, std::vector<ChunkResult> &res, int n_threads) {
    const size_t nc = res.size();
    if (nc == 1) {                              // one chunk: its table is the batch's
        hgx_batch &lb = res[0].local;
        B.pieces.swap(lb.pieces);
        B.masks.swap(lb.masks);
        B.pair_off.swap(lb.pair_off);
        B.pair_ref.swap(lb.pair_ref);
        B.n_reads = lb.n_reads;
        B.trace.swap(lb.trace);
        return;
    }
    const int P = std::max(1, std::min(n_threads, 64));
    struct Ent { uint32_t local; uint64_t hash; };
    std::vector<std::vector<std::vector<Ent>>> bucket(nc, std::vector<std::vector<Ent>>(P));     // [chunk][partition]
    hgx_par_tasks(n_threads, nc, [&](int, size_t c) {
        const hgx_batch &lb = res[c].local;
        for (size_t k = 0; k < lb.pieces.size(); ++k) {
            const hgx_piece &pc = lb.pieces[k];
            const uint64_t h = PieceTable::hash(pc.lo_word, pc.n_words, &lb.masks[pc.mask_off]);
            bucket[c][(h >> 40) % P].push_back(Ent{(uint32_t)k, h});
        }
    });
    std::vector<hgx_batch> part(P);                                        // partition-private distinct pieces
    std::vector<std::vector<uint32_t>> remap(nc);                          // [chunk][local id] -> (partition-local id, patched below)
    for (size_t c = 0; c < nc; ++c) remap[c].resize(res[c].local.pieces.size());
    hgx_par_tasks(n_threads, (size_t)P, [&](int, size_t p) {
        for (size_t c = 0; c < nc; ++c) {
            const hgx_batch &lb = res[c].local;
            for (const Ent &e : bucket[c][p]) {
                const hgx_piece &pc = lb.pieces[e.local];
                remap[c][e.local] = hgx_intern_masks(part[p], pc.lo_word, pc.n_words, &lb.masks[pc.mask_off]);
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
    std::vector<size_t> roff(nc + 1, 0), poff(nc + 1, 0);
    for (size_t c = 0; c < nc; ++c) {
        roff[c + 1] = roff[c] + res[c].local.pair_ref.size();
        poff[c + 1] = poff[c] + res[c].local.pair_off.size() - 1;
        B.n_reads += res[c].local.n_reads;
    }
    B.pair_ref.resize(roff[nc]);
    B.pair_off.resize(poff[nc] + 1);
    B.pair_off[0] = 0;
    hgx_par_tasks(n_threads, nc, [&](int, size_t c) {
        const hgx_batch &lb = res[c].local;
        // which partition a local piece went to: recomputed from the buckets (remap holds the partition-local id)
        std::vector<uint32_t> gid(lb.pieces.size());
        for (int p = 0; p < P; ++p)
            for (const Ent &e : bucket[c][p]) gid[e.local] = pbase[p] + remap[c][e.local];
        uint32_t *dst = &B.pair_ref[roff[c]];
        for (size_t k = 0; k < lb.pair_ref.size(); ++k) {
            const uint32_t ref = lb.pair_ref[k];
            dst[k] = (ref & 0x80000000u) | gid[ref & 0x7fffffffu];
        }
        for (size_t k = 1; k < lb.pair_off.size(); ++k) B.pair_off[poff[c] + k] = (int32_t)(roff[c] + lb.pair_off[k]);
    });
    for (auto &r : res)
        for (auto &t : r.local.trace) B.trace.push_back(std::move(t));
}

template <class F>
void parallel_for(int n_threads, size_t n, F fn) {   // fn(thread, begin, end), on the persistent worker pool
    hgx_par_ranges(n_threads, n, fn);
}

}   // namespace

static int parse_lines(hgx_batch **out, const hgx_locus *Lc, hgx_line *lines, size_t n, const hgx_parse_opts *opts);

// SAM text (name-grouped) -> private writable copy + line table -> parse_lines
extern "C" int hgx_parse_sam(hgx_batch **out, const hgx_locus *Lc, const char *sam, size_t n_bytes, const hgx_parse_opts *opts) {
    HARGCHK(out && Lc && (sam || n_bytes == 0) && opts);
    try {
        int n_threads = opts->n_threads > 0 ? opts->n_threads : (int)std::thread::hardware_concurrency();
        n_threads = std::max(1, std::min(n_threads, 512));
        struct PoolFree { void operator()(char *p) const { hgx_host_free(p); } };
        std::unique_ptr<char, PoolFree> text((char *)hgx_host_alloc(n_bytes + 1));      // tokens are NUL-terminated in place
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
        return parse_lines(out, Lc, lines.data(), lines.size(), opts);
    } catch (const std::exception &e) {
        hgx_set_error("%s", e.what());
        return HGX_EINVAL;
    }
}

extern "C" int hgx_parse_alignment_file(hgx_batch **out, const hgx_locus *Lc, const char *path, const char *regions,
                                        const hgx_parse_opts *opts) {
    HARGCHK(out && Lc && path && opts);
    const bool prof = getenv("HGX_PARSE_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    try {
        const double t0 = now();
        hgx_align_lines al;                     // the reader's buffers are tokenised in place (no copy, no trip through the caller)
        int rc = hgx_read_alignment_lines(path, regions, opts->n_threads, al);
        if (rc) return rc;
        const double t1 = now();
        rc = parse_lines(out, Lc, al.lines.data(), al.lines.size(), opts);
        if (prof) fprintf(stderr, "[hgx_parse_alignment_file] read %.1f ms, parse %.1f ms\n", (t1 - t0) * 1e3, (now() - t1) * 1e3);
        return rc;
    } catch (const std::exception &e) {
        hgx_set_error("hgx_parse_alignment_file: %s", e.what());
        return HGX_ENOMEM;
    }
}

// lines: name-grouped records; lines[i].p[lines[i].len] is writable (it becomes the record's terminator)
static int parse_lines(hgx_batch **out, const hgx_locus *Lc, hgx_line *lines, size_t n, const hgx_parse_opts *opts) {
    HARGCHK(out && Lc && (lines || n == 0) && opts);
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
        int n_threads = opts->n_threads > 0 ? opts->n_threads : (int)std::thread::hardware_concurrency();
        n_threads = std::max(1, std::min(n_threads, 512));
        if (opts->keep_trace) n_threads = 1;                // traces (and novel-variant numbering) follow stream order
        if (n < 20000) n_threads = 1;
        // per-record tables: written by the split below, never read before (no zero fill of ~100 bytes per record)
        struct PoolFree { void operator()(void *p) const { hgx_host_free(p); } };
        std::unique_ptr<void, PoolFree> recs_mem(hgx_host_alloc(std::max<size_t>(n, 1) * sizeof(Fields)));
        std::unique_ptr<void, PoolFree> ok_mem(hgx_host_alloc(std::max<size_t>(n, 1)));
        Fields *recs = (Fields *)recs_mem.get();
        uint8_t *ok = (uint8_t *)ok_mem.get();
        // field split + pass 1 (pileup over all records, common:1076-1134), both embarrassingly parallel over lines
        std::vector<std::vector<uint32_t>> tcounts(n_threads);
        std::vector<std::string> terr(n_threads);
        parallel_for(n_threads, n, [&](int t, size_t b, size_t e) {
            std::vector<uint32_t> &cnt = tcounts[t];
            cnt.assign((size_t)n_ref * 6, 0u);
            for (size_t i = b; i < e; ++i) {
                char *line = lines[i].p, *lend = line + lines[i].len;
                *lend = 0;
                ok[i] = split_line(line, lend, recs[i]) ? 1 : 0;
                if (!ok[i]) continue;
                const Fields &f = recs[i];
                if (f.flag & 0x4) continue;
                const int pos = f.pos - (opts->base_locus + 1);
                if (pos < 0) continue;
                if (!opts->allow_discordant && !(f.flag & 0x2)) continue;
                int rp = 0, gp = pos;
                for (const char *p = f.cigar; *p;) {
                    char *q;
                    const long len = strtol(p, &q, 10);
                    if (q == p || !*q) break;
                    const char op = *q;
                    if (op == 'M' || op == 'D') {
                        for (long j = 0; j < len; ++j) {
                            if (gp + j >= n_ref) break;
                            int slot = 5;
                            if (op == 'M') {
                                if ((size_t)(rp + j) >= f.seq_len) { terr[t] = "IndexError: read shorter than CIGAR"; break; }
                                const char c = f.seq[rp + j];
                                slot = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
                            }
                            cnt[(size_t)(gp + j) * 6 + slot]++;
                        }
                    }
                    if (op == 'M' || op == 'N' || op == 'D') gp += (int)len;
                    if (op == 'M' || op == 'I' || op == 'S') rp += (int)len;
                    p = q + 1;
                }
            }
        });
        lap("split + pileup");
        for (auto &e : terr) if (!e.empty()) throw RefError(e);
        B->counts.assign((size_t)n_ref * 6, 0u);
        B->nt_set.assign(n_ref, 0);
        {   // partial pileups summed in worker (= stream) order, columns split among the workers
            const size_t cells = B->counts.size();
            parallel_for(n_threads, cells, [&](int, size_t b, size_t e) {
                for (auto &cnt : tcounts)
                    if (!cnt.empty())
                        for (size_t k = b; k < e; ++k) B->counts[k] += cnt[k];
            });
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
        const long expected = opts->codis_choose_pairs ? pair_interdist(recs, ok, n, opts->simulation != 0) : -1;
        // pass 2: chunks that start where the read id changes
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
        std::vector<ChunkResult> res(nc);
        lap("pileup merge + cuts");
        hgx_par_tasks(n_threads, nc, [&](int, size_t c) {
            process_chunk(L, *opts, alts, recs, ok, cut[c], cut[c + 1], *B, c + 1 == nc, expected, res[c]);
        });
        for (auto &r : res)
            if (r.error_code) {
                hgx_set_error("%s", r.error.c_str());
                const int code = r.error_code;
                delete B;
                return code;
            }
        lap("streaming loop");
        merge_chunks(*B, res, n_threads);
        lap("merge");
        hgx_finalize_batch(*B, n_threads);
        lap("finalize");
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
