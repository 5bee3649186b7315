"""pyref -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Pure-Python (dict / set / str) restatement of the reference's per-locus typing body, used only
as a checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
hisat-genotype_amd/ imports it.

It follows the reference's data model on purpose (variant ids are the strings "hv<n>" / "nv<k>" /
"unknown", haplotypes are '-'-joined strings, classes are '-'.join(sorted(names))) so that every
intermediate can be compared literally with the golden vectors recorded from the real reference
(tests/golden/NAME.json.gz).  Each function cites the reference lines it restates:

  lower_bound            hisatgenotype_typing_common.py:406-422
  rep_alleles            hisatgenotype_typing_core.py:67-115
  pileup                 hisatgenotype_typing_common.py:1059-1134 (annotation part 1137-1183 feeds only the assembly graph)
  Alternatives           hisatgenotype_typing_common.py:1424-1657
  ambiguous_diffs        hisatgenotype_typing_common.py:1663-1955 (+ validation_check.py:313-341)
  error_correct          hisatgenotype_typing_core.py:119-243
  RefLocus.decode        hisatgenotype_typing_core.py:800-1164
  RefLocus.exon_pieces   hisatgenotype_typing_core.py:718-792
  RefLocus.add_count     hisatgenotype_typing_core.py:626-677
  RefLocus.add_stat      hisatgenotype_typing_core.py:1171-1236
  RefLocus.run           hisatgenotype_typing_core.py:384-401, 476-491, 559-596, 1238-1406, 1545-1593, 1650-1789, 2076-2121
  single_abundance       hisatgenotype_typing_common.py:1272-1410

Parity status: PINNED by tests/test_pyref_golden.py against the golden vectors.
"""
import math
import re

_CIGAR_RE = re.compile(r"\d+\w")


class ReferenceError_(Exception):
    """The reference would assert / exit(1) / raise on this input."""


def lower_bound(lst, pos):
    low, high = 0, len(lst)
    while low < high:
        m = (low + high) // 2
        if lst[m][0] < pos:
            low = m + 1
        elif lst[m][0] > pos:
            high = m
        else:
            while m > 0 and lst[m - 1][0] >= pos:
                m -= 1
            return m
    return low


def var_right(var):
    t, p, d = var
    return p + int(d) - 1 if t == "deletion" else p


def exonic_vars(Vars, exons):
    out = set()
    for vid, var in Vars.items():
        l, r = var[1], var_right(var)
        for el, er in exons:
            if l >= el and r <= er:
                out.add(vid)
    return out


def rep_alleles(Links, exon_vars, in_alleles=None):
    """Group alleles by identical exonic variant set; representative = first member met."""
    per_allele = {}
    for vid, alleles in Links.items():
        if vid not in exon_vars:
            continue
        for a in alleles:
            if in_alleles is not None and a not in in_alleles:
                continue
            per_allele.setdefault(a, set()).add(vid)
    groups = {}
    for a, vs in per_allele.items():
        groups.setdefault(frozenset(vs), []).append(a)
    reps, rep_groups = {}, {}
    for members in groups.values():
        rep_groups[members[0]] = members
        for m in members:
            reps[m] = members[0]
    return reps, rep_groups


def pileup(records, ref_len, allow_discordant):
    """records: iterable of (flag, pos0, cigar_str, seq).  Returns (counts[L] dicts, nt_set[L] lists)."""
    counts = pileup_counts(records, ref_len, allow_discordant)
    return counts, nt_sets_from_counts(counts)


def pileup_counts(records, ref_len, allow_discordant):
    """The counting half of get_mpileup (common:1059-1122).  Counts are sums over records, so the tables of the shards of a
    stream add up position by position (tests/oracle_util.py runs the oracle over a big sample shard by shard on the host cores)."""
    counts = [dict() for _ in range(ref_len)]
    for flag, pos, cigar_str, seq in records:
        if flag & 0x4:
            continue
        if pos < 0:
            continue
        if not allow_discordant and not (flag & 0x2):
            continue
        rp, gp = 0, pos
        for c in _CIGAR_RE.findall(cigar_str):
            op, n = c[-1], int(c[:-1])
            if op in "MD":
                for j in range(n):
                    nt = seq[rp + j] if op == "M" else "D"
                    if gp + j < ref_len:
                        counts[gp + j][nt] = counts[gp + j].get(nt, 0) + 1
            if op in "MND":
                gp += n
            if op in "MIS":
                rp += n
    return counts


def nt_sets_from_counts(counts):
    """The 20 % / >= 7 rule per position (common:1124-1134)."""
    nt_sets = []
    for d in counts:
        tot = sum(d.values())
        s = []
        if tot >= 20:
            for nt, c in d.items():
                if nt in "ACGT" and (c >= tot * 0.2 or c >= 7):
                    s.append(nt)
        nt_sets.append(s)
    return nt_sets


class Alternatives:
    """Equivalent spellings around deletions (left / right tables of haplotype strings)."""

    def __init__(self, ref_seq, allele_vars, Vars, Var_list):
        self.ref, self.Vars, self.Var_list = ref_seq, Vars, Var_list
        self.second = set()
        for vs in allele_vars.values():
            for i in range(len(vs) - 1):
                self.second.add(vs[i] + "-" + vs[i + 1])
        rev = []
        for _, vid in Var_list:
            t, p, d = Vars[vid]
            if t == "deletion":
                p = p + int(d) - 1
            elif t == "insertion":
                p += 1
            rev.append([p, vid])
        self.rev = sorted(rev, key=lambda x: x[0])
        self.left, self.right = {}, {}
        for _, vid in Var_list:
            t, p, d = Vars[vid]
            if p == 0 or t != "deletion":
                continue
            n = int(d)
            if p + n >= len(ref_seq):
                continue
            self._recur(vid, [p, vid, p + n - 1], [p + n, p + n - 1], True, 0)
            self._recur(vid, [p, vid, p + n - 1], [p, p - 1], False, 0)

    def _next(self, ht, left, exclude=()):
        ref, Vars = self.ref, self.Vars
        pos = int(ht[0]) - 1 if left else ht[-1] + 1
        if pos < 0 or pos >= len(ref):
            return []
        if left:
            out = [[[pos] + ht[1:], ref[pos]]]
            prev = ht[1] if len(ht) > 2 else None
            hi = lower_bound(self.rev, pos + 1)
            for j in range(hi - 1, -1, -1):
                vid = self.rev[j][1]
                t, p, d = Vars[vid]
                if t == "deletion":
                    if p == 0:
                        continue
                    p = p + int(d) - 1
                if p > pos:
                    continue
                if p < pos:
                    break
                if vid in exclude:
                    continue
                if prev and (vid + "-" + prev) not in self.second:
                    continue
                if t == "single":
                    out.append([[p, vid] + ht[1:], d])
                elif t == "deletion":
                    out += self._next([p - int(d) + 1, vid] + ht[1:], left, exclude)
        else:
            out = [[ht[:-1] + [pos], ref[pos]]]
            prev = ht[-2] if len(ht) > 2 else None
            for j in range(lower_bound(self.Var_list, pos), len(self.Var_list)):
                vid = self.Var_list[j][1]
                t, p, d = Vars[vid]
                if p < pos:
                    continue
                if p > pos:
                    break
                if vid in exclude:
                    continue
                if prev and (prev + "-" + vid) not in self.second:
                    continue
                if t == "single":
                    out.append([ht[:-1] + [vid, p], d])
                elif t == "deletion":
                    out += self._next(ht[:-1] + [vid, p + int(d) - 1], left, exclude)
        return out

    @staticmethod
    def _spell(ht):
        if len(ht) <= 2:
            return "%d-%d" % (ht[0], ht[1])
        return "%d-%s-%d" % (ht[0], "-".join(ht[1:-1]), ht[-1])

    def _recur(self, orig, ht, alt, left, dep):
        b1 = self._next(ht, left)
        b2 = self._next(alt, left, (orig,))
        found = False
        for nh, bp in b1:
            for na, bp2 in b2:
                if bp != bp2:
                    continue
                if left:
                    if int(nh[0]) == int(na[0]):
                        continue
                elif int(nh[-1]) == int(na[-1]):
                    continue
                found = True
                self._recur(orig, nh, na, left, dep + 1)
        if dep > 0 and not found:
            table = self.left if left else self.right
            s, a = self._spell(ht), self._spell(alt)
            table.setdefault(s, set()).add(a)
            table.setdefault(a, set()).add(s)

    def sorted_list(self, left):
        table = self.left if left else self.right
        lst = [[int(k.split("-")[-1 if left else 0]), k] for k in table.keys()]
        return sorted(lst, key=lambda x: x[0])


def _ht_and_len(ref_seq, cmp_list):
    ht, n = [], 0
    for c in cmp_list:
        if c[0] == "match":
            n += len(ref_seq[c[1]:c[1] + c[2]])
        elif c[0] == "mismatch":
            n += 1
        vid = c[3] if len(c) > 3 else ""
        if vid != "" and vid != "unknown":
            ht.append(vid)
    return ht, n


def ambiguous_diffs(ref_seq, Vars, alts_l, alts_r, list_l, list_r, cmp_list):
    cmp_left, cmp_right = 0, len(cmp_list) - 1
    left = cmp_list[0][1]
    right = cmp_list[-1][1] + cmp_list[-1][2] - 1
    lset, rset = set(), set()

    found = False
    for i in range(len(cmp_list) - 1, -1, -1):
        t, cur_left, length = cmp_list[i][:3]
        vid = cmp_list[i][3] if t in ("mismatch", "deletion") else ""
        if t in ("mismatch", "deletion", "insertion") and not vid.startswith("hv"):
            continue
        cur_right = cur_left + length - 1 if t in ("match", "deletion") else cur_left
        cur_ht, seqlen = _ht_and_len(ref_seq, cmp_list[:i + 1])
        cur_str = str(left) if not cur_ht else "%d-%s" % (left, "-".join(cur_ht))
        i_found = False
        hi = lower_bound(list_l, cur_right + 1)
        for j in range(min(hi + 1, len(list_l)) - 1, -1, -1):
            hp, key = list_l[j]
            if hp < cur_left:
                break
            if hp > cur_right:
                continue
            if cur_ht and key.find("-".join(cur_ht)) == -1:
                continue
            f = key.split("-")[:-1]
            if len(cur_ht) + 1 == len(f):
                if left < int(f[0]):
                    continue
            else:
                v2 = f[len(f) - len(cur_ht) - 1]
                if v2 not in Vars:
                    raise ReferenceError_("KeyError %s" % v2)
                if left <= var_right(Vars[v2]):
                    continue
            i_found = True
            for alt_str in alts_l[key]:
                af = alt_str.split("-")
                a_right = int(af[-1])
                if a_right > cur_right:
                    raise ReferenceError_("assert alt_ht_right <= cur_right")
                seq_pos, cur_pos, part = cur_right - a_right, a_right, []
                for v in reversed(af[1:-1]):
                    vt, vp, vd = Vars[v]
                    if vt == "deletion":
                        vp = vp + int(vd) - 1
                    if vp > cur_pos:
                        raise ReferenceError_("assert var_pos_ <= cur_pos")
                    nsp = seq_pos + (cur_pos - vp)
                    if nsp >= seqlen:
                        break
                    if vt == "single":
                        nsp += 1
                        ncp = vp - 1
                    elif vt == "deletion":
                        ncp = vp - int(vd)
                    else:
                        raise ReferenceError_("insertion in alternative")
                    part.insert(0, v)
                    if nsp >= seqlen:
                        break
                    seq_pos, cur_pos = nsp, ncp
                if part:
                    seq_left = seqlen - seq_pos - 1
                    tail = ""
                    if found:
                        ids = [c[3] for c in cmp_list[i + 1:cmp_left]
                               if c[0] in ("mismatch", "deletion", "insertion") and c[3].startswith("hv")]
                        if ids:
                            tail = "-" + "-".join(ids)
                    lset.add("%d-%s" % (cur_pos - seq_left, "-".join(part)) + tail)
        if i_found:
            if not found:
                cmp_left = i + 1
                lset.add(cur_str)
            found = True
    if not found:
        lset.add(str(left))

    found = False
    for i in range(len(cmp_list)):
        t, cur_left, length = cmp_list[i][:3]
        vid = cmp_list[i][3] if t in ("mismatch", "deletion") else ""
        if t in ("mismatch", "deletion", "insertion") and not vid.startswith("hv"):
            continue
        cur_right = cur_left + length - 1 if t in ("match", "deletion") else cur_left
        cur_ht, seqlen = _ht_and_len(ref_seq, cmp_list[i:])
        cur_str = str(right) if not cur_ht else "%s-%d" % ("-".join(cur_ht), right)
        i_found = False
        for j in range(lower_bound(list_r, cur_left), len(list_r)):
            hp, key = list_r[j]
            if hp > cur_right:
                break
            if hp < cur_left:
                continue
            if cur_ht and key.find("-".join(cur_ht)) == -1:
                continue
            f = key.split("-")[1:]
            if len(cur_ht) + 1 == len(f):
                if right > int(f[-1]):
                    continue
            else:
                v2 = f[len(cur_ht)] if len(cur_ht) < len(f) else None
                if v2 is None:
                    raise ReferenceError_("IndexError")
                if v2 not in Vars:
                    raise ReferenceError_("KeyError %s" % v2)
                if right >= Vars[v2][1]:
                    continue
            i_found = True
            for alt_str in alts_r[key]:
                af = alt_str.split("-")
                a_left = int(af[0])
                if cur_left > a_left:
                    raise ReferenceError_("assert cur_left <= alt_ht_left")
                seq_pos, cur_pos, part = a_left - cur_left, a_left, []
                for v in af[1:-1]:
                    vt, vp, vd = Vars[v]
                    if vp < cur_pos:
                        raise ReferenceError_("assert var_pos_ >= cur_pos")
                    nsp = seq_pos + (vp - cur_pos)
                    if nsp >= seqlen:
                        break
                    if vt == "single":
                        nsp += 1
                        ncp = vp + 1
                    elif vt == "deletion":
                        ncp = vp + int(vd)
                    else:
                        raise ReferenceError_("insertion in alternative")
                    part.append(v)
                    if nsp >= seqlen:
                        break
                    seq_pos, cur_pos = nsp, ncp
                if part:
                    seq_left = seqlen - seq_pos - 1
                    if seq_left < 0:
                        raise ReferenceError_("assert seq_left >= 0")
                    head = ""
                    if found:
                        ids = [c[3] for c in cmp_list[cmp_right + 1:i]
                               if c[0] in ("mismatch", "deletion", "insertion") and c[3].startswith("hv")]
                        if ids:
                            head = "-".join(ids) + "-"
                    rset.add(head + "%s-%d" % ("-".join(part), cur_pos + seq_left))
        if i_found:
            if not found:
                cmp_right = i - 1
                rset.add(cur_str)
            found = True
    if not found:
        rset.add(str(right))

    if cmp_right < cmp_left:
        cmp_left = 0
        lset = {str(left)}

    # always-on sanity check (devel/settings.json holds the truthy string "False", quirk Q1)
    seen = set()
    for h in lset:
        h = "-".join(h.split("-")[1:])
        if h == "":
            continue
        if h in seen:
            raise ReferenceError_("check_amb_uniqueness")
        seen.add(h)
    for h in rset:
        h = "-".join(h.split("-")[:-1])
        if h == "":
            continue
        if h in seen:
            raise ReferenceError_("check_amb_uniqueness")
        seen.add(h)
    return cmp_left, cmp_right, sorted(lset), sorted(rset)


def _known_single(Vars, Var_list, pos, base):
    j = lower_bound(Var_list, pos)
    while j < len(Var_list):
        p, vid = Var_list[j]
        if p > pos:
            break
        if p == pos:
            t, _, d = Vars[vid]
            if t == "single" and d == base:
                return vid
        j += 1
    return None


def error_correct(ref_seq, read_seq, read_pos, nt_sets, Vars, Var_list, cmp_list):
    ncorr = 0
    i = 0
    while i < len(cmp_list):
        t, left, length = cmp_list[i][:3]
        if left >= len(ref_seq):
            break
        if t == "match":
            mid, last = [], 0
            for j in range(length):
                if read_pos + j >= len(read_seq) or left + j >= len(ref_seq):
                    continue
                b = read_seq[read_pos + j]
                s = nt_sets[left + j]
                if len(s) > 0 and b not in s:
                    b = "N" if len(s) > 1 else s[0]
                    read_seq = read_seq[:read_pos + j] + b + read_seq[read_pos + j + 1:]
                    if b == ref_seq[left + j]:
                        raise ReferenceError_("assert read_bp != ref_bp")
                    new = ["mismatch", left + j, 1, "unknown"]
                    ncorr += 1
                    if b != "N":
                        k = _known_single(Vars, Var_list, left + j, b)
                        if k is not None:
                            new[3] = k
                    if j > last:
                        mid.append(["match", left + last, j - last])
                    mid.append(new)
                    last = j + 1
            if last < length:
                mid.append(["match", left + last, length - last])
            cmp_list = cmp_list[:i] + mid + cmp_list[i + 1:]
            i += len(mid) - 1
        else:
            b = read_seq[read_pos]
            s = nt_sets[left]
            if len(s) > 0 and b not in s:
                b = "N" if len(s) > 1 else s[0]
                read_seq = read_seq[:read_pos] + b + read_seq[read_pos + 1:]
                if b == "N":
                    cmp_list[i][3] = "unknown"
                elif b == ref_seq[left]:
                    cmp_list[i] = ["match", left, 1]
                    ncorr += 1
                else:
                    k = _known_single(Vars, Var_list, left, b)
                    cmp_list[i][3] = k if k is not None else "unknown"
        read_pos += length
        i += 1
    out = []
    for c in cmp_list:
        if c[0] == "match" and out and out[-1][0] == "match":
            out[-1] = ["match", out[-1][1], out[-1][2] + c[2]]
        else:
            out.append(c)
    return out, read_seq, ncorr


def single_abundance(Gene_cmpt, remove_low=False, Gene_length=None, stats=None):
    """dict[str,int] -> [[allele, prob]] sorted descending (stable); FP64; SQUAREM EM."""
    Gene_length = Gene_length or {}

    def norm(p):
        if Gene_length:
            tot = 0
            for a, m in p.items():
                tot += m / Gene_length[a]
            for a, m in p.items():
                p[a] = m / Gene_length[a] / tot
        else:
            tot = sum(p.values())
            for a, m in p.items():
                p[a] = m / tot

    split = [(k.split("-"), float(c)) for k, c in Gene_cmpt.items()]
    prob = {}
    for names, c in split:
        for a in names:
            prob[a] = prob.get(a, 0.0) + c / len(names)
    norm(prob)

    def step(p):
        q = {}
        for names, c in split:
            s = 0.0
            for a in names:
                if a in p:
                    s += p[a]
            if s <= 0.0:
                continue
            for a in names:
                if a in p:
                    q[a] = q.get(a, 0.0) + c * p[a] / s
        norm(q)
        return q

    def prune(p):
        if not p:
            return p
        mx = max(p.values())
        return {a: v for a, v in p.items() if v >= mx / 10.0}

    diff, it = 1.0, 0
    while diff > 0.0001 and it < 1000:
        p1 = step(prob)
        p2 = step(p1)
        sr = sv = 0.0
        r, v = {}, {}
        for a in prob:
            r[a] = p1[a] - prob[a]
            sr += r[a] * r[a]
            v[a] = p2[a] - p1[a] - r[a]
            sv += v[a] * v[a]
        if sv > 0.0:
            g = -math.sqrt(sr / sv)
            for a in prob:
                p2[a] = max(0.0, prob[a] - 2 * g * r[a] + g * g * v[a])
            p1 = step(p2)
        diff = 0.0
        for a in prob:
            diff += abs(prob[a] - p1[a]) if a in p1 else prob[a]
        prob = p1
        if it >= 10 and remove_low:
            prob = prune(prob)
        it += 1
    if remove_low:
        prob = prune(prob)
    norm(prob)
    if stats is not None:
        stats["n_iter"] = it
    return sorted([[a, p] for a, p in prob.items()], key=lambda x: x[1], reverse=True)


def pair_interdist(lines, simulation):
    """Median inner distance of uniquely, concordantly aligned mates (typing_common.py:1187-1265)."""
    dists, prev, reads = [], None, []
    for l in lines:
        c = l.split()
        read_id, flag, pos, cigar_str = c[0], int(c[1]), int(c[3]), c[5]
        if flag & 0x4:
            continue
        if simulation:
            read_id = read_id.split("|")[0]
        NH, YT = None, ""
        for col in c[11:]:
            if col.startswith("NH"):
                NH = int(col[5:])
            elif col.startswith("YT"):
                YT = col[5:]
        if NH is None or NH > 1 or YT != "CP":
            continue
        if prev is not None and read_id != prev:
            if len(reads) == 2:
                (l1, r1), (l2, r2) = reads
                dists.append(l2 - r1 - 1 if l1 <= l2 else l1 - r2 - 1)
            reads = []
        right = pos
        for cg in _CIGAR_RE.findall(cigar_str):
            if cg[-1] in "MND":
                right += int(cg[:-1])
        reads.append([pos, right - 1])
        prev = read_id
    dists.sort()
    return dists[len(dists) // 2] if dists else -1


def choose_pairs(lhts, rhts, expected):
    """Keep the mate haplotype pairs whose inner distance is closest to the expected one (typing_core.py:680-716)."""
    if len(lhts) > 0 and len(rhts) > 0 and max(len(lhts), len(rhts)) >= 2:
        best, picked = None, []
        for lh in lhts:
            f = lh.split("-")
            ll, lr = int(f[0]), int(f[-1])
            for rh in rhts:
                g = rh.split("-")
                rl, rr = int(g[0]), int(g[-1])
                inter = rl - lr - 1 if lr < rr else ll - rr - 1
                cur = abs(expected - inter)
                if best is None or best > cur:
                    best, picked = cur, [[lh, rh]]
                elif best == cur:
                    picked.append([lh, rh])
        lhts = {a for a, _ in picked}
        rhts = {b for _, b in picked}
    return lhts, rhts


class RefLocus:
    """Per-locus state + the streaming loop of typing() for one locus."""

    def __init__(self, locus, num_editdist=2, error_correction=True, allow_discordant=False,
                 remove_low=True, simulation=False):
        d = locus.reference_dicts()
        g = locus.gene
        self.gene, self.base = g, locus.base_fname
        self.ref_allele = locus.ref_allele
        self.ref_seq = locus.backbone
        self.exons = [list(e) for e in locus.exons]
        self.Genes = d["Genes"][g]
        self.Gene_names = d["Gene_names"][g]
        self.Gene_lengths = d["Gene_lengths"][g]
        self.Links = d["Links"]
        self.Vars = {k: list(v) for k, v in d["Vars"][g].items()}
        self.Var_list = [list(x) for x in d["Var_list"][g]]
        self.opts = dict(num_editdist=num_editdist, error_correction=error_correction,
                         allow_discordant=allow_discordant, remove_low=remove_low, simulation=simulation)
        self.maxright = {}
        cur = -1
        for _, vid in self.Var_list:
            cur = max(cur, var_right(self.Vars[vid]))
            self.maxright[vid] = cur
        self.allele_vars = {}
        for _, vid in self.Var_list:
            for a in self.Links.get(vid, []):
                if a in self.Genes:
                    self.allele_vars.setdefault(a, []).append(vid)
        ev = exonic_vars(self.Vars, self.exons)
        self.allele_reps, self.rep_groups = rep_alleles(self.Links, ev)
        self.rep_set = set(self.allele_reps.values())
        self.alts = Alternatives(self.ref_seq, self.allele_vars, self.Vars, self.Var_list)
        self.list_l = self.alts.sorted_list(True)
        self.list_r = self.alts.sorted_list(False)
        self.novel = 0
        self.trace = None          # optional list collecting per-record intermediates
        self.score = True          # False: front-end only (collect the pairs' haplotypes, skip add_count/EM)
        # Optional: {} = remember the outcome of decode + cmp_list2 + ambiguous_diffs per distinct (pos, CIGAR, SEQ, Zs, MD).  The
        # outcome is a function of exactly those and of the pileup (novel variants get the id of their first creation either way),
        # and 70 % of a deep sample's records repeat an earlier key: a 1 M-read oracle run in minutes instead of tens of minutes.
        # Off by default; tests/test_pyref_golden.py checks that it changes nothing.
        self.memo = None

    # -- novel variants (core:404-431) ------------------------------------------------------
    def _add_novel(self, vtype, pos, data):
        j = lower_bound(self.Var_list, pos)
        while j < len(self.Var_list):
            p, vid = self.Var_list[j]
            if p > pos:
                break
            if p == pos:
                t, _, d = self.Vars[vid]
                if t == vtype and d == data:
                    raise ReferenceError_("assert novel variant is new")
                if t != vtype:
                    if vtype == "insertion" or (vtype == "single" and t == "deletion"):
                        break
                elif data < d:
                    break
            j += 1
        vid = "nv%d" % self.novel
        self.novel += 1
        self.Vars[vid] = [vtype, pos, data]
        self.Var_list.insert(j, [pos, vid])
        return vid

    def _lookup(self, pos, pred):
        j = lower_bound(self.Var_list, pos)
        while j < len(self.Var_list):
            p, vid = self.Var_list[j]
            if p > pos:
                break
            if p == pos and pred(self.Vars[vid]):
                return vid
            j += 1
        return "unknown"

    # -- one SAM record -> cmp_list (core:876-1164) ---------------------------------------------
    def decode(self, pos, cigar_str, read_seq, Zs, MD, counts, nt_sets):
        o = self.opts
        zs = []
        if Zs:
            for item in Zs.split(","):
                f = item.split("|")
                zs.append([int(f[0]), f[1], f[2]])
        if MD == "":
            raise ReferenceError_("assert MD != ''")
        md_i = md_len = 0
        zs_i = 0
        zs_pos = zs[0][0] if zs else 0
        rp, gp = 0, pos
        cigars = [[c[-1], int(c[:-1])] for c in _CIGAR_RE.findall(cigar_str)]
        cmp_list, n_ec, bad = [], 0, False
        clip = [0, 0]
        for ci, (op, n) in enumerate(cigars):
            if op == "M":
                first, used, start = True, 0, len(cmp_list)
                while True:
                    if not first or md_len == 0:
                        if MD[md_i].isdigit():
                            num = 0
                            while md_i < len(MD) and MD[md_i].isdigit():
                                num = num * 10 + int(MD[md_i])
                                md_i += 1
                            md_len += num
                    if md_len >= n:
                        md_len -= n
                        if n > used:
                            cmp_list.append(["match", gp + used, n - used])
                        break
                    first = False
                    base = read_seq[rp + md_len]
                    if MD[md_i] not in "ACGT":
                        raise ReferenceError_("assert MD_ref_base in ACGT")
                    md_i += 1
                    if md_len > used:
                        cmp_list.append(["match", gp + used, md_len - used])
                    if rp + md_len == zs_pos and zs_i < len(zs):
                        if zs[zs_i][1] != "S":
                            raise ReferenceError_("assert Zs type S")
                        vid = zs[zs_i][2]
                        zs_i += 1
                        zs_pos += 1
                        if zs_i < len(zs):
                            zs_pos += zs[zs_i][0]
                    else:
                        vid = self._lookup(gp + md_len, lambda v: v[0] == "single" and v[2] == base)
                    cmp_list.append(["mismatch", gp + md_len, 1, vid])
                    used = md_len + 1
                    md_len += 1
                    if md_len == n:
                        md_len = 0
                        break
                if o["error_correction"]:
                    new, read_seq, k = error_correct(self.ref_seq, read_seq, rp, nt_sets, self.Vars,
                                                     self.Var_list, cmp_list[start:])
                    cmp_list = cmp_list[:start] + new
                    n_ec += k
            elif op == "I":
                if rp == zs_pos and zs_i < len(zs):
                    if zs[zs_i][1] != "I":
                        raise ReferenceError_("assert Zs type I")
                    vid = zs[zs_i][2]
                    zs_i += 1
                    if zs_i < len(zs):
                        zs_pos += zs[zs_i][0]
                else:
                    vid = self._lookup(gp, lambda v: v[0] == "insertion" and len(v[2]) == n)
                cmp_list.append(["insertion", gp, n, vid])
                if "N" in read_seq[rp:rp + n]:
                    bad = True
            elif op == "D":
                if MD[md_i] == "0":
                    md_i += 1
                if MD[md_i] != "^":
                    raise ReferenceError_("assert MD ^")
                md_i += 1
                while md_i < len(MD) and MD[md_i] in "ACGT":
                    md_i += 1
                if rp == zs_pos and zs_i < len(zs) and zs[zs_i][1] == "D":
                    vid = zs[zs_i][2]
                    zs_i += 1
                    if zs_i < len(zs):
                        zs_pos += zs[zs_i][0]
                else:
                    vid = self._lookup(gp, lambda v: v[0] == "deletion" and int(v[2]) == n)
                cmp_list.append(["deletion", gp, n, vid])
                if gp < len(counts):
                    dc = nc = 0
                    for nt, c in counts[gp].items():
                        if nt == "D":
                            dc += c
                        else:
                            nc += c
                    if self.base == "hla" and dc * 6 < nc:
                        bad = True
            elif op == "S":
                if ci == 0:
                    clip[0] = n
                    zs_pos += n
                else:
                    if ci + 1 != len(cigars):
                        raise ReferenceError_("assert softclip at end")
                    clip[1] = n
            else:
                raise ReferenceError_("assert cigar op")
            if op in "MND":
                gp += n
            if op in "MIS":
                rp += n
        if clip[0] > 0:
            read_seq = read_seq[clip[0]:]
        if clip[1] > 0:
            read_seq = read_seq[:-clip[1]]
        if gp > len(self.ref_seq):
            return None
        if n_ec > max(1, o["num_editdist"]):
            return None
        if bad:
            return None
        # novel variants (core:1126-1164)
        rp = 0
        for c in cmp_list:
            t, p, n = c[:3]
            if t != "match" and c[3] == "unknown":
                add = True
                if t == "mismatch":
                    data = read_seq[rp]
                    if data == "N":
                        add = False
                elif t == "deletion":
                    data = str(n)
                else:
                    data = read_seq[rp:rp + n]
                if add:
                    c[3] = self._add_novel("single" if t == "mismatch" else t, p, data)
            if t != "deletion":
                rp += n
        return cmp_list, gp

    # -- exon clipping of a haplotype (core:718-792) --------------------------------------------
    def exon_pieces(self, ht_str):
        f = ht_str.split("-")
        ht = [int(f[0])] + f[1:-1] + [int(f[-1])]
        out = []
        for el, er in self.exons:
            hl, hr = ht[0], ht[-1]
            if el > hr or er < hl:
                continue
            new = list(ht)
            if hl < el:
                done = False
                for i in range(1, len(new) - 1):
                    t, p, d = self.Vars[new[i]]
                    if (t != "deletion" and p >= el) or (t == "deletion" and p - 1 >= el):
                        hl = el
                        new = [hl] + new[i:]
                        done = True
                        break
                    if t == "deletion":
                        r = p + int(d)
                        if r >= el:
                            hl = r
                            new = [r] + new[i + 1:]
                            done = True
                            break
                if not done:
                    hl = el
                    new = [hl, hr]
            if hl < el:
                raise ReferenceError_("assert ht_left >= e_left")
            if hr > er:
                done = False
                for i in range(len(new) - 2, 0, -1):
                    t, r, d = self.Vars[new[i]]
                    if t == "deletion":
                        r = r + int(d) - 1
                    if (t != "deletion" and r <= er) or (t == "deletion" and r + 1 <= er):
                        hr = er
                        new = new[:i + 1] + [hr]
                        done = True
                        break
                    if t == "deletion":
                        l = r - int(d)
                        if l <= er:
                            hr = l
                            new = new[:i] + [hr]
                            done = True
                            break
                if not done:
                    hr = er
                    new = [hl, hr]
            if hl > hr:
                raise ReferenceError_("assert ht_left <= ht_right")
            if len(new) == 2:
                out.append("%d-%d" % (new[0], new[-1]))
            else:
                out.append("%d-%s-%d" % (new[0], "-".join(new[1:-1]), new[-1]))
        return out

    # -- add_count / add_stat ----------------------------------------------------------------------
    def add_count(self, count, ht_str):
        f = ht_str.split("-")
        left, right = int(f[0]), int(f[-1])
        if left > right:
            raise ReferenceError_("assert left <= right")
        ids = f[1:-1]
        alleles = set(self.Genes.keys()) - {self.ref_allele}
        for v in ids:
            if v.startswith("nv") or v not in self.Links:
                continue
            alleles &= set(self.Links[v])
        ids = set(ids)
        bad = set()
        j = min(lower_bound(self.Var_list, right + 1), len(self.Var_list) - 1)
        while j >= 0:
            v = self.Var_list[j][1]
            if v.startswith("nv") or v in ids or v not in self.Links:
                j -= 1
                continue
            if v in self.maxright and self.maxright[v] < left:
                break
            vl, vr = self.Vars[v][1], var_right(self.Vars[v])
            if left <= vl <= right or left <= vr <= right:
                bad |= set(self.Links[v])
            j -= 1
        alleles -= bad
        alleles &= set(count.keys())
        for a in alleles:
            count[a] += 1

    @staticmethod
    def add_stat(cmpt, counts, per_read, include=()):
        if not per_read:
            return ""
        mx = max(per_read.values())
        cur = set()
        for a, c in per_read.items():
            if c < mx:
                continue
            if include and a not in include:
                continue
            cur.add(a)
            counts[a] = counts.get(a, 0) + 1
        if not cur:
            return ""
        key = "-".join(sorted(cur))
        cmpt[key] = cmpt.get(key, 0) + 1
        return key

    # -- the streaming loop -------------------------------------------------------------------------
    def pileup_records(self, sam_text, base_locus=0):
        """(flag, pos0, cigar, seq) of every record: what pass 1 feeds the pileup."""
        recs = []
        for l in sam_text.split("\n"):
            if l and not l.startswith("@"):
                c = l.split()
                recs.append((int(c[1]), int(c[3]) - (base_locus + 1), c[5], c[9]))
        return recs

    def run(self, sam_text, base_locus=0, pileup_tables=None):
        """`pileup_tables` = (counts, nt_sets) of the WHOLE stream when `sam_text` is one shard of it (cut at a read-id boundary)."""
        o = self.opts
        lines = [l for l in sam_text.split("\n") if l and not l.startswith("@")]
        # pass 1: pileup over all records (core:445-449)
        if pileup_tables is not None:
            counts, nt_sets = pileup_tables
        else:
            counts, nt_sets = pileup(self.pileup_records(sam_text, base_locus), len(self.ref_seq), o["allow_discordant"])
        self.pileup_counts, self.nt_sets = counts, nt_sets
        interdist = pair_interdist(lines, o["simulation"]) if self.base == "codis" else None   # core:451-456

        hla = self.base == "hla"
        exons_cmpt, gene_cmpt, exons_counts, gene_counts = {}, {}, {}, {}
        num_reads = num_pairs = 0
        prev_id = None
        seen_l, seen_r, seen_u = set(), set(), set()
        lhts, rhts = set(), set()
        per_exon, per_gene = {}, {}
        pair_log = []

        def flush():
            pieces_e, pieces_g = [], []
            for ht in lhts | rhts:
                for eh in self.exon_pieces(ht):
                    if self.score:
                        self.add_count(per_exon, eh)
                    pieces_e.append(eh)
                if self.score:
                    self.add_count(per_gene, ht)
                pieces_g.append(ht)
            if not self.score:
                pair_log.append({"exon": sorted(pieces_e), "gene": sorted(pieces_g)})
                return
            ke = ""
            if hla:
                ke = self.add_stat(exons_cmpt, exons_counts, per_exon, self.rep_set)
            kg = self.add_stat(gene_cmpt, gene_counts, per_gene)
            pair_log.append({"exon": sorted(pieces_e), "gene": sorted(pieces_g), "exon_cls": ke, "gene_cls": kg})

        for l in lines:
            cols = l.split()
            read_id, flag, pos, cigar_str = cols[0], int(cols[1]), int(cols[3]), cols[5]
            if o["simulation"]:
                read_id = read_id.split("|")[0]
            read_seq = cols[9]
            pos -= base_locus + 1
            if pos < 0:
                continue
            if flag & 0x4:
                continue
            NM = Zs = MD = NH = ""
            for col in cols[11:]:
                if col.startswith("Zs"):
                    Zs = col[5:]
                elif col.startswith("MD"):
                    MD = col[5:]
                elif col.startswith("NM"):
                    NM = int(col[5:])
                elif col.startswith("NH"):
                    NH = int(col[5:])
            if NM == "" or NH == "":
                raise ReferenceError_("TypeError: missing NM/NH tag (quirk Q8)")
            if NM > o["num_editdist"]:
                continue
            if NH > 1:
                continue
            if not o["allow_discordant"] and not (flag & 0x2):
                continue
            is_left = bool(flag & 0x40)
            if is_left:
                if read_id in seen_l:
                    continue
                seen_l.add(read_id)
            elif flag & 0x80:
                if read_id in seen_r:
                    continue
                seen_r.add(read_id)
            else:
                if not o["allow_discordant"]:
                    raise ReferenceError_("assert allow_discordant")
                if read_id in seen_u:
                    continue
                seen_u.add(read_id)
            mkey = hit = None
            if self.memo is not None:
                mkey = (pos, cigar_str, read_seq, Zs, MD)
                hit = self.memo.get(mkey)
            if hit is not None:
                dec = hit[0]
            else:
                dec = self.decode(pos, cigar_str, read_seq, Zs, MD, counts, nt_sets)
                if dec is None and mkey is not None:
                    self.memo[mkey] = (None,)
            if dec is None:
                continue
            cmp_list, right_pos = dec
            num_reads += 1
            if read_id != prev_id:
                if prev_id is not None:
                    num_pairs += 1
                    flush()
                lhts, rhts = set(), set()
                per_exon, per_gene = {}, {}
                for a in self.Gene_names:
                    if a.find("BACKBONE") != -1:
                        continue
                    if self.base == "genome" and a.find("GRCh38") != -1:
                        continue
                    if a in self.rep_set:
                        per_exon[a] = 0
                    per_gene[a] = 0
            if hit is not None:
                c2, (cl, cr, la, ra) = hit[1], hit[2]
            else:
                # cmp_list2 (core:1351-1368)
                c2 = []
                for c in cmp_list:
                    c = list(c)
                    if c[0] == "match":
                        if c2 and c2[-1][0] == "match":
                            c2[-1][2] += c[2]
                        else:
                            c2.append(c)
                    elif c[0] == "mismatch" and (c[3] == "unknown" or c[3].startswith("nv")):
                        if c2 and c2[-1][0] == "match":
                            c2[-1][2] += 1
                        else:
                            c2.append(["match", c[1], 1])
                    else:
                        c2.append(c)
                cl, cr, la, ra = ambiguous_diffs(self.ref_seq, self.Vars, self.alts.left, self.alts.right,
                                                 self.list_l, self.list_r, c2)
                if mkey is not None:
                    self.memo[mkey] = (dec, c2, (cl, cr, la, ra))
            if self.trace is not None:
                self.trace.append({"cmp": [list(x) for x in c2], "iad": [cl, cr, la, ra]})
            mid = [c[3] for c in c2[cl:cr + 1] if c[0] in ("mismatch", "deletion", "insertion")]
            for lh in la:
                for rh in ra:
                    ht = "-".join(lh.split("-") + mid + rh.split("-"))
                    (lhts if is_left else rhts).add(ht)
            prev_id = read_id
        if prev_id is not None:
            num_pairs += 1
            if self.base == "codis" and self.gene == "D18S51":                      # core:1547-1552
                lhts, rhts = choose_pairs(lhts, rhts, interdist)
            flush()

        res = dict(num_reads=num_reads, num_pairs=num_pairs, pairs=pair_log, exons_cmpt=exons_cmpt,
                   gene_cmpt=gene_cmpt, gene_counts=gene_counts, em=[])
        if num_reads <= 0 or not self.score:
            return res
        res["counts_sorted"] = sorted([[a, c] for a, c in gene_counts.items()], key=lambda x: x[1], reverse=True)

        def em(cmpt, low, lengths):
            st = {}
            out = single_abundance(cmpt, low, lengths, st)
            res["em"].append({"cmpt": [[k, v] for k, v in cmpt.items()], "remove_low": low,
                              "use_length": bool(lengths), "result": out, "n_iter": st["n_iter"]})
            return out

        if hla:
            exon_prob = gene_prob = em(exons_cmpt, o["remove_low"], None)
            exon_alleles, psum = set(), 0.0
            for i, (a, p) in enumerate(exon_prob):
                if i >= 10 and p < 0.03:
                    break
                if len(self.rep_groups[a]) <= 1:
                    continue
                psum += p
                exon_alleles |= set(self.rep_groups[a])
            if exon_alleles:
                cm2 = {}
                for k, v in gene_cmpt.items():
                    k2 = "-".join(a for a in k.split("-") if a in exon_alleles)
                    if not k2:
                        continue
                    cm2[k2] = cm2.get(k2, 0) + v
                gp = em(cm2, True, self.Gene_lengths)
                comb = {}
                for a, p in exon_prob:
                    if a not in exon_alleles:
                        comb[a] = p
                for a, p in gp:
                    comb[a] = p * psum
                gene_prob = sorted([[a, p] for a, p in comb.items()], key=lambda x: x[1], reverse=True)
        else:
            if len(gene_cmpt) <= 1:
                if len(gene_cmpt) == 1:
                    raise ReferenceError_("TypeError: dict_keys is not subscriptable (quirk Q3)")
                gene_prob = []
            else:
                gene_prob = em(gene_cmpt, False, None)
        res["gene_prob"] = gene_prob
        return res


def report_lines(res, simulation=False, true_alleles=(), output_allele_counts=True):
    """Body of the report (core:1593, 1650-1677, 2076-2121)."""
    out = ["\t\t\t%d reads and %d pairs are aligned" % (res["num_reads"], res["num_pairs"])]
    for i, (a, c) in enumerate(res["counts_sorted"]):
        if simulation:
            found = False
            for t in true_alleles:
                if a == t:
                    out.append("\t\t\t*** %d ranked %s (count: %d)" % (i + 1, t, c))
                    found = True
            if i < 5 and not found:
                out.append("\t\t\t\t%d %s (count: %d)" % (i + 1, a, c))
        else:
            out.append("\t\t\t\t%d %s (count: %d)" % (i + 1, a, c))
            if i >= 9 and not output_allele_counts:
                break
    out.append("\n")
    found_list = [False] * len(true_alleles)
    for i, (a, p) in enumerate(res["gene_prob"]):
        if p < 0.01:
            break
        found = False
        if simulation:
            for k, t in enumerate(true_alleles):
                if a == t:
                    out.append("\t\t\t*** %d ranked %s (abundance: %.2f%%)" % (i + 1, t, p * 100.0))
                    found_list[k] = True
                    found = True
            if False not in found_list and i >= 10:
                break
        if not found:
            out.append("\t\t\t\t%d ranked %s (abundance: %.2f%%)" % (i + 1, a, p * 100.0))
        if not simulation and i >= 9:
            break
        if i >= 19:
            break
    return out
