"""Synthetic workload generator: HLA-like / STR-like loci, read pairs and HISAT2-dialect SAM.

No aligner (hisat2) and no IMGT-derived database exist in the build or on the GPU
box, so every input of the typing hot path is synthesised here from fixed seeds
(SURVEY.md section 8d).  The generator knows the *truth* alignment of every read, so the
SAM it writes carries exactly the fields the reference loop consumes
(hisatgenotype_typing_core.py:800-898): FLAG, POS, CIGAR, SEQ and the tags
``NM:i`` (edits NOT explained by known graph variants, quirk Q8), ``MD:Z``,
``Zs:Z`` (``gap|S/D/I|var_id`` items, typing_common.py:780-843 shows the producer
side of the same grammar), ``NH:i`` and ``YT:Z``.

This module is plain Python/numpy; it is workload tooling, not part of the timed path.
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# exon intervals printed by the reference's own golden assembly report for HLA-A
# (devel/hg_test3_assembly report "exons" line)
HLA_A_EXONS = [[300, 372], [503, 772], [1014, 1289], [1869, 2144],
               [2247, 2363], [2806, 2838], [2981, 3028], [3198, 3202]]
HLA_A_PRIMARY = [[503, 772], [1014, 1289]]

_BASES = "ACGT"


@dataclass
class Locus:
    """One typing locus in the vocabulary of the reference's index files.

    ``var_*`` lists are in ``Var_list`` order (sorted by position, stable), ids are
    ``hv<n>``.  ``links[var_id]`` is the allele list of the ``.link`` file.
    ``allele_names[0]`` is ``<gene>*BACKBONE`` (order = ``Gene_names[gene]``).
    """
    gene: str
    backbone: str
    var_ids: List[str]
    var_type: List[str]
    var_pos: List[int]
    var_data: List[str]
    links: Dict[str, List[str]]
    allele_names: List[str]
    exons: List[List[int]]
    primary_exons: List[List[int]]
    base_fname: str = "hla"
    # allele name -> sorted list of indices into var_* (its variants)
    allele_vars: Dict[str, List[int]] = field(default_factory=dict)

    @property
    def ref_allele(self) -> str:
        return "%s*BACKBONE" % self.gene

    def allele_sequence(self, name: str) -> str:
        """Spell an allele from the backbone and its variants
        (same construction as typing_core.py:2199-2237)."""
        seq, prev = [], 0
        bb = self.backbone
        for vi in self.allele_vars.get(name, []):
            t, p, d = self.var_type[vi], self.var_pos[vi], self.var_data[vi]
            assert prev <= p
            if p > prev:
                seq.append(bb[prev:p])
            if t == "single":
                seq.append(d)
                prev = p + 1
            elif t == "deletion":
                prev = p + int(d)
            else:
                seq.append(d)
                prev = p
        if prev < len(bb):
            seq.append(bb[prev:])
        return "".join(seq)

    def allele_length(self, name: str) -> int:
        n = len(self.backbone)
        for vi in self.allele_vars.get(name, []):
            t, d = self.var_type[vi], self.var_data[vi]
            if t == "deletion":
                n -= int(d)
            elif t == "insertion":
                n += len(d)
        return n

    def reference_dicts(self):
        """The dict-of-str arguments ``typing()`` takes (typing_core.py:249-286)."""
        g = self.gene
        Vars = {g: {}}
        Var_list = {g: []}
        for i, vid in enumerate(self.var_ids):
            Vars[g][vid] = [self.var_type[i], self.var_pos[i], self.var_data[i]]
            Var_list[g].append([self.var_pos[i], vid])
        Links = {vid: list(al) for vid, al in self.links.items()}
        Genes = {g: {}}
        for name in self.allele_names:
            Genes[g][name] = self.backbone if name == self.ref_allele else self.allele_sequence(name)
        Gene_names = {g: list(self.allele_names)}
        Gene_lengths = {g: {n: len(s) for n, s in Genes[g].items()}}
        refGenes = {g: self.ref_allele}
        refGene_loci = {g: [self.ref_allele, "6", 0, len(self.backbone) - 1,
                            [list(e) for e in self.exons], [list(e) for e in self.primary_exons]]}
        return dict(Vars=Vars, Var_list=Var_list, Links=Links, Genes=Genes, Gene_names=Gene_names,
                    Gene_lengths=Gene_lengths, refGenes=refGenes, refGene_loci=refGene_loci)

    def to_json(self) -> dict:
        return dict(gene=self.gene, backbone=self.backbone, var_ids=self.var_ids, var_type=self.var_type,
                    var_pos=self.var_pos, var_data=self.var_data, links=self.links,
                    allele_names=self.allele_names, exons=self.exons, primary_exons=self.primary_exons,
                    base_fname=self.base_fname)

    @staticmethod
    def from_json(d: dict) -> "Locus":
        loc = Locus(**d)
        loc._index_alleles()
        return loc

    def _index_alleles(self):
        idx = {vid: i for i, vid in enumerate(self.var_ids)}
        av: Dict[str, List[int]] = {}
        for vid, alleles in self.links.items():
            if vid not in idx:
                continue
            for a in alleles:
                av.setdefault(a, []).append(idx[vid])
        for a in av:
            av[a].sort()
        self.allele_vars = av


def _allele_name(gene: str, group: int, member: int) -> str:
    f1, f2 = group // 150 + 1, group % 150 + 1
    return "%s*%02d:%02d:01:%02d" % (gene, f1, f2, member + 1)


def _exonic(pos: int, right: int, exons) -> bool:
    for el, er in exons:
        if pos >= el and right <= er:
            return True
    return False


def make_hla_like_locus(gene: str = "A", n_alleles: int = 7000, length: int = 3569, n_vars: int = 2500,
                        seed: int = 101, sibling_frac: float = 0.35, deletion_frac: float = 0.07,
                        multi_allelic_frac: float = 0.03, n_backbone_equal: int = 1,
                        exons=None, primary_exons=None, var_id_base: int = 0,
                        unlinked_vars: int = 0, insertion_frac: float = 0.0) -> Locus:
    """HLA-A-like locus following SURVEY.md section 8d: ``n_vars`` sites in [30, L-30),
    93 % single / 7 % short deletions, carrier spectrum 70 % U[1,3], 20 % U[4,50],
    10 % U[2 %, 30 %] of the alleles; ``sibling_frac`` of the alleles are intron-only
    siblings of another allele (identical exonic variant set), so the exon->gene
    hand-off (typing_core.py:1739-1767) has multi-member representative groups.
    ``unlinked_vars`` adds variants present in ``.snp`` but absent from ``.link``
    (exercises the ``var_id not in Links`` branches of add_count, core:644-657).
    """
    rng = random.Random(seed)
    exons = [list(e) for e in (exons if exons is not None else HLA_A_EXONS) if e[1] < length]
    primary_exons = [list(e) for e in (primary_exons if primary_exons is not None else HLA_A_PRIMARY) if e[1] < length]
    backbone = "".join(rng.choice(_BASES) for _ in range(length))

    # allele groups: one exon profile per group, members differ by intronic variants only
    n_real = n_alleles - n_backbone_equal
    n_groups = max(1, int(round(n_real * (1.0 - sibling_frac))))
    group_of = list(range(n_groups)) + [rng.randrange(n_groups) for _ in range(n_real - n_groups)]
    members: Dict[int, List[int]] = {}
    names: List[str] = []
    for a, g in enumerate(group_of):
        members.setdefault(g, []).append(a)
        names.append(_allele_name(gene, g, len(members[g]) - 1))
    bb_equal = ["%s*%02d:%02d:01:%02d" % (gene, 99, 99 - i, 1) for i in range(n_backbone_equal)]

    # variant sites
    lo, hi = 30, length - 30
    n_sites = min(n_vars, hi - lo)
    sites = sorted(rng.sample(range(lo, hi), n_sites))
    blocked = set()
    raw = []  # (pos, type, data)
    for p in sites:
        if p in blocked:
            continue
        if insertion_frac > 0 and rng.random() < insertion_frac:
            raw.append((p, "insertion", "".join(rng.choice(_BASES) for _ in range(rng.randint(1, 3)))))
            blocked.add(p)
            blocked.add(p + 1)
        elif rng.random() < deletion_frac:
            dl = rng.randint(1, 4)
            if any((p + k) in blocked for k in range(dl)) or p + dl >= hi:
                dl = 1
            raw.append((p, "deletion", str(dl)))
            for k in range(dl + 1):          # +1: never two adjacent deletions (one CIGAR D op each)
                blocked.add(p + k)
        else:
            alts = [b for b in _BASES if b != backbone[p]]
            rng.shuffle(alts)
            raw.append((p, "single", alts[0]))
            if rng.random() < multi_allelic_frac:
                raw.append((p, "single", alts[1]))
            blocked.add(p)
    # a single inside someone's deletion span is not allowed
    del_cover = set()
    for p, t, d in raw:
        if t == "deletion":
            for k in range(int(d)):
                del_cover.add(p + k)
    raw = [(p, t, d) for (p, t, d) in raw if t == "deletion" or p not in del_cover]
    # no insertion directly behind a deletion's last base (would give adjacent D and I ops)
    del_next = set(p + int(d) for p, t, d in raw if t == "deletion")
    raw = [(p, t, d) for (p, t, d) in raw if not (t == "insertion" and (p in del_next or p in del_cover))]
    order = {"insertion": 0, "single": 1, "deletion": 2}
    raw.sort(key=lambda x: (x[0], order[x[1]], x[2]))

    def carriers_count() -> int:
        r = rng.random()
        if r < 0.70:
            return rng.randint(1, 3)
        if r < 0.90:
            return rng.randint(4, 50)
        return rng.randint(max(1, int(0.02 * n_real)), max(1, int(0.30 * n_real)))

    var_ids, var_type, var_pos, var_data = [], [], [], []
    links: Dict[str, List[str]] = {}
    used_at_pos: Dict[int, set] = {}
    n_unlinked_left = unlinked_vars
    for i, (p, t, d) in enumerate(raw):
        vid = "hv%d" % (var_id_base + i)
        right = p + int(d) - 1 if t == "deletion" else p
        k = min(carriers_count(), n_real)
        if _exonic(p, right, exons):
            # exonic variants are carried by whole groups
            ng = max(1, min(n_groups, int(round(k * n_groups / float(n_real)))))
            gs = rng.sample(range(n_groups), ng)
            car = sorted(a for g in gs for a in members[g])
        else:
            car = sorted(rng.sample(range(n_real), k))
        taken = used_at_pos.setdefault(p, set())
        car = [a for a in car if a not in taken]
        taken.update(car)
        var_ids.append(vid)
        var_type.append(t)
        var_pos.append(p)
        var_data.append(d)
        if n_unlinked_left > 0 and not _exonic(p, right, exons) and rng.random() < 0.05:
            n_unlinked_left -= 1
            continue
        links[vid] = [names[a] for a in car]

    loc = Locus(gene=gene, backbone=backbone, var_ids=var_ids, var_type=var_type, var_pos=var_pos,
                var_data=var_data, links=links, allele_names=[], exons=exons, primary_exons=primary_exons,
                base_fname="hla")
    loc._index_alleles()
    # Gene_names order: backbone, then alleles by first appearance scanning Var_list x Links
    # (typing_core.py:2199-2237), then the backbone-equal alleles (core:2463-2467)
    seen, ordered = set(), []
    for vid in var_ids:
        for a in links.get(vid, []):
            if a not in seen:
                seen.add(a)
                ordered.append(a)
    no_var = [n for n in names if n not in seen]
    loc.allele_names = [loc.ref_allele] + ordered + no_var + bb_equal
    return loc


def make_str_like_locus(gene: str = "D8S1179", unit: str = "TCTA", max_repeats: int = 19, min_repeats: int = 7,
                        flank: int = 200, seed: int = 7, var_id_base: int = 0) -> Locus:
    """CODIS-like STR locus: backbone = flank + unit x max_repeats + flank; every shorter
    allele is one left-shifted deletion of k units at the first repeat position
    (``leftshift=True`` for codis, typing_common.py:564)."""
    rng = random.Random(seed)
    def rnd(n):
        s = []
        while len(s) < n:
            b = rng.choice(_BASES)
            s.append(b)
        return "".join(s)
    left = rnd(flank)
    # make sure the flank does not extend the repeat
    while left.endswith(unit[-1]):
        left = left[:-1] + rng.choice([b for b in _BASES if b != unit[-1]])
    right = rnd(flank)
    while right.startswith(unit[0]):
        right = rng.choice([b for b in _BASES if b != unit[0]]) + right[1:]
    backbone = left + unit * max_repeats + right
    p0 = len(left)
    var_ids, var_type, var_pos, var_data, links = [], [], [], [], {}
    names = []
    i = 0
    for rep in range(max_repeats - 1, min_repeats - 1, -1):
        dl = (max_repeats - rep) * len(unit)
        vid = "hv%d" % (var_id_base + i)
        i += 1
        name = "%s*%d" % (gene, rep)
        var_ids.append(vid); var_type.append("deletion"); var_pos.append(p0); var_data.append(str(dl))
        links[vid] = [name]
        names.append(name)
    # Var_list order for equal positions: file order; the reference numbers deletions by data
    loc = Locus(gene=gene, backbone=backbone, var_ids=var_ids, var_type=var_type, var_pos=var_pos,
                var_data=var_data, links=links, allele_names=[], exons=[[0, len(backbone) - 1]],
                primary_exons=[[0, len(backbone) - 1]], base_fname="codis")
    loc._index_alleles()
    loc.allele_names = [loc.ref_allele] + names + ["%s*%d" % (gene, max_repeats)]
    return loc


# --------------------------------------------------------------------------------------
# reads
# --------------------------------------------------------------------------------------
@dataclass
class Alignment:
    qname: str
    flag: int
    pos: int                      # 0-based backbone position of the first aligned base
    cigar: List[Tuple[str, int]]
    seq: str                      # forward-strand read bases (incl. soft clips)
    md: str
    zs: str
    nm: int
    nh: int = 1
    yt: str = "CP"
    mate_pos: int = 0

    def cigar_str(self) -> str:
        return "".join("%d%s" % (n, op) for op, n in self.cigar)

    def sam_line(self, rname: str, base_locus: int = 0) -> str:
        tags = ["NM:i:%d" % self.nm, "MD:Z:%s" % self.md]
        if self.zs:
            tags.append("Zs:Z:%s" % self.zs)
        tags += ["NH:i:%d" % self.nh, "YT:Z:%s" % self.yt]
        return "\t".join([self.qname, str(self.flag), rname, str(self.pos + 1 + base_locus), "60",
                          self.cigar_str(), "=", str(self.mate_pos + 1 + base_locus), "0", self.seq,
                          "I" * len(self.seq)] + tags)


class AlleleMap:
    """Allele sequence annotated with backbone coordinates and the variant under each base."""

    def __init__(self, locus: Locus, name: str):
        bb = locus.backbone
        seq: List[str] = []
        bpos: List[int] = []
        vid: List[int] = []
        dels: Dict[int, int] = {}   # allele index i -> deletion var index between base i-1 and i
        ins: Dict[int, int] = {}    # allele index i -> insertion var index if base i is an inserted base
        prev = 0
        for vi in locus.allele_vars.get(name, []):
            t, p, d = locus.var_type[vi], locus.var_pos[vi], locus.var_data[vi]
            if p > prev:
                seq.extend(bb[prev:p]); bpos.extend(range(prev, p)); vid.extend([-1] * (p - prev))
            if t == "single":
                seq.append(d); bpos.append(p); vid.append(vi)
                prev = p + 1
            elif t == "deletion":
                dels[len(seq)] = vi
                prev = p + int(d)
            else:
                # inserted bases sit in front of backbone base p (typing_core.py:2228-2231)
                for ch in d:
                    seq.append(ch); bpos.append(p); vid.append(-1); ins[len(seq) - 1] = vi
                prev = p
        if prev < len(bb):
            seq.extend(bb[prev:]); bpos.extend(range(prev, len(bb))); vid.extend([-1] * (len(bb) - prev))
        self.seq = "".join(seq)
        self.bpos = bpos
        self.vid = vid
        self.dels = dels
        self.ins = ins
        self.locus = locus


def _align_read(amap: AlleleMap, start: int, read_len: int, rng: random.Random, err_rate: float,
                softclip: Tuple[int, int] = (0, 0), novel_del_at: int = -1, novel_ins_at: int = -1):
    """Truth alignment of allele bases [start, start+read_len) against the backbone."""
    loc = amap.locus
    bb = loc.backbone
    cigar: List[List] = []
    md: List[str] = []
    zs: List[str] = []
    read: List[str] = []
    nm = 0
    md_run = 0
    zs_gap = 0         # read bases since the previous Zs item
    last_was_del = False

    def push(op, n=1):
        if cigar and cigar[-1][0] == op:
            cigar[-1][1] += n
        else:
            cigar.append([op, n])

    sc_l, sc_r = softclip
    end = start + read_len
    i0, i1 = start + sc_l, end - sc_r
    for k in range(sc_l):
        read.append(rng.choice(_BASES))
    if sc_l:
        push("S", sc_l)
    ins_map = getattr(amap, "ins", {})
    # never start or stop inside a known insertion (an aligner would soft-clip or report a partial, novel one)
    while i0 in ins_map and i0 < i1:
        i0 += 1
    while (i1 - 1) in ins_map and i1 > i0:
        i1 -= 1
    for i in range(i0, i1):
        if i in ins_map:
            # known insertion: CIGAR I, Zs "gap|I|id" once per run; the consumer advances read_pos over the
            # inserted bases but not Zs_pos (core:994-1001), so they count into the NEXT item's gap
            vi = ins_map[i]
            if (i - 1) not in ins_map or ins_map[i - 1] != vi or i == i0:
                zs.append("%d|I|%s" % (zs_gap, loc.var_ids[vi])); zs_gap = 0
            read.append(amap.seq[i])
            push("I")
            zs_gap += 1
            last_was_del = False
            continue
        if i == novel_ins_at and i > i0 + 5 and i + 5 < i1 and not last_was_del and amap.vid[i] < 0 and i not in amap.dels:
            k = rng.randint(1, 2)
            for _ in range(k):
                read.append(rng.choice(_BASES))
            push("I", k)
            nm += k
            zs_gap += k
        if i > i0 and i in amap.dels:
            vi = amap.dels[i]
            dl = int(loc.var_data[vi])
            p = loc.var_pos[vi]
            md.append(str(md_run)); md_run = 0
            md.append("^" + bb[p:p + dl])
            push("D", dl)
            zs.append("%d|D|%s" % (zs_gap, loc.var_ids[vi])); zs_gap = 0
            last_was_del = True
        elif (i > i0 and i == novel_del_at and i + 1 < i1 and not last_was_del and amap.vid[i] < 0
              and (i + 1) not in amap.dels and amap.bpos[i] == amap.bpos[i - 1] + 1):
            # novel 1-bp deletion: skip allele base i entirely (read lacks it)
            p = amap.bpos[i]
            md.append(str(md_run)); md_run = 0
            md.append("^" + bb[p])
            push("D", 1)
            nm += 1
            last_was_del = True
            continue
        b = amap.seq[i]
        p = amap.bpos[i]
        vi = amap.vid[i]
        if vi < 0 and err_rate > 0 and rng.random() < err_rate:
            b = rng.choice([x for x in _BASES if x != b])
        read.append(b)
        push("M")
        if b != bb[p]:
            md.append(str(md_run)); md_run = 0
            md.append(bb[p])
            if vi >= 0:
                zs.append("%d|S|%s" % (zs_gap, loc.var_ids[vi])); zs_gap = 0
            else:
                nm += 1
                zs_gap += 1
            last_was_del = False
        else:
            md_run += 1
            zs_gap += 1
            last_was_del = False
    md.append(str(md_run))
    for k in range(sc_r):
        read.append(rng.choice(_BASES))
    if sc_r:
        push("S", sc_r)
    # standard MD: "0" between two adjacent non-match items is kept (e.g. ^AC0T)
    return amap.bpos[i0], [(op, n) for op, n in cigar], "".join(read), "".join(md), ",".join(zs), nm


def simulate_pairs(locus: Locus, sample_alleles: Sequence[str], n_pairs: int, read_len: int = 150,
                   frag_len: Tuple[int, int] = (400, 400), err_rate: float = 0.0, seed: int = 1,
                   simulation_names: bool = False, softclip_frac: float = 0.0, novel_del_frac: float = 0.0,
                   tile_interval: Optional[int] = None, single_end: bool = False,
                   multi_hit_frac: float = 0.0, discordant_frac: float = 0.0,
                   unaligned_frac: float = 0.0, dup_frac: float = 0.0, novel_ins_frac: float = 0.0) -> List[Alignment]:
    """Draw fragments from the sample's alleles and return truth alignments, two per pair.

    ``tile_interval``: if set, fragments start every ``tile_interval`` bases of each allele
    (like typing_common.simulate_reads, common:848); ``n_pairs`` is then ignored.
    The *_frac knobs exercise the record filters of typing_core.py:815-872.
    """
    rng = random.Random(seed)
    maps = [AlleleMap(locus, a) for a in sample_alleles]
    out: List[Alignment] = []
    starts: List[Tuple[int, int, int]] = []
    if tile_interval:
        for mi, m in enumerate(maps):
            fl = frag_len[0]
            for s in range(0, len(m.seq) - fl + 1, tile_interval):
                starts.append((mi, s, fl))
    else:
        for _ in range(n_pairs):
            mi = rng.randrange(len(maps))
            fl = rng.randint(frag_len[0], frag_len[1])
            s = rng.randrange(0, len(maps[mi].seq) - fl + 1)
            starts.append((mi, s, fl))
    for k, (mi, s, fl) in enumerate(starts):
        m = maps[mi]
        recs = []
        mates = [(s, True)] if single_end else [(s, True), (s + fl - read_len, False)]
        fwd_first = rng.random() < 0.5
        for (st, is_leftmost) in mates:
            sc = (0, 0)
            if softclip_frac > 0 and rng.random() < softclip_frac:
                sc = (rng.randint(1, 6), 0) if rng.random() < 0.5 else (0, rng.randint(1, 6))
            nd = -1
            if novel_del_frac > 0 and rng.random() < novel_del_frac:
                nd = st + rng.randint(20, read_len - 20)
            ni = -1
            if novel_ins_frac > 0 and rng.random() < novel_ins_frac:
                ni = st + rng.randint(20, read_len - 20)
            pos, cigar, seq, md, zs, nm = _align_read(m, st, read_len, rng, err_rate, sc, nd, ni)
            recs.append((pos, cigar, seq, md, zs, nm, is_leftmost))
        for j, (pos, cigar, seq, md, zs, nm, is_leftmost) in enumerate(recs):
            if single_end:
                flag = 0
            else:
                first_in_pair = (is_leftmost == fwd_first)
                flag = 0x1 | 0x2 | (0x40 if first_in_pair else 0x80) | (0x20 if is_leftmost else 0x10)
            if simulation_names:
                side = "L" if (flag & 0x40 or single_end) else "R"
                info = "%d_%s" % (pos + 1, "".join("%d%s" % (n, op) for op, n in cigar))
                if zs:
                    info += "_" + zs
                qname = ("%d|%s_%s" % (k + 1, side, info))[:251]
            else:
                qname = "r%07d" % (k + 1)
            al = Alignment(qname=qname, flag=flag, pos=pos, cigar=cigar, seq=seq, md=md, zs=zs, nm=nm,
                           mate_pos=recs[1 - j][0] if len(recs) == 2 else pos)
            r = rng.random()
            if r < multi_hit_frac:
                al.nh = 2
            elif r < multi_hit_frac + discordant_frac and not single_end:
                al.flag &= ~0x2
                al.yt = "DP"
            elif r < multi_hit_frac + discordant_frac + unaligned_frac:
                al.flag |= 0x4
            out.append(al)
            if dup_frac > 0 and rng.random() < dup_frac:
                out.append(al)   # secondary line of the same mate: dropped by the duplicate filter
    return out


def sam_text(locus: Locus, alignments: Sequence[Alignment], name_sorted: bool = True,
             base_locus: int = 0, header: bool = False) -> str:
    """SAM body in the order the reference's loop sees it: coordinate order (BAM), then the
    stable by-name sort of ``sort -k1,1 -s`` (typing_core.py:458-468, C locale)."""
    als = sorted(alignments, key=lambda a: a.pos)            # samtools view of a sorted BAM
    if name_sorted:
        als = sorted(als, key=lambda a: a.qname.encode())    # stable, bytewise (LC_ALL=C)
    lines = []
    if header:
        lines.append("@SQ\tSN:%s\tLN:%d" % (locus.ref_allele, len(locus.backbone)))
    rname = locus.ref_allele
    for a in als:
        lines.append(a.sam_line(rname, base_locus))
    return "\n".join(lines) + "\n"


def pick_sample(locus: Locus, seed: int, n: int = 2, with_siblings: bool = True) -> List[str]:
    """Choose the sample's true alleles with ``random.Random(seed)`` among alleles that carry variants."""
    rng = random.Random(seed)
    cand = [a for a in locus.allele_names[1:] if a in locus.allele_vars]
    return sorted(rng.sample(cand, n))


# --------------------------------------------------------------------------------------
# fast read simulator for bench-size inputs (event based: O(variants in the read), not O(read length))
# --------------------------------------------------------------------------------------
class _FastAllele:
    def __init__(self, locus: Locus, name: str):
        import bisect
        self.bisect = bisect
        m = AlleleMap(locus, name)
        self.seq = m.seq
        self.bpos = np.asarray(m.bpos, dtype=np.int64)
        # events in allele coordinates: (allele index, kind, var index); deletions sit before base i
        ev = [(i, 0, v) for i, v in enumerate(m.vid) if v >= 0] + [(i, 1, v) for i, v in m.dels.items()]
        ev.sort(key=lambda e: (e[0], -e[1]))      # a deletion before base i precedes a single at base i
        self.ev_pos = [e[0] for e in ev]
        self.ev = ev
        self.is_var = set(i for i, v in enumerate(m.vid) if v >= 0)
        self.locus = locus

    def align(self, start: int, read_len: int, err_pos):
        """(pos0, cigar_str, seq, md, zs, nm) of allele bases [start, start+read_len) with substitution
        errors at the given allele offsets (never on variant bases)."""
        loc, bb = self.locus, self.locus.backbone
        end = start + read_len
        lo = self.bisect.bisect_left(self.ev_pos, start)
        hi = self.bisect.bisect_left(self.ev_pos, end)
        events = []
        for k in range(lo, hi):
            i, kind, v = self.ev[k]
            if kind == 1 and i == start:
                continue                      # deletion in front of the first base is outside the read
            events.append((i, -1 if kind == 1 else 0, v))
        seq = self.seq[start:end]
        if err_pos:
            s = list(seq)
            for e in err_pos:
                if e in self.is_var:
                    continue
                alt = _BASES[(_BASES.index(s[e - start]) + 1 + (e % 3)) % 4]
                s[e - start] = alt
                events.append((e, 1, -1))
            seq = "".join(s)
            events.sort(key=lambda x: (x[0], x[1]))
        cigar, md, zs = [], [], []
        nm = 0
        m_run = 0            # current M run
        md_run = 0
        zs_gap = 0
        cur = start
        for i, kind, v in events:
            n = i - cur       # matching bases before the event
            m_run += n; md_run += n; zs_gap += n
            cur = i
            if kind == -1:    # known deletion before base i
                dl = int(loc.var_data[v]); p = loc.var_pos[v]
                cigar.append("%dM%dD" % (m_run, dl)); m_run = 0
                md.append("%d^%s" % (md_run, bb[p:p + dl])); md_run = 0
                zs.append("%d|D|%s" % (zs_gap, loc.var_ids[v])); zs_gap = 0
            else:
                p = int(self.bpos[i])
                md.append("%d%s" % (md_run, bb[p])); md_run = 0
                m_run += 1
                cur = i + 1
                if kind == 0:
                    zs.append("%d|S|%s" % (zs_gap, loc.var_ids[v])); zs_gap = 0
                else:
                    nm += 1
                    zs_gap += 1
        n = end - cur
        m_run += n; md_run += n
        cigar.append("%dM" % m_run)
        md.append("%d" % md_run)
        return int(self.bpos[start]), "".join(cigar), seq, "".join(md), ",".join(zs), nm


def simulate_sam_fast(locus: Locus, sample_alleles: Sequence[str], n_pairs: int, read_len: int = 150,
                      frag_len: Tuple[int, int] = (350, 450), err_rate: float = 0.0, seed: int = 1) -> str:
    """Name-grouped SAM text for ``n_pairs`` random fragments (same dialect as ``sam_text``); sized for
    the 1M-read bench configuration (about 15 us per read)."""
    rng = np.random.RandomState(seed)
    maps = [_FastAllele(locus, a) for a in sample_alleles]
    which = rng.randint(0, len(maps), n_pairs)
    flen = rng.randint(frag_len[0], frag_len[1] + 1, n_pairs)
    u = rng.random_sample(n_pairs)
    orient = rng.random_sample(n_pairs) < 0.5
    n_err = rng.binomial(read_len, err_rate, (n_pairs, 2)) if err_rate > 0 else None
    rname = locus.ref_allele
    qual = "I" * read_len
    out = []
    for k in range(n_pairs):
        m = maps[which[k]]
        fl = int(flen[k])
        s = int(u[k] * (len(m.seq) - fl + 1))
        recs = []
        for j, st in enumerate((s, s + fl - read_len)):
            errs = None
            if n_err is not None and n_err[k, j]:
                errs = sorted(set(int(x) for x in rng.randint(st, st + read_len, n_err[k, j])))
            recs.append(m.align(st, read_len, errs))
        qname = "r%07d" % (k + 1)
        for j, (pos, cigar, seq, md, zs, nm) in enumerate(recs):
            leftmost = j == 0
            first = leftmost == bool(orient[k])
            flag = 0x1 | 0x2 | (0x40 if first else 0x80) | (0x20 if leftmost else 0x10)
            tags = "NM:i:%d\tMD:Z:%s\t" % (nm, md)
            if zs:
                tags += "Zs:Z:%s\t" % zs
            out.append("%s\t%d\t%s\t%d\t60\t%s\t=\t%d\t0\t%s\t%s\t%sNH:i:1\tYT:Z:CP" % (
                qname, flag, rname, pos + 1, cigar, recs[1 - j][0] + 1, seq, qual, tags))
    return "\n".join(out) + "\n"


def write_index(loci: Sequence[Locus], ix_dir: str, base_fname: str) -> None:
    """Write loci as the index files the reference reads (formats: hisatgenotype_amd.indexio docstring;
    writer side in the reference: typing_process.py:1055-1108)."""
    import os
    os.makedirs(ix_dir, exist_ok=True)
    full = os.path.join(ix_dir, base_fname)
    with open(full + "_backbone.fa", "w") as fa, open(full + ".locus", "w") as lo, open(full + ".snp", "w") as sn, \
            open(full + ".link", "w") as li, open(full + ".allele", "w") as al, open(full + ".partial", "w") as pa:
        for loc in loci:
            fa.write(">%s\n" % loc.ref_allele)
            for i in range(0, len(loc.backbone), 60):
                fa.write(loc.backbone[i:i + 60] + "\n")
            prim = {tuple(e) for e in loc.primary_exons}
            exon_str = ",".join("%d-%d%s" % (e[0], e[1], "p" if tuple(e) in prim else "") for e in loc.exons)
            lo.write("%s\t6\t0\t%d\t%d\t%s\t+\n" % (loc.ref_allele, len(loc.backbone) - 1, len(loc.backbone), exon_str))
            for i, vid in enumerate(loc.var_ids):
                sn.write("%s\t%s\t%s\t%d\t%s\n" % (vid, loc.var_type[i], loc.ref_allele, loc.var_pos[i], loc.var_data[i]))
            for vid, alleles in loc.links.items():
                li.write("%s\t%s\n" % (vid, " ".join(alleles)))
            for name in loc.allele_names[1:]:
                al.write(name + "\n")


def write_genome_index(loci: Sequence[Locus], ix_dir: str, genome_name: str, family: str, chrom: str = "6",
                       gap: int = 500, seed: int = 1) -> Dict[str, Tuple[int, int]]:
    """Write loci as a GENOTYPE-GENOME index `<ix_dir>/<genome_name>.*` (consumer: typing_core.py:2326-2397): one chromosome
    = random spacer, backbone, spacer, backbone, ...; `.fa` + `.fa.fai`; `.locus` rows `FAMILY allele chrom left right exons
    strand` in chromosome coordinates; `.snp` positions on the chromosome; `.allele` / `.partial` rows `family<TAB>name`.
    Returns {gene: (left, right)} (0-based, inclusive)."""
    import os
    os.makedirs(ix_dir, exist_ok=True)
    rng = random.Random(seed)
    full = os.path.join(ix_dir, genome_name)
    seq, spans = [], {}
    pos = 0
    for loc in loci:
        spacer = "".join(rng.choice(_BASES) for _ in range(gap))
        seq.append(spacer)
        pos += gap
        spans[loc.gene] = (pos, pos + len(loc.backbone) - 1)
        seq.append(loc.backbone)
        pos += len(loc.backbone)
    seq.append("".join(rng.choice(_BASES) for _ in range(gap)))
    chrom_seq = "".join(seq)
    header = ">%s\n" % chrom
    with open(full + ".fa", "w") as fa:
        fa.write(header)
        for i in range(0, len(chrom_seq), 60):
            fa.write(chrom_seq[i:i + 60] + "\n")
    with open(full + ".fa.fai", "w") as fai:
        fai.write("%s\t%d\t%d\t60\t61\n" % (chrom, len(chrom_seq), len(header)))
    with open(full + ".locus", "w") as lo, open(full + ".snp", "w") as sn, open(full + ".link", "w") as li, \
            open(full + ".allele", "w") as al, open(full + ".partial", "w") as pa:
        for loc in loci:
            left, right = spans[loc.gene]
            prim = {tuple(e) for e in loc.primary_exons}
            exon_str = ",".join("%d-%d%s" % (e[0], e[1], "p" if tuple(e) in prim else "") for e in loc.exons)
            lo.write("%s\t%s\t%s\t%d\t%d\t%s\t+\n" % (family.upper(), loc.ref_allele, chrom, left, right, exon_str))
            for i, vid in enumerate(loc.var_ids):
                sn.write("%s\t%s\t%s\t%d\t%s\n" % (vid, loc.var_type[i], chrom, loc.var_pos[i] + left, loc.var_data[i]))
            for vid, alleles in loc.links.items():
                li.write("%s\t%s\n" % (vid, " ".join(alleles)))
            for name in loc.allele_names[1:]:
                al.write("%s\t%s\n" % (family, name))
        pa.write("")
    return spans
