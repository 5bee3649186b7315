"""Readers for the on-disk typing index that `hisatgenotype_extract_vars` writes.

Formats (one record per line, tab separated unless noted), as consumed by the reference:
  <base>.locus        allele_name chrom left right length exon_str strand      typing_common.py:279-309
                      (genotype-genome variant: gene allele_name chrom left right exon_str strand)
                      exon_str = "l-r,l-r,..." with a trailing "p" on primary exons
  <base>.snp          var_id type allele_name pos data                          typing_common.py:339-368
  <base>.link         var_id <tab or space> allele allele ...                   typing_common.py:388-403
  <base>.allele       allele_name                                               typing_core.py:2416-2418
  <base>.partial      allele_name                                               typing_core.py:2420-2422
  <base>_backbone.fa  FASTA, one backbone allele per gene                       typing_common.py:313-334

`load_index` assembles the dict arguments `typing()` takes, the way `genotyping_locus` does for a stand-alone
(non genotype-genome) index (typing_core.py:2399-2480), without spelling out the ~7 000 allele sequences: only
their names and lengths are needed on this path.
"""
import os


def read_locus(fname, isgenome=False, target="", refGenes=None, refGene_loci=None):
    refGenes = {} if refGenes is None else refGenes
    refGene_loci = {} if refGene_loci is None else refGene_loci
    with open(fname) as f:
        lines = f.read().strip("\n").split("\n")
    for line in lines:
        if not line:
            continue
        fields = line.split()
        if isgenome:
            gene, gene_name, chrom, left, right, exon_str, strand = fields
            if gene.lower() != target:
                continue
        else:
            gene_name, chrom, left, right, _, exon_str, strand = fields
        g = gene_name.split("*")[0]
        if g in refGenes:
            raise ValueError("duplicate gene %s in %s" % (g, fname))
        refGenes[g] = gene_name
        exons, primary = [], []
        for ex in exon_str.split(","):
            is_p = ex.endswith("p")
            if is_p:
                ex = ex[:-1]
            l, r = ex.split("-")
            exons.append([int(l), int(r)])
            if is_p:
                primary.append([int(l), int(r)])
        refGene_loci[g] = [gene_name, chrom, int(left), int(right), exons, primary]
    return refGenes, refGene_loci


def read_allele_seq(fname, dic=None, genes=False):
    dic = {} if dic is None else dic
    with open(fname) as f:
        chunks = f.read().strip("\n").split(">")[1:]
    for ch in chunks:
        ix = ch.find("\n")
        name, seq = ch[:ix], ch[ix:].replace("\n", "")
        tgt = dic.setdefault(name.split("*")[0], {}) if genes else dic
        if name in tgt:
            raise ValueError("non-unique sequence name: %s" % name)
        tgt[name] = seq
    return dic


def read_variants(fname, genes=True):
    vardata, varlist = {}, {}
    with open(fname) as f:
        lines = f.read().strip("\n").split("\n")
    for line in lines:
        if not line:
            continue
        var_id, var_type, name, pos, var = line.split("\t")
        if var_type == "Deletion":          # (the reference compares against a capitalised literal that never occurs)
            var = int(var)
        pos = int(pos)
        gene = name.split("*")[0] if genes else name
        vardata.setdefault(gene, {})
        varlist.setdefault(gene, [])
        if genes:
            if var_id in vardata[gene]:
                raise ValueError("duplicate variant id %s" % var_id)
            vardata[gene][var_id] = [var_type, pos, var]
            varlist[gene].append([pos, var_id])
        else:
            varlist[gene].append([pos, var_type, var, var_id])
    for gene in varlist:
        varlist[gene].sort(key=lambda x: x[0])
    return (vardata, varlist) if genes else varlist


def read_links(fname):
    links = {}
    with open(fname) as f:
        lines = f.read().strip("\n").split("\n")
    for line in lines:
        if not line:
            continue
        fields = line.replace(" ", "\t").split("\t")
        if fields[0] in links:
            raise ValueError("duplicate link id %s" % fields[0])
        links[fields[0]] = fields[1:]
    return links


def read_names(fname):
    if not os.path.exists(fname):
        return []
    with open(fname) as f:
        return [l.strip() for l in f if l.strip()]


def _allele_length(backbone_len, var_ids, Vars):
    n = backbone_len
    for v in var_ids:
        t, _, d = Vars[v]
        if t == "deletion":
            n -= int(d)
        elif t == "insertion":
            n += len(d)
    return n


def load_index(ix_dir, base_fname):
    """Dict arguments of typing() for a stand-alone index `<ix_dir>/<base_fname>.*`.

    ``Genes[gene]`` maps every allele name to None except the backbone (its sequence); ``Gene_names`` follows the
    reference's order (backbone, alleles by first appearance scanning Var_list x Links, then alleles equal to the
    backbone -- the reference iterates a set there, quirk Q7; here they come sorted)."""
    full = os.path.join(ix_dir, base_fname)
    alleles = read_names(full + ".allele")
    partial_alleles = set(read_names(full + ".partial"))
    refGenes, refGene_loci = read_locus(full + ".locus", False, base_fname)
    Vars, Var_list = read_variants(full + ".snp", True)
    Links = read_links(full + ".link")
    backbones = read_allele_seq(full + "_backbone.fa", {}, True)
    Genes, Gene_names, Gene_lengths = {}, {}, {}
    for gene in refGene_loci:
        Vars.setdefault(gene, {})
        Var_list.setdefault(gene, [])
    for gene, bb in backbones.items():
        (bname, bseq), = bb.items()
        per_allele = {}
        for _, vid in Var_list.get(gene, []):
            for a in Links.get(vid, []):
                per_allele.setdefault(a, []).append(vid)
        names = [bname] + list(per_allele.keys())
        if len(names) <= 1:
            names.append("%s*GRCh38" % gene)
        seen = set(names)
        for a in sorted(x for x in alleles if x.split("*")[0] == gene and x not in seen):
            names.append(a)
        Genes[gene] = {n: None for n in names}
        Genes[gene][bname] = bseq
        Gene_names[gene] = names
        Gene_lengths[gene] = {n: _allele_length(len(bseq), per_allele.get(n, []), Vars.get(gene, {})) for n in names}
    dbversion = open(full + ".version").read() if os.path.exists(full + ".version") else "NONE"
    return dict(refGenes=refGenes, refGene_loci=refGene_loci, Genes=Genes, Gene_names=Gene_names, Gene_lengths=Gene_lengths,
                Vars=Vars, Var_list=Var_list, Links=Links, partial_alleles=partial_alleles, alleles=set(alleles),
                dbversion=dbversion)


def allele_sequence(index, gene, name):
    """Sequence of one allele, built from the backbone and the allele's variants the way
    typing_core.read_Gene_alleles_from_vars does (typing_core.py:2198-2236); `load_index` keeps only names and lengths, the
    self-test loop of genotyping_locus needs the bases of the few alleles it simulates reads from.  Cached in
    index["Genes"]."""
    got = index["Genes"][gene].get(name)
    if got is not None:
        return got
    per = index.get("_allele_vars", {}).get(gene)
    if per is None:
        per = {}
        for _, vid in index["Var_list"].get(gene, []):
            for a in index["Links"].get(vid, []):
                per.setdefault(a, []).append(vid)
        index.setdefault("_allele_vars", {})[gene] = per
    bb = index["Genes"][gene][index["refGenes"][gene]]
    gv = index["Vars"].get(gene, {})
    out, prev = [], 0
    for vid in per.get(name, []):
        t, pos, data = gv[vid]
        assert prev <= pos
        out.append(bb[prev:pos])
        if t == "single":
            out.append(data)
            prev = pos + 1
        elif t == "deletion":
            prev = pos + int(data)
        else:
            assert t == "insertion"
            out.append(data)
            prev = pos
    out.append(bb[prev:])
    seq = "".join(out)
    index["Genes"][gene][name] = seq
    return seq


def _faidx_fetch(fa_path, chrom, left0, right0):
    """`samtools faidx <fa> chrom:left-right` (typing_core.py:2175-2195) from the .fai index: name, length, offset, bases per
    line, bytes per line."""
    entry = None
    with open(fa_path + ".fai") as f:
        for line in f:
            t = line.rstrip("\n").split("\t")
            if t[0] == chrom:
                entry = (int(t[1]), int(t[2]), int(t[3]), int(t[4]))
                break
    if entry is None:
        raise ValueError("%s is not in %s.fai" % (chrom, fa_path))
    length, offset, per_line, line_bytes = entry
    right0 = min(right0, length - 1)
    start = offset + (left0 // per_line) * line_bytes + left0 % per_line
    end = offset + (right0 // per_line) * line_bytes + right0 % per_line + 1
    with open(fa_path, "rb") as f:
        f.seek(start)
        raw = f.read(end - start)
    return raw.replace(b"\n", b"").replace(b"\r", b"").decode()


def load_genome_index(ix_dir, genotype_genome, base_fname):
    """Dict arguments of typing() for a genotype-genome index `<ix_dir>/<genotype_genome>.*` restricted to the database
    `base_fname`, as genotyping_locus assembles them (typing_core.py:2326-2397): `.allele` / `.partial` rows are
    `family<TAB>name`; `.locus` rows carry the family first and genome coordinates; `.snp` positions are chromosome
    coordinates, re-based to the locus (read_Gene_vars_genotype_genome, typing_core.py:2238-2275); the backbone of a
    locus is its span of the genome FASTA (read_backbone_alleles, typing_core.py:2175-2195; here through the .fai index,
    no samtools)."""
    full = os.path.join(ix_dir, genotype_genome)
    alleles, partial_alleles = set(), set()
    for ext, dst in ((".allele", alleles), (".partial", partial_alleles)):
        with open(full + ext) as f:
            for line in f:
                if not line.strip():
                    continue
                family, name = line.strip().split("\t")
                if family == base_fname:
                    dst.add(name)
    refGenes, refGene_loci = read_locus(full + ".locus", True, base_fname)
    by_chr = {}
    for gene, v in refGene_loci.items():
        by_chr.setdefault(v[1], []).append((v[0], v[2], v[3]))
    Vars, Var_list = {}, {}
    with open(full + ".snp") as f:
        for line in f:
            if not line.strip():
                continue
            var_id, var_type, var_chr, pos, data = line.rstrip("\n").split("\t")
            pos = int(pos)
            hit = next(((n, l, r) for n, l, r in by_chr.get(var_chr, []) if l <= pos <= r), None)
            if hit is None:
                continue
            gene = hit[0].split("*")[0]
            gv = Vars.setdefault(gene, {})
            if var_id in gv:
                raise ValueError("duplicate variant id %s" % var_id)
            gv[var_id] = [var_type, pos - hit[1], data]
            Var_list.setdefault(gene, []).append([pos - hit[1], var_id])
    for gene in Var_list:
        Var_list[gene].sort()
    Links = read_links(full + ".link")
    Genes, Gene_names, Gene_lengths = {}, {}, {}
    for gene, (bname, chrom, left, right, _, _) in refGene_loci.items():
        seq = _faidx_fetch(full + ".fa", chrom, left, right)
        if len(seq) != right - left + 1:
            raise ValueError("locus %s reaches past the end of %s" % (gene, chrom))
        Vars.setdefault(gene, {})
        Var_list.setdefault(gene, [])
        per_allele = {}
        for _, vid in Var_list[gene]:
            for a in Links.get(vid, []):
                per_allele.setdefault(a, []).append(vid)
        names = [bname] + list(per_allele.keys())
        if len(names) <= 1:
            names.append("%s*GRCh38" % gene)
        seen = set(names)
        for a in sorted(x for x in alleles if x.split("*")[0] == gene and x not in seen):
            names.append(a)
        Genes[gene] = {n: None for n in names}
        Genes[gene][bname] = seq
        Gene_names[gene] = names
        Gene_lengths[gene] = {n: _allele_length(len(seq), per_allele.get(n, []), Vars[gene]) for n in names}
    dbversion = open(full + ".version").read() if os.path.exists(full + ".version") else "NONE"
    return dict(refGenes=refGenes, refGene_loci=refGene_loci, Genes=Genes, Gene_names=Gene_names, Gene_lengths=Gene_lengths,
                Vars=Vars, Var_list=Var_list, Links=Links, partial_alleles=partial_alleles, alleles=alleles,
                dbversion=dbversion)


_MEMO = {}


def _stamp(ix_dir, stem):
    out = []
    for f in sorted(os.listdir(ix_dir)):
        if (f.startswith(stem + ".") or f.startswith(stem + "_")) and not f.endswith((".hgx.npz", ".report", ".bam", ".sam")):
            st = os.stat(os.path.join(ix_dir, f))
            out.append((f, st.st_mtime_ns, st.st_size))
    return tuple(out)


def load_index_memo(ix_dir, base_fname, genotype_genome=""):
    """load_index / load_genome_index, remembered per process while the index files keep their names, sizes and modification
    times: driver.genotyping_locus is called once per sample on the same index (/root/reference/hisatgenotype:616-665), and the
    SAME dict objects coming back let typing() find its packed loci by identity (locus.LocusCache)."""
    key = (os.path.abspath(ix_dir), base_fname, genotype_genome)
    stamp = _stamp(ix_dir, genotype_genome or base_fname)
    hit = _MEMO.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    ix = load_genome_index(ix_dir, genotype_genome, base_fname) if genotype_genome else load_index(ix_dir, base_fname)
    _MEMO[key] = (stamp, ix)
    return ix


def packed_locus(ix_dir, base_fname, gene, index=None, use_cache=True):
    """PackedLocus of `gene`, through the packed binary cache `<ix_dir>/<base_fname>.<gene>.hgx.npz`: the
    cache is used when it is newer than every text file of the index, rebuilt (and rewritten, best effort) otherwise."""
    from .locus import PackedLocus
    full = os.path.join(ix_dir, base_fname)
    cache = "%s.%s.hgx.npz" % (full, gene)
    sources = [full + ext for ext in (".snp", ".link", ".locus", ".allele", ".partial", "_backbone.fa") if os.path.exists(full + ext)]
    if use_cache and os.path.exists(cache) and all(os.path.getmtime(cache) >= os.path.getmtime(f) for f in sources):
        try:
            return PackedLocus.load_cache(cache)
        except Exception:
            pass                                  # stale or foreign file: rebuild below
    ix = index if index is not None else load_index(ix_dir, base_fname)
    pl = PackedLocus.from_reference_dicts(gene, base_fname, ix["refGenes"], ix["Genes"], ix["Gene_names"], ix["Gene_lengths"],
                                          ix["refGene_loci"], ix["Vars"], ix["Var_list"], ix["Links"])
    if use_cache:
        try:
            pl.save_cache(cache)
        except OSError:
            pass                                  # read-only index directory
    return pl
