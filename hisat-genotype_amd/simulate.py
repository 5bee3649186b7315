"""Read simulation for the self-test loop of genotyping_locus, and the stand-in for the aligner.

  simulate_reads(...)    same signature, same FASTA files, same use of the global `random` stream as
                         typing_common.simulate_reads (typing_common.py:697-983): fragments tiled every
                         `simulate_interval` bases over each allele of the test, read names that spell the true alignment
                         ("<n>|<L/R>_<pos>_<CIGAR>_<gap|T|id,...>"), optional per-base sequencing errors.
  truth_align(...)       SAM records for such reads from their NAMES: HISAT2 (typing_common.align_reads, common:985-1056)
                         is an absent submodule of the reference, so for SIMULATED reads the alignment the read was cut
                         from is written out instead of being searched for.  It is the only place where this package
                         substitutes for a tool of the pipeline, and only in the self-test: real reads need a real aligner.
  align_reads(...)       the reference's align_reads signature: runs `hisat2` with the reference's options when it is on
                         PATH (SAM text straight to `out_fname`; hgx_read_alignments needs no samtools), else truth_align
                         for simulated reads, else NotImplementedError.

Host-side, off the hot path; nothing here touches the GPU.
"""
import errno
import os
import random
import shutil
import subprocess
import sys

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


class _AlleleLayout:
    """An allele laid over its backbone (typing_common.py:878-935): `ex_seq` is the backbone with the allele's singles
    substituted, deleted bases marked 'D' and inserted bases marked 'I'; `ex_desc` names the variant under every such
    position; `to_ex[i]` / `to_backbone[i]` map allele base i to its `ex_seq` index / backbone coordinate."""

    def __init__(self, allele_name, allele_seq, backbone_seq, gene_vars, Links):
        ids = [v for v, carriers in Links.items() if allele_name in carriers]
        for v in ids:
            assert v.startswith("hv")
        ids.sort(key=lambda v: int(v[2:]))
        ex_seq, ex_desc = list(backbone_seq), [""] * len(backbone_seq)
        shift = 0
        for v in ids:
            vtype, vpos, vdata = gene_vars[v]
            vpos += shift
            if vtype == "single":
                ex_seq[vpos], ex_desc[vpos] = vdata, v
            elif vtype == "deletion":
                n = int(vdata)
                assert vpos + n <= len(ex_seq)
                ex_seq[vpos:vpos + n] = ["D"] * n
                ex_desc[vpos:vpos + n] = [v] * n
            else:
                assert vtype == "insertion"
                n = len(vdata)
                ex_seq[vpos:vpos] = ["I"] * n
                ex_desc[vpos:vpos] = [v] * n
                shift += n
        assert len(backbone_seq) + shift == len(ex_seq)
        self.ex_seq, self.ex_desc = "".join(ex_seq), ex_desc
        self.to_ex, self.to_backbone = [0] * len(allele_seq), [0] * len(allele_seq)
        j, inserted = 0, 0
        for i in range(len(allele_seq)):
            while self.ex_seq[j] == "D":
                j += 1
            if self.ex_seq[j] == "I":
                inserted += 1
            self.to_ex[i], self.to_backbone[i] = j, j - inserted
            j += 1

    def name_of_read(self, read_seq, pos):
        """"<pos>_<CIGAR>[_<items>]" of allele bases [pos, pos + len(read_seq)) (typing_common.py:772-843).  Inserted bases
        count as M; a known insertion is reported when the first base after it is reached; a base that differs from the
        allele (a sequencing error) is the bare item "unknown"."""
        ex_seq, ex_desc = self.ex_seq, self.ex_desc
        cigar, items = "", []
        run = gap = 0
        open_ins = ""
        last = len(read_seq) - 1
        for k, base in enumerate(read_seq):
            j = self.to_ex[pos + k]
            sym = ex_seq[j]
            assert sym != "D"
            run += 1
            if sym == "I":
                assert open_ins in ("", ex_desc[j])
                open_ins = ex_desc[j]
            else:
                if open_ins:
                    items.append("%s|I|%s" % (gap, open_ins))
                    open_ins, gap = "", 0
                if ex_desc[j] != "" or base != sym:
                    items.append("%d|S|%s" % (gap, ex_desc[j]) if ex_desc[j] != "" else "unknown")
                    gap = 0
                else:
                    gap += 1
            if k < last and ex_seq[j + 1] == "D":
                n = 1
                while j + 1 + n < len(ex_seq) and ex_seq[j + 1 + n] == "D":
                    n += 1
                cigar += "%dM%dD" % (run, n)
                run = 0
                items.append("%s|D|%s" % (gap, ex_desc[j + 1]))
                gap = 0
        assert run > 0
        info = "%d_%s%dM" % (self.to_backbone[pos] + 1, cigar, run)
        if items:
            info += "_" + ",".join(items)
        return info


def _other_bases(base):
    assert base in "ACGT"
    return [b for b in "ACGT" if b != base]


def _mkdir_p(path):
    try:
        os.makedirs(path)
    except OSError as exc:
        if not (exc.errno == errno.EEXIST and os.path.isdir(path)):
            raise


def simulate_reads(seq_dic, base_fname, allele_list, Vars, Links, simulate_interval=1, read_len=100, frag_len=250,
                   perbase_errorrate=0.0, perbase_snprate=0.0, skip_fragment_regions=[], out_dir=".", test_i=0):
    """Write `<base_fname>_input_{1,2}.fa` (current directory, and a copy under
    `<out_dir>/dir_<gene>/dir_test-<test_i>_<alleles>/`) for every allele group of `allele_list`; returns the number of
    pairs per allele, `[[n, ...] per group]`.  Draws from the module-level `random` generator in the reference's order, so a
    caller that seeded it gets the reference's reads.  As in the reference the read lists are NOT reset between groups:
    each group's files also hold the earlier groups' reads, and the files left behind are those of the last group."""
    reads_1, reads_2, num_pairs = [], [], []
    for allele_names in allele_list:
        gene = allele_names[0].split("*")[0]
        num_pairs.append([])
        for allele_name in allele_names:
            allele_seq = seq_dic[gene][allele_name]
            if perbase_snprate > 0:
                # the reference mutates a COPY it never reads again (typing_common.py:890-891): only the draws matter
                for base in allele_seq:
                    if random.random() * 100 < perbase_snprate:
                        random.shuffle(_other_bases(base))
            lay = _AlleleLayout(allele_name, allele_seq, seq_dic[gene]["%s*BACKBONE" % gene], Vars[gene], Links)

            def cut(pos):
                seq = allele_seq[pos:pos + read_len]
                if perbase_errorrate > 0.0:
                    out = list(seq)
                    for k in range(read_len):
                        if lay.ex_desc[lay.to_backbone[pos + k]] != "":     # (backbone coordinate into the ex_seq annotation: sic)
                            continue
                        if random.random() * 100 < perbase_errorrate:
                            alts = _other_bases(out[k])
                            random.shuffle(alts)
                            out[k] = alts[0]
                    seq = "".join(out)
                return seq, lay.name_of_read(seq, pos)

            n_here = 0
            for start in range(0, len(allele_seq) - frag_len + 1, simulate_interval):
                if any(start <= right and start + frag_len > left for left, right in skip_fragment_regions):
                    continue
                seq1, info1 = cut(start)
                seq2, info2 = cut(start + frag_len - read_len)
                reads_1.append([seq1, info1])
                reads_2.append(["".join(_COMP.get(b, b) for b in reversed(seq2)), info2])
                n_here += 1
            num_pairs[-1].append(n_here)
        ident = "_".join(allele_names).replace("*", "-")
        read_dir = "%s/dir_%s/dir_test-%d_%s" % (out_dir, gene, test_i, ident)
        _mkdir_p(read_dir)
        for idx, reads in ((1, reads_1), (2, reads_2)):
            fname = "%s_input_%d.fa" % (base_fname, idx)
            text = "".join(">%s\n%s\n" % (("%d|%s_%s" % (n + 1, "LR"[idx - 1], info))[:251], seq)
                           for n, (seq, info) in enumerate(reads))
            for path in (fname, "%s/%s" % (read_dir, fname)):
                with open(path, "w") as f:
                    f.write(text)
    return num_pairs


# ------------------------------------------------------------------------------------------------------------------
# the aligner's stand-in for simulated reads
# ------------------------------------------------------------------------------------------------------------------
def _read_fasta(path):
    name, out = None, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                name = line[1:]
            elif name is not None:
                out.append((name, line))
                name = None
    return out


def _truth_record(qname, seq, Genes, Vars, refGenes):
    """(gene, pos0, cigar, md, zs, nm) of a simulated read from its name: the named variants sit at their database
    positions, everything else is M; bases that differ from the backbone without a named single are edits (NM)."""
    info = qname.split("|", 1)[1].split("_", 1)[1]            # "<pos>_<CIGAR>[_items]"
    f = info.split("_", 2)
    pos0 = int(f[0]) - 1
    named = []
    if len(f) > 2:
        for item in f[2].split(","):
            t = item.split("|")
            if len(t) == 3 and t[2].startswith("hv"):
                named.append(t[2])
    genes = [g for g in Genes if named and named[0] in Vars.get(g, {})] or list(Genes)
    best = None
    for gene in genes:
        bb = Genes[gene][refGenes[gene]]
        gv = Vars.get(gene, {})
        if any(v not in gv for v in named):
            continue
        events = {}
        for v in named:
            vtype, vpos, vdata = gv[v]
            if vtype != "single":
                events.setdefault(vpos, []).append((vtype, vdata, v))
        singles = {gv[v][1]: v for v in named if gv[v][0] == "single"}
        cigar, md, zs = [], [], []
        nm = md_run = gap = 0
        r, p, ok = 0, pos0, True

        def push(op, n):
            if cigar and cigar[-1][0] == op:
                cigar[-1][1] += n
            else:
                cigar.append([op, n])

        while r < len(seq):
            for vtype, vdata, v in events.pop(p, []) if r > 0 else []:
                if vtype == "deletion":
                    n = int(vdata)
                    md.append("%d^%s" % (md_run, bb[p:p + n]))
                    md_run = 0
                    push("D", n)
                    zs.append("%d|D|%s" % (gap, v))
                    gap = 0
                    p += n
                else:                                           # known insertion in front of backbone base p
                    n = min(len(vdata), len(seq) - r)
                    zs.append("%d|I|%s" % (gap, v))
                    gap = n                                     # the consumer does not advance Zs_pos over inserted bases
                    push("I", n)
                    r += n
            if r >= len(seq):
                break
            if p >= len(bb):
                ok = False
                break
            if seq[r] == bb[p]:
                md_run += 1
                gap += 1
            else:
                md.append("%d%s" % (md_run, bb[p]))
                md_run = 0
                if p in singles:
                    zs.append("%d|S|%s" % (gap, singles[p]))
                    gap = 0
                else:
                    nm += 1
                    gap += 1
            push("M", 1)
            r += 1
            p += 1
        if not ok:
            continue
        md.append("%d" % md_run)
        cand = (nm, gene, pos0, "".join("%d%s" % (n, op) for op, n in cigar), "".join(md), ",".join(zs))
        if best is None or cand[0] < best[0]:
            best = cand
    if best is None:
        raise ValueError("simulated read %r fits no locus" % qname)
    nm, gene, pos0, cigar, md, zs = best
    return gene, pos0, cigar, md, zs, nm


def truth_align(read_fname, out_fname, Genes, Vars, refGenes):
    """Write the SAM text (header + one record per read, input order) for simulated reads `read_fname` = [mate-1 FASTA,
    mate-2 FASTA] or [single-end FASTA] to `out_fname`.  Dialect = what the reference's loop reads from HISAT2's graph
    alignments: FLAG 99 / 147 (0 single-end), MD, NM = edits not explained by a known variant, Zs = known variants,
    NH:i:1, YT:Z:CP (YT:Z:UU single-end)."""
    mates = [_read_fasta(p) for p in read_fname]
    paired = len(mates) == 2
    lines = ["@SQ\tSN:%s\tLN:%d" % (refGenes[g], len(Genes[g][refGenes[g]])) for g in Genes]
    for k in range(len(mates[0])):
        recs = []
        for m, reads in enumerate(mates):
            qname, seq = reads[k]
            if m == 1:
                seq = "".join(_COMP.get(b, b) for b in reversed(seq))          # back on the forward strand, as SAM stores it
            recs.append((qname, seq) + _truth_record(qname, seq, Genes, Vars, refGenes))
        for m, (qname, seq, gene, pos0, cigar, md, zs, nm) in enumerate(recs):
            flag = (99 if m == 0 else 147) if paired else 0
            mate_pos = recs[1 - m][3] if paired else pos0
            tags = ["NM:i:%d" % nm, "MD:Z:%s" % md] + (["Zs:Z:%s" % zs] if zs else []) + ["NH:i:1", "YT:Z:%s" % ("CP" if paired else "UU")]
            lines.append("\t".join([qname, str(flag), refGenes[gene], str(pos0 + 1), "60", cigar, "=" if paired else "*",
                                    str(mate_pos + 1) if paired else "0", "0", seq, "I" * len(seq)] + tags))
    _store_as_the_reference_does(out_fname, "\n".join(lines) + "\n")


def _store_as_the_reference_does(out_fname, sam_text):
    """The reference pipes the aligner through `samtools view -bS - | samtools sort` (typing_common.py:1038-1051): the
    alignment file it later reads (`samtools view F | sort -k1,1 -s`, typing_core.py:458-468) is a COORDINATE-sorted BAM, so the
    records of one read name -- mates, secondary hits -- reach the decode loop in coordinate order, not in the aligner's
    (ADVICE r2).  Same here: the SAM text becomes a coordinate-sorted BAM (libhgx's writer; no samtools)."""
    from . import bamio
    refs = []
    for line in sam_text.split("\n"):
        if not line.startswith("@"):
            break
        if line.startswith("@SQ"):
            f = dict(t.split(":", 1) for t in line.split("\t")[1:] if ":" in t)
            refs.append((f["SN"], int(f["LN"])))
    body = "\n".join(l for l in sam_text.split("\n") if l and not l.startswith("@")) + "\n"
    bamio.write_bam_native(out_fname, body, refs, sort_by_coordinate=True)


def align_reads(aligner, simulation, index_name, index_type, base_fname, read_fname, fastq, threads, out_fname, verbose,
                truth=None):
    """typing_common.align_reads (common:985-1056).  With the aligner on PATH: the reference's command line, SAM text
    written to `out_fname` (no samtools: hgx_read_alignments reads SAM text and groups by name itself).  Without it:
    simulated reads carry their alignment in their names and `truth` = (Genes, Vars, refGenes) lets truth_align write it
    out; real reads cannot be aligned here."""
    if shutil.which(aligner):
        if aligner == "hisat2":
            cmd = [aligner, "--mm"] + ([] if simulation else ["--no-unal"]) + ["--no-spliced-alignment", "-X", "1000"]
            if index_type == "linear":
                cmd += ["-k", "10"]
            else:
                cmd += ["--max-altstried", "64", "--haplotype"]
                if base_fname == "codis":
                    cmd += ["--enable-codis", "--no-softclip"]
        elif aligner == "bowtie2":
            cmd = [aligner, "--no-unal", "-k", "10"]
        else:
            raise AssertionError(aligner)
        cmd += ["-x", index_name, "-p", str(threads)] + ([] if fastq else ["-f"])
        assert len(read_fname) in (1, 2)
        cmd += ["-U", read_fname[0]] if len(read_fname) == 1 else ["-1", read_fname[0], "-2", read_fname[1]]
        if verbose >= 1:
            print(" ".join(cmd), file=sys.stderr)
        with open(out_fname + ".sam", "w") as out, open(os.devnull, "w") as null:
            subprocess.check_call(cmd, stdout=out, stderr=null)
        with open(out_fname + ".sam") as f:
            _store_as_the_reference_does(out_fname, f.read())
        os.remove(out_fname + ".sam")
        return
    if simulation and truth is not None:
        truth_align(read_fname, out_fname, *truth)
        return
    raise NotImplementedError("%s is not installed (an absent submodule of the reference): aligning real reads is outside "
                              "the accelerated path -- pass alignment_fname" % aligner)
