#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (``/root/reference``) in the build container.

Nothing of the reference is stored in this repo: the script imports it from a scratch copy
(needed only to add the missing ``hisat2/VERSION`` file, typing_core.py:312), feeds
``typing()`` (typing_core.py:249) synthetic loci + SAM written by ``hisatgenotype_amd.synth``
through a stub ``samtools`` (``view F ..`` = the SAM body, ``index`` = touch), and records what
the reference computed:

  G3  per kept record: the ``cmp_list2`` given to ``identify_ambigious_diffs`` and its result
      (typing_common.py:1663-1955), plus every ``error_correct`` result (typing_core.py:119-243)
  G4  ``get_alternatives`` tables (typing_common.py:1424-1657)
  G5  ``get_mpileup`` nt_sets / counts (typing_common.py:1059-1184)
  G6  per pair: every ``add_count(count_per_read, ht, 1)`` call by level (typing_core.py:626-677,
      captured with ``sys.setprofile`` because the closure cannot be imported) and the class key
      each ``add_stat`` returned (typing_core.py:1171-1236)
  G7  every ``single_abundance`` call: ordered class dict, flags, result at full repr precision and
      the number of outer iterations (typing_common.py:1282-1410)
  G8  the report text (typing_core.py:1593, 1659-1672, 2097-2111)

Fixtures are data only (inputs + expected outputs), written as ``tests/golden/<name>.json.gz``.
Run:  PYTHONHASHSEED=0 python tests/golden/make_golden.py [scenario ...]
"""
import copy
import gzip
import json
import os
import shutil
import stat
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"
    sys.exit(subprocess.call([sys.executable] + sys.argv, env=os.environ))

sys.path.insert(0, ROOT)
import hisatgenotype_amd  # noqa: E402
from hisatgenotype_amd import synth  # noqa: E402


def setup_reference():
    tmp = tempfile.mkdtemp(prefix="hgref_")
    dst = os.path.join(tmp, "hgref")
    shutil.copytree(REF, dst, symlinks=True)
    os.makedirs(os.path.join(dst, "hisat2"), exist_ok=True)
    with open(os.path.join(dst, "hisat2", "VERSION"), "w") as f:
        f.write("2.2.1\n")
    bindir = os.path.join(tmp, "bin")
    os.makedirs(bindir)
    stub = os.path.join(bindir, "samtools")
    with open(stub, "w") as f:
        f.write("#!/bin/sh\n"
                "case \"$1\" in\n"
                "  view) grep -v '^@' \"$2\" ;;\n"
                "  index) touch \"$2.bai\" ;;\n"
                "  *) exit 1 ;;\n"
                "esac\n")
    os.chmod(stub, os.stat(stub).st_mode | stat.S_IEXEC)
    os.environ["PATH"] = bindir + os.pathsep + os.environ["PATH"]
    os.environ["LC_ALL"] = "C"
    sys.path.insert(0, os.path.join(dst, "hisatgenotype_modules"))
    return tmp


class Capture:
    """Profile hook + wrappers recording what the reference computes."""

    def __init__(self, core, common):
        self.core, self.common = core, common
        self.records = []      # G3
        self.ec = []           # error_correct results
        self.pairs = []        # G6
        self.cur = None
        self.em = []           # G7
        self.alts = None
        self.mpileup = None
        self._n_diff = 0

    # ---- sys.setprofile hook for the closures of typing() --------------------------------
    def profile(self, frame, event, arg):
        co = frame.f_code
        if not co.co_filename.endswith("hisatgenotype_typing_core.py"):
            return
        name = co.co_name
        if name == "add_count" and event == "call":
            caller = frame.f_back.f_locals
            cpr = frame.f_locals["count_per_read"]
            if cpr is caller.get("Gene_exons_count_per_read"):
                lvl = "exon"
            elif cpr is caller.get("Gene_count_per_read"):
                lvl = "gene"
            else:
                lvl = "primary"
            if self.cur is None:
                self.cur = {"exon": [], "gene": [], "primary": [], "cls": {}}
            self.cur[lvl].append(frame.f_locals["ht"])
        elif name == "add_stat" and event == "return":
            caller = frame.f_back.f_locals
            gc = frame.f_locals["Gene_cmpt"]
            if gc is caller.get("Gene_exons_cmpt"):
                lvl = "exon"
            elif gc is caller.get("Gene_cmpt"):
                lvl = "gene"
            else:
                lvl = "primary"
            if self.cur is None:
                self.cur = {"exon": [], "gene": [], "primary": [], "cls": {}}
            self.cur["cls"][lvl] = arg
            if lvl == "gene":      # gene-level add_stat is the last call of every flush
                self.pairs.append(self.cur)
                self.cur = None

    def install(self):
        core, common = self.core, self.common
        cap = self
        orig_iad = common.identify_ambigious_diffs
        orig_em = common.single_abundance
        orig_pd = common.prob_diff
        orig_alt = common.get_alternatives
        orig_mp = common.get_mpileup
        orig_ec = core.error_correct

        def iad(ref_seq, Vars, Al, Ar, All, Arl, cmp_list, verbose, debug=False):
            res = orig_iad(ref_seq, Vars, Al, Ar, All, Arl, cmp_list, verbose, debug)
            cap.records.append({"cmp": copy.deepcopy(cmp_list),
                                "iad": [res[0], res[1], sorted(res[2]), sorted(res[3])]})
            return res

        def pd(a, b):
            cap._n_diff += 1
            return orig_pd(a, b)

        def em(Gene_cmpt, remove_low_abundance_allele=False, Gene_length={}):
            cap._n_diff = 0
            res = orig_em(Gene_cmpt, remove_low_abundance_allele, Gene_length)
            cap.em.append({"cmpt": [[k, v] for k, v in Gene_cmpt.items()],
                           "remove_low": bool(remove_low_abundance_allele),
                           "use_length": len(Gene_length) > 0,
                           "result": [[a, repr(p)] for a, p in res],
                           "n_iter": cap._n_diff})
            return res

        def alt(ref_seq, allele_vars, Vars, Var_list, verbose):
            l, r = orig_alt(ref_seq, allele_vars, Vars, Var_list, verbose)
            cap.alts = {"left": {k: sorted(v) for k, v in l.items()},
                        "right": {k: sorted(v) for k, v in r.items()}}
            return l, r

        def mp(cmd, ref_seq, base_locus, vars, allow_discordant):
            res = orig_mp(cmd, ref_seq, base_locus, vars, allow_discordant)
            cap.mpileup = {"nt_set": ["".join(sorted(x[0])) for x in res],
                           "counts": [{nt: v[0] for nt, v in x[1].items()} for x in res]}
            return res

        def ec(ref_seq, read_seq, read_pos, mpileup, Vars, Var_list, cmp_list, debug=False):
            res = orig_ec(ref_seq, read_seq, read_pos, mpileup, Vars, Var_list, cmp_list, debug)
            cap.ec.append({"cmp": copy.deepcopy(res[0]), "n": res[2]})
            return res

        common.identify_ambigious_diffs = iad
        common.single_abundance = em
        common.prob_diff = pd
        common.get_alternatives = alt
        common.get_mpileup = mp
        core.error_correct = ec
        self._restore = lambda: (setattr(common, "identify_ambigious_diffs", orig_iad),
                                 setattr(common, "single_abundance", orig_em),
                                 setattr(common, "prob_diff", orig_pd),
                                 setattr(common, "get_alternatives", orig_alt),
                                 setattr(common, "get_mpileup", orig_mp),
                                 setattr(core, "error_correct", orig_ec))

    def restore(self):
        self._restore()


def run_reference(core, common, locus, sam, *, simulation, sample, num_editdist=2, error_correction=True,
                  allow_discordant=False, remove_low=True, read_len=150, frag_len=400, workdir,
                  profile_closures=True):
    d = locus.reference_dicts()
    gene = locus.gene
    base = locus.base_fname
    os.makedirs(workdir, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(workdir)
    bam = os.path.join(workdir, "syn.bam")
    with open(bam, "w") as f:
        f.write(sam)
    for ext in (".bai",):
        if os.path.exists(bam + ext):
            os.remove(bam + ext)
    cap = Capture(core, common)
    cap.install()
    if simulation:
        locus_list = [list(sample)]
    else:
        locus_list = [gene]
    err = None
    stderr_fd = os.dup(2)
    devnull = os.open(os.devnull, os.O_WRONLY)
    try:
        os.dup2(devnull, 2)
        if profile_closures:
            sys.setprofile(cap.profile)
        try:
            core.typing(simulation, os.path.join(workdir, base), locus_list, "", True, set(), d["refGenes"],
                        d["Genes"], d["Gene_names"], d["Gene_lengths"], d["refGene_loci"], d["Vars"],
                        d["Var_list"], d["Links"], [["hisat2", "graph"]], num_editdist, False,
                        "assembly_graph", error_correction, True, allow_discordant, False, remove_low, [],
                        False, ["reads_1.fa", "reads_2.fa"], bam, [], read_len, frag_len, 1, False, 0, False,
                        workdir, "NONE", True, 0)
        except SystemExit as e:      # the reference exit(1)s on violated sanity checks
            err = "SystemExit(%s)" % (e.code,)
        except Exception as e:       # e.g. quirk Q3 (TypeError) -- part of the observable behaviour
            err = "%s: %s" % (type(e).__name__, e)
        finally:
            sys.setprofile(None)
    finally:
        os.dup2(stderr_fd, 2)
        os.close(devnull)
        os.close(stderr_fd)
        cap.restore()
        os.chdir(cwd)
    rep = [f for f in os.listdir(workdir) if f.endswith(".report")]
    report = open(os.path.join(workdir, rep[0])).read() if rep else ""
    # drop the header lines carrying paths/versions (typing_core.py:315-325)
    body = report.split("\n")
    try:
        k = next(i for i, l in enumerate(body) if l.startswith("\t\thisat2"))
        report = "\n".join(body[k:])
    except StopIteration:
        pass
    return cap, report, err


def encode_classes(locus, keys):
    """Class keys are huge ('-'.join of up to 7 000 names): store each distinct key once as a hex bitset
    over Gene_names order."""
    idx = {n: i for i, n in enumerate(locus.allele_names)}
    table, ids = {}, []
    for k in keys:
        if k not in table:
            table[k] = len(table)
        ids.append(table[k])
    enc = []
    for k in table:
        bits = 0
        if k:
            for n in k.split("-"):
                bits |= 1 << idx[n]
        enc.append("%x" % bits)
    return enc, ids


SCENARIOS = {}


def scenario(fn):
    SCENARIOS[fn.__name__] = fn
    return fn


@scenario
def hla_small_pair():
    loc = synth.make_hla_like_locus(n_alleles=48, n_vars=260, seed=11, sibling_frac=0.4)
    sample = synth.pick_sample(loc, 5)
    al = synth.simulate_pairs(loc, sample, 0, read_len=100, frag_len=(350, 350), seed=3,
                              simulation_names=True, tile_interval=23)
    return dict(locus=loc, sample=sample, al=al, simulation=True, read_len=100, frag_len=350)


@scenario
def hla_small_single():
    loc = synth.make_hla_like_locus(n_alleles=64, n_vars=300, seed=12, sibling_frac=0.3)
    sample = synth.pick_sample(loc, 9, n=1)
    al = synth.simulate_pairs(loc, sample, 0, read_len=100, frag_len=(350, 350), seed=4,
                              simulation_names=True, tile_interval=17)
    return dict(locus=loc, sample=sample, al=al, simulation=True, read_len=100, frag_len=350)


@scenario
def hla_errors_filters():
    loc = synth.make_hla_like_locus(n_alleles=150, n_vars=500, seed=13, sibling_frac=0.35, unlinked_vars=4,
                                    deletion_frac=0.12)
    sample = synth.pick_sample(loc, 21)
    al = synth.simulate_pairs(loc, sample, 700, read_len=150, frag_len=(350, 450), err_rate=0.005, seed=5,
                              softclip_frac=0.05, novel_del_frac=0.03, multi_hit_frac=0.02,
                              discordant_frac=0.02, unaligned_frac=0.01, dup_frac=0.02)
    return dict(locus=loc, sample=sample, al=al, simulation=False)


@scenario
def hla_mid_real():
    loc = synth.make_hla_like_locus(n_alleles=512, n_vars=1200, seed=14, sibling_frac=0.4)
    sample = synth.pick_sample(loc, 33)
    al = synth.simulate_pairs(loc, sample, 600, read_len=150, frag_len=(350, 450), err_rate=0.002, seed=6)
    return dict(locus=loc, sample=sample, al=al, simulation=False)


@scenario
def hla_keep_low():
    loc = synth.make_hla_like_locus(n_alleles=256, n_vars=800, seed=15, sibling_frac=0.4)
    sample = synth.pick_sample(loc, 34)
    al = synth.simulate_pairs(loc, sample, 400, read_len=150, frag_len=(400, 400), err_rate=0.0, seed=7)
    return dict(locus=loc, sample=sample, al=al, simulation=False, remove_low=False, error_correction=False)


@scenario
def hla_single_end():
    loc = synth.make_hla_like_locus(n_alleles=100, n_vars=400, seed=16)
    sample = synth.pick_sample(loc, 35)
    al = synth.simulate_pairs(loc, sample, 300, read_len=150, err_rate=0.003, seed=8, single_end=True)
    return dict(locus=loc, sample=sample, al=al, simulation=False, allow_discordant=True)


@scenario
def hla_novel_sample():
    """Reads from a recombinant allele absent from the database: many pairs are compatible with
    no allele, which the reference turns into a class of ALL alleles (quirk Q4, core:1177-1190)."""
    loc = synth.make_hla_like_locus(n_alleles=80, n_vars=420, seed=17)
    a, b = synth.pick_sample(loc, 36)
    hyb = copy.deepcopy(loc)
    name = "A*98:01:01:01"
    va = [v for v in loc.allele_vars[a] if loc.var_pos[v] < 1700]
    vb = [v for v in loc.allele_vars[b] if loc.var_pos[v] >= 1700]
    hyb.allele_vars[name] = sorted(va + vb)
    al = synth.simulate_pairs(hyb, [name], 250, read_len=150, frag_len=(380, 420), seed=9)
    return dict(locus=loc, sample=[a, b], al=al, simulation=False)


@scenario
def hla_insertions():
    """Known insertions (CIGAR I + Zs gap|I|id), novel insertions and deletions, soft clips."""
    loc = synth.make_hla_like_locus(n_alleles=120, n_vars=500, seed=31, insertion_frac=0.08, deletion_frac=0.1)
    sample = synth.pick_sample(loc, 3)
    al = synth.simulate_pairs(loc, sample, 500, err_rate=0.003, seed=4, novel_ins_frac=0.05, novel_del_frac=0.03,
                              softclip_frac=0.05)
    return dict(locus=loc, sample=sample, al=al, simulation=False)


@scenario
def hla_7000():
    loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
    sample = synth.pick_sample(loc, 101)
    al = synth.simulate_pairs(loc, sample, 300, read_len=150, frag_len=(350, 450), err_rate=0.001, seed=100)
    return dict(locus=loc, sample=sample, al=al, simulation=False, store_locus=False,
                locus_params=dict(n_alleles=7000, n_vars=2500, seed=101))


@scenario
def hla_7000_10k():
    """BASELINE.json configs[0]: HLA-A-like, ~7 000 alleles, 10 k simulated 2x150 bp reads through the reference's own
    CPU path.  Lean fixture (SAM, EM calls with their class dicts, report; no per-record traces) + the reference's wall
    time on this container's CPU: the true-reference baseline of BASELINE.md."""
    loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
    sample = synth.pick_sample(loc, 101)
    sam = synth.simulate_sam_fast(loc, sample, 5000, err_rate=0.002, seed=100)
    return dict(locus=loc, sample=sample, sam=sam, simulation=False, store_locus=False, lean=True,
                locus_params=dict(n_alleles=7000, n_vars=2500, seed=101))


@scenario
def codis_10k():
    """BASELINE.json configs[4] shape through the reference's own CPU path, timed: one CODIS STR locus (D8S1179-like ladder),
    10 k simulated 2x100 bp reads.  Lean fixture (SAM, EM call with its class dict, report) + the reference's wall time."""
    loc = synth.make_str_like_locus(gene="D8S1179", unit="TCTA", max_repeats=19, min_repeats=7, flank=200, seed=908)
    sample = ["D8S1179*10", "D8S1179*14"]
    sam = synth.simulate_sam_fast(loc, sample, 5000, read_len=100, frag_len=(230, 270), err_rate=0.002, seed=48)
    return dict(locus=loc, sample=sample, sam=sam, simulation=False, lean=True, read_len=100, frag_len=250)


@scenario
def codis_like():
    loc = synth.make_str_like_locus()
    sample = ["D8S1179*10", "D8S1179*13"]
    al = synth.simulate_pairs(loc, sample, 0, read_len=100, frag_len=(250, 250), seed=10,
                              simulation_names=True, tile_interval=3)
    return dict(locus=loc, sample=sample, al=al, simulation=True, read_len=100, frag_len=250)


@scenario
def codis_d18s51():
    """CODIS locus D18S51: the only place choose_pairs / get_pair_interdist act (final flush, core:1547-1552).
    The tiling makes the name-sorted LAST pair ("9|...") end inside the repeat, so it has 14 alternative haplotypes
    that choose_pairs cuts down to the pair matching the median inner distance."""
    loc = synth.make_str_like_locus(gene="D18S51", unit="AGAA", max_repeats=22, min_repeats=9, flank=180, seed=2)
    sample = ["D18S51*12", "D18S51*17"]
    al = synth.simulate_pairs(loc, sample, 0, read_len=100, frag_len=(250, 250), seed=2, simulation_names=True,
                              tile_interval=12)
    return dict(locus=loc, sample=sample, al=al, simulation=True, read_len=100, frag_len=250)


def main():
    names = sys.argv[1:] or list(SCENARIOS)
    tmp = setup_reference()
    import hisatgenotype_typing_common as common
    import hisatgenotype_typing_core as core
    try:
        for name in names:
            sc = SCENARIOS[name]()
            loc = sc["locus"]
            sam = sc["sam"] if "sam" in sc else synth.sam_text(loc, sc["al"])
            lean = bool(sc.get("lean"))
            t_ref = time.perf_counter()
            cap, report, err = run_reference(
                core, common, loc, sam, simulation=sc["simulation"], sample=sc["sample"],
                error_correction=sc.get("error_correction", True),
                allow_discordant=sc.get("allow_discordant", False), remove_low=sc.get("remove_low", True),
                read_len=sc.get("read_len", 150), frag_len=sc.get("frag_len", 400),
                workdir=os.path.join(tmp, "work_" + name), profile_closures=not lean)
            t_ref = time.perf_counter() - t_ref
            if lean:                         # keep the EM calls and the report only
                cap.pairs, cap.records, cap.ec, cap.alts, cap.mpileup = [], [], [], None, None
            cls_keys = []
            for p in cap.pairs:
                cls_keys += [p["cls"].get("exon", ""), p["cls"].get("gene", "")]
            for e in cap.em:
                cls_keys += [k for k, _ in e["cmpt"]]
            enc, ids = encode_classes(loc, cls_keys)
            it = iter(ids)
            pairs = []
            for p in cap.pairs:
                pairs.append({"exon": sorted(p["exon"]), "gene": sorted(p["gene"]),
                              "primary": sorted(p["primary"]), "exon_cls": next(it), "gene_cls": next(it)})
            ems = []
            for e in cap.em:
                ems.append({"cmpt": [[next(it), v] for _, v in e["cmpt"]], "remove_low": e["remove_low"],
                            "use_length": e["use_length"], "result": e["result"], "n_iter": e["n_iter"]})
            fx = {
                "name": name,
                "options": {"simulation": sc["simulation"], "sample": sc["sample"],
                            "error_correction": sc.get("error_correction", True),
                            "allow_discordant": sc.get("allow_discordant", False),
                            "remove_low": sc.get("remove_low", True), "num_editdist": 2,
                            "read_len": sc.get("read_len", 150), "frag_len": sc.get("frag_len", 400)},
                "locus": loc.to_json() if sc.get("store_locus", True) else None,
                "locus_params": sc.get("locus_params"),
                "allele_names": loc.allele_names,
                "sam": sam,
                "error": err,
                "alts": cap.alts,
                "mpileup": cap.mpileup,
                "records": cap.records,
                "error_correct": cap.ec,
                "classes": enc,
                "pairs": pairs,
                "em": ems,
                "report": report,
            }
            if lean:
                n_rec = sum(1 for l in sam.split("\n") if l)
                cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
                fx["reference_timing"] = {
                    "what": "hisatgenotype_typing_core.typing() on this SAM (samtools stub = grep), one process, wrappers on "
                            "single_abundance / get_mpileup / identify_ambigious_diffs / error_correct only (no profile hook)",
                    "seconds": round(t_ref, 2), "sam_records": n_rec, "records_per_s": round(n_rec / t_ref, 1),
                    "cpu": cpu[0] if cpu else "unknown", "cores_used": 1, "python": sys.version.split()[0]}
                print("   reference wall time %.1f s for %d records = %.1f records/s on %s" % (
                    t_ref, n_rec, n_rec / t_ref, fx["reference_timing"]["cpu"]))
            out = os.path.join(HERE, name + ".json.gz")
            with gzip.GzipFile(out, "wb", mtime=0) as f:
                f.write(json.dumps(fx, separators=(",", ":")).encode())
            print("%-22s records=%d pairs=%d em_calls=%d classes=%d err=%s size=%.1f KB" % (
                name, len(cap.records), len(pairs), len(ems), len(enc), err, os.path.getsize(out) / 1024.0))
            first = [l for l in report.split("\n") if "aligned" in l or "ranked" in l][:6]
            print("   " + "\n   ".join(l.strip() for l in first))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
