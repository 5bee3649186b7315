"""N repeated typing() calls on the configs[0] fixture (the drop-in's steady state), for rocprofv3 --kernel-trace --stats."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
import hisatgenotype_amd as hgx
name = sys.argv[2] if len(sys.argv) > 2 else "hla_7000_10k"
fx = gu.load(name)
loc, o = fx["_locus"], fx["options"]
d = loc.reference_dicts()
tmp = tempfile.mkdtemp(dir="/dev/shm")
sam = os.path.join(tmp, "sample.sam")
open(sam, "w").write(fx["sam"])
def call():
    hgx.typing(False, os.path.join(tmp, loc.base_fname), [loc.gene], "", True, set(), d["refGenes"], d["Genes"], d["Gene_names"], d["Gene_lengths"],
               d["refGene_loci"], d["Vars"], d["Var_list"], d["Links"], [["hisat2", "graph"]], o["num_editdist"], False, "assembly_graph",
               o["error_correction"], True, o["allow_discordant"], False, o["remove_low"], [], False, ["sample.fq"], sam, [], o["read_len"], o["frag_len"], 1,
               False, 0, False, tmp, "NONE", True, 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    call()
t0 = time.perf_counter()
for _ in range(n):
    call()
print("%s: %.3f ms per typing() call over %d calls (3 warm-up calls before)" % (name, (time.perf_counter() - t0) / n * 1e3, n))
