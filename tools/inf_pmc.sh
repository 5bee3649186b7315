cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/infpmc
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ[C]*_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/infpmc/avail.txt
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/infpmc/$n -o t -- python3 tools/inflate_probe.py run > gpurun_out/infpmc/$n.log 2>&1
  python3 - "$n" <<'P'
import csv, sys, glob, collections
n = sys.argv[1]
fs = glob.glob("gpurun_out/infpmc/%s/**/t_counter_collection.csv" % n, recursive=True)
if not fs: print(n, "no output"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if "k_bgzf_inflate" in r["Kernel_Name"]:
        acc[int(r["Grid_Size"]) // 64][r["Counter_Name"]].append(float(r["Counter_Value"]))
for nb, d in sorted(acc.items()):
    print(nb, "blocks:", "  ".join("%s %.4g/block" % (k, sum(v) / len(v) / nb) for k, v in sorted(d.items())))
P
done
