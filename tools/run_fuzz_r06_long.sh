#!/bin/bash
# Round 6, the longer campaigns (other seeds than tools/run_fuzz_r06.sh): tools/run_fuzz_r06_long.sh <name> -> gpurun_out/<name>/*.log
D=gpurun_out/$1
mkdir -p $D
python3 tools/fuzz_alignment.py 2500 1660000 > $D/fuzz_alignment_2500.log 2>&1
python3 tools/fuzz_front.py 4000 4990000 bam > $D/fuzz_front_bam_4000.log 2>&1
python3 tools/fuzz_parity.py 4000 141000 > $D/fuzz_parity_4000.log 2>&1
python3 tools/fuzz_many.py 1500 1850000 > $D/fuzz_many_1500.log 2>&1
python3 tools/fuzz_inflate.py 30000 131 > $D/fuzz_inflate_30000.log 2>&1
python3 tools/fuzz_em_large.py 40 1530000 > $D/fuzz_em_large_40.log 2>&1
tail -n 2 $D/*.log
