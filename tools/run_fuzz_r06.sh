#!/bin/bash
# Round 6 fuzz campaigns on the GPU box (final tree): tools/run_fuzz_r06.sh <name> -> gpurun_out/<name>/*.log
D=gpurun_out/$1
mkdir -p $D
python3 tools/fuzz_alignment.py 600 660000 > $D/fuzz_alignment_600.log 2>&1
python3 tools/fuzz_front.py 1500 1990000 bam > $D/fuzz_front_bam_1500.log 2>&1
python3 tools/fuzz_front.py 1500 2990000 > $D/fuzz_front_sam_1500.log 2>&1
python3 tools/fuzz_parity.py 1500 41000 > $D/fuzz_parity_1500.log 2>&1
python3 tools/fuzz_many.py 600 850000 > $D/fuzz_many_600.log 2>&1
python3 tools/fuzz_inflate.py 10000 31 > $D/fuzz_inflate_10000.log 2>&1
python3 tools/fuzz_em_large.py 30 530000 > $D/fuzz_em_large_30.log 2>&1
tail -n 2 $D/*.log
