#!/bin/bash
# The fuzz campaigns once more on the round's LAST tree (after D18S51 on the kernels, key.find on ids, SAM text in two parts, the
# container walk in ranges): other seeds again.  tools/run_fuzz_r06_last.sh <name> -> gpurun_out/<name>/*.log
D=gpurun_out/$1
mkdir -p $D
python3 tools/fuzz_sam_parts.py 150 990000 > $D/fuzz_sam_parts_150.log 2>&1
python3 tools/fuzz_alignment.py 500 2660000 > $D/fuzz_alignment_500.log 2>&1
python3 tools/fuzz_front.py 2500 5990000 bam > $D/fuzz_front_bam_2500.log 2>&1
python3 tools/fuzz_front.py 2500 6990000 > $D/fuzz_front_sam_2500.log 2>&1
python3 tools/fuzz_parity.py 3000 91000 > $D/fuzz_parity_3000.log 2>&1
python3 tools/fuzz_many.py 600 2850000 > $D/fuzz_many_600.log 2>&1
python3 tools/fuzz_inflate.py 10000 77 > $D/fuzz_inflate_10000.log 2>&1
tail -n 2 $D/*.log
