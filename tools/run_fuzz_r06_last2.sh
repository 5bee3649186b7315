#!/bin/bash
# After the round's very last changes (distinct pieces by hash table, key order by counting, one sort for the canonical order):
# the front-end campaigns once more, other seeds.  tools/run_fuzz_r06_last2.sh <name> -> gpurun_out/<name>/*.log
D=gpurun_out/$1
mkdir -p $D
python3 tools/fuzz_front.py 2500 11990000 bam > $D/fuzz_front_bam_2500.log 2>&1
python3 tools/fuzz_front.py 1500 12990000 > $D/fuzz_front_sam_1500.log 2>&1
python3 tools/fuzz_parity.py 2000 191000 > $D/fuzz_parity_2000.log 2>&1
python3 tools/fuzz_many.py 500 4850000 > $D/fuzz_many_500.log 2>&1
python3 tools/fuzz_alignment.py 300 3660000 > $D/fuzz_alignment_300.log 2>&1
python3 tools/fuzz_sam_parts.py 60 2990000 > $D/fuzz_sam_parts_60.log 2>&1
tail -n 2 $D/*.log
