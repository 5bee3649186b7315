#!/bin/bash
# The measurements behind round 6's last changes (SAM text in two parts, a BAM's container walk in ranges), on the GPU box:
# tools/run_evidence_r06_late.sh <name> -> gpurun_out/<name>/*.txt   (the rocprofv3 traces themselves are not kept)
D=gpurun_out/$1
mkdir -p $D
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
echo "== tools/numa_h2d_probe.hip: host -> device rate of 410 MB of registered memory, by NUMA node of its pages"
hipcc --offload-arch=gfx950 -O2 -pthread tools/numa_h2d_probe.hip -o /tmp/numa_h2d_probe 2>/dev/null && /tmp/numa_h2d_probe
} > $D/numa_h2d_probe.txt 2>&1
for form in parts sam_whole; do
  rm -rf $D/tr
  if [ $form = parts ]; then
    rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $D/tr -o t -- python3 tools/e2e_file.py 500000 0 > $D/run_$form.log 2>&1
  else
    HGX_E2E_FRONT=sam_whole rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $D/tr -o t -- python3 tools/e2e_file.py 500000 0 > $D/run_$form.log 2>&1
  fi
  { echo "== hgx_type_file on SAM text, 1 M reads, form: $form (tools/sam_copy_timeline.py, then tools/bam_gap_timeline.py with k_fe_decode as the call's marker)"
    python3 tools/sam_copy_timeline.py $D/tr 398 | tail -12
    python3 tools/bam_gap_timeline.py $D/tr k_fe_decode 25 150
  } >> $D/sam_copy_timeline.txt 2>&1
done
rm -rf $D/tr
rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $D/tr -o t -- python3 tools/e2e_bam.py 500000 > $D/run_bam.log 2>&1
{ echo "== hgx_type_file on a coordinate-sorted BAM, 1 M reads (tools/bam_gap_timeline.py): kernels of 40 us and more, idle stretches of 25 us and more"
  grep "^run" $D/run_bam.log
  python3 tools/bam_gap_timeline.py $D/tr
} > $D/bam_gap_timeline.txt 2>&1
rm -rf $D/tr $D/run_*.log
python3 tools/sam_split_ab2.py 3 2>&1 | grep "^parts\|^whole" > $D/sam_split_ab.txt
tail -n 8 $D/*.txt
