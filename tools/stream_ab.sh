# A/B of the stream placement (hgx_type.hip): measured classes ("placed", the default) against creation order ("unplaced")
for sw in placed unplaced placed; do
  HGX_STREAMS=$sw python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$sw configs1', d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['samples_in_flight'].items() if k!='note'}, 'class1', d['workloads']['class1']['ms_per_step'], d['workloads']['class1']['ms_per_step_spread'], 'panel64', d['workloads']['panel64']['ms_per_step'], d['workloads']['class1']['stream_sets'])"
done
