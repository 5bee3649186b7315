for cfg in "0 0" "11 0" "11 8" "4 8" "15 8" "11 16" "22 8"; do
  set -- $cfg
  HGX_EM_BATCH1=$1 HGX_EM_BATCH=$2 python bench.py --no-workloads --no-e2e --no-cpu-baseline --steps 60 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch1=$1 batch=$2', d['ms_per_step'], d['em_iters_per_s'])"
done
