for n in 0 64 96 128 160 192 224; do
  echo "== HGX_GENE_CUS=$n"
  HGX_GENE_CUS=$n python bench.py --no-cpu-baseline --no-e2e --no-workloads --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"
done
