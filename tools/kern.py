import sys, json
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        k = d["roofline"]["kernels"]
        print(d["ms_per_step"], {n: (v["avg_ms"], v["total_ms_per_step"]) for n, v in k.items()})
