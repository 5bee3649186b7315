import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(d["ms_per_step"], round(d["value"] / 1e6, 1), "Mreads/s", d["roofline"]["kernel"])
