import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print(d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["roofline_detail"].items() if k.endswith("_all")})
