#!/usr/bin/env python3
"""The numbers of one bench.py line at a glance:  python tools/show_bench.py FILE  (the last line of FILE is the JSON)."""
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("step_ms_spread"), d["roofline"]["frac"], d.get("cpu_baseline"))
for k,v in d.items():
    if k.startswith("end_to_end"):
        print(k, {a:b for a,b in v.items() if not isinstance(b,(dict,list)) and a not in ("what","parity")})
        for a,b in v.items():
            if isinstance(b,dict): print("   ", a, {x:y for x,y in b.items() if x not in ("what","parity")})
