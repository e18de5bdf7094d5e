import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l)
        if "config" in d: print(d["config"], d["lds_nodes"], d["mrays_per_s"], d["closest_ms_per_frame"], d["shadow_ms_per_frame"])
