import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"])
print(json.dumps(d["roofline"].get("configs"), indent=1)[:3000])
print(d["roofline"].get("file_to_file_ms"))
print(json.dumps(d.get("fmm"), indent=1)[:1500])
print(json.dumps(d.get("c5"), indent=1)[:900])
print(d.get("c4_strong",{}).get("gpu_state",{}).get("before"))
