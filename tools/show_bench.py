"""Print the headline numbers of bench.py JSON lines: python tools/show_bench.py file.json ..."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(f, "value %.4g %s  ms/step %.4f  launch_ms %s  dominant %s" % (d["value"], d["unit"], d["ms_per_step"], r.get("launch_ms"), r.get("dominant_kernel_ms")))
    for k in ("roofline_policy", "roofline_ppo_update"):
        if k in d:
            print("   ", k, {q: d[k][q] for q in ("achieved", "frac", "launch_ms") if q in d[k]})
