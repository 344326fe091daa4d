"""print the essentials of bench.py JSON lines read from stdin"""
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l)
    except Exception:
        if l.strip():
            print(l[:300].rstrip())
        continue
    r = d["roofline"]
    print("%-62s | decode %8.1f tok/s | prefill %9.1f tok/s | launch frac %s | step frac %.3f" % (
        d["config"]["workload"][:62], d["value"], d.get("prefill_tok_s", 0.0), ("%.3f" % r["launch_frac"]) if r.get("launch_frac") else "-", r.get("step_frac", 0.0)))
