import json, sys
for n in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(n) if l.startswith("{")][-1])
        print(n.split("/")[-1], "ms/step", d["ms_per_step"], "value", d["value"], "roofline", d.get("roofline", {}).get("frac"),
              "stages", {k: v.get("ms") if isinstance(v, dict) else v for k, v in (d.get("stages") or {}).items()})
    except Exception as e:
        print(n, "failed", e)
