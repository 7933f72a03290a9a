import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d.get("roofline") or {}
    print(f.split("/")[-1], "| ms/step", round(d["ms_per_step"], 4), "| kernel_ms", round(d.get("kernel_ms", 0), 4),
          "| value %.4g %s" % (d["value"], d["unit"]), "| roofline %.1f GB/s frac %.3f" % (r.get("achieved", 0), r.get("frac", 0)),
          "| stages", d.get("stages_ms"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("gpu_matches_cpu"))
