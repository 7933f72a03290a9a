"""Per-kernel summary of the PMC passes + kernel trace of one field-vector workload (scripts/archive/gpu_r3_profile.sh): launches,
mean duration, VALU instructions, wave cycles, HBM bytes -- and the VALU-issue ceiling they imply."""
import collections, csv, glob, json, sys
src, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "nmx::" not in k:
            continue
        agg[k.split("(")[0].replace("void ", "")[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(src + "/trace/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "nmx::" in k:
            dur[k.split("(")[0].replace("void ", "")[:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"_note": "rocprofv3 --pmc (separate passes, --kernel-trace only) per-launch means; FETCH_SIZE / WRITE_SIZE in KiB, reads doubled "
                "per the guide's gfx950 correction; lane_ops = SQ_INSTS_VALU x 64; valu_floor_us = lane_ops / 30e12 (the measured "
                "VOP3 issue rate of the chip, profiles/r01_ubench_instruction_rates.jsonl)"}
for k, d in agg.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    if k in dur:
        xs = dur[k][len(dur[k]) // 3:]  # skip warm-up launches
        e["launches_traced"] = len(dur[k])
        e["mean_us"] = sum(xs) / len(xs)
    if "FETCH_SIZE" in e:
        e["hbm_read_bytes_corrected"] = e["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in e:
        e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
    if "SQ_INSTS_VALU" in e:
        e["lane_ops"] = e["SQ_INSTS_VALU"] * 64
        e["valu_floor_us"] = e["lane_ops"] / 30e12 * 1e6
    out[k] = e
json.dump(out, open(dst, "w"), indent=1)
for k, e in out.items():
    if isinstance(e, dict):
        print(k[:70], {c: (round(x, 1) if isinstance(x, float) else x) for c, x in e.items()
                       if c in ("mean_us", "valu_floor_us", "SQ_INSTS_VALU", "hbm_read_bytes_corrected", "hbm_write_bytes", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")})
