"""Summarise the PMC passes of scripts/gpu_pmc.sh into profiles/<dir>/pmc_traffic.json (per-launch means)."""
import collections, csv, glob, json, sys
src, dst = sys.argv[1], sys.argv[2]
names = {"AccumFn": "accum", "AccumSegFn": "accum", "ReducePair": "reduce", "FoldFn": "fold", "FoldRaw": "fold_raw", "FinalSeg": "final_seg",
         "radix_sort": "sort", "DigitsFn": "digits", "k_hist_hi": "hist_hi", "k_part_hi": "part_hi", "k_hist_lo": "hist_lo",
         "k_part_lo": "part_lo", "k_reduce_tree": "reduce_tree", "k_big_all": "big_all"}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for pat, nm in names.items():
            if pat in r["Kernel_Name"]:
                agg[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_note": "rocprofv3 --pmc, separate passes, per-launch means. FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE "
                "reports half of the bytes of dword/dwordx4 streaming reads (MI355X_MICROARCH.md, HBM section) -- calibrated here on "
                "a streaming kernel whose bytes are known exactly (round 1: DigitsFn, 96 B per pair; round 2: k_part_lo, which reads "
                "5 B and writes 4 B per sorted entry)."}
for nm, d in agg.items():
    out[nm] = {c: sum(v) / len(v) for c, v in d.items()}
    if "FETCH_SIZE" in out[nm]:
        out[nm]["hbm_read_bytes_corrected"] = out[nm]["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in out[nm]:
        out[nm]["hbm_write_bytes"] = out[nm]["WRITE_SIZE"] * 1024
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: {c: round(x) for c, x in v.items()} if isinstance(v, dict) else v for k, v in out.items() if k != "_note"}, indent=0)[:1500])
