#!/bin/bash
# round 6: counters on the round-6 build.  (1) the MSM at 2^20 (kernel trace + three --pmc passes, each its own run: profiles/r06_msm/),
# (2) the Spartan replay at 2^20 (kernel trace + the same passes: the k_sc_* kernels had no counters in round 5: profiles/r06_spartan/)
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for what in msm spartan; do
  OUT=gpurun_out/r6pmc_$what; mkdir -p "$OUT"
  if [ $what = msm ]; then ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extras"; else ARGS="--workload spartan_replay --log2n 20 --steps 3 --warmup 1 --no-cpu-baseline"; fi
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -o t -- python "$R/bench.py" $ARGS > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/trace.err" )
  cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv" 2>/dev/null
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/$OUT/p$i" -o pmc -- python "$R/bench.py" $ARGS > /dev/null 2> "$R/$OUT/p$i.err" )
  done
  python3 - "$OUT" $what <<'PY'
import collections, csv, glob, json, sys
src, what = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(src + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k.split("(")[0].replace("void ", "").replace("nmx::", "")[:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(src + "/p1/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nmx::", "")[:70]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"_note": "rocprofv3 --pmc, separate passes (never combined with other trace domains), per-launch means over every launch of the run. "
                "FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of dword/dwordx4 streaming reads "
                "(MI355X_MICROARCH.md, HBM section): hbm_read_bytes_corrected = FETCH_SIZE x 1024 x 2. us = mean duration under the FETCH_SIZE pass."}
for k, d in sorted(agg.items()):
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e["launches"] = max(len(v) for v in d.values())
    if "FETCH_SIZE" in e: e["hbm_read_bytes_corrected"] = e["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in e: e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
    if k in dur: e["us"] = sum(dur[k]) / len(dur[k])
    out[k] = e
json.dump(out, open(src + "/pmc.json", "w"), indent=1)
for k, e in out.items():
    if k != "_note" and e.get("us", 0) * e["launches"] > 20:
        print(k[:60], {c: round(x, 1) for c, x in e.items() if c in ("us", "launches", "hbm_read_bytes_corrected", "hbm_write_bytes", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")})
PY
  rm -rf "$OUT"/p*/ "$OUT"/trace   # (raw csvs are large; the summaries stay)
done
