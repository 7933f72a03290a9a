#!/bin/bash
# Round 3: counters of the single-pass suffix Horner kernel (separate --pmc passes, --kernel-trace only).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-hpmc}; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
LG=${LG:-24}
for sub in ${SUBS:-1 4}; do
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"; do
    i=$((i+1))
    ( cd /tmp && NMX_TUNE_HORNER_SUB=$sub timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/$OUT/sub$sub/p$i" -o pmc -- python "$R/bench.py" --workload horner --log2n $LG --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$R/$OUT/sub${sub}_p$i.err" ) || { echo "pass $i rc=$?"; tail -3 "$OUT/sub${sub}_p$i.err"; }
  done
  ( cd /tmp && NMX_TUNE_HORNER_SUB=$sub timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/sub$sub/trace" -o t -- python "$R/bench.py" --workload horner --log2n $LG --steps 5 --warmup 2 --no-cpu-baseline > "$R/$OUT/sub${sub}_bench.json" 2> /dev/null )
  echo "== horner 2^$LG sub=$sub"
  python - "$OUT/sub$sub" "$OUT/horner_scan_sub${sub}_pmc.json" <<'PY'
import collections, csv, glob, json, sys
src, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "nmx::" in k:
            agg[k.split("(")[0].replace("void ", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(src + "/trace/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "nmx::" in k:
            dur[k.split("(")[0].replace("void ", "")[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k, d in agg.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    if k in dur:
        xs = dur[k][len(dur[k]) // 3:]
        e["mean_us"] = sum(xs) / len(xs)
    out[k] = e
    print(k, {c: round(x, 1) for c, x in e.items()})
json.dump(out, open(dst, "w"), indent=1)
PY
done
echo "== done"
