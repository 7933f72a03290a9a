#!/bin/bash
# Round 5, call B: the provers on HostFp4 + host tail + one index per thread, transposed SpMV with block-summed split columns,
# two-launch eq tables; tail threshold sweep; small-MSM stage profile
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5b
mkdir -p "$OUT"
echo "== spartan tests"; timeout 1500 python -m pytest tests/test_gpu_spartan.py -q --maxfail=6 > "$OUT/pytest_spartan.txt" 2>&1; tail -12 "$OUT/pytest_spartan.txt"
echo "== fieldvec tests (eq tables changed)"; timeout 1500 python -m pytest tests/test_gpu_fieldvec.py tests/test_gpu_fieldvec_large.py -q --maxfail=6 > "$OUT/pytest_fieldvec.txt" 2>&1; tail -4 "$OUT/pytest_fieldvec.txt"
echo "== spartan replay"
for l in 14 17 20; do
  timeout 900 python bench.py --workload spartan_replay --log2n $l --steps 5 --warmup 2 > "$OUT/spartan_$l.json" 2> "$OUT/spartan_$l.err"; echo "l=$l rc=$?"
done
for t in 0 4 5 7 8; do
  NMX_SC_HOST_TAIL=$t timeout 900 python bench.py --workload spartan_replay --log2n 20 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/spartan_20_tail$t.json" 2> "$OUT/spartan_20_tail$t.err"
  NMX_SC_HOST_TAIL=$t timeout 900 python bench.py --workload spartan_replay --log2n 14 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/spartan_14_tail$t.json" 2> "$OUT/spartan_14_tail$t.err"
done
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/spartan_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms", round(d["value"],3), "cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("gpu_matches_cpu"), all(d["proof_verifies"].values()))
        print("   breakdown", d["breakdown_ms"]); print("   provers", d["provers"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-1500:])
PY
echo "== rocprof spartan 2^20"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_spartan20" -o sp20 -- python "$GRAFT_REPO_ROOT/bench.py" --workload spartan_replay --log2n 20 --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$OUT/prof_spartan20.log" 2>&1 ); echo "rc=$?"
find "$OUT/prof_spartan20" -name "*kernel_stats.csv" | head -1 | xargs -r head -30
echo "== small MSM stages"
timeout 600 python scripts/gpu_small_msm_stages.py 30 > "$OUT/small_msm_stages.txt" 2>&1; cat "$OUT/small_msm_stages.txt"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_smallmsm" -o sm -- python "$GRAFT_REPO_ROOT/scripts/gpu_small_msm_stages.py" 10 > /dev/null 2>&1 ); echo "rc=$?"
echo "== done"
