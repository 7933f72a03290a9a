#!/bin/bash
# Round 5, call A: the Spartan provers + transposed SpMV (tests, replay with breakdown, rocprof), the 8-way multidev tests,
# same-box A/B of the r03 field kernels against HEAD
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5a
mkdir -p "$OUT"
echo "== spartan tests"; timeout 1200 python -m pytest tests/test_gpu_spartan.py -q --maxfail=6 > "$OUT/pytest_spartan.txt" 2>&1; tail -15 "$OUT/pytest_spartan.txt"
echo "== multidev tests"; timeout 1200 python -m pytest tests/test_gpu_multidev.py -q --maxfail=4 > "$OUT/pytest_multidev.txt" 2>&1; tail -4 "$OUT/pytest_multidev.txt"
echo "== spartan replay"
for l in 14 17 20; do
  timeout 900 python bench.py --workload spartan_replay --log2n $l --steps 5 --warmup 2 > "$OUT/spartan_$l.json" 2> "$OUT/spartan_$l.err"; echo "l=$l rc=$?"
done
NMX_SC_POLL_US=0 timeout 900 python bench.py --workload spartan_replay --log2n 20 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/spartan_20_nopoll.json" 2> "$OUT/spartan_20_nopoll.err"
NMX_SC_POLL_US=0 NMX_SYNC_SPIN_US=200 timeout 900 python bench.py --workload spartan_replay --log2n 20 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/spartan_20_spin.json" 2> "$OUT/spartan_20_spin.err"
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/spartan_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms", round(d["value"],3), "cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("gpu_matches_cpu"), d["proof_verifies"])
        print("   breakdown", d["breakdown_ms"]); print("   provers", d["provers"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-1500:])
PY
echo "== rocprof spartan 2^20"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_spartan20" -o sp20 -- python bench.py --workload spartan_replay --log2n 20 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/prof_spartan20.log" 2>&1; echo "rc=$?"
find "$OUT/prof_spartan20" -name "*kernel_stats.csv" | head -1 | xargs -r head -40
echo "== A/B r03 vs HEAD"
for rep in 1 2; do
  for wl in sumcheck3 quad_prod cross_term round3; do
    # build/r03_tree: a checkout of the round-3 tree (git worktree add build/r03_tree 5613f68, built there); removed after this A/B
    (cd build/r03_tree && timeout 400 python bench.py --workload $wl --log2n 24 --steps 10 --warmup 3 --no-cpu-baseline) > "$OUT/ab_r03_${wl}_$rep.json" 2> "$OUT/ab_r03_${wl}_$rep.err"
    timeout 400 python bench.py --workload $wl --log2n 24 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/ab_head_${wl}_$rep.json" 2> "$OUT/ab_head_${wl}_$rep.err"
  done
done
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/ab_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), "kernel_ms", round(d["kernel_ms"],4), "frac", round(d["roofline"]["frac"],4))
    except Exception as e:
        print(f, "ERR", e)
PY
echo "== done"
