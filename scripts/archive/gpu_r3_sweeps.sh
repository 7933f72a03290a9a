#!/bin/bash
# Round 3 closing measurements with the final build: GPU suite (log kept), sizes, curves, scalar distributions, field-kernel PMC.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3sweeps}
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
one() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), round(d["value"]/1e6,1), d.get("stages_ms"))
except Exception as e:
    print("ERR", e)
PY
}
echo "== pytest gpu"; timeout 1500 python -m pytest tests -q -m gpu --maxfail=5 -x > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -2
for lg in 10 12 13 14 15 16 17 18 19 20 21 22 23 24; do
  st=30; [ $lg -ge 22 ] && st=5
  echo "== size 2^$lg"; timeout 600 python bench.py --steps $st --warmup 3 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/size_$lg.json" 2>/dev/null; one "$OUT/size_$lg.json"
done
for cv in 1 2 3; do echo "== curve $cv 2^20"; timeout 300 python bench.py --steps 20 --warmup 5 --curve $cv --no-extras --no-cpu-baseline > "$OUT/curve_$cv.json" 2>/dev/null; one "$OUT/curve_$cv.json"; done
for dist in u1 u10 u16 u32 u64 equal zero_rm1; do echo "== dist $dist 2^20"; timeout 300 python bench.py --steps 20 --warmup 5 --dist $dist --no-extras --no-cpu-baseline > "$OUT/dist_$dist.json" 2>/dev/null; one "$OUT/dist_$dist.json"; done
for wl in sumcheck3:24 mle_eval:24 lincomb8:22; do
  name=${wl%%:*}; lg=${wl##*:}; i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"; do
    i=$((i+1))
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/$OUT/fv_$name/p$i" -o pmc -- python "$R/bench.py" --workload $name --log2n $lg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 )
  done
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/fv_$name/trace" -o t -- python "$R/bench.py" --workload $name --log2n $lg --steps 5 --warmup 2 --no-cpu-baseline > "$R/$OUT/fv_${name}_bench.json" 2>/dev/null )
  echo "== fieldvec $name 2^$lg"; python scripts/pmc_fieldvec.py "$OUT/fv_$name" "$OUT/fv_${name}_pmc.json" | cut -c1-300
done
echo "== done"
