#!/bin/bash
# Round 3 final evidence (third pass: single-pass Horner scan, shared-reduction field kernels, sympy KATs):
# GPU suite, smoke, default line, kernel trace of the headline command, field-kernel counters, the Horner sizes.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3final3}
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 1500 python -m pytest tests -q -m gpu --maxfail=5 -x > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -2
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== default line"; timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"; cut -c1-200 "$OUT/bench_default.json"
echo "== kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -o msm -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/trace.err" )
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && cut -d, -f1-4 "$f" | sed 's/void nmx:://; s/(nmx::[^"]*"/"/; s/(nmx::.*)//' | cut -c1-110 | head -12
rm -rf "$OUT/trace"
echo "== field-kernel counters"
for wl in ${FV:-lincomb8:22 mle_eval:24 sumcheck3:24 horner:22}; do
  name=${wl%%:*}; lg=${wl##*:}
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/$OUT/fv_$name/p$i" -o pmc -- python "$R/bench.py" --workload $name --log2n $lg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$R/$OUT/fv_${name}_p$i.err" )
  done
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/fv_$name/trace" -o t -- python "$R/bench.py" --workload $name --log2n $lg --steps 5 --warmup 2 --no-cpu-baseline > "$R/$OUT/fv_${name}_bench.json" 2> /dev/null )
  echo "-- $name 2^$lg"
  python scripts/pmc_fieldvec.py "$OUT/fv_$name" "$OUT/${name}_pmc.json" | cut -c1-260
  f=$(find "$OUT/fv_$name/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv"
  rm -rf "$OUT/fv_$name"
done
echo "== horner sizes"
for lg in 14 16 18 20 21 22 24; do
  timeout 300 python bench.py --workload horner --log2n $lg --steps 20 --warmup 3 $([ $lg -ge 24 ] && echo --no-cpu-baseline) 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^$lg kernel %.4f ms frac %.3f call %.4f ms matches=%s' % (d['kernel_ms'], d['roofline']['frac'], d['ms_per_step'], d.get('cpu_baseline',{}).get('gpu_matches_cpu')))"
done
echo "== hyperkzg / prove_step replays"
timeout 300 python bench.py --workload hyperkzg_replay --log2n 20 --steps 5 --warmup 2 > "$OUT/hkzg_20.json" 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/hkzg_20.json').read().strip().splitlines()[-1]); print('hkzg 2^20', round(d['ms_per_step'],3), 'ms', d.get('cpu_baseline',{}).get('gpu_matches_cpu'))"
echo "== done"
