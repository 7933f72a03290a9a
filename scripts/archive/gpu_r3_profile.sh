#!/bin/bash
# Round 3 profiles: kernel trace + PMC passes of the headline MSM, PMC passes of the field-vector kernels that sit near or
# under 0.40 of the HBM roofline.  PMC passes are separate runs with --kernel-trace only (gpurun's rule).  Outputs under
# gpurun_out/<tag>; the summaries are copied to profiles/r03_* by hand afterwards.
set -u
export TMPDIR=/tmp
TAG=${1:-r3prof}
OUT=gpurun_out/$(date +%H%M%S)_$TAG; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest gpu"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=5 -x -s > "$OUT/pytest_gpu.txt" 2>&1; tail -6 "$OUT/pytest_gpu.txt"; grep -h "IPA-shaped" "$OUT/pytest_gpu.txt"
fi
echo "== bench (plain, for box_variance)"
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"; cut -c1-400 "$OUT/bench_plain.json"
echo "== kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -o msm -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/trace.err" )
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4 "$f" | sed 's/void nmx:://; s/(nmx::[^"]*"/"/; s/(nmx::.*)//' | cut -c1-110 | head -24
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/$OUT/msm/p$i" -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> "$R/$OUT/msm_p$i.err" )
  echo "== msm pmc pass $i done: $(find "$OUT/msm/p$i" -name '*counter_collection.csv' | head -1)"
done
python scripts/pmc_summary.py "$OUT/msm" "$OUT/pmc_traffic.json" | head -40
for wl in ${FV:-horner:22 sumcheck3:24 mle_eval:24 lincomb8:22}; do
  name=${wl%%:*}; lg=${wl##*:}
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"; do
    i=$((i+1))
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/$OUT/fv_$name/p$i" -o pmc -- python "$R/bench.py" --workload $name --log2n $lg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$R/$OUT/fv_${name}_p$i.err" )
  done
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/fv_$name/trace" -o t -- python "$R/bench.py" --workload $name --log2n $lg --steps 5 --warmup 2 --no-cpu-baseline > "$R/$OUT/fv_${name}_bench.json" 2> /dev/null )
  echo "== fieldvec $name 2^$lg"
  python scripts/pmc_fieldvec.py "$OUT/fv_$name" "$OUT/fv_${name}_pmc.json"
done
echo "== done"
