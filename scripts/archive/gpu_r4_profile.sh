#!/bin/bash
# Round 4 evidence run: the whole GPU suite + smoke + the default line + the in-process multi-GPU mode, then the kernel trace and
# PMC passes of the headline MSM and of the field kernels that sit near or under 0.40 of the HBM roofline.  PMC passes are separate
# runs with --kernel-trace only (gpurun's rule).  Outputs under gpurun_out/<tag>; summaries are copied to profiles/r04_* afterwards.
set -u
export TMPDIR=/tmp
TAG=${1:-r4prof}
OUT=gpurun_out/$(date +%H%M%S)_$TAG; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest gpu"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=5 > "$OUT/pytest_gpu.txt" 2>&1; tail -4 "$OUT/pytest_gpu.txt"
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
echo "== default line"
timeout 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"; echo "rc=$?"; cut -c1-300 "$OUT/bench_default_line.json"
echo "== in-process --gpus 2 (falls back to the one device), then 2 logical devices oversubscribed with RCCL required"
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 > "$OUT/inproc_gpus2.json" 2> "$OUT/inproc_gpus2.err"; echo "rc=$?"
NMX_BENCH_OVERSUB=1 NMX_BENCH_COMBINE=2 timeout 600 python bench.py --gpus 2 --total-log2n 22 --steps 5 --warmup 2 > "$OUT/inproc_oversub2.json" 2> "$OUT/inproc_oversub2.err"; echo "rc=$?"
python - "$OUT" <<'PY'
import json,sys
for f in ("inproc_gpus2","inproc_oversub2"):
    try:
        d=json.loads(open(f"{sys.argv[1]}/{f}.json").read().strip().splitlines()[-1])
        print(f, "ms", round(d["ms_per_step"],3), "n_gpus", d["n_gpus"], "rccl_ranks", d["rccl_ranks"], "combine_ms", d["combine_ms"], "matches_cpu", d["cpu_baseline"]["gpu_matches_cpu"], "gpu0", (d.get("scalars_on_gpu0") or {}).get("matches"))
    except Exception as e:
        print(f, "ERR", e)
PY
echo "== kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -o msm -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/trace.err" )
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && cut -d, -f1-4 "$f" | sed 's/void nmx:://; s/(nmx::[^"]*"/"/; s/(nmx::.*)//' | cut -c1-110 | head -22
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/$OUT/msm/p$i" -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> "$R/$OUT/msm_p$i.err" )
  echo "== msm pmc pass $i done: $(find "$OUT/msm/p$i" -name '*counter_collection.csv' | head -1)"
done
python scripts/pmc_summary.py "$OUT/msm" "$OUT/pmc_traffic.json" | head -30
for wl in ${FV:-horner:22 sumcheck3:24 round3:24 spmv:22 mle_eval:24 mle_eval:20 lincomb8:22 quad_prod:24 cross_term:24}; do
  name=${wl%%:*}; lg=${wl##*:}
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"; do
    i=$((i+1))
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/$OUT/fv_${name}_$lg/p$i" -o pmc -- python "$R/bench.py" --workload $name --log2n $lg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$R/$OUT/fv_${name}_${lg}_p$i.err" )
  done
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/fv_${name}_$lg/trace" -o t -- python "$R/bench.py" --workload $name --log2n $lg --steps 5 --warmup 2 --no-cpu-baseline > "$R/$OUT/fv_${name}_${lg}_bench.json" 2> /dev/null )
  f=$(find "$OUT/fv_${name}_$lg/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/fv_${name}_${lg}_kernel_stats.csv"
  echo "== fieldvec $name 2^$lg"
  python scripts/pmc_fieldvec.py "$OUT/fv_${name}_$lg" "$OUT/fv_${name}_${lg}_pmc.json"
done
# drop the raw counter dumps (the summaries stay): the merge-back limit is 64 MiB
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT" | cut -f1
echo "== done"
