#!/bin/bash
# Round 3: the fused MSM tail (k_big_all, k_reduce_tree, plan inside the accumulate kernel) against round 2's
# one-launch-per-level tree (NMX_TUNE_NO_TREE_FUSE=1), per size and per scalar distribution + kernel trace.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r3tail}
mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d["stages_ms"])
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest gpu"
timeout 1200 python -m pytest tests -q -m gpu --maxfail=5 -x ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.txt" 2>&1; tail -8 "$OUT/pytest_gpu.txt"
fi
for lg in ${SIZES:-20 21 18 16 13}; do
  for mode in ${MODES:-T F}; do
    # T: one launch per reduction level; F: fused tree (default); N / NT: the same with the chained products in the tail
    # passes (scripts/build_variant.sh nolat -DNMX_LAT_TAIL=0)
    case $mode in T) envs="NMX_TUNE_NO_TREE_FUSE=1";; F) envs="NMX_X=0";;
      N) envs="NMX_SO=$PWD/nova_amd/libnova_mi355x_nolat.so";; NT) envs="NMX_SO=$PWD/nova_amd/libnova_mi355x_nolat.so NMX_TUNE_NO_TREE_FUSE=1";; esac
    echo "== $mode log2n=$lg"
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --log2n $lg --no-extras --no-cpu-baseline > "$OUT/bench_${mode}_$lg.json" 2> "$OUT/bench_${mode}_$lg.err"
    show "$OUT/bench_${mode}_$lg.json"
  done
done
for dist in ${DISTS:-u1 u16 equal zero_rm1}; do
  echo "== F dist=$dist"
  timeout 300 python bench.py --steps 10 --warmup 3 --dist $dist --no-extras --no-cpu-baseline > "$OUT/bench_F_$dist.json" 2> "$OUT/bench_F_$dist.err"
  show "$OUT/bench_F_$dist.json"
done
if [ "${SKIP_PROF:-0}" != "1" ]; then
echo "== rocprof kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o msm -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4 "$f" | sed 's/void nmx:://; s/(nmx::[^"]*"/"/; s/(nmx::.*)//' | cut -c1-120 | head -26
fi

if [ "${SKIP_INPROC:-0}" != "1" ]; then
echo "== in-process multi-GPU mode on this box (fallback expected when fewer GPUs are visible)"
timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --total-log2n ${INPROC_LOG2N:-22} > "$OUT/bench_inproc_gpus2.json" 2> "$OUT/bench_inproc_gpus2.err"; echo "rc=$?"; tail -2 "$OUT/bench_inproc_gpus2.err"; cut -c1-900 "$OUT/bench_inproc_gpus2.json"
fi
echo "== done"
