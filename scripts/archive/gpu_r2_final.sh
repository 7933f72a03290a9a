#!/bin/bash
# Round-2 final numbers: all GPU tests, default bench line, other sizes / curves / distributions, concurrency.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-final}
mkdir -p "$OUT"
nproc > "$OUT/nproc.txt"; lscpu | grep -E "Model name|^CPU\(s\)" >> "$OUT/nproc.txt"; cat /sys/fs/cgroup/cpu.max >> "$OUT/nproc.txt" 2>/dev/null
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== pytest gpu"
timeout 1500 python -m pytest tests -q -m gpu --maxfail=5 --durations=12 > "$OUT/pytest_gpu.txt" 2>&1; tail -22 "$OUT/pytest_gpu.txt"
echo "== default bench"
t0=$(date +%s); timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "rc=$? wall=$(( $(date +%s) - t0 ))s"; cat "$OUT/bench.json"
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(f"{sys.argv[2]:>26}: {d['ms_per_step']:.4f} ms  {d['value']/1e6:8.1f} M/s ", d.get("stages_ms"))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
for lg in 10 12 13 14 15 16 17 18 19 21 22 24; do
  timeout 600 python bench.py --log2n $lg --steps 10 --warmup 3 --no-extras --no-cpu-baseline > "$OUT/bench_2p${lg}_single_gpu.json" 2>/dev/null; show "$OUT/bench_2p${lg}_single_gpu.json" "bn254 2^$lg"
done
for cv in 1 2 3; do
  timeout 300 python bench.py --curve $cv --steps 10 --warmup 3 --no-extras > "$OUT/bench_curve$cv.json" 2>/dev/null; show "$OUT/bench_curve$cv.json" "curve $cv 2^20"
done
for dist in u1 u10 u16 u32 u64 equal zero_rm1; do
  timeout 300 python bench.py --dist $dist --steps 10 --warmup 3 --no-extras --no-cpu-baseline > "$OUT/dist_$dist.json" 2>/dev/null; show "$OUT/dist_$dist.json" "dist $dist"
done
for it in 1024 65536; do
  timeout 300 python bench.py --workload prove_step_replay --iters $it --steps 10 --warmup 3 > "$OUT/replay_$it.json" 2>/dev/null; python - "$OUT/replay_$it.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("prove_step replay", d["config"]["workload"][:60], round(d["value"],4), "ms  cpu", round(d["cpu_baseline"]["value"],1), d["cpu_baseline"]["gpu_matches_cpu"])
PY
done
for lg in 14 16 20; do
  timeout 600 python bench.py --workload hyperkzg_replay --log2n $lg --steps 5 --warmup 2 > "$OUT/hkzg_$lg.json" 2>/dev/null; python - "$OUT/hkzg_$lg.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("hyperkzg replay", d["config"]["workload"][:40], round(d["value"],4), "ms  cpu", round(d["cpu_baseline"]["value"],1), d["cpu_baseline"]["gpu_matches_cpu"])
PY
done
echo "== concurrent callers"; timeout 300 python scripts/gpu_concurrent.py 2>/dev/null | tee "$OUT/concurrent_callers.txt"
echo "== small MSMs (slice form, cached)"; timeout 300 python scripts/gpu_smallmsm.py 2>/dev/null | tee "$OUT/small_msm.txt"
echo "== done"
