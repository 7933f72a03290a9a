#!/bin/bash
# Round 5, call F: DPF ubench (rounding mode ordered), finer grids of the reduction passes below 2^22, host_split by size
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5f
mkdir -p "$OUT"
echo "== dpf ubench"; timeout 300 bench/dpf_ubench | tee "$OUT/dpf_ubench.jsonl"
echo "== fieldvec + spartan tests"; timeout 1800 python -m pytest tests/test_gpu_fieldvec.py tests/test_gpu_fieldvec_large.py tests/test_gpu_spartan.py -q --maxfail=6 > "$OUT/pytest_a.txt" 2>&1; tail -4 "$OUT/pytest_a.txt"
echo "== trait form / host_split"; timeout 900 python scripts/gpu_trait_form.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/host_split.txt"
echo "== fieldvec workloads at 2^20 and 2^24"
for wl in sumcheck3 round3 quad_prod mle_eval; do for lg in 20 24; do
  timeout 400 python bench.py --workload $wl --log2n $lg --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', $lg, 'kernel_ms', round(d['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4), d.get('cpu_baseline',{}).get('gpu_matches_cpu'))"
done; done
echo "== spartan replay"; for l in 14 20; do timeout 900 python bench.py --workload spartan_replay --log2n $l --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spartan', $l, round(d['value'],3), d['breakdown_ms'])"; done
echo "== done"
