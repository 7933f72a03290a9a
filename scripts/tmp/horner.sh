cd $GRAFT_REPO_ROOT
for v in 8 4 1; do
export NMX_TUNE_HORNER_TOP=$v
echo "== horner_top=$v"
timeout 900 python -m pytest tests -q -m gpu -k "horner or streaming_kernels" -x 2>&1 | tail -2
for lg in 16 20 22 24; do
  timeout 300 python bench.py --workload horner --log2n $lg --steps 10 --warmup 3 > gpurun_out/h.json 2>/dev/null
  python - gpurun_out/h.json $lg <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"horner 2^{sys.argv[2]}: kernel {d['kernel_ms']:.4f} ms  {d['roofline']['achieved']:.0f} GB/s frac {d['roofline']['frac']:.3f} matches={d.get('cpu_baseline',{}).get('gpu_matches_cpu')}")
PY
done
done
