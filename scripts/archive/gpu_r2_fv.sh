#!/bin/bash
# field-vector kernels: tests + roofline lines at the sizes DESIGN.md quotes.  Outputs under gpurun_out/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-fv}
mkdir -p "$OUT"
echo "== pytest fieldvec"
timeout 900 python -m pytest tests -q -m gpu -k "fieldvec or field or spmv or mle or horner or sumcheck or keyfiles or replay" --maxfail=5 > "$OUT/pytest_fv.txt" 2>&1; tail -6 "$OUT/pytest_fv.txt"
for spec in "mle_eval 20" "mle_eval 24" "spmv 22" "spmv 20" "horner 20" "horner 24" "sumcheck3 24" "axpy 24" "cross_term 24" "bind 24" "quad_prod 24" "lincomb8 22" "round3 24"; do
  set -- $spec
  timeout 300 python bench.py --workload $1 --log2n $2 --steps 10 --warmup 3 > "$OUT/$1_2p$2.json" 2> "$OUT/$1_2p$2.err"
  python - "$OUT/$1_2p$2.json" $1 $2 <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]; c=d.get("cpu_baseline",{})
    print(f"{sys.argv[2]:>10} 2^{sys.argv[3]}: kernel {d['kernel_ms']:.4f} ms  {r['achieved']:.0f} GB/s  frac {r['frac']:.3f}  matches_cpu={c.get('gpu_matches_cpu')}")
except Exception as e:
    print(sys.argv[2], sys.argv[3], "ERR", e)
PY
done
echo "== done"
echo "== horner A/B (NMX_TUNE_HORNER_TOP: 8 = register-resident 8-coefficient levels (shipped), 4, 1 = round-1 chunk-per-lane only)"
for lg in 16 20 22 24; do for v in 8 4 1; do
  NMX_TUNE_HORNER_TOP=$v timeout 300 python bench.py --workload horner --log2n $lg --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/h.json" 2>/dev/null
  python - "$OUT/h.json" $lg $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"horner 2^{sys.argv[2]} top={sys.argv[3]}: kernel {d['kernel_ms']:.4f} ms  frac {d['roofline']['frac']:.3f}")
PY
done; done
echo "== N>1 code path on one GPU: RCCL world 1, strong sharding of 2^22"
NMX_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --strong --total-log2n 22 --steps 5 --warmup 2 --no-extras 2>/dev/null | tail -1 | cut -c1-900
echo "== done2"
