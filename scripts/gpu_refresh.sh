#!/bin/bash
# regenerate the bench JSONs kept under profiles/r01_msm_2p20 and profiles/r01_fieldvec with the current build
set -u
O=gpurun_out/refresh; mkdir -p $O
b() { timeout 300 python bench.py "$@" 2>/dev/null | tail -1; }
for d in random u1 u10 u16 u64 equal zero_rm1 pm_small; do b --dist $d --no-cpu-baseline --steps 20 > $O/dist_$d.json; done
for l in 10 12 13 14 16 22 24; do b --log2n $l --no-cpu-baseline --steps 20 > $O/bench_2p${l}_single_gpu.json; done
for it in 1024 65536; do b --workload prove_step_replay --iters $it --steps 20 > $O/replay_$it.json; done
for l in 14 16 20; do b --workload hyperkzg_replay --log2n $l --steps 5 > $O/hkzg_$l.json; done
for c in 1 2 3; do b --curve $c --no-cpu-baseline --steps 20 > $O/bench_curve$c.json; done
mkdir -p $O/fv
for w in axpy cross_term bind sumcheck3 round3 quad_prod lincomb8 horner mle_eval spmv; do for l in 20 24; do
  [ $w = spmv -a $l = 24 ] && l=22; [ $w = lincomb8 -a $l = 24 ] && l=22
  b --workload $w --log2n $l --steps 10 > $O/fv/bench_${w}_$l.json; done; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/refresh/*.json') + glob.glob('gpurun_out/refresh/fv/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'UNREADABLE', e); continue
    r = d.get('roofline') or {}
    print(os.path.basename(f), round(d['ms_per_step'], 4), 'ms', '%.4g' % d['value'], d['unit'], 'frac', round(r.get('frac', 0), 3),
          (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('gpu_matches_cpu'))
PY
