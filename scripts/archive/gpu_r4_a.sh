#!/bin/bash
# Round 4, first pass: the whole GPU suite on the new build (shard-resident scalars, RCCL combine, forced peer copies, full
# slice verification, plain-key partition), the default line, the in-process multi-GPU mode on this 1-GPU box, plain-path A/B.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)_${1:-r4a}
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d.get("stages_ms"))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
echo "== pytest gpu"; timeout 1800 python -m pytest tests -q -m gpu --maxfail=8 > "$OUT/pytest_gpu.txt" 2>&1; grep -E "passed|failed|error" "$OUT/pytest_gpu.txt" | tail -12
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== plain-key path at 2^20 / 2^16 / 2^13: hand-written partition vs rocPRIM"
for lg in 20 16 13; do
  for mode in part sort; do
    case $mode in part) envs="NMX_X=0";; sort) envs="NMX_TUNE_NO_PARTITION=1";; esac
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --log2n $lg --no-tables --no-extras --no-cpu-baseline > "$OUT/plain_${mode}_$lg.json" 2> "$OUT/plain_${mode}_$lg.err"
    echo "-- $mode 2^$lg"; show "$OUT/plain_${mode}_$lg.json"
  done
done
echo "== in-process --gpus 2 on this box (falls back to the devices visible; full-size oracle check)"
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 > "$OUT/inproc_gpus2.json" 2> "$OUT/inproc_gpus2.err"; echo "rc=$?"; cut -c1-1500 "$OUT/inproc_gpus2.json"; tail -3 "$OUT/inproc_gpus2.err"
echo "== in-process, 2 logical devices oversubscribed (NMX_BENCH_OVERSUB=1), RCCL required, 2^22 total"
NMX_BENCH_OVERSUB=1 NMX_BENCH_COMBINE=2 timeout 600 python bench.py --gpus 2 --total-log2n 22 --steps 5 --warmup 2 > "$OUT/inproc_oversub2.json" 2> "$OUT/inproc_oversub2.err"; echo "rc=$?"; cut -c1-2500 "$OUT/inproc_oversub2.json"; tail -3 "$OUT/inproc_oversub2.err"
echo "== default line"; timeout 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"; cut -c1-400 "$OUT/bench_default.json"
python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("stages", d["stages_ms"]); print("trait_form", d.get("trait_form")); print("prove", d.get("prove_step_replay_ms")); print("hkzg", d.get("hyperkzg_replay_ms"))
print({k:(v.get("frac") if isinstance(v,dict) else v) for k,v in d.get("fieldvec",{}).items()})
PY
echo "== done"
