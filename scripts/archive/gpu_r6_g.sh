#!/bin/bash
# round 6, lease g: indices per lane in the big sum-check passes (NMX_SC_BIG_IPT x NMX_SC_BIG_FROM), Spartan replay at 2^20, alternating
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6g
mkdir -p "$OUT"
for rep in 1 2; do
for cfg in "0 16" "1 17" "1 16" "1 15" "1 14" "2 17" "2 16" "0 16"; do
  set -- $cfg
  NMX_SC_BIG_IPT=$1 NMX_SC_BIG_FROM=$2 timeout 300 python bench.py --workload spartan_replay --log2n 20 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/s.json" 2>> "$OUT/err.txt"
  python - "$OUT/s.json" "$cfg" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"ipt/from {sys.argv[2]}: {d['value']:.3f} ms", {k: v for k, v in d['breakdown_ms'].items() if k.startswith('sumcheck')})
PY
done; done | tee "$OUT/big_ipt_ab.txt"
