#!/bin/bash
# round 6, lease h: the inner-claim form of derive_from_claim (no per-round inversion): parity, then the Spartan replay and the chained replay
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6h
mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_spartan.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 | tee "$OUT/pytest.txt"
for l in 14 20 14 20; do
  timeout 300 python bench.py --workload spartan_replay --log2n $l --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/spartan_$l.json" 2>> "$OUT/err.txt"
  python - "$OUT/spartan_$l.json" $l <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"2^{sys.argv[2]}: {d['value']:.3f} ms", {k: v for k, v in d['breakdown_ms'].items() if k.startswith('sumcheck')},
      {k: (v['wait_ms'], v['host_algebra_ms']) for k, v in d['provers'].items()})
PY
done | tee "$OUT/spartan.txt"
for l in 14 20; do
  timeout 900 python bench.py --workload compressed_snark_replay --log2n $l --steps 5 --warmup 2 > "$OUT/csnark_$l.json" 2> "$OUT/csnark_$l.err"
  python - "$OUT/csnark_$l.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"][:64], "->", round(d["value"], 3), "ms; matches", d["cpu_baseline"]["gpu_matches_cpu"], "cpp", {k: v for k, v in d["cpp_driver"].items() if k != "what"})
PY
done | tee "$OUT/csnark.txt"
