export TMPDIR=/tmp
for ser in 0 1 0 1; do e=""; [ $ser = 1 ] && e="--serial-snarks"
  for l in 14 20; do timeout 900 python bench.py --workload compressed_snark_replay --log2n $l --steps 5 --warmup 2 $e 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial_snarks $ser 2^$l:', round(d['value'],3), 'ms py;', d['cpu_baseline']['gpu_matches_cpu'], '; cpp', d['cpp_driver']['ms'], d['cpp_driver']['gpu_matches_cpu'], d['cpp_driver']['groups_ms'])"; done; done
