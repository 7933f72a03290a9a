#!/bin/bash
# round 6: rocprofv3 --kernel-trace --stats of the inner-product argument at 2^14 (bench.py --workload ipa_replay)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6ipastats
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ipa_stats -- python $GRAFT_REPO_ROOT/bench.py --workload ipa_replay --log2n 14 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> /tmp/ipa_stats.err
f=$(find /tmp/ipa_stats -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT/kernel_stats.csv"
head -25 "$OUT/kernel_stats.csv" | cut -c1-150
tail -1 "$OUT/bench_under_rocprof.json" | cut -c1-200
