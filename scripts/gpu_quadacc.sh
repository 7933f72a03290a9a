#!/bin/bash
# quad mixed-add accumulate: parity, then small / mid sizes with and without it.  Short timeouts: DPP code.
set -u
OUT=gpurun_out/quadacc; mkdir -p $OUT
timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blitzar or property" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for l in 10 13 14 16 17 18 20; do for q in 1 0; do
  NMX_TUNE_NO_QUAD_ACCUM=$q timeout 100 python bench.py --log2n $l --steps 30 --warmup 5 --no-cpu-baseline > $OUT/b.json 2>/dev/null; echo "2^$l no_quad=$q"; python scripts/show.py $OUT/b.json
done; done
timeout 200 python bench.py --workload prove_step_replay --iters 1024 --steps 20 --warmup 3 > $OUT/replay_1024.json 2>/dev/null; tail -c 600 $OUT/replay_1024.json
