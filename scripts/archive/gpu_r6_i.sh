export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fieldvec.py tests/test_gpu_large.py -x -q -m gpu -k "fold_chain or compressed or hyperkzg" -p no:cacheprovider 2>&1 | tail -4
for sep in 0 1 0 1; do e=""; [ $sep = 1 ] && e="--separate-folds"
  for l in 14 20; do timeout 600 python bench.py --workload hyperkzg_replay --log2n $l --steps 10 --warmup 3 --no-cpu-baseline $e 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('separate_folds $sep', d['config']['workload'][:34], round(d['value'],3))"; done; done
