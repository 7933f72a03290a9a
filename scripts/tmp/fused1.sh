cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_batch_fused.py tests/test_gpu_parity.py -x -q 2>&1 | tail -15
echo "== parts (fused)"
timeout 300 python scripts/tmp/hkzg_parts.py 2>&1 | grep -v amdgpu.ids | grep -E "batch_commit|folds"
echo "== parts (no fuse)"
NMX_TUNE_NO_BATCH_FUSE=1 timeout 300 python scripts/tmp/hkzg_parts.py 2>&1 | grep -v amdgpu.ids | grep -E "batch_commit all"
