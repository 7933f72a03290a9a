cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_batch_fused.py -x -q 2>&1 | tail -5
timeout 600 python scripts/tmp/fused_prof.py 2>&1 | grep -v amdgpu.ids
echo "== parts (fused)"
timeout 300 python scripts/tmp/hkzg_parts.py 2>&1 | grep -v amdgpu.ids | grep -E "batch_commit"
