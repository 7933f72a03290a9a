cd $GRAFT_REPO_ROOT
timeout 600 python scripts/tmp/fused_prof.py 2>&1 | grep -v amdgpu.ids | head -8
echo "== parts (fused)"
timeout 300 python scripts/tmp/hkzg_parts.py 2>&1 | grep -v amdgpu.ids | grep -E "batch_commit all|3 opens"
bash scripts/gpu_r2_check.sh fused
