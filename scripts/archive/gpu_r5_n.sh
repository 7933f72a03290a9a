#!/bin/bash
# round 5, call n: full GPU suite + smoke + default bench line after batch_invert / commit_begin
mkdir -p gpurun_out/r5zh
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r5zh/pytest_gpu.txt
cat gpurun_out/r5zh/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r5zh/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r5zh/bench_default_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5zh/bench_default_line.json"))
print("headline ms", d["ms_per_step"], "value", d["value"])
print("prove_step", d["prove_step_replay_ms"]["ms"], d["prove_step_replay_ms"].get("overlap"))
print("spartan", d["spartan_replay_ms"].get("ms"), "hyperkzg", d["hyperkzg_replay_ms"]["ms"], "trait", d["trait_form"]["ms"])
PY
