cd ${GRAFT_REPO_ROOT:-.}
for set in "tests/test_fault_injection.py" "tests/test_gpu_dist.py" "tests/test_direct_solver.py" "tests/test_device_anm.py" "tests/test_device_ops.py tests/test_cli.py"; do
  rm -f gpurun_out/parity_steps_human_arap16.json
  python -m pytest $set "tests/test_gpu_fullsize.py::test_named_config_against_oracle[human_arap16]" -m gpu -q -x > /tmp/o.log 2>&1
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/parity_steps_human_arap16.json")); print("$set", "->", d["device_steps"], d["first_divergence"], d["device"][0])
except Exception as e: print("$set", "-> no record", e)
PY
  grep -E "passed|failed" /tmp/o.log | tail -1
done
