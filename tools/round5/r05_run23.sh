#!/bin/bash
# full validation of the tree: GPU suite, smoke, default bench line
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > gpurun_out/r05_t23_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r05_t23_smoke.txt
timeout 900 python bench.py 2>gpurun_out/r05_t23_bench.err | tail -1 > gpurun_out/r05_v1_bench.json
cat gpurun_out/r05_t23_pytest.txt gpurun_out/r05_t23_smoke.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_v1_bench.json'))
print(d['value'], d['ms_per_step'], d['step_frac_of_mfma_peak'], d['parity'])
for v in d['other_configs']: print(v['config'], v['ms_per_step'])
print([(s['batch_per_gpu'], round(s['ms_per_step'],3)) for s in d['sweep']])
print(d['roofline'])
PY
