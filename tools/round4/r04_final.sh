#!/bin/bash
# final validation of the round-4 tree: full GPU suite, smoke(), default bench line
python -m pytest tests -m gpu -q > gpurun_out/r04_final.log 2>&1
tail -n 6 gpurun_out/r04_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err
tail -c 600 gpurun_out/r04_final_bench.json
