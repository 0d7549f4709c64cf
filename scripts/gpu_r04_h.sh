#!/bin/bash
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -k "ssr or SSR or sem or dropin or unfiltered or full_size" ) > gpurun_out/r04h_pytest.log 2>&1
tail -6 gpurun_out/r04h_pytest.log
for i in 1 2 3; do timeout 300 python scripts/bench_ssr_frame.py --frames 5 2>&1 | tail -2; done > gpurun_out/r04h_ssr_frame.txt
cat gpurun_out/r04h_ssr_frame.txt
