#!/bin/bash
# Round 6, lease 4: whole GPU suite + bench on the code with per-chunk range words
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build(); g.smoke()' > $OUT/r06_l4_smoke.txt 2>&1; tail -1 $OUT/r06_l4_smoke.txt
( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -25 ) > $OUT/r06_l4_pytest_gpu.txt 2>&1; tail -12 $OUT/r06_l4_pytest_gpu.txt
timeout 300 python -m pytest tests/test_unfiltered_parity.py -m gpu -q -s -k "most_rays" 2>&1 | grep "rays end to end" > $OUT/r06_l4_trained_fraction.txt; cat $OUT/r06_l4_trained_fraction.txt
( time python bench.py --steps 5 --warmup 1 > $OUT/r06_l4_bench.json 2> $OUT/r06_l4_bench.err ) 2>&1 | tail -3; tail -c 800 $OUT/r06_l4_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06_l4_bench.json").read().strip().split("\n")[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["kernel"])
    print("ssr", d["configs"]["ssr_room0_320x240"]["ms_per_step"], d["configs"]["ssr_room0_320x240"]["roofline"]["frac"])
    f = d["f16_range_fallback"]; print("fallback", f["vs_f16x3_frame"], f["merged_chunks"])
    print("train", d["train_step"]["ms_per_step"], d["train_step"]["graphed_ms_per_step"])
    print("verdicts", {k: v["verdict"] for k, v in d["parity"]["psnr_verdicts"].items()}, {k: v["verdict"] for k, v in d["parity"]["trained"]["psnr_verdicts"].items()})
except Exception as e:
    print("bench json:", repr(e))
PY
