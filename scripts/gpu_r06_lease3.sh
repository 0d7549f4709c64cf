#!/bin/bash
# Round 6, lease 3: the new default kernel + chunk coalescing through the tests they touch, the bench line, then the whole GPU suite
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build(); g.smoke()' > $OUT/r06_l3_smoke.txt 2>&1; tail -2 $OUT/r06_l3_smoke.txt
( time timeout 900 python -m pytest tests/test_coalesce_gpu.py tests/test_range_fallback_gpu.py tests/test_train_masks_gpu.py "tests/test_gpu_parity.py::test_object_kernel_tile_forms_are_bit_identical" tests/test_unfiltered_parity.py -m gpu -q -x -s 2>&1 | grep -v "amdgpu.ids" | tail -40 ) > $OUT/r06_l3_targeted.txt 2>&1; tail -25 $OUT/r06_l3_targeted.txt
( time python bench.py --steps 5 --warmup 1 > $OUT/r06_l3_bench.json 2> $OUT/r06_l3_bench.err ) 2>&1 | tail -3; tail -c 1500 $OUT/r06_l3_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06_l3_bench.json").read().strip().split("\n")[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["kernel"])
    print("ref chunking", d["reference_chunking"])
    print("verdicts", {k: (v["verdict"], v["within_the_plain_budget"]) for k, v in d["parity"]["psnr_verdicts"].items()})
    print("per map", {k: (v["systematic_db"], v["expected_db"], v["mean_delta_db_over_targets"], v["delta_db"]) for k, v in d["parity"]["psnr_delta_db_per_map"].items()})
    print("f32k   ", {k: (v["systematic_db"], v["expected_db"], v["mean_delta_db_over_targets"], v["delta_db"]) for k, v in (d["parity"]["psnr_delta_db_per_map_exact_f32_kernel"] or {}).items()})
    t = d["parity"]["trained"]
    print("trained", t["frame"], t["seconds"], {k: (v["systematic_db"], v["expected_db"], v["mean_delta_db_over_targets"], v["delta_db"]) for k, v in t["psnr_delta_db_per_map"].items()})
    print("trained verdicts", {k: v["verdict"] for k, v in t["psnr_verdicts"].items()}, t["fraction_of_rays_within_the_plain_tolerance"], t["device_fp32_oracle_vs_host_fp32_oracle_max_abs"])
    print("ssr", d["configs"]["ssr_room0_320x240"]["ms_per_step"], d["configs"]["ssr_room0_320x240"]["roofline"]["frac"])
    print("fallback", d["f16_range_fallback"])
    print("train", d["train_step"]["ms_per_step"], d["train_step"]["graphed_ms_per_step"])
except Exception as e:
    print("bench json:", repr(e))
PY
( time timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -15 ) > $OUT/r06_l3_pytest_gpu.txt 2>&1; tail -8 $OUT/r06_l3_pytest_gpu.txt
