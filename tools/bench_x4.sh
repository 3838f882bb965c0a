#!/bin/bash
# Three fresh bench.py processes with placement selection and one without, one line each.
set -o pipefail
for i in 1 2 3; do timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_tuned$i.json 2>gpurun_out/r2_bench_tuned$i.err || exit 1; done
timeout -k 10 200 python bench.py --no-cpu-baseline --no-tune-placement > gpurun_out/r2_bench_untuned1.json 2>/dev/null || exit 1
python - <<PY
import json
for f in ("tuned1","tuned2","tuned3","untuned1"):
    d=json.load(open("gpurun_out/r2_bench_%s.json"%f))
    print(f, round(d["ms_per_step"],3), round(d["value"]/1e9,1), "finest", round(d["roofline"]["avg_launch_ms"],3), "jacobi", round(d["roofline"]["single_jacobi_sweep"]["avg_launch_ms"],3), d["config"]["placement_sampling_seconds"], d["config"]["placement"])
PY
