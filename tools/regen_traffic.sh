#!/bin/bash
# HBM traffic per launch of the finest-level kernels: two rocprofv3 counter passes over tools/pmc_probe.py
# (FETCH_SIZE and WRITE_SIZE apart: one run cannot hold both), summarised into gpurun_out/traffic.json, which is
# copied to profiles/traffic.json by hand.  Run on the GPU box from the repository root.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2_traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/pmc_probe.py > /dev/null
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/pmc_probe.py > /dev/null
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cfetch -- python3 $R/tools/pmc_probe.py --cycle > /dev/null
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cwrite -- python3 $R/tools/pmc_probe.py --cycle > /dev/null
cd $R
python3 tools/pmc_summary.py $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) \
    --cycle-fetch $(ls $O/cfetch/*/*counter_collection.csv | head -1) --cycle-write $(ls $O/cwrite/*/*counter_collection.csv | head -1) > gpurun_out/traffic.json
cp $(ls $O/cfetch/*/*counter_collection.csv | head -1) gpurun_out/r2_cycle_pmc_fetch_size.csv
cp $(ls $O/cwrite/*/*counter_collection.csv | head -1) gpurun_out/r2_cycle_pmc_write_size.csv
cp $(ls $O/fetch/*/*counter_collection.csv | head -1) gpurun_out/r2_probe_pmc_fetch_size.csv
cp $(ls $O/write/*/*counter_collection.csv | head -1) gpurun_out/r2_probe_pmc_write_size.csv
python3 - <<PY
import json
d=json.load(open("gpurun_out/traffic.json"))
for k,v in d["kernels"].items():
    print(k, "fetch %.2f GB write %.2f GB" % (v["fetch_bytes"]/1e9, v["write_bytes"]/1e9), "over own %.3f" % v.get("hbm_over_own", 0))
print("cycle", d.get("cycle"))
PY
