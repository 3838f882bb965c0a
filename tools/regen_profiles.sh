#!/bin/bash
# Regenerates the measurements kept under profiles/ (run on the GPU box from the repository root;
# everything lands in gpurun_out/, from where the summaries are copied by hand).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $O/bench_n14.json
echo bench done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --no-cpu-baseline > /dev/null
echo stats done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/pmc_probe.py > /dev/null
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/pmc_probe.py > /dev/null
echo pmc done
timeout -k 10 200 python3 $R/bench.py --batch 1024 --exp 7 --jacobi-levels 4 > $O/bench_batch1024_n7.json
timeout -k 10 200 python3 $R/bench.py --exp 13 --no-cpu-baseline > $O/bench_n13.json
timeout -k 10 200 python3 $R/bench.py --exp 11 --no-cpu-baseline > $O/bench_n11.json
timeout -k 10 200 python3 $R/bench.py --exp 8 --jacobi-levels 4 --no-cpu-baseline > $O/bench_n8.json
echo benches done
timeout -k 10 400 python3 $R/tools/tune_fused.py --unroll 1 --rows 32 64 128 > $O/tune_fused.txt 2>&1
echo tune done
