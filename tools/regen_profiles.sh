#!/bin/bash
# Regenerates the measurements kept under profiles/ (run on the GPU box from the repository root; everything
# lands in gpurun_out/, from where the summaries are copied into profiles/ by hand).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/regen
mkdir -p $O
cd $R
if [ -z "$SKIP_TRAFFIC" ]; then   # SKIP_TRAFFIC=1: profiles/traffic.json already belongs to these kernel sources
bash tools/regen_traffic.sh > $O/traffic.txt 2>&1
cp gpurun_out/traffic.json gpurun_out/r2_probe_pmc_fetch_size.csv gpurun_out/r2_probe_pmc_write_size.csv $O/
cp gpurun_out/traffic.json profiles/traffic.json   # bench.py prints the table only when it fits the kernel sources
echo pmc done
fi
timeout -k 10 400 python3 bench.py > $O/bench_n14.json
echo bench done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --no-cpu-baseline --no-untuned-leg > /dev/null
cd $R
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/bench_n14_kernel_stats.csv
python3 tools/cycle_timeline.py $(ls $O/prof_bench/*/*kernel_trace.csv | head -1) > $O/cycle_timeline.txt 2>&1 || true
rm -rf $O/prof_bench
echo stats done
timeout -k 10 200 python3 bench.py --batch 1024 --exp 7 --jacobi-levels 4 > $O/bench_batch1024_n7.json
timeout -k 10 200 python3 bench.py --exp 13 --no-cpu-baseline > $O/bench_n13.json
timeout -k 10 200 python3 bench.py --exp 11 --no-cpu-baseline > $O/bench_n11.json
timeout -k 10 200 python3 bench.py --exp 8 --jacobi-levels 4 --no-cpu-baseline > $O/bench_n8.json
echo benches done
timeout -k 10 400 python3 tools/tune_fused.py --unroll 1 --rows 0 32 64 128 > $O/tune_fused.txt 2>&1
timeout -k 10 300 python3 tools/convergence.py --exps 12 14 --eps 0.025 2.5e-6 > $O/convergence.txt 2>&1
timeout -k 10 200 examples/poisson_main 14 10 20 > $O/poisson_main_14_10.txt 2>&1
echo all done
