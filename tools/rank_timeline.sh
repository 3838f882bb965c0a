#!/bin/bash
# Kernel and copy trace of rank 0 of a 2-rank run that shares one GPU (gloo, messages staged through the
# host): shows the interior launch of a cut pass running while the halo rows travel.  Run on the GPU box
# from the repository root; rank 0 runs under rocprofv3 directly, rank 1 beside it.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 WORLD_SIZE=2
ARGS="--gpus 2 --backend gloo --exp 12 --steps 4 --warmup 2 --no-tune-placement"
RANK=1 LOCAL_RANK=1 timeout -k 10 280 python3 $R/bench.py $ARGS > $O/tl_rank1.log 2>&1 &
P1=$!
RANK=0 LOCAL_RANK=0 timeout -k 10 280 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_ranks -- python3 $R/bench.py $ARGS > $O/tl_rank0.log 2>&1
wait $P1
echo timeline done
