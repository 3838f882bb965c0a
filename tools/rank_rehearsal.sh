#!/bin/bash
# One rank of an N-rank weak-scaling job rehearsed alone on one GPU (tools/rank_rehearsal.py): cycle time on one
# rank, on a middle rank of 8 with plain copies and with RCCL in the place of the links (the library's raw calls, and
# through torch.distributed), on rank 0 of 2, with solve()'s order of norm reads, with the whole cycle as one graph, through
# the C handle, and on a rank of BASELINE's 16385^2 cut into 8 slabs; then the kernel trace of the 8-rank case and the
# timeline of one of its cycles.
# Run on the GPU box from the repository root; results under gpurun_out/reh/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/reh
mkdir -p $O
rm -rf $O/prof_w8 $O/rehearsal.txt
cd /tmp && export TMPDIR=/tmp
for a in "--world 1" "--world 8 --comm copy" "--world 8 --comm rccl" "--world 8 --comm rccl-torch" "--world 4 --comm rccl" "--world 2 --rank 0 --comm rccl" \
         "--world 8 --comm rccl --solve-order" "--world 8 --comm rccl --engine-args {\"graph_comm\":true}" "--world 8 --native" \
         "--world 8 --comm rccl --strong --exp 13" "--world 8 --comm rccl-torch --strong --exp 13" "--world 8 --native --strong --exp 13" "--world 1"; do
  echo "== tools/rank_rehearsal.py $a" >> $O/rehearsal.txt
  timeout -k 10 170 python3 $R/tools/rank_rehearsal.py $a 2> $O/err.txt | grep '"world"' >> $O/rehearsal.txt || { tail -20 $O/err.txt; exit 1; }
done
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $O/prof_w8 -- python3 $R/tools/rank_rehearsal.py --world 8 --comm rccl --steps 6 > $O/trace_w8.log 2>&1 || { tail -20 $O/trace_w8.log; exit 1; }
python3 $R/tools/rank_cycle_timeline.py $O/prof_w8/*/*_kernel_trace.csv > $O/timeline_w8.txt
cat $O/rehearsal.txt | cut -c1-260
tail -1 $O/timeline_w8.txt
