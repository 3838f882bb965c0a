set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 $R/tools/placement_pmc.py --spacer 40 20 60 > $O/plain.txt 2>&1 || exit 1
cat $O/plain.txt
timeout -s KILL 150 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d $O/tlb -- python3 $R/tools/placement_pmc.py --spacer 40 > $O/tlb.txt 2>&1 || exit 2
timeout -s KILL 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum --output-format csv -d $O/rd -- python3 $R/tools/placement_pmc.py --spacer 40 > $O/rd.txt 2>&1 || exit 3
timeout -s KILL 150 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum --output-format csv -d $O/wr -- python3 $R/tools/placement_pmc.py --spacer 40 > $O/wr.txt 2>&1 || exit 4
cat $O/tlb.txt $O/rd.txt $O/wr.txt | grep state
find $O -name "*.csv" | head
