# SQ / TA counters of the bench kernels at the FULL headline batch (1000 instructions); usage on the GPU box: bash tools/pmc_r2.sh [tag]
# few counters per pass (SQ has 8 slots); every pass under its own timeout: an oversize request can hang rocprofv3
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
B="python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0"
run() { n=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- $B > $O/$n.log 2>&1; echo "$n rc=$?"; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS &&
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR &&
run p3 SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LEVEL_WAVES GRBM_GUI_ACTIVE &&
run p4 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
python3 $R/tools/pmc_summary.py $O k_ > $O/summary.txt 2>&1
tail -n 40 $O/summary.txt
