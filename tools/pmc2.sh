# memory-path counters of the bench kernels (300 instructions); usage on the GPU box: bash tools/pmc2.sh
# few counters per pass: a TA/TCP request that does not fit the hardware aborts rocprofv3 (and then hangs): hence the timeouts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc2
mkdir -p $O
B="python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --instructions 300"
run() { n=$1; shift; timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- $B > $O/$n.log 2>&1; echo "$n rc=$?"; }
run a TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run b TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
run c TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run d SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LEVEL_WAVES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES
run e SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
