cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc/p1 -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --instructions 300 > $R/gpurun_out/pmc/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc/p2 -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --instructions 300 > $R/gpurun_out/pmc/p2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc/p3 -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --instructions 300 > $R/gpurun_out/pmc/p3.log 2>&1
ls -R $R/gpurun_out/pmc | head -30
