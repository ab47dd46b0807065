# usage (on the GPU box, via gpurun): bash tools/profile_round.sh <tag> [bench args]
# One round's profile evidence of the default bench command (the full 1000-instruction headline batch):
#   bench line, rocprofv3 kernel trace + stats (make_profiles.py also writes steady-state averages: the first batch of the process dropped), HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), SQ counters (three
#   passes of <= 8 counters; TA_* passes hang rocprofv3 on this pool and are left out).  Every pass has its own timeout.
# Raw output goes to gpurun_out/<tag>/; tools/make_profiles.py <tag> turns it into the committed summaries under profiles/.
TAG=${1:-r2}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-sample 0 --no-copy-ceiling $*"
python3 $R/bench.py --steps 5 --warmup 2 $* > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 6 --warmup 2 > $OUT/trace.log 2>&1; echo "trace rc=$?"
run() { n=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- $B --steps 1 --warmup 0 > $OUT/$n.log 2>&1; echo "$n rc=$?"; }
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS &&
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR &&
run p3 SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LEVEL_WAVES GRBM_GUI_ACTIVE
find $OUT -name "*.csv" | wc -l
