# usage (on the GPU box, via gpurun): bash tools/profile_round.sh <tag>
# kernel trace + HBM traffic counters of the default bench command; summaries go to gpurun_out/<tag>/
TAG=${1:-r1}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head
