set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o run -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
cd $ROOT
python scripts/step_timeline.py $OUT/trace 3 --all > $OUT/timeline_all.txt 2>&1
python scripts/step_timeline.py $OUT/trace 3 > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
