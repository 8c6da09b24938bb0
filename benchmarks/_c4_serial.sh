cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/c4_serial
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o s -- python $GRAFT_REPO_ROOT/benchmarks/c4_serial_trace.py > $O/log.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/c4_timeline.py $(find $O/t -name "*kernel_trace.csv") 3 > $O/timeline.txt
tail -3 $O/log.txt; cat $O/timeline.txt
