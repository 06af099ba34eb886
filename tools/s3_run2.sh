mkdir -p gpurun_out/s3
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s3/gputest.log 2>&1; tail -4 gpurun_out/s3/gputest.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o b8 -- python $R/tools/profile_forward.py --batch 8 --graph --reps 5 > /tmp/kt.log 2>&1; tail -2 /tmp/kt.log
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -60 "$f" > $R/gpurun_out/s3/b8_kernel_stats.csv
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_iteration.py "$f" > $R/gpurun_out/s3/b8_iteration.txt 2>&1
python $R/tools/trace_frame.py "$f" encoder > $R/gpurun_out/s3/b8_encoder.txt 2>&1
