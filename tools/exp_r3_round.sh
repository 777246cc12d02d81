set -e
mkdir -p gpurun_out/r3i
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3i/gputest.log 2>&1 || { tail -40 gpurun_out/r3i/gputest.log; exit 1; }
tail -2 gpurun_out/r3i/gputest.log
bash tools/prof_round.sh > gpurun_out/r3i/prof_round.log 2>&1 || { tail -20 gpurun_out/r3i/prof_round.log; exit 1; }
tail -30 gpurun_out/r3i/prof_round.log
bash tools/exp_round_latency.sh > gpurun_out/r3i/latency.log 2>&1 || true
tail -12 gpurun_out/r3i/latency.log
