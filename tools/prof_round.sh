#!/bin/bash
# Produces the judged profile artefacts of a round under gpurun_out/round/ (copy to profiles/ afterwards):
#   bench.json            python bench.py (default flags)
#   bench_c3.json / _c5   python bench.py --config 3 / 5 (BASELINE configs #3, #5)
#   bench_2ranks.json     python bench.py --gpus 2 --backend gloo --share-gpu (rehearsal of the multi-rank path on one GPU)
#   bench_cv24.json       python bench.py --semantics cv24 (the OpenCV-2.4 evaluation of the same call, 720p, alt)
#   bench_rccl1.json      python bench.py --force-dist --backend nccl (the RCCL code path with one rank)
#   phase.txt             share of wave time per k_cascade phase (HC_PHASE build, if ugproj-detector_amd/libhip_cascade_PH.so exists)
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of the default command
#   pmc_*                 separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ+LDS counters), no tracing mixed in
#   lds.json              LDS-array / bank-conflict / VALU shares of k_cascade from those counters (bench.py roofline.lds)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/round
rm -rf $O; mkdir -p $O
cd $R && python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench done" > $O/progress.txt
timeout -k 10 400 python bench.py --config 3 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?" >> $O/progress.txt
timeout -k 10 600 python bench.py --config 5 --steps 10 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/progress.txt
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --share-gpu --batch 256 --steps 20 --cpu-sample 0 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "2ranks rc=$?" >> $O/progress.txt
timeout -k 10 300 python bench.py --semantics cv24 --cpu-sample 0 > $O/bench_cv24.json 2> $O/bench_cv24.err; echo "cv24 rc=$?" >> $O/progress.txt
# the N > 1 code path (process group, RCCL all-gather of box records, gather pipeline) with ONE rank: must match the plain run
MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 timeout -k 10 300 python bench.py --force-dist --backend nccl --cpu-sample 0 --no-extra > $O/bench_rccl1.json 2> $O/bench_rccl1.err; echo "rccl1 rc=$?" >> $O/progress.txt
if [ -f $R/ugproj-detector_amd/libhip_cascade_PH.so ]; then
  HC_LIB_PATH=$R/ugproj-detector_amd/libhip_cascade_PH.so timeout -k 10 200 python bench.py --steps 3 --cpu-sample 0 --no-extra 2>&1 >/dev/null | grep "^\[phase" | tail -2 > $O/phase.txt
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --cpu-sample 0 --no-extra > $O/trace.log 2>&1 || exit 1
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv
echo "trace done" >> $O/progress.txt
ARGS="$R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-timing-events --no-extra"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_fetch -- python3 $ARGS > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -- python3 $ARGS > $O/pmc_write.log 2>&1 || exit 1
echo "traffic passes done" >> $O/progress.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq1 -- python3 $ARGS > $O/pmc_sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq2 -- python3 $ARGS > $O/pmc_sq2.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $O > $O/pmc_summary.txt
python3 $R/tools/lds_from_pmc.py $O/pmc_summary.txt $O/lds.json > /dev/null    # roofline.lds of bench.py reads profiles/rNN_lds.json
FRAMES=$(python3 -c "import json; print(json.loads(open('$O/bench.json').read().strip().splitlines()[-1])['config']['frames_per_gpu_per_step'])")
python3 $R/tools/traffic_from_pmc.py $O $O/traffic.json $FRAMES > /dev/null
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2    # raw csv are large; the summaries stay
cat $O/bench.json; head -8 $O/kernel_stats.csv | cut -c1-160; cat $O/traffic.json; cat $O/progress.txt
