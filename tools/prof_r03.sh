# Round-3 profiling runs on the GPU box (outputs under gpurun_out/, summaries are copied to profiles/ afterwards):
#   bash tools/prof_r03.sh stats     kernel trace + --stats of the default bench command
#   bash tools/prof_r03.sh pmc       HBM-traffic PMC passes of the headline (own run, kernel trace only)
#   bash tools/prof_r03.sh sec       PMC passes of the secondary kernels k_iter4 / k_icgn_iter
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
what=${1:-stats}
if [ "$what" = stats ]; then
  mkdir -p gpurun_out/prof_stats
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o r03 -- python3 bench.py --no-secondary --cpu-seconds 0 > gpurun_out/r03_bench_under_rocprof.json 2> gpurun_out/r03_bench_under_rocprof.err
  ls gpurun_out/prof_stats
fi
if [ "$what" = pmc ]; then
  mkdir -p gpurun_out/pmc
  ICTR_TEAM_TIMEOUT_S=1 timeout -k 10 500 python3 tools/check_pmc.py --run profiles/pmc_traffic.txt -d gpurun_out/pmc -- python3 bench.py --steps 2 --warmup 1 --no-secondary --cpu-seconds 0 --no-events > gpurun_out/r03_pmc_run.log 2>&1
  python3 tools/summarize_pmc_kernel.py gpurun_out/pmc k_level_res > gpurun_out/r03_pmc_resident.txt
  python3 tools/summarize_pmc_kernel.py gpurun_out/pmc k_ref8 >> gpurun_out/r03_pmc_resident.txt
  cat gpurun_out/r03_pmc_resident.txt
fi
if [ "$what" = sec ]; then
  mkdir -p gpurun_out/pmc_sec
  cat profiles/pmc_traffic.txt profiles/pmc_ea.txt > gpurun_out/pmc_sec_in.txt
  echo "pmc: SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" >> gpurun_out/pmc_sec_in.txt
  echo "pmc: TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" >> gpurun_out/pmc_sec_in.txt
  timeout -k 10 600 python3 tools/check_pmc.py --run gpurun_out/pmc_sec_in.txt -d gpurun_out/pmc_sec -- python3 tools/sec_one.py psz4 c3 c5 > gpurun_out/r03_pmc_sec_run.log 2>&1
  for k in k_iter4 k_icgn_iter k_icgn_hess; do python3 tools/summarize_pmc_kernel.py gpurun_out/pmc_sec $k; done > gpurun_out/r03_pmc_secondary.txt
  tail -40 gpurun_out/r03_pmc_secondary.txt
fi
