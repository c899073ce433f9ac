#!/bin/bash
# Round 6's evidence in one gpurun call (on ONE build): rocprofv3 trace + PMC passes of the bench command
# (profiles/collect.sh), the bench line, the per-row timings (section 8 rows + the reference's own benchmarks), the C5
# share's bench line, one hop of the ring through host memory / device memory of one process (ring_hop) and through
# another PROCESS's device memory (ipc_hop), four Fits in flight (conc4_probe), the one-launch Fit's own measurements
# (the reference's benchmark shapes, where the launch's iteration goes, the default's choice against the general path,
# the two microbenchmarks its design rests on).  Summaries are copied under gpurun_out/
# for the way back (gpurun merges gpurun_out/ only); the raw CSVs are dropped (tens of MB).
TAG=${1:-r06g}
mkdir -p gpurun_out
bash profiles/collect.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
echo collect rc=$?
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.json profiles/${TAG}_fetch_probe.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_$TAG
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
echo bench rc=$?
timeout -k 10 900 python tests/perf_rows.py > gpurun_out/${TAG}_rows.json 2> gpurun_out/${TAG}_rows.err
echo rows rc=$?
timeout -k 10 600 python tests/perf_rows_ref.py >> gpurun_out/${TAG}_rows.json 2>> gpurun_out/${TAG}_rows.err
echo rows_ref rc=$?
timeout -k 10 600 python bench.py --workload c5 --steps 40 --warmup 20 > gpurun_out/${TAG}_bench_c5_n1.json 2> gpurun_out/${TAG}_bench_c5_n1.err
echo c5 rc=$?
{ timeout -k 5 120 tools/micro/ring_hop.bin; timeout -k 5 120 tools/micro/ipc_hop.bin; } > gpurun_out/${TAG}_ring_hop.txt 2>&1
echo hop rc=$?
timeout -k 10 200 python tools/conc4_probe.py 4 2>&1 | tail -1 > gpurun_out/${TAG}_conc4.txt
echo conc4 rc=$?
{
  for n in 1024 4096 16384; do timeout -k 10 100 python tools/small_fit_probe.py $n 10 2>&1 | grep points; done
  if [ -f experiments/ab/libpcgx_stamps.so ]; then
    for n in 1024 4096 16384; do echo "stamps, $n points:"; PCGX_STAMPS_POINTS=$n PCGX_LIB=experiments/ab/libpcgx_stamps.so timeout -k 10 120 python tools/stamps.py small_fit 2>&1 | grep -v amdgpu.ids; done
  fi
  for s in 1 0; do PCGX_ICP_SMALL=$s timeout -k 10 300 python tools/small_vs_general.py 2>&1 | grep PCGX_ICP; done
  timeout -k 5 120 tools/micro/xcd_pingpong.bin
  timeout -k 5 120 tools/micro/dpp_chain.bin
} > gpurun_out/${TAG}_small_fit.txt 2>&1
echo small rc=$?
# rocprofv3's own durations of the one-launch Fit's kernels at the three benchmark sizes (host-pointer Fits only)
export TMPDIR=/tmp
: > gpurun_out/${TAG}_small_kernel_stats.csv
for n in 1024 4096 16384; do
  rm -rf gpurun_out/prof_small_$n
  PCGX_PROBE_HOST_ONLY=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small_$n -- python3 tools/small_fit_probe.py $n 10 > /dev/null 2>&1
  f=$(find gpurun_out/prof_small_$n -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && { echo "# $n points, ten host-pointer Fits of 10 iterations"; head -1 "$f"; grep -E 'icp_small_fit|small_prepare|small_order' "$f"; } >> gpurun_out/${TAG}_small_kernel_stats.csv
  rm -rf gpurun_out/prof_small_$n
done
echo small_stats rc=$?
for mem in dev host; do
  for n in 2 4 5; do
    [ $mem = host ] && [ $n = 5 ] && continue
    PCGX_RING_MEM=$mem PCGX_BENCH_REHEARSE=1 timeout -k 10 200 python bench.py --gpus $n --steps 100 --warmup 20 --points 125000 > gpurun_out/${TAG}_rehearse_${mem}_n${n}.json 2> gpurun_out/${TAG}_rehearse_${mem}_n${n}.err
    echo "rehearsal $mem n=$n rc=$?"
  done
done
tail -3 gpurun_out/${TAG}_collect.log
