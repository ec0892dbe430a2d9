#!/bin/bash
# One parametrised evidence script (replaces the round-5 one-off call scripts tools/r05_*.sh).  Run on the GPU box through gpurun:
#   gpurun --timeout 3000 -- 'bash tools/run_evidence.sh <tag> <step> [<step> ...]'
# Everything lands under gpurun_out/ as <tag>_<what>; copy what is to be judged into profiles/ (python tools/install_round.py <tag>).
# Steps:
#   suite        python -m pytest tests -x -q -m gpu                       -> <tag>_gpu_suite.txt
#   smoke        __graft_entry__.smoke()                                   -> <tag>_smoke.txt
#   bench        the driver's command (--steps 20 --warmup 5), PMC rows kept -> <tag>_bench_final.json/.log, <tag>_pmc_final/
#   trace        the bench under rocprofv3 --kernel-trace --stats (resident legs only: every launch of the k = 21 instance but the
#                cpu-baseline sample's is a full-size one, so the stats row's average is the kernel time) -> <tag>_final_bench_kernel_stats.csv
#   repeats      tools/repeats_rates.py at 3 Gb, k = 21 and 31, levels 0/1/3/10 -> <tag>_repeats_rates.txt
#   repeats_trace  main / rest kernel split of the repeats worlds (rocprofv3 --kernel-trace --stats) -> <tag>_repeats_trace.txt
#   valu         wave-level VALU / SALU / LDS instructions per launch (PMC)  -> <tag>_valu.txt
#   rehearse     bench.py --gpus 2 on one device (MFX_BENCH_REHEARSE=1: plumbing of the N > 1 line) -> <tag>_bench_rehearse_2ranks.json
#   isa          tools/isa_report.py (no GPU needed)                          -> <tag>_isa_hot_loop.txt
set -u
TAG=$1; shift
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for STEP in "$@"; do
  case $STEP in
    suite)  ( timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -vE "^(ROCm|Hostname|Librccl|RCCL|HIP)" | tail -15 ) > $OUT/${TAG}_gpu_suite.txt; tail -2 $OUT/${TAG}_gpu_suite.txt ;;
    smoke)  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; tail -1 $OUT/${TAG}_smoke.txt ;;
    bench)  rm -rf $OUT/${TAG}_pmc_final; S=$SECONDS
            MFX_BENCH_KEEP_PMC=$OUT/${TAG}_pmc_final python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_final.json 2> $OUT/${TAG}_bench_final.log
            echo "bench wall $((SECONDS - S)) s" | tee -a $OUT/${TAG}_bench_final.log ;;
    trace)  D=/tmp/kt_bench; rm -rf $D
            ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o kt -- python $ROOT/bench.py --steps 10 --warmup 3 --no-pmc --no-cpu-baseline --no-e2e --no-full-index --no-streamed --no-k31 --no-repeats ) > $OUT/${TAG}_final_trace.log 2>&1
            F=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/${TAG}_final_bench_kernel_stats.csv; rm -rf $D
            grep -E "mfx_hist" $OUT/${TAG}_final_bench_kernel_stats.csv | cut -c1-200 ;;
    repeats) ( for K in 21 31; do echo "== k = $K"; python tools/repeats_rates.py --bases 3e9 --levels 0,1,3,10 --k $K 2>&1 | grep -E "^level|^\{" ; done ) > $OUT/${TAG}_repeats_rates.txt; grep "^level\|^==" $OUT/${TAG}_repeats_rates.txt ;;
    repeats_trace) rm -f $OUT/${TAG}_repeats_trace.txt; GRAFT_REPO_ROOT=$ROOT bash tools/repeats_trace.sh gpurun_out/${TAG}_repeats_trace.txt 21 0,3,10; GRAFT_REPO_ROOT=$ROOT bash tools/repeats_trace.sh gpurun_out/${TAG}_repeats_trace.txt 31 0,3,10; cat $OUT/${TAG}_repeats_trace.txt ;;
    valu)   rm -f $OUT/${TAG}_valu.txt; GRAFT_REPO_ROOT=$ROOT bash tools/pmc_valu.sh gpurun_out/${TAG}_valu.txt 21 k21=default; GRAFT_REPO_ROOT=$ROOT bash tools/pmc_valu.sh gpurun_out/${TAG}_valu.txt 31 k31=default; cat $OUT/${TAG}_valu.txt ;;
    rehearse) MFX_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --bases 1e9 --no-pmc --no-e2e --no-k31 --no-full-index --no-repeats > $OUT/${TAG}_bench_rehearse_2ranks.json 2> $OUT/${TAG}_bench_rehearse_2ranks.log; tail -c 400 $OUT/${TAG}_bench_rehearse_2ranks.json ;;
    isa)    python tools/isa_report.py > $OUT/${TAG}_isa_hot_loop.txt; grep -A3 "21, 4, 6, false\|31, 4, 4, false" $OUT/${TAG}_isa_hot_loop.txt | head -10 ;;
    *) echo "unknown step $STEP" ;;
  esac
done
