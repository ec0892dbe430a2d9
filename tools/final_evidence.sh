#!/bin/bash
# The round's closing evidence in one gpurun call: the GPU suite, then tools/profile.sh (default bench + the same under rocprofv3 --kernel-trace --stats).
# bash tools/final_evidence.sh <tag>   ->  gpurun_out/gpu_pytest.log, gpurun_out/prof_<tag>/ ; then: python tools/install_evidence.py <tag> <name>
TAG=${1:-final}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gpu_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_pytest.log
tail -4 gpurun_out/gpu_pytest.log
/usr/bin/time -v -o gpurun_out/bench_wall_$TAG.txt bash tools/profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
grep -E "Elapsed|Maximum resident" gpurun_out/bench_wall_$TAG.txt
tail -3 gpurun_out/profile_$TAG.log
