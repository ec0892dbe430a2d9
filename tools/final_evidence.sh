#!/bin/bash
# The round's closing evidence in one gpurun call: the GPU suite, then tools/profile.sh (default bench + the same under rocprofv3 --kernel-trace --stats).
# bash tools/final_evidence.sh <tag>   ->  gpurun_out/gpu_pytest.log, gpurun_out/prof_<tag>/ ; then: python tools/install_evidence.py <tag> <name>
TAG=${1:-final}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd $REPO
if [ -z "$MFX_SKIP_PYTEST" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gpu_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_pytest.log
  tail -4 gpurun_out/gpu_pytest.log
fi
T0=$SECONDS
bash tools/profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
echo "tools/profile.sh (the default bench run + the same under rocprofv3): $((SECONDS - T0)) s" | tee gpurun_out/bench_wall_$TAG.txt
tail -3 gpurun_out/profile_$TAG.log
