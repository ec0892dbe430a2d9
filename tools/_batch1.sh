mkdir -p gpurun_out
g++ -O2 -std=c++17 tools/variants_host_bench.cpp -Imerfin_amd/csrc -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Lmerfin_amd -lmerfin_amd -Wl,-rpath,$PWD/merfin_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/vhb 2> gpurun_out/s3_vhb.log
for i in 1 2; do MFX_VAR_TIMING=1 /tmp/vhb 3e9 5 1 >> gpurun_out/s3_vhb.log 2>&1; done
MFX_VAR_TIMING=1 timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s3_pytest2.log 2>&1; echo rc=$? >> gpurun_out/s3_pytest2.log
tail -5 gpurun_out/s3_pytest2.log; tail -8 gpurun_out/s3_vhb.log
