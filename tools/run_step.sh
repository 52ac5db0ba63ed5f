cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python tools/gpu_variants.py c2 256 - GATLING_TRACE_BLOCKS_PER_CU=5 GATLING_TRACE_BLOCKS_PER_CU=7 > gpurun_out/c2_variants.log 2>&1; grep -v amdgpu.ids gpurun_out/c2_variants.log
timeout 600 python tools/gpu_variants.py c1 64 - > gpurun_out/c1_variants.log 2>&1; grep -v amdgpu.ids gpurun_out/c1_variants.log
