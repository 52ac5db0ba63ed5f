cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 900 python tools/gpu_variants.py c3 256 - GATLING_POOL_SLOTS=67108864 GATLING_POOL_SLOTS=16777216 > gpurun_out/c3_variants.log 2>&1; grep -v amdgpu.ids gpurun_out/c3_variants.log | tail -5
timeout 900 python tools/gpu_variants.py c4 256 - GATLING_POOL_SLOTS=67108864 GATLING_POOL_SLOTS=16777216 > gpurun_out/c4_variants.log 2>&1; grep -v amdgpu.ids gpurun_out/c4_variants.log | tail -5
