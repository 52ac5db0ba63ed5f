cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python tools/gpu_variants.py c2 256 - GATLING_SAMPLE_BUFFER_MB=4096 GATLING_SAMPLE_BUFFER_MB=8192 GATLING_SAMPLE_BUFFER_MB=8192,GATLING_POOL_SLOTS=2097152 GATLING_SAMPLE_BUFFER_MB=8192,GATLING_POOL_SLOTS=8388608 GATLING_SAMPLE_BUFFER_MB=1024 > gpurun_out/c2_variants.log 2>&1; grep -v amdgpu.ids gpurun_out/c2_variants.log
