cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python tools/gpu_variants.py c2 256 - - > gpurun_out/c2_variants.log 2>&1; grep -v amdgpu.ids gpurun_out/c2_variants.log
timeout 600 python tools/gpu_variants.py c3 64 - - > gpurun_out/c3_variants.log 2>&1; grep -v amdgpu.ids gpurun_out/c3_variants.log
