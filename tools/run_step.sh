cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
for L in 0 1; do
echo "== balanced bottom $L"
GATLING_BVH_BALANCED_BOTTOM=$L timeout 600 python tools/gpu_variants.py c3 64 - - > gpurun_out/c3_variants_$L.log 2>&1; grep -v amdgpu.ids gpurun_out/c3_variants_$L.log | tail -2
GATLING_BVH_BALANCED_BOTTOM=$L timeout 600 python tools/gpu_variants.py c4 64 - - > gpurun_out/c4_variants_$L.log 2>&1; grep -v amdgpu.ids gpurun_out/c4_variants_$L.log | tail -2
GATLING_BVH_BALANCED_BOTTOM=$L timeout 600 python tools/gpu_variants.py c5 8 - - > gpurun_out/c5_variants_$L.log 2>&1; grep -v amdgpu.ids gpurun_out/c5_variants_$L.log | tail -2
done
