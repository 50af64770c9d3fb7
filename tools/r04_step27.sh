cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04w
SBMC_SMALL_KERNELS= timeout 600 python tools/find_small_ops.py 8 > gpurun_out/r04w/small_kernels_rank8.txt 2>&1
grep -v "^MIOpen\|^\[W\|amdgpu.ids" gpurun_out/r04w/small_kernels_rank8.txt | head -90
