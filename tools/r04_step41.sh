cd $GRAFT_REPO_ROOT
bash tools/prof_pointwise.sh > /tmp/pp.log 2>&1; grep "pw_bwd_kernel<128" gpurun_out/profiles_pw/r02_pointwise_pmc.txt | cut -c1-230
