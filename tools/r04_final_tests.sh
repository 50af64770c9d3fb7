cd $GRAFT_REPO_ROOT
o=gpurun_out/r04t; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $o/tests_all.txt; cat $o/tests_all.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
