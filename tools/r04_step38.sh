cd $GRAFT_REPO_ROOT
timeout 60 tools/dev/tr16_probe | head -40
