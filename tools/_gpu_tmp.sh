cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 tools/wstat.py 1920 1080 64 2>&1 | grep wstat
