cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for m in 0 1; do echo "MM_FUSE_TAIL=$m"; MM_FUSE_TAIL=$m timeout -k 10 120 python tools/time_workload.py $1 200 2>&1 | grep -v amdgpu.ids; done; done
