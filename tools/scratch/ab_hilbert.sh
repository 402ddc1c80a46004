cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in "$@"; do MODMFCC_LIB=$PWD/$lib python tools/scratch/hilbert_time.py 2>&1 | grep -v amdgpu.ids; done; done
