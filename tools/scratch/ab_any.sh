cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in modulation_mfcc_amd/libmodmfcc_A.so modulation_mfcc_amd/libmodmfcc.so; do echo $lib; MODMFCC_LIB=$PWD/$lib python tools/scratch/any_pre.py 2>&1 | grep -v amdgpu.ids; done; done
