cd $GRAFT_REPO_ROOT
for lib in "$@"; do echo "== $lib"; MODMFCC_LIB=$PWD/$lib timeout -k 10 200 python tools/scratch/typed_refdefault.py 2>&1 | grep -v amdgpu.ids | grep "any-length"; MODMFCC_LIB=$PWD/$lib timeout -k 10 200 python tools/any_time.py 2>&1 | grep "form 0" | grep "^400\|^600\|^800\|^1000\|^2000"; done
