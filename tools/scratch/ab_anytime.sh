cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in "$@"; do echo "== $lib"; MODMFCC_LIB=$PWD/$lib timeout -k 10 200 python tools/any_time.py 2>&1 | grep "form 0" | grep -v "^512\|502\|499"; done; done
