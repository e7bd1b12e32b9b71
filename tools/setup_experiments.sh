cd $GRAFT_REPO_ROOT
echo shared; timeout -k 10 200 python3 tools/kernel_split.py | grep -E "frame|setup|gather|any" || exit 1
echo unshared; RT_SHARE_SHADOW_RAYS=0 timeout -k 10 200 python3 tools/kernel_split.py | grep -E "frame|setup|gather|any"
