R=$GRAFT_REPO_ROOT; cd $R
MOGAN_WGRAD2=1 MOGAN_WINO=0 python tools/check_dconv2.py 2>&1 | grep -v "amdgpu.ids\|Warning\|Consider\|ef = \|ew = " | tail -14
echo "== wgrad2"; MOGAN_WGRAD2=1 python tools/time_dconv.py 2>&1 | grep -v amdgpu.ids | cut -c1-140 | head -4
echo "== default"; python tools/time_dconv.py 2>&1 | grep -v amdgpu.ids | cut -c1-140 | head -4
