# gpurun -- bash tools/experiments/hazard_hunt/run_hx.sh   (after `make -C dmhomo_amd/csrc hx` in the build container)
cd $GRAFT_REPO_ROOT
for v in hip hx hx_const hx_lb1 hx_pad hx hx_const; do
  DMH_LIB_PATH=$GRAFT_REPO_ROOT/dmhomo_amd/libdmhomo_$v.so python tools/experiments/hazard_hunt/run_hx.py 30 2>&1 | grep -v amdgpu.ids
done
