# same-box A/B of the fused expert MLP alone: libmdm_hip_base.so (kept beside the library) against the current build
cd $GRAFT_REPO_ROOT
P=motiondiffusion-moe_amd
cp $P/libmdm_hip.so /tmp/new.so
for r in 1 2; do
  for v in base new; do
    if [ $v = base ]; then cp $P/libmdm_hip_base.so $P/libmdm_hip.so; else cp /tmp/new.so $P/libmdm_hip.so; fi
    python tools/mlp_ko.py 0 only16 2>/dev/null | sed "s/^/$v /"
  done
done
cp /tmp/new.so $P/libmdm_hip.so
