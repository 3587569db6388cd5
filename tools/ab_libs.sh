# same-box A/B of two library builds: motiondiffusion-moe_amd/libmdm_hip_base.so (build it from the older sources first) against the
# current libmdm_hip.so; usage (on the GPU box): bash tools/ab_libs.sh "<bench args>" ["<bench args>" ...]
cd $GRAFT_REPO_ROOT
P=motiondiffusion-moe_amd
cp $P/libmdm_hip.so /tmp/new.so
for args in "$@"; do
  for r in 1 2; do
    for v in base new; do
      if [ $v = base ]; then cp $P/libmdm_hip_base.so $P/libmdm_hip.so; else cp /tmp/new.so $P/libmdm_hip.so; fi
      python bench.py $args --no-modes --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v [$args]', j['ms_per_step'])"
    done
  done
done
cp /tmp/new.so $P/libmdm_hip.so
