cd $GRAFT_REPO_ROOT
P=motiondiffusion-moe_amd
cp $P/libmdm_hip.so /tmp/new.so
for r in 1 2; do
  for v in base new; do
    if [ $v = base ]; then cp $P/libmdm_hip_base.so $P/libmdm_hip.so; else cp /tmp/new.so $P/libmdm_hip.so; fi
    for cfg in small; do
      python bench.py --config $cfg --no-modes --no-cpu-baseline --no-other-configs --steps 30 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $cfg', j['ms_per_step'])"
    done
  done
done
cp /tmp/new.so $P/libmdm_hip.so
