# same-box A/B of kernel-selection knobs: tools/ab_variants.sh <config> <variant> [<variant> ...]   (two alternating rounds)
cd $GRAFT_REPO_ROOT
cfg=$1; shift
for r in 1 2; do
  for v in "$@"; do
    python bench.py --config $cfg --variant $v --no-modes --no-cpu-baseline --no-other-configs --steps 30 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg variant $v', j['ms_per_step'])"
  done
done
