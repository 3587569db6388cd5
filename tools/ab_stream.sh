A="--steps 20 --warmup 5 --no-cpu-baseline --no-modes --no-other-configs"
for v in 0 63 0 63; do
  python bench.py --config big $A --variant $v 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('big variant $v', j['ms_per_step'])"
done
for v in 0 63 0 63; do
  python bench.py --config big16 --batch 8 --precision 2 $A --variant $v 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('big16 B=8 f16 variant $v', j['ms_per_step'])"
done
