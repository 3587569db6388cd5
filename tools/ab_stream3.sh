A="--precision 3 --steps 20 --warmup 5 --no-cpu-baseline --no-modes --no-other-configs"
for v in 0 69 70 0 69; do
  python bench.py $A --variant $v 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parity variant $v', j['ms_per_step'], j['roofline']['frac'])"
done
