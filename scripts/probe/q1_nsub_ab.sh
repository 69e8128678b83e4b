# Q1 back-transformation with super-blocks of 2048 (VIVIT_BT_NSUB=16, default) and 4096 reflectors: bench phases, two rounds
for v in 16 32 16 32; do
  VIVIT_BT_NSUB=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ph = {p['stage'][:14]: round(p['seconds'], 4) for p in d['roofline_phases']}
print('nsub $v', d['ms_per_step'], ph.get('Q1 back-transf'), ph.get('Q2 back-transf'))
"
done
