#!/bin/bash
# Same-box stage times of the headline step under different environment settings (bench.py without the extras, 3 steps each).
#   scripts/probe/bench_env_ab.sh "VIVIT_BT_NSUB=32" "VIVIT_GEMM_SPLIT_KC=8192" ...   (the empty setting runs first and last)
cd "$(dirname "$0")/../.."
run() { env $1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-configs 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[%s]' % (sys.argv[1] or 'default'), '%.0f eigenpairs/s' % d['value'], ' '.join('%s=%.3f' % (p['stage'].split(' ')[0].rstrip(':') + ('/' + p['stage'].split(' ')[1] if p['stage'].startswith('sy2sb') else ''), p['seconds']) for p in d['roofline_phases']))
" "$1"; }
run ""
for s in "$@"; do run "$s"; done
run ""
