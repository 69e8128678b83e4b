#!/bin/bash
# accuracy / time of the bf16-pipe Gram SYRK against the length of its MFMA accumulation chain (VIVIT_BX_FLUSH)
mkdir -p gpurun_out/flush
for f in 8192 4096 2048 1024 512; do
  VIVIT_PREC_TIME=1 VIVIT_BX_FLUSH=$f python tests/gram_precision_child.py gpurun_out/flush/bx_$f.json 5120:401408 1280:401408 8192:65536 || exit 1
done
VIVIT_PREC_TIME=1 VIVIT_GEMM_SPLIT=0 python tests/gram_precision_child.py gpurun_out/flush/f32.json 5120:401408 1280:401408 8192:65536 || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/flush/*.json')):
    d=json.load(open(f))
    for c in d['cases']:
        print(f.split('/')[-1], c['n'], c['K'], 'off_rms %.2e slope %+.2e diag_rms %.2e diag_mean %+.2e ms %.2f' % (c['offdiag_rms'], c['offdiag_slope'], c['diag_rms'], c['diag_mean'], c['ms']))
PY
