#!/bin/bash
# A/B of library variants on the bench's timed region: tools/probe/ab/run_ab.sh <variant>...   ("base" = product build)
for v in "$@"; do
  if [ "$v" = base ]; then unset ECOZ2VQ_LIB; else export ECOZ2VQ_LIB=$PWD/tools/probe/ab/$v/libecoz2vq.so; fi
  python bench.py --no-cpu-baseline --steps 21 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v', 'Gframes/s %.3f' % (d['value']/1e9), 'step %.3f ms' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline']['kernel_ms'], 'steady %.3f' % d['config']['steady_state']['ms_per_step'], 'e2e %.1f ms' % (1e3*d['config']['learn_end_to_end']['seconds']), 'quant %.2f G/s' % (d['config']['quantize_frames_per_sec_device_resident']/1e9))"
done
