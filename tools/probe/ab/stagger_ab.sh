for st in 1 0 2 3 1; do
  ECOZ2_VQ_PRE_STAGGER=$st python bench.py --no-cpu-baseline --steps 21 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('stagger $st', 'Gframes/s %.3f' % (d['value']/1e9), 'step %.3f ms' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline']['kernel_ms'], 'steady %.3f' % d['config']['steady_state']['ms_per_step'])"
done
