python bench.py --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['roofline']['frac'], d.get('other_workloads'))"
python tools/sampler_modes.py 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['stars'], d['walkers'], round(d['stepwise_us_per_step'],1), round(d['persistent_us_per_step'],1), round(d['auto_us_per_step'],1))"
