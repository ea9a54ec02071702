#!/bin/bash
# usage: scripts/opt_sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...: one short bench per option set (context options via their env names)
for e in "$@"; do
  env $e python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-other-configs --no-other-legs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms']; print('$e', 'step', round(d['ms_per_step'],2), 'unet', round(s['unet'],2), 'geom', round(s['geometry_wall'],2), 'cconv', round(s['continuous_conv'],2), 'frac', round(d['roofline']['frac'],4))"
done
