#!/bin/bash
# usage: tools/benchexp.sh <tag>...   headline bench (no CPU baseline) on experimental builds under gridcodegenerator_amd/_build_x/<tag>; "-" = shipped build
for tag in "$@"; do
  if [ "$tag" = "-" ]; then bd=""; else bd="--build-dir gridcodegenerator_amd/_build_x/$tag"; fi
  echo -n "$tag: "; python bench.py --steps 300 --warmup 30 --no-cpu-baseline $bd $BENCH_EXTRA | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msolves/s  %.2f us/launch'%(d['value']/1e6, d['roofline']['launch_us']))" || exit 1
done
