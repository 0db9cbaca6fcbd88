#!/bin/bash
timeout -k 10 900 python -m pytest tests -q -m gpu 2>&1 | tail -2 && timeout -k 10 300 python -c "
import __graft_entry__ as g
g.smoke()
" 2>&1 | tail -1 && timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1),'M/s', d['ms_per_step'], d['roofline']['frac'])"
