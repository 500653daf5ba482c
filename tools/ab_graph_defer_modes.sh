#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--no-breakdown --no-cpu-baseline --no-extras --no-roofline --steps 30 --warmup 8"
for m in "graph:loader:0:aux,dense,pcr" "graph:loader:0:dense" "graph:loader:0:aux,pcr" "graph:loader:0:pcr" "graph:loader:0:dense,pcr" "graph:loader:0:0" "eager:loader:aux,dense,sparse" "graph:loader:sparse:aux,dense,pcr"; do
  python bench.py --mode "$m" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['value'])"
done
