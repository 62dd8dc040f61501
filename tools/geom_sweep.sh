#!/bin/bash
# Pipelined (23 streams) and serial throughput for conv launch geometries: SPS_GEOM_L<level>="<ntw>,<S>".
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-stages --no-h2d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1'.ljust(40), d['value'], d['roofline']['gpu_ms_per_step'])"; }
run base
for g in "2,4" "4,4" "4,2" "2,2" "4,1"; do SPS_GEOM_L3=$g SPS_GEOM_L4=$g run "L3=L4=$g"; done
for g in "2,4" "2,2" "2,1" "1,2"; do SPS_GEOM_L2=$g run "L2=$g"; done
for g in "1,2" "1,4"; do SPS_GEOM_L1=$g run "L1=$g"; done
for g in "1,2"; do SPS_GEOM_L0=$g run "L0=$g"; done
