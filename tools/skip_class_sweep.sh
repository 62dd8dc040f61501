#!/bin/bash
# What does each kernel class cost the PIPELINED rate?  Diagnostic build (tools/ab_build.sh diag -DSPS_DIAG
# -DSPS_FUSE_UP=0), bench.py with one class of convolution launches skipped at a time (wrong results, of course).
# usage (GPU box): bash tools/skip_class_sweep.sh
mkdir -p gpurun_out
for m in 0 1 2 4 8 16 3 31; do
  SPS_LIB=tools/ab/lib_diag.so SPS_DIAG_SKIP_CLASS=$m python3 bench.py --no-cpu-baseline --no-stages > gpurun_out/skip_$m.json 2>> gpurun_out/skip.err
done
python3 - <<'PY'
import json
names = {0: "nothing skipped", 1: "3^4 layers of levels 2-4 (10 launches)", 2: "3^4 layers of levels 0-1 (6)", 4: "strided convs (4)",
         8: "transposed convs (4)", 16: "conv0", 3: "all 3^4 layers", 31: "every convolution"}
base = None
for m in (0, 1, 2, 4, 8, 16, 3, 31):
    d = json.loads(open(f"gpurun_out/skip_{m}.json").read().strip().splitlines()[-1])
    r = d["resident_value"]
    if base is None: base = r
    print(f"skip {names[m]:42s} resident {r:8.1f} scans/s  = {1e6/r:6.1f} us/scan   saves {1e6/base - 1e6/r:6.1f} us/scan")
PY
