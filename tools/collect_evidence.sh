#!/bin/bash
# Evidence set of one build, collected on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into profiles/<tag>/):
#   bench lines (default, driver protocol, serial, configs 3 and 4), rocprofv3 kernel stats (serial and pipelined),
#   HBM traffic from separate FETCH_SIZE / WRITE_SIZE passes, SQ / TA / TCP counters per conv layer, training-step and
#   online-filter timings.
# The HBM traffic passes run first, so that the bench lines of the set carry the traffic measured on this very build.
# usage: gpurun -- bash tools/collect_evidence.sh <tag> [pmc|bench|all]
#   pmc   = the counter / trace passes whose summaries the bench lines quote (traffic.json, kernel_durations.json,
#           vector_pipe_budget.json: written to gpurun_out/<tag>/ AND profiles/; copy them into profiles/ of the working tree,
#           then run `bench`), bench = everything else; all (default) = both in one call (needs ~20 min of box time)
tag=${1:-round2}; phase=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag; mkdir -p $o
: >> $o/bench.err
if [ "$phase" != bench ]; then
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $o/pmc_$ctr -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stages --no-h2d --streams 1 --pipelined-geometry > /dev/null 2>> $o/bench.err
done
# the tracked profiles/traffic.json (what `roofline.traffic` of every later bench line reports) is replaced only by a
# complete measurement of this build: both passes present and the derivation successful
if python3 tools/traffic_pmc.py $o/pmc_FETCH_SIZE $o/pmc_WRITE_SIZE $o/traffic.json > $o/traffic_summary.txt 2>> $o/bench.err && [ -s $o/traffic.json ]; then
  cp $o/traffic.json profiles/traffic.json   # (on the GPU box: the bench lines below then carry this build's `roofline.traffic`)
else
  echo "traffic_pmc.py failed: profiles/traffic.json left as it was" | tee -a $o/bench.err
fi
# rocprofv3 kernel durations of this build first: the bench lines below then carry `dominant_kernel_rocprof_us`
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_serial -- python3 bench.py --streams 1 --pipelined-geometry --steps 80 --warmup 10 --no-cpu-baseline --no-h2d > $o/bench_under_rocprof_serial.json 2>> $o/bench.err
if python3 tools/kernel_durations.py $o/prof_serial $o/kernel_durations.json >> $o/traffic_summary.txt 2>> $o/bench.err && [ -s $o/kernel_durations.json ]; then
  cp $o/kernel_durations.json profiles/kernel_durations.json
fi
cp $(ls $o/prof_serial/*/*kernel_stats.csv | head -1) $o/kernel_stats.csv
python3 tools/kernel_table.py $o/prof_serial > $o/kernel_table.txt
cp $(ls $o/prof_serial/*/*kernel_trace.csv | head -1) $o/kernel_trace_serial.csv
# the binding budget: MFMA + VALU time per SIMD and scan (six --pmc passes) -> vector_pipe_budget.{txt,json}
bash tools/pmc_stalls.sh $tag > /dev/null 2>> $o/bench.err
rm -rf $o/prof_serial $o/pmc_FETCH_SIZE $o/pmc_WRITE_SIZE
cat $o/traffic_summary.txt; tail -1 $o/vector_pipe_budget.txt
fi
[ "$phase" = pmc ] && exit 0
python3 bench.py > $o/bench.json 2>> $o/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_driver_protocol.json 2>> $o/bench.err
python3 bench.py --streams 1 --steps 200 --warmup 20 --no-cpu-baseline > $o/bench_serial_1stream.json 2>> $o/bench.err
python3 bench.py --config 3 --steps 150 --warmup 20 > $o/bench_config3.json 2>> $o/bench.err
python3 bench.py --config 4 --steps 150 --warmup 20 > $o/bench_config4.json 2>> $o/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_pipelined -- python3 bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-stages --no-h2d > $o/bench_under_rocprof_pipelined.json 2>> $o/bench.err
cp $(ls $o/prof_pipelined/*/*kernel_stats.csv | head -1) $o/kernel_stats_pipelined.csv
if [ ! -f $o/kernel_trace_serial.csv ]; then   # (phase `bench` on a box of its own: the serial trace overlap.py compares with)
  rocprofv3 --kernel-trace --output-format csv -d $o/prof_serial2 -- python3 bench.py --streams 1 --pipelined-geometry --steps 40 --warmup 10 --no-cpu-baseline --no-h2d --no-stages > /dev/null 2>> $o/bench.err
  cp $(ls $o/prof_serial2/*/*kernel_trace.csv | head -1) $o/kernel_trace_serial.csv; rm -rf $o/prof_serial2
fi
python3 tools/overlap.py $(ls $o/prof_pipelined/*/*kernel_trace.csv | head -1) $o/kernel_trace_serial.csv > $o/overlap.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $o/pmc_a -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stages --no-h2d --streams 1 --pipelined-geometry > /dev/null 2>> $o/bench.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $o/pmc_b -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stages --no-h2d --streams 1 --pipelined-geometry > /dev/null 2>> $o/bench.err
python3 tools/pmc_table.py $o/pmc_a $o/pmc_b > $o/pmc_conv_layers.txt 2>> $o/bench.err
python3 tools/pmc_derive.py $o/pmc_conv_layers.txt > $o/pmc_conv_layers_derived.txt 2>> $o/bench.err
python3 tools/train_timing.py --steps 50 2>> $o/bench.err | tail -1 > $o/train_timing.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_train -- python3 tools/train_timing.py --steps 20 > /dev/null 2>> $o/bench.err
cp $(ls $o/prof_train/*/*kernel_stats.csv | head -1) $o/kernel_stats_train.csv
python3 tools/train_roofline.py $o/kernel_stats_train.csv $o/bench.json >> $o/train_timing.txt 2>> $o/bench.err
python3 tools/filter_timing.py 2>> $o/bench.err | tail -4 > $o/filter_timing.txt
rm -rf $o/prof_train $o/prof_pipelined $o/pmc_a $o/pmc_b $o/kernel_trace_serial.csv
ls -la $o; tail -3 $o/bench.err; cat $o/train_timing.txt $o/filter_timing.txt; cat $o/traffic_summary.txt; cat $o/pmc_conv_layers_derived.txt | head -30
python3 -c "
import json
for f in ('bench','bench_driver_protocol','bench_serial_1stream','bench_config3','bench_config4'):
    d=json.loads(open('$o/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], (d.get('cpu_baseline') or {}).get('value'), (d.get('h2d_inclusive') or {}).get('value'))
"
