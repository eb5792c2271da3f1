#!/bin/bash
# The C5 step (16 feet x 50 002-vertex template, opt-in fp16 mode) with the side streams OFF (bwd_streams = fwd_streams = 0): every kernel runs alone,
# so the trace gives each kernel's own time (in the product's step they overlap and stretch each other).  usage: tools/prof_c5_serial.sh [out-name]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/${1:-r06}; mkdir -p $O
export FIND_TUNING=bwd_streams=0,fwd_streams=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5ser -- python3 $R/bench.py --c5 --fp16 --steps 10 --warmup 3 --no-cpu-baseline --headline-only > $O/c5ser_line.json 2> $O/c5ser.err
python3 $R/tools/step_stats.py $O/c5ser/*/*kernel_trace.csv head_out_fwd_h16_kernel 6 > $O/c5_fp16_serial_step_stats.csv
head -24 $O/c5_fp16_serial_step_stats.csv | cut -c1-150; tail -1 $O/c5_fp16_serial_step_stats.csv
rm -rf $O/c5ser/
