#!/bin/bash
O=gpurun_out/r2d; mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== bwd_streams=0" > $O/log.txt
FIND_TUNING=bwd_streams=0 timeout 120 python tools/graph_bisect.py mlp_free_pts >> $O/log.txt 2>&1
echo "== reduce_stream=0" >> $O/log.txt
FIND_TUNING=reduce_stream=0 timeout 120 python tools/graph_bisect.py mlp_free_pts >> $O/log.txt 2>&1
echo "== default with API log" >> $O/log.txt
AMD_LOG_LEVEL=3 timeout 120 python tools/graph_bisect.py mlp_free_pts > $O/api.log 2>&1
grep -n "hipStreamBeginCapture" $O/api.log | tail -1 >> $O/log.txt
L=$(grep -n "hipStreamBeginCapture" $O/api.log | tail -1 | cut -d: -f1)
tail -n +$L $O/api.log | grep -v "hipGetDevice\|hipGetLastError\|hipPeekAtLastError\|__hipPushCallConfiguration\|__hipPopCallConfiguration" | cut -c1-220 > $O/api_capture.log
wc -l $O/api_capture.log >> $O/log.txt
rm -f $O/api.log
echo "== b1 debug" >> $O/log.txt
HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 300 python tools/debug_b1.py >> $O/log.txt 2>&1
tail -40 $O/log.txt
