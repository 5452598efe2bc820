#!/bin/bash
# Socket power / clocks (rocm-smi) while a kernel loops: scripts/probes/power_sample.sh "<python command that runs ~6 s>" <tag>
CMD=$1; TAG=$2
( eval "$CMD" > /dev/null 2>&1 ) &
PID=$!
sleep 2.5
for i in 1 2 3 4; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor junction\)" | tr '\n' ';' | sed "s/^/[$TAG] /"; echo
  sleep 0.7
done
wait $PID
