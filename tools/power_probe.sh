#!/bin/bash
# Samples clocks and socket power (rocm-smi, read-only) while a saturated launch loop runs: is the kernel held at a
# power / clock limit?   usage (GPU box, repo root): bash tools/power_probe.sh "<command>" <out-prefix>
CMD=$1; OUT=$2
mkdir -p "$(dirname "$OUT")"
( for i in $(seq 1 400); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.25; done ) > "$OUT.smi.log" &
SMI=$!
$CMD > "$OUT.cmd.log" 2>&1
RC=$?
kill $SMI 2>/dev/null
wait $SMI 2>/dev/null
exit $RC
