#!/bin/bash
# Poll power / clocks / temperature while a sustained bench runs:  bash scripts/power_poll.sh <out-prefix> <bench args...>
OUT=$1; shift
( for i in $(seq 1 40); do echo "--- t=$i"; rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge|memory)|Performance" ; sleep 0.5; done ) > $OUT.smi.txt 2>&1 &
POLL=$!
python bench.py --steps 2500 --no-cpu-baseline --no-extras "$@" > $OUT.json 2>&1
kill $POLL 2>/dev/null; wait $POLL 2>/dev/null
grep '^{' $OUT.json | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])'
