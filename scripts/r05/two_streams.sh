#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_two_streams
mkdir -p $OUT
python3 scripts/r05/two_streams.py 8 50 1 2 3 > $OUT/two_streams.json 2> $OUT/two_streams.err; cat $OUT/two_streams.json; tail -3 $OUT/two_streams.err
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
