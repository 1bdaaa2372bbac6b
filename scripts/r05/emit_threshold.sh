#!/bin/bash
# round 5: where forming the rows inside k_cgi_rows (every workgroup fences, the last one compacts) stops paying: the fence of
# a workgroup writes the XCD's L2 back, ~1 us each and one after another per XCD
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_cgi_tail
mkdir -p $OUT
for nq in 1 4 8 16 32; do
  for e in 16384 256; do
    FA_ROWS_EMIT_MAX=$e python3 scripts/time_pass.py 20 $nq 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nq', d['queries'], 'emit_max', d['env'].get('FA_ROWS_EMIT_MAX'), 'cgi_ms %.4f total_ms %.4f' % (d['cgi_ms'], d['total_ms']), d['rows_sha'])"
  done
done | tee $OUT/emit_threshold.txt
