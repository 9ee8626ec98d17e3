#!/bin/bash
# Runs ON THE GPU BOX: the hardware queue of every kernel of the single-commitment loop (rocprofv3's Queue_Id), with the runtime's
# placement (LG_PICK_STREAMS=0) and with the library's chosen streams, after 0 / 1 / 2 extra streams made by the host application.
export TMPDIR=/tmp
for S in 0 1 2; do
  for P in 0 1; do
    D=/tmp/qm_${S}_$P; rm -rf $D
    LG_PICK_STREAMS=$P rocprofv3 --kernel-trace --output-format csv -d $D -- python3 tools/stream_order_probe.py --child $S > /tmp/qm.json 2>/dev/null
    echo "== extra streams $S, chosen streams $([ $P = 1 ] && echo on || echo off): $(grep '^{' /tmp/qm.json)"
    python3 tools/queue_map.py $(find $D -name '*kernel_trace.csv' | head -1) 1500
  done
done
