#!/bin/bash
# Diagnostic build of the library with in-kernel cycle timers in conv_ring_kernel (-DPWS_RING_TIMERS), then tools/ring_timers.py.
# Run from the repo root after `python -m pwstablenet_amd.build` (reuses the other objects).
set -e
mkdir -p tools/_bin
O=pwstablenet_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPWS_RING_TIMERS=${TIMERS_VARIANT:-1} -x hip -c pwstablenet_amd/csrc/conv_ring.hip -o /tmp/conv_ring_timers.o
OBJS=$(ls $O/*.o | grep -v "conv_ring.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ${TIMERS_OUT:-tools/_bin/ring_timers.so} $OBJS /tmp/conv_ring_timers.o
echo built ${TIMERS_OUT:-tools/_bin/ring_timers.so}
