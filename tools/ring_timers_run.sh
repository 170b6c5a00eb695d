#!/bin/bash
# usage: bash tools/ring_timers_run.sh [lib ...]   (default tools/_bin/ring_timers.so; build with tools/ring_timers.sh)
for lib in ${@:-tools/_bin/ring_timers.so}; do
  echo "== $lib"
  export PWS_LIB_PATH=$PWD/$lib
  python tools/ring_timers.py CONV_K3S1 64 256 256 64 64
  python tools/ring_timers.py CONV_K3S1 64 128 128 128 128
  python tools/ring_timers.py CONVT_K4S2 64 128 128 256 64
  python tools/ring_timers.py CONVT_K4S2 64 128 128 256 64 dgrad
done
