for lib in ${LIBS:-ab_base.so pwstablenet_amd/libpwstable_hip.so ab_base.so pwstablenet_amd/libpwstable_hip.so}; do
  echo "== $lib"
  PWS_LIB_PATH=$PWD/$lib EXPS=0 bash tools/ring_bench.sh
  PWS_LIB_PATH=$PWD/$lib python tools/configs2_step.py 2>&1 | tail -2
done
