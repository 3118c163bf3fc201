# Parity suites under forced configurations (experiment knobs that change geometry, not results beyond tolerance).
# Stops at the first run that was killed by its timeout; ordinary test failures are listed and the next run starts.
#   bash tools/hygiene.sh > gpurun_out/hygiene.txt
run() {
  echo "== $*"
  env "$@" timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pyramid_patch.py tests/test_gpu_errors.py tests/test_gpu_robust.py -q 2>&1 | tail -4
  rc=${PIPESTATUS[0]}
  if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "killed by timeout: stopping"; exit 1; fi
}
run ICTR_REF8_CPW_BY_LEVEL=0
run ICTR_RESIDENT_SLOTS=2
run ICTR_RESIDENT_NP=32
run ICTR_RESIDENT_NP=16
run ICTR_CPW=64
run ICTR_RESIDENT_PRIO=0
