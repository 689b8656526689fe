#!/bin/bash
# c1r tuning sweep (round 4): block size / blocks per CU / per-tile barrier, timed with tools/kbench.py --only c1
mkdir -p gpurun_out/c1r_knobs
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python tools/kbench.py --only c1 > gpurun_out/c1r_knobs/$name.txt 2>&1
  echo "== $name ($*) rc=$?"
  grep "c1r" gpurun_out/c1r_knobs/$name.txt | grep -v "^fwd\|^dgrad"
}
run base WFAE_C1R_WAVES=8
run sync WFAE_C1R_WAVES=8 WFAE_C1R_SYNC=1
run w4b3 WFAE_C1R_WAVES=4 WFAE_C1R_BPC=3
run w4b3sync WFAE_C1R_WAVES=4 WFAE_C1R_BPC=3 WFAE_C1R_SYNC=1
run w4b2 WFAE_C1R_WAVES=4 WFAE_C1R_BPC=2
