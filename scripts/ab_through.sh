#!/bin/bash
# A/B of the expansion's data gradient inside the merge backward's apply pass (default) against its own launch (SSMI_NO_THROUGH=1):
# whole step, step without any weight gradient (SSMI_ABLATE_WGRAD=1, results meaningless), step with the weight gradients on the main chain
cd "$GRAFT_REPO_ROOT"
run() { env "$@" timeout -k 10 300 python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$* $EXTRA', d['ms_per_step'])"; }
for rep in 1 2; do
  EXTRA="" run A=1 || exit 1
  EXTRA="" run SSMI_NO_THROUGH=1 || exit 1
  EXTRA="" run SSMI_THROUGH_C=64 || exit 1
  EXTRA="" run SSMI_THROUGH_C=128 || exit 1
done
