#!/bin/bash
# A/B of the register-weight kernel on the 16-channel ENet layers (batch 16): LDS kernel vs conv_regw at several grid caps
cd "$GRAFT_REPO_ROOT"
export SSMI_SHAPES=2,3,4,14
echo "== LDS kernel"; SSMI_NO_REGW=1 timeout -k 10 120 python scripts/bench_convs.py 16 2>/dev/null || exit 1
for cfg in "1 512" "1 1024" "1 2048" "2 512" "4 512" "4 768" "4 1024"; do set -- $cfg; echo "== conv_regw, MT $1, at most $2 workgroups"; SSMI_REGW_MT=$1 SSMI_REGW_WGS=$2 timeout -k 10 120 python scripts/bench_convs.py 16 2>/dev/null || exit 1; done
