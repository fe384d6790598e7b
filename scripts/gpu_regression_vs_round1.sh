#!/bin/bash
# Run ON THE GPU BOX: the configuration table of scripts/gpu_config_regression.py for this tree and for a checkout of round 1's
# tree (38e3921, prepared HERE before the gpurun call by scripts/prepare_round1_tree.sh into _r01_tree/, built in place).
cd "$(dirname "$0")/.."
out=gpurun_out/config_regression.txt
: > $out
if [ -d _r01_tree/sunray_amd ]; then
  (cd _r01_tree && SUNRAY_REF_ASSETS=$PWD/../tests/golden/ref_assets timeout -k 10 400 python scripts/gpu_config_regression.py round1 2>&1 | grep -E "^round1|Error|error" ) >> $out
fi
timeout -k 10 400 python scripts/gpu_config_regression.py HEAD 2>&1 | grep -E "^HEAD|Error|error" >> $out
SUNRAY_PRIMARY_REUSE=0 timeout -k 10 400 python scripts/gpu_config_regression.py HEAD-nohandoff 2>&1 | grep -E "^HEAD|Error|error" >> $out
cat $out
