#!/bin/bash
# Checks round 1's tree (38e3921) out into _r01_tree/ (git-ignored; travels to the GPU box with its built library) and
# builds its HIP library there, for scripts/gpu_regression_vs_round1.sh.
set -e
cd "$(dirname "$0")/.."
rm -rf _r01_tree && mkdir _r01_tree
git archive 38e3921 | tar -x -C _r01_tree
cp scripts/gpu_config_regression.py _r01_tree/scripts/
(cd _r01_tree && python -m sunray_amd.build > /dev/null && ls -la sunray_amd/libsunray_hip.so)
rm -rf _r01_tree/profiles _r01_tree/tests/golden _r01_tree/gpurun_out
