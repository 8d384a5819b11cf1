#!/bin/bash
# build the HIP library; non-zero exit (and the compiler errors) on failure
cd "$(dirname "$0")/.." || exit 1
out=$(python -m crimac_classifiers_unet_amd.build 2>&1)
rc=$?
echo "$out" | grep -E "error|\.so$" | head -20
if [ $rc -ne 0 ] || echo "$out" | grep -q "error"; then echo "BUILD FAILED"; exit 1; fi
python -c "from crimac_classifiers_unet_amd import hip; hip.load_library()" || exit 1
