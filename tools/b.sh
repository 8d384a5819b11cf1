#!/bin/bash
# build the HIP library; non-zero exit (and the compiler's diagnostics) on failure
cd "$(dirname "$0")/.." || exit 1
python -m crimac_classifiers_unet_amd.build > /tmp/crimac_build.log 2>&1
rc=$?
grep -E "error:|warning: .*(uninit|overflow)|\.so$" /tmp/crimac_build.log | head -30
if [ $rc -ne 0 ]; then echo "BUILD FAILED (rc=$rc)"; tail -30 /tmp/crimac_build.log; exit 1; fi
python -c "from crimac_classifiers_unet_amd import hip; hip.load_library()" || exit 1
