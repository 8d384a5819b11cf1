#!/bin/bash
set -e
CRIMAC_LIB=$PWD/gpurun_exp_diagconv.so timeout -k 10 300 python tools/diag_wch_phases.py h3p 2>&1 | grep -v amdgpu
