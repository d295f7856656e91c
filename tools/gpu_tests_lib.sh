#!/bin/bash
# run the GPU parity tests against a given HIP build: tools/gpu_tests_lib.sh name
QPALM_GFX950_LIB=$PWD/qpalm_amd/lib/libqpalm_gfx950_$1.so timeout 300 python -m pytest tests/test_parity.py -x -q -m gpu 2>&1 | tail -1
