#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "epilogue or conv_fwd_dgrad or batchnorm" 2>&1 | tail -15
