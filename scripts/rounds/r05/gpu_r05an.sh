#!/bin/bash
# round 5, GPU pass an: the new smoke-entry test (and its neighbours) on the final build
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -k "smoke or kernel_choice or golden" 2>&1 | tail -4
