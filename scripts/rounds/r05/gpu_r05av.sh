#!/bin/bash
# round 5, GPU pass av: the at-scale tests (ranged host call of a 1024 window: same translation unit as the folded 2048 shape) on the final build
timeout 100 python -m pytest tests/test_parity_at_scale_gpu.py -m gpu -q --no-header 2>&1 | tail -3
