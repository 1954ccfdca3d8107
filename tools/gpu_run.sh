#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "dynamic_quant or uint8" 2>&1 | tail -25
