#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "eval_scenes or evaluation_loops" 2>&1 | tail -15
