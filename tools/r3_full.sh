#!/bin/bash
# full GPU suite + a short bench.py run
set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 || exit 1
python bench.py --steps 2 --warmup 1 --no-cpu-baseline
