#!/bin/bash
python tools/ab_bench.py c2 2 latency=1 finish=1 2>&1 | tail -3 | cut -c1-3500
python tools/ab_bench.py c2 2 census=1 2>&1 | tail -2 | head -1 | cut -c1-3000
