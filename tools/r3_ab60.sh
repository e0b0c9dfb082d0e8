#!/bin/bash
python tools/ab_bench.py c4 1 latency=1 finish=1 2>&1 | tail -3 | head -2 | cut -c1-1800
python tools/ab_bench.py c4 1 timeline=1 2>&1 | tail -2 | head -1 | cut -c1-1500
