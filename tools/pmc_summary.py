#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output dirs for one kernel: python tools_pmc_summary.py <kernel substring> <dir> [<dir> ...]"""
import collections, csv, glob, json, sys

def main():
    kern = sys.argv[1]
    agg = collections.OrderedDict()
    durs = []
    for d in sys.argv[2:]:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                if kern not in r["Kernel_Name"]:
                    continue
                agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    der = {}
    g = agg.get
    if g("TCC_HIT_sum") is not None:
        der["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        der["valu_lane_utilization"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)
    if g("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if g(k):
                der[k + "/WAVE_CYCLES"] = g(k) / g("SQ_WAVE_CYCLES")
    if g("FETCH_SIZE"):
        der["fetch_bytes_raw"] = g("FETCH_SIZE") * 1024
    print(json.dumps({"kernel": kern, "durations_ms": durs, "counters": agg, "derived": der}, indent=1))

main()
