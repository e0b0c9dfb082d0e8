#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output dirs for one kernel: python tools/pmc_summary.py <kernel substring> <dir> [<dir> ...]
Counters are reported per launch of the LONGEST dispatch of that kernel in each run (the main launch of a frame; the cost
pre-pass launch of the same kernel is listed separately under "other_dispatches")."""
import collections, csv, glob, json, sys

def main():
    kern = sys.argv[1]
    agg = collections.OrderedDict()
    durs, others = [], []
    for d in sys.argv[2:]:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            disp = collections.OrderedDict()  # dispatch id -> {"dur": ns, counters}
            for r in csv.DictReader(open(f)):
                if kern not in r["Kernel_Name"]:
                    continue
                e = disp.setdefault(r["Dispatch_Id"], {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
                e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if not disp:
                continue
            main_id = max(disp, key=lambda k: disp[k]["dur"])
            for k, e in disp.items():
                if k == main_id:
                    durs.append(e["dur"] / 1e6)
                    for n, v in e["c"].items():
                        agg.setdefault(n, []).append(v)
                else:
                    others.append({"ms": e["dur"] / 1e6, **e["c"]})
    agg = collections.OrderedDict((k, sum(v) / len(v)) for k, v in agg.items())
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
        der["fetch_bytes"] = g("FETCH_SIZE") * 1024
    if g("WRITE_SIZE"):
        der["write_bytes"] = g("WRITE_SIZE") * 1024
    print(json.dumps({"kernel": kern, "main_launch_ms": durs, "counters_per_main_launch": agg, "derived": der, "other_dispatches": others}, indent=1))

main()
