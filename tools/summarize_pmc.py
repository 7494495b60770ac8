"""Summarise rocprofv3 PMC passes of bench.py into profiles/<tag>_conv_traffic.json.

Usage: python tools/summarize_pmc.py <gpurun_out/dir> <tag> <steps+warmup of the pmc runs>

Per the MI355X guide (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are collected in separate
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports exactly half of the bytes of wide
coalesced 16-B-per-lane streams (all conv loads here are global_load[_lds]_dwordx4) -> doubled;
WRITE_SIZE is exact (checked: it equals the algorithmic output bytes of every conv launch).
"""
import collections
import csv
import glob
import json
import sys


def per_dispatch(dirname, counter):
    f = glob.glob(f"{dirname}/pmc_{counter}*/runc/*counter_collection.csv")[0]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        out.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
    return out


def main():
    d, tag, n_fw = sys.argv[1], sys.argv[2], int(sys.argv[3])
    res = {"source": d, "forwards_in_pmc_run": n_fw, "kernels": {}}
    for counter, scale, key in (("FETCH_SIZE", 2.0 * 1024, "fetch_bytes"), ("WRITE_SIZE", 1024.0, "write_bytes")):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for _, (name, val) in per_dispatch(d, counter).items():
            short = name.split("(")[0].replace("void ", "")
            agg[short][0] += 1
            agg[short][1] += val * scale
        for k, (calls, tot) in agg.items():
            e = res["kernels"].setdefault(k, {})
            e["calls"] = calls
            e[key + "_per_launch"] = tot / calls
            e[key + "_per_forward"] = tot / n_fw
    conv = [v for k, v in res["kernels"].items() if "conv3x3_mfma" in k or "conv3x3_wino" in k or "conv3x3_w16" in k]
    res["conv3x3_mfma"] = {
        "launches_per_forward": sum(v["calls"] for v in conv) / n_fw,
        "hbm_bytes_per_forward": sum(v.get("fetch_bytes_per_forward", 0) + v.get("write_bytes_per_forward", 0) for v in conv),
        "fetch_bytes_per_forward": sum(v.get("fetch_bytes_per_forward", 0) for v in conv),
        "write_bytes_per_forward": sum(v.get("write_bytes_per_forward", 0) for v in conv),
    }
    res["conv3x3_mfma"]["hbm_bytes_per_launch"] = res["conv3x3_mfma"]["hbm_bytes_per_forward"] / res["conv3x3_mfma"]["launches_per_forward"]
    json.dump(res, open(f"profiles/{tag}_conv_traffic.json", "w"), indent=1)
    print(json.dumps(res["conv3x3_mfma"], indent=1))
    for k, v in res["kernels"].items():
        print(f"{k[:60]:60s} calls {v['calls']:4d}  fetch/launch {v.get('fetch_bytes_per_launch',0)/1e6:10.1f} MB  write/launch {v.get('write_bytes_per_launch',0)/1e6:10.1f} MB")


if __name__ == "__main__":
    main()
