"""Reduce two rocprofv3 PMC passes (FETCH_SIZE in one run, WRITE_SIZE in another; the two
do not fit one pass) of `bench.py --clips C` to HBM bytes per launch of the chain's kernels.

    python profiles/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> C out.json

Units and corrections: both counters are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read
requests as 64 bytes, so it is doubled; the doubling is checked in the same run on k_ch_apply
(reads exactly clips * frames * 2 channels * 4 B, coalesced): its corrected fetch must match
that byte count within a few percent.  WRITE_SIZE is exact for the dword-per-lane stores used here
(k_ch_apply writes exactly the same byte count)."""
import collections, csv, json, sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    # launches differ in size (phase 1 vs phase 2c share kernels): keep the largest class
    return {k: max(v) for k, v in agg.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    clips = int(sys.argv[3])
    import os
    out = {"clips": clips, "commit": os.environ.get("ADN_COMMIT", "?"), "unit": "bytes per launch (largest launch of each kernel)",
           "correction": "FETCH_SIZE x 2 (gfx950), WRITE_SIZE x 1, KiB -> bytes", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f = 2.0 * fetch.get(k, 0.0) * 1024.0
        w = write.get(k, 0.0) * 1024.0
        if f + w < 64e6:
            continue
        out["kernels"][k] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w,
                             "hbm_bytes_per_clip": (f + w) / clips}
    cal = out["kernels"].get("void adn::k_ch_apply<float, 2, 3>")
    if cal:
        expect = clips * 60 * 48000 * 2 * 4
        out["calibration"] = {"kernel": "k_ch_apply<float, 2, 3>", "expected_read_bytes": expect,
                              "fetch_ratio": cal["fetch_bytes"] / expect, "write_ratio": cal["write_bytes"] / expect}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
