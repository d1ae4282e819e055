"""Turn rocprofv3 --pmc counter_collection CSVs into profiles/<tag>_pmc_traffic.json.

usage: python tools/pmc_collect.py <out.json> <particles>:<fetch.csv>:<write.csv> [...]
FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section); the first dispatch of every kernel is dropped."""
import json
import sys

import pandas as pd


def per_kernel(csv, counter, skip_first=1):
    d = pd.read_csv(csv)
    d = d[d.Counter_Name == counter]
    out = {}
    for name, g in d.groupby("Kernel_Name"):
        if not name.startswith("rbpf::"):
            continue
        vals = g.sort_values("Dispatch_Id").Counter_Value.to_numpy()[skip_first:]
        if len(vals):
            out[name.split("(")[0]] = (float(vals.mean()), int(len(vals)))
    return out


def main():
    out = {}
    for spec in sys.argv[2:]:
        P, fcsv, wcsv = spec.split(":")
        f = per_kernel(fcsv, "FETCH_SIZE")
        w = per_kernel(wcsv, "WRITE_SIZE")
        out[P] = {k: {"fetch_raw_bytes": f[k][0] * 1024.0, "write_bytes": w.get(k, (0.0, 0))[0] * 1024.0, "launches": f[k][1]}
                  for k in f}
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
