"""Turn rocprofv3 --pmc passes with SQ counters into profiles/<tag>_pmc_sq.json: per kernel, the mean counter value per
launch (first launch of every kernel dropped).

usage: python tools/pmc_sq_collect.py <out.json> <counter_collection.csv> [...]"""
import json
import sys

import pandas as pd


def main():
    out = {}
    for csv in sys.argv[2:]:
        d = pd.read_csv(csv)
        for (name, counter), g in d.groupby(["Kernel_Name", "Counter_Name"]):
            if not name.startswith("rbpf::"):
                continue
            vals = g.sort_values("Dispatch_Id").Counter_Value.to_numpy()[1:]
            if len(vals):
                out.setdefault(name.split("(")[0], {})[counter] = float(vals.mean())
    json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
