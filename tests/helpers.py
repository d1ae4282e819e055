"""Shared test helpers (oracle-side map construction from golden dumps)."""
import numpy as np

from oracle import rbpf_oracle as orc


def oracle_map_from_dump(g, prefix, cs):
    """Rebuild an OracleHybridMap from a `dump_map` record in a golden file."""
    hm = orc.OracleHybridMap(cs)
    hm.tiles = []
    centres, offs = g[prefix + "centres"], g[prefix + "offs"]
    for t in range(len(centres)):
        cx, cy = centres[t]
        cx = int(cx) if float(cx).is_integer() else float(cx)
        cy = int(cy) if float(cy).is_integer() else float(cy)
        tile = orc.OracleTile(cx, cy, 40, cs)
        s, e = offs[t], offs[t + 1]
        tile.map[g[prefix + "xs"][s:e], g[prefix + "ys"][s:e]] = g[prefix + "vals"][s:e]
        hm.tiles.append(tile)
    return hm


def dump_oracle_map(hm):
    out = {}
    for t in hm.tiles:
        xs, ys = np.nonzero(t.map)
        out[(float(t.cx), float(t.cy))] = (xs, ys, t.map[xs, ys])
    return out


def golden_dump_as_dict(g, prefix):
    out = {}
    centres, offs = g[prefix + "centres"], g[prefix + "offs"]
    for t in range(len(centres)):
        s, e = offs[t], offs[t + 1]
        out[(float(centres[t][0]), float(centres[t][1]))] = (
            g[prefix + "xs"][s:e], g[prefix + "ys"][s:e], g[prefix + "vals"][s:e])
    return out
