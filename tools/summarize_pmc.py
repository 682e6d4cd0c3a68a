"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files to per-kernel-class HBM traffic
per launch (bytes), applying the gfx950 correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE reports exactly half
of the bytes of wide (16 B/lane) coalesced reads, so kernels whose reads are 16-B wide are doubled; WRITE_SIZE
is taken as reported.  Counter unit: KiB-like units of 1024 B? -> rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KB."""
import collections
import csv
import glob
import json
import re
import sys


def conv_epi(name):
    m = re.search(r"conv_(?:split|igemm)_kernel<([^>]*)>", name)
    if not m:
        return None
    args = [a.strip() for a in m.group(1).split(",")]
    return args[3] if len(args) >= 4 else args[-1]  # conv_split_kernel<KS, TW, BN, EPI, NSUB>, conv_igemm_kernel<KS, TW, EPI>

# kernel-name substring, optional epilogue template arg, -> (scope name, reads are 16 B/lane wide?)
RULES = [
    ("lookup_fwd_quad_kernel", None, "lookup", True),
    ("lookup_fwd_coop_kernel", None, "lookup", True),
    ("corr_build_lds_kernel", None, "corr_build", False),
    ("corr_build_f16x3_kernel", None, "corr_build", False),
    ("corr_build_kernel", None, "corr_build", False),
    ("geo_pyramid_kernel", None, "geo_pyramid", False),
    ("gwc_kernel", None, "gwc_volume", False),
]


def classify(name, grid, grids_by_epi):
    for sub, _, scope, wide in RULES:
        if sub in name:
            return scope, wide
    if "conv_split_kernel" in name or "conv_igemm_kernel" in name:
        epi = conv_epi(name)
        # the GRU z|r conv has the largest grid of its epilogue; DispHead.conv1 (LINEAR, 128 -> 256 at the same resolution)
        # has exactly that grid too — the larger LINEAR grids belong to the full-resolution context-net convs
        biggest = max(grids_by_epi.get("1" if epi == "0" else epi, [0]))
        if grid == biggest:
            # HBM-side fetches of the convs are the dword halo-patch loads (the 16-B weight loads hit L2): no doubling
            return {"1": "gru04_zr_conv", "2": "gru04_q_conv", "0": "disp_head_conv1"}.get(epi), False
    return None, False


def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")
    rows = list(csv.DictReader(open(f[0]))) if f else []
    return rows


def main():
    pf, pw, out = sys.argv[1:4]
    res = collections.defaultdict(lambda: {"fetch": [], "write": []})
    for key, rows in (("fetch", load(pf)), ("write", load(pw))):
        grids = collections.defaultdict(list)
        for r in rows:
            n = r["Kernel_Name"]
            if "conv_split_kernel" in n or "conv_igemm_kernel" in n:
                grids[conv_epi(n)].append(int(r["Grid_Size"]))
        for r in rows:
            scope, wide = classify(r["Kernel_Name"], int(r["Grid_Size"]), grids)
            if scope is None:
                continue
            v = float(r["Counter_Value"]) * 1024.0
            if key == "fetch" and wide:
                v *= 2.0
            res[scope][key].append(v)
    summary = {}
    for scope, d in res.items():
        fe = sum(d["fetch"]) / len(d["fetch"]) if d["fetch"] else None
        wr = sum(d["write"]) / len(d["write"]) if d["write"] else None
        summary[scope] = {"fetch_bytes": fe, "write_bytes": wr,
                          "hbm_bytes": (fe or 0) + (wr or 0), "launches_sampled": max(len(d["fetch"]), len(d["write"]))}
    # raw per-kernel averages (counter units as reported, x1024 B), for transparency
    raw = collections.defaultdict(lambda: {"fetch": [], "write": []})
    for key, rows in (("fetch", load(pf)), ("write", load(pw))):
        for r in rows:
            raw[(r["Kernel_Name"][:110], int(r["Grid_Size"]))][key].append(float(r["Counter_Value"]) * 1024.0)
    top = sorted(raw.items(), key=lambda kv: -(sum(kv[1]["fetch"]) + sum(kv[1]["write"])))[:30]
    summary["_raw_per_kernel"] = [
        {"kernel": k[0], "grid": k[1], "launches": max(len(v["fetch"]), len(v["write"])),
         "fetch_bytes_avg": sum(v["fetch"]) / len(v["fetch"]) if v["fetch"] else None,
         "write_bytes_avg": sum(v["write"]) / len(v["write"]) if v["write"] else None} for k, v in top]
    json.dump(summary, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
