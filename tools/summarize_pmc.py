"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files (two SEPARATE passes of the same bench.py
command, tools/profile_round.sh) to per-kernel-class HBM traffic per launch (bytes).

gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly HALF of the bytes of a wide (16 B per lane)
coalesced streaming read — `global_load_dwordx4` and `buffer_load ... lds` alike — so every kernel class whose reads are
16 B per lane is doubled; WRITE_SIZE is taken as reported.  Both counters are in KiB.  A sanity check follows the
correction: the corrected fetch must be at least the class's COMPULSORY input bytes (it cannot read less than its
operands); classes that fail are flagged `below_compulsory` instead of being trusted.

    python tools/summarize_pmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [h w]
"""
import collections
import csv
import datetime
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def src_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "any-stereo_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def conv_args(name):
    m = re.search(r"conv_(?:split|igemm)_kernel<([^>]*)>", name)
    return [a.strip() for a in m.group(1).split(",")] if m else None


# kernel-name substring -> (class, reads are 16 B per lane?)
RULES = [
    ("lookup_convc1_direct_kernel", "lookup_convc1", True),  # register-direct form (the one the loop launches): float4 per tap
    ("lookup_convc1_kernel", "lookup_convc1", True),   # geometry windows: float4 per tap
    ("lookup_fwd_quad_kernel", "lookup", True),
    ("lookup_fwd_coop_kernel", "lookup", True),
    ("corr_build_lds_kernel", "corr_build", True),     # batched buffer loads of the 128-wide f2 slab, 16 B per lane
    ("corr_build_f16x3_kernel", "corr_build", True),
    ("corr_build_kernel", "corr_build", True),
    ("geo_pyramid_kernel", "geo_pyramid", True),
    ("gwc_kernel", "gwc_volume", True),
    ("liif_tail_kernel", "liif_tail", True),           # 16-B groups of the channels-last rows
    ("liif_lowres_cl_kernel", "liif_lowres", False),   # dword loads along pixels
    ("sf_partial_kernel", "liif_affinity", False),
]


def compulsory_inputs(h, w, C=96, G=8, D=48, L=2, r=4):
    """Operand bytes a launch cannot avoid reading at the bench's cfg-2 shapes (B = 1)."""
    P = h * w
    return {
        "corr_build": 4 * 2 * C * P,
        "gwc_volume": 4 * 2 * C * P,
        "geo_pyramid": 4 * G * D * P,
        "lookup": 4 * P * (L * (G + 1) * (2 * r + 2) + 1),
        "lookup_convc1": 4 * P * (L * (G + 1) * (2 * r + 2) + 1),
        # blocked split-fp16 sources: 2 x 2 B per element, (h, mf, up) = 384 channels; weights come from L2
        "gru04_zr_conv": 4 * 384 * P,
        "gru04_q_conv": 4 * 384 * P,
        "disp_head_conv1": 4 * 128 * P,
    }


def classify(name, grid, grids_by_epi):
    for sub, cls, wide in RULES:
        if sub in name:
            return cls, wide
    a = conv_args(name)
    if a and "conv_split_kernel" in name:
        epi = a[3]
        # the GRU z|r conv has the largest grid of its epilogue; DispHead.conv1 (RELU_TAPS = 4, or LINEAR) shares that grid;
        # larger LINEAR grids belong to the full-resolution context-net convs
        biggest = max(grids_by_epi.get(epi, [0]))
        if grid == biggest and epi in ("1", "2", "4"):
            # the loop's convs stage both operand images by 16-B LDS-DMA from blocked split-fp16 tensors (all-DMA path)
            return {"1": "gru04_zr_conv", "2": "gru04_q_conv", "4": "disp_head_conv1"}[epi], True
    return None, False


def blocks(r):
    """Launch size in BLOCKS: the 8-wave and the 4-wave (lean) form of a convolution have the same block grid but 512 / 256
    threads per block, and one pass runs both (the first iteration's sources are not blocked yet)."""
    return int(r["Grid_Size"]) // max(1, int(r.get("Workgroup_Size", 0) or 1))


def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []


def main():
    pf, pw, out = sys.argv[1:4]
    h, w = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (136, 240)
    res = collections.defaultdict(lambda: {"fetch": [], "write": [], "wide": False})
    for key, rows in (("fetch", load(pf)), ("write", load(pw))):
        grids = collections.defaultdict(list)
        for r in rows:
            a = conv_args(r["Kernel_Name"])
            if a and "conv_split_kernel" in r["Kernel_Name"]:
                grids[a[3]].append(blocks(r))
        for r in rows:
            cls, wide = classify(r["Kernel_Name"], blocks(r), grids)
            if cls is None:
                continue
            v = float(r["Counter_Value"]) * 1024.0
            res[cls][key].append(v * (2.0 if (key == "fetch" and wide) else 1.0))
            res[cls]["wide"] = wide
    need = compulsory_inputs(h, w)
    summary = {"_meta": {"src_hash": src_hash(), "collected": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
                         "command": "bench.py --no-cpu-baseline --no-batched --no-extras --no-graph --steps 1 --warmup 1 under "
                                    "rocprofv3 --pmc FETCH_SIZE and (separately) --pmc WRITE_SIZE",
                         "correction": "FETCH_SIZE x2 for classes with 16 B/lane reads (MI355X_MICROARCH.md §HBM); "
                                       "KiB -> bytes x1024", "quarter_res_map": [h, w]}}
    for cls, d in res.items():
        fe = sum(d["fetch"]) / len(d["fetch"]) if d["fetch"] else None
        wr = sum(d["write"]) / len(d["write"]) if d["write"] else None
        ent = {"fetch_bytes": fe, "write_bytes": wr, "hbm_bytes": (fe or 0) + (wr or 0), "fetch_doubled": d["wide"],
               "launches_sampled": max(len(d["fetch"]), len(d["write"]))}
        if cls in need and fe is not None:
            ent["compulsory_input_bytes"] = need[cls]
            ent["below_compulsory"] = bool(fe < 0.97 * need[cls])
            if ent["below_compulsory"]:  # the counter cannot be trusted for this access pattern: do not publish a total
                ent["hbm_bytes"] = None
        summary[cls] = ent
    raw = collections.defaultdict(lambda: {"fetch": [], "write": []})
    for key, rows in (("fetch", load(pf)), ("write", load(pw))):
        for r in rows:
            raw[(r["Kernel_Name"][:110], int(r["Grid_Size"]))][key].append(float(r["Counter_Value"]) * 1024.0)
    top = sorted(raw.items(), key=lambda kv: -(sum(kv[1]["fetch"]) + sum(kv[1]["write"])))[:30]
    summary["_raw_per_kernel"] = [
        {"kernel": k[0], "grid": k[1], "launches": max(len(v["fetch"]), len(v["write"])),
         "fetch_bytes_avg_uncorrected": sum(v["fetch"]) / len(v["fetch"]) if v["fetch"] else None,
         "write_bytes_avg": sum(v["write"]) / len(v["write"]) if v["write"] else None} for k, v in top]
    json.dump(summary, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if not k.startswith("_raw")}, indent=1))


if __name__ == "__main__":
    main()
