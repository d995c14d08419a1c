"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, see MI355X_MICROARCH.md 'HBM') into
per-kernel average HBM bytes per launch.  gfx950 corrections from that guide: FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE reports half of the bytes of wide coalesced streaming reads (16 B/lane, global loads and LDS-DMA alike), so it is
doubled; WRITE_SIZE is exact for 16-byte streaming stores (our epilogues issue 8-byte stores: uncalibrated, reported as is).
    python tools/summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN_hbm_traffic.json"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.match(r"(?:void )?lds::(\w+?)(?:_kernel)?(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        nf, vf = f.get(k, [0, 0.0])
        nw, vw = w.get(k, [0, 0.0])
        out[k] = {"launches": max(nf, nw), "fetch_bytes_per_launch": 2.0 * 1024.0 * vf / max(nf, 1),
                  "write_bytes_per_launch": 1024.0 * vw / max(nw, 1)}
        out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
    json.dump({"note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as reported; KiB units", "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
