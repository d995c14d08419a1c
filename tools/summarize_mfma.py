"""Summarise a rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY ...
pass into per-kernel matrix-pipe utilisation.
Normalisation (checked against the algorithmic MFMA count: v_mfma_f32_32x32x2_f32 holds a SIMD's pipe for 64 cycles, so
busy = FLOP / 64 per launch up to tile padding): SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, SQ_BUSY_CYCLES over the
32 shader engines, hence  mfma_busy_frac = MFMA_BUSY / (BUSY * 32)  = share of SIMD-cycles, while the kernel is resident, in which the
matrix pipe is busy.  SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY are given as fractions of SQ_WAVE_CYCLES (wave-parked at a
wait or barrier / issue-stalled / issuing).
    python tools/summarize_mfma.py <counter_collection.csv> > profiles/rNN_mfma_util.json"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.match(r"(?:void )?lds::(\w+?)(?:_kernel)?(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            cnt[k] += 1
    out = {}
    for k, c in agg.items():
        b, w, m = c["SQ_BUSY_CYCLES"], c["SQ_WAVE_CYCLES"], c["SQ_VALU_MFMA_BUSY_CYCLES"]
        if not b or not w or not k.startswith(("conv_", "attention", "gn_", "lm_", "voc_")):
            continue
        out[k] = {"launches": cnt[k], "mfma_busy_frac": m / (b * 32.0), "mfma_busy_cycles_per_launch": m / cnt[k],
                  "wait_any_frac": c["SQ_WAIT_ANY"] / w, "wait_inst_frac": c["SQ_WAIT_INST_ANY"] / w, "active_inst_frac": c["SQ_ACTIVE_INST_ANY"] / w}
    json.dump({"note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES); wait/active fractions of SQ_WAVE_CYCLES", "kernels": out},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
