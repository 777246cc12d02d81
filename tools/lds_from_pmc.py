#!/usr/bin/env python3
"""LDS / VALU utilisation of the cascade kernels from a tools/pmc_summary.py text ("report rocprof HBM GB/s, LDS
bank-conflict rate and VALU utilisation next to the roofline fraction") -> JSON for bench.py's roofline.lds:
  {"k_cascade": {"idx_active_frac": LDS-array cycles / CU cycles, "bank_conflict_frac": conflict cycles / LDS-array cycles,
                 "valu_frac": vector instructions x 4 cycles / SIMD cycles, "kernels": [...]}}
Counters of all k_cascade launches of a batch (scan-step-1 and scan-step-2 kernels) are summed; CU cycles = GRBM_GUI_ACTIVE / 8
XCDs x 256 CUs, SIMD cycles = 4 x that (MI355X_MICROARCH.md)."""
import json, re, sys

def parse(path):
    d, k = {}, None
    for ln in open(path):
        if ln.startswith("hc::"):
            k = ln.strip(); d.setdefault(k, {})
        else:
            m = re.match(r"\s+(\S+)\s+mean\s+([\d.]+)", ln)
            if m and k:
                d[k][m.group(1)] = float(m.group(2))
    return d

def main():
    src, dst = sys.argv[1], sys.argv[2]
    d = parse(src)
    ks = [k for k in d if "k_cascade" in k and "GRBM_GUI_ACTIVE" in d[k] and "SQ_LDS_IDX_ACTIVE" in d[k]]
    tot = lambda c: sum(d[k].get(c, 0.0) for k in ks)
    cu = tot("GRBM_GUI_ACTIVE") / 8.0 * 256.0
    out = {"k_cascade": {
        "idx_active_frac": round(tot("SQ_LDS_IDX_ACTIVE") / cu, 4),
        "bank_conflict_frac": round(tot("SQ_LDS_BANK_CONFLICT") / tot("SQ_LDS_IDX_ACTIVE"), 4),
        "valu_frac": round(tot("SQ_INSTS_VALU") * 4.0 / (4.0 * cu), 4),
        "per_launch": {"SQ_LDS_IDX_ACTIVE": tot("SQ_LDS_IDX_ACTIVE"), "SQ_LDS_BANK_CONFLICT": tot("SQ_LDS_BANK_CONFLICT"),
                       "SQ_INSTS_VALU": tot("SQ_INSTS_VALU"), "SQ_INSTS_LDS": tot("SQ_INSTS_LDS"), "cu_cycles": cu},
        "kernels": [k.split("<")[0] + "<" + k.split("<", 1)[1][:60] for k in ks],
        "note": "valu_frac prices a vector instruction at 4 SIMD cycles"}}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["k_cascade"]))

if __name__ == "__main__":
    main()
