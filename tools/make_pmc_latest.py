#!/usr/bin/env python3
"""profiles/pmc_latest.json from a tools/collect_pmc.sh run: HBM bytes (FETCH_SIZE x2 per the gfx950 correction of
MI355X_MICROARCH.md HBM section, + WRITE_SIZE) and VALU wave-instructions per launch of the dominant kernel, for the
default bench.py segment workload, stamped with the git commit the profile was taken at.
Usage: make_pmc_latest.py <per-kernel summary csv from tools/pmc_summary.py> <tag> <git commit> [existing json]"""
import csv
import json
import sys


def main():
    path, tag, commit = sys.argv[1:4]
    rows = {r["kernel"]: r for r in csv.DictReader(l for l in open(path) if not l.startswith("#"))}
    k = next(n for n in rows if "poseidon_hash_rows_kernel<false>" in n)
    r = rows[k]
    f, w = float(r["FETCH_SIZE"]), float(r["WRITE_SIZE"])
    out = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else {}
    gui = float(r["GRBM_GUI_ACTIVE"])
    out["segment"] = {
        "source": "profiles/%s_pmc_segment_per_kernel.csv (rocprofv3 --pmc passes, tools/collect_pmc.sh %s --commit-steps 0: "
                  "bench.py default segment workload, %s dispatches; means per dispatch)" % (tag, tag, r["dispatches"]),
        "git_commit": commit, "kernel": k, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
        "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read -> doubled; WRITE_SIZE as is; x1024 B",
        "leaf_hash_hbm_bytes_per_launch": (2 * f + w) * 1024,
        "leaf_hash_valu_wave_insts_per_launch": float(r["SQ_INSTS_VALU"]),
        "GRBM_GUI_ACTIVE_sum_over_8_XCDs": gui,
        "valu_issue_utilisation": float(r["SQ_INSTS_VALU"]) * 4 / (gui / 8 * 1024),
    }
    for name in rows:
        if "ntt_pass_kernel" in name or "ntt_strided_swap_kernel" in name or "ntt_contig_wave_kernel" in name:
            rr = rows[name]
            out.setdefault("ntt", {})[name.strip()] = {
                "dispatches": rr["dispatches"], "valu_wave_insts": float(rr["SQ_INSTS_VALU"]),
                "valu_issue_utilisation": float(rr["SQ_INSTS_VALU"]) * 4 / (float(rr["GRBM_GUI_ACTIVE"]) / 8 * 1024),
                "lds_bank_conflict_cycles": float(rr["SQ_LDS_BANK_CONFLICT"]), "lds_active_cycles": float(rr["SQ_LDS_IDX_ACTIVE"])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
