#!/usr/bin/env python3
"""Instruction issue per launch from rocprofv3 --pmc SQ passes (tools/pmc_sq.sh: separate runs, --kernel-trace only).

usage: pmc_issue.py "<glob of the passes' results .db>" <out.json> <config> [commit] [date]

For every tbz_ kernel: wave-instructions issued per launch by type (SQ_INSTS_VALU / _SALU / _LDS / _VMEM_RD / _VMEM_WR /
_BRANCH: summed over the counter's instances — one per shader engine — of a dispatch, averaged over the dispatches) and
the kernel's busy time in cycles (SQ_BUSY_CYCLES, the largest instance).  The ISSUE ROOF: a SIMD hands one wavefront one
instruction per quad-cycle — a wave64 vector instruction occupies the 16-lane SIMD for four cycles — so 1024 SIMDs x
busy cycles / 4 slots per launch.  issue_frac = (VALU + SALU + LDS) / slots says how far the kernel is from issuing an
instruction on every SIMD in every slot (instructions of different types CAN issue side by side from different waves:
the VALU share alone, valu_frac, is the pipe that saturates first here); parked / stalled / issuing are the shares of
the resident waves' time (SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES)."""
import glob
import json
import sqlite3
import sys

N_SIMD = 256 * 4


def collect(pattern):
    acc = {}
    paths = sorted(set(glob.glob(pattern)) | set(glob.glob(pattern.replace("/*.db", "/*/*.db"))))
    for p in paths:
        db = sqlite3.connect(p)
        cur = db.cursor()
        t = lambda like: [r[0] for r in cur.execute("select name from sqlite_master where name like '%s%%'" % like)][0]
        sym, dsp, info, ev = t("rocpd_info_kernel_symbol"), t("rocpd_kernel_dispatch"), t("rocpd_info_pmc"), t("rocpd_pmc_event")
        q = (f"select s.kernel_name, i.name, d.id, sum(e.value), max(e.value) from {ev} e join {info} i on e.pmc_id=i.id "
             f"join {dsp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id group by s.kernel_name, i.name, d.id")
        for k, c, _, tot, mx in cur.execute(q):
            acc.setdefault(k.replace(".kd", ""), {}).setdefault(c, []).append((tot, mx))
    out = {}
    for k, cs in acc.items():
        if not k.startswith("tbz_"):
            continue
        out[k] = {c: {"sum": sum(v[0] for v in vs) / len(vs), "max": sum(v[1] for v in vs) / len(vs)} for c, vs in cs.items()}
    return out


def main(pattern, out, config, commit=None, date=None):
    raw = collect(pattern)
    ker = {}
    for k, c in raw.items():
        if "SQ_BUSY_CYCLES" not in c or "SQ_INSTS_VALU" not in c:
            continue
        g = lambda n: c.get(n, {"sum": 0.0})["sum"]
        busy = c["SQ_BUSY_CYCLES"]["max"]
        slots = N_SIMD * busy / 4.0
        wc = g("SQ_WAVE_CYCLES") or 1.0
        ker[k] = {
            "busy_cycles": busy, "issue_slots": slots,
            "valu": g("SQ_INSTS_VALU"), "salu": g("SQ_INSTS_SALU"), "lds": g("SQ_INSTS_LDS"),
            "vmem": g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR"), "branch": g("SQ_INSTS_BRANCH"),
            "issue_frac": (g("SQ_INSTS_VALU") + g("SQ_INSTS_SALU") + g("SQ_INSTS_LDS")) / slots if slots else None,
            "valu_frac": g("SQ_INSTS_VALU") / slots if slots else None,
            "waves_per_simd": wc / (busy / 4.0) / N_SIMD if busy else None,
            "parked": g("SQ_WAIT_ANY") / wc, "stalled": g("SQ_WAIT_INST_ANY") / wc, "issuing": g("SQ_ACTIVE_INST_ANY") / wc,
        }
    json.dump({"note": __doc__.split("\n\n", 1)[1].replace("\n", " "), "config": str(config), "commit": commit, "date": date,
               "how": "rocprofv3 --kernel-trace --pmc <8 SQ counters>, three separate passes (tools/pmc_sq.sh)",
               "kernels": ker}, open(out, "w"), indent=1)
    for k in sorted(ker, key=lambda k: -ker[k]["busy_cycles"])[:8]:
        v = ker[k]
        print("%-28s busy %.3g cyc  VALU %.3g SALU %.3g LDS %.3g  issue %.2f  valu %.2f  waves/SIMD %.2f  parked %.2f" % (
            k, v["busy_cycles"], v["valu"], v["salu"], v["lds"], v["issue_frac"], v["valu_frac"], v["waves_per_simd"], v["parked"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], *sys.argv[4:6])
