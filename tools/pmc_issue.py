#!/usr/bin/env python3
"""Instruction issue of the decode-stage kernels against a CALIBRATED roof.

usage: pmc_issue.py "<glob of the passes' results .db>" <out.json> <config> [commit] [date] [issue_roof.json]

From rocprofv3 --pmc SQ passes (tools/pmc_sq.sh: separate runs, --kernel-trace only), for every tbz_ kernel: the
wave-instructions issued per launch by type (SQ_INSTS_VALU / _SALU / _LDS / _VMEM_RD / _VMEM_WR / _BRANCH: summed over the
counter's instances of a dispatch, averaged over the dispatches), the waves per SIMD (SQ_WAVE_CYCLES over the busy
quad-cycles), the parked / stalled / issuing shares of a wave's life, and the launch's duration (the kernel trace of the
same passes).

The roof is MEASURED (tools/issue_roof.hip -> profiles/issue_roof.json; round 3 assumed "one wave-instruction per SIMD per
quad-cycle whatever its type" and came out at 1.03 - 1.15 of it: a ceiling the measurement exceeds is no ceiling).  What
one gfx950 SIMD retires per NANOSECOND — the shader clock gives way under load, so cycles are the wrong unit — was taken for
streams of the decoders' own instructions (v_alignbit / v_bfe / v_cndmask; s_and_b64 / s_cselect_b64; ds_read_u16) at
VALU : SALU = 1 : 0, 2 : 1 and 1 : 1, with 1 .. 8 waves per SIMD on all 1 024 SIMDs.  A kernel's roof is that table read at
the kernel's own (SALU + LDS) : VALU ratio and waves per SIMD (bilinear); issue_frac = (VALU + SALU + LDS per launch) / (1024 SIMDs x
duration x roof), which cannot exceed 1 by construction of the roof.  valu_frac is the vector pipe alone against the
1 : 0 stream at the same occupancy."""
import glob
import json
import os
import sqlite3
import sys

N_SIMD = 256 * 4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(pattern):
    acc, dur = {}, {}
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
        for k, a in cur.execute(f"select s.kernel_name, avg(d.end - d.start) from {dsp} d join {sym} s on d.kernel_id=s.id "
                                f"group by s.kernel_name"):
            dur.setdefault(k.replace(".kd", ""), []).append(a)
    out = {}
    for k, cs in acc.items():
        if not k.startswith("tbz_"):
            continue
        out[k] = {c: {"sum": sum(v[0] for v in vs) / len(vs), "max": sum(v[1] for v in vs) / len(vs)} for c, vs in cs.items()}
        out[k]["_dur_ns"] = sum(dur.get(k, [0.0])) / max(1, len(dur.get(k, [0.0])))
    return out


class Roof:
    """profiles/issue_roof.json as a function: wave-instructions per SIMD per ns at (SALU per VALU, waves per SIMD)"""

    def __init__(self, path):
        rows = json.load(open(path))["rows"]
        self.tab = {}  # salu-per-valu -> [(W, total rate, valu rate)]
        for r in rows:
            m = r["mix"]
            ratio = 0.0 if m.startswith("V ") else 0.5 if m.startswith("VS ") else 1.0 if m.startswith("VS11") else None
            if ratio is None:
                continue
            self.tab.setdefault(ratio, []).append((r["waves_per_simd"], r["wave_inst_per_simd_ns"], r["first_type_per_simd_ns"]))
        for v in self.tab.values():
            v.sort()

    @staticmethod
    def _at(rows, w, col):
        if w <= rows[0][0]:
            return rows[0][col] * (w / rows[0][0])  # fewer waves than one per SIMD: proportionally fewer instructions
        for a, b in zip(rows, rows[1:]):
            if w <= b[0]:
                f = (w - a[0]) / (b[0] - a[0])
                return a[col] + f * (b[col] - a[col])
        return rows[-1][col]

    def rate(self, salu_per_valu, w, col=1):
        ks = sorted(self.tab)
        x = min(max(salu_per_valu, ks[0]), ks[-1])
        for a, b in zip(ks, ks[1:]):
            if x <= b:
                f = (x - a) / (b - a)
                return self._at(self.tab[a], w, col) * (1 - f) + self._at(self.tab[b], w, col) * f
        return self._at(self.tab[ks[-1]], w, col)


def main(pattern, out, config, commit=None, date=None, roof_path=None):
    raw = collect(pattern)
    roof = Roof(roof_path or os.path.join(ROOT, "profiles", "issue_roof.json"))
    ker = {}
    for k, c in raw.items():
        if "SQ_BUSY_CYCLES" not in c or "SQ_INSTS_VALU" not in c:
            continue
        g = lambda n: c.get(n, {"sum": 0.0})["sum"]
        busy = c["SQ_BUSY_CYCLES"]["max"]
        wc = g("SQ_WAVE_CYCLES") or 1.0
        dur = c["_dur_ns"]
        valu, salu, lds = g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS")
        w = wc / (busy / 4.0) / N_SIMD if busy else 0.0
        # scalar AND LDS instructions ride along with the vector ones (issue_roof: VS, VS11, VSL rows): the table is read at
        # (SALU + LDS) per VALU (reading it at SALU per VALU alone put tbz_k0b_scan, a kernel AT the vector roof with an LDS
        # lookup per four code lengths, at 1.11)
        spv = (salu + lds) / valu if valu else 0.0
        peak = roof.rate(spv, w) if w else None          # wave-instructions per SIMD per ns, this mix, this occupancy
        vpeak = roof.rate(0.0, w, 2) if w else None      # the vector pipe alone
        ach = (valu + salu + lds) / (N_SIMD * dur) if dur else None
        ker[k] = {
            "duration_ns": dur, "busy_cycles": busy,
            "valu": valu, "salu": salu, "lds": lds,
            "vmem": g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR"), "branch": g("SQ_INSTS_BRANCH"),
            "salu_per_valu": spv, "waves_per_simd": w,
            "issue_achieved": ach, "issue_roof": peak,
            "issue_frac": ach / peak if ach and peak else None,
            "valu_achieved": valu / (N_SIMD * dur) if dur else None, "valu_roof": vpeak,
            "valu_frac": valu / (N_SIMD * dur) / vpeak if dur and vpeak else None,
            "parked": g("SQ_WAIT_ANY") / wc, "stalled": g("SQ_WAIT_INST_ANY") / wc, "issuing": g("SQ_ACTIVE_INST_ANY") / wc,
            "lds_bank_conflict_share": (g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")) if g("SQ_LDS_IDX_ACTIVE") else None,
            "lanes_active": (g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))) if g("SQ_ACTIVE_INST_VALU") else None,
        }
    json.dump({"note": __doc__.split("\n\n", 1)[1].replace("\n", " "), "config": str(config), "commit": commit, "date": date,
               "how": "rocprofv3 --kernel-trace --pmc <8 SQ counters>, three separate passes (tools/pmc_sq.sh); roof: tools/issue_roof.hip",
               "unit": "wave-instructions per SIMD per ns",
               "kernels": ker}, open(out, "w"), indent=1)
    for k in sorted(ker, key=lambda k: -ker[k]["duration_ns"])[:8]:
        v = ker[k]
        f = lambda x: "  n/a" if x is None else "%5.2f" % x
        print("%-28s %8.1f us  VALU %.3g SALU %.3g LDS %.3g  waves/SIMD %.2f  issue %s of %s = %s   valu %s of %s = %s   parked %.2f  "
              "bank-conflict share %s  lanes active %s" % (
                  k, v["duration_ns"] / 1e3, v["valu"], v["salu"], v["lds"], v["waves_per_simd"], f(v["issue_achieved"]),
                  f(v["issue_roof"]), f(v["issue_frac"]), f(v["valu_achieved"]), f(v["valu_roof"]), f(v["valu_frac"]), v["parked"],
                  f(v["lds_bank_conflict_share"]), f(v["lanes_active"])))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], *sys.argv[4:7])
