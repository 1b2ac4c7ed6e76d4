#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only).

usage: pmc_traffic.py <fetch.db> <write.db> <raw_out.json> <traffic_out.json> <C> <U> [commit] [date]

Counters are in KB per dispatch, one row per counter instance: summed over the instances of a dispatch, then
averaged over the dispatches of a kernel.  Corrections as MI355X_MICROARCH.md §HBM prescribes for gfx950:
FETCH_SIZE reports half of a wide coalesced streaming read (16 B per lane) -> doubled for K2 (its token loads are that
pattern).  K1's loads (one 16-byte window read per lane, every lane in a different line) and K1's stores (4 bytes per
lane per token, every lane in a different line) are patterns the guide calls uncalibrated: they are taken as reported,
and say more as ratios between versions than as absolutes.  WRITE_SIZE is exact for K2's 16-byte streaming stores."""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    cur = db.cursor()
    t = lambda like: [r[0] for r in cur.execute("select name from sqlite_master where name like '%s%%'" % like)][0]
    sym, dsp, info, ev = t("rocpd_info_kernel_symbol"), t("rocpd_kernel_dispatch"), t("rocpd_info_pmc"), t("rocpd_pmc_event")
    q = (f"select s.kernel_name, d.id, sum(e.value) from {ev} e join {info} i on e.pmc_id=i.id "
         f"join {dsp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id where i.name='{counter}' "
         f"group by s.kernel_name, d.id")
    acc = {}
    for k, _, v in cur.execute(q):
        acc.setdefault(k.replace(".kd", ""), []).append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main(fetch_db, write_db, raw_out, out, C, U, commit=None, date=None):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    raw = {k: {"FETCH_SIZE": f.get(k, 0.0), "WRITE_SIZE": w.get(k, 0.0)} for k in sorted(set(f) | set(w)) if k.startswith("tbz_")}
    json.dump(raw, open(raw_out, "w"), indent=1)
    KB = 1024.0
    k1 = [k for k in raw if k.startswith("tbz_k1")]
    k2 = [k for k in raw if k.startswith("tbz_k2")]
    k1f = sum(raw[k]["FETCH_SIZE"] for k in k1) * KB
    k1w = sum(raw[k]["WRITE_SIZE"] for k in k1) * KB
    k2f = sum(raw[k]["FETCH_SIZE"] for k in k2) * KB * 2
    k2w = sum(raw[k]["WRITE_SIZE"] for k in k2) * KB
    tot = k1f + k1w + k2f + k2w
    alg = C + U
    json.dump({
        "note": __doc__.split("\n\n", 1)[1].replace("\n", " "),
        "workload": "bench.py config 2, 1 GiB",
        "config": "2",
        "commit": commit,
        "date": date,
        "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/refresh_profiles.sh)",
        "C": C, "U": U,
        "k1_fetch_bytes": k1f, "k1_write_bytes": k1w, "k2_fetch_bytes_corrected": k2f, "k2_write_bytes": k2w,
        "decode_stage_hbm_bytes_per_launch": tot,
        "algorithmic_bytes_decode_stage": alg,
        "ratio_traffic_over_algorithmic": tot / alg,
    }, open(out, "w"), indent=1)
    print(json.dumps(raw, indent=1))
    print("decode stage traffic %.3f GB vs algorithmic %.3f GB" % (tot / 1e9, alg / 1e9))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), *sys.argv[7:9])
