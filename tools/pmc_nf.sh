#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_nf.sh <tag> [commit] — the counters of the 1 GiB no-flush stream
# (bench.py --config nf --size-mib 1024): the three SQ passes (tools/pmc_sq.sh -> gpurun_out/pmc_issue_<tag>nf.json,
# pmc_<tag>nf.txt) and FETCH_SIZE / WRITE_SIZE in passes of their own -> gpurun_out/<tag>_pmc_traffic_nf.json
TAG=$1; COMMIT=${2:-unknown}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
bash $R/tools/pmc_sq.sh ${TAG}nf 1024 nf $COMMIT > $O/${TAG}nf_pmc_sq.log 2>&1 || { echo "SQ passes failed"; tail -3 $O/${TAG}nf_pmc_sq.log; }
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/${TAG}nf_pmc_$C -o p -- python3 $R/bench.py --config nf --size-mib 1024 --corpus-cache /tmp/tbz_corpus_$(id -u) --steps 2 --warmup 0 --no-cpu-baseline --no-h2h > $O/${TAG}nf_pmc_$C.log 2>&1 || { echo "pmc $C failed"; tail -3 $O/${TAG}nf_pmc_$C.log; exit 1; }
done
F=$(ls $O/${TAG}nf_pmc_FETCH_SIZE/*results.db $O/${TAG}nf_pmc_FETCH_SIZE/*/*results.db 2>/dev/null | head -1)
W=$(ls $O/${TAG}nf_pmc_WRITE_SIZE/*results.db $O/${TAG}nf_pmc_WRITE_SIZE/*/*results.db 2>/dev/null | head -1)
python3 - "$F" "$W" "$O/${TAG}_pmc_traffic_nf.json" "$COMMIT" <<'PY'
import json, sqlite3, sys, datetime
def per_kernel(path, counter):
    db = sqlite3.connect(path); cur = db.cursor()
    t = lambda like: [r[0] for r in cur.execute("select name from sqlite_master where name like '%s%%'" % like)][0]
    sym, dsp, info, ev = t("rocpd_info_kernel_symbol"), t("rocpd_kernel_dispatch"), t("rocpd_info_pmc"), t("rocpd_pmc_event")
    q = (f"select s.kernel_name, d.id, sum(e.value) from {ev} e join {info} i on e.pmc_id=i.id join {dsp} d on e.event_id=d.event_id "
         f"join {sym} s on d.kernel_id=s.id where i.name='{counter}' group by s.kernel_name, d.id")
    acc = {}
    for k, _, v in cur.execute(q):
        acc.setdefault(k.replace(".kd", ""), []).append(v)
    return acc
f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
calls = 2.0
out = {"workload": "bench.py --config nf --size-mib 1024 (one ordinary zlib stream, 1 GiB of text)", "commit": sys.argv[4],
       "date": datetime.datetime.utcnow().strftime("%Y-%m-%d"),
       "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 2 calls each (tools/pmc_nf.sh); KB per launch as reported "
              "(gfx950: FETCH_SIZE reports half of a wide streaming read: MI355X_MICROARCH.md)", "kernels": {}}
for k in sorted(set(f) | set(w)):
    if not k.startswith("tbz_"):
        continue
    fv, wv = f.get(k, []), w.get(k, [])
    out["kernels"][k] = {"launches_per_call": len(fv or wv) / calls,
                         "fetch_KB_per_launch_as_reported": float("%.4g" % (sum(fv) / len(fv))) if fv else None,
                         "write_KB_per_launch": float("%.4g" % (sum(wv) / len(wv))) if wv else None}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -(kv[1]["fetch_KB_per_launch_as_reported"] or 0) * kv[1]["launches_per_call"])[:8]:
    print(k, v)
PY
rm -rf $O/${TAG}nf_pmc_FETCH_SIZE $O/${TAG}nf_pmc_WRITE_SIZE $O/pmc_${TAG}nf_[0-9]
