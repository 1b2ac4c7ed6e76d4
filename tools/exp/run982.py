import sys, pickle, importlib, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests import parity_cases as P
T = importlib.import_module("3bz_amd")
eng = T.Engine(0, lib_path=os.environ.get("EMU_LIB"))
fmt, blob, steps, sizes = pickle.load(open(os.path.join(os.path.dirname(__file__), 'cf982.pkl'),'rb'))
for rep in range(3):
    out = bytearray(sizes[0]); r = eng.inflate(blob[:steps[0]], 1, out); print("fresh", rep, r.status, r.out_len, r.out_total, r.in_consumed, r.segments, flush=True)
# warm the pools with a bigger call, then again
big = bytearray(200000); r = eng.inflate(blob, 1, big); print("big", r.status, r.out_len)
for rep in range(2):
    out = bytearray(sizes[0]); r = eng.inflate(blob[:steps[0]], 1, out); print("after big", rep, r.status, r.out_len, r.out_total, r.in_consumed, r.segments, flush=True)
for cap in (1361, 1362, 1363, 4085, 4086, 4087):
    out = bytearray(cap); r = eng.inflate(blob[:steps[0]], 1, out); print("cap", cap, r.status, r.out_len, r.out_total)
