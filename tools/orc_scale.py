"""Thread scaling of the CPU oracle on the host it runs on (bench.py's cpu_baseline picks its thread count from this).
    python tools/orc_scale.py [fast]      fast = the oracle's fast hash (sparse partial rounds, SIMD lanes)"""
import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as g, oracle_lib as O
pkg = g.load_package()
b = pkg.CircuitBuilder(); t = pkg.AesGcmTarget.build(b, 4, 10, 1024, False); data = b.build()
key, nonce, pt = bytes([42]*16), bytes([111]*12), bytes([42]*1024)
ct, tag = pkg.native.gcm_encrypt(key, nonce, pt)
pw = pkg.PartialWitness(); t.set_targets(pw, key, nonce, pt, ct, tag)
oc = O.OracleCircuit(data.blob)
fast = len(sys.argv) > 1 and sys.argv[1] == "fast"
O.lib().orc_set_fast_hash(int(fast))
print("fast hash", fast, "SIMD lanes", O.lib().orc_set_simd_lanes(8), "host threads", O.lib().orc_num_threads(), flush=True)
for nt in (128, 64, 32, 16, 8, 1):
    if nt > O.lib().orc_num_threads() and nt != 1:
        continue
    O.lib().orc_set_num_threads(nt)
    oc.prove(pw.map)
    t0 = time.time(); oc.prove(pw.map); dt = time.time() - t0
    print(nt, round(dt, 3), {k: round(v, 3) for k, v in O.OracleCircuit.last_stage_seconds().items()}, flush=True)
