#!/usr/bin/env python3
"""Headline benchmark: proofs/sec on the AES-GCM 1 KiB circuit (BASELINE.json configs[2]).

A "step" is one pass of the hot path -- p2_prove_batch_device: witness generation, commitments, quotient, FRI --
over one batch of synthetic PartialWitness inputs that are already resident in HBM.  One process per GPU; with
N > 1 each rank proves its own shard of independent proofs (no data-path collective), weak scaling.
Prints ONE JSON line (rank 0).  Only the cpu_baseline leg touches oracle/.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_TAG = "r03"     # profiles/<tag>_traffic.json (PMC bytes) and profiles/<tag>_valu.json (PMC VALU instructions) of this round
NUM_SIMD, LANES_PER_CYCLE = 256 * 4, 32  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles


def splitmix64(seed):
    """SplitMix64 byte stream (proof i uses seed 0x5EED + i)."""
    x = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
        for k in range(8):
            yield (z >> (8 * k)) & 0xFF


def synth_inputs(pkg, target, index, L):
    g = splitmix64(0x5EED + index)
    key = bytes(next(g) for _ in range(16))
    nonce = bytes(next(g) for _ in range(12))
    pt = bytes(next(g) for _ in range(L))
    ct, tag = pkg.native.gcm_encrypt(key, nonce, pt)
    pw = pkg.PartialWitness()
    target.set_targets(pw, key, nonce, pt, ct, tag)
    return pw


def synth_elgamal_inputs(pkg, targets, index):
    """ElGamal workload (BASELINE.json configs[3]; ecgfp5/src/elgamal/circuit.rs:78-96): key, message point and nonce
    from seeds 0x5EED + 3 i + {0, 1, 2}."""
    pk_t, nonce_t, msg_t, ct_t = targets
    E = pkg.ecgfp5
    sk = pkg.ECGFP5SecretKey.rand(0x5EED + 3 * index)
    pk, msg, nonce = sk.public_key(), E.new_rand_from_subgroup(0x5EED + 3 * index + 1), E.random_scalar(0x5EED + 3 * index + 2)
    ct = E.elgamal_encrypt(pk, nonce, msg)
    pw = pkg.PartialWitness()
    pw.set_point_target(pk_t, pk)
    pw.set_point_target(msg_t, msg)
    pw.set_biguint320_target(nonce_t, nonce)
    pw.set_point_target(ct_t[0], ct[0])
    pw.set_point_target(ct_t[1], ct[1])
    return pw


def path_bytes_per_proof(info, live_wires):
    """bytes/proof = 64 n [4.25 c_w + 4.125 (c_z + c_q) + 2 c_p] + 1536 n with this build's column counts
    (c_w = wire columns that carry data, c_p = constants + 80 sigmas)."""
    n = 1 << info["degree_bits"]
    c_p = info["num_constants_cols"] + info["num_routed_wires"]
    return 64 * n * (4.25 * live_wires + 4.125 * (info["num_zs_cols"] + info["num_quotient_cols"]) + 2 * c_p) + 1536 * n


def kernel_bytes(name, info, active_wires=80):
    """Algorithmic HBM bytes of ONE launch of a kernel, per proof (DESIGN.md 'Kernels')."""
    n = 1 << info["degree_bits"]
    N = 8 * n
    if name == "hash_leaves":  # dominant launch = the wires tree: 80 live columns read once + digests written
        return 8 * N * active_wires + 32 * N
    if name == "lde":
        return 8 * n * active_wires + 8 * N * active_wires
    if name == "quotient":
        return 8 * N * (80 + info["num_constants_cols"] + 80 + info["num_zs_cols"] + 16) + 16 * N
    return None


def cpu_quota_cores():
    """CPU share of this container in cores (cgroup v2 cpu.max, v1 cfs quota), or None when unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def load_profile(kind, root=ROOT, tag=None):
    """profiles/<tag>_<kind>.json (written by tools/pmc_traffic.py / pmc_valu.py from rocprofv3 PMC passes) and where it came from.
    The counter files are evidence taken on ONE build: each records the identity of the kernel sources it was measured on
    (tools/src_id.py) and is used only while that matches the sources this run is built from -- otherwise the data is None and
    the source record says "stale", instead of silently pricing another binary."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from src_id import csrc_id
    here = csrc_id()["csrc_sha16"]
    path = os.path.join("profiles", (tag or PROFILE_TAG) + "_" + kind + ".json")
    try:
        j = json.load(open(os.path.join(root, path)))
    except Exception:  # noqa: BLE001
        return None, {"file": path, "status": "missing"}
    src = j.get("source", {})
    ok = src.get("csrc_sha16") == here
    return (j if ok else None), {"file": path, "measured_on_csrc_sha16": src.get("csrc_sha16"), "git_head": src.get("git_head"), "this_build_csrc_sha16": here,
                                 "status": "current" if ok else "stale: measured on other kernel sources"}


def spawn_ranks(n):
    """One process per GPU on this node: python -m torch.distributed.run --nproc-per-node n bench.py <same argv>."""
    import socket
    import subprocess
    with socket.socket() as so:  # a free rendezvous port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def device_identity(torch, index):
    """PCI bus id (or uuid) of the device this rank proves on: the driver can see N distinct GPUs in the JSON line."""
    p = torch.cuda.get_device_properties(index)
    dom, bus, dev = (getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    if bus is not None:
        return "%04x:%02x:%02x.0" % (dom or 0, bus, dev or 0)
    return str(getattr(p, "uuid", index))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="proofs per step per GPU (two chunks, one per proving stream)")
    ap.add_argument("--plaintext-bytes", type=int, default=1024)
    ap.add_argument("--workload", choices=["aes-gcm", "elgamal"], default="aes-gcm",
                    help="aes-gcm = BASELINE.json's metric workload (default); elgamal = configs[3]'s circuit")
    ap.add_argument("--pcie-steps", type=int, default=2, help="extra untimed-for-`value` steps through the host path (value_pcie_inclusive); 0 = skip")
    ap.add_argument("--config", type=int, choices=[3, 4, 5], default=None,
                    help="BASELINE.json configs[i-1] as one flag: 3 = AES-GCM 1 KiB (the default workload), 4 = ElGamal with the per-GPU share of "
                         "batch 1024 over 8 GPUs (--workload elgamal --batch 128), 5 = AES-GCM 64 KiB with the per-GPU share of batch 256 over "
                         "8 GPUs (--plaintext-bytes 65536 --batch 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=32, help="most OpenMP threads of the cpu_baseline leg (the port stops scaling at 16-32 on the GPU boxes' hosts; a cgroup CPU quota below this lowers it)")
    ap.add_argument("--cpu-sample", type=int, default=12, help="proofs timed on the host for cpu_baseline (median, after one warm-up)")
    args = ap.parse_args()

    if args.config == 4:
        args.workload, args.batch = "elgamal", 128
    elif args.config == 5:
        args.plaintext_bytes, args.batch = 65536, 32

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD torch.distributed.run (never an
    # exec -- and before anything in this process touches the GPU), relay rank 0's JSON line and leave with the child's code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    lib = pkg.lib()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the prover has no CPU fallback")
    # Rehearsal on a one-GPU box (the 8-GPU run is the driver's): P2AES_BENCH_REHEARSAL=1 puts every rank on device 0 and
    # uses gloo for the barrier / max-over-ranks, since RCCL refuses two ranks on one device.  Never set by the driver.
    rehearsal = os.environ.get("P2AES_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    ident = device_identity(torch, local_rank)
    devices = [ident]
    if dist:
        devices = [None] * world
        dist.all_gather_object(devices, ident)

    L, B = args.plaintext_bytes, args.batch
    if L > 4096 and args.batch == 256:
        B = 16  # deep circuits: ~7 GB of workspace per proof -> two chunks of 8, one per stream, so witness generation overlaps
    builder = pkg.CircuitBuilder()
    if args.workload == "elgamal":
        pk_t, nonce_t, msg_t = builder.add_virtual_point_target(), builder.add_virtual_biguint320_target(), builder.add_virtual_point_target()
        target = (pk_t, nonce_t, msg_t, builder.elgamal_encrypt(pk_t, nonce_t, msg_t))
    else:
        target = pkg.AesGcmTarget.build(builder, 4, 10, L, False)  # AesGcm128Target<L>, aes-gcm/src/lib.rs:19
    data = pkg.CircuitData(builder.build().blob, device=local_rank)
    info = data.info
    h = data.gpu()
    pb = data.proof_bytes

    # synthetic witnesses: rank r proves proofs [r*B, (r+1)*B); inputs are placed in HBM before the timed region
    if args.workload == "elgamal":
        pws = [synth_elgamal_inputs(pkg, target, rank * B + i) for i in range(B)]
        label = "ecGFp5 ElGamal encryption circuit (elgamal/circuit.rs:28), n=2^%d rows, standard_recursion_config" % info["degree_bits"]
        metric = "proofs/sec (ElGamal circuit)"
    else:
        pws = [synth_inputs(pkg, target, rank * B + i, L) for i in range(B)]
        label = "AES-GCM-128 %d-byte plaintext circuit (AesGcm128Target<%d>, TAG=false), n=2^%d rows, standard_recursion_config" % (L, L, info["degree_bits"])
        metric = "proofs/sec (AES-GCM 1 KiB circuit)" if L == 1024 else "proofs/sec (AES-GCM %d B circuit)" % L
    targets = list(pws[0].map.keys())
    nt = len(targets)
    vals = torch.tensor([[pw.map[t] - (1 << 64) if pw.map[t] >= (1 << 63) else pw.map[t] for t in targets] for pw in pws],
                        dtype=torch.int64, device="cuda")
    proofs = torch.zeros(B * pb, dtype=torch.uint8, device="cuda")
    status = torch.zeros(B, dtype=torch.int32, device="cuda")
    tarr = (C.c_uint64 * nt)(*targets)

    # A call is ordered with the stream it is given: that stream waits for the proofs, so calls issued on ONE stream drain
    # the chip at every step boundary.  Successive steps therefore alternate between two caller streams (both ordered
    # after the input tensors), the way a service keeps two batches in flight: step k+1's witness generation runs under
    # step k's commitments.  Everything is complete before the clock stops (synchronize below).
    callers = [torch.cuda.Stream(), torch.cuda.Stream()]
    for cs in callers:
        cs.wait_stream(torch.cuda.current_stream())
    torch_stream = torch.cuda.current_stream().cuda_stream
    step_no = [0]

    def step():
        cs = callers[step_no[0] % 2].cuda_stream
        step_no[0] += 1
        rc = lib.p2_prove_batch_device(h, B, tarr, nt, vals.data_ptr(), proofs.data_ptr(), status.data_ptr(), cs)
        if rc:
            raise RuntimeError(lib.p2_last_error().decode())

    def sync():
        if lib.p2_circuit_synchronize(h):
            raise RuntimeError(lib.p2_last_error().decode())
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    assert int(status.abs().sum().item()) == 0, "warm-up proofs failed: %s" % status.tolist()
    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if dist:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert int(status.abs().sum().item()) == 0

    # PCIe-inclusive rate (never `value`): the same batch through p2_prove_batch -- host assignments in, proof bytes out,
    # via the handle's pinned staging buffers -- the shape of the reference's `data.prove(pw)` returning bytes to the host
    pcie = None
    if rank == 0 and args.pcie_steps > 0:
        asg = (pkg.api._Assignment * B)()
        keep = []
        for i, pw in enumerate(pws):
            ts, vs = (C.c_uint64 * nt)(*pw.map.keys()), (C.c_uint64 * nt)(*pw.map.values())
            keep.append((ts, vs))
            asg[i].targets, asg[i].values, asg[i].count = ts, vs, nt
        hbuf = C.create_string_buffer(B * pb)
        hstat = (C.c_int * B)()
        assert lib.p2_prove_batch(h, B, asg, hbuf, hstat) == 0, lib.p2_last_error()   # warm-up: staging buffers are allocated here
        t1 = time.perf_counter()
        for _ in range(args.pcie_steps):
            assert lib.p2_prove_batch(h, B, asg, hbuf, hstat) == 0, lib.p2_last_error()
        pdt = time.perf_counter() - t1
        assert not any(hstat)
        assert hbuf.raw[:pb] == bytes(proofs[:pb].cpu().numpy().tobytes()), "host-path proof differs from the device-path proof"
        pcie = {"value": round(B * args.pcie_steps / pdt, 3), "unit": "proofs/s", "steps": args.pcie_steps,
                "note": "p2_prove_batch: host assignments in, proof bytes out (pinned staging, async copies); calls are synchronous, so consecutive batches do not overlap"}

    # per-kernel launch durations, measured live with HIP events on the proving stream (separate, untimed pass)
    roofline = None
    roofline_valu = None
    kernels = {}
    if rank == 0:
        lib.p2_circuit_set_timing(h, 1)
        chunk = int(lib.p2_circuit_chunk_proofs(h)) or (B + 1) // 2  # the size of the chunks the timed steps ran (equal chunks, capped by free HBM)
        rc = lib.p2_prove_batch_device(h, chunk, tarr, nt, vals.data_ptr(), proofs.data_ptr(), status.data_ptr(), torch_stream)  # one chunk = one stream
        assert rc == 0
        sync()
        arr = (pkg.api._KernelTime * 64)()
        k = lib.p2_circuit_get_timing(h, arr, 64)
        lib.p2_circuit_set_timing(h, 0)
        for i in range(min(k, 64)):
            kernels[arr[i].name.decode()] = (arr[i].ms, arr[i].count)
        total = sum(ms for ms, _ in kernels.values())
        dom = max(kernels, key=lambda n: kernels[n][0])
        ms, cnt = kernels[dom]
        if dom == "hash_leaves":
            # launches per chunk: wires (80 live cols), zs (34), quotient (16) trees -> algorithmic bytes averaged
            n, N = 1 << info["degree_bits"], 8 << info["degree_bits"]
            per_chunk = sum(8 * N * c + 32 * N for c in (80, info["num_zs_cols"], info["num_quotient_cols"])) * chunk
            nbytes = per_chunk / 3.0
        else:
            kb = kernel_bytes(dom, info)
            nbytes = kb * chunk if kb else None
        avg_ms = ms / cnt
        # The counter files are evidence taken on ONE build: each records the identity of the kernel sources it was measured on
        # (tools/src_id.py) and is used only while that matches the sources this run is built from -- otherwise the field is
        # null and *_source says "stale", instead of silently pricing another binary.
        def profile(kind):
            return load_profile(kind)
        traffic = None
        tj, traffic_source = profile("traffic")
        try:  # HBM bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE x2 per calibration, WRITE_SIZE), profiles/
            key = {"hash_leaves": "k_hash_leaves", "lde": "k_ntt_r16<true>", "quotient": "k_quotient<false, true>"}.get(dom)
            if tj and key and L == 1024 and chunk == tj.get("chunk") and args.workload == "aes-gcm":
                ent = tj["per_launch_avg_bytes"][key]
                traffic = ent.get("chunk_launches_avg", ent["total"])
        except Exception:  # noqa: BLE001
            traffic = None
        if nbytes:
            ach = nbytes / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_ms": round(avg_ms, 4),
                        "share_of_gpu_time": round(ms / total, 3),
                        "note": "kernel is VALU-issue bound (a Poseidon permutation is 15.5 k VALU instructions), not HBM bound: see roofline_valu and DESIGN.md section 5",
                        "poseidon_perm_per_s": round((24 * (8 << info["degree_bits"]) * chunk / 3.0) / (avg_ms * 1e-3)) if dom == "hash_leaves" else None}
        # The roofline that actually bounds the path: VALU issue.  Instruction counts per launch are a property of the
        # kernel and the workload (rocprofv3 --pmc SQ_INSTS_VALU, profiles/<tag>_valu.json via tools/pmc_valu.py); the
        # duration is this run's own HIP-event measurement; the clock is the one the chip held in the counter pass.
        vjj, valu_source = profile("valu")
        try:
            vj = vjj["kernels"] if vjj else {}
            key = {"hash_leaves": "k_hash_leaves", "lde": "k_ntt_r16<true>", "quotient": "k_quotient<false, true>"}.get(dom)
            if key in vj and L == 1024 and args.workload == "aes-gcm" and chunk == 128:
                e = vj[key]
                clock = e["clock_GHz"] * 1e9
                ach = e["valu_insts"] * 64 / (avg_ms * 1e-3)
                peak = NUM_SIMD * LANES_PER_CYCLE * clock
                perms = 24 * (8 << info["degree_bits"]) * chunk / 3.0
                roofline_valu = {"bound": "valu_issue", "kernel": dom, "achieved": round(ach / 1e12, 2), "peak": round(peak / 1e12, 2), "unit": "T lane-ops/s",
                                 "frac": round(ach / peak, 4), "clock_GHz": e["clock_GHz"], "valu_insts_per_launch": e["valu_insts"], "valu_source": valu_source,
                                 "insts_per_perm": round(e["valu_insts"] * 64 / perms) if dom == "hash_leaves" else None,
                                 "cycles_per_inst": round(NUM_SIMD * clock * avg_ms * 1e-3 / e["valu_insts"], 3),
                                 "note": "peak = 256 CU x 4 SIMD x 32 lanes x clock (one wave64 instruction per 2 cycles); tools/microbench/valu_rates.hip measures 2.3 cycles for "
                                         "plain two-source ops and 4.1 for v_mad_u64_u32 / carry / select / three-source ops, so this instruction mix cannot exceed ~0.6"}
        except Exception:  # noqa: BLE001
            roofline_valu = None
        if roofline_valu is None and L == 1024 and args.workload == "aes-gcm":
            roofline_valu = {"bound": "valu_issue", "kernel": dom, "achieved": None, "valu_source": valu_source}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # the CPU baseline is reported at N = 1 only
        import oracle_lib  # the checker, used here only as the reported CPU baseline
        oc = oracle_lib.OracleCircuit(data.blob)
        # A one-GPU box gives its container 16 of the host's 2 x 64 EPYC 9575F cores (cgroup cpu.max), and that is where the port
        # stops scaling: one proof takes 0.41 s on 16 threads, 0.44 s on 32, 0.63 s on 64 or 8, 1.0 s on 128, 3.5 s on one
        # (tools/orc_scale.py fast, profiles/r03_cpu_baseline_scaling.log).  Threads = the quota, at most --cpu-threads.
        all_cores = oracle_lib.lib().orc_num_threads()
        quota = cpu_quota_cores()  # the GPU boxes hand a container a share of the host's cores: threads beyond it only add switches
        cores = max(1, min(all_cores, args.cpu_threads, int(quota + 0.999) if quota else all_cores))
        oracle_lib.lib().orc_set_num_threads(cores)
        # the oracle's faster form of its own hash (sparse partial rounds, lazy reduction, one leaf per SIMD lane:
        # oracle/oracle_poseidon_sparse.h, oracle_poseidon_simd.h; held to the textbook permutation by tests/test_oracle_kat.py):
        # the baseline should not be handicapped by a 30-round scalar loop
        oracle_lib.lib().orc_set_fast_hash(1)
        lanes = oracle_lib.lib().orc_set_simd_lanes(8)  # 8 = AVX-512, 4 = AVX2, 1 = scalar: whatever the host has
        st, ref = oc.prove(pws[0].map)  # warm-up (page faults, OpenMP team start-up)
        times = []
        for i in range(args.cpu_sample):
            t1 = time.perf_counter()
            st, ref = oc.prove(pws[i].map)
            times.append(time.perf_counter() - t1)
            assert st == 0
        cdt = sorted(times)[len(times) // 2] * args.cpu_sample  # median proof time
        got = bytes(proofs[: args.cpu_sample * pb].cpu().numpy().tobytes())
        assert got[(args.cpu_sample - 1) * pb: args.cpu_sample * pb] == ref, "GPU proof differs from the oracle's"
        stages = {k: round(v, 4) for k, v in oracle_lib.OracleCircuit.last_stage_seconds().items()}  # of the last proof timed
        # thread scaling: ONE proof on one thread (BASELINE.md asks for the core count used and how the port scales)
        oracle_lib.lib().orc_set_num_threads(1)
        t1 = time.perf_counter()
        st, _ = oc.prove(pws[0].map)
        one = time.perf_counter() - t1
        stages1 = {k: round(v, 4) for k, v in oracle_lib.OracleCircuit.last_stage_seconds().items()}
        oracle_lib.lib().orc_set_num_threads(cores)
        oracle_lib.lib().orc_set_fast_hash(0)
        cpu_baseline = {"value": round(args.cpu_sample / cdt, 4), "unit": "proofs/s", "cores": cores,
                        "kind": "port", "sample": "median of %d proofs after 1 warm-up, same workload (%s; C++ restatement with its fast hash -- sparse partial rounds, %d leaves per SIMD call -- and OpenMP; not the Rust reference)" % (args.cpu_sample, label.split(" (")[0], lanes),
                        "simd_lanes": lanes, "cpu_quota_cores": quota,
                        "stage_seconds": stages, "single_thread": {"value": round(1.0 / one, 4), "unit": "proofs/s", "stage_seconds": stages1,
                                                                   "speedup_at_cores": round(one * args.cpu_sample / cdt, 2), "host_threads_available": all_cores}}

    if rank == 0:
        total_proofs = B * args.steps * world
        pbp = path_bytes_per_proof(info, 80)  # neither workload has Poseidon gates: only the 80 routed wire columns carry data
        whole_path = {"bytes_per_proof_formula": int(pbp), "achieved_GBps_per_gpu": round(pbp * total_proofs / dt / world / 1e9, 1),
                      "frac_of_hbm_peak": round(pbp * total_proofs / dt / world / 1e9 / HBM_PEAK_GBS, 4),
                      "note": "SURVEY.md 8(d) formula with this build's column counts; the path is Poseidon/VALU bound, see roofline.note"}
        out = {
            "metric": metric, "value": round(total_proofs / dt, 3), "unit": "proofs/s",
            "n_gpus": world, "ranks_seen": dist.get_world_size() if dist else 1, "devices": devices, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 (Goldilocks field)", "data": "synthetic",
            "config": {"workload": label, "proofs_per_step_per_gpu": B, "proof_bytes": pb, "parallelism": "independent proofs sharded by index"},
            "roofline": roofline, "roofline_valu": roofline_valu, "cpu_baseline": cpu_baseline,
            "value_pcie_inclusive": pcie,
            "whole_path": whole_path,
            "chunk_proofs": chunk if rank == 0 else None, "kernels_ms_per_chunk": {k: round(v[0], 3) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1][0])},
        }
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
