#!/usr/bin/env python3
"""bench.py -- ML-KEM-768 encapsulations/sec on MI355X (BASELINE.json's metric), plus every other BASELINE config.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mode M] [--extras contract|all|none] [--no-cpu-baseline] [--no-pmc] [--sample-parity]

OUTPUT: the LAST stdout line is the bench contract's JSON line and nothing but the contract (contract_line(): metric / value / config /
parity / roofline / cpu_baseline / value_host_abi / strong + one figure and verdict per other BASELINE config; < 6 KB, strict JSON,
tests/test_bench_line.py).  The full record -- every config's kernel times, rooflines, parity details, probes, notes and definitions --
is written to bench_extras.json next to this file (--extras-file), which the line names as "extras".

A "step" is one pass of the hot path over one batch of synthetic inputs that are already resident in HBM when the timed
region starts.  The headline (default --mode encaps) is BASELINE.json configs[1]: "ML-KEM-768 Encapsulate batch=2^20 on
1xMI355X", distinct-key form of SURVEY.md section 8(d): ek_i are valid keys from seeded keygen
(d||z = SHAKE256("circl-hip/keygen" || LE64(i))[:64]), ONE KEY PER ITEM (2^20 keys made by the GPU keygen, untimed);
m_i = SHAKE256("circl-hip/m" || LE64(i))[:32] are all distinct.  Every ciphertext and shared secret of the batch is compared with the
oracle (--sample-parity: a 2^16 sample instead).  `configs.pooled` keeps the figure of earlier rounds (a pool of 2^16 keys cycled 16x).

The full record carries, under "configs", a measured figure (own HIP-event kernel times, own roofline against SURVEY
8(d)'s algorithmic bytes, own sampled oracle parity) for every other BASELINE config on this rank's GPU:

    decaps        ML-KEM-768 Decapsulate 2^20 of the ciphertexts just produced (config 3's second half; all ss_dec == ss_enc)
    config3       Encaps + Decaps pairs, 2^20 per GPU (2^23 over 8 GPUs)
    config4       ML-DSA-65 Verify 2^18, DISTINCT keys (GPU keygen + GPU deterministic signing), >= 1 % corrupted signatures
    config5       ML-KEM-1024 Encapsulate + ML-DSA-87 Verify submitted concurrently on two streams (per-GPU share 2^16 + 2^16)
    host_abi      the host-buffer C ABI end to end (H2D + kernels + D2H), page-locked and ordinary pageable buffers
and with --extras all (opt-in: the default run is the contract's):
    pooled               the figure of rounds 1-4 (keys from a pool of 2^16)
    shared_key / keyed   one key for the batch / a table of 1000 keys (the reference's parsed-key cache)
    hybrid               X-Wing and X25519MLKEM768 on resident arrays, 2^18 per GPU
    small_batches / concurrent_callers   per-call cost of small batches; T one-item callers through the coalescer and R reactors on the asynchronous form (tools/bin/concurrent_bench)

--mode config3 | config4 | config5 | host makes that workload the headline (metric / value / ms_per_step) instead.

N > 1: one rank per GPU.  `python bench.py --gpus N` without a launcher re-executes itself as N ranks under torch.distributed.run
(127.0.0.1, a free port); under a launcher it refuses to run (exit code 3) unless WORLD_SIZE == N and the box has N GPUs.  Every rank
owns its own batch of --batch items (`value`: weak scaling, per-GPU work fixed as N grows); `strong` in the same line is ONE batch of
--batch items split n/G per rank (the metric read literally: SURVEY.md 8e); `value_host_abi` is the same metric through the
host-pointer C ABI (PCIe-inclusive).  There is no data-path collective (the only collectives are the timing barrier and the
reductions of timings: RCCL); rank 0 prints ONE JSON line with whole-job aggregates and per-rank rates.

`cpu_baseline` (rank 0, N = 1): the same workload on the host cores this process may use -- `value` = oracle/vec, a batch-vectorised
port (16 / 32 items per AVX2 / AVX-512 vector, Keccak on 4 / 8 states; bytes checked against the scalar oracle in the run), with the
scalar oracle (the restatement of CIRCL's generic Go the parity legs use) as `scalar_oracle`; CIRCL's own AVX2 path needs a Go toolchain
(`cpu_baseline_reference` runs it when $CIRCL_REFERENCE and `go` exist).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
POOL = 1 << 20                    # keys per rank: one per item at the default batch (SURVEY 8d allowed a pool of 2^16 "for keygen cost reasons":
POOLED = 1 << 16                  # GPU keygen makes 2^20 keys in 7 ms, so the reason is gone; the old figure stays as configs.pooled)
# algorithmic bytes per operation = inputs read once + outputs written once
BYTES = {"mlkem768_encaps": 1184 + 32 + 1088 + 32, "mlkem768_encaps_shared": 32 + 1088 + 32, "mlkem768_decaps": 2400 + 1088 + 32,
         "mlkem1024_encaps": 1568 + 32 + 1568 + 32, "mldsa65_verify": 1952 + 3309 + 32 + 1, "mldsa87_verify": 2592 + 4627 + 32 + 1}


def shake_seeds(label, count, outlen, start=0):
    out = np.empty((count, outlen), np.uint8)
    lab = label.encode()
    for i in range(count):
        out[i] = np.frombuffer(hashlib.shake_256(lab + (start + i).to_bytes(8, "little")).digest(outlen), np.uint8)
    return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def lib_sha256():
    from circl_amd import _native
    h = hashlib.sha256()
    with open(_native.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


class Timer:
    """K steps bracketed by barrier + synchronize on both sides; per-kernel HIP-event times from the library's own brackets."""

    def __init__(self, ranks, kernels):
        from circl_amd import device as cdev
        self.ranks, self.kernels, self.cdev = ranks, kernels, cdev

    def run(self, step, steps, warmup):
        cdev = self.cdev
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        for k in self.kernels:
            cdev.profile_read(k)
        cdev.profile_enable(True)
        self.ranks.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.ranks.barrier()
        elapsed = time.perf_counter() - t0
        cdev.profile_enable(False)
        kern = {}
        for k in self.kernels:
            ms, cnt = cdev.profile_read(k)
            kern[k] = {"total_ms": ms, "launch_groups": cnt, "ms_per_step": ms / max(steps, 1)}
        return elapsed, kern


def roofline(kernel_name, ops_per_step, bytes_per_op, kernel_ms_per_step, note=None, traffic=None):
    achieved = ops_per_step * bytes_per_op / (kernel_ms_per_step * 1e-3) / 1e9 if kernel_ms_per_step else None
    r = {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBPS if achieved else None, "traffic": traffic,
         "algorithmic_bytes_per_launch": ops_per_step * bytes_per_op, "avg_launch_ms": kernel_ms_per_step}
    if note:
        r["note"] = note
    return r


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------
class KemWork:
    """ML-KEM encaps / decaps over B resident items (distinct keys: pool of min(2^16, B) GPU-generated keys, tiled)."""

    def __init__(self, param, B, rank, dev, shake_inputs=True, window=None, pool_max=None):
        """window = (lo, total): this rank's B = hi - lo items are items [lo, hi) of ONE batch of `total` items (strong scaling:
        the batch is the one rank 0 of a weak run owns -- item i has key pool[i mod pool] and message SHAKE256(label || LE64(i)))."""
        from circl_amd import device as cdev
        self.param, self.B, self.dev = param, B, dev
        pool = min(pool_max or POOL, B if window is None else window[1])
        if shake_inputs:
            seeds = torch.from_numpy(shake_seeds("circl-hip/keygen", pool, 64, start=(rank * POOL if window is None else 0))).to(dev)
            self.m = torch.from_numpy(shake_seeds("circl-hip/m", B, 32, start=(rank * B if window is None else window[0]))).to(dev)
        else:
            g = torch.Generator(device=dev).manual_seed(1000 + rank)
            seeds = torch.randint(0, 256, (pool, 64), dtype=torch.uint8, device=dev, generator=g)
            self.m = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device=dev, generator=g)
        kg = cdev.MLKEMDevice(param, pool, dev)
        ek_pool, dk_pool = kg.keygen(seeds)   # GPU keygen is itself parity-pinned (tests/test_gpu_mlkem.py: ACVP keyGen, KAT hashes)
        torch.cuda.synchronize()
        first = 0 if window is None else window[0] % pool
        reps = (first + B + pool - 1) // pool
        self.pool = pool
        self.seeds = seeds
        if reps == 1 and first == 0 and pool == B:  # one key per item: the keygen's own arrays
            self.ek, self.dk = ek_pool, dk_pool
        else:
            self.ek = ek_pool.repeat(reps, 1)[first:first + B].contiguous()
            self.dk = dk_pool.repeat(reps, 1)[first:first + B].contiguous()
        self.distinct_keys = bool(pool >= B)
        self.eng = cdev.MLKEMDevice(param, B, dev)
        self.ss_dec = torch.empty((B, 32), dtype=torch.uint8, device=dev)
        self.st_dec = torch.empty(B, dtype=torch.uint8, device=dev)

    def encaps(self):
        self.eng.encaps(self.ek, self.m)

    def decaps(self):
        self.eng.decaps(self.dk, self.eng.ct, self.ss_dec, self.st_dec)

    def parity_encaps(self, sample):
        """sample = None: EVERY item of the batch against the oracle (north_star: "every ciphertext, shared secret ... bit-exact")"""
        from oracle import orc
        t = time.perf_counter()
        if sample is None or sample >= self.B:
            ct0, ss0, st0 = orc.mlkem_encaps(self.param, self.ek.cpu().numpy(), self.m.cpu().numpy())
            ct, ss, n = self.eng.ct.cpu().numpy(), self.eng.ss.cpu().numpy(), self.B
        else:
            idx = torch.from_numpy(np.random.default_rng(0).choice(self.B, size=sample, replace=False)).to(self.dev)
            ct0, ss0, st0 = orc.mlkem_encaps(self.param, self.ek[idx].cpu().numpy(), self.m[idx].cpu().numpy())
            ct, ss, n = self.eng.ct[idx].cpu().numpy(), self.eng.ss[idx].cpu().numpy(), int(idx.numel())
        ok = bool((ct == ct0).all() and (ss == ss0).all() and not st0.any())
        return {"sampled_items": n, "of_items": self.B, "whole_batch": n == self.B, "bit_exact_vs_oracle": ok,
                "status_nonzero": int(self.eng.status.sum().item()), "oracle_seconds": time.perf_counter() - t}

    def parity_decaps(self, sample):
        from oracle import orc
        idx = torch.from_numpy(np.random.default_rng(1).choice(self.B, size=min(self.B, sample), replace=False)).to(self.dev)
        ss0, st0 = orc.mlkem_decaps(self.param, self.dk[idx].cpu().numpy(), self.eng.ct[idx].cpu().numpy())
        ok = bool((self.ss_dec[idx].cpu().numpy() == ss0).all() and not st0.any())
        return {"sampled_items": int(idx.numel()), "bit_exact_vs_oracle": ok, "status_nonzero": int(self.st_dec.sum().item()),
                "all_items_ss_dec_equals_ss_enc": bool((self.ss_dec == self.eng.ss).all().item())}


class DsaWork:
    """ML-DSA verify over n resident items with DISTINCT keys: GPU keygen from seeded seeds, GPU deterministic signing
    (both parity-pinned on the GPU: ACVP keyGen / sigGen, Wycheproof, KAT hashes), >= 1 % of the signatures corrupted
    (bit flip in z, bit flip in c~, non-canonical hint) so that the false paths run."""

    def __init__(self, param, n, rank, dev):
        from circl_amd import device as cdev
        self.param, self.n, self.dev = param, n, dev
        g = torch.Generator(device=dev).manual_seed(4000 + 17 * rank + param)
        seeds = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
        self.msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device=dev, generator=g)
        self.eng = cdev.MLDSADevice(param, n, dev, msg_len=32, sign=True)
        self.pk, sk = self.eng.keygen(seeds)
        t = time.perf_counter()
        self.sig = self.eng.sign(sk, self.msg)
        torch.cuda.synchronize()
        self.sign_s = time.perf_counter() - t
        self.sig_good = self.sig.clone()
        self.sk = sk
        ct = {44: 32, 65: 48, 87: 64}[param]
        bad = torch.arange(0, n, 85, device=dev)          # 1.18 % of the items
        kinds = torch.arange(bad.numel(), device=dev) % 3
        self.sig[bad[kinds == 0], ct + 100] ^= 8          # bit flip in z
        self.sig[bad[kinds == 1], 3] ^= 1                 # bit flip in c~
        self.sig[bad[kinds == 2], self.eng.SIG - 1] = 0xFF  # non-canonical hint (count byte > omega)
        self.want = torch.ones(n, dtype=torch.uint8, device=dev)
        self.want[bad] = 0
        self.n_bad = int(bad.numel())
        self.eng.sws = None  # the signing workspace (60 KB per item) is not needed any more
        torch.cuda.empty_cache()

    def verify(self):
        self.eng.verify(self.pk, self.sig, self.msg)

    def parity(self, sample, sign_sample):
        from oracle import orc
        rng = np.random.default_rng(2)
        t0 = time.perf_counter()
        idx = np.arange(self.n) if (sample is None or sample >= self.n) else np.sort(rng.choice(self.n, size=sample, replace=False))
        ti = torch.from_numpy(idx).to(self.dev)
        msgs_all = self.msg[:self.n * 32].view(self.n, 32)
        msgs = [bytes(r) for r in msgs_all[ti].cpu().numpy()]
        ok0 = orc.mldsa_verify(self.param, self.pk[ti].cpu().numpy(), self.sig[ti].cpu().numpy(), msgs)
        got = self.eng.ok
        res = {"sampled_items": int(len(idx)), "of_items": self.n, "whole_batch": len(idx) == self.n, "oracle_seconds": time.perf_counter() - t0,
               "bit_exact_vs_oracle": bool((got[ti].cpu().numpy() == ok0).all()),
               "all_items_as_expected": bool((got == self.want).all().item()), "corrupted_items": self.n_bad,
               "rejected_items": int((got == 0).sum().item())}
        # the GPU-made inputs themselves: signatures equal the oracle's deterministic signatures on a sub-sample
        sidx = idx[:sign_sample]
        si = torch.from_numpy(sidx).to(self.dev)
        sig0 = orc.mldsa_sign(self.param, self.sk[si].cpu().numpy(), msgs[:len(sidx)])
        res["gpu_signatures_equal_oracle"] = {"sampled_items": int(len(sidx)), "bit_exact_vs_oracle": bool((self.sig_good[si].cpu().numpy() == sig0).all())}
        return res


def small_batches(dev):
    """Cost per call of small / medium batches on resident arrays (stream-ordered, 20 calls per sample, best of 3): the routes of
    DESIGN.md 4.4 / 4.5.  Every size is checked: all decapsulated secrets equal the encapsulated ones, a sample of each batch
    against the oracle; ML-DSA signatures verify and a sample equals the oracle's."""
    from circl_amd import device as cdev, hostapi
    from oracle import orc

    def best(fn, reps=20):  # stream-ordered calls back to back, one synchronisation per `reps` (tests/gpu_microbench.py measures the same way)
        fn()
        torch.cuda.synchronize()
        b = 1e9
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            b = min(b, (time.perf_counter() - t) / reps)
        return b * 1e6

    out = {"unit": "us per call", "mlkem768": {}, "mldsa65": {}, "bit_exact_vs_oracle": True}
    g = torch.Generator(device=dev).manual_seed(77)
    for n in (1, 1 << 10, 1 << 12, 1 << 14):
        eng = cdev.MLKEMDevice(768, n, dev)
        ek, dk = eng.keygen(torch.randint(0, 256, (n, 64), dtype=torch.uint8, device=dev, generator=g))
        m = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
        ct = torch.empty((n, eng.CT), dtype=torch.uint8, device=dev)
        ss = torch.empty((n, 32), dtype=torch.uint8, device=dev)
        ss2 = torch.empty_like(ss)
        te = best(lambda: eng.encaps(ek, m, ct, ss))
        td = best(lambda: eng.decaps(dk, ct, ss2))
        t1 = best(lambda: eng.encaps_shared(ek[:1], m, ct, ss))
        tab = hostapi.KeyTable("mlkem-public", 768, ek[:1].cpu().numpy())   # the key parsed ONCE and resident (circl_hip_mlkem_keytable_new)
        tt = best(lambda: eng.encaps_table(tab, m, ct=ct, ss=ss))
        ct_t = ct.clone()
        eng.encaps_shared(ek[:1], m, ct, ss)
        torch.cuda.synchronize()
        out["bit_exact_vs_oracle"] &= bool((ct_t == ct).all().item())
        tab.close()
        eng.encaps(ek, m, ct, ss)
        torch.cuda.synchronize()
        k = min(n, 64)
        ct0, ss0, _ = orc.mlkem_encaps(768, ek[:k].cpu().numpy(), m[:k].cpu().numpy())
        ok = bool((ss2 == ss).all().item()) and bool((ct[:k].cpu().numpy() == ct0).all()) and bool((ss[:k].cpu().numpy() == ss0).all())
        out["bit_exact_vs_oracle"] &= ok
        out["mlkem768"][str(n)] = {"encaps": te, "decaps": td, "encaps_one_key": t1, "encaps_resident_key": tt, "encaps_per_s": n / te * 1e6,
                                   "encaps_resident_key_per_s": n / tt * 1e6}
    for n in (1, 1 << 10):
        eng = cdev.MLDSADevice(65, n, dev, sign=True)
        kseeds = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
        pk, sk = eng.keygen(kseeds)
        tk = best(lambda: eng.keygen(kseeds), 10)
        pk0, sk0 = orc.mldsa_keygen(65, kseeds[:min(n, 4)].cpu().numpy())
        out["bit_exact_vs_oracle"] &= bool((pk[:len(pk0)].cpu().numpy() == pk0).all() and (sk[:len(sk0)].cpu().numpy() == sk0).all())
        msg = torch.randint(0, 256, (n * 32 + 16,), dtype=torch.uint8, device=dev, generator=g)
        sig = eng.sign(sk, msg)
        ts = best(lambda: eng.sign(sk, msg, sig), 10)
        tv = best(lambda: eng.verify(pk, sig, msg))
        vtab = hostapi.KeyTable("mldsa-public", 65, pk[:1].cpu().numpy())
        stab = hostapi.KeyTable("mldsa-private", 65, sk[:1].cpu().numpy())
        sig1 = eng.sign_table(stab, msg)
        ok1 = bool(eng.verify_table(vtab, sig1, msg).all().item())
        tsr = best(lambda: eng.sign_table(stab, msg, sig1), 10)
        tvr = best(lambda: eng.verify_table(vtab, sig1, msg))
        out["bit_exact_vs_oracle"] &= ok1 and bool((orc.mldsa_sign(65, sk[:1].cpu().numpy(), [bytes(msg[:32].cpu().numpy())]) == sig1[:1].cpu().numpy()).all())
        vtab.close()
        stab.close()
        k = min(n, 4)
        msgs = [bytes(msg[32 * i:32 * i + 32].cpu().numpy()) for i in range(k)]
        ok = bool(eng.verify(pk, sig, msg).all().item()) and bool((orc.mldsa_sign(65, sk[:k].cpu().numpy(), msgs) == sig[:k].cpu().numpy()).all())
        out["bit_exact_vs_oracle"] &= ok
        out["mldsa65"][str(n)] = {"keygen": tk, "sign": ts, "verify": tv, "sign_resident_key": tsr, "verify_resident_key": tvr}
    return out


def concurrent_callers(seconds=1.0):
    """The shape of every existing kem.Scheme consumer (kem/hybrid/hybrid.go:95-99, hpke/algs.go:283-285): T host threads, each calling ONE
    resident-key encapsulation at a time through host buffers, closed loop -- with and without cross-caller coalescing
    (circl_hip_keytable_set_coalesce).  tools/bin/concurrent_bench (C++, built by __graft_entry__.build()) does the calling and compares every
    result with an ordinary batch call; this only parses its lines.  None when the tool is not there."""
    import re
    exe = os.path.join(ROOT, "tools", "bin", "concurrent_bench")
    if not os.path.exists(exe):
        return None
    out = {"unit": "ops/s", "what": "ML-KEM-768, one item per call, 8 resident keys, T caller threads in a closed loop (aggregate <= T / latency), "
                                    "%.0f s per point; latencies in us; host CPUs usable by this process: %d" % (seconds, len(os.sched_getaffinity(0)))}
    for name, args in (("encaps_coalesced", ["encaps", "256", "0"]), ("encaps_uncoalesced", ["encaps", "0", "0"]), ("decaps_coalesced", ["decaps", "256", "0"])):
        try:
            r = subprocess.run([exe] + args + ["1", str(seconds), "1", "32", "64"], capture_output=True, text=True, timeout=120)
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": str(e)[-200:]}
            continue
        pts = {}
        for ln in r.stdout.splitlines():
            m = re.match(r"T=(\d+)\s+(\d+) ops/s.*p50\s+([0-9.]+)\s+p99\s+([0-9.]+).*mismatches (\d+)", ln)
            if m:
                pts["T%s" % m.group(1)] = {"ops_per_s": float(m.group(2)), "p50_us": float(m.group(3)), "p99_us": float(m.group(4)), "mismatches": int(m.group(5))}
                cpu = re.search(r"CPU ([0-9.]+) us per call", ln)
                if cpu:
                    pts["T%s" % m.group(1)]["host_cpu_us_per_call"] = float(cpu.group(1))
        out[name] = pts if (r.returncode == 0 and pts) else {"error": (r.stdout + r.stderr)[-300:]}
    # the asynchronous form (circl_hip_keytable_async_start / *_table_submit / circl_hip_poll; circl_hip_queue for keys with the call): R reactor
    # threads, W one-item requests outstanding each, nobody asleep inside the library per call
    asy = {}
    for op in ("encaps", "decaps", "encaps_call"):
        for R, W in ((1, 256), (4, 128)):
            try:
                r = subprocess.run([exe, "--async", op, str(R), str(W), str(seconds)], capture_output=True, text=True, timeout=120)
                m = re.search(r"(\d+) items/s\s+latency us p50\s+([0-9.]+) p99\s+([0-9.]+).*?host CPU ([0-9.]+) us per item.*?mismatches (\d+)", r.stdout)
                asy["%s_R%d_W%d" % (op, R, W)] = ({"items_per_s": float(m.group(1)), "p50_us": float(m.group(2)), "p99_us": float(m.group(3)),
                                                   "host_cpu_us_per_item": float(m.group(4)), "mismatches": int(m.group(5))}
                                                  if (r.returncode == 0 and m) else {"error": (r.stdout + r.stderr)[-300:]})
            except Exception as e:  # noqa: BLE001
                asy["%s_R%d_W%d" % (op, R, W)] = {"error": str(e)[-200:]}
    out["async_reactors"] = asy
    return out


def host_abi(work, dev_index, runs=3):
    """End to end through circl_hip_mlkem_encaps (host pointers): H2D + kernels + D2H, first with buffers from
    circl_hip_alloc_host (page-locked), then with ordinary pageable numpy memory, as a Go caller's []byte would be."""
    from circl_amd import _native as nat
    L = nat.lib()
    B, EK, CT = work.B, work.eng.EK, work.eng.CT
    ek_np, m_np = work.ek.cpu().numpy(), work.m.cpu().numpy()
    ct_ref = work.eng.ct.cpu().numpy()
    out = {}

    def run(p_ek, p_m, p_ct, p_ss, p_st):
        best, tot = 1e9, 0.0
        for _ in range(runs):
            t = time.perf_counter()
            rc = L.circl_hip_mlkem_encaps(work.param, p_ek, p_m, p_ct, p_ss, p_st, B, dev_index)
            dt = time.perf_counter() - t
            assert rc == 0, rc
            best, tot = min(best, dt), tot + dt
        return best, tot / runs

    def figure(best, mean):
        h2d, d2h = B * (EK + 32), B * (CT + 32 + 1)
        return {"value": B / best, "unit": "encaps/s", "best_s": best, "mean_s": mean, "h2d_GBps": h2d / best / 1e9, "d2h_GBps": d2h / best / 1e9,
                "pcie_gen5_x16_GBps_per_direction": 64.0}

    # page-locked
    bufs = []

    def pinned(nbytes):
        p = L.circl_hip_alloc_host(nbytes)
        assert p
        bufs.append(p)
        return p, np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p))
    p_ek, a_ek = pinned(B * EK); p_m, a_m = pinned(B * 32); p_ct, a_ct = pinned(B * CT); p_ss, a_ss = pinned(B * 32); p_st, a_st = pinned(B)
    a_ek[:] = ek_np.reshape(-1); a_m[:] = m_np.reshape(-1)
    L.circl_hip_mlkem_encaps(work.param, p_ek, p_m, p_ct, p_ss, p_st, B, dev_index)  # warm: staging slots, streams
    best, mean = run(p_ek, p_m, p_ct, p_ss, p_st)
    out["pinned"] = figure(best, mean)
    out["pinned"]["all_ct_equal_device_resident_run"] = bool((a_ct.reshape(B, CT) == ct_ref).all())
    for p in bufs:
        L.circl_hip_free_host(p)
    # pageable, deliberately byte-misaligned sub-slices (a Go sub-slice is 1-byte aligned)
    def pageable(nbytes, off):
        raw = np.zeros(nbytes + 64, np.uint8)  # touched: no first-touch page faults inside the timed call
        return raw[off:off + nbytes]
    g_ek, g_m = pageable(B * EK, 1), pageable(B * 32, 3)
    g_ct, g_ss, g_st = pageable(B * CT, 5), pageable(B * 32, 7), pageable(B, 9)
    g_ek[:] = ek_np.reshape(-1); g_m[:] = m_np.reshape(-1)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    L.circl_hip_mlkem_encaps(work.param, ptr(g_ek), ptr(g_m), ptr(g_ct), ptr(g_ss), ptr(g_st), B, dev_index)
    best, mean = run(ptr(g_ek), ptr(g_m), ptr(g_ct), ptr(g_ss), ptr(g_st))
    out["pageable"] = figure(best, mean)
    out["pageable"]["misaligned_by_bytes"] = [1, 3, 5, 7, 9]
    out["pageable"]["all_ct_equal_device_resident_run"] = bool((g_ct.reshape(B, CT) == ct_ref).all() and not g_st.any())
    out["note"] = ("circl_hip_mlkem_encaps with host pointers, %d items, one device: staged through the library's page-locked slots by its "
                   "host thread pool (pageable) or DMA-ed directly (page-locked); PCIe-inclusive, never the headline value" % B)
    return out


def cpu_baseline(work, budget_s=10.0):
    """The CPU on the same workload, on the host cores this process may use: `value` = oracle/vec (the batch-vectorised port: W items per
    AVX2 / AVX-512 vector, Keccak on 4 / 8 states -- the CPU programmed the way the GPU is, and faster than an AVX2 path that vectorises
    inside one operation), checked here against the scalar oracle; `scalar_oracle` = oracle/kyber.c, the line-by-line restatement of CIRCL's
    generic Go that the parity tests use (distinct-key and shared-key).  Neither is CIRCL's own AVX2 assembler (no Go toolchain)."""
    from oracle import orc
    cores = orc.ncpu()
    ek_np, m_np = work.ek.cpu().numpy(), work.m.cpu().numpy()
    probe = min(len(ek_np), 512 * cores)
    t = time.perf_counter()
    ct_p, ss_p, st_p = orc.mlkem_encaps(work.param, ek_np[:probe], m_np[:probe], threads=cores)
    rate = probe / (time.perf_counter() - t)
    sample = int(min(len(ek_np), max(probe, rate * budget_s * 0.5)))
    t = time.perf_counter()
    orc.mlkem_encaps(work.param, ek_np[:sample], m_np[:sample], threads=cores)
    dt = time.perf_counter() - t
    s2 = int(min(len(m_np), 2.5 * sample * 0.4))
    t = time.perf_counter()
    orc.mlkem_encaps_shared(work.param, ek_np[:1], m_np[:s2], threads=cores)
    dt2 = time.perf_counter() - t
    where = (f"{cores} pthreads = the CPUs this container may use (affinity {len(os.sched_getaffinity(0))}, capped by the cgroup CPU quota)")
    scalar = {"value": sample / dt, "unit": "encaps/s", "cores": cores, "per_thread": sample / dt / cores,
              "shared_key": {"value": s2 / dt2, "unit": "encaps/s", "per_thread": s2 / dt2 / cores,
                             "sample": f"first {s2} messages to one parsed key (orc_mlkem_encaps_cached: A^T, t-hat and H(ek) once per thread), {dt2:.1f} s"},
              "sample": f"first {sample} items of the same batch, {dt:.1f} s, oracle/liborc.so (-O3 -march=x86-64-v3) with {where}; scalar C "
                        "restatement of CIRCL's generic Go"}
    vec = None
    try:
        isa = orc.vec_isa() if work.param in (768, 1024) else 0
        if isa:
            ct_v, ss_v, st_v = orc.mlkem_encaps_vec(work.param, ek_np[:probe], m_np[:probe], threads=cores)
            same = bool((ct_v == ct_p).all() and (ss_v == ss_p).all() and (st_v == st_p).all())
            t = time.perf_counter()
            orc.mlkem_encaps_vec(work.param, ek_np[:probe * 4], m_np[:probe * 4], threads=cores)
            vrate = min(len(ek_np), probe * 4) / (time.perf_counter() - t)
            vs = int(min(len(ek_np), max(probe, vrate * budget_s * 0.4)))
            reps = max(1, min(16, int(vrate * budget_s * 0.4 / vs)))  # the batch bounds the sample: pass over it again until ~4 s are spent
            t = time.perf_counter()
            for _ in range(reps):
                orc.mlkem_encaps_vec(work.param, ek_np[:vs], m_np[:vs], threads=cores)
            dtv = (time.perf_counter() - t) / reps
            # one parsed key for the batch: the shape of the reference's own BenchmarkEncapsulate (kem/schemes/schemes_test.go:28-38)
            ct_s0, ss_s0 = orc.mlkem_encaps_shared(work.param, ek_np[:1], m_np[:probe], threads=cores)
            ct_s, ss_s, st_s = orc.mlkem_encaps_vec(work.param, ek_np[:1], m_np[:probe], threads=cores, shared=True)
            same_s = bool((ct_s == ct_s0).all() and (ss_s == ss_s0).all() and not st_s.any())
            reps_s = max(1, min(32, int(2.5 * vrate * budget_s * 0.2 / len(m_np))))
            t = time.perf_counter()
            for _ in range(reps_s):
                orc.mlkem_encaps_vec(work.param, ek_np[:1], m_np, threads=cores, shared=True)
            dts = (time.perf_counter() - t) / reps_s
            vec = {"value": vs / dtv, "per_thread": vs / dtv / cores, "isa": {1: "AVX2 (16 items per vector, Keccak x4)", 2: "AVX-512 (32 items per vector, Keccak x8)"}[isa],
                   "equals_scalar_oracle_on_first_items": [probe, same],
                   "shared_key": {"value": len(m_np) / dts, "unit": "encaps/s", "per_thread": len(m_np) / dts / cores, "equals_scalar_oracle_on_first_items": [probe, same_s],
                                  "sample": f"all {len(m_np)} messages to one parsed key (th, A^T, H(ek) once per thread), {reps_s} pass(es) of {dts:.2f} s"},
                   "sample": f"first {vs} items of the same batch, {reps} pass(es) of {dtv:.2f} s, oracle/liborcvec.so with {where}"}
    except Exception as e:  # the scalar figure stands on its own
        vec = {"error": repr(e)[:200]}
    # every figure ONCE: `value` / `per_thread` / `shared_key` are the vector port's when it ran and agreed with the scalar oracle (then the
    # scalar oracle's own figures sit under `scalar_oracle`), the scalar oracle's otherwise
    out = {"value": scalar["value"], "unit": "encaps/s", "cores": cores, "kind": "port", "per_thread": scalar["per_thread"], "cpu": cpu_model(),
           "isa": "scalar C (-O3 -march=x86-64-v3)", "shared_key": scalar["shared_key"],
           "sample": scalar["sample"] + " (Go toolchain absent, so not CIRCL's AVX2 path)"}
    if vec and vec.get("value") and vec["equals_scalar_oracle_on_first_items"][1]:
        out.update({"value": vec["value"], "per_thread": vec["per_thread"], "isa": vec["isa"],
                    "equals_scalar_oracle_on_first_items": vec["equals_scalar_oracle_on_first_items"],
                    "sample": vec["sample"] + "; oracle/vec/mlkem_vec.c, the batch-vectorised port (items side by side in the vector lanes; bytes equal "
                              "to the scalar oracle's, tests/test_oracle_vec.py).  Not CIRCL's own AVX2 assembler (no Go toolchain on any box), which vectorises "
                              "inside one operation; the scalar restatement of its generic Go is `scalar_oracle`",
                    "scalar_oracle": {k: scalar[k] for k in ("value", "unit", "per_thread", "sample", "shared_key")}})
        if vec["shared_key"]["equals_scalar_oracle_on_first_items"][1]:
            out["shared_key"] = vec["shared_key"]
    elif vec:
        out["vectorized_failed"] = vec
    return out


GO_HARNESS = r'''// Written by circl-hip's bench.py and placed into kem/schemes by `go test -overlay` (nothing is written into the
// reference tree).  BASELINE.md section 5: all cores, distinct key: UnmarshalBinaryPublicKey + EncapsulateDeterministically
// per item on the same synthetic arrays the GPU run used.
package schemes_test

import (
	"encoding/binary"
	"os"
	"sync/atomic"
	"testing"

	"github.com/cloudflare/circl/kem/schemes"
)

func circlHipInputs(tb testing.TB) (int, []byte, []byte) {
	raw, err := os.ReadFile(os.Getenv("CIRCL_HIP_BENCH_INPUT"))
	if err != nil {
		tb.Fatal(err)
	}
	n := int(binary.LittleEndian.Uint64(raw[:8]))
	return n, raw[8 : 8+n*1184], raw[8+n*1184 : 8+n*1184+n*32]
}

func BenchmarkCirclHipDistinctKey(b *testing.B) {
	s := schemes.ByName("ML-KEM-768")
	n, ek, m := circlHipInputs(b)
	var ctr int64
	b.ResetTimer()
	b.RunParallel(func(pb *testing.PB) {
		for pb.Next() {
			i := int(atomic.AddInt64(&ctr, 1)) % n
			pk, err := s.UnmarshalBinaryPublicKey(ek[i*1184 : (i+1)*1184])
			if err != nil {
				b.Fatal(err)
			}
			if _, _, err = s.EncapsulateDeterministically(pk, m[i*32:(i+1)*32]); err != nil {
				b.Fatal(err)
			}
		}
	})
}

func TestCirclHipParity(t *testing.T) {
	s := schemes.ByName("ML-KEM-768")
	n, ek, m := circlHipInputs(t)
	if n > 256 {
		n = 256
	}
	out := make([]byte, 0, n*(1088+32))
	for i := 0; i < n; i++ {
		pk, err := s.UnmarshalBinaryPublicKey(ek[i*1184 : (i+1)*1184])
		if err != nil {
			t.Fatal(err)
		}
		ct, ss, err := s.EncapsulateDeterministically(pk, m[i*32:(i+1)*32])
		if err != nil {
			t.Fatal(err)
		}
		out = append(append(out, ct...), ss...)
	}
	if err := os.WriteFile(os.Getenv("CIRCL_HIP_BENCH_OUTPUT"), out, 0o600); err != nil {
		t.Fatal(err)
	}
}
'''


def cpu_baseline_reference(work, budget_s=8.0):
    """BASELINE.md section 5 option A: CIRCL itself, when the box has a Go toolchain and $CIRCL_REFERENCE points at a checkout
    (nothing here reads /root/reference): the reference's own BenchmarkEncapsulate/ML-KEM-768 (one core, parsed key; AVX2 and
    -tags purego) and an all-cores distinct-key harness over the first items of this very batch, placed into kem/schemes by
    `go test -overlay`; its first 256 ciphertexts / shared secrets are compared with the GPU's.  Returns None (and says why on
    stderr) when any of that is unavailable: the caller then times the oracle ('port')."""
    import re
    import tempfile
    go, ref = shutil.which("go"), os.environ.get("CIRCL_REFERENCE")
    if not go or not ref or not os.path.isdir(os.path.join(ref, "kem", "schemes")):
        return None
    tmp = tempfile.mkdtemp(prefix="circl_go_", dir="/tmp")
    try:
        nsamp = min(work.B, 1 << 16)
        ek, m = work.ek[:nsamp].cpu().numpy(), work.m[:nsamp].cpu().numpy()
        with open(os.path.join(tmp, "in.bin"), "wb") as f:
            f.write(nsamp.to_bytes(8, "little") + ek.tobytes() + m.tobytes())
        with open(os.path.join(tmp, "zz_circl_hip_test.go"), "w") as f:
            f.write(GO_HARNESS)
        with open(os.path.join(tmp, "overlay.json"), "w") as f:
            json.dump({"Replace": {os.path.join(ref, "kem", "schemes", "zz_circl_hip_test.go"): os.path.join(tmp, "zz_circl_hip_test.go")}}, f)
        env = dict(os.environ, CIRCL_HIP_BENCH_INPUT=os.path.join(tmp, "in.bin"), CIRCL_HIP_BENCH_OUTPUT=os.path.join(tmp, "out.bin"),
                   GOFLAGS=os.environ.get("GOFLAGS", "-mod=mod"), GOCACHE=os.path.join(tmp, "gocache"))
        cores = len(os.sched_getaffinity(0))

        def gotest(args, timeout):
            r = subprocess.run([go, "test"] + args + ["./kem/schemes"], cwd=ref, env=env, capture_output=True, text=True, timeout=timeout)
            if r.returncode != 0:
                raise RuntimeError((r.stdout + r.stderr)[-600:])
            return r.stdout

        def ns_per_op(out, name):
            mm = re.search(re.escape(name) + r"\S*\s+\d+\s+([0-9.]+) ns/op", out)
            if not mm:
                raise RuntimeError("no ns/op for " + name)
            return float(mm.group(1))
        bt = "%ds" % max(1, int(budget_s / 4))
        single = ns_per_op(gotest(["-run", "^$", "-bench", "BenchmarkEncapsulate/ML-KEM-768$", "-benchtime", bt], 300), "BenchmarkEncapsulate/ML-KEM-768")
        purego = ns_per_op(gotest(["-tags", "purego", "-run", "^$", "-bench", "BenchmarkEncapsulate/ML-KEM-768$", "-benchtime", bt], 300),
                           "BenchmarkEncapsulate/ML-KEM-768")
        ov = ["-overlay", os.path.join(tmp, "overlay.json")]
        gotest(ov + ["-run", "TestCirclHipParity", "-count", "1"], 300)
        ref_out = np.fromfile(os.path.join(tmp, "out.bin"), np.uint8).reshape(-1, 1088 + 32)
        k = len(ref_out)
        same = bool((ref_out[:, :1088] == work.eng.ct[:k].cpu().numpy()).all() and (ref_out[:, 1088:] == work.eng.ss[:k].cpu().numpy()).all())
        par = ns_per_op(gotest(ov + ["-run", "^$", "-bench", "BenchmarkCirclHipDistinctKey", "-benchtime", bt, "-cpu", str(cores)], 300), "BenchmarkCirclHipDistinctKey")
        ver = subprocess.run([go, "version"], capture_output=True, text=True).stdout.strip()
        return {"value": 1e9 / par, "unit": "encaps/s", "cores": cores, "kind": "reference", "cpu": cpu_model(), "go": ver,
                "gpu_equals_reference_on_first_items": {"items": int(k), "bit_exact": same},
                "single_core_parsed_key": {"avx2_ns_per_op": single, "purego_ns_per_op": purego, "value": 1e9 / single, "unit": "encaps/s",
                                           "what": "go test -bench BenchmarkEncapsulate/ML-KEM-768 ./kem/schemes (kem/schemes/schemes_test.go:28-38)"},
                "sample": f"cloudflare/circl at $CIRCL_REFERENCE, {ver}: b.RunParallel over the first {nsamp} items of the same batch (UnmarshalBinaryPublicKey + "
                          f"EncapsulateDeterministically per item), GOMAXPROCS={cores}, benchtime {bt}"}
    except Exception as e:  # noqa: BLE001 -- the reference path is optional; the port is the fallback
        print("bench.py: the Go reference baseline was attempted and failed (%s: %s); timing the C oracle instead" % (type(e).__name__, str(e)[-400:]), file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------------------------------------
# PMC: HBM-side traffic and VALU instruction counts of the dominant kernel, measured by a fresh rocprofv3 pass
# ------------------------------------------------------------------------------------------------------------------
PMC_KERNELS = {  # name in the JSON line -> substring of the rocprofv3 kernel name
    "mlkem768_encrypt": "mlkem_encrypt_kernel<3, 0",
    "mlkem1024_encrypt": "mlkem_encrypt_kernel<4, 0",
    "mldsa65_verify": "mldsa_verify_kernel<65",
    "mldsa87_verify": "mldsa_verify_kernel<87",
}


def pmc_live(batch, timeout_s=300):
    """Runs `bench.py --pmc-child` under rocprofv3 --pmc (own passes, kernel-trace only, as MI355X_MICROARCH.md prescribes)
    and returns per-launch medians for the dominant kernel of every BASELINE config (PMC_KERNELS): read bytes =
    2 x FETCH_SIZE x 1024 (gfx950 reports half of a wide coalesced stream), write bytes = WRITE_SIZE x 1024 (uncalibrated),
    SQ_INSTS_VALU.  The child launches each kernel at the batch size its config is quoted on."""
    import csv
    import glob
    import tempfile
    rp = shutil.which("rocprofv3")
    if not rp:
        return None, "rocprofv3 not found"
    res = {k: {} for k in PMC_KERNELS}
    base = tempfile.mkdtemp(prefix="circl_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    t_end = time.time() + timeout_s
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            left = t_end - time.time()
            if left < 20:
                return None, "time budget of the PMC passes exhausted"
            d = os.path.join(base, counter)
            cmd = [rp, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--batch", str(batch)]
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=left, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            vals = {k: [] for k in PMC_KERNELS}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r.get("Counter_Name") != counter:
                            continue
                        name = r.get("Kernel_Name", "")
                        for k, pat in PMC_KERNELS.items():
                            if pat in name:
                                vals[k].append(float(r["Counter_Value"]))
            if not vals["mlkem768_encrypt"]:
                return None, f"no {counter} rows for the encrypt kernel"
            for k, v in vals.items():
                if v:
                    res[k][counter] = float(np.median(v))  # (the child's launches of a kernel all have the same batch size)
    except Exception as e:  # noqa: BLE001 -- any failure of the optional pass falls back to the committed figures
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(base, ignore_errors=True)
    out = {}
    for k, c in res.items():
        if len(c) == 3:
            out[k] = {"read_bytes": 2.0 * c["FETCH_SIZE"] * 1024.0, "write_bytes": c["WRITE_SIZE"] * 1024.0,
                      "bytes": 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0, "valu_insts": c["SQ_INSTS_VALU"]}
    out["method"] = ("live rocprofv3 --pmc passes of this very library inside this bench run (kernel-trace only, one counter per pass); "
                     "read = 2 x FETCH_SIZE KB (gfx950 correction), write = WRITE_SIZE KB (uncalibrated); L2<->fabric bytes incl. Infinity-Cache hits")
    return out, None


def pmc_committed():
    """profiles/traffic.json + valu.json: the figures of an earlier live pass (written by a run with CIRCL_BENCH_WRITE_PMC set),
    used only when they carry the SHA-256 of the library that is loaded now."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        with open(os.path.join(ROOT, "profiles", "valu.json")) as f:
            v = json.load(f)
        kern = {}
        for k in PMC_KERNELS:
            tk, vk = (t.get("kernels") or {}).get(k), (v.get("kernels") or {}).get(k)
            if tk and vk:
                kern[k] = {"bytes": tk["bytes"], "read_bytes": tk.get("read_bytes"), "write_bytes": tk.get("write_bytes"), "valu_insts": vk["valu_insts"]}
        return {"bytes": t.get("mlkem768_encrypt_bytes_per_launch_2p20"), "valu_insts": v.get("mlkem768_encrypt_valu_insts_per_launch_2p20"),
                "lib_sha256": t.get("lib_sha256"), "kernels": kern}
    except Exception:  # noqa: BLE001
        return None


def pmc_write(live_all, sha, batch):
    """CIRCL_BENCH_WRITE_PMC=<dir>: keep the live figures as traffic.json / valu.json (copied into profiles/ by hand)."""
    d = os.environ.get("CIRCL_BENCH_WRITE_PMC")
    if not d or not live_all:
        return
    os.makedirs(d, exist_ok=True)
    kern = {k: v for k, v in live_all.items() if isinstance(v, dict)}
    with open(os.path.join(d, "traffic.json"), "w") as f:
        json.dump({"mlkem768_encrypt_bytes_per_launch_2p20": kern["mlkem768_encrypt"]["bytes"] if batch == 1 << 20 else None,
                   "batch": batch, "kernels": {k: {kk: v[kk] for kk in ("bytes", "read_bytes", "write_bytes")} for k, v in kern.items()},
                   "lib_sha256": sha, "method": live_all["method"]}, f, indent=1)
    with open(os.path.join(d, "valu.json"), "w") as f:
        json.dump({"mlkem768_encrypt_valu_insts_per_launch_2p20": kern["mlkem768_encrypt"]["valu_insts"] if batch == 1 << 20 else None,
                   "batch": batch, "kernels": {k: {"valu_insts": v["valu_insts"]} for k, v in kern.items()}, "lib_sha256": sha,
                   "method": "rocprofv3 --pmc SQ_INSTS_VALU (own pass, kernel-trace only), median over the launches of bench.py --pmc-child"}, f, indent=1)


VALU_PEAK_WAVE_INSTS_PER_S = 1024 * 2.4e9 / 2.0   # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2 cycles per wave64 VALU op, 2.4 GHz = 1.229e12


KECCAK_ROUND_INSTS = 180   # keccak_dev.h: VALU instructions per round; 24 rounds = 4 320 per wave-level permutation (64 sponges)
# LOWER bounds on the Keccak-f[1600] wave-instructions of one launch over n items: wave-level permutations x 4 320.  (Sponge glue, the
# rare fourth SHAKE128 block and SampleInBall are not counted: they only raise the Keccak share, i.e. LOWER the ceiling's rate.)
KECCAK_WAVE_PERMS = {
    "mlkem768_encrypt": (7, 4),    # a group of 64 / K^2 = 7 items: 3 SHAKE128 blocks of its 63 matrix streams + 1 pass of its 49 PRF streams
    "mlkem1024_encrypt": (4, 4),   # 4 items: 3 blocks of 64 matrix streams + 1 pass of 36 PRF streams
    "mldsa65_verify": (2, 5),      # 2 items x K.L = 30 ExpandA streams, 5 SHAKE128 blocks each
    "mldsa87_verify": (1, 5),      # 1 item x 56 streams
}


def valu_issue(insts, launch_ms, kernel=None, items=None, probe=None):
    """VALU issue of one launch: against the nominal peak, and against a ceiling built from THIS run's live probe
    (circl_hip_profile_valu_probe, same process, same chip, 4 waves per SIMD like the kernels): the launch's Keccak instructions
    (an algorithmic lower bound) cannot issue faster than the library's own Keccak round does on register-resident states, and
    none of its other VALU instructions faster than two-operand integer ops do:
        t_min = (N_keccak / R_keccak + (N - N_keccak) / R_simple) / SIMDs ;  frac_of_mix_ceiling = t_min / t_launch  (<= 1 by construction)."""
    if not insts or not launch_ms:
        return None
    simds, nominal_hz = 1024, 2.4e9
    per_s = insts / (launch_ms * 1e-3)
    out = {"wave_insts_per_launch": insts, "achieved_Ginst_per_s": per_s / 1e9,
           "peak_Ginst_per_s": VALU_PEAK_WAVE_INSTS_PER_S / 1e9, "frac": per_s / VALU_PEAK_WAVE_INSTS_PER_S,
           "peak_definition": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)",
           "cycles_per_inst_per_simd_at_2.4GHz": simds * nominal_hz / per_s, "resident_waves_per_simd": 4}
    if probe and kernel in KECCAK_WAVE_PERMS and items:
        group, perms = KECCAK_WAVE_PERMS[kernel]
        n_k = min(float(insts), -(-items // group) * perms * 24 * KECCAK_ROUND_INSTS)
        r_k, r_s = probe["keccak_insts_per_s_per_simd"], probe["simple_insts_per_s_per_simd"]
        t_min = (n_k / r_k + (insts - n_k) / r_s) / simds
        out.update({"ceiling_source": "live", "probe": probe, "keccak_wave_insts_lower_bound": n_k, "keccak_share_of_valu_insts": n_k / insts,
                    "mix_ceiling_ms": t_min * 1e3, "frac_of_mix_ceiling": t_min / (launch_ms * 1e-3),
                    "frac_of_mix_ceiling_definition": "t_min / t_launch, t_min = (N_keccak / R_keccak + (N - N_keccak) / R_simple) / 1024 SIMDs: N = SQ_INSTS_VALU of "
                                                      "the launch, N_keccak = %d wave-level permutations per %d items x 4320 (a lower bound), R = this run's "
                                                      "circl_hip_profile_valu_probe rates at 4 waves per SIMD; <= 1 by construction" % (perms, group)})
    else:
        out["ceiling_source"] = None
    return out


def live_valu_probe(dev_index):
    from circl_amd import device as cdev
    try:
        k, s_ = cdev.valu_probe(dev_index, 4)
        return {"keccak_insts_per_s_per_simd": k, "simple_insts_per_s_per_simd": s_, "waves_per_simd": 4,
                "keccak_cycles_per_inst_at_2.4GHz": 2.4e9 / k, "simple_cycles_per_inst_at_2.4GHz": 2.4e9 / s_,
                "what": "circl_hip_profile_valu_probe in this process, right after the timed steps: 512 register-resident Keccak-f[1600] permutations "
                        "per lane / 2M two-operand integer instructions per lane, every SIMD, HIP-event timed, best of 3"}
    except Exception as e:  # noqa: BLE001
        print("bench.py: the live VALU probe failed (%s)" % e, file=sys.stderr)
        return None


# ------------------------------------------------------------------------------------------------------------------
# the stdout line: the contract's keys only (a 20 KB line was more than the driver's parser keeps); everything else
# goes to the extras file
# ------------------------------------------------------------------------------------------------------------------
LINE_MAX = 6144


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) else None


def _num(x, digits=6):
    """Floats to 6 significant digits (the line stays short; the extras file keeps full precision)."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _num(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, digits) for v in x]
    return x


def _roof(r):
    if not isinstance(r, dict):
        return None
    o = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "traffic_over_algorithmic"))
    if isinstance(r.get("valu"), dict):
        o["valu"] = _pick(r["valu"], ("frac", "frac_of_mix_ceiling", "ceiling_source"))
    src = r["pmc"].get("source") if isinstance(r.get("pmc"), dict) else r.get("pmc_source")
    o["pmc"] = {"source": src}
    return o


def contract_line(out, extras_name="bench_extras.json"):
    """The ONE stdout line of the bench contract built from the full record `out`: metric / value / config / parity / roofline /
    cpu_baseline / value_host_abi / strong, one short summary per other BASELINE config, and the name of the file that holds the rest.
    Always strict JSON and shorter than LINE_MAX (tests/test_bench_line.py)."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = out.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "key_pool", "parallelism", "mode"))
    if len(line["config"].get("workload") or "") > 400:
        line["config"]["workload"] = line["config"]["workload"][:397] + "..."
    line["parity"] = _pick(out.get("parity"), ("sampled_items", "whole_batch", "bit_exact_vs_oracle", "ranks_failing"))
    line["roofline"] = _roof(out.get("roofline"))
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "cpu", "per_thread", "isa", "go"))
        c["sample"] = (cb.get("sample") or "")[:240]
        for sub in ("scalar_oracle", "shared_key", "port", "single_core_parsed_key"):
            if isinstance(cb.get(sub), dict) and cb[sub].get("value") is not None:
                c[sub] = _pick(cb[sub], ("value", "kind"))
        if isinstance(cb.get("gpu_equals_reference_on_first_items"), dict):
            c["gpu_equals_reference_on_first_items"] = cb["gpu_equals_reference_on_first_items"]
        line["cpu_baseline"] = c
    if isinstance(out.get("value_host_abi"), dict):
        line["value_host_abi"] = _pick(out["value_host_abi"], ("value", "pinned", "unit"))
    if isinstance(out.get("strong"), dict):
        line["strong"] = _pick(out["strong"], ("value", "unit", "ms_per_step", "items_per_rank"))
        par = out["strong"].get("parity")
        if isinstance(par, dict):
            line["strong"]["parity"] = _pick(par, ("sampled_items", "bit_exact_vs_oracle", "ranks_failing"))
    pr = (out.get("per_rank") or {}).get("encaps_per_s")
    if pr and len(pr) > 1:
        line["per_rank"] = {"encaps_per_s": pr}
    # the other BASELINE configs: one figure + verdict each (their rooflines, kernel times and parity details are in the extras file)
    summ = {}
    for name, c in (out.get("configs") or {}).items():
        if not isinstance(c, dict) or name not in ("decaps", "config3", "config4", "config5"):
            continue
        e = _pick(c, ("value", "unit", "ms_per_step"))
        for rk in ("roofline", "roofline_mlkem1024", "roofline_mldsa87"):
            if isinstance(c.get(rk), dict):
                e[rk] = _pick(c[rk], ("frac", "traffic_over_algorithmic"))
                if isinstance(c[rk].get("valu"), dict):
                    e[rk]["valu_frac_of_mix_ceiling"] = c[rk]["valu"].get("frac_of_mix_ceiling")
        par = c.get("parity")
        if isinstance(par, dict):
            if "bit_exact_vs_oracle" in par:
                e["parity"] = _pick(par, ("sampled_items", "whole_batch", "bit_exact_vs_oracle", "all_items_as_expected", "all_items_ss_dec_equals_ss_enc", "ranks_failing"))
            else:
                e["parity"] = {"bit_exact_vs_oracle": all(bool(v.get("bit_exact_vs_oracle")) for v in par.values() if isinstance(v, dict)),
                               "ranks_failing": par.get("ranks_failing")}
        summ[name] = e
    if summ:
        line["configs"] = summ
    line["extras"] = extras_name
    line["bench_wall_s"] = out.get("bench_wall_s")
    text = json.dumps(_num(line), allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_MAX:   # cannot happen with the fields above; never print a line the driver cannot keep
        for k in ("configs", "per_rank", "strong"):
            line.pop(k, None)
            text = json.dumps(_num(line), allow_nan=False, separators=(",", ":"))
            if len(text) < LINE_MAX:
                break
    return text


def _finite(x):
    """NaN / Infinity -> None, so that both files are strict JSON."""
    if isinstance(x, float):
        return x if x == x and x not in (float("inf"), float("-inf")) else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def emit(out, extras_path):
    """Writes the full record to `extras_path` and returns the contract line (which names that file)."""
    out = _finite(out)
    name = os.path.basename(extras_path)
    try:
        with open(extras_path, "w") as f:
            json.dump(out, f, indent=1, allow_nan=False)
            f.write("\n")
    except OSError as e:
        print("bench.py: could not write %s (%s)" % (extras_path, e), file=sys.stderr)
        name = None
    return contract_line(out, name)


# ------------------------------------------------------------------------------------------------------------------
def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks (one per GPU) under
    torch.distributed.run on 127.0.0.1 and a free port, exactly as the driver does for N > 1; returns its exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes: the host driver only has dmabuf IPC
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps of the headline workload (300 x 7.4 ms = 2.2 s)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1 << 20, help="ML-KEM items per GPU per step")
    ap.add_argument("--mode", default="encaps", choices=["encaps", "config3", "config4", "config5", "host"])
    ap.add_argument("--extras", default="contract", choices=["contract", "all", "none"],
                    help="contract (default): the headline + strong + host ABI + BASELINE configs 3/4/5 + roofline + cpu_baseline; all: also the pooled / "
                         "shared-key / keyed / hybrid / small-batch / concurrent-caller legs; none: only the headline workload")
    ap.add_argument("--no-extras", action="store_true", help="same as --extras none")
    ap.add_argument("--extras-file", default=os.path.join(ROOT, "bench_extras.json"),
                    help="where the full record goes (configs, notes, probes, definitions); the stdout line names it as `extras`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (traffic / valu then come from profiles/*.json if they match this build)")
    ap.add_argument("--sample-parity", action="store_true", help="compare 2^16-item samples with the oracle instead of whole batches (headline: 2^20, config 4: 2^18)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    share = bool(os.environ.get("CIRCL_BENCH_SHARE_GPU"))  # test aid: several ranks on one device (then with CIRCL_DIST_BACKEND=gloo; RCCL wants one GPU per rank)
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(3)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print("bench.py: launched with WORLD_SIZE=%s but --gpus %d: the line would report the wrong n_gpus; refusing" %
              (os.environ["WORLD_SIZE"], args.gpus), file=sys.stderr)
        sys.exit(3)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (circl_amd has no CPU path)", file=sys.stderr)
        sys.exit(2)
    if not share and torch.cuda.device_count() < args.gpus:
        print("bench.py: --gpus %d but this process sees %d GPU(s): refusing to time fewer devices than asked for" %
              (args.gpus, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(3)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.pmc_child:
        sys.exit(launch_ranks(args.gpus))      # one rank per GPU, this very command line in every rank
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if share:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from circl_amd import parallel
    ranks = parallel.Ranks("nccl", dev)  # nccl == RCCL on ROCm; used for the barrier / reductions only
    world, rank = ranks.world, ranks.rank
    B = args.batch
    if args.no_extras:
        args.extras = "none"
    extras = args.extras != "none"      # the BASELINE configs' legs (3, 4, 5), strong, host ABI
    more = args.extras == "all"         # opt-in legs: pooled keys, shared key / key table, hybrids, small batches, concurrent callers

    if args.pmc_child:  # a few launches of the dominant kernel of every BASELINE config for the counter passes, nothing else
        w = KemWork(768, B, 0, dev, shake_inputs=False)
        for _ in range(3):
            w.encaps()
        torch.cuda.synchronize()
        del w
        torch.cuda.empty_cache()
        d65 = DsaWork(65, max(B // 4, 64), 0, dev)
        for _ in range(3):
            d65.verify()
        torch.cuda.synchronize()
        del d65
        torch.cuda.empty_cache()
        k1024 = KemWork(1024, max(B // 16, 64), 0, dev, shake_inputs=False)
        d87 = DsaWork(87, max(B // 16, 64), 0, dev)
        for _ in range(3):
            k1024.encaps()
            d87.verify()
        torch.cuda.synchronize()
        return

    t_start = time.perf_counter()
    kem = KemWork(768, B, rank, dev)
    timer_kem = Timer(ranks, ["mlkem_hash", "mlkem_encrypt", "mlkem_decrypt"])
    out_cfg = {}

    # ---- sustained run: >= 2 s of back-to-back steps (clocks ramped, the SMI sampler sees the GPU busy) ----
    for _ in range(3):
        kem.encaps()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        kem.encaps()
    torch.cuda.synchronize()
    est = (time.perf_counter() - t) / 10
    sus_steps = max(args.steps, int(2.2 / est) + 1)
    ranks.barrier()
    t = time.perf_counter()
    for _ in range(sus_steps):
        kem.encaps()
    ranks.barrier()
    sus_s = ranks.max(time.perf_counter() - t)
    sustained = {"steps": sus_steps, "seconds": sus_s, "value": ranks.sum(B * sus_steps) / sus_s, "unit": "encaps/s",
                 "note": "untimed-by-contract pre-run of the same step, >= 2 s, so that clocks and the SMI sampler settle"}

    # ---- the contract's timed region: W warm-up steps, then exactly K steps ----
    headline = {}
    if args.mode in ("encaps", "config3"):
        if args.mode == "encaps":
            step, ops = kem.encaps, B
        else:
            def step():
                kem.encaps()
                kem.decaps()
            ops = B
        elapsed, kern = timer_kem.run(step, args.steps, args.warmup)
        value, elapsed = parallel.whole_job_rate(ranks, ops * args.steps, elapsed)
        headline = {"elapsed": elapsed, "value": value, "kern": kern}
    enc_kern = headline["kern"] if args.mode == "encaps" else None
    if enc_kern is None and extras:  # the encaps figures are wanted in any mode
        el, enc_kern = timer_kem.run(kem.encaps, max(5, min(args.steps, 20)), 2)
        v, el = parallel.whole_job_rate(ranks, B * max(5, min(args.steps, 20)), el)
        out_cfg["encaps"] = {"value": v, "unit": "encaps/s", "ms_per_step": el / max(5, min(args.steps, 20)) * 1e3}
    status_sum = int(kem.eng.status.sum().item())
    full = not args.sample_parity
    parity_enc = kem.parity_encaps((None if full else 1 << 16) if rank == 0 else 1 << 12)
    parity_fail = ranks.sum(0 if parity_enc["bit_exact_vs_oracle"] else 1)
    parity_enc["ranks_failing"] = int(parity_fail)
    per_rank = {"encaps_per_s": ranks.gather(B * args.steps / headline["elapsed"]) if args.mode == "encaps" else None}

    # ---- strong scaling: the metric read literally -- ONE batch of B items, n/G contiguous items per rank ----
    strong = None
    if args.mode == "encaps":
        lo, hi = parallel.shard_bounds(B, world, rank)
        if world == 1:
            strong = {"value": headline["value"], "ms_per_step": headline["elapsed"] / args.steps * 1e3, "items_per_rank": [B],
                      "note": "one rank: the strong and the weak figure are the same measurement"}
        else:
            ks = KemWork(768, hi - lo, rank, dev, window=(lo, B))
            el_s, _ = Timer(ranks, []).run(ks.encaps, args.steps, args.warmup)
            v_s, el_s = parallel.whole_job_rate(ranks, (hi - lo) * args.steps, el_s)
            ps = ks.parity_encaps(1 << 12)
            strong = {"value": v_s, "ms_per_step": el_s / args.steps * 1e3, "items_per_rank": [int(x) for x in ranks.gather(hi - lo)],
                      "per_rank_encaps_per_s": ranks.gather((hi - lo) * args.steps / el_s),
                      "parity": dict(ps, ranks_failing=int(ranks.sum(0 if ps["bit_exact_vs_oracle"] else 1)))}
            del ks
            torch.cuda.empty_cache()
        strong.update({"unit": "encaps/s", "scaling": "strong", "batch_total": B, "steps": args.steps, "warmup": args.warmup,
                       "workload": "ONE batch of %d ML-KEM-768 encapsulations (distinct-key), contiguous split n/G per rank, same barrier / "
                                   "max-over-ranks protocol as `value`" % B})

    # ---- the figure of rounds 1-4, for continuity: the same step with keys drawn from a pool of 2^16 (77 MB of ek: fits the 256 MB
    # Infinity Cache, which 1.24 GB of distinct keys does not) ----
    if more and args.mode == "encaps" and kem.pool > POOLED:
        kp = KemWork(768, B, rank, dev, pool_max=POOLED)
        el_p, kern_p = Timer(ranks, ["mlkem_hash", "mlkem_encrypt"]).run(kp.encaps, 10, 2)
        v_p, el_p = parallel.whole_job_rate(ranks, B * 10, el_p)
        pp = kp.parity_encaps(1 << 12)
        out_cfg["pooled"] = {"value": v_p, "unit": "encaps/s", "ms_per_step": el_p / 10 * 1e3, "key_pool": kp.pool,
                             "kernel_ms_per_step": {k: v["ms_per_step"] for k, v in kern_p.items()}, "parity": pp,
                             "note": "keys from a pool of %d cycled %dx (what `value` was in rounds 1-4); `value` now has one key per item" % (kp.pool, B // kp.pool)}
        del kp
        torch.cuda.empty_cache()

    # ---- decaps + config 3 (second half of Encaps + Decaps on the same items) ----
    if extras or args.mode == "config3":
        ksteps = args.steps if args.mode == "config3" else 10
        if args.mode != "config3":
            el, kd = timer_kem.run(kem.decaps, ksteps, 2)
            vdec, el = parallel.whole_job_rate(ranks, B * ksteps, el)
            dec_ms = el / ksteps * 1e3
        else:
            kd = headline["kern"]
            dec_ms = None
            vdec = None
        par_dec = kem.parity_decaps(1 << 16 if rank == 0 else 1 << 12)
        par_dec["ranks_failing"] = int(ranks.sum(0 if (par_dec["bit_exact_vs_oracle"] and par_dec["all_items_ss_dec_equals_ss_enc"]) else 1))
        dom_ms = kd["mlkem_encrypt"]["ms_per_step"]
        if args.mode == "config3":
            # the step holds encaps + decaps: split the encrypt-kernel time evenly is wrong, so report the whole step's kernels
            out_cfg["decaps"] = {"parity": par_dec, "kernel_ms_per_step": {k: v["ms_per_step"] for k, v in kd.items()}}
        else:
            out_cfg["decaps"] = {"value": vdec, "unit": "decaps/s", "ms_per_step": dec_ms, "n_per_gpu": B,
                                 "kernel_ms_per_step": {k: v["ms_per_step"] for k, v in kd.items()},
                                 "roofline": roofline("mlkem_encrypt_kernel<3, REENCRYPT> (dominant of decrypt / hash / re-encrypt)", B, BYTES["mlkem768_decaps"], dom_ms,
                                                      note="frac of the whole decapsulation (all three kernels): %.4f" %
                                                      (B * BYTES["mlkem768_decaps"] / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS)),
                                 "parity": par_dec}
            enc_ms = (headline["elapsed"] / args.steps * 1e3) if args.mode == "encaps" else out_cfg["encaps"]["ms_per_step"]
            out_cfg["config3"] = {"value": world * B / ((enc_ms + dec_ms) * 1e-3), "unit": "encaps+decaps pairs/s",
                                  "workload": "ML-KEM-768 Encaps + Decaps, %d items per GPU (BASELINE configs[2]: 2^23 over 8 GPUs = 2^20 per GPU), "
                                              "all ss_dec == ss_enc" % B,
                                  "ms_per_pair_step": enc_ms + dec_ms, "per_rank_decaps_per_s": ranks.gather(B * ksteps / (dec_ms * 1e-3 * ksteps))}

    # ---- shared key / key table (the reference's parsed-key cache) ----
    if more and args.mode == "encaps":
        ct_s, ss_s, st_s = torch.empty_like(kem.eng.ct), torch.empty_like(kem.eng.ss), torch.empty_like(kem.eng.status)
        tm = Timer(ranks, ["mlkem_hash", "mlkem_encrypt", "mlkem_keytable"])
        el, ks = tm.run(lambda: kem.eng.encaps_shared(kem.ek[:1], kem.m, ct_s, ss_s, st_s), 5, 1)
        out_cfg["shared_key"] = {"value": world * B * 5 / ranks.max(el), "unit": "encaps/s", "ms_per_step": ranks.max(el) / 5 * 1e3,
                                 "roofline": roofline("mlkem_encrypt_kernel<3, ENCAPS, shared>", B, BYTES["mlkem768_encaps_shared"], ks["mlkem_encrypt"]["ms_per_step"]),
                                 "note": "one ek for the whole batch (circl_hip_mlkem_encaps_shared): A^T and H(ek) amortised"}
        nkeys = min(1000, B)
        idx = torch.randint(0, nkeys, (B,), dtype=torch.int32, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
        table = kem.ek[:nkeys].contiguous()
        el, kk = tm.run(lambda: kem.eng.encaps_keyed(table, idx, kem.m, ct_s, ss_s, st_s), 5, 1)
        # bit-exact against the per-item API on the gathered keys
        kem.eng.encaps(table[idx.long()].contiguous(), kem.m)
        torch.cuda.synchronize()
        same = bool((ct_s == kem.eng.ct).all().item() and (ss_s == kem.eng.ss).all().item())
        kem.encaps()  # restore the headline outputs for the checks below
        torch.cuda.synchronize()
        out_cfg["keyed"] = {"value": world * B * 5 / ranks.max(el), "unit": "encaps/s", "ms_per_step": ranks.max(el) / 5 * 1e3, "table_keys": nkeys,
                            "keytable_ms_per_step": kk["mlkem_keytable"]["ms_per_step"], "equals_per_item_api_on_gathered_keys": same,
                            "note": "circl_hip_mlkem_encaps_keyed: 1000-key table + uniform random index per item; expansion once per table entry"}
        del ct_s, ss_s, st_s

    # ---- the hybrid KEMs that carry ML-KEM-768 (X-Wing, X25519MLKEM768), composed on the device ----
    if more and args.mode == "encaps":
        from circl_amd import device as cdev
        nh = max(B // 4, 64)
        gh = torch.Generator(device=dev).manual_seed(9000 + rank)
        hyb = {}
        for scheme, name in ((cdev.XWING, "xwing"), (cdev.X25519MLKEM768, "x25519mlkem768")):
            h = cdev.HybridDevice(scheme, nh, dev)
            seeds = torch.randint(0, 256, (nh, h.S["seed"]), dtype=torch.uint8, device=dev, generator=gh)
            es = torch.randint(0, 256, (nh, h.S["eseed"]), dtype=torch.uint8, device=dev, generator=gh)
            pk, sk = h.keygen(seeds)
            th = Timer(ranks, ["x25519", "mlkem_hash", "mlkem_encrypt"])
            el_e, ke = th.run(lambda: h.encaps(pk, es), 5, 1)
            el_d, _ = th.run(lambda: h.decaps(sk, h.ct), 5, 1)
            torch.cuda.synchronize()
            agree = bool((h.ss == h.ss2).all().item()) and not bool(h.status.any().item())
            from oracle import hybrid as ohyb
            ns = min(nh, 1 << 10 if rank == 0 else 1 << 6)
            pkc, esc, skc = pk[:ns].cpu().numpy(), es[:ns].cpu().numpy(), sk[:ns].cpu().numpy()
            ct0, ss0, st0 = (ohyb.xwing_encaps if scheme == cdev.XWING else ohyb.hybrid_encaps)(pkc, esc)
            exact = bool((h.ct[:ns].cpu().numpy() == ct0).all() and (h.ss[:ns].cpu().numpy() == ss0).all() and not st0.any())
            hyb[name] = {"encaps_per_s": world * nh * 5 / ranks.max(el_e), "decaps_per_s": world * nh * 5 / ranks.max(el_d), "n_per_gpu": nh,
                         "x25519_kernel_ms_per_encaps_step": ke["x25519"]["ms_per_step"],
                         "parity": {"sampled_items": ns, "bit_exact_vs_oracle": exact, "all_items_ss_dec_equals_ss_enc": agree}}
            del h
        out_cfg["hybrid"] = dict(hyb, note="X25519 on the GPU (one ladder / fixed-base comb per lane), ML-KEM-768, SHAKE256 / SHA3-256 glue: "
                                           "circl_hip_hybrid_*_dev on resident arrays; replaces kem/xwing and kem/hybrid's X25519MLKEM768")
        torch.cuda.empty_cache()

    # ---- host-buffer ABI, end to end ----
    host = None
    if (extras and args.mode == "encaps") or args.mode == "host":
        host = host_abi(kem, local)
        host["pageable"]["whole_job_value"] = ranks.sum(host["pageable"]["value"])
        host["pinned"]["whole_job_value"] = ranks.sum(host["pinned"]["value"])
        host["per_rank_pageable_per_s"] = ranks.gather(host["pageable"]["value"])
        out_cfg["host_abi"] = host

    # ---- config 4: ML-DSA-65 verify, 2^18 distinct keys ----
    if extras or args.mode == "config4":
        n4 = max(B // 4, 64)   # 2^18 at the default batch (BASELINE configs[3])
        d65 = DsaWork(65, n4, rank, dev)
        tm = Timer(ranks, ["mldsa_hash", "mldsa_verify"])
        k4 = args.steps if args.mode == "config4" else 5
        el, kv = tm.run(d65.verify, k4, args.warmup if args.mode == "config4" else 1)
        v4, el = parallel.whole_job_rate(ranks, n4 * k4, el)
        par4 = d65.parity((None if full else 1 << 16) if rank == 0 else 1 << 10, 1 << 11)
        par4["ranks_failing"] = int(ranks.sum(0 if (par4["bit_exact_vs_oracle"] and par4["all_items_as_expected"]) else 1))
        cfg4 = {"value": v4, "unit": "verifications/s", "ms_per_step": el / k4 * 1e3, "n_per_gpu": n4, "steps": k4,
                "workload": "ML-DSA-65 Verify, %d items per GPU, DISTINCT keys (GPU keygen + GPU deterministic signing at %.2e sig/s incl. setup), "
                            "32-byte messages, empty context, %.2f %% corrupted signatures" % (n4, n4 / d65.sign_s, 100.0 * d65.n_bad / n4),
                "kernel_ms_per_step": {k: v["ms_per_step"] for k, v in kv.items()},
                "roofline": roofline("mldsa_verify_kernel<65>", n4, BYTES["mldsa65_verify"], kv["mldsa_verify"]["ms_per_step"]),
                "parity": par4, "per_rank_per_s": ranks.gather(n4 * k4 / el)}
        if args.mode == "config4":
            headline = {"elapsed": el, "value": v4, "kern": kv}
        out_cfg["config4"] = cfg4
        del d65
        torch.cuda.empty_cache()

    # ---- config 5: ML-KEM-1024 encaps + ML-DSA-87 verify, concurrently on two streams ----
    if extras or args.mode == "config5":
        n5 = max(B // 16, 64)  # 2^16 at the default batch: the per-GPU share of BASELINE configs[4] (2^20 mixed items over 8 GPUs = 2^16 + 2^16 per GPU)
        k1024 = KemWork(1024, n5, rank, dev, shake_inputs=False)
        d87 = DsaWork(87, n5, rank, dev)
        s_kem, s_dsa = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

        def mixed():
            with torch.cuda.stream(s_kem):
                k1024.encaps()
            with torch.cuda.stream(s_dsa):
                d87.verify()
        tm = Timer(ranks, ["mlkem_hash", "mlkem_encrypt", "mldsa_hash", "mldsa_verify"])
        k5 = args.steps if args.mode == "config5" else 10
        el, km = tm.run(mixed, k5, args.warmup if args.mode == "config5" else 2)
        v5, el = parallel.whole_job_rate(ranks, 2 * n5 * k5, el)
        tk = Timer(ranks, ["mlkem_hash", "mlkem_encrypt"])
        el_k, kk = tk.run(k1024.encaps, 10, 1)
        td = Timer(ranks, ["mldsa_hash", "mldsa_verify"])
        el_d, kd87 = td.run(d87.verify, 10, 1)
        pk5 = k1024.parity_encaps(1 << 14 if rank == 0 else 1 << 10)
        pd5 = d87.parity(1 << 14 if rank == 0 else 1 << 9, 1 << 9)
        fails = ranks.sum(0 if (pk5["bit_exact_vs_oracle"] and pd5["bit_exact_vs_oracle"] and pd5["all_items_as_expected"]) else 1)
        cfg5 = {"value": v5, "unit": "mixed items/s", "ms_per_step": el / k5 * 1e3, "n_per_gpu": [n5, n5], "steps": k5,
                "workload": "ML-KEM-1024 Encapsulate (%d, distinct keys) + ML-DSA-87 Verify (%d, distinct keys, %.2f %% corrupted) submitted on two streams at once" %
                            (n5, n5, 100.0 * d87.n_bad / n5),
                "alone": {"mlkem1024_encaps_per_s": n5 * 10 / el_k, "mldsa87_verify_per_s": n5 * 10 / el_d,
                          "serial_sum_ms": (el_k + el_d) / 10 * 1e3},
                "kernel_ms_per_step": {k: v["ms_per_step"] for k, v in km.items()},
                "roofline_mlkem1024": roofline("mlkem_encrypt_kernel<4>", n5, BYTES["mlkem1024_encaps"], kk["mlkem_encrypt"]["ms_per_step"]),
                "roofline_mldsa87": roofline("mldsa_verify_kernel<87>", n5, BYTES["mldsa87_verify"], kd87["mldsa_verify"]["ms_per_step"]),
                "parity": {"mlkem1024": pk5, "mldsa87": pd5, "ranks_failing": int(fails)}, "per_rank_per_s": ranks.gather(2 * n5 * k5 / el)}
        if args.mode == "config5":
            headline = {"elapsed": el, "value": v5, "kern": km}
        out_cfg["config5"] = cfg5
        del k1024, d87
        torch.cuda.empty_cache()

    if args.mode == "host":
        headline = {"elapsed": B / host["pageable"]["value"] * args.steps, "value": host["pageable"]["whole_job_value"], "kern": {}}

    if more and rank == 0 and world == 1:
        out_cfg["small_batches"] = small_batches(dev)
        cc = concurrent_callers()
        if cc:
            out_cfg["concurrent_callers"] = cc

    # ---- PMC: traffic / VALU instructions of the dominant kernels ----
    traffic, valu, pmc_note = None, None, None
    if rank == 0 and enc_kern is not None:
        enc_avg_ms = enc_kern["mlkem_encrypt"]["ms_per_step"]
        probe = live_valu_probe(local)
        live_all, why = (None, "disabled (--no-pmc)") if (args.no_pmc or world > 1) else pmc_live(B)
        live = live_all.get("mlkem768_encrypt") if live_all else None
        committed = pmc_committed()
        sha = lib_sha256()
        pmc_write(live_all, sha, B)
        other = live_all if live_all else ((committed or {}).get("kernels") if (committed or {}).get("lib_sha256") == sha else None)
        if other:  # the other configs' dominant kernels: traffic and VALU fraction next to their HBM rooflines
            live_all_or_committed = other
            for cfg_key, roof_key, pk in (("config4", "roofline", "mldsa65_verify"), ("config5", "roofline_mlkem1024", "mlkem1024_encrypt"),
                                          ("config5", "roofline_mldsa87", "mldsa87_verify")):
                r = (out_cfg.get(cfg_key) or {}).get(roof_key)
                if r and live_all_or_committed.get(pk):
                    r["traffic"] = live_all_or_committed[pk]["bytes"]
                    r["traffic_over_algorithmic"] = r["traffic"] / r["algorithmic_bytes_per_launch"]
                    r["valu"] = valu_issue(live_all_or_committed[pk]["valu_insts"], r["avg_launch_ms"], pk,
                                           {"mldsa65_verify": max(B // 4, 64)}.get(pk, max(B // 16, 64)), probe)
                    r["pmc_source"] = "live" if live_all else "profiles/traffic.json + valu.json (this very build)"
        if live:
            traffic, valu = live["bytes"], valu_issue(live["valu_insts"], enc_avg_ms, "mlkem768_encrypt", B, probe)
            pmc_note = {"source": "live", "method": live_all["method"], "read_bytes": live["read_bytes"], "write_bytes": live["write_bytes"]}
            if committed and committed.get("bytes"):
                dev_pct = abs(committed["bytes"] - traffic) / traffic * 100.0
                pmc_note["committed_profiles_traffic_json_deviates_pct"] = dev_pct
                if dev_pct > 5.0:
                    print("bench.py: WARNING profiles/traffic.json is STALE: %.3e B committed vs %.3e B measured now (%.1f %%)" %
                          (committed["bytes"], traffic, dev_pct), file=sys.stderr)
        elif committed and committed.get("lib_sha256") == sha:
            traffic, valu = committed["bytes"], valu_issue(committed["valu_insts"], enc_avg_ms, "mlkem768_encrypt", B, probe)
            pmc_note = {"source": "profiles/traffic.json + valu.json (taken from this very build of libcirclhip.so)", "live_pass": why}
        else:
            pmc_note = {"source": None, "live_pass": why,
                        "committed": "profiles/traffic.json was measured on a different build of libcirclhip.so: not reported"}
            print("bench.py: WARNING no valid PMC figures for this build (%s)" % why, file=sys.stderr)

    if rank == 0:
        K = args.steps
        metric = {"encaps": "ML-KEM-768 encapsulations/sec (whole node), batch=2^20",
                  "config3": "ML-KEM-768 encaps+decaps pairs/sec (whole node), 2^20 per GPU",
                  "config4": "ML-DSA-65 verifications/sec (whole node), batch=2^18 per GPU",
                  "config5": "ML-KEM-1024 encaps + ML-DSA-87 verify mixed items/sec (whole node), 2^16+2^16 per GPU",
                  "host": "ML-KEM-768 encapsulations/sec through the host-buffer C ABI (pageable memory, PCIe-inclusive)"}[args.mode]
        unit = {"encaps": "encaps/s", "config3": "pairs/s", "config4": "verifications/s", "config5": "items/s", "host": "encaps/s"}[args.mode]
        out = {
            "metric": metric, "value": headline["value"], "unit": unit, "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": headline["elapsed"] / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": {"encaps": "ML-KEM-768 Encapsulate, distinct-key (one GPU-generated key per item), batch=%d per GPU, inputs resident in HBM "
                                              "(SURVEY 8d scope i; the PCIe-inclusive host-ABI rate, scope ii, is `value_host_abi`)" % B,
                                    "config3": "ML-KEM-768 Encapsulate + Decapsulate, distinct-key, batch=%d per GPU, inputs resident in HBM" % B,
                                    "config4": "ML-DSA-65 Verify, distinct-key, batch=%d per GPU, inputs resident in HBM" % max(B // 4, 64),
                                    "config5": "ML-KEM-1024 Encapsulate + ML-DSA-87 Verify on two streams, %d + %d per GPU, inputs resident in HBM" % (max(B // 16, 64), max(B // 16, 64)),
                                    "host": "ML-KEM-768 Encapsulate through circl_hip_mlkem_encaps (host pointers, pageable), batch=%d per GPU" % B}[args.mode],
                       "key_pool": kem.pool, "parallelism": "batch split per device, no collectives", "mode": args.mode},
            "value_is": "weak: every rank times its own batch of %d (per-GPU work fixed as N grows); `strong` = one batch of %d split over the ranks; "
                        "`value_host_abi` = the same metric through the host-pointer C ABI (PCIe-inclusive, pageable caller memory)" % (B, B),
            "strong": strong,
            "value_host_abi": ({"value": host["pageable"]["whole_job_value"], "pinned": host["pinned"]["whole_job_value"], "unit": "encaps/s",
                                "what": "circl_hip_mlkem_encaps with host pointers (SURVEY 8d scope ii): H2D + kernels + D2H, pageable / page-locked caller buffers"}
                               if host else None),
            "sustained": sustained,
            "parity": parity_enc,
            "per_rank": per_rank,
            "configs": out_cfg,
            "bench_wall_s": time.perf_counter() - t_start,
        }
        if enc_kern is not None:
            enc_avg_ms = enc_kern["mlkem_encrypt"]["ms_per_step"]
            r = roofline("mlkem_encrypt_kernel<3>", B, BYTES["mlkem768_encaps"], enc_avg_ms, traffic=traffic,
                         note="integer-VALU bound (Keccak-f[1600] as 2x u32 bit ops, Z_3329 arithmetic on V_MUL_LO/HI_U32), not HBM bound; "
                              "traffic is L2<->fabric bytes incl. the Infinity-Cache-resident matrix scratch: see DESIGN.md 4.4/5")
            r["launches"] = enc_kern["mlkem_encrypt"]["launch_groups"]
            r["hash_kernel_avg_ms"] = enc_kern["mlkem_hash"]["ms_per_step"]
            r["valu"] = valu
            if traffic:
                r["traffic_over_algorithmic"] = traffic / r["algorithmic_bytes_per_launch"]
            r["pmc"] = pmc_note
            r["status_nonzero"] = status_sum
            out["roofline"] = r
        else:
            out["roofline"] = (out_cfg.get(args.mode) or {}).get("roofline")
        if world == 1 and not args.no_cpu_baseline:
            # the reference itself when this box can run it (Go toolchain + $CIRCL_REFERENCE), the C oracle otherwise; with the
            # reference, the oracle's figure stays in the line as `port` for comparison across boxes
            refb = cpu_baseline_reference(kem)
            port = cpu_baseline(kem, budget_s=10.0 if refb is None else 4.0)
            out["cpu_baseline"] = dict(refb, port=port) if refb else port
        line = emit(out, args.extras_file)
        print(line, flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
