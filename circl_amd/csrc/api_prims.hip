// api_prims.hip -- the unit-level primitives (Keccak-f, ring transforms, sponges) and the batched XOF / KangarooTwelve
// service of the C ABI (include/circl_hip.h).  No CPU compute path: the KangarooTwelve host code only lays out the tree's
// nodes; every permutation runs on the device.
#include "host_common.h"
#include "prim_kernels.h"
#include "sampler_prims.h"

using namespace circl::host;

namespace {
PipeOpts prim_opts(size_t row_bytes) {
    PipeOpts o;
    o.chunk_items = host_chunk_items(std::max<size_t>(size_t(1) << 10, (size_t(32) << 20) / std::max<size_t>(row_bytes, 1)));  // ~32 MB per chunk
    return o;
}
const std::function<size_t(size_t)> no_ws = [](size_t) { return size_t(0); };
}  // namespace

extern "C" {

int circl_hip_keccak_f1600(uint64_t *states, size_t n, int rounds, int device) {
    if (rounds != 24 && rounds != 12) return CIRCL_HIP_EPARAM;
    uint8_t *p = reinterpret_cast<uint8_t *>(states);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{p + lo * 200, 200}}, {}, {{p + lo * 200, 200}}, no_ws, prim_opts(200), [&](Chunk &c) {
            HIP_TRY(hipMemcpyAsync(c.out[0], c.in[0], c.cnt * 200, hipMemcpyDeviceToDevice, c.st));
            hipLaunchKernelGGL(circl::prim::keccak_f1600_kernel, dim3((unsigned)((c.cnt + 255) / 256)), dim3(256), 0, c.st,
                               reinterpret_cast<uint64_t *>(c.out[0]), c.cnt, 24 - rounds);
            HIP_TRY(hipGetLastError());
            return CIRCL_HIP_OK;
        });
    });
}

/* Live issue-rate probe (bench.py prices the kernels' VALU time against it): wave-instructions per second PER SIMD sustained by
 * (a) the library's Keccak-f[1600] round and (b) two-operand integer VALU ops, on every SIMD of `device` with `waves_per_simd`
 * wavefronts resident each.  HIP-event timed; ~10 ms of GPU time. */
int circl_hip_profile_valu_probe(int device, int waves_per_simd, double *keccak_insts_per_s_per_simd, double *simple_insts_per_s_per_simd) {
    if (device < 0 || device >= ndev()) return CIRCL_HIP_ENODEV;
    if (waves_per_simd < 1 || waves_per_simd > 8) return CIRCL_HIP_EPARAM;
    HIP_TRY(hipSetDevice(physical_device(device)));
    const int cus = dev_info(device).cus;
    uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sink), 256));
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st = nullptr;
    auto cleanup = [&] {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        if (st) (void)hipStreamDestroy(st);
        (void)hipFree(sink);
    };
    auto timed = [&](auto kern, int iters, double insts_per_wave, double *out) -> int {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {  // the first repetition also warms the clocks
            HIP_TRY(hipEventRecord(a, st));
            hipLaunchKernelGGL(kern, dim3((unsigned)(cus * waves_per_simd)), dim3(256), 0, st, sink, iters);
            HIP_TRY(hipEventRecord(b, st));
            HIP_TRY(hipEventSynchronize(b));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, a, b));
            if (rep > 0) best = std::min(best, ms);
        }
        if (out) *out = insts_per_wave * waves_per_simd / (best * 1e-3);  // every SIMD ran waves_per_simd wavefronts side by side
        return CIRCL_HIP_OK;
    };
    int rc = CIRCL_HIP_OK;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        rc = CIRCL_HIP_EHIP;
    }
    constexpr int kKeccakIters = 512, kSimpleIters = 8192;
    if (rc == CIRCL_HIP_OK) rc = timed(circl::prim::keccak_rate_probe_kernel, kKeccakIters, 4320.0 * kKeccakIters, keccak_insts_per_s_per_simd);
    if (rc == CIRCL_HIP_OK) rc = timed(circl::prim::simple_rate_probe_kernel, kSimpleIters, (double)circl::prim::kSimpleProbePerIter * kSimpleIters, simple_insts_per_s_per_simd);
    cleanup();
    return rc;
}

int circl_hip_keccak_f1600_coop(uint64_t *states, size_t n, int device) {
    uint8_t *p = reinterpret_cast<uint8_t *>(states);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{p + lo * 200, 200}}, {}, {{p + lo * 200, 200}}, no_ws, prim_opts(200), [&](Chunk &c) {
            HIP_TRY(hipMemcpyAsync(c.out[0], c.in[0], c.cnt * 200, hipMemcpyDeviceToDevice, c.st));
            hipLaunchKernelGGL(circl::prim::keccak_f1600_coop_kernel, dim3((unsigned)c.cnt), dim3(64), 0, c.st, reinterpret_cast<uint64_t *>(c.out[0]));
            HIP_TRY(hipGetLastError());
            return CIRCL_HIP_OK;
        });
    });
}

int circl_hip_keccak_f1600_split(uint64_t *states, size_t n, int device) {
    uint8_t *p = reinterpret_cast<uint8_t *>(states);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{p + lo * 200, 200}}, {}, {{p + lo * 200, 200}}, no_ws, prim_opts(200), [&](Chunk &c) {
            HIP_TRY(hipMemcpyAsync(c.out[0], c.in[0], c.cnt * 200, hipMemcpyDeviceToDevice, c.st));
            hipLaunchKernelGGL(circl::prim::keccak_f1600_split_kernel, dim3((unsigned)((2 * c.cnt + 255) / 256)), dim3(256), 0, c.st,
                               reinterpret_cast<uint64_t *>(c.out[0]), c.cnt);
            HIP_TRY(hipGetLastError());
            return CIRCL_HIP_OK;
        });
    });
}

int circl_hip_kyber_ntt(int16_t *polys, size_t n, int inverse, int device) {
    uint8_t *p = reinterpret_cast<uint8_t *>(polys);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{p + lo * 512, 512}}, {}, {{p + lo * 512, 512}}, no_ws, prim_opts(512), [&](Chunk &c) {
            HIP_TRY(hipMemcpyAsync(c.out[0], c.in[0], c.cnt * 512, hipMemcpyDeviceToDevice, c.st));
            hipLaunchKernelGGL(circl::prim::kyber_ntt_kernel, dim3((unsigned)c.cnt), dim3(64), 0, c.st, reinterpret_cast<int16_t *>(c.out[0]), inverse);
            HIP_TRY(hipGetLastError());
            return CIRCL_HIP_OK;
        });
    });
}

int circl_hip_dilithium_ntt(uint32_t *polys, size_t n, int inverse, int device) {
    uint8_t *p = reinterpret_cast<uint8_t *>(polys);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{p + lo * 1024, 1024}}, {}, {{p + lo * 1024, 1024}}, no_ws, prim_opts(1024), [&](Chunk &c) {
            HIP_TRY(hipMemcpyAsync(c.out[0], c.in[0], c.cnt * 1024, hipMemcpyDeviceToDevice, c.st));
            hipLaunchKernelGGL(circl::prim::dilithium_ntt_kernel, dim3((unsigned)c.cnt), dim3(64), 0, c.st, reinterpret_cast<uint32_t *>(c.out[0]), inverse);
            HIP_TRY(hipGetLastError());
            return CIRCL_HIP_OK;
        });
    });
}

int circl_hip_kyber_mulhat(int16_t *out, const int16_t *a, const int16_t *b, size_t n, int device) {
    uint8_t *po = reinterpret_cast<uint8_t *>(out);
    const uint8_t *pa = reinterpret_cast<const uint8_t *>(a), *pb = reinterpret_cast<const uint8_t *>(b);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{pa + lo * 512, 512}, {pb + lo * 512, 512}}, {}, {{po + lo * 512, 512}}, no_ws, prim_opts(512), [&](Chunk &c) {
            hipLaunchKernelGGL(circl::prim::kyber_mulhat_kernel, dim3((unsigned)c.cnt), dim3(64), 0, c.st, reinterpret_cast<int16_t *>(c.out[0]),
                               reinterpret_cast<const int16_t *>(c.in[0]), reinterpret_cast<const int16_t *>(c.in[1]));
            HIP_TRY(hipGetLastError());
            return CIRCL_HIP_OK;
        });
    });
}

int circl_hip_shake(int rate, int ds, const uint8_t *in, size_t inlen, uint8_t *out, size_t outlen, size_t n, int device) {
    if ((rate != 168 && rate != 136 && rate != 72) || (ds != 0x1f && ds != 0x06) || outlen == 0) return CIRCL_HIP_EPARAM;
    const size_t il = inlen ? inlen : 1;  // keep the per-item stride non-zero for empty messages
    std::vector<uint8_t> pad;
    const uint8_t *src = in;
    if (!inlen) { pad.assign(n, 0); src = pad.data(); }
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{src + lo * il, il}}, {}, {{out + lo * outlen, outlen}}, no_ws, prim_opts(il + outlen), [&](Chunk &c) {
            // the kernel strides inputs by `inlen`; empty inputs never dereference
            hipLaunchKernelGGL(circl::prim::sponge_kernel, dim3((unsigned)((c.cnt + 255) / 256)), dim3(256), 0, c.st, rate / 8, (uint32_t)ds, 0,
                               (const uint8_t *)c.in[0], inlen, (const uint64_t *)nullptr, c.out[0], outlen, c.cnt);
            HIP_TRY(hipGetLastError());
            return CIRCL_HIP_OK;
        });
    });
}

// ---- lane-local arithmetic and the samplers as unit-level primitives (parity tests only; see prim_kernels.h / sampler_prims.h) ----
int circl_hip_lane_op(int op, int arg, const uint32_t *a, const uint32_t *b, uint32_t *out0, uint32_t *out1, size_t n, int device) {
    if (op < 1 || op >= CIRCL_HIP_LANE_OP_COUNT || !a || !out0) return CIRCL_HIP_EPARAM;
    const uint8_t *pa = reinterpret_cast<const uint8_t *>(a), *pb = reinterpret_cast<const uint8_t *>(b);
    uint8_t *p0 = reinterpret_cast<uint8_t *>(out0), *p1 = reinterpret_cast<uint8_t *>(out1);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        std::vector<HIn> ins = {{pa + lo * 4, 4}};
        if (pb) ins.push_back({pb + lo * 4, 4});
        return run_pipeline(dev, cnt, ins, {}, {{p0 + lo * 4, 4}, {p1 ? p1 + lo * 4 : nullptr, 4}}, no_ws, prim_opts(16), [&](Chunk &c) {
            hipLaunchKernelGGL(circl::prim::lane_op_kernel, dim3((unsigned)((c.cnt + 255) / 256)), dim3(256), 0, c.st, op, arg,
                               reinterpret_cast<const uint32_t *>(c.in[0]), pb ? reinterpret_cast<const uint32_t *>(c.in[1]) : nullptr,
                               reinterpret_cast<uint32_t *>(c.out[0]), p1 ? reinterpret_cast<uint32_t *>(c.out[1]) : nullptr, c.cnt);
            HIP_TRY(hipGetLastError());
            return CIRCL_HIP_OK;
        });
    });
}

int circl_hip_kyber_sample_uniform(const uint8_t *seed32, const uint8_t *xy, int16_t *polys, size_t n, int device) {
    uint8_t *po = reinterpret_cast<uint8_t *>(polys);
    PipeOpts o = prim_opts(512);
    o.chunk_items = (o.chunk_items + 63) & ~size_t(63);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{seed32 + lo * 32, 32}, {xy + lo * 2, 2}}, {}, {{po + lo * 512, 512}}, no_ws, o, [&](Chunk &c) {
            hipLaunchKernelGGL(circl::prim::kyber_uniform_prim_kernel, dim3((unsigned)((c.cnt + 63) / 64)), dim3(64), circl::mlkem::Geom<3>::LDS_FIFO, c.st,
                               (const uint8_t *)c.in[0], (const uint8_t *)c.in[1], reinterpret_cast<int16_t *>(c.out[0]), c.cnt);
            HIP_TRY(hipGetLastError());
            return CIRCL_HIP_OK;
        });
    });
}

int circl_hip_kyber_sample_cbd(int eta, const uint8_t *seed32, int16_t *polys, size_t n, int device) {
    if (eta != 2 && eta != 3) return CIRCL_HIP_EPARAM;
    constexpr size_t ROW = 64 * 512;  // 64 nonces x 256 int16
    uint8_t *po = reinterpret_cast<uint8_t *>(polys);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{seed32 + lo * 32, 32}}, {}, {{po + lo * ROW, ROW}}, no_ws, prim_opts(ROW), [&](Chunk &c) {
            if (eta == 2)
                hipLaunchKernelGGL(circl::prim::kyber_cbd_prim_kernel<3>, dim3((unsigned)c.cnt), dim3(64), 0, c.st, (const uint8_t *)c.in[0],
                                   reinterpret_cast<int16_t *>(c.out[0]), c.cnt);
            else
                hipLaunchKernelGGL(circl::prim::kyber_cbd_prim_kernel<2>, dim3((unsigned)c.cnt), dim3(64), 0, c.st, (const uint8_t *)c.in[0],
                                   reinterpret_cast<int16_t *>(c.out[0]), c.cnt);
            HIP_TRY(hipGetLastError());
            return CIRCL_HIP_OK;
        });
    });
}

int circl_hip_mldsa_sample_uniform(const uint8_t *seed32, const uint16_t *nonce, uint32_t *polys, size_t n, int device) {
    uint8_t *po = reinterpret_cast<uint8_t *>(polys);
    const uint8_t *pn = reinterpret_cast<const uint8_t *>(nonce);
    PipeOpts o = prim_opts(1024);
    o.chunk_items = (o.chunk_items + 63) & ~size_t(63);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {{seed32 + lo * 32, 32}, {pn + lo * 2, 2}}, {}, {{po + lo * 1024, 1024}},
                            [](size_t k) { return ((k + 63) & ~size_t(63)) * circl::mldsa::kPackedRowDwords * 4; }, o, [&](Chunk &c) {
                                hipLaunchKernelGGL(circl::prim::mldsa_uniform_prim_kernel, dim3((unsigned)((c.cnt + 63) / 64)), dim3(64),
                                                   circl::mldsa::DG<44>::LDS_FIFO, c.st, (const uint8_t *)c.in[0], reinterpret_cast<const uint16_t *>(c.in[1]),
                                                   reinterpret_cast<uint32_t *>(c.ws), reinterpret_cast<uint32_t *>(c.out[0]), c.cnt);
                                HIP_TRY(hipGetLastError());
                                return CIRCL_HIP_OK;
                            });
    });
}

// Batched XOF service: n sponges over variable-length messages, 24 or 12 rounds.
int circl_hip_xof(int rate, int ds, int rounds, const uint8_t *in_blob, const uint64_t *in_off, uint8_t *out, size_t outlen, size_t n,
                  int device) {
    if ((rate != 168 && rate != 136 && rate != 72 && rate != 104 && rate != 144) || ds < 1 || ds > 0x7f || (rounds != 24 && rounds != 12) ||
        outlen == 0)
        return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    static const uint8_t dummy[16] = {0};
    if (!in_blob) in_blob = dummy;  // all messages empty
    PipeOpts o;
    // messages are ragged: bound a chunk by its item count only (blob bytes of a chunk are whatever its offsets span)
    o.chunk_items = host_chunk_items(size_t(1) << 16);
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        return run_pipeline(dev, cnt, {}, {{in_blob, in_off + lo}}, {{out + lo * outlen, outlen}}, no_ws, o, [&](Chunk &c) {
            hipLaunchKernelGGL(circl::prim::sponge_kernel, dim3((unsigned)((c.cnt + 255) / 256)), dim3(256), 0, c.st, rate / 8, (uint32_t)ds, 24 - rounds,
                               c.blob[0], (size_t)0, c.off[0], c.out[0], outlen, c.cnt);
            HIP_TRY(hipGetLastError());
            return CIRCL_HIP_OK;
        });
    });
}

}  // extern "C"

// KangarooTwelve draft -10 (xof/k12/k12.go), n independent computations.  Tree hashing maps onto the batched sponge service
// as two or three TurboSHAKE128 batches: every 8192-byte leaf of every long message (D = 0x0B, k12.go:136-160), then the
// final nodes of the long messages (D = 0x06, :141-142, :383-395) and the short messages (|M| + |C| + |length_encode(|C|)|
// <= 8192, D = 0x07, :60-66).
// The message bytes are not re-arranged on the host: the whole message blob goes to the device once and the leaves are
// hashed in place as (offset, length) ranges of it -- with an empty context the trailing length_encode(0) byte is a
// suffix the kernel appends, so short messages and last leaves are hashed in place too; only what does not lie inside a
// message as it stands -- ranges that run into a non-empty C || length_encode(|C|), and the long messages' final nodes
// (first chunk || chaining values) -- is assembled into a small side blob (at most ~8 KB + 32 B per leaf per message).
namespace {
void k12_length_encode(std::vector<uint8_t> &v, uint64_t x) {  // k12.go:333-342
    uint8_t be[8];
    int nz = 0;
    for (int i = 0; i < 8; i++) be[i] = (uint8_t)(x >> (56 - 8 * i));
    while (nz < 8 && be[nz] == 0) nz++;
    v.insert(v.end(), be + nz, be + 8);
    v.push_back((uint8_t)(8 - nz));
}

// TurboSHAKE128 (12 rounds) over `cnt` ranges (off, len) of the device buffer `d_buf` -> outlen bytes each at d_out
int k12_ranges(hipStream_t st, const uint8_t *d_buf, const uint64_t *d_off, const uint64_t *d_len, uint32_t ds, uint8_t *d_out, size_t outlen, size_t cnt) {
    if (cnt == 0) return CIRCL_HIP_OK;
    // suffix = 0x00 = length_encode(0): ranges flagged with one suffix byte end a message whose context is empty
    hipLaunchKernelGGL(circl::prim::sponge_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, 168 / 8, ds, 12, d_buf, (size_t)0, d_off, d_out,
                       outlen, cnt, d_len, (uint64_t)0);
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}
}  // namespace

extern "C" int circl_hip_k12(const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, uint8_t *out,
                             size_t outlen, size_t n, int device) {
    constexpr size_t CHUNK = 8192;
    if (outlen == 0) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    if (ndev() <= 0) return CIRCL_HIP_ENODEV;
    const int dev = device < 0 ? 0 : device;  // one tree per message: the batch is not split across devices
    if (dev >= ndev()) return CIRCL_HIP_ENODEV;
    HIP_TRY(hipSetDevice(physical_device(dev)));
    const size_t base = (size_t)msg_off[0], msg_bytes = (size_t)(msg_off[n] - msg_off[0]);
    // ---- pass 0 (host): where every leaf lives ----
    // device buffer = [message blob (msg_bytes)] [side blob]; offsets below are relative to its start
    std::vector<uint8_t> side, tail;
    std::vector<uint64_t> leaf_off, leaf_len;       // every leaf of every long message, in message order
    std::vector<size_t> nleaves(n, 0);
    std::vector<uint64_t> head_off(n), head_len(n); // S_0 of each message (whole S for short ones)
    auto side_at = [&]() { return (uint64_t)(msg_bytes + 16 + side.size()); };  // (16 bytes of slack between the two parts)
    for (size_t i = 0; i < n; i++) {
        const size_t mo = (size_t)msg_off[i] - base, ml = (size_t)(msg_off[i + 1] - msg_off[i]);
        const uint8_t *m = msg_blob + msg_off[i];
        const uint8_t *c = ctx_blob ? ctx_blob + ctx_off[i] : nullptr;
        const size_t cl = ctx_blob ? (size_t)(ctx_off[i + 1] - ctx_off[i]) : 0;
        tail.clear();
        if (cl) tail.insert(tail.end(), c, c + cl);
        k12_length_encode(tail, cl);
        const size_t total = ml + tail.size();
        // S[lo, hi): a range of the message as it lies in the blob, or assembled into the side blob when it reaches the tail
        auto place = [&](size_t lo, size_t hi, uint64_t &o, uint64_t &l) {
            l = hi - lo;
            if (hi <= ml) { o = mo + lo; return; }
            if (cl == 0 && lo <= ml) {  // the tail is the one byte length_encode(0) = 00: hashed in place with a suffix byte
                o = mo + lo;
                l = (uint64_t)(ml - lo) | (uint64_t(1) << 56);
                return;
            }
            while (side.size() & 7) side.push_back(0);  // 8-byte aligned ranges: the sponge kernel's fast path
            o = side_at();
            if (lo < ml) side.insert(side.end(), m + lo, m + ml);
            side.insert(side.end(), tail.begin() + (std::max(lo, ml) - ml), tail.begin() + (hi - ml));
        };
        place(0, std::min(total, CHUNK), head_off[i], head_len[i]);
        for (size_t lo = CHUNK; lo < total; lo += CHUNK) {
            uint64_t o, l;
            place(lo, std::min(total, lo + CHUNK), o, l);
            leaf_off.push_back(o);
            leaf_len.push_back(l);
            nleaves[i]++;
        }
    }
    const size_t total_leaves = leaf_off.size();
    size_t n_long = 0;
    for (size_t i = 0; i < n; i++) n_long += nleaves[i] != 0;
    // ---- device staging: one slot holds everything ----
    const size_t side0 = msg_bytes + 16;
    const size_t final_bytes = n_long * (CHUNK + 64) + 32 * total_leaves + 64;  // upper bound of the long messages' final nodes
    const size_t o_side = side0, o_final = up256(o_side + side.size() + 16), o_desc = up256(o_final + final_bytes + 16);
    const size_t desc_cnt = std::max(total_leaves, n);
    const size_t o_out = up256(o_desc + 2 * 8 * desc_cnt), o_cv = up256(o_out + n * outlen), need = up256(o_cv + 32 * total_leaves + 16);
    Slot *slot = slot_acquire(dev);
    if (!slot) return CIRCL_HIP_EHIP;
    struct Release {
        Slot *s; hipStream_t st;
        ~Release() { if (st) (void)hipStreamSynchronize(st); slot_release(s); }
    } rel{slot, nullptr};
    if (int rc = slot->ensure(need, 0, 0)) return rc;
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t st; ~StreamGuard() { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } } sg{st};
    uint8_t *d = slot->d;
    if (msg_bytes) HIP_TRY(hipMemcpyAsync(d, msg_blob + base, msg_bytes, hipMemcpyHostToDevice, st));
    if (!side.empty()) HIP_TRY(hipMemcpyAsync(d + o_side, side.data(), side.size(), hipMemcpyHostToDevice, st));
    uint64_t *d_off = reinterpret_cast<uint64_t *>(d + o_desc), *d_len = d_off + desc_cnt;
    // ---- pass 1: chaining values of all leaves ----
    std::vector<uint8_t> cv(32 * total_leaves);
    if (total_leaves) {
        HIP_TRY(hipMemcpyAsync(d_off, leaf_off.data(), 8 * total_leaves, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_len, leaf_len.data(), 8 * total_leaves, hipMemcpyHostToDevice, st));
        if (int rc = k12_ranges(st, d, d_off, d_len, 0x0B, d + o_cv, 32, total_leaves)) return rc;
        HIP_TRY(hipMemcpyAsync(cv.data(), d + o_cv, cv.size(), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    // ---- pass 2: final nodes.  Short messages hash their S as it lies (D = 0x07); long ones S_0 || 03 0^7 || CVs ||
    // length_encode(leaves) || FF FF (D = 0x06), assembled on the host around the first chunk ----
    std::vector<uint64_t> f_off, f_len;
    std::vector<size_t> f_idx;
    std::vector<uint8_t> res(n * outlen);
    for (int pass = 0; pass < 2; pass++) {
        std::vector<uint8_t> fin;
        f_off.clear(); f_len.clear(); f_idx.clear();
        size_t cvpos = 0;
        for (size_t i = 0; i < n; i++) {
            const bool is_long = nleaves[i] != 0;
            if (is_long == (pass == 0)) {
                if (!is_long) {
                    f_off.push_back(head_off[i]);
                    f_len.push_back(head_len[i]);
                } else {
                    while (fin.size() & 7) fin.push_back(0);
                    f_off.push_back(o_final + fin.size());
                    const size_t start = fin.size();
                    const uint8_t *m = msg_blob + msg_off[i];  // a long message's first chunk lies wholly inside M or reaches the tail
                    const size_t ml = (size_t)(msg_off[i + 1] - msg_off[i]);
                    if (ml >= CHUNK) fin.insert(fin.end(), m, m + CHUNK);
                    else {  // |M| < 8192 < |S|: rebuild S_0 from M and the tail
                        tail.clear();
                        const size_t cl = ctx_blob ? (size_t)(ctx_off[i + 1] - ctx_off[i]) : 0;
                        if (cl) tail.insert(tail.end(), ctx_blob + ctx_off[i], ctx_blob + ctx_off[i] + cl);
                        k12_length_encode(tail, cl);
                        fin.insert(fin.end(), m, m + ml);
                        fin.insert(fin.end(), tail.begin(), tail.begin() + (CHUNK - ml));
                    }
                    static const uint8_t sep[8] = {3, 0, 0, 0, 0, 0, 0, 0};
                    fin.insert(fin.end(), sep, sep + 8);
                    fin.insert(fin.end(), cv.begin() + 32 * cvpos, cv.begin() + 32 * (cvpos + nleaves[i]));
                    k12_length_encode(fin, nleaves[i]);
                    fin.push_back(0xff);
                    fin.push_back(0xff);
                    f_len.push_back(fin.size() - start);
                }
                f_idx.push_back(i);
            }
            cvpos += nleaves[i];
        }
        if (f_idx.empty()) continue;
        const size_t cnt = f_idx.size();
        if (!fin.empty()) {
            if (fin.size() > final_bytes) { g_err = "k12: final-node staging overflow"; return CIRCL_HIP_EHIP; }
            HIP_TRY(hipMemcpyAsync(d + o_final, fin.data(), fin.size(), hipMemcpyHostToDevice, st));
        }
        HIP_TRY(hipMemcpyAsync(d_off, f_off.data(), 8 * cnt, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_len, f_len.data(), 8 * cnt, hipMemcpyHostToDevice, st));
        if (int rc = k12_ranges(st, d, d_off, d_len, pass == 0 ? 0x06 : 0x07, d + o_out, outlen, cnt)) return rc;
        HIP_TRY(hipMemcpyAsync(res.data(), d + o_out, cnt * outlen, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t k = 0; k < cnt; k++) std::memcpy(out + f_idx[k] * outlen, &res[k * outlen], outlen);
    }
    return CIRCL_HIP_OK;
}
