// api_prims.hip -- the unit-level primitives (Keccak-f, ring transforms, sponges) and the batched XOF / KangarooTwelve
// service of the C ABI (include/circl_hip.h).  No CPU compute path: the KangarooTwelve host code only lays out the tree's
// nodes; every permutation runs on the device.
#include "host_common.h"
#include "prim_kernels.h"

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

// KangarooTwelve draft -10 (xof/k12/k12.go), n independent computations.  Tree hashing maps onto the batched
// sponge service as two or three TurboSHAKE128 batches: every 8192-byte leaf of every long message (D = 0x0B,
// k12.go:136-160), then the final nodes of the long messages (D = 0x06, :141-142, :383-395) and the short
// messages (|M| + |C| + |length_encode(|C|)| <= 8192, D = 0x07, :60-66).
namespace {
void k12_length_encode(std::vector<uint8_t> &v, uint64_t x) {  // k12.go:333-342
    uint8_t be[8];
    int nz = 0;
    for (int i = 0; i < 8; i++) be[i] = (uint8_t)(x >> (56 - 8 * i));
    while (nz < 8 && be[nz] == 0) nz++;
    v.insert(v.end(), be + nz, be + 8);
    v.push_back((uint8_t)(8 - nz));
}
}  // namespace

extern "C" int circl_hip_k12(const uint8_t *msg_blob, const uint64_t *msg_off, const uint8_t *ctx_blob, const uint64_t *ctx_off, uint8_t *out,
                             size_t outlen, size_t n, int device) {
    constexpr size_t CHUNK = 8192;
    if (outlen == 0) return CIRCL_HIP_EPARAM;
    if (n == 0) return CIRCL_HIP_OK;
    // S_i = M_i || C_i || length_encode(|C_i|), first chunks and leaves gathered separately
    std::vector<uint8_t> leaves, tail;
    std::vector<uint64_t> leaf_off{0};
    std::vector<std::vector<uint8_t>> head(n);   // S_0 of each message (whole S for short ones)
    std::vector<size_t> nleaves(n, 0);
    for (size_t i = 0; i < n; i++) {
        const uint8_t *m = msg_blob + msg_off[i];
        const size_t ml = (size_t)(msg_off[i + 1] - msg_off[i]);
        const uint8_t *c = ctx_blob ? ctx_blob + ctx_off[i] : nullptr;
        const size_t cl = ctx_blob ? (size_t)(ctx_off[i + 1] - ctx_off[i]) : 0;
        tail.clear();
        if (cl) tail.insert(tail.end(), c, c + cl);
        k12_length_encode(tail, cl);
        const size_t total = ml + tail.size();
        auto byte_range = [&](size_t lo, size_t hi, std::vector<uint8_t> &dst) {  // S[lo, hi)
            if (lo < ml) dst.insert(dst.end(), m + lo, m + std::min(hi, ml));
            if (hi > ml) dst.insert(dst.end(), tail.begin() + (std::max(lo, ml) - ml), tail.begin() + (hi - ml));
        };
        byte_range(0, std::min(total, CHUNK), head[i]);
        for (size_t off = CHUNK; off < total; off += CHUNK) {
            byte_range(off, std::min(total, off + CHUNK), leaves);
            leaf_off.push_back(leaves.size());
            nleaves[i]++;
        }
    }
    const size_t total_leaves = leaf_off.size() - 1;
    std::vector<uint8_t> cv(32 * total_leaves);
    if (total_leaves) {
        leaves.resize(leaves.size() + 16);
        const int rc = circl_hip_xof(168, 0x0B, 12, leaves.data(), leaf_off.data(), cv.data(), 32, total_leaves, device);
        if (rc) return rc;
    }
    // final nodes: long and short messages go out as two batches (different domain bytes)
    for (int pass = 0; pass < 2; pass++) {
        std::vector<uint8_t> blob;
        std::vector<uint64_t> off{0};
        std::vector<size_t> idx;
        size_t cvpos = 0;
        for (size_t i = 0; i < n; i++) {
            const bool is_long = nleaves[i] != 0;
            if (is_long == (pass == 0)) {
                blob.insert(blob.end(), head[i].begin(), head[i].end());
                if (is_long) {
                    static const uint8_t sep[8] = {3, 0, 0, 0, 0, 0, 0, 0};
                    blob.insert(blob.end(), sep, sep + 8);
                    blob.insert(blob.end(), cv.begin() + 32 * cvpos, cv.begin() + 32 * (cvpos + nleaves[i]));
                    k12_length_encode(blob, nleaves[i]);
                    blob.push_back(0xff);
                    blob.push_back(0xff);
                }
                off.push_back(blob.size());
                idx.push_back(i);
            }
            cvpos += nleaves[i];
        }
        if (idx.empty()) continue;
        blob.resize(blob.size() + 16);
        std::vector<uint8_t> res(outlen * idx.size());
        const int rc = circl_hip_xof(168, pass == 0 ? 0x06 : 0x07, 12, blob.data(), off.data(), res.data(), outlen, idx.size(), device);
        if (rc) return rc;
        for (size_t k = 0; k < idx.size(); k++) std::memcpy(out + idx[k] * outlen, &res[k * outlen], outlen);
    }
    return CIRCL_HIP_OK;
}
