// api_x25519.hip -- batch X25519 behind the C ABI (include/circl_hip.h): the Diffie-Hellman half of X25519MLKEM768 and
// X-Wing.  No CPU compute path.
#include "host_common.h"
#include "x25519_kernels.h"

using namespace circl::host;

namespace circl {
namespace host {
// internal (api_hybrid.hip): out_base[i] = X25519(scalar_i, 9) and out_shared[i] = X25519(scalar_i, point_i) in one launch
int x25519_pair_dev(const uint8_t *d_scalar, const uint8_t *d_point, uint8_t *d_out_base, uint8_t *d_out_shared, uint8_t *d_ok, size_t n,
                    hipStream_t st) {
    if (n == 0) return CIRCL_HIP_OK;
    const unsigned nb = (unsigned)((n + 63) / 64);
    ProfScope ps(CIRCL_HIP_KERNEL_X25519, st);
    hipLaunchKernelGGL(circl::x25519::x25519_pair_kernel, dim3(2 * nb), dim3(64), 0, st, reinterpret_cast<const uint32_t *>(d_scalar),
                       reinterpret_cast<const uint32_t *>(d_point), reinterpret_cast<uint32_t *>(d_out_base), reinterpret_cast<uint32_t *>(d_out_shared), d_ok,
                       n, nb);
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}
}  // namespace host
}  // namespace circl

extern "C" {

int circl_hip_x25519_dev(const uint8_t *d_scalar, const uint8_t *d_point, uint8_t *d_out, uint8_t *d_ok, size_t n, void *stream) {
    if (ndev() <= 0) return CIRCL_HIP_ENODEV;
    if (!d_scalar || !d_out) return CIRCL_HIP_EPARAM;
    if ((reinterpret_cast<uintptr_t>(d_scalar) | reinterpret_cast<uintptr_t>(d_point) | reinterpret_cast<uintptr_t>(d_out)) & 3) return CIRCL_HIP_EWORKSPACE;
    if (n == 0) return CIRCL_HIP_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
    ProfScope ps(CIRCL_HIP_KERNEL_X25519, st);
    if (d_point)
        hipLaunchKernelGGL(circl::x25519::x25519_kernel<false>, grid, block, 0, st, reinterpret_cast<const uint32_t *>(d_scalar),
                           reinterpret_cast<const uint32_t *>(d_point), reinterpret_cast<uint32_t *>(d_out), d_ok, n);
    else
        hipLaunchKernelGGL(circl::x25519::x25519_kernel<true>, grid, block, 0, st, reinterpret_cast<const uint32_t *>(d_scalar),
                           static_cast<const uint32_t *>(nullptr), reinterpret_cast<uint32_t *>(d_out), d_ok, n);
    HIP_TRY(hipGetLastError());
    return CIRCL_HIP_OK;
}

int circl_hip_x25519(const uint8_t *scalar, const uint8_t *point, uint8_t *out, uint8_t *ok, size_t n, int device) {
    if (!scalar || !out) return n ? CIRCL_HIP_EPARAM : CIRCL_HIP_OK;
    PipeOpts opts;
    opts.chunk_items = host_chunk_items(size_t(1) << 16);  // 0.7 ms of ladder per chunk
    opts.wipe_device = true;
    const std::function<size_t(size_t)> no_ws = [](size_t) { return size_t(0); };
    return shard(n, device, [&](int dev, size_t lo, size_t cnt) {
        std::vector<HIn> ins = {{scalar + lo * 32, 32, true}};
        if (point) ins.push_back({point + lo * 32, 32});
        return run_pipeline(dev, cnt, ins, {}, {{out + lo * 32, 32, true}, {ok ? ok + lo : nullptr, 1}}, no_ws, opts, [&](Chunk &c) {
            return circl_hip_x25519_dev(c.in[0], point ? c.in[1] : nullptr, c.out[0], c.out[1], c.cnt, c.st);
        });
    }, kHeavyOneDeviceMax);
}

}  // extern "C"
