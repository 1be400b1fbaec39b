// tools/pcie_probe.hip -- measurement aid for the host-buffer path: what moves bytes fastest between page-locked host
// memory and HBM on this box?  hipMemcpyAsync (SDMA engines) against copy KERNELS that read / write the mapped host memory
// directly, one direction at a time and both at once, whole buffers and pipelined chunks.
//   hipcc --offload-arch=gfx950 -O3 -o build/pcie_probe tools/pcie_probe.hip && build/pcie_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

__global__ void __launch_bounds__(256) copy_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const size_t total = (size_t)1 << 30;  // 1 GiB per direction
    uint8_t *h_in, *h_out, *d_in, *d_out;
    CK(hipHostMalloc((void **)&h_in, total, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&h_out, total, hipHostMallocDefault));
    CK(hipMalloc((void **)&d_in, total));
    CK(hipMalloc((void **)&d_out, total));
    memset(h_in, 1, total);
    memset(h_out, 2, total);
    CK(hipMemset(d_out, 3, total));
    hipStream_t s[8];
    for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    auto sync_all = [&] { CK(hipDeviceSynchronize()); };
    auto report = [&](const char *name, double bytes, double t) { printf("%-64s %7.2f GB/s  (%.2f ms)\n", name, bytes / t / 1e9, t * 1e3); fflush(stdout); };

    for (int rep = 0; rep < 2; rep++) {  // second round = warm
        printf("---- round %d ----\n", rep);
        double t;
        sync_all(); t = now(); CK(hipMemcpyAsync(d_in, h_in, total, hipMemcpyHostToDevice, s[0])); sync_all(); report("memcpyAsync H2D whole", total, now() - t);
        sync_all(); t = now(); CK(hipMemcpyAsync(h_out, d_out, total, hipMemcpyDeviceToHost, s[1])); sync_all(); report("memcpyAsync D2H whole", total, now() - t);
        sync_all(); t = now();
        CK(hipMemcpyAsync(d_in, h_in, total, hipMemcpyHostToDevice, s[0]));
        CK(hipMemcpyAsync(h_out, d_out, total, hipMemcpyDeviceToHost, s[1]));
        sync_all(); report("memcpyAsync H2D + D2H concurrently (2 streams), both dirs", 2.0 * total, now() - t);
        for (size_t chunk : {(size_t)4 << 20, (size_t)32 << 20}) {
            for (int ns : {2, 4, 8}) {
                sync_all(); t = now();
                int k = 0;
                for (size_t off = 0; off < total; off += chunk, k++) {
                    CK(hipMemcpyAsync(d_in + off, h_in + off, chunk, hipMemcpyHostToDevice, s[k % ns]));
                    CK(hipMemcpyAsync(h_out + off, d_out + off, chunk, hipMemcpyDeviceToHost, s[k % ns]));
                }
                sync_all();
                char nm[128];
                snprintf(nm, sizeof nm, "memcpyAsync chunks of %zu MB, H2D then D2H per stream, %d streams", chunk >> 20, ns);
                report(nm, 2.0 * total, now() - t);
            }
            sync_all(); t = now();
            for (size_t off = 0; off < total; off += chunk) CK(hipMemcpyAsync(d_in + off, h_in + off, chunk, hipMemcpyHostToDevice, s[0]));
            for (size_t off = 0; off < total; off += chunk) CK(hipMemcpyAsync(h_out + off, d_out + off, chunk, hipMemcpyDeviceToHost, s[1]));
            sync_all();
            char nm[128];
            snprintf(nm, sizeof nm, "memcpyAsync chunks of %zu MB, H2D stream + D2H stream", chunk >> 20);
            report(nm, 2.0 * total, now() - t);
        }
        for (int blocks : {256, 1024, 4096}) {
            char nm[128];
            sync_all(); t = now();
            hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s[0], (uint4 *)d_in, (const uint4 *)h_in, total / 16);
            sync_all(); snprintf(nm, sizeof nm, "copy kernel H2D (reads mapped host memory), %d blocks", blocks); report(nm, total, now() - t);
            sync_all(); t = now();
            hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s[1], (uint4 *)h_out, (const uint4 *)d_out, total / 16);
            sync_all(); snprintf(nm, sizeof nm, "copy kernel D2H (writes mapped host memory), %d blocks", blocks); report(nm, total, now() - t);
            sync_all(); t = now();
            hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s[0], (uint4 *)d_in, (const uint4 *)h_in, total / 16);
            hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s[1], (uint4 *)h_out, (const uint4 *)d_out, total / 16);
            sync_all(); snprintf(nm, sizeof nm, "copy kernels H2D + D2H concurrently, %d blocks each, both dirs", blocks); report(nm, 2.0 * total, now() - t);
        }
        // mixed: SDMA one way, kernel the other
        sync_all(); t = now();
        CK(hipMemcpyAsync(d_in, h_in, total, hipMemcpyHostToDevice, s[0]));
        hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, s[1], (uint4 *)h_out, (const uint4 *)d_out, total / 16);
        sync_all(); report("memcpyAsync H2D + copy kernel D2H concurrently, both dirs", 2.0 * total, now() - t);
        sync_all(); t = now();
        hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, s[0], (uint4 *)d_in, (const uint4 *)h_in, total / 16);
        CK(hipMemcpyAsync(h_out, d_out, total, hipMemcpyDeviceToHost, s[1]));
        sync_all(); report("copy kernel H2D + memcpyAsync D2H concurrently, both dirs", 2.0 * total, now() - t);
    }
    // host-side: how fast do T threads copy pageable -> page-locked (the staging the library does)?
    uint8_t *pg = (uint8_t *)malloc(total);
    memset(pg, 5, total);
    for (int T : {1, 4, 8, 16, 32}) {
        double t = now();
        std::vector<std::thread> th;
        for (int i = 0; i < T; i++)
            th.emplace_back([&, i] { const size_t lo = total * i / T, hi = total * (i + 1) / T; memcpy(h_in + lo, pg + lo, hi - lo); });
        for (auto &x : th) x.join();
        char nm[64];
        snprintf(nm, sizeof nm, "host memcpy pageable -> pinned, %d threads", T);
        report(nm, total, now() - t);
    }
    return 0;
}
