#!/usr/bin/env python3
"""Generates build/valu_rate_gen.hip: an issue-rate probe for the VALU instructions our kernels use.

Each kernel runs ITER x 64 copies of one instruction over 8 independent accumulator registers; 2 and 8
waves per SIMD on every SIMD of the chip.  {d} = accumulator (dest and one source), {b}/{c} = other VGPRs,
{s} = an SGPR.  Output: cycles per wave-instruction per SIMD (wall time x 2.4 GHz).
"""
import sys

TESTS = [
    ("xor v,v", "v_xor_b32 {d}, {d}, {b}"),
    ("xor sgpr,v", "v_xor_b32 {d}, {s}, {d}"),
    ("xor const,v", "v_xor_b32 {d}, 5, {d}"),
    ("xor literal,v", "v_xor_b32 {d}, 0x12345, {d}"),
    ("and v,v", "v_and_b32 {d}, {d}, {b}"),
    ("and const,v", "v_and_b32 {d}, 15, {d}"),
    ("or v,v", "v_or_b32 {d}, {d}, {b}"),
    ("add v,v", "v_add_u32 {d}, {d}, {b}"),
    ("add const,v", "v_add_u32 {d}, 7, {d}"),
    ("sub v,v", "v_sub_u32 {d}, {d}, {b}"),
    ("mov v", "v_mov_b32 {d}, {b}"),
    ("min_u32 v,v", "v_min_u32 {d}, {d}, {b}"),
    ("max_i32 v,v", "v_max_i32 {d}, {d}, {b}"),
    ("cndmask v,v,vcc", "v_cndmask_b32 {d}, {d}, {b}, vcc"),
    ("lshlrev const", "v_lshlrev_b32 {d}, 3, {d}"),
    ("lshlrev v", "v_lshlrev_b32 {d}, {b}, {d}"),
    ("lshrrev const", "v_lshrrev_b32 {d}, 3, {d}"),
    ("ashrrev const", "v_ashrrev_i32 {d}, 3, {d}"),
    ("bfe_u32", "v_bfe_u32 {d}, {d}, 3, 12"),
    ("bfe_i32", "v_bfe_i32 {d}, {d}, 0, 16"),
    ("bitop3 3 vgpr", "v_bitop3_b32 {d}, {d}, {b}, {c} bitop3:0x96"),
    ("bitop3 dup src", "v_bitop3_b32 {d}, {d}, {b}, {b} bitop3:0x96"),
    ("bitop3 sgpr", "v_bitop3_b32 {d}, {d}, {b}, {s} bitop3:0x96"),
    ("alignbit const", "v_alignbit_b32 {d}, {d}, {b}, 7"),
    ("alignbit v", "v_alignbit_b32 {d}, {d}, {b}, {c}"),
    ("alignbit rot", "v_alignbit_b32 {d}, {d}, {d}, 7"),
    ("perm_b32", "v_perm_b32 {d}, {d}, {b}, {c}"),
    ("add3", "v_add3_u32 {d}, {d}, {b}, {c}"),
    ("lshl_add", "v_lshl_add_u32 {d}, {d}, 2, {b}"),
    ("lshl_or", "v_lshl_or_b32 {d}, {d}, 8, {b}"),
    ("and_or", "v_and_or_b32 {d}, {d}, {b}, {c}"),
    ("xad", "v_xad_u32 {d}, {d}, {b}, {c}"),
    ("mul_u32_u24", "v_mul_u32_u24 {d}, {d}, {b}"),
    ("mul_i32_i24", "v_mul_i32_i24 {d}, {d}, {b}"),
    ("mul_hi_u32_u24", "v_mul_hi_u32_u24 {d}, {d}, {b}"),
    ("mad_u32_u24", "v_mad_u32_u24 {d}, {d}, {b}, {c}"),
    ("mad_i32_i24", "v_mad_i32_i24 {d}, {d}, {b}, {c}"),
    ("mul_lo_u32", "v_mul_lo_u32 {d}, {d}, {b}"),
    ("mul_hi_u32", "v_mul_hi_u32 {d}, {d}, {b}"),
    ("mul_hi_i32", "v_mul_hi_i32 {d}, {d}, {b}"),
    ("mad_u64_u32", None),
    ("dot2_i32_i16", "v_dot2_i32_i16 {d}, {b}, {c}, {d}"),
    ("pk_add_u16", "v_pk_add_u16 {d}, {d}, {b}"),
    ("pk_mul_lo_u16", "v_pk_mul_lo_u16 {d}, {d}, {b}"),
    ("pk_mad_u16", "v_pk_mad_u16 {d}, {d}, {b}, {c}"),
    ("pk_sub_i16", "v_pk_sub_i16 {d}, {d}, {b}"),
    ("mul_lo_u16", "v_mul_lo_u16 {d}, {d}, {b}"),
    ("mad_u16", "v_mad_u16 {d}, {d}, {b}, {c}"),
    ("cvt_f32_u32", "v_cvt_f32_u32 {d}, {d}"),
    ("fma_f32", "v_fma_f32 {d}, {d}, {b}, {c}"),
    ("pk_fma_f32", None),
    ("mul_f32", "v_mul_f32 {d}, {d}, {b}"),
    ("cmp+addc", None),
]

out = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>',
       '#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)']
names = []
for idx, (label, fmt) in enumerate(TESTS):
    name = f"k{idx}"
    if fmt is None:
        if label == "mad_u64_u32":
            body = "\\n ".join(f"v_mad_u64_u32 %{2*i}, vcc, %16, %17, %{2*i}" for i in range(4))
            # 4 x 64-bit accumulators
            asm = (f'asm volatile("{body}" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(b), "v"(c) : "vcc");')
            body = "\\n ".join(f"v_mad_u64_u32 %{i}, vcc, %4, %5, %{i}" for i in range(4))
            asm = f'asm volatile("{body}" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(c) : "vcc");'
            per = 4
        elif label == "pk_fma_f32":
            body = "\\n ".join(f"v_pk_fma_f32 %{i}, %{i}, %4, %4" for i in range(4))
            asm = f'asm volatile("{body}" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(w4) : );'
            per = 4
        else:  # cmp + addc pair
            body = "\\n ".join(f"v_cmp_lt_u32 vcc, %{i}, %8\\n v_addc_co_u32 %{i}, vcc, %{i}, %9, vcc" for i in range(8))
            asm = f'asm volatile("{body}" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");'
            per = 16
    else:
        body = "\\n ".join(fmt.format(d=f"%{i}", b="%8", c="%9", s="%10") for i in range(8))
        asm = f'asm volatile("{body}" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sg) : "vcc");'
        per = 8
    reps = 64 // per if per <= 8 else 4
    out.append(f'''__global__ void __launch_bounds__(256) {name}(unsigned *out, int iters, unsigned sg) {{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned b = threadIdx.x * 2654435761u, c = b ^ 0x9e3779b9u;
    unsigned long long w0 = a0, w1 = a1, w2 = a2, w3 = a3, w4 = ((unsigned long long)b << 32) | c;
    for (int i = 0; i < iters; i++) {{
''' + "\n".join("        " + asm for _ in range(reps)) + f'''
    }}
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)w0 ^ (unsigned)w1 ^ (unsigned)w2 ^ (unsigned)w3) == 0x12345678u) out[0] = a0;
}}''')
    names.append((name, label, per * reps))

out.append('''template <class K> double run(K kern, int wps, int per_iter) {
    static unsigned *out = nullptr; if (!out) CK(hipMalloc(&out, 64));
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * wps);
}
int main() {
    printf("%-18s %8s %8s %8s   (cycles per wave-instruction per SIMD at 2.4 GHz nominal)\\n", "instruction", "1 w/SIMD", "2 w/SIMD", "8 w/SIMD");''')
for name, label, per in names:
    out.append(f'    printf("%-18s %8.2f %8.2f %8.2f\\n", "{label}", run({name}, 1, {per}), run({name}, 2, {per}), run({name}, 8, {per}));')
out.append("    return 0;\n}")
open(sys.argv[1] if len(sys.argv) > 1 else "build/valu_rate_gen.hip", "w").write("\n".join(out) + "\n")
