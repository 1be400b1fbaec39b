#!/usr/bin/env python3
"""Generates build/bank_probe_gen.hip: does the VGPR bank (register number mod 4) of the operands change the
issue cost of a VALU instruction on gfx950, and how long is the dependent-issue latency?

Every test is a block of 64 instructions on PHYSICAL registers (v16..v47, clobbered), so the operand banks are
what the test says and not what the register allocator happened to pick.  Output: cycles per wave-instruction
per SIMD at 1 / 2 / 4 / 8 waves per SIMD (wall time x 2.4 GHz nominal).
"""
import sys


def bank_reg(base, bank, k=0):
    """k-th register of bank `bank` at or above `base` (base is a multiple of 4)"""
    return base + 4 * k + bank


TESTS = []


def add(label, insts):
    assert len(insts) == 64, (label, len(insts))
    TESTS.append((label, insts))


# 8 independent accumulators d_i = v16+i (bank i % 4); other sources from v32..v47
def two_src(op, same):
    out = []
    for rep in range(8):
        for i in range(8):
            d = 16 + i
            b = bank_reg(32, (i % 4) if same else (i + 1) % 4, rep % 4)
            out.append(f"{op} v{d}, v{d}, v{b}")
    return out


def three_src(op, mode, tail=""):
    out = []
    for rep in range(8):
        for i in range(8):
            d = 16 + i
            bd = i % 4
            if mode == "distinct":
                b, c = bank_reg(32, (bd + 1) % 4, rep % 4), bank_reg(32, (bd + 2) % 4, (rep + 1) % 4)
            elif mode == "bc_same":
                b, c = bank_reg(32, (bd + 1) % 4, rep % 4), bank_reg(32, (bd + 1) % 4, (rep + 1) % 4)
            elif mode == "db_same":
                b, c = bank_reg(32, bd, rep % 4), bank_reg(32, (bd + 2) % 4, (rep + 1) % 4)
            elif mode == "all_same":
                b, c = bank_reg(32, bd, rep % 4), bank_reg(32, bd, (rep + 1) % 4)
            out.append(f"{op} v{d}, v{d}, v{b}, v{c}{tail}")
    return out


def dst_other(op, mode):
    """dst is not a source: v_xor v(24+i), v(16+i), vB -- dst bank equal / different from src banks"""
    out = []
    for rep in range(8):
        for i in range(8):
            s = 16 + i
            bs = i % 4
            b = bank_reg(32, (bs + 1) % 4, rep % 4)
            d = bank_reg(24, bs if mode == "dst_eq_src0" else (bs + 2) % 4, i // 4)
            out.append(f"{op} v{d}, v{s}, v{b}")
    return out


def chain(op_fmt, nacc):
    out = []
    for k in range(64):
        d = 16 + (k % nacc)
        out.append(op_fmt.format(d=f"v{d}", b=f"v{bank_reg(32, (d + 1) % 4, 0)}", c=f"v{bank_reg(32, (d + 2) % 4, 1)}"))
    return out


add("xor 2src, banks differ", two_src("v_xor_b32", False))
add("xor 2src, same bank", two_src("v_xor_b32", True))
add("add 2src, banks differ", two_src("v_add_u32", False))
add("add 2src, same bank", two_src("v_add_u32", True))
add("xor dst bank = src0 bank", dst_other("v_xor_b32", "dst_eq_src0"))
add("xor dst bank differs", dst_other("v_xor_b32", "dst_ne"))
add("bitop3 3 banks", three_src("v_bitop3_b32", "distinct", " bitop3:0x96"))
add("bitop3 b,c same bank", three_src("v_bitop3_b32", "bc_same", " bitop3:0x96"))
add("bitop3 d,b same bank", three_src("v_bitop3_b32", "db_same", " bitop3:0x96"))
add("bitop3 all same bank", three_src("v_bitop3_b32", "all_same", " bitop3:0x96"))
add("alignbit 2 banks (+const)", [s.replace(", vC", "") for s in [f"v_alignbit_b32 v{16 + i}, v{16 + i}, v{bank_reg(32, (i + 1) % 4, r % 4)}, 7" for r in range(8) for i in range(8)]])
add("alignbit same bank", [f"v_alignbit_b32 v{16 + i}, v{16 + i}, v{bank_reg(32, i % 4, r % 4)}, 7" for r in range(8) for i in range(8)])
add("fma_f32 3 banks", three_src("v_fma_f32", "distinct"))
add("fma_f32 all same bank", three_src("v_fma_f32", "all_same"))
add("mul_u32_u24 banks differ", two_src("v_mul_u32_u24", False))
add("mul_u32_u24 same bank", two_src("v_mul_u32_u24", True))
for n in (1, 2, 3, 4, 8):
    add(f"xor chain, {n} accumulators", chain("v_xor_b32 {d}, {d}, {b}", n))
for n in (1, 2, 4):
    add(f"alignbit chain, {n} acc", chain("v_alignbit_b32 {d}, {d}, {b}, 7", n))
for n in (1, 2, 4):
    add(f"bitop3 chain, {n} acc", chain("v_bitop3_b32 {d}, {d}, {b}, {c} bitop3:0x96", n))


# ---- mixes of the fast and the slow class: does switching cost anything, and is a slow-class result late? ----
def mix(group, fast="v_xor_b32 v{d}, v{d}, v{b}", slow="v_alignbit_b32 v{d}, v{d}, v{b}, 7"):
    out = []
    k = 0
    while len(out) < 64:
        for cls in (fast, slow):
            for g in range(group):
                i = (k + g) % 8
                d = (16 if cls is fast else 24) + i
                out.append(cls.format(d=d, b=bank_reg(32, (d + 1) % 4, g % 4), c=bank_reg(32, (d + 2) % 4, (g + 1) % 4)))
            k += group
    return out[:64]


for g in (1, 2, 4, 8, 32):
    add(f"xor/alignbit groups of {g}", mix(g))
for g in (1, 4, 32):
    add(f"bitop3/alignbit groups of {g}", mix(g, fast="v_bitop3_b32 v{d}, v{d}, v{b}, v{c} bitop3:0x96"))


def cross_chain(nacc):
    """bitop3 -> alignbit -> bitop3 ... each consuming the previous result of the same accumulator"""
    out = []
    for k in range(32):
        d = 16 + (k % nacc)
        out.append(f"v_bitop3_b32 v{d}, v{d}, v{bank_reg(32, (d + 1) % 4, 0)}, v{bank_reg(32, (d + 2) % 4, 1)} bitop3:0x96")
        out.append(f"v_alignbit_b32 v{d}, v{d}, v{bank_reg(32, (d + 1) % 4, 2)}, 7")
    return out


for n in (1, 2, 4, 8):
    add(f"bitop3->alignbit chain, {n} acc", cross_chain(n))

out = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>',
       '#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)']
clob = ", ".join(f'"v{r}"' for r in range(16, 48))
init = "\\n ".join(f"v_mov_b32 v{r}, %0" for r in range(16, 48))
for idx, (label, insts) in enumerate(TESTS):
    body = "\\n ".join(insts)
    out.append(f'''__global__ void __launch_bounds__(256) k{idx}(unsigned *out, int iters) {{
    unsigned seed = threadIdx.x * 2654435761u, res;
    asm volatile("{init}" : : "v"(seed) : {clob});
    for (int i = 0; i < iters; i++) {{
        asm volatile("{body}" : : : {clob});
    }}
    asm volatile("v_xor_b32 %0, v16, v17\\n v_xor_b32 %0, %0, v18\\n v_xor_b32 %0, %0, v19" : "=v"(res) : : {clob});
    if (res == 0x12345678u) out[0] = res;
}}''')

out.append('''template <class K> double run(K kern, int wps) {
    static unsigned *out = nullptr; if (!out) CK(hipMalloc(&out, 64));
    const int iters = 4000, blocks = 256 * wps;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e-3 * 2.4e9 / ((double)iters * 64 * wps);
}
int main() {
    printf("%-28s %8s %8s %8s %8s   (cycles per wave-instruction per SIMD at 2.4 GHz nominal)\\n", "test", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");''')
for idx, (label, _) in enumerate(TESTS):
    out.append(f'    printf("%-28s %8.2f %8.2f %8.2f %8.2f\\n", "{label}", run(k{idx}, 1), run(k{idx}, 2), run(k{idx}, 4), run(k{idx}, 8));')
out.append("    return 0;\n}")
open(sys.argv[1] if len(sys.argv) > 1 else "build/bank_probe_gen.hip", "w").write("\n".join(out) + "\n")
