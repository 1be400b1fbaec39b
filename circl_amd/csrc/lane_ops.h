// lane_ops.h -- the coefficient-level functions of kyber_dev.h / dilithium_dev.h behind one switch, for the unit-level primitive
// circl_hip_lane_op (prim_kernels.h: the DEVICE instantiation, swept on the GPU by tests/test_gpu_lane_prims.py) and for
// tests/hostsim (the HOST instantiation of the same source, tests/test_hostsim.py).  Op codes: include/circl_hip.h.
#pragma once
#include "../../include/circl_hip.h"
#include "kyber_dev.h"
#include "dilithium_dev.h"

namespace circl {
namespace prim {

CIRCL_HD void lane_op_eval(int op, int arg, uint32_t x, uint32_t y, uint32_t &r0_out, uint32_t &r1_out) {
    uint32_t r0 = 0, r1 = 0;
    switch (op) {
    case CIRCL_HIP_LANE_KYBER_COMPRESS:
        r0 = arg == 4 ? kyber::compress_coeff<4>((int)x) : arg == 5 ? kyber::compress_coeff<5>((int)x) : arg == 10 ? kyber::compress_coeff<10>((int)x)
                                                                                                                       : kyber::compress_coeff<11>((int)x);
        break;
    case CIRCL_HIP_LANE_KYBER_DECOMPRESS:
        r0 = (uint32_t)(arg == 4 ? kyber::decompress_coeff<4>(x) : arg == 5 ? kyber::decompress_coeff<5>(x) : arg == 10 ? kyber::decompress_coeff<10>(x)
                                 : arg == 11 ? kyber::decompress_coeff<11>(x) : kyber::decompress_coeff<1>(x));
        break;
    case CIRCL_HIP_LANE_KYBER_MSG_BIT: r0 = kyber::msg_bit((int)x); break;
    case CIRCL_HIP_LANE_KYBER_MULC: r0 = kyber::mulc(x, kyber::mulc_const(y)); break;
    case CIRCL_HIP_LANE_KYBER_REDUCE32: r0 = kyber::reduce32(x); break;
    case CIRCL_HIP_LANE_KYBER_NORMALIZE:
        r0 = (uint32_t)kyber::normalize((int)(int16_t)x);
        r1 = (uint32_t)kyber::barrett((int)(int16_t)x);
        break;
    case CIRCL_HIP_LANE_KYBER_CBD2_WORD: r0 = kyber::cbd2_bias8_word(x); break;
    case CIRCL_HIP_LANE_KYBER_DOT2: r0 = (uint32_t)kyber::dot2(x, y, arg); break;
    case CIRCL_HIP_LANE_DIL_DECOMPOSE:
        if (arg == 95232) dilithium::decompose<95232>(x, r0, r1);
        else dilithium::decompose<261888>(x, r0, r1);
        break;
    case CIRCL_HIP_LANE_DIL_USE_HINT: r0 = arg == 95232 ? dilithium::use_hint<95232>(x, y) : dilithium::use_hint<261888>(x, y); break;
    case CIRCL_HIP_LANE_DIL_MAKE_HINT: r0 = arg == 95232 ? dilithium::make_hint<95232>(x, y) : dilithium::make_hint<261888>(x, y); break;
    case CIRCL_HIP_LANE_DIL_POWER2ROUND: dilithium::power2round(x, r0, r1); break;
    case CIRCL_HIP_LANE_DIL_MONT32: r0 = dilithium::mont32(x, y); break;
    case CIRCL_HIP_LANE_DIL_MONT64: r0 = dilithium::mont64(((uint64_t)y << 32) | x); break;
    case CIRCL_HIP_LANE_DIL_NORMALIZE:
        r0 = dilithium::normalize(x);
        r1 = dilithium::fold(x);
        break;
    case CIRCL_HIP_LANE_DIL_EXCEEDS: r0 = dilithium::exceeds(x, y) ? 1u : 0u; break;
    }
    r0_out = r0;
    r1_out = r1;
}

}  // namespace prim
}  // namespace circl
