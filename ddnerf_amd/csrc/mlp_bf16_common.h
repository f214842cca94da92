// Shared device helpers of the kernels on the bf16 matrix cores (mlp_bf16.hip, mlp_x3_fwd.hip, mlp_x3_fwd_train.hip, mlp_x3_train.hip).
#pragma once
#include <utility>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// k-order: inside every 16 columns the quads are stored [0,2,1,3]
__host__ __device__ __forceinline__ constexpr int korder(int p) {  // packed position -> original column (an involution)
    return (p & ~15) | ((p & 3) + 4 * (((p >> 2) & 1) * 2 + ((p >> 3) & 1)));
}

// k-order of the 16x16x32 kernel (mlp_bf16.hip): inside every 32 columns, packed position 8g + e (lane group g, fragment
// element e) holds original column 16(e>>2) + 4g + (e&3)
__host__ __device__ __forceinline__ constexpr int korder32(int p) {  // packed position -> original column
    return (p & ~31) | (16 * ((p >> 2) & 1) + 4 * ((p >> 3) & 3) + (p & 3));
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// LDS-DMA of the very first stage only (nothing to overlap it with); asm so that hipcc's wait-count pass does not see a
// pending LDS write (it would degrade every later counted lgkmcnt(N) to lgkmcnt(0)).
__device__ __forceinline__ void dma_piece(const char *__restrict__ gsrc_lane, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr_of(const char *p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p;
}

// One of the two B files (HA; this layer's / the next layer's activations ping-pong between HA and HB, 128
// registers each) is homed in the accumulator half of the unified register file: MFMA reads its B operand from there directly, and the arch VGPRs stay free
// for the accumulator tiles, the A ring and the VALU temporaries.  Left to itself hipcc keeps shuttling them
// (5,600 v_accvgpr moves and 109 spills measured); defining every packed word through this one-instruction asm
// gives it the AGPR register class from birth.
__device__ __forceinline__ unsigned to_agpr(unsigned v) {
    unsigned a;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
    return a;
}
