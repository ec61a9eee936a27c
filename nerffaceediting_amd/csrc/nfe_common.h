// Shared helpers for libnfe_render.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "nfe_render.h"

namespace nfe {

int fail(int code, const char* fmt, ...);   // records the thread-local message, returns code
const char* last_error();

#define NFE_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return ::nfe::fail(NFE_EINVAL, __VA_ARGS__); \
    } while (0)

#define NFE_CHECK_LAUNCH(what)                                                                   \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) return ::nfe::fail(NFE_ELAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Packed decoder blob layout (floats); see nfe_decoder_pack in nfe_render.hip.
constexpr int DEC_A_G0 = 0;      // [mb 2][ks4 4][lane 64][4]
constexpr int DEC_A_A0 = 2048;   // same
constexpr int DEC_A_G1 = 4096;   // [ks4 8][lane 64][4]
constexpr int DEC_A_A1 = 6144;   // same
constexpr int DEC_B_G0 = 8192;   // [64] hidden-unit order
constexpr int DEC_B_A0 = 8256;   // [64]
constexpr int DEC_B_G1 = 8320;   // [32] MFMA row order
constexpr int DEC_B_A1 = 8352;   // [32] MFMA row order
constexpr int DEC_FLOATS = 8384;  // fp32 fragments + biases (also the LDS footprint of either math mode)
// Split-bf16 fragments: 32 fragments x 64 lanes x 4 words (8 bf16).  Fragment index:
//   layer 0: ((net*2 + mb)*2 + ks)*2 + part        net 0=geo 1=app, ks 0..1, part 0=hi 1=lo
//   layer 1: 16 + (net*4 + ks)*2 + part            ks 0..3
constexpr int DEC_BF16 = DEC_FLOATS;
constexpr int DEC_BF16_WORDS = 32 * 64 * 4;
constexpr int DEC_TOTAL = DEC_FLOATS + DEC_BF16_WORDS;
static_assert(DEC_TOTAL == NFE_DECODER_PACKED_FLOATS, "decoder blob size");
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// order-preserving float <-> uint map for atomicMin/atomicMax on depths
__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// Philox4x32-10 (Salmon et al. 2011), counter = (c0,c1,c2,0), key = seed.
struct u32x4 { unsigned x, y, z, w; };
__device__ __forceinline__ u32x4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                               unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(unsigned bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }

}  // namespace nfe
