// Shared helpers for libnfe_render.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>

#include "nfe_render.h"

namespace nfe {

int fail(int code, const char* fmt, ...);   // records the thread-local message, returns code
// nfe_render.hip: the evaluation pass of nfe_render_backward on the forward kernel (see there)
int render_eval_pass(const nfe_render_backward_args* a, const float* decoder_packed, float* rec_sig, float* rec_a, hipStream_t st);
int render_color_dot_pass(const nfe_render_backward_args* a, float* rec_sig, float* rec_a, hipStream_t st);
const char* last_error();
unsigned long long* handoff_status_word();   // nfe_api.cpp: the process's sticky (poisoned calls << 32 | lost hand-offs) word, or null

#define NFE_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return ::nfe::fail(NFE_EINVAL, __VA_ARGS__); \
    } while (0)

#define NFE_CHECK_LAUNCH(what)                                                                   \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) return ::nfe::fail(NFE_ELAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

// ---- host: device properties and the LDS opt-in, per device (a process may use several)
constexpr int MAX_DEVICES = 64;
static inline int current_device() { int dev = 0; return hipGetDevice(&dev) == hipSuccess && dev >= 0 ? dev : 0; }
// CU count of the CURRENT device (a process may render on several: one cached value per device id)
static inline int num_cus() {
    static std::atomic<int> cached[MAX_DEVICES];
    const int dev = current_device();
    int cus = dev < MAX_DEVICES ? cached[dev].load(std::memory_order_relaxed) : 0;
    if (!cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
        if (dev < MAX_DEVICES) cached[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

// Opt a kernel into more than 64 KB of dynamic LDS.  The attribute belongs to the (kernel, device) pair, so callers apply it per
// device (LdsOptIn below) and a failure is an error of the launch, not something to find out from the launch's own failure.
template <typename K>
static inline hipError_t allow_lds(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
struct LdsOptIn {          // one per kernel instantiation (function-local static): which devices have the attribute
    std::atomic<unsigned long long> done{0};
    template <typename K> hipError_t apply(K kernel, int bytes) {
        const int dev = current_device();
        if (dev < MAX_DEVICES && (done.load(std::memory_order_acquire) >> dev & 1ull)) return hipSuccess;
        const hipError_t e = allow_lds(kernel, bytes);
        if (e == hipSuccess && dev < MAX_DEVICES) done.fetch_or(1ull << dev, std::memory_order_release);
        return e;
    }
};

struct LdsOptInMax {       // the same for a kernel whose LDS size depends on the call: the largest size each device has been given
    std::atomic<int> have[MAX_DEVICES];
    template <typename K> hipError_t apply(K kernel, int bytes) {
        const int dev = current_device();
        if (dev < MAX_DEVICES && have[dev].load(std::memory_order_acquire) >= bytes) return hipSuccess;
        const hipError_t e = allow_lds(kernel, bytes);
        if (e == hipSuccess && dev < MAX_DEVICES) {
            int cur = have[dev].load(std::memory_order_relaxed);
            while (cur < bytes && !have[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
        }
        return e;
    }
};

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
// Per-sample records of the backward (sigma / dL/dsigma, a / omega, T, depth copy) live in the order its kernels walk them:
// [view][64-ray tile][sample][lane of the tile], a tile being 8 x 8 pixels of a square image whose side is a multiple of 8, else 64
// consecutive rays - a wave of bwd_ray_kernel / bwd_scatter_sorted_kernel reads and writes 256-byte rows.  Slot of sample 0 of ray m;
// sample k is 64 * k further.
__device__ __forceinline__ long long bwd_slot_base(int R, int M, int S, int n, int m) {
    int t, l;
    if (R > 0 && (R & 7) == 0 && (long long)R * R == M) { const int px = m % R, py = m / R; t = (py >> 3) * (R >> 3) + (px >> 3); l = (py & 7) * 8 + (px & 7); }
    else { t = m >> 6; l = m & 63; }
    return (((long long)n * ((M + 63) >> 6) + t) * S) * 64 + l;
}
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

struct Taps { int xc0, xc1, yc0, yc1; float w[4]; float wdef; };   // wdef = (sum of weights) - 1, exactly 0 when all 4 taps are inside

// One axis of the bilinear footprint: clamped tap coordinates, 1-D weights with out-of-range taps zeroed, and whether both
// taps are inside (as 1.0 / 0.0).  The 2-D weights are separable ((vx ? ex : 0) * (vy ? ey : 0) == (vx && vy) ? ex * ey : 0,
// bit for bit: the factors are in [0, 1]), so on SQUARE planes the three projections (x,y), (x,z), (z,x) of a sample share
// their axes: three axis computations instead of six.
// Every per-lane condition here is consumed by the v_cndmask that follows its v_cmp; none is combined with another one.  Rounds
// 2-3 blamed the combined form `(vx && vy) ? w : 0` for zeros in lanes 48-63 (r02_lane_mask.md); the cause was the packed multiply
// the vectoriser made of the weight products in BOTH forms (v_pk_mul_f32 ... op_sel:[0,1]: profiles/experiments/r04_pk_opsel_hazard.md),
// which the build's assembly pass (csrc/pk_opsel_fix.py) now commutes.  -DNFE_TAPS_COMBINED=1 keeps the old form as a reproducer.
struct Axis {
    int c0, c1; float a0, a1; float in;
};
// Hide a value from the optimiser (no instruction): used where it would otherwise re-combine two per-lane conditions into one
// scalar-unit mask operation (see above).
__device__ __forceinline__ float opaque_f(float v) { asm volatile("; nfe_launder %0" : "+v"(v)); return v; }
__device__ __forceinline__ Axis axis_geometry(int size, float g) {
    Axis a;
    const float i = (g + 1.0f) * (0.5f * (float)size) - 0.5f;
    const float f0 = floorf(i);
    const float d = i - f0, e = 1.0f - d;
    const int x0 = (int)fminf(fmaxf(f0, -2.0f), (float)(size + 1));
    const int x1 = x0 + 1;
    a.a0 = (unsigned)x0 < (unsigned)size ? e : 0.0f;
    a.a1 = (unsigned)x1 < (unsigned)size ? d : 0.0f;
    a.c0 = min(max(x0, 0), size - 1); a.c1 = min(max(x1, 0), size - 1);
    a.in = opaque_f((unsigned)x0 < (unsigned)(size - 1) ? 1.0f : 0.0f);   // 0 <= x0 and x0 + 1 < size; opaque: see taps_from_axes
    return a;
}
__device__ __forceinline__ Taps taps_from_axes(const Axis& u, const Axis& v) {      // u indexes W, v indexes H
    Taps t;
    t.xc0 = u.c0; t.xc1 = u.c1; t.yc0 = v.c0; t.yc1 = v.c1;
    t.w[0] = u.a0 * v.a0; t.w[1] = u.a1 * v.a0; t.w[2] = u.a0 * v.a1; t.w[3] = u.a1 * v.a1;
    // (sum - 1) when a tap is outside, exactly 0 when all four are inside - as arithmetic on the two 1.0 / 0.0 flags.  The select
    // form `u.in * v.in != 0 ? 0 : sum - 1` was folded back by the compiler into v_cmp, v_cmp, s_and_b64, v_cndmask: the shape the
    // comment above keeps out of these kernels (tools/lint_lane_masks.py, S1).
    t.wdef = (1.0f - u.in * v.in) * (((t.w[0] + t.w[1]) + (t.w[2] + t.w[3])) - 1.0f);
    return t;
}
// Bilinear tap geometry of one sample on one plane: F.grid_sample(bilinear, zeros, align_corners=False), renderer.py:64,
// unnormalised as ATen's CPU kernel does.  Clamped coordinates are always addressable; out-of-range taps carry weight 0.
__device__ __forceinline__ Taps tap_geometry(int H, int W, float u, float v) {
    return taps_from_axes(axis_geometry(W, u), axis_geometry(H, v));
}

}  // namespace nfe
