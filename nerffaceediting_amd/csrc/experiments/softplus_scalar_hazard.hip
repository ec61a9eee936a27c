// EXPERIMENT, KNOWN WRONG, never linked into libnfe_render.so: the reproducer of profiles/experiments/r04_asm_trans_hazard.md as a
// stand-alone translation unit (rounds 3-4 carried it as -DNFE_SOFTPLUS_SCALAR=1 inside nfe_render.hip).
//
// A plain `v_add_f32` written as inline asm reads a v_exp_f32 / v_log_f32 result one instruction after the transcendental wrote it.
// hipcc pads the transcendental -> VALU wait state only for instructions it emits itself, not for the text of an asm statement, so
// the add can read the register before the result has landed: run-dependent results on MI355X.  tools/asm_audit.py rule TRNS must
// flag every such statement; tests/test_lint_cpu.py::test_asm_audit_detects_the_reproduced_hazard audits THIS file to prove the
// detector sees what the hardware punished.  The product's softplus (nfe_render.hip, softplus_log2_x16) uses compiler-emitted
// v_pk_add_f32 and is clean under the same rule.
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float add_f32_plain(float a, float b) {      // an add the SLP vectoriser cannot pair into v_pk_add_f32
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float relu_bits(float y) { return __int_as_float(max(__float_as_int(y), 0)); }

// softplus(x) / ln 2 on y = x log2(e), as max(y, 0) + log2(1 + 2^-|y|), with the hazardous adds
__device__ __forceinline__ void softplus_log2_x16_scalar_hazard(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const float e0 = add_f32_plain(__builtin_amdgcn_exp2f(-__builtin_fabsf(a[r])), 1.0f);
        const float e1 = add_f32_plain(__builtin_amdgcn_exp2f(-__builtin_fabsf(a[r + 1])), 1.0f);
        a[r] = add_f32_plain(__builtin_amdgcn_logf(e0), relu_bits(a[r]));
        a[r + 1] = add_f32_plain(__builtin_amdgcn_logf(e1), relu_bits(a[r + 1]));
    }
}

__global__ void softplus_scalar_hazard_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = in[(long long)r * n + i];
    softplus_log2_x16_scalar_hazard(a);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(long long)r * n + i] = a[r];
}
