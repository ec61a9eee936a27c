// Error channel, version and the hand-off status words of libnfe_render.so (no device code here).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "nfe_common.h"
#include "nfe_render.h"

// The kernels of this library must go through csrc/Makefile: its assembly pass (pk_opsel_fix.py) removes a packed-fp32 operand form the
// compiler's vectoriser emits and MI355X misreads beside MFMAs (profiles/experiments/r04_pk_opsel_hazard.md).  A library linked from
// plain `hipcc -c *.hip` objects carries that hazard silently, so the one file every link needs refuses to compile outside the Makefile
// (which passes -DNFE_BUILT_BY_MAKEFILE=1 after it has patched and re-assembled the device code of the four .hip files).
static_assert(NFE_BUILT_BY_MAKEFILE == 1, "build libnfe_render.so with `make -C nerffaceediting_amd/csrc` (see the comment above)");

namespace nfe {
static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
const char* last_error() { return g_err; }

// Lost hand-offs (render_ws_kernel, bwd_decoder_kernel): ONE pinned, device-visible 64-bit word per process - low half = abandoned
// waits, high half = poisoned calls.  The kernel that closes a call adds both with a single system-scope atomic; the host reads or
// exchanges the word with a single atomic, so (lost, calls) can never be seen or cleared half-way (two 32-bit words and
// two exchanges could report calls = 0 and leave a count behind).  Sticky diagnostic only: no call looks at it before launching.
// Allocation failure (no GPU) leaves the pointer null: outputs are still poisoned with NaN and the per-call words still count.
unsigned long long* handoff_status_word() {
    static unsigned long long* word = [] {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return (unsigned long long*)nullptr; }
        memset(p, 0, 64);
        return (unsigned long long*)p;
    }();
    return word;
}

// the per-call count(s): `n_words` 32-bit words at word offsets off[] of the call's workspace, read after the stream has drained
static int call_status(const char* who, const void* workspace, const int* off, int n_words, nfe_stream_t stream, uint32_t* lost_out) {
    if (!workspace) return fail(NFE_EINVAL, "%s: workspace is null", who);
    unsigned words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipError_t e = hipMemcpyAsync(words, workspace, sizeof(words), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(NFE_ELAUNCH, "%s: %s", who, hipGetErrorString(e));
    unsigned lost = 0;
    for (int i = 0; i < n_words; ++i) lost += words[off[i]];
    if (lost_out) *lost_out = lost;
    if (lost) return fail(NFE_EHANDOFF, "%s: the last call on this workspace lost %u wave hand-offs (a producer / consumer wave gave up waiting for "
                                        "its partner); every output of that call was set to NaN - repeat it (NFE_RENDER_WS=0 / NFE_BWD_DECODER=single "
                                        "select the kernels without a hand-off)", who, lost);
    return NFE_OK;
}
}  // namespace nfe

extern "C" int nfe_abi_version(void) { return NFE_ABI_VERSION; }
extern "C" const char* nfe_last_error(void) { return nfe::last_error(); }

extern "C" int nfe_render_status(uint32_t* lost_handoffs, uint32_t* poisoned_calls, int clear) {
    unsigned long long* status = nfe::handoff_status_word();
    unsigned long long v = 0;
    if (status) v = clear ? __atomic_exchange_n(status, 0ull, __ATOMIC_RELAXED) : __atomic_load_n(status, __ATOMIC_RELAXED);
    if (lost_handoffs) *lost_handoffs = (uint32_t)(v & 0xffffffffull);
    if (poisoned_calls) *poisoned_calls = (uint32_t)(v >> 32);
    return NFE_OK;
}

// words 2 and 6 of the render workspace: final pass and coarse pass (minmax_init_kernel zeroes them at the start of every call)
extern "C" int nfe_render_call_status(const void* workspace, nfe_stream_t stream, uint32_t* lost_handoffs) {
    static const int off[2] = {2, 6};
    return nfe::call_status("nfe_render_call_status", workspace, off, 2, stream, lost_handoffs);
}
// word 0 of the backward workspace (zeroed at the start of every call)
extern "C" int nfe_render_backward_call_status(const void* workspace, nfe_stream_t stream, uint32_t* lost_handoffs) {
    static const int off[1] = {0};
    return nfe::call_status("nfe_render_backward_call_status", workspace, off, 1, stream, lost_handoffs);
}
