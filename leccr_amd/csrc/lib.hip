// lib.hip - library plumbing: version, thread-local error text, launch check.
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"
#include <algorithm>
#include <utility>
#include <vector>

namespace leccr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return -1000 - (int)e;
    }
    return LECCR_OK;
}

// ---- profiling: event pairs around selected launches, summed per kind on read ---------------------------------
static bool g_prof_on = false;
static constexpr int kProfCap = 1 << 17;       // 131 072 launches per window (~50 steps of the config-E model)
static int64_t g_prof_dropped = 0;              // launches of the current window that found the buffer full
struct ProfRec { hipEvent_t a, b; int kind; double work; };
static ProfRec* g_prof = nullptr;
static int g_prof_n = 0;

ProfScope::ProfScope(int kind, double work, hipStream_t stream) : slot(-1), st(stream) {
    if (!g_prof_on) return;
    if (g_prof_n >= kProfCap) { ++g_prof_dropped; return; }
    slot = g_prof_n++;
    ProfRec& r = g_prof[slot];
    if (!r.a) {
        (void)hipEventCreate(&r.a);
        (void)hipEventCreate(&r.b);
    }
    r.kind = kind;
    r.work = work;
    (void)hipEventRecord(r.a, st);
}
ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, st);
}
}  // namespace leccr

extern "C" int leccr_prof_enable(int on) {
    using namespace leccr;
    if (on && !g_prof) g_prof = static_cast<ProfRec*>(calloc(kProfCap, sizeof(ProfRec)));
    g_prof_on = on != 0;
    if (on) { g_prof_n = 0; g_prof_dropped = 0; }
    return LECCR_OK;
}

// Launches of the current window that were NOT recorded because the buffer was full (0 = the read-outs are complete).
extern "C" int64_t leccr_prof_dropped(void) { return leccr::g_prof_dropped; }

// ms[k], work[k] (FLOPs), launches[k] for k < 5 (gemm_nt, gemm_tn, gemm_generic, attn_fwd, attn_bwd). Synchronises.
extern "C" int leccr_prof_read(double* ms, double* work, int64_t* launches) {
    return leccr_prof_read_n(leccr::PROF_KINDS_V1, ms, work, launches);
}

// The same for the first n kinds of ProfKind (n <= 11: + layernorm fwd / bwd, colsum, embedding stages, AdamW + copies, dropout,
// whose `work` is algorithmic HBM bytes).
extern "C" int leccr_prof_read_n(int n, double* ms, double* work, int64_t* launches) {
    using namespace leccr;
    if (n < 1 || n > PROF_KINDS) { set_error("prof_read_n: n=%d", n); return LECCR_EINVAL; }
    for (int k = 0; k < n; ++k) { ms[k] = 0; work[k] = 0; launches[k] = 0; }
    for (int i = 0; i < g_prof_n; ++i) {
        ProfRec& r = g_prof[i];
        if (r.kind >= n) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
        ms[r.kind] += t;
        work[r.kind] += r.work;
        launches[r.kind] += 1;
    }
    return LECCR_OK;
}

// Launches of the kinds in `kind_mask` (bit k = kind k) as ONE busy time: the union of their [start, stop] intervals, so that
// launches overlapping on two streams (weight-gradient GEMMs beside the dX chain) are not counted twice.  Synchronises.
extern "C" int leccr_prof_read_union(int kind_mask, double* busy_ms, double* work, int64_t* launches) {
    using namespace leccr;
    *busy_ms = 0; *work = 0; *launches = 0;
    std::vector<std::pair<float, float>> iv;
    hipEvent_t base = nullptr;
    for (int i = 0; i < g_prof_n; ++i) {
        ProfRec& r = g_prof[i];
        if (!((kind_mask >> r.kind) & 1)) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        if (!base) base = r.a;
        float t0 = 0.f, t1 = 0.f;
        if (hipEventElapsedTime(&t0, base, r.a) != hipSuccess || hipEventElapsedTime(&t1, base, r.b) != hipSuccess) continue;
        iv.emplace_back(t0, t1);
        *work += r.work;
        *launches += 1;
    }
    std::sort(iv.begin(), iv.end());
    float lo = 0.f, hi = 0.f;
    bool open = false;
    for (auto& p : iv) {
        if (!open) { lo = p.first; hi = p.second; open = true; }
        else if (p.first <= hi) hi = p.second > hi ? p.second : hi;
        else { *busy_ms += hi - lo; lo = p.first; hi = p.second; }
    }
    if (open) *busy_ms += hi - lo;
    return LECCR_OK;
}

// Diagnostic: `blocks` workgroups of `threads` lanes that do nothing for `us` microseconds and exit (every wave watches the constant 100 MHz
// clock, so the grid drains by itself).  Stands in for a collective's resident workgroups when there is no second GPU to talk to: what a
// communication kernel does to the persistent GEMMs beside it is first of all to HOLD compute units (scripts/r5_rccl_cus.sh).
__global__ void occupy_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

extern "C" int leccr_debug_occupy(int blocks, int threads, int64_t us, void* stream) {
    if (blocks < 1 || blocks > 256 || threads < 64 || threads > 1024 || us < 0 || us > 100000) return LECCR_EINVAL;
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (long long)us * 100);
    return hipGetLastError() == hipSuccess ? LECCR_OK : LECCR_EINVAL;
}

extern "C" int leccr_version(void) { return 3; }      // 3: leccr_gemm_params.c3 / c3_form, leccr_split3, leccr_debug_set_stamps declared
extern "C" const char* leccr_last_error(void) { return leccr::g_err; }
