// Host-side runtime of the library: thread-local error string, launch check and the
// optional per-launch hipEvent bracketing that bench.py uses for its live roofline.
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "common.h"

namespace aau {

static thread_local char g_err[512] = "";

// traversal hint (aau_traverse): 0 = off (always forward), 1 = alternate; g_rev = direction of the next launch
static thread_local int g_trav_mode = 0, g_rev = 0;

int next_traversal() {
    if (!g_trav_mode) return 0;
    const int r = g_rev;
    g_rev ^= 1;
    return r;
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return AAU_E_HIP;
    }
    return AAU_OK;
}

struct ProfRec {
    hipEvent_t a, b;
    int family;
    double flops, bytes;
    char tag[48], label[48];
};
static thread_local char g_prof_label[48] = "";   // caller-side name of the next launch (aau_prof_label)
static thread_local int g_prof_cur = -1;   // record of the innermost live ProfScope of this thread
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(int family, double flops, hipStream_t s) : idx(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.a = get_event();
    r.b = get_event();
    r.family = family;
    r.flops = flops;
    r.bytes = 0.0;
    r.tag[0] = 0;
    memcpy(r.label, g_prof_label, sizeof(r.label));
    g_prof_label[0] = 0;
    hipEventRecord(r.a, s);
    g_prof.push_back(r);
    idx = (int)g_prof.size() - 1;
    g_prof_cur = idx;
}

void prof_tag(const char* tag, double bytes) {
    if (g_prof_cur < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_cur >= (int)g_prof.size()) return;
    ProfRec& r = g_prof[g_prof_cur];
    if (tag) snprintf(r.tag, sizeof(r.tag), "%s", tag);
    if (bytes > 0.0) r.bytes = bytes;
}

ProfScope::~ProfScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (idx < (int)g_prof.size()) hipEventRecord(g_prof[idx].b, stream);
    g_prof_cur = -1;
}

}  // namespace aau

extern "C" const char* aau_last_error(void) { return aau::g_err; }
extern "C" int aau_version(void) { return 1; }

extern "C" int aau_traverse(int alternate) {
    aau::g_trav_mode = alternate != 0;
    aau::g_rev = 0;
    return AAU_OK;
}

extern "C" int aau_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(aau::g_prof_mu);
    aau::g_prof_on = on != 0;
    return AAU_OK;
}

extern "C" int aau_prof_label(const char* label) {
    snprintf(aau::g_prof_label, sizeof(aau::g_prof_label), "%s", label ? label : "");
    return AAU_OK;
}

extern "C" int aau_prof_collect_launches(int cap, int* n_out, char* tags, double* ms, double* flops, double* bytes,
                                         int* family) {
    using namespace aau;
    AAU_REQUIRE(n_out && (cap == 0 || (tags && ms && flops && bytes && family)), "aau_prof_collect_launches: null pointer");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& r : g_prof) {
        float t = 0.f;
        const bool ok = hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess;
        if (ok && n < cap) {
            snprintf(tags + (size_t)n * AAU_PROF_TAG_LEN, AAU_PROF_TAG_LEN, "%s|%s", r.label, r.tag);
            ms[n] = t; flops[n] = r.flops; bytes[n] = r.bytes; family[n] = r.family;
            ++n;
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_prof.clear();
    *n_out = n;
    return AAU_OK;
}

extern "C" int aau_prof_collect(double ms[AAU_PROF_FAMILIES], int64_t launches[AAU_PROF_FAMILIES],
                                double flops[AAU_PROF_FAMILIES]) {
    using namespace aau;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < AAU_PROF_FAMILIES; ++i) { ms[i] = 0; launches[i] = 0; flops[i] = 0; }
    for (auto& r : g_prof) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms[r.family] += t;
            launches[r.family] += 1;
            flops[r.family] += r.flops;
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_prof.clear();
    return AAU_OK;
}
