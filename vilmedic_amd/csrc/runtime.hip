// runtime.hip -- error state, version, and the per-family HIP-event profiler of libvmhip.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void vm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int vm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vm_set_error("%s: %s", what, hipGetErrorString(e));
        return VM_EHIP;
    }
    return VM_OK;
}

extern "C" const char* vm_last_error(void) { return g_err; }

// ---------------------------------------------------------------- environment switches (cached)
static VmEnv g_env;
static bool g_env_loaded = false;
static void load_env() {
    const char* v;
    g_env.gemm_variant = (v = getenv("VM_GEMM_VARIANT")) ? atoi(v) : -1;
    g_env.gemm_debug = (v = getenv("VM_GEMM_DEBUG")) ? atoi(v) : 0;
    g_env.gemm_groupw = (v = getenv("VM_GEMM_GROUPW")) ? atoi(v) : 0;
    g_env.gemm_p8_mf = (v = getenv("VM_GEMM_P8_MF")) ? atoi(v) : 0;
    g_env.wgrad_p8 = (v = getenv("VM_WGRAD_P8")) ? atoi(v) : 2;
    g_env.wgrad_p8_min = (v = getenv("VM_WGRAD_P8_MIN")) ? atoi(v) : 64;
    g_env.gemm_generic = getenv("VM_GEMM_GENERIC") != nullptr;
    g_env.gemm_no_skinny = getenv("VM_GEMM_NO_SKINNY") != nullptr;
    g_env.attn_tile = getenv("VM_ATTN_TILE") != nullptr;
    g_env.attn_stream = getenv("VM_ATTN_STREAM") != nullptr;
    g_env_loaded = true;
}
const VmEnv& vm_env() {
    if (!g_env_loaded) load_env();
    return g_env;
}
extern "C" void vm_reload_env(void) { load_env(); }
extern "C" int vm_version(void) { return 100; }
// the digest of the sources this library was built from (vilmedic_amd/build.py writes build_digest.inc before compiling); build.py finds the
// marker in the file's bytes to decide whether the library is current
#if __has_include("build_digest.inc")
#include "build_digest.inc"
#endif
#ifndef VM_BUILD_DIGEST
#define VM_BUILD_DIGEST "unknown"
#endif
extern "C" const char* vm_build_digest(void) { return "VMDIGEST:" VM_BUILD_DIGEST + 9; }
extern "C" int vm_sizeof_gemm_epilogue(void) { return (int)sizeof(vm_gemm_epilogue); }

// ---------------------------------------------------------------- profiler
struct ProfSlot { hipEvent_t a, b; int fam; double work; char tag[96]; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfSlot*> g_prof_used, g_prof_free;

VmProfScope::VmProfScope(int family, double work, hipStream_t stream, const char* tag_fmt, ...) : fam(family), s(stream), slot(nullptr) {
    if (!g_prof_on) return;
    char tag[96] = "";
    if (tag_fmt) { va_list ap; va_start(ap, tag_fmt); vsnprintf(tag, sizeof(tag), tag_fmt, ap); va_end(ap); }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfSlot* p;
    if (!g_prof_free.empty()) { p = g_prof_free.back(); g_prof_free.pop_back(); }
    else { p = new ProfSlot(); hipEventCreate(&p->a); hipEventCreate(&p->b); }
    p->fam = family; p->work = work; memcpy(p->tag, tag, sizeof(tag));
    hipEventRecord(p->a, stream);
    slot = p;
}
VmProfScope::~VmProfScope() {
    if (!slot) return;
    ProfSlot* p = (ProfSlot*)slot;
    hipEventRecord(p->b, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_used.push_back(p);
}

extern "C" int vm_prof_enable(int on) { g_prof_on = on != 0; return VM_OK; }
extern "C" int vm_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto* p : g_prof_used) g_prof_free.push_back(p);
    g_prof_used.clear();
    return VM_OK;
}
extern "C" int vm_prof_read(int family, double* ms_total, double* work_total, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0, w = 0; int64_t n = 0;
    for (auto* p : g_prof_used) {
        if (p->fam != family) continue;
        hipEventSynchronize(p->b);
        float t = 0;
        hipEventElapsedTime(&t, p->a, p->b);
        ms += t; w += p->work; ++n;
    }
    if (ms_total) *ms_total = ms;
    if (work_total) *work_total = w;
    if (launches) *launches = n;
    return VM_OK;
}

// per-tag breakdown of the recorded launches: "family tag launches total_ms total_work" lines (diagnostics for bench.py)
extern "C" int vm_prof_dump(const char* path) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    struct Acc { double ms = 0, work = 0; long n = 0; };
    std::map<std::string, Acc> acc;
    for (auto* p : g_prof_used) {
        hipEventSynchronize(p->b);
        float t = 0;
        hipEventElapsedTime(&t, p->a, p->b);
        char key[128];
        snprintf(key, sizeof(key), "%d %s", p->fam, p->tag[0] ? p->tag : "-");
        Acc& a = acc[key]; a.ms += t; a.work += p->work; a.n += 1;
    }
    FILE* f = fopen(path, "w");
    if (!f) { vm_set_error("vm_prof_dump: cannot open %s", path); return VM_EINVAL; }
    for (auto& kv : acc) fprintf(f, "%s %ld %.4f %.6g\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.work);
    fclose(f);
    return VM_OK;
}
