// Library-wide pieces of the C ABI (include/gfs3d.h): version + error text.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace gfs3d {
bool g_deterministic = false;
bool g_bn_tail = [] { const char *e = getenv("GFS3D_PW_TAIL"); return !(e && e[0] == '0'); }();     // GFS3D_PW_TAIL=0: field bisection

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct ProfRec { int id; hipEvent_t a, b; };
static std::mutex g_pm;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;          // events are recycled: none is created in steady state
static bool g_prof = false;
constexpr size_t PROF_MAX_RECS = 1u << 16;      // un-read records are bounded: beyond this, launches go un-timed

static bool pool_take(hipEvent_t &e)
{
    if (!g_pool.empty()) { e = g_pool.back(); g_pool.pop_back(); return true; }
    return hipEventCreate(&e) == hipSuccess;
}

int prof_begin(int kernel_id, hipStream_t st)
{
    if (!g_prof) return -1;
    std::lock_guard<std::mutex> lk(g_pm);
    if (g_recs.size() >= PROF_MAX_RECS) return -1;
    ProfRec r;
    r.id = kernel_id;
    if (!pool_take(r.a)) return -1;
    if (!pool_take(r.b)) { g_pool.push_back(r.a); return -1; }
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}
void prof_end(int token, hipStream_t st)
{
    if (token < 0) return;
    std::lock_guard<std::mutex> lk(g_pm);
    if (token < (int)g_recs.size()) (void)hipEventRecord(g_recs[token].b, st);
}
}  // namespace gfs3d

extern "C" int gfs3d_profile_enable(int on)
{
    using namespace gfs3d;
    std::lock_guard<std::mutex> lk(g_pm);
    if (on && !g_prof)                                  // fill the pool up front: no hipEventCreate between the brackets
        while (g_pool.size() < 2048) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) break;
            g_pool.push_back(e);
        }
    g_prof = on != 0;
    return GFS3D_OK;
}

extern "C" int gfs3d_profile_read(int kernel_id, double *total_ms, long long *launches)
{
    using namespace gfs3d;
    GFS3D_REQUIRE(total_ms && launches && kernel_id >= 0 && kernel_id < GFS3D_PROF_KERNELS, "gfs3d_profile_read: bad argument");
    std::lock_guard<std::mutex> lk(g_pm);
    double ms = 0.0;
    long long n = 0;
    std::vector<ProfRec> keep;
    for (auto &r : g_recs) {
        if (r.id != kernel_id) { keep.push_back(r); continue; }
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms += t; ++n; }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.swap(keep);
    *total_ms = ms;
    *launches = n;
    return GFS3D_OK;
}

extern "C" int gfs3d_set_bn_tail(int on)
{
    gfs3d::g_bn_tail = on != 0;
    return GFS3D_OK;
}

extern "C" int gfs3d_version(void) { return 100; }   // 0.1.0
extern "C" const char *gfs3d_last_error(void) { return gfs3d::g_err; }
