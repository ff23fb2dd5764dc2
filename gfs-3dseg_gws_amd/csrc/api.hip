// Library-wide pieces of the C ABI (include/gfs3d.h): version + error text.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace gfs3d {
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
static bool g_prof = false;

int prof_begin(int kernel_id, hipStream_t st)
{
    if (!g_prof) return -1;
    std::lock_guard<std::mutex> lk(g_pm);
    ProfRec r;
    r.id = kernel_id;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
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
    std::lock_guard<std::mutex> lk(gfs3d::g_pm);
    gfs3d::g_prof = on != 0;
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
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.swap(keep);
    *total_ms = ms;
    *launches = n;
    return GFS3D_OK;
}

extern "C" int gfs3d_version(void) { return 100; }   // 0.1.0
extern "C" const char *gfs3d_last_error(void) { return gfs3d::g_err; }
