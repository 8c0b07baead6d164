// Library-level entry points of the slam_amd C ABI.
#include <stdlib.h>

#include <atomic>

#include "common.h"

namespace slam {
char *error_buffer()
{
    static thread_local char buf[512] = {0};
    return buf;
}

namespace {
int knn_from_env()
{
    const char *v = getenv("SLAM_KNN_IMPL");
    return (v && v[0] == 'v') ? SLAM_KNN_VALU : SLAM_KNN_FP4;
}
int score_from_env()
{
    const char *v = getenv("SLAM_SCORE_IMPL");
    if (!v)
        return SLAM_SCORE_AUTO;
    if (v[0] == 'f' && v[1] == '6')
        return SLAM_SCORE_F64;
    if (v[0] == 's')
        return SLAM_SCORE_SPLIT;
    if (v[0] == 'w')
        return SLAM_SCORE_WIDE;
    if (v[0] == 'm')
        return SLAM_SCORE_MFMA;
    return SLAM_SCORE_AUTO;
}
// the environment is read here, once, when the library is loaded
std::atomic<int> g_knn{knn_from_env()};
std::atomic<int> g_score{score_from_env()};
}  // namespace

int knn_impl() { return g_knn.load(std::memory_order_relaxed); }
int score_impl() { return g_score.load(std::memory_order_relaxed); }
}  // namespace slam

extern "C" int slam_version(void) { return 2; }

extern "C" const char *slam_last_error(void) { return slam::error_buffer(); }

extern "C" int slam_select_kernels(int knn, int score)
{
    if (knn > SLAM_KNN_VALU || score > SLAM_SCORE_MFMA)
        return slam::fail(SLAM_ERR_BAD_ARG, "slam_select_kernels: unknown variant (%d, %d)", knn, score);
    if (knn >= 0)
        slam::g_knn.store(knn);
    if (score >= 0)
        slam::g_score.store(score);
    return SLAM_OK;
}
