// Shared helpers of the slam_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/slam_amd.h"

namespace slam {

// Thread-local message behind slam_last_error().
char *error_buffer();

inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(SLAM_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SLAM_OK;
}

constexpr int WAVE = 64;

// Kernel variants in force (slam_select_kernels; defaults from SLAM_KNN_IMPL / SLAM_SCORE_IMPL read once
// when the library is loaded).
int knn_impl();
int score_impl();

}  // namespace slam
