// Library-level entry points of the slam_amd C ABI.
#include "common.h"

namespace slam {
char *error_buffer()
{
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace slam

extern "C" int slam_version(void) { return 1; }

extern "C" const char *slam_last_error(void) { return slam::error_buffer(); }
