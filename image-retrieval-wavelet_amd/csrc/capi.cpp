// Error string + ABI version of libwvhash.  Everything else lives next to its kernels.
#include "common.hpp"

#include <string.h>

namespace wv {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace wv

extern "C" const char *wv_last_error(void) { return wv::g_err; }
extern "C" int wv_abi_version(void) { return 5; }
