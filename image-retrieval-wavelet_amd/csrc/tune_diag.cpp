// Diagnostic build (libwvhash_diag.so): kernel-selection switches are read from the environment on every call, so that tests
// can pin each code path and tools/ can A/B variants.  Never loaded by the product path (wvhash/_lib.py: diagnostic()).
#include <cstdlib>
namespace wv {
const char *tune(const char *name) { return getenv(name); }
}  // namespace wv
