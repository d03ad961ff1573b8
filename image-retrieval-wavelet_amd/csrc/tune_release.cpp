// Release build of libwvhash.so: the kernel-selection switches the tests and the tuning tools use (WV_SWT_PATH,
// WV_HEAD_FRONT, WV_TOPK_V2, ...) do not exist -- no entry point reads the environment (include/wvhash.h: "no global mutable
// state").  libwvhash_diag.so links tune_diag.cpp instead; every other object is shared between the two libraries.
namespace wv {
const char *tune(const char *) { return nullptr; }
}  // namespace wv
