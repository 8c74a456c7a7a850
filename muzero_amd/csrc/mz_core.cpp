// Error reporting and version of the C ABI.
#include "mz_common.h"

namespace mz {
char *error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace mz

extern "C" {
const char *mz_last_error(void) { return mz::error_buffer(); }
int mz_abi_version(void) { return 3; }
}
