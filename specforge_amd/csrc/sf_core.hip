// Library-level entry points of the C-ABI: version, build flavour, last-error string.
#include "sf_api_internal.h"

char* sf_error_buffer() {
    static thread_local char buf[SF_ERROR_BUFFER_BYTES] = {0};
    return buf;
}

extern "C" int sf_abi_version(void) { return SF_ABI_VERSION; }

extern "C" int sf_is_emulated(void) {
#ifdef SF_EMU
    return 1;
#else
    return 0;
#endif
}

extern "C" const char* sf_last_error(void) { return sf_error_buffer(); }
