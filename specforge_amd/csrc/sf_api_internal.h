// Internal glue of the C-ABI: status codes, last-error string, launch checking.
#pragma once
#include <stdio.h>
#include <string.h>

#include "sf_platform.h"
#include "../../include/specforge_amd.h"

// thread-local last error message (returned by sf_last_error)
char* sf_error_buffer();
#define SF_ERROR_BUFFER_BYTES 512

#define SF_CHECK_ARG(cond, msg)                                                              \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            snprintf(sf_error_buffer(), SF_ERROR_BUFFER_BYTES, "%s (failed check: %s)", msg, #cond); \
            return SF_ERR_INVALID;                                                           \
        }                                                                                    \
    } while (0)

static inline int sf_check_launch(const char* what) {
    const char* e = sf_launch_error();
    if (e) {
        snprintf(sf_error_buffer(), SF_ERROR_BUFFER_BYTES, "%s: launch failed: %s", what, e);
        return SF_ERR_LAUNCH;
    }
    return SF_OK;
}

// Tuning knobs.  The PRODUCT library reads no environment variable: every knob is the compile-time default (the name
// is dropped by the preprocessor, so `strings libsfhip.so` shows none).  Only the tools build (-DSF_ABLATE,
// `python specforge_amd/build.py ablate` -> tools/experiments/libsfhip_ablate.so) turns them into getenv lookups for
// A/B timing of measured-and-rejected variants.
#ifdef SF_ABLATE
#include <stdlib.h>
static inline int sf_knob_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#define sf_knob(name, dflt) sf_knob_env(name, dflt)
#else
#define sf_knob(name, dflt) (dflt)
#endif
