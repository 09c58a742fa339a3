// capi_guard.h -- nothing may unwind through an extern "C" body: the caller may be C (or a cgo / JNI /
// ctypes stub) and an exception leaving the library there is std::terminate.  Bodies that allocate with
// the standard containers or start threads run inside guarded(): std::bad_alloc / std::length_error
// become RSBWT_ENOMEM, anything else (std::system_error of a thread that could not be started, ...)
// RSBWT_ESYS, with the message kept for rsbwt_last_error().
#ifndef RSBWT_CAPI_GUARD_H
#define RSBWT_CAPI_GUARD_H

#include <exception>
#include <new>
#include <stdexcept>

#include "../../include/rsbwt.h"

namespace rsb {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

template <class F>
int guarded(const char *what, F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(RSBWT_ENOMEM, "%s: host allocation failed", what);
    } catch (const std::length_error &) {
        return fail(RSBWT_ENOMEM, "%s: host allocation failed (size out of range)", what);
    } catch (const std::exception &e) {
        return fail(RSBWT_ESYS, "%s: %s", what, e.what());
    } catch (...) {
        return fail(RSBWT_ESYS, "%s: unknown exception", what);
    }
}

}  // namespace rsb
#endif
