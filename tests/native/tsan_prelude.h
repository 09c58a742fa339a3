// tsan_prelude.h -- force-included (-include) in the ThreadSanitizer build only.  GCC 11's libtsan has
// no interceptor for pthread_cond_clockwait, which libstdc++ uses for condition_variable::wait_for on
// the steady clock: TSan then believes the mutex is still held across the wait and reports "double
// lock" and races between two holders of the same mutex.  Without this macro libstdc++ falls back to
// pthread_cond_timedwait, which TSan understands.  Test builds only; the library is built without it.
#pragma once
#include <bits/c++config.h>
#undef _GLIBCXX_USE_PTHREAD_COND_CLOCKWAIT
