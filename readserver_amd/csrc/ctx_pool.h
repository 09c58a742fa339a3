// ctx_pool.h -- the bounded pool of per-call contexts behind a handle (capi_internal.h: a stream pair and a staging
// buffer per concurrent host caller), as a template over the context type and its maker so that the waiting logic
// can be held to its contract on the CPU with a maker that fails (tests/native/ctx_pool_test.cpp).
//
// Contract: acquire() returns a free context, or makes one while fewer than MAX exist, or waits for a release.
// A maker that fails does not strand anyone: the thread that failed waits for a release only while other contexts
// exist (and sees a release that happened while it was making), every waiter is woken when a maker backs out, and
// with no context left acquire() returns nullptr instead of waiting for a release that cannot come.
#ifndef RSBWT_CTX_POOL_H
#define RSBWT_CTX_POOL_H

#include <condition_variable>
#include <mutex>
#include <vector>

namespace rsb {

template <class Ctx, int MAX>
struct bounded_pool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Ctx *> free_;
    int created = 0;  // contexts that exist or are being made

    // make(): a new context or nullptr; called without the lock held
    template <class Make>
    Ctx *acquire(Make &&make) {
        std::unique_lock<std::mutex> lock(mu);
        for (;;) {
            if (!free_.empty()) {
                Ctx *c = free_.back();
                free_.pop_back();
                return c;
            }
            if (created < MAX) {
                ++created;
                lock.unlock();
                Ctx *c = make();
                if (c) return c;
                lock.lock();
                --created;
                cv.notify_all();  // a waiter may now be the one to make a context -- or has to learn that none is left
                // others exist: wait for one of them instead of asking the maker again at once (a release that
                // happened while the lock was dropped is seen: the predicate looks at free_ first)
                cv.wait(lock, [&] { return !free_.empty() || created == 0; });
                if (free_.empty()) return nullptr;  // not even one context: report it
                continue;
            }
            cv.wait(lock, [&] { return !free_.empty() || created < MAX; });
        }
    }
    void release(Ctx *c) {
        if (!c) return;
        {
            std::lock_guard<std::mutex> lock(mu);
            free_.push_back(c);
        }
        cv.notify_one();
    }
};

}  // namespace rsb
#endif
