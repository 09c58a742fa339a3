// ctx_pool_test.cpp -- the per-call context pool's waiting logic (readserver_amd/csrc/ctx_pool.h) with a maker that
// fails, on the CPU (built with -fsanitize=thread by tests/test_capi_boundary.py).  Held to:
//   1. a maker that always fails: every caller gets nullptr, nobody waits for a release that cannot come
//      (two callers failing at once included: the second to back out used to leave the first asleep);
//   2. a maker that fails after the first success: callers share the one context, and a release that happens while
//      a caller is inside its failing maker is not missed;
//   3. no failures: at most MAX contexts exist, every caller is served.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../readserver_amd/csrc/ctx_pool.h"

struct ctx { int id; };
using pool_t = rsb::bounded_pool<ctx, 4>;

static int run(int threads, int rounds, int succeed_first, bool slow_maker, int *served, int *refused, int *made_out) {
    pool_t pool;
    std::atomic<int> made{0}, ok{0}, no{0}, in_use{0}, worst{0};
    auto worker = [&] {
        for (int r = 0; r < rounds; ++r) {
            ctx *c = pool.acquire([&]() -> ctx * {
                if (slow_maker) std::this_thread::sleep_for(std::chrono::microseconds(200));
                const int k = made.fetch_add(1);
                return k < succeed_first ? new ctx{k} : nullptr;
            });
            if (!c) { ++no; continue; }
            const int now = ++in_use;
            int w = worst.load();
            while (now > w && !worst.compare_exchange_weak(w, now)) {}
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            --in_use;
            ++ok;
            pool.release(c);
        }
    };
    std::vector<std::thread> ts;
    for (int i = 0; i < threads; ++i) ts.emplace_back(worker);
    for (auto &t : ts) t.join();
    for (ctx *c : pool.free_) delete c;
    *served = ok;
    *refused = no;
    *made_out = made;
    return worst;
}

int main() {
    int ok, no, made;
    // 1. nothing can be made: all refused, and the run ENDS (a hang here is the bug)
    run(8, 50, 0, true, &ok, &no, &made);
    if (ok != 0 || no != 8 * 50) { std::printf("always-failing maker: served %d refused %d\n", ok, no); return 1; }
    // 2. one context, then failures: everybody is served through it or refused -- nobody is lost
    int worst = run(8, 200, 1, true, &ok, &no, &made);
    if (ok + no != 8 * 200 || ok == 0 || worst > 1) { std::printf("one context: served %d refused %d worst %d\n", ok, no, worst); return 1; }
    // 3. no failures: bounded by MAX
    worst = run(16, 200, 1 << 30, false, &ok, &no, &made);
    if (ok != 16 * 200 || no != 0 || worst > 4 || made > 4) { std::printf("no failures: served %d refused %d worst %d made %d\n", ok, no, worst, made); return 1; }
    std::printf("ctx_pool ok\n");
    return 0;
}
