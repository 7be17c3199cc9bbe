// pgv_gate.h -- a counting gate on one futex word (plain host C++, no HIP): at most `width` holders, everybody else
// asleep.  Used by pgv_query_scan to bound the single-query scans a process has in flight (pgv_abi_ivf.hip) and compiled
// on its own by tests/c/gate_stress.cpp.
//
// Round 3's gate woke a sleeper only when the leaver saw the count AT the width (`before >= width`).  Two holders
// leaving back to back open two slots with ONE wake; from then on the count tops out at width - 1, no later leave
// wakes anybody, and every remaining sleeper stays asleep for good once the threads that still run have finished
// their work -- a host-side hang with an idle GPU (BENCH_r03: 900 s inside concurrent_backends).  This one counts
// its sleepers and wakes one per leave whenever there is any, and a sleeper never waits longer than a millisecond
// for a wake-up that cannot come.
#pragma once

#include <linux/futex.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

struct PgvGate {
    int inflight = 0;  // holders; the futex word
    int waiters = 0;   // threads inside (or on their way into) FUTEX_WAIT

    // bounded_nap = false only in the stress test (the counting alone must be enough)
    void enter(int width, bool bounded_nap = true) {
        if (width <= 0) return;
        for (;;) {
            int cur = __atomic_load_n(&inflight, __ATOMIC_SEQ_CST);
            if (cur < width) {
                if (__atomic_compare_exchange_n(&inflight, &cur, cur + 1, true, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return;
                continue;
            }
            // announce first, sleep second: a leaver that misses the announcement has already changed the word, and
            // the kernel compares the word with `cur` before it puts us to sleep
            __atomic_add_fetch(&waiters, 1, __ATOMIC_SEQ_CST);
            const struct timespec nap = {0, 1000000};
            syscall(SYS_futex, &inflight, FUTEX_WAIT_PRIVATE, cur, bounded_nap ? &nap : nullptr, nullptr, 0);
            __atomic_sub_fetch(&waiters, 1, __ATOMIC_SEQ_CST);
        }
    }

    void leave(int width) {
        if (width <= 0) return;
        __atomic_sub_fetch(&inflight, 1, __ATOMIC_SEQ_CST);
        if (__atomic_load_n(&waiters, __ATOMIC_SEQ_CST) > 0) syscall(SYS_futex, &inflight, FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
    }
};
