// gate_stress.cpp -- the admission gate of pgv_query_scan (pgvector_amd/csrc/pgv_gate.h) under the schedule that hung
// BENCH_r03: more threads than slots, every thread a fixed number of passes, then it exits (nobody re-enters at the
// end to refill the count).  `new` runs PgvGate with UNBOUNDED sleeps -- the sleeper count alone must be enough;
// `legacy` runs round 3's rule (wake one only when the leaver saw the count at the width) and is expected to strand
// sleepers.  Prints "ok <rounds>" or "stuck round <r>: inflight <i> sleeping <s>" and exits 0 / 3.
//   g++ -O2 -pthread -I pgvector_amd/csrc tests/c/gate_stress.cpp -o build/tests/gate_stress
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pgv_gate.h"

static int g_width, g_legacy, g_passes;
static PgvGate g_gate;
static int g_sleeping;  // legacy: threads inside FUTEX_WAIT (diagnostics only)

static void legacy_enter() {
    for (;;) {
        int cur = __atomic_load_n(&g_gate.inflight, __ATOMIC_RELAXED);
        if (cur < g_width) {
            if (__atomic_compare_exchange_n(&g_gate.inflight, &cur, cur + 1, true, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED)) return;
            continue;
        }
        __atomic_add_fetch(&g_sleeping, 1, __ATOMIC_RELAXED);
        syscall(SYS_futex, &g_gate.inflight, FUTEX_WAIT_PRIVATE, cur, nullptr, nullptr, 0);
        __atomic_sub_fetch(&g_sleeping, 1, __ATOMIC_RELAXED);
    }
}
static void legacy_leave() {
    const int before = __atomic_fetch_sub(&g_gate.inflight, 1, __ATOMIC_RELEASE);
    if (before >= g_width) syscall(SYS_futex, &g_gate.inflight, FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
}

static void *worker(void *arg) {
    unsigned seed = (unsigned)(size_t)arg * 2654435761u + 1;
    for (int p = 0; p < g_passes; p++) {
        if (g_legacy) legacy_enter(); else g_gate.enter(g_width, false);
        seed = seed * 1664525u + 1013904223u;
        for (volatile unsigned spin = 0; spin < (seed >> 19); spin++) {}   // "the scan": 0..8191 iterations
        if (g_legacy) legacy_leave(); else g_gate.leave(g_width);
        seed = seed * 1664525u + 1013904223u;
        for (volatile unsigned spin = 0; spin < (seed >> 21); spin++) {}   // "the next rank": shorter
    }
    return nullptr;
}

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: gate_stress new|legacy threads width passes rounds\n"); return 2; }
    g_legacy = strcmp(argv[1], "legacy") == 0;
    const int threads = atoi(argv[2]);
    g_width = atoi(argv[3]);
    g_passes = atoi(argv[4]);
    const int rounds = atoi(argv[5]);
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int r = 0; r < rounds; r++) {
        for (int i = 0; i < threads; i++) pthread_create(&th[i], nullptr, worker, (void *)(size_t)(r * threads + i));
        struct timespec until;
        clock_gettime(CLOCK_REALTIME, &until);
        until.tv_sec += 5;
        for (int i = 0; i < threads; i++)
            if (pthread_timedjoin_np(th[i], nullptr, &until) != 0) {
                printf("stuck round %d: inflight %d sleeping %d\n", r, g_gate.inflight, g_legacy ? g_sleeping : g_gate.waiters);
                fflush(stdout);
                _exit(3);   // the sleepers cannot be woken: leave with them
            }
    }
    printf("ok %d\n", rounds);
    return 0;
}
